/*
 * spec_study.c -- FEASIBILITY STUDY (analysis tool, test infrastructure, never shipped).
 *
 * Question: can one row's serial chain be cut into K segments that run concurrently, each segment warm-started L pixels
 * early from a COLD state (zero carried error, original left pixel, histogram = committed histogram without this row's
 * earlier bumps), and still be provably bit-exact after a cheap validation?  Measured here on the CPU, per segment:
 *   sync    : after the warm-up the speculative state (left bytes, rem, thr x2, per channel) equals the true state
 *   robust  : every decision inside the segment is provably unaffected by the bumps D of the earlier pixels of the row
 *             (margin test: Hwin - H2 > max_{v in band, v != win} D[v] - D[win], or no bumped bin in the band at all)
 *   changed : decisions that REALLY differ when D is added (ground truth, for calibration of the margin test)
 * usage: spec_study W H mode strength bleed K L
 */
#include "pngloss_port.c"

extern void pngloss_synth_rgba(unsigned char *rgba, uint32_t width, uint32_t height, int mode, uint64_t frame);

typedef struct { int left[4], rem[4], thr_prev[4], thr_cur[4]; } chain_state;

/* one pixel of the chain; returns decisions through out pointers; H is read (and bumped if bump) */
static void pixel_step(const engine *e, uint32_t y, int f, unsigned s, long bleed, uint32_t x, chain_state *st, uint32_t *Hs,
                       int bump, const uint32_t *D, unsigned char *outb, int *nonrobust, int *changed)
{
    const uint32_t bpp = e->bpp;
    const size_t stride = (size_t)e->W * bpp;
    const unsigned char *orig = e->pix + (size_t)y * stride;
    const unsigned char *nabove = y ? orig - stride : NULL;
    const uint32_t *O = e->orig_hist[f];
    const int q = (int)s + 1;
    const bool has_alpha = (bpp % 2) == 0;
    int d16[4] = { 0, 0, 0, 0 };
    const bool transparent = has_alpha && orig[(size_t)x * bpp + bpp - 1] == 0;
    for (uint32_t c = 0; c < bpp; c++) {
        const size_t o = (size_t)x * bpp + c;
        const int pl = plane_of(bpp, c);
        const int ov = orig[o];
        const int above = nabove ? nabove[o] : 0;
        const int diag = (nabove && x) ? nabove[o - bpp] : 0;
        const int left = x ? st->left[c] : 0;
        const int pred = predict(f, above, diag, left);
        int back, sym;
        if (transparent && c == bpp - 1) { back = 0; sym = (0 - pred) & 255; d16[pl] = 0; }
        else {
            const int err = sext16(e->E0[(size_t)x * 4 + pl] + st->rem[pl] + st->thr_prev[pl]);
            const int osym = sext8(ov - pred);
            const int predc = ov - osym;
            const int filt = osym + err;
            int vmin, vmax;
            if (filt < 0) { vmax = -((-filt) - ((-filt) % q)); vmin = vmax - (int)s; }
            else { vmin = filt - (filt % q); vmax = vmin + (int)s; }
            const int lo = -predc, hi = 255 - predc;
            vmin = med3(vmin, lo, hi); vmax = med3(vmax, lo, hi);
            int best = vmin; uint32_t bh = Hs[vmin & 255], bo = O[vmin & 255]; int bflag = (vmin == osym);
            for (int v = vmin + 1; v <= vmax; v++) {
                uint32_t h = Hs[v & 255], oo = O[v & 255]; int fl = (v == osym);
                if (better(h, oo, fl, bh, bo, bflag)) { best = v; bh = h; bo = oo; bflag = fl; }
            }
            if (D) {
                /* margin test + ground truth under H + D */
                uint32_t h2 = 0, X = 0; int any = 0;
                for (int v = vmin; v <= vmax; v++) if (v != best) { any = 1; if (Hs[v & 255] > h2) h2 = Hs[v & 255]; if (D[v & 255] > X) X = D[v & 255]; }
                if (any) {
                    const long M = (long)bh - (long)h2;
                    const long need = (long)X - (long)D[best & 255];
                    if (!(X == 0 || M > need)) (*nonrobust)++;
                    int b2 = vmin; uint32_t bh2 = Hs[vmin & 255] + D[vmin & 255], bo2 = O[vmin & 255]; int bf2 = (vmin == osym);
                    for (int v = vmin + 1; v <= vmax; v++) {
                        uint32_t h = Hs[v & 255] + D[v & 255], oo = O[v & 255]; int fl = (v == osym);
                        if (better(h, oo, fl, bh2, bo2, bf2)) { b2 = v; bh2 = h; bo2 = oo; bf2 = fl; }
                    }
                    if (b2 != best) (*changed)++;
                }
            }
            back = best + predc; sym = best & 255; d16[pl] = sext16(filt - best);
        }
        outb[c] = (unsigned char)back;
        if (bump) Hs[sym]++;
    }
    for (uint32_t c = 0; c < bpp; c++) st->left[c] = outb[c];
    for (int pl = 0; pl < 4; pl++) {
        int parts[5];
        port_sierra_split(d16[pl], bleed, parts);
        st->thr_prev[pl] = st->thr_cur[pl]; st->thr_cur[pl] = parts[1]; st->rem[pl] = parts[4];
    }
}

int main(int argc, char **argv)
{
    if (argc < 8) { fprintf(stderr, "usage: %s W H mode strength bleed K L\n", argv[0]); return 2; }
    const uint32_t W = atoi(argv[1]), Hh = atoi(argv[2]); const int mode = atoi(argv[3]); const unsigned s = atoi(argv[4]);
    const long bleed = atol(argv[5]); const int K = atoi(argv[6]); const int L = atoi(argv[7]);
    unsigned char *rgba = malloc((size_t)W * Hh * 4);
    pngloss_synth_rgba(rgba, W, Hh, mode, 0);
    /* class detection like port_optimize_with_rows (simplified: use the mode) */
    uint32_t bpp = mode == 2 ? 3 : (mode == 3 ? 2 : (mode == 4 ? 1 : 4));
    unsigned char *pix = malloc((size_t)W * Hh * bpp);
    for (size_t i = 0; i < (size_t)W * Hh; i++) {
        const unsigned char *p = rgba + i * 4; unsigned char *d = pix + i * bpp;
        if (bpp == 1) d[0] = p[1]; else if (bpp == 2) { d[0] = p[1]; d[1] = p[3]; } else if (bpp == 3) { d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; } else memcpy(d, p, 4);
    }
    const size_t stride = (size_t)W * bpp;
    engine e; memset(&e, 0, sizeof e);
    e.W = W; e.H = Hh; e.bpp = bpp; e.pix = pix;
    e.old_above = calloc(stride, 1); e.E0 = calloc((size_t)W * 4, 2); e.E1 = calloc((size_t)W * 4, 2);
    candidate cand[F_COUNT];
    for (int f = 0; f < F_COUNT; f++) { cand[f].bytes = calloc(stride, 1); cand[f].diff16 = calloc((size_t)W * 4, 2); }
    port_orig_histograms(pix, W, Hh, bpp, e.orig_hist);
    chain_state *truth = malloc(sizeof(chain_state) * (W + 1));
    uint32_t *Dk = malloc(sizeof(uint32_t) * 256);
    long seg_total = 0, seg_sync = 0, seg_ok = 0, dec_nonrobust = 0, dec_changed = 0, dec_total = 0;
    const int bands = 8; long band_tot[8] = {0}, band_ok[8] = {0}, band_sync[8] = {0};
    long f_tot[5] = {0}, f_sync[5] = {0};
    unsigned char outb[4];
    for (uint32_t y = 0; y < Hh; y++) {
        const unsigned char *nabove = y ? pix + (size_t)(y - 1) * stride : NULL;
        int winner = -1; uint64_t best_cost = UINT64_MAX;
        for (int f = 0; f < F_COUNT; f++) {
            /* true chain with per-pixel state log */
            uint32_t Ht[256]; memcpy(Ht, e.hist, sizeof Ht);
            chain_state st; memset(&st, 0, sizeof st);
            truth[0] = st;
            for (uint32_t x = 0; x < W; x++) {
                pixel_step(&e, y, f, s, bleed, x, &st, Ht, 1, NULL, outb, NULL, NULL);
                memcpy(cand[f].bytes + (size_t)x * bpp, outb, bpp);
                truth[x + 1] = st;
            }
            /* segments */
            memset(Dk, 0, 1024);
            for (int k = 1; k < K; k++) {
                const uint32_t xk = (uint32_t)((uint64_t)W * k / K), xe = (uint32_t)((uint64_t)W * (k + 1) / K);
                const uint32_t xs = xk > (uint32_t)L ? xk - L : 0;
                /* D = bumps of true pixels [0, xk) */
                {   uint32_t Hp[256]; memcpy(Hp, e.hist, sizeof Hp);
                    /* recompute prefix by replaying symbols: cheaper: accumulate from cand bytes */
                    memset(Dk, 0, 1024);
                    for (uint32_t x = 0; x < xk; x++)
                        for (uint32_t c = 0; c < bpp; c++) {
                            const size_t o = (size_t)x * bpp + c;
                            int left = x ? cand[f].bytes[o - bpp] : 0, above = nabove ? nabove[o] : 0, diag = (nabove && x) ? nabove[o - bpp] : 0;
                            Dk[(cand[f].bytes[o] - predict(f, above, diag, left)) & 255]++;
                        }
                }
                /* cold warm-up: zero error, original left pixel, committed histogram, no bumps */
                chain_state sp; memset(&sp, 0, sizeof sp);
                if (xs) for (uint32_t c = 0; c < bpp; c++) sp.left[c] = (pix + (size_t)y * stride)[(size_t)(xs - 1) * bpp + c];
                uint32_t Hs[256]; memcpy(Hs, e.hist, sizeof Hs);
                for (uint32_t x = xs; x < xk; x++) pixel_step(&e, y, f, s, bleed, x, &sp, Hs, 0, NULL, outb, NULL, NULL);
                const int sync = memcmp(&sp, &truth[xk], sizeof sp) == 0;
                /* segment proper from the TRUE state (what a synced run would do), histogram = committed + own bumps */
                chain_state sq = truth[xk];
                int nonrobust = 0, changed = 0;
                for (uint32_t x = xk; x < xe; x++) pixel_step(&e, y, f, s, bleed, x, &sq, Hs, 1, Dk, outb, &nonrobust, &changed);
                seg_total++; seg_sync += sync; seg_ok += (sync && nonrobust == 0);
                dec_nonrobust += nonrobust; dec_changed += changed; dec_total += (long)(xe - xk) * bpp;
                const int b = (int)((uint64_t)y * bands / Hh);
                band_tot[b]++; band_ok[b] += (sync && nonrobust == 0); band_sync[b] += sync;
                f_tot[f]++; f_sync[f] += sync;
            }
            /* finish the candidate like the port does */
            memcpy(cand[f].hist, Ht, sizeof Ht);
            {   /* diff16 for the commit: rerun plain chain (cheap enough) */
                run_chain(&e, y, f, s, bleed, &cand[f]);
            }
            const bool adaptive = (y == 0);
            if (adaptive && port_adaptive_filter(nabove, cand[f].bytes, W, bpp) != f) cand[f].cost = UINT64_MAX;
            else cand[f].cost = derivative_error(&e, y, &cand[f]) / 128 + entropy_cost(&e, y, f, &cand[f]);
            if (cand[f].cost < best_cost) { best_cost = cand[f].cost; winner = f; }
        }
        if (winner < 0) { fprintf(stderr, "row %u needs the strength retry; study skips it\n", y); winner = 0; }
        memcpy(e.old_above, pix + (size_t)y * stride, stride);
        memcpy(pix + (size_t)y * stride, cand[winner].bytes, stride);
        memcpy(e.hist, cand[winner].hist, sizeof e.hist);
        commit_error_rows(&e, &cand[winner], bleed);
    }
    printf("W=%u H=%u mode=%d s=%u b=%ld K=%d L=%d\n", W, Hh, mode, s, bleed, K, L);
    printf("segments %ld: state-sync %.2f%%, fully ok (sync & all decisions robust) %.2f%%\n", seg_total, 100.0 * seg_sync / seg_total, 100.0 * seg_ok / seg_total);
    printf("decisions %ld: flagged non-robust %.4f%%, really changed %.4f%%\n", dec_total, 100.0 * dec_nonrobust / dec_total, 100.0 * dec_changed / dec_total);
    for (int b = 0; b < bands; b++) printf("  rows %4d%%..: sync %.1f%%  ok %.1f%%\n", b * 100 / bands, 100.0 * band_sync[b] / band_tot[b], 100.0 * band_ok[b] / band_tot[b]);
    for (int f = 0; f < 5; f++) printf("  filter %d: sync %.1f%%\n", f, 100.0 * f_sync[f] / f_tot[f]);
    return 0;
}
