/*
 * frozen_study.c -- FEASIBILITY STUDY (analysis tool, test infrastructure, never shipped).
 *
 * Question behind the round-3 "segment-parallel" row engine: if every decision of a row is taken against the histogram
 * FROZEN at the start of the row (so that a channel's chain is a finite-state machine over (left byte, diff, thr) with
 * static per-pixel transition tables, which can be cut into segments and enumerated in parallel), how often does a decision
 * differ from the reference's, which reads the RUNNING histogram?  Measured along the true trajectory:
 *   mism     decisions whose frozen arg-max differs from the running one (after each one the histogram is re-frozen there,
 *            so mism + 1 = number of "epochs" the row needs)
 *   cons     rows a conservative validation with block snapshots (counts known only per block of PB pixels) could not
 *            prove although they hold no mismatch (false rejects)
 *   states   the chain state (diff of the previous pixel, rem+thr carried in, thr of the pixel before) -- how large the
 *            enumerated state set must be
 * usage: frozen_study W H mode strength bleed [PB]
 */
#include "pngloss_port.c"

extern void pngloss_synth_rgba(unsigned char *rgba, uint32_t width, uint32_t height, int mode, uint64_t frame);

#define MAXD 64
typedef struct {
    unsigned long long rows, rows_clean, mism, decisions, cons_reject_clean, cons_reject_rows;
    unsigned long long mism_by_rowband[8];   /* rows 0-15, 16-63, 64-255, 256-1023, 1024-.. */
    unsigned long long rows_by_rowband[8];
    unsigned long long diffhist[2 * MAXD + 1], rhist[33], bigdiff;
    unsigned long long first_mism_pos_sum;
    unsigned long long clamp_dec, tie_dec;
} fstats;
static fstats FS[F_COUNT];
static int PB = 16;

static int rowband(uint32_t y) { return y < 16 ? 0 : y < 64 ? 1 : y < 256 ? 2 : y < 1024 ? 3 : 4; }

static int argmax_band(const uint32_t *Hs, const uint32_t *O, int vmin, int vmax, int osym, int *tie)
{
    int best = vmin; uint32_t bh = Hs[vmin & 255], bo = O[vmin & 255]; int bflag = (vmin == osym);
    int t = 0;
    for (int v = vmin + 1; v <= vmax; v++) {
        uint32_t h = Hs[v & 255], oo = O[v & 255]; int fl = (v == osym);
        if (h == bh && oo == bo) t = 1;
        if (better(h, oo, fl, bh, bo, bflag)) { best = v; bh = h; bo = oo; bflag = fl; }
    }
    if (tie) *tie = t;
    return best;
}

static void study_chain(const engine *e, uint32_t y, int f, unsigned s, long bleed, candidate *cd)
{
    const uint32_t W = e->W, bpp = e->bpp;
    const size_t stride = (size_t)W * bpp;
    const unsigned char *orig = e->pix + (size_t)y * stride;
    const unsigned char *nabove = y ? orig - stride : NULL;
    const uint32_t *O = e->orig_hist[f];
    uint32_t *Hs = cd->hist;
    uint32_t Hf[256];
    const int q = (int)s + 1;
    const bool has_alpha = (bpp % 2) == 0;
    fstats *st = &FS[f];
    memcpy(Hs, e->hist, sizeof(e->hist));
    memcpy(Hf, e->hist, sizeof(e->hist));
    int rem[4] = { 0, 0, 0, 0 }, thr_prev[4] = { 0, 0, 0, 0 }, thr_cur[4] = { 0, 0, 0, 0 };
    unsigned long long mism_row = 0;
    /* for the conservative block validation: decisions of the row */
    static int *dec_vmin, *dec_vmax, *dec_best; static uint32_t dec_cap;
    if (dec_cap < W * 4) { dec_cap = W * 4; dec_vmin = realloc(dec_vmin, dec_cap * sizeof(int)); dec_vmax = realloc(dec_vmax, dec_cap * sizeof(int)); dec_best = realloc(dec_best, dec_cap * sizeof(int)); }
    long first_mism = -1;

    for (uint32_t x = 0; x < W; x++) {
        int d16[4] = { 0, 0, 0, 0 };
        const bool transparent = has_alpha && orig[(size_t)x * bpp + bpp - 1] == 0;
        for (uint32_t c = 0; c < bpp; c++) {
            const size_t o = (size_t)x * bpp + c;
            const int pl = plane_of(bpp, c);
            const int ov = orig[o];
            const int above = nabove ? nabove[o] : 0;
            const int diag = (nabove && x) ? nabove[o - bpp] : 0;
            const int left = x ? cd->bytes[o - bpp] : 0;
            const int pred = predict(f, above, diag, left);
            int back, sym;
            dec_vmin[x * 4 + c] = 1; dec_vmax[x * 4 + c] = 0; dec_best[x * 4 + c] = 0;
            if (transparent && c == bpp - 1) {
                back = 0; sym = (0 - pred) & 255; d16[pl] = 0;
                dec_best[x * 4 + c] = sym; /* vmin > vmax: no competitors */
            } else {
                const int carried = rem[pl] + thr_prev[pl];
                if (carried >= -16 && carried <= 16) st->rhist[carried + 16]++;
                const int err = sext16(e->E0[(size_t)x * 4 + pl] + carried);
                const int osym = sext8(ov - pred);
                const int predc = ov - osym;
                const int filt = osym + err;
                int vmin, vmax;
                if (filt < 0) { vmax = -((-filt) - ((-filt) % q)); vmin = vmax - (int)s; }
                else          { vmin = filt - (filt % q);          vmax = vmin + (int)s; }
                const int lo = -predc, hi = 255 - predc;
                const int uvmin = vmin, uvmax = vmax;
                vmin = med3(vmin, lo, hi);
                vmax = med3(vmax, lo, hi);
                if (vmin != uvmin || vmax != uvmax) st->clamp_dec++;
                int tie = 0;
                const int best = argmax_band(Hs, O, vmin, vmax, osym, NULL);
                const int bestf = argmax_band(Hf, O, vmin, vmax, osym, &tie);
                if (tie) st->tie_dec++;
                st->decisions++;
                if (best != bestf) {
                    if (getenv("FS_VERBOSE") && y > 64) fprintf(stderr, "mism y=%u f=%d x=%u c=%u orig=%d filt=%d osym=%d band=[%d,%d] best=%d (H %u->%u O %u) frozen=%d (H %u->%u O %u)\n", y, f, x, c, ov, filt, osym, vmin, vmax, best, Hf[best&255], Hs[best&255], O[best&255], bestf, Hf[bestf&255], Hs[bestf&255], O[bestf&255]);
                    st->mism++; mism_row++;
                    if (first_mism < 0) first_mism = x;
                    memcpy(Hf, Hs, sizeof Hf);      /* re-freeze here */
                }
                dec_vmin[x * 4 + c] = vmin; dec_vmax[x * 4 + c] = vmax; dec_best[x * 4 + c] = best;
                back = best + predc;
                sym = best & 255;
                d16[pl] = sext16(filt - best);
                const int dd = d16[pl];
                if (dd >= -MAXD && dd <= MAXD) st->diffhist[dd + MAXD]++; else st->bigdiff++;
            }
            cd->bytes[o] = (unsigned char)back;
            Hs[sym]++;
        }
        for (int pl = 0; pl < 4; pl++) {
            cd->diff16[(size_t)x * 4 + pl] = (int16_t)d16[pl];
            int parts[5];
            port_sierra_split(d16[pl], bleed, parts);
            thr_prev[pl] = thr_cur[pl];
            thr_cur[pl] = parts[1];
            rem[pl] = parts[4];
        }
    }
    st->rows++;
    st->rows_by_rowband[rowband(y)]++;
    st->mism_by_rowband[rowband(y)] += mism_row;
    if (!mism_row) st->rows_clean++;
    else st->first_mism_pos_sum += (unsigned long long)first_mism;

    /* conservative validation against the histogram frozen at the row start, counts known per block of PB pixels:
     * a decision in block k is PROVEN if for every other v' of its clamped band
     *     H0[v'] + cnt_through_block_k[v'] <  H0[best] + cnt_before_block_k[best]
     * (strictly below on the frequency alone).  Ties at the top under the frozen histogram are "proven" only if no bin of the
     * band is bumped in the row up to and including this block (then frozen == running for that band). */
    {
        uint32_t cb[256], ce[256];
        memset(cb, 0, sizeof cb);
        int reject = 0;
        for (uint32_t x0 = 0; x0 < W && !reject; x0 += (uint32_t)PB) {
            const uint32_t x1 = x0 + (uint32_t)PB < W ? x0 + (uint32_t)PB : W;
            memcpy(ce, cb, sizeof ce);
            for (uint32_t x = x0; x < x1; x++) for (uint32_t c = 0; c < bpp; c++) ce[dec_best[x * 4 + c] & 255]++;
            for (uint32_t x = x0; x < x1 && !reject; x++)
                for (uint32_t c = 0; c < bpp; c++) {
                    const int vmin = dec_vmin[x * 4 + c], vmax = dec_vmax[x * 4 + c], best = dec_best[x * 4 + c];
                    if (vmin > vmax) continue;
                    const uint32_t hb = e->hist[best & 255] + cb[best & 255];
                    int quiet = 1;
                    for (int v = vmin; v <= vmax; v++) if (ce[v & 255]) quiet = 0;
                    if (quiet) continue;    /* nothing in the band was bumped: frozen == running */
                    for (int v = vmin; v <= vmax; v++) {
                        if (v == best) continue;
                        if (!(e->hist[v & 255] + ce[v & 255] < hb)) { reject = 1; break; }
                    }
                    if (reject) break;
                }
            memcpy(cb, ce, sizeof cb);
        }
        if (reject) { st->cons_reject_rows++; if (!mism_row) st->cons_reject_clean++; }
    }
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s W H mode strength bleed [PB]\n", argv[0]); return 2; }
    const uint32_t W = (uint32_t)atoi(argv[1]), H = (uint32_t)atoi(argv[2]);
    const int mode = atoi(argv[3]);
    const unsigned strength = (unsigned)atoi(argv[4]);
    const long bleed = atol(argv[5]);
    if (argc > 6) PB = atoi(argv[6]);
    unsigned char *rgba = malloc((size_t)W * H * 4);
    if (getenv("FS_FILE")) { FILE *fp = fopen(getenv("FS_FILE"), "rb"); if (!fp || fread(rgba, 4, (size_t)W * H, fp) != (size_t)W * H) { fprintf(stderr, "cannot read %s\n", getenv("FS_FILE")); return 1; } fclose(fp); }
    else pngloss_synth_rgba(rgba, W, H, mode, 0);
    /* classify + pack like port_optimize_with_rows */
    int gray = 1, opaque = 1;
    for (size_t i = 0; i < (size_t)W * H; i++) { const unsigned char *p = rgba + i * 4; gray &= (p[0] == p[1]) & (p[1] == p[2]); opaque &= p[3] == 255; }
    const uint32_t bpp = gray ? (opaque ? 1 : 2) : (opaque ? 3 : 4);
    unsigned char *pix = malloc((size_t)W * H * bpp);
    for (size_t i = 0; i < (size_t)W * H; i++) {
        const unsigned char *s = rgba + i * 4; unsigned char *d = pix + i * bpp;
        switch (bpp) { case 1: d[0] = s[1]; break; case 2: d[0] = s[1]; d[1] = s[3]; break; case 3: d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; break; default: memcpy(d, s, 4); }
    }
    const size_t stride = (size_t)W * bpp;
    engine e; memset(&e, 0, sizeof e);
    e.W = W; e.H = H; e.bpp = bpp; e.pix = pix;
    e.old_above = calloc(stride, 1);
    e.E0 = calloc((size_t)W * 4, sizeof(int16_t)); e.E1 = calloc((size_t)W * 4, sizeof(int16_t));
    candidate cand[F_COUNT];
    for (int f = 0; f < F_COUNT; f++) { cand[f].bytes = calloc(stride, 1); cand[f].diff16 = calloc((size_t)W * 4, sizeof(int16_t)); }
    port_orig_histograms(pix, W, H, bpp, e.orig_hist);
    unsigned long long any_rows_dirty = 0;
    for (uint32_t y = 0; y < H; y++) {
        const bool adaptive = y == 0;
        const unsigned char *nabove = y ? pix + (size_t)(y - 1) * stride : NULL;
        unsigned s = strength; int winner = -1; uint64_t best_cost = UINT64_MAX;
        for (;;) {
            unsigned long long before[F_COUNT]; for (int f = 0; f < F_COUNT; f++) before[f] = FS[f].mism;
            for (int f = 0; f < F_COUNT; f++) {
                study_chain(&e, y, f, s, bleed, &cand[f]);
                if (adaptive && port_adaptive_filter(nabove, cand[f].bytes, W, bpp) != f) cand[f].cost = UINT64_MAX;
                else cand[f].cost = derivative_error(&e, y, &cand[f]) / 128 + entropy_cost(&e, y, f, &cand[f]);
                if (cand[f].cost < best_cost) { best_cost = cand[f].cost; winner = f; }
            }
            int dirty = 0; for (int f = 0; f < F_COUNT; f++) dirty |= FS[f].mism != before[f];
            any_rows_dirty += dirty;
            if (winner >= 0) break;
            if (s == 0) abort();
            s--;
        }
        memcpy(e.old_above, pix + (size_t)y * stride, stride);
        memcpy(pix + (size_t)y * stride, cand[winner].bytes, stride);
        memcpy(e.hist, cand[winner].hist, sizeof e.hist);
        commit_error_rows(&e, &cand[winner], bleed);
    }
    static const char *fn[F_COUNT] = { "none", "sub", "up", "avg", "paeth" };
    printf("W=%u H=%u mode=%d bpp=%u s=%u b=%ld PB=%d\n", W, H, mode, bpp, strength, bleed, PB);
    printf("rows with a mismatch in ANY candidate: %llu of %u\n", any_rows_dirty, H);
    for (int f = 0; f < F_COUNT; f++) {
        fstats *st = &FS[f];
        printf("%-5s rows %llu clean %llu (%.2f%%)  mism %llu (%.4f per row, %.3g of decisions)  cons-reject rows %llu (of clean: %llu)  clamp %.2f%% tie %.2f%%\n", fn[f], st->rows, st->rows_clean,
               100.0 * st->rows_clean / st->rows, st->mism, (double)st->mism / st->rows, (double)st->mism / st->decisions, st->cons_reject_rows, st->cons_reject_clean,
               100.0 * st->clamp_dec / st->decisions, 100.0 * st->tie_dec / st->decisions);
        printf("      mism/row by row band [0-15,16-63,64-255,256-1023,1024+]:");
        for (int b = 0; b < 5; b++) printf(" %.3f", st->rows_by_rowband[b] ? (double)st->mism_by_rowband[b] / st->rows_by_rowband[b] : 0.0);
        printf("\n      diff range:");
        int dmin = MAXD, dmax = -MAXD; for (int d = -MAXD; d <= MAXD; d++) if (st->diffhist[d + MAXD]) { if (d < dmin) dmin = d; if (d > dmax) dmax = d; }
        unsigned long long inband = 0, tot = st->bigdiff; for (int d = -MAXD; d <= MAXD; d++) { tot += st->diffhist[d + MAXD]; if (d >= -(int)strength && d <= (int)strength) inband += st->diffhist[d + MAXD]; }
        printf(" [%d, %d] beyond +-%d: %llu; |diff|<=s: %.4f%%;  carried rem+thr:", dmin, dmax, MAXD, st->bigdiff, 100.0 * inband / tot);
        for (int r = 0; r < 33; r++) if (st->rhist[r]) printf(" %d:%llu", r - 16, st->rhist[r]);
        printf("\n");
    }
    return 0;
}
