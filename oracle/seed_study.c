/*
 * seed_study.c -- FEASIBILITY STUDY (analysis tool, test infrastructure, never shipped).
 *
 * Question behind the round-4 "seeded" enumeration of the segment-parallel row engine: for (strength, bleed) pairs whose chain-state
 * set (delta, cn, th) is far too large to enumerate (s = 85, bleed 1: ~10^5 states), is it enough to start K pixels IN FRONT of a
 * segment boundary from a small seed set -- every possible left byte, with nothing carried (cn = th = 0) -- and let the carried terms
 * contract?  Measured along the reference's own trajectory (running histogram) with the seeds stepped against the histogram frozen at
 * the start of the row, as the engine does:
 *   miss(K)   boundaries whose TRUE state is not among the states the seeds reach after K steps
 *   sets(K)   distinct states the seeds reach (how many lanes run on, how wide the dense tables must be)
 * usage: seed_study W H mode strength bleed [L]        (FS_FILE=raw rgba file instead of the generator; KSET=a,b,c,d the four run-in lengths; ROWSTEP=n every n-th row;
 *        VARIANT=0 seeds without carried terms, 1 (default) with the carried terms of the diff that explains the left byte, 2 / 3 a second / third family per diff -- the steady
 *        state of flat regions; SHOWMISS=k lists the first misses at the k-th run-in length).
 * Round 6 (profiles/r06_seed_study.txt): the same question for EXHAUSTIVE state sets (s = 19 b = 2): may the units of a batch start from ~47 seeds eight pixels in front of them
 * instead of from all 253 states?  Photographic content: no miss in 10^4 .. 10^5 boundaries; flat regions hold fixed points the seeds miss (the engine falls back).
 */
#include "pngloss_port.c"

extern void pngloss_synth_rgba(unsigned char *rgba, uint32_t width, uint32_t height, int mode, uint64_t frame);

typedef struct { int left, cn, th; } cstate;
#define NK 4
static int KS[NK] = { 8, 16, 24, 32 };
static unsigned long long n_bound[F_COUNT], n_miss[F_COUNT][NK], n_sets[F_COUNT][NK], max_set[F_COUNT][NK], n_seedless[F_COUNT], n_dirty[F_COUNT][NK];
static int VARIANT = 1, CMAX, TMAX;
static int ROWSTEP = 1;
static int SEGL = 32;

/* decision tables of the frozen histogram (per row and filter): leader = arg-max of (H, O), lowest v among equals, of the band's
 * prefix [bandlo, v] and suffix [v, bandhi]; a clamp cuts a band from one side only */
#define TOFF 320
static int16_t PRE[2][2 * TOFF], SUF[2][2 * TOFF];
static void build_tables(const uint32_t *Hf, const uint32_t *O, int s)
{
    const int q = s + 1;
    for (int sgn = 0; sgn < 2; sgn++)
        for (int v = -TOFF; v < TOFF; v++) {
            PRE[sgn][v + TOFF] = SUF[sgn][v + TOFF] = 0;
            if (sgn ? v > 0 : v < 0) continue;
            const int t = (sgn ? -v : v) / q, blo = sgn ? -(t * q) - s : t * q, bhi = blo + s;
            int L = blo;
            for (int u = blo + 1; u <= v; u++) if (Hf[u & 255] > Hf[L & 255] || (Hf[u & 255] == Hf[L & 255] && O[u & 255] > O[L & 255])) L = u;
            PRE[sgn][v + TOFF] = (int16_t)L;
            L = v;
            for (int u = v + 1; u <= bhi; u++) if (Hf[u & 255] > Hf[L & 255] || (Hf[u & 255] == Hf[L & 255] && O[u & 255] > O[L & 255])) L = u;
            SUF[sgn][v + TOFF] = (int16_t)L;
        }
}
static int g_oob;
/* one step of channel c of filter f at pixel x against the frozen histogram Hf (optimize_state.c:131-254 for one channel) */
static void fstep(const engine *e, uint32_t y, int f, unsigned s, long bleed, const uint32_t *Hf, uint32_t x, uint32_t c, cstate *st)
{
    const uint32_t bpp = e->bpp;
    const size_t stride = (size_t)e->W * bpp;
    const unsigned char *orig = e->pix + (size_t)y * stride;
    const unsigned char *nabove = y ? orig - stride : NULL;
    const uint32_t *O = e->orig_hist[f];
    const int q = (int)s + 1;
    const size_t o = (size_t)x * bpp + c;
    const int pl = plane_of(bpp, c);
    const int ov = orig[o];
    const int above = nabove ? nabove[o] : 0;
    const int diag = (nabove && x) ? nabove[o - bpp] : 0;
    const int pred = predict(f, above, diag, st->left);
    const bool transparent = (bpp % 2) == 0 && orig[(size_t)x * bpp + bpp - 1] == 0;
    int back, d;
    if (transparent && c == bpp - 1) { back = 0; d = 0; }
    else {
        const int err = sext16(e->E0[(size_t)x * 4 + pl] + st->cn);
        const int osym = sext8(ov - pred);
        const int predc = ov - osym;
        const int filt = osym + err;
        int vmin, vmax;
        if (filt < 0) { vmax = -((-filt) - ((-filt) % q)); vmin = vmax - (int)s; }
        else          { vmin = filt - (filt % q);          vmax = vmin + (int)s; }
        const int blo = vmin;
        const int lo = -predc, hi = 255 - predc;
        vmin = med3(vmin, lo, hi); vmax = med3(vmax, lo, hi);
        int best;
        if (vmin < -TOFF || vmax >= TOFF) { g_oob++; best = vmin; }
        else {
            const int sgn = filt < 0;
            best = vmin > blo ? SUF[sgn][vmin + TOFF] : PRE[sgn][vmax + TOFF];
            if (vmin == vmax) best = vmin;
            else if (osym >= vmin && osym <= vmax && Hf[osym & 255] == Hf[best & 255] && O[osym & 255] == O[best & 255]) best = osym;
        }
        back = best + predc;
        d = sext16(filt - best);
    }
    int parts[5];
    port_sierra_split(d, bleed, parts);
    st->left = back; st->cn = parts[4] + st->th; st->th = parts[1];
}

static int cmp_state(const void *a, const void *b)
{
    const cstate *p = a, *q = b;
    if (p->left != q->left) return p->left - q->left;
    if (p->cn != q->cn) return p->cn - q->cn;
    return p->th - q->th;
}

static void study_row(const engine *e, uint32_t y, int f, unsigned s, long bleed, const candidate *cd, int dmax)
{
    const uint32_t W = e->W, bpp = e->bpp;
    const size_t stride = (size_t)W * bpp;
    const unsigned char *orig = e->pix + (size_t)y * stride;
    static cstate *seeds; static int cap;
    const int nseed = 1024;
    if (cap < nseed) { cap = nseed; seeds = realloc(seeds, sizeof(cstate) * cap); }
    for (uint32_t c = 0; c < bpp; c++) {
        const int pl = plane_of(bpp, c);
        for (uint32_t b = (uint32_t)SEGL; b < W; b += (uint32_t)SEGL) {
            /* the true state in front of pixel b, from the reference's outputs of b-1 and b-2 */
            int p1[5], p2[5];
            port_sierra_split(cd->diff16[(size_t)(b - 1) * 4 + pl], bleed, p1);
            port_sierra_split(b >= 2 ? cd->diff16[(size_t)(b - 2) * 4 + pl] : 0, bleed, p2);
            const cstate truth = { cd->bytes[(size_t)(b - 1) * bpp + c], p1[4] + p2[1], p1[1] };
            n_bound[f]++;
            for (int ki = 0; ki < NK; ki++) {
                const int K = KS[ki];
                const uint32_t x0 = (int)b >= K ? b - (uint32_t)K : 0u;
                {
                    /* does the reference's own trajectory follow the FROZEN machine over the window?  (if not, the row has an epoch there anyway) */
                    int dirty = 0;
                    for (uint32_t x = x0; x < b && !dirty; x++) {
                        int q1[5], q2[5], q0[5];
                        port_sierra_split(x >= 1 ? cd->diff16[(size_t)(x - 1) * 4 + pl] : 0, bleed, q1);
                        port_sierra_split(x >= 2 ? cd->diff16[(size_t)(x - 2) * 4 + pl] : 0, bleed, q2);
                        port_sierra_split(cd->diff16[(size_t)x * 4 + pl], bleed, q0);
                        cstate t = { x >= 1 ? cd->bytes[(size_t)(x - 1) * bpp + c] : 0, q1[4] + q2[1], q1[1] };
                        fstep(e, y, f, s, bleed, e->hist, x, c, &t);
                        const cstate want = { cd->bytes[(size_t)x * bpp + c], q0[4] + q1[1], q0[1] };
                        if (cmp_state(&t, &want)) dirty = 1;
                    }
                    if (dirty) { n_dirty[f][ki]++; continue; }
                }
                int n = 0;
                if (x0 == 0) { seeds[n++] = (cstate){ 0, 0, 0 }; }
                else {
                    const int centre = orig[(size_t)(x0 - 1) * bpp + c] + e->E0[(size_t)(x0 - 1) * 4 + pl];
                    const bool tr = (bpp % 2) == 0 && c == bpp - 1 && orig[(size_t)(x0 - 1) * bpp + bpp - 1] == 0;
                    if (tr) { for (int t = -TMAX; t <= TMAX; t++) seeds[n++] = (cstate){ 0, t, 0 }; }
                    else if (f == 0 || f == 2) {
                        if (VARIANT == 0) seeds[n++] = (cstate){ 0, 0, 0 };
                        else {
                            /* every cn, th on a grid that keeps the set within 256 */
                            int tstep = 1;
                            while ((2 * CMAX + 1) * (2 * (TMAX / tstep) + 1) > 256) tstep++;
                            for (int cn = -CMAX; cn <= CMAX; cn++) for (int th = -(TMAX / tstep) * tstep; th <= TMAX; th += tstep) seeds[n++] = (cstate){ 0, cn, th };
                        }
                    }
                    else if (VARIANT >= 2) {
                        /* by the diff d of the boundary pixel: nothing carried in (fresh); the same diff in the pixels before (steady: flat regions sit in such fixed points);
                         * VARIANT 3: also diff d before with diff 0 two before, ... */
                        for (int d = -(int)s; d <= (int)s; d++) {
                            int p[5]; port_sierra_split(d, bleed, p);
                            const int fam = VARIANT == 2 ? 2 : VARIANT;
                            for (int k = 0; k < fam; k++) {
                                const int carry = k == 0 ? 0 : (k == 1 ? p[4] + p[1] : p[4]);
                                const int thp = k == 0 ? 0 : (k == 1 ? p[1] : 0);
                                const int l = centre + carry - d;
                                if (l < 0 || l > 255) continue;
                                seeds[n++] = (cstate){ l, p[4] + thp, p[1] };
                            }
                        }
                    }
                    else for (int l = 0; l <= 255; l++) {
                        if (VARIANT == 0) { if (abs(l - centre) <= dmax) seeds[n++] = (cstate){ l, 0, 0 }; }
                        else {
                            /* the carried terms that go with this left byte when nothing was carried into the boundary pixel */
                            const int D = centre - l;
                            if (abs(D) > dmax) continue;
                            int p[5]; port_sierra_split(D, bleed, p);
                            seeds[n++] = (cstate){ l, p[4], p[1] };
                        }
                    }
                }
                if (!n) n_seedless[f]++;
                { static int once; if (!once && n > 1) { once = 1; fprintf(stderr, "seeds per boundary: %d\n", n); } }
                for (int i = 0; i < n; i++)
                    for (uint32_t x = x0; x < b; x++) fstep(e, y, f, s, bleed, e->hist, x, c, &seeds[i]);
                qsort(seeds, (size_t)n, sizeof(cstate), cmp_state);
                int u = 0, hit = 0;
                for (int i = 0; i < n; i++) {
                    if (i == 0 || cmp_state(&seeds[i], &seeds[i - 1])) u++;
                    if (!cmp_state(&seeds[i], &truth)) hit = 1;
                }
                if (!hit) n_miss[f][ki]++;
                if (!hit && getenv("SHOWMISS") && ki == atoi(getenv("SHOWMISS")) && n_miss[f][ki] <= 40) {
                    const int cb = orig[(size_t)(b - 1) * bpp + c] + e->E0[(size_t)(b - 1) * 4 + pl];
                    printf("MISS f=%d y=%u b=%u c=%u truth: dleft=%d cn=%d th=%d | reached:", f, y, b, c, truth.left - cb, truth.cn, truth.th);
                    for (int i = 0; i < n; i++) if (i == 0 || cmp_state(&seeds[i], &seeds[i - 1])) printf(" (%d,%d,%d)", seeds[i].left - cb, seeds[i].cn, seeds[i].th);
                    printf("  orig:");
                    for (uint32_t x = b - 6; x < b; x++) printf(" %d", orig[(size_t)x * bpp + c]);
                    printf("\n");
                }
                n_sets[f][ki] += (unsigned long long)u;
                if ((unsigned long long)u > max_set[f][ki]) max_set[f][ki] = (unsigned long long)u;
            }
        }
    }
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s W H mode strength bleed [L]\n", argv[0]); return 2; }
    const uint32_t W = (uint32_t)atoi(argv[1]), H = (uint32_t)atoi(argv[2]);
    const int mode = atoi(argv[3]);
    const unsigned strength = (unsigned)atoi(argv[4]);
    const long bleed = atol(argv[5]);
    if (argc > 6) SEGL = atoi(argv[6]);
    if (getenv("ROWSTEP")) ROWSTEP = atoi(getenv("ROWSTEP"));
    if (getenv("KSET")) sscanf(getenv("KSET"), "%d,%d,%d,%d", &KS[0], &KS[1], &KS[2], &KS[3]);
    unsigned char *rgba = malloc((size_t)W * H * 4);
    if (getenv("FS_FILE")) { FILE *fp = fopen(getenv("FS_FILE"), "rb"); if (!fp || fread(rgba, 4, (size_t)W * H, fp) != (size_t)W * H) { fprintf(stderr, "cannot read %s\n", getenv("FS_FILE")); return 1; } fclose(fp); }
    else pngloss_synth_rgba(rgba, W, H, mode, 0);
    int gray = 1, opaque = 1;
    for (size_t i = 0; i < (size_t)W * H; i++) { const unsigned char *p = rgba + i * 4; gray &= (p[0] == p[1]) & (p[1] == p[2]); opaque &= p[3] == 255; }
    const uint32_t bpp = gray ? (opaque ? 1 : 2) : (opaque ? 3 : 4);
    unsigned char *pix = malloc((size_t)W * H * bpp);
    for (size_t i = 0; i < (size_t)W * H; i++) {
        const unsigned char *s = rgba + i * 4; unsigned char *d = pix + i * bpp;
        switch (bpp) { case 1: d[0] = s[1]; break; case 2: d[0] = s[1]; d[1] = s[3]; break; case 3: d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; break; default: memcpy(d, s, 4); }
    }
    /* reach of the left byte around orig + incoming error: |diff| <= s plus the largest carried term */
    int rmax = 0, tmax = 0;
    for (int d = -(int)strength; d <= (int)strength; d++) { int p[5]; port_sierra_split(d, bleed, p); if (abs(p[4]) > rmax) rmax = abs(p[4]); if (abs(p[1]) > tmax) tmax = abs(p[1]); }
    const int dmax = (int)strength + rmax + tmax;
    CMAX = rmax + tmax; TMAX = tmax;
    if (getenv("VARIANT")) VARIANT = atoi(getenv("VARIANT"));
    const size_t stride = (size_t)W * bpp;
    engine e; memset(&e, 0, sizeof e);
    e.W = W; e.H = H; e.bpp = bpp; e.pix = pix;
    e.old_above = calloc(stride, 1);
    e.E0 = calloc((size_t)W * 4, sizeof(int16_t)); e.E1 = calloc((size_t)W * 4, sizeof(int16_t));
    candidate cand[F_COUNT];
    for (int f = 0; f < F_COUNT; f++) { cand[f].bytes = calloc(stride, 1); cand[f].diff16 = calloc((size_t)W * 4, sizeof(int16_t)); }
    port_orig_histograms(pix, W, H, bpp, e.orig_hist);
    for (uint32_t y = 0; y < H; y++) {
        const bool adaptive = y == 0;
        const unsigned char *nabove = y ? pix + (size_t)(y - 1) * stride : NULL;
        unsigned s = strength; int winner = -1; uint64_t best_cost = UINT64_MAX;
        for (;;) {
            for (int f = 0; f < F_COUNT; f++) {
                run_chain(&e, y, f, s, bleed, &cand[f]);
                if (s == strength && y % (uint32_t)ROWSTEP == 0) build_tables(e.hist, e.orig_hist[f], (int)s);
                if (s == strength && y % (uint32_t)ROWSTEP == 0) study_row(&e, y, f, s, bleed, &cand[f], dmax);
                if (adaptive && port_adaptive_filter(nabove, cand[f].bytes, W, bpp) != f) cand[f].cost = UINT64_MAX;
                else cand[f].cost = derivative_error(&e, y, &cand[f]) / 128 + entropy_cost(&e, y, f, &cand[f]);
                if (cand[f].cost < best_cost) { best_cost = cand[f].cost; winner = f; }
            }
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
    printf("W=%u H=%u mode=%d bpp=%u s=%u b=%ld L=%d variant %d reach +-%d cmax %d tmax %d oob %d\n", W, H, mode, bpp, strength, bleed, SEGL, VARIANT, dmax, CMAX, TMAX, g_oob);
    for (int f = 0; f < F_COUNT; f++) {
        printf("%-5s %llu;", fn[f], n_bound[f]);
        for (int ki = 0; ki < NK; ki++) printf("  K=%d: miss %llu dirty %llu, states %.1f (max %llu)", KS[ki], n_miss[f][ki], n_dirty[f][ki], n_bound[f] ? (double)n_sets[f][ki] / n_bound[f] : 0.0, max_set[f][ki]);
        printf("\n");
    }
    return 0;
}
