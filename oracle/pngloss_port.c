/*
 * pngloss_port.c -- CPU ORACLE (test infrastructure only; see pngloss_port.h for the rules).
 *
 * Restatement of the reference's per-scanline optimiser, written from the semantics, not
 * from the reference text.  Structure per row y:
 *
 *   pre   : nothing (E0 holds the incoming Sierra error for row y)
 *   chain : for each of the 5 candidate filters, a strictly serial walk over x that only carries
 *             left pixel, rem(x-1), thr(x-2) and the running symbol histogram
 *           and records, per pixel, the chosen byte and the int16 quantisation error ("diff16")
 *   post  : (vectorisable over x) derivative error metric, libpng heuristic filter, entropy cost
 *   commit: winner's bytes -> image, winner's diff16 -> next two error rows, winner's histogram -> state
 *
 * Algebra used here and in the HIP kernels, each proven against the real reference by tests/test_oracle.py:
 *   (1) reference: predicted +-= 256 so that orig-predicted in [-128,127]   (optimize_state.c:175-182)
 *       here     : osym = sext8(orig - pred),  pred' = orig - osym,  filtered = here - pred' = osym + err
 *   (2) reference: three-way clamp of [min,max] to reconstructable bytes    (optimize_state.c:195-210)
 *       here     : min = med3(min, lo, hi), max = med3(max, lo, hi) with lo = -pred', hi = 255 - pred'
 *   (3) reference: ascending scan with replace-rules                         (optimize_state.c:212-244)
 *       here     : lexicographic arg-max of (H[v], O_f[v], v==osym, -v)
 *   (4) reference: ulog2(UINTMAX_MAX / freq) by shifting                     (optimize_state.c:338,565-572)
 *       here     : 33 + clz32(freq)
 *   (5) reference: 10 "+=" into three int16 error rows per pixel            (optimize_state.c:445-467)
 *       here     : rem/thr carried in registers along the chain, the eight next-row terms summed after the row
 *                  from the stored diff16 (int16 wrap is additive, so the order does not matter)
 *   (6) reference: three optimize_state copies + original_frequency x3       (pngloss_image.c:172-189)
 *       here     : original_frequency once; candidates write into their own scratch, commit by copy of the winner
 */
#include "pngloss_port.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* self-check of variant 2 (test aid): n > 0 verifies before every pixel that each usable band's recorded leader is the unique
 * maximum of its bins, and reports up to n violations on stderr */
static int g_lead_selfcheck = 0;
void port_lead_selfcheck(int n) { g_lead_selfcheck = n; }
int port_lead_selfcheck_left(void) { return g_lead_selfcheck; }
enum { F_NONE = 0, F_SUB, F_UP, F_AVG, F_PAETH, F_COUNT };

static const unsigned char png_filter_flag[F_COUNT] = { 0x08, 0x10, 0x20, 0x40, 0x80 };

/* ------------------------------------------------------------------ small pure helpers */

static inline int paeth(int above, int diag, int left)
{
    int p = above - diag, pd = left - diag;
    int pl = p < 0 ? -p : p;             /* distance to left  */
    int pa = pd < 0 ? -pd : pd;          /* distance to above */
    int pg = (p + pd) < 0 ? -(p + pd) : (p + pd);
    if (pl <= pa && pl <= pg) return left;
    return pa <= pg ? above : diag;
}

static inline int predict(int f, int above, int diag, int left)
{
    switch (f) {
    case F_SUB:   return left;
    case F_UP:    return above;
    case F_AVG:   return (above + left) >> 1;
    case F_PAETH: return paeth(above, diag, left);
    default:      return 0;
    }
}

static inline int sext8(int v)  { return (int)(int8_t)(uint8_t)(v & 0xff); }
static inline int sext16(int v) { return (int)(int16_t)(uint16_t)(v & 0xffff); }
static inline int med3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* which of the four error planes a pixel channel uses (color_delta.c:4-41, optimize_state.c:167-171) */
static inline int plane_of(uint32_t bpp, uint32_t c) { return (bpp == 2 && c == 1) ? 3 : (int)c; }
/* how many of the four expanded delta lanes carry this channel (gray is replicated into r,g,b) */
static inline int weight_of(uint32_t bpp, uint32_t c) { return (bpp <= 2 && c == 0) ? 3 : 1; }

unsigned port_symbol_cost(uint32_t freq)
{
    /* bit length of floor((2^64-1)/freq) == 64 - floor(log2 freq) for 1 <= freq < 2^32 */
    return freq ? 33u + (unsigned)__builtin_clz(freq) : 0u;
}

void port_sierra_split(int diff16, long bleed, int parts[5])
{
    long d = (long)diff16 / bleed;       /* C division truncates toward zero */
    long t = d / 16;  d -= 4 * t;
    long h = d / 8;   d -= 2 * h;
    long f = (d * 2) / 9; d -= 2 * f;
    long v = d / 2;   d -= v;
    parts[0] = (int)t; parts[1] = (int)h; parts[2] = (int)f; parts[3] = (int)v; parts[4] = (int)d;
}

/* ------------------------------------------------------------------ fully parallel pieces */

void port_classify(const unsigned char *const *rows, uint32_t width, uint32_t height, int *grayscale, int *opaque)
{
    int g = 1, o = 1;
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) {
            const unsigned char *p = rows[y] + (size_t)x * 4;
            g &= (p[0] == p[1]) & (p[1] == p[2]);
            o &= (p[3] == 255);
        }
    *grayscale = g;
    *opaque = o;
}

void port_orig_histograms(const unsigned char *pix, uint32_t width, uint32_t height, uint32_t bpp, uint32_t out[5][256])
{
    memset(out, 0, sizeof(uint32_t) * 5 * 256);
    const size_t stride = (size_t)width * bpp;
    for (uint32_t y = 0; y < height; y++) {
        const unsigned char *row = pix + y * stride;
        const unsigned char *up = y ? row - stride : NULL;
        for (size_t i = 0; i < stride; i++) {
            int here = row[i];
            int left = i >= bpp ? row[i - bpp] : 0;
            int above = up ? up[i] : 0;
            int diag = (up && i >= bpp) ? up[i - bpp] : 0;
            for (int f = 0; f < F_COUNT; f++)
                out[f][(here - predict(f, above, diag, left)) & 255]++;
        }
    }
}

int port_adaptive_filter(const unsigned char *above_row, const unsigned char *row, uint32_t width, uint32_t bpp)
{
    uint32_t sum[F_COUNT] = { 0, 0, 0, 0, 0 };
    const size_t n = (size_t)width * bpp;
    for (size_t i = 0; i < n; i++) {
        int here = row[i];
        int left = i >= bpp ? row[i - bpp] : 0;
        int above = above_row ? above_row[i] : 0;
        int diag = (above_row && i >= bpp) ? above_row[i - bpp] : 0;
        for (int f = 0; f < F_COUNT; f++) {
            int b = (here - predict(f, above, diag, left)) & 255;
            sum[f] += (uint32_t)(b < 128 ? b : 256 - b);
        }
    }
    int best = 0;
    for (int f = 1; f < F_COUNT; f++)
        if (sum[f] < sum[best]) best = f;      /* strict: first minimum in order none,sub,up,avg,paeth */
    return best;
}

/* ------------------------------------------------------------------ the serial chain */

typedef struct {
    unsigned char *bytes;   /* [W*bpp]  candidate row                               */
    int16_t *diff16;        /* [W*4]    quantisation error per error plane, int16   */
    uint32_t hist[256];     /* running symbol histogram after this candidate's row  */
    uint64_t cost;
} candidate;

typedef struct {
    uint32_t W, H, bpp;
    unsigned char *pix;       /* packed image: rows < y already optimised, rows >= y original */
    unsigned char *old_above; /* original row y-1 */
    int16_t *E0, *E1;         /* [(W)*4] incoming error for rows y and y+1 (x-major, 4 planes) */
    uint32_t hist[256];
    uint32_t orig_hist[5][256];
} engine;

static inline int better(uint32_t h, uint32_t o, int flag, uint32_t bh, uint32_t bo, int bflag)
{
    if (h != bh) return h > bh;
    if (o != bo) return o > bo;
    return flag > bflag;        /* equal keys: the later (larger) v only wins if it is the original symbol */
}

static void run_chain(const engine *e, uint32_t y, int f, unsigned s, long bleed, candidate *cd)
{
    const uint32_t W = e->W, bpp = e->bpp;
    const size_t stride = (size_t)W * bpp;
    const unsigned char *orig = e->pix + (size_t)y * stride;
    const unsigned char *nabove = y ? orig - stride : NULL;
    const uint32_t *O = e->orig_hist[f];
    uint32_t *Hs = cd->hist;
    const int q = (int)s + 1;
    const bool has_alpha = (bpp % 2) == 0;

    memcpy(Hs, e->hist, sizeof(e->hist));
    int rem[4] = { 0, 0, 0, 0 }, thr_prev[4] = { 0, 0, 0, 0 }, thr_cur[4] = { 0, 0, 0, 0 };

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
            if (transparent && c == bpp - 1) {
                back = 0;                      /* keep fully transparent pixels fully transparent */
                sym = (0 - pred) & 255;
                d16[pl] = 0;
            } else {
                const int err = sext16(e->E0[(size_t)x * 4 + pl] + rem[pl] + thr_prev[pl]);
                const int osym = sext8(ov - pred);
                const int predc = ov - osym;
                const int filt = osym + err;
                int vmin, vmax;
                if (filt < 0) { vmax = -((-filt) - ((-filt) % q)); vmin = vmax - (int)s; }
                else          { vmin = filt - (filt % q);          vmax = vmin + (int)s; }
                const int lo = -predc, hi = 255 - predc;
                vmin = med3(vmin, lo, hi);
                vmax = med3(vmax, lo, hi);
                int best = vmin;
                uint32_t bh = Hs[vmin & 255], bo = O[vmin & 255];
                int bflag = (vmin == osym);
                for (int v = vmin + 1; v <= vmax; v++) {
                    uint32_t h = Hs[v & 255], oo = O[v & 255];
                    int fl = (v == osym);
                    if (better(h, oo, fl, bh, bo, bflag)) { best = v; bh = h; bo = oo; bflag = fl; }
                }
                back = best + predc;
                if (back < 0 || back > 255) { fprintf(stderr, "port: reconstruction %d out of range\n", back); abort(); }
                sym = best & 255;
                d16[pl] = sext16(filt - best);   /* == (int16)(here - back) */
            }
            cd->bytes[o] = (unsigned char)back;
            Hs[sym]++;
        }
        /* replicated gray lanes 1,2 and the unused alpha lane are never read back; only real planes are kept */
        for (int pl = 0; pl < 4; pl++) {
            cd->diff16[(size_t)x * 4 + pl] = (int16_t)d16[pl];
            int parts[5];
            port_sierra_split(d16[pl], bleed, parts);
            thr_prev[pl] = thr_cur[pl];
            thr_cur[pl] = parts[1];
            rem[pl] = parts[4];
        }
    }
}

/* ------------------------------------------------------------------ GPU-shaped variant of the chain
 *
 * port_set_chain_variant(1) switches run_chain to the formulation the HIP row engine uses, so that the algorithm
 * itself (not only its CPU-friendly restatement) is checked against the real reference on the CPU:
 *   - the secondary key is the 8-bit RANK of original_frequency, folded with "is the original symbol" and the
 *     candidate index into one word (key2), arg-maxed after the primary key (running frequency);
 *   - the channels of a pixel are evaluated SPECULATIVELY against the histogram as it was before the pixel, then
 *     repaired in channel order by re-evaluating only the bins the earlier channels incremented.
 */
static int g_chain_variant = 0;
void port_set_chain_variant(int v) { g_chain_variant = v; }
static int g_force_filter = -1;   /* debugging aid: >= 0 makes that candidate the winner of every row */
void port_set_force_filter(int f) { g_force_filter = f; }

static inline uint32_t key2(uint32_t rank, int jj, int josym)
{
    return ((rank << 9) | ((jj == josym) ? 256u : 0u) | (uint32_t)(255 - jj)) + 1u;
}

static void run_chain_spec(const engine *e, uint32_t y, int f, unsigned s, long bleed, candidate *cd)
{
    const uint32_t W = e->W, bpp = e->bpp;
    const size_t stride = (size_t)W * bpp;
    const unsigned char *orig = e->pix + (size_t)y * stride;
    const unsigned char *nabove = y ? orig - stride : NULL;
    const uint32_t *O = e->orig_hist[f];
    uint32_t rank[256];
    for (int b = 0; b < 256; b++) { uint32_t r = 0; for (int k = 0; k < 256; k++) r += O[k] < O[b]; rank[b] = r; }
    uint32_t *Hs = cd->hist;
    const int q = (int)s + 1;
    const bool has_alpha = (bpp % 2) == 0;
    memcpy(Hs, e->hist, sizeof(e->hist));
    int rem[4] = { 0, 0, 0, 0 }, thr_prev[4] = { 0, 0, 0, 0 }, thr_cur[4] = { 0, 0, 0, 0 };

    for (uint32_t x = 0; x < W; x++) {
        int vmin[4], span[4], josym[4], predc[4], filt[4], tr[4], jwin[4];
        uint32_t Hwin[4], K[4], Rwin[4];
        const bool transparent = has_alpha && orig[(size_t)x * bpp + bpp - 1] == 0;
        for (uint32_t c = 0; c < bpp; c++) {           /* "all four DPP rows at once": only pre-pixel state is read */
            const size_t o = (size_t)x * bpp + c;
            const int pl = plane_of(bpp, c);
            const int ov = orig[o];
            const int above = nabove ? nabove[o] : 0;
            const int diag = (nabove && x) ? nabove[o - bpp] : 0;
            const int left = x ? cd->bytes[o - bpp] : 0;
            const int pred = predict(f, above, diag, left);
            const int osym = sext8(ov - pred);
            predc[c] = ov - osym;
            const int err = sext16(e->E0[(size_t)x * 4 + pl] + rem[pl] + thr_prev[pl]);
            filt[c] = osym + err;
            const int af = filt[c] < 0 ? -filt[c] : filt[c];
            const int base = (af / q) * q;
            int lo_v = filt[c] < 0 ? -base - (int)s : base, hi_v = lo_v + (int)s;
            const int lo = -predc[c], hi = lo + 255;
            lo_v = med3(lo_v, lo, hi); hi_v = med3(hi_v, lo, hi);
            tr[c] = transparent && c == bpp - 1;
            if (tr[c]) { lo_v = hi_v = -pred; predc[c] = pred; }
            vmin[c] = lo_v; span[c] = hi_v - lo_v; josym[c] = osym - lo_v;
            uint32_t m = 0;
            for (int jj = 0; jj <= span[c]; jj++) { uint32_t h = Hs[(lo_v + jj) & 255]; if (h > m) m = h; }
            uint32_t kk = 0;
            for (int jj = 0; jj <= span[c]; jj++) {
                int b = (lo_v + jj) & 255;
                if (Hs[b] == m) { uint32_t k2 = key2(rank[b], jj, josym[c]); if (k2 > kk) kk = k2; }
            }
            Hwin[c] = m; K[c] = kk;
            jwin[c] = 255 - (int)((kk - 1u) & 255u); Rwin[c] = (kk - 1u) >> 9;
        }
        for (uint32_t cp = 0; cp + 1 < bpp; cp++) {    /* exact repair, in channel order */
            const uint32_t sb = (uint32_t)(vmin[cp] + jwin[cp]) & 255u, sH = Hwin[cp] + 1u, sR = Rwin[cp];
            for (uint32_t c = cp + 1; c < bpp; c++) {
                const int jj2 = ((int)sb - vmin[c]) & 255;
                const uint32_t K2 = key2(sR, jj2, josym[c]);
                if (jj2 <= span[c] && (sH > Hwin[c] || (sH == Hwin[c] && K2 > K[c]))) {
                    Hwin[c] = sH; K[c] = K2; jwin[c] = jj2; Rwin[c] = sR;
                }
            }
        }
        int d16[4] = { 0, 0, 0, 0 };
        for (uint32_t c = 0; c < bpp; c++) {
            const int vwin = vmin[c] + jwin[c];
            cd->bytes[(size_t)x * bpp + c] = (unsigned char)(vwin + predc[c]);
            d16[plane_of(bpp, c)] = tr[c] ? 0 : sext16(filt[c] - vwin);
            Hs[vwin & 255]++;
        }
        for (int pl = 0; pl < 4; pl++) {
            cd->diff16[(size_t)x * 4 + pl] = (int16_t)d16[pl];
            int parts[5];
            port_sierra_split(d16[pl], bleed, parts);
            thr_prev[pl] = thr_cur[pl]; thr_cur[pl] = parts[1]; rem[pl] = parts[4];
        }
    }
}

/* ------------------------------------------------------------------ "band leader" variant of the chain (variant 2)
 *
 * The formulation of the round-2 HIP row engine (pl_engine.hip, chain_lead), proven here on the CPU first.
 * Observation: without the clamp the candidate set of a channel is one of a FIXED partition of v-space into bands
 * [tq, tq+s] (filt >= 0) and [-tq-s, -tq] (filt < 0) (optimize_state.c:186-193), and the choice inside a band is the
 * lexicographic arg-max of (H[v], O_f[v], v==osym, -v) (:212-244).  Keep, per tracked band, its LEADER
 * L = argmax (H, O_f, -v) and whether that maximum of (H, O_f) is unique.  Then:
 *   - a pixel channel whose band is usable and whose leader can be reconstructed (lo <= L <= hi, the clamp of
 *     :195-210) chooses exactly L: the clamped range is a subset of the band that contains the band's unique maximum,
 *     and "v == osym" only breaks (H, O_f) ties, of which there are none;
 *   - bumping the leader of a usable band changes no usable band's state, so the four channels of a pixel decouple
 *     and NO per-pixel gather, reduction or channel repair is needed -- one table lookup per channel;
 *   - anything else takes the exact sequential evaluation and then rescans the bands its bumps touched.
 * Bands of opposite sign OVERLAP in histogram bins (bin b is v = b in a positive band and v = b - 256 in a negative
 * one), so "changes no usable band's state" has one exception: the leader bin u of band B may lie in a band A of the other
 * sign whose leader is another bin.  Bumping u through B's fast path then raises a NON-leader of A, which is harmless
 * exactly as long as u stays strictly below A's leader in (H, O_f).  Every such (u, leader of A) is a WATCHED RELATION; the
 * moment a bump would make u catch up, A is rescanned (here: immediately, the counts being current; on the GPU, where the
 * bumps of a chunk are applied later, the first pixel at which a relation breaks is found when they are applied and that
 * pixel is redone exactly, which rescans A through the ordinary slow path -- same results, both being exact).
 * LIGHT pixels.  If the leader is clamped away and what the clamp leaves of the band is a SINGLE value, that value is the answer
 * whatever the histogram says (saturated pixels).  Its bin need not lead any band; the rule that keeps every usable band's
 * state true is the same one, stated for any bump: the bumped bin stays strictly below the leader of every usable band that
 * holds it, unless it is that leader (band_watch_breaks checks exactly this for all bins a pixel bumps).
 * Geometry.  Filters with a data dependent clamp (sub, up, average, paeth): bands t < 256/q, the clamp is checked per
 * pixel.  Filter none: the prediction is 0, so the re-centred prediction is 0 (orig <= 127, "P pixels", v = byte) or 256
 * (orig >= 128, "N pixels", v = byte - 256) and the clamp is static: positive bands are cut to [.., 255], negative ones to
 * [-256, -1] (so v = 0 is not in the negative zero band), a P pixel with filt < 0 can only choose v = 0 and an N pixel
 * with filt >= 0 only v = -1; these two are served by the zero bands when those are led by exactly that value.
 * port_lead_stats() reports how often each case occurs (the GPU engine's speed is the fast fraction).
 */
typedef struct { int L; int ok; int usable; } band_state;   /* ok: L is the unique (H,O) maximum AND the scan is fresh */
static unsigned long long g_lead_stats[8 * 6];   /* [0..7] all chains, [8+8f..] chain f: pixels, fast, light (among the fast), slow:untracked/unusable/watched relation, slow:clamp, slow:forced, scans, rows */
#define LSTAT(i) do { g_lead_stats[i]++; g_lead_stats[8 + 8 * g_lead_f + (i)]++; } while (0)
static int g_lead_f;
void port_lead_stats(unsigned long long out[48], int reset)
{
    if (out) memcpy(out, g_lead_stats, sizeof g_lead_stats);
    if (reset) memset(g_lead_stats, 0, sizeof g_lead_stats);
}

typedef struct { int q, s, NP, none; const uint32_t *O; } band_geo;
/* band ids: 0..NP-1 = positive bands t, NP..2NP-1 = negative bands t */
static inline int band_lo(const band_geo *g, int id)
{
    if (id < g->NP) return id * g->q;
    const int t = id - g->NP, lo = -t * g->q - g->s;
    return (g->none && lo < -256) ? -256 : lo;
}
static inline int band_hi(const band_geo *g, int id)
{
    if (id < g->NP) { const int hi = id * g->q + g->s; return (g->none && hi > 255) ? 255 : hi; }
    const int t = id - g->NP;
    return (g->none && t == 0) ? -1 : -t * g->q;
}
static inline int band_of_bin(const band_geo *g, int bin, int neg)   /* the band of that sign holding the bin, or -1 */
{
    if (!neg) { const int t = bin / g->q; return t < g->NP ? t : -1; }
    int v;
    if (bin) v = bin - 256; else if (g->none) v = -256; else return g->NP ? g->NP : -1;   /* general: v = 0 sits in negative band 0 too */
    const int t = (-v) / g->q;
    return t < g->NP ? g->NP + t : -1;
}
static inline int band_has_bin(const band_geo *g, int id, int bin)
{
    int v;
    if (id < g->NP) v = bin;
    else v = bin ? bin - 256 : (g->none ? -256 : 0);
    return v >= band_lo(g, id) && v <= band_hi(g, id);
}
static void band_scan(const uint32_t *Hs, const band_geo *g, int id, band_state *b)
{
    const int v0 = band_lo(g, id), v1 = band_hi(g, id);
    int L = v0; uint32_t bh = Hs[v0 & 255], bo = g->O[v0 & 255]; int uniq = 1;
    for (int v = v0 + 1; v <= v1; v++) {
        const uint32_t h = Hs[v & 255], o = g->O[v & 255];
        if (h > bh || (h == bh && o > bo)) { L = v; bh = h; bo = o; uniq = 1; }
        else if (h == bh && o == bo) uniq = 0;
    }
    b->L = L; b->ok = uniq; b->usable = 0;
    LSTAT(6);
}
static void band_rebuild_for_bin(const uint32_t *Hs, const band_geo *g, band_state *B, int bin)
{
    for (int neg = 0; neg < 2; neg++) {
        const int id = band_of_bin(g, bin, neg);
        if (id < 0) continue;
        /* cheap test: a fresh band whose leader is not this bin and still beats it strictly keeps its state */
        const band_state *b = &B[id];
        const int lbin = b->L & 255;
        const uint32_t hnew = Hs[bin], onew = g->O[bin];
        const int need = !b->ok || (lbin == bin ? 0 : (hnew > Hs[lbin] || (hnew == Hs[lbin] && onew >= g->O[lbin])));
        if (need) { band_scan(Hs, g, id, &B[id]); B[id].usable = B[id].ok; }
    }
}
/* Would one of the bumps of a fast pixel (bins u[0..n-1], in channel order) make a bin catch up with the leader of the band of
 * the other sign that holds it -- a watched relation break?  Then the pixel is not fast: a later channel may have looked at
 * that band.  Counts the pixel's own earlier bumps. */
static inline int band_watch_breaks(const uint32_t *Hs, const band_geo *g, const band_state *B, const int *u, int n)
{
    for (int c = 0; c < n; c++)
        for (int neg = 0; neg < 2; neg++) {
            const int a = band_of_bin(g, u[c], neg);
            if (a < 0 || !B[a].usable) continue;
            const int l = B[a].L & 255;
            if (l == u[c]) continue;
            uint32_t hu = Hs[u[c]] + 1, hl = Hs[l];
            for (int k = 0; k < c; k++) { hu += u[k] == u[c]; hl += u[k] == l; }
            if (!(hu < hl || (hu == hl && g->O[u[c]] < g->O[l]))) return 1;
        }
    return 0;
}
/* the band a lookup with this filt lands in (-1: not tracked) and, for filter none's two one-value cases, the value it is forced to */
static inline int band_of_lookup(const band_geo *g, int filt, int npixel, int *forced, int *fv)
{
    *forced = 0;
    if (g->none) {
        if (!npixel && filt < 0) { *forced = 1; *fv = 0; return g->NP ? 0 : -1; }
        if (npixel && filt >= 0) { *forced = 1; *fv = -1; return g->NP ? g->NP : -1; }
    }
    const int t = (filt < 0 ? -filt : filt) / g->q;
    if (t >= g->NP) return -1;
    if (g->none && filt > 255) return -1;
    if (g->none && filt < -256) return -1;
    return filt < 0 ? g->NP + t : t;
}

static void run_chain_lead(const engine *e, uint32_t y, int f, unsigned s, long bleed, candidate *cd)
{
    const uint32_t W = e->W, bpp = e->bpp;
    const size_t stride = (size_t)W * bpp;
    const unsigned char *orig = e->pix + (size_t)y * stride;
    const unsigned char *nabove = y ? orig - stride : NULL;
    const uint32_t *O = e->orig_hist[f];
    uint32_t *Hs = cd->hist;
    const int q = (int)s + 1;
    band_geo g = { q, (int)s, f == F_NONE ? (256 + q - 1) / q : 256 / q, f == F_NONE, O };
    const int NP = g.NP;
    const bool has_alpha = (bpp % 2) == 0;
    band_state B[2 * 256 + 2];
    memcpy(Hs, e->hist, sizeof(e->hist));
    g_lead_f = f;
    for (int id = 0; id < 2 * NP; id++) band_scan(Hs, &g, id, &B[id]);
    for (int id = 0; id < 2 * NP; id++) B[id].usable = B[id].ok;
    LSTAT(7);
    int rem[4] = { 0, 0, 0, 0 }, thr_prev[4] = { 0, 0, 0, 0 }, thr_cur[4] = { 0, 0, 0, 0 };

    for (uint32_t x = 0; x < W; x++) {
        int d16[4] = { 0, 0, 0, 0 };
        const bool transparent = has_alpha && orig[(size_t)x * bpp + bpp - 1] == 0;
        int pred[4], osym[4], filt[4], lo[4], vfast[4], tr[4];
        int why = 0, light = 0;
        /* fast attempt: every channel looks only at the band states as they were before the pixel */
        for (uint32_t c = 0; c < bpp; c++) {
            const size_t o = (size_t)x * bpp + c;
            const int pl = plane_of(bpp, c);
            const int ov = orig[o];
            const int above = nabove ? nabove[o] : 0;
            const int diag = (nabove && x) ? nabove[o - bpp] : 0;
            const int left = x ? cd->bytes[o - bpp] : 0;
            pred[c] = predict(f, above, diag, left);
            tr[c] = transparent && c == bpp - 1;
            osym[c] = sext8(ov - pred[c]);
            lo[c] = osym[c] - ov;
            const int err = sext16(e->E0[(size_t)x * 4 + pl] + rem[pl] + thr_prev[pl]);
            filt[c] = osym[c] + err;
            int forced, fv = 0, id;
            if (tr[c]) {
                /* forced symbol (0 - pred) mod 256 (optimize_state.c:158-164), looked up as v = sext8(-pred) (filter none:
                 * pred = 0, a P pixel): fine iff that band is usable and led by exactly this v */
                fv = sext8(-pred[c]);
                id = band_of_lookup(&g, fv, 0, &forced, &fv);
                forced = 1;
                vfast[c] = -pred[c];
                lo[c] = -pred[c];
            } else {
                id = band_of_lookup(&g, filt[c], ov >= 128, &forced, &fv);
            }
            const int usable = id >= 0 && B[id].usable;
            const int L = usable ? B[id].L : 0;
            if (usable && forced && L == fv) {
                if (!tr[c]) vfast[c] = fv;
                continue;
            }
            if (tr[c] && !g.none) {
                /* a forced symbol does not depend on the histogram either: light, under the same rule (vfast is set above) */
                light = 1;
                continue;
            }
            if (usable && forced) { why = why ? why : 5; continue; }
            if (usable && L >= lo[c] && L <= lo[c] + 255) { vfast[c] = L; continue; }
            /* no usable leader inside the clamp.  If what the clamp leaves of the band is a single value, that value is the
             * answer whatever the histogram says ("light" pixel: the bump goes to a bin that need not lead any band, and
             * band_watch_breaks below holds it to the same rule as every other bump) */
            if (!tr[c] && !g.none) {
                const int fl = filt[c];
                int vmin, vmax;
                if (fl < 0) { vmax = -((-fl) - ((-fl) % q)); vmin = vmax - (int)s; }
                else        { vmin = fl - (fl % q);          vmax = vmin + (int)s; }
                vmin = med3(vmin, lo[c], lo[c] + 255);
                vmax = med3(vmax, lo[c], lo[c] + 255);
                if (vmin == vmax) { vfast[c] = vmin; light = 1; continue; }
            }
            why = why ? why : (id < 0 ? 2 : !usable ? 3 : 4);
            continue;
        }
        if (!why) {
            int ub[4];
            for (uint32_t c = 0; c < bpp; c++) ub[c] = vfast[c] & 255;
            if (band_watch_breaks(Hs, &g, B, ub, (int)bpp)) why = 6;      /* a watched relation would break inside this pixel */
        }
        if (g_lead_selfcheck > 0) {
            for (int id = 0; id < 2 * NP; id++) {
                if (!B[id].usable) continue;
                band_state tmp; const unsigned long long k1 = g_lead_stats[6], k2 = g_lead_stats[8 + 8 * g_lead_f + 6];
                band_scan(Hs, &g, id, &tmp); g_lead_stats[6] = k1; g_lead_stats[8 + 8 * g_lead_f + 6] = k2;
                if (tmp.L != B[id].L || !tmp.ok) { g_lead_selfcheck--; fprintf(stderr, "stale usable band: y=%u f=%d x=%u id=%d stored L=%d true L=%d uniq=%d (q=%d NP=%d)\n", y, f, x, id, B[id].L, tmp.L, tmp.ok, q, NP); break; }
            }
        }
        LSTAT(0);
        if (!why) {
            LSTAT(1);
            if (light) g_lead_stats[8 + 8 * g_lead_f + 2]++, g_lead_stats[2]++;   /* slot 2: light pixels (counted among the fast ones) */
            for (uint32_t c = 0; c < bpp; c++) {
                const int pl = plane_of(bpp, c);
                cd->bytes[(size_t)x * bpp + c] = (unsigned char)(vfast[c] - lo[c]);
                d16[pl] = tr[c] ? 0 : sext16(filt[c] - vfast[c]);
                Hs[vfast[c] & 255]++;
            }
        } else {
            LSTAT((why == 6 || why == 2) ? 3 : why);
            /* exact sequential evaluation (the reference's own order), then rescan what the bumps touched */
            for (uint32_t c = 0; c < bpp; c++) {
                const int pl = plane_of(bpp, c);
                int best;
                if (tr[c]) { best = -pred[c]; d16[pl] = 0; }
                else {
                    const int fl = filt[c];
                    int vmin, vmax;
                    if (fl < 0) { vmax = -((-fl) - ((-fl) % q)); vmin = vmax - (int)s; }
                    else        { vmin = fl - (fl % q);          vmax = vmin + (int)s; }
                    vmin = med3(vmin, lo[c], lo[c] + 255);
                    vmax = med3(vmax, lo[c], lo[c] + 255);
                    best = vmin;
                    uint32_t bh = Hs[vmin & 255], bo = O[vmin & 255];
                    int bflag = (vmin == osym[c]);
                    for (int v = vmin + 1; v <= vmax; v++) {
                        uint32_t h = Hs[v & 255], oo = O[v & 255];
                        int fg = (v == osym[c]);
                        if (better(h, oo, fg, bh, bo, bflag)) { best = v; bh = h; bo = oo; bflag = fg; }
                    }
                    d16[pl] = sext16(fl - best);
                }
                cd->bytes[(size_t)x * bpp + c] = (unsigned char)(best - lo[c]);
                Hs[best & 255]++;
                band_rebuild_for_bin(Hs, &g, B, best & 255);
            }
        }
        for (int pl = 0; pl < 4; pl++) {
            cd->diff16[(size_t)x * 4 + pl] = (int16_t)d16[pl];
            int parts[5];
            port_sierra_split(d16[pl], bleed, parts);
            thr_prev[pl] = thr_cur[pl]; thr_cur[pl] = parts[1]; rem[pl] = parts[4];
        }
    }
}

/* ------------------------------------------------------------------ per-row post pass */

static uint64_t derivative_error(const engine *e, uint32_t y, const candidate *cd)
{
    const uint32_t W = e->W, bpp = e->bpp;
    const size_t stride = (size_t)W * bpp;
    const unsigned char *orig = e->pix + (size_t)y * stride;
    const unsigned char *nabove = y ? orig - stride : NULL;
    const unsigned char *oabove = y ? e->old_above : NULL;
    uint64_t total = 0;
    for (uint32_t x = 0; x < W; x++)
        for (uint32_t c = 0; c < bpp; c++) {
            const size_t o = (size_t)x * bpp + c;
            const int ov = orig[o], back = cd->bytes[o];
            const int na = nabove ? nabove[o] : 0, oa = oabove ? oabove[o] : 0;
            const int nd = (nabove && x) ? nabove[o - bpp] : 0, od = (oabove && x) ? oabove[o - bpp] : 0;
            const int nl = x ? cd->bytes[o - bpp] : 0, ol = x ? orig[o - bpp] : 0;
            const int da = (oa - ov) - (na - back);
            const int dd = (od - ov) - (nd - back);
            const int dl = (ol - ov) - (nl - back);
            total += (uint64_t)weight_of(bpp, c) * (uint64_t)(da * da + dd * dd + dl * dl);
        }
    return total;
}

static uint32_t entropy_cost(const engine *e, uint32_t y, int f, const candidate *cd)
{
    const uint32_t W = e->W, bpp = e->bpp;
    const size_t stride = (size_t)W * bpp;
    const unsigned char *nabove = y ? e->pix + (size_t)(y - 1) * stride : NULL;
    uint32_t total = 0;
    for (size_t i = 0; i < stride; i++) {
        int left = i >= bpp ? cd->bytes[i - bpp] : 0;
        int above = nabove ? nabove[i] : 0;
        int diag = (nabove && i >= bpp) ? nabove[i - bpp] : 0;
        total += port_symbol_cost(cd->hist[(cd->bytes[i] - predict(f, above, diag, left)) & 255]);
    }
    return total;
}

static void commit_error_rows(engine *e, const candidate *win, long bleed)
{
    /* row+1 target x gets t(x+2)+f(x+1)+v(x)+f(x-1)+t(x-2); row+2 target x gets t(x+1)+h(x)+t(x-1) */
    const uint32_t W = e->W;
    for (uint32_t x = 0; x < W; x++)
        for (int pl = 0; pl < 4; pl++) {
            int c1 = 0, c2 = 0;
            for (int dx = -2; dx <= 2; dx++) {
                long sx = (long)x + dx;
                if (sx < 0 || sx >= (long)W) continue;
                int parts[5];
                port_sierra_split(win->diff16[(size_t)sx * 4 + pl], bleed, parts);
                int ad = dx < 0 ? -dx : dx;
                c1 += ad == 2 ? parts[0] : (ad == 1 ? parts[2] : parts[3]);
                if (ad <= 1) c2 += ad == 1 ? parts[0] : parts[1];
            }
            e->E0[(size_t)x * 4 + pl] = (int16_t)sext16(e->E1[(size_t)x * 4 + pl] + c1);
            e->E1[(size_t)x * 4 + pl] = (int16_t)sext16(c2);
        }
}

/* ------------------------------------------------------------------ image driver */

int port_optimize_packed(unsigned char *pix, uint32_t width, uint32_t height, uint32_t bpp,
                         unsigned char *row_filters, unsigned strength, long bleed, port_trace *trace)
{
    if (!width || !height) return 0;
    const size_t stride = (size_t)width * bpp;
    engine e;
    memset(&e, 0, sizeof e);
    e.W = width; e.H = height; e.bpp = bpp; e.pix = pix;
    e.old_above = calloc(stride, 1);
    e.E0 = calloc((size_t)width * 4, sizeof(int16_t));
    e.E1 = calloc((size_t)width * 4, sizeof(int16_t));
    candidate cand[F_COUNT];
    int ok = e.old_above && e.E0 && e.E1;
    for (int f = 0; f < F_COUNT; f++) {
        cand[f].bytes = calloc(stride, 1);
        cand[f].diff16 = calloc((size_t)width * 4, sizeof(int16_t));
        ok = ok && cand[f].bytes && cand[f].diff16;
    }
    if (ok) {
        port_orig_histograms(pix, width, height, bpp, e.orig_hist);
        for (uint32_t y = 0; y < height; y++) {
            const bool adaptive = !row_filters || y == 0;   /* PNG: first row is always filtered adaptively */
            const unsigned char *nabove = y ? pix + (size_t)(y - 1) * stride : NULL;
            unsigned s = strength;
            int winner = -1;
            uint64_t best_cost = UINT64_MAX;
            uint64_t costs[F_COUNT];
            for (;;) {
                for (int f = 0; f < F_COUNT; f++) {
                    if (g_chain_variant == 2) run_chain_lead(&e, y, f, s, bleed, &cand[f]);
                    else if (g_chain_variant) run_chain_spec(&e, y, f, s, bleed, &cand[f]);
                    else run_chain(&e, y, f, s, bleed, &cand[f]);
                    if (adaptive && port_adaptive_filter(nabove, cand[f].bytes, width, bpp) != f) {
                        cand[f].cost = UINT64_MAX;
                    } else {
                        cand[f].cost = derivative_error(&e, y, &cand[f]) / 128 + entropy_cost(&e, y, f, &cand[f]);
                    }
                    if (g_force_filter >= 0) cand[f].cost = f == g_force_filter ? 0 : UINT64_MAX;
                    costs[f] = cand[f].cost;
                    if (cand[f].cost < best_cost) { best_cost = cand[f].cost; winner = f; }
                }
                if (winner >= 0) break;
                if (s == 0) { fprintf(stderr, "port: no acceptable filter at row %u\n", y); abort(); }
                s--;
            }
            if (trace) {
                if (trace->cost) memcpy(trace->cost + (size_t)y * F_COUNT, costs, sizeof costs);
                if (trace->strength_used) trace->strength_used[y] = (uint8_t)s;
                if (trace->winner) trace->winner[y] = (uint8_t)winner;
            }
            memcpy(e.old_above, pix + (size_t)y * stride, stride);
            memcpy(pix + (size_t)y * stride, cand[winner].bytes, stride);
            memcpy(e.hist, cand[winner].hist, sizeof e.hist);
            commit_error_rows(&e, &cand[winner], bleed);
            if (row_filters) row_filters[y] = png_filter_flag[winner];
        }
        if (trace && trace->final_hist) memcpy(trace->final_hist, e.hist, sizeof e.hist);
    }
    for (int f = 0; f < F_COUNT; f++) { free(cand[f].bytes); free(cand[f].diff16); }
    free(e.old_above); free(e.E0); free(e.E1);
    return ok ? 0 : 17;
}

int port_optimize_with_rows(unsigned char **rows, uint32_t width, uint32_t height, unsigned char *row_filters,
                            bool verbose, uint_fast8_t quantization_strength, int_fast16_t bleed_divider)
{
    (void)verbose;
    if (!width || !height) return 0;
    int gray, opaque;
    port_classify((const unsigned char *const *)rows, width, height, &gray, &opaque);
    const uint32_t bpp = gray ? (opaque ? 1 : 2) : (opaque ? 3 : 4);
    unsigned char *pix = malloc((size_t)width * height * bpp);
    if (!pix) return 17;
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) {
            const unsigned char *s = rows[y] + (size_t)x * 4;
            unsigned char *d = pix + ((size_t)y * width + x) * bpp;
            switch (bpp) {
            case 1: d[0] = s[1]; break;
            case 2: d[0] = s[1]; d[1] = s[3]; break;
            case 3: d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; break;
            default: memcpy(d, s, 4); break;
            }
        }
    int rc = port_optimize_packed(pix, width, height, bpp, row_filters, quantization_strength, bleed_divider, NULL);
    if (rc == 0)
        for (uint32_t y = 0; y < height; y++)
            for (uint32_t x = 0; x < width; x++) {
                unsigned char *d = rows[y] + (size_t)x * 4;
                const unsigned char *s = pix + ((size_t)y * width + x) * bpp;
                switch (bpp) {
                case 1: d[0] = d[1] = d[2] = s[0]; d[3] = 255; break;
                case 2: d[0] = d[1] = d[2] = s[0]; d[3] = s[1]; break;
                case 3: d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = 255; break;
                default: memcpy(d, s, 4); break;
                }
            }
    free(pix);
    return rc;
}
