/*
 * pl_engine.hip -- the row engine: pngloss's per-scanline optimiser as ONE persistent gfx950 kernel.
 *
 * Replaces, for a whole image, the reference's
 *     optimize_image            /root/reference/src/pngloss_image.c:159-333   (row loop, 5-filter search, retry, commit)
 *     optimize_state_row / _run /root/reference/src/optimize_state.c:292-361, 114-290
 *     diffuse_color_error       /root/reference/src/optimize_state.c:390-490
 *     adaptive_filter_for_rows  /root/reference/src/optimize_state.c:492-562
 *     color_delta.c             /root/reference/src/color_delta.c:4-66
 *
 * One workgroup = one image (blockIdx.x = image of the batch), 8 wavefronts, persistent over all rows.  The five candidate
 * filters of a row are independent (pngloss_image.c:213,240): waves 0..4 run their serial chains concurrently (up, sub,
 * average, paeth, none); all eight waves run the passes that are parallel over x (post pass, commit).
 *
 * Two formulations of the chain live here, both exact:
 *   band-leader chains (round 2, chain_lead; section "BAND-LEADER CHAINS" below): per pixel and channel ONE
 *     dependent LDS lookup in a decision table built from the leaders of the quantisation bands, hand-scheduled loops
 *     (pl_lead_asm.h) on four lanes, histogram bumps deferred and applied 64 pixels at a time under a rule that keeps the table
 *     true (watched relations), exceptions settled without the histogram where it cannot matter (light pixels) and by the
 *     round-1 evaluation otherwise;
 *   round-1 chains (chain_row): lanes 16c..16c+15 = channel c of the current pixel (a DPP row) hold the <= s+1 candidate symbols
 *     of that channel's band (optimize_state.c:186-214); per pixel: predict -> re-centre -> band -> clamp, gather of {running
 *     frequency, rank of original frequency}, two DPP arg-max reductions of the reference's 4-level key, exact repair of the
 *     coupling between the channels of a pixel (optimize_state.c:221,253).  They take the rows the band-leader chains cannot
 *     (q > 128, exploding errors) and the rows where those are measurably slower (adaptive choice per row, from cycle counts).
 * The Sierra terms that feed the same row (x+1, x+2) ride in the table entries / registers; the eight terms for the next two
 * rows, the derivative error metric, libpng's heuristic and the entropy cost are deferred to the passes parallel over x.
 *
 * The x-chain and the row-to-row dependence through the winner's histogram are inherently serial (SURVEY.md
 * Appendix C): this kernel is bound by that dependency chain, not by HBM and not by MFMA.
 */
#include "pl_device.h"

#include <atomic>
#include <cstdio>
#include <type_traits>

#ifndef PL_SEGPROF
#define PL_SEGPROF 0   /* timing experiments only: s_memtime stamps inside the chain loop (perturbs it) */
#endif
#ifndef PL_ABLATE
#define PL_ABLATE 0   /* timing experiments only (tools/ablate.sh): >0 removes pieces of the chain, results become wrong */
#endif

namespace {

/* ---- DPP helpers (wave64, 16-lane rows) ---------------------------------------------------------------- */
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
/* max over the 16 lanes of a DPP row, result in every lane of the row: row_ror 8,4,2,1 butterfly */
__device__ __forceinline__ uint32_t rowmax_u32(uint32_t v)
{
    v = max(v, dpp_u32<0x128>(v));
    v = max(v, dpp_u32<0x124>(v));
    v = max(v, dpp_u32<0x122>(v));
    v = max(v, dpp_u32<0x121>(v));
    return v;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float recip_up(int d)
{
    /* smallest float strictly above the correctly rounded 1/d: see pl_truncdiv_f */
    return __uint_as_float(__float_as_uint(1.0f / (float)d) + 2u);   /* +2 ulp: safe even if the device division were 1 ulp low */
}

/* The 17-bit secondary key: rank of original frequency, "is the original symbol", then lowest candidate index.
 * +1 so that 0 can mean "not a candidate". */
__device__ __forceinline__ uint32_t key2(uint32_t rank, int jj, int josym)
{
    return ((rank << 9) | ((jj == josym) ? 256u : 0u) | (uint32_t)(255 - jj)) + 1u;
}

/* LDS pointers keep their address space across the (non-inlined) per-filter chain functions, so the compiler emits
 * ds_* instead of flat_* accesses */
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x2 lds_uint2;
typedef __attribute__((address_space(3))) u32x4 lds_uint4;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
/* pointers loaded from the job table are generic; accesses through them would be FLAT instructions, which count in
 * lgkmcnt and make every wait for an LDS result also wait for outstanding global stores.  Name the address space. */
typedef __attribute__((address_space(1))) uint32_t glb_u32;
typedef __attribute__((address_space(1))) const uint32_t glb_cu32;
typedef __attribute__((address_space(1))) const u32x2 glb_cu32x2;

#define PL_TBL_N 512               /* 256 bins + 64 per-lane dummy slots, padded so that every table is 4 KB aligned */
#define PL_CHUNK 32                /* pixels per vector pre-phase; small so that several workgroups fit one CU's LDS */

struct RowCtx {
    const uint32_t *row;      /* original row y (slots)                                  */
    const uint32_t *nabove;   /* optimised row y-1 or nullptr                            */
    const uint2 *err0;        /* incoming error for row y                                */
    uint4 *cand;              /* cand[5][W]: per candidate, per pixel: 4 x (byte | diff16<<8) */
    lds_uint2 *tbl;           /* tbl[5][PL_TBL_N] {H, rank<<9} in LDS                    */
    lds_uint4 *rec;           /* this WAVE's chunk records: [PL_CHUNK][4][1 or 2]         */
    lds_u32 *lut;             /* Sierra split table: [diff+256] -> rem | thr<<16, |diff|<=255 */
    uint32_t W, y, bpp;
    int s;
    float rq, rbleed, r29;
    uint32_t slow;            /* out: pixels that took the exact-repair slow path        */
    unsigned long long seg[4];/* out (PL_SEGPROF): cycles in head+gather | reductions | check | tail */
};

__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));   /* |a - b| */
    return r;
}

__device__ __forceinline__ int med3_i32(int v, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}

/* max over an aligned group of GL (16 or 8) lanes, result in every lane of the group */
template <int GL>
__device__ __forceinline__ uint32_t groupmax_u32(uint32_t v)
{
    if (GL == 16) return rowmax_u32(v);
    v = max(v, dpp_u32<0xB1>(v));    /* quad_perm [1,0,3,2] */
    v = max(v, dpp_u32<0x4E>(v));    /* quad_perm [2,3,0,1] */
    v = max(v, dpp_u32<0x141>(v));   /* row_half_mirror     */
    return v;
}

/* ---------------------------------------------------------------------------------------------------------
 * The serial chain of one wave over one row.
 *   MODE 1,3,4  one candidate filter (sub, average, paeth): 16 lanes per channel
 *   MODE 5      TWO candidate filters whose prediction does not depend on the left neighbour (none = half 0, up =
 *               half 1 of every 16-lane DPP row): 8 lanes per (channel, filter).  Five chains on four SIMDs would
 *               make two waves share a SIMD, and the younger of two DPP-heavy waves on one SIMD runs at half rate
 *               (profiles/r01_ubench_simd_sharing.txt) -- so the two cheapest chains share one wave instead.
 *   NCT  candidates per lane known at compile time (1..4) or 0 for the generic two-sweep loop
 *   TR   the image class has alpha (2 or 4 B/px): honour optimize_state.c:158-164 for fully transparent pixels
 *   WRAP false when every incoming error of this row is <= 8000 in magnitude (decided by the commit pass): then no
 *        intermediate can leave int16 (DESIGN.md "int16 wrap") and the sign-extensions are dropped
 * Per pixel (all channels at once): band -> candidates -> gather -> 2 DPP arg-max reductions, speculative w.r.t. the
 * histogram bumps of the earlier channels of the same pixel; a 1-pass test (one ds_bpermute) proves the speculation
 * harmless -- the common case -- or sends the pixel down the exact sequential repair.
 * --------------------------------------------------------------------------------------------------------- */
template <int MODE, int NCT, bool TR, bool WRAP>
__device__ __forceinline__ void chain_row(RowCtx &k, const int lane)
{
    constexpr bool PAIR = MODE == 5;
    constexpr int GL = PAIR ? 8 : 16;                     /* lanes per (channel[, filter]) group */
    const int c = lane >> 4, jl = lane & (GL - 1);
    const int half = PAIR ? (lane >> 3) & 1 : 0;
    const int filt_id = PAIR ? 2 * half : MODE;           /* which candidate filter this lane works for */
    const uint32_t bpp = (uint32_t)__builtin_amdgcn_readfirstlane((int)k.bpp);
    const uint32_t W = (uint32_t)__builtin_amdgcn_readfirstlane((int)k.W);
    glb_cu32 *const row = (glb_cu32 *)k.row, *const nabove = (glb_cu32 *)k.nabove;
    glb_cu32x2 *const err0 = (glb_cu32x2 *)k.err0;
    glb_u32 *const outp = (glb_u32 *)(k.cand + (size_t)filt_id * W);
    const bool active = (uint32_t)c < bpp;
    const int s = __builtin_amdgcn_readfirstlane(k.s), q = s + 1;
    const int nc = NCT ? NCT : (q + GL - 1) / GL;
    const float rq = k.rq, rbleed = k.rbleed, r29 = k.r29;
    lds_uint2 *const T = k.tbl + filt_id * PL_TBL_N;
    lds_uint4 *const R = k.rec;
    lds_u32 *const LUT = k.lut;
    /* lane constants */
    const bool upd = active && jl == 0;                 /* the one lane per group that bumps the histogram        */
    const uint32_t inc = upd ? 1u : 0u;
    const int dummy = PL_NSYM + lane;
    const bool chk = active && jl < c && jl < 3;        /* lane (c, k<c) checks channel k's bump against channel c */
    /* the same facts as sign/bit masks, so that the per-pixel test stays in the VALU: a VALU-written lane mask that
     * is read by the SALU costs ~16-28 cycles on gfx950 (profiles/r01_ubench_mask_traffic.txt) */
    const int nochk_neg = chk ? 0 : (int)0x80000000;    /* forces "no conflict" in lanes that do not check            */
    const int inact_neg = active ? 0 : (int)0x80000000; /* forces "in range" in lanes of unused channel rows          */
    const int upd_and = upd ? 255 : 0, upd_or = upd ? 0 : dummy;
    const bool sel0 = jl == 0, sel1 = jl == 1;          /* which earlier channel this lane checks                  */
    uint32_t slow = 0;
    unsigned long long seg0 = 0, seg1 = 0, seg2 = 0, seg3 = 0, tprev = PL_SEGPROF ? __builtin_readcyclecounter() : 0;

    int left = 0, rem = 0, thr_prev = 0, thr_cur = 0;

    constexpr int RS = PAIR ? 2 : 1;                    /* records per (pixel, channel) */
    for (uint32_t x0 = 0; x0 < W; x0 += PL_CHUNK) {
        /* ---- vector pre-phase: lane = pixel x0+lane; everything that does not depend on the chain ---- */
        bool chunk_has_transparent = false;
        if (lane < PL_CHUNK) {
            const uint32_t xl = x0 + lane;
            const bool ok = xl < W;
            const uint32_t o = ok ? row[xl] : 0u;
            const uint32_t a = (ok && nabove) ? nabove[xl] : 0u;
            const uint32_t d = (ok && nabove && xl) ? nabove[xl - 1] : 0u;
            const u32x2 e = ok ? err0[xl] : (u32x2){ 0u, 0u };
            const bool alpha0 = TR && (bpp & 1u) == 0u && ((o >> (8u * (bpp - 1u))) & 255u) == 0u;   /* only 2 and 4 B/px have alpha */
            chunk_has_transparent = alpha0;
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                const int p = pl_plane_of_channel(bpp, cc);
                const int e0 = pl_sext16((int)(p < 2 ? (e.x >> (16 * p)) : (e.y >> (16 * (p - 2)))));
                const int orig = (o >> (8 * cc)) & 255, above = (a >> (8 * cc)) & 255, diag = (d >> (8 * cc)) & 255;
                const uint32_t trbit = (alpha0 && (uint32_t)cc == bpp - 1u) ? (1u << 16) : 0u;
                if (PAIR) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int pred = h ? above : 0;
                        const int osym = pl_sext8(orig - pred);
                        R[(lane * 4 + cc) * RS + h] = (u32x4){ (uint32_t)osym, (uint32_t)(osym - orig), (uint32_t)pred | trbit,
                                                              (uint32_t)(WRAP ? e0 : osym + e0) };
                    }
                } else {
                    R[(lane * 4 + cc) * RS] = (u32x4){ (uint32_t)orig, (uint32_t)above, (uint32_t)diag | trbit, (uint32_t)e0 };
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        const int n = (int)min((uint32_t)PL_CHUNK, W - x0);
        /* the serial part of the chunk, compiled twice: with and without the fully-transparent-pixel handling, chosen per
         * chunk (most chunks of most images have no alpha == 0 pixel, and the handling costs ~6 issue slots per pixel) */
        auto serial_part = [&](auto trx_tag) {
        constexpr bool TRX = decltype(trx_tag)::value;
        u32x4 r = R[c * RS + half];
        for (int g = 0; g < n; g += GL) {
        const int m = min(GL, n - g);
        /* one pixel step; called twice per loop iteration (manual 2x unroll: halves the loop-control and back-edge cost
         * and lets the record registers alternate instead of being copied) */
        auto pixel = [&](const int ii) {
            const int i = g + ii;
            const u32x4 rn = R[(((i + 1) & (PL_CHUNK - 1)) * 4 + c) * RS + half];   /* prefetch the next pixel's record */
            unsigned long long tA = 0, tC = 0, tD = 0, tE = 0;
            if (PL_SEGPROF) { tA = __builtin_readcyclecounter(); seg3 += tA - tprev; }

            /* ---- uniform per group: optimize_state.c:157-210 ---- */
            int osym, lo, predraw, filt;
            if (PAIR) {
                osym = (int)r.x; lo = (int)r.y; predraw = (int)(r.z & 0xffffu);
                filt = WRAP ? osym + pl_sext16((int)r.w + rem + thr_prev) : (int)r.w + rem + thr_prev;
            } else {
                const int orig = (int)r.x;
                if (MODE == 4) {
                    /* Paeth with three v_sad_u32 (|a-b|+0) instead of sub/neg/max triplets */
                    const uint32_t ab = r.y, dg = r.z & 0xffffu, lf = (uint32_t)left;
                    const uint32_t dl = sad_u32(ab, dg), da = sad_u32(lf, dg), dd = sad_u32(lf + ab, dg + dg);
                    predraw = (dl <= da && dl <= dd) ? left : (da <= dd ? (int)ab : (int)dg);
                } else {
                    predraw = pl_predict<MODE>((int)r.y, (int)(r.z & 0xffffu), left);
                }
                osym = pl_sext8(orig - predraw);
                lo = osym - orig;                       /* = -(re-centred prediction), optimize_state.c:175-182 */
                int err = (int)r.w + rem + thr_prev;
                if (WRAP) err = pl_sext16(err);
                filt = osym + err;
            }
            const int tq = (int)((float)filt * rq);     /* trunc(filt / q), exact (pl_device.h) */
            int vmin = __mul24(tq, q) - ((filt >> 31) & s);
            int vmax = vmin + s;
            const int hi = lo + 255;
            vmin = med3_i32(vmin, lo, hi);
            vmax = med3_i32(vmax, lo, hi);
            bool tr = false;
            if (TRX) {
                tr = (r.z >> 16) != 0;                  /* optimize_state.c:158-164 */
                vmin = tr ? -predraw : vmin;
                vmax = tr ? -predraw : vmax;
                lo = tr ? -predraw : lo;
            }
            const int span = vmax - vmin, josym = osym - vmin;

            /* ---- candidates: gather and two-level arg-max (optimize_state.c:212-244).  Lanes beyond the band
             *      re-evaluate its last member, which cannot change a max or an arg-max. ---- */
            uint32_t Hwin, K;
            if (NCT) {
                int jj[NCT ? NCT : 1];
                u32x2 e[NCT ? NCT : 1];
                uint32_t kk[NCT ? NCT : 1];
#pragma unroll
                for (int t = 0; t < NCT; t++) {
                    jj[t] = min(jl + GL * t, span);
                    e[t] = PL_ABLATE >= 5 ? (u32x2){ (uint32_t)jj[t], 0u } : T[(vmin + jj[t]) & 255];
                }
                uint32_t hm = e[0].x;
#pragma unroll
                for (int t = 0; t < NCT; t++) {
                    kk[t] = e[t].y + ((jj[t] == josym) ? 256u : 0u) + (uint32_t)(256 - jj[t]);
                    hm = max(hm, e[t].x);
                }
                if (PL_SEGPROF) { asm volatile("" : "+v"(hm)); tC = __builtin_readcyclecounter(); seg0 += tC - tA; }
                Hwin = PL_ABLATE >= 4 ? hm : groupmax_u32<GL>(hm);
                uint32_t km = 0;
#pragma unroll
                for (int t = 0; t < NCT; t++) km = max(km, e[t].x == Hwin ? kk[t] : 0u);
                K = PL_ABLATE >= 3 ? (km | 1u) : groupmax_u32<GL>(km);
            } else {
                uint32_t hm = 0;
                for (int t = 0; t < nc; t++) hm = max(hm, T[(vmin + min(jl + GL * t, span)) & 255].x);
                Hwin = groupmax_u32<GL>(hm);
                uint32_t km = 0;
                for (int t = 0; t < nc; t++) {
                    const int j2 = min(jl + GL * t, span);
                    const u32x2 e = T[(vmin + j2) & 255];
                    km = max(km, e.x == Hwin ? e.y + ((j2 == josym) ? 256u : 0u) + (uint32_t)(256 - j2) : 0u);
                }
                K = groupmax_u32<GL>(km);
            }
            if (PL_SEGPROF) { asm volatile("" : "+v"(K)); tD = __builtin_readcyclecounter(); seg1 += tD - tC; }
            int jwin = (int)((0u - K) & 255u);          /* K-1 = rank<<9 | flag<<8 | 255-j */
            int vwin = vmin + jwin;

            /* histogram bump, issued NOW with the speculative winner so that the LDS atomic retires behind the check
             * below instead of in front of the next pixel's record wait; the rare repair path moves it.  EXEC is not
             * touched: non-owner lanes add 0 to a private dummy slot. */
            const int bump_idx = (vwin & upd_and) | upd_or;
            if (PL_ABLATE < 6) __hip_atomic_fetch_add((lds_u32 *)&T[bump_idx], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

            /* ---- did an earlier channel of this pixel bump a bin of my band that is not my winner?
             *      lane (c, k) looks at channel k's (speculative) winner: readlane broadcasts + lane-constant selects,
             *      no LDS round trip; the split-table read issued below overlaps with this arithmetic ---- */
            int sv;
            if (PL_ABLATE >= 1) sv = vwin;
            else if (PAIR) {
                const int a0 = __builtin_amdgcn_readlane(vwin, 0), a1 = __builtin_amdgcn_readlane(vwin, 16), a2 = __builtin_amdgcn_readlane(vwin, 32);
                const int b0 = __builtin_amdgcn_readlane(vwin, 8), b1 = __builtin_amdgcn_readlane(vwin, 24), b2 = __builtin_amdgcn_readlane(vwin, 40);
                const int sa = sel0 ? a0 : (sel1 ? a1 : a2), sb = sel0 ? b0 : (sel1 ? b1 : b2);
                sv = half ? sb : sa;
            } else {
                const int a0 = __builtin_amdgcn_readlane(vwin, 0), a1 = __builtin_amdgcn_readlane(vwin, 16), a2 = __builtin_amdgcn_readlane(vwin, 32);
                sv = sel0 ? a0 : (sel1 ? a1 : a2);
            }

            /* ---- reconstruct (optimize_state.c:251-260); the Sierra terms that stay in this row (:455,467) come
             *      from a 511-entry LDS table of the split, fetched alongside the ds_bpermute above ---- */
            int back = vwin - lo;
            int diff = filt - vwin;
            if (WRAP) diff = pl_sext16(diff);
            if (TRX) diff = tr ? 0 : diff;
            uint32_t le = PL_ABLATE >= 2 ? 0u : LUT[(diff + 256) & 511];
            int remv, thrv;
            int sv2 = sv;

            /* bad  <=>  (checking lane: channel k's winner lies in my band and is not my winner) or |diff| > 255.
             * z >= 0 <=> conflict, w >= 0 <=> table index out of range; one compare, one branch. */
            const int tt = (sv2 - vmin) & 255;
            const int z = ((tt != jwin) ? span - tt : -1) | nochk_neg;
            const int w = (max(diff, -diff) - 256) | inact_neg;
            const bool bad = PL_ABLATE >= 1 ? false : ((z & w) >= 0);
            /* consume the split-table entry HERE: otherwise the compiler sinks its ds_read below the branch, where the
             * whole LDS latency lands on the critical path; issued ~15 slots ago it has (nearly) arrived by now */
            asm volatile("" : "+v"(le));
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad) != 0, 0)) {
                /* exact repair in channel order: channel cp chose bin sb and bumped it to sH */
                slow++;
                uint32_t Rwin = (K - 1u) >> 9;
#pragma unroll
                for (int cp = 0; cp < 3; cp++) {
                    if ((uint32_t)cp + 1u < bpp) {
                        int sb; uint32_t sH, sR;
                        if (PAIR) {
                            const int b0 = __builtin_amdgcn_readlane(vwin, 16 * cp), b1 = __builtin_amdgcn_readlane(vwin, 16 * cp + 8);
                            const int h0 = __builtin_amdgcn_readlane((int)Hwin, 16 * cp), h1 = __builtin_amdgcn_readlane((int)Hwin, 16 * cp + 8);
                            const int r0 = __builtin_amdgcn_readlane((int)Rwin, 16 * cp), r1 = __builtin_amdgcn_readlane((int)Rwin, 16 * cp + 8);
                            sb = half ? b1 : b0; sH = (uint32_t)(half ? h1 : h0) + 1u; sR = (uint32_t)(half ? r1 : r0);
                        } else {
                            sb = __builtin_amdgcn_readlane(vwin, 16 * cp);
                            sH = (uint32_t)__builtin_amdgcn_readlane((int)Hwin, 16 * cp) + 1u;
                            sR = (uint32_t)__builtin_amdgcn_readlane((int)Rwin, 16 * cp);
                        }
                        const int jj2 = (sb - vmin) & 255;
                        const uint32_t K2 = (sR << 9) + ((jj2 == josym) ? 256u : 0u) + (uint32_t)(256 - jj2);
                        const bool better = (c > cp) & (jj2 <= span) & ((sH > Hwin) | ((sH == Hwin) & (K2 > K)));
                        Hwin = better ? sH : Hwin;
                        K = better ? K2 : K;
                        jwin = better ? jj2 : jwin;
                        Rwin = better ? sR : Rwin;
                        vwin = vmin + jwin;
                    }
                }
                /* move the speculative bump to the repaired winner (no-op where nothing changed) */
                __hip_atomic_fetch_add((lds_u32 *)&T[bump_idx], 0u - inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add((lds_u32 *)&T[(vwin & upd_and) | upd_or], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                back = vwin - lo;
                diff = filt - vwin;
                if (WRAP) diff = pl_sext16(diff);
                if (TRX) diff = tr ? 0 : diff;
                const PlSplit sp = pl_sierra_split(diff, rbleed, r29);
                remv = (int)sp.rem;
                thrv = (int)sp.h;
            } else {
                remv = pl_sext16((int)le);
                thrv = (int)le >> 16;
            }
            if (PL_SEGPROF) { asm volatile("" : "+v"(remv), "+v"(thrv)); tE = __builtin_readcyclecounter(); seg2 += tE - tD; tprev = tE; }
            thr_prev = thr_cur;
            thr_cur = thrv;
            rem = remv;
            left = back;
            /* output capture: the pixel's record is dead once `r` holds it, so its first word takes the result (one
             * ds_write, all lanes of the group the same word); lane jl picks up pixel g+jl when the group is done */
            const uint32_t packed = (uint32_t)back | ((uint32_t)diff << 8);
            *(lds_u32 *)&R[((i & (PL_CHUNK - 1)) * 4 + c) * RS + half] = packed;
            r = rn;
        };
        int ii = 0;
        for (; ii + 1 < m; ii += 2) { pixel(ii); pixel(ii + 1); }
        if (ii < m) pixel(ii);
        if (active && jl < m) outp[(size_t)(x0 + g + jl) * 4 + c] = *(lds_u32 *)&R[(((g + jl) & (PL_CHUNK - 1)) * 4 + c) * RS + half];
        }
        };
        if (TR && __builtin_amdgcn_ballot_w64(chunk_has_transparent) != 0) serial_part(std::true_type{});
        else serial_part(std::false_type{});
    }
    k.slow = slow;
    k.seg[0] = seg0; k.seg[1] = seg1; k.seg[2] = seg2; k.seg[3] = seg3;
}

template <int MODE, bool TR, bool WRAP>
__device__ __forceinline__ void chain_dispatch_nc(RowCtx &k, int lane)
{
    const int q = k.s + 1;
    const int per_lane = (q + (MODE == 5 ? 7 : 15)) / (MODE == 5 ? 8 : 16);
    switch (per_lane) {
    case 1: chain_row<MODE, 1, TR, WRAP>(k, lane); break;   /* s <= 15 (paired wave: s <= 7)  */
    case 2: chain_row<MODE, 2, TR, WRAP>(k, lane); break;   /* s <= 31 (<= 15)                */
    case 3: chain_row<MODE, 3, TR, WRAP>(k, lane); break;   /* s <= 47 (<= 23): the default s=19 pairs here */
    case 4: chain_row<MODE, 4, TR, WRAP>(k, lane); break;   /* s <= 63 (<= 31)                */
    case 5: chain_row<MODE, 5, TR, WRAP>(k, lane); break;
    case 6: chain_row<MODE, 6, TR, WRAP>(k, lane); break;   /* s <= 95 (<= 47)                */
    default: chain_row<MODE, 0, TR, WRAP>(k, lane); break;  /* generic two-sweep loop         */
    }
}

template <int MODE>
__device__ __noinline__ void chain_dispatch(RowCtx &k, int lane, bool wrap)
{
    const bool tr = (k.bpp & 1u) == 0;
    if (wrap) {
        /* rare (needs |error| > 8000): one careful variant is enough */
        chain_row<MODE, 0, true, true>(k, lane);
    } else if (tr) chain_dispatch_nc<MODE, true, false>(k, lane);
    else chain_dispatch_nc<MODE, false, false>(k, lane);
}

/* =========================================================================================================
 * Round 2: the BAND-LEADER chain (chain_lead).  Proven on the CPU first: oracle/pngloss_port.c:run_chain_lead.
 *
 * Without the clamp, the candidate set of a channel is one member of a FIXED partition of v-space into bands
 * [tq, tq+s] (filt >= 0) / [-tq-s, -tq] (filt < 0) (optimize_state.c:186-193) and the choice inside it is the arg-max
 * of (H[v], O_f[v], v==osym, -v) (:212-244).  Per band the chain keeps the band's LEADER L = argmax (H, O_f, -v) and whether that
 * maximum of (H, O_f) is unique, folded into a decision table indexed by filt:  T[filt] = { 8*L | 8*rem(filt-L) << 16, 8*thr(filt-L) }.
 * A channel whose band is usable and whose leader is reconstructable (lo <= L <= hi, the clamp of :195-210) chooses
 * exactly L, and bumping a unique leader changes no band's leader: the four channels of a pixel decouple, the
 * histogram is not read at all on this path (its bumps are applied later, 64 pixels per instruction), and one pixel
 * step is ONE dependent LDS lookup plus a handful of adds -- instead of a gather, two DPP reductions and a coupling
 * check.  Everything is kept scaled by 8 (the byte size of a table entry) so that filt IS the table address.
 * Bands of opposite sign overlap in histogram bins; the rule that keeps every usable band's state true under deferred bumps is
 * the WATCHED RELATIONS rule further down (checked when the bumps are applied, lead_flush / flush_verify).
 * A pixel whose table entry is unusable (tie at the top, band not tracked, leader clamped away, forced transparent-alpha
 * symbol that is not its band's leader) shows up as a reconstructed byte outside [0,255]; it is detected per burst of 8-16
 * pixels, and from the first such pixel on the slow section of chain_lead classifies pixel by pixel: back to the loop, LIGHT
 * (the answer does not depend on the histogram: a clamp that leaves one value, a forced symbol) or exact (the gather / arg-max /
 * repair step of round 1, after which the bands its bump touched are rescanned and their table entries rewritten).
 * Preconditions (else the row runs the round-1 chain): q <= 128 and every incoming Sierra error of the row
 * |E0| <= 88, which bounds |filt| <= 252 -- so table addresses need no clamp.
 * ========================================================================================================= */
#define PL_LT_N 512                 /* decision-table entries per chain: filt in [-256, 255] */
#define PL_LT_BADV 0x4000           /* 8*v marker of an unusable entry: reconstructs to a byte far outside 0..255 */
#define PL_LCHUNK 64                /* pixels per vector phase */
#define PL_LGROUP 16                /* pixels per speculative group */
#define PL_LWORK_N 128
#define PL_LREC_N (PL_LCHUNK + 8)   /* the hand-scheduled loop runs up to 4 pixels past the chunk and fetches 2 ahead */
#define PL_E0_LEAD_MAX 88           /* rows with larger incoming |error| take the round-1 chain */

struct LeadCtx {
    const uint32_t *row, *nabove;
    const uint2 *err0;
    uint4 *cand;              /* this chain's candidate row cand[f][W] */
    lds_uint2 *tbl;           /* this chain's {H, rank<<9}[PL_TBL_N] */
    lds_uint2 *T;             /* this chain's decision table [PL_LT_N] */
    lds_u32 *bs;              /* this chain's band states [512]: L+256 | ok<<9 | usable<<10 */
    lds_u32 *work;            /* PL_LWORK_N words: [0..7] ids of rescanned bands, [41] number of watched relations, [48..111] the
                                 relations (leader bin u | bin l of the other band's leader << 8) */
    lds_uint4 *crec;          /* chain records of the chunk: [PL_LCHUNK][4][RW] */
    lds_uint2 *out;           /* results of the chunk: [(2 + PL_LCHUNK)][4] {8*byte (checked) | 8*v << 16, table address} (lead_rec_diff) */
    lds_u32 *lut;             /* Sierra split table [diff+256] -> rem | thr<<16 */
    uint32_t W, bpp;
    int s;
    float rq;
    uint32_t slow;            /* out: pixels redone exactly */
    uint32_t rebuilds;        /* out: band rescans */
    uint32_t light;           /* out: light pixels (single-valued clamp, no histogram read) */
    unsigned long long cyc[7]; /* out (diagnostics): cycles in vector phases | fast groups | exact redo | rescan */
};

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* ---- band geometry (oracle/pngloss_port.c: band_geo and friends) ------------------------------------------------
 * ids 0..NP-1 = positive bands t, NP..2NP-1 = negative bands t.  Filters with a data dependent clamp: bands t < 256/q.
 * Filter none (static clamp, see the oracle's comment): positive bands cut to [.., 255], negative ones to [-256, -1],
 * two decision tables (P pixels: orig <= 127, N pixels: orig >= 128). */
struct LeadGeo { int q, s, NP; bool none; float rq; };
__device__ __forceinline__ LeadGeo lead_geo(int s, bool none, float rq)
{
    LeadGeo g; g.q = s + 1; g.s = s; g.none = none; g.rq = rq;
    g.NP = none ? (256 + s) / (s + 1) : 256 / (s + 1);
    return g;
}
__device__ __forceinline__ int geo_div(const LeadGeo &g, int x) { return (int)((float)x * g.rq); }   /* x / q, 0 <= x <= 512 */
__device__ __forceinline__ int band_lo(const LeadGeo &g, int id)
{
    if (id < g.NP) return id * g.q;
    const int lo = -(id - g.NP) * g.q - g.s;
    return (g.none && lo < -256) ? -256 : lo;
}
__device__ __forceinline__ int band_hi(const LeadGeo &g, int id)
{
    if (id < g.NP) { const int hi = id * g.q + g.s; return (g.none && hi > 255) ? 255 : hi; }
    const int t = id - g.NP;
    return (g.none && t == 0) ? -1 : -t * g.q;
}
__device__ __forceinline__ int band_of_bin(const LeadGeo &g, int bin, bool neg)   /* the band of that sign holding the bin, or -1 */
{
    if (!neg) { const int t = geo_div(g, bin); return t < g.NP ? t : -1; }
    int v;
    if (bin) v = bin - 256; else if (g.none) v = -256; else return g.NP ? g.NP : -1;   /* v = 0 sits in negative band 0 too */
    const int t = geo_div(g, -v);
    return t < g.NP ? g.NP + t : -1;
}
__device__ __forceinline__ bool band_has_bin(const LeadGeo &g, int id, int bin)
{
    const int v = id < g.NP ? bin : (bin ? bin - 256 : (g.none ? -256 : 0));
    return v >= band_lo(g, id) && v <= band_hi(g, id);
}
/* band states in LDS: L + 256 | ok << 9 | usable << 10 | touches-to-skip << 11 | back-off level << 16   (ok: L is the unique
 * (H, rank) maximum and the scan is fresh; usable = ok, unless the band's watched relation found no room in the list) */

/* one decision-table entry.  tab 0: the table (filter none: of the P pixels), tab 1: filter none's N pixels */
__device__ __forceinline__ u32x2 lead_entry_at(const LeadGeo &g, lds_u32 *bs, lds_u32 *LUT, int filt, int tab)
{
    const u32x2 badent = (u32x2){ (uint32_t)PL_LT_BADV, 0u };
    int id; bool forced = false; int fv = 0;
    const int af = abs(filt);
    /* the one-value cases of filter none are kept for |filt| < 64 only (beyond that: exact path), so that a change of a
     * zero band rewrites 64 entries, not 256 */
    if (g.none && tab == 0 && filt < 0) { forced = true; fv = 0; id = (g.NP && filt >= -64) ? 0 : -1; }
    else if (g.none && tab == 1 && filt >= 0) { forced = true; fv = -1; id = (g.NP && filt < 64) ? g.NP : -1; }
    else {
        const int t = geo_div(g, af);
        id = t < g.NP ? (filt < 0 ? g.NP + t : t) : -1;
        if (g.none && (tab == 1) != (filt < 0)) id = -1;
    }
    if (id < 0) return badent;
    const uint32_t st = bs[id];
    if (!(st & 1024u)) return badent;
    const int L = (int)(st & 511u) - 256;
    if (forced && L != fv) return badent;
    const int diff = filt - L;
    if (diff < -256 || diff > 255) return badent;
    const uint32_t le = LUT[diff + 256];
    const int rem = pl_sext16((int)le), thr = (int)le >> 16;
    return (u32x2){ ((uint32_t)(L * 8) & 0xffffu) | ((uint32_t)(rem * 8) << 16), (uint32_t)(thr * 8) };
}

/* rewrite every table entry that depends on band id (all lanes of the wave cooperate) */
__device__ __forceinline__ void lead_write_band_entries(const LeadCtx &k, const LeadGeo &g, int lane, int id)
{
    const bool neg = id >= g.NP;
    const int tab = (g.none && neg) ? 1 : 0;
    lds_uint2 *const T = k.T + tab * PL_LT_N;
    int flo = band_lo(g, id), fhi = band_hi(g, id);
    if (!g.none && neg && fhi == 0) fhi = -1;             /* filt = 0 belongs to positive band 0 */
    for (int f = flo + lane; f <= fhi; f += 64) T[f + 256] = lead_entry_at(g, k.bs, k.lut, f, tab);
    if (g.none && (id == 0 || id == g.NP)) {
        /* the one-value cases served by the zero bands: P pixels with filt < 0, N pixels with filt >= 0 */
        const int f = (id == 0 ? -64 : 0) + lane;
        T[f + 256] = lead_entry_at(g, k.bs, k.lut, f, tab);
    }
}

/* scan of band id by one lane (row start) */
__device__ __forceinline__ uint32_t lead_scan_serial(const LeadGeo &g, lds_uint2 *H, int id)
{
    const int v0 = band_lo(g, id), v1 = band_hi(g, id);
    u32x2 e = H[v0 & 255];
    int L = v0; uint32_t bh = e.x, br = e.y; bool uniq = true;
    /* four bins per round trip (the loads do not depend on the comparisons) */
    for (int v = v0 + 1; v <= v1; v += 4) {
        u32x2 e4[4];
#pragma unroll
        for (int i = 0; i < 4; i++) e4[i] = H[min(v + i, v1) & 255];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (v + i <= v1) {
                if (e4[i].x > bh || (e4[i].x == bh && e4[i].y > br)) { L = v + i; bh = e4[i].x; br = e4[i].y; uniq = true; }
                else if (e4[i].x == bh && e4[i].y == br) uniq = false;
            }
        }
    }
    return (uint32_t)(L + 256) | (uniq ? 512u : 0u);
}

/* WATCHED RELATIONS (oracle: band_watch_breaks).  Bands of opposite sign overlap in histogram bins: the leader bin u of a usable
 * band may lie in the band A of the other sign, whose leader l is another bin.  Bumping u on the fast path then raises a
 * non-leader of A -- harmless exactly as long as u stays strictly below l in (H, rank).  The list work[48..] = (u | l << 8) of
 * all such pairs (work[41] = count) is rebuilt after every change of band states; lead_flush checks it against the bumps it is
 * about to apply.  A band whose relation finds no room in the list is made unusable (safe; only with very many tiny bands). */
#define PL_LREL_MAX 64
#ifndef PL_POST_EARLY
#define PL_POST_EARLY 2u        /* how many of the first chains to finish do their own candidate's post pass (0: none) */
#endif
#define PL_PAETH_BIAS (1u << 27)
#ifndef PL_ADAPT_SLOW
#define PL_ADAPT_SLOW 600u      /* band-leader rows slower than this many cycles per pixel make the kernel try the round-1 chains */
#endif
#ifndef PL_ADAPT_REPROBE
#define PL_ADAPT_REPROBE 32u
#endif
#ifndef PL_PREFETCH_FIX
#define PL_PREFETCH_FIX 1
#endif
/* per-phase cycle counters of the chains (PNGLOSS_HIP_DEBUG prints them).  Every reading drains the LDS queue, eight per chunk:
 * off in the product build (make HIPFLAGS+=-DPL_LEAD_PROF=1 for a profiling build) */
#ifndef PL_LEAD_PROF
#define PL_LEAD_PROF 0
#endif
#define LTIME() (PL_LEAD_PROF ? __builtin_readcyclecounter() : 0ull)
#ifndef PL_LEAD_BURST0_CLEAN
#define PL_LEAD_BURST0_CLEAN 4
#endif
#ifndef PL_LEAD_BURST0_RESTART
#define PL_LEAD_BURST0_RESTART 2
#endif
/* A relation is CLOSE while fewer than PL_LREL_CLOSE bumps of u could break it.  work[112..119] = bitmap of the bins u of the
 * relations that were close when the row stood at pixel work[42]; every pixel bumps at most 4 bins, so until
 * 4 * (pixels since) reaches PL_LREL_CLOSE a range of bumps that touches no marked bin cannot break any relation and
 * lead_flush skips the check.  Re-marked whenever the list is rebuilt and when the pixel budget runs out. */
#ifndef PL_LREL_CLOSE
#define PL_LREL_CLOSE 1024u
#endif
__device__ __forceinline__ void lead_mark_close(const LeadCtx &k, int lane, int xnow)
{
    if (lane < 8) k.work[112 + lane] = 0u;
    if (lane == 8) k.work[42] = (uint32_t)xnow;
    wave_lds_sync();
    const int nrel = (int)k.work[41];
    if (lane < nrel) {
        const uint32_t rel = k.work[48 + lane], ub = rel & 255u;
        const u32x2 eu = k.tbl[ub], el = k.tbl[(rel >> 8) & 255u];
        if (eu.x + PL_LREL_CLOSE >= el.x) __hip_atomic_fetch_or(&k.work[112 + (ub >> 5)], 1u << (ub & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    wave_lds_sync();
}

__device__ __forceinline__ void lead_collect_relations(const LeadCtx &k, const LeadGeo &g, int lane, int xnow)
{
    const int nb = 2 * g.NP;
    if (lane == 0) k.work[41] = 0u;
    wave_lds_sync();
    for (int base = 0; base < nb; base += 64) {
        const int id = base + lane;
        bool dropped = false;
        if (id < nb) {
            const uint32_t st = k.bs[id];
            if (st & 1024u) {
                const int u = ((int)(st & 511u) - 256) & 255;
                const int a = band_of_bin(g, u, id < g.NP);
                if (a >= 0) {
                    const uint32_t sa = k.bs[a];
                    const int l = ((int)(sa & 511u) - 256) & 255;
                    if ((sa & 1024u) && l != u) {
                        const uint32_t slot = __hip_atomic_fetch_add(&k.work[41], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (slot < PL_LREL_MAX) k.work[48 + slot] = (uint32_t)u | ((uint32_t)l << 8);
                        else { k.bs[id] = st & ~1024u; dropped = true; }
                    }
                }
            }
        }
        unsigned long long m = __builtin_amdgcn_ballot_w64(dropped);
        if (m) {
            wave_lds_sync();
            while (m) { const int j = (int)__builtin_ctzll(m); m &= m - 1; lead_write_band_entries(k, g, lane, base + j); }
        }
    }
    wave_lds_sync();
    if (lane == 0 && k.work[41] > PL_LREL_MAX) k.work[41] = PL_LREL_MAX;
    lead_mark_close(k, lane, xnow);
}

/* whole-table rebuild from the chain's histogram (row start / new strength) */
__device__ __forceinline__ void lead_build_table(const LeadCtx &k, const LeadGeo &g, int lane)
{
    const int nb = 2 * g.NP;
    for (int id = lane; id < nb; id += 64) {
        const uint32_t st = lead_scan_serial(g, k.tbl, id);
        k.bs[id] = st | ((st & 512u) << 1);               /* usable = ok */
    }
    wave_lds_sync();
    const int ntab = g.none ? 2 : 1;
    /* (four entries per lane and pass: their band-state and split-table reads are in flight together) */
    for (int idx = lane; idx < ntab * PL_LT_N; idx += 256) {
        u32x2 ent[4];
#pragma unroll
        for (int u = 0; u < 4; u++) ent[u] = lead_entry_at(g, k.bs, k.lut, ((idx + 64 * u) & 511) - 256, (idx + 64 * u) >> 9);
#pragma unroll
        for (int u = 0; u < 4; u++) k.T[idx + 64 * u] = ent[u];
    }
    wave_lds_sync();
    lead_collect_relations(k, g, lane, 0);
}

/* After a slow pixel: row c of the wave (16 lanes) looks after the bin its channel just bumped.  Bands (one per sign)
 * that hold the bin are rescanned unless the cheap test proves their state unchanged (the bin is not the leader and
 * still strictly below it in (H, rank), or it is the leader of a band that was ok); then one lane settles the usable
 * bits of the rescanned bands in priority order and the table entries of every band that changed are rewritten. */
__device__ __forceinline__ void lead_rescan(LeadCtx &k, const LeadGeo &g, int lane, int bin, bool rowactive, lds_u32 *work, int xnow)
{
    const int jl = lane & 15, c = lane >> 4;
    lds_uint2 *const H = k.tbl;
    const int nc = (g.q + 15) >> 4;
    /* work[0..7] = ids of the bands that were rescanned (-1: none), slot 2c + sign */
    /* both tests first (their LDS reads overlap); most slow pixels end here */
    int idt[2]; bool needt[2]; uint32_t levt[2];
    {
        uint32_t stt[2];
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            idt[pass] = rowactive ? band_of_bin(g, bin, pass == 1) : -1;
            stt[pass] = idt[pass] >= 0 ? k.bs[idt[pass]] : 512u;
        }
        const u32x2 eb = H[bin];
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const uint32_t st = stt[pass];
            const int lbin = ((int)(st & 511u) - 256) & 255;
            const u32x2 el = H[lbin];
            levt[pass] = (st >> 16) & 7u;
            bool need = false;
            if (idt[pass] >= 0) {
                if (!(st & 512u)) {
                    /* a band that is not ok (tie at the top, or demoted by a conflict) is rescanned when touched -- with an
                     * exponential back-off while rescans keep finding it unusable: bits 11..15 count the touches to skip */
                    const uint32_t cool = (st >> 11) & 31u;
                    need = cool == 0;
                    if (cool && jl == 0) k.bs[idt[pass]] = st - (1u << 11);
                } else need = lbin == bin ? false : (eb.x > el.x || (eb.x == el.x && eb.y >= el.y));
            }
            needt[pass] = need;
        }
    }
    if (__builtin_amdgcn_ballot_w64(needt[0] || needt[1]) == 0) return;
    if (lane < 8) work[lane] = 0xffffffffu;
    wave_lds_sync();
    bool any = false;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int id0 = idt[pass];
        const bool need = needt[pass];
        const uint32_t level = levt[pass];
        if (__builtin_amdgcn_ballot_w64(need) == 0) continue;
        any = true;
        const int id = need ? id0 : 0;
        const int v0 = band_lo(g, id), n1 = band_hi(g, id) - v0;     /* last index */
        uint32_t hm = 0;
        for (int t = 0; t < nc; t++) hm = max(hm, H[(v0 + min(jl + 16 * t, n1)) & 255].x);
        const uint32_t Hmax = rowmax_u32(hm);
        uint32_t km = 0;
        for (int t = 0; t < nc; t++) {
            const int j = min(jl + 16 * t, n1);
            const u32x2 e = H[(v0 + j) & 255];
            km = max(km, e.x == Hmax ? (e.y >> 1) + (uint32_t)(255 - j) + 1u : 0u);    /* rank<<8 | 255-j, +1 */
        }
        const uint32_t K = rowmax_u32(km) - 1u;
        const int jL = 255 - (int)(K & 255u);
        const uint32_t rL = K >> 8;
        uint32_t dup = 0;
        for (int t = 0; t < nc; t++) {
            const int j = jl + 16 * t;
            if (j <= n1 && j != jL) {
                const u32x2 e = H[(v0 + j) & 255];
                dup |= (e.x == Hmax && (e.y >> 9) == rL) ? 1u : 0u;
            }
        }
        dup = rowmax_u32(dup);
        if (need && jl == 0) {
            /* usable = ok; a band that stays unusable backs off: the next 2^level - 1 touches do not rescan it */
            const uint32_t lv = dup ? min(level + 1u, 5u) : 0u;
            k.bs[id] = (uint32_t)(v0 + jL + 256) | (dup ? 0u : 1536u) | (lv << 16) | (((1u << lv) - 1u) << 11);
            work[2 * c + pass] = (uint32_t)id;
        }
        k.rebuilds += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(need && jl == 0));
    }
    if (!any) return;
    wave_lds_sync();
    /* the table entries of the rescanned bands, then the watched relations (leaders may have moved) */
#pragma unroll 1
    for (int j = 0; j < 8; j++) {
        const int id = (int)work[j];
        if (id >= 0) lead_write_band_entries(k, g, lane, id);
    }
    wave_lds_sync();
    lead_collect_relations(k, g, lane, xnow);
}

/* Result record of a pixel and channel: { 8*byte | 8*v << 16, table address of its lookup = 8*filt + TB (+ 4 KB for filter none's
 * N table) }.  The Sierra difference filt - v: 12 bits of the address difference, which also drops the 4 KB. */
__device__ __forceinline__ int lead_rec_diff(const u32x2 r, const int TB)
{
    return __builtin_amdgcn_sbfe((int)r.y - pl_sext16((int)(r.x >> 16)) - TB, 0, 12) >> 3;
}

/* per-lane state of the speculative fast path (only lanes 0,16,32,48 -- one per channel -- run it) */
struct LeadState {
    uint32_t e0;     /* previous pixel's entry word 0: 8*v (low 16, signed) | 8*rem << 16 */
    int h1;          /* 8*thr of the previous pixel  (entry word 1) */
    int h2;          /* 8*thr of the pixel before it               */
    int lo8;         /* 8*lo of the previous pixel: byte*8 = 8*v - lo8 */
    int addr;        /* table address the previous pixel looked up (8*filt + TB) */
    uint32_t mul;    /* 1, or 4096 when the previous pixel's channel was a forced transparent alpha */
    int bad;         /* OR of the checked 8*byte values of the group */
};

#include "pl_lead_asm.h"

/* the exact evaluation of ONE pixel, all channels at once in 16-lane rows: gather, two DPP arg-max reductions and
 * the sequential repair of the histogram coupling between the channels -- the round-1 pixel step (chain_row) with
 * the generic candidate loop.  Bumps the histogram.  o,a,d,ex,ey: the pixel's raw words (uniform).
 * Hout/Rout: the winner's frequency BEFORE its bump and its rank (for the rescan test). */
template <int MODE>
__device__ __forceinline__ void lead_exact_pixel(const LeadCtx &k, int lane, uint32_t o, uint32_t a, uint32_t d, uint32_t ex, uint32_t ey,
                                                 int left, int rem, int thr_prev, int &back_out, int &diff_out, int &bin_out,
                                                 uint32_t &Hout, uint32_t &Rout)
{
    const int c = lane >> 4, jl = lane & 15;
    const uint32_t bpp = k.bpp;
    const bool active = (uint32_t)c < bpp;
    const int s = k.s, q = s + 1, nc = (q + 15) >> 4;
    lds_uint2 *const T = k.tbl;
    const int p = pl_plane_of_channel(bpp, c);
    const int e0 = pl_sext16((int)(p < 2 ? (ex >> (16 * p)) : (ey >> (16 * (p - 2)))));
    const int orig = (o >> (8 * c)) & 255, above = (a >> (8 * c)) & 255, diag = (d >> (8 * c)) & 255;
    const bool tr = (bpp & 1u) == 0u && ((o >> (8u * (bpp - 1u))) & 255u) == 0u && (uint32_t)c == bpp - 1u;
    const int predraw = pl_predict<MODE>(above, diag, left);
    const int osym = pl_sext8(orig - predraw);
    int lo = osym - orig;
    const int filt = osym + e0 + rem + thr_prev;
    const int tq = (int)((float)filt * k.rq);
    int vmin = tq * q - ((filt >> 31) & s);
    int vmax = vmin + s;
    const int hi = lo + 255;
    vmin = med3_i32(vmin, lo, hi);
    vmax = med3_i32(vmax, lo, hi);
    if (tr) { vmin = -predraw; vmax = -predraw; lo = -predraw; }
    const int span = vmax - vmin, josym = osym - vmin;
    uint32_t Hwin, K;
    if (nc <= 4) {
        /* up to four candidates per lane (q <= 64): every gather is issued before the first wait */
        u32x2 e[4]; int jj[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            jj[t] = min(jl + 16 * min(t, nc - 1), span);
            e[t] = T[(vmin + jj[t]) & 255];
        }
        uint32_t hm = max(max(e[0].x, e[1].x), max(e[2].x, e[3].x));
        Hwin = rowmax_u32(hm);
        uint32_t km = 0;
#pragma unroll
        for (int t = 0; t < 4; t++)
            km = max(km, e[t].x == Hwin ? e[t].y + ((jj[t] == josym) ? 256u : 0u) + (uint32_t)(256 - jj[t]) : 0u);
        K = rowmax_u32(km);
    } else {
        uint32_t hm = 0;
        for (int t = 0; t < nc; t++) hm = max(hm, T[(vmin + min(jl + 16 * t, span)) & 255].x);
        Hwin = rowmax_u32(hm);
        uint32_t km = 0;
        for (int t = 0; t < nc; t++) {
            const int j2 = min(jl + 16 * t, span);
            const u32x2 e = T[(vmin + j2) & 255];
            km = max(km, e.x == Hwin ? e.y + ((j2 == josym) ? 256u : 0u) + (uint32_t)(256 - j2) : 0u);
        }
        K = rowmax_u32(km);
    }
    int jwin = (int)((0u - K) & 255u);
    int vwin = vmin + jwin;
    uint32_t Rwin = (K - 1u) >> 9;
#pragma unroll
    for (int cp = 0; cp < 3; cp++) {
        if ((uint32_t)cp + 1u < bpp) {
            const int sb = __builtin_amdgcn_readlane(vwin, 16 * cp);
            const uint32_t sH = (uint32_t)__builtin_amdgcn_readlane((int)Hwin, 16 * cp) + 1u;
            const uint32_t sR = (uint32_t)__builtin_amdgcn_readlane((int)Rwin, 16 * cp);
            const int jj2 = (sb - vmin) & 255;
            const uint32_t K2 = (sR << 9) + ((jj2 == josym) ? 256u : 0u) + (uint32_t)(256 - jj2);
            const bool better = (c > cp) & (jj2 <= span) & ((sH > Hwin) | ((sH == Hwin) & (K2 > K)));
            Hwin = better ? sH : Hwin;
            K = better ? K2 : K;
            jwin = better ? jj2 : jwin;
            Rwin = better ? sR : Rwin;
            vwin = vmin + jwin;
        }
    }
    if (active && jl == 0) __hip_atomic_fetch_add((lds_u32 *)&T[vwin & 255], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    back_out = vwin - lo;
    diff_out = tr ? 0 : filt - vwin;
    bin_out = vwin & 255;
    Hout = Hwin;
    Rout = Rwin;
}

/* The speculative run of one channel lane over the pixels [pos, end) of the chunk.  The chain state in front of pixel
 * pos is re-derived from the result records of pixels pos-1 and pos-2 (ring slots, so this also works across chunks and
 * behind an exactly redone pixel).  Writes the result record {8*byte | 8*v << 16, table address} of every pixel it passes and returns
 * how many pixels of the chunk now have one; `bad` tells that some byte it produced lies outside 0..255 (= a table
 * entry was unusable or a leader clamped away; everything behind the first such pixel is garbage, and memory-safe).
 * The hand-scheduled loop runs whole groups of four, up to 4 pixels past `end` (neutral records there), and notices a
 * bad byte with a lag of up to 17 pixels.  The histogram is NOT touched here: the caller applies the bumps 64 pixels
 * at a time. */
template <int MODE, bool TRX>
__device__ __forceinline__ int lead_fast_run(lds_uint4 *R, lds_uint2 *OUT, lds_u32 *LUT, const int c, const int TB,
                                             const int pos, const int end, const int burst0, bool &bad)
{
    constexpr int RW = MODE == 4 ? 2 : 1;
    LeadState t;
    {
        const u32x2 r1 = OUT[(pos + 1) * 4 + c], r2 = OUT[pos * 4 + c];
        const uint32_t le1 = LUT[(lead_rec_diff(r1, TB) + 256) & 511];
        const uint32_t le2 = LUT[(lead_rec_diff(r2, TB) + 256) & 511];
        /* the first step writes the record of pixel pos-1 once more: give it back exactly what it holds (that pixel's bump
         * may still be deferred, and its bin rides in the record's upper half) */
        const int v8p0 = pl_sext16((int)(r1.x >> 16));
        t.e0 = (r1.x >> 16) | ((uint32_t)(pl_sext16((int)le1) * 8) << 16);
        t.h1 = ((int)le1 >> 16) * 8;
        t.h2 = ((int)le2 >> 16) * 8;
        t.lo8 = v8p0 - (int)(r1.x & 0xffffu);
        t.addr = (int)r1.y;
        t.mul = 1u;
        t.bad = 0;
    }
    bad = false;
    u32x4 ra0 = R[(pos * 4 + c) * RW], ra1 = RW == 2 ? R[(pos * 4 + c) * RW + 1] : ra0;
    u32x4 rb0 = R[((pos + 1) * 4 + c) * RW], rb1 = RW == 2 ? R[((pos + 1) * 4 + c) * RW + 1] : rb0;
    if (!TRX) {
        /* the hand-scheduled loop (pl_lead_asm.h) */
        const int iters0 = (end - pos + 4) >> 2;              /* ceil((end - pos + 1) / 4): the step of pixel `end` writes the record of end-1 */
        int iters = iters0;
        const uint32_t rptr = (uint32_t)(uintptr_t)&R[(pos * 4 + c) * RW], optr = (uint32_t)(uintptr_t)&OUT[(pos + 1) * 4 + c];
        uint32_t acc;
        if (MODE == 0 || MODE == 2) acc = lead_asm_noneup(t, rptr, optr, iters, burst0, ra0, rb0);
        else if (MODE == 1) acc = lead_asm_sub(t, rptr, optr, iters, burst0, ra0, rb0);
        else if (MODE == 3) acc = lead_asm_avg(t, rptr, optr, iters, burst0, ra0, rb0);
        else acc = lead_asm_paeth(t, rptr, optr, iters, burst0, ra0, rb0, ra1, rb1);
        bad = __builtin_amdgcn_ballot_w64(acc > 2047u) != 0;
        return min(end, pos + 4 * (iters0 - iters) - 1);
    }
    /* chunks that hold a fully transparent pixel: plain C++ steps */
    u32x4 rc0 = rb0, rc1 = rb1;
    /* look up pixel i (record r0/r1), write the record of pixel i-1, fetch record i+2 into rn0/rn1; true = pixel i-1 is bad */
    auto step = [&](const int i, const u32x4 r0, const u32x4 r1, u32x4 &rn0, u32x4 &rn1) -> bool {
        const int v8p = pl_sext16((int)t.e0), rem8p = (int)t.e0 >> 16;
        int back8p = v8p - t.lo8;
        back8p = (int)__umul24((uint32_t)back8p, t.mul);   /* forced symbol: any mismatch -> out of range */
        int addr, lo8; uint32_t trf = 0;
        if (MODE == 0 || MODE == 2) {
            addr = (int)r0.x + t.h2 + rem8p;
            lo8 = (int)r0.y;
            trf = r0.w; addr = trf ? (int)r0.z : addr; lo8 = trf ? (int)r0.z - TB : lo8;
        } else if (MODE == 1) {
            const int osym8 = __builtin_amdgcn_sbfe((int)r0.x - back8p, 0, 11);
            addr = osym8 + ((int)r0.y + t.h2 + rem8p);
            lo8 = osym8 - (int)r0.x;
            trf = r0.z; const int f8 = __builtin_amdgcn_sbfe(-back8p, 0, 11); addr = trf ? f8 + TB : addr; lo8 = trf ? f8 : lo8;
        } else if (MODE == 3) {
            const int pred = (int)__builtin_amdgcn_ubfe((uint32_t)(back8p + (int)r0.z), 4, 8);
            const int osym = __builtin_amdgcn_sbfe((int)r0.x - pred, 0, 8);
            addr = (osym << 3) + ((int)r0.y + t.h2 + rem8p);
            lo8 = (osym - (int)r0.x) << 3;
            trf = r0.w; const int f8 = pl_sext8(-pred) << 3; addr = trf ? f8 + TB : addr; lo8 = trf ? f8 : lo8;
        } else {
            /* Paeth without compares: key = distance << 14 | priority << 12 | (8*(orig - candidate) + 2048); the minimum
             * key is the predictor the reference picks (left, then above, then upper-left on ties, optimize_state.c:600-613)
             * and its low 11 bits are 8*osym already */
            const uint32_t b14 = ((uint32_t)back8p << 14) + PL_PAETH_BIAS;
            const uint32_t kA = sad_u32(b14, r0.x) + r1.x, kD = sad_u32(b14, r0.y) + r1.y, kL = r0.w - (uint32_t)back8p;
            const uint32_t m3 = min(min(kL, kA), kD);
            const int orig8 = (int)(r1.z & 0x7fffffffu);
            const int osym8 = __builtin_amdgcn_sbfe((int)m3, 0, 11);
            addr = osym8 + ((int)r1.w + t.h2 + rem8p);
            lo8 = osym8 - orig8;
            trf = r1.z >> 31; const int f8 = __builtin_amdgcn_sbfe(osym8 - orig8, 0, 11); addr = trf ? f8 + TB : addr; lo8 = trf ? f8 : lo8;
        }
        const u32x2 en = *(lds_uint2 *)(uintptr_t)(uint32_t)addr;
        OUT[(i + 1) * 4 + c] = (u32x2){ min((uint32_t)back8p, 0xffffu) | (t.e0 << 16), (uint32_t)t.addr };
        rn0 = R[((i + 2) * 4 + c) * RW];
        if (RW == 2) rn1 = R[((i + 2) * 4 + c) * RW + 1];
        const bool b = __builtin_amdgcn_ballot_w64((uint32_t)back8p > 2047u) != 0;
        t.h2 = t.h1; t.lo8 = lo8; t.addr = addr;
        t.mul = trf ? 4096u : 1u;
        t.h1 = (int)en.y; t.e0 = en.x;
        return b;
    };
    /* (validity tested per pixel here: chunks with transparent pixels are dense in forced symbols that do not lead their band, and
     * what a burst does behind one is wasted -- bursts of 8 were 13 % slower on the checkerboard frame) */
    for (int i = pos; i <= end; i++) {                        /* the step of pixel `end` (a neutral record) writes the record of end-1 */
        if (__builtin_expect(step(i, ra0, ra1, rc0, rc1), 0)) { bad = true; return i; }
        ra0 = rb0; ra1 = rb1; rb0 = rc0; rb1 = rc1;
    }
    return end;
}

/* ---------------------------------------------------------------------------------------------------------
 * One row of one candidate filter, band-leader formulation.  MODE = filter (0 none, 1 sub, 2 up, 3 average, 4 paeth).
 * Chain records (written by the vector pre-phase, lane = pixel; everything scaled by 8, TB = byte address of T[256]):
 *   none/up : { 8*osym + 8*e0 + TB, 8*lo, forced address (TB + 8*sext8(-pred)), tr }
 *   sub     : { 8*orig, 8*e0 + TB, tr, - }
 *   average : { orig, 8*e0 + TB, 8*above, -8*orig (tr in chunks that hold a transparent pixel) }
 *   paeth   : { (8*diag << 14) + BIAS, (8*(2*diag-above) << 14) + BIAS, -, (8*|above-diag| << 14) + 8*orig + 2048 }
 *             { (1<<12) + 8*(orig-above) + 2048, (2<<12) + 8*(orig-diag) + 2048, 8*orig | tr<<31, 8*e0 + TB }
 * --------------------------------------------------------------------------------------------------------- */
template <int MODE, bool TR>
__device__ __forceinline__ void chain_lead(LeadCtx &kref, const int lane)
{
    LeadCtx k = kref;     /* a register copy: kref lives in the caller's frame, and every field is read many times */
    const int c = lane >> 4, jl = lane & 15;
    const uint32_t bpp = (uint32_t)__builtin_amdgcn_readfirstlane((int)k.bpp);
    const uint32_t W = (uint32_t)__builtin_amdgcn_readfirstlane((int)k.W);
    glb_cu32 *const row = (glb_cu32 *)k.row, *const nabove = (glb_cu32 *)k.nabove;
    glb_cu32x2 *const err0 = (glb_cu32x2 *)k.err0;
    const bool active = (uint32_t)c < bpp;
    lds_uint4 *const R = k.crec;
    lds_uint2 *const OUT = k.out;
    lds_u32 *const LUT = k.lut;
    const int TB = (int)(uint32_t)(uintptr_t)(k.T + 256);   /* filter none: of the P table; the N table sits 4 KB behind */
    const LeadGeo geo = lead_geo(k.s, MODE == 0, k.rq);
    const bool chainlane = active && jl == 0;

    /* result ring slots 0,1 = the two pixels before the chunk: byte 0, diff 0 */
    if (lane < 8) OUT[lane] = (u32x2){ 0u, (uint32_t)TB };
    uint32_t slow = 0, light = 0;
    unsigned long long cyc_vec = 0, cyc_fast = 0, cyc_exact = 0, cyc_rescan = 0, cyc_clean = 0, cyc_flush = 0;
    uint32_t px_clean = 0;

    /* prefetch of the first chunk's raw words (lane = pixel) */
    uint32_t no = 0, na = 0, nd = 0; u32x2 ne = (u32x2){ 0u, 0u };
    {
        const uint32_t xl = lane;
        if (xl < W) {
            no = row[xl];
            if (nabove) { na = nabove[xl]; nd = xl ? nabove[xl - 1] : 0u; }
            ne = err0[xl];
        }
    }
#if PL_PREFETCH_FIX
    /* (the first chunk's words are complete before the loop: otherwise the compiler, which sees them used inside the serial
     * loop, flushes vmcnt in front of that loop in EVERY chunk -- right behind the prefetch loads it is supposed to overlap) */
    __builtin_amdgcn_s_waitcnt(0x0F70);   /* vmcnt(0) */
#endif
    u32x4 pend = (u32x4){ 0u, 0u, 0u, 0u }; uint32_t pend_x = 0; int pend_n = 0;
    for (uint32_t x0 = 0; x0 < W; x0 += PL_LCHUNK) {
        const int n = (int)min((uint32_t)PL_LCHUNK, W - x0);
        const uint32_t o = no, a = na, d = nd; const u32x2 e = ne;
        {   /* issue the next chunk's loads now; they land while the serial part runs */
            const uint32_t xl = x0 + PL_LCHUNK + lane;
            no = 0; na = 0; nd = 0; ne = (u32x2){ 0u, 0u };
            if (xl < W) {
                no = row[xl];
                if (nabove) { na = nabove[xl]; nd = nabove[xl - 1]; }
                ne = err0[xl];
            }
        }
        /* the previous chunk's candidate words go out BEHIND the prefetch loads: a store issued at the end of the chunk would sit
         * in front of them in the vmcnt queue, and the wait for the prefetched words at the next chunk's start would wait for
         * the store's acknowledgement too */
        if (pend_n && lane < pend_n) ((__attribute__((address_space(1))) u32x4 *)k.cand)[pend_x + lane] = pend;
        const unsigned long long tv0 = LTIME();
        /* ---- vector pre-phase: lane = pixel ---- */
        const bool alpha0 = TR && lane < n && ((o >> (8u * (bpp - 1u))) & 255u) == 0u;
        const bool chunk_tr = TR && __builtin_amdgcn_ballot_w64(alpha0) != 0;
        auto write_records = [&](const int idx, const uint32_t po, const uint32_t pa, const uint32_t pd, const u32x2 pe, const bool palpha0) {
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
            const int p = pl_plane_of_channel(bpp, cc);
            const int e0 = pl_sext16((int)(p < 2 ? (pe.x >> (16 * p)) : (pe.y >> (16 * (p - 2)))));
            const int orig = (po >> (8 * cc)) & 255, above = (pa >> (8 * cc)) & 255, diag = (pd >> (8 * cc)) & 255;
            const uint32_t trf = (palpha0 && (uint32_t)cc == bpp - 1u) ? 1u : 0u;
            const int e0tb = e0 * 8 + TB + ((MODE == 0 && orig >= 128) ? PL_LT_N * 8 : 0);
            if (MODE == 0 || MODE == 2) {
                const int pred = MODE == 2 ? above : 0;
                const int osym = pl_sext8(orig - pred);
                R[idx * 4 + cc] = (u32x4){ (uint32_t)(osym * 8 + e0tb), (uint32_t)((osym - orig) * 8),
                                           (uint32_t)(pl_sext8(-pred) * 8 + TB), trf };
            } else if (MODE == 1) {
                R[idx * 4 + cc] = (u32x4){ (uint32_t)(orig * 8), (uint32_t)e0tb, trf, 0u };
            } else if (MODE == 3) {
                R[idx * 4 + cc] = (u32x4){ (uint32_t)orig, (uint32_t)e0tb, (uint32_t)(above * 8), chunk_tr ? trf : (uint32_t)(-8 * orig) };
            } else {
                /* distances in key position already: |8*left - 8*diag| << 14 = |(8*left << 14) + BIAS - x|, the same for
                 * |8*left + 8*above - 16*diag| with y (BIAS keeps both operands of the unsigned v_sad_u32 positive) */
                R[(idx * 4 + cc) * 2] = (u32x4){ ((uint32_t)(diag * 8) << 14) + PL_PAETH_BIAS, ((uint32_t)((2 * diag - above) * 8) << 14) + PL_PAETH_BIAS, 0u,
                                                 ((uint32_t)(abs(above - diag) * 8) << 14) + (uint32_t)(orig * 8 + 2048) };
                R[(idx * 4 + cc) * 2 + 1] = (u32x4){ (1u << 12) + (uint32_t)((orig - above) * 8 + 2048), (2u << 12) + (uint32_t)((orig - diag) * 8 + 2048),
                                                     (uint32_t)(orig * 8) | (trf << 31), (uint32_t)e0tb };
            }
        }
        };
        write_records(lane, o, a, d, e, alpha0);
        /* the loop overshoots the chunk by up to 4 pixels and fetches 2 more: neutral records (a black pixel, no error) */
        if (lane < PL_LREC_N - PL_LCHUNK) write_records(PL_LCHUNK + lane, 0u, 0u, 0u, (u32x2){ 0u, 0u }, false);
        wave_lds_sync();
        cyc_vec += LTIME() - tv0;

        /* ---- serial part ---- */
        int pos = 0, flushed = 0;
        unsigned long long fmask = 0;   /* light pixels of the chunk whose bumps are still deferred */
        /* The deferred histogram bumps of the pixels [from, to) of the chunk, lane = pixel -- after checking them against the
         * watched relations: the first pixel whose bumps would let a relation's bin u catch up with its l (counting, in pixel
         * order, u's bumps up to and including that pixel against l's before it) must not be applied: it and everything behind
         * it is void, and it is redone exactly (which rescans the band through the ordinary slow path).  Returns that pixel's
         * index, or `to`; the bumps in front of it are applied. */
        const uint32_t chmask = (1u << bpp) - 1u;
        auto flush_verify = [&](int from, int to) -> int {
            const bool mine = lane >= from && lane < to;
            /* every LDS read below is unconditional (lanes and channels that do not take part are masked afterwards): the reads
             * of one stage are in flight together instead of one round trip each.  The bin of every bump rides in the upper
             * half of the result record's first word: (8v >> 3) & 255 */
            uint32_t bin4[4];
#pragma unroll
            for (int cc = 0; cc < 4; cc++) bin4[cc] = (OUT[(lane + 2) * 4 + cc].x >> 19) & 255u;
            const int nrel = __builtin_amdgcn_readfirstlane((int)k.work[41]);
            const int markx = __builtin_amdgcn_readfirstlane((int)k.work[42]);
            int kv = to;
            /* no bump of the range goes to a bin marked close (and the marking is still good for this range) and no light pixel
             * is in it: nothing to check -- one decision for the common case */
            if (nrel && to > from && 4u * (uint32_t)((int)x0 + to - markx) >= PL_LREL_CLOSE) lead_mark_close(k, lane, (int)x0 + from);
            uint32_t hitbits = 0;
#pragma unroll
            for (int cc = 0; cc < 4; cc++) hitbits |= ((k.work[112 + (bin4[cc] >> 5)] >> (bin4[cc] & 31u)) & 1u) << cc;
            const uint32_t lightlane = (uint32_t)(fmask >> lane) & 1u;
            const bool special = __builtin_amdgcn_ballot_w64(mine && ((hitbits & chmask) | lightlane) != 0u) != 0;
            bool check = false;
            if (special) check = nrel && __builtin_amdgcn_ballot_w64(mine && (hitbits & chmask) != 0u) != 0;
            int sym[4];
#pragma unroll
            for (int cc = 0; cc < 4; cc++) sym[cc] = (mine && ((chmask >> cc) & 1u)) ? (int)bin4[cc] : -1;
            if (check) {
                /* quick reject, lane = relation: u is not bumped in this range at all (bitmap of the bumped bins, work[32..39]),
                 * or even if every bump of the range went to u it would stay below l */
                if (lane < 8) k.work[32 + lane] = 0u;
                wave_lds_sync();
                if (mine) {
#pragma unroll
                    for (int cc = 0; cc < 4; cc++)
                        if ((uint32_t)cc < bpp) __hip_atomic_fetch_or(&k.work[32 + (sym[cc] >> 5)], 1u << (sym[cc] & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                wave_lds_sync();
                bool close = false;
                uint32_t rel = 0;
                if (lane < nrel) {
                    rel = k.work[48 + lane];
                    const uint32_t ub = rel & 255u;
                    if ((k.work[32 + (ub >> 5)] >> (ub & 31u)) & 1u) {
                        const u32x2 eu = k.tbl[ub], el = k.tbl[(rel >> 8) & 255u];
                        close = !(eu.x + 4u * (uint32_t)(to - from) < el.x);
                    }
                }
                unsigned long long m = __builtin_amdgcn_ballot_w64(close);
                while (m) {
                    const int r = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(m));
                    m &= m - 1;
                    const uint32_t rr2 = (uint32_t)__builtin_amdgcn_readlane((int)rel, r);
                    const int u = (int)(rr2 & 255u), l = (int)((rr2 >> 8) & 255u);
                    const u32x2 eu = k.tbl[u], el = k.tbl[l];
                    uint32_t cu = 0, cl = 0;
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) {
                        const unsigned long long mu = __builtin_amdgcn_ballot_w64(sym[cc] == u), ml = __builtin_amdgcn_ballot_w64(sym[cc] == l);
                        cu += __builtin_amdgcn_mbcnt_hi((uint32_t)(mu >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mu, 0u)) + (sym[cc] == u ? 1u : 0u);
                        cl += __builtin_amdgcn_mbcnt_hi((uint32_t)(ml >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ml, 0u));
                    }
                    const uint32_t hu = eu.x + cu, hl = el.x + cl;
                    const bool viol = mine && (hu > hl || (hu == hl && eu.y >= el.y));
                    const unsigned long long mv = __builtin_amdgcn_ballot_w64(viol);
                    if (mv) kv = min(kv, (int)__builtin_ctzll(mv));
                }
            }
            if (special && __builtin_amdgcn_ballot_w64(mine && lightlane != 0u) != 0) {
                /* light pixels in the range: a bin they bump must stay strictly below the leader of every usable band that
                 * holds it (unless it is that leader: then the watched relations above cover it) -- decided on the safe side,
                 * as if every bump of the range went to it */
                bool viol = false;
#pragma unroll
                for (int cc = 0; cc < 4; cc++) {
#pragma unroll
                    for (int sg = 0; sg < 2; sg++) {
                        const int b = (int)bin4[cc];
                        const int id = band_of_bin(geo, b, sg == 1);
                        const uint32_t st = k.bs[max(id, 0)];
                        const int L = ((int)(st & 511u) - 256) & 255;
                        const uint32_t hb = k.tbl[b].x, hl = k.tbl[L].x;
                        viol |= ((chmask >> cc) & 1u) && id >= 0 && (st & 1024u) && L != b && !(hb + 4u * (uint32_t)(to - from) < hl);
                    }
                }
                const unsigned long long mv = __builtin_amdgcn_ballot_w64(mine && lightlane != 0u && viol);
                if (mv) kv = min(kv, (int)__builtin_ctzll(mv));
            }
            {
                /* (an add of 0 for what does not take part: no branches around the four adds) */
                const bool on = mine && lane < kv;
#pragma unroll
                for (int cc = 0; cc < 4; cc++)
                    __hip_atomic_fetch_add((lds_u32 *)&k.tbl[bin4[cc]], (on && ((chmask >> cc) & 1u)) ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            return kv;
        };
        for (;;) {
            const unsigned long long tf0 = LTIME();
            int limit = n, ixb = n;                            /* records up to `limit` are good; ixb: first bad pixel, if any */
            if (pos < n) {
                int cur = n; bool bad = false;
                const int burst0 = pos == 0 ? PL_LEAD_BURST0_CLEAN : PL_LEAD_BURST0_RESTART;   /* first burst of the run, in groups of 4 pixels */
                if (chainlane) {
                    if (TR && chunk_tr) cur = lead_fast_run<MODE, true>(R, OUT, LUT, c, TB, pos, n, burst0, bad);
                    else cur = lead_fast_run<MODE, false>(R, OUT, LUT, c, TB, pos, n, burst0, bad);
                }
                cur = __builtin_amdgcn_readfirstlane(cur);    /* lane 0 is channel 0's chain lane: always active */
                const bool anybad = __builtin_amdgcn_ballot_w64(bad) != 0;
                const unsigned long long tf1 = LTIME();
                cyc_fast += tf1 - tf0;
                if (pos == 0 && !anybad) { cyc_clean += tf1 - tf0; px_clean += (uint32_t)n; }   /* diagnostics: undisturbed whole-chunk runs */
                wave_lds_sync();
                if (__builtin_expect(anybad, 0)) {
                    /* the first pixel whose reconstruction left 0..255 is among the last records (a record trails its lookup by
                     * one pixel, so the window behind a 16-pixel burst is 17 wide: look at 32) */
                    const int w0 = cur - 32 + jl, w1 = cur - 16 + jl;
                    const bool f0 = active && w0 >= pos && ((OUT[(max(w0, 0) + 2) * 4 + c].x & 0xffffu) > 2047u);
                    const bool f1 = active && w1 >= pos && ((OUT[(max(w1, 0) + 2) * 4 + c].x & 0xffffu) > 2047u);
                    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(f0), m1 = __builtin_amdgcn_ballot_w64(f1);
                    const uint32_t a16 = (uint32_t)((m0 | (m0 >> 16) | (m0 >> 32) | (m0 >> 48)) & 0xffffull);
                    const uint32_t b16 = (uint32_t)((m1 | (m1 >> 16) | (m1 >> 32) | (m1 >> 48)) & 0xffffull);
                    ixb = a16 ? cur - 32 + (int)__builtin_ctz(a16) : (b16 ? cur - 16 + (int)__builtin_ctz(b16) : n);
                    /* (only the neutral pixels behind the chunk out of range: ixb = n, everything up to min(cur, n) is good) */
                    limit = min(ixb, min(cur, n));
                }
                pos = limit;
            }
            /* ---- slow section.  Entered at the first bad pixel of the run WITHOUT flushing, or behind a flush at the pixel
             * whose bumps would break a watched relation.  Every pixel is classified from the table as it stands (lane row =
             * channel):
             *   0  every channel's entry reconstructs a byte inside 0..255: hand back to the fast run
             *   1  LIGHT: the channels that fail have a band the clamp [lo, lo+255] cuts down to a single value -- the answer
             *      whatever the histogram says (saturated pixels, 70% of the slow ones).  Result record written here, bump
             *      deferred like a fast pixel's; its bins need not lead any band, so lead_flush holds them to the rule the
             *      watched relations enforce for leaders (fmask marks the pixel)
             *   2  anything else: the pending bumps are flushed and the pixel is evaluated exactly
             * (oracle: run_chain_lead's light / slow cases) */
            int ix; bool force_exact = false;
            const unsigned long long tf1b = LTIME();
            if (ixb < n && ixb == limit) ix = ixb;
            else {
                const int kv = flush_verify(flushed, limit);
                wave_lds_sync();
                cyc_flush += LTIME() - tf1b;
                flushed = kv;
                if (kv >= limit) { if (limit >= n) break; continue; }
                ix = kv; force_exact = true;
                fmask &= ~(~0ull << kv);
            }
            const unsigned long long tf1 = LTIME();
            uint32_t le1, le2; int left;
            auto derive = [&](const int at) {
                /* chain state in front of pixel `at`, from the results of at-1 and at-2 */
                const u32x2 r1 = OUT[(at + 1) * 4 + c], r2 = OUT[(at + 0) * 4 + c];
                le1 = LUT[(lead_rec_diff(r1, TB) + 256) & 511];
                le2 = LUT[(lead_rec_diff(r2, TB) + 256) & 511];
                left = (int)(r1.x & 0xffffu) >> 3;
            };
            derive(ix);
            bool first = true;
            const uint32_t actbits = active ? ~0u : 0u;
            unsigned long long cyc_inner = 0;                    /* flush + rescan time inside the section (accounted separately) */
            for (;;) {
                const uint32_t po = (uint32_t)__builtin_amdgcn_readlane((int)o, ix), pa = (uint32_t)__builtin_amdgcn_readlane((int)a, ix),
                               pd = (uint32_t)__builtin_amdgcn_readlane((int)d, ix);
                const uint32_t pex = (uint32_t)__builtin_amdgcn_readlane((int)e.x, ix), pey = (uint32_t)__builtin_amdgcn_readlane((int)e.y, ix);
                int back, diff, bin;
                bool anyheavy, anylight;
                {
                    const int p = pl_plane_of_channel(bpp, c);
                    const int e0 = pl_sext16((int)(p < 2 ? (pex >> (16 * p)) : (pey >> (16 * (p - 2)))));
                    const int orig = (po >> (8 * c)) & 255, above = (pa >> (8 * c)) & 255, diag = (pd >> (8 * c)) & 255;
                    const int pred = pl_predict<MODE>(above, diag, left);
                    const int osym = pl_sext8(orig - pred), lo = osym - orig;
                    const int filt = osym + e0 + pl_sext16((int)le1) + ((int)le2 >> 16);
                    const int fcl = med3_i32(filt, -256, 255);
                    /* the alpha channel of a fully transparent pixel takes the forced symbol -pred (optimize_state.c:158-164): fine
                     * iff the band that holds it is usable and led by exactly this value (what the fast run tests, too) */
                    const bool trlane = TR && chunk_tr && (uint32_t)c == bpp - 1u && ((po >> (8u * (bpp - 1u))) & 255u) == 0u;
                    const int fsym = pl_sext8(-pred);
                    const u32x2 ent = k.T[(trlane ? fsym : fcl) + 256 + ((MODE == 0 && !trlane && orig >= 128) ? PL_LT_N : 0)];
                    const int v8 = pl_sext16((int)ent.x);
                    /* per-lane integers instead of lane masks (no VALU -> SALU -> VALU round trips in front of the one decision):
                     * notok != 0: no usable leader inside the clamp;  several != 0: the clamp leaves more than one value */
                    const uint32_t notok = (trlane ? (uint32_t)(v8 ^ (fsym * 8)) : ((uint32_t)(fcl ^ filt) | ((uint32_t)(v8 - lo * 8) >> 11))) & actbits;
                    const int q = k.s + 1, tq = (int)((float)filt * k.rq);
                    const int vmin = tq * q - ((filt >> 31) & k.s), hi = lo + 255;
                    const int cmin = med3_i32(vmin, lo, hi), cmax = med3_i32(vmin + k.s, lo, hi);
                    /* (a forced symbol is as independent of the histogram as a single value the clamp leaves) */
                    const uint32_t several = MODE != 0 ? (trlane ? 0u : (uint32_t)(cmin ^ cmax)) : 1u;
                    const int v = trlane ? fsym : (notok ? cmin : (v8 >> 3));
                    back = trlane ? 0 : v - lo; diff = trlane ? 0 : filt - v; bin = v & 255;
                    const uint32_t heavyv = notok ? several : 0u;
                    anylight = __builtin_amdgcn_ballot_w64(notok != 0u) != 0;
                    /* (first && !anylight cannot happen -- the run stopped here -- but never hand back without progress) */
                    anyheavy = __builtin_amdgcn_ballot_w64(heavyv != 0u) != 0 || force_exact || (first && !anylight);
                }
                first = false;
                if (!anyheavy) {
                    if (!anylight) break;                        /* class 0: back to the fast run */
                    fmask |= 1ull << ix;
                    light++;
                } else {
                    if (flushed < ix) {
                        const unsigned long long tfa = LTIME();
                        wave_lds_sync();
                        const int kv = flush_verify(flushed, ix);
                        wave_lds_sync();
                        const unsigned long long tfb = LTIME();
                        cyc_flush += tfb - tfa; cyc_inner += tfb - tfa;
                        flushed = kv;
                        if (kv < ix) {                          /* a relation breaks in front of this pixel: that one first */
                            ix = kv; force_exact = true;
                            fmask &= ~(~0ull << kv);
                            derive(ix);
                            continue;
                        }
                    }
                    uint32_t Hw, Rw; (void)Hw; (void)Rw;
                    lead_exact_pixel<MODE>(k, lane, po, pa, pd, pex, pey, left, pl_sext16((int)le1), (int)le2 >> 16, back, diff, bin, Hw, Rw);
                    slow++;
                    force_exact = false;
                }
                if (chainlane) OUT[(ix + 2) * 4 + c] = (u32x2){ (uint32_t)(back * 8) | ((uint32_t)(bin * 8) << 16), (uint32_t)((diff + bin) * 8 + TB) };
                const uint32_t le0 = LUT[(diff + 256) & 511];
                if (anyheavy) {
                    const unsigned long long te1 = LTIME();
                    wave_lds_sync();
                    lead_rescan(k, geo, lane, bin, active, k.work, x0 + ix);
                    flushed = ix + 1;
                    const unsigned long long te2 = LTIME();
                    cyc_rescan += te2 - te1; cyc_inner += te2 - te1;
                }
                left = back; le2 = le1; le1 = le0;
                ix++;
                if (ix >= n) break;
            }
            cyc_exact += LTIME() - tf1 - cyc_inner;
            wave_lds_sync();
            if (ix < 64) fmask &= ~(~0ull << ix);
            pos = ix;
        }
        wave_lds_sync();
        const unsigned long long tv1 = LTIME();
        /* ---- vector post-phase: candidate row (byte | diff16 << 8 per channel), lane = pixel ---- */
        {
            uint32_t w[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
            for (uint32_t cc = 0; cc < 4; cc++) {
                const u32x2 r = OUT[(lane + 2) * 4 + cc];
                if (cc < bpp) w[cc] = ((r.x >> 3) & 255u) | ((uint32_t)lead_rec_diff(r, TB) << 8);
            }
            pend = (u32x4){ w[0], w[1], w[2], w[3] };   /* stored at the top of the next chunk (or behind the loop) */
            pend_x = x0; pend_n = n;
#if !PL_PREFETCH_FIX
            if (lane < pend_n) ((__attribute__((address_space(1))) u32x4 *)k.cand)[pend_x + lane] = pend;
            pend_n = 0;
#endif
        }
        /* the chunk's last two results become slots 0,1 of the next chunk */
        u32x2 keep = (u32x2){ 0u, 0u };
        if (lane < 8) keep = OUT[n * 4 + lane];
        wave_lds_sync();
        if (lane < 8) OUT[lane] = keep;
        wave_lds_sync();
        cyc_vec += LTIME() - tv1;
    }
    if (pend_n && lane < pend_n) ((__attribute__((address_space(1))) u32x4 *)k.cand)[pend_x + lane] = pend;
    kref.slow = slow;
    kref.light = light;
    kref.rebuilds = k.rebuilds;
    kref.cyc[0] = cyc_vec + (cyc_flush << 0) * 0; kref.cyc[6] = cyc_flush; kref.cyc[1] = cyc_fast; kref.cyc[2] = cyc_exact; kref.cyc[3] = cyc_rescan;
    kref.cyc[4] = cyc_clean; kref.cyc[5] = px_clean;
}

template <int MODE>
__device__ __noinline__ void chain_lead_dispatch(LeadCtx &k, int lane)
{
    if ((k.bpp & 1u) == 0) chain_lead<MODE, true>(k, lane);
    else chain_lead<MODE, false>(k, lane);
}

/* ---------------------------------------------------------------------------------------------------------
 * Per-candidate post pass, parallel over x (lane = pixel): derivative error (optimize_state.c:265-287),
 * libpng's heuristic filter (optimize_state.c:492-562) and entropy cost (optimize_state.c:326-342).
 * Returns the row cost of optimize_state_row (optimize_state.c:360) or UINT64_MAX if rejected (:319-324).
 * --------------------------------------------------------------------------------------------------------- */
/* All five candidates at once, parallel over x with every thread of the workgroup: the neighbourhood of a pixel (original
 * and optimised rows) is loaded once for the five candidates; per-wave partial sums go to the LDS accumulators
 * acc[f] = { derr (u64 as two u32 adds: low, high), cost, hs[5] } (8 words per candidate, zeroed by the caller). */
/* (first, stride): the pixels this thread takes -- (tid, workgroup size) for the joint pass, (lane, 64) when one wave does a
 * candidate of its own while the other chains still run; fmask: the candidates to do */
__device__ void post_pass_all(const PlJob &j, uint32_t y, uint32_t bpp, const uint2 (*tbl)[PL_TBL_N], bool adaptive, int tid, uint32_t *acc,
                              uint32_t first, uint32_t stride, uint32_t fmask)
{
    const uint32_t W = j.width;
    const uint32_t *row = j.img + (size_t)y * W;
    const uint32_t *nab = y ? row - W : nullptr;
    uint64_t derr[PL_NFILT] = { 0, 0, 0, 0, 0 };
    uint32_t cost[PL_NFILT] = { 0, 0, 0, 0, 0 };
    uint32_t hs[PL_NFILT][PL_NFILT] = { { 0 } };
    for (uint32_t x = first; x < W; x += stride) {
        const uint32_t o = row[x], ol = x ? row[x - 1] : 0u;
        const uint32_t na = nab ? nab[x] : 0u, nd = (nab && x) ? nab[x - 1] : 0u;
        const uint32_t oa = y ? j.old_above[x] : 0u, od = (y && x) ? j.old_above[x - 1] : 0u;
#pragma unroll
        for (int f = 0; f < PL_NFILT; f++) {
            if (!((fmask >> f) & 1u)) continue;
            const uint4 cw = j.cand[(size_t)f * W + x];
            const uint4 cl = x ? j.cand[(size_t)f * W + x - 1] : make_uint4(0, 0, 0, 0);
            const uint32_t cws[4] = { cw.x, cw.y, cw.z, cw.w }, cls[4] = { cl.x, cl.y, cl.z, cl.w };
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if ((uint32_t)c < bpp) {
                    const int sh = 8 * c;
                    const int back = cws[c] & 255, nl = x ? (int)(cls[c] & 255) : 0;
                    const int ov = (o >> sh) & 255, olv = (ol >> sh) & 255;
                    const int nav = (na >> sh) & 255, ndv = (nd >> sh) & 255, oav = (oa >> sh) & 255, odv = (od >> sh) & 255;
                    const int da = (oav - ov) - (nav - back);
                    const int dd = (odv - ov) - (ndv - back);
                    const int dl = (olv - ov) - (nl - back);
                    const uint32_t w = (bpp <= 2 && c == 0) ? 3u : 1u;      /* gray is replicated into r,g,b (color_delta.c:11-26) */
                    derr[f] += (uint64_t)(w * (uint32_t)(da * da + dd * dd + dl * dl));
                    const int pred = f == 0 ? 0 : (f == 1 ? nl : (f == 2 ? nav : (f == 3 ? (nav + nl) >> 1 : pl_paeth(nav, ndv, nl))));
                    cost[f] += 33u + (uint32_t)__clz((int)tbl[f][(back - pred) & 255].x);
                    if (adaptive) {
                        const int preds[PL_NFILT] = { 0, nl, nav, (nav + nl) >> 1, pl_paeth(nav, ndv, nl) };
#pragma unroll
                        for (int g = 0; g < PL_NFILT; g++) {
                            const int bb = (back - preds[g]) & 255;
                            hs[f][g] += (uint32_t)(bb < 128 ? bb : 256 - bb);
                        }
                    }
                }
            }
        }
    }
    const int lane = tid & 63;
#pragma unroll
    for (int f = 0; f < PL_NFILT; f++) {
        if (!((fmask >> f) & 1u)) continue;
        const uint64_t d = wave_sum_u64(derr[f]);
        const uint32_t cs = wave_sum_u32(cost[f]);
        if (lane == 0) {
            atomicAdd((unsigned long long *)&acc[8 * f], (unsigned long long)d);
            atomicAdd(&acc[8 * f + 2], cs);
        }
        if (adaptive) {
#pragma unroll
            for (int g = 0; g < PL_NFILT; g++) {
                const uint32_t v = wave_sum_u32(hs[f][g]);
                if (lane == 0) atomicAdd(&acc[8 * f + 3 + g], v);
            }
        }
    }
}

/* next-row Sierra terms of pixel sx of the winner, error plane via channel ch (optimize_state.c:446-465) */
__device__ __forceinline__ PlSplit split_at(const uint4 *cd, long sx, uint32_t W, int ch, float rbleed, float r29)
{
    if (sx < 0 || sx >= (long)W) { PlSplit z = { 0.f, 0.f, 0.f, 0.f, 0.f }; return z; }
    const uint4 v = cd[sx];
    const uint32_t w = ch == 0 ? v.x : (ch == 1 ? v.y : (ch == 2 ? v.z : v.w));
    return pl_sierra_split(pl_sext16((int)(w >> 8)), rbleed, r29);
}

} // namespace

/* LDS layout of pl_engine (bytes) */
#define PL_SM_TBL 0
#define PL_SM_HC (PL_SM_TBL + PL_NFILT * PL_TBL_N * 8)
#define PL_SM_LUT (PL_SM_HC + PL_NSYM * 4)
#define PL_SM_LUT2 (PL_SM_LUT + 512 * 4)            /* [diff+256] -> twos | fours << 8 | five << 16 | threes << 24 (int8 each) */
#define PL_SM_COSTS (PL_SM_LUT2 + 512 * 4)
#define PL_SM_FLAGS (PL_SM_COSTS + 64 + PL_NFILT * 8 * 4)
#define PL_SM_UNION (PL_SM_FLAGS + 64)
#define PL_SM_LEGACY_BYTES (PL_CHUNK * 4 * (2 + 4) * 16)
#define PL_SM_L_BS ((PL_NFILT + 1) * PL_LT_N * 8)             /* six decision tables: filter none has two */
#define PL_SM_L_WORK (PL_SM_L_BS + PL_NFILT * 512 * 4)
#define PL_SM_L_REC (PL_SM_L_WORK + PL_NFILT * PL_LWORK_N * 4)
#define PL_SM_L_REC_WAVE (PL_LREC_N * 4 * 16)                  /* per chain; the paeth chain takes two */
#define PL_SM_L_OUT (PL_SM_L_REC + (PL_NFILT + 1) * PL_SM_L_REC_WAVE)
#define PL_SM_L_OUT_WAVE ((PL_LCHUNK + 2 + 8) * 4 * 8)
#define PL_SM_LEAD_BYTES (PL_SM_L_OUT + PL_NFILT * PL_SM_L_OUT_WAVE)
#define PL_SM_COMMIT_BYTES (PL_SM_LEAD_BYTES > PL_SM_LEGACY_BYTES ? PL_SM_LEAD_BYTES : PL_SM_LEGACY_BYTES)   /* what the commit pass may use of the chains' region */
#define PL_SM_TOTAL (4096 + PL_SM_UNION + (PL_SM_LEAD_BYTES > PL_SM_LEGACY_BYTES ? PL_SM_LEAD_BYTES : PL_SM_LEGACY_BYTES))

__global__ __launch_bounds__(PL_ENGINE_THREADS) void pl_engine(const PlJob *jobs, const uint32_t *sel, PlEngineParams prm)
{
    /* LDS carve-up (dynamic: the band-leader tables push the total past the 64 KB static limit; gfx950 has 160 KB) */
    extern __shared__ __align__(16) unsigned char smem_raw[];
    /* the histogram tables sit 4 KB aligned (the band-leader chain ORs the bin offset into the table address) */
    unsigned char *const smem = smem_raw + ((4096u - ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw & 4095u)) & 4095u);
    uint2 (*const tbl)[PL_TBL_N] = (uint2 (*)[PL_TBL_N])(smem + PL_SM_TBL);   /* {running symbol_frequency, rank(original_frequency)<<9} per candidate (+64 dummy slots) */
    uint32_t *const Hc = (uint32_t *)(smem + PL_SM_HC);                               /* committed symbol_frequency */
    uint32_t *const split_lut = (uint32_t *)(smem + PL_SM_LUT);                       /* [diff+256] -> rem | thr<<16 of the Sierra split, |diff| <= 255 */
    uint32_t *const split_lut2 = (uint32_t *)(smem + PL_SM_LUT2);                     /* the next-rows terms of the split, for the commit pass */
    unsigned long long *const costs = (unsigned long long *)(smem + PL_SM_COSTS);
    uint32_t *const pacc = (uint32_t *)(smem + PL_SM_COSTS + 64);                     /* post pass accumulators: 8 words per candidate */
    uint32_t &big_err = *(uint32_t *)(smem + PL_SM_FLAGS);                            /* some |incoming error| of the coming row exceeds 8000 (see WRAP) */
    uint32_t &big_lead = *(uint32_t *)(smem + PL_SM_FLAGS + 4);                       /* ... exceeds PL_E0_LEAD_MAX: the row takes the round-1 chain */
    uint32_t &uniq = *(uint32_t *)(smem + PL_SM_FLAGS + 8);
    uint32_t &simd_map = *(uint32_t *)(smem + PL_SM_FLAGS + 12);                      /* diagnostics */
    uint32_t &chains_done = *(uint32_t *)(smem + PL_SM_FLAGS + 20);                   /* chain waves that finished this row attempt */
    uint32_t &post_done = *(uint32_t *)(smem + PL_SM_FLAGS + 24);                     /* candidates whose post pass their own wave already did */
    uint32_t &rowcyc = *(uint32_t *)(smem + PL_SM_FLAGS + 16);                        /* cycles of the slowest chain wave of this row attempt */
    uint4 *const rec = (uint4 *)(smem + PL_SM_UNION);                                 /* round-1 chain: chunk records (wave 0 two filters, waves 1..4 one) */
    /* band-leader chain (same region): decision tables, band states, chain records, result rings */
    uint2 *const ltab = (uint2 *)(smem + PL_SM_UNION);
    uint32_t *const lbs = (uint32_t *)(smem + PL_SM_UNION + PL_SM_L_BS);
    unsigned char *const lrec = smem + PL_SM_UNION + PL_SM_L_REC;
    uint2 *const lout = (uint2 *)(smem + PL_SM_UNION + PL_SM_L_OUT);

    const PlJob j = jobs[sel ? sel[blockIdx.x] : blockIdx.x];   /* (sel: the images of a mixed batch that take this engine) */
    const uint32_t W = j.width, H = j.height;
    const uint32_t bpp = pl_job_bpp(j);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float r29 = 2.0f * __uint_as_float(__float_as_uint(1.0f / 9.0f) + 1u);

    for (int i = tid; i < PL_NSYM; i += PL_ENGINE_THREADS) Hc[i] = 0;
    for (int i = tid; i < PL_NFILT * PL_NSYM; i += PL_ENGINE_THREADS)
        tbl[i >> 8][i & 255] = make_uint2(0u, j.orig_rank[i] << 9);
    for (int i = tid; i < PL_NFILT * 64; i += PL_ENGINE_THREADS) tbl[i >> 6][PL_NSYM + (i & 63)] = make_uint2(0u, 0u);
    for (int i = tid; i < 512; i += PL_ENGINE_THREADS) {
        const PlSplit sp = pl_sierra_split(i - 256, prm.rbleed, r29);
        split_lut[i] = ((uint32_t)(int)sp.rem & 0xffffu) | ((uint32_t)(int)sp.h << 16);
        split_lut2[i] = ((uint32_t)(int)sp.t & 255u) | (((uint32_t)(int)sp.f & 255u) << 8) | (((uint32_t)(int)sp.v & 255u) << 16) | ((uint32_t)(int)sp.h << 24);
    }
    if (tid == 0) { big_err = 0; big_lead = 0; simd_map = 0; }
    __syncthreads();

    uint32_t retried = 0, slow_px = 0, light_px = 0, lead_rows = 0, lead_rebuilds = 0;
    /* Which chains a row takes where both could: the band-leader chains unless the rows they just did were slow (dense slow pixels:
     * noise with ties everywhere, transparency patterns) AND the round-1 chains, tried on one row, were faster.  Both are exact, so
     * the choice -- made from cycle counts -- never shows in the results.  est_*: cycles per pixel of the slowest chain wave of
     * the last row each kind ran (0 = not known); the loser is tried again every PL_ADAPT_REPROBE rows. */
    uint32_t est_lead = 0, est_legacy = 0, adapt_since = 0, adapt_legacy_rows = 0;
    bool use_legacy = false, adapt_prev_lead = true, adapt_was_legacy = false;
    uint32_t adapt_backoff = 2u, adapt_last_lead = ~0u;
    unsigned long long lead_cyc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long cyc_post = 0, cyc_commit = 0;      /* diagnostics */   /* diagnostics: vector | fast | exact | rescan | table build */
    unsigned long long chain_cycles = 0, segs[4] = { 0, 0, 0, 0 };
    int status = 0;
    for (uint32_t y = 0; y < H && !status; y++) {
        const bool adaptive = !j.row_filters || y == 0;   /* pngloss_image.c:210 */
        int s = prm.strength;
        int winner = -1;
        const bool wrap = big_err != 0 || prm.force_careful;
        for (;;) {
            /* every candidate starts from the committed histogram (optimize_state_copy, pngloss_image.c:240) */
            for (int i = tid; i < PL_NFILT * PL_NSYM; i += PL_ENGINE_THREADS) tbl[i >> 8][i & 255].x = Hc[i & 255];
            for (int i = tid; i < PL_NFILT * 8; i += PL_ENGINE_THREADS) pacc[i] = 0u;
            if (tid == 0) { rowcyc = 0u; chains_done = 0u; post_done = 0u; }
            __syncthreads();
            /* chain phase.  Band-leader chains (round 2): five waves, one per candidate filter -- their fast path has
             * no DPP, and plain VALU/LDS waves sharing a SIMD do not slow each other (profiles/r01_ubench_simd_sharing.txt).
             * Rows that do not meet its preconditions take the round-1 chains below. */
            const bool lead_ok = (prm.engine_mode & 15) != 1 && s + 1 <= 128 && !wrap && big_lead == 0;
            /* engine_mode 3 ("mix", a test hook): the two kinds of chains take turns every four rows, whatever the clocks say -- the
             * adaptive choice (mode 0) depends on measured cycles and would not reproduce a mismatch of one of them */
            if (lead_ok && (prm.engine_mode & 15) == 3) use_legacy = ((y >> 2) & 1u) != 0u;
            const bool lead = lead_ok && !(use_legacy && ((prm.engine_mode & 15) == 0 || (prm.engine_mode & 15) == 3));
            if ((prm.engine_mode & 15) == 3 && lead_ok && !lead) adapt_legacy_rows++;
            const int lead_f = wave == 0 ? 2 : (wave == 1 ? 1 : (wave == 2 ? 3 : (wave == 3 ? 4 : 0)));   /* up | sub | average | paeth | none */
            const bool paired = s + 1 <= 48;
            if (lead) {
              if (wave < PL_NFILT) {
                LeadCtx k;
                k.row = j.img + (size_t)y * W;
                k.nabove = y ? k.row - W : nullptr;
                k.err0 = j.err0;
                k.cand = j.cand + (size_t)lead_f * W;
                k.tbl = (lds_uint2 *)&tbl[lead_f][0];
                k.T = (lds_uint2 *)(ltab + (lead_f ? lead_f + 1 : 0) * PL_LT_N);   /* none: tables 0 (P pixels) and 1 (N pixels) */
                k.bs = (lds_u32 *)(lbs + lead_f * 512);
                k.work = (lds_u32 *)(smem + PL_SM_UNION + PL_SM_L_WORK) + lead_f * PL_LWORK_N;
                k.crec = (lds_uint4 *)(lrec + lead_f * PL_SM_L_REC_WAVE);   /* none, sub, up, average, paeth (two slots) */
                k.out = (lds_uint2 *)(lout + lead_f * (PL_SM_L_OUT_WAVE / 8));
                k.lut = (lds_u32 *)&split_lut[0];
                k.W = W; k.bpp = bpp; k.s = s; k.rq = recip_up(s + 1);
                k.slow = 0; k.rebuilds = 0; k.light = 0;
                const unsigned long long t0 = __builtin_readcyclecounter();
                lead_build_table(k, lead_geo(s, lead_f == 0, k.rq), lane);
                const unsigned long long tb1 = __builtin_readcyclecounter();
                switch (lead_f) {
                case 0: chain_lead_dispatch<0>(k, lane); break;
                case 1: chain_lead_dispatch<1>(k, lane); break;
                case 2: chain_lead_dispatch<2>(k, lane); break;
                case 3: chain_lead_dispatch<3>(k, lane); break;
                default: chain_lead_dispatch<4>(k, lane); break;
                }
                const unsigned long long dtc = __builtin_readcyclecounter() - t0;
                chain_cycles += dtc;
                if (lane == 0) atomicMax(&rowcyc, (uint32_t)min(dtc, 0xffffffffull));
#if PL_POST_EARLY
                {   /* the first chains to finish (usually up and none, well ahead of sub / average / paeth) do the post pass of their
                     * own candidate while the others still run: that much less for the joint pass behind the barrier */
                    uint32_t order = 0;
                    if (lane == 0) order = atomicAdd(&chains_done, 1u);
                    order = (uint32_t)__builtin_amdgcn_readfirstlane((int)order);
                    if (order < PL_POST_EARLY) {
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      /* this wave's candidate row is complete and visible to itself */
                        post_pass_all(j, y, bpp, tbl, adaptive, tid, pacc, (uint32_t)lane, 64u, 1u << lead_f);
                        if (lane == 0) atomicOr(&post_done, 1u << lead_f);
                    }
                }
#endif
                slow_px += k.slow;
                light_px += k.light;
                lead_rebuilds += k.rebuilds;
                lead_rows++;
                for (int qq = 0; qq < 4; qq++) lead_cyc[qq] += k.cyc[qq];
                lead_cyc[4] += tb1 - t0;
                lead_cyc[5] += k.cyc[4]; lead_cyc[6] += k.cyc[5]; lead_cyc[7] += k.cyc[6];
              }
            } else
            /* round-1 chains: four waves on the four SIMDs -- wave 0 runs the 'none' and 'up' chains side by side,
             * waves 1..3 run sub, average, paeth; wave 4 only takes part in the data-parallel passes */
            /* wide bands (s > 47) would need > 6 candidates per lane in the paired wave: there the five chains run as
             * five waves instead (wave 4 = 'up'), accepting that two of them share a SIMD */
            if (wave < PL_NFILT && (wave < 4 || !paired)) {
                RowCtx k;
                k.row = j.img + (size_t)y * W;
                k.nabove = y ? k.row - W : nullptr;
                k.err0 = j.err0;
                k.cand = j.cand;
                k.tbl = (lds_uint2 *)&tbl[0][0];
                k.rec = (lds_uint4 *)&rec[wave == 0 ? 0 : PL_CHUNK * 4 * (1 + wave)];
                k.lut = (lds_u32 *)&split_lut[0];
                k.W = W; k.y = y; k.bpp = bpp; k.s = s;
                k.rq = recip_up(s + 1); k.rbleed = prm.rbleed; k.r29 = r29;
                k.slow = 0;
                const unsigned long long t0 = __builtin_readcyclecounter();
                switch (wave) {
                case 0: if (paired) chain_dispatch<5>(k, lane, wrap); else chain_dispatch<0>(k, lane, wrap); break;
                case 1: chain_dispatch<1>(k, lane, wrap); break;
                case 2: chain_dispatch<3>(k, lane, wrap); break;
                case 3: chain_dispatch<4>(k, lane, wrap); break;
                default: chain_dispatch<2>(k, lane, wrap); break;
                }
                const unsigned long long dtc = __builtin_readcyclecounter() - t0;
                chain_cycles += dtc;
                if (lane == 0) atomicMax(&rowcyc, (uint32_t)min(dtc, 0xffffffffull));
                slow_px += k.slow;
                if (PL_SEGPROF) for (int q = 0; q < 4; q++) segs[q] += k.seg[q];
            }
            __syncthreads();   /* candidate rows (global, same CU) and histograms (LDS) complete and visible */
            if (lead_ok && (prm.engine_mode & 15) == 0) {
                const uint32_t pp = rowcyc / W + 1u;
                /* (band-leader rows: the faster of the last two, so that a single slow row -- they exist, 7x the average -- is not a reason to switch) */
                if (lead) { est_lead = adapt_prev_lead ? min(pp, adapt_last_lead) : pp; adapt_last_lead = pp; }
                else { est_legacy = (!adapt_prev_lead && est_legacy) ? (est_legacy + pp) >> 1 : pp; adapt_legacy_rows++; }

                adapt_prev_lead = lead;
                if (est_lead <= PL_ADAPT_SLOW) { use_legacy = false; adapt_since = 0; adapt_backoff = 2u; }
                else if (est_legacy == 0) use_legacy = true;                       /* first try of the round-1 chains */
                else {
                    /* The loser runs a row now and then.  While the round-1 chains win, the band-leader chains are tried again after
                     * 2, 4, 8 .. PL_ADAPT_REPROBE rows (a single slow row must not cost 32 rows of the slower engine; a slow region
                     * is probed rarely); the round-1 chains are retried every PL_ADAPT_REPROBE rows when close behind, else rarely. */
                    const bool legacy_better = 21u * est_legacy < 20u * est_lead;
                    if (lead && legacy_better && adapt_was_legacy) adapt_backoff = min(2u * adapt_backoff, PL_ADAPT_REPROBE);   /* a probe lost */
                    const uint32_t every = legacy_better ? adapt_backoff : (10u * est_legacy < 13u * est_lead ? PL_ADAPT_REPROBE : 256u);
                    if (++adapt_since >= every) { use_legacy = !legacy_better; adapt_since = 0; }
                    else use_legacy = legacy_better;
                }
                adapt_was_legacy = !lead;
            }
            const unsigned long long tpp0 = __builtin_readcyclecounter();
            /* post pass: all threads, all candidates (the accumulators were zeroed before the chain phase) */
            post_pass_all(j, y, bpp, tbl, adaptive, tid, pacc, (uint32_t)tid, PL_ENGINE_THREADS, 31u & ~post_done);
            cyc_post += __builtin_readcyclecounter() - tpp0;
            __syncthreads();
            if (tid < PL_NFILT) {
                const uint32_t *a8 = pacc + 8 * tid;
                uint64_t cst = (((uint64_t)a8[1] << 32) | a8[0]) / 128u + a8[2];
                if (adaptive) {
                    int bestg = 0;
#pragma unroll
                    for (int g = 1; g < PL_NFILT; g++) if (a8[3 + g] < a8[3 + bestg]) bestg = g;
                    if (bestg != tid) cst = ~0ull;           /* libpng's heuristic would not have picked this filter (optimize_state.c:319-324) */
                }
                costs[tid] = (prm.engine_mode >> 8) ? (tid == (prm.engine_mode >> 8) - 1 ? 0ull : ~0ull) : cst;   /* debugging aid: force one candidate */
            }
            __syncthreads();
            uint64_t best = ~0ull;
#pragma unroll
            for (int f = 0; f < PL_NFILT; f++) {
                const uint64_t cf = costs[f];
                if (cf < best) { best = cf; winner = f; }     /* strict <: lowest filter index wins ties (pngloss_image.c:257) */
            }
            __syncthreads();
            if (winner >= 0) break;
            if (s == 0) { status = 65; break; }                /* pngloss_image.c:268-271 aborts here */
            s--;                                               /* pngloss_image.c:274 */
            retried += (s == prm.strength - 1);
        }
        if (status) break;

        /* ---- commit (pngloss_image.c:277-308), parallel over x ---- */
        const unsigned long long tcm0 = __builtin_readcyclecounter();
        if (tid == 0) { big_err = 0; big_lead = 0; }
        __syncthreads();
        bool big = false, bigl = false;
        const uint4 *__restrict__ cd = j.cand + (size_t)winner * W;
        uint32_t *__restrict__ rowp = j.img + (size_t)y * W;
        uint32_t *__restrict__ oldab = j.old_above;
        uint2 *__restrict__ perr0 = j.err0, *__restrict__ perr1 = j.err1;
        const uint32_t keep = bpp >= 4 ? 0xffffffffu : ((1u << (8 * bpp)) - 1u);
        /* next-rows Sierra terms of one pixel and channel (optimize_state.c:446-465): t | f << 8 | v << 16 | h << 24 (int8 each) from the
         * 512-entry LDS table of the split for |diff| <= 255, else by the float arithmetic of pl_sierra_split */
        auto terms_of = [&](const uint4 v, const int ch) -> uint32_t {
            const uint32_t w = ch == 0 ? v.x : (ch == 1 ? v.y : (ch == 2 ? v.z : v.w));
            const int diff = pl_sext16((int)(w >> 8));
            if (diff >= -256 && diff <= 255) return split_lut2[diff + 256];
            const PlSplit sp = pl_sierra_split(diff, prm.rbleed, r29);
            return ((uint32_t)(int)sp.t & 255u) | (((uint32_t)(int)sp.f & 255u) << 8) | (((uint32_t)(int)sp.v & 255u) << 16) | ((uint32_t)(int)sp.h << 24);
        };
        const auto T_ = [](uint32_t e) { return __builtin_amdgcn_sbfe((int)e, 0, 8); };
        const auto F_ = [](uint32_t e) { return __builtin_amdgcn_sbfe((int)e, 8, 8); };
        const auto V_ = [](uint32_t e) { return __builtin_amdgcn_sbfe((int)e, 16, 8); };
        const auto H_ = [](uint32_t e) { return (int)e >> 24; };
        if ((size_t)W * 16u <= (size_t)PL_SM_COMMIT_BYTES) {
            /* Two passes over the row through LDS (the chains' region is free now): every pixel's terms are looked up ONCE -- four
             * words, one per error plane -- and its four neighbours read them from there, instead of five pixels x four planes of
             * lookups per pixel. */
            uint4 *const tterms = (uint4 *)(smem + PL_SM_UNION);
#pragma unroll 4
            for (uint32_t x = tid; x < W; x += PL_ENGINE_THREADS) {
                const uint4 cw = cd[x];
                const uint32_t np = ((cw.x & 255u) | ((cw.y & 255u) << 8) | ((cw.z & 255u) << 16) | ((cw.w & 255u) << 24)) & keep;
                oldab[x] = rowp[x];
                rowp[x] = np;
                uint32_t tw[4];
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const int ch = pl_channel_of_plane(bpp, p);
                    tw[p] = ch >= 0 ? terms_of(cw, ch) : 0u;
                }
                tterms[x] = make_uint4(tw[0], tw[1], tw[2], tw[3]);
            }
            __syncthreads();
#pragma unroll 4
            for (uint32_t x = tid; x < W; x += PL_ENGINE_THREADS) {
                const uint2 e1 = perr1[x];
                const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);      /* (outside the row: diff 0, whose terms are 0) */
                const uint4 m2 = x >= 2 ? tterms[x - 2] : z4, m1 = x >= 1 ? tterms[x - 1] : z4, z0 = tterms[x];
                const uint4 p1 = x + 1 < W ? tterms[x + 1] : z4, p2 = x + 2 < W ? tterms[x + 2] : z4;
                const uint32_t am2[4] = { m2.x, m2.y, m2.z, m2.w }, am1[4] = { m1.x, m1.y, m1.z, m1.w }, az0[4] = { z0.x, z0.y, z0.z, z0.w };
                const uint32_t ap1[4] = { p1.x, p1.y, p1.z, p1.w }, ap2[4] = { p2.x, p2.y, p2.z, p2.w };
                uint32_t n0[4], n1[4];
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const uint32_t e1p = p < 2 ? (e1.x >> (16 * p)) : (e1.y >> (16 * (p - 2)));
                    const int c1 = T_(ap2[p]) + F_(ap1[p]) + V_(az0[p]) + F_(am1[p]) + T_(am2[p]);
                    const int c2 = T_(ap1[p]) + H_(az0[p]) + T_(am1[p]);
                    n0[p] = (uint32_t)((int)e1p + c1) & 0xffffu;   /* int16 wrap-on-store */
                    big |= abs(pl_sext16((int)n0[p])) > 8000;
                    bigl |= abs(pl_sext16((int)n0[p])) > PL_E0_LEAD_MAX;
                    n1[p] = (uint32_t)c2 & 0xffffu;
                }
                perr0[x] = make_uint2(n0[0] | (n0[1] << 16), n0[2] | (n0[3] << 16));
                perr1[x] = make_uint2(n1[0] | (n1[1] << 16), n1[2] | (n1[3] << 16));
            }
        } else {
        /* (the pointers do not alias: telling the compiler lets it keep the loads of several iterations in flight) */
#pragma unroll 4
        for (uint32_t x = tid; x < W; x += PL_ENGINE_THREADS) {
            const uint4 cw = cd[x];
            const uint32_t np = ((cw.x & 255u) | ((cw.y & 255u) << 8) | ((cw.z & 255u) << 16) | ((cw.w & 255u) << 24)) & keep;
            oldab[x] = rowp[x];
            rowp[x] = np;
            const uint2 e1 = perr1[x];
            /* the five source pixels of the next-rows terms, loaded once for the four planes */
            const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
            const uint4 cm2 = x >= 2 ? cd[x - 2] : z4, cm1 = x >= 1 ? cd[x - 1] : z4;
            const uint4 cp1 = x + 1 < W ? cd[x + 1] : z4, cp2 = x + 2 < W ? cd[x + 2] : z4;
            uint32_t n0[4], n1[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int ch = pl_channel_of_plane(bpp, p);
                const uint32_t e1p = p < 2 ? (e1.x >> (16 * p)) : (e1.y >> (16 * (p - 2)));
                int c1 = 0, c2 = 0;
                if (ch >= 0) {
                    /* (rows wider than the LDS region: the five source pixels x-2..x+2 looked up per pixel) */
                    auto terms = [&](const uint4 v) -> uint32_t { return terms_of(v, ch); };
                    /* (a pixel outside the row has no record: all-zero words give diff 0, whose terms are 0) */
                    const uint32_t m2 = terms(cm2), m1 = terms(cm1), z0 = terms(cw), p1 = terms(cp1), p2 = terms(cp2);
                    c1 = T_(p2) + F_(p1) + V_(z0) + F_(m1) + T_(m2);
                    c2 = T_(p1) + H_(z0) + T_(m1);
                }
                n0[p] = (uint32_t)((int)e1p + c1) & 0xffffu;   /* int16 wrap-on-store */
                big |= abs(pl_sext16((int)n0[p])) > 8000;
                bigl |= abs(pl_sext16((int)n0[p])) > PL_E0_LEAD_MAX;
                n1[p] = (uint32_t)c2 & 0xffffu;
            }
            perr0[x] = make_uint2(n0[0] | (n0[1] << 16), n0[2] | (n0[3] << 16));
            perr1[x] = make_uint2(n1[0] | (n1[1] << 16), n1[2] | (n1[3] << 16));
        }
        }
        if (big) big_err = 1;
        if (bigl) big_lead = 1;
        for (int b = tid; b < PL_NSYM; b += PL_ENGINE_THREADS) Hc[b] = tbl[winner][b].x;
        if (tid == 0 && j.row_filters) j.row_filters[y] = (uint8_t)(0x08u << winner);   /* PNG_FILTER_* flags */
        if (tid == 0) j.row_ids[y] = (uint8_t)winner;
        if (tid == 0 && j.progress) __hip_atomic_store(j.progress, y + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        cyc_commit += __builtin_readcyclecounter() - tcm0;
        __syncthreads();
    }

    /* ---- epilogue: final histogram + result record (pngloss_image.c:311-325) ---- */
    uint32_t nz = 0;
    for (int b = tid; b < PL_NSYM; b += PL_ENGINE_THREADS) {
        j.final_hist[b] = Hc[b];
        nz += Hc[b] != 0;
    }
    if (tid == 0) uniq = 0;
    __syncthreads();
    if (nz) atomicAdd(&uniq, nz);
    __syncthreads();
    /* diagnostics: per chain wave, cycles spent in the serial chain (>>10) and pixels that needed the exact repair */
    if (lane == 0 && wave < 4) {
        j.result[8 + wave] = (int32_t)(chain_cycles >> 10);
        j.result[12 + wave] = (int32_t)slow_px;
        if (PL_SEGPROF) for (int q = 0; q < 4; q++) j.result[16 + wave * 4 + q] = (int32_t)(segs[q] >> 10);
    }
    if (lane == 0 && wave < PL_NFILT) {
        for (int qq = 0; qq < 5; qq++) j.result[32 + wave * 5 + qq] = (int32_t)(lead_cyc[qq] >> 10);
        j.result[57 + wave] = (int32_t)(lead_cyc[6] ? lead_cyc[5] / lead_cyc[6] : 0);
        j.result[27 + wave] = (int32_t)(lead_cyc[7] >> 10);   /* flush + relation check */
        if (!PL_SEGPROF) j.result[16 + wave] = (int32_t)light_px;   /* light pixels */
    }
    if (lane == 0 && wave == 0) { j.result[62] = (int32_t)(cyc_post >> 10); j.result[63] = (int32_t)(cyc_commit >> 10); }
    {   /* diagnostics: which SIMD each wave of the workgroup sits on (HW_ID bits 5:4), two bits per wave */
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if (lane == 0) atomicOr(&simd_map, ((hwid >> 4) & 3u) << (2 * wave));
    }
    __syncthreads();
    if (tid == 0) j.result[7] = (int32_t)simd_map;
    if (tid == 0) { j.result[22] = (int32_t)est_lead; j.result[23] = (int32_t)est_legacy; }
    if (tid == 0) j.result[21] = (int32_t)adapt_legacy_rows;   /* rows that took the round-1 chains by the adaptive choice */
    if (lane == 0 && wave == 4) {
        j.result[24] = (int32_t)(chain_cycles >> 10);
        j.result[25] = (int32_t)slow_px;
        j.result[26] = (int32_t)lead_rebuilds;
    }
    if (tid == 0) {
        j.result[0] = status;
        j.result[1] = (int32_t)bpp;
        j.result[2] = (int32_t)uniq;
        j.result[3] = (int32_t)retried;
        j.result[4] = (int32_t)slow_px;   /* wave 0's count of pixels that needed the exact channel repair / exact redo */
        j.result[5] = (int32_t)lead_rows;  /* row attempts that ran the band-leader chains */
        j.result[6] = (int32_t)lead_rebuilds;
    }
}

int pl_engine_occupancy(void)
{
    int n = -1;
    if (hipFuncSetAttribute((const void *)pl_engine, hipFuncAttributeMaxDynamicSharedMemorySize, PL_SM_TOTAL) != hipSuccess) return -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pl_engine, PL_ENGINE_THREADS, PL_SM_TOTAL) != hipSuccess) return -1;
    return n;
}

hipError_t pl_launch_engine(const PlJob *d_jobs, const uint32_t *d_sel, size_t n, PlEngineParams prm, hipStream_t stream)
{
    if (!n) return hipSuccess;
    {   /* the attribute belongs to the function ON THE CURRENT DEVICE: remembered per device (a node has up to 8) */
        static std::atomic<unsigned> done{ 0 };
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        if (dev < 0 || dev >= 32 || !(done.load(std::memory_order_acquire) & (1u << dev))) {
            const hipError_t e = hipFuncSetAttribute((const void *)pl_engine, hipFuncAttributeMaxDynamicSharedMemorySize, PL_SM_TOTAL);
            if (e != hipSuccess) {
                /* 104 KB of dynamic LDS per workgroup: a gfx950-class CU (160 KB) has it, the 64 KB parts do not */
                fprintf(stderr, "pngloss_hip: the row engine needs %d bytes of LDS per workgroup (gfx950-class CU with 160 KB); device %d refused: %s\n",
                        (int)PL_SM_TOTAL, dev, hipGetErrorString(e));
                return e;
            }
            if (dev >= 0 && dev < 32) done.fetch_or(1u << dev, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL(pl_engine, dim3((unsigned)n), dim3(PL_ENGINE_THREADS), PL_SM_TOTAL, stream, d_jobs, d_sel, prm);
    return hipGetLastError();
}
