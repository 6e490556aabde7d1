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
 * Mapping (one workgroup = one image, blockIdx.x = image of the batch; 5 wavefronts of 64):
 *
 *   wave f (0..4)      = candidate PNG filter f (none, sub, up, average, paeth) -- the five candidates of a row are
 *                        independent (pngloss_image.c:213,240), so they run concurrently on the CU's SIMDs.
 *   lanes 16c..16c+15  = channel c of the current pixel (a DPP "row"); the 16 lanes hold the <= s+1 candidate
 *                        symbols of that channel's band (optimize_state.c:186-214), ceil((s+1)/16) per lane.
 *
 * Per pixel the wave does: predict -> re-centre -> band -> clamp (uniform per 16-lane row), gathers
 * {running frequency, rank of original frequency} for its candidates from the wave's private LDS table, arg-maxes
 * the reference's 4-level key with two 4-step DPP row reductions, then repairs the only coupling between the four
 * channels of a pixel -- the histogram increments of the earlier channels (optimize_state.c:221,253) -- exactly, by
 * re-evaluating just the <= 3 bins those channels incremented.  The Sierra terms that feed the same row
 * (x+1, x+2) stay in registers; the eight terms for the next two rows, the derivative error metric, libpng's
 * heuristic and the entropy cost are deferred to passes that are parallel over x (64 pixels per instruction).
 *
 * The x-chain and the row-to-row dependence through the winner's histogram are inherently serial (SURVEY.md
 * Appendix C): this kernel is bound by that dependency chain, not by HBM and not by MFMA.
 */
#include "pl_device.h"

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

struct RowCtx {
    const uint32_t *row;      /* original row y (slots)                    */
    const uint32_t *nabove;   /* optimised row y-1 or nullptr              */
    const uint2 *err0;        /* incoming error for row y                  */
    uint32_t *out;            /* this candidate's cand[] as words: [x][4]  */
    uint2 *tbl;               /* this candidate's {H, rank}[256] in LDS    */
    uint2 (*rec)[4];          /* this candidate's [64][4] chunk records    */
    uint32_t W, y, bpp;
    int s;
    float rq, rbleed, r29;
};

/* ---------------------------------------------------------------------------------------------------------
 * The serial chain for one candidate filter F over one row.  NCT = candidates per lane known at compile time
 * (1: s<=15, 2: s<=31) or 0 for the generic two-sweep loop.
 * --------------------------------------------------------------------------------------------------------- */
template <int F, int NCT>
__device__ __forceinline__ void chain_row(const RowCtx &k, const int lane)
{
    const int c = lane >> 4, jl = lane & 15;
    const uint32_t bpp = k.bpp, W = k.W;
    const bool active = (uint32_t)c < bpp;
    const bool has_alpha = (bpp & 1u) == 0;
    const int s = k.s, q = s + 1;
    const int nc = NCT ? NCT : (q + 15) >> 4;
    const float rq = k.rq, rbleed = k.rbleed, r29 = k.r29;
    uint2 *const T = k.tbl;

    int left = 0, rem = 0, thr_prev = 0, thr_cur = 0;

    for (uint32_t x0 = 0; x0 < W; x0 += 64) {
        /* ---- vector pre-phase: lane = pixel x0+lane; everything that does not depend on the chain ---- */
        {
            const uint32_t xl = x0 + lane;
            const bool ok = xl < W;
            const uint32_t o = ok ? k.row[xl] : 0u;
            const uint32_t a = (ok && k.nabove) ? k.nabove[xl] : 0u;
            const uint32_t d = (ok && k.nabove && xl) ? k.nabove[xl - 1] : 0u;
            const uint2 e = ok ? k.err0[xl] : make_uint2(0u, 0u);
            const bool alpha0 = has_alpha && ((o >> (8u * (bpp - 1u))) & 255u) == 0u;
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                const int p = pl_plane_of_channel(bpp, cc);
                const uint32_t ev = p < 2 ? (e.x >> (16 * p)) : (e.y >> (16 * (p - 2)));
                uint32_t w0 = ((o >> (8 * cc)) & 255u) | (((a >> (8 * cc)) & 255u) << 8) | (((d >> (8 * cc)) & 255u) << 16);
                if (alpha0 && (uint32_t)cc == bpp - 1u) w0 |= 1u << 24;
                k.rec[lane][cc] = make_uint2(w0, (uint32_t)pl_sext16((int)ev));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        const int n = (int)min(64u, W - x0);
        uint2 r = k.rec[0][c];
        for (int i = 0; i < n; i++) {
            const uint2 rn = k.rec[(i + 1) & 63][c];   /* prefetch the next pixel's record */
            const int orig = r.x & 255, above = (r.x >> 8) & 255, diag = (r.x >> 16) & 255;
            const bool tr = has_alpha && (r.x >> 24);

            /* ---- uniform per channel row: optimize_state.c:157-210 ---- */
            const int pred = pl_predict<F>(above, diag, left);
            const int osym = pl_sext8(orig - pred);
            int predc = orig - osym;
            const int err = pl_sext16((int)r.y + rem + thr_prev);
            const int filt = osym + err;
            const int af = abs(filt);
            const int base = (int)((float)af * rq) * q;
            int vmin = filt < 0 ? -base - s : base;
            int vmax = vmin + s;
            const int lo = -predc, hi = lo + 255;
            vmin = pl_med3(vmin, lo, hi);
            vmax = pl_med3(vmax, lo, hi);
            if (tr) { vmin = -pred; vmax = -pred; predc = pred; }   /* optimize_state.c:158-164 */
            const int span = vmax - vmin, josym = osym - vmin;

            /* ---- candidates: gather and two-level arg-max (optimize_state.c:212-244) ---- */
            uint32_t Hwin, K;
            if (NCT == 1) {
                const uint2 e0 = T[(vmin + jl) & 255];
                const bool v0 = jl <= span;
                Hwin = rowmax_u32(v0 ? e0.x : 0u);
                K = rowmax_u32((v0 && e0.x == Hwin) ? key2(e0.y, jl, josym) : 0u);
            } else if (NCT == 2) {
                const uint2 e0 = T[(vmin + jl) & 255];
                const uint2 e1 = T[(vmin + jl + 16) & 255];
                const bool v0 = jl <= span, v1 = jl + 16 <= span;
                Hwin = rowmax_u32(max(v0 ? e0.x : 0u, v1 ? e1.x : 0u));
                const uint32_t k0 = (v0 && e0.x == Hwin) ? key2(e0.y, jl, josym) : 0u;
                const uint32_t k1 = (v1 && e1.x == Hwin) ? key2(e1.y, jl + 16, josym) : 0u;
                K = rowmax_u32(max(k0, k1));
            } else {
                uint32_t m = 0;
                for (int t = 0; t < nc; t++) {
                    const int jj = jl + 16 * t;
                    const uint32_t h = T[(vmin + jj) & 255].x;
                    m = max(m, jj <= span ? h : 0u);
                }
                Hwin = rowmax_u32(m);
                uint32_t kk = 0;
                for (int t = 0; t < nc; t++) {
                    const int jj = jl + 16 * t;
                    const uint2 e = T[(vmin + jj) & 255];
                    kk = max(kk, (jj <= span && e.x == Hwin) ? key2(e.y, jj, josym) : 0u);
                }
                K = rowmax_u32(kk);
            }
            int jwin = 255 - (int)((K - 1u) & 255u);
            uint32_t Rwin = (K - 1u) >> 9;

            /* ---- exact repair of the channel coupling: channel cp chose bin sb and bumped it to sH ---- */
#pragma unroll
            for (int cp = 0; cp < 3; cp++) {
                if ((uint32_t)cp + 1u < bpp) {
                    const int binp = (vmin + jwin) & 255;
                    const uint32_t sb = (uint32_t)__builtin_amdgcn_readlane(binp, 16 * cp);
                    const uint32_t sH = (uint32_t)__builtin_amdgcn_readlane((int)Hwin, 16 * cp) + 1u;
                    const uint32_t sR = (uint32_t)__builtin_amdgcn_readlane((int)Rwin, 16 * cp);
                    const int jj2 = ((int)sb - vmin) & 255;
                    const uint32_t K2 = key2(sR, jj2, josym);
                    const bool better = (c > cp) && (jj2 <= span) && (sH > Hwin || (sH == Hwin && K2 > K));
                    if (better) { Hwin = sH; K = K2; jwin = jj2; Rwin = sR; }
                }
            }

            /* ---- reconstruct, carry the in-row Sierra terms (optimize_state.c:251-260,455,467) ---- */
            const int vwin = vmin + jwin;
            const int back = vwin + predc;
            const int diff = tr ? 0 : pl_sext16(filt - vwin);
            const PlSplit sp = pl_sierra_split(diff, rbleed, r29);
            thr_prev = thr_cur;
            thr_cur = (int)sp.h;
            rem = (int)sp.rem;
            left = back;
            if (jl == 0 && active) {
                atomicAdd(&T[vwin & 255].x, 1u);
                k.out[(size_t)(x0 + i) * 4 + c] = (uint32_t)(back & 255) | ((uint32_t)(diff & 0xffff) << 8);
            }
            r = rn;
        }
    }
}

template <int F>
__device__ __forceinline__ void chain_dispatch(const RowCtx &k, int lane)
{
    const int q = k.s + 1;
    if (q <= 16) chain_row<F, 1>(k, lane);
    else if (q <= 32) chain_row<F, 2>(k, lane);
    else chain_row<F, 0>(k, lane);
}

/* ---------------------------------------------------------------------------------------------------------
 * Per-candidate post pass, parallel over x (lane = pixel): derivative error (optimize_state.c:265-287),
 * libpng's heuristic filter (optimize_state.c:492-562) and entropy cost (optimize_state.c:326-342).
 * Returns the row cost of optimize_state_row (optimize_state.c:360) or UINT64_MAX if rejected (:319-324).
 * --------------------------------------------------------------------------------------------------------- */
__device__ uint64_t post_pass(const PlJob &j, uint32_t y, uint32_t bpp, int f, const uint2 *T, bool adaptive, int lane)
{
    const uint32_t W = j.width;
    const uint32_t *row = j.img + (size_t)y * W;
    const uint32_t *nab = y ? row - W : nullptr;
    const uint4 *cd = j.cand + (size_t)f * W;
    uint64_t derr = 0;
    uint32_t cost = 0;
    uint32_t hs[PL_NFILT] = { 0, 0, 0, 0, 0 };
    for (uint32_t x = lane; x < W; x += 64) {
        const uint4 cw = cd[x];
        const uint4 cl = x ? cd[x - 1] : make_uint4(0, 0, 0, 0);
        const uint32_t o = row[x], ol = x ? row[x - 1] : 0u;
        const uint32_t na = nab ? nab[x] : 0u, nd = (nab && x) ? nab[x - 1] : 0u;
        const uint32_t oa = y ? j.old_above[x] : 0u, od = (y && x) ? j.old_above[x - 1] : 0u;
        const uint32_t cws[4] = { cw.x, cw.y, cw.z, cw.w }, cls[4] = { cl.x, cl.y, cl.z, cl.w };
        for (uint32_t c = 0; c < bpp; c++) {
            const int sh = 8 * c;
            const int back = cws[c] & 255, nl = x ? (int)(cls[c] & 255) : 0;
            const int ov = (o >> sh) & 255, olv = (ol >> sh) & 255;
            const int nav = (na >> sh) & 255, ndv = (nd >> sh) & 255, oav = (oa >> sh) & 255, odv = (od >> sh) & 255;
            const int da = (oav - ov) - (nav - back);
            const int dd = (odv - ov) - (ndv - back);
            const int dl = (olv - ov) - (nl - back);
            const uint32_t w = (bpp <= 2 && c == 0) ? 3u : 1u;      /* gray is replicated into r,g,b (color_delta.c:11-26) */
            derr += (uint64_t)(w * (uint32_t)(da * da + dd * dd + dl * dl));
            cost += 33u + (uint32_t)__clz((int)T[(back - pl_predict_rt(f, nav, ndv, nl)) & 255].x);
            if (adaptive) {
                const int preds[PL_NFILT] = { 0, nl, nav, (nav + nl) >> 1, pl_paeth(nav, ndv, nl) };
#pragma unroll
                for (int g = 0; g < PL_NFILT; g++) {
                    const int b = (back - preds[g]) & 255;
                    hs[g] += (uint32_t)(b < 128 ? b : 256 - b);
                }
            }
        }
    }
    derr = wave_sum_u64(derr);
    cost = wave_sum_u32(cost);
    if (adaptive) {
        int best = 0;
        uint32_t bs = wave_sum_u32(hs[0]);
#pragma unroll
        for (int g = 1; g < PL_NFILT; g++) {
            const uint32_t v = wave_sum_u32(hs[g]);
            if (v < bs) { bs = v; best = g; }
        }
        if (best != f) return ~0ull;
    }
    return derr / 128u + cost;
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

__global__ __launch_bounds__(PL_ENGINE_THREADS) void pl_engine(const PlJob *jobs, PlEngineParams prm)
{
    __shared__ uint2 tbl[PL_NFILT][PL_NSYM];     /* {running symbol_frequency, rank(original_frequency)} per candidate */
    __shared__ uint32_t Hc[PL_NSYM];             /* committed symbol_frequency                                        */
    __shared__ uint2 rec[PL_NFILT][64][4];       /* per-candidate chunk records                                       */
    __shared__ unsigned long long costs[PL_NFILT];

    const PlJob j = jobs[blockIdx.x];
    const uint32_t W = j.width, H = j.height;
    const uint32_t bpp = pl_job_bpp(j);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float r29 = 2.0f * __uint_as_float(__float_as_uint(1.0f / 9.0f) + 1u);

    for (int i = tid; i < PL_NSYM; i += PL_ENGINE_THREADS) Hc[i] = 0;
    for (int i = tid; i < PL_NFILT * PL_NSYM; i += PL_ENGINE_THREADS)
        tbl[i >> 8][i & 255] = make_uint2(0u, j.orig_rank[i]);
    __syncthreads();

    uint32_t retried = 0;
    int status = 0;
    for (uint32_t y = 0; y < H && !status; y++) {
        const bool adaptive = !j.row_filters || y == 0;   /* pngloss_image.c:210 */
        int s = prm.strength;
        int winner = -1;
        for (;;) {
            /* every candidate starts from the committed histogram (optimize_state_copy, pngloss_image.c:240) */
            for (int b = lane; b < PL_NSYM; b += 64) tbl[wave][b].x = Hc[b];
            RowCtx k;
            k.row = j.img + (size_t)y * W;
            k.nabove = y ? k.row - W : nullptr;
            k.err0 = j.err0;
            k.out = reinterpret_cast<uint32_t *>(j.cand + (size_t)wave * W);
            k.tbl = tbl[wave];
            k.rec = rec[wave];
            k.W = W; k.y = y; k.bpp = bpp; k.s = s;
            k.rq = recip_up(s + 1); k.rbleed = prm.rbleed; k.r29 = r29;
            switch (wave) {
            case 0: chain_dispatch<0>(k, lane); break;
            case 1: chain_dispatch<1>(k, lane); break;
            case 2: chain_dispatch<2>(k, lane); break;
            case 3: chain_dispatch<3>(k, lane); break;
            default: chain_dispatch<4>(k, lane); break;
            }
            /* the post pass reads what this wave's own lanes stored: drain them (same CU, same L1) */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint64_t cst = post_pass(j, y, bpp, wave, tbl[wave], adaptive, lane);
            if (lane == 0) costs[wave] = cst;
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
        const uint4 *cd = j.cand + (size_t)winner * W;
        uint32_t *rowp = j.img + (size_t)y * W;
        const uint32_t keep = bpp >= 4 ? 0xffffffffu : ((1u << (8 * bpp)) - 1u);
        for (uint32_t x = tid; x < W; x += PL_ENGINE_THREADS) {
            const uint4 cw = cd[x];
            const uint32_t np = ((cw.x & 255u) | ((cw.y & 255u) << 8) | ((cw.z & 255u) << 16) | ((cw.w & 255u) << 24)) & keep;
            j.old_above[x] = rowp[x];
            rowp[x] = np;
            const uint2 e1 = j.err1[x];
            uint32_t n0[4], n1[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int ch = pl_channel_of_plane(bpp, p);
                const uint32_t e1p = p < 2 ? (e1.x >> (16 * p)) : (e1.y >> (16 * (p - 2)));
                float c1 = 0.f, c2 = 0.f;
                if (ch >= 0) {
                    const PlSplit m2 = split_at(cd, (long)x - 2, W, ch, prm.rbleed, r29);
                    const PlSplit m1 = split_at(cd, (long)x - 1, W, ch, prm.rbleed, r29);
                    const PlSplit z0 = split_at(cd, (long)x, W, ch, prm.rbleed, r29);
                    const PlSplit p1 = split_at(cd, (long)x + 1, W, ch, prm.rbleed, r29);
                    const PlSplit p2 = split_at(cd, (long)x + 2, W, ch, prm.rbleed, r29);
                    c1 = p2.t + p1.f + z0.v + m1.f + m2.t;
                    c2 = p1.t + z0.h + m1.t;
                }
                n0[p] = (uint32_t)((int)e1p + (int)c1) & 0xffffu;   /* int16 wrap-on-store */
                n1[p] = (uint32_t)((int)c2) & 0xffffu;
            }
            j.err0[x] = make_uint2(n0[0] | (n0[1] << 16), n0[2] | (n0[3] << 16));
            j.err1[x] = make_uint2(n1[0] | (n1[1] << 16), n1[2] | (n1[3] << 16));
        }
        for (int b = tid; b < PL_NSYM; b += PL_ENGINE_THREADS) Hc[b] = tbl[winner][b].x;
        if (tid == 0 && j.row_filters) j.row_filters[y] = (uint8_t)(0x08u << winner);   /* PNG_FILTER_* flags */
        __syncthreads();
    }

    /* ---- epilogue: final histogram + result record (pngloss_image.c:311-325) ---- */
    uint32_t nz = 0;
    for (int b = tid; b < PL_NSYM; b += PL_ENGINE_THREADS) {
        j.final_hist[b] = Hc[b];
        nz += Hc[b] != 0;
    }
    __shared__ uint32_t uniq;
    if (tid == 0) uniq = 0;
    __syncthreads();
    if (nz) atomicAdd(&uniq, nz);
    __syncthreads();
    if (tid == 0) {
        j.result[0] = status;
        j.result[1] = (int32_t)bpp;
        j.result[2] = (int32_t)uniq;
        j.result[3] = (int32_t)retried;
    }
}

hipError_t pl_launch_engine(const PlJob *d_jobs, size_t n, PlEngineParams prm, hipStream_t stream)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(pl_engine, dim3((unsigned)n), dim3(PL_ENGINE_THREADS), 0, stream, d_jobs, prm);
    return hipGetLastError();
}
