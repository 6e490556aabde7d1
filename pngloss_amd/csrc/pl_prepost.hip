/*
 * pl_prepost.hip -- the fully data-parallel, HBM-bound kernels around the row engine (gfx950).
 *
 *   pl_init      zero/seed the per-image workspace
 *   pl_classify  gray / opaque detection            (replaces pngloss_image.c:64-80, without the early exit)
 *   pl_repack    RGBA8 -> "slots" layout in place   (replaces pngloss_image.c:81-124)
 *   pl_hist      original_frequency[5][256]         (replaces optimize_state.c:66-83, computed once, not 3x)
 *   pl_rank      8-bit order-preserving rank of each original_frequency table (new: lets the row engine compare
 *                the secondary key of optimize_state.c:228-231 inside one 32-bit arg-max word)
 *   pl_unpack    "slots" -> RGBA8 in place           (replaces pngloss_image.c:125-148)
 *
 * One launch covers the whole batch: blockIdx.y = image.  The class (bytes per pixel) is decided on the device and
 * read back from PlJob::flags by every later kernel, so the host never synchronises inside the pipeline.
 */
#include "pl_device.h"
#include <cstdlib>

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void pl_init(const PlJob *jobs)
{
    const PlJob j = jobs[blockIdx.x];
    const uint32_t t = threadIdx.x;
    if (t == 0) {
        *j.flags = PL_FLAG_GRAY | PL_FLAG_OPAQUE;
        *j.out_flags = PL_FLAG_GRAY | PL_FLAG_OPAQUE;
        for (int r = 0; r < 64; r++) j.result[r] = 0;
    }
    for (uint32_t i = t; i < PL_NFILT * PL_NSYM; i += kThreads) j.orig_hist[i] = 0;
    for (uint32_t i = t; i < j.width; i += kThreads) {
        j.err0[i] = make_uint2(0, 0);
        j.err1[i] = make_uint2(0, 0);
        j.old_above[i] = 0;
    }
}

__global__ __launch_bounds__(kThreads) void pl_classify(const PlJob *jobs)
{
    const PlJob j = jobs[blockIdx.y];
    if (j.forced_bpp) return;
    const size_t n = (size_t)j.width * j.height;
    uint32_t gray = 1, opaque = 1;
    /* 16 B per lane per load when the pixel count allows it */
    const size_t n4 = (reinterpret_cast<uintptr_t>(j.img) & 15u) ? 0 : n / 4;
    const uint4 *p4 = reinterpret_cast<const uint4 *>(j.img);
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (size_t)gridDim.x * kThreads) {
        const uint4 v = p4[i];
        const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t r = w[k] & 255u, g = (w[k] >> 8) & 255u, b = (w[k] >> 16) & 255u, a = w[k] >> 24;
            gray &= (r == g) & (g == b);
            opaque &= (a == 255u);
        }
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
        const uint32_t w = j.img[i];
        const uint32_t r = w & 255u, g = (w >> 8) & 255u, b = (w >> 16) & 255u, a = w >> 24;
        gray &= (r == g) & (g == b);
        opaque &= (a == 255u);
    }
    /* one atomic per WORKGROUP at most (8192 same-address atomics, one per wave, used to cost 80 us of a 100 us kernel),
     * and none at all once the flag word already says "neither gray nor opaque" */
    __shared__ uint32_t wg_clear;
    if (threadIdx.x == 0) wg_clear = 0;
    __syncthreads();
    const bool all_gray = __all(gray), all_opaque = __all(opaque);
    if ((threadIdx.x & 63) == 0) {
        const uint32_t clear = (all_gray ? 0u : PL_FLAG_GRAY) | (all_opaque ? 0u : PL_FLAG_OPAQUE);
        if (clear) atomicOr(&wg_clear, clear);
    }
    __syncthreads();
    if (threadIdx.x == 0 && wg_clear) {
        const uint32_t now = __hip_atomic_load(j.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (now & wg_clear) atomicAnd(j.flags, ~wg_clear);
    }
}

__global__ __launch_bounds__(kThreads) void pl_repack(const PlJob *jobs)
{
    const PlJob j = jobs[blockIdx.y];
    if (j.forced_bpp) return;
    const uint32_t bpp = pl_bpp_from_flags(*j.flags);
    if (bpp == 4) return;
    const size_t n = (size_t)j.width * j.height;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
        const uint32_t w = j.img[i];
        uint32_t o;
        if (bpp == 3) o = w & 0x00ffffffu;                       /* r,g,b,0 */
        else if (bpp == 2) o = ((w >> 8) & 255u) | ((w >> 24) << 8); /* g,a,0,0 : gray is the G channel (pngloss_image.c:112-115) */
        else o = (w >> 8) & 255u;                                 /* g,0,0,0 */
        j.img[i] = o;
    }
}

__global__ __launch_bounds__(kThreads) void pl_unpack(const PlJob *jobs)
{
    const PlJob j = jobs[blockIdx.y];
    if (j.forced_bpp) return;
    const uint32_t bpp = pl_bpp_from_flags(*j.flags);
    if (bpp == 4) return;
    const size_t n = (size_t)j.width * j.height;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
        const uint32_t w = j.img[i];
        uint32_t o;
        if (bpp == 3) o = w | 0xff000000u;
        else {
            const uint32_t g = w & 255u;
            const uint32_t a = bpp == 2 ? ((w >> 8) & 255u) : 255u;
            o = g | (g << 8) | (g << 16) | (a << 24);
        }
        j.img[i] = o;
    }
}

/* original_frequency: for every channel byte and each of the five predictors, count (byte - prediction) mod 256,
 * predictions taken from the ORIGINAL neighbours.  LDS-privatised per workgroup, one global atomic per non-empty
 * bin per workgroup at the end.
 *
 * What bounds it is the rate at which a CU retires LDS atomics (20 per pixel) and, next to that, the VALU work per pixel -- not
 * HBM.  So: REP lane-interleaved replicas of the 1280 counters, replicas of one bin in CONSECUTIVE words (different banks; natural
 * images put most residuals into a handful of bins, a wave's lanes would serialise on one word otherwise); a thread takes four
 * neighbouring pixels (one 16-byte load per row, left / diagonal neighbours from registers); sub, up and average residuals of the
 * four channel bytes are taken in one 32-bit word (byte-wise arithmetic without carries across bytes). */
__device__ __forceinline__ uint32_t pl_sub4(uint32_t a, uint32_t b)      /* per byte (a - b) mod 256 */
{
    return ((a | 0x80808080u) - (b & 0x7f7f7f7fu)) ^ (~(a ^ b) & 0x80808080u);
}
__device__ __forceinline__ uint32_t pl_avg4(uint32_t a, uint32_t b)      /* per byte floor((a + b) / 2) */
{
    return (a & b) + (((a ^ b) & 0xfefefefeu) >> 1);
}
template <int REP, int NCH>
__device__ __forceinline__ void pl_hist_count(uint32_t *mine, uint32_t here, uint32_t left, uint32_t above, uint32_t diag)
{
    const uint32_t r1 = pl_sub4(here, left), r2 = pl_sub4(here, above), r3 = pl_sub4(here, pl_avg4(above, left));
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int hv = (here >> (8 * c)) & 255, lv = (left >> (8 * c)) & 255;
        const int av = (above >> (8 * c)) & 255, dv = (diag >> (8 * c)) & 255;
        atomicAdd(&mine[(0 * PL_NSYM + (uint32_t)hv) * REP], 1u);
        atomicAdd(&mine[(1 * PL_NSYM + ((r1 >> (8 * c)) & 255u)) * REP], 1u);
        atomicAdd(&mine[(2 * PL_NSYM + ((r2 >> (8 * c)) & 255u)) * REP], 1u);
        atomicAdd(&mine[(3 * PL_NSYM + ((r3 >> (8 * c)) & 255u)) * REP], 1u);
        atomicAdd(&mine[(4 * PL_NSYM + (uint32_t)((hv - pl_paeth(av, dv, lv)) & 255)) * REP], 1u);
    }
}
template <int REP, int NT, int NCH>
__device__ __forceinline__ void pl_hist_loop(uint32_t *mine, const uint32_t *__restrict__ img, uint32_t W, size_t n)
{
    if (!(W & 3u) && !(reinterpret_cast<uintptr_t>(img) & 15u) && n < 0xffffffffull) {
        /* four pixels per thread and step; a row is a whole number of such quads, so a quad never straddles two rows */
        const uint4 *__restrict__ img4 = reinterpret_cast<const uint4 *>(img);
        const uint32_t W4 = W >> 2, nq = (uint32_t)(n >> 2);
        for (uint32_t q = blockIdx.x * NT + threadIdx.x; q < nq; q += gridDim.x * NT) {
            const uint32_t qx = q % W4;
            const bool has_up = q >= W4;
            const uint4 hq = img4[q];
            uint4 aq = make_uint4(0u, 0u, 0u, 0u);
            uint32_t left = 0u, diag = 0u;
            if (has_up) aq = img4[q - W4];
            if (qx) left = img[(size_t)q * 4 - 1];
            if (has_up && qx) diag = img[(size_t)(q - W4) * 4 - 1];
            pl_hist_count<REP, NCH>(mine, hq.x, left, aq.x, diag);
            pl_hist_count<REP, NCH>(mine, hq.y, hq.x, aq.y, aq.x);
            pl_hist_count<REP, NCH>(mine, hq.z, hq.y, aq.z, aq.y);
            pl_hist_count<REP, NCH>(mine, hq.w, hq.z, aq.w, aq.z);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
            const uint32_t x = (uint32_t)(i % W);
            const bool has_up = i >= W;
            const uint32_t here = img[i];
            const uint32_t left = x ? img[i - 1] : 0u;
            const uint32_t above = has_up ? img[i - W] : 0u;
            const uint32_t diag = (has_up && x) ? img[i - W - 1] : 0u;
            pl_hist_count<REP, NCH>(mine, here, left, above, diag);
        }
    }
}
template <int REP, int NT>
__global__ __launch_bounds__(NT) void pl_hist(const PlJob *__restrict__ jobs)
{
    extern __shared__ __align__(16) uint32_t pl_hist_lds[];               /* [PL_NFILT * PL_NSYM][REP] */
    uint32_t *const h = pl_hist_lds;
    const PlJob j = jobs[blockIdx.y];
    const uint32_t bpp = pl_job_bpp(j);
    for (uint32_t i = threadIdx.x; i < REP * PL_NFILT * PL_NSYM; i += NT) h[i] = 0;
    __syncthreads();
    uint32_t *const mine = h + (threadIdx.x & (REP - 1));
    const size_t n = (size_t)j.width * j.height;
    if (bpp == 4) pl_hist_loop<REP, NT, 4>(mine, j.img, j.width, n);
    else if (bpp == 3) pl_hist_loop<REP, NT, 3>(mine, j.img, j.width, n);
    else if (bpp == 2) pl_hist_loop<REP, NT, 2>(mine, j.img, j.width, n);
    else pl_hist_loop<REP, NT, 1>(mine, j.img, j.width, n);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < PL_NFILT * PL_NSYM; i += NT) {
        uint32_t v = 0;
#pragma unroll
        for (int r = 0; r < REP; r++) v += h[i * REP + r];
        if (v) atomicAdd(&j.orig_hist[i], v);
    }
}

/* rank[f][b] = #{ b' : orig_hist[f][b'] < orig_hist[f][b] }  (0..255): a < b  <=>  rank(a) < rank(b), and
 * a == b <=> rank(a) == rank(b), which is all optimize_state.c:228-235 uses the secondary key for. */
__global__ __launch_bounds__(PL_NSYM) void pl_rank(const PlJob *jobs)
{
    __shared__ uint32_t h[PL_NSYM];
    const PlJob j = jobs[blockIdx.x];
    for (int f = 0; f < PL_NFILT; f++) {
        const uint32_t mine = j.orig_hist[f * PL_NSYM + threadIdx.x];
        __syncthreads();
        h[threadIdx.x] = mine;
        __syncthreads();
        uint32_t r = 0;
        for (int k = 0; k < PL_NSYM; k++) r += h[k] < mine;
        j.orig_rank[f * PL_NSYM + threadIdx.x] = r;
    }
}

inline dim3 batch_grid(const PlJob *h_jobs, size_t n, uint32_t px_per_thread)
{
    size_t max_px = 1;
    for (size_t i = 0; i < n; i++) {
        size_t px = (size_t)h_jobs[i].width * h_jobs[i].height;
        if (px > max_px) max_px = px;
    }
    size_t blocks = (max_px + (size_t)kThreads * px_per_thread - 1) / ((size_t)kThreads * px_per_thread);
    /* enough workgroups to fill 256 CUs several times over, but never more than needed for the batch */
    size_t cap = (2048 + n - 1) / n;
    if (cap < 8) cap = 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return dim3((unsigned)blocks, (unsigned)n, 1);
}



/* pl_hist: 16 replicas (80 KB) and 1024 threads, one workgroup per CU, three rounds of workgroups on a large frame -- measured
 * against 4 / 8 replicas, 256 / 512 threads and 256 ... 2048 workgroups in profiles/r03_hist_variants.txt */
template <int REP, int NT>
hipError_t launch_hist_v(const PlJob *d_jobs, const PlJob *h_jobs, size_t n, hipStream_t stream, unsigned wgs_per_image_cap)
{
    constexpr size_t lds = (size_t)REP * PL_NFILT * PL_NSYM * sizeof(uint32_t);
    static std::atomic<unsigned> optin{ 0 };                  /* (per device: pl_lds_optin) */
    if (lds > 65536) { const hipError_t attr = pl_lds_optin(reinterpret_cast<const void *>(&pl_hist<REP, NT>), lds, optin); if (attr != hipSuccess) return attr; }
    size_t max_px = 1;
    for (size_t i = 0; i < n; i++) {
        const size_t px = (size_t)h_jobs[i].width * h_jobs[i].height;
        if (px > max_px) max_px = px;
    }
    size_t blocks = (max_px + (size_t)NT * 16 - 1) / ((size_t)NT * 16);
    size_t cap = (wgs_per_image_cap + n - 1) / n;
    if (cap < 4) cap = 4;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((pl_hist<REP, NT>), dim3((unsigned)blocks, (unsigned)n), dim3(NT), lds, stream, d_jobs);
    return hipGetLastError();
}
hipError_t launch_hist(const PlJob *d_jobs, const PlJob *h_jobs, size_t n, hipStream_t stream)
{
    static const int variant = [] { const char *e = getenv("PNGLOSS_HIP_HIST"); return e ? atoi(e) : 0; }();
    if (variant == 1) return launch_hist_v<8, 256>(d_jobs, h_jobs, n, stream, 2048);      /* round 2's shape, for comparison */
    return launch_hist_v<16, 1024>(d_jobs, h_jobs, n, stream, 768);
}

} // namespace

hipError_t pl_launch_prepare(const PlJob *d_jobs, const PlJob *h_jobs, size_t n, hipStream_t stream, bool with_hist)
{
    if (!n) return hipSuccess;
    hipError_t e = hipSuccess;
    hipLaunchKernelGGL(pl_init, dim3((unsigned)n), dim3(kThreads), 0, stream, d_jobs);
    hipLaunchKernelGGL(pl_classify, batch_grid(h_jobs, n, 16), dim3(kThreads), 0, stream, d_jobs);
    hipLaunchKernelGGL(pl_repack, batch_grid(h_jobs, n, 8), dim3(kThreads), 0, stream, d_jobs);
    if (!with_hist) return hipGetLastError();
    {
        e = launch_hist(d_jobs, h_jobs, n, stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(pl_rank, dim3((unsigned)n), dim3(PL_NSYM), 0, stream, d_jobs);
    return hipGetLastError();
}

hipError_t pl_launch_finish(const PlJob *d_jobs, const PlJob *h_jobs, size_t n, hipStream_t stream)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(pl_unpack, batch_grid(h_jobs, n, 8), dim3(kThreads), 0, stream, d_jobs);
    return hipGetLastError();
}
