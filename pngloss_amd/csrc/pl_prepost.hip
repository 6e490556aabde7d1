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
 * bin per workgroup at the end. */
__global__ __launch_bounds__(kThreads) void pl_hist(const PlJob *jobs)
{
    /* natural images put most residuals into a handful of bins, so the 64 lanes of a wave keep hitting the same LDS
     * word; kRep lane-interleaved replicas cut that serialisation kRep-fold (replica = lane & (kRep-1)) */
    constexpr int kRep = 8;
    /* replicas of one bin sit in CONSECUTIVE words = different banks (a [replica][bin] layout would put all of them into one
     * bank: 1280 words apart) */
    __shared__ uint32_t h[PL_NFILT * PL_NSYM][kRep];
    const PlJob j = jobs[blockIdx.y];
    const uint32_t bpp = pl_job_bpp(j);
    for (uint32_t i = threadIdx.x; i < kRep * PL_NFILT * PL_NSYM; i += kThreads) (&h[0][0])[i] = 0;
    __syncthreads();
    uint32_t *const mine = &h[0][threadIdx.x & (kRep - 1)];
    const uint32_t W = j.width;
    const size_t n = (size_t)W * j.height;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
        const uint32_t x = (uint32_t)(i % W);
        const bool has_up = i >= W;
        const uint32_t here = j.img[i];
        const uint32_t left = x ? j.img[i - 1] : 0u;
        const uint32_t above = has_up ? j.img[i - W] : 0u;
        const uint32_t diag = (has_up && x) ? j.img[i - W - 1] : 0u;
        for (uint32_t c = 0; c < bpp; c++) {
            const int hv = (here >> (8 * c)) & 255, lv = (left >> (8 * c)) & 255;
            const int av = (above >> (8 * c)) & 255, dv = (diag >> (8 * c)) & 255;
            atomicAdd(&mine[(0 * PL_NSYM + (hv & 255)) * kRep], 1u);
            atomicAdd(&mine[(1 * PL_NSYM + ((hv - lv) & 255)) * kRep], 1u);
            atomicAdd(&mine[(2 * PL_NSYM + ((hv - av) & 255)) * kRep], 1u);
            atomicAdd(&mine[(3 * PL_NSYM + ((hv - ((av + lv) >> 1)) & 255)) * kRep], 1u);
            atomicAdd(&mine[(4 * PL_NSYM + ((hv - pl_paeth(av, dv, lv)) & 255)) * kRep], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < PL_NFILT * PL_NSYM; i += kThreads) {
        uint32_t v = 0;
#pragma unroll
        for (int r = 0; r < kRep; r++) v += h[i][r];
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

} // namespace

hipError_t pl_launch_prepare(const PlJob *d_jobs, const PlJob *h_jobs, size_t n, hipStream_t stream)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(pl_init, dim3((unsigned)n), dim3(kThreads), 0, stream, d_jobs);
    hipLaunchKernelGGL(pl_classify, batch_grid(h_jobs, n, 16), dim3(kThreads), 0, stream, d_jobs);
    hipLaunchKernelGGL(pl_repack, batch_grid(h_jobs, n, 8), dim3(kThreads), 0, stream, d_jobs);
    hipLaunchKernelGGL(pl_hist, batch_grid(h_jobs, n, 8), dim3(kThreads), 0, stream, d_jobs);
    hipLaunchKernelGGL(pl_rank, dim3((unsigned)n), dim3(PL_NSYM), 0, stream, d_jobs);
    return hipGetLastError();
}

hipError_t pl_launch_finish(const PlJob *d_jobs, const PlJob *h_jobs, size_t n, hipStream_t stream)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(pl_unpack, batch_grid(h_jobs, n, 8), dim3(kThreads), 0, stream, d_jobs);
    return hipGetLastError();
}
