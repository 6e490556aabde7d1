/*
 * pl_emit.hip -- first piece of the PNG write side on the device: turn the optimised RGBA8
 * image + the chosen per-row filters into the filtered scanlines a PNG encoder deflates.
 *
 * Replaces what the reference gets from libpng inside rwpng_write_image24 / rwpng_write_end
 * (/root/reference/src/rwpng.c:558-609 colour-type detection and gray repack, :477-501 png_write_row with
 * png_set_filter per row; row 0 -- and every row when row_filters is NULL -- with PNG_ALL_FILTERS, i.e. libpng's
 * minimum-sum-of-absolute-differences heuristic).  The host only deflates and frames chunks (cli/png_stream_writer.c).
 *
 *   pl_emit_classify   gray / opaque detection of the OPTIMISED pixels (it can differ from the input's class)
 *   pl_emit_rows       one workgroup per scanline: heuristic filter if the row is adaptive, then filter + repack
 *                      to 1/2/3/4 bytes per pixel; loads are whole RGBA dwords (16 B per lane), stores whole dwords
 * Both are pure streaming kernels: HBM-bound, ~4 B/px read + <=4 B/px written (the previous row comes from L2).
 */
#include "pl_device.h"

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void pl_emit_classify(const PlJob *jobs)
{
    const PlJob j = jobs[blockIdx.y];
    if (!j.emit_rows) return;
    const size_t n = (size_t)j.width * j.height;
    uint32_t gray = 1, opaque = 1;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
        const uint32_t w = j.img[i];
        const uint32_t r = w & 255u, g = (w >> 8) & 255u, b = (w >> 16) & 255u, a = w >> 24;
        gray &= (r == g) & (g == b);
        opaque &= (a == 255u);
    }
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
        const uint32_t now = __hip_atomic_load(j.out_flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (now & wg_clear) atomicAnd(j.out_flags, ~wg_clear);
    }
}

/* channel k of the OUTPUT pixel format taken from an RGBA dword: gray -> G[,A]; colour -> R,G,B[,A] */
__device__ __forceinline__ int out_channel(uint32_t rgba, uint32_t och, int k)
{
    if (och <= 2) return k == 0 ? (int)((rgba >> 8) & 255u) : (int)(rgba >> 24);
    return (int)((rgba >> (8 * k)) & 255u);
}

__global__ __launch_bounds__(kThreads) void pl_emit_rows(const PlJob *jobs)
{
    __shared__ uint32_t sums[PL_NFILT];
    __shared__ int chosen;
    const PlJob j = jobs[blockIdx.y];
    if (!j.emit_rows) return;
    const uint32_t y = blockIdx.x;
    if (y >= j.height) return;
    const uint32_t W = j.width;
    const uint32_t och = pl_bpp_from_flags(*j.out_flags);
    const uint32_t *row = j.img + (size_t)y * W;
    const uint32_t *up = y ? row - W : nullptr;
    const bool adaptive = y == 0 || j.emit_adaptive_all;
    /* libpng (png_write_start_row / png_set_filter) drops the filters that have no neighbour to predict from: sub, average and
     * paeth in 1-pixel-wide images, up, average and paeth in 1-pixel-high ones; a row asked to use one is written as "none" */
    const uint32_t allowed = (W == 1 ? 0x05u : 0x1fu) & (j.height == 1 ? 0x03u : 0x1fu);

    int f = j.row_ids[y];
    if (!((allowed >> f) & 1u)) f = 0;
    if (adaptive) {
        /* libpng's png_write_find_filter with all five filters: least sum of |signed residual|, first minimum wins */
        if (threadIdx.x < PL_NFILT) sums[threadIdx.x] = 0;
        __syncthreads();
        uint32_t s[PL_NFILT] = { 0, 0, 0, 0, 0 };
        for (uint32_t x = threadIdx.x; x < W; x += kThreads) {
            const uint32_t cur = row[x], lf = x ? row[x - 1] : 0u, ab = up ? up[x] : 0u, dg = (up && x) ? up[x - 1] : 0u;
            for (uint32_t k = 0; k < och; k++) {
                const int c = out_channel(cur, och, k), l = out_channel(lf, och, k), a = out_channel(ab, och, k), d = out_channel(dg, och, k);
                const int preds[PL_NFILT] = { 0, l, a, (a + l) >> 1, pl_paeth(a, d, l) };
#pragma unroll
                for (int g = 0; g < PL_NFILT; g++) {
                    const int v = (c - preds[g]) & 255;
                    s[g] += (uint32_t)(v < 128 ? v : 256 - v);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < PL_NFILT; g++) {
            uint32_t v = s[g];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(&sums[g], v);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int best = 0;                                   /* "none" is always allowed */
            for (int g = 1; g < PL_NFILT; g++)
                if (((allowed >> g) & 1u) && sums[g] < sums[best]) best = g;
            chosen = best;
        }
        __syncthreads();
        f = chosen;
    }
    if (threadIdx.x == 0) j.emit_ids[y] = (uint8_t)f;

    /* filter + repack, four pixels per lane: 16 B in, och dwords out */
    uint32_t *dst = reinterpret_cast<uint32_t *>(j.emit_rows + (size_t)y * j.emit_pitch);
    for (uint32_t x4 = threadIdx.x; x4 * 4 < W; x4 += kThreads) {
        uint32_t cur[4], ab[4], prev_cur, prev_ab;
        const uint32_t x0 = x4 * 4;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t x = x0 + k;
            cur[k] = x < W ? row[x] : 0u;
            ab[k] = (up && x < W) ? up[x] : 0u;
        }
        prev_cur = x0 ? row[x0 - 1] : 0u;
        prev_ab = (up && x0) ? up[x0 - 1] : 0u;
        unsigned char bytes[16];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t lf = k ? cur[k - 1] : prev_cur, dg = k ? ab[k - 1] : prev_ab;
            for (uint32_t ch = 0; ch < och; ch++) {
                const int c = out_channel(cur[k], och, (int)ch), l = out_channel(lf, och, (int)ch);
                const int a = out_channel(ab[k], och, (int)ch), d = out_channel(dg, och, (int)ch);
                bytes[k * och + ch] = (unsigned char)(c - pl_predict_rt(f, a, d, l));
            }
        }
        for (uint32_t wd = 0; wd < och; wd++) {
            const uint32_t v = (uint32_t)bytes[4 * wd] | ((uint32_t)bytes[4 * wd + 1] << 8) | ((uint32_t)bytes[4 * wd + 2] << 16) | ((uint32_t)bytes[4 * wd + 3] << 24);
            dst[(size_t)x4 * och + wd] = v;
        }
    }
}

} // namespace

hipError_t pl_launch_emit(const PlJob *d_jobs, const PlJob *h_jobs, size_t n, hipStream_t stream)
{
    bool any = false;
    uint32_t max_h = 1;
    size_t max_px = 1;
    for (size_t i = 0; i < n; i++)
        if (h_jobs[i].emit_rows) {
            any = true;
            if (h_jobs[i].height > max_h) max_h = h_jobs[i].height;
            const size_t px = (size_t)h_jobs[i].width * h_jobs[i].height;
            if (px > max_px) max_px = px;
        }
    if (!any) return hipSuccess;
    size_t blocks = (max_px + (size_t)kThreads * 16 - 1) / ((size_t)kThreads * 16);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pl_emit_classify, dim3((unsigned)blocks, (unsigned)n), dim3(kThreads), 0, stream, d_jobs);
    hipLaunchKernelGGL(pl_emit_rows, dim3(max_h, (unsigned)n), dim3(kThreads), 0, stream, d_jobs);
    return hipGetLastError();
}
