/*
 * pl_rows.hip -- the row engine for STRENGTH 0 ("lossless": pngloss -s0 only searches the filter of every row).
 *
 * At quantization strength 0 the band of every value is the value itself (q = s + 1 = 1: /root/reference/src/optimize_state.c:186-193), so
 * optimize_state_run stores every byte unchanged, every difference is zero, nothing is diffused (optimize_state.c:390-467) and the five candidate
 * rows of a row are all the ORIGINAL row: what is left of optimize_image (/root/reference/src/pngloss_image.c:201-287) is the choice of the row's
 * filter -- the candidate whose symbols (the plain PNG residuals of the original pixels under that filter) are cheapest under the running symbol
 * histogram, cost = sum over bins of n * (33 + clz(H + n)) (optimize_state.c:326-342: every pixel is charged 64 - floor(log2 H[symbol]) under the
 * histogram AFTER the row), lowest index on ties (pngloss_image.c:257); on adaptive rows (row 0, or every row when the caller passes no
 * row_filters) only the candidate that libpng's heuristic picks for the row it wrote is acceptable (optimize_state.c:319-324, 492-562) -- and the
 * row it wrote is the original row whatever the candidate, so that is one filter for all five.  The winner's residuals join the histogram.
 *
 * Nothing couples the rows but that histogram, and the residual COUNTS of a row do not depend on it.  So:
 *
 *   pl_rows_stats    one workgroup per (image, row): the five residual histograms of the row (LDS atomics, four replicas) and the five sums of
 *                    libpng's heuristic -- every row of every image at once, the image read twice (as the row and as the row above): HBM /
 *                    LDS-atomic bound like pl_hist;
 *   pl_rows_decide   ONE WAVE per image, the rows in series, everything in registers (four bins a lane): cost of the five candidates against the
 *                    running histogram, winner, histogram += the winner's counts; the next row's counts are requested while this row is decided;
 *                    the counts of four rows ahead in registers;
 *                    no shared memory, no barrier.
 *
 * The segment engine takes the same images at strength 0 (its state set has one state; tests pin it with PNGLOSS_HIP_ENGINE=seg): 139 Mpixels/s on
 * an 8192 x 8192 frame against this file's ~4 Gpixels/s (profiles/r05_rows_engine.txt).  Results are the reference's bit for bit either way.
 */
#include "pl_device.h"

namespace {

constexpr int kStatThreads = 256;
constexpr int kStatReplicas = 4;

/* (PlJob::rowstat: PL_ROWSTAT_WORDS words per row -- [5][256] counts, [5] heuristic sums, 3 spare) */

__global__ __launch_bounds__(kStatThreads) void pl_rows_stats(const PlJob *__restrict__ jobs)
{
    __shared__ uint32_t hist[kStatReplicas][PL_NFILT][PL_NSYM];
    __shared__ uint32_t hsum[PL_NFILT];
    const PlJob j = jobs[blockIdx.y];
    const uint32_t y = blockIdx.x, W = j.width;
    if (y >= j.height || !j.rowstat) return;
    const uint32_t bpp = pl_job_bpp(j);
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < kStatReplicas * PL_NFILT * PL_NSYM; i += kStatThreads) (&hist[0][0][0])[i] = 0u;
    if (tid < PL_NFILT) hsum[tid] = 0u;
    __syncthreads();
    const uint32_t *row = j.img + (size_t)y * W, *abv = y ? j.img + (size_t)(y - 1u) * W : nullptr;
    uint32_t (*myh)[PL_NSYM] = hist[tid & (kStatReplicas - 1)];
    uint32_t hs[PL_NFILT] = { 0, 0, 0, 0, 0 };
    for (uint32_t x = (uint32_t)tid; x < W; x += kStatThreads) {
        const uint32_t o = row[x], l = x ? row[x - 1] : 0u;
        const uint32_t a = abv ? abv[x] : 0u, d = (abv && x) ? abv[x - 1] : 0u;
        for (uint32_t c = 0; c < bpp; c++) {
            const int sh = 8 * (int)c;
            const int v = (int)((o >> sh) & 255u), lv = (int)((l >> sh) & 255u), av = (int)((a >> sh) & 255u), dv = (int)((d >> sh) & 255u);
            const int pred[PL_NFILT] = { 0, lv, av, (av + lv) >> 1, pl_paeth(av, dv, lv) };      /* optimize_state.c:575-613 */
#pragma unroll
            for (int f = 0; f < PL_NFILT; f++) {
                const int r = (v - pred[f]) & 255;
                atomicAdd(&myh[f][r], 1u);
                hs[f] += (uint32_t)(r < 128 ? r : 256 - r);                                  /* libpng's heuristic: optimize_state.c:492-562 */
            }
        }
    }
#pragma unroll
    for (int f = 0; f < PL_NFILT; f++) {
        uint32_t s = hs[f];
        for (int o2 = 32; o2 > 0; o2 >>= 1) s += __shfl_xor(s, o2, 64);
        if ((tid & 63) == 0 && s) atomicAdd(&hsum[f], s);
    }
    __syncthreads();
    uint32_t *out = j.rowstat + (size_t)y * PL_ROWSTAT_WORDS;
    for (int i = tid; i < PL_NFILT * PL_NSYM; i += kStatThreads) {
        uint32_t s = 0;
#pragma unroll
        for (int r = 0; r < kStatReplicas; r++) s += (&hist[r][0][0])[i];
        out[i] = s;
    }
    if (tid < PL_NFILT) out[PL_NFILT * PL_NSYM + tid] = hsum[tid];
}

/* sum over the wave, the same value in every lane's view as a scalar: four DPP adds inside the 16-lane rows, then the four rows' sums through scalar registers
 * (no LDS crossbar: __shfl_xor is a ds_bpermute a step, and this kernel is ONE dependent chain) */
template <int CTRL>
__device__ __forceinline__ uint32_t rows_dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t rows_wave_sum(uint32_t v)
{
    v += rows_dpp<0x128>(v);        /* row_ror 8, 4, 2, 1: every lane of a 16-lane row holds the row's sum */
    v += rows_dpp<0x124>(v);
    v += rows_dpp<0x122>(v);
    v += rows_dpp<0x121>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

/* ONE wave per image, no shared memory, no barrier: lane l holds bins l, l + 64, l + 128, l + 192 of the running histogram and of the row's five count
 * vectors in registers (the next row's are requested while this row is decided); every lane takes the (uniform) decision.  A row is one dependent chain of
 * ~150 instructions: ~0.4 us (the first version -- five waves, a candidate each, three barriers and a shuffle reduction a row -- took 1.4 us a row). */
__global__ __launch_bounds__(64) void pl_rows_decide(const PlJob *__restrict__ jobs)
{
    const PlJob j = jobs[blockIdx.x];
    const uint32_t height = j.height;
    const int lane = (int)threadIdx.x;
    const uint32_t bpp = pl_job_bpp(j);
    uint32_t H[4] = { 0, 0, 0, 0 }, O[PL_NFILT][4];
    /* the counts of the next kDepth rows are in registers or on their way: a row's decision takes ~0.4 us, a load from device memory 1 - 2 us */
    constexpr int kDepth = 4;
    uint32_t buf[kDepth][PL_NFILT][4], bhs[kDepth][PL_NFILT];
#pragma unroll
    for (int f = 0; f < PL_NFILT; f++) { for (int q = 0; q < 4; q++) O[f][q] = 0; }
#pragma unroll
    for (int k = 0; k < kDepth; k++) {
        const bool in = (uint32_t)k < height && j.rowstat;
        const uint32_t *nx = j.rowstat + (size_t)(in ? k : 0) * PL_ROWSTAT_WORDS;
#pragma unroll
        for (int f = 0; f < PL_NFILT; f++) {
#pragma unroll
            for (int q = 0; q < 4; q++) buf[k][f][q] = in ? nx[(size_t)f * PL_NSYM + lane + 64 * q] : 0u;
            bhs[k][f] = in ? nx[PL_NFILT * PL_NSYM + f] : 0u;
        }
    }
    uint32_t status = 0;
    bool done = false;
    for (uint32_t y0 = 0; y0 < height && !done; y0 += kDepth) {
#pragma unroll
        for (int k = 0; k < kDepth; k++) {
            const uint32_t y = y0 + (uint32_t)k;
            if (y >= height || done) break;
            uint32_t n[PL_NFILT][4], hs[PL_NFILT];
#pragma unroll
            for (int f = 0; f < PL_NFILT; f++) { hs[f] = (uint32_t)__builtin_amdgcn_readfirstlane((int)bhs[k][f]); for (int q = 0; q < 4; q++) n[f][q] = buf[k][f][q]; }    /* (every lane loaded the same word: a scalar from here on, and so is the decision) */
            if (y + kDepth < height) {
                const uint32_t *nx = j.rowstat + (size_t)(y + kDepth) * PL_ROWSTAT_WORDS;
#pragma unroll
                for (int f = 0; f < PL_NFILT; f++) {
#pragma unroll
                    for (int q = 0; q < 4; q++) buf[k][f][q] = nx[(size_t)f * PL_NSYM + lane + 64 * q];
                    bhs[k][f] = nx[PL_NFILT * PL_NSYM + f];
                }
            }
            /* the row's entropy cost under the histogram after the row (optimize_state.c:326-342); a row's cost fits 32 bits (rows up to 2^20 pixels: 4 * 2^20 * 65) */
            uint32_t cost[PL_NFILT];
#pragma unroll
            for (int f = 0; f < PL_NFILT; f++) {
                uint32_t c = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t h = H[q] + n[f][q];
                    c += n[f][q] * (33u + (uint32_t)__clz((int)h));  /* (n = 0 adds nothing whatever clz says of h = 0) */
                    O[f][q] += n[f][q];                         /* (original_frequency of the image, optimize_state.c:66-83: the sum of the rows' counts) */
                }
                cost[f] = rows_wave_sum(c);
            }
            const bool adaptive = !j.row_filters || y == 0;     /* pngloss_image.c:210 */
            int bestg = 0;
#pragma unroll
            for (int g = 1; g < PL_NFILT; g++) if (hs[g] < hs[bestg]) bestg = g;
            uint64_t best = ~0ull; int w = -1;
#pragma unroll
            for (int g = 0; g < PL_NFILT; g++) {
                const uint64_t cg = (adaptive && g != bestg) ? ~0ull : (uint64_t)cost[g];              /* optimize_state.c:319-324 */
                if (cg < best) { best = cg; w = g; }                                                    /* strict <: pngloss_image.c:257 */
            }
            if (w < 0) { status = 65u; done = true; break; }    /* no acceptable row at strength 0: the reference abort()s (pngloss_image.c:268-271) */
            switch (w) {                                        /* (w is a scalar: one branch, four additions) */
            case 0: for (int q = 0; q < 4; q++) H[q] += n[0][q]; break;
            case 1: for (int q = 0; q < 4; q++) H[q] += n[1][q]; break;
            case 2: for (int q = 0; q < 4; q++) H[q] += n[2][q]; break;
            case 3: for (int q = 0; q < 4; q++) H[q] += n[3][q]; break;
            default: for (int q = 0; q < 4; q++) H[q] += n[4][q]; break;
            }
            if (lane == 0) {
                if (j.row_filters) j.row_filters[y] = (uint8_t)(0x08u << w);                           /* PNG_FILTER_* flags, pngloss_image.c:288-308 */
                j.row_ids[y] = (uint8_t)w;
                if (j.progress) __hip_atomic_store(j.progress, y + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    uint32_t nz = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { j.final_hist[lane + 64 * q] = H[q]; nz += H[q] != 0u; }
#pragma unroll
    for (int f = 0; f < PL_NFILT; f++) {
#pragma unroll
        for (int q = 0; q < 4; q++) j.orig_hist[f * PL_NSYM + lane + 64 * q] = O[f][q];                /* (pl_hist is not run for this engine: pngloss_hip_last_histogram reads this) */
    }
    nz = rows_wave_sum(nz);
    if (lane < 64) { j.result[lane] = 0; }
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) {
        j.result[0] = (int32_t)status; j.result[1] = (int32_t)bpp; j.result[2] = (int32_t)nz; j.result[3] = 0;
        j.result[5] = (int32_t)height;                           /* "row attempts": one per row */
        j.result[20] = 4;                                        /* engine id: row statistics (strength 0) */
    }
}

} // namespace

hipError_t pl_launch_rows(const PlJob *d_jobs, const PlJob *h_jobs, size_t n, hipStream_t stream)
{
    if (!n) return hipSuccess;
    uint32_t max_h = 0;
    for (size_t i = 0; i < n; i++) max_h = h_jobs[i].height > max_h ? h_jobs[i].height : max_h;
    if (max_h) hipLaunchKernelGGL(pl_rows_stats, dim3(max_h, (unsigned)n), dim3(kStatThreads), 0, stream, d_jobs);
    hipLaunchKernelGGL(pl_rows_decide, dim3((unsigned)n), dim3(64), 0, stream, d_jobs);
    return hipGetLastError();
}
