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
 *   pl_rows_decide   one workgroup per image, five waves (a candidate each, four bins a lane), the rows in series: cost of the five candidates
 *                    against the running histogram, winner, histogram += the winner's counts; the next row's counts are requested while this
 *                    row is decided.  ~1 us a row.
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

constexpr int kDecideThreads = PL_NFILT * 64;

__global__ __launch_bounds__(kDecideThreads) void pl_rows_decide(const PlJob *__restrict__ jobs)
{
    __shared__ uint32_t H[PL_NSYM];
    __shared__ unsigned long long cost[PL_NFILT];
    __shared__ int win;
    const PlJob j = jobs[blockIdx.x];
    const uint32_t height = j.height;
    const int tid = (int)threadIdx.x, f = tid >> 6, lane = tid & 63;
    const uint32_t bpp = pl_job_bpp(j);
    for (int i = tid; i < PL_NSYM; i += kDecideThreads) H[i] = 0u;
    __syncthreads();
    uint32_t status = 0;
    uint32_t nn[4] = { 0, 0, 0, 0 }, nhs[PL_NFILT] = { 0, 0, 0, 0, 0 };
    if (height && j.rowstat) {
#pragma unroll
        for (int q = 0; q < 4; q++) nn[q] = j.rowstat[(size_t)f * PL_NSYM + lane + 64 * q];
        if (tid == 0) { for (int g = 0; g < PL_NFILT; g++) nhs[g] = j.rowstat[PL_NFILT * PL_NSYM + g]; }
    }
    for (uint32_t y = 0; y < height; y++) {
        uint32_t n[4], hs[PL_NFILT];
#pragma unroll
        for (int q = 0; q < 4; q++) n[q] = nn[q];
#pragma unroll
        for (int g = 0; g < PL_NFILT; g++) hs[g] = nhs[g];
        if (y + 1 < height) {                                   /* the next row's counts are on their way while this row is decided */
            const uint32_t *nx = j.rowstat + (size_t)(y + 1) * PL_ROWSTAT_WORDS;
#pragma unroll
            for (int q = 0; q < 4; q++) nn[q] = nx[(size_t)f * PL_NSYM + lane + 64 * q];
            if (tid == 0) { for (int g = 0; g < PL_NFILT; g++) nhs[g] = nx[PL_NFILT * PL_NSYM + g]; }
        }
        /* the row's entropy cost under the histogram after the row (optimize_state.c:326-342) */
        unsigned long long c = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t h = H[lane + 64 * q] + n[q];
            c += n[q] ? (unsigned long long)n[q] * (33u + (uint32_t)__builtin_clz(h)) : 0ull;
        }
        for (int o2 = 32; o2 > 0; o2 >>= 1) c += __shfl_xor(c, o2, 64);
        if (lane == 0) cost[f] = c;
        __syncthreads();
        if (tid == 0) {
            const bool adaptive = !j.row_filters || y == 0;         /* pngloss_image.c:210 */
            int bestg = 0;
            for (int g = 1; g < PL_NFILT; g++) if (hs[g] < hs[bestg]) bestg = g;
            unsigned long long best = ~0ull; int w = -1;
            for (int g = 0; g < PL_NFILT; g++) {
                const unsigned long long cg = (adaptive && g != bestg) ? ~0ull : cost[g];          /* optimize_state.c:319-324 */
                if (cg < best) { best = cg; w = g; }                                                /* strict <: pngloss_image.c:257 */
            }
            win = w;
            if (w >= 0) {
                if (j.row_filters) j.row_filters[y] = (uint8_t)(0x08u << w);                       /* PNG_FILTER_* flags, pngloss_image.c:288-308 */
                j.row_ids[y] = (uint8_t)w;
                if (j.progress) __hip_atomic_store(j.progress, y + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __syncthreads();
        const int w = win;
        if (w < 0) { status = 65u; break; }                     /* no acceptable row at strength 0: the reference abort()s (pngloss_image.c:268-271) */
        if (f == w) {
#pragma unroll
            for (int q = 0; q < 4; q++) H[lane + 64 * q] += n[q];
        }
        __syncthreads();
    }
    __syncthreads();
    uint32_t nz = 0;
    for (int i = tid; i < PL_NSYM; i += kDecideThreads) { j.final_hist[i] = H[i]; nz += H[i] != 0u; }
    for (int o2 = 32; o2 > 0; o2 >>= 1) nz += __shfl_xor(nz, o2, 64);
    if (tid < PL_NFILT) cost[tid] = 0ull;
    __syncthreads();
    if (lane == 0 && nz) atomicAdd(&cost[0], (unsigned long long)nz);
    __syncthreads();
    if (tid == 0) {
        for (int i = 0; i < 64; i++) j.result[i] = 0;
        j.result[0] = (int32_t)status; j.result[1] = (int32_t)bpp; j.result[2] = (int32_t)cost[0]; j.result[3] = 0;
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
    hipLaunchKernelGGL(pl_rows_decide, dim3((unsigned)n), dim3(kDecideThreads), 0, stream, d_jobs);
    return hipGetLastError();
}
