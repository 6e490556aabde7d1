/*
 * pl_pngread.hip -- the PNG READ side on the device: inflated IDAT bytes -> RGBA8, what libpng does for
 * the reference's rwpng_read_image24_libpng (/root/reference/src/rwpng.c:179-400) behind the inflate: the inverse scanline filters and
 * the expansion to RGBA8 (pl_pngread_core.h).  The inflate itself -- a serial bit stream per file -- stays with zlib on host threads
 * (pngloss_amd/cli/png_stream_reader.c); interlaced files stay with libpng.
 *
 * Unfiltering is a recurrence: byte (x, y) needs the reconstructed bytes at (x - bpp, y), (x, y - 1), (x - bpp, y - 1).  Rows are NOT
 * independent, but row y can run one pixel behind row y - 1: a wavefront.  One wave per BAND of 64 rows (blockIdx.x = band,
 * blockIdx.y = image); lane = row of the band, staggered one pixel; band b runs one block behind band b - 1, whose last row it
 * reads from a per-band row buffer once that band's progress word says the block is there (release / acquire at device scope;
 * workgroups are dispatched in blockIdx order, so the band waited for is always running or done; a bounded wait turns a broken
 * assumption into an error status instead of a hang).  A 4096 x 4096 RGBA file: 64 bands in flight instead of one wave walking
 * them in turn (449 ms -> see profiles/r03_read_side.txt); the band's data goes through LDS in blocks of 960 bytes per row (a multiple of every pixel size
 * 1, 2, 3, 4, 6, 8, so blocks cut between pixels), 65 rows (the row above the band first) x 976 bytes = 62 KB.  Neighbouring lanes
 * exchange the "above" bytes through that tile one step apart (wave-synchronous: same wave, program order).  Behind every block all
 * 64 lanes expand its pixels to RGBA8 with coalesced stores.  Images of a batch are independent.
 */
#include "pl_device.h"
#include "pl_pngread.h"

namespace {

#define PR_BLK 960
#define PR_PAD 16
#define PR_STRIDE (PR_PAD + PR_BLK)

__device__ __forceinline__ void pr_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(64) void pr_k_decode(const PrJob *jobs)
{
    extern __shared__ __align__(16) uint8_t pr_tile[];          /* [(PR_ROWS + 1)][PR_STRIDE]: row 0 = the row above the band */
    const PrJob &j = jobs[blockIdx.y];
    const PrFormat &F = j.F;
    const int lane = threadIdx.x;
    const uint32_t W = F.width, H = F.height, rowbytes = F.rowbytes, bppf = F.bppf;
    const size_t S = (size_t)rowbytes + 1;
    int bad = 0;
    const uint32_t band = blockIdx.x;
    if (band >= j.nbands) return;
    const uint8_t *const above_row = band ? j.lastrow + (size_t)(band - 1) * j.lastpitch : nullptr;
    uint8_t *const below_row = j.lastrow + (size_t)band * j.lastpitch;
    uint32_t blk = 0;
    {
        const uint32_t y0 = band * PR_ROWS;
        const int nrows = (int)min((uint32_t)PR_ROWS, H - y0);
        int ft = 0;
        if (lane < nrows) { ft = j.raw[(size_t)(y0 + lane) * S]; if (ft > 4) { bad = 1; ft = 0; } }
        /* left margins: the pixel in front of the row is zero */
        {
            uint8_t *m = pr_tile + (size_t)(lane + 1) * PR_STRIDE;
            for (int k = 0; k < PR_PAD; k += 4) *(uint32_t *)(m + k) = 0u;
            if (lane == 0) for (int k = 0; k < PR_PAD; k += 4) *(uint32_t *)(pr_tile + k) = 0u;
        }
        for (uint32_t b0 = 0; b0 < rowbytes; b0 += PR_BLK, blk++) {
            const int nb = (int)min((uint32_t)PR_BLK, rowbytes - b0);
            /* raw bytes of the band's rows -> tile rows 1.. (dword loads; the source rows start at odd addresses: unaligned loads).  Eight
             * rows' requests are issued before the first store, so that a block costs 8 round trips to memory, not 256 */
            for (int r0 = 0; r0 < nrows; r0 += 8) {
                uint32_t v[8][4];
#pragma unroll
                for (int rr = 0; rr < 8; rr++) {
                    const uint8_t *src = j.raw + (size_t)(y0 + min(r0 + rr, nrows - 1)) * S + 1 + b0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int i = lane * 4 + q * 256;
                        v[rr][q] = 0u;
                        if (i + 4 <= nb) __builtin_memcpy(&v[rr][q], src + i, 4);
                    }
                }
#pragma unroll
                for (int rr = 0; rr < 8; rr++) {
                    if (r0 + rr >= nrows) continue;
                    uint8_t *dst = pr_tile + (size_t)(r0 + rr + 1) * PR_STRIDE + PR_PAD;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int i = lane * 4 + q * 256;
                        if (i + 4 <= nb) *(uint32_t *)(dst + i) = v[rr][q];
                    }
                }
            }
            if (nb & 3) {                                   /* the last, partial dword of every row: lane = row */
                if (lane < nrows) {
                    const uint8_t *src = j.raw + (size_t)(y0 + lane) * S + 1 + b0;
                    uint8_t *dst = pr_tile + (size_t)(lane + 1) * PR_STRIDE + PR_PAD;
                    for (int k = nb & ~3; k < nb; k++) dst[k] = src[k];
                }
            }
            /* the row above the band: the band above has to be past this block */
            if (band) {
                if (lane == 0) {
                    uint32_t spins = 0;
                    while (__hip_atomic_load(&j.progress[band - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) <= blk) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > (1u << 24)) { bad = 2; break; }        /* (seconds: the band above is not running -- give up loudly) */
                    }
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            {
                uint8_t *dst = pr_tile + PR_PAD;
                for (int i = lane * 4; i < nb; i += 256) {
                    uint32_t v = 0u;
                    if (y0) { if (i + 4 <= nb) __builtin_memcpy(&v, above_row + b0 + i, 4); else for (int k = i; k < nb; k++) v |= (uint32_t)above_row[b0 + k] << (8 * (k - i)); }
                    if (i + 4 <= nb) *(uint32_t *)(dst + i) = v; else for (int k = i; k < nb; k++) dst[k] = (uint8_t)(v >> (8 * (k - i)));
                }
            }
            pr_wave_sync();
            /* wavefront: lane r reconstructs pixel (t - r) of its row at step t */
            const int npg = (nb + (int)bppf - 1) / (int)bppf;
            uint8_t *mine = pr_tile + (size_t)(lane + 1) * PR_STRIDE + PR_PAD;
            const uint8_t *up = pr_tile + (size_t)lane * PR_STRIDE + PR_PAD;
            for (int t = 0; t < npg + nrows - 1; t++) {
                const int xi = t - lane;
                if (lane < nrows && xi >= 0 && xi < npg) {
                    const int o = xi * (int)bppf;
                    if (bppf == 4) {
                        const uint32_t x4 = *(const uint32_t *)(mine + o), a4 = *(const uint32_t *)(mine + o - 4), b4 = *(const uint32_t *)(up + o), c4 = *(const uint32_t *)(up + o - 4);
                        *(uint32_t *)(mine + o) = pr_recon4(ft, x4, a4, b4, c4);
                    } else {
                        for (int k = 0; k < (int)bppf && o + k < nb; k++)
                            mine[o + k] = (uint8_t)pr_recon(ft, mine[o + k], mine[o + k - (int)bppf], up[o + k], up[o + k - (int)bppf]);
                    }
                }
                pr_wave_sync();
            }
            /* expansion of the block's pixels to RGBA8, row after row, lanes along x */
            uint32_t px0, px1;
            if (F.bit_depth >= 8) { px0 = b0 / bppf; px1 = (b0 + (uint32_t)nb) / bppf; }
            else { px0 = b0 * 8u / F.bit_depth; px1 = min(W, (b0 + (uint32_t)nb) * 8u / F.bit_depth); }
            for (int r = 0; r < nrows; r++) {
                const uint8_t *rowp = pr_tile + (size_t)(r + 1) * PR_STRIDE + PR_PAD - b0;       /* so that absolute byte offsets index it */
                uint32_t *out = j.rgba + (size_t)(y0 + r) * W;
                for (uint32_t x = px0 + (uint32_t)lane; x < px1; x += 64) out[x] = pr_expand(F, rowp, x);
            }
            /* the band's last row for the next band; the last pixel of every row becomes the margin of the next block */
            if (nrows == PR_ROWS && y0 + PR_ROWS < H) {
                const uint8_t *src = pr_tile + (size_t)PR_ROWS * PR_STRIDE + PR_PAD;
                for (int i = lane; i < nb; i += 64) below_row[b0 + i] = src[i];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) __hip_atomic_store(&j.progress[band], blk + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            uint8_t keep[8];
            {
                const uint8_t *src = pr_tile + (size_t)(lane + 1) * PR_STRIDE + PR_PAD + nb - (int)bppf;
                for (int k = 0; k < (int)bppf; k++) keep[k] = src[k];
            }
            uint8_t keep0[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
            if (lane == 0) { const uint8_t *src = pr_tile + PR_PAD + nb - (int)bppf; for (int k = 0; k < (int)bppf; k++) keep0[k] = src[k]; }
            pr_wave_sync();
            {
                uint8_t *m = pr_tile + (size_t)(lane + 1) * PR_STRIDE + PR_PAD - (int)bppf;
                for (int k = 0; k < (int)bppf; k++) m[k] = keep[k];
                if (lane == 0) { uint8_t *m0 = pr_tile + PR_PAD - (int)bppf; for (int k = 0; k < (int)bppf; k++) m0[k] = keep0[k]; }
            }
            pr_wave_sync();
        }
    }
    if (__builtin_amdgcn_ballot_w64(bad != 0) && lane == 0) *j.status = 25;       /* LIBPNG_FATAL_ERROR (rwpng.h:33): a filter type beyond 4 */
}

} // namespace

hipError_t pl_launch_png_decode(const PrJob *d_jobs, size_t n, uint32_t max_bands, hipStream_t stream)
{
    if (!n || !max_bands) return hipSuccess;
    hipLaunchKernelGGL(pr_k_decode, dim3(max_bands, (unsigned)n), dim3(64), (PR_ROWS + 1) * PR_STRIDE, stream, d_jobs);
    return hipGetLastError();
}
