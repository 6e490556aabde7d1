/*
 * pl_pngread.hip -- the PNG READ side on the device: inflated IDAT bytes -> RGBA8, what libpng does for
 * the reference's rwpng_read_image24_libpng (/root/reference/src/rwpng.c:179-400) behind the inflate: the inverse scanline filters and
 * the expansion to RGBA8 (pl_pngread_core.h).  The inflate itself -- a serial bit stream per file -- stays with zlib on host threads
 * (pngloss_amd/cli/png_stream_reader.c); interlaced files stay with libpng.
 *
 * Unfiltering is a recurrence: byte (x, y) needs the reconstructed bytes at (x - bpp, y), (x, y - 1), (x - bpp, y - 1).  Rows are NOT
 * independent, but row y can run one pixel behind row y - 1: a wavefront.  One wave per BAND of 64 rows (blockIdx.x = band,
 * blockIdx.y = image); lane = row of the band, staggered one step (4, 8 or 12 bytes: a whole number of pixels and of words, pr_step);
 * band b runs one block behind band b - 1, whose last row it
 * reads from a per-band row buffer once that band's progress word says the block is there (release / acquire at device scope;
 * workgroups are dispatched in blockIdx order, so the band waited for is always running or done; a bounded wait turns a broken
 * assumption into an error status instead of a hang).  A 4096 x 4096 RGBA file: 64 bands in flight instead of one wave walking
 * them in turn (449 ms -> 7.5 ms, profiles/r03_read_side.txt).  The band's data goes through LDS in blocks of 960 bytes per row (a multiple
 * of every pixel size 1, 2, 3, 4, 6, 8 and of every step, so blocks cut between pixels), 65 rows (the row above the band first) x 976 bytes = 62 KB.  Neighbouring lanes
 * exchange the "above" bytes through that tile one step apart (wave-synchronous: same wave, program order).  Behind every block the
 * band's last row is published, then all 64 lanes expand the block's pixels to RGBA8 with coalesced stores.  Images of a batch are independent.
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

/* One wavefront step of a lane: the next CH bytes of its row -- a whole number of pixels and of 32-bit words (4 bytes for 1-, 2- and
 * 4-byte pixels, 12 for 3 and 6, 8 for 8) -- read as words, reconstructed byte after byte in registers (a byte's left neighbour is BPP
 * bytes back: in this chunk, or in the words in front of it), written back as words.  The row above is complete up to the end of the
 * same chunk (its lane is one step ahead).  BPP is a template parameter so that every byte index is a constant. */
template <int BPP>
__device__ __forceinline__ void pr_step(int ft, uint8_t *mine, const uint8_t *up, int o)
{
    constexpr int CH = (BPP == 3 || BPP == 6) ? 12 : (BPP == 8 ? 8 : 4), NW = CH / 4, PW = (BPP + 3) / 4;
    if (BPP == 4) {
        const uint32_t x4 = *(const uint32_t *)(mine + o), a4 = *(const uint32_t *)(mine + o - 4), b4 = *(const uint32_t *)(up + o), c4 = *(const uint32_t *)(up + o - 4);
        *(uint32_t *)(mine + o) = pr_recon4(ft, x4, a4, b4, c4);
        return;
    }
    uint32_t X[NW], B[NW], R[NW], AP[PW], CP[PW];
#pragma unroll
    for (int w = 0; w < NW; w++) { X[w] = *(const uint32_t *)(mine + o + 4 * w); B[w] = *(const uint32_t *)(up + o + 4 * w); R[w] = 0u; }
#pragma unroll
    for (int w = 0; w < PW; w++) { AP[w] = *(const uint32_t *)(mine + o - 4 * PW + 4 * w); CP[w] = *(const uint32_t *)(up + o - 4 * PW + 4 * w); }
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const int x = (int)((X[k >> 2] >> (8 * (k & 3))) & 255u), b = (int)((B[k >> 2] >> (8 * (k & 3))) & 255u);
        int a, c;
        if (k >= BPP) {
            const int li = k - BPP;
            a = (int)((R[li >> 2] >> (8 * (li & 3))) & 255u); c = (int)((B[li >> 2] >> (8 * (li & 3))) & 255u);
        } else {
            const int li = 4 * PW + k - BPP;
            a = (int)((AP[li >> 2] >> (8 * (li & 3))) & 255u); c = (int)((CP[li >> 2] >> (8 * (li & 3))) & 255u);
        }
        R[k >> 2] |= (uint32_t)pr_recon(ft, x, a, b, c) << (8 * (k & 3));
    }
#pragma unroll
    for (int w = 0; w < NW; w++) *(uint32_t *)(mine + o + 4 * w) = R[w];
}

template <int BPP>
__device__ __forceinline__ void pr_wavefront(int ft, uint8_t *mine, const uint8_t *up, int npg, int nrows, int lane)
{
    constexpr int CH = (BPP == 3 || BPP == 6) ? 12 : (BPP == 8 ? 8 : 4);
    for (int t = 0; t < npg + nrows - 1; t++) {
        const int xi = t - lane;
        if (lane < nrows && xi >= 0 && xi < npg) pr_step<BPP>(ft, mine, up, xi * CH);
        pr_wave_sync();
    }
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
            /* wavefront: lane r reconstructs chunk (t - r) of its row at step t (pr_step: 4, 8 or 12 bytes, a whole number of pixels).  A
             * partial last chunk of a row's last block runs over stale tile bytes behind nb: never read as pixels, never past the row's 960 */
            const int ch = (bppf == 3 || bppf == 6) ? 12 : (bppf == 8 ? 8 : 4);
            const int npg = (nb + ch - 1) / ch;
            uint8_t *mine = pr_tile + (size_t)(lane + 1) * PR_STRIDE + PR_PAD;
            const uint8_t *up = pr_tile + (size_t)lane * PR_STRIDE + PR_PAD;
            switch (bppf) {                                 /* (outside the loop: one specialised loop per pixel size) */
            case 1: pr_wavefront<1>(ft, mine, up, npg, nrows, lane); break;
            case 2: pr_wavefront<2>(ft, mine, up, npg, nrows, lane); break;
            case 3: pr_wavefront<3>(ft, mine, up, npg, nrows, lane); break;
            case 4: pr_wavefront<4>(ft, mine, up, npg, nrows, lane); break;
            case 6: pr_wavefront<6>(ft, mine, up, npg, nrows, lane); break;
            default: pr_wavefront<8>(ft, mine, up, npg, nrows, lane); break;
            }
            /* the band's last row for the band below, published before this block's pixels are expanded: the band below starts on the block meanwhile */
            if (nrows == PR_ROWS && y0 + PR_ROWS < H) {
                const uint8_t *src = pr_tile + (size_t)PR_ROWS * PR_STRIDE + PR_PAD;
                for (int i = lane; i < nb; i += 64) below_row[b0 + i] = src[i];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) __hip_atomic_store(&j.progress[band], blk + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
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
            /* the last pixel of every row becomes the margin of the next block */
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
    /* status of the image: 25 = LIBPNG_FATAL_ERROR (rwpng.h:33): a filter type beyond 4; 64 = the band above never reported (an internal
     * error, not a damaged file) -- the larger code wins when bands disagree */
    if (__builtin_amdgcn_ballot_w64(bad == 2) && lane == 0) atomicMax(j.status, 64);
    else if (__builtin_amdgcn_ballot_w64(bad == 1) && lane == 0) atomicMax(j.status, 25);
}

} // namespace

hipError_t pl_launch_png_decode(const PrJob *d_jobs, size_t n, uint32_t max_bands, hipStream_t stream)
{
    if (!n || !max_bands) return hipSuccess;
    hipLaunchKernelGGL(pr_k_decode, dim3(max_bands, (unsigned)n), dim3(64), (PR_ROWS + 1) * PR_STRIDE, stream, d_jobs);
    return hipGetLastError();
}
