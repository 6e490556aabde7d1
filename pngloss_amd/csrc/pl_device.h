/*
 * pl_device.h -- shared device-side definitions for the gfx950 pngloss hot path (internal, not part of the C ABI).
 *
 * Working layout ("slots"): every pixel is one 32-bit word in HBM, channel c of the packed image in byte c.
 *   4 B/px class (rgba)       : the caller's RGBA8 buffer as is
 *   3 B/px class (rgb)        : (r,g,b,0)   -- alpha byte parked at 0, restored to 255 by the unpack kernel
 *   2 B/px class (gray+alpha) : (g,a,0,0)
 *   1 B/px class (gray)       : (g,0,0,0)
 * so the row engine always moves whole dwords (coalesced 256 B per wave load) and the four channel rows of a wave
 * pick their byte with one v_bfe.  This replaces the malloc+repack of /root/reference/src/pngloss_image.c:81-124.
 */
#ifndef PL_DEVICE_H
#define PL_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PL_NFILT 5
#define PL_NSYM 256
#define PL_ROWSTAT_WORDS (PL_NFILT * PL_NSYM + 8)
#define PL_ENGINE_THREADS 512   /* 8 waves: five run the candidate chains, all eight the row passes (two waves per SIMD keep 256 VGPRs each) */

/* class flag bits produced by the classify kernel */
#define PL_FLAG_GRAY 1u
#define PL_FLAG_OPAQUE 2u

/* Everything the kernels need to know about one image of the batch.  Lives in device memory. */
struct PlJob {
    uint32_t *img;        /* slots image, width*height words, in place over the caller's RGBA8 buffer            */
    uint8_t *row_filters; /* device, height bytes, or nullptr                                                    */
    uint32_t width, height;
    uint32_t forced_bpp;  /* 0 = detect (optimize_with_rows); 1..4 = caller says so (optimize_image seam)         */
    /* per-image workspace */
    uint32_t *flags;      /* [1]  PL_FLAG_* (AND-reduced)                                                        */
    uint32_t *orig_hist;  /* [5][256] original_frequency (optimize_state.c:66-83)                                */
    uint32_t *orig_rank;  /* [5][256] order/equality preserving 8-bit rank of orig_hist[f][*]                    */
    uint4 *cand;          /* [5][width] per candidate, per pixel, per channel: byte | (diff16 << 8)              */
    uint2 *err0;          /* [width] 4 x int16 incoming Sierra error for the current row  (color_error row 0)    */
    uint2 *err1;          /* [width] same for the next row                                 (color_error row 1)    */
    uint32_t *old_above;  /* [width] original (pre-optimisation) previous row = last_row_pixels                  */
    uint32_t *final_hist; /* [256]                                                                               */
    uint8_t *row_ids;     /* [height] winning filter 0..4 of every row (always written, also when row_filters is null) */
    uint32_t *out_flags;  /* [1]  PL_FLAG_* of the OPTIMISED image (what the PNG writer side detects, rwpng.c:558-573)  */
    uint8_t *emit_ids;    /* [height] or null: PNG filter type actually used per scanline of the emitted stream         */
    uint8_t *emit_rows;   /* [height][emit_pitch] or null: filtered scanline bytes in the output colour type             */
    uint32_t emit_pitch;  /* bytes between emitted rows (multiple of 16, >= width*4)                                     */
    uint32_t emit_adaptive_all; /* 1: every row takes libpng's heuristic filter (row_filters == NULL mode), 0: only row 0 */
    uint32_t *progress;   /* null, or a host-visible word that receives the number of finished rows (the -v progress display) */
    uint32_t *rowstat;    /* null, or (strength 0: pl_rows.hip) [height][PL_ROWSTAT_WORDS]: per row the residual counts [5][256] of the five filters and libpng's heuristic sums [5] */
    int32_t *result;      /* [64]: [0] status, [1] bpp, [2] unique symbols, [3] retried rows; [20] engine (0 workgroup per image, 3 segment-parallel, 4 row statistics = strength 0).
                             Workgroup engine: [4] pixels redone exactly (chain wave 0 = 'up' on band-leader rows), [5] row attempts on the band-leader
                             chains, [6] band rescans, [7] SIMD map, [8..15] chain kilo-cycles / repaired pixels per chain wave, [16..20] light pixels
                             per chain wave (or PL_SEGPROF stamps [16..31]), [21] rows on the round-1 chains by the adaptive choice, [22..23] last
                             cycles per pixel, [24..31] wave 4 + flush diagnostics, [32..63] phase kilo-cycles per chain (pl_engine.hip epilogue).
                             Segment engine: [4] epochs, [5] row attempts, [6] rows finished serially, [7] rows in which 'none' was ruled out,
                             [24..58] phase clocks when PNGLOSS_HIP_SEGPROF is set (pl_seg_core.h)                                                  */
};

__device__ __forceinline__ uint32_t pl_bpp_from_flags(uint32_t fl)
{
    const bool g = fl & PL_FLAG_GRAY, o = fl & PL_FLAG_OPAQUE;
    return g ? (o ? 1u : 2u) : (o ? 3u : 4u);
}

__device__ __forceinline__ uint32_t pl_job_bpp(const PlJob &j)
{
    return j.forced_bpp ? j.forced_bpp : pl_bpp_from_flags(*j.flags);
}

/* PNG predictors (optimize_state.c:575-613).  All operands 0..255. */
__device__ __forceinline__ int pl_paeth(int above, int diag, int left)
{
    const int p = above - diag, pd = left - diag;
    const int pl = abs(p), pa = abs(pd), pg = abs(p + pd);
    return (pl <= pa && pl <= pg) ? left : (pa <= pg ? above : diag);
}

template <int F>
__device__ __forceinline__ int pl_predict(int above, int diag, int left)
{
    if (F == 1) return left;
    if (F == 2) return above;
    if (F == 3) return (above + left) >> 1;
    if (F == 4) return pl_paeth(above, diag, left);
    return 0;
}

__device__ __forceinline__ int pl_predict_rt(int f, int above, int diag, int left)
{
    switch (f) {
    case 1: return left;
    case 2: return above;
    case 3: return (above + left) >> 1;
    case 4: return pl_paeth(above, diag, left);
    default: return 0;
    }
}

__device__ __forceinline__ int pl_sext8(int v) { return __builtin_amdgcn_sbfe(v, 0, 8); }
__device__ __forceinline__ int pl_sext16(int v) { return __builtin_amdgcn_sbfe(v, 0, 16); }
__device__ __forceinline__ int pl_med3(int v, int lo, int hi) { return min(max(v, lo), hi); }

/* Exact truncating division of a small signed integer (|n| < 2^17) by a positive constant d <= 32767 using the
 * float reciprocal rd = nextafterf(1.0f/d, +inf): trunc(n * rd) == trunc(n / d).  Proof sketch and exhaustive
 * check: tests/test_host_logic.py::test_float_reciprocal_division. */
__device__ __forceinline__ float pl_truncdiv_f(float n, float rd) { return truncf(n * rd); }

/* Sierra split of one error lane (optimize_state.c:397-401,445-467), all in exact small-integer float arithmetic.
 * in : diff16 (already int16-wrapped), rbleed = recip of bleed_divider, r29 = 2*nextafterf(1/9,+inf)
 * out: t "twos", h "threes", f "fours", v "five", rem (what stays for x+1) */
struct PlSplit { float t, h, f, v, rem; };
__device__ __forceinline__ PlSplit pl_sierra_split(int diff16, float rbleed, float r29)
{
    PlSplit s;
    float d = truncf((float)diff16 * rbleed);
    s.t = truncf(d * 0.0625f);   d = fmaf(s.t, -4.0f, d);
    s.h = truncf(d * 0.125f);    d = fmaf(s.h, -2.0f, d);
    s.f = truncf(d * r29);       d = fmaf(s.f, -2.0f, d);
    s.v = truncf(d * 0.5f);      s.rem = d - s.v;
    return s;
}

/* which channel feeds error plane p (color_delta.c:4-41 expand + optimize_state.c:167-171): -1 = plane unused */
__device__ __forceinline__ int pl_channel_of_plane(uint32_t bpp, int p)
{
    if (bpp == 2) return p == 0 ? 0 : (p == 3 ? 1 : -1);
    return p < (int)bpp ? p : -1;
}
__device__ __forceinline__ int pl_plane_of_channel(uint32_t bpp, int c) { return (bpp == 2 && c == 1) ? 3 : c; }

/* A kernel launched with more than 64 KB of dynamic LDS must be opted in with hipFuncAttributeMaxDynamicSharedMemorySize -- an attribute of
 * the function ON THE CURRENT DEVICE: remembered per device in a bit mask the call site owns (a node has up to 8 devices). */
#include <atomic>
inline hipError_t pl_lds_optin(const void *func, size_t bytes, std::atomic<unsigned> &done)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev < 0 || dev >= 32 || !(done.load(std::memory_order_acquire) & (1u << dev))) {
        const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 32) done.fetch_or(1u << dev, std::memory_order_release);
    }
    return hipSuccess;
}

/* launchers implemented in pl_prepost.hip / pl_engine.hip (host side) */
struct PlEngineParams {
    int strength;
    float rq;       /* nextafterf(1/(strength+1), +inf) */
    float rbleed;   /* nextafterf(1/bleed, +inf)        */
    float r29;      /* 2*nextafterf(1/9, +inf)          */
    int engine_mode;   /* low 4 bits, test hook: 0 = band-leader chains where their preconditions hold, round-1 chains where those are
                          measurably slow (default), 1 = round-1 chains only, 2 = band-leader chains wherever possible;
                          bits 8..: debugging aid, 1 + the candidate that wins every row */
    int force_careful; /* test hook: always run the chain variant with explicit int16 wrap handling (normally only
                          rows whose incoming |error| exceeds 8000 use it) */
};

hipError_t pl_launch_prepare(const PlJob *d_jobs, const PlJob *h_jobs, size_t n, hipStream_t stream, bool with_hist = true);   /* with_hist = false: no original_frequency pass (the strength-0 engine sums it from its rows' counts) */
hipError_t pl_launch_engine(const PlJob *d_jobs, const uint32_t *d_sel, size_t n, PlEngineParams prm, hipStream_t stream);   /* d_sel: n job indices, or null = jobs 0..n-1 */
int pl_engine_occupancy(void);   /* workgroups of the row engine per CU according to the HIP occupancy query */
hipError_t pl_launch_rows(const PlJob *d_jobs, const PlJob *h_jobs, size_t n, hipStream_t stream);      /* the row engine of strength 0 (pl_rows.hip) */
hipError_t pl_launch_finish(const PlJob *d_jobs, const PlJob *h_jobs, size_t n, hipStream_t stream);
hipError_t pl_launch_emit(const PlJob *d_jobs, const PlJob *h_jobs, size_t n, hipStream_t stream);

#endif
