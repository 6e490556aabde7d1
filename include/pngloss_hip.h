/*
 * pngloss_hip.h -- C ABI of the MI355X (gfx950) implementation of pngloss's filter+quantise hot path.
 *
 * libpngloss_hip.so is a drop-in for the ONE seam the reference has on this path:
 *
 *     pngloss_file_internal()  --calls-->  optimize_with_rows()        /root/reference/src/pngloss.c:266
 *                                                                      /root/reference/src/pngloss_image.h:21-25
 *
 * Section 1 re-exports that seam (and its two legacy siblings) with the reference's exact names, argument meaning,
 * ownership rules and return codes, so that relinking pngloss.c against this library instead of
 * pngloss_image.c/optimize_state.c/color_delta.c changes nothing but the speed.  Section 2 is the device-resident /
 * batched extension the reference does not have (its per-file loop, pngloss.c:173, is sequential).
 *
 * Plain C, plain pointers and sizes; no HIP or torch types appear in any signature (streams are passed as void*).
 * There is NO CPU fallback behind these entry points: if the HIP runtime, a gfx950 device or the kernels are
 * unavailable they fail loudly (message on stderr + PNGLOSS_HIP_ERROR), they never silently compute on the host.
 */
#ifndef PNGLOSS_HIP_H
#define PNGLOSS_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- return codes: the subset of the reference's pngloss_error (rwpng.h:23-38) this path can produce ---------- */
#ifndef PNGLOSS_ERROR_CODES
#define PNGLOSS_ERROR_CODES
#define PNGLOSS_SUCCESS             0   /* SUCCESS              rwpng.h:24 */
#define PNGLOSS_OUT_OF_MEMORY_ERROR 17  /* OUT_OF_MEMORY_ERROR  rwpng.h:30 (host or device allocation failed) */
#define PNGLOSS_INVALID_ARGUMENT    4   /* INVALID_ARGUMENT     rwpng.h:27 (bleed outside 1..32767 etc.; the CLI
                                           validates this at pngloss.c:123-131, the library re-checks) */
#define PNGLOSS_HIP_ERROR           64  /* new: HIP runtime / device / kernel failure (no reference equivalent) */
#define PNGLOSS_INTERNAL_ABORT      65  /* new: the invariant the reference abort()s on (optimize_state.c:216-219,
                                           245-248; pngloss_image.c:268-271) was violated on the device */
#endif

/* =====================================================================================================
 * 1. Drop-in seam (host pointers).  Replaces /root/reference/src/pngloss_image.h:14-29.
 * ===================================================================================================== */

/* pngloss_image.h:7-11 -- packed image with 1..4 bytes per pixel, rows need not be contiguous. */
typedef struct {
    unsigned char **rows;
    uint32_t width, height;
    uint_fast8_t bytes_per_pixel;
} pngloss_image;

/* Replaces optimize_with_rows(), pngloss_image.h:21-25 / pngloss_image.c:52-156.
 *   rows          height pointers to width*4 bytes of RGBA8 each, caller-owned, rewritten IN PLACE
 *   row_filters   caller-allocated height bytes or NULL.  Non-NULL: receives libpng filter FLAG values
 *                 0x08,0x10,0x20,0x40,0x80 (PNG_FILTER_NONE..PAETH, pngloss_image.c:290-306) and only row 0 is
 *                 forced to libpng's heuristic filter.  NULL: every row is (pngloss_image.c:210), no IDs returned.
 *   verbose       prints the progress display of pngloss_image.c:214-237 (spinner + percentage of finished rows at 10 Hz, fed
 *                 by a host-mapped word the engine writes per row), then "compression complete" / "used N unique symbols"
 *                 to stderr like :309-325
 *   quantization_strength 0..255, bleed_divider 1..32767 (pngloss.c:123-131)
 * Returns PNGLOSS_SUCCESS or an error above.  Output bytes and filter IDs are bit-identical to the reference. */
int optimize_with_rows(unsigned char **rows, uint32_t width, uint32_t height, unsigned char *row_filters,
                       bool verbose, uint_fast8_t quantization_strength, int_fast16_t bleed_divider);

/* Replaces optimize_with_stride(), pngloss_image.h:17-20 / pngloss_image.c:40-50 (row_filters = NULL mode). */
void optimize_with_stride(unsigned char *pixels, uint32_t width, uint32_t height, uint32_t stride,
                          bool verbose, uint_fast8_t quantization_strength, int_fast16_t bleed_divider);

/* Replaces optimizeForAverageFilter(), pngloss_image.h:14-16 / pngloss_image.c:29-38 (RGBA, bleed fixed at 2). */
void optimizeForAverageFilter(unsigned char pixels[], int width, int height, int quantization);

/* Replaces optimize_image(), pngloss_image.h:26-29 / pngloss_image.c:159-333: the lower seam on an already packed
 * 1/2/3/4 bytes-per-pixel image (no gray/alpha detection). */
int optimize_image(pngloss_image *image, unsigned char *row_filters, bool verbose,
                   uint_fast8_t quantization_strength, int_fast16_t bleed_divider);

/* =====================================================================================================
 * 2. Device-resident, batched extension (new; the natural batching point is pngloss.c:173).
 * ===================================================================================================== */

typedef struct pngloss_hip_ctx pngloss_hip_ctx;

/* One RGBA8 image resident in device memory. */
typedef struct {
    void    *d_rgba;         /* device pointer, width*height*4 bytes, rows contiguous; rewritten in place       */
    void    *d_row_filters;  /* device pointer to height bytes, or NULL (=> all rows adaptive, no IDs)           */
    uint32_t width, height;
} pngloss_hip_image_desc;

/* Per-image result record (host memory, filled after the stream has been synchronised by _finish). */
typedef struct {
    int32_t  status;            /* PNGLOSS_SUCCESS / PNGLOSS_INTERNAL_ABORT                                        */
    uint32_t bytes_per_pixel;   /* 1 gray, 2 gray+alpha, 3 rgb, 4 rgba -- what pngloss_image.c:64-96 detects       */
    uint32_t unique_symbols;    /* non-zero bins of the final histogram (pngloss_image.c:311-325)                  */
    uint32_t retried_rows;      /* rows that needed the strength-decrement retry (pngloss_image.c:266-274)         */
    uint32_t repaired_pixels;   /* diagnostics, no reference equivalent: segment engine: validation restarts (epochs); workgroup engine:
                                   pixels its first chain wave redid exactly                          */
} pngloss_hip_result;

/* Number of HIP devices visible, or a negative PNGLOSS_HIP_ERROR-style code if the runtime is unusable. */
int pngloss_hip_device_count(void);

/* Create / destroy a context bound to one device (device < 0: the current device).  NULL on failure. */
pngloss_hip_ctx *pngloss_hip_create(int device);
void pngloss_hip_destroy(pngloss_hip_ctx *ctx);

/* Enqueue the whole hot path for n device-resident images on `stream` (a hipStream_t passed as void*, NULL = the
 * default stream).  Images are independent and run concurrently.
 * Asynchronous: returns without waiting for the device; call pngloss_hip_finish() to synchronise and collect results.
 * Work the caller enqueues on `stream` behind this call runs behind the batch.  (Two row engines, chosen per batch: one
 * workgroup per image -- everything is enqueued on `stream` before the call returns; one image spread over the whole device -- the
 * number of row attempts depends on the data, so a helper thread of the context feeds them to a stream of the context's own and
 * `stream` waits for the device-written "images finished" word (hipStreamWaitValue32).  On a device without stream memory operations
 * the call waits for that thread instead: then, and only then, it blocks for the duration of the row engine.)
 * One batch per context at a time: the next call must follow pngloss_hip_finish(). */
int pngloss_hip_optimize_batch_async(pngloss_hip_ctx *ctx, const pngloss_hip_image_desc *images, size_t n,
                                     unsigned quantization_strength, long bleed_divider, void *stream);

/* Wait for the last enqueued batch, copy back its n result records (results may be NULL). */
int pngloss_hip_finish(pngloss_hip_ctx *ctx, pngloss_hip_result *results, size_t n);

/* The synchronous form: enqueue + finish in one call.  Its caller waits on the host anyway, so `stream` gets no device-side wait for the segment engine's "finished"
 * word (the helper thread is joined inside the call and `stream` is ordered behind the engine's streams by an event): the engine runs 1-7 % faster without a queue
 * polling that word, and batches of twelve and more frames run as three launch sequences instead of two.  Results are the same either way. */
int pngloss_hip_optimize_batch(pngloss_hip_ctx *ctx, const pngloss_hip_image_desc *images, size_t n,
                               unsigned quantization_strength, long bleed_divider, void *stream,
                               pngloss_hip_result *results);

/* The same for images in HOST memory (the batch form of optimize_with_rows: this is what a multi-file CLI loop calls
 * instead of pngloss.c:173-208's one-file-at-a-time loop).  rgba: width*height*4 contiguous bytes, rewritten in place;
 * row_filters: height bytes or NULL.  Uploads, runs one batch, downloads; synchronous. */
typedef struct {
    unsigned char *rgba;
    unsigned char *row_filters;
    uint32_t width, height;
} pngloss_hip_host_image;

int pngloss_hip_optimize_batch_host(pngloss_hip_ctx *ctx, const pngloss_hip_host_image *images, size_t n,
                                    unsigned quantization_strength, long bleed_divider, pngloss_hip_result *results);

/* ---- Every GPU of the node.  The reference's command line walks its files one at a time (/root/reference/src/pngloss.c:
 * 173-208); a node with several GPUs deals them out instead: one context per device, images split by size (longest first
 * to the least loaded device), one host thread per context, no collective -- images are independent.
 * devices: NULL/"" = $PNGLOSS_DEVICES if set, else every visible device; otherwise a comma separated list of device ordinals,
 * which may repeat ("0,0" = two contexts sharing device 0: the split logic without a second GPU). */
typedef struct pngloss_hip_multi pngloss_hip_multi;
pngloss_hip_multi *pngloss_hip_multi_create(const char *devices);
void pngloss_hip_multi_destroy(pngloss_hip_multi *multi);
int pngloss_hip_multi_count(const pngloss_hip_multi *multi);
/* owner[i] = index of the context image i goes to (deterministic LPT split; exported so that callers can plan and tests can
 * check it against pngloss_amd/shard.py) */
void pngloss_hip_multi_split(const pngloss_hip_host_image *images, size_t n, int parts, int *owner);

/* As above, and additionally returns, per image, the FILTERED SCANLINES a PNG encoder deflates -- the first piece of
 * the PNG write side done on the device (replaces libpng's png_write_row filtering inside rwpng_write_image24,
 * /root/reference/src/rwpng.c:477-501,558-609): the colour type is re-detected from the optimised pixels, gray images
 * are repacked, row 0 (every row when row_filters is NULL) takes libpng's heuristic filter, the others the filter the
 * optimiser chose.  The host then only deflates and frames chunks (pngloss_amd/cli/png_stream_writer.c).
 *   filter_types  caller-allocated height bytes, receives the PNG filter type 0..4 of every scanline
 *   scanlines     caller-allocated height*pitch bytes, receives width*channels filtered bytes per row
 *   pitch         in: bytes between rows of `scanlines`, >= width*4
 *   color_type    out: 0 gray, 4 gray+alpha, 2 RGB, 6 RGBA (8 bits per sample)
 * Entries whose buffers are NULL are skipped. */
typedef struct {
    unsigned char *filter_types;
    unsigned char *scanlines;
    size_t pitch;
    int color_type;
} pngloss_hip_scanlines;

int pngloss_hip_optimize_batch_host_emit(pngloss_hip_ctx *ctx, const pngloss_hip_host_image *images, size_t n,
                                         unsigned quantization_strength, long bleed_divider, pngloss_hip_result *results,
                                         pngloss_hip_scanlines *scanlines);

/* Same again, but the device also DEFLATES the scanlines: per image the caller gets the complete zlib stream of the
 * PNG's IDAT data (what /root/reference/src/rwpng.c:477-637 obtains from libpng + zlib level 9 on the CPU, and where
 * the reference tool spends its time once the hot path is fast).  The stream inflates to exactly the scanlines the
 * _emit call returns -- so the decoded PNG is identical -- but it is not the byte sequence zlib would write:
 * the encoder is the GPU one of pngloss_amd/csrc/pl_deflate_core.h (multi-level match search, optimal parse, 256 KiB
 * blocks that each end byte-aligned).  On the files of the reference's suite its output is 6-10 % smaller than zlib
 * level 9 / Z_FILTERED.
 * `data` must have room for pngloss_hip_zlib_bound(width, height) bytes; `size` = 0 for an empty image.  One image may
 * have at most 1 GiB of scanlines ((4*width+1)*height; positions are 32-bit on the device): larger ones make the call
 * return PNGLOSS_INVALID_ARGUMENT -- use the _emit form and a CPU deflate for those. */
typedef struct {
    unsigned char *data;      /* in: caller's buffer; out: zlib stream (header 78 DA ... Adler-32) */
    size_t capacity;          /* in */
    size_t size;              /* out */
    int color_type;           /* out: 0, 2, 4 or 6 */
    uint32_t blocks[3];       /* out: deflate blocks written as stored / fixed / dynamic */
    uint32_t flags;           /* in: PNGLOSS_HIP_Z_* */
} pngloss_hip_zstream;

/* the caller only wants the stream: the optimised pixels (and row_filters) are not copied back to the host image */
#define PNGLOSS_HIP_Z_STREAM_ONLY 1u

size_t pngloss_hip_zlib_bound(uint32_t width, uint32_t height);

/* pngloss_hip_optimize_batch_host / _emit / _zlib over every context of `multi`: scanlines and streams may be NULL
 * (independently); results[], scanlines[], streams[] are indexed like images[].  Returns the worst status; a batch in which
 * single images failed (results[i].status != 0) returns PNGLOSS_INTERNAL_ABORT with every other image done. */
int pngloss_hip_multi_optimize_batch_host(pngloss_hip_multi *multi, const pngloss_hip_host_image *images, size_t n,
                                          unsigned quantization_strength, long bleed_divider, pngloss_hip_result *results,
                                          pngloss_hip_scanlines *scanlines, pngloss_hip_zstream *streams);

int pngloss_hip_optimize_batch_host_zlib(pngloss_hip_ctx *ctx, const pngloss_hip_host_image *images, size_t n,
                                         unsigned quantization_strength, long bleed_divider, pngloss_hip_result *results,
                                         pngloss_hip_zstream *streams);

/* Milliseconds the last _zlib call spent in the deflate stage (device work + the block-size round trip). */
double pngloss_hip_last_deflate_ms(const pngloss_hip_ctx *ctx);

/* Duration in milliseconds of the row-engine kernel of the last finished batch, measured with hipEvents recorded
 * on the launch stream immediately around that kernel (what bench.py's roofline block reports).  < 0 if none. */
double pngloss_hip_last_engine_ms(const pngloss_hip_ctx *ctx);
/* Same for the whole enqueued pipeline (classify + histograms + repack + engine + unpack). */
double pngloss_hip_last_total_ms(const pngloss_hip_ctx *ctx);

/* Final 256-bin symbol histogram of image `index` of the last finished batch (host buffer of 256 uint32). */
int pngloss_hip_last_histogram(pngloss_hip_ctx *ctx, size_t index, uint32_t *hist256);

/* ---- PNG read side behind the inflate.  Replaces what libpng does for rwpng_read_image24_libpng
 * (/root/reference/src/rwpng.c:179-400) between "inflated IDAT bytes" and "RGBA8 rows": the inverse scanline filters (a recurrence over
 * x and y, run as a row wavefront on the device) and the transformations that reader registers -- palette / low bit depths / tRNS
 * expanded, 16-bit samples stripped to their high byte, gray to RGB, alpha 255 filled in (rwpng.c:239-258).  The caller parses the
 * chunks and inflates IDAT (zlib; pngloss_amd/cli/png_stream_reader.c does, on host threads) and passes
 *   scanlines     height * (1 + rowbytes) inflated bytes of a NON-INTERLACED image, filter type byte first in every row
 *   color_type, bit_depth   as in IHDR;  palette / palette_entries   the PLTE payload (RGB triples);  trns / trns_bytes   the tRNS payload
 *   rgba          out: width * height * 4 bytes, exactly what rwpng_read_image24 returns in rgba_data
 * Returns PNGLOSS_SUCCESS, PNGLOSS_INVALID_ARGUMENT (not a PNG format) or 25 (LIBPNG_FATAL_ERROR, rwpng.h:33: a filter type beyond 4).
 * Interlaced files are not taken (Adam7 passes have their own geometry): the tool reads those with libpng. */
typedef struct {
    const unsigned char *scanlines;
    uint32_t width, height;
    uint8_t color_type, bit_depth;
    const unsigned char *palette;
    uint32_t palette_entries;
    const unsigned char *trns;
    uint32_t trns_bytes;
    unsigned char *rgba;
} pngloss_hip_png_source;

int pngloss_hip_png_decode_batch_host(pngloss_hip_ctx *ctx, const pngloss_hip_png_source *src, size_t n);
/* The same with a status per image (status: n ints or NULL): every image is decoded and downloaded whatever happens to the others, a damaged
 * one gets 25 in its slot (an internal failure PNGLOSS_HIP_ERROR) and the call returns the worst code -- the reference, too, fails only the
 * damaged file (/root/reference/src/pngloss.c:196-204).  Not while a batch is in flight on the context (PNGLOSS_INVALID_ARGUMENT).
 * A failure of the batch AS A WHOLE (bad argument, allocation, copy, launch, device fault: nothing was decoded) puts the call's return value
 * into EVERY status[i] (and NULL into every d_rgba[i] of the device forms): status[i] == 0 always means "image i is decoded". */
int pngloss_hip_png_decode_batch_host_status(pngloss_hip_ctx *ctx, const pngloss_hip_png_source *src, size_t n, int *status);

/* DEVICE-RESIDENT hand-over ("fuse with K0"): the same decode, but the RGBA8 images STAY in device memory -- in a frame
 * arena of the context, apart from the workspace the optimiser uses -- and d_rgba[i] receives their device pointers (width * height * 4 bytes,
 * 256-byte aligned), which go straight into pngloss_hip_image_desc.d_rgba of pngloss_hip_optimize_batch[_async] on the same context: the pixels
 * never travel back to the host between the reader and the optimiser (the classify / histogram kernels read what the expansion kernel
 * wrote).  src[i].rgba is not used (may be NULL).  The frames are valid until the next decode on this context, or its destruction; the
 * optimiser rewrites them in place.  Everything is enqueued on `stream` (a hipStream_t, NULL = the default stream); the call returns when the
 * statuses have arrived (the decode itself takes milliseconds).  status: n ints or NULL, as above.  Replaces the part of
 * /root/reference/src/rwpng.c:179-400 behind the inflate, like pngloss_hip_png_decode_batch_host. */
int pngloss_hip_png_decode_batch_device(pngloss_hip_ctx *ctx, const pngloss_hip_png_source *src, size_t n, void **d_rgba, int *status, void *stream);

/* The same from the COMPRESSED image data: the inflate, too, runs on the device -- one wave per file (pngloss_amd/csrc/pl_inflate_core.h: stored,
 * fixed and dynamic blocks, the 32 KB window in shared memory, Adler-32 checked), for the files of a window, whose streams are independent.
 * This replaces ALL of the reader behind the chunk walk (/root/reference/src/rwpng.c:179-400: libpng's png_read_image = zlib inflate + inverse
 * filters + transformations); what goes up is the file's compressed bytes, what comes out stays on the device.
 *   zstream   the concatenated payloads of the file's IDAT chunks (one zlib stream), zbytes of them
 * A stream the device inflater does not take (damaged, preset dictionary, size mismatch ...) gets status 25 and the call returns 25: read that
 * file on the host (so does a stream shorter than 6 bytes or beyond 4 GiB -- that file only).  One wave decodes ~3 MB/s (profiles/r04_read_side.txt; zlib: ~250 MB/s per host thread): this pays only when a call brings well over
 * a thousand files; otherwise use the scanline form above, with zlib on host threads. */
typedef struct {
    const unsigned char *zstream;
    size_t zbytes;
    uint32_t width, height;
    uint8_t color_type, bit_depth;
    const unsigned char *palette;
    uint32_t palette_entries;
    const unsigned char *trns;
    uint32_t trns_bytes;
} pngloss_hip_png_zsource;

int pngloss_hip_png_decode_batch_device_z(pngloss_hip_ctx *ctx, const pngloss_hip_png_zsource *zsrc, size_t n, void **d_rgba, int *status, void *stream);

/* Page-locked host memory for what goes up to the device (the inflated scanlines handed to the decode calls, images handed to the host
 * batches): a copy from it is one DMA, a copy from pageable memory is staged by the runtime (33 ms against 1.3 ms for 64 MiB, bench.py
 * `transfers`).  NULL when the runtime has none to give; pngloss_hip_pinned_free(NULL) is a no-op. */
void *pngloss_hip_pinned_alloc(size_t bytes);
void pngloss_hip_pinned_free(void *p);

/* What the row engine did for image `index` of the last finished batch (diagnostics; bench.py reports it):
 *   info[0]  engine: 3 = segment-parallel (the image spread over the whole GPU: few large images, latency), 0 = one workgroup per image
 *            (batches), 4 = row statistics (every image of a batch at strength 0, where nothing is quantised and only the filter search of
 *            pngloss_image.c:201-287 is left: info[1] = rows).  Chosen per batch by pngloss_hip_optimize_batch_async from a cost model -- fitted on the reference kind of box, scaled by the device's CU count;
 *            PNGLOSS_HIP_CALIB=1 also calibrates it once per device and process on a small synthetic frame (~30 ms; off by default: the probe is sensitive to the clock governor) -- (wide images and small batches go to
 *            the segment-parallel engine, narrow images and large batches to the other; state sets of up to 1024 chain states, i.e.
 *            most strength / bleed pairs, rows up to 8192 pixels); PNGLOSS_HIP_ENGINE=seg|wg|lead|legacy|mix pins it (test hook).
 *   info[1]  row attempts (engine 3) / rows on the band-leader chains (engine 0)
 *   info[2]  validation restarts (engine 3) / pixels redone exactly (engine 0)
 *   info[3]  rows finished serially (engine 3) / rows on the round-1 chains by the adaptive choice (engine 0)
 *   info[4]  rows in which candidate none was ruled out by its cost bound (engine 3)
 *   info[5]  segments whose entry state was in no enumerated set (engine 3): walked step by step by the chain kernel (seeded state sets), or the
 *            places where a row was broken off and resumed in an epoch (exhaustive state sets: next to never from every state; a batch whose units or segments start from
 *            seeds -- round 6 -- breaks a row off where no seed reached its state and finishes it from every state)
 *   info[6]  launch groups the batch ran as (engine 3; see the option "launch_groups"), info[7] 1 when the call put a device-side wait for the engine on the
 *            caller's stream (the asynchronous entry's non-blocking variant), 0 when it waited on the host
 * No reference equivalent. */
int pngloss_hip_last_engine_info(pngloss_hip_ctx *ctx, size_t index, int32_t info[8]);

/* Options of a context (not while a batch is in flight).  Known: name "engine", value
 *   "auto"    (default) the row engine is chosen image by image from the cost model (see pngloss_hip_last_engine_info)
 *   "seg"     the segment-parallel engine for every image it takes (any strength / bleed; rows up to 2^20 pixels)
 *   "wg"      one workgroup per image (band-leader chains); "lead" / "legacy": its chain variants (diagnostics)
 *   "mix"     alternate the two engines over the images of a batch (diagnostics)
 *   "rows"    like "auto" (the row-statistics engine takes strength 0 under "auto" and "rows" alike; "seg" / "wg" run strength 0 the long way)
 * The environment variable PNGLOSS_HIP_ENGINE (the tests' hook) is read only while this option is at "auto" (once per call).
 * Name "launch_groups", value "auto" | "2" (default) | "3": how many launch sequences a large batch gets on the segment-parallel engine through the
 *   SYNCHRONOUS entry point pngloss_hip_optimize_batch.  "3" is 3-6 % faster at 32-64 frames of 1080p and is meant for a process that never gives
 *   pngloss_hip_optimize_batch_async a stream of its own: a third engine stream in the process slows every later engine run that waits on a caller's stream,
 *   so the library then runs the asynchronous entry in its blocking variant (it returns when the engine is done), and it never creates a third stream once
 *   any context of the process has used such a wait.
 * Returns PNGLOSS_SUCCESS or PNGLOSS_INVALID_ARGUMENT (unknown name or value).
 * Results never depend on an option, on the engine, or on the environment: the remaining environment hooks (PNGLOSS_HIP_SEG_GROUPS, _SEG_UNIT, _ENUM_NT,
 * _NO_STREAM_WAIT, _SEGPROF, _DEBUG ...: timing and test pins) are read once, when a context is created, and none of them changes a byte; the debugging aid
 * that does ("candidate f wins every row") exists in builds made with -DPL_DEBUG_FORCE_FILTER=f only, and pngloss_hip_version() of such a build says so. */
int pngloss_hip_set_option(pngloss_hip_ctx *ctx, const char *name, const char *value);

/* Library / device identification string (static storage). */
const char *pngloss_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PNGLOSS_HIP_H */
