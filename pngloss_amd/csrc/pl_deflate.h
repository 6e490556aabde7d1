/* pl_deflate.h -- host entry of the GPU deflate (pl_deflate.hip); internal to libpngloss_hip.so */
#ifndef PL_DEFLATE_H
#define PL_DEFLATE_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "pl_deflate_core.h"

#define PL_DEFLATE_BLOCK_BYTES DFL_DEFAULT_BLOCK_BYTES
#define PL_DEFLATE_MAX_CHAIN   DFL_DEFAULT_MAX_CHAIN
#define PL_DEFLATE_MAX_STREAM  (1ull << 30)     /* scanline bytes handled per group (32-bit positions, ~40 B/position of workspace) */

typedef struct {
    const uint8_t *d_filter_types;   /* device: one filter type (0..4) per row */
    const uint8_t *d_scanlines;      /* device: filtered rows, `pitch` bytes apart */
    uint32_t pitch, rowbytes, height;
    unsigned char *out;              /* host: receives the zlib stream */
    size_t out_capacity, out_size;
    uint32_t blocks_stored, blocks_fixed, blocks_dynamic;
} pl_deflate_image;

size_t pl_deflate_bound(uint32_t width, uint32_t height);     /* capacity that always suffices (4 channels) */
hipError_t pl_deflate_images(pl_deflate_image *imgs, size_t n, hipStream_t stream);

#endif
