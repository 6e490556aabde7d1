/* pl_pngread.h -- device job of the PNG read side (pl_pngread.hip).  Internal. */
#ifndef PL_PNGREAD_H
#define PL_PNGREAD_H

#include <hip/hip_runtime.h>

#include "pl_pngread_core.h"

struct PrJob {
    const uint8_t *raw;     /* device: height * (1 + rowbytes) inflated bytes, filter type first */
    uint32_t *rgba;         /* device: width * height RGBA8 */
    uint8_t *lastrow;       /* device scratch: rowbytes */
    int32_t *status;        /* device: 0 or 25 */
    PrFormat F;
};

hipError_t pl_launch_png_decode(const PrJob *d_jobs, size_t n, hipStream_t stream);

#endif
