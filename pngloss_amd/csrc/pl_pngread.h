/* pl_pngread.h -- device job of the PNG read side (pl_pngread.hip).  Internal. */
#ifndef PL_PNGREAD_H
#define PL_PNGREAD_H

#include <hip/hip_runtime.h>

#include "pl_pngread_core.h"

struct PrJob {
    const uint8_t *raw;     /* device: height * (1 + rowbytes) inflated bytes, filter type first */
    uint32_t *rgba;         /* device: width * height RGBA8 */
    uint8_t *lastrow;       /* device scratch: one row of `lastpitch` bytes per band of PR_ROWS rows (the band's last row, for the band below) */
    uint32_t *progress;     /* device, zeroed before the launch: per band, the number of blocks whose last row is in `lastrow` */
    uint32_t lastpitch, nbands;
    int32_t *status;        /* device: 0 or 25 */
    PrFormat F;
};

#define PR_ROWS 64          /* rows per band: one wave, lane = row */
hipError_t pl_launch_png_decode(const PrJob *d_jobs, size_t n, uint32_t max_bands, hipStream_t stream);

#endif
