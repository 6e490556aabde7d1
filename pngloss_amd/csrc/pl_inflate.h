/* pl_inflate.h -- device job of the PNG inflate (pl_inflate.hip).  Internal. */
#ifndef PL_INFLATE_H
#define PL_INFLATE_H

#include <hip/hip_runtime.h>

#include "pl_inflate_core.h"

/* one wave per stream; jobs[i].status receives 0 or a PLI_E_* code */
hipError_t pl_launch_inflate(const PliStream *d_jobs, size_t n, hipStream_t stream);

#endif
