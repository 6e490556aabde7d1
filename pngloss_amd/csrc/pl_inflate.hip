/*
 * pl_inflate.hip -- the inflate of a window's PNG files on the device: one wave per zlib stream (pl_inflate_core.h has the decoder and says
 * what it replaces: zlib under libpng's png_read_image, /root/reference/src/rwpng.c:179-400).  The streams of a window are independent,
 * so the serial bit stream of one file costs latency, not throughput: ~10 MB/s of scanlines for a lone wave on photographs, 30 - 40 on flat content (round 6: rounds of speculative
 * look-ups by all lanes, the chain followed in scalar registers; 3.3 MB/s before; profiles/r06_inflate.txt), 50 KB of shared memory a stream (32 KB window, 4 KB staged input,
 * tables) = three streams per CU, 768 in flight on the device: 8 GB/s of scanlines in aggregate.
 */
#include "pl_inflate.h"

namespace {

__global__ __launch_bounds__(PLI_NL) void pl_inflate_k(const PliStream *__restrict__ jobs, unsigned n)
{
    __shared__ PliShared S;
    if (blockIdx.x >= n) return;
    const PliStream st = jobs[blockIdx.x];
    pli_inflate(st, S);
}

} // namespace

hipError_t pl_launch_inflate(const PliStream *d_jobs, size_t n, hipStream_t stream)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(pl_inflate_k, dim3((unsigned)n), dim3(PLI_NL), 0, stream, d_jobs, (unsigned)n);
    return hipGetLastError();
}
