// ubench_boundary.hip -- what a kernel boundary costs the segment engine: latency of dependent loads in a kernel that
// follows a producer kernel on the same stream (gfx950, one MI355X).  Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_boundary tools/ubench_boundary.hip
//   case A  data written by the previous kernel (all CUs, so mostly by other XCDs), read as a chain of dependent loads
//   case B  data written long ago and not touched since (read-only for this kernel)
//   case C  the same addresses a second time inside the kernel (L2 / L1 hit)
//   case D  N independent loads issued back to back, then one wait (what a burst costs)
// plus the duration of an empty kernel and of a kernel that does one load, from HIP events over 2000 back-to-back launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_produce(uint32_t *buf, unsigned n, unsigned stride, unsigned salt)
{
    // element i * stride holds the index of the next element of the chain (a permutation with one cycle: i -> i + 1)
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    // salt 0: i -> i + 1 (the consumer's first 64 steps stay inside what producer workgroup 0 wrote);
    // salt != 0: i -> i + 257 (every step lands in what ANOTHER producer workgroup -- mostly on another XCD -- wrote)
    if (i < n) buf[(size_t)i * stride] = ((i + (salt ? 257u : 1u)) % n) * stride;
}
__global__ void k_empty() {}
// barriers: nb rounds of (one LDS write, barrier, one LDS read of a neighbour's word) in a workgroup of blockDim.x threads
__global__ __launch_bounds__(1024) void k_barriers(unsigned nb, unsigned long long *res, uint32_t *sink)
{
    __shared__ uint32_t lds[1024];
    uint32_t v = threadIdx.x;
    __syncthreads();
    unsigned long long t0 = wall_clock64();
    for (unsigned k = 0; k < nb; k++) {
        lds[threadIdx.x] = v + k;
        __syncthreads();
        v = lds[(threadIdx.x + 67) % blockDim.x];
        __syncthreads();
    }
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { res[0] = t1 - t0; sink[0] = v; }
}
// the same with a pending global store in front of every barrier
__global__ __launch_bounds__(1024) void k_barriers_st(unsigned nb, unsigned long long *res, uint32_t *sink, uint32_t *glob)
{
    __shared__ uint32_t lds[1024];
    uint32_t v = threadIdx.x;
    __syncthreads();
    unsigned long long t0 = wall_clock64();
    for (unsigned k = 0; k < nb; k++) {
        lds[threadIdx.x] = v + k;
        glob[(size_t)k * 1024 + threadIdx.x] = v;
        __syncthreads();
        v = lds[(threadIdx.x + 67) % blockDim.x];
        __syncthreads();
    }
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { res[0] = t1 - t0; sink[0] = v; }
}
// producer whose working workgroups are those with blockIdx.x % 8 == xcd (workgroups go to the XCDs round robin): fills 32 KB per 256 KB block
__global__ __launch_bounds__(1024) void k_produce_xcd(uint32_t *buf, unsigned xcd, unsigned salt)
{
    if (blockIdx.x % 8 != xcd) return;
    const unsigned q = blockIdx.x / 8;      // which 256 KB block
    buf[(size_t)q * 1024 * 64 + threadIdx.x] = threadIdx.x + salt;
}
// a chain whose every step lands in another 2 MB page (and another 64 KB block): what address translation adds after a kernel boundary
__global__ void k_produce_far(uint32_t *buf, unsigned n, size_t stride_words)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[(size_t)i * stride_words] = (i + 1) % n;
}
__global__ void k_chase_far(const uint32_t *buf, size_t stride_words, unsigned nsteps, unsigned long long *res)
{
    if (threadIdx.x) return;
    unsigned long long t0 = wall_clock64();
    uint32_t p = 0;
    for (unsigned k = 0; k < nsteps; k++) p = buf[(size_t)p * stride_words];
    unsigned long long t1 = wall_clock64();
    uint32_t q = p == 12345u ? 1u : 0u;
    for (unsigned k = 0; k < nsteps; k++) q = buf[(size_t)q * stride_words];
    unsigned long long t2 = wall_clock64();
    res[0] = t1 - t0; res[1] = t2 - t1; res[5] = p + q;
}
// a burst: every thread of a 1024-thread workgroup loads `per` words (coalesced) written by the previous kernel, stores them to LDS
template <unsigned per>
__global__ __launch_bounds__(1024) void k_burst(const uint32_t *buf, unsigned long long *res, uint32_t *sink)
{
    __shared__ uint32_t lds[8192];
    unsigned long long t0 = wall_clock64();
    uint32_t v[per];
#pragma unroll
    for (unsigned q = 0; q < per; q++) v[q] = buf[(size_t)q * 1024 * 64 + (size_t)blockIdx.x * 1024 + threadIdx.x];     // every q another 256 KB away
#pragma unroll
    for (unsigned q = 0; q < per; q++) lds[q * 1024 + threadIdx.x] = v[q];
    __syncthreads();
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { res[0] = t1 - t0; sink[0] = lds[5]; }
}
__global__ void k_oneload(const uint32_t *buf, uint32_t *out) { if (threadIdx.x == 0) out[0] = buf[0]; }
__global__ void k_chase(const uint32_t *fresh, const uint32_t *old, unsigned nsteps, unsigned long long *res)
{
    if (threadIdx.x) return;
    unsigned long long t0 = wall_clock64();
    uint32_t p = 0;
    for (unsigned k = 0; k < nsteps; k++) p = fresh[p];
    unsigned long long t1 = wall_clock64();
    uint32_t q = p & 1u ? 0u : 0u;
    for (unsigned k = 0; k < nsteps; k++) q = old[q];
    unsigned long long t2 = wall_clock64();
    uint32_t r = (q & 1u) ? 0u : 0u;
    for (unsigned k = 0; k < nsteps; k++) r = fresh[r];
    unsigned long long t3 = wall_clock64();
    uint32_t acc = 0;
    for (unsigned k = 0; k < nsteps; k++) acc += old[(size_t)(nsteps + k) * 64];     // independent, never touched lines
    res[5] = acc + p + q + r;
    unsigned long long t4 = wall_clock64();
    res[0] = t1 - t0; res[1] = t2 - t1; res[2] = t3 - t2; res[3] = t4 - t3;
}

int main()
{
    const unsigned n = 1 << 16, stride = 64;   // one element per 256 B: every step a new cache line
    uint32_t *fresh, *old; unsigned long long *res; uint32_t *out;
    CK(hipMalloc(&fresh, (size_t)n * stride * 4)); CK(hipMalloc(&old, (size_t)n * stride * 4)); CK(hipMalloc(&res, 64)); CK(hipMalloc(&out, 64));
    CK(hipMemset(old, 0, (size_t)n * stride * 4));
    hipLaunchKernelGGL(k_produce, dim3(n / 256), dim3(256), 0, 0, old, n, stride, 0u);
    CK(hipDeviceSynchronize());
    const unsigned nsteps = 64;
    const int reps = 200;
    const double tick_ns = 10.0;   // wall_clock64: 100 MHz
    for (unsigned far_chain = 0; far_chain < 2; far_chain++) {
    double sum[4] = { 0, 0, 0, 0 };
    for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL(k_produce, dim3(n / 256), dim3(256), 0, 0, fresh, n, stride, far_chain);
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, fresh, old, nsteps, res);
        unsigned long long h[6];
        CK(hipMemcpy(h, res, sizeof h, hipMemcpyDeviceToHost));
        if (r >= 8) for (int q = 0; q < 4; q++) sum[q] += (double)h[q];
        // keep "old" cold for the next repetition: walk a different part of it next time would need more memory; instead flush by a big memset elsewhere
        CK(hipMemsetAsync(out, 0, 64, 0));
    }
    printf("dependent load, line written by the previous kernel (%s): %.0f ns per load\n", far_chain ? "by other workgroups, other XCDs" : "by one workgroup", sum[0] / (reps - 8) / nsteps * tick_ns);
    printf("dependent load, line not written since (first touch): %.0f ns per load\n", sum[1] / (reps - 8) / nsteps * tick_ns);
    printf("dependent load, same lines again inside the kernel  : %.0f ns per load\n", sum[2] / (reps - 8) / nsteps * tick_ns);
    printf("%u independent loads of untouched lines, one wait   : %.0f ns in all\n", nsteps, sum[3] / (reps - 8) * tick_ns);
    }
    {
        const size_t far_words = (2u << 20) / 4 + (64u << 10) / 4 + 64;   // 2 MB + 64 KB + 256 B
        const unsigned nfar = 100;
        uint32_t *far; CK(hipMalloc(&far, far_words * 4 * nfar));
        double s0 = 0, s1 = 0;
        for (int r = 0; r < 60; r++) {
            hipLaunchKernelGGL(k_produce_far, dim3(1), dim3(128), 0, 0, far, nfar, far_words);
            hipLaunchKernelGGL(k_chase_far, dim3(1), dim3(64), 0, 0, far, far_words, 64u, res);
            unsigned long long h[6]; CK(hipMemcpy(h, res, sizeof h, hipMemcpyDeviceToHost));
            if (r >= 4) { s0 += (double)h[0]; s1 += (double)h[1]; }
        }
        printf("dependent load, every step in another 2 MB page      : %.0f ns per load (again inside the kernel: %.0f ns)\n", s0 / 56 / 64 * tick_ns, s1 / 56 / 64 * tick_ns);
        double sb[3] = { 0, 0, 0 };
        for (int k = 0; k < 3; k++)
            for (int r = 0; r < 60; r++) {
                hipLaunchKernelGGL(k_produce, dim3(n / 256), dim3(256), 0, 0, fresh, n, stride, (unsigned)r);
                if (k == 0) hipLaunchKernelGGL(k_burst<1>, dim3(1), dim3(1024), 0, 0, fresh, res, out);
                else if (k == 1) hipLaunchKernelGGL(k_burst<4>, dim3(1), dim3(1024), 0, 0, fresh, res, out);
                else hipLaunchKernelGGL(k_burst<8>, dim3(1), dim3(1024), 0, 0, fresh, res, out);
                unsigned long long h[6]; CK(hipMemcpy(h, res, sizeof h, hipMemcpyDeviceToHost));
                if (r >= 4) sb[k] += (double)h[0];
            }
        for (unsigned nwg = 1; nwg <= 256; nwg *= 4) {
            double sw = 0;
            for (int r = 0; r < 40; r++) {
                hipLaunchKernelGGL(k_produce, dim3(n / 256), dim3(256), 0, 0, fresh, n, stride, (unsigned)r);
                hipLaunchKernelGGL(k_burst<8>, dim3(nwg), dim3(1024), 0, 0, fresh, res, out);
                unsigned long long h[6]; CK(hipMemcpy(h, res, sizeof h, hipMemcpyDeviceToHost));
                if (r >= 4) sw += (double)h[0];
            }
            printf("  %3u such workgroups at once (each its own 32 KB), workgroup 0: %.0f ns\n", nwg, sw / 36 * tick_ns);
        }
        for (unsigned xcd = 0; xcd < 8; xcd += 1) {
            double sw = 0;
            for (int r = 0; r < 40; r++) {
                hipLaunchKernelGGL(k_produce_xcd, dim3(64), dim3(1024), 0, 0, fresh, xcd, (unsigned)r);
                hipLaunchKernelGGL(k_burst<8>, dim3(1), dim3(1024), 0, 0, fresh, res, out);
                unsigned long long h[6]; CK(hipMemcpy(h, res, sizeof h, hipMemcpyDeviceToHost));
                if (r >= 4) sw += (double)h[0];
            }
            printf("  8-word burst by workgroup 0 of lines produced by workgroups with id %% 8 == %u: %.0f ns\n", xcd, sw / 36 * tick_ns);
        }
        printf("1024-thread workgroup, 1 / 4 / 8 coalesced words per thread from fresh lines to LDS + barrier: %.0f / %.0f / %.0f ns\n", sb[0] / 56 * tick_ns, sb[1] / 56 * tick_ns, sb[2] / 56 * tick_ns);
    }
    for (unsigned nt = 64; nt <= 1024; nt *= 4) {
        double a = 0, b2 = 0;
        for (int r = 0; r < 30; r++) {
            unsigned long long h[6];
            hipLaunchKernelGGL(k_barriers, dim3(1), dim3(nt), 0, 0, 32u, res, out);
            CK(hipMemcpy(h, res, sizeof h, hipMemcpyDeviceToHost)); if (r >= 2) a += (double)h[0];
            hipLaunchKernelGGL(k_barriers_st, dim3(1), dim3(nt), 0, 0, 32u, res, out, fresh);
            CK(hipMemcpy(h, res, sizeof h, hipMemcpyDeviceToHost)); if (r >= 2) b2 += (double)h[0];
        }
        printf("workgroup of %4u threads: LDS write, barrier, LDS read, barrier: %.0f ns per round; with a global store in front of the barrier: %.0f ns\n", nt, a / 28 / 32 * tick_ns, b2 / 28 / 32 * tick_ns);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int w = 0; w < 2; w++) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("empty kernel, back to back on one stream            : %.2f us per launch\n", ms * 1000.0 / 2000);
    for (int w = 0; w < 2; w++) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(k_oneload, dim3(1), dim3(64), 0, 0, old, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("one load + one store kernel, back to back           : %.2f us per launch\n", ms * 1000.0 / 2000);
    for (int w = 0; w < 2; w++) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(k_produce, dim3(256), dim3(256), 0, 0, fresh, n, stride, (unsigned)i);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("256-workgroup store kernel, back to back            : %.2f us per launch\n", ms * 1000.0 / 2000);
    return 0;
}
