// ubench_gridsync.hip -- what an in-kernel grid-wide hand-over costs next to a kernel boundary (gfx950, one MI355X).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_gridsync tools/ubench_gridsync.hip
// G co-resident workgroups (G <= 256: one per CU at most) run R rounds of: every workgroup stores 256 words, signals (release at device scope),
// waits until all G have signalled (acquire), then reads 256 words that ANOTHER workgroup (blockIdx + G/2: mostly another XCD) stored in this
// round and checks them.  Reported: microseconds per round, from wall_clock64 in workgroup 0, and the number of wrong words (must be 0).
// The wait is bounded: a workgroup that spins for more than ~2 s gives up and the run reports it (no hang on a broken assumption).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NT>
__global__ __launch_bounds__(NT) void k_rounds(uint32_t *data, uint32_t *counter, unsigned rounds, unsigned long long *res, uint32_t *bad, int mode)
{
    const unsigned G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    __shared__ uint32_t give_up;
    if (t == 0) give_up = 0;
    __syncthreads();
    uint32_t wrong = 0;
    unsigned long long t0 = 0;
    for (unsigned r = 0; r < rounds; r++) {
        if (r == 8 && b == 0 && t == 0) t0 = wall_clock64();
        if (t < 256) data[(size_t)b * 256 + t] = r * 1000003u + b * 257u + t;
        if (mode == 0) {
            // counter barrier: one release-add per workgroup, everybody polls the same word
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            if (t == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (r + 1) * G) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 24)) { give_up = 1; break; }
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else {
            // flag per producer: a consumer waits only for the workgroup whose data it reads
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            if (t == 0) {
                __hip_atomic_store(counter + 64 * (size_t)b, r + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned p = (b + G / 2) % G;
                unsigned spins = 0;
                while (__hip_atomic_load(counter + 64 * (size_t)p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < r + 1) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 24)) { give_up = 1; break; }
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        if (give_up) break;
        const unsigned p = (b + G / 2) % G;
        if (t < 256) wrong += data[(size_t)p * 256 + t] != r * 1000003u + p * 257u + t;
        __syncthreads();        // (nobody overwrites its slot before its reader is done?  the NEXT round's barrier orders that: a reader of round r
                                //  has read before it signals round r + 1, and a writer of round r + 1 ... writes BEFORE the barrier: so double-buffer)
        data += (size_t)G * 256 * ((r & 1) ? -1 : 1);
    }
    if (wrong || give_up) atomicAdd(bad, wrong + (give_up ? 1000000u : 0u));
    if (b == 0 && t == 0) res[0] = wall_clock64() - t0;
}

int main()
{
    uint32_t *data, *counter, *bad; unsigned long long *res;
    CK(hipMalloc(&data, 2 * 256 * 256 * 4)); CK(hipMalloc(&counter, 64 * 256 * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&res, 8));
    const unsigned rounds = 2008;
    for (int mode = 0; mode < 2; mode++)
        for (int nt : { 256, 1024 })
            for (unsigned G : { 20u, 40u, 80u, 160u, 256u }) {
                CK(hipMemset(counter, 0, 64 * 256 * 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(data, 0, 2 * 256 * 256 * 4));
                if (nt == 256) hipLaunchKernelGGL(k_rounds<256>, dim3(G), dim3(256), 0, 0, data, counter, rounds, res, bad, mode);
                else hipLaunchKernelGGL(k_rounds<1024>, dim3(G), dim3(1024), 0, 0, data, counter, rounds, res, bad, mode);
                CK(hipDeviceSynchronize());
                unsigned long long h; uint32_t hb;
                CK(hipMemcpy(&h, res, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
                printf("%s, %4d threads, %3u workgroups: %.2f us per round (store 1 KB, hand over, read 1 KB of another workgroup), wrong words %u\n",
                       mode ? "flag per producer" : "counter barrier  ", nt, G, (double)h * 10.0 / 1e3 / (rounds - 8), hb);
            }
    return 0;
}
