// tools/ubench4.hip -- how many 320-thread workgroups does a CU really host, as a function of static LDS and scratch?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
template <int LDS_BYTES, int SCRATCH>
__global__ __launch_bounds__(320) void k(uint64_t *out, int spin)
{
    __shared__ uint32_t lds[LDS_BYTES / 4];
    volatile uint32_t priv[SCRATCH ? SCRATCH : 1];
    uint32_t hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint64_t t0 = wall_clock64();
    uint32_t a = threadIdx.x;
    for (int i = 0; i < spin; i++) { lds[(a + i) % (LDS_BYTES / 4)] = a; a = a * 1664525u + 1013904223u; if (SCRATCH) priv[a % SCRATCH] = a; }
    __syncthreads();
    uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = t0; out[blockIdx.x * 4 + 1] = t1; out[blockIdx.x * 4 + 2] = hwid | ((uint64_t)(xcc & 15) << 32); out[blockIdx.x * 4 + 3] = lds[a % (LDS_BYTES / 4)]; }
}
template <int L, int S> void run(const char *name)
{
    const int n = 1024;
    uint64_t *d; (void)hipMalloc(&d, n * 32);
    hipLaunchKernelGGL((k<L, S>), dim3(n), dim3(320), 0, 0, d, 20000);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> h(n * 4); (void)hipMemcpy(h.data(), d, n * 32, hipMemcpyDeviceToHost);
    // max overlap per (xcc, se, cu)
    int best = 0;
    for (int i = 0; i < n; i++) {
        int c = 0;
        for (int j = 0; j < n; j++) {
            bool same = ((h[i * 4 + 2] >> 32) == (h[j * 4 + 2] >> 32)) && (((h[i * 4 + 2] >> 8) & 0xff) == ((h[j * 4 + 2] >> 8) & 0xff)) && (((h[i*4+2] >> 13) & 7) == ((h[j*4+2] >> 13) & 7));
            if (same && h[j * 4] <= h[i * 4] && h[j * 4 + 1] > h[i * 4]) c++;
        }
        best = std::max(best, c);
    }
    uint64_t tmin = ~0ull, tmax = 0; for (int i = 0; i < n; i++) { tmin = std::min(tmin, h[i*4]); tmax = std::max(tmax, h[i*4+1]); }
    printf("%-28s max co-resident workgroups on one CU: %d   total %.2f ms\n", name, best, (tmax - tmin) / 100000.0);
    (void)hipFree(d);
}
int main()
{
    run<8192, 0>("LDS 8 KB");
    run<32768, 0>("LDS 32 KB");
    run<49152, 0>("LDS 48 KB");
    run<57344, 0>("LDS 56 KB");
    run<65536, 0>("LDS 64 KB");
    run<57344, 34>("LDS 56 KB + scratch 136 B");
    return 0;
}
