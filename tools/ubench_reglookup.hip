// ubench_reglookup.hip -- a dependent table chase out of REGISTERS next to the same chase through LDS (gfx950, one wave alone on a CU).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_reglookup tools/ubench_reglookup.hip
//   lds      512 x 4-byte table in LDS, idx = table[idx]                                        (what the engines do today)
//   lane64    64-entry table in ONE VGPR (entry = lane), idx = readlane(v, idx)                 (the cheapest register form)
//   reg512   512-entry table in EIGHT VGPRs, idx = readlane(v[idx >> 6], idx & 63): all eight readlanes + a scalar select tree
//            (v_readlane cannot index the register dynamically; the alternative, s_set_gpr_idx_on + v_mov + v_readlane, has the same
//             three dependent stages: scalar index -> vector move -> readlane)
// Every chase is a single cycle through all entries (a random permutation), 4096 links, timed with s_memtime around the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k_chase(const uint32_t *perm512, const uint32_t *perm64, unsigned links, unsigned long long *res, uint32_t *sink)
{
    __shared__ uint32_t lds[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = perm512[i];
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = perm512[k * 64 + lane];
    const uint32_t v64 = perm64[lane];
    __syncthreads();
    // 1. LDS
    uint32_t idx = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (unsigned i = 0; i < links; i++) idx = lds[idx];
    unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = idx;
    // 2. one VGPR, 64 entries
    uint32_t s = 0;
    unsigned long long t2 = __builtin_readcyclecounter();
    for (unsigned i = 0; i < links; i++) s = __builtin_amdgcn_readlane(v64, s);
    unsigned long long t3 = __builtin_readcyclecounter();
    acc += s;
    // 3. eight VGPRs, 512 entries
    s = 0;
    unsigned long long t4 = __builtin_readcyclecounter();
    for (unsigned i = 0; i < links; i++) {
        const uint32_t l = s & 63u, r = s >> 6;
        const uint32_t a0 = __builtin_amdgcn_readlane(v[0], l), a1 = __builtin_amdgcn_readlane(v[1], l), a2 = __builtin_amdgcn_readlane(v[2], l), a3 = __builtin_amdgcn_readlane(v[3], l);
        const uint32_t a4 = __builtin_amdgcn_readlane(v[4], l), a5 = __builtin_amdgcn_readlane(v[5], l), a6 = __builtin_amdgcn_readlane(v[6], l), a7 = __builtin_amdgcn_readlane(v[7], l);
        const uint32_t b0 = (r & 1u) ? a1 : a0, b1 = (r & 1u) ? a3 : a2, b2 = (r & 1u) ? a5 : a4, b3 = (r & 1u) ? a7 : a6;
        const uint32_t c0 = (r & 2u) ? b1 : b0, c1 = (r & 2u) ? b3 : b2;
        s = (r & 4u) ? c1 : c0;
    }
    unsigned long long t5 = __builtin_readcyclecounter();
    acc += s;
    if (lane == 0) { res[0] = t1 - t0; res[1] = t3 - t2; res[2] = t5 - t4; sink[0] = acc; }
}

int main()
{
    std::mt19937 rng(7);
    auto cycle = [&](unsigned n) {          // a permutation that is one cycle: i -> next in a shuffled order
        std::vector<uint32_t> order(n), p(n);
        std::iota(order.begin(), order.end(), 0u);
        std::shuffle(order.begin() + 1, order.end(), rng);
        for (unsigned i = 0; i < n; i++) p[order[i]] = order[(i + 1) % n];
        return p;
    };
    std::vector<uint32_t> p512 = cycle(512), p64 = cycle(64);
    uint32_t *d512, *d64, *sink; unsigned long long *res;
    CK(hipMalloc(&d512, 2048)); CK(hipMalloc(&d64, 256)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&res, 24));
    CK(hipMemcpy(d512, p512.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(d64, p64.data(), 256, hipMemcpyHostToDevice));
    const unsigned links = 4096;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, d512, d64, links, res, sink);
        CK(hipDeviceSynchronize());
    }
    unsigned long long h[3]; uint32_t hs;
    CK(hipMemcpy(h, res, 24, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hs, sink, 4, hipMemcpyDeviceToHost));
    // s_memtime / readcyclecounter: shader clock cycles on gfx9
    printf("dependent lookup, one wave alone (cycles per link over %u links; checksum %u):\n", links, hs);
    printf("  512 x 4 B table in LDS (ds_read_b32)                      : %.1f\n", (double)h[0] / links);
    printf("  64 entries in one VGPR (v_readlane, lane = index)         : %.1f\n", (double)h[1] / links);
    printf("  512 entries in eight VGPRs (8 v_readlane + select tree)   : %.1f\n", (double)h[2] / links);
    return 0;
}
