// tools/ubench2.hip -- do two waves that share a SIMD slow each other down?  5 waves of one workgroup run the same
// dependent VALU chain; each wave reports its own cycle count and its SIMD id (HW_ID register).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))
__global__ void k(uint64_t *out, uint32_t *sink, int mode)
{
    const int wave = threadIdx.x >> 6;
    uint32_t a = threadIdx.x, b = 3;
    uint32_t hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    __syncthreads();
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 16; it++) {
        if (mode == 0) asm volatile(R256("v_add_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        else if (mode == 1) asm volatile(R256("v_max_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\nv_add_u32 %1, %1, %1\nv_add_u32 %1, %1, %1\n") : "+v"(a), "+v"(b));
        else asm volatile(R256("v_mul_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) { out[wave * 2] = t1 - t0; out[wave * 2 + 1] = hwid; }
    sink[threadIdx.x] = a + b;
}
int main()
{
    uint64_t *d; uint32_t *s;
    (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&s, 4096 * 4);
    for (int mode = 0; mode < 3; mode++)
        for (int waves : {1, 4, 5, 8}) {
            for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, d, s, mode); (void)hipDeviceSynchronize(); }
            uint64_t h[64]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            printf("mode %d, %d waves:", mode, waves);
            for (int w = 0; w < waves; w++) printf("  w%d simd%llu %.2f", w, (unsigned long long)((h[2 * w + 1] >> 4) & 3), (double)h[2 * w] / (16.0 * 256 * (mode == 1 ? 3 : 1)));
            printf("  (cycles per instruction)\n");
        }
    return 0;
}
