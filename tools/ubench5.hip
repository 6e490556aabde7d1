// tools/ubench5.hip -- what does a loop back-edge cost a lone wave, as a function of the loop body size?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))
template <int BODY> __device__ void body(uint32_t &a, uint32_t b)
{
    if (BODY == 16) asm volatile(R16("v_add3_u32 %0, %0, %1, %1\n") : "+v"(a) : "v"(b));
    if (BODY == 64) asm volatile(R64("v_add3_u32 %0, %0, %1, %1\n") : "+v"(a) : "v"(b));
    if (BODY == 256) asm volatile(R256("v_add3_u32 %0, %0, %1, %1\n") : "+v"(a) : "v"(b));
    if (BODY == 1024) asm volatile(R256("v_add3_u32 %0, %0, %1, %1\n") R256("v_add3_u32 %0, %0, %1, %1\n") R256("v_add3_u32 %0, %0, %1, %1\n") R256("v_add3_u32 %0, %0, %1, %1\n") : "+v"(a) : "v"(b));
}
template <int BODY> __global__ void k(uint64_t *out, uint32_t *sink, int iters)
{
    uint32_t a = threadIdx.x, b = 3;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) body<BODY>(a, b);
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = a;
}
template <int BODY> void run(uint64_t *d, uint32_t *s)
{
    const int iters = 4096 / BODY * 16;
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k<BODY>, dim3(1), dim3(64), 0, 0, d, s, iters); (void)hipDeviceSynchronize(); }
    uint64_t h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("body %4d x v_add3_u32 (8-byte VOP3, %5d bytes): %.2f cycles/instr, %.1f cycles per iteration beyond 4.16/instr\n", BODY, BODY * 8,
           (double)h / ((double)iters * BODY), (double)h / iters - 4.16 * BODY);
}
int main()
{
    uint64_t *d; uint32_t *s; (void)hipMalloc(&d, 64); (void)hipMalloc(&s, 4096);
    run<16>(d, s); run<64>(d, s); run<256>(d, s); run<1024>(d, s);
    return 0;
}
