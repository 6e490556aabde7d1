// tools/ubench3.hip -- cost of taken branches, VALU<->SALU mask traffic and s_nop for a lone wave (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
__global__ void k(uint64_t *out, uint32_t *sink)
{
    uint32_t a = threadIdx.x, b = 3, c = 5;
    uint64_t t0, t1;
    // 0: 64 x (16 valu) straight line = 1024 valu
    t0 = __builtin_readcyclecounter();
    asm volatile(R64(R16("v_add_u32 %0, %0, %1\n")) : "+v"(a) : "v"(b));
    t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[0] = t1 - t0;
    // 1: 64 x (16 valu + 1 taken branch)
    t0 = __builtin_readcyclecounter();
    asm volatile(R64(R16("v_add_u32 %0, %0, %1\n") "s_branch 1f\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\n1:\n") : "+v"(a) : "v"(b));
    t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[1] = t1 - t0;
    // 2: 64 x (16 valu + not-taken cbranch)
    t0 = __builtin_readcyclecounter();
    asm volatile("s_cmp_eq_u32 0, 1\n" R64(R16("v_add_u32 %0, %0, %1\n") "s_cbranch_scc1 1f\n1:\n") : "+v"(a) : "v"(b) : "scc");
    t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[2] = t1 - t0;
    // 3: 64 x (v_cmp -> s_and -> v_cndmask) dependent through the mask
    t0 = __builtin_readcyclecounter();
    asm volatile(R64(R4("v_cmp_lt_u32 vcc, %0, %1\ns_and_b64 vcc, vcc, exec\nv_cndmask_b32 %0, %0, %2, vcc\nv_add_u32 %0, 1, %0\n")) : "+v"(a) : "v"(b), "v"(c) : "vcc");
    t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[3] = t1 - t0;
    // 4: 64 x 4 x (v_cmp -> v_cndmask) no salu
    t0 = __builtin_readcyclecounter();
    asm volatile(R64(R4("v_cmp_lt_u32 vcc, %0, %1\nv_cndmask_b32 %0, %0, %2, vcc\nv_add_u32 %0, 1, %0\n")) : "+v"(a) : "v"(b), "v"(c) : "vcc");
    t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[4] = t1 - t0;
    // 5: s_nop 1 x 256
    t0 = __builtin_readcyclecounter();
    asm volatile(R64(R4("s_nop 1\n")));
    t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[5] = t1 - t0;
    // 6: v_cmp -> s_cbranch_vccz (not taken) x 256 with valu between
    t0 = __builtin_readcyclecounter();
    asm volatile(R64(R4("v_cmp_lt_u32 vcc, %0, %1\ns_cbranch_vccnz 1f\nv_add_u32 %0, 1, %0\n1:\n")) : "+v"(a) : "v"(b) : "vcc");
    t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[6] = t1 - t0;
    // 7: sdwa ops x 1024
    t0 = __builtin_readcyclecounter();
    asm volatile(R64(R16("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n")) : "+v"(a) : "v"(b));
    t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[7] = t1 - t0;
    // 8: v_mul_i32_i24 / cvt chain x 1024
    t0 = __builtin_readcyclecounter();
    asm volatile(R64(R16("v_mul_i32_i24 %0, %0, %1\n")) : "+v"(a) : "v"(b));
    t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[8] = t1 - t0;
    // 9: v_med3_i32 x 1024
    t0 = __builtin_readcyclecounter();
    asm volatile(R64(R16("v_med3_i32 %0, %0, %1, %2\n")) : "+v"(a) : "v"(b), "v"(c));
    t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[9] = t1 - t0;
    // 10: v_add3 / lshl_add x 1024
    t0 = __builtin_readcyclecounter();
    asm volatile(R64(R16("v_add3_u32 %0, %0, %1, %2\n")) : "+v"(a) : "v"(b), "v"(c));
    t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[10] = t1 - t0;
    // 11: v_cmp_sdwa -> sgpr then s_and x 256
    t0 = __builtin_readcyclecounter();
    asm volatile(R64(R4("v_cmp_ne_u32_sdwa s[10:11], %0, %1 src0_sel:BYTE_0 src1_sel:DWORD\ns_and_b64 s[10:11], s[10:11], exec\nv_cndmask_b32_e64 %0, %0, %2, s[10:11]\n")) : "+v"(a) : "v"(b), "v"(c) : "s10", "s11");
    t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[11] = t1 - t0;
    sink[threadIdx.x] = a;
}
int main()
{
    uint64_t *d; uint32_t *s;
    (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&s, 4096 * 4);
    const char *names[] = {"1024 valu straight", "1024 valu + 64 taken s_branch", "1024 valu + 64 untaken cbranch", "256x(v_cmp,s_and,v_cndmask,v_add)", "256x(v_cmp,v_cndmask,v_add)",
                           "256 x s_nop 1", "256x(v_cmp,cbranch_vcc untaken,v_add)", "1024 sdwa add", "1024 v_mul_i32_i24", "1024 v_med3_i32", "1024 v_add3_u32", "256x(v_cmp_sdwa->sgpr,s_and,v_cndmask)"};
    for (int r = 0; r < 3; r++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s); (void)hipDeviceSynchronize(); }
    uint64_t h[64]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int i = 0; i < 12; i++) printf("%-44s %8llu cycles\n", names[i], (unsigned long long)h[i]);
    return 0;
}
