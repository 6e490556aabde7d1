// tools/ubench.hip -- single-wave latency / issue-rate microbenchmarks for the primitives the row engine's serial
// chain is built from (gfx950).  Not part of the product; results are recorded in DESIGN.md.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench && ./tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))

#define TIMED(name, idx, N, BODY)                                      \
    {                                                                  \
        __syncthreads();                                               \
        uint64_t t0 = __builtin_readcyclecounter();                    \
        BODY;                                                          \
        uint64_t t1 = __builtin_readcyclecounter();                    \
        if (threadIdx.x == 0) { out[2 * idx] = t1 - t0; out[2 * idx + 1] = N; } \
    }

__global__ void ubench(uint64_t *out, uint32_t *sink, const uint32_t *chase)
{
    __shared__ uint32_t lds[1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = ((i * 4 + 64 * 4) & 4095);   // byte offset of next
    __syncthreads();
    uint32_t a = lane, b = lane * 3 + 1, c = 7, d = 9;
    float f = (float)lane, g = 1.0001f;

    // 0: dependent v_add_u32
    TIMED("dep v_add", 0, 256, asm volatile(R256("v_add_u32 %0, %0, %1\n") : "+v"(a) : "v"(b)));
    // 1: 4 independent v_add chains (1024 ops)
    TIMED("indep v_add x4", 1, 1024,
          asm volatile(R256("v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %4\nv_add_u32 %2, %2, %4\nv_add_u32 %3, %3, %4\n")
                       : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(lane)));
    // 2: dependent DPP max (row_ror:1)
    TIMED("dep v_max_u32_dpp", 2, 256, asm volatile(R256("s_nop 1\nv_max_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n") : "+v"(a)));
    // 3: dependent LDS pointer chase ds_read_b32
    {
        uint32_t p = lane * 4;
        TIMED("dep ds_read_b32", 3, 64, asm volatile(R64("ds_read_b32 %0, %0\ns_waitcnt lgkmcnt(0)\n") : "+v"(p)));
        a += p;
    }
    // 4: dependent ds_bpermute
    {
        uint32_t p = ((lane + 1) & 63) * 4, v = lane;
        TIMED("dep ds_bpermute", 4, 64, asm volatile(R64("ds_bpermute_b32 %0, %1, %0\ns_waitcnt lgkmcnt(0)\n") : "+v"(v) : "v"(p)));
        a += v;
    }
    // 5: readlane -> valu -> readlane chain
    {
        uint32_t v = lane; uint32_t s;
        TIMED("dep readlane+v_add", 5, 64, asm volatile(R64("v_readlane_b32 %1, %0, 5\ns_nop 3\nv_add_u32 %0, %1, %0\n") : "+v"(v), "=s"(s)));
        a += v + s;
    }
    // 6: SALU dependent chain
    {
        uint32_t s = 3;
        TIMED("dep s_add", 6, 256, asm volatile(R256("s_add_u32 %0, %0, 7\n") : "+s"(s)));
        a += s;
    }
    // 7: dependent float chain mul+trunc
    TIMED("dep v_mul_f32+v_trunc", 7, 512, asm volatile(R256("v_mul_f32 %0, %0, %1\nv_trunc_f32 %0, %0\n") : "+v"(f) : "v"(g)));
    // 8: dependent global load chase (L2/L1 hit)
    {
        uint64_t p = (uint64_t)chase;
        uint32_t off = lane * 4;
        TIMED("dep global_load", 8, 16, asm volatile(R16("global_load_dword %0, %0, %1\ns_waitcnt vmcnt(0)\n") : "+v"(off) : "s"(p)));
        a += off;
    }
    // 9: ds_add (no return) then dependent ds_read of same address
    {
        uint32_t p = lane * 4, v = 1, r;
        TIMED("ds_add+ds_read", 9, 64, asm volatile(R64("ds_add_u32 %1, %2\nds_read_b32 %0, %1\ns_waitcnt lgkmcnt(0)\nv_and_b32 %1, 0xffc, %0\n") : "=&v"(r), "+v"(p) : "v"(v)));
        a += r + p;
    }
    // 10: v_cvt chain
    TIMED("dep cvt i2f,f2i", 10, 512, asm volatile(R256("v_cvt_f32_i32 %0, %1\nv_cvt_i32_f32 %1, %0\n") : "+v"(f), "+v"(a)));
    // 11: v_cmp -> s_and_saveexec -> v_mov -> s_or (masked region)
    {
        uint32_t v = lane;
        TIMED("masked region", 11, 64, asm volatile(R64("v_cmp_eq_u32 vcc, 0, %1\ns_and_saveexec_b64 s[10:11], vcc\nv_add_u32 %0, 1, %0\ns_or_b64 exec, exec, s[10:11]\n") : "+v"(v) : "v"(lane) : "vcc", "s10", "s11"));
        a += v;
    }
    // 12: dependent ds_read_b32 broadcast (all lanes same address)
    {
        uint32_t p = 0, r;
        TIMED("dep ds_read_b32 bcast", 12, 64, asm volatile(R64("ds_read_b32 %0, %1\ns_waitcnt lgkmcnt(0)\nv_and_b32 %1, 0xffc, %0\n") : "=&v"(r), "+v"(p)));
        a += r;
    }
    // 13: independent VALU, 2 waves? (skipped) -- v_med3 / v_bfe dependent
    TIMED("dep v_bfe_i32+v_med3", 13, 512, asm volatile(R256("v_bfe_i32 %0, %0, 0, 8\nv_med3_i32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c)));
    // 14: v_readfirstlane -> s_mul -> v_mov
    {
        uint32_t v = lane, s;
        TIMED("readfirstlane+s_mul+v_add", 14, 64, asm volatile(R64("v_readfirstlane_b32 %1, %0\ns_mul_i32 %1, %1, 3\ns_nop 0\nv_add_u32 %0, %1, %0\n") : "+v"(v), "=s"(s)));
        a += v + s;
    }
    sink[threadIdx.x] = a + b + c + d + (uint32_t)f;
}

int main()
{
    uint64_t *d_out; uint32_t *d_sink, *d_chase;
    hipMalloc(&d_out, 64 * sizeof(uint64_t)); hipMalloc(&d_sink, 1024 * 4); hipMalloc(&d_chase, 4096);
    uint32_t h_chase[1024];
    for (int i = 0; i < 1024; i++) h_chase[i] = ((i * 4 + 256) & 4095);
    hipMemcpy(d_chase, h_chase, 4096, hipMemcpyHostToDevice);
    const char *names[] = {"dep v_add_u32", "indep v_add x4 (per op)", "dep v_max_u32_dpp(+s_nop1)", "dep ds_read_b32 chase", "dep ds_bpermute",
                           "readlane+s_nop3+v_add", "dep s_add_u32", "dep v_mul_f32|v_trunc (per op)", "dep global_load chase", "ds_add+ds_read+and",
                           "dep cvt i2f|f2i (per op)", "masked region (4 instr)", "dep ds_read_b64 bcast+and", "dep v_bfe|v_med3 (per op)", "readfirstlane+s_mul+nop+v_add"};
    for (int waves = 1; waves <= 2; waves++) {
        for (int rep = 0; rep < 2; rep++) {
            hipMemset(d_out, 0, 64 * 8);
            hipLaunchKernelGGL(ubench, dim3(1), dim3(64 * waves), 0, 0, d_out, d_sink, d_chase);
            hipDeviceSynchronize();
        }
        uint64_t h[64];
        hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
        printf("== %d wave(s) in the workgroup (wave 0 timed)\n", waves);
        for (int i = 0; i < 15; i++) printf("%-34s %8.2f cycles/op  (%llu cycles / %llu)\n", names[i], (double)h[2 * i] / (double)h[2 * i + 1], (unsigned long long)h[2 * i], (unsigned long long)h[2 * i + 1]);
    }
    return 0;
}
