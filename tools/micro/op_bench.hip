// SUPERSEDED by op_bench2.hip (round 4): this one divides WALL time by a nominal 2.4 GHz, and the shader clock under
// these loads is 1.9 - 2.1 GHz - its 3.2 "cycles" for v_fma_f32 are 1.94 real ones.  Kept for the round-3 tables.
// Microbenchmark (developer tool): issue cost of the VALU instructions the compositing kernels are made of,
// in SIMD cycles per wave64 instruction (nominal 2.4 GHz), 16 independent instances per loop iteration,
// 16 waves per SIMD.  build: hipcc --offload-arch=gfx950 -O3 -o op_bench tools/micro/op_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s0, float s1) {
    float a[16];
    const float t = (float)threadIdx.x * 1e-3f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = t + i;
    float m = s0 + t * 1e-9f, c = s1;
    asm volatile("" : "+v"(m), "+v"(c));
    unsigned long long sm = threadIdx.x & 1 ? 0x5555555555555555ull : 0xAAAAAAAAAAAAAAAAull;
    sm = __builtin_amdgcn_readfirstlane((unsigned)sm) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(sm >> 32)) << 32);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 5) asm volatile("s_mov_b64 vcc, %0" : : "s"(sm) : "vcc");       // one scalar move per 16 selects
#define OP(i)                                                                                                   \
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));                    \
        else if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(s0), "v"(c));              \
        else if (MODE == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));                           \
        else if (MODE == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                           \
        else if (MODE == 4) asm volatile("v_add_f32 %0, 0x3f800347, %0" : "+v"(a[i]));                            \
        else if (MODE == 5) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));              \
        else if (MODE == 6) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(sm));          \
        else if (MODE == 7) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");               \
        else if (MODE == 8) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));                                        \
        else if (MODE == 9) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));                                        \
        else if (MODE == 10) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                          \
        else if (MODE == 11) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(m));                              \
        else if (MODE == 12) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i])); \
        else if (MODE == 13) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a[i])); \
        else if (MODE == 14) asm volatile("v_sub_f32 %0, |%0|, |%1|" : "+v"(a[i]) : "v"(c));                      \
        else if (MODE == 15) asm volatile("v_fma_f32 %0, -%0, %1, %1" : "+v"(a[i]) : "v"(m));                     \
        else if (MODE == 16) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&a[i & ~1]) : "v"(*(double*)&m), "v"(*(double*)&c));
        REP16(OP)
#undef OP
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
    const int blocks = 4096, iters = 1000;
    float* out; (void)hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const char* names[16] = {"v_fma_f32 (VGPR operands)", "v_fma_f32 (one SGPR operand)", "v_mul_f32", "v_add_f32", "v_add_f32 (literal)",
                             "v_cndmask_b32 (vcc)", "v_cndmask_b32 (SGPR pair)", "v_cmp_le_f32 -> vcc", "v_exp_f32", "v_rcp_f32",
                             "v_max_f32", "v_mov_b32", "v_add_f32_dpp quad_perm", "v_add_f32_dpp row_ror:4", "v_sub_f32 |a|,|b|",
                             "v_fma_f32 with neg modifier"};
#define RUN(M) case M: hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-7f); break;
    for (int mode = 0; mode < 16; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(a);
            switch (mode) { RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) }
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
        }
        const double waves = (double)blocks * 4;
        printf("%-32s %.2f SIMD-cycles per instruction\n", names[mode], best * 1e-3 * 2.4e9 * 1024 / (waves * iters * 16));
    }
    return 0;
}
