// Microbenchmark (developer tool): issue cost of the cross-lane instructions a wave reduction can be built from,
// in SIMD cycles per wave64 instruction (nominal 2.4 GHz), 16 independent instances per loop iteration, 16 waves
// per SIMD.  build: hipcc --offload-arch=gfx950 -O3 -o xlane_bench tools/micro/xlane_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define REP8(X) X(0) X(2) X(4) X(6) X(8) X(10) X(12) X(14)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s0) {
    float a[16];
    const float t = (float)threadIdx.x * 1e-3f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = t + i;
    int addr = ((threadIdx.x ^ 16) & 63) * 4;
    asm volatile("" : "+v"(addr));
    for (int it = 0; it < iters; ++it) {
#define OP(i)                                                                                                     \
        if (MODE == 0) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 1]));                   \
        else if (MODE == 1) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 1]));
#define OP1(i)                                                                                                    \
        if (MODE == 2) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(12)" : "+v"(a[i]) : "v"(addr)); \
        else if (MODE == 3) asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));  \
        else if (MODE == 4) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(a[i])); \
        else if (MODE == 5) asm volatile("v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xa" : "+v"(a[i])); \
        else if (MODE == 6) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a[i])); \
        else if (MODE == 7) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));                                      \
        else if (MODE == 8) { int s_; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s_) : "v"(a[i])); asm volatile("" :: "s"(s_)); } \
        else if (MODE == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
        if (MODE <= 1) { REP8(OP) REP8(OP) } else { REP16(OP1) }
#undef OP
#undef OP1
        if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)");
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
    const char* names[10] = {"v_permlane32_swap_b32", "v_permlane16_swap_b32", "ds_bpermute_b32", "v_mov_b32_dpp row_ror:8",
                             "v_add_f32_dpp row_ror:8 bank_mask:0xc", "v_add_f32_dpp row_half_mirror bank_mask:0xa",
                             "v_add_f32_dpp row_bcast:15 row_mask:0xa", "v_add_f32 (reference)", "v_readlane_b32", "v_cndmask_b32 vcc (VOP2, two VGPRs)"};
#define RUN(M) case M: hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f); break;
    for (int mode = 0; mode < 10; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(a);
            switch (mode) { RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) }
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
        }
        const double waves = (double)blocks * 4;
        printf("%-44s %.2f SIMD-cycles per instruction\n", names[mode], best * 1e-3 * 2.4e9 * 1024 / (waves * iters * 16));
    }
    return 0;
}
