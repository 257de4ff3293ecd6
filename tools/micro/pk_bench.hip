// Microbenchmark (developer tool): issue rate of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 vs the scalar forms.
// build: hipcc --offload-arch=gfx950 -O3 -o pk_bench tools/micro/pk_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s0, float s1) {
    const float t = (float)threadIdx.x * 1e-3f;
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (f2){t + i, t - i};
    f2 m = (f2){s0, s1}, c = (f2){s1, s0};
    asm volatile("" : "+v"(m), "+v"(c));            // multiplicand / addend live in VGPRs in every mode
    float dep = t;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {            // 16 scalar FMAs (kept scalar: the asm barrier stops the SLP packer)
                float x = a[i].x, y = a[i].y;
                x = __builtin_fmaf(x, m.x, c.x);
                asm volatile("" : "+v"(x));
                y = __builtin_fmaf(y, m.y, c.y);
                asm volatile("" : "+v"(y));
                a[i].x = x; a[i].y = y;
            } else if (MODE == 1) {     // 8 packed FMAs (same flops)
                a[i] = __builtin_elementwise_fma(a[i], m, c);
            } else if (MODE == 2) {     // 8 packed FMAs with a broadcast scalar operand
                a[i] = __builtin_elementwise_fma(a[i], (f2){s0, s0}, c);
            } else if (MODE == 3) {     // 8 packed mul + 8 packed add (not contracted)
#pragma clang fp contract(off)
                a[i] = a[i] * m;
                asm volatile("" : "+v"(a[i]));
                a[i] = a[i] + c;
            } else {                    // 16 DEPENDENT scalar FMAs: latency of one VALU op
                dep = __builtin_fmaf(dep, m.x, c.x);
                asm volatile("" : "+v"(dep));
                dep = __builtin_fmaf(dep, m.y, c.y);
                asm volatile("" : "+v"(dep));
            }
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    r += dep;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
    const int blocks = 4096, iters = 2000;
    float* out; (void)hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const char* names[5] = {"16 x v_fma_f32", "8 x v_pk_fma_f32", "8 x v_pk_fma_f32 (bcast)", "8 x v_pk_mul + 8 x v_pk_add", "16 dependent v_fma_f32, 1 wave/SIMD"};
    for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(a);
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-7f); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-7f); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-7f); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-7f); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, iters, 1.0001f, 1e-7f); break;
            }
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
        }
        const double waves = (mode == 4 ? 1024.0 : (double)blocks * 4), per_iter_cycles = best * 1e-3 * 2.4e9 * 1024 / (waves * iters);
        printf("%-30s %.3f ms  = %.1f SIMD-cycles per iteration\n", names[mode], best, per_iter_cycles);
    }
    return 0;
}
