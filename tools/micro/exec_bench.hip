// Microbenchmark (developer tool), round 5: does a wave64 VALU instruction cost less when part of EXEC is off?
// 42 % of the lanes of an executed compositing body are valid, and 38 % of the valid bodies have all their valid pixels in
// one half of the 8x8 block (profiles/r04o_lane_packing_counters.txt) - if an all-inactive half (or quarter) of a
// wave were skipped by the vector pipe, masking invalid lanes with EXEC instead of computing them with alpha = 0
// would pay.  Every wave brackets a loop of 64 v_fma_f32 (16 independent registers) with s_memtime under a given EXEC
// mask; occupancy pinned with dynamic LDS.  cycles / instruction / SIMD = d(memtime) * waves per SIMD / instructions.
// build: hipcc --offload-arch=gfx950 -O3 -o exec_bench tools/micro/exec_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void k(float* out, unsigned long long* times, int iters, unsigned long long mask, float s0) {
    extern __shared__ float lds_pad[];
    float a[16];
    const float t = (float)threadIdx.x * 1e-3f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = t + i;
    float m = s0 + t * 1e-9f, c = 0.25f;
    asm volatile("" : "+v"(m), "+v"(c));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_mov_b64 exec, %0" : : "s"(mask) : "exec");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#define OPX(i)                                                                                            \
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));          \
            else if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));                              \
            else asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            REP16(OPX)
#undef OPX
        }
    }
    asm volatile("s_mov_b64 exec, -1" : : : "exec");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63) == 0) times[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
    (void)lds_pad;
}

template <int OP>
static void run(const char* name, unsigned long long mask, int waves_per_simd) {
    // occupancy: W workgroups of 256 threads (one wave per SIMD) per CU, capped by W equal shares of the LDS
    const int iters = 2000, cus = 256, threads = 256;
    const size_t lds = 160 * 1024 / waves_per_simd - 1024;
    float* out; unsigned long long* times;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&times, 8 * cus * 64);
    (void)hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = cus * waves_per_simd, nw = grid * threads / 64;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(threads), lds, 0, out, times, iters, mask, 1.0001f);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(threads), lds, 0, out, times, iters, mask, 1.0001f);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nw);
    (void)hipMemcpy(h.data(), times, 8 * nw, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cyc = (double)h[nw / 2] / (iters * 64.0);
    // wave64 instructions per SIMD per microsecond from the launch's wall time (1024 SIMDs)
    const double rate = (double)nw * iters * 64.0 / (ms * 1e3) / 1024.0;
    printf("%-10s exec %016llx  %d waves/SIMD: %6.2f cycles per instruction per wave (%5.2f / waves), launch %.3f ms = %.0f "
           "instructions per SIMD per us\n", name, mask, waves_per_simd, cyc, cyc / waves_per_simd, ms, rate);
    (void)hipFree(out); (void)hipFree(times);
}

int main() {
    const unsigned long long masks[] = {~0ull, 0x00000000ffffffffull, 0x000000000000ffffull};
    for (int wps : {1, 2, 3, 4, 5, 6, 8})
        for (unsigned long long m : masks) run<0>("v_fma_f32", m, wps);
    for (int wps : {4, 5, 8}) run<1>("v_exp_f32", ~0ull, wps);
    for (int wps : {4, 5, 8}) run<2>("dpp add", ~0ull, wps);
    return 0;
}
