// Microbenchmark (developer tool): what does a wave-uniform 48-byte record cost per iteration when it
// comes (a) from LDS with broadcast ds_read_b128s, (b) from global memory through scalar loads?
// build: hipcc --offload-arch=gfx950 -O3 -o lds_bench tools/micro/lds_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kWaves = 4, kThreads = 256;

template <int MODE>
__global__ __launch_bounds__(kThreads) void k(const float4* __restrict__ recs, float* __restrict__ out, int iters) {
    __shared__ float4 lds[kWaves][64 * 3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4* l = lds[wave];
    for (int i = lane; i < 64 * 3; i += 64) l[i] = recs[(blockIdx.x * kWaves + wave) % 64 * 192 + i];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const float X = (float)(lane & 7) - 3.5f, Y = (float)(lane >> 3) - 3.5f, XX = X * X, XY = X * Y, YY = Y * Y;
    float T = 1.0f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
    int fidx = 0;
    const float4* g = recs + (size_t)((blockIdx.x * kWaves + wave) % 64) * 192;
    float mine[12];                                   // MODE 5: lane j keeps record j in registers
    {
        const float4 m0 = g[3 * lane], m1 = g[3 * lane + 1], m2 = g[3 * lane + 2];
        mine[0] = m0.x; mine[1] = m0.y; mine[2] = m0.z; mine[3] = m0.w; mine[4] = m1.x; mine[5] = m1.y;
        mine[6] = m1.z; mine[7] = m1.w; mine[8] = m2.x; mine[9] = m2.y; mine[10] = m2.z; mine[11] = m2.w;
    }
    for (int it = 0; it < iters; ++it) {
        const int j = it & 63;
        float4 r0, r1, r2;
        if (MODE == 0 || MODE == 2) { r0 = l[3 * j]; r1 = l[3 * j + 1]; r2 = l[3 * j + 2]; }
        else if (MODE == 3 || MODE == 4) {           // uniform address in global memory -> scalar loads
            typedef float f4 __attribute__((ext_vector_type(4)));
            typedef const __attribute__((address_space(4))) f4* cptr;          // constant address space -> s_load
            const unsigned long long addr = (unsigned long long)(g + 3 * j);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr), hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
            cptr p = (cptr)(((unsigned long long)hi << 32) | lo);
            const f4 t0 = p[0], t1 = p[1], t2 = p[2];
            r0 = make_float4(t0.x, t0.y, t0.z, t0.w); r1 = make_float4(t1.x, t1.y, t1.z, t1.w);
            r2 = make_float4(t2.x, t2.y, t2.z, t2.w);
        } else if (MODE == 5) {                       // v_readlane of the staging lane's registers -> SGPRs
            const int sj = __builtin_amdgcn_readfirstlane(j);
            float t[12];
#pragma unroll
            for (int c = 0; c < 11; ++c) t[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine[c]), sj));
            r0 = make_float4(t[0], t[1], t[2], t[3]); r1 = make_float4(t[4], t[5], t[6], t[7]);
            r2 = make_float4(t[8], 0.f, t[10], 0.f);
        } else { r0 = make_float4(0.1f, 0.01f * it, 0.02f, 0.03f); r1 = make_float4(0.01f, 0.02f, 0.5f, 0.4f); r2 = make_float4(0.3f, 0.f, __int_as_float(it), 0.f); }
        if (MODE == 2 || MODE == 4) { a0 += r0.x + r1.y + r2.z; continue; }      // loads only
        float s = __builtin_fmaf(X, r0.y, r0.x);
        s = __builtin_fmaf(Y, r0.z, s); s = __builtin_fmaf(XX, r0.w, s);
        s = __builtin_fmaf(XY, r1.x, s); s = __builtin_fmaf(YY, r1.y, s);
        const float a = __builtin_amdgcn_exp2f(-s);
        const bool ok = a >= 0.00392f;
        const float ae = ok ? a : 0.0f;
        const float nT = __builtin_fmaf(-ae, T, T);
        const bool stop = (nT <= 1e-4f) & ok;
        const float Tn = stop ? -__builtin_fabsf(T) : nT;
        const float vis = __builtin_fabsf(T) - __builtin_fabsf(Tn);
        a0 = __builtin_fmaf(r1.z, vis, a0); a1 = __builtin_fmaf(r1.w, vis, a1); a2 = __builtin_fmaf(r2.x, vis, a2);
        fidx = (ok & !stop) ? __float_as_int(r2.z) : fidx;
        T = Tn;
    }
    out[blockIdx.x * kThreads + threadIdx.x] = a0 + a1 + a2 + T + (float)fidx;
}

int main() {
    const int blocks = 2040, iters = 600;
    float4* recs; float* out;
    std::vector<float4> h(64 * 192);
    for (size_t i = 0; i < h.size(); ++i) h[i] = make_float4(0.5f + 0.001f * (i % 97), 0.01f, 0.02f, 0.03f);
    hipMalloc(&recs, h.size() * sizeof(float4)); hipMalloc(&out, blocks * kThreads * 4);
    hipMemcpy(recs, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char* names[6] = {"lds b128 x3 + body", "body only (constants)", "lds b128 x3 only", "scalar loads + body", "scalar loads only", "v_readlane x11 + body"};
    for (int mode = 0; mode < 6; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(a);
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(kThreads), 0, 0, recs, out, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(kThreads), 0, 0, recs, out, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(kThreads), 0, 0, recs, out, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(kThreads), 0, 0, recs, out, iters); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(kThreads), 0, 0, recs, out, iters); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(kThreads), 0, 0, recs, out, iters); break;
            }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
        }
        const double recs_total = (double)blocks * kWaves * iters;
        printf("%-26s %.3f ms  = %.1f SIMD-cycles per record per wave-slot (1024 SIMDs @2.4GHz), %.1f CU-clk per record\n",
               names[mode], best, best * 1e-3 * 2.4e9 * 1024 / recs_total, best * 1e-3 * 2.4e9 * 256 / recs_total);
    }
    return 0;
}
