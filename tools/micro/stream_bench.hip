// Microbenchmark (developer tool): which shape of a streaming read reaches the HBM read ceiling of an MI355X
// (guide: ~6.3 TB/s achievable of 8 TB/s).  build: hipcc --offload-arch=gfx950 -O3 -o stream_bench tools/micro/stream_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: grid-stride, U loads of 16 B in flight per lane.  MODE 1: every block owns a contiguous span.
// MODE 2: as 0 with non-temporal loads.
template <int MODE, int U, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const float4* __restrict__ src, size_t n16, float* __restrict__ sink) {
    float4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 1) {
        const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
        const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
        for (size_t i = b0 + threadIdx.x; i + (size_t)(U - 1) * THREADS < b1; i += (size_t)U * THREADS) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 v = src[i + (size_t)u * THREADS];
                acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
            }
        }
    } else {
        const size_t stride = (size_t)gridDim.x * THREADS;
        for (size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x; i + (size_t)(U - 1) * stride < n16; i += (size_t)U * stride) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float4 v;
                if (MODE == 2) {
                    const float* p = reinterpret_cast<const float*>(src + i + (size_t)u * stride);
                    v.x = __builtin_nontemporal_load(p); v.y = __builtin_nontemporal_load(p + 1);
                    v.z = __builtin_nontemporal_load(p + 2); v.w = __builtin_nontemporal_load(p + 3);
                } else {
                    v = src[i + (size_t)u * stride];
                }
                acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
            }
        }
    }
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) t += (acc[u].x + acc[u].y) + (acc[u].z + acc[u].w);
    if (t == 123456.789f) sink[0] = t;
}

template <int MODE, int U, int THREADS>
void run(const char* name, const float4* buf, size_t n16, float* sink, int grid) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k<MODE, U, THREADS>), dim3(grid), dim3(THREADS), 0, 0, buf, n16, sink);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("%-52s grid %6d  %.0f GB/s\n", name, grid, 16.0 * n16 / (best * 1e-3) / 1e9);
}

int main() {
    const size_t bytes = (size_t)2 << 30, n16 = bytes / 16;
    float4* buf; float* sink;
    (void)hipMalloc(&buf, bytes); (void)hipMalloc(&sink, 64);
    (void)hipMemset(buf, 0, bytes);
    for (int g : {2048, 4096, 8192, 16384}) {
        run<0, 4, 256>("grid-stride, 4 x 16 B, 256 threads", buf, n16, sink, g);
        run<0, 8, 256>("grid-stride, 8 x 16 B, 256 threads", buf, n16, sink, g);
        run<0, 16, 256>("grid-stride, 16 x 16 B, 256 threads", buf, n16, sink, g);
    }
    run<0, 8, 512>("grid-stride, 8 x 16 B, 512 threads", buf, n16, sink, 2048);
    run<0, 8, 1024>("grid-stride, 8 x 16 B, 1024 threads", buf, n16, sink, 1024);
    run<0, 4, 1024>("grid-stride, 4 x 16 B, 1024 threads", buf, n16, sink, 2048);
    for (int g : {2048, 8192, 32768}) {
        run<1, 4, 256>("contiguous span per block, 4 x 16 B", buf, n16, sink, g);
        run<1, 8, 256>("contiguous span per block, 8 x 16 B", buf, n16, sink, g);
    }
    run<2, 8, 256>("grid-stride, non-temporal 4-byte loads x 8", buf, n16, sink, 4096);
    return 0;
}
