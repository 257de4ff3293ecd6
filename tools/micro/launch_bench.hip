// Microbenchmark (developer tool): host cost of the HIP calls a frame is made of.
// build: hipcc --offload-arch=gfx950 -O3 -o launch_bench tools/micro/launch_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Cam { float a[4]; int b[6]; float c[2]; };
__global__ void k_small(int n, float* p) { if (n < 0) p[0] = 1.f; }
__global__ void k_args(Cam cam, int n, const float* a, const float* b, const float* c, const float* d, float* e,
                       float* f, int* g, int* h, float* p) { if (n < 0) p[0] = cam.a[0]; }
template <typename F> double per_call_us(F f, int reps, hipStream_t s) {
    for (int i = 0; i < 50; ++i) f();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) f();
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    float* d; hipMalloc(&d, 1 << 20);
    int* host; hipHostMalloc(&host, 64);
    Cam cam{};
    printf("empty kernel, 2 args          : %.2f us/launch\n", per_call_us([&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, 1, d); }, 2000, s));
    printf("kernel, 48-byte struct + 10 args: %.2f us/launch\n", per_call_us([&] { hipLaunchKernelGGL(k_args, dim3(1), dim3(64), 0, s, cam, 1, d, d, d, d, d, d, (int*)d, (int*)d, d); }, 2000, s));
    printf("big grid kernel (4096 blocks)  : %.2f us/launch\n", per_call_us([&] { hipLaunchKernelGGL(k_small, dim3(4096), dim3(256), 0, s, 1, d); }, 2000, s));
    printf("hipMemsetAsync 4 B             : %.2f us/call\n", per_call_us([&] { hipMemsetAsync(d, 0, 4, s); }, 2000, s));
    printf("hipMemsetAsync 1 MB            : %.2f us/call\n", per_call_us([&] { hipMemsetAsync(d, 0, 1 << 20, s); }, 2000, s));
    printf("hipMemcpyAsync D2H 4 B (pinned): %.2f us/call\n", per_call_us([&] { hipMemcpyAsync(host, d, 4, hipMemcpyDeviceToHost, s); }, 2000, s));
    hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming);
    printf("hipEventRecord                 : %.2f us/call\n", per_call_us([&] { hipEventRecord(e, s); }, 2000, s));
    return 0;
}
