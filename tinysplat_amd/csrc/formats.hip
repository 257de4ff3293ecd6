// formats.hip - the on-disk PLY row layout of tinysplat's export.
//
// Reference: /root/reference/tinysplat/splatting/model_gaussian.py:330-361 (export_ply) builds, per
// Gaussian, the float32 record
//     x y z | nx ny nz (zeros) | f_dc_0..2 | f_rest_0..3(K-1)-1 | opacity | scale_0..2 | rot_0..3
// where f_rest is CHANNEL-major (colors_rest[N, K-1, 3] transposed to [N, 3, K-1], :351), through
// seven device->host copies, a numpy concatenate and a Python-level tuple per row.  Here one kernel
// interleaves the six tensors into the [N, 17 + 3 (K-1)] record array in device memory (and one
// kernel de-interleaves it again for import); the host then moves a single contiguous buffer.
// HBM streaming: 4 W bytes read + 4 W bytes written per Gaussian (W = 62 floats at SH degree 3).
#include <hip/hip_runtime.h>

#include "../../include/tinysplat_hip.h"

namespace {

constexpr int kThreads = 256;
inline int launch_status() { return (int)hipGetLastError(); }

struct PlyTensors {
    float* means; float* dc; float* rest; float* opac; float* scales; float* quats;
    int k_rest;
};

// address of the tensor element that record column c of row r maps to; nullptr for the normals
__device__ __forceinline__ float* element(const PlyTensors& t, long long r, int c) {
    if (c < 3) return t.means + 3 * r + c;
    if (c < 6) return nullptr;
    if (c < 9) return t.dc + 3 * r + (c - 6);
    const int nrest = 3 * t.k_rest;
    if (c < 9 + nrest) {
        const int j = c - 9, ch = j / t.k_rest, k = j - ch * t.k_rest;      // channel-major on disk
        return t.rest + (3 * r * t.k_rest) + 3 * k + ch;
    }
    c -= 9 + nrest;
    if (c < 1) return t.opac + r;
    if (c < 4) return t.scales + 3 * r + (c - 1);
    return t.quats + 4 * r + (c - 4);
}

template <bool PACK>
__global__ __launch_bounds__(kThreads) void ply_rows_kernel(long long n, PlyTensors t,
                                                            float* __restrict__ rows) {
    const int w = 17 + 3 * t.k_rest;
    const long long total = n * w;
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < total; e += stride) {
        const long long r = e / w;
        const int c = (int)(e - r * w);
        float* p = element(t, r, c);
        if (PACK) rows[e] = p ? *p : 0.0f;
        else if (p) *p = rows[e];
    }
}

template <bool PACK>
int run(int32_t n, int32_t k_rest, float* means, float* dc, float* rest, float* opac, float* scales,
        float* quats, float* rows, void* stream) {
    if (n < 0 || k_rest < 0) return TS_E_BADARG;
    if (n == 0) return 0;
    if (!means || !dc || (k_rest > 0 && !rest) || !opac || !scales || !quats || !rows)
        return TS_E_BADARG;
    const PlyTensors t{means, dc, rest, opac, scales, quats, k_rest};
    const long long total = (long long)n * (17 + 3 * k_rest);
    long long blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(ply_rows_kernel<PACK>, dim3((unsigned)blocks), dim3(kThreads), 0,
                       (hipStream_t)stream, (long long)n, t, rows);
    return launch_status();
}

}  // namespace

extern "C" {

int32_t ts_ply_row_floats(int32_t k_rest) { return k_rest < 0 ? 0 : 17 + 3 * k_rest; }

int ts_ply_pack_rows(int32_t n, int32_t k_rest, const float* means, const float* colors_dc,
                     const float* colors_rest, const float* opacities, const float* scales,
                     const float* quats, float* rows, void* stream) {
    return run<true>(n, k_rest, (float*)means, (float*)colors_dc, (float*)colors_rest,
                     (float*)opacities, (float*)scales, (float*)quats, rows, stream);
}

int ts_ply_unpack_rows(int32_t n, int32_t k_rest, const float* rows, float* means, float* colors_dc,
                       float* colors_rest, float* opacities, float* scales, float* quats,
                       void* stream) {
    return run<false>(n, k_rest, means, colors_dc, colors_rest, opacities, scales, quats,
                      (float*)rows, stream);
}

}  // extern "C"
