// (part of tools/experiments/r04_fused_params_bwd.patch: lives at tinysplat_amd/csrc/reduce_rows.h)
// reduce_rows.h - the row walk of a Gaussian's gradient rows (one 48-byte row per (tile, Gaussian), written by
// raster_bwd_kernel) and the step from the raw sums to the 2-D gradients, shared by reduce_partials_kernel
// (raster.hip) and the fused parameter-stage backward (project.hip: reduce_params_bwd_kernel) so that both produce
// the same bits.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/tinysplat_hip.h"

#ifndef TS_REDUCE_AHEAD
#define TS_REDUCE_AHEAD 4                // rows of a Gaussian requested together
#endif

namespace ts {

constexpr int kRowF4 = TS_PARTIAL_ROW_FLOATS / 4;
static_assert(TS_PARTIAL_ROW_FLOATS % 4 == 0 && TS_PARTIAL_ROW_FLOATS >= 12, "row = whole float4s, >= 10 values");

__device__ __forceinline__ unsigned int row_flag_gen(int flags) {
    return ((unsigned int)flags >> 8) & 0xffu ? ((unsigned int)flags >> 8) & 0xffu : 1u;
}

__device__ __forceinline__ void sum_rows_plain(int cnt, long long end, unsigned int gen,
                                               const float4* __restrict__ partials,
                                               const unsigned char* __restrict__ row_flags, float4& a0, float4& a1,
                                               float4& a2) {
    constexpr int kAhead = TS_REDUCE_AHEAD;
    for (long long s0 = end - cnt; s0 < end; s0 += kAhead) {
        bool f[kAhead];
        float4 p0[kAhead], p1[kAhead], p2[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) f[u] = (s0 + u < end) && row_flags[s0 + u] == gen;
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            if (f[u]) {
                p0[u] = partials[kRowF4 * (s0 + u)]; p1[u] = partials[kRowF4 * (s0 + u) + 1];
                p2[u] = partials[kRowF4 * (s0 + u) + 2];
            }
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            if (f[u]) {
                a0.x += p0[u].x; a0.y += p0[u].y; a0.z += p0[u].z; a0.w += p0[u].w;
                a1.x += p1[u].x; a1.y += p1[u].y; a1.z += p1[u].z; a1.w += p1[u].w;
                a2.x += p2[u].x; a2.y += p2[u].y;
            }
        }
    }
}

__device__ __forceinline__ void grad2d_from_sums(int flags, const float4 q0, const float4 q1, const float4 a0, float& vx,
                                                 float& vy, float& vop) {
#pragma clang fp contract(fast)
    const float A = q0.w, B = q1.x, C = q1.y, op = q0.z;
    vx = A * a0.y + B * a0.z;
    vy = B * a0.y + C * a0.z;
    vop = op > 0.0f ? -a0.x / op : 0.0f;
    if (flags & TS_RASTER_LOGIT_OPACITY) vop *= op * (1.0f - op);   // through the sigmoid
}

}  // namespace ts
