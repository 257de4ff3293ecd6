// project_inputs.h - how the projection kernels read the camera and the adapter-side argument preparation
// (project.hip; shard.hip's fused owner stage).  Included inside the including file's anonymous namespace.
#pragma once

__device__ __forceinline__ ts::Cam load_cam(const float* __restrict__ viewmat,
                                            const float* __restrict__ projmat, const ts_camera c) {
    ts::Cam C;
#pragma unroll
    for (int i = 0; i < 12; ++i) C.v[i] = viewmat[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) C.p[i] = projmat[i];
    C.fx = c.fx; C.fy = c.fy; C.cx = c.cx; C.cy = c.cy;
    C.W = c.img_width; C.H = c.img_height; C.tbx = c.tile_bounds_x; C.tby = c.tile_bounds_y;
    C.row0 = c.tile_row0; C.rows = c.tile_rows; C.gs = c.glob_scale; C.clip = c.clip_thresh;
    return C;
}

// The adapter-side argument preparation of rasterize.py:72-73 folded into the kernels on request:
//   TS_PROJECT_LOG_SCALES  scales hold log-scales; use exp(scales)            (rasterize.py:72)
//   TS_PROJECT_RAW_QUATS   quats are unnormalised; use quats / |quats|        (rasterize.py:73)
// (the projection itself normalises its quaternion argument once more, as upstream does).
__device__ __forceinline__ void prep_inputs(int flags, float s[3], float q[4], float* inv_norm) {
    if (flags & TS_PROJECT_LOG_SCALES) { s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]); }
    if (flags & TS_PROJECT_RAW_QUATS) {
        const float n = sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
        q[0] = q[0] / n; q[1] = q[1] / n; q[2] = q[2] / n; q[3] = q[3] / n;
        if (inv_norm) *inv_norm = 1.0f / n;
    }
}

