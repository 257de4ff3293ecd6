// pack.h - the 48-byte packed record of one Gaussian (see ts_pack_splats in tinysplat_hip.h), written
// either by pack_splats_kernel (binning.hip) or straight from the colour stage (project.hip:
// ts_colors_pack_fwd), which has the colours in registers and saves the colours array's round trip
// through memory and one launch.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/tinysplat_hip.h"
#include "splat_math.h"

namespace ts {

struct PackArgs {            // by value into the kernels; splats == nullptr: nothing is packed
    int channels, flags;     // flags: TS_RASTER_LOGIT_OPACITY
    const float* xys;
    const int* radii;
    const float* conics;
    const float* opacity;
    const int* cum_tiles_hit;
    const float* depths;     // channel 3 of an RGB + depth frame (channels == 4), else unused
    float4* splats;
    ts_camera cam;
};

// Record of Gaussian i with colour (c0, c1, c2) [and depths[i] as channel 3].  A Gaussian that is not
// listed in this launch (culled, or outside the tile-row stripe) gets no record: it has no reader -
// bin_count / bin_scatter skip it on the same test, the compositing kernels see listed ids only and
// ts_reduce_partials reads the record of a Gaussian with num_tiles_hit > 0 only.
__device__ __forceinline__ void pack_one(const PackArgs& a, int i, float c0, float c1, float c2, float c3) {
    const int r = a.radii[i];
    if (r <= 0) return;
    const float2 xy = reinterpret_cast<const float2*>(a.xys)[i];
    const TileBox b = tile_bbox(xy.x, xy.y, (float)r, a.cam.tile_bounds_x, a.cam.tile_bounds_y,
                                a.cam.tile_row0, a.cam.tile_rows);
    const int w = b.maxx - b.minx, h = b.maxy - b.miny;
    const int cnt = h > 0 ? w * h : 0;
    if (cnt <= 0) return;            // visible, but not in this stripe
    const int excl = a.cum_tiles_hit[i] - cnt;
    const int slot_base = excl - b.miny * w - b.minx;
    float op = a.opacity[i];
    if (a.flags & TS_RASTER_LOGIT_OPACITY) op = 1.0f / (1.0f + expf(-op));   // sigmoid, rasterize.py:86
    a.splats[3 * (size_t)i] = make_float4(xy.x, xy.y, op, a.conics[3 * i]);
    a.splats[3 * (size_t)i + 1] = make_float4(a.conics[3 * i + 1], a.conics[3 * i + 2], c0, c1);
    a.splats[3 * (size_t)i + 2] =
        make_float4(c2, c3, __int_as_float(slot_base), __int_as_float(w | (b.minx << 16)));
}

}  // namespace ts
