// shard.hip - record exchange of the Gaussian-sharded multi-GPU frame (no counterpart in
// the reference, which is single-device: /root/reference/tinysplat/splatting/rasterize.py:17).
//
// One rank per GPU owns a contiguous range of the Gaussians (parameters and optimiser state) AND one stripe
// of tile rows of the image.  Per frame a rank
//   1. projects and colours ITS Gaussians only (ts_project_fwd, ts_colors_pack_fwd with the full-frame camera),
//   2. routes the 64-byte export record of every visible Gaussian to the rank(s) whose stripe its tile box
//      reaches (ts_route_count -> counts per destination; ts_route_pack -> records grouped by destination,
//      ascending Gaussian index inside a group),
//   3. receives the records its own stripe lists (torch.distributed all_to_all, RCCL over xGMI), turns them
//      into the arrays the binning and compositing kernels take (ts_import_records, scan, ts_import_pack),
//      bins, sorts and composites its stripe exactly as a single GPU does,
//   4. in backward composites back to front, reduces the rows per imported record (ts_reduce_partials_rows),
//      returns the 48-byte gradient rows to the owners with the reverse all_to_all, and
//   5. sums the rows a Gaussian got back from its destinations in ascending stripe order
//      (ts_route_accumulate: the same walk as the packing, so no index list is kept) before the colour stage and
//      projection backward run on the owned Gaussians.
// Every per-Gaussian stage of a rank is proportional to what it owns (N / G) or to what its stripe lists
// (~N / G plus the Gaussians straddling a stripe boundary); nothing is replicated, and what crosses xGMI is
// 64 + 48 bytes per (Gaussian, stripe) pair - a few MB per rank and frame at 1 M Gaussians on 8 GPUs, instead
// of a dense 40 N-byte all-reduce.
//
// Ordering: a destination receives the groups of the source ranks in rank order, ranks own ascending index
// ranges and a group is packed in ascending index order, so the local index of a record at its destination is
// monotone in the global Gaussian index: ties of equal depth break exactly as on one GPU, and the stripes of a
// sharded frame tile the single-GPU image bit for bit.
#include <hip/hip_runtime.h>

#include "../../include/tinysplat_hip.h"
#include "splat_math.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
#include "project_inputs.h"

// destination ranks [d0, d1] of a Gaussian (d0 > d1: none)
struct DestRange { int d0, d1; };

__device__ __forceinline__ int stripe_of_row(const ts_stripes& st, int row) {
    int d = 0;
    while (d + 1 < st.num && row >= st.row[d + 1]) ++d;
    return d;
}

__device__ __forceinline__ DestRange dest_range(int i, int n, const float* __restrict__ xys,
                                                const int* __restrict__ radii, const ts_camera& cam,
                                                const ts_stripes& st) {
    DestRange r{1, 0};
    if (i >= n) return r;
    const int rad = radii[i];
    if (rad <= 0) return r;
    const float2 xy = reinterpret_cast<const float2*>(xys)[i];
    const ts::TileBox b = ts::tile_bbox(xy.x, xy.y, (float)rad, cam.tile_bounds_x, cam.tile_bounds_y, 0,
                                        cam.tile_bounds_y);
    if (b.maxx <= b.minx || b.maxy <= b.miny) return r;
    // rows outside every stripe (st.row[0] > 0 or st.row[num] < tile_bounds_y) have no destination
    const int lo = max(b.miny, st.row[0]), hi = min(b.maxy, st.row[st.num]);
    if (hi <= lo) return r;
    r.d0 = stripe_of_row(st, lo);
    r.d1 = stripe_of_row(st, hi - 1);
    return r;
}

// block_counts[d * B + b] = Gaussians of block b with a record for destination d
__global__ __launch_bounds__(kThreads) void route_count_kernel(int n, const float* __restrict__ xys,
                                                               const int* __restrict__ radii,
                                                               const ts_camera cam, const ts_stripes st,
                                                               int* __restrict__ block_counts) {
    __shared__ int cnt[kWaves][TS_MAX_RANKS];
    const int i = blockIdx.x * kThreads + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const DestRange r = dest_range(i, n, xys, radii, cam, st);
    for (int d = 0; d < st.num; ++d) {
        const unsigned long long m = __ballot(r.d0 <= d && d <= r.d1);
        if (lane == 0) cnt[wave][d] = __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < st.num) {
        int s = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) s += cnt[w][threadIdx.x];
        block_counts[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s;
    }
}

// FUSED OWNER FORWARD (small N: a rank of 8 owns ~125 k Gaussians and its stage is bound by launch floors, not by
// bytes): projection (project.hip: project_fwd_kernel), colour stage + packed record (sh_colors_fwd_sparse_kernel +
// pack_one, from registers) and the per-block destination counts (route_count_kernel) of one owned Gaussian per lane,
// in ONE launch.  The same device functions in the same order: every array the three launches wrote holds the same
// bits (the clamp mask of a culled Gaussian, which nothing reads, is 0 here).
template <int DEG>
__global__ __launch_bounds__(kThreads) void owner_fwd_kernel(
    int n, int num_bases, const float* __restrict__ means3d, const float* __restrict__ scales,
    const float* __restrict__ quats, const float* __restrict__ viewmat, const float* __restrict__ projmat,
    const ts_camera cam, const int pflags, const float* __restrict__ origin, const float* __restrict__ dc,
    const float* __restrict__ rest, const float* __restrict__ opacity, const int channels, const int rflags,
    float* __restrict__ xys, float* __restrict__ depths, int* __restrict__ radii, float* __restrict__ conics,
    int* __restrict__ num_tiles_hit, unsigned char* __restrict__ mask, float4* __restrict__ splats,
    const ts_stripes st, int* __restrict__ block_counts) {
    constexpr int KA = (DEG + 1) * (DEG + 1);
    __shared__ int cnt[kWaves][TS_MAX_RANKS];
    const int i = blockIdx.x * kThreads + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    DestRange r{1, 0};
    if (i < n) {
        const ts::Cam C = load_cam(viewmat, projmat, cam);
        const float m[3] = {means3d[3 * i], means3d[3 * i + 1], means3d[3 * i + 2]};
        float sc[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
        const float4 qv = reinterpret_cast<const float4*>(quats)[i];
        float q[4] = {qv.x, qv.y, qv.z, qv.w};
        prep_inputs(pflags, sc, q, nullptr);
        ts::ProjOut o;
        ts::project_one(C, m, sc, q, o);
        reinterpret_cast<float2*>(xys)[i] = make_float2(o.x, o.y);
        depths[i] = o.depth;
        radii[i] = o.radius;
        conics[3 * i] = o.conic[0]; conics[3 * i + 1] = o.conic[1]; conics[3 * i + 2] = o.conic[2];
        num_tiles_hit[i] = o.tiles;
        if (o.radius > 0) {
            // colour stage: the association order of sh_colors_fwd_kernel / _sparse_kernel (identical bits)
            float Y[KA];
            ts::sh_basis(DEG, m[0] - origin[0], m[1] - origin[1], m[2] - origin[2], Y);
            float c0 = Y[0] * dc[3 * i], c1 = Y[0] * dc[3 * i + 1], c2 = Y[0] * dc[3 * i + 2];
            const float* row = rest + (size_t)i * 3 * (num_bases - 1);
#pragma unroll
            for (int k = 1; k < KA; ++k) {
                c0 = c0 + Y[k] * row[3 * (k - 1)];
                c1 = c1 + Y[k] * row[3 * (k - 1) + 1];
                c2 = c2 + Y[k] * row[3 * (k - 1) + 2];
            }
            c0 = c0 + 0.5f; c1 = c1 + 0.5f; c2 = c2 + 0.5f;
            if (mask) mask[i] = (unsigned char)((c0 >= 0.0f ? 1 : 0) | (c1 >= 0.0f ? 2 : 0) | (c2 >= 0.0f ? 4 : 0));
            c0 = fmaxf(c0, 0.0f); c1 = fmaxf(c1, 0.0f); c2 = fmaxf(c2, 0.0f);
            // packed record (pack.h: pack_one, full-frame camera) from the registers
            const ts::TileBox b = ts::tile_bbox(o.x, o.y, (float)o.radius, cam.tile_bounds_x, cam.tile_bounds_y,
                                                cam.tile_row0, cam.tile_rows);
            const int w = b.maxx - b.minx, h = b.maxy - b.miny;
            const int cntb = h > 0 ? w * h : 0;
            if (cntb > 0) {
                const int slot_base = (o.tiles - cntb) - b.miny * w - b.minx;      // (cum_tiles_hit := num_tiles_hit here)
                float op = opacity[i];
                if (rflags & TS_RASTER_LOGIT_OPACITY) op = 1.0f / (1.0f + expf(-op));
                splats[3 * (size_t)i] = make_float4(o.x, o.y, op, o.conic[0]);
                splats[3 * (size_t)i + 1] = make_float4(o.conic[1], o.conic[2], c0, c1);
                splats[3 * (size_t)i + 2] = make_float4(c2, channels == 4 ? o.depth : 0.0f, __int_as_float(slot_base),
                                                        __int_as_float(w | (b.minx << 16)));
            }
            // destinations: dest_range on the registers
            const ts::TileBox fb = ts::tile_bbox(o.x, o.y, (float)o.radius, cam.tile_bounds_x, cam.tile_bounds_y, 0,
                                                 cam.tile_bounds_y);
            if (fb.maxx > fb.minx && fb.maxy > fb.miny) {
                const int lo = max(fb.miny, st.row[0]), hi = min(fb.maxy, st.row[st.num]);
                if (hi > lo) { r.d0 = stripe_of_row(st, lo); r.d1 = stripe_of_row(st, hi - 1); }
            }
        } else if (mask) {
            mask[i] = 0;
        }
    }
    for (int d = 0; d < st.num; ++d) {
        const unsigned long long mm = __ballot(r.d0 <= d && d <= r.d1);
        if (lane == 0) cnt[wave][d] = __popcll(mm);
    }
    __syncthreads();
    if (threadIdx.x < st.num) {
        int sum = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) sum += cnt[w][threadIdx.x];
        block_counts[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = sum;
    }
}

// single workgroup: per destination an exclusive scan over the blocks (in place), then the exclusive scan of
// the destination totals.  counts[d] = records for destination d, seg[d] = first record of its group.
// PADDED groups (ts_route_count_padded): seg[d] is not the running sum of the counts but a base the caller fixed
// BEFORE the counts were known (capacities from the previous frame), so that the exchange needs no host read of this
// frame's counts; the rows between a group's count and its capacity are the caller's (zeroed) padding.
struct GroupBase { int padded; int base[TS_MAX_RANKS + 1]; };
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void route_scan_kernel(int num_blocks, int num_dest,
                                                                  int* __restrict__ block_counts,
                                                                  int* __restrict__ seg,
                                                                  int* __restrict__ counts, const GroupBase gb) {
    __shared__ int wsum[kScanThreads / 64];
    __shared__ int carry, total[TS_MAX_RANKS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = 0; d < num_dest; ++d) {
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        int* row = block_counts + (size_t)d * num_blocks;
        for (int base = 0; base < num_blocks; base += kScanThreads) {
            const int b = base + threadIdx.x;
            const int v = b < num_blocks ? row[b] : 0;
            int inc = v;
#pragma unroll
            for (int k = 1; k < 64; k <<= 1) {
                const int u = __shfl_up(inc, k, 64);
                if (lane >= k) inc += u;
            }
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            int add = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < kScanThreads / 64; ++w) {
                const int s = wsum[w];
                if (w < wave) add += s;
                tot += s;
            }
            if (b < num_blocks) row[b] = carry + add + inc - v;
            __syncthreads();
            if (threadIdx.x == 0) carry += tot;
            __syncthreads();
        }
        if (threadIdx.x == 0) total[d] = carry;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int run = 0;
        for (int d = 0; d < num_dest; ++d) {
            seg[d] = gb.padded ? gb.base[d] : run;
            counts[d] = total[d];
            run += total[d];
        }
        seg[num_dest] = gb.padded ? gb.base[num_dest] : run;
    }
}

// The ordered position of (Gaussian i, destination d) in the send buffer is
//   seg[d] + base of the block for d + Gaussians of earlier waves of the block + earlier lanes of the wave;
// packing and the gradient accumulation replay the same walk, so they agree without an index list.
__global__ __launch_bounds__(kThreads) void route_pack_kernel(
    int n, int gid_base, const float* __restrict__ xys, const int* __restrict__ radii,
    const float* __restrict__ depths, const float4* __restrict__ splats, const ts_camera cam,
    const ts_stripes st, const int* __restrict__ block_base, const int* __restrict__ seg,
    float4* __restrict__ records) {
    __shared__ int cnt[kWaves][TS_MAX_RANKS];
    const int i = blockIdx.x * kThreads + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const DestRange r = dest_range(i, n, xys, radii, cam, st);
    float4 q0, q1, q2, q3;
    if (r.d0 <= r.d1) {
        q0 = splats[3 * (size_t)i]; q1 = splats[3 * (size_t)i + 1]; q2 = splats[3 * (size_t)i + 2];
        q2.z = depths[i];
        q2.w = __int_as_float(radii[i]);
        q3 = make_float4(__int_as_float(gid_base + i), 0.f, 0.f, 0.f);
    }
    for (int d = 0; d < st.num; ++d) {               // wave-uniform: ballots by the whole wave
        const bool mine = r.d0 <= d && d <= r.d1;
        const unsigned long long m = __ballot(mine);
        if (lane == 0) cnt[wave][d] = __popcll(m);
    }
    __syncthreads();
    // No lane leaves an iteration early: every ballot of this loop is executed by the whole wave (a `continue`
    // before the next iteration's ballot would rely on the compiler reconverging the wave at the loop latch,
    // which HIP does not promise - ADVICE r3); only the stores are predicated.
    for (int d = 0; d < st.num; ++d) {
        const bool mine = r.d0 <= d && d <= r.d1;
        const unsigned long long m = __ballot(mine);
        int pos = seg[d] + block_base[(size_t)d * gridDim.x + blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) pos += cnt[w][d];
        // (padded groups: a group that outgrew its capacity is cut off at the next group's base - the caller sees the
        // count, drops the frame and runs it again with exact sizes)
        if (mine && pos < seg[d + 1]) {
            float4* o = records + 4 * (size_t)pos;
            o[0] = q0; o[1] = q1; o[2] = q2; o[3] = q3;
        }
    }
}

// Sums, per owned Gaussian, the gradient rows that came back from its destinations (ascending stripe: a fixed
// order, so the result is reproducible), applies the colour stage's clamp mask and the sigmoid's derivative
// (the rows carry dL/d opacity), and writes the dense 2-D gradients of the owned Gaussians (zeros for the
// ones that reached no stripe).  Row: {v_x, v_y, v_conic xx, xy | yy, v_c0, v_c1, v_c2 | v_depth, v_opacity, -, -}.
__global__ __launch_bounds__(kThreads) void route_accumulate_kernel(
    int n, int channels, const float* __restrict__ xys, const int* __restrict__ radii,
    const float4* __restrict__ splats, const unsigned char* __restrict__ color_mask, const ts_camera cam,
    const ts_stripes st, const int* __restrict__ block_base, const int* __restrict__ seg,
    const float4* __restrict__ rows, float* __restrict__ v_xy, float* __restrict__ v_conic,
    float* __restrict__ v_colors, float* __restrict__ v_depth, float* __restrict__ v_opacity) {
    __shared__ int cnt[kWaves][TS_MAX_RANKS];
    const int i = blockIdx.x * kThreads + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const DestRange r = dest_range(i, n, xys, radii, cam, st);
    for (int d = 0; d < st.num; ++d) {
        const unsigned long long m = __ballot(r.d0 <= d && d <= r.d1);
        if (lane == 0) cnt[wave][d] = __popcll(m);
    }
    __syncthreads();
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
    for (int d = 0; d < st.num; ++d) {
        const bool mine = r.d0 <= d && d <= r.d1;
        const unsigned long long m = __ballot(mine);          // whole wave, every iteration (see route_pack_kernel)
        int pos = seg[d] + block_base[(size_t)d * gridDim.x + blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) pos += cnt[w][d];
        if (mine) {
            const float4 p0 = rows[3 * (size_t)pos], p1 = rows[3 * (size_t)pos + 1], p2 = rows[3 * (size_t)pos + 2];
            a0.x += p0.x; a0.y += p0.y; a0.z += p0.z; a0.w += p0.w;
            a1.x += p1.x; a1.y += p1.y; a1.z += p1.z; a1.w += p1.w;
            a2.x += p2.x; a2.y += p2.y;
        }
    }
    if (i >= n) return;
    if (r.d0 <= r.d1) {
        const float op = splats[3 * (size_t)i].z;
        a2.y *= op * (1.0f - op);                        // through the sigmoid of rasterize.py:86
        if (color_mask) {
            const int mk = color_mask[i];
            if (!(mk & 1)) a1.y = 0.0f;
            if (!(mk & 2)) a1.z = 0.0f;
            if (!(mk & 4)) a1.w = 0.0f;
        }
    }
    reinterpret_cast<float2*>(v_xy)[i] = make_float2(a0.x, a0.y);
    v_conic[3 * i] = a0.z; v_conic[3 * i + 1] = a0.w; v_conic[3 * i + 2] = a1.x;
    v_colors[3 * i] = a1.y; v_colors[3 * i + 1] = a1.z; v_colors[3 * i + 2] = a1.w;
    if (channels == 4 && v_depth) v_depth[i] = a2.x;
    v_opacity[i] = a2.y;
}

// record -> the per-Gaussian arrays binning takes, for the importing rank's stripe
// FUSED OWNER BACKWARD (small N, as owner_fwd_kernel): the rows that came back summed per owned Gaussian
// (route_accumulate_kernel), the colour stage's backward (project.hip: sh_colors_bwd_kernel without a mask - it was
// applied to the sums) and the projection's backward (project_bwd_kernel, flags 3) in the lane that owns the Gaussian:
// the 2-D gradients stay in registers, two launch boundaries less.  Same device functions, same order: same bits.
template <int DEG>
__global__ __launch_bounds__(kThreads) void owner_bwd_kernel(
    int n, int channels, int num_bases, const float* __restrict__ means3d, const float* __restrict__ scales,
    const float* __restrict__ quats, const float* __restrict__ viewmat, const float* __restrict__ projmat,
    const float* __restrict__ origin, const float* __restrict__ xys, const int* __restrict__ radii,
    const float4* __restrict__ splats, const unsigned char* __restrict__ color_mask, const ts_camera cam,
    const ts_stripes st, const int* __restrict__ block_base, const int* __restrict__ seg,
    const float4* __restrict__ rows, float* __restrict__ v_xy, float* __restrict__ v_conic,
    float* __restrict__ v_colors, float* __restrict__ v_depth, float* __restrict__ v_opacity,
    float* __restrict__ v_dc, float* __restrict__ v_rest, float* __restrict__ v_means3d,
    float* __restrict__ v_scales, float* __restrict__ v_quats) {
    constexpr int KA = (DEG + 1) * (DEG + 1);
    __shared__ int cnt[kWaves][TS_MAX_RANKS];
    const int i = blockIdx.x * kThreads + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const DestRange r = dest_range(i, n, xys, radii, cam, st);
    for (int d = 0; d < st.num; ++d) {
        const unsigned long long m = __ballot(r.d0 <= d && d <= r.d1);
        if (lane == 0) cnt[wave][d] = __popcll(m);
    }
    __syncthreads();
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
    for (int d = 0; d < st.num; ++d) {
        const bool mine = r.d0 <= d && d <= r.d1;
        const unsigned long long m = __ballot(mine);          // whole wave, every iteration (see route_pack_kernel)
        int pos = seg[d] + block_base[(size_t)d * gridDim.x + blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) pos += cnt[w][d];
        if (mine) {
            const float4 p0 = rows[3 * (size_t)pos], p1 = rows[3 * (size_t)pos + 1], p2 = rows[3 * (size_t)pos + 2];
            a0.x += p0.x; a0.y += p0.y; a0.z += p0.z; a0.w += p0.w;
            a1.x += p1.x; a1.y += p1.y; a1.z += p1.z; a1.w += p1.w;
            a2.x += p2.x; a2.y += p2.y;
        }
    }
    if (i >= n) return;
    if (r.d0 <= r.d1) {
        const float op = splats[3 * (size_t)i].z;
        a2.y *= op * (1.0f - op);                        // through the sigmoid of rasterize.py:86
        if (color_mask) {
            const int mk = color_mask[i];
            if (!(mk & 1)) a1.y = 0.0f;
            if (!(mk & 2)) a1.z = 0.0f;
            if (!(mk & 4)) a1.w = 0.0f;
        }
    }
    reinterpret_cast<float2*>(v_xy)[i] = make_float2(a0.x, a0.y);
    v_conic[3 * i] = a0.z; v_conic[3 * i + 1] = a0.w; v_conic[3 * i + 2] = a1.x;
    v_colors[3 * i] = a1.y; v_colors[3 * i + 1] = a1.z; v_colors[3 * i + 2] = a1.w;
    const bool with_depth = channels == 4 && v_depth;
    if (with_depth) v_depth[i] = a2.x;
    v_opacity[i] = a2.y;
    const float m3[3] = {means3d[3 * i], means3d[3 * i + 1], means3d[3 * i + 2]};
    {   // colour stage, backward: v_dc = Y0 v, row k of v_rest = Y_k v (inactive bands: zeros)
        float Y[KA];
        ts::sh_basis(DEG, m3[0] - origin[0], m3[1] - origin[1], m3[2] - origin[2], Y);
        const float v0 = a1.y, v1 = a1.z, v2 = a1.w;
        v_dc[3 * i] = Y[0] * v0; v_dc[3 * i + 1] = Y[0] * v1; v_dc[3 * i + 2] = Y[0] * v2;
        const int RS = 3 * (num_bases - 1);
        float* row = v_rest + (size_t)i * RS;
#pragma unroll
        for (int k = 1; k < KA; ++k) {
            row[3 * (k - 1)] = Y[k] * v0; row[3 * (k - 1) + 1] = Y[k] * v1; row[3 * (k - 1) + 2] = Y[k] * v2;
        }
        for (int j = 3 * (KA - 1); j < RS; ++j) row[j] = 0.0f;
    }
    ts::ProjGrad g;
    for (int k = 0; k < 3; ++k) { g.v_mean[k] = 0.0f; g.v_scale[k] = 0.0f; }
    for (int k = 0; k < 4; ++k) g.v_quat[k] = 0.0f;
    if (radii[i] > 0) {
        const ts::Cam C = load_cam(viewmat, projmat, cam);
        float sc[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
        const float4 qv = reinterpret_cast<const float4*>(quats)[i];
        float q[4] = {qv.x, qv.y, qv.z, qv.w};
        float inv_n = 1.0f;
        prep_inputs(3, sc, q, &inv_n);
        const float vx[2] = {a0.x, a0.y};
        const float vc[3] = {a0.z, a0.w, a1.x};
        ts::project_one_vjp(C, m3, sc, q, vx, with_depth ? a2.x : 0.0f, vc, nullptr, g);
        g.v_scale[0] *= sc[0]; g.v_scale[1] *= sc[1]; g.v_scale[2] *= sc[2];            // d exp(x) = exp(x) dx
        const float dotp = q[0] * g.v_quat[0] + q[1] * g.v_quat[1] + q[2] * g.v_quat[2] + q[3] * g.v_quat[3];
        for (int k = 0; k < 4; ++k) g.v_quat[k] = (g.v_quat[k] - q[k] * dotp) * inv_n;  // q_hat = q / |q|
    }
    v_means3d[3 * i] = g.v_mean[0]; v_means3d[3 * i + 1] = g.v_mean[1]; v_means3d[3 * i + 2] = g.v_mean[2];
    v_scales[3 * i] = g.v_scale[0]; v_scales[3 * i + 1] = g.v_scale[1]; v_scales[3 * i + 2] = g.v_scale[2];
    reinterpret_cast<float4*>(v_quats)[i] = make_float4(g.v_quat[0], g.v_quat[1], g.v_quat[2], g.v_quat[3]);
}

__global__ __launch_bounds__(kThreads) void import_records_kernel(int m, const float4* __restrict__ records,
                                                                  const ts_camera cam, float* __restrict__ xys,
                                                                  float* __restrict__ depths,
                                                                  int* __restrict__ radii,
                                                                  int* __restrict__ num_tiles_hit) {
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= m) return;
    const float4 q0 = records[4 * (size_t)j], q2 = records[4 * (size_t)j + 2];
    const int rad = __float_as_int(q2.w);
    const ts::TileBox b = ts::tile_bbox(q0.x, q0.y, (float)rad, cam.tile_bounds_x, cam.tile_bounds_y,
                                        cam.tile_row0, cam.tile_rows);
    const int h = b.maxy - b.miny, w = b.maxx - b.minx;
    reinterpret_cast<float2*>(xys)[j] = make_float2(q0.x, q0.y);
    depths[j] = q2.z;
    radii[j] = rad;
    num_tiles_hit[j] = (h > 0 && w > 0) ? w * h : 0;
}

// record + cum_tiles_hit -> the 48-byte compositing record (pack.h layout) of the importing rank
__global__ __launch_bounds__(kThreads) void import_pack_kernel(int m, const float4* __restrict__ records,
                                                               const int* __restrict__ cum_tiles_hit,
                                                               const ts_camera cam, float4* __restrict__ splats) {
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= m) return;
    const float4 q0 = records[4 * (size_t)j], q1 = records[4 * (size_t)j + 1], q2 = records[4 * (size_t)j + 2];
    const int rad = __float_as_int(q2.w);
    const ts::TileBox b = ts::tile_bbox(q0.x, q0.y, (float)rad, cam.tile_bounds_x, cam.tile_bounds_y,
                                        cam.tile_row0, cam.tile_rows);
    const int w = b.maxx - b.minx, h = b.maxy - b.miny;
    const int cnt = (h > 0 && w > 0) ? w * h : 0;
    if (cnt <= 0) return;                       // no reader (see pack.h)
    const int slot_base = cum_tiles_hit[j] - cnt - b.miny * w - b.minx;
    splats[3 * (size_t)j] = q0;
    splats[3 * (size_t)j + 1] = q1;
    splats[3 * (size_t)j + 2] = make_float4(q2.x, q2.y, __int_as_float(slot_base), __int_as_float(w | (b.minx << 16)));
}

inline int launch_status() { return (int)hipGetLastError(); }

inline bool bad_stripes(const ts_stripes* st, const ts_camera* cam) {
    if (!st || !cam || st->num < 1 || st->num > TS_MAX_RANKS) return true;
    for (int d = 0; d < st->num; ++d)
        if (st->row[d] > st->row[d + 1]) return true;
    return st->row[0] < 0 || st->row[st->num] > cam->tile_bounds_y;
}

inline int route_blocks(int n) { return n > 0 ? (n + kThreads - 1) / kThreads : 1; }

}  // namespace

extern "C" {

int64_t ts_route_ws_ints(int32_t n, int32_t num_ranks) {
    if (num_ranks < 1) num_ranks = 1;
    return (int64_t)route_blocks(n) * num_ranks + num_ranks + 1;
}

int ts_route_count_padded(int32_t n, const float* xys, const int32_t* radii, const ts_camera* cam,
                          const ts_stripes* stripes, const int32_t* group_base, int32_t* route_ws, int32_t* counts,
                          void* stream) {
    if (n < 0 || bad_stripes(stripes, cam) || !route_ws || !counts) return TS_E_BADARG;
    if (n > 0 && (!xys || !radii)) return TS_E_BADARG;
    GroupBase gb;
    gb.padded = group_base != nullptr;
    for (int d = 0; d <= TS_MAX_RANKS; ++d) gb.base[d] = 0;
    if (group_base) {
        for (int d = 0; d <= stripes->num; ++d) {
            if (group_base[d] < 0 || (d > 0 && group_base[d] < group_base[d - 1])) return TS_E_BADARG;
            gb.base[d] = group_base[d];
        }
    }
    const int blocks = route_blocks(n);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(route_count_kernel, dim3(blocks), dim3(kThreads), 0, s, n, xys, radii, *cam, *stripes,
                       route_ws);
    hipLaunchKernelGGL(route_scan_kernel, dim3(1), dim3(kScanThreads), 0, s, blocks, (int)stripes->num, route_ws,
                       route_ws + (size_t)blocks * stripes->num, counts, gb);
    return launch_status();
}

int ts_route_count(int32_t n, const float* xys, const int32_t* radii, const ts_camera* cam,
                   const ts_stripes* stripes, int32_t* route_ws, int32_t* counts, void* stream) {
    return ts_route_count_padded(n, xys, radii, cam, stripes, nullptr, route_ws, counts, stream);
}

int ts_route_pack(int32_t n, int32_t gid_base, const float* xys, const int32_t* radii, const float* depths,
                  const float* splats, const ts_camera* cam, const ts_stripes* stripes, const int32_t* route_ws,
                  float* records, void* stream) {
    if (n < 0 || bad_stripes(stripes, cam) || !route_ws) return TS_E_BADARG;
    if (n == 0) return 0;
    if (!xys || !radii || !depths || !splats || !records) return TS_E_BADARG;
    const int blocks = route_blocks(n);
    hipLaunchKernelGGL(route_pack_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, n, (int)gid_base,
                       xys, radii, depths, reinterpret_cast<const float4*>(splats), *cam, *stripes, route_ws,
                       route_ws + (size_t)blocks * stripes->num, reinterpret_cast<float4*>(records));
    return launch_status();
}

int ts_route_accumulate(int32_t n, int32_t channels, const float* xys, const int32_t* radii, const float* splats,
                        const uint8_t* color_mask, const ts_camera* cam, const ts_stripes* stripes,
                        const int32_t* route_ws, const float* grad_rows, float* v_xy, float* v_conic,
                        float* v_colors, float* v_depth, float* v_opacity, void* stream) {
    if (n < 0 || (channels != 3 && channels != 4) || bad_stripes(stripes, cam) || !route_ws) return TS_E_BADARG;
    if (n == 0) return 0;
    if (!xys || !radii || !splats || !grad_rows || !v_xy || !v_conic || !v_colors || !v_opacity)
        return TS_E_BADARG;
    const int blocks = route_blocks(n);
    hipLaunchKernelGGL(route_accumulate_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, n,
                       (int)channels, xys, radii, reinterpret_cast<const float4*>(splats), color_mask, *cam,
                       *stripes, route_ws, route_ws + (size_t)blocks * stripes->num,
                       reinterpret_cast<const float4*>(grad_rows), v_xy, v_conic, v_colors, v_depth, v_opacity);
    return launch_status();
}

int ts_import_records(int32_t m, const float* records, const ts_camera* cam, float* xys, float* depths,
                      int32_t* radii, int32_t* num_tiles_hit, void* stream) {
    if (m < 0 || !cam) return TS_E_BADARG;
    if (m == 0) return 0;
    if (!records || !xys || !depths || !radii || !num_tiles_hit) return TS_E_BADARG;
    hipLaunchKernelGGL(import_records_kernel, dim3((m + kThreads - 1) / kThreads), dim3(kThreads), 0,
                       (hipStream_t)stream, m, reinterpret_cast<const float4*>(records), *cam, xys, depths, radii,
                       num_tiles_hit);
    return launch_status();
}

int ts_import_pack(int32_t m, const float* records, const int32_t* cum_tiles_hit, const ts_camera* cam,
                   float* splats, void* stream) {
    if (m < 0 || !cam) return TS_E_BADARG;
    if (m == 0) return 0;
    if (!records || !cum_tiles_hit || !splats) return TS_E_BADARG;
    hipLaunchKernelGGL(import_pack_kernel, dim3((m + kThreads - 1) / kThreads), dim3(kThreads), 0,
                       (hipStream_t)stream, m, reinterpret_cast<const float4*>(records), cum_tiles_hit, *cam,
                       reinterpret_cast<float4*>(splats));
    return launch_status();
}

int ts_shard_owner_fwd_fused(int32_t n, int32_t degrees_to_use, int32_t num_bases, const float* means3d,
                             const float* scales, const float* quats, const float* viewmat, const float* projmat,
                             const ts_camera* cam, int32_t project_flags, const float* origin, const float* colors_dc,
                             const float* colors_rest, const float* opacity, int32_t channels, int32_t raster_flags,
                             float* xys, float* depths, int32_t* radii, float* conics, int32_t* num_tiles_hit,
                             uint8_t* clamp_mask, float* splats, const ts_stripes* stripes, const int32_t* group_base,
                             int32_t* route_ws, int32_t* counts, void* stream) {
    if (n < 0 || !cam || bad_stripes(stripes, cam) || !route_ws || !counts || (channels != 3 && channels != 4) ||
        degrees_to_use < 0 || degrees_to_use > 4 || num_bases < (degrees_to_use + 1) * (degrees_to_use + 1))
        return TS_E_BADARG;
    if (n > 0 && (!means3d || !scales || !quats || !viewmat || !projmat || !origin || !colors_dc || !opacity || !xys ||
                  !depths || !radii || !conics || !num_tiles_hit || !splats || (num_bases > 1 && !colors_rest)))
        return TS_E_BADARG;
    GroupBase gb;
    gb.padded = group_base != nullptr;
    for (int d = 0; d <= TS_MAX_RANKS; ++d) gb.base[d] = 0;
    if (group_base) {
        for (int d = 0; d <= stripes->num; ++d) {
            if (group_base[d] < 0 || (d > 0 && group_base[d] < group_base[d - 1])) return TS_E_BADARG;
            gb.base[d] = group_base[d];
        }
    }
    const int blocks = route_blocks(n);
    hipStream_t s = (hipStream_t)stream;
#define TS_OWNER_FWD(D)                                                                                            \
    hipLaunchKernelGGL(owner_fwd_kernel<D>, dim3(blocks), dim3(kThreads), 0, s, n, num_bases, means3d, scales, quats, \
                       viewmat, projmat, *cam, (int)project_flags, origin, colors_dc, colors_rest, opacity,        \
                       (int)channels, (int)raster_flags, xys, depths, radii, conics, num_tiles_hit, clamp_mask,    \
                       reinterpret_cast<float4*>(splats), *stripes, route_ws)
    switch (degrees_to_use) {
        case 0: TS_OWNER_FWD(0); break;
        case 1: TS_OWNER_FWD(1); break;
        case 2: TS_OWNER_FWD(2); break;
        case 3: TS_OWNER_FWD(3); break;
        default: TS_OWNER_FWD(4); break;
    }
#undef TS_OWNER_FWD
    hipLaunchKernelGGL(route_scan_kernel, dim3(1), dim3(kScanThreads), 0, s, blocks, (int)stripes->num, route_ws,
                       route_ws + (size_t)blocks * stripes->num, counts, gb);
    return launch_status();
}

int ts_shard_owner_bwd_fused(int32_t n, int32_t channels, int32_t degrees_to_use, int32_t num_bases, const float* means3d,
                             const float* scales, const float* quats, const float* viewmat, const float* projmat,
                             const float* origin, const float* xys, const int32_t* radii, const float* splats,
                             const uint8_t* color_mask, const ts_camera* cam, const ts_stripes* stripes,
                             const int32_t* route_ws, const float* grad_rows, float* v_xy, float* v_conic,
                             float* v_colors, float* v_depth, float* v_opacity, float* v_colors_dc, float* v_colors_rest,
                             float* v_means3d, float* v_scales, float* v_quats, void* stream) {
    if (n < 0 || !cam || bad_stripes(stripes, cam) || !route_ws || (channels != 3 && channels != 4) ||
        degrees_to_use < 0 || degrees_to_use > 4 || num_bases < (degrees_to_use + 1) * (degrees_to_use + 1))
        return TS_E_BADARG;
    if (n == 0) return 0;
    if (!means3d || !scales || !quats || !viewmat || !projmat || !origin || !xys || !radii || !splats || !grad_rows ||
        !v_xy || !v_conic || !v_colors || !v_opacity || !v_colors_dc || (num_bases > 1 && !v_colors_rest) || !v_means3d ||
        !v_scales || !v_quats)
        return TS_E_BADARG;
    const int blocks = route_blocks(n);
    hipStream_t s = (hipStream_t)stream;
#define TS_OWNER_BWD(D)                                                                                            \
    hipLaunchKernelGGL(owner_bwd_kernel<D>, dim3(blocks), dim3(kThreads), 0, s, n, (int)channels, (int)num_bases,  \
                       means3d, scales, quats, viewmat, projmat, origin, xys, radii,                               \
                       reinterpret_cast<const float4*>(splats), color_mask, *cam, *stripes, route_ws,              \
                       route_ws + (size_t)blocks * stripes->num, reinterpret_cast<const float4*>(grad_rows), v_xy, \
                       v_conic, v_colors, v_depth, v_opacity, v_colors_dc, v_colors_rest, v_means3d, v_scales, v_quats)
    switch (degrees_to_use) {
        case 0: TS_OWNER_BWD(0); break;
        case 1: TS_OWNER_BWD(1); break;
        case 2: TS_OWNER_BWD(2); break;
        case 3: TS_OWNER_BWD(3); break;
        default: TS_OWNER_BWD(4); break;
    }
#undef TS_OWNER_BWD
    return launch_status();
}

}  // extern "C"
