// raster.hip - tile compositing kernels for gfx950 (forward, backward, gradient row reduce).
//
// What gsplat's rasterize_forward / rasterize_backward compute for tinysplat's calls at
// /root/reference/tinysplat/splatting/rasterize.py:44,50 (front-to-back alpha compositing of the
// depth-sorted per-tile lists, and the back-to-front replay that produces v_xy / v_conic /
// v_colors / v_opacity), re-designed around the 64-wide wavefront:
//
//   * ONE WAVE OWNS ONE 16x16 TILE.  Lane l covers column (l & 15) and the four rows
//     (l >> 4) + {0,4,8,12}; four pixels per lane.  There is no workgroup barrier anywhere: the
//     four waves of a 256-thread workgroup run four different tiles independently, early-out is a
//     wave ballot, and per-Gaussian terms that depend only on the column (dx, A dx^2, B dx) are
//     computed once per lane instead of once per pixel.
//   * Per 64-entry chunk of the tile's sorted list each lane gathers ONE Gaussian's 48-byte packed
//     record (3 x 16 B loads), tests it exactly against the tile rectangle (minimum of the conic
//     form over the rectangle vs. the alpha >= 1/255 level set - conservative, so results are
//     unchanged), and survivors are compacted into LDS with a wave ballot + prefix count.  The
//     inner loop then reads each survivor back with wave-uniform (broadcast) ds_read_b128s.
//   * Backward: per-lane partial sums over its 4 pixels, then a DPP butterfly (quad_perm,
//     row_half_mirror, row_mirror, row_bcast15/31) reduces 6+C values across the wave and one lane
//     writes one 48-byte row per (tile, Gaussian) into a slot that is contiguous per Gaussian.
//     reduce_partials then sums each Gaussian's rows in a fixed order: no float atomics, and the
//     gradients are bit-reproducible run to run.
#include <hip/hip_runtime.h>

#include "../../include/tinysplat_hip.h"
#include "splat_math.h"

namespace {

constexpr int kWaves = 4;              // tiles per workgroup
constexpr int kThreads = 64 * kWaves;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLog2_255 = 7.994353436858858f;

#define TS_WAVE_SYNC()                                            \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
    } while (0)

__device__ __forceinline__ float dpp_add(float v, const int ctrl, const int row_mask) {
    // v + (v moved by the DPP control); lanes of disabled rows / without a source add 0
    int moved;
    switch (ctrl) {  // the builtin needs literal immediates
        case 0xB1: moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true); break;
        case 0x4E: moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true); break;
        case 0x141: moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true); break;
        case 0x140: moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true); break;
        case 0x142: moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, true); break;
        default: moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, true); break;
    }
    (void)row_mask;
    return v + __int_as_float(moved);
}

// sum over the 64 lanes; the total is valid in lanes 48..63 (read it from lane 63)
__device__ __forceinline__ float wave_sum_hi(float v) {
    v = dpp_add(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
    v = dpp_add(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
    v = dpp_add(v, 0x141, 0xF);   // row_half_mirror
    v = dpp_add(v, 0x140, 0xF);   // row_mirror      -> every lane holds its row-of-16 sum
    v = dpp_add(v, 0x142, 0xA);   // row_bcast15     -> rows 1,3 += previous row
    v = dpp_add(v, 0x143, 0xC);   // row_bcast31     -> rows 2,3 += rows 0+1
    return v;
}

__device__ __forceinline__ int wave_max_int(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}

// Minimum over the rectangle dx in [xlo,xhi], dy in [ylo,yhi] of hA dx^2 + B dx dy + hC dy^2
// (hA, hC > 0 assumed by the caller).
__device__ __forceinline__ float min_form_on_rect(float hA, float B, float hC, float xlo, float xhi,
                                                  float ylo, float yhi) {
    if (xlo <= 0.0f && xhi >= 0.0f && ylo <= 0.0f && yhi >= 0.0f) return 0.0f;
    float best = 3.0e38f;
    const float inv2C = 0.5f / hC, inv2A = 0.5f / hA;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float dx = e ? xhi : xlo;
        const float dy = fminf(fmaxf(-B * dx * inv2C, ylo), yhi);
        best = fminf(best, hA * dx * dx + dy * (B * dx + hC * dy));
        const float ey = e ? yhi : ylo;
        const float ex = fminf(fmaxf(-B * ey * inv2A, xlo), xhi);
        best = fminf(best, hC * ey * ey + ex * (B * ey + hA * ex));
    }
    return best;
}

// sigma * log2(e) for one pixel; explicit fmas so that forward and backward (which must replay the
// forward's alpha >= 1/255 decisions) evaluate bit-identical values whatever the optimiser does.
__device__ __forceinline__ float sigma_l2(float Adx2, float Bdx, float hC, float dy) {
    return __builtin_fmaf(dy, __builtin_fmaf(hC, dy, Bdx), Adx2);
}

struct Staged {          // what a lane derives from the Gaussian it gathered
    bool keep;
    float gx, gy, hA, B, hC, lo;   // conic and opacity in the log2 domain
};

// gather + exact tile cull.  rect = pixel sample extents of the tile.
__device__ __forceinline__ Staged stage_splat(bool have, const float4 q0, const float4 q1,
                                              float X0, float X1, float Y0, float Y1) {
    Staged s;
    s.gx = q0.x; s.gy = q0.y;
    const float A = q0.w, Bc = q1.x, Cc = q1.y, op = q0.z;
    s.hA = 0.5f * kLog2e * A;
    s.B = kLog2e * Bc;
    s.hC = 0.5f * kLog2e * Cc;
    s.lo = __log2f(op);
    s.keep = have && (op > 0.0f);
    if (s.keep) {
        const float tau = s.lo + kLog2_255;            // sigma' <= tau  <=>  alpha >= 1/255
        if (tau < -0.02f) {
            s.keep = false;
        } else if (s.hA > 0.0f && s.hC > 0.0f) {
            const float xlo = s.gx - X1, xhi = s.gx - X0, ylo = s.gy - Y1, yhi = s.gy - Y0;
            const float m = min_form_on_rect(s.hA, s.B, s.hC, xlo, xhi, ylo, yhi);
            const float dxm = fmaxf(fabsf(xlo), fabsf(xhi)), dym = fmaxf(fabsf(ylo), fabsf(yhi));
            const float mag = s.hA * dxm * dxm + s.hC * dym * dym + fabsf(s.B) * dxm * dym;
            s.keep = (m <= tau + 0.02f + 4.0e-6f * mag);
        }
    }
    return s;
}

template <int CH>
__global__ __launch_bounds__(kThreads) void raster_fwd_kernel(
    const ts_camera cam, const int num_tiles, const int* __restrict__ tile_bins,
    const int* __restrict__ ids_sorted, const float4* __restrict__ splats,
    const float* __restrict__ background, float* __restrict__ out_img,
    float* __restrict__ final_Ts, int* __restrict__ final_index) {
    __shared__ float4 lds_all[kWaves][64 * 3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x * kWaves + wave;
    if (tile >= num_tiles) return;
    float4* lds = lds_all[wave];
    const int tbx = cam.tile_bounds_x;
    const int tx = tile % tbx, ty = tile / tbx + cam.tile_row0;
    const int px = tx * 16 + (lane & 15);
    const int py0 = ty * 16 + (lane >> 4);
    const float fpx = (float)px + ts::kPixOff, fpy0 = (float)py0 + ts::kPixOff;
    const float X0 = (float)(tx * 16) + ts::kPixOff, X1 = X0 + 15.0f;
    const float Y0 = (float)(ty * 16) + ts::kPixOff, Y1 = Y0 + 15.0f;
    const int W = cam.img_width, H = cam.img_height;

    float T[4], acc[4][CH];
    int fidx[4];
    bool done[4], inside[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        inside[k] = (px < W) && (py0 + 4 * k < H);
        done[k] = !inside[k];
        T[k] = 1.0f;
        fidx[k] = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[k][c] = 0.0f;
    }

    const int2 range = reinterpret_cast<const int2*>(tile_bins)[tile];
    bool all_done = __all(done[0] && done[1] && done[2] && done[3]);

    for (int base = range.x; base < range.y && !all_done; base += 64) {
        const int i = base + lane;
        const bool have = i < range.y;
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
        if (have) {
            const int g = ids_sorted[i];
            q0 = splats[3 * (size_t)g];
            q1 = splats[3 * (size_t)g + 1];
            q2 = splats[3 * (size_t)g + 2];
        }
        const Staged s = stage_splat(have, q0, q1, X0, X1, Y0, Y1);
        const unsigned long long mask = __ballot(s.keep);
        const int cnt = __popcll(mask);
        if (s.keep) {
            const int pos = __popcll(mask & ((1ull << lane) - 1ull));
            lds[3 * pos] = make_float4(s.gx, s.gy, s.hA, s.B);
            lds[3 * pos + 1] = make_float4(s.hC, s.lo, q1.z, q1.w);
            lds[3 * pos + 2] = make_float4(q2.x, q2.y, __int_as_float(i), 0.0f);
        }
        TS_WAVE_SYNC();
        for (int j = 0; j < cnt; ++j) {
            const float4 r0 = lds[3 * j], r1 = lds[3 * j + 1], r2 = lds[3 * j + 2];
            const float dx = r0.x - fpx;
            const float Adx2 = (r0.z * dx) * dx, Bdx = r0.w * dx;
            const float dy0 = r0.y - fpy0;
            const int idx = __float_as_int(r2.z);
            float col[CH];
            col[0] = r1.z; col[1] = r1.w; col[2] = r2.x;
            if (CH == 4) col[CH - 1] = r2.y;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dy = dy0 - 4.0f * (float)k;
                const float sg = sigma_l2(Adx2, Bdx, r1.x, dy);
                const float a = fminf(ts::kAlphaMax, __builtin_amdgcn_exp2f(r1.y - sg));
                const bool valid = !done[k] && (sg >= 0.0f) && (a >= ts::kAlphaMin);
                const float nT = T[k] * (1.0f - a);
                const bool stop = valid && (nT <= ts::kTEps);
                const bool hit = valid && !stop;
                const float vis = hit ? a * T[k] : 0.0f;
#pragma unroll
                for (int c = 0; c < CH; ++c) acc[k][c] += col[c] * vis;
                T[k] = hit ? nT : T[k];
                fidx[k] = hit ? idx : fidx[k];
                done[k] = done[k] || stop;
            }
            all_done = __all(done[0] && done[1] && done[2] && done[3]);
            if (all_done) break;
        }
        TS_WAVE_SYNC();
    }

    float bg[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) bg[c] = background[c];
    const int row_off = cam.tile_row0 * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!inside[k]) continue;
        const size_t pix = (size_t)(py0 + 4 * k - row_off) * W + px;
        final_Ts[pix] = T[k];
        final_index[pix] = fidx[k];
        float* o = out_img + pix * CH;
#pragma unroll
        for (int c = 0; c < CH; ++c) o[c] = acc[k][c] + T[k] * bg[c];
    }
}

template <int CH>
__global__ __launch_bounds__(kThreads) void raster_bwd_kernel(
    const ts_camera cam, const int num_tiles, const long long num_isects,
    const int* __restrict__ tile_bins, const int* __restrict__ ids_sorted,
    const float4* __restrict__ splats, const float* __restrict__ background,
    const float* __restrict__ final_Ts, const int* __restrict__ final_index,
    const float* __restrict__ v_out_img, const float* __restrict__ v_out_alpha,
    float4* __restrict__ partials) {
    __shared__ float4 lds_all[kWaves][64 * 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x * kWaves + wave;
    if (tile >= num_tiles) return;
    const int2 range = reinterpret_cast<const int2*>(tile_bins)[tile];
    if (range.y <= range.x) return;
    float4* lds = lds_all[wave];
    const int tbx = cam.tile_bounds_x;
    const int tx = tile % tbx, ty = tile / tbx + cam.tile_row0;
    const int px = tx * 16 + (lane & 15);
    const int py0 = ty * 16 + (lane >> 4);
    const float fpx = (float)px + ts::kPixOff, fpy0 = (float)py0 + ts::kPixOff;
    const float X0 = (float)(tx * 16) + ts::kPixOff, X1 = X0 + 15.0f;
    const float Y0 = (float)(ty * 16) + ts::kPixOff, Y1 = Y0 + 15.0f;
    const int W = cam.img_width, H = cam.img_height;
    const int row_off = cam.tile_row0 * 16;

    float bg[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) bg[c] = background[c];

    float T[4], tb[4], buf[4][CH], vo[4][CH];
    int fidx[4];
    int fmax = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool inside = (px < W) && (py0 + 4 * k < H);
        fidx[k] = -1;
        T[k] = 1.0f;
        tb[k] = 0.0f;
#pragma unroll
        for (int c = 0; c < CH; ++c) { buf[k][c] = 0.0f; vo[k][c] = 0.0f; }
        if (inside) {
            const size_t pix = (size_t)(py0 + 4 * k - row_off) * W + px;
            fidx[k] = final_index[pix];
            T[k] = final_Ts[pix];
            float dotbg = 0.0f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                vo[k][c] = v_out_img[pix * CH + c];
                dotbg += bg[c] * vo[k][c];
            }
            const float va = v_out_alpha ? v_out_alpha[pix] : 0.0f;
            tb[k] = T[k] * (va - dotbg);
        }
        fmax = max(fmax, fidx[k]);
    }
    fmax = wave_max_int(fmax);
    const int last = min(range.y - 1, fmax);

    for (int hi = last; hi >= range.x; hi -= 64) {
        const int i = hi - lane;
        const bool have = i >= range.x;
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
        if (have) {
            const int g = ids_sorted[i];
            q0 = splats[3 * (size_t)g];
            q1 = splats[3 * (size_t)g + 1];
            q2 = splats[3 * (size_t)g + 2];
        }
        const Staged s = stage_splat(have, q0, q1, X0, X1, Y0, Y1);
        const unsigned long long mask = __ballot(s.keep);
        const int cnt = __popcll(mask);
        if (s.keep) {
            const int pos = __popcll(mask & ((1ull << lane) - 1ull));
            const int slot = __float_as_int(q2.z) + ty * __float_as_int(q2.w) + tx;
            lds[4 * pos] = make_float4(s.gx, s.gy, s.hA, s.B);
            lds[4 * pos + 1] = make_float4(s.hC, s.lo, q1.z, q1.w);
            lds[4 * pos + 2] = make_float4(q2.x, q2.y, __int_as_float(i), __int_as_float(slot));
            lds[4 * pos + 3] = make_float4(q0.w, q1.x, q1.y, q0.z);      // A, B, C, opacity
        }
        TS_WAVE_SYNC();
        for (int j = 0; j < cnt; ++j) {
            const float4 r0 = lds[4 * j], r1 = lds[4 * j + 1], r2 = lds[4 * j + 2];
            const float dx = r0.x - fpx;
            const float Adx2 = (r0.z * dx) * dx, Bdx = r0.w * dx;
            const float dy0 = r0.y - fpy0;
            const int idx = __float_as_int(r2.z);
            float col[CH];
            col[0] = r1.z; col[1] = r1.w; col[2] = r2.x;
            if (CH == 4) col[CH - 1] = r2.y;

            float araw[4], dyv[4];
            bool valid[4];
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dyv[k] = dy0 - 4.0f * (float)k;
                const float sg = sigma_l2(Adx2, Bdx, r1.x, dyv[k]);
                araw[k] = __builtin_amdgcn_exp2f(r1.y - sg);                 // opacity * exp(-sigma)
                valid[k] = (idx <= fidx[k]) && (sg >= 0.0f) &&
                           (fminf(ts::kAlphaMax, araw[k]) >= ts::kAlphaMin);
                any = any || valid[k];
            }
            if (!__any(any)) continue;

            float s_ = 0.0f, sy = 0.0f, syy = 0.0f, vc[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) vc[c] = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (valid[k]) {
                    const float a = fminf(ts::kAlphaMax, araw[k]);
                    const float ra = __builtin_amdgcn_rcpf(1.0f - a);
                    const float Tk = T[k] * ra;                 // transmittance in front of g
                    const float fac = a * Tk;
                    float v_a = ra * tb[k];
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        vc[c] += fac * vo[k][c];
                        v_a += (col[c] * Tk - buf[k][c] * ra) * vo[k][c];
                        buf[k][c] += col[c] * fac;
                    }
                    T[k] = Tk;
                    // d alpha / d sigma = -araw unless the 0.999 clamp is active (then 0)
                    const float v_sig = (araw[k] > ts::kAlphaMax) ? 0.0f : -araw[k] * v_a;
                    s_ += v_sig;
                    sy += v_sig * dyv[k];
                    syy += v_sig * dyv[k] * dyv[k];
                }
            }
            float red[6 + CH];
            red[0] = wave_sum_hi(s_);
            red[1] = wave_sum_hi(dx * s_);
            red[2] = wave_sum_hi(sy);
            red[3] = wave_sum_hi(dx * dx * s_);
            red[4] = wave_sum_hi(dx * sy);
            red[5] = wave_sum_hi(syy);
#pragma unroll
            for (int c = 0; c < CH; ++c) red[6 + c] = wave_sum_hi(vc[c]);
            if (lane == 63) {
                const float4 r3 = lds[4 * j + 3];
                const long long slot = (long long)__float_as_int(r2.w);
                if (slot >= 0 && slot < num_isects) {
                    const float v_x = r3.x * red[1] + r3.y * red[2];
                    const float v_y = r3.y * red[1] + r3.z * red[2];
                    const float v_op = -red[0] / r3.w;
                    float4* row = partials + 3 * slot;
                    row[0] = make_float4(v_x, v_y, v_op, 0.5f * red[3]);
                    row[1] = make_float4(red[4], 0.5f * red[5], red[6], red[7]);
                    row[2] = make_float4(red[8], CH == 4 ? red[6 + CH - 1] : 0.0f, 0.0f, 0.0f);
                }
            }
        }
        TS_WAVE_SYNC();
    }
}

template <int CH>
__global__ __launch_bounds__(256) void reduce_partials_kernel(
    int n, const int* __restrict__ num_tiles_hit, const int* __restrict__ cum_tiles_hit,
    const float4* __restrict__ partials, float* __restrict__ v_xy, float* __restrict__ v_conic,
    float* __restrict__ v_colors, float* __restrict__ v_opacity) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int cnt = num_tiles_hit[i];
    const long long end = cum_tiles_hit[i];
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
    for (long long s = end - cnt; s < end; ++s) {
        const float4 p0 = partials[3 * s], p1 = partials[3 * s + 1], p2 = partials[3 * s + 2];
        a0.x += p0.x; a0.y += p0.y; a0.z += p0.z; a0.w += p0.w;
        a1.x += p1.x; a1.y += p1.y; a1.z += p1.z; a1.w += p1.w;
        a2.x += p2.x; a2.y += p2.y;
    }
    reinterpret_cast<float2*>(v_xy)[i] = make_float2(a0.x, a0.y);
    v_opacity[i] = a0.z;
    v_conic[3 * i] = a0.w; v_conic[3 * i + 1] = a1.x; v_conic[3 * i + 2] = a1.y;
    if (CH == 4) {
        reinterpret_cast<float4*>(v_colors)[i] = make_float4(a1.z, a1.w, a2.x, a2.y);
    } else {
        v_colors[3 * i] = a1.z; v_colors[3 * i + 1] = a1.w; v_colors[3 * i + 2] = a2.x;
    }
}

inline int launch_status() { return (int)hipGetLastError(); }

}  // namespace

extern "C" {

int ts_raster_fwd(int32_t channels, const ts_camera* cam, const int32_t* tile_bins,
                  const int32_t* gaussian_ids_sorted, const float* splats, const float* background,
                  float* out_img, float* final_Ts, int32_t* final_index, void* stream) {
    if (!cam || (channels != 3 && channels != 4)) return TS_E_BADARG;
    const int nt = cam->tile_rows * cam->tile_bounds_x;
    if (nt <= 0) return 0;
    if (!tile_bins || !background || !out_img || !final_Ts || !final_index) return TS_E_BADARG;
    const int grid = (nt + kWaves - 1) / kWaves;
    hipStream_t s = (hipStream_t)stream;
    const float4* sp = reinterpret_cast<const float4*>(splats);
    if (channels == 3)
        hipLaunchKernelGGL(raster_fwd_kernel<3>, dim3(grid), dim3(kThreads), 0, s, *cam, nt,
                           tile_bins, gaussian_ids_sorted, sp, background, out_img, final_Ts,
                           final_index);
    else
        hipLaunchKernelGGL(raster_fwd_kernel<4>, dim3(grid), dim3(kThreads), 0, s, *cam, nt,
                           tile_bins, gaussian_ids_sorted, sp, background, out_img, final_Ts,
                           final_index);
    return launch_status();
}

int ts_raster_bwd(int32_t channels, int64_t num_intersects, const ts_camera* cam,
                  const int32_t* tile_bins, const int32_t* gaussian_ids_sorted, const float* splats,
                  const float* background, const float* final_Ts, const int32_t* final_index,
                  const float* v_out_img, const float* v_out_alpha, float* partials, void* stream) {
    if (!cam || (channels != 3 && channels != 4) || num_intersects < 0) return TS_E_BADARG;
    const int nt = cam->tile_rows * cam->tile_bounds_x;
    if (nt <= 0 || num_intersects == 0) return 0;
    if (!tile_bins || !gaussian_ids_sorted || !splats || !background || !final_Ts || !final_index ||
        !v_out_img || !partials)
        return TS_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(partials, 0,
                                  (size_t)num_intersects * TS_PARTIAL_ROW_FLOATS * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    const int grid = (nt + kWaves - 1) / kWaves;
    const float4* sp = reinterpret_cast<const float4*>(splats);
    float4* pr = reinterpret_cast<float4*>(partials);
    if (channels == 3)
        hipLaunchKernelGGL(raster_bwd_kernel<3>, dim3(grid), dim3(kThreads), 0, s, *cam, nt,
                           (long long)num_intersects, tile_bins, gaussian_ids_sorted, sp,
                           background, final_Ts, final_index, v_out_img, v_out_alpha, pr);
    else
        hipLaunchKernelGGL(raster_bwd_kernel<4>, dim3(grid), dim3(kThreads), 0, s, *cam, nt,
                           (long long)num_intersects, tile_bins, gaussian_ids_sorted, sp,
                           background, final_Ts, final_index, v_out_img, v_out_alpha, pr);
    return launch_status();
}

int ts_reduce_partials(int32_t n, int32_t channels, const int32_t* num_tiles_hit,
                       const int32_t* cum_tiles_hit, const float* partials, float* v_xy,
                       float* v_conic, float* v_colors, float* v_opacity, void* stream) {
    if (n < 0 || (channels != 3 && channels != 4)) return TS_E_BADARG;
    if (n == 0) return 0;
    if (!num_tiles_hit || !cum_tiles_hit || !v_xy || !v_conic || !v_colors || !v_opacity)
        return TS_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const float4* pr = reinterpret_cast<const float4*>(partials);
    const int grid = (n + 255) / 256;
    if (channels == 3)
        hipLaunchKernelGGL(reduce_partials_kernel<3>, dim3(grid), dim3(256), 0, s, n, num_tiles_hit,
                           cum_tiles_hit, pr, v_xy, v_conic, v_colors, v_opacity);
    else
        hipLaunchKernelGGL(reduce_partials_kernel<4>, dim3(grid), dim3(256), 0, s, n, num_tiles_hit,
                           cum_tiles_hit, pr, v_xy, v_conic, v_colors, v_opacity);
    return launch_status();
}

}  // extern "C"
