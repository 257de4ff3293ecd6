// raster.hip - tile compositing kernels for gfx950 (forward, backward, gradient row reduce).
//
// What gsplat's rasterize_forward / rasterize_backward compute for tinysplat's calls at
// /root/reference/tinysplat/splatting/rasterize.py:44,50 (front-to-back alpha compositing of the
// depth-sorted per-tile lists, and the back-to-front replay that produces v_xy / v_conic /
// v_colors / v_opacity), re-designed around the 64-wide wavefront.  Both kernels are VALU-issue bound
// (one wave64 VALU instruction = 4 cycles of a SIMD), so the design minimises instructions per listed
// (tile, Gaussian) pair:
//
//   * ONE WAVE OWNS ONE 16x16 TILE, seen as four 8x8 pixel blocks.  Lane l sits at (l & 7, l >> 3)
//     of every block: four pixels per lane, and one VALU instruction covers exactly one block, so
//     a block is the unit of skipping.  No workgroup barrier anywhere: the four waves of a
//     256-thread workgroup run four different tiles independently; early-out is a wave ballot.
//   * Per 64-entry chunk of the tile's sorted list each lane gathers ONE Gaussian's 48-byte packed
//     record and tests it exactly against the pixels of each block that still matter (minimum of the
//     conic form over their bounding rectangle vs. the alpha >= 1/255 level set - conservative, so
//     results are unchanged).
//   * BLOCK RECORDS WITH A POLYNOMIAL EXPONENT.  For every (Gaussian, block) that survives, the
//     Gaussian's lane expands  sigma*log2e - log2(opacity)  around the block centre:
//         s(X, Y) = c0 + c1 X + c2 Y + c3 X^2 + c4 XY + c5 Y^2,   X, Y = lane position - 3.5,
//     and appends {c0..c5, colour, list index} to an LDS list of that block (ballot + prefix count).
//     X, Y, X^2, XY, Y^2 are five per-lane constants of the whole kernel, so the inner loop evaluates
//     the exponent with 5 FMAs on wave-uniform (LDS-broadcast) coefficients and nothing else has to be
//     set up per entry and lane (the previous design spent 17 issues per entry on dx, dy and the
//     partial products of both halves of the tile, whether or not a half was touched).  The four blocks
//     are walked one after the other ("k-major"): pixels of different blocks are independent, and
//     inside a block the list order is preserved.
//   * Per-pixel bodies are branch free inside a block and lean: log2(opacity) is folded into c0, the
//     compositing weight is T_old - T_new (telescoping), a finished pixel is marked by the sign of
//     T, and the sigma >= 0 / 0.999-clamp tests exist only in the instantiation used when a staged
//     Gaussian can trigger them.
//   * BACKWARD WITHOUT A PER-ENTRY CROSS-LANE REDUCTION.  The gradient of a (tile, Gaussian) is a
//     sum over pixels, and pixels are lanes: the previous design reduced nine values over the wave
//     per entry with a merged DPP butterfly (~37 issues per entry, the largest single cost).  Now
//     pass A (lane = pixel) only resolves the transmittance chain and stores two numbers per pixel -
//     v_sigma and alpha*T - as one 64-float line per (Gaussian, block) in LDS; every 16 lines, pass B
//     runs TRANSPOSED (lane = line x pixel quarter): it reads its 16 pixels of both lines and
//     accumulates the six geometric moments (against compile-time pixel offsets) and the colour sums
//     (against the block's v_out, kept in an LDS table) in registers, converts the moments to the
//     Gaussian-centred sums of the row format, combines the four quarters with 2-issue
//     v_permlane16/32_swap merges and writes 16 rows with three stores.  ~9 issues per line instead
//     of ~30, no float atomics, and the gradients stay bit-reproducible run to run.
//   * One partial row per (tile, Gaussian, block): slot 4 s + k.  reduce_partials sums each
//     Gaussian's flagged rows in a fixed order and applies the conic / opacity factors once.
#include <hip/hip_runtime.h>

#include "../../include/tinysplat_hip.h"
#include "splat_math.h"

namespace {

// tuning knobs (overridable with -D for the ablation runs of tools/time_raster.py)
#ifndef TS_FWD_MIN_WAVES
#define TS_FWD_MIN_WAVES 1             // __launch_bounds__ 2nd argument = min waves per SIMD
#endif
#ifndef TS_BWD_MIN_WAVES
#define TS_BWD_MIN_WAVES 1
#endif

#ifndef TS_ABLATE
#define TS_ABLATE 0                    // timing experiments only (results are wrong when != 0)
#endif

#ifndef TS_RASTER_WAVES
#define TS_RASTER_WAVES 4
#endif
constexpr int kWaves = TS_RASTER_WAVES;   // tiles (= waves) per workgroup; waves never synchronise
constexpr int kThreads = 64 * kWaves;
constexpr int kRows = 16;                 // (Gaussian, block) lines per pass B of the backward kernel
constexpr int kRowStride = 68;            // floats per line: 4 * odd -> the strided b128 reads of pass B
                                          // start in 16 different bank groups
constexpr int kVoStride = 65;             // float4s per block of the v_out table (bank stagger)
using ts::kLog2e;
using ts::kLog2_255;

#define TS_WAVE_SYNC()                                            \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
    } while (0)

__device__ __forceinline__ int wave_max_int(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}

// XCD-aware workgroup -> tile-group mapping.  Workgroup b is observed to run on XCD (b % 8), and
// every XCD has a private 4 MiB L2.  Handing XCD x the x-th contiguous eighth of the tile groups
// (a band of tile rows) keeps the packed-record gathers of neighbouring tiles, which share most of
// their Gaussians, in one L2.  Placement only affects speed.  Grid = 8 * ceil(groups / 8).
#ifndef TS_XCD_MAP
#define TS_XCD_MAP 1
#endif
__device__ __forceinline__ int xcd_tile_group(int num_groups) {
    if (!TS_XCD_MAP) return (int)blockIdx.x;
    const int per_xcd = (num_groups + 7) >> 3;
    return (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
}

// Bounding rectangle (in lane coordinates 0..7) of the pixels of an 8x8 block selected by a 64-bit
// lane mask (bit = ly*8 + lx).  Pure scalar bit arithmetic on the ballot.  Returns false if empty.
__device__ __forceinline__ bool mask_rect(unsigned long long m, int& xmin, int& xmax, int& ymin,
                                          int& ymax) {
    if (m == 0ull) return false;
    ymin = __builtin_ctzll(m) >> 3;
    ymax = (63 - __builtin_clzll(m)) >> 3;
    unsigned int c = (unsigned int)(m | (m >> 32));
    c |= c >> 16;
    c |= c >> 8;
    c &= 0xffu;
    xmin = __builtin_ctz(c);
    xmax = 31 - __builtin_clz(c);
    return true;
}

// Writes rects[k] = {x0, x1, y0, y1} (sample-position bounds, inclusive) of the pixels selected by
// `sel[k]` (one bool per lane and block) and returns the mask of non-empty blocks.  Lane 0 stores;
// callers fence before reading.
__device__ __forceinline__ int update_rects(const bool sel[4], float X0, float Y0, float4* rects,
                                            int lane) {
    int blocks = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long m = __ballot(sel[k]);
        int xmin, xmax, ymin, ymax;
        if (mask_rect(m, xmin, xmax, ymin, ymax)) {
            blocks |= (1 << k);
            const float bx = X0 + (float)(8 * (k & 1)), by = Y0 + (float)(8 * (k >> 1));
            if (lane == 0)
                rects[k] = make_float4(bx + (float)xmin, bx + (float)xmax, by + (float)ymin,
                                       by + (float)ymax);
        }
    }
    return blocks;
}

using mask64 = unsigned long long;
#define TS_BALLOT(c) __builtin_amdgcn_ballot_w64(c)          // lane condition -> 64-bit scalar mask
#define TS_LANE(m) __builtin_amdgcn_inverse_ballot_w64(m)    // scalar mask -> lane condition

// Per-lane constants: the lane's sample position relative to the centre of its 8x8 block (the same
// in all four blocks) and the monomials the exponent polynomial needs.
struct Geom { float X, Y, XX, XY, YY; };
__device__ __forceinline__ Geom lane_geom(int lane) {
    Geom g;
    g.X = (float)(lane & 7) - 3.5f;
    g.Y = (float)(lane >> 3) - 3.5f;
    g.XX = g.X * g.X; g.XY = g.X * g.Y; g.YY = g.Y * g.Y;
    return g;
}

// What a lane derives once per chunk from the Gaussian it gathered (lane = list entry).
struct Staged {
    bool cand;           // may contribute at all (opacity > 0, level set not empty)
    bool psd;            // positive-definite conic: the geometric cull applies
    bool general;        // the per-pixel code must test sigma >= 0 and apply the 0.999 clamp
    float gx, gy, hA, B, hC, lo, tau, inv2A, inv2C;
};

__device__ __forceinline__ Staged stage_gaussian(bool have, const float4 q0, const float4 q1) {
    Staged s;
    s.gx = q0.x; s.gy = q0.y;
    const float op = q0.z;
    s.hA = 0.5f * kLog2e * q0.w;
    s.B = kLog2e * q1.x;
    s.hC = 0.5f * kLog2e * q1.y;
    s.lo = __log2f(op);
    s.tau = s.lo + kLog2_255;                      // sigma' <= tau  <=>  alpha >= 1/255
    s.psd = s.hA > 0.0f && s.hC > 0.0f;
    s.inv2A = 0.5f / s.hA; s.inv2C = 0.5f / s.hC;
    // For a positive-definite conic and opacity <= 0.99 neither the sigma >= 0 test nor the 0.999
    // clamp can trigger (sigma >= 0 up to rounding, alpha <= opacity): the lean loop is exact.
    s.general = !(s.psd && 4.0f * s.hA * s.hC > s.B * s.B && op <= 0.99f);
    s.cand = have && op > 0.0f && s.tau >= -0.02f;
    return s;
}

// Can the Gaussian reach alpha >= 1/255 inside the rectangle r = {x0, x1, y0, y1} of a block?
__device__ __forceinline__ bool block_hit(const Staged& s, const float4 r) {
    if (!s.cand) return false;
    if (!s.psd) return true;                       // not a PSD conic: no geometric cull
    return ts::rect_may_contribute(s.hA, s.B, s.hC, s.inv2A, s.inv2C, s.tau, s.gx, s.gy, r.x, r.y, r.z, r.w);
}

// Expansion of  hA dx^2 + B dx dy + hC dy^2 - log2(opacity),  d = xy - pixel,  around a block centre:
// with (dxc, dyc) = xy - centre and pixel = centre + (X, Y):  dx = dxc - X, dy = dyc - Y.
struct Coefs { float c0, c1, c2; };
__device__ __forceinline__ Coefs block_coefs(const Staged& s, float dxc, float dyc) {
#pragma clang fp contract(off)
    Coefs c;
    const float t = s.B * dxc;
    c.c0 = ((s.hA * dxc) * dxc + t * dyc) + ((s.hC * dyc) * dyc - s.lo);
    c.c1 = -(2.0f * (s.hA * dxc) + s.B * dyc);
    c.c2 = -(2.0f * (s.hC * dyc) + t);
    return c;
}

// s(X, Y) of one pixel: explicit FMAs in a fixed order, so that forward and backward (which must
// replay the forward's alpha >= 1/255 decisions) evaluate bit-identical values.
__device__ __forceinline__ float poly_sigma(const Geom& g, const float4 r0, const float4 r1) {
    float s = __builtin_fmaf(g.X, r0.y, r0.x);
    s = __builtin_fmaf(g.Y, r0.z, s);
    s = __builtin_fmaf(g.XX, r0.w, s);
    s = __builtin_fmaf(g.XY, r1.x, s);
    return __builtin_fmaf(g.YY, r1.y, s);
}

// Composites the `cnt` records of one block into its 64 pixels (forward).
//   T > 0 = transmittance of an unfinished pixel; T < 0 = finished (or outside the image) with final
//   transmittance |T|; fidx = list index of the last Gaussian composited; acc = colour.
// record = { c0 c1 c2 c3 | c4 c5 col0 col1 | col2 col3 idx log2(opacity) }
// vis = |T_old| - |T_new| equals alpha*T up to one rounding of T, is 0 for the stopping Gaussian
// and for finished pixels, and makes the weights telescope (sum of vis = 1 - T_final exactly).
template <int CH, bool GENERAL>
__device__ __forceinline__ void fwd_records(const float4* __restrict__ lds, int cnt, const Geom g,
                                            float& T, int& fidx, float (&acc)[CH]) {
#pragma clang fp contract(off)          // both instantiations must round alike
    for (int j = 0; j < (TS_ABLATE == 3 ? 0 : cnt); ++j) {
        const float4 r0 = lds[3 * j], r1 = lds[3 * j + 1], r2 = lds[3 * j + 2];
        const int idx = __float_as_int(r2.z);
        // sgl = sigma*log2(e) - log2(opacity), so alpha = exp2(-sgl)
        const float sgl = poly_sigma(g, r0, r1);
        float a = __builtin_amdgcn_exp2f(-sgl);
        bool ok = a >= ts::kAlphaMin;
        if (GENERAL) {
            a = fminf(ts::kAlphaMax, a);
            ok = ok & (sgl >= -r2.w);                                 // sigma >= 0
        }
        // A finished pixel (T < 0) needs no test of its own: nT = T (1 - ae) stays negative, so
        // either `stop` fires and -|T| puts T back, or ae = 0 and nT = T; vis is 0 both ways.
        const float ae = ok ? a : 0.0f;
        const float nT = __builtin_fmaf(-ae, T, T);
        const bool stop = (nT <= ts::kTEps) & ok;
        const float Tn = stop ? -__builtin_fabsf(T) : nT;            // the stopping Gaussian is not composited
        const float vis = __builtin_fabsf(T) - __builtin_fabsf(Tn);
        acc[0] = __builtin_fmaf(r1.z, vis, acc[0]);
        acc[1] = __builtin_fmaf(r1.w, vis, acc[1]);
        acc[2] = __builtin_fmaf(r2.x, vis, acc[2]);
        if (CH == 4) acc[CH - 1] = __builtin_fmaf(r2.y, vis, acc[CH - 1]);
        fidx = (ok & !stop) ? idx : fidx;
        T = Tn;
    }
}

// Pixel layout of a wave: lane l -> (lx, ly) = (l & 7, l >> 3) inside an 8x8 block; the lane owns
// that position in each of the four blocks k of the 16x16 tile (block k: bx = k & 1, by = k >> 1).
// SPLIT: four waves per tile, each owning ONE 8x8 block (the other three count as outside the image).
// Same list, same arithmetic per pixel; used when a launch has fewer tiles than the GPU has SIMDs (a
// tile-row stripe of a multi-GPU frame, a small image), where one wave per tile leaves the vector ALUs
// without a second wave to switch to and the launch takes as long as the longest tile list.
template <int CH, bool SPLIT>
__global__ __launch_bounds__(kThreads, TS_FWD_MIN_WAVES) void raster_fwd_kernel(
    const ts_camera cam, const int num_tiles, const int* __restrict__ tile_bins,
    const int* __restrict__ ids_sorted, const float4* __restrict__ splats,
    const float* __restrict__ background, float* __restrict__ out_img,
    float* __restrict__ final_Ts, int* __restrict__ final_index, const int clamp_rgb,
    unsigned char* __restrict__ clamp_mask) {
    __shared__ float4 lds_all[kWaves][64 * 3];
    __shared__ float4 rect_all[kWaves][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int units = SPLIT ? 4 * num_tiles : num_tiles;
    const int unit = xcd_tile_group((units + kWaves - 1) / kWaves) * kWaves + wave;
    if (unit >= units) return;
    const int tile = SPLIT ? unit >> 2 : unit;
    const int only = SPLIT ? unit & 3 : -1;
    float4* lds = lds_all[wave];
    float4* rects = rect_all[wave];
    const int tbx = cam.tile_bounds_x;
    const int tx = tile % tbx, ty = tile / tbx + cam.tile_row0;
    const int px0 = tx * 16 + (lane & 7), py0 = ty * 16 + (lane >> 3);
    const float X0 = (float)(tx * 16) + ts::kPixOff, Y0 = (float)(ty * 16) + ts::kPixOff;
    const int W = cam.img_width, H = cam.img_height;
    const Geom geom = lane_geom(lane);

    // T > 0: transmittance of an unfinished pixel; T < 0: finished or outside, final value |T|
    float T[4], acc[4][CH];
    int fidx[4];
    bool inside[4];
    int live = 0;                                   // blocks that still have unfinished pixels
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        inside[k] = (px0 + 8 * (k & 1) < W) && (py0 + 8 * (k >> 1) < H) && (!SPLIT || k == only);
        T[k] = inside[k] ? 1.0f : -1.0f;
        fidx[k] = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[k][c] = 0.0f;
        if (__any(inside[k])) live |= (1 << k);
    }

    const int2 range = reinterpret_cast<const int2*>(tile_bins)[tile];

    // Software pipeline over 64-entry chunks: the id of chunk c+2 and the packed record of chunk
    // c+1 are in flight while chunk c is composited (two dependent gathers = ~2 us of latency
    // that a wave with few co-resident waves per SIMD cannot hide otherwise).
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 n0 = zero4, n1 = zero4, n2 = zero4;
    int id_next = 0;
    if (range.x + lane < range.y) {
        const int g = ids_sorted[range.x + lane];
        n0 = splats[3 * (size_t)g]; n1 = splats[3 * (size_t)g + 1]; n2 = splats[3 * (size_t)g + 2];
    }
    if (range.x + 64 + lane < range.y) id_next = ids_sorted[range.x + 64 + lane];

    for (int base = range.x; base < range.y && live != 0; base += 64) {
        const int i = base + lane;
        const bool have = i < range.y;
        const float4 q0 = n0, q1 = n1, q2 = n2;
        if (i + 64 < range.y) {
            const int g = id_next;
            n0 = splats[3 * (size_t)g]; n1 = splats[3 * (size_t)g + 1]; n2 = splats[3 * (size_t)g + 2];
        }
        if (i + 128 < range.y) id_next = ids_sorted[i + 128];
        {   // rectangle of the still-unfinished pixels of each block: saturated pixels need no more
            // Gaussians, so late in the list most (Gaussian, block) pairs are culled here
            bool sel[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) sel[k] = T[k] > 0.0f;
            live = update_rects(sel, X0, Y0, rects, lane);
            TS_WAVE_SYNC();
            if (live == 0) break;
        }
        const Staged s = stage_gaussian(have, q0, q1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!(live & (1 << k))) continue;                         // wave-uniform
            const bool hit = block_hit(s, rects[k]);
            const mask64 mask = __ballot(hit);
            if (mask == 0ull) continue;
            const int cnt = __popcll(mask);
            if (hit) {
                const float bcx = X0 + (float)(8 * (k & 1)) + 3.5f, bcy = Y0 + (float)(8 * (k >> 1)) + 3.5f;
                const Coefs c = block_coefs(s, s.gx - bcx, s.gy - bcy);
                const int pos = __popcll(mask & ((1ull << lane) - 1ull));
                lds[3 * pos] = make_float4(c.c0, c.c1, c.c2, s.hA);
                lds[3 * pos + 1] = make_float4(s.B, s.hC, q1.z, q1.w);
                lds[3 * pos + 2] = make_float4(q2.x, q2.y, __int_as_float(i), s.lo);
            }
            TS_WAVE_SYNC();
            // the general per-pixel code is chosen once per block list, so that the common case runs
            // a loop without the sigma >= 0 / clamp tests
            if (__ballot(hit && s.general) != 0ull)
                fwd_records<CH, true>(lds, cnt, geom, T[k], fidx[k], acc[k]);
            else
                fwd_records<CH, false>(lds, cnt, geom, T[k], fidx[k], acc[k]);
            TS_WAVE_SYNC();
        }
    }

    float bg[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) bg[c] = background[c];
    const int row_off = cam.tile_row0 * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!inside[k]) continue;
        const size_t pix = (size_t)(py0 + 8 * (k >> 1) - row_off) * W + (px0 + 8 * (k & 1));
        const float Tf = __builtin_fabsf(T[k]);
        if (final_Ts) {             // null in the forward-only (viewer) mode: nothing is kept for backward
            final_Ts[pix] = Tf;
            final_index[pix] = fidx[k];
        }
        float* o = out_img + pix * CH;
        int pass = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float val = acc[k][c] + Tf * bg[c];
            if (clamp_rgb && c < 3) {          // the adapter's clamp(rgb, max=1), rasterize.py:45
                pass |= (val <= 1.0f) ? (1 << c) : 0;          // torch's rule: gradient passes at x <= max
                val = fminf(val, 1.0f);
            }
            o[c] = val;
        }
        if (clamp_mask) clamp_mask[pix] = (unsigned char)pass;
    }
}

// gfx950 row / half swaps: v_permlane16_swap exchanges the odd 16-lane rows of x with the even rows of
// y, v_permlane32_swap the upper half of x with the lower half of y; x' + y' is then a butterfly MERGE
// of two values across rows in 2 VALU issues (DPP cannot cross rows at all).
__device__ __forceinline__ float swap16_add(float x, float y) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);   // rows: [x0+x1, y0+y1, x2+x3, y2+y3]
}
__device__ __forceinline__ float swap32_add(float x, float y) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);   // halves: [x_lo+x_hi, y_lo+y_hi]
}

// Pass B of the backward kernel: turns up to kRows (Gaussian, block) lines of per-pixel {v_sigma,
// alpha*T} into partial rows.  Lane = (line r = lane & 15, pixel quarter q = lane >> 4): the lane sums
// the 16 pixels (two pixel rows) of its quarter.
//   geometric part: moments of v against the pixel offsets (X = lx - 3.5 is a compile-time constant
//     per step; the two pixel rows of a quarter are kept apart and shifted to the block-centred Y
//     afterwards), then converted to the row format's Gaussian-centred sums with dx = dxc - X,
//     dy = dyc - Y (dxc, dyc = Gaussian centre - block centre, from the line's meta record);
//   colour part: sum of alpha*T times the pixel's v_out, read from the wave's LDS table of its tile;
//   the four quarters are merged with v_permlane16/32_swap (2 issues per merge), after which the lane
//     of quarter p holds values p, 4 + p and 8 + (p & 1) of line r, and stores them.
// meta[r] = { dxc, dyc, row slot (int), block k (int) }.
template <int CH>
__device__ __forceinline__ void pass_b(const float* __restrict__ vbuf, const float* __restrict__ fbuf,
                                       const float4* __restrict__ vo_tab, const float4* __restrict__ meta,
                                       int nrows, float* __restrict__ partials,
                                       unsigned char* __restrict__ row_flags, int lane) {
    const int r = lane & 15, q = lane >> 4;
    const float4 m = meta[r];
    const float dxc = m.x, dyc = m.y;
    const int slot = __float_as_int(m.z), k = __float_as_int(m.w) & 3;
    const float4* vl = reinterpret_cast<const float4*>(vbuf + r * kRowStride + 16 * q);
    const float4* fl = reinterpret_cast<const float4*>(fbuf + r * kRowStride + 16 * q);
    const float4* vo = vo_tab + k * kVoStride + 16 * q;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;   // pixel row 2q | 2q + 1
    float col[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) col[c] = 0.0f;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float4 v4 = vl[h], f4 = fl[h];
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w}, ff[4] = {f4.x, f4.y, f4.z, f4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int s = 4 * h + e;
            const float X = (float)(s & 7) - 3.5f;
            if (s < 8) {
                a0 += vv[e];
                a1 = __builtin_fmaf(vv[e], X, a1);
                a2 = __builtin_fmaf(vv[e], X * X, a2);
            } else {
                b0 += vv[e];
                b1 = __builtin_fmaf(vv[e], X, b1);
                b2 = __builtin_fmaf(vv[e], X * X, b2);
            }
            const float4 w = vo[s];
            col[0] = __builtin_fmaf(ff[e], w.x, col[0]);
            col[1] = __builtin_fmaf(ff[e], w.y, col[1]);
            col[2] = __builtin_fmaf(ff[e], w.z, col[2]);
            if (CH == 4) col[CH - 1] = __builtin_fmaf(ff[e], w.w, col[CH - 1]);
        }
    }
    // block-centred moments of this quarter: Y = y' + Yq with y' in {0, 1}
    const float Yq = (float)(2 * q) - 3.5f;
    const float M0 = a0 + b0, Mx = a1 + b1, Mxx = a2 + b2;
    const float My = __builtin_fmaf(Yq, M0, b0);
    const float Mxy = __builtin_fmaf(Yq, Mx, b1);
    const float Myy = __builtin_fmaf(Yq, __builtin_fmaf(Yq, M0, 2.0f * b0), b0);
    // Gaussian-centred sums (dx = dxc - X, dy = dyc - Y):  S v, S v dx, S v dy, S v dx^2, S v dx dy, S v dy^2
    float val[10];
    val[0] = M0;
    val[1] = __builtin_fmaf(dxc, M0, -Mx);
    val[2] = __builtin_fmaf(dyc, M0, -My);
    val[3] = __builtin_fmaf(dxc, __builtin_fmaf(dxc, M0, -2.0f * Mx), Mxx);
    val[4] = __builtin_fmaf(dxc, __builtin_fmaf(dyc, M0, -My), __builtin_fmaf(-dyc, Mx, Mxy));
    val[5] = __builtin_fmaf(dyc, __builtin_fmaf(dyc, M0, -2.0f * My), Myy);
#pragma unroll
    for (int c = 0; c < 4; ++c) val[6 + c] = c < CH ? col[c] : 0.0f;
    // merge the four quarters: quarter p ends with values p, 4 + p, 8 + (p & 1)
    const float q0 = swap32_add(swap16_add(val[0], val[1]), swap16_add(val[2], val[3]));
    const float q1 = swap32_add(swap16_add(val[4], val[5]), swap16_add(val[6], val[7]));
    const float t = swap16_add(val[8], CH == 4 ? val[9] : val[8]);
    const float q2 = t + __shfl_xor(t, 32, 64);
    if (r < nrows) {
        float* p = partials + (size_t)slot * TS_PARTIAL_ROW_FLOATS;
        p[q] = q0;
        p[4 + q] = q1;
        if (q < 2 && 8 + q < 6 + CH) p[8 + q] = q2;
        if (q == 3) row_flags[slot] = 1;                       // this row now holds data
    }
}

// Pass A of the backward kernel for the records of one block, record index j0 .. cnt-1 (records are
// staged back to front).  Resolves the transmittance chain per pixel and stores one line per record
// with a contributing pixel; stops when kRows lines are pending and returns the next record index.
//   T = transmittance behind the Gaussian being replayed, R = T_final * (v_alpha - bg . v_out) -
//   sum over the Gaussians already replayed of fac * (colour . v_out), vo = v_out, fidx = index of
//   the last Gaussian the forward pass composited.
// record = { c0 c1 c2 c3 | c4 c5 col0 col1 | col2 col3 idx log2(opacity) | dxc dyc slot k }
// A lane that is not valid uses alpha = 0 (ra = 1, fac = 0, v_sig = 0) and changes nothing.
template <int CH, bool GENERAL>
__device__ __forceinline__ int bwd_records(const float4* __restrict__ lds, int j0, int cnt, const Geom g,
                                           float& T, float& R, const float (&vo)[CH], const int fidx,
                                           float* __restrict__ vbuf, float* __restrict__ fbuf,
                                           float4* __restrict__ meta, int& nrows, int lane) {
    // Every fused multiply-add below is written out; implicit contraction is switched off so that the
    // GENERAL and the lean instantiation round identically.
#pragma clang fp contract(off)
    int j = j0;
    for (; j < (TS_ABLATE == 3 ? 0 : cnt); ++j) {
        const float4 r0 = lds[4 * j], r1 = lds[4 * j + 1], r2 = lds[4 * j + 2];
        const int idx = __float_as_int(r2.z);
        const float sgl = poly_sigma(g, r0, r1);
        const float araw = __builtin_amdgcn_exp2f(-sgl);             // opacity * exp(-sigma)
        float a = araw;
        mask64 validm = TS_BALLOT(araw >= ts::kAlphaMin) & TS_BALLOT(idx <= fidx);
        if (GENERAL) {
            a = fminf(ts::kAlphaMax, araw);
            validm &= TS_BALLOT(sgl >= -r2.w);                       // sigma >= 0
        }
        if (validm == 0ull) continue;                                // wave-uniform
        const float am = TS_LANE(validm) ? a : 0.0f;
        const float ra = __builtin_amdgcn_rcpf(1.0f - am);
        const float Tk = T * ra;                        // transmittance in front of the Gaussian
        const float fac = am * Tk;
        float cv = r1.z * vo[0];                        // colour . v_out of this pixel
        cv = __builtin_fmaf(r1.w, vo[1], cv);
        cv = __builtin_fmaf(r2.x, vo[2], cv);
        if (CH == 4) cv = __builtin_fmaf(r2.y, vo[CH - 1], cv);
        // dL/dalpha = Tk (c . v_out) + ra (T_final (v_alpha - bg . v_out) - S_behind fac' c' . v_out)
        const float v_a = __builtin_fmaf(Tk, cv, ra * R);
        R = __builtin_fmaf(-fac, cv, R);
        T = Tk;
        // d alpha / d sigma = -alpha, or 0 where the 0.999 clamp is active
        float v_sig = -am * v_a;
        if (GENERAL) v_sig = TS_LANE(TS_BALLOT(araw > ts::kAlphaMax)) ? 0.0f : v_sig;
        vbuf[nrows * kRowStride + lane] = v_sig;
        fbuf[nrows * kRowStride + lane] = fac;
        if (lane == 0) meta[nrows] = lds[4 * j + 3];
        ++nrows;
        if (nrows == kRows) { ++j; break; }
    }
    return j;
}

template <int CH, bool SPLIT>
__global__ __launch_bounds__(kThreads, TS_BWD_MIN_WAVES) void raster_bwd_kernel(
    const ts_camera cam, const int num_tiles, const long long num_isects,
    const int* __restrict__ tile_bins, const int* __restrict__ ids_sorted,
    const float4* __restrict__ splats, const float* __restrict__ background,
    const float* __restrict__ final_Ts, const int* __restrict__ final_index,
    const float* __restrict__ v_out_img, const float* __restrict__ v_out_alpha,
    const unsigned char* __restrict__ clamp_mask,
    float* __restrict__ partials, unsigned char* __restrict__ row_flags) {
    __shared__ float4 lds_all[kWaves][64 * 4];
    __shared__ float4 rect_all[kWaves][4];
    __shared__ float vbuf_all[kWaves][kRows * kRowStride];
    __shared__ float fbuf_all[kWaves][kRows * kRowStride];
    __shared__ float4 vo_all[kWaves][4 * kVoStride];
    __shared__ float4 meta_all[kWaves][kRows];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int units = SPLIT ? 4 * num_tiles : num_tiles;
    const int unit = xcd_tile_group((units + kWaves - 1) / kWaves) * kWaves + wave;
    if (unit >= units) return;
    const int tile = SPLIT ? unit >> 2 : unit;
    const int only = SPLIT ? unit & 3 : -1;
    float4* rects = rect_all[wave];
    const int2 range = reinterpret_cast<const int2*>(tile_bins)[tile];
    if (range.y <= range.x) return;
    float4* lds = lds_all[wave];
    float* vbuf = vbuf_all[wave];
    float* fbuf = fbuf_all[wave];
    float4* vo_tab = vo_all[wave];
    float4* meta = meta_all[wave];
    const int tbx = cam.tile_bounds_x;
    const int tx = tile % tbx, ty = tile / tbx + cam.tile_row0;
    const int px0 = tx * 16 + (lane & 7), py0 = ty * 16 + (lane >> 3);
    const float X0 = (float)(tx * 16) + ts::kPixOff, Y0 = (float)(ty * 16) + ts::kPixOff;
    const int W = cam.img_width, H = cam.img_height;
    const int row_off = cam.tile_row0 * 16;
    const Geom geom = lane_geom(lane);

    float bg[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) bg[c] = background[c];

    // per pixel: T = transmittance behind the Gaussian being replayed; R = T_final*(v_alpha - bg.v_out)
    // - sum over the Gaussians already replayed of fac * (colour . v_out)   (see bwd_records)
    float T[4], R[4], vo[4][CH];
    int fidx[4], bmax[4];
    int fmax = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int px = px0 + 8 * (k & 1), py = py0 + 8 * (k >> 1);
        const bool inside = (px < W) && (py < H) && (!SPLIT || k == only);
        fidx[k] = -1;
        T[k] = 1.0f;
        R[k] = 0.0f;
#pragma unroll
        for (int c = 0; c < CH; ++c) vo[k][c] = 0.0f;
        if (inside) {
            const size_t pix = (size_t)(py - row_off) * W + px;
            fidx[k] = final_index[pix];
            T[k] = final_Ts[pix];
            float dotbg = 0.0f;
            const int pass = clamp_mask ? clamp_mask[pix] : 7;   // backward of the fused clamp(max=1)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                vo[k][c] = (c >= 3 || (pass & (1 << c))) ? v_out_img[pix * CH + c] : 0.0f;
                dotbg += bg[c] * vo[k][c];
            }
            const float va = v_out_alpha ? v_out_alpha[pix] : 0.0f;
            R[k] = T[k] * (va - dotbg);
        }
        // the tile's v_out, for pass B (which reads pixels of other lanes)
        vo_tab[k * kVoStride + lane] = make_float4(vo[k][0], vo[k][1], vo[k][2], CH == 4 ? vo[k][CH - 1] : 0.0f);
        bmax[k] = wave_max_int(fidx[k]);            // last list index any pixel of block k used
        fmax = max(fmax, bmax[k]);
    }
    const int last = min(range.y - 1, fmax);
    int nrows = 0;                                  // lines waiting for pass B (carried across blocks and chunks)

    // same software pipeline as the forward kernel, walking the list back to front
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 n0 = zero4, n1 = zero4, n2 = zero4;
    int id_next = 0;
    if (last - lane >= range.x) {
        const int g = ids_sorted[last - lane];
        n0 = splats[3 * (size_t)g]; n1 = splats[3 * (size_t)g + 1]; n2 = splats[3 * (size_t)g + 2];
    }
    if (last - 64 - lane >= range.x) id_next = ids_sorted[last - 64 - lane];

    for (int hi = last; hi >= range.x; hi -= 64) {
        const int i = hi - lane;
        const bool have = i >= range.x;
        const float4 q0 = n0, q1 = n1, q2 = n2;
        if (i - 64 >= range.x) {
            const int g = id_next;
            n0 = splats[3 * (size_t)g]; n1 = splats[3 * (size_t)g + 1]; n2 = splats[3 * (size_t)g + 2];
        }
        if (i - 128 >= range.x) id_next = ids_sorted[i - 128];
        int blocks;
        {   // rectangle of the pixels whose forward list reaches into this chunk (fidx >= chunk low)
            bool sel[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) sel[k] = fidx[k] >= hi - 63;
            blocks = update_rects(sel, X0, Y0, rects, lane);
            TS_WAVE_SYNC();
        }
        const Staged s = stage_gaussian(have, q0, q1);
        // one partial row per (tile, Gaussian, block): slot 4 * pair + k
        const int pair_slot = __float_as_int(q2.z) + ty * __float_as_int(q2.w) + tx;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!(blocks & (1 << k))) continue;                       // wave-uniform
            // nothing in block k got further than bmax[k] in forward
            const bool hit = (i <= bmax[k]) && block_hit(s, rects[k]);
            const mask64 mask = __ballot(hit);
            if (mask == 0ull) continue;
            const int cnt = __popcll(mask);
            if (hit) {
                const float bcx = X0 + (float)(8 * (k & 1)) + 3.5f, bcy = Y0 + (float)(8 * (k >> 1)) + 3.5f;
                const float dxc = s.gx - bcx, dyc = s.gy - bcy;
                const Coefs c = block_coefs(s, dxc, dyc);
                const int pos = __popcll(mask & ((1ull << lane) - 1ull));
                lds[4 * pos] = make_float4(c.c0, c.c1, c.c2, s.hA);
                lds[4 * pos + 1] = make_float4(s.B, s.hC, q1.z, q1.w);
                lds[4 * pos + 2] = make_float4(q2.x, q2.y, __int_as_float(i), s.lo);
                lds[4 * pos + 3] = make_float4(dxc, dyc, __int_as_float(4 * pair_slot + k), __int_as_float(k));
            }
            TS_WAVE_SYNC();
            const bool general = __ballot(hit && s.general) != 0ull;
            int j = 0;
            while (j < cnt) {
                j = general ? bwd_records<CH, true>(lds, j, cnt, geom, T[k], R[k], vo[k], fidx[k], vbuf, fbuf,
                                                    meta, nrows, lane)
                            : bwd_records<CH, false>(lds, j, cnt, geom, T[k], R[k], vo[k], fidx[k], vbuf, fbuf,
                                                     meta, nrows, lane);
                if (nrows == kRows) {
                    TS_WAVE_SYNC();
                    pass_b<CH>(vbuf, fbuf, vo_tab, meta, kRows, partials, row_flags, lane);
                    TS_WAVE_SYNC();
                    nrows = 0;
                }
            }
            TS_WAVE_SYNC();
        }
    }
    if (nrows > 0) {
        TS_WAVE_SYNC();
        pass_b<CH>(vbuf, fbuf, vo_tab, meta, nrows, partials, row_flags, lane);
    }
    (void)num_isects;
}

// Row layout (raw sums over the pixels of one block of one tile, see pass_b):
//   [ S v_sigma, S v_sigma dx, S v_sigma dy, S v_sigma dx^2, S v_sigma dx dy, S v_sigma dy^2, c0..c3, -, - ]
// with d = xy - pixel.  Every (tile, Gaussian) owns four rows (slot 4 s + block) and one 32-bit flag
// word; the conic / opacity factors are applied once per Gaussian here.
template <int CH>
__global__ __launch_bounds__(256) void reduce_partials_kernel(
    int n, int flags, const int* __restrict__ num_tiles_hit, const int* __restrict__ cum_tiles_hit,
    const float4* __restrict__ partials, const unsigned char* __restrict__ row_flags,
    const float4* __restrict__ splats, float* __restrict__ v_xy, float* __restrict__ v_conic,
    float* __restrict__ v_colors, float* __restrict__ v_opacity) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int cnt = num_tiles_hit[i];
    const long long end = cum_tiles_hit[i];
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
    auto add_row = [&](long long s) {
        const float4 p0 = partials[3 * s], p1 = partials[3 * s + 1], p2 = partials[3 * s + 2];
        a0.x += p0.x; a0.y += p0.y; a0.z += p0.z; a0.w += p0.w;
        a1.x += p1.x; a1.y += p1.y; a1.z += p1.z; a1.w += p1.w;
        a2.x += p2.x; a2.y += p2.y;
    };
    const unsigned int* flags4 = reinterpret_cast<const unsigned int*>(row_flags);
    for (long long s = end - cnt; s < end; ++s) {
        const unsigned int f = flags4[s];
        if (f == 0u) continue;                    // never written this pass (stale contents)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (f & (0xffu << (8 * k))) add_row(4 * s + k);
    }
    float vx = 0.f, vy = 0.f, vop = 0.f;
    if (cnt > 0) {
        const float4 q0 = splats[3 * (size_t)i], q1 = splats[3 * (size_t)i + 1];
        const float A = q0.w, B = q1.x, C = q1.y, op = q0.z;
        vx = A * a0.y + B * a0.z;
        vy = B * a0.y + C * a0.z;
        vop = op > 0.0f ? -a0.x / op : 0.0f;
        if (flags & TS_RASTER_LOGIT_OPACITY) vop *= op * (1.0f - op);   // through the sigmoid
    }
    reinterpret_cast<float2*>(v_xy)[i] = make_float2(vx, vy);
    v_opacity[i] = vop;
    v_conic[3 * i] = 0.5f * a0.w; v_conic[3 * i + 1] = a1.x; v_conic[3 * i + 2] = 0.5f * a1.y;
    if (CH == 4) {
        reinterpret_cast<float4*>(v_colors)[i] = make_float4(a1.z, a1.w, a2.x, a2.y);
    } else {
        v_colors[3 * i] = a1.z; v_colors[3 * i + 1] = a1.w; v_colors[3 * i + 2] = a2.x;
    }
}

inline int launch_status() { return (int)hipGetLastError(); }

}  // namespace

extern "C" {

int ts_raster_fwd(int32_t channels, int32_t flags, const ts_camera* cam, const int32_t* tile_bins,
                  const int32_t* gaussian_ids_sorted, const float* splats, const float* background,
                  float* out_img, float* final_Ts, int32_t* final_index, uint8_t* clamp_mask,
                  void* stream) {
    if (!cam || (channels != 3 && channels != 4)) return TS_E_BADARG;
    const int nt = cam->tile_rows * cam->tile_bounds_x;
    if (nt <= 0) return 0;
    if (!tile_bins || !background || !out_img || (!final_Ts != !final_index)) return TS_E_BADARG;
    const bool split = (flags & TS_RASTER_SPLIT_BLOCKS) != 0;
    const int units = split ? 4 * nt : nt;
    const int grid = 8 * (((units + kWaves - 1) / kWaves + 7) / 8);      // see xcd_tile_group
    hipStream_t s = (hipStream_t)stream;
    const float4* sp = reinterpret_cast<const float4*>(splats);
    const int clamp = (flags & TS_RASTER_CLAMP_RGB) ? 1 : 0;
#define TS_LAUNCH_FWD(C, S)                                                                        \
    hipLaunchKernelGGL((raster_fwd_kernel<C, S>), dim3(grid), dim3(kThreads), 0, s, *cam, nt,       \
                       tile_bins, gaussian_ids_sorted, sp, background, out_img, final_Ts,           \
                       final_index, clamp, clamp ? clamp_mask : nullptr)
    if (channels == 3) { if (split) TS_LAUNCH_FWD(3, true); else TS_LAUNCH_FWD(3, false); }
    else { if (split) TS_LAUNCH_FWD(4, true); else TS_LAUNCH_FWD(4, false); }
#undef TS_LAUNCH_FWD
    return launch_status();
}

int ts_raster_bwd(int32_t channels, int32_t flags, int64_t num_intersects, const ts_camera* cam,
                  const int32_t* tile_bins, const int32_t* gaussian_ids_sorted, const float* splats,
                  const float* background, const float* final_Ts, const int32_t* final_index,
                  const float* v_out_img, const float* v_out_alpha, const uint8_t* clamp_mask,
                  float* partials, uint8_t* row_flags, void* stream) {
    if (!cam || (channels != 3 && channels != 4) || num_intersects < 0) return TS_E_BADARG;
    const int nt = cam->tile_rows * cam->tile_bounds_x;
    if (nt <= 0 || num_intersects == 0) return 0;
    if (!tile_bins || !gaussian_ids_sorted || !splats || !background || !final_Ts || !final_index ||
        !v_out_img || !partials || !row_flags)
        return TS_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const bool split = (flags & TS_RASTER_SPLIT_BLOCKS) != 0;
    hipError_t e = hipMemsetAsync(row_flags, 0, (size_t)num_intersects * TS_ROWS_PER_PAIR, s);
    if (e != hipSuccess) return (int)e;
    const int units = split ? 4 * nt : nt;
    const int grid = 8 * (((units + kWaves - 1) / kWaves + 7) / 8);      // see xcd_tile_group
    const float4* sp = reinterpret_cast<const float4*>(splats);
#define TS_LAUNCH_BWD(C, S)                                                                        \
    hipLaunchKernelGGL((raster_bwd_kernel<C, S>), dim3(grid), dim3(kThreads), 0, s, *cam, nt,       \
                       (long long)num_intersects, tile_bins, gaussian_ids_sorted, sp, background,  \
                       final_Ts, final_index, v_out_img, v_out_alpha, clamp_mask, partials, row_flags)
    if (channels == 3) { if (split) TS_LAUNCH_BWD(3, true); else TS_LAUNCH_BWD(3, false); }
    else { if (split) TS_LAUNCH_BWD(4, true); else TS_LAUNCH_BWD(4, false); }
#undef TS_LAUNCH_BWD
    return launch_status();
}

int ts_reduce_partials(int32_t n, int32_t channels, int32_t flags, const int32_t* num_tiles_hit,
                       const int32_t* cum_tiles_hit, const float* partials,
                       const uint8_t* row_flags, const float* splats, float* v_xy, float* v_conic,
                       float* v_colors, float* v_opacity, void* stream) {
    if (n < 0 || (channels != 3 && channels != 4)) return TS_E_BADARG;
    if (n == 0) return 0;
    if (!num_tiles_hit || !cum_tiles_hit || !row_flags || !splats || !v_xy || !v_conic || !v_colors || !v_opacity)
        return TS_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const float4* pr = reinterpret_cast<const float4*>(partials);
    const float4* sp = reinterpret_cast<const float4*>(splats);
    const int grid = (n + 255) / 256;
    if (channels == 3)
        hipLaunchKernelGGL(reduce_partials_kernel<3>, dim3(grid), dim3(256), 0, s, n, (int)flags,
                           num_tiles_hit,
                           cum_tiles_hit, pr, row_flags, sp, v_xy, v_conic, v_colors, v_opacity);
    else
        hipLaunchKernelGGL(reduce_partials_kernel<4>, dim3(grid), dim3(256), 0, s, n, (int)flags,
                           num_tiles_hit,
                           cum_tiles_hit, pr, row_flags, sp, v_xy, v_conic, v_colors, v_opacity);
    return launch_status();
}

}  // extern "C"
