// train.hip - the two ops that sit directly around the render path in tinysplat's training step
// (/root/reference/scripts/train.py:58-63, :97):
//
//   * photometric loss  (1 - lambda) * L1 + lambda * (1 - SSIM)  and its gradient w.r.t. the rendered
//     image.  SSIM follows pytorch_msssim.SSIM(data_range=1, size_average=True, channel=3) as
//     constructed at tinysplat/splatting/model_gaussian.py:57: 11-tap Gaussian window (sigma 1.5),
//     separable "valid" filtering, K = (0.01, 0.03), mean over the (H-10) x (W-10) x 3 map.
//     Images are HWC float32 exactly as the rasterizer writes them (no permute / unsqueeze copy).
//   * Adam update of the six parameter tensors (torch.optim.Adam defaults, train.py:26) in ONE
//     launch over a table of tensors with per-tensor learning rates (model_gaussian.py:112-120).
//
// SSIM pass 1 (the five filtered maps, the SSIM sum, three partial-derivative maps) is a sliding window: one wave per
// 64 map columns and channel, the filtered rows in a register ring (ssim_fwd_rows_kernel; the tiled kernel of rounds
// 1 - 5 stays behind -DTS_SSIM_ROWS=0).  Pass 2 (the image gradient) stages a (32+10) x (32+10) patch of the maps in
// LDS and runs the two 11-tap passes from there, so each map byte is read from HBM ~1.7x.  Adam streams at HBM rate.
#include <hip/hip_runtime.h>

#include "../../include/tinysplat_hip.h"
#include "adam_math.h"

namespace {

constexpr int kWin = 11, kHalo = kWin - 1;     // valid filtering: output is (H-10) x (W-10)
constexpr int kTile = 32;                      // output tile edge
constexpr int kPatch = kTile + kHalo;          // 42
constexpr int kThreads = 256;
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

// pytorch_msssim._fspecial_gauss_1d(11, 1.5) as torch evaluates it in float32 (exp(-(i-5)^2 / 4.5), normalised by the
// float32 sum): the eleven values the reference's SSIM module holds, as literals
#define TS_GAUSS_WINDOW 0x1.0d957p-10f, 0x1.f1fe02p-8f, 0x1.26eb18p-5f, 0x1.bff0fep-4f, 0x1.b43c3ep-3f, 0x1.10656p-2f, \
                        0x1.b43c3ep-3f, 0x1.bff0fep-4f, 0x1.26eb18p-5f, 0x1.f1fe02p-8f, 0x1.0d957p-10f
__device__ __forceinline__ void gauss_window(float* g) {
    constexpr float w[kWin] = {TS_GAUSS_WINDOW};
#pragma unroll
    for (int i = 0; i < kWin; ++i) g[i] = w[i];
}

constexpr int kStrip = 4;   // outputs per thread and pass: inputs are read from LDS once per strip

// out[j] = sum_k g[k] * h[r0 + j + k][q], j < kStrip: a column strip with a register sliding window
__device__ __forceinline__ void vertical_strip(const float (*h)[kTile + 1], int r0, int q,
                                               const float* g, float* out) {
    float col[kWin + kStrip - 1];
#pragma unroll
    for (int k = 0; k < kWin + kStrip - 1; ++k) col[k] = h[r0 + k][q];
#pragma unroll
    for (int j = 0; j < kStrip; ++j) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < kWin; ++k) s += g[k] * col[j + k];
        out[j] = s;
    }
}

// h[r][q0 + j] = sum_k g[k] * p[r][q0 + j + k], j < kStrip
__device__ __forceinline__ void horizontal_strip(const float (*p)[kPatch + 1], int r, int q0,
                                                 const float* g, float (*h)[kTile + 1]) {
    float row[kWin + kStrip - 1];
#pragma unroll
    for (int k = 0; k < kWin + kStrip - 1; ++k) row[k] = p[r][q0 + k];
#pragma unroll
    for (int j = 0; j < kStrip; ++j) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < kWin; ++k) s += g[k] * row[j + k];
        h[r][q0 + j] = s;
    }
}

__device__ __forceinline__ float block_sum(float v, float* scratch) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    const float t = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    __syncthreads();
    return t;
}

// Pass 1: per 32x32 tile of the SSIM map: the five filtered maps, the SSIM value (summed per block
// into sums[block*2+0]) and the three partial-derivative maps dS/d filt(X), dS/d filt(X^2),
// dS/d filt(XY) written to dmaps[3][3 channels][Ho][Wo] (planar).  Also the L1 sum of the tile's own 32x32 pixels
// (image tiles of the same grid cover the whole image; sums[block*2+1]).
// X has `xs` floats per pixel (3: an RGB image; 4: the compositing kernels' RGB+depth output, whose
// channel 3 is compared with the depth target D when D != nullptr, train.py:65-69).
// Xd (with xs == 3): the depth as its own [H,W] plane (ts_photometric_loss_planes).
__global__ __launch_bounds__(kThreads) void ssim_fwd_kernel(int H, int W, int xs,
                                                            const float* __restrict__ X,
                                                            const float* __restrict__ Xd,
                                                            const float* __restrict__ Y,
                                                            const float* __restrict__ D,
                                                            float* __restrict__ dmaps,
                                                            float* __restrict__ sums) {
    __shared__ float px[kPatch][kPatch + 1], py[kPatch][kPatch + 1];
    __shared__ float h0[kPatch][kTile + 1], h1[kPatch][kTile + 1], h2[kPatch][kTile + 1],
        h3[kPatch][kTile + 1], h4[kPatch][kTile + 1];
    __shared__ float scratch[4];
    float g[kWin];
    gauss_window(g);
    const int Ho = H - kHalo, Wo = W - kHalo;
    const int ox0 = blockIdx.x * kTile, oy0 = blockIdx.y * kTile;
    float ssim_sum = 0.0f, l1_sum = 0.0f, depth_sum = 0.0f;
    // every thread's share of the 42 x 42 pixel patch is fetched ONCE for the three channels (one
    // 16-byte load per pixel of a 4-float image) instead of one strided 4-byte load per channel pass
    constexpr int kShare = (kPatch * kPatch + kThreads - 1) / kThreads;
    float xa[kShare][3], ya[kShare][3];
#pragma unroll
    for (int u = 0; u < kShare; ++u) {
        const int i = threadIdx.x + u * kThreads;
        const int r = i / kPatch, q = i % kPatch;
        const int y = oy0 + r, x = ox0 + q;
#pragma unroll
        for (int c = 0; c < 3; ++c) xa[u][c] = ya[u][c] = 0.0f;
        if (i < kPatch * kPatch && y < H && x < W) {
            const size_t pix = (size_t)y * W + x;
            float x3 = 0.0f;
            if (xs == 4) {
                const float4 v = reinterpret_cast<const float4*>(X)[pix];
                xa[u][0] = v.x; xa[u][1] = v.y; xa[u][2] = v.z; x3 = v.w;
            } else {
                xa[u][0] = X[pix * 3]; xa[u][1] = X[pix * 3 + 1]; xa[u][2] = X[pix * 3 + 2];
                if (Xd && D && r < kTile && q < kTile) x3 = Xd[pix];
            }
            ya[u][0] = Y[pix * 3]; ya[u][1] = Y[pix * 3 + 1]; ya[u][2] = Y[pix * 3 + 2];
            if (r < kTile && q < kTile) {
                l1_sum += fabsf(xa[u][0] - ya[u][0]);
                l1_sum += fabsf(xa[u][1] - ya[u][1]);
                l1_sum += fabsf(xa[u][2] - ya[u][2]);
                if (D) depth_sum += fabsf(x3 - D[pix]);
            }
        }
    }
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int u = 0; u < kShare; ++u) {
            const int i = threadIdx.x + u * kThreads;
            if (i < kPatch * kPatch) {
                const int r = i / kPatch, q = i % kPatch;
                px[r][q] = c == 0 ? xa[u][0] : (c == 1 ? xa[u][1] : xa[u][2]);
                py[r][q] = c == 0 ? ya[u][0] : (c == 1 ? ya[u][1] : ya[u][2]);
            }
        }
        __syncthreads();
        // horizontal pass: 42 rows x 32 columns, five quantities.  One item = kStrip adjacent
        // outputs of a row: its kWin + kStrip - 1 inputs are read from LDS once and reused from
        // registers (14 reads for 4 outputs instead of 44).
        for (int i = threadIdx.x; i < kPatch * (kTile / kStrip); i += kThreads) {
            const int r = i / (kTile / kStrip), q0 = (i % (kTile / kStrip)) * kStrip;
            float a[kWin + kStrip - 1], b[kWin + kStrip - 1];
#pragma unroll
            for (int k = 0; k < kWin + kStrip - 1; ++k) { a[k] = px[r][q0 + k]; b[k] = py[r][q0 + k]; }
#pragma unroll
            for (int j = 0; j < kStrip; ++j) {
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
                for (int k = 0; k < kWin; ++k) {
                    const float av = a[j + k], bv = b[j + k], w = g[k];
                    s0 += w * av; s1 += w * bv; s2 += w * av * av; s3 += w * bv * bv; s4 += w * av * bv;
                }
                h0[r][q0 + j] = s0; h1[r][q0 + j] = s1; h2[r][q0 + j] = s2; h3[r][q0 + j] = s3;
                h4[r][q0 + j] = s4;
            }
        }
        __syncthreads();
        // vertical pass + SSIM
        for (int i = threadIdx.x; i < kTile * (kTile / kStrip); i += kThreads) {
            // one item = kStrip vertically adjacent outputs of a column (register sliding window)
            const int q = i % kTile, r0 = (i / kTile) * kStrip;
            float f[5][kStrip];
            vertical_strip(h0, r0, q, g, f[0]); vertical_strip(h1, r0, q, g, f[1]);
            vertical_strip(h2, r0, q, g, f[2]); vertical_strip(h3, r0, q, g, f[3]);
            vertical_strip(h4, r0, q, g, f[4]);
#pragma unroll
          for (int j = 0; j < kStrip; ++j) {
            const int r = r0 + j;
            const int y = oy0 + r, x = ox0 + q;
            if (y >= Ho || x >= Wo) continue;
            const float m1 = f[0][j], m2 = f[1][j], q1 = f[2][j], q2 = f[3][j], r12 = f[4][j];
            const float s1 = q1 - m1 * m1, s2 = q2 - m2 * m2, s12 = r12 - m1 * m2;
            const float A1 = 2.0f * m1 * m2 + kC1, A2 = 2.0f * s12 + kC2;
            const float B1 = m1 * m1 + m2 * m2 + kC1, B2 = s1 + s2 + kC2;
            const float inv = 1.0f / (B1 * B2);
            const float S = A1 * A2 * inv;
            ssim_sum += S;
            const size_t o = ((size_t)c * Ho + y) * Wo + x;      // planar per channel: coalesced in x
            const size_t plane = (size_t)Ho * Wo * 3;
            dmaps[o] = 2.0f * m2 * (A2 - A1) * inv - 2.0f * m1 * S * (1.0f / B1 - 1.0f / B2);  // d/d filt(X)
            dmaps[plane + o] = -S / B2;                                                        // d/d filt(X^2)
            dmaps[2 * plane + o] = 2.0f * A1 * inv;                                            // d/d filt(XY)
          }
        }
        __syncthreads();
    }
    const float ts = block_sum(ssim_sum, scratch);
    const float tl = block_sum(l1_sum, scratch);
    const float td = block_sum(depth_sum, scratch);
    if (threadIdx.x == 0) {
        const int b = blockIdx.y * gridDim.x + blockIdx.x;
        sums[3 * b] = ts;
        sums[3 * b + 1] = tl;
        sums[3 * b + 2] = td;
    }
}

// Pass 1 as a SLIDING WINDOW (round 6; the tiled kernel above stays for -DTS_SSIM_ROWS=0).  One wave owns 64 columns
// of the SSIM map for one colour channel and walks `seg` map rows from the top: a pixel row is read once (64 + 10
// columns, the next rows already in flight), filtered horizontally through a wave-private LDS line - no workgroup
// barrier anywhere - and the eleven most recent horizontally filtered rows of the five maps stay in REGISTERS (a ring,
// the row loop unrolled by the window length so that every index is static), so the vertical pass costs no memory
// traffic at all.  Image bytes read 1.16 x (columns) x (seg + 10) / seg (rows) instead of 1.72 x, a third of the
// LDS traffic and none of the nine barriers of the tiled kernel, which ran at 1.8 TB/s of its own traffic (95 us at
// 1920 x 1080: three workgroups per CU, 2.7 rounds).  The three channel waves of a column block share a workgroup (and
// so the L1 lines of the interleaved pixels); its fourth wave sums the depth term.  sums: {ssim, l1, depth l1} per wave at
// sums[3 * ((seg * column_blocks + block) * 4 + wave)]; every pixel's L1 term is counted by exactly one wave.
#ifndef TS_SSIM_ROWS
#define TS_SSIM_ROWS 1
#endif
constexpr int kCols = 64;              // map columns per wave
// Map rows per wave, chosen per image so that the launch is ONE round of resident workgroups (three per CU at 140 - 168
// VGPRs): 1080p on 256 CUs: 30 column blocks x 25 segments of 43 rows = 750 <= 768.  A small image gets short segments
// (more waves, more halo rows each) instead of a handful of waves walking hundreds of rows one after the other.
constexpr int kSegMin = 8, kSegMax = 64;
inline int ssim_segment_rows(int ho, int wo) {
    static int resident = 0;                                   // workgroups resident at once (per process: one device kind)
    if (resident == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
            (void)hipGetLastError();
            cus = 256;
        }
        resident = 3 * cus;
    }
    const int blocks_x = (wo + kCols - 1) / kCols;
    const int segments = resident / blocks_x > 0 ? resident / blocks_x : 1;
    const int seg = (ho + segments - 1) / segments;
    return seg < kSegMin ? kSegMin : (seg > kSegMax ? kSegMax : seg);
}
constexpr int kLine = kCols + 16;      // LDS line: 74 pixels used
constexpr int kRowsWaves = 4;          // waves per workgroup: three channels and the depth term

#define TS_TRAIN_WAVE_SYNC()                                      \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

struct RowRegs { float xm, xe, ym, ye; };              // one pixel row of a wave: main / extra columns of X and Y

__global__ __launch_bounds__(64 * kRowsWaves) void ssim_fwd_rows_kernel(int H, int W, int xs, int seg,
                                                                         const float* __restrict__ X,
                                                                         const float* __restrict__ Xd,
                                                                         const float* __restrict__ Y,
                                                                         const float* __restrict__ D,
                                                                         float* __restrict__ dmaps,
                                                                         float* __restrict__ sums) {
    __shared__ float line[3][2][2][kLine];                     // [channel wave][parity][X | Y][column]
    constexpr float g[kWin] = {TS_GAUSS_WINDOW};
    const int lane = threadIdx.x & 63, c = threadIdx.x >> 6;
    const int Ho = H - kHalo, Wo = W - kHalo;
    const int x0 = blockIdx.x * kCols, o0 = blockIdx.y * seg;
    const int rows = min(seg, Ho - o0);                        // map rows of this wave (>= 1)
    const bool last_block = blockIdx.x == gridDim.x - 1, last_seg = blockIdx.y == gridDim.y - 1;
    const int xa = x0 + lane, xb = x0 + kCols + lane;
    const bool in_a = xa < W, in_b = lane < kHalo && xb < W;
    const size_t slot = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kRowsWaves + c;
    if (c == 3) {
        // the fourth wave: the depth L1 of the pixels this workgroup owns (its own rows and main columns; the last
        // segment / column block also takes the ten trailing rows / columns nobody else starts at)
        float depth_sum = 0.0f;
        if (D) {
            // branch-free loads, eleven rows in flight: one row per round trip made this wave the slowest of the
            // workgroup (a 44-row chain of dependent memory latencies)
            const int own_rows = last_seg ? rows + kHalo : rows;
            const int xa_d = min(xa, W - 1), xb_d = min(xb, W - 1);
            const float* const da = xs == 4 ? X + (size_t)xa_d * 4 + 3 : Xd + xa_d;
            const float* const db = xs == 4 ? X + (size_t)xb_d * 4 + 3 : Xd + xb_d;
            const size_t stride = xs == 4 ? (size_t)W * 4 : (size_t)W;
            const bool use_b = last_block && in_b;
            for (int t0 = 0; t0 < own_rows; t0 += kWin) {
                float va[kWin], vb[kWin], ta[kWin], tb[kWin];
#pragma unroll
                for (int i = 0; i < kWin; ++i) {
                    const size_t y = (size_t)(o0 + min(t0 + i, own_rows - 1));
                    va[i] = da[y * stride]; vb[i] = db[y * stride];
                    ta[i] = D[y * W + xa_d]; tb[i] = D[y * W + xb_d];
                }
#pragma unroll
                for (int i = 0; i < kWin; ++i) {
                    if (t0 + i < own_rows) {
                        depth_sum += in_a ? fabsf(va[i] - ta[i]) : 0.0f;
                        depth_sum += use_b ? fabsf(vb[i] - tb[i]) : 0.0f;
                    }
                }
            }
        }
        depth_sum = wave_sum(depth_sum);
        if (lane == 0) { sums[3 * slot] = 0.0f; sums[3 * slot + 1] = 0.0f; sums[3 * slot + 2] = depth_sum; }
        return;
    }
    float ssim_sum = 0.0f, l1_sum = 0.0f;
    // branch-free loads (columns beyond the image read the last pixel of the row and are zeroed): a load inside a
    // divergent branch makes the compiler wait for ALL outstanding memory operations before the next use, which
    // serialises the rows in flight
    const int xa_c = min(xa, W - 1), xb_c = min(xb, W - 1);
    // the lane's part of every address once; a row adds a wave-uniform offset
    const float* const xa_p = X + (size_t)xa_c * xs + c;
    const float* const xb_p = X + (size_t)xb_c * xs + c;
    const float* const ya_p = Y + (size_t)xa_c * 3 + c;
    const float* const yb_p = Y + (size_t)xb_c * 3 + c;
    float* const dmap_lane = dmaps + (size_t)c * Ho * Wo + min(xa, Wo - 1);
    auto fetch = [&](int y) -> RowRegs {
        const size_t rx = (size_t)y * W * xs, ry = (size_t)y * W * 3;
        RowRegs r;
        r.xm = xa_p[rx]; r.ym = ya_p[ry];
        r.xe = xb_p[rx]; r.ye = yb_p[ry];
        return r;                                              // (zeroed where they are consumed: a select here would wait for the load)
    };

    const int total = rows + kHalo;                            // pixel rows o0 .. o0 + total - 1 (all < H)
    // Rows are fetched kBatch at a time into registers with STATIC indices (the row loop is unrolled by kRing = 12, one
    // slot more than the window needs; 12 = 2 kBatch: two batches, one being filtered and one in flight).  On gfx9-class
    // targets loads and stores share one counter and may complete out of order, so with the map stores of earlier rows
    // in flight the compiler waits for EVERYTHING before the first use of a loaded register - a `cur = next` copy or a
    // three-row ring end in s_waitcnt vmcnt(0) per row (111 us).  With one wait per SIX rows the other waves of the SIMD
    // cover it: one batch without look-ahead 61 us, the next batch issued right after the first row of the current one
    // has been consumed (its wait then finds loads a whole batch old) 59 - 62 us - the launch is ~0.6 VALU-bound by then.
    constexpr int kBatch = 6, kRing = kWin + 1;
    static_assert(kRing == 2 * kBatch, "two batches per unrolled body");
    RowRegs rb[2 * kBatch];
#pragma unroll
    for (int i = 0; i < kBatch; ++i) rb[i] = fetch(o0 + min(i, total - 1));
    float ring[5][kRing];
    const size_t plane = (size_t)Ho * Wo * 3;
    for (int base = 0; base < total; base += kRing) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) {
            const int t = base + s;
            if (t >= total) continue;                          // wave-uniform (no break: the unrolled body keeps static ring indices)
            const int y = o0 + t;
            const RowRegs cur = rb[s];
            // this wave's share of the L1 sums: its own rows and main columns; the last segment / column block also
            // takes the ten trailing rows / columns nobody else starts at
            const float xm = in_a ? cur.xm : 0.0f, ym = in_a ? cur.ym : 0.0f;
            const float xe = in_b ? cur.xe : 0.0f, ye = in_b ? cur.ye : 0.0f;
            if (t < rows || last_seg) l1_sum += fabsf(xm - ym) + (last_block ? fabsf(xe - ye) : 0.0f);
            float* lx = line[c][t & 1][0];
            float* ly = line[c][t & 1][1];
            lx[lane] = xm; ly[lane] = ym;
            if (lane < 16) { lx[kCols + lane] = xe; ly[kCols + lane] = ye; }
            if (s % kBatch == 0 && t + kBatch < total) {       // the batch after this one, into the other half of rb
#pragma unroll
                for (int i = 0; i < kBatch; ++i) rb[(s + kBatch + i) % kRing] = fetch(o0 + min(t + kBatch + i, total - 1));
            }
            TS_TRAIN_WAVE_SYNC();
            float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, h4 = 0.f;
#pragma unroll
            for (int k = 0; k < kWin; ++k) {
                const float a = lx[lane + k], b = ly[lane + k];
                const float wa = g[k] * a, wb = g[k] * b;
                h0 += wa; h1 += wb;
                h2 = __builtin_fmaf(wa, a, h2); h3 = __builtin_fmaf(wb, b, h3); h4 = __builtin_fmaf(wa, b, h4);
            }
            ring[0][s] = h0; ring[1][s] = h1; ring[2][s] = h2; ring[3][s] = h3; ring[4][s] = h4;
            if (t >= kHalo) {
                // map row o = y - 10: the rows y - 10 .. y sit in ring slots (s + 2 + k) % 12
                float f[5];
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    float v = 0.f;
#pragma unroll
                    for (int k = 0; k < kWin; ++k) v = __builtin_fmaf(g[k], ring[m][(s + 2 + k) % kRing], v);
                    f[m] = v;
                }
                if (xa < Wo) {
                    const float m1 = f[0], m2 = f[1], q1 = f[2], q2 = f[3], r12 = f[4];
                    const float s1 = q1 - m1 * m1, s2 = q2 - m2 * m2, s12 = r12 - m1 * m2;
                    const float A1 = 2.0f * m1 * m2 + kC1, A2 = 2.0f * s12 + kC2;
                    const float B1 = m1 * m1 + m2 * m2 + kC1, B2 = s1 + s2 + kC2;
                    // two hardware reciprocals (1 ulp) instead of four IEEE divisions (~10 instructions each): the maps
                    // move by ~1e-7 relative, the loss by < 1e-7
                    const float i1 = __builtin_amdgcn_rcpf(B1), i2 = __builtin_amdgcn_rcpf(B2);
                    const float inv = i1 * i2;
                    const float S = A1 * A2 * inv;
                    ssim_sum += S;
                    float* o = dmap_lane + (size_t)(y - kHalo) * Wo;              // planar per channel: coalesced in x
                    o[0] = 2.0f * m2 * (A2 - A1) * inv - 2.0f * m1 * S * (i1 - i2);   // d/d filt(X)
                    o[plane] = -S * i2;                                               // d/d filt(X^2)
                    o[2 * plane] = 2.0f * A1 * inv;                                   // d/d filt(XY)
                }
            }
        }
    }
    const float ts = wave_sum(ssim_sum), tl = wave_sum(l1_sum);
    if (lane == 0) { sums[3 * slot] = ts; sums[3 * slot + 1] = tl; sums[3 * slot + 2] = 0.0f; }
}

// Pass 2: gradient w.r.t. X.  The transpose of the valid filter is a full correlation of the three
// partial maps (zero outside the (Ho, Wo) map):  gX = F^T dm + 2 X F^T dq + Y F^T dr, scaled by
// w_ssim, plus w_l1 * sign(X - Y).
__global__ __launch_bounds__(kThreads) void ssim_bwd_kernel(int H, int W, int xs,
                                                            const float* __restrict__ X,
                                                            const float* __restrict__ Xd,
                                                            const float* __restrict__ Y,
                                                            const float* __restrict__ D,
                                                            const float* __restrict__ dmaps,
                                                            float w_l1, float w_ssim, float w_depth,
                                                            float* __restrict__ gX, float* __restrict__ gXd) {
    __shared__ float p0[kPatch][kPatch + 1], p1[kPatch][kPatch + 1], p2[kPatch][kPatch + 1];
    __shared__ float h0[kPatch][kTile + 1], h1[kPatch][kTile + 1], h2[kPatch][kTile + 1];
    float g[kWin];
    gauss_window(g);
    const int Ho = H - kHalo, Wo = W - kHalo;
    const size_t plane = (size_t)Ho * Wo * 3;
    const int x0 = blockIdx.x * kTile, y0 = blockIdx.y * kTile;     // image tile
    // each thread owns kStrip vertically adjacent pixels of the 32 x 32 tile: their gradients are
    // collected over the three channel passes and stored once (one 16-byte store per pixel of a
    // 4-float image) instead of three strided 4-byte stores
    static_assert(kTile * (kTile / kStrip) == kThreads, "one strip per thread");
    float gacc[kStrip][3];
    for (int c = 0; c < 3; ++c) {
        // map patch needed: rows y0-10 .. y0+31, cols x0-10 .. x0+31
        for (int i = threadIdx.x; i < kPatch * kPatch; i += kThreads) {
            const int r = i / kPatch, q = i % kPatch;
            const int y = y0 - kHalo + r, x = x0 - kHalo + q;
            float a = 0.f, b = 0.f, d = 0.f;
            if (y >= 0 && y < Ho && x >= 0 && x < Wo) {
                const size_t o = ((size_t)c * Ho + y) * Wo + x;
                a = dmaps[o]; b = dmaps[plane + o]; d = dmaps[2 * plane + o];
            }
            p0[r][q] = a; p1[r][q] = b; p2[r][q] = d;
        }
        __syncthreads();
        // pixel (y, x) receives from map locations (y - k, x - l): flipped window == same (symmetric)
        for (int i = threadIdx.x; i < kPatch * (kTile / kStrip); i += kThreads) {
            const int r = i / (kTile / kStrip), q0 = (i % (kTile / kStrip)) * kStrip;
            horizontal_strip(p0, r, q0, g, h0); horizontal_strip(p1, r, q0, g, h1);
            horizontal_strip(p2, r, q0, g, h2);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < kTile * (kTile / kStrip); i += kThreads) {
            const int q = i % kTile, r0 = (i / kTile) * kStrip;
            float fa[kStrip], fb[kStrip], fd[kStrip];
            vertical_strip(h0, r0, q, g, fa); vertical_strip(h1, r0, q, g, fb);
            vertical_strip(h2, r0, q, g, fd);
#pragma unroll
          for (int j = 0; j < kStrip; ++j) {
            const int y = y0 + r0 + j, x = x0 + q;
            if (y >= H || x >= W) continue;
            const float a = fa[j], b = fb[j], d = fd[j];
            const size_t pix = (size_t)y * W + x;
            const float xv = X[pix * xs + c], yv = Y[pix * 3 + c];
            const float diff = xv - yv;
            const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
            const float gv = w_l1 * sgn + w_ssim * (a + 2.0f * xv * b + yv * d);
            if (c == 0) gacc[j][0] = gv; else if (c == 1) gacc[j][1] = gv; else gacc[j][2] = gv;
          }
        }
        __syncthreads();
    }
    {
        const int q = threadIdx.x % kTile, r0 = (threadIdx.x / kTile) * kStrip;
#pragma unroll
        for (int j = 0; j < kStrip; ++j) {
            const int y = y0 + r0 + j, x = x0 + q;
            if (y >= H || x >= W) continue;
            const size_t pix = (size_t)y * W + x;
            if (xs == 4) {
                float gd = 0.0f;                     // gradient of the depth channel (zero without a target)
                if (D) {
                    const float dd = X[pix * 4 + 3] - D[pix];
                    gd = w_depth * (dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f));
                }
                reinterpret_cast<float4*>(gX)[pix] = make_float4(gacc[j][0], gacc[j][1], gacc[j][2], gd);
            } else {
                gX[pix * 3] = gacc[j][0]; gX[pix * 3 + 1] = gacc[j][1]; gX[pix * 3 + 2] = gacc[j][2];
                if (gXd) {                           // the depth plane's gradient (zero without a target)
                    float gd = 0.0f;
                    if (D) {
                        const float dd = Xd[pix] - D[pix];
                        gd = w_depth * (dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f));
                    }
                    gXd[pix] = gd;
                }
            }
        }
    }
}

struct AdamTable {
    float* p[TS_ADAM_MAX_TENSORS];
    const float* g[TS_ADAM_MAX_TENSORS];
    float* m[TS_ADAM_MAX_TENSORS];
    float* v[TS_ADAM_MAX_TENSORS];
    long long n[TS_ADAM_MAX_TENSORS];
    float step_size[TS_ADAM_MAX_TENSORS];     // lr / (1 - beta1^step)
    float bc2_sqrt[TS_ADAM_MAX_TENSORS];      // sqrt(1 - beta2^step)
    int count;
};

// grid.y = tensor index; 16-byte accesses when the element count allows (all six tinysplat tensors
// have element counts divisible by 4 for N % 4 == 0; the tail is handled element-wise)
__global__ __launch_bounds__(kThreads) void adam_kernel(const AdamTable t, float beta1, float beta2,
                                                        float eps) {
    const int ti = blockIdx.y;
    const long long n = t.n[ti];
    float* __restrict__ p = t.p[ti];
    const float* __restrict__ g = t.g[ti];
    float* __restrict__ m = t.m[ti];
    float* __restrict__ v = t.v[ti];
    const ts::AdamCoef coef = {beta1, beta2, eps, t.step_size[ti], t.bc2_sqrt[ti]};     // (adam_math.h: one update, every caller)
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* P = &pp.x; const float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) ts::adam_update(P[k], G[k], M[k], V[k], coef);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0) {
        for (long long i = 4 * n4 + threadIdx.x; i < n; i += kThreads) {
            float pi = p[i], mi = m[i], vi = v[i];
            ts::adam_update(pi, g[i], mi, vi, coef);
            p[i] = pi; m[i] = mi; v[i] = vi;
        }
    }
}

// The loss value from the partial sums of pass 1 (ABI 8): {loss, mean|X - Y|, SSIM, mean|depth - D|} as four floats,
// summed in double in a fixed order by one workgroup.  Replaces a dozen one-element PyTorch launches per training step
// (reduce, three divisions, the weighted sum, four casts) that sat between the loss kernels and the frame's backward.
__global__ __launch_bounds__(kThreads) void loss_reduce_kernel(int H, int W, int triples, float w_l1, float w_ssim,
                                                               float w_depth, const float* __restrict__ sums,
                                                               float* __restrict__ out) {
    __shared__ double part[3][kThreads];
    double a = 0.0, b = 0.0, d = 0.0;
    for (int i = threadIdx.x; i < triples; i += kThreads) {
        a += (double)sums[3 * i]; b += (double)sums[3 * i + 1]; d += (double)sums[3 * i + 2];
    }
    part[0][threadIdx.x] = a; part[1][threadIdx.x] = b; part[2][threadIdx.x] = d;
    __syncthreads();
    for (int step = kThreads / 2; step >= 1; step >>= 1) {
        if ((int)threadIdx.x < step)
            for (int k = 0; k < 3; ++k) part[k][threadIdx.x] += part[k][threadIdx.x + step];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n_map = 3.0 * (double)(H - kHalo) * (double)(W - kHalo), n_img = 3.0 * (double)H * (double)W;
        // the caller's gradient scales carry the weights: w_l1 = (1 - lambda) / (3 H W), w_ssim = -lambda / (3 Ho Wo),
        // w_depth = lambda_depth / (H W)
        const double lambda = -(double)w_ssim * n_map;
        const double loss = (double)w_l1 * part[1][0] + lambda + (double)w_ssim * part[0][0] + (double)w_depth * part[2][0];
        out[0] = (float)loss;
        out[1] = (float)(part[1][0] / n_img);
        out[2] = (float)(part[0][0] / n_map);
        out[3] = (float)(part[2][0] / ((double)H * (double)W));
    }
}

inline int launch_status() { return (int)hipGetLastError(); }

}  // namespace

extern "C" {

int64_t ts_photometric_ws_floats(int32_t height, int32_t width) {
    if (height <= kHalo || width <= kHalo) return 0;
    const int64_t ho = height - kHalo, wo = width - kHalo;
    const int64_t seg = TS_SSIM_ROWS ? ssim_segment_rows((int)ho, (int)wo) : 1;
    const int64_t tiles = TS_SSIM_ROWS ? kRowsWaves * ((wo + kCols - 1) / kCols) * ((ho + seg - 1) / seg)      // one triple per wave
                                       : (int64_t)((width + kTile - 1) / kTile) * ((height + kTile - 1) / kTile);
    return 3 * ho * wo * 3 + 3 * tiles;
}

namespace {
// pass 1 of the loss: per-wave (per-tile) sums behind the three partial-derivative maps in ws
void launch_ssim_fwd(int height, int width, int xs, const float* image, const float* depth, const float* target,
                     const float* depth_target, float* ws, hipStream_t s) {
    const size_t plane3 = (size_t)3 * (height - kHalo) * (width - kHalo) * 3;
    if (TS_SSIM_ROWS) {
        const int seg = ssim_segment_rows(height - kHalo, width - kHalo);
        const dim3 grid((width - kHalo + kCols - 1) / kCols, (height - kHalo + seg - 1) / seg);
        hipLaunchKernelGGL(ssim_fwd_rows_kernel, grid, dim3(64 * kRowsWaves), 0, s, height, width, xs, seg, image, depth,
                           target, depth_target, ws, ws + plane3);
    } else {
        const dim3 grid((width + kTile - 1) / kTile, (height + kTile - 1) / kTile);
        hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(kThreads), 0, s, height, width, xs, image, depth, target,
                           depth_target, ws, ws + plane3);
    }
}
}  // namespace

int ts_photometric_loss_rgbd(int32_t height, int32_t width, int32_t pixel_floats, const float* image,
                             const float* target, const float* depth_target, float w_l1,
                             float w_ssim, float w_depth, float* ws, float* v_image, void* stream) {
    if (height <= kHalo || width <= kHalo) return TS_E_BADARG;
    if (pixel_floats != 3 && pixel_floats != 4) return TS_E_BADARG;
    if (depth_target && pixel_floats != 4) return TS_E_BADARG;
    if (!image || !target || !ws) return TS_E_BADARG;
    const dim3 grid((width + kTile - 1) / kTile, (height + kTile - 1) / kTile);
    const size_t plane3 = (size_t)3 * (height - kHalo) * (width - kHalo) * 3;
    hipStream_t s = (hipStream_t)stream;
    launch_ssim_fwd(height, width, (int)pixel_floats, image, nullptr, target, depth_target, ws, s);
    if (v_image)
        hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(kThreads), 0, s, height, width,
                           (int)pixel_floats, image, (const float*)nullptr, target, depth_target, ws, w_l1, w_ssim,
                           w_depth, v_image, (float*)nullptr);
    return launch_status();
}

int ts_photometric_loss_planes(int32_t height, int32_t width, const float* image, const float* depth,
                               const float* target, const float* depth_target, float w_l1, float w_ssim,
                               float w_depth, float* ws, float* v_image, float* v_depth, void* stream) {
    if (height <= kHalo || width <= kHalo) return TS_E_BADARG;
    if (!image || !target || !ws || (depth_target && !depth) || (v_depth && (!depth || !v_image))) return TS_E_BADARG;
    const dim3 grid((width + kTile - 1) / kTile, (height + kTile - 1) / kTile);
    const size_t plane3 = (size_t)3 * (height - kHalo) * (width - kHalo) * 3;
    hipStream_t s = (hipStream_t)stream;
    launch_ssim_fwd(height, width, 3, image, depth, target, depth_target, ws, s);
    if (v_image)
        hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(kThreads), 0, s, height, width, 3, image, depth, target,
                           depth_target, ws, w_l1, w_ssim, w_depth, v_image, v_depth);
    return launch_status();
}

int ts_photometric_loss_reduce(int32_t height, int32_t width, float w_l1, float w_ssim, float w_depth,
                               const float* ws, float* out4, void* stream) {
    if (height <= kHalo || width <= kHalo || !ws || !out4) return TS_E_BADARG;
    const int64_t maps = (int64_t)9 * (height - kHalo) * (width - kHalo);
    const int64_t triples = (ts_photometric_ws_floats(height, width) - maps) / 3;
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, height, width, (int)triples,
                       w_l1, w_ssim, w_depth, ws + maps, out4);
    return launch_status();
}

int ts_photometric_loss(int32_t height, int32_t width, const float* image, const float* target,
                        float w_l1, float w_ssim, float* ws, float* v_image, void* stream) {
    return ts_photometric_loss_rgbd(height, width, 3, image, target, nullptr, w_l1, w_ssim, 0.0f, ws,
                                    v_image, stream);
}

int ts_adam_step(int32_t num_tensors, float* const* params, const float* const* grads,
                 float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                 const float* lr, const int32_t* steps, float beta1, float beta2, float eps,
                 void* stream) {
    if (num_tensors < 0 || num_tensors > TS_ADAM_MAX_TENSORS) return TS_E_BADARG;
    if (num_tensors == 0) return 0;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr || !steps) return TS_E_BADARG;
    AdamTable t;
    long long maxn = 0;
    t.count = num_tensors;
    for (int i = 0; i < TS_ADAM_MAX_TENSORS; ++i) {
        const bool on = i < num_tensors;
        t.p[i] = on ? params[i] : nullptr; t.g[i] = on ? grads[i] : nullptr;
        t.m[i] = on ? exp_avg[i] : nullptr; t.v[i] = on ? exp_avg_sq[i] : nullptr;
        t.n[i] = on ? numel[i] : 0;
        t.step_size[i] = 0.0f; t.bc2_sqrt[i] = 1.0f;
        if (!on) continue;
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] < 0 || steps[i] < 1)
            return TS_E_BADARG;
        // double-precision bias corrections as torch.optim.Adam computes them on the host; the step
        // count is per tensor because torch skips (and does not age) tensors without a gradient
        const ts::AdamCoef c = ts::adam_coef(lr[i], steps[i], beta1, beta2, eps);
        t.step_size[i] = c.step_size;
        t.bc2_sqrt[i] = c.bc2_sqrt;
        if (t.n[i] > maxn) maxn = t.n[i];
    }
    if (maxn == 0) return 0;
    long long blocks = (maxn / 4 + kThreads - 1) / kThreads;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks, num_tensors), dim3(kThreads), 0,
                       (hipStream_t)stream, t, beta1, beta2, eps);
    return launch_status();
}

}  // extern "C"
