// binning.hip - tile binning and per-tile depth sort for gfx950.
//
// What gsplat's rasterize_gaussians does before compositing (the part reached from
// /root/reference/tinysplat/splatting/rasterize.py:44,50): cumsum(num_tiles_hit), emit one
// (tile<<32 | depth bits, gaussian id) pair per covered tile, global 64-bit sort, bin edges.
// Here the same final order (tile-major, then depth bits, ties by ascending Gaussian id) is produced
// without a global sort:
//   scan_tiles      wave-scan (DPP-free __shfl_up ladder inside a wave, LDS across the 4 waves)
//   bin_count       per-chunk tile histograms in LDS (ds_add), B x T count matrix, no global atomics
//   tile_offsets    column scan of the matrix (per-chunk bases, one launch) + exclusive scan over tiles -> tile_bins
//   bin_scatter     replays each chunk with LDS cursors preloaded from its bases, 4-byte id stores
//                   (the sort gathers depth by id from the 4 N-byte, L2-resident depth array;
//                   sort key = depth_bits << 32 | gaussian_id, unique inside a tile)
//   sort_tiles      one workgroup per tile.  <= 1024 keys: bitonic network with the keys in registers
//                   (in-thread stages), 64-bit lane exchanges (in-wave stages) and LDS only for the
//                   few cross-wave stages.  Larger tiles: per-tile sample sort (splitters from a
//                   sorted sample, LDS histogram + cursors, then every wave sorts sub-buckets of
//                   <= 256 keys on its own) - O(n log n), no size limit
//   pack_splats     gathers the compositing operands of a Gaussian into one 48-byte record
// Traffic: 4 I written + 4 I read (+ gathers) + 4 I written (+ 8 B T for the count matrix) against the 36 I a 3-pass 64-bit
// LSD radix sort of key+payload would move at minimum.
#include <hip/hip_runtime.h>

#include "../../include/tinysplat_hip.h"
#include "pack.h"
#include "splat_math.h"

namespace {

constexpr int kThreads = 256;
constexpr int kScanItems = 4;                       // per thread
constexpr int kScanBlock = kThreads * kScanItems;   // 1024 elements per block

__device__ __forceinline__ int wave_inclusive_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(v, d, 64);
        if (lane >= d) v += u;
    }
    return v;
}

// block-wide inclusive scan of one int per thread (WAVES waves; 256 threads = 4 by default); returns the
// inclusive value, *total = block sum.
template <int WAVES = 4>
__device__ __forceinline__ int block_inclusive_scan(int v, int* total) {
    __shared__ int wave_sums[WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = wave_inclusive_scan(v);
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    int add = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const int s = wave_sums[w];
        if (w < wave) add += s;
        tot += s;
    }
    *total = tot;
    __syncthreads();
    return inc + add;
}

// total_out (may be null): device-visible address - pinned host memory - that receives out[n-1], the
// grand total, from the thread that produces it (system-scope store): the host reads it there without
// a copy operation behind the scan
__global__ __launch_bounds__(kThreads) void scan_local_kernel(int n, const int* __restrict__ in,
                                                              int* __restrict__ out,
                                                              int* __restrict__ block_sums,
                                                              int* __restrict__ total_out) {
    const int base = blockIdx.x * kScanBlock + threadIdx.x * kScanItems;
    int v[kScanItems];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        sum += v[k];
        v[k] = sum;
    }
    int total;
    const int inc = block_inclusive_scan(sum, &total);
    const int excl = inc - sum;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) {
            out[base + k] = v[k] + excl;
            if (total_out && gridDim.x == 1 && base + k == n - 1)
                __hip_atomic_store(total_out, v[k] + excl, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// adds to every element of block b the sum of the blocks before it, which the block reduces itself from
// the <= n / 1024 block sums (a separate single-workgroup scan of those sums would be one more launch on
// the critical path of the frame: ~5 us each on MI355X however small the kernel)
__global__ __launch_bounds__(kThreads) void scan_add_kernel(int n, int* __restrict__ out,
                                                            const int* __restrict__ block_sums,
                                                            int* __restrict__ total_out) {
    __shared__ int red[kThreads / 64];
    int part = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += kThreads) part += block_sums[j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    int add = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) add += red[w];
    const int base = blockIdx.x * kScanBlock + threadIdx.x * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) {
            const int v = out[base + k] + add;
            if (blockIdx.x != 0) out[base + k] = v;
            if (total_out && base + k == n - 1)
                __hip_atomic_store(total_out, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
}

// ---- tile bucketing without global atomics --------------------------------------------------
// Device-scope atomics on MI355X are executed memory-side (the per-XCD L2s are not coherent), and
// measured ~25 G atomics/s: 6 M intersections cost 0.25 ms per pass.  Instead the Gaussians are cut
// into B contiguous chunks; workgroup b builds the tile histogram of its chunk in LDS (ds_add, no
// memory traffic), the B x T count matrix is column-scanned into per-(chunk,tile) bases, and the
// scatter pass replays the chunk with LDS cursors preloaded from those bases.  The order inside a
// bucket is arbitrary (LDS atomic order); sort_tiles makes it canonical.
// Block (b, w) replays chunk b and touches the tiles of window w.  Measured on config 3 / config 5:
// one window per chunk (as long as the tile count fits the LDS) with 4096-Gaussian chunks is the
// fastest split - larger chunks with several tile windows make the runs written into a bucket
// longer but re-read every Gaussian once per window and lost 10-30 %.
#ifndef TS_BIN_THREADS
#define TS_BIN_THREADS 1024
#endif
constexpr int kBinThreads = TS_BIN_THREADS;
#ifndef TS_BIN_CHUNK
#define TS_BIN_CHUNK 4096
#endif
constexpr int kBinChunkMin = TS_BIN_CHUNK;  // Gaussians per chunk (at least)
constexpr int kBinMaxChunks = 512;
constexpr int kBinWindowMax = 36864;      // tiles per LDS window (144 KiB of the 160 KiB LDS)
constexpr int kBinTargetBlocks = 1;       // chunks x windows aimed at (1 = no extra windows)

// Chunk size: kBinChunkMin Gaussians for scenes of ~1 M and more (one workgroup per CU); smaller scenes get
// smaller chunks, down to 1024, so that a 100 k-Gaussian frame still spreads over ~100 workgroups
// (config 2: bin_count 30 -> 15 us, bin_scatter 33 -> 17 us).
__host__ __device__ inline int bin_num_chunks(int n) {
    int per = n / 192;
    per = per < 1024 ? 1024 : (per > kBinChunkMin ? kBinChunkMin : per);
    int b = (n + per - 1) / per;
    if (b < 1) b = 1;
    if (b > kBinMaxChunks) b = kBinMaxChunks;
    return b;
}

// tiles per window: enough windows to reach kBinTargetBlocks workgroups, never above the LDS budget
inline int bin_window_tiles(int n, int num_tiles) {
    const int chunks = bin_num_chunks(n);
    int nwin = (kBinTargetBlocks + chunks - 1) / chunks;
    const int max_win = (num_tiles + 255) / 256;           // keep windows >= 256 tiles
    if (nwin > max_win) nwin = max_win;
    if (nwin < 1) nwin = 1;
    int window = (num_tiles + nwin - 1) / nwin;
    window = (window + 63) & ~63;
    if (window > kBinWindowMax) window = kBinWindowMax;
    return window;
}

// Tight binning (splats != nullptr): a (Gaussian, tile) pair of the bounding box is dropped when the
// packed record proves that no pixel of the 16x16 tile can reach alpha >= 1/255.  Dropped pairs
// contribute exactly nothing to the image or to any gradient, so the frame is bit-identical; on the
// random scenes ~36 % of the bounding-box pairs go away before the scatter, the sort and the
// compositing kernels ever see them.  Count and scatter make the same decisions (same code, same
// inputs).  With splats == nullptr the lists are gsplat's bounding-box lists.
//
// The level set {alpha >= 1/255} is the ellipse  hA dx^2 + B dx dy + hC dy^2 <= tau  (log2 domain,
// d = pixel - centre).  Instead of testing every tile of the box, each tile ROW gets the x-interval
// of the ellipse over the row's y-band in closed form: for a fixed dy the ellipse is the interval
// (-B dy -+ sqrt(disc(dy))) / (2 hA) with disc = 4 hA tau - D4 dy^2, D4 = 4 hA hC - B^2; its left
// end is convex and its right end concave in dy, so over a band the union is spanned by the band's
// two ends and, when they lie in the band, the ellipse's leftmost / rightmost points.  tau carries
// the same slack as ts::rect_may_contribute (evaluated for the farthest pixel of the box) and the
// interval is widened by kTightEps pixels, which makes the kept set a superset of every pixel whose
// alpha test can pass in the compositing kernels.
constexpr int kTilePix = 16;              // tile edge in pixels (rasterize.py:19-20)
constexpr float kTightEps = 0.02f;
struct TightTest {
    bool cull_all, geometric;
    float gx, gy, hA, B, tau4A, D4, inv2A, dymax, dxext, dy_left;
    __device__ __forceinline__ TightTest(bool tight, const float4 q0, const float4 q1, float radius) {
        cull_all = false; geometric = false;
        gx = gy = hA = B = tau4A = D4 = inv2A = dymax = dxext = dy_left = 0.0f;
        if (!tight) return;
        gx = q0.x - ts::kPixOff; gy = q0.y - ts::kPixOff;      // (row_range works on pixel INDICES: sample = index + off)
        hA = 0.5f * ts::kLog2e * q0.w; B = ts::kLog2e * q1.x;
        const float hC = 0.5f * ts::kLog2e * q1.y;
        const float op = q0.z;
        float tau = __log2f(op) + ts::kLog2_255;
        if (!(op > 0.0f) || !(tau >= -0.02f)) { cull_all = true; return; }
        D4 = 4.0f * hA * hC - B * B;
        // D4 = 4 hA hC (1 - rho^2) cancels for a rotated needle: its float32 relative error is ~2e-7 / (1 - rho^2),
        // and the ellipse's extent sqrt(tau / D4) inherits half of it.  Below 1 - rho^2 = 1e-2 (axis ratio > 20 at
        // 45 degrees) that error is no longer small against the slack of the test: keep the bounding box.
        if (!(hA > 0.0f && hC > 0.0f && D4 > 1e-2f * (4.0f * hA * hC))) return;
        const float far = radius + (float)kTilePix;                            // farthest pixel offset
        tau += 0.02f + 4.0e-6f * (hA + hC + fabsf(B)) * far * far;
        geometric = true;
        inv2A = 0.5f / hA;
        tau4A = 4.0f * hA * tau;
        dymax = sqrtf(tau4A / D4) + kTightEps;
        dxext = sqrtf(4.0f * hC * tau / D4);
        dy_left = B * dxext / (2.0f * hC);           // dy of the leftmost point (rightmost: -dy_left)
    }
    // tiles [lo, hi) of tile row ty (clipped to [minx, maxx)) the ellipse can reach
    __device__ __forceinline__ void row_range(int ty, int minx, int maxx, int& lo, int& hi) const {
        lo = minx; hi = maxx;
        if (!geometric) return;
        const float a = fmaxf((float)(ty * kTilePix) - gy - kTightEps, -dymax);
        const float b = fminf((float)(ty * kTilePix + kTilePix - 1) - gy + kTightEps, dymax);
        if (a > b) { hi = lo; return; }
        const float sa = sqrtf(fmaxf(tau4A - D4 * a * a, 0.0f)), sb = sqrtf(fmaxf(tau4A - D4 * b * b, 0.0f));
        float left = fminf((-B * a - sa) * inv2A, (-B * b - sb) * inv2A);
        float right = fmaxf((-B * a + sa) * inv2A, (-B * b + sb) * inv2A);
        if (dy_left >= a && dy_left <= b) left = -dxext;
        if (-dy_left >= a && -dy_left <= b) right = dxext;
        left = fminf(left, right);                                   // rounding near a tangent band
        const float xl = gx + left - kTightEps, xr = gx + right + kTightEps;
        // tile tx holds sample positions 16 tx .. 16 tx + 15
        const int tlo = (int)ceilf((xl - (float)(kTilePix - 1)) * (1.0f / kTilePix));
        const int thi = (int)floorf(xr * (1.0f / kTilePix)) + 1;
        lo = max(lo, tlo); hi = min(hi, thi);
        if (hi < lo) hi = lo;
    }
};

// Walks the Gaussians of a chunk and calls emit(t) for every list (tile of the lists' shape, index t
// inside the launch) a Gaussian is entered in.  Count and scatter replay the same walk, so they agree.
// A thread handles kBinThreads-strided Gaussians; the operands of kBinPre of them (radius, centre, the
// two record words of the tight test) are loaded before the first one is processed, and a workgroup is
// 1024 threads: with one workgroup per CU the walk is a chain of dependent loads and square roots per
// Gaussian, and more waves / more loads in flight are what hide it.  Measured against 512 threads without
// prefetch: bin_count 32.5 -> 25.7 us on config 3, 125 -> 111 us on config 5; bin_scatter 0.63 -> 0.49 ms
// on config 5 (63 us either way on config 3, where the scattered 4-byte stores bind).
#ifndef TS_BIN_PRE
#define TS_BIN_PRE 2
#endif
constexpr int kBinPre = TS_BIN_PRE;
template <typename Emit>
__device__ __forceinline__ void walk_chunk(int g0, int g1, const float* __restrict__ xys,
                                           const int* __restrict__ radii,
                                           const float4* __restrict__ splats, const ts_camera& cam,
                                           Emit emit) {
    const int ws = cam.wide_tiles ? 1 : 0;                         // list column = 16x16 column >> ws
    const int tbx = (cam.tile_bounds_x + ws) >> ws;
    const bool tight_lists = splats != nullptr;
    for (int base = g0 + threadIdx.x; base < g1; base += kBinPre * kBinThreads) {
        int r[kBinPre];
        float2 xy[kBinPre];
        float4 q0[kBinPre], q1[kBinPre];
#pragma unroll
        for (int u = 0; u < kBinPre; ++u) {
            const int i = base + u * kBinThreads;
            r[u] = i < g1 ? radii[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < kBinPre; ++u) {
            const int i = base + u * kBinThreads;
            xy[u] = make_float2(0.f, 0.f);
            q0[u] = q1[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r[u] > 0) {
                xy[u] = reinterpret_cast<const float2*>(xys)[i];
                if (tight_lists) { q0[u] = splats[3 * (size_t)i]; q1[u] = splats[3 * (size_t)i + 1]; }
            }
        }
#pragma unroll
        for (int u = 0; u < kBinPre; ++u) {
            if (r[u] <= 0) continue;
            const ts::TileBox b = ts::tile_bbox(xy[u].x, xy[u].y, (float)r[u], cam.tile_bounds_x,
                                                cam.tile_bounds_y, cam.tile_row0, cam.tile_rows);
            if (b.maxy <= b.miny || b.maxx <= b.minx) continue;  // not in this stripe (its record was never written)
            const TightTest tight(tight_lists, q0[u], q1[u], (float)r[u]);
            if (tight.cull_all) continue;
            const int i = base + u * kBinThreads;
            for (int ty = b.miny; ty < b.maxy; ++ty) {
                int lo, hi;
                tight.row_range(ty, b.minx, b.maxx, lo, hi);
                if (hi <= lo) continue;
                // list columns: 16x16 tiles lo..hi-1, or the wide tiles that contain them
                for (int tx = lo >> ws; tx <= (hi - 1) >> ws; ++tx) emit((ty - cam.tile_row0) * tbx + tx, i);
            }
        }
    }
}

// LOAD-BALANCED WALK (round 4).  walk_chunk gives every lane one Gaussian and lets it loop over
// its tile rows and the tiles of a row: a Gaussian of the random scenes has 1-3 rows, the slowest lane of a wave ~5,
// and the 35-instruction row_range runs at ~40 % lane use (bin_count: 670 wave instructions per 64 Gaussians).  Here
// a batch of 1024 Gaussians (one per thread) is expanded into ITEMS = (Gaussian, tile row): phase 1 computes the box
// and the TightTest once per Gaussian and stashes what a row needs in LDS; a workgroup scan of the row counts gives
// every Gaussian its item range; the items are written out in windows of kItemCap and phase 2 hands consecutive
// items to consecutive lanes.  Same decisions as walk_chunk (same TightTest members, same row_range), different
// ORDER of the emits - which no caller depends on (LDS histogram / cursors; the per-tile sort follows).
#ifndef TS_BIN_BALANCED
#define TS_BIN_BALANCED 1
#endif
constexpr int kItemCap = 4096;                    // items per window
constexpr int kStashWords = 11;                   // per Gaussian: 10 TightTest floats + (minx | maxx << 16)
constexpr size_t kBalancedLds = (size_t)(kStashWords * kBinThreads + kItemCap + 32) * 4;      // static LDS it needs
template <typename Emit>
__device__ __forceinline__ void walk_chunk_balanced(int g0, int g1, const float* __restrict__ xys,
                                                    const int* __restrict__ radii,
                                                    const float4* __restrict__ splats, const ts_camera& cam,
                                                    Emit emit) {
    static_assert(kBinThreads == 1024, "one Gaussian per thread and batch, items packed as lt | row << 10");
    __shared__ float stash[kStashWords][kBinThreads];
    __shared__ int items[kItemCap];
    __shared__ int wave_sum[kBinThreads / 64 + 1];
    const int ws = cam.wide_tiles ? 1 : 0;
    const int tbx = (cam.tile_bounds_x + ws) >> ws;
    const bool tight_lists = splats != nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int base = g0; base < g1; base += kBinThreads) {
        // ---- phase 1: one Gaussian per thread
        const int i = base + tid;
        int rows = 0, miny = 0;
        if (i < g1) {
            const int r = radii[i];
            if (r > 0) {
                const float2 xy = reinterpret_cast<const float2*>(xys)[i];
                float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
                if (tight_lists) { q0 = splats[3 * (size_t)i]; q1 = splats[3 * (size_t)i + 1]; }
                const ts::TileBox b = ts::tile_bbox(xy.x, xy.y, (float)r, cam.tile_bounds_x, cam.tile_bounds_y,
                                                    cam.tile_row0, cam.tile_rows);
                if (b.maxy > b.miny && b.maxx > b.minx) {
                    const TightTest t(tight_lists, q0, q1, (float)r);
                    if (!t.cull_all) {
                        rows = b.maxy - b.miny;
                        miny = b.miny;
                        stash[0][tid] = t.geometric ? t.dymax : -1.0f;          // dymax >= 0 when geometric
                        stash[1][tid] = t.gx; stash[2][tid] = t.gy; stash[3][tid] = t.B; stash[4][tid] = t.D4;
                        stash[5][tid] = t.inv2A; stash[6][tid] = t.tau4A; stash[7][tid] = t.dxext;
                        stash[8][tid] = t.dy_left; stash[9][tid] = __int_as_float(b.miny);
                        stash[10][tid] = __int_as_float(b.minx | (b.maxx << 16));
                    }
                }
            }
        }
        // ---- exclusive scan of the row counts over the workgroup
        int incl = rows;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d, 64);
            if (lane >= d) incl += v;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        if (wave == 0) {
            int v = lane < kBinThreads / 64 ? wave_sum[lane] : 0;
#pragma unroll
            for (int d = 1; d < kBinThreads / 64; d <<= 1) {
                const int u = __shfl_up(v, d, 64);
                if (lane >= d) v += u;
            }
            if (lane < kBinThreads / 64) wave_sum[lane] = v;                     // inclusive over waves
        }
        __syncthreads();
        const int total = wave_sum[kBinThreads / 64 - 1];
        const int first = incl - rows + (wave > 0 ? wave_sum[wave - 1] : 0);   // this Gaussian's first item
        // ---- items in windows of kItemCap: written by their Gaussians, consumed by consecutive lanes
        for (int w0 = 0; w0 < total; w0 += kItemCap) {
            const int lo_k = max(0, w0 - first), hi_k = min(rows, w0 + kItemCap - first);
            for (int k = lo_k; k < hi_k; ++k) items[first + k - w0] = tid | (k << 10);
            __syncthreads();
            const int m = min(kItemCap, total - w0);
            for (int j = tid; j < m; j += kBinThreads) {
                const int it = items[j], lt = it & 1023, k = it >> 10;
                TightTest t(false, make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), 0.0f);
                const float d0 = stash[0][lt];
                t.geometric = d0 >= 0.0f;
                t.dymax = d0;
                t.gx = stash[1][lt]; t.gy = stash[2][lt]; t.B = stash[3][lt]; t.D4 = stash[4][lt];
                t.inv2A = stash[5][lt]; t.tau4A = stash[6][lt]; t.dxext = stash[7][lt]; t.dy_left = stash[8][lt];
                const int ty = __float_as_int(stash[9][lt]) + k;
                const int xx = __float_as_int(stash[10][lt]);
                int lo, hi;
                t.row_range(ty, xx & 0xffff, xx >> 16, lo, hi);
                if (hi <= lo) continue;
                const int id = base + lt;
                for (int tx = lo >> ws; tx <= (hi - 1) >> ws; ++tx) emit((ty - cam.tile_row0) * tbx + tx, id);
            }
            __syncthreads();
        }
        __syncthreads();          // wave_sum and the stash change hands (a batch without items has no barrier above)
    }
}

// grid = (chunks, windows); counts[b * T + t]
template <bool BAL>
__global__ __launch_bounds__(kBinThreads) void bin_count_kernel(
    int n, int chunk, const float* __restrict__ xys, const int* __restrict__ radii,
    const float4* __restrict__ splats, const ts_camera cam, int num_tiles, int window,
    int* __restrict__ counts) {
    extern __shared__ int hist[];
    const int t0 = blockIdx.y * window;
    const int tw = min(num_tiles - t0, window);
    for (int j = threadIdx.x; j < tw; j += kBinThreads) hist[j] = 0;
    __syncthreads();
    const int g0 = blockIdx.x * chunk, g1 = min(n, g0 + chunk);
    auto emit = [&](int t, int) {
        t -= t0;
        if ((unsigned)t < (unsigned)tw) atomicAdd(&hist[t], 1);
    };
    if (BAL) walk_chunk_balanced(g0, g1, xys, radii, splats, cam, emit);
    else walk_chunk(g0, g1, xys, radii, splats, cam, emit);
    __syncthreads();
    int* dst = counts + (size_t)blockIdx.x * num_tiles + t0;
    for (int j = threadIdx.x; j < tw; j += kBinThreads) dst[j] = hist[j];
}

constexpr int kScanGroups = 16;      // chunk groups of the column scan (waves of its workgroup)

// Column scan of the B x T count matrix in ONE launch: a workgroup owns 64 tile columns, its 16 waves own the
// 16 chunk groups; thread (tile, group) loads its <= kColPer counts (coalesced across the tiles of a wave, all
// loads issued before the first use), scans them in registers, the group sums of a tile are exchanged through LDS,
// and the thread writes the exclusive prefixes back.  tile_total[t] = column sum.  Replaces three launches
// (group sums, scan of the group sums, finish: ~6 us each on an in-order stream whatever their size).
constexpr int kColTiles = 64;
constexpr int kColPer = (kBinMaxChunks + kScanGroups - 1) / kScanGroups;      // 32 chunks per group at most
__global__ __launch_bounds__(kColTiles * kScanGroups) void column_scan_kernel(int num_tiles, int chunks,
                                                                              int per_group,
                                                                              int* __restrict__ counts,
                                                                              int* __restrict__ tile_total) {
    __shared__ int gsum[kScanGroups][kColTiles];
    const int tl = threadIdx.x & (kColTiles - 1), g = threadIdx.x / kColTiles;
    const int t = blockIdx.x * kColTiles + tl;
    const int b0 = g * per_group, b1 = min(chunks, b0 + per_group);
    int c[kColPer];
    int sum = 0;
    if (t < num_tiles) {
#pragma unroll
        for (int j = 0; j < kColPer; ++j) c[j] = (b0 + j < b1) ? counts[(size_t)(b0 + j) * num_tiles + t] : 0;
#pragma unroll
        for (int j = 0; j < kColPer; ++j) {
            const int v = c[j];
            c[j] = sum;
            sum += v;
        }
    }
    gsum[g][tl] = sum;
    __syncthreads();
    if (t >= num_tiles) return;
    int base = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < kScanGroups; ++q) {
        const int v = gsum[q][tl];
        if (q < g) base += v;
        tot += v;
    }
#pragma unroll
    for (int j = 0; j < kColPer; ++j)
        if (b0 + j < b1) counts[(size_t)(b0 + j) * num_tiles + t] = base + c[j];
    if (g == 0) tile_total[t] = tot;
}

// single workgroup of 1024 threads, 8 tiles per lane and step (a 1080p frame's 8160 tiles in one step):
// exclusive scan over tiles -> tile_bins; tile_total[t] becomes start[t]
constexpr int kOffsetsThreads = 1024;
// CAPACITY GUARD: when the caller sized the per-intersection buffers from an estimate (capacity >= 0) instead of
// waiting for the count, a frame whose bounding-box total cum[n-1] exceeds it must not touch them: the guard word
// (spare[-1]) is set, every list is left empty (tile_bins = 0, tile_start = 0), ts_bin_scatter returns at once, and
// the compositing kernels find nothing to do; the host sees the same total when it reads the count and runs the
// frame again with exact sizes.
__global__ __launch_bounds__(kOffsetsThreads) void tile_offsets_kernel(int num_tiles,
                                                                       int* __restrict__ tile_total,
                                                                       int* __restrict__ tile_bins,
                                                                       int* __restrict__ spare,
                                                                       const int* __restrict__ total_ptr,
                                                                       long long capacity, int* __restrict__ longest_out) {
    constexpr int kPer = 8;
    __shared__ int carry, over, longest;
    int my_longest = 0;
    if (threadIdx.x == 0) {
        carry = 0;
        longest = 0;
        *spare = 0;                                       // the workspace's last word: ts_sort_tiles' tile counter
        over = (total_ptr && capacity >= 0 && ((long long)*total_ptr > capacity || *total_ptr < 0)) ? 1 : 0;
        spare[-1] = over;
    }
    __syncthreads();
    if (over) {
        for (int t = threadIdx.x; t <= num_tiles; t += kOffsetsThreads) {
            tile_total[t] = 0;
            if (t < num_tiles) reinterpret_cast<int2*>(tile_bins)[t] = make_int2(0, 0);
        }
        return;
    }
    for (int base = 0; base < num_tiles; base += kOffsetsThreads * kPer) {
        const int t0 = base + threadIdx.x * kPer;
        int v[kPer], sum = 0;
#pragma unroll
        for (int e = 0; e < kPer; ++e) {
            v[e] = (t0 + e < num_tiles) ? tile_total[t0 + e] : 0;
            sum += v[e];
            my_longest = max(my_longest, v[e]);
        }
        int total;
        const int inc = block_inclusive_scan<kOffsetsThreads / 64>(sum, &total);
        int start = carry + inc - sum;
#pragma unroll
        for (int e = 0; e < kPer; ++e) {
            if (t0 + e < num_tiles) {
                tile_total[t0 + e] = start;
                reinterpret_cast<int2*>(tile_bins)[t0 + e] =
                    v[e] > 0 ? make_int2(start, start + v[e]) : make_int2(0, 0);
            }
            start += v[e];
        }
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_total[num_tiles] = carry;     // tile_start[T]: end of the last bucket
    if (longest_out) {          // the longest list of the frame (a scene statistic for the caller's launch policy; may be mapped host memory)
        atomicMax(&longest, my_longest);                     // (LDS)
        __syncthreads();
        if (threadIdx.x == 0) *longest_out = longest;
    }
}

__global__ __launch_bounds__(kBinThreads) void bin_scatter_kernel(
    int n, int chunk, const float* __restrict__ xys,
    const int* __restrict__ radii, const float4* __restrict__ splats, const ts_camera cam,
    int num_tiles, int window, const int* __restrict__ bases, const int* __restrict__ tile_start,
    int* __restrict__ bucket_ids) {
    extern __shared__ int cursor[];
    if (tile_start[num_tiles + 1] != 0) return;              // capacity guard (tile_offsets_kernel)
    const int t0 = blockIdx.y * window;
    const int tw = min(num_tiles - t0, window);
    const int* src = bases + (size_t)blockIdx.x * num_tiles + t0;
    for (int j = threadIdx.x; j < tw; j += kBinThreads) cursor[j] = tile_start[t0 + j] + src[j];
    __syncthreads();
    const int g0 = blockIdx.x * chunk, g1 = min(n, g0 + chunk);
    walk_chunk(g0, g1, xys, radii, splats, cam, [&](int t, int i) {
        t -= t0;
        if ((unsigned)t < (unsigned)tw) bucket_ids[atomicAdd(&cursor[t], 1)] = i;
    });
}

// Two-level scatter.  The direct scatter above writes every 4-byte id to a different cache line (a chunk of
// 4096 Gaussians puts ~2 entries into each of 8160 buckets) from whichever XCD runs the chunk: the L2s hold
// partial lines of the same bucket and write them back separately - WRITE_SIZE measured 8x the id bytes
// (131 MB for 16 MB on config 3).  With a scratch buffer of I words the ids instead travel in two coalesced
// hops:
//   coarse  each chunk appends (id | tile-in-group << 27) to the region of the tile GROUP (kCoarseTiles = 32
//           consecutive lists) - a group's region is the concatenation of its tiles' buckets, so a chunk's
//           entries for it form one run of ~64 words;
//   fine    one workgroup per group streams its region and places the ids in the tiles' buckets with LDS
//           cursors: all of a bucket's lines are written by one workgroup, merge in one L2 and leave as full lines.
#ifndef TS_COARSE_SHIFT
#define TS_COARSE_SHIFT 5
#endif
constexpr int kCoarseShift = TS_COARSE_SHIFT;
constexpr int kCoarseTiles = 1 << kCoarseShift;
constexpr int kCoarseIdBits = 32 - kCoarseShift;            // ids below 2^27
#ifndef TS_TWO_HOP_FROM
#define TS_TWO_HOP_FROM (1 << 18)
#endif
constexpr int kTwoHopFrom = TS_TWO_HOP_FROM;

template <bool BAL>
__global__ __launch_bounds__(kBinThreads) void bin_scatter_coarse_kernel(
    int n, int chunk, const float* __restrict__ xys, const int* __restrict__ radii,
    const float4* __restrict__ splats, const ts_camera cam, int num_tiles, const int* __restrict__ bases,
    const int* __restrict__ tile_start, int* __restrict__ scratch) {
    extern __shared__ int cursor[];
    if (tile_start[num_tiles + 1] != 0) return;              // capacity guard (tile_offsets_kernel)
    const int groups = (num_tiles + kCoarseTiles - 1) >> kCoarseShift;
    // cursor[g] = start of the group's region + the entries earlier chunks put into the group's tiles; the chunk's
    // row of the base matrix is read coalesced by the whole workgroup and summed per group with LDS atomics
    const int* src = bases + (size_t)blockIdx.x * num_tiles;
    for (int g = threadIdx.x; g < groups; g += kBinThreads) cursor[g] = tile_start[g << kCoarseShift];
    __syncthreads();
    for (int t = threadIdx.x; t < num_tiles; t += kBinThreads) {
        const int v = src[t];
        if (v) atomicAdd(&cursor[t >> kCoarseShift], v);
    }
    __syncthreads();
    const int g0 = blockIdx.x * chunk, g1 = min(n, g0 + chunk);
    auto emit = [&](int t, int i) {
        scratch[atomicAdd(&cursor[t >> kCoarseShift], 1)] = i | ((t & (kCoarseTiles - 1)) << kCoarseIdBits);
    };
    if (BAL) walk_chunk_balanced(g0, g1, xys, radii, splats, cam, emit);
    else walk_chunk(g0, g1, xys, radii, splats, cam, emit);
}

#ifndef TS_FINE_THREADS
#define TS_FINE_THREADS 1024
#endif
#ifndef TS_FINE_AHEAD
#define TS_FINE_AHEAD 8
#endif
#ifndef TS_FINE_REORDER
#define TS_FINE_REORDER 1
#endif
constexpr int kFineThreads = TS_FINE_THREADS;
constexpr int kFineAhead = TS_FINE_AHEAD;
constexpr int kFinePass = kFineThreads * kFineAhead;     // entries handled together
// Fine hop: one workgroup per tile group streams the group's region in passes of kFinePass entries.  A pass is
// reordered by tile in LDS before it is written (rank inside the tile from a returning LDS atomic, tile offsets
// from a 32-entry scan), so that consecutive lanes store consecutive words of one bucket: with one 4-byte store
// per entry straight from the stream a wave's store touched ~32 different lines, and the kernel was bound by
// those write transactions (137 us for 250 MB on config 5), not by bytes.  The order inside a bucket is
// arbitrary, as before; ts_sort_tiles follows.
__global__ __launch_bounds__(kFineThreads) void bin_scatter_fine_kernel(int num_tiles,
                                                                        const int* __restrict__ tile_start,
                                                                        const int* __restrict__ scratch,
                                                                        int* __restrict__ bucket_ids) {
    __shared__ int cursor[kCoarseTiles];                 // next free word of every bucket of the group
    const int t0 = blockIdx.x << kCoarseShift, t1 = min(num_tiles, t0 + kCoarseTiles);
    if ((int)threadIdx.x < t1 - t0) cursor[threadIdx.x] = tile_start[t0 + threadIdx.x];
    const int begin = tile_start[t0], end = tile_start[t1];
#if TS_FINE_REORDER
    __shared__ int hist[kCoarseTiles], loff[kCoarseTiles + 1], gbase[kCoarseTiles];
    __shared__ int ids[kFinePass];
    __shared__ unsigned char tiles[kFinePass];
    if ((int)threadIdx.x < kCoarseTiles) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int c0 = begin; c0 < end; c0 += kFinePass) {
        const int m = min(kFinePass, end - c0);                       // entries of this pass
        unsigned int w[kFineAhead];
        int rank[kFineAhead];
#pragma unroll
        for (int u = 0; u < kFineAhead; ++u) {
            const int k = u * kFineThreads + (int)threadIdx.x;
            if (k < m) w[u] = (unsigned int)scratch[c0 + k];
        }
#pragma unroll
        for (int u = 0; u < kFineAhead; ++u) {
            const int k = u * kFineThreads + (int)threadIdx.x;
            if (k < m) rank[u] = atomicAdd(&hist[w[u] >> kCoarseIdBits], 1);
        }
        __syncthreads();
        if ((int)threadIdx.x < 64) {                                  // one wave: offsets of the pass, bases, reset
            const int t = threadIdx.x;
            const int c = t < kCoarseTiles ? hist[t] : 0;
            int inc = c;                                              // inclusive scan over the first 32 lanes
#pragma unroll
            for (int d = 1; d < kCoarseTiles; d <<= 1) {
                const int o = __shfl_up(inc, d, 64);
                if (t >= d) inc += o;
            }
            if (t < kCoarseTiles) {
                loff[t] = inc - c;
                gbase[t] = cursor[t];
                cursor[t] += c;
                hist[t] = 0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kFineAhead; ++u) {
            const int k = u * kFineThreads + (int)threadIdx.x;
            if (k < m) {
                const int t = (int)(w[u] >> kCoarseIdBits);
                const int lp = loff[t] + rank[u];
                ids[lp] = (int)(w[u] & ((1u << kCoarseIdBits) - 1u));
                tiles[lp] = (unsigned char)t;
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kFineAhead; ++u) {
            const int k = u * kFineThreads + (int)threadIdx.x;
            if (k < m) {
                const int t = tiles[k];
                bucket_ids[gbase[t] + (k - loff[t])] = ids[k];
            }
        }
        __syncthreads();                                              // ids / tiles / loff are rewritten by the next pass
    }
#else
    __syncthreads();
    for (int j = begin + threadIdx.x; j < end; j += kFineAhead * kFineThreads) {
        unsigned int w[kFineAhead];
#pragma unroll
        for (int u = 0; u < kFineAhead; ++u)
            if (j + u * kFineThreads < end) w[u] = (unsigned int)scratch[j + u * kFineThreads];
#pragma unroll
        for (int u = 0; u < kFineAhead; ++u)
            if (j + u * kFineThreads < end)
                bucket_ids[atomicAdd(&cursor[w[u] >> kCoarseIdBits], 1)] = (int)(w[u] & ((1u << kCoarseIdBits) - 1u));
    }
#endif
}

// ---- per-tile bitonic sort ---------------------------------------------------------------------
constexpr int kSortCap = 4096;   // keys held in LDS (32 KiB); larger buckets sort in global memory

#include "sort_network.h"

struct IdKeyArray {                 // in-place view of a bucket of ids compared through their keys
    volatile int* ids;
    const float* depths;
};

__device__ __forceinline__ void cmpswap(IdKeyArray a, int i, int j) {
    const int x = a.ids[i], y = a.ids[j];
    if (make_key(a.depths, x) > make_key(a.depths, y)) { a.ids[i] = y; a.ids[j] = x; }
}

template <typename Ptr>
__device__ __forceinline__ void bitonic_network(Ptr a, int n) {
    int lg = 0;                                   // npad = 2^lg >= n
    while ((1 << lg) < n) ++lg;
    const int half = (1 << lg) >> 1;
    // all strides are powers of two: index arithmetic is shifts and masks (no integer division)
    for (int lk = 1; lk <= lg; ++lk) {            // merge blocks of size k = 2^lk
        const int k = 1 << lk, hk = k >> 1;
        for (int p = threadIdx.x; p < half; p += kThreads) {          // flip
            const int base = (p >> (lk - 1)) << lk, off = p & (hk - 1);
            const int i = base + off, j = base + k - 1 - off;
            if (j < n) cmpswap(a, i, j);
        }
        __syncthreads();
        for (int ld = lk - 2; ld >= 0; --ld) {                        // disperse, stride d = 2^ld
            const int d = 1 << ld;
            for (int p = threadIdx.x; p < half; p += kThreads) {
                const int i = ((p >> ld) << (ld + 1)) | (p & (d - 1)), j = i + d;
                if (j < n) cmpswap(a, i, j);
            }
            __syncthreads();
        }
    }
}

// Register-resident bitonic sort of one bucket of ids by the whole workgroup (n <= 256 E): in-thread
// stages are pure register work, in-wave stages 64-bit lane exchanges through the LDS crossbar, and
// only partner distances >= 64 E (at most 3 of the 55 stages of a 1024-key sort) take an LDS round
// trip with a barrier.  `g` and `out` may alias (everything is loaded before anything is stored).
template <int E>
__device__ __forceinline__ void sort_tile_regs(const int* g, const float* __restrict__ depths,
                                               int* out, int n, unsigned long long* lds) {
    const int t = threadIdx.x;
    unsigned long long k[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = t * E + e;
        k[e] = i < n ? make_key(depths, g[i]) : ~0ull;
    }
    __syncthreads();                                  // aliasing callers: all loads before any store
    bitonic_regs<E, kThreads>(k, t, min(pow2_at_least(n), kThreads * E), lds);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = t * E + e;
        if (i < n) out[i] = (int)(unsigned int)k[e];
    }
}

__device__ __forceinline__ void sort_bucket_block(const int* g, const float* __restrict__ depths,
                                                  int* out, int n, unsigned long long* lds) {
    if (n <= kThreads) sort_tile_regs<1>(g, depths, out, n, lds);
    else if (n <= 2 * kThreads) sort_tile_regs<2>(g, depths, out, n, lds);
    else if (n <= 4 * kThreads) sort_tile_regs<4>(g, depths, out, n, lds);
    else if (n <= 8 * kThreads) sort_tile_regs<8>(g, depths, out, n, lds);
    else sort_tile_regs<16>(g, depths, out, n, lds);
}

// Tiles of more than kSampleMin keys (beyond the register network): per-tile SAMPLE SORT,
// O(n log n), streaming the tile's ids from global memory.  Measured on config 5 (2.5 k keys per
// tile) the network with its 16 independent keys per lane is 3x faster than this latency-chained
// scheme, so the network keeps every tile it can hold; the sample sort is the robust path for very
// dense tiles, replacing a global-memory network whose every stage is a barrier plus HBM round trips.
//   1. a regular sample of NS keys (256, or 1024 for n > 8192) is sorted with the network;
//   2. B - 1 splitters taken evenly from the sorted sample cut the key range into B sub-buckets of
//      ~96 keys on average (keys are unique, so equal depths cannot pile up in one sub-bucket);
//   3. two passes over the tile's ids: LDS histogram of sub-bucket sizes, then LDS cursors scatter
//      the ids into `out` grouped by sub-bucket;
//   4. each wave sorts whole sub-buckets (<= 256 keys) on its own, in place; the rare larger ones are
//      sorted afterwards by the whole workgroup.
// LDS (u64 units): scratch[4096] sample[1024] tables[1024] = 48 KiB.
// tiles up to kSortCap keys use the plain network (one wave up to kWaveSortMax, the workgroup beyond)
constexpr int kMaxSub = 512;              // sub-buckets per tile
constexpr int kSubTarget = 96;            // mean keys per sub-bucket aimed at
constexpr int kLargeLdsU64 = kSortCap + 2048;

__device__ __forceinline__ void sort_tile_sample(const int* __restrict__ g,
                                                 const float* __restrict__ depths,
                                                 int* __restrict__ out, int n,
                                                 unsigned long long* lds) {
    unsigned long long* scratch = lds;
    unsigned long long* smp = lds + kSortCap;
    int* start = reinterpret_cast<int*>(lds + kSortCap + 1024);       // [kMaxSub + 1]
    int* cursor = start + (kMaxSub + 1);                              // [kMaxSub]
    int* big = cursor + kMaxSub;                                      // [kMaxSub] oversize list
    __shared__ int nbig;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ns = n > 8192 ? 1024 : 256;
    int nb = n / kSubTarget;
    nb = nb < 2 ? 2 : (nb > kMaxSub ? kMaxSub : nb);

    if (ns == 1024) {                                                 // sorted sample
        unsigned long long k[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            k[e] = make_key(depths, g[(int)(((long long)(t * 4 + e) * n) >> 10)]);
        bitonic_regs<4, kThreads>(k, t, 1024, scratch);
#pragma unroll
        for (int e = 0; e < 4; ++e) smp[t * 4 + e] = k[e];
    } else {
        unsigned long long k[1];
        k[0] = make_key(depths, g[(int)(((long long)t * n) >> 8)]);
        bitonic_regs<1, kThreads>(k, t, 256, scratch);
        smp[t] = k[0];
    }
    for (int b = t; b < nb; b += kThreads) cursor[b] = 0;
    if (t == 0) nbig = 0;
    __syncthreads();

    // sub-bucket of a key = number of splitters <= key; splitter j = smp[((j + 1) * ns) / nb]
    auto sub_of = [&](unsigned long long key) {
        int lo = 0, hi = nb - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (smp[((mid + 1) * ns) / nb] <= key) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    for (int i = t; i < n; i += kThreads) atomicAdd(&cursor[sub_of(make_key(depths, g[i]))], 1);
    __syncthreads();
    {   // exclusive scan of nb <= 512 counts: two per thread
        const int c0 = (2 * t < nb) ? cursor[2 * t] : 0, c1 = (2 * t + 1 < nb) ? cursor[2 * t + 1] : 0;
        int total;
        const int inc = block_inclusive_scan(c0 + c1, &total);
        const int ex = inc - (c0 + c1);
        if (2 * t < nb) start[2 * t] = ex;
        if (2 * t + 1 < nb) start[2 * t + 1] = ex + c0;
        if (t == 0) start[nb] = n;
    }
    __syncthreads();
    for (int b = t; b < nb; b += kThreads) cursor[b] = start[b];
    __syncthreads();
    for (int i = t; i < n; i += kThreads) {
        const int id = g[i];
        out[atomicAdd(&cursor[sub_of(make_key(depths, id))], 1)] = id;
    }
    __threadfence_block();
    __syncthreads();
    for (int b = wave; b < nb; b += kThreads / 64) {                 // wave-level sorts, in place
        const int s0 = start[b], m = start[b + 1] - s0;
        if (m <= 1) continue;
        if (m <= 256) {
            unsigned long long k[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = lane * 4 + e;
                k[e] = i < m ? make_key(depths, out[s0 + i]) : ~0ull;
            }
            bitonic_regs<4, 64>(k, lane, min(pow2_at_least(m), 256), nullptr);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = lane * 4 + e;
                if (i < m) out[s0 + i] = (int)(unsigned int)k[e];
            }
        } else if (lane == 0) {
            big[atomicAdd(&nbig, 1)] = b;
        }
    }
    __syncthreads();
    const int nbg = nbig;
    for (int q = 0; q < nbg; ++q) {                                   // rare: oversize sub-buckets
        const int b = big[q];
        const int s0 = start[b], m = start[b + 1] - s0;
        __syncthreads();
        if (m <= kSortCap) {
            sort_bucket_block(out + s0, depths, out + s0, m, scratch);
        } else {
            IdKeyArray a{out + s0, depths};
            bitonic_network(a, m);
        }
        __syncthreads();
    }
}

// Three launches: the first kernel gives every tile of up to kWaveSortMax keys to one wave (four tiles per
// workgroup) and queues those beyond kSortCap; the second sorts the tiles in between with one workgroup per
// tile; the third walks the queue (sample sort).
__global__ __launch_bounds__(kThreads) void sort_tiles_small_kernel(
    int num_tiles, const int* __restrict__ tile_bins, const float* __restrict__ depths,
    const int* __restrict__ bucket_ids, int* __restrict__ ids_sorted, int* __restrict__ large_count,
    int* __restrict__ large_list) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (tile >= num_tiles) return;
    const int2 range = reinterpret_cast<const int2*>(tile_bins)[tile];
    const int n = range.y - range.x;
    if (n <= 0) return;
    if (n > kWaveSortMax) {                      // (kWaveSortMax, kSortCap]: sort_tiles_mid_kernel; beyond: queued
        if (n > kSortCap && lane == 0) large_list[atomicAdd(large_count, 1)] = tile;
        return;
    }
    const int* g = bucket_ids + range.x;
    int* out = ids_sorted + range.x;
    if (n <= 64) sort_tile_wave<1>(g, depths, out, n, lane);
    else if (n <= 128) sort_tile_wave<2>(g, depths, out, n, lane);
    else if (n <= 256) sort_tile_wave<4>(g, depths, out, n, lane);
    else if (n <= 512) sort_tile_wave<8>(g, depths, out, n, lane);
    else sort_tile_wave<16>(g, depths, out, n, lane);
}

// Tiles beyond one wave's network.  Up to kSortCap keys: one workgroup per tile, the register network with LDS
// for the cross-wave stages (32 KiB of LDS: five workgroups per CU), launched over ALL tiles - a workgroup whose
// tile belongs to another kernel leaves at once.  Walking a queue of such tiles from a persistent grid instead
// (one launch for everything above a wave's network, tried in round 3) saved 5 us on config 3 and ran config 5's
// sort - 14 k tiles of ~1 900 keys - at 0.78 - 0.89 instead of 0.55 ms.  Beyond kSortCap: the sample sort
// (48 KiB), a small persistent grid over the queue that sort_tiles_small_kernel filled.
// large_count / large_list (ts_sort_tiles_above: no sort_tiles_small_kernel runs): this kernel queues the tiles
// beyond kSortCap for the sample sort itself.
__global__ __launch_bounds__(kThreads) void sort_tiles_mid_kernel(
    const int* __restrict__ tile_bins, const float* __restrict__ depths,
    const int* __restrict__ bucket_ids, int* __restrict__ ids_sorted, int* __restrict__ large_count,
    int* __restrict__ large_list) {
    __shared__ unsigned long long lk[kSortCap / 2];
    const int2 range = reinterpret_cast<const int2*>(tile_bins)[blockIdx.x];
    const int n = range.y - range.x;
    if (n > kSortCap && large_count && threadIdx.x == 0) large_list[atomicAdd(large_count, 1)] = blockIdx.x;
    if (n <= kWaveSortMax || n > kSortCap) return;
    sort_bucket_block(bucket_ids + range.x, depths, ids_sorted + range.x, n, lk);
}

__global__ __launch_bounds__(kThreads) void sort_tiles_large_kernel(
    const int* __restrict__ tile_bins, const float* __restrict__ depths,
    const int* __restrict__ bucket_ids, int* __restrict__ ids_sorted,
    const int* __restrict__ large_count, const int* __restrict__ large_list) {
    __shared__ unsigned long long lds_large[kLargeLdsU64];
    const int count = *large_count;
    for (int q = blockIdx.x; q < count; q += gridDim.x) {
        const int2 range = reinterpret_cast<const int2*>(tile_bins)[large_list[q]];
        const int n = range.y - range.x;
        if (n <= kSortCap) continue;
        __syncthreads();
        sort_tile_sample(bucket_ids + range.x, depths, ids_sorted + range.x, n, lds_large);
    }
}

// ts_sort_tiles_above in ONE launch (the companion of ts_raster_fwd_sort, which sorts the lists of up to kWaveSortMax
// entries itself): one workgroup per tile, which leaves at once unless its list is longer - then the register
// network with LDS stages (up to kSortCap keys) or the sample sort, on the same 48 KiB.  On frames that take this path
// (16x16 lists: fewer than ~1 000 pairs per tile on average) long lists are the exception, so the sample sort's LDS
// footprint costs nothing here, and the queue of oversized tiles with its second launch is gone.
__global__ __launch_bounds__(kThreads) void sort_tiles_above_kernel(
    const int* __restrict__ tile_bins, const float* __restrict__ depths,
    const int* __restrict__ bucket_ids, int* __restrict__ ids_sorted) {
    __shared__ unsigned long long lds_u64[kLargeLdsU64 > kSortCap / 2 ? kLargeLdsU64 : kSortCap / 2];
    const int2 range = reinterpret_cast<const int2*>(tile_bins)[blockIdx.x];
    const int n = range.y - range.x;
    if (n <= kWaveSortMax) return;
    if (n <= kSortCap) sort_bucket_block(bucket_ids + range.x, depths, ids_sorted + range.x, n, lds_u64);
    else sort_tile_sample(bucket_ids + range.x, depths, ids_sorted + range.x, n, lds_u64);
}

__global__ __launch_bounds__(kThreads) void pack_splats_kernel(int n, const ts::PackArgs a,
                                                              const float* __restrict__ colors) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    if (a.radii[i] <= 0) return;
    float c0, c1, c2, c3 = 0.0f;
    if (a.channels == 4 && a.depths == nullptr) {
        const float4 c = reinterpret_cast<const float4*>(colors)[i];
        c0 = c.x; c1 = c.y; c2 = c.z; c3 = c.w;
    } else {             // [n,3] colours; channel 3 of an RGB + depth frame = depths (rasterize.py:48-50)
        c0 = colors[3 * i]; c1 = colors[3 * i + 1]; c2 = colors[3 * i + 2];
        if (a.channels == 4) c3 = a.depths[i];
    }
    ts::pack_one(a, i, c0, c1, c2, c3);
}

inline int launch_status() { return (int)hipGetLastError(); }

}  // namespace

extern "C" {

int32_t ts_num_tiles(const ts_camera* cam) {
    if (!cam) return 0;
    const int tbx = cam->wide_tiles ? (cam->tile_bounds_x + 1) / 2 : cam->tile_bounds_x;
    return cam->tile_rows * tbx;
}

int64_t ts_scan_ws_ints(int32_t n) {
    return n <= 0 ? 1 : (int64_t)((n + kScanBlock - 1) / kScanBlock);
}

int ts_scan_tiles(int32_t n, const int32_t* num_tiles_hit, int32_t* cum_tiles_hit,
                  int32_t* scan_ws, int32_t* total_out, void* stream) {
    if (n < 0) return TS_E_BADARG;
    if (n == 0) return 0;
    if (!num_tiles_hit || !cum_tiles_hit || !scan_ws) return TS_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (n + kScanBlock - 1) / kScanBlock;
    hipLaunchKernelGGL(scan_local_kernel, dim3(nb), dim3(kThreads), 0, s, n, num_tiles_hit,
                       cum_tiles_hit, scan_ws, total_out);
    if (nb > 1)
        hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(kThreads), 0, s, n, cum_tiles_hit, scan_ws,
                           total_out);
    return launch_status();
}

int64_t ts_bin_ws_ints(int32_t n, int32_t num_tiles) {
    if (num_tiles < 0) num_tiles = 0;
    return (int64_t)(bin_num_chunks(n) + 1) * num_tiles + 3;     // counts | tile_start[T + 1] | guard | the spare word
}



int ts_bin_count(int32_t n, const float* xys, const int32_t* radii, const float* splats,
                 const ts_camera* cam, int32_t* bin_ws, void* stream) {
    if (n < 0 || !cam || !bin_ws) return TS_E_BADARG;
    const int nt = ts_num_tiles(cam);
    if (nt <= 0) return 0;
    if (n > 0 && (!xys || !radii)) return TS_E_BADARG;
    const int chunks = bin_num_chunks(n);
    const int chunk = n > 0 ? (n + chunks - 1) / chunks : 1;
    const int window = bin_window_tiles(n, nt);
    const int windows = (nt + window - 1) / window;
    const size_t lds = (size_t)window * sizeof(int);
    // (the load-balanced walk does not pay for the count: 26 -> 28 us on config 3, 111 -> 149 us on config 5)
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bin_count_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(bin_count_kernel<false>, dim3(chunks, windows), dim3(kBinThreads), lds,
                       (hipStream_t)stream, n, chunk, xys, radii,
                       reinterpret_cast<const float4*>(splats), *cam, nt, window, bin_ws);
    return launch_status();
}

int ts_tile_offsets(int32_t n, int32_t num_tiles, int32_t* bin_ws, int32_t* tile_bins,
                    const int32_t* cum_tiles_hit, int64_t capacity, void* stream) {
    return ts_tile_offsets_stats(n, num_tiles, bin_ws, tile_bins, cum_tiles_hit, capacity, nullptr, stream);
}

int ts_tile_offsets_stats(int32_t n, int32_t num_tiles, int32_t* bin_ws, int32_t* tile_bins,
                          const int32_t* cum_tiles_hit, int64_t capacity, int32_t* longest_list, void* stream) {
    if (n < 0 || num_tiles < 0) return TS_E_BADARG;
    if (num_tiles == 0) return 0;
    if (!bin_ws || !tile_bins) return TS_E_BADARG;
    const int chunks = bin_num_chunks(n);
    int* tile_total = bin_ws + (size_t)chunks * num_tiles;     // [T + 1]: becomes tile_start, [T] = grand total
    const int per_group = (chunks + kScanGroups - 1) / kScanGroups;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(column_scan_kernel, dim3((num_tiles + kColTiles - 1) / kColTiles),
                       dim3(kColTiles * kScanGroups), 0, s, num_tiles, chunks, per_group, bin_ws, tile_total);
    hipLaunchKernelGGL(tile_offsets_kernel, dim3(1), dim3(kOffsetsThreads), 0, s, num_tiles, tile_total,
                       tile_bins, bin_ws + (ts_bin_ws_ints(n, num_tiles) - 1),
                       (cum_tiles_hit && n > 0) ? cum_tiles_hit + (n - 1) : nullptr, (long long)capacity, longest_list);
    return launch_status();
}

int ts_bin_scatter(int32_t n, const float* xys, const int32_t* radii, const float* splats,
                   const ts_camera* cam, const int32_t* bin_ws, int32_t* bucket_ids, int32_t* scratch,
                   void* stream) {
    if (n < 0 || !cam) return TS_E_BADARG;
    if (n == 0) return 0;
    if (!xys || !radii || !bin_ws || !bucket_ids) return TS_E_BADARG;
    const int nt = ts_num_tiles(cam);
    if (nt <= 0) return 0;
    const int chunks = bin_num_chunks(n);
    const int chunk = (n + chunks - 1) / chunks;
    const int* tile_start = bin_ws + (size_t)chunks * nt;
    // two coalesced hops (see bin_scatter_coarse_kernel) where the write amplification of the direct scatter is what
    // binds; a small launch (a 100 k scene, the ~125 k records of one rank of a sharded frame) is latency-bound and
    // the second launch costs more than it saves (34 vs 12 us on a 1/8 stripe of config 3)
    if (scratch && n >= kTwoHopFrom && n < (1 << kCoarseIdBits)) {
        const int groups = (nt + kCoarseTiles - 1) >> kCoarseShift;
        // load-balanced walk where the caller says a Gaussian covers many tiles (ts_camera.hints & TS_HINT_BALANCED_WALK):
        // config 5 (16 bounding-box tiles per Gaussian) coarse hop 369 -> 274 us; config 3 (6 tiles) 51 -> 56 us
        if (TS_BIN_BALANCED && (cam->hints & TS_HINT_BALANCED_WALK))
            hipLaunchKernelGGL(bin_scatter_coarse_kernel<true>, dim3(chunks), dim3(kBinThreads),
                               (size_t)groups * sizeof(int), (hipStream_t)stream, n, chunk, xys, radii,
                               reinterpret_cast<const float4*>(splats), *cam, nt, bin_ws, tile_start, scratch);
        else
            hipLaunchKernelGGL(bin_scatter_coarse_kernel<false>, dim3(chunks), dim3(kBinThreads),
                               (size_t)groups * sizeof(int), (hipStream_t)stream, n, chunk, xys, radii,
                               reinterpret_cast<const float4*>(splats), *cam, nt, bin_ws, tile_start, scratch);
        hipLaunchKernelGGL(bin_scatter_fine_kernel, dim3(groups), dim3(kFineThreads), 0, (hipStream_t)stream, nt,
                           tile_start, scratch, bucket_ids);
        return launch_status();
    }
    const int window = bin_window_tiles(n, nt);
    const int windows = (nt + window - 1) / window;
    const size_t lds = (size_t)window * sizeof(int);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bin_scatter_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(bin_scatter_kernel, dim3(chunks, windows), dim3(kBinThreads), lds,
                       (hipStream_t)stream, n, chunk, xys, radii,
                       reinterpret_cast<const float4*>(splats), *cam, nt, window, bin_ws,
                       tile_start, bucket_ids);
    return launch_status();
}

int ts_sort_tiles(int32_t num_tiles, const int32_t* tile_bins, const float* depths,
                  const int32_t* bucket_ids, int32_t* gaussian_ids_sorted, int32_t* sort_ws,
                  int32_t* zeroed_counter, void* stream) {
    if (num_tiles < 0) return TS_E_BADARG;
    if (num_tiles == 0) return 0;
    if (!tile_bins || !depths || !bucket_ids || !gaussian_ids_sorted || !sort_ws) return TS_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    // counter + list of the tiles beyond the sorting network.  zeroed_counter: a word the caller knows to
    // be zero (ts_tile_offsets leaves the last word of its workspace so) - saves a 4-byte memset launch
    int32_t* counter = zeroed_counter ? zeroed_counter : sort_ws;
    int32_t* list = zeroed_counter ? sort_ws : sort_ws + 1;
    if (!zeroed_counter) {
        hipError_t e = hipMemsetAsync(sort_ws, 0, sizeof(int32_t), s);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(sort_tiles_small_kernel, dim3((num_tiles + kThreads / 64 - 1) / (kThreads / 64)),
                       dim3(kThreads), 0, s, (int)num_tiles, tile_bins, depths, bucket_ids,
                       gaussian_ids_sorted, counter, list);
    hipLaunchKernelGGL(sort_tiles_mid_kernel, dim3(num_tiles), dim3(kThreads), 0, s, tile_bins, depths,
                       bucket_ids, gaussian_ids_sorted, (int*)nullptr, (int*)nullptr);
    const int grid = num_tiles < 768 ? num_tiles : 768;
    hipLaunchKernelGGL(sort_tiles_large_kernel, dim3(grid), dim3(kThreads), 0, s, tile_bins, depths,
                       bucket_ids, gaussian_ids_sorted, counter, list);
    return launch_status();
}

int ts_sort_tiles_above(int32_t num_tiles, const int32_t* tile_bins, const float* depths,
                        const int32_t* bucket_ids, int32_t* gaussian_ids_sorted, int32_t* sort_ws,
                        int32_t* zeroed_counter, void* stream) {
    if (num_tiles < 0) return TS_E_BADARG;
    if (num_tiles == 0) return 0;
    if (!tile_bins || !depths || !bucket_ids || !gaussian_ids_sorted) return TS_E_BADARG;
    (void)sort_ws; (void)zeroed_counter;        // (no queue of oversized tiles any more: every tile has its workgroup)
    hipLaunchKernelGGL(sort_tiles_above_kernel, dim3(num_tiles), dim3(kThreads), 0, (hipStream_t)stream, tile_bins,
                       depths, bucket_ids, gaussian_ids_sorted);
    return launch_status();
}

int ts_pack_splats(int32_t n, int32_t channels, int32_t flags, const float* xys, const int32_t* radii,
                   const float* conics, const float* colors, const float* opacity,
                   const int32_t* cum_tiles_hit, const ts_camera* cam, const float* depths, float* splats,
                   void* stream) {
    if (n < 0 || !cam || (channels != 3 && channels != 4)) return TS_E_BADARG;
    if (n == 0) return 0;
    if (!xys || !radii || !conics || !colors || !opacity || !cum_tiles_hit || !splats)
        return TS_E_BADARG;
    ts::PackArgs a;
    a.channels = channels; a.flags = (int)flags; a.xys = xys; a.radii = radii; a.conics = conics;
    a.opacity = opacity; a.cum_tiles_hit = cum_tiles_hit; a.depths = depths;
    a.splats = reinterpret_cast<float4*>(splats); a.cam = *cam;
    hipLaunchKernelGGL(pack_splats_kernel, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0,
                       (hipStream_t)stream, n, a, colors);
    return launch_status();
}

}  // extern "C"
