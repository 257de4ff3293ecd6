// raster.hip - tile compositing kernels for gfx950 (forward, backward, gradient row reduce).
//
// What gsplat's rasterize_forward / rasterize_backward compute for tinysplat's calls at
// /root/reference/tinysplat/splatting/rasterize.py:44,50 (front-to-back alpha compositing of the
// depth-sorted per-tile lists, and the back-to-front replay that produces v_xy / v_conic /
// v_colors / v_opacity), re-designed around the 64-wide wavefront:
//
//   * ONE WAVE OWNS ONE 16x16 TILE, seen as four 8x8 pixel blocks.  Lane l sits at (l & 7, l >> 3)
//     of every block: four pixels per lane, and one VALU instruction covers exactly one block, so
//     a block is the unit of skipping.  There is no workgroup barrier anywhere: the four waves of a
//     256-thread workgroup run four different tiles independently; early-out is a wave ballot.
//   * Per 64-entry chunk of the tile's sorted list each lane gathers ONE Gaussian's 48-byte packed
//     record (3 x 16 B loads) and tests it exactly against each 8x8 block (minimum of the conic
//     form over the block rectangle vs. the alpha >= 1/255 level set - conservative, so results
//     are unchanged).  Gaussians that can reach a still-unfinished block are compacted into LDS
//     with a wave ballot + prefix count, carrying a 4-bit block mask; the inner loop reads each
//     survivor back with wave-uniform (broadcast) ds_read_b128s and runs the per-pixel math only
//     for the blocks in its mask (scalar branches).  The rectangle a Gaussian is tested against is
//     not the whole block but the bounding box of the block's pixels that still matter (forward:
//     not yet saturated; backward: list already started), recomputed per chunk from a ballot with
//     scalar bit arithmetic, so dense scenes shed work pixel row by pixel row and stop early.
//   * Per-pixel bodies are branch free inside a block and lean: log2(opacity) is folded into the
//     exponent, the compositing weight is T_old - T_new (telescoping), a finished pixel is marked
//     by the sign of T, and Gaussians that need the sigma >= 0 / 0.999-clamp tests are flagged at
//     staging so that the common chunk runs a loop without them.  Nothing is set up per entry: a
//     flagged block evaluates dx, dy and the exponent itself (8 issues, sigma_l2).  What bounds both kernels
//     (profiles/HISTORY.md round 4): a wave issues one instruction per ~5 cycles whatever it
//     is, a tile is ~60 k of them, and 8 160 tiles on 4 096 - 5 120 wave slots are two rounds whose tail idles a
//     fifth of the launch; with four waves per SIMD the vector pipe (a wave64 instruction per 2 cycles: 78.6 T
//     lane-operations/s, of which these kernels reach 0.44 - 0.48) saturates in the full phase, so instruction
//     count AND kind set the time.  Packed fp32 does not help (a v_pk_fma_f32 costs 3.5 pipe cycles for two FMAs
//     and packing the natural pairs of the bodies - (dx, dy), (hA dx, hC dy), (v dx, v dy), the accumulators -
//     removed 12 % of the instructions and none of the time).  The file is built with -fno-slp-vectorize
//     because the SLP packer's register shuffles cost issue slots on top.
//   * Backward: per-lane partial sums over its <= 4 pixels, then a DPP butterfly that merges
//     the value vectors while it reduces (row_ror / row_half_mirror DPP under bank masks inside rows of 16,
//     v_permlane16_swap / v_permlane32_swap across rows) and a few lanes write one 48-byte row of raw sums per
//     (tile, Gaussian) into a slot that is contiguous per Gaussian.  reduce_partials sums each Gaussian's rows in a fixed order
//     and applies the conic / opacity factors once: no float atomics, and the gradients are
//     bit-reproducible run to run.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/tinysplat_hip.h"
#include "splat_math.h"

namespace {
#include "sort_network.h"

// tuning knobs (overridable with -D for the ablation runs of tools/time_raster.py)
#ifndef TS_FWD_MIN_WAVES
#define TS_FWD_MIN_WAVES 1             // __launch_bounds__ 2nd argument = min waves per SIMD
#endif
#ifndef TS_FWD_MIN_WAVES_RGB
#define TS_FWD_MIN_WAVES_RGB 5         // the 16x16, three-channel forward kernel keeps its five waves per SIMD (<= 96 VGPRs)
#endif
#ifndef TS_FWD_MIN_WAVES_RGBD
#define TS_FWD_MIN_WAVES_RGBD 5        // ... and so does the four-channel one since the next chunk's records land in LDS (TS_LDS_DMA):
                                       // the hybrid instantiation spills 160 bytes at chunk level, raster_fwd 327 -> 304 us (round 5)
#endif
#ifndef TS_BWD_MIN_WAVES
#define TS_BWD_MIN_WAVES 1
#endif
#ifndef TS_BWD_MIN_WAVES_16
#define TS_BWD_MIN_WAVES_16 5          // the 16x16 backward kernels at five waves per SIMD (96 VGPRs: 5 - 11 dwords spilled at
#endif                                 // chunk level with TS_LDS_DMA; they took 107 / 115).  Round 5: raster_bwd 465 -> 450 us

#ifndef TS_ABLATE
#define TS_ABLATE 0                    // timing experiments only (results are wrong when != 0): 3 = no per-entry loop at all
                                       // (what remains: prologue, sort, staging, epilogue); 4 = backward: the row's lane sums
                                       // added within the lane, no cross-lane reduction; 5 = the per-entry loop (record read,
                                       // mask dispatch, backward: `any` test) WITHOUT the block bodies and flushes; 6 = no
                                       // in-kernel sort (forward); 7 = backward: bodies, no flush at all; 8 = forward: the exponent evaluated twice per body
                                       // (tools/ablate_frame.sh; profiles/r06d_compositing_ablation.txt)
#endif

// TS_STATS=1 (developer build, tools/raster_stats.py; slow): dynamic work counters of the compositing kernels,
// ts_stats = {fwd staged entries, fwd block bodies, bwd staged entries, bwd block bodies entered, bwd bodies with
// a valid lane, bwd rows flushed, bwd valid lanes, bwd list entries walked, bwd bodies valid in one half only,
// bwd bodies valid in 1 / 2 / 3 of the four 4x4 quadrants}.
#ifndef TS_STATS
#define TS_STATS 0
#endif
#if TS_STATS
__device__ unsigned long long ts_stats[12];
#define TS_STAT(i, v)                                                                       \
    do {                                                                                    \
        if ((threadIdx.x & 63) == 0) atomicAdd(&ts_stats[i], (unsigned long long)(v));      \
    } while (0)
#else
#define TS_STAT(i, v) ((void)0)
#endif

// TS_TIMELINE=1 (developer build, tools/raster_timeline.py): every wave of the two compositing kernels records
// {start, end} on the 100 MHz constant clock, its hardware slot (HW_ID | XCC_ID << 16) and its list length.
#ifndef TS_TIMELINE
#define TS_TIMELINE 0
#endif
#if TS_TIMELINE
constexpr int kTimelineMax = 1 << 17;
constexpr int kTimelineRow = 12;
__device__ unsigned long long ts_timeline[2][kTimelineMax][kTimelineRow];
struct WaveClock {
    unsigned long long t0, c0;
    unsigned long long seg[4] = {0ull, 0ull, 0ull, 0ull};     // shader-clock ticks per segment (TS_SEG_*)
    unsigned int work[3] = {0u, 0u, 0u};                       // staged entries, block bodies, rows flushed
    int which, unit, n;
    __device__ WaveClock(int which_, int unit_, int n_) : which(which_), unit(unit_), n(n_) {
        t0 = __builtin_amdgcn_s_memrealtime();
        c0 = __builtin_amdgcn_s_memtime();
    }
    __device__ ~WaveClock() {
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if ((threadIdx.x & 63) == 0 && unit < kTimelineMax) {
            ts_timeline[which][unit][0] = t0;
            ts_timeline[which][unit][1] = t1;
            ts_timeline[which][unit][2] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
            ts_timeline[which][unit][3] = (unsigned long long)n;
            for (int i = 0; i < 4; ++i) ts_timeline[which][unit][4 + i] = seg[i];
            ts_timeline[which][unit][8] = c1 - c0;
            for (int i = 0; i < 3; ++i) ts_timeline[which][unit][9 + i] = work[i];
        }
    }
};
#define TS_WAVE_CLOCK(which, unit, n) WaveClock ts_wave_clock_(which, unit, n)
#define TS_SEG_PARAM , unsigned long long (&ts_seg_)[4], unsigned int (&ts_work_)[3]
#define TS_SEG_ARG , ts_wave_clock_.seg, ts_wave_clock_.work
#define TS_WORK(i, v) ts_work_[i] += (unsigned int)(v)
#define TS_SEG_T0(var)                                               \
    __builtin_amdgcn_sched_barrier(0);                               \
    const unsigned long long var = __builtin_amdgcn_s_memtime();     \
    __builtin_amdgcn_sched_barrier(0)
#define TS_SEG_ADD(segs, i, var)                                     \
    __builtin_amdgcn_sched_barrier(0);                               \
    segs[i] += __builtin_amdgcn_s_memtime() - var;                   \
    __builtin_amdgcn_sched_barrier(0)
#else
#define TS_WAVE_CLOCK(which, unit, n) ((void)0)
#define TS_SEG_PARAM
#define TS_SEG_ARG
#define TS_WORK(i, v) ((void)0)
#define TS_SEG_T0(var) ((void)0)
#define TS_SEG_ADD(segs, i, var) ((void)0)
#endif

// TS_LDS_PAD=<bytes> (developer build): extra LDS per workgroup, to pin the number of resident waves per SIMD
// (a workgroup's kWaves waves sit on different SIMDs, so waves per SIMD = workgroups per CU = 160 KiB / LDS).
#ifndef TS_LDS_PAD
#define TS_LDS_PAD 0
#endif
#if TS_LDS_PAD
#define TS_LDS_PAD_DECL() __shared__ int lds_pad_[TS_LDS_PAD / 4]; if (threadIdx.x == 9999) lds_pad_[blockIdx.x & 7] = 1; asm volatile("" : : "v"(&lds_pad_[0]) : "memory")
#else
#define TS_LDS_PAD_DECL() ((void)0)
#endif

// TS_LDS_DMA: the packed records of the NEXT chunk travel from global memory straight into LDS
// (global_load_lds_dwordx4: 16 bytes per lane to base + lane * 16) instead of through twelve VGPRs that stay live
// across the whole chunk - the registers that kept raster_bwd at four and raster_fwd at five waves per SIMD.
#ifndef TS_LDS_DMA
#define TS_LDS_DMA 1                   // (round 4: no gain on its own; round 5: what lets raster_bwd fit five waves per SIMD)
#endif
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__device__ __forceinline__ void dma_record(const float4* __restrict__ splats, int g, float4* raw) {
    const float4* p = splats + 3 * (size_t)g;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)p, (lds_ptr_t)raw, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p + 1), (lds_ptr_t)(raw + 64), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p + 2), (lds_ptr_t)(raw + 128), 16, 0, 0);
}
#define TS_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define TS_LDS_WAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

#ifndef TS_FLAG_WITH_ROW
#define TS_FLAG_WITH_ROW (CH == 4)     // measured: four channels 561 -> 544 us, three channels 507 -> 512 us
#endif
#ifndef TS_SELECT_SGPR
#define TS_SELECT_SGPR 1
#endif
#ifndef TS_BWD_EXEC_MASK
#define TS_BWD_EXEC_MASK 0             // measured (round 5): the body's second half under EXEC = valid lanes: 450 -> 459 us
#endif
#ifndef TS_BWD_EARLY_RECORD
#define TS_BWD_EARLY_RECORD 0          // measured (round 5): next record requested in front of the row flush: 451 -> 458 - 465 us
#endif

#ifndef TS_NT_ROWS
#define TS_NT_ROWS 0                    // gradient rows written (raster_bwd) / read (reduce_partials) non-temporally
#endif

#ifndef TS_REDUCE_AHEAD
#define TS_REDUCE_AHEAD 4                // rows of a Gaussian requested together by reduce_partials
#endif

#ifndef TS_RASTER_WAVES
#define TS_RASTER_WAVES 4
#endif
constexpr int kWaves = TS_RASTER_WAVES;   // tiles (= waves) per workgroup; waves never synchronise
constexpr int kThreads = 64 * kWaves;
// float4s per gradient row slot.  (Round 4 tried 64-byte slots, whole-sector stores: WRITE_SIZE 218 -> 221 MB,
// reduce_partials 57 -> 60 us - the 1.9x "write amplification" of round 3 is the row's FLAG BYTE, a 32-byte sector
// write of its own per row (2.41 M x (64 + 32) B = 231 MB), not rows straddling lines.)
constexpr int kRowF4 = TS_PARTIAL_ROW_FLOATS / 4;
static_assert(TS_PARTIAL_ROW_FLOATS % 4 == 0 && TS_PARTIAL_ROW_FLOATS >= 12, "row = whole float4s, >= 10 values");
using ts::kLog2e;
using ts::kLog2_255;

#define TS_WAVE_SYNC()                                            \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
    } while (0)

// workgroup barrier that orders LDS traffic only: global loads (a prefetch) and stores of the wave stay in flight,
// where __syncthreads() waits for all of them
#define TS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ int wave_max_int(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}

// sigma * log2(e) - log2(opacity) of one pixel from d = xy - pixel: six explicit operations in a
// fixed order, so that forward and backward (which must replay the forward's alpha >= 1/255
// decisions) evaluate bit-identical values whatever the optimiser does.
// Evaluated PER FLAGGED BLOCK (8 issues with dx, dy), with nothing set up per entry.  The first
// version shared dx, dy, hA dx^2, B dx and hC dy^2 - lo between the two halves of the tile: 17 issues
// per entry plus 2 per block, which pays from 2.8 blocks per entry up - but with tight tile lists and
// the per-block cull an entry reaches 1.2 blocks on average, so the shared set-up was mostly spent on
// halves that are never evaluated.
__device__ __forceinline__ float sigma_l2(float hA, float B, float hC, float neg_lo, float dx, float dy) {
    float s = __builtin_fmaf(hC * dy, dy, neg_lo);      // hC dy^2 - log2(opacity)
    s = __builtin_fmaf(hA * dx, dx, s);
    return __builtin_fmaf(B * dx, dy, s);
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

struct Staged {          // what a lane derives from the Gaussian it gathered
    int mask;            // bit k set: the alpha >= 1/255 level set may reach 8x8 block k of the tile (bit NB: see below)
    float gx, gy, hA, B, hC, lo;   // conic and opacity in the log2 domain
};

// Bounding rectangle (in lane coordinates 0..7) of the pixels of an 8x8 block selected by a 64-bit
// lane mask (bit = ly*8 + lx).  Pure scalar bit arithmetic on the ballot.  Returns false if empty.
struct BlockRect { float x0, x1, y0, y1; };          // inclusive sample-position bounds
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

// Tile shapes.  NBX = 8x8 blocks per tile row: 2 -> the 16x16 tile of gsplat's lists (four blocks, four
// pixels per lane); 4 -> a WIDE 32x16 tile (eight blocks, eight pixels per lane) made of two
// horizontally adjacent 16x16 tiles whose lists were binned as one (ts_camera.wide_tiles).  A Gaussian
// that reaches both halves is then listed, staged, reduced over the wave and written as a gradient row
// ONCE instead of twice: on the random scenes the wide lists hold 0.73x the pairs, and the per-entry
// cost (the cross-lane reduction of the backward pass above all) shrinks with them, while the number
// of block bodies - the per-pixel work - is unchanged.  Block k sits at (bx, by) = (k % NBX, k / NBX).
//
// gather + exact cull against the NB 8x8 pixel blocks of the tile.
// rects[k] = sample-position bounding rectangle of the pixels of block k that still matter (all of
// the block at first; it shrinks as pixels saturate in the forward pass / covers only the pixels
// whose lists have started in the backward pass), kept in LDS so that the rolled loop can index it.
// blocks = bit mask of the blocks whose rectangle is non-empty.
template <int NB>
__device__ __forceinline__ Staged stage_splat(bool have, const float4 q0, const float4 q1,
                                              const float4* __restrict__ rects, int blocks) {
    Staged s;
    s.gx = q0.x; s.gy = q0.y;
    const float A = q0.w, Bc = q1.x, Cc = q1.y, op = q0.z;
    s.hA = 0.5f * kLog2e * A;
    s.B = kLog2e * Bc;
    s.hC = 0.5f * kLog2e * Cc;
    s.lo = __log2f(op);
    s.mask = 0;
    // bit NB ("general"): the per-pixel code must test sigma >= 0 and apply the 0.999 clamp.  For a
    // positive-definite conic and opacity <= 0.99 neither can trigger (sigma >= 0 up to rounding,
    // alpha = opacity * exp(-sigma) <= opacity), and the kernels take a leaner wave-uniform path.
    const bool general = !(s.hA > 0.0f && s.hC > 0.0f && 4.0f * s.hA * s.hC > s.B * s.B && op <= 0.99f);
    if (have && op > 0.0f) {
        const float tau = s.lo + kLog2_255;            // sigma' <= tau  <=>  alpha >= 1/255
        if (tau >= -0.02f) {
            if (s.hA > 0.0f && s.hC > 0.0f) {
                const float inv2A = 0.5f / s.hA, inv2C = 0.5f / s.hC;
#pragma unroll 1
                for (int k = 0; k < NB; ++k) {          // rolled: runs once per 64 entries, keeps VGPRs low
                    if (!(blocks & (1 << k))) continue;
                    const float4 r = rects[k];          // {x0, x1, y0, y1}, wave-uniform
                    if (ts::rect_may_contribute(s.hA, s.B, s.hC, inv2A, inv2C, tau, s.gx, s.gy, r.x, r.y, r.z,
                                                r.w))
                        s.mask |= (1 << k);
                }
            } else {
                s.mask = blocks;                        // not a PSD conic: no geometric cull
            }
            if (s.mask != 0 && general) s.mask |= (1 << NB);
        }
    }
    return s;
}

// Writes rects[k] for the pixels selected by `sel[k]` (one bool per lane and block) and returns the
// mask of non-empty blocks.  Lane 0 stores; callers fence before reading.
template <int NBX>
__device__ __forceinline__ int update_rects(const bool (&sel)[2 * NBX], float X0, float Y0, float4* rects,
                                            int lane) {
    int blocks = 0;
#pragma unroll
    for (int k = 0; k < 2 * NBX; ++k) {
        const unsigned long long m = __ballot(sel[k]);
        int xmin, xmax, ymin, ymax;
        if (mask_rect(m, xmin, xmax, ymin, ymax)) {
            blocks |= (1 << k);
            const float bx = X0 + (float)(8 * (k % NBX)), by = Y0 + (float)(8 * (k / NBX));
            if (lane == 0)
                rects[k] = make_float4(bx + (float)xmin, bx + (float)xmax, by + (float)ymin,
                                       by + (float)ymax);
        }
    }
    return blocks;
}

// Blocks of a wide tile a Gaussian may be composited into: gsplat lists a Gaussian in the 16x16 tiles
// of its tile box only, so a half of the wide tile that lies outside the box [minx, minx + w) must not
// see it even where its level set reaches in (record: bbox_w | bbox_minx << 16).  Bit NB is kept.
// WL (16x16 waves reading WIDE lists): the wave's tile column tx must itself lie in the box.
template <int NBX, bool WL>
__device__ __forceinline__ int bbox_blocks(int wm, int tx) {
    const int minx = wm >> 16, w = wm & 0xffff;
    if (NBX == 2) return (!WL || (tx >= minx && tx < minx + w)) ? ~0 : 0;
    return ((2 * tx >= minx) ? 0x33 : 0) | ((2 * tx + 1 < minx + w) ? 0xCC : 0) | (1 << (2 * NBX));
}

using mask64 = unsigned long long;
#define TS_BALLOT(c) __builtin_amdgcn_ballot_w64(c)          // lane condition -> 64-bit scalar mask
#define TS_LANE(m) __builtin_amdgcn_inverse_ballot_w64(m)    // scalar mask -> lane condition

// LIST SEGMENTS (bits 8..11 of ts_camera.hints = S > 1; 16x16 lists).  The backward pass of a tile is a chain over
// its sorted list.  A launch of a few hundred tiles - a rank's stripe of a sharded frame, a small image - cannot
// fill 1 024 SIMDs with one chain per tile; four waves per tile (TS_RASTER_SPLIT_BLOCKS) fill them, but every one
// of the four walks and stages the WHOLE list for a quarter of the pixels and writes its own gradient row.  The
// chain can be CUT instead: the forward pass keeps, per pixel, its state at the segment boundaries b_s - the
// transmittance T_s in front of entry b_s and the colour B_s that the entries at and behind b_s contribute - and the
// replay of the segment [b_s, b_s+1) starts from
//     T = T_(s+1),   R = T_fin (v_alpha - bg . v_out) - v_out . B_(s+1)
// which is what walking the entries behind it would have left (S_behind fac c . v_out = v_out . sum of their colour
// contributions), with the forward pass's own T instead of T_fin divided back through hundreds of (1 - alpha).
// B_s is NOT a difference of the running colour sum (that would know it only to a rounding of the whole pixel
// colour): the forward pass sums every segment's contributions on their own and adds them up from the back.
// A tile becomes up to S independent work items of ONE wave each over all four blocks; every (tile, Gaussian)
// belongs to exactly one of them, so there is one row per pair and the reduction is the plain one.  Boundaries are
// a function of the list range alone (equal entry counts, multiples of 64: seg_bound), so both passes - and the
// four waves of a split forward launch - agree on them without a table.  The gradients differ from the uncut
// pass's by rounding only.
// On a FULL frame (8 160 tiles) segments gain nothing (profiles/HISTORY.md, round 4: the machine is saturated and
// every extra item re-reads its pixel state); callers enable them for launches that would otherwise be split.
// Layout behind final_Ts (P = pixels of the launch, float planes; ts_final_planes):
//   plane 0: T_fin | s = 1..S-1: planes 1 + (s-1)(1+CH) + {0: T_s, 1+c: B_s[c]}
#define TS_CAM_SEGS(cam) (((cam).hints >> 8) & 15)
#ifndef TS_SEG_MIN_LIST
#define TS_SEG_MIN_LIST 65
#endif
constexpr int kSegMinList = TS_SEG_MIN_LIST;      // shorter lists (one chunk) stay one segment
constexpr int kSegMax = 8;
// Boundary s (1 .. S-1) of a list of `list_len` entries, in entries from its start (a multiple of 64), or a value
// >= list_len where the list has fewer segments; boundary 0 is the start.  Integer arithmetic on the list length
// alone: both passes (and the four waves of a split forward launch) compute the same values.  The FRONT segments are
// the expensive ones - every pixel is still alive there, late in the list most (Gaussian, block) pairs are culled -
// so the boundaries are not equidistant (TS_SEG_SHAPE: 0 = equal counts, 1 = halfway, 2 = quadratic).
#ifndef TS_SEG_SHAPE
#define TS_SEG_SHAPE 2
#endif
#ifndef TS_FWD_LATE_PREFETCH
#define TS_FWD_LATE_PREFETCH 0
#endif
#ifndef TS_FWD_CUT_FIRST
#define TS_FWD_CUT_FIRST 1
#endif
#ifndef TS_LOC_LDS
#define TS_LOC_LDS 0                   // 1: the per-segment sums of a whole-tile wave in LDS (LocLds) instead of registers: slower
#endif
#ifndef TS_SEG_CAP_CHUNKS
#define TS_SEG_CAP_CHUNKS 0
#endif
__device__ __forceinline__ int seg_bound(int list_len, int S, int s) {
    if (S <= 1 || list_len < kSegMinList) return 0x7fffff00;
    const int C = TS_SEG_CAP_CHUNKS > 0 ? min((list_len + 63) >> 6, TS_SEG_CAP_CHUNKS) : (list_len + 63) >> 6;
    int b = 0;
    for (int k = 1; k <= s; ++k) {                 // S <= 8: a few scalar operations
        int f;
        if (TS_SEG_SHAPE == 0) f = (C * k + S - 1) / S;
        else if (TS_SEG_SHAPE == 1) f = (C * (k * k + k * S) + 2 * S * S - 1) / (2 * S * S);
        else f = (C * k * k + S * S - 1) / (S * S);
        b = max(f, b + 1);                         // strictly increasing, at least one chunk per segment
    }
    return b << 6;
}
__host__ __device__ __forceinline__ size_t seg_plane_stride(const ts_camera& cam) {
    const int rows = min(16 * cam.tile_rows, cam.img_height - 16 * cam.tile_row0);
    return (size_t)cam.img_width * (size_t)max(rows, 0);
}

// HYBRID LAUNCH (bits 12..15 of ts_camera.hints = W16 in 1..15, together with S > 1; one wave per 16x16 tile on 16x16
// lists).  A full frame is two rounds of ~200-us waves (8 160 tiles on 4 096 / 5 120 wave slots) whose last third runs
// half empty; cutting EVERY list costs as much as the tail it removes (profiles/HISTORY.md, round 4).  So only the tiles
// that are dispatched LAST are cut: the tiles are handed out in eight bands (one per XCD, see xcd_tile_group), the first
// W16/16 of a band as whole tiles exactly as without segments, the rest as S list-segment items each - the small items
// fill the slots the whole tiles leave behind.  Which tile is cut is a function of (tile, num_tiles, W16) alone, so the
// forward pass (which keeps the boundary state for the cut tiles only) and the backward pass agree without a table.
// W16 = 0 with S > 1: every tile is cut (the small launches).
// CHECKPOINTS live behind final_Ts at float offset ckpt_offset(P): one block per cut tile of S RECORDS of (1 + CH) 256
// floats, record r = 1 .. S:  {T_r = transmittance in front of boundary r (r < S), D_(r-1) = colour the entries of
// segment r - 1 contributed} per pixel - what the forward wave knows when it reaches boundary r, written with one
// 16-byte store per lane and block ([block k][lane] float4 {T, D_0, D_1, D_2}; a fourth channel in a plane of 256 floats
// behind them).  The backward item of segment s starts from T_(s+1) and the colour BEHIND it, D_(s+1) + ... + D_(S-1)
// = records s + 2 .. S added from the back (independent loads: one round trip; a running suffix sum kept by the
// forward wave was a chain of dependent load -> add -> store per boundary at the end of every cut tile).
#define TS_CAM_WHOLE16(cam) (((cam).hints >> 12) & 15)
struct CutTiles {
    int band, whole;                                   // tiles per band | of which composited whole (multiples of 4)
    __host__ __device__ int cut() const { return band - whole; }
    // -> index of the tile's checkpoint block, or -1 for a whole tile
    __host__ __device__ int block_of(int tile) const {
        const int b = tile / band, j = tile - b * band;
        return j < whole ? -1 : b * (band - whole) + (j - whole);
    }
    // work items of a band's backward launch
    __host__ __device__ int items(int S) const { return whole + (band - whole) * S; }
};
__host__ __device__ __forceinline__ CutTiles cut_tiles(int num_tiles, int hints) {
    CutTiles m;
    const int whole16 = (hints >> 12) & 15;
    m.band = 4 * ((num_tiles + 31) / 32);
    m.whole = whole16 > 0 ? ((m.band * whole16) / 16) & ~3 : 0;
    return m;
}
__host__ __device__ __forceinline__ size_t ckpt_offset(size_t plane) { return (plane + 63) & ~(size_t)63; }
// float offset of record r (1 .. S) of checkpoint block b
template <int CH>
__host__ __device__ __forceinline__ size_t ckpt_record(int block, int S, int r) {
    return ((size_t)block * (size_t)S + (size_t)(r - 1)) * (size_t)((1 + CH) * 256);
}
// record at `rec`: this lane's {T, D} of block k
template <int CH>
__device__ __forceinline__ void ckpt_store(float* rec, int k, int lane, float T, const float (&D)[CH]) {
    reinterpret_cast<float4*>(rec)[64 * k + lane] = make_float4(T, D[0], D[1], D[2]);
    if (CH == 4) rec[1024 + 64 * k + lane] = D[CH - 1];
}

// Composites the `cnt` staged Gaussians of one chunk into the wave's tile (forward).
//   per pixel and block k: T > 0 = transmittance of an unfinished pixel; T < 0 = finished (or
//   outside the image) with final transmittance |T|; fidx = list index of the last Gaussian
//   composited; acc = colour.
// vis = |T_old| - |T_new| equals alpha*T up to one rounding of T, is 0 for the stopping Gaussian
// and for finished pixels, and makes the weights telescope (sum of vis = 1 - T_final exactly).
// GENERAL adds the sigma >= 0 test and the 0.999 clamp, which cannot trigger for a
// positive-definite conic with opacity <= 0.99 (bit 4 of the staged mask).
// Per-segment colour sums of the forward pass (LIST SEGMENTS): what the entries of the CURRENT segment contributed to
// each pixel, kept beside the running sum.  LocRegs<CH, N>: N sets in registers (0 = none; 1 = a wave of a split
// launch owns ONE block).  LocLds<CH>: one set per block of a whole-tile wave in LDS, accumulated with ds_add_f32 -
// twelve more live registers cost the forward kernel its fifth wave per SIMD (122 VGPRs, or 31 spilled), the LDS
// form costs three multiplies and three LDS atomics per body of a cut tile and no register.
template <int CH, int N>
struct LocRegs {
    static constexpr bool on = N > 0;
    float v[N > 0 ? N : 1][CH];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < (N > 0 ? N : 1); ++k)
#pragma unroll
            for (int c = 0; c < CH; ++c) v[k][c] = 0.0f;
    }
    __device__ __forceinline__ void add(int k, const float (&col)[CH], float vis) {
#pragma unroll
        for (int c = 0; c < CH; ++c) v[N == 1 ? 0 : k][c] = __builtin_fmaf(col[c], vis, v[N == 1 ? 0 : k][c]);
    }
    __device__ __forceinline__ void get(int k, float (&D)[CH]) const {
#pragma unroll
        for (int c = 0; c < CH; ++c) D[c] = v[N == 1 ? 0 : k][c];
    }
};
template <int CH, int NB>
struct LocLds {
    static constexpr bool on = true;
    float4* p;                                       // this lane's sums of block 0; block k at p[64 k] (16 bytes per lane)
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < NB; ++k) p[64 * k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ __forceinline__ void add(int k, const float (&col)[CH], float vis) {
        float4 v = p[64 * k];
        v.x = __builtin_fmaf(col[0], vis, v.x); v.y = __builtin_fmaf(col[1], vis, v.y); v.z = __builtin_fmaf(col[2], vis, v.z);
        if (CH == 4) v.w = __builtin_fmaf(col[CH - 1], vis, v.w);
        p[64 * k] = v;
    }
    __device__ __forceinline__ void get(int k, float (&D)[CH]) const {
        const float4 v = p[64 * k];
        D[0] = v.x; D[1] = v.y; D[2] = v.z;
        if (CH == 4) D[CH - 1] = v.w;
    }
};

// The per-pixel update of the forward pass behind the exponent, hand-scheduled (round 6, TS_FWD_ASM).  The compiler's
// code for the C++ form below spent 5 wait states (s_nop 0 / 1 / 1) and one register copy on every block body:
// it folds the negation of `-ae` and the `-|T|` of the stop into v_cndmask_b32_e64 source modifiers, and a VOP3 select
// that reads vcc as a CONSTANT needs two wait states behind the v_cmp that wrote it (an e32 select, which reads vcc
// implicitly, needs none); the new T was built in a temporary and moved.  Here: both value selects are e32 (the
// negation moves into the FMA, vis becomes the select of T - nT - the same bits as |T| - |Tn|: an unfinished pixel has
// T, nT > 0, and a stopping or finished one selects 0), the one select that keeps its modifiers (T <- -|T| on a stop)
// stands three instructions behind its compare, and T is updated in place: 21 VALU issues + 1 wait state per body where
// the compiler had 21 + 5 (measured: no change of the launch - a wait state is an issue slot of one wave that its four
// neighbours fill; what the launch pays for is the pipe cost of the 21 instructions, 56 cycles: DESIGN.md section 4).  Same operations on the same operands: image, final_Ts and final_index are bit for bit the
// C++ form's (tests/test_gpu_parity.py compares the frame with the oracle; tools/variant_check.py the two builds).
#ifndef TS_FWD_ASM
#define TS_FWD_ASM 1
#endif
// ae = exp2(-sgl) >= 1/255 ? exp2(-sgl) : 0
__device__ __forceinline__ float fwd_alpha(float sgl) {
#if TS_FWD_ASM
    float a, ae;
    asm("v_exp_f32_e64 %0, -%2\n\t"
        "s_nop 0\n\t"
        "v_cmp_le_f32_e32 vcc, %3, %0\n\t"
        "v_cndmask_b32_e32 %1, 0, %0, vcc"
        : "=&v"(a), "=v"(ae)
        : "v"(sgl), "s"(ts::kAlphaMin)
        : "vcc");
    return ae;
#else
    const float a = __builtin_amdgcn_exp2f(-sgl);
    return a >= ts::kAlphaMin ? a : 0.0f;
#endif
}
// the lean path (no clamp, no sigma >= 0 test between the two): ONE block, so that the compiler puts no wait state of
// its own between two asm statements
__device__ __forceinline__ float fwd_alpha_update(float sgl, float& T, int& fidx, int idx, float& a0, float& a1, float& a2,
                                                  float c0, float c1, float c2) {
    float a, ae, nT, d, vis;
    asm("v_exp_f32_e64 %0, -%10\n\t"
        "s_nop 0\n\t"
        "v_cmp_le_f32_e32 vcc, %15, %0\n\t"
        "v_cndmask_b32_e32 %1, 0, %0, vcc\n\t"
        "v_fma_f32 %2, -%1, %5, %5\n\t"
        "v_sub_f32_e32 %3, %5, %2\n\t"
        "v_cmp_nge_f32_e32 vcc, %16, %2\n\t"
        "v_cndmask_b32_e32 %4, 0, %3, vcc\n\t"
        "v_fmac_f32_e32 %6, %11, %4\n\t"
        "v_fmac_f32_e32 %7, %12, %4\n\t"
        "v_cndmask_b32_e64 %5, -|%5|, %2, vcc\n\t"
        "v_fmac_f32_e32 %8, %13, %4\n\t"
        "v_cmp_lt_f32_e32 vcc, 0, %4\n\t"
        "v_cndmask_b32_e32 %9, %9, %14, vcc"
        : "=&v"(a), "=&v"(ae), "=&v"(nT), "=&v"(d), "=&v"(vis), "+v"(T), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(fidx)
        : "v"(sgl), "v"(c0), "v"(c1), "v"(c2), "v"(idx), "s"(ts::kAlphaMin), "s"(ts::kTEps)
        : "vcc");
    return vis;
}
// T, fidx and the first three channels of acc updated for one pixel; returns vis (the weight of the colour)
__device__ __forceinline__ float fwd_update(float ae, float& T, int& fidx, int idx, float& a0, float& a1, float& a2,
                                            float c0, float c1, float c2) {
#if TS_FWD_ASM
    float nT, d, vis;
    asm("v_fma_f32 %0, -%8, %3, %3\n\t"                    // nT = T - ae T
        "v_sub_f32_e32 %1, %3, %0\n\t"                      // d = T - nT
        "v_cmp_nge_f32_e32 vcc, %13, %0\n\t"                // vcc = !(nT <= kTEps): the pixel goes on
        "v_cndmask_b32_e32 %2, 0, %1, vcc\n\t"              // vis = goes on ? d : 0
        "v_fmac_f32_e32 %4, %9, %2\n\t"
        "v_fmac_f32_e32 %5, %10, %2\n\t"
        "v_cndmask_b32_e64 %3, -|%3|, %0, vcc\n\t"          // T = goes on ? nT : -|T|   (three issues behind the v_cmp)
        "v_fmac_f32_e32 %6, %11, %2\n\t"
        "v_cmp_lt_f32_e32 vcc, 0, %2\n\t"                   // composited <=> vis > 0
        "v_cndmask_b32_e32 %7, %7, %12, vcc"
        : "=&v"(nT), "=&v"(d), "=&v"(vis), "+v"(T), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(fidx)
        : "v"(ae), "v"(c0), "v"(c1), "v"(c2), "v"(idx), "s"(ts::kTEps)
        : "vcc");
    return vis;
#else
    const float nT = __builtin_fmaf(-ae, T, T);
    const float Tn = nT <= ts::kTEps ? -__builtin_fabsf(T) : nT;   // the stopping Gaussian is not composited
    const float vis = __builtin_fabsf(T) - __builtin_fabsf(Tn);
    a0 = __builtin_fmaf(c0, vis, a0); a1 = __builtin_fmaf(c1, vis, a1); a2 = __builtin_fmaf(c2, vis, a2);
    fidx = vis > 0.0f ? idx : fidx;
    T = Tn;
    return vis;
#endif
}

// FINISHED PIXELS AS A SCALAR MASK (round 6, TS_FWD_ALIVE).  With instructions priced per class a
// forward body is 56 pipe cycles of which the three compare -> select decisions are 21: alpha >= 1/255, the stop test
// with its TWO selects (vis, T), and "was it composited" for fidx.  A pixel stops ONCE in its life, and "finished" was
// kept in the sign of T, which every body had to restore.  Now a block's unfinished pixels are a 64-bit mask in SGPRs:
//   m1 = ballot(alpha >= 1/255) & alive            (scalar and)          ae = m1 ? alpha : 0
//   nT = T - ae T,  d = T - nT  (= vis: 0 wherever ae = 0),   go = ballot(!(nT <= kTEps))
//   common case, no pixel of the block stops in this body:  acc += colour * d;  fidx = m1 ? idx : fidx;  T = nT
// - no select for vis or T, no compare for fidx (m1 is the answer): 46 cycles.  A body in which a pixel stops (wave-uniform
// scalar branch) applies the selects and clears the pixel's alive bit.  Same operations on the same operands for every
// pixel that is still alive, no update of the others: image, final_Ts, final_index bit for bit.
// MEASURED (round 6, profiles/r06h_fwd_alive_mask.txt): bit for bit, and NO gain - raster_fwd 298 / 302 us against 295 / 300
// (config 3), 311 against 307 (RGB + depth), 628 against 619 (config 5): the 8 pipe cycles a body loses are paid back by
// two more scalar instructions and a v_cmp -> s_and -> select dependency in every body's chain.  Off; kept as a knob.
#ifndef TS_FWD_ALIVE
#define TS_FWD_ALIVE 0
#endif
template <int CH, bool LOC>
__device__ __forceinline__ void fwd_update_alive(float a, mask64 m1, float& T, mask64& alive, int& fidx, int idx,
                                                 float (&acc)[CH], const float (&col)[CH], float (&loc)[CH]) {
#pragma clang fp contract(off)
    float ae;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(ae) : "v"(a), "s"(m1));
    float nT = __builtin_fmaf(-ae, T, T);
    float d = T - nT;
    const mask64 go = TS_BALLOT(!(nT <= ts::kTEps));
    if (__builtin_expect(~go != 0ull, 0)) {
        // a pixel of this block stops here (once per pixel): d, nT and m1 are corrected IN PLACE (asm with "+v": the
        // common path below must not be re-emitted with its values in fresh registers and copied at the join)
        asm("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(d) : "s"(go));
        asm("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(nT) : "v"(T), "s"(go));
        m1 &= go;
        alive &= go;
    }
    asm volatile("" : "+v"(d), "+v"(nT), "+s"(m1));
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_fmaf(col[c], d, acc[c]);
    if (LOC) {
#pragma unroll
        for (int c = 0; c < CH; ++c) loc[c] = __builtin_fmaf(col[c], d, loc[c]);
    }
    asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(fidx) : "v"(idx), "s"(m1));
    T = nT;
}

template <int CH, bool GENERAL, int NBX, class Loc>
__device__ __forceinline__ void fwd_chunk(const float4* __restrict__ lds, int cnt, const float (&fpx)[NBX],
                                          const float (&fpy)[2], float (&T)[2 * NBX], int (&fidx)[2 * NBX],
                                          float (&acc)[2 * NBX][CH], Loc& loc, mask64 (&alive)[2 * NBX] TS_SEG_PARAM) {
#pragma clang fp contract(off)          // as in bwd_chunk: both instantiations must round alike
    constexpr bool LOC = Loc::on;
    TS_WORK(0, cnt);
    for (int j = 0; j < (TS_ABLATE == 3 ? 0 : cnt); ++j) {
        TS_SEG_T0(tseg_a);
        const float4 r0 = lds[3 * j], r1 = lds[3 * j + 1], r2 = lds[3 * j + 2];
        const int bm = __builtin_amdgcn_readfirstlane(__float_as_int(r2.w));
        const int idx = __float_as_int(r2.z);
        const float neg_lo = -r1.y;
        float col[CH];
        col[0] = r1.z; col[1] = r1.w; col[2] = r2.x;
        if (CH == 4) col[CH - 1] = r2.y;
#if TS_TIMELINE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        TS_SEG_ADD(ts_seg_, 1, tseg_a);
        TS_SEG_T0(tseg_b);
        TS_WORK(1, __popc(bm & ((1 << (2 * NBX)) - 1)));
#pragma unroll
        for (int k = 0; k < 2 * NBX; ++k) {
            if (!(bm & (1 << k))) continue;                           // wave-uniform
            if (TS_ABLATE == 5) { asm volatile("" ::: "memory"); continue; }
            TS_STAT(1, 1);
            // sgl = sigma*log2(e) - log2(opacity), so alpha = exp2(-sgl)
            const float sgl = sigma_l2(r0.z, r0.w, r1.x, neg_lo, r0.x - fpx[k % NBX], r0.y - fpy[k / NBX]);
            if (TS_ABLATE == 8) {          // timing experiment: the exponent's nine instructions once more per body (result unused)
                float dup = sigma_l2(r0.z, r0.w, r1.x, neg_lo, r0.y - fpx[k % NBX], r0.x - fpy[k / NBX]);
                dup = __builtin_amdgcn_exp2f(-dup);
                asm volatile("" ::"v"(dup));
            }
            // Every decision below is ONE compare feeding ONE select, with no scalar mask arithmetic in between:
            // a v_cmp -> s_and / s_xor -> v_cndmask chain costs a wave 36 cycles and a select on a vcc that the
            // scalar unit wrote 19 (tools/micro/lat_bench.hip), a compare -> select pair 11.
            float vis;
            if constexpr (TS_FWD_ALIVE && !Loc::on) {
                // (segment sums - Loc - keep the sign form: the cut tiles of a hybrid launch are a sixth of the frame)
                const float a = __builtin_amdgcn_exp2f(-sgl);
                mask64 m1 = TS_BALLOT(a >= ts::kAlphaMin) & alive[k];
                float ag = a;
                if (GENERAL) {
                    ag = fminf(ts::kAlphaMax, a);
                    m1 &= TS_BALLOT(sgl >= neg_lo);                   // sigma >= 0
                }
                float no_loc_[CH];
                fwd_update_alive<CH, false>(ag, m1, T[k], alive[k], fidx[k], idx, acc[k], col, no_loc_);
                continue;
            }
            if (TS_FWD_ASM && !GENERAL) {
                vis = fwd_alpha_update(sgl, T[k], fidx[k], idx, acc[k][0], acc[k][1], acc[k][2], col[0], col[1], col[2]);
            } else {
            float ae = fwd_alpha(sgl);                                // alpha >= 1/255 ? alpha : 0
            if (GENERAL) {
                ae = fminf(ts::kAlphaMax, ae);
                ae = sgl >= neg_lo ? ae : 0.0f;                       // sigma >= 0
            }
            // A finished pixel (T < 0) needs no test of its own: nT = T (1 - ae) stays negative, `stop` fires and
            // -|T| puts T back; vis is 0.  An unfinished pixel always has T > kTEps (it would have stopped
            // otherwise), so with ae = 0 (Gaussian below 1/255) nT = T and `stop` cannot fire: no `& ok` needed.
            // composited <=> alpha >= 1/255 and not stopped <=> vis = alpha T > 0 (alpha >= 1/255, T > 1e-4)
            vis = fwd_update(ae, T[k], fidx[k], idx, acc[k][0], acc[k][1], acc[k][2], col[0], col[1], col[2]);
            }
            if (CH == 4) acc[k][CH - 1] = __builtin_fmaf(col[CH - 1], vis, acc[k][CH - 1]);
            if (LOC) loc.add(k, col, vis);      // the same contribution summed per list segment (LIST SEGMENTS)
        }
        TS_SEG_ADD(ts_seg_, 2, tseg_b);
    }
}

// COOPERATIVE TILES (bits 16..19 of ts_camera.hints = C16; forward launch, one wave per 16x16 tile on 16x16 lists).
// The forward launch is two rounds of ~130-us waves as well, and its items cannot be list segments (the transmittance is
// a chain) - but a tile's list can be STAGED by four waves and its four 8x8 blocks composited by four waves: the tiles
// that are dispatched LAST (the first C16/16 of every band; whole tiles are handed out from the band's end) become
// workgroups of their own.  Per round of 256 list entries every wave gathers and culls 64 of them against the four
// blocks' rectangles (stage_splat, as a whole-tile wave does per chunk) and compacts its survivors into the shared
// record array; after a barrier wave w walks the staged entries whose mask names block w - found with ONE vector read
// of the mask words and a ballot per source wave, then a scalar loop over the set bits - and runs the same per-pixel
// body on one pixel per lane.  The tile's sort is shared too: four runs sorted in registers, merged by ranking every
// key in the three other runs (binary searches in LDS; keys are unique).  Against a whole-tile wave the walk and the
// sort are divided by four instead of replicated (which is what made TS_RASTER_SPLIT_BLOCKS cost 2.4x per tile), the
// record of an entry is read once per BLOCK it touches instead of once (LDS bandwidth: the reason why this is not
// the mapping of every tile), the rectangles of the unfinished pixels follow per 256 instead of per 64 entries.
// Every pixel sees the same entries in the same order through the same arithmetic: image, final_Ts and final_index
// are bitwise those of the whole-tile waves.
#define TS_CAM_COOP16(cam) (((cam).hints >> 16) & 15)
#ifndef TS_COOP
#define TS_COOP (TS_RASTER_WAVES == 4)
#endif
#ifndef TS_COOP_AHEAD
#define TS_COOP_AHEAD 4
#endif
constexpr int kCoopAhead = TS_COOP_AHEAD;
// TS_SEGS_COOP=1 (measured, not the default): in a launch that keeps boundary state (hybrid launch, S > 1) the CUT tiles
// are the cooperative ones - a cooperative wave has one pixel per lane, so the per-segment colour sums are three
// registers there, where they are twelve (spilled at the five-wave budget) in the whole-tile waves of such a launch;
// C16 is ignored then.  The whole-tile waves lose their spills (96 VGPRs + 92 bytes -> 94), but the boundary records
// and the three extra FMAs per body then sit in the items that END the launch instead of the ones that start it:
// raster_fwd 290 -> 300 us on config 3.
#ifndef TS_SEGS_COOP
#define TS_SEGS_COOP 0
#endif
constexpr bool kSegsCoop = TS_SEGS_COOP && TS_COOP;
// internal (set by ts_raster_fwd* in the camera copy it hands to the kernel, never by callers): the launch asked for
// TS_RASTER_SPLIT_BLOCKS under TS_HINT_COOP_SPLIT - every tile (S > 1: every cut tile) is a cooperative workgroup
constexpr int kHintCoopAll = 1 << 21;
struct FwdPlan {
    int per_xcd;          // tile groups (of kWaves tiles) per band
    int coop, coop_lo;    // groups coop_lo .. coop_lo + coop - 1 of every band: one workgroup per TILE, handed out last
    bool descending;      // groups (and cooperative tiles) are handed out from the band's end: the cut tiles first
    __host__ __device__ int grid() const { return 8 * (per_xcd + (kWaves - 1) * coop); }
};
// segs: the launch keeps boundary state (the SEGS instantiations with S > 1)
__host__ __device__ __forceinline__ FwdPlan fwd_plan(int num_tiles, int hints, bool segs, bool coop_ok) {
    FwdPlan p;
    const int groups = (num_tiles + kWaves - 1) / kWaves;
    p.per_xcd = (groups + 7) >> 3;
    const int w16 = (hints >> 12) & 15, c16 = (hints >> 16) & 15;
    const CutTiles m = cut_tiles(num_tiles, hints);           // (band = kWaves * per_xcd tiles for kWaves == 4)
    if (TS_COOP && coop_ok && (hints & kHintCoopAll) && !segs) {      // a small launch, no boundary state: every tile
        p.coop = p.per_xcd;
        p.coop_lo = 0;
        p.descending = false;
        return p;
    }
    if (segs && coop_ok && (kSegsCoop || (TS_COOP && (hints & kHintCoopAll)))) {
        p.coop = (m.band - m.whole) / kWaves;                 // W16 = 0: every tile is cut
        p.coop_lo = m.whole / kWaves;
        p.descending = false;
        return p;
    }
    p.coop = (TS_COOP && coop_ok) ? (p.per_xcd * c16) / 16 : 0;
    if (segs) p.coop = w16 > 0 ? min(p.coop, m.whole / kWaves) : 0;     // a cut tile keeps its boundary state: whole-tile wave
    p.coop_lo = 0;
    p.descending = (segs && TS_FWD_CUT_FIRST && w16 > 0) || p.coop > 0;
    return p;
}

// the per-pixel body of fwd_chunk on ONE pixel per lane (same operations in the same order: identical bits)
template <int CH, bool GENERAL, bool LOC>
__device__ __forceinline__ void fwd_body1(const float4 r0, const float4 r1, const float4 r2, float fpx, float fpy,
                                          float& T, int& fidx, float (&acc)[CH], float (&loc)[CH]) {
#pragma clang fp contract(off)
    const int idx = __float_as_int(r2.z);
    const float neg_lo = -r1.y;
    float col[CH];
    col[0] = r1.z; col[1] = r1.w; col[2] = r2.x;
    if (CH == 4) col[CH - 1] = r2.y;
    const float sgl = sigma_l2(r0.z, r0.w, r1.x, neg_lo, r0.x - fpx, r0.y - fpy);
    float vis;
    if (TS_FWD_ASM && !GENERAL) {
        vis = fwd_alpha_update(sgl, T, fidx, idx, acc[0], acc[1], acc[2], col[0], col[1], col[2]);
    } else {
        float ae = fwd_alpha(sgl);
        if (GENERAL) {
            ae = fminf(ts::kAlphaMax, ae);
            ae = sgl >= neg_lo ? ae : 0.0f;
        }
        vis = fwd_update(ae, T, fidx, idx, acc[0], acc[1], acc[2], col[0], col[1], col[2]);
    }
    if (CH == 4) acc[CH - 1] = __builtin_fmaf(col[CH - 1], vis, acc[CH - 1]);
    if (LOC) {          // the same contribution summed per list segment (LIST SEGMENTS)
#pragma unroll
        for (int c = 0; c < CH; ++c) loc[c] = __builtin_fmaf(col[c], vis, loc[c]);
    }
}

// one sorted run of a cooperative tile's list: n_run <= 64 E keys of g[] sorted in registers, position lane * E + e
template <int E>
__device__ __forceinline__ void coop_sort_run(const int* __restrict__ g, const float* __restrict__ depths, int n_run,
                                              int lane, unsigned long long (&k)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * 64 + lane;
        k[e] = i < n_run ? make_key(depths, g[i]) : ~0ull;
    }
    // (key i starts at network position (i % 64) * E + i / 64: with E > 1 the whole network must run, whatever n_run -
    // a run can be shorter than 32 E + 1 keys, which sort_tile_wave's choice of E rules out)
    bitonic_regs<E, 64>(k, lane, E == 1 ? min(pow2_at_least(max(n_run, 1)), 64) : 64 * E, nullptr);
}
// ... and its merge with the three other runs: rank of a key = its position in its own run + the keys below it in the
// others.  keys: [4][256] in LDS, run v holds nv(v) keys (the rest is padding that is never ranked).
template <int E>
__device__ __forceinline__ void coop_sort_tile(const int* __restrict__ g, const float* __restrict__ depths,
                                               int* __restrict__ out, int* ids_l, int n, int wave, int lane,
                                               unsigned long long* keys) {
    const int q = (n + 3) >> 2;                              // run length (the last runs may be shorter or empty)
    const int n_own = max(0, min(q, n - wave * q));
    unsigned long long k[E];
    coop_sort_run<E>(g + wave * q, depths, n_own, lane, k);
#pragma unroll
    for (int e = 0; e < E; ++e) keys[wave * 256 + lane * E + e] = k[e];
    TS_LDS_BARRIER();
    // branch-free lower bounds with power-of-two steps, the 3 E searches of a lane in lockstep: every step has all its
    // LDS reads in flight together (one after the other they were ~50 dependent round trips per lane)
    int lo[E][3], len[3];
#pragma unroll
    for (int dv = 0; dv < 3; ++dv) len[dv] = max(0, min(q, n - ((wave + dv + 1) & 3) * q));
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
        for (int dv = 0; dv < 3; ++dv) lo[e][dv] = 0;
#pragma unroll
    for (int step = 64 * E; step >= 1; step >>= 1) {             // runs hold at most 64 E keys (a key may lie above all of them)
#pragma unroll
        for (int e = 0; e < E; ++e) {
#pragma unroll
            for (int dv = 0; dv < 3; ++dv) {
                const unsigned long long* run = keys + ((wave + dv + 1) & 3) * 256;
                const int p = lo[e][dv] + step;
                // (p - 1 <= 64 E - 1 lies inside the run's 256 slots; beyond its length the slot holds padding or
                // stale keys, which the length test rules out)
                if (p <= len[dv] && run[p - 1] < k[e]) lo[e][dv] = p;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int p = lane * E + e;
        if (p < n_own) {
            const int rank = p + lo[e][0] + lo[e][1] + lo[e][2];
            out[rank] = (int)(unsigned int)k[e];            // for the backward pass
            ids_l[rank] = (int)(unsigned int)k[e];          // for this workgroup's own walk
        }
    }
}

// One 16x16 tile composited by the four waves of a workgroup (see COOPERATIVE TILES).  rec: 256 x 3 float4 staged
// records (the whole-tile waves' lds_all), also the key array of the shared sort; rect_sh[4]: rectangle of the
// unfinished pixels of block k; cnt_sh[4]: entries wave s staged in this round; alive_sh[4]: block k has unfinished pixels.
template <int CH, bool SORT, bool SEGS>
__device__ __forceinline__ void coop_fwd_tile(
    const ts_camera& cam, int tile, int num_tiles, const int* __restrict__ tile_bins, const int* __restrict__ ids_sorted,
    const int* __restrict__ bucket_ids, const float* __restrict__ depths, int* ids_rw,
    const float4* __restrict__ splats, const float* __restrict__ background, float* __restrict__ out_img,
    float* __restrict__ out_depth, float* __restrict__ final_Ts, int* __restrict__ final_index, int clamp_rgb,
    unsigned char* __restrict__ clamp_mask, float4* rec, float4* rect_sh, int* cnt_sh, int* alive_sh, int clock_unit) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tbx = cam.tile_bounds_x;
    const int tx = tile % tbx, ty = tile / tbx + cam.tile_row0;
    const int px = tx * 16 + 8 * (wave & 1) + (lane & 7), py = ty * 16 + 8 * (wave >> 1) + (lane >> 3);
    const float fpx = (float)px + ts::kPixOff, fpy = (float)py + ts::kPixOff;
    const float BX0 = (float)(tx * 16 + 8 * (wave & 1)) + ts::kPixOff, BY0 = (float)(ty * 16 + 8 * (wave >> 1)) + ts::kPixOff;
    const int W = cam.img_width, H = cam.img_height;
    const bool inside = px < W && py < H;
    float T = inside ? 1.0f : -1.0f, acc[CH];
    int fidx = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = 0.0f;
    const int2 range = reinterpret_cast<const int2*>(tile_bins)[tile];
    const int n = range.y - range.x;
    TS_WAVE_CLOCK(0, clock_unit + wave, n);          // (timeline builds: the cooperative waves follow the tiles' rows)
    (void)clock_unit;

    // list segments: a cut tile's boundary records (CHECKPOINTS), block `wave` of every record by this wave
    const int S_seg = max(1, min(TS_CAM_SEGS(cam), kSegMax));
    int ck_block = -1;
    if (SEGS && final_Ts != nullptr && S_seg > 1 && n >= kSegMinList) ck_block = cut_tiles(num_tiles, cam.hints).block_of(tile);
    const bool seg_on = SEGS && ck_block >= 0;
    float* ckp = nullptr;
    if (seg_on) ckp = final_Ts + ckpt_offset(seg_plane_stride(cam)) + ckpt_record<CH>(ck_block, S_seg, 1);
    int next_ck = seg_on ? range.x + seg_bound(n, S_seg, 1) : 0x7fffffff;       // list index of the next boundary
    int ck = 1;                                                                 // its number (1 .. S-1)
    float loc[CH];                              // colour the entries of the CURRENT segment contributed to this pixel
#pragma unroll
    for (int c = 0; c < CH; ++c) loc[c] = 0.0f;
    auto store_ck = [&](int number) {
        ckpt_store<CH>(ckp + (number - 1) * ((1 + CH) * 256), wave, lane, __builtin_fabsf(T), loc);
#pragma unroll
        for (int c = 0; c < CH; ++c) loc[c] = 0.0f;
    };

    // rectangle of this wave's unfinished pixels -> LDS (read by the four staging waves after the next barrier)
    auto publish = [&]() {
        const unsigned long long m = __ballot(T > 0.0f);
        int xmin = 0, xmax = 0, ymin = 0, ymax = 0;
        const bool any = mask_rect(m, xmin, xmax, ymin, ymax);
        if (lane == 0) {
            alive_sh[wave] = any ? 1 : 0;
            if (any) rect_sh[wave] = make_float4(BX0 + (float)xmin, BX0 + (float)xmax, BY0 + (float)ymin, BY0 + (float)ymax);
        }
    };
    publish();

    int idr[4] = {0, 0, 0, 0};               // shared sort: this lane's ids of rounds 0 .. 3
    bool from_lds = false;
    TS_SEG_T0(tseg_s);
    if (SORT && TS_ABLATE == 6 && n <= kWaveSortMax) {       // timing experiment: a copy instead of the sort
        for (int i = threadIdx.x; i < n; i += 256) ids_rw[range.x + i] = bucket_ids[range.x + i];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __syncthreads();
    } else
    if (SORT && n > 0 && n <= kWaveSortMax) {
        const int* g = bucket_ids + range.x;
        int* out = ids_rw + range.x;
        if (n <= 128) {                       // short list: one wave's network, the others wait
            if (wave == 0) {
                if (n <= 64) sort_tile_wave<1>(g, depths, out, n, lane);
                else sort_tile_wave<2>(g, depths, out, n, lane);
            }
        } else {
            unsigned long long* keys = reinterpret_cast<unsigned long long*>(rec);
            int* ids_l = reinterpret_cast<int*>(rec) + 2048;          // 8 KiB of keys, then 4 KiB of sorted ids
            if (n <= 256) coop_sort_tile<1>(g, depths, out, ids_l, n, wave, lane, keys);
            else if (n <= 512) coop_sort_tile<2>(g, depths, out, ids_l, n, wave, lane, keys);
            else coop_sort_tile<4>(g, depths, out, ids_l, n, wave, lane, keys);
        }
        if (n <= 128) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the list is read back by other waves
            __syncthreads();
        } else {
            // the shared sort left the sorted ids in LDS as well (behind the keys): every lane takes the ids of its
            // entries of all (<= 4) rounds from there - no wait for the global stores, no round trip through memory
            TS_LDS_BARRIER();
            const int* ids_l = reinterpret_cast<const int*>(rec) + 2048;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 256 * r + 64 * wave + lane;
                if (i < n) idr[r] = ids_l[i];
            }
            from_lds = true;
        }
    }
    TS_SEG_ADD(ts_wave_clock_.seg, 3, tseg_s);
    const int* ids = SORT ? ids_rw : ids_sorted;

    // software pipeline over rounds of 256 entries: this wave stages entries base + 64 wave + lane
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 n0 = zero4, n1 = zero4, n2 = zero4;
    int id_next = 0;
    {
        const int i0 = range.x + 64 * wave + lane;
        if (i0 < range.y) {
            const int g = from_lds ? idr[0] : ids[i0];
            n0 = splats[3 * (size_t)g]; n1 = splats[3 * (size_t)g + 1]; n2 = splats[3 * (size_t)g + 2];
        }
        if (i0 + 256 < range.y) id_next = from_lds ? idr[1] : ids[i0 + 256];
    }
    int round = 0;
    for (int base = range.x; base < range.y; base += 256) {
        TS_SEG_T0(tseg_w0);
        TS_LDS_BARRIER();                     // rectangles / alive flags of the last round are in LDS, rec[] is free
        TS_SEG_ADD(ts_wave_clock_.seg, 1, tseg_w0);
        TS_SEG_T0(tseg_p);
        const int live = (alive_sh[0] ? 1 : 0) | (alive_sh[1] ? 2 : 0) | (alive_sh[2] ? 4 : 0) | (alive_sh[3] ? 8 : 0);
        if (live == 0) break;                 // (the same value in all four waves)
        const int i = base + 64 * wave + lane;
        const bool have = i < range.y;
        const float4 q0 = n0, q1 = n1, q2 = n2;
        if (i + 256 < range.y) {
            const int g = id_next;
            n0 = splats[3 * (size_t)g]; n1 = splats[3 * (size_t)g + 1]; n2 = splats[3 * (size_t)g + 2];
        }
        if (i + 512 < range.y) id_next = from_lds ? (round == 0 ? idr[2] : idr[3]) : ids[i + 512];
        ++round;
        Staged s = stage_splat<4>(have, q0, q1, rect_sh, live);
        const bool keep = (s.mask & 15) != 0;
        const unsigned long long kmask = __ballot(keep);
        if (keep) {
            const int pos = 64 * wave + __popcll(kmask & ((1ull << lane) - 1ull));
            rec[3 * pos] = make_float4(s.gx, s.gy, s.hA, s.B);
            rec[3 * pos + 1] = make_float4(s.hC, s.lo, q1.z, q1.w);
            rec[3 * pos + 2] = make_float4(q2.x, q2.y, __int_as_float(i), __int_as_float(s.mask));
        }
        if (lane == 0) cnt_sh[wave] = __popcll(kmask);
        TS_SEG_ADD(ts_wave_clock_.seg, 0, tseg_p);
        TS_SEG_T0(tseg_w1);
        TS_LDS_BARRIER();
        TS_SEG_ADD(ts_wave_clock_.seg, 1, tseg_w1);
        TS_SEG_T0(tseg_b);
        // block `wave`: the staged entries of the four source waves in list order
#pragma unroll 1
        for (int src = 0; src < 4; ++src) {
            if (SEGS && base + 64 * src == next_ck) {     // a segment boundary (never the list's first entry)
                store_ck(ck);
                ++ck;
                next_ck = ck < S_seg ? range.x + seg_bound(n, S_seg, ck) : 0x7fffffff;
            }
            const int c = cnt_sh[src];
            const float4* rs = rec + 3 * 64 * src;
            const int m = lane < c ? __float_as_int(rs[3 * lane + 2].w) : 0;
            unsigned long long bits = __ballot(((m >> wave) & 1) != 0);
            if (bits == 0ull) continue;
            const bool general = __ballot(((m >> wave) & 1) != 0 && (m & 16) != 0) != 0ull;
            // one loop per variant (the choice is made once per source chunk, as a whole-tile wave makes it per chunk)
            auto run = [&](auto gen_tag, auto loc_tag) {
                constexpr bool G = decltype(gen_tag)::value, L = decltype(loc_tag)::value;
                // the records of up to kCoopAhead entries are requested together: one LDS round trip per group, not
                // one per body (a wave has ONE pixel per lane here: nothing else of its own to overlap the wait with)
                while (bits != 0ull) {
                    float4 r0[kCoopAhead], r1[kCoopAhead], r2[kCoopAhead];
                    int got = 0;
#pragma unroll
                    for (int u = 0; u < kCoopAhead; ++u) {
                        if (bits != 0ull) {
                            const int j = __builtin_ctzll(bits);
                            bits &= bits - 1ull;
                            r0[u] = rs[3 * j]; r1[u] = rs[3 * j + 1]; r2[u] = rs[3 * j + 2];
                            got = u + 1;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kCoopAhead; ++u)
                        if (u < got) fwd_body1<CH, G, L>(r0[u], r1[u], r2[u], fpx, fpy, T, fidx, acc, loc);
                }
            };
            if (SEGS && seg_on) {
                if (general) run(std::true_type{}, std::true_type{});
                else run(std::false_type{}, std::true_type{});
            } else {
                if (general) run(std::true_type{}, std::false_type{});
                else run(std::false_type{}, std::false_type{});
            }
        }
        publish();
        TS_SEG_ADD(ts_wave_clock_.seg, 2, tseg_b);
    }

    if (SEGS && seg_on) {
        // the pass ended in segment ck - 1: its colour goes into record ck; the records behind hold no colour (see the
        // whole-tile waves of the split launches)
        store_ck(ck);
#pragma unroll 1
        for (int r = ck + 1; r <= S_seg; ++r) store_ck(r);
    }
    if (!inside) return;
    float bg[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) bg[c] = background[c];
    const size_t pix = (size_t)(py - cam.tile_row0 * 16) * W + px;
    const float Tf = __builtin_fabsf(T);
    if (final_Ts) {
        final_Ts[pix] = Tf;
        final_index[pix] = fidx;
    }
    const bool planes = CH == 4 && out_depth != nullptr;
    float* o = out_img + pix * (planes ? 3 : CH);
    int pass = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        float val = acc[c] + Tf * bg[c];
        if (clamp_rgb && c < 3) {
            pass |= (val <= 1.0f) ? (1 << c) : 0;
            val = fminf(val, 1.0f);
        }
        if (c == 3 && planes) out_depth[pix] = val;
        else o[c] = val;
    }
    if (clamp_mask) clamp_mask[pix] = (unsigned char)pass;
}

// Pixel layout of a wave: lane l -> (lx, ly) = (l & 7, l >> 3) inside an 8x8 block; the lane owns
// that position in each of the four blocks k of the 16x16 tile.
// SPLIT: four waves per tile, each owning ONE 8x8 block (the other three count as outside the image).
// Same list, same arithmetic per pixel; used when a launch has fewer tiles than the GPU has SIMDs (a
// tile-row stripe of a multi-GPU frame, a small image), where one wave per tile leaves the vector ALUs
// without a second wave to switch to and the launch takes as long as the longest tile list.
// WL (NBX == 2 only): the lists are those of wide tiles (ts_camera.wide_tiles) but every 16x16 tile
// keeps a wave of its own, which walks the list of the wide tile it belongs to and drops the Gaussians
// whose tile box does not contain it - binning and sorting at the wide tiles' price, compositing at the
// register footprint of four pixels per lane.
// SORT (one wave per 16x16 tile on 16x16 lists only): the wave first sorts its tile's list itself - bucket_ids ->
// ids_rw, the network of sort_tiles_small_kernel - when the list has at most kWaveSortMax entries (longer lists were
// sorted by ts_sort_tiles_above before this launch).  The per-tile sort on its own is latency-bound (VALU 40 %, LDS
// 59 % busy on config 3); inside this VALU-bound kernel its stalls are filled by other tiles' compositing.
template <int CH, bool SPLIT, int NBX, bool WL, bool SORT = false, bool SEGS = false>
__global__ __launch_bounds__(kThreads, (NBX == 2 && !SPLIT) ? (CH == 3 ? TS_FWD_MIN_WAVES_RGB : TS_FWD_MIN_WAVES_RGBD) : TS_FWD_MIN_WAVES) void raster_fwd_kernel(
    const ts_camera cam, const int num_tiles, const int* __restrict__ tile_bins,
    const int* __restrict__ ids_sorted, const int* __restrict__ bucket_ids, const float* __restrict__ depths,
    int* ids_rw, const float4* __restrict__ splats,
    const float* __restrict__ background, float* __restrict__ out_img, float* __restrict__ out_depth,
    float* __restrict__ final_Ts, int* __restrict__ final_index, const int clamp_rgb,
    unsigned char* __restrict__ clamp_mask) {
    constexpr int NB = 2 * NBX;
    TS_LDS_PAD_DECL();
    __shared__ float4 lds_all[kWaves][64 * 3];
    __shared__ float4 rect_all[kWaves][NB];
    __shared__ float4 raw_all[TS_LDS_DMA ? kWaves : 1][3 * 64];      // landing zone of the next chunk's records
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int units = SPLIT ? NB * num_tiles : num_tiles;
    int unit = xcd_tile_group((units + kWaves - 1) / kWaves) * kWaves + wave;
    constexpr bool kCoop = TS_COOP && !SPLIT && NBX == 2 && !WL;
    if constexpr (!SPLIT && NBX == 2 && !WL) {
        // hybrid launch: the cut tiles of a band - the ones that also keep their boundary state, the longest items of
        // this launch - are handed out FIRST here (the backward launch hands them out last, as small items); the
        // tiles handed out last - C16 / 16 of a band from its start - are composited by a workgroup each (COOPERATIVE
        // TILES; TS_SEGS_COOP=1: the cut tiles themselves)
        const FwdPlan pl = fwd_plan(num_tiles, cam.hints, SEGS && final_Ts != nullptr && TS_CAM_SEGS(cam) > 1, kCoop);
        const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
        if (slot < pl.per_xcd - pl.coop) {
            if (pl.descending) unit = (xcd * pl.per_xcd + (pl.per_xcd - 1 - slot)) * kWaves + wave;
        } else if constexpr (kCoop) {
            const int t = slot - (pl.per_xcd - pl.coop);                 // 0 .. kWaves * coop - 1
            const int ctile = (xcd * pl.per_xcd + pl.coop_lo) * kWaves + (pl.descending ? kWaves * pl.coop - 1 - t : t);
            if (ctile >= num_tiles) return;
            __shared__ int coop_cnt[4], coop_alive[4];
            coop_fwd_tile<CH, SORT, SEGS>(cam, ctile, num_tiles, tile_bins, ids_sorted, bucket_ids, depths, ids_rw, splats,
                                          background, out_img, out_depth, final_Ts, final_index, clamp_rgb, clamp_mask,
                                          &lds_all[0][0], &rect_all[0][0], coop_cnt, coop_alive,
                                          num_tiles + kWaves * (xcd * kWaves * pl.coop + t));
            return;
        } else {
            return;
        }
    }
    if (unit >= units) return;
    const int tile = SPLIT ? unit / NB : unit;
    const int only = SPLIT ? unit % NB : -1;
    float4* lds = lds_all[wave];
    float4* rects = rect_all[wave];
    const int tbx = NBX == 2 ? cam.tile_bounds_x : (cam.tile_bounds_x + 1) >> 1;     // tiles of this shape per row
    const int tx = tile % tbx, ty = tile / tbx + cam.tile_row0;
    const int px0 = tx * (8 * NBX) + (lane & 7), py0 = ty * 16 + (lane >> 3);
    // sample positions of the lane's pixel in the block columns / the upper and lower block row
    float fpx[NBX];
#pragma unroll
    for (int c = 0; c < NBX; ++c) fpx[c] = (float)(px0 + 8 * c) + ts::kPixOff;
    const float fpy[2] = {(float)py0 + ts::kPixOff, (float)(py0 + 8) + ts::kPixOff};
    const float X0 = (float)(tx * (8 * NBX)) + ts::kPixOff, Y0 = (float)(ty * 16) + ts::kPixOff;
    const int W = cam.img_width, H = cam.img_height;

    // T > 0: transmittance of an unfinished pixel; T < 0: finished or outside, final value |T|
    float T[NB], acc[NB][CH];
    mask64 alive[NB];
    int fidx[NB];
    bool inside[NB];
    int live = 0;                                   // blocks that still have unfinished pixels
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        inside[k] = (px0 + 8 * (k % NBX) < W) && (py0 + 8 * (k / NBX) < H) && (!SPLIT || k == only);
        T[k] = inside[k] ? 1.0f : -1.0f;
        alive[k] = TS_BALLOT(inside[k]);              // (TS_FWD_ALIVE: the unfinished pixels of block k)
        fidx[k] = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[k][c] = 0.0f;
        if (__any(inside[k])) live |= (1 << k);
    }

    const int list = WL ? (ty - cam.tile_row0) * ((tbx + 1) >> 1) + (tx >> 1) : tile;
    const int2 range = reinterpret_cast<const int2*>(tile_bins)[list];
    TS_WAVE_CLOCK(0, unit, range.y - range.x);

    if (SORT) {
        // one list per wave, or (SPLIT) per workgroup: its four waves composite the four 8x8 blocks of ONE tile,
        // wave 0 sorts the list and the others wait at the barrier
        static_assert(!SORT || (!WL && NBX == 2), "16x16 lists");
        TS_SEG_T0(tseg_s);                // (timeline builds: the sort is segment 3 of the forward kernel)
        const int n = range.y - range.x;
        if (TS_ABLATE == 6 && n <= kWaveSortMax && (!SPLIT || wave == 0)) {      // timing experiment: a copy instead of the sort
            for (int i = lane; i < n; i += 64) ids_rw[range.x + i] = bucket_ids[range.x + i];
        } else
        if (n > 0 && n <= kWaveSortMax && (!SPLIT || wave == 0)) {
            const int* g = bucket_ids + range.x;
            int* out = ids_rw + range.x;
            if (n <= 64) sort_tile_wave<1>(g, depths, out, n, lane);
            else if (n <= 128) sort_tile_wave<2>(g, depths, out, n, lane);
            else if (n <= 256) sort_tile_wave<4>(g, depths, out, n, lane);
            else if (n <= 512) sort_tile_wave<8>(g, depths, out, n, lane);
            else sort_tile_wave<16>(g, depths, out, n, lane);
        }
        // the list is read back below by other lanes (SPLIT: other waves): stores done before the loads are issued
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (SPLIT) __syncthreads();       // units = 4 * tiles: the four waves of a workgroup are valid together
        TS_SEG_ADD(ts_wave_clock_.seg, 3, tseg_s);
    }
    // (SORT: reads go through the pointer the sort wrote through; ids_sorted is the same buffer)
    const int* ids = SORT ? ids_rw : ids_sorted;

    // Software pipeline over 64-entry chunks: the id of chunk c+2 and the packed record of chunk
    // c+1 are in flight while chunk c is composited (two dependent gathers = ~2 us of latency
    // that a wave with ~3 co-resident waves per SIMD cannot hide otherwise).
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* raw = raw_all[TS_LDS_DMA ? wave : 0];
#if !TS_LDS_DMA
    float4 n0 = zero4, n1 = zero4, n2 = zero4;
#endif
    int id_next = 0;
    if (range.x + lane < range.y) {
        const int g = ids[range.x + lane];
#if TS_LDS_DMA
        dma_record(splats, g, raw);
#else
        n0 = splats[3 * (size_t)g]; n1 = splats[3 * (size_t)g + 1]; n2 = splats[3 * (size_t)g + 2];
#endif
    }
    if (range.x + 64 + lane < range.y) id_next = ids[range.x + 64 + lane];

    // list segments: this pass leaves the per-pixel state at the segment boundaries for the backward pass - in a SPLIT
    // launch for every tile (each of the four waves keeps the pixels of its block), with one wave per tile for the CUT
    // tiles of a hybrid launch (or, W16 = 0, for all of them)
    // (one wave per tile, TS_SEGS_COOP: the cut tiles are cooperative workgroups - no whole-tile wave keeps boundary state)
    constexpr bool kSegsF = SEGS && NBX == 2 && !WL && (SPLIT || !kSegsCoop);
    const int S_seg = max(1, min(TS_CAM_SEGS(cam), kSegMax));
    int ck_block = -1;                                               // the tile's checkpoint block, -1 = none
    if (kSegsF && final_Ts != nullptr && S_seg > 1 && range.y - range.x >= kSegMinList)
        ck_block = cut_tiles(num_tiles, cam.hints).block_of(tile);
    const bool seg_on = ck_block >= 0;
    float* ckp = nullptr;                                            // record 1 of the tile's checkpoint block (see CHECKPOINTS)
    if (kSegsF && seg_on) ckp = final_Ts + ckpt_offset(seg_plane_stride(cam)) + ckpt_record<CH>(ck_block, S_seg, 1);
    int next_ck = seg_on ? range.x + seg_bound(range.y - range.x, S_seg, 1) : 0x7fffffff;      // list index of the next boundary
    int ck = 1;                                                      // its number (1 .. S-1)
    // colour the entries of the CURRENT segment contributed (the running sum `acc` rounds at the magnitude of the whole
    // pixel colour: differences of it would know what lies behind a boundary only to that rounding)
    constexpr bool kLocLds = kSegsF && !SPLIT && TS_LOC_LDS;
    using LocT = std::conditional_t<kLocLds, LocLds<CH, NB>, LocRegs<CH, !kSegsF ? 0 : (SPLIT ? 1 : NB)>>;
    LocT loc;
    if constexpr (kLocLds) {
        __shared__ float4 loc_all[kWaves][kLocLds ? NB * 64 : 1];
        loc.p = loc_all[wave] + lane;
        if (seg_on) loc.zero();
    } else {
        loc.zero();
    }
    LocRegs<CH, 0> no_loc;
    // boundary `number` (1 .. S-1) is reached: record `number` = T in front of it and the colour of the segment that ends
    auto store_ck = [&](int number) {
        float* rec = ckp + (number - 1) * ((1 + CH) * 256);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            if (SPLIT && k != only) continue;
            float D[CH];
            loc.get(k, D);
            ckpt_store<CH>(rec, k, lane, __builtin_fabsf(T[k]), D);
        }
        loc.zero();
    };

    // the walk over the list, with (LOCP) or without the per-segment colour sums - two instantiations, so that the
    // whole tiles of a hybrid launch run the loop they always ran
    auto walk = [&](auto& loc_) {
    constexpr bool LOCP = std::remove_reference_t<decltype(loc_)>::on;
    for (int base = range.x; base < range.y && live != 0; base += 64) {
        TS_SEG_T0(tseg_p);
        const int i = base + lane;
        const bool have = i < range.y;
#if TS_LDS_DMA
        TS_DMA_WAIT();                               // this chunk's records have landed (and id_next has arrived)
        const float4 q0 = have ? raw[lane] : zero4, q1 = have ? raw[64 + lane] : zero4,
                     q2 = have ? raw[128 + lane] : zero4;
        TS_LDS_WAIT();                               // read out before the next chunk's records may overwrite them
        if (i + 64 < range.y) dma_record(splats, id_next, raw);
#else
        // (LOCP: the per-segment sums need twelve registers more; there the next chunk's records are requested AFTER
        // this chunk's have been staged, so that the two sets are never live together - the chunk's bodies still
        // cover the round trip)
        constexpr bool kLate = LOCP && TS_FWD_LATE_PREFETCH;
        const float4 q0 = n0, q1 = n1, q2 = n2;
        auto prefetch = [&]() {
            if (i + 64 < range.y) {
                const int g = id_next;
                n0 = splats[3 * (size_t)g]; n1 = splats[3 * (size_t)g + 1]; n2 = splats[3 * (size_t)g + 2];
            }
            if (i + 128 < range.y) id_next = ids[i + 128];
        };
        if constexpr (!kLate) prefetch();
#endif
#if TS_LDS_DMA
        if (i + 128 < range.y) id_next = ids[i + 128];
#endif
        if constexpr (LOCP) {
            // a segment boundary (wave-uniform; never the list's first entry).  The record's stores are issued BEHIND the
            // loads of the next chunk's records: the wait for those loads then leaves the stores in flight
            if (base == next_ck) {
                store_ck(ck);
                ++ck;
                next_ck = ck < S_seg ? range.x + seg_bound(range.y - range.x, S_seg, ck) : 0x7fffffff;
            }
        }
        {   // rectangle of the still-unfinished pixels of each block: saturated pixels need no more
            // Gaussians, so late in the list most (Gaussian, block) pairs are culled here
            bool sel[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) sel[k] = (TS_FWD_ALIVE && !LOCP) ? TS_LANE(alive[k]) : T[k] > 0.0f;
            live = update_rects<NBX>(sel, X0, Y0, rects, lane);
            TS_WAVE_SYNC();
            if (live == 0) break;
        }
        Staged s = stage_splat<NB>(have, q0, q1, rects, live);
        s.mask &= bbox_blocks<NBX, WL>(__float_as_int(q2.w), tx);
        const bool keep = (s.mask & ((1 << NB) - 1)) != 0;
        const unsigned long long mask = __ballot(keep);
        const int cnt = __popcll(mask);
        if (keep) {
            const int pos = __popcll(mask & ((1ull << lane) - 1ull));
            lds[3 * pos] = make_float4(s.gx, s.gy, s.hA, s.B);
            lds[3 * pos + 1] = make_float4(s.hC, s.lo, q1.z, q1.w);
            lds[3 * pos + 2] = make_float4(q2.x, q2.y, __int_as_float(i), __int_as_float(s.mask));
        }
        TS_WAVE_SYNC();
        TS_STAT(0, cnt);
        const bool general = __ballot(keep && (s.mask & (1 << NB))) != 0ull;
#if !TS_LDS_DMA
        if constexpr (kLate) prefetch();
#endif
        // bit NB of a staged mask = that Gaussian needs the general per-pixel code; the choice is made
        // once per chunk so that the common case runs a loop without those tests
        TS_SEG_ADD(ts_wave_clock_.seg, 0, tseg_p);
        if (general)
            fwd_chunk<CH, true, NBX>(lds, cnt, fpx, fpy, T, fidx, acc, loc_, alive TS_SEG_ARG);
        else
            fwd_chunk<CH, false, NBX>(lds, cnt, fpx, fpy, T, fidx, acc, loc_, alive TS_SEG_ARG);
        TS_WAVE_SYNC();
    }
    };
    if constexpr (!kSegsF) walk(no_loc);
    else if constexpr (SPLIT) walk(loc);
    else if (seg_on) walk(loc);
    else walk(no_loc);

    if constexpr (kSegsF) if (seg_on) {
        // the pass ended in segment ck - 1: its colour goes into record ck; the records behind hold no colour (the
        // wave's pixels are finished there - no segment replays them, fidx lies in front - or the list is shorter),
        // and their T only has to be a finite number
        store_ck(ck);
#pragma unroll 1
        for (int r = ck + 1; r <= S_seg; ++r) store_ck(r);          // (the sums are zero by now)
    }

    float bg[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) bg[c] = background[c];
    const int row_off = cam.tile_row0 * 16;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        if (!inside[k]) continue;
        const size_t pix = (size_t)(py0 + 8 * (k / NBX) - row_off) * W + (px0 + 8 * (k % NBX));
        const float Tf = __builtin_fabsf(T[k]);
        if (final_Ts) {             // null in the forward-only (viewer) mode: nothing is kept for backward
            final_Ts[pix] = Tf;
            final_index[pix] = fidx[k];
        }
        // out_depth (CH == 4 only): the image as two planes - 3 floats per pixel in out_img, channel 3 apart
        const bool planes = CH == 4 && out_depth != nullptr;
        float* o = out_img + pix * (planes ? 3 : CH);
        int pass = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float val = acc[k][c] + Tf * bg[c];
            if (clamp_rgb && c < 3) {          // the adapter's clamp(rgb, max=1), rasterize.py:45
                pass |= (val <= 1.0f) ? (1 << c) : 0;          // torch's rule: gradient passes at x <= max
                val = fminf(val, 1.0f);
            }
            if (c == 3 && planes) out_depth[pix] = val;
            else o[c] = val;
        }
        if (clamp_mask) clamp_mask[pix] = (unsigned char)pass;
    }
}

// Butterfly merge of two per-lane partial vectors: lanes whose `bit` is clear keep a, the others
// keep b, and each adds the kept quantity of its partner lane (partner given by the DPP control).
typedef float f2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float merge2(float a, float b, bool bit) {
    const float keep = bit ? b : a, send = bit ? a : b;
    return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), CTRL, 0xF, 0xF, true));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_t(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}

// Reduces ten per-lane values over the wave in one merged butterfly.  On return lane l holds the wave
// sum of value (l & 7) if bit 3 of l is clear, of value 8 + (l & 1) otherwise.
//   level 1  partner l ^ 1 (quad_perm):  5 merges, lanes with bit 0 keep the odd value of each pair
//   level 2  partner l ^ 2 (quad_perm):  2 merges (bit 1) + 1 plain add for the {8,9} pair
//   level 3  row_ror:4:                  1 merge (bit 2) + 1 plain add
//   level 4  row_ror:8:                  1 merge (bit 3): values 0..7 | values 8,9  -> row-of-16 totals
//   rows are combined lane-wise through the LDS crossbar (ds_bpermute; the single-lane row_bcast forms
//   cannot be used because lanes of a row hold different values).
// 31 VALU issues + 2 ds_bpermute for 10 values (a plain DPP reduction is 6-8 per value).
template <bool HAVE9>
__device__ __forceinline__ float wave_sum10(const float v[10], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    const float u0 = merge2<0xB1>(v[0], v[1], b0), u1 = merge2<0xB1>(v[2], v[3], b0);
    const float u2 = merge2<0xB1>(v[4], v[5], b0), u3 = merge2<0xB1>(v[6], v[7], b0);
    // without a tenth value the {8,9} "pair" is a plain add (both lanes of a pair then hold value 8)
    float t = HAVE9 ? merge2<0xB1>(v[8], v[9], b0) : dpp_add_t<0xB1, 0xF>(v[8]);
    const float w0 = merge2<0x4E>(u0, u1, b1), w1 = merge2<0x4E>(u2, u3, b1);
    t = dpp_add_t<0x4E, 0xF>(t);
    float x = merge2<0x124>(w0, w1, b2);     // row_ror:4  (source lane differs in bit 2, same bits 1:0)
    t = dpp_add_t<0x124, 0xF>(t);
    x = merge2<0x128>(x, t, b3);             // row_ror:8
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}

// The same reduction with the levels reordered so that the two in-row levels which DPP can write under a
// BANK MASK run first, while there are most values: lane bit 3 (partner lane ^ 8 = row_ror:8, banks 2-3 =
// bank_mask 0xC) and lane bit 2 (row_half_mirror pairs the two quads of a half-row, banks 1,3 = 0xA).  A merge
// of two values is then two masked DPP adds (all lanes take a + perm(a), the lanes of the set bit are
// overwritten with b + perm(b)) instead of two selects and a DPP add.  The compiler cannot be made to emit a
// DPP add that keeps its destination in the masked-off lanes, hence the one asm block; inside it every DPP read
// of a register lies at least two instructions behind the write (the VALU -> DPP hazard), and it begins and ends
// with s_nop 1 because the compiler's hazard recogniser does not look into it.  15 masked adds take the ten
// values to three registers; the cross-row levels and the two quad levels follow on those.
//   c0: value 2 * bit2 + bit3 | c1: value 4 + 2 * bit2 + bit3 | c2: value 8 + bit3
// returns x with lane l holding the wave sum of value 4 * bit4 + 2 * bit2 + bit3 (l < 32) or 8 + bit3 (l >= 32).
#ifndef TS_FLUSH_SWAP32
#define TS_FLUSH_SWAP32 1
#endif
#ifndef TS_FLUSH_ASM
#define TS_FLUSH_ASM 1
#endif
#define TS_DPP_ROR8 "row_ror:8 row_mask:0xf bank_mask:0xf"
#define TS_DPP_ROR8_HI "row_ror:8 row_mask:0xf bank_mask:0xc"
#define TS_DPP_HM "row_half_mirror row_mask:0xf bank_mask:0xf"
#define TS_DPP_HM_HI "row_half_mirror row_mask:0xf bank_mask:0xa"
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
// `zero` (TS_FLUSH_ZERO_EARLY): called right behind the asm block, which is the last reader of the caller's accumulators -
// the caller's zeroing moves then stand where the permlane / DPP hazards below would otherwise need s_nops
template <bool HAVE9, class Zero>
__device__ __forceinline__ float wave_sum10_masked(const float v[10], int lane, Zero zero) {
    float c0, c1, c2, t1, t3;
    if (HAVE9) {
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %5, %5 " TS_DPP_ROR8 "\n\t"
            "v_add_f32_dpp %3, %7, %7 " TS_DPP_ROR8 "\n\t"
            "v_add_f32_dpp %1, %9, %9 " TS_DPP_ROR8 "\n\t"
            "v_add_f32_dpp %4, %11, %11 " TS_DPP_ROR8 "\n\t"
            "v_add_f32_dpp %2, %13, %13 " TS_DPP_ROR8 "\n\t"
            "v_add_f32_dpp %0, %6, %6 " TS_DPP_ROR8_HI "\n\t"
            "v_add_f32_dpp %3, %8, %8 " TS_DPP_ROR8_HI "\n\t"
            "v_add_f32_dpp %1, %10, %10 " TS_DPP_ROR8_HI "\n\t"
            "v_add_f32_dpp %4, %12, %12 " TS_DPP_ROR8_HI "\n\t"
            "v_add_f32_dpp %2, %14, %14 " TS_DPP_ROR8_HI "\n\t"
            "v_add_f32_dpp %0, %0, %0 " TS_DPP_HM "\n\t"
            "v_add_f32_dpp %1, %1, %1 " TS_DPP_HM "\n\t"
            "v_add_f32_dpp %2, %2, %2 " TS_DPP_HM "\n\t"
            "v_add_f32_dpp %0, %3, %3 " TS_DPP_HM_HI "\n\t"
            "v_add_f32_dpp %1, %4, %4 " TS_DPP_HM_HI "\n\t"
            "s_nop 1"
            : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(t1), "=&v"(t3)
            : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]),
              "v"(v[9]));
    } else {        // nine values: both halves of a row end up holding value 8 in c2
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %5, %5 " TS_DPP_ROR8 "\n\t"
            "v_add_f32_dpp %3, %7, %7 " TS_DPP_ROR8 "\n\t"
            "v_add_f32_dpp %1, %9, %9 " TS_DPP_ROR8 "\n\t"
            "v_add_f32_dpp %4, %11, %11 " TS_DPP_ROR8 "\n\t"
            "v_add_f32_dpp %2, %13, %13 " TS_DPP_ROR8 "\n\t"
            "v_add_f32_dpp %0, %6, %6 " TS_DPP_ROR8_HI "\n\t"
            "v_add_f32_dpp %3, %8, %8 " TS_DPP_ROR8_HI "\n\t"
            "v_add_f32_dpp %1, %10, %10 " TS_DPP_ROR8_HI "\n\t"
            "v_add_f32_dpp %4, %12, %12 " TS_DPP_ROR8_HI "\n\t"
            "v_add_f32_dpp %0, %0, %0 " TS_DPP_HM "\n\t"
            "v_add_f32_dpp %1, %1, %1 " TS_DPP_HM "\n\t"
            "v_add_f32_dpp %2, %2, %2 " TS_DPP_HM "\n\t"
            "v_add_f32_dpp %0, %3, %3 " TS_DPP_HM_HI "\n\t"
            "v_add_f32_dpp %1, %4, %4 " TS_DPP_HM_HI "\n\t"
            "s_nop 1"
            : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(t1), "=&v"(t3)
            : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]));
    }
    zero();
    // lane bit 4: odd / even rows exchanged in one issue (v_permlane16_swap), c0 stays in the even rows
    const u2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(c0), __float_as_uint(c1), false, false);
    const float d = __uint_as_float(r.x) + __uint_as_float(r.y);
#if TS_FLUSH_SWAP32
    // c2: the same exchange with itself; lane bit 5: v_permlane32_swap hands the upper half of d and the lower half
    // of c2 across in one issue, so lanes 0-31 end with d's total and lanes 32-63 with c2's - the same pairs of
    // summands as the ds_bpermute form below (bit-identical rows) without three trips through the LDS crossbar
    // (~65 cycles each, two of them dependent) on every row
    const u2v q = __builtin_amdgcn_permlane16_swap(__float_as_uint(c2), __float_as_uint(c2), false, false);
    const float e = __uint_as_float(q.x) + __uint_as_float(q.y);
    const u2v h = __builtin_amdgcn_permlane32_swap(__float_as_uint(d), __float_as_uint(e), false, false);
    float x = __uint_as_float(h.x) + __uint_as_float(h.y);
#else
    float dd = d + __shfl_xor(d, 32, 64);            // lane bit 5
    c2 += __shfl_xor(c2, 16, 64);
    c2 += __shfl_xor(c2, 32, 64);
    float x = (lane & 32) ? c2 : dd;                 // one register for the two quad levels
#endif
    x = dpp_add_t<0x4E, 0xF>(x);
    return dpp_add_t<0xB1, 0xF>(x);
}

// Wave-reduces the 6+CH per-lane sums of one (tile, Gaussian) and writes its row of `partials`
// (lanes 48..57 each store one float of the 40/48-byte row, lane 58 sets the flag; TS_FLUSH_ASM: the first lane
// of ten quads stores, lane 1 sets the flag).
#ifndef TS_FLUSH_ZERO_EARLY
#define TS_FLUSH_ZERO_EARLY 1
#endif
// ROWS THROUGH LDS (round 6, TS_ROWS_LDS).  The ablation table of round 6 (profiles/r06d_compositing_ablation.txt) put
// a number on the flush for the first time: raster_bwd 453 us, without any flush 183 us, with the stores but without the
// cross-lane butterfly 383 us - the two scattered stores per row (ten dwords from ten lanes + a flag byte: 4.85 M
// vector-memory instructions per launch, each one a trip through the CU's one address unit at a quarter rate for
// 64-bit addresses) cost THREE times what the 24-issue butterfly costs.  So a row is no longer stored when it is
// reduced: the writer lanes park it in LDS - in the 48 bytes of the entry's own staged record, which is dead once its
// bodies have read it; the row slot stays where the record keeps it (word 11), word 10 (the list index) becomes the
// mark "this entry has a row" - and when the chunk is done lane j stores row j with three 16-byte stores and its flag
// byte: four vector-memory instructions per chunk of up to 64 rows instead of two per row.  Same values in the same
// places: gradients bit for bit.
#ifndef TS_ROWS_LDS
#define TS_ROWS_LDS 1
#endif
constexpr int kRowMark = -1;               // word 10 of a staged record once its row has been parked there
// word of the record this lane fills when a row is parked (-1: none): the reduction leaves value w in the first lane of
// ten quads (see wave_sum10_masked); lane 36 - first lane of an idle quad - writes the mark
__device__ __forceinline__ int row_word_of_lane(int lane, int values) {
    const int b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1;
    const int w = (lane & 32) ? 8 + b3 : 4 * b4 + 2 * b2 + b3;
    const bool writer = (lane & 3) == 0 && ((lane & 32) == 0 || (lane & 0x14) == 0);
    if (lane == 36) return 10;
    return (writer && w < values) ? w : -1;
}
template <int CH>
__device__ __forceinline__ void flush_row(float (&v)[6 + CH], int slot_i, long long num_isects,
                                          float* __restrict__ partials,
                                          unsigned char* __restrict__ row_flags, int lane,
                                          float* lds_row = nullptr, int row_word = -1) {
    // num_isects carries the row-flag value in its top byte (see ts_raster_bwd: TS_RASTER_FLAG_GEN)
    const unsigned char flag_val = (unsigned char)((unsigned long long)num_isects >> 56);
    float r;
    if (TS_ABLATE == 4) {               // timing experiment: no cross-lane reduction
        r = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])) + v[8];
    } else {
        float v10[10];
#pragma unroll
        for (int c = 0; c < 10; ++c) v10[c] = c < 6 + CH ? v[c] : 0.0f;
        auto zero = [&]() {
            if (!TS_FLUSH_ZERO_EARLY) return;
            // zero the accumulators two at a time (v_mov_b64 on a register pair)
#pragma unroll
            for (int c = 0; c + 1 < 6 + CH; c += 2) {
                f2 z = (f2)(0.0f);
                asm volatile("" : "+v"(z));
                v[c] = z.x; v[c + 1] = z.y;
            }
            if ((6 + CH) & 1) v[5 + CH] = 0.0f;
        };
        if (TS_FLUSH_ASM) r = wave_sum10_masked<(CH == 4)>(v10, lane, zero);
        else { r = wave_sum10<(CH == 4)>(v10, lane); zero(); }
    }
    if (TS_ROWS_LDS && TS_FLUSH_ASM && lds_row != nullptr) {   // park the row in the entry's staged record
        if (row_word >= 0) lds_row[row_word] = row_word == 10 ? __int_as_float(kRowMark) : r;
        return;
    }
    const long long slot = (long long)slot_i;                  // < num_isects by construction (pack_splats)
    (void)num_isects;
    if (TS_FLUSH_ASM) {
        const int b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1;
        const int w = (lane & 32) ? 8 + b3 : 4 * b4 + 2 * b2 + b3;
        const bool writer = (lane & 3) == 0 && ((lane & 32) == 0 || (lane & 0x14) == 0);
        if (writer && w < 6 + CH) {
            partials[slot * TS_PARTIAL_ROW_FLOATS + w] = r;
            // "this row now holds data": stored by every writer lane (same byte, one transaction) inside the row's
            // exec region - a second region for one lane cost a saveexec / restore / branch per row
            if (TS_FLAG_WITH_ROW) row_flags[slot] = flag_val;
        }
        if (!TS_FLAG_WITH_ROW && lane == 1) row_flags[slot] = flag_val;
        return;
    }
    const int w = lane - 48;                                   // row 3: lane 48+w holds value w, w < 10
    if (w >= 0) {
        if (w < 6 + CH) {
            if (TS_NT_ROWS) __builtin_nontemporal_store(r, partials + slot * TS_PARTIAL_ROW_FLOATS + w);
            else partials[slot * TS_PARTIAL_ROW_FLOATS + w] = r;
        } else if (w == 10) {
            row_flags[slot] = flag_val;                        // this row now holds data
        }
    }
}

// Replays the `cnt` staged Gaussians of one chunk back to front (backward).
//   per pixel and block k: T = transmittance behind the Gaussian being replayed, R = T_final *
//   (v_alpha - bg . v_out) - sum over the Gaussians already replayed of fac * (colour . v_out),
//   vo = v_out, fidx = index of the last Gaussian the forward pass composited.
// Inside a block the body is full-exec and branch free: a lane that is not valid uses alpha = 0
// (ra = 1, fac = 0, v_sig = 0) and changes nothing.
// STARTED (round 6, TS_BWD_STARTED): every pixel of every block that takes part in this chunk has its whole list in
// front of it (fidx >= the chunk's highest index - true for all but the last one or two chunks of a list, since nearly
// every pixel's last contribution lies there), so `idx <= fidx[k]` holds for every entry and lane: the body drops the
// compare and the s_and_b64 of the two ballots (2 of its ~43 issue slots).  Same sums, bit for bit.
#ifndef TS_BWD_STARTED
#define TS_BWD_STARTED 0          // measured (round 6, config 3): raster_bwd 447 - 453 us either way; off
#endif
template <int CH, bool GENERAL, int NBX, bool STARTED = false>
__device__ __forceinline__ void bwd_chunk(const float4* __restrict__ lds, int cnt, const float (&fpx)[NBX],
                                          const float (&fpy)[2], float (&T)[2 * NBX], float (&R)[2 * NBX],
                                          const float (&vo)[2 * NBX][CH], const int (&fidx)[2 * NBX],
                                          float (&acc)[6 + CH], long long num_isects,
                                          float* __restrict__ partials,
                                          unsigned char* __restrict__ row_flags, int lane TS_SEG_PARAM) {
    // Every fused multiply-add below is written out; implicit contraction is switched off so that the
    // GENERAL and the lean instantiation round identically (otherwise `acc[0] += -am * v_a` fuses
    // in one and not in the other, and a gradient would depend on which chunk an entry lands in).
#pragma clang fp contract(off)
    TS_WORK(0, cnt);
    // staged record: three float4 for three channels (the block mask rides in the unused fourth colour word),
    // four for RGB + depth.  TS_BWD_EARLY_RECORD (an experiment, off): the NEXT entry's record requested in front of the
    // row flush of this one - its registers are dead by then, and the flush (no LDS access in it) would cover the round
    // trip that the wave waits out at the top of every iteration (7.6 % of a backward wave's cycles); slower (above)
    constexpr int RS = CH == 3 ? 3 : 4;
    const int row_word = row_word_of_lane(lane, 6 + CH);      // (TS_ROWS_LDS) the word of a parked row this lane writes
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    float bm_f = 0.0f;
    if (TS_BWD_EARLY_RECORD && cnt > 0 && TS_ABLATE != 3) {
        r0 = lds[0]; r1 = lds[1]; r2 = lds[2];
        bm_f = CH == 3 ? r2.y : lds[3].x;
    }
    for (int j = 0; j < (TS_ABLATE == 3 ? 0 : cnt); ++j) {
        TS_SEG_T0(tseg_a);
        if (!TS_BWD_EARLY_RECORD) {
            r0 = lds[RS * j]; r1 = lds[RS * j + 1]; r2 = lds[RS * j + 2];
            bm_f = CH == 3 ? r2.y : lds[RS * j + 3].x;
        }
        const int bm = __builtin_amdgcn_readfirstlane(__float_as_int(bm_f));
        const int idx = __float_as_int(r2.z);
        const float neg_lo = -r1.y;
        float col[CH];
        col[0] = r1.z; col[1] = r1.w; col[2] = r2.x;
        if (CH == 4) col[CH - 1] = r2.y;
#if TS_TIMELINE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (so that the record read is timed on its own)
#endif
        TS_SEG_ADD(ts_seg_, 1, tseg_a);
        TS_SEG_T0(tseg_b);
        TS_WORK(1, __popc(bm & ((1 << (2 * NBX)) - 1)));

        int any = 0;
#pragma unroll
        for (int k = 0; k < 2 * NBX; ++k) {
            if (!(bm & (1 << k))) continue;                           // wave-uniform
            if (TS_ABLATE == 5) { asm volatile("" ::: "memory"); continue; }
            TS_STAT(3, 1);
            const float dx = r0.x - fpx[k % NBX], dy = r0.y - fpy[k / NBX];
            const float sgl = sigma_l2(r0.z, r0.w, r1.x, neg_lo, dx, dy);
            const float araw = __builtin_amdgcn_exp2f(-sgl);           // opacity * exp(-sigma)
            float a = araw;
            mask64 validm = TS_BALLOT(araw >= ts::kAlphaMin);
            if (!STARTED) validm &= TS_BALLOT(idx <= fidx[k]);
            if (GENERAL) {
                a = fminf(ts::kAlphaMaxBwd, araw);
                validm &= TS_BALLOT(sgl >= neg_lo);                    // sigma >= 0
            }
            if (validm == 0ull) continue;                              // wave-uniform
            TS_STAT(4, 1);
            TS_STAT(6, __popcll(validm));
            // lane packing: how many bodies have their valid pixels in ONE half (rows 0-3 / 4-7) or ONE quarter
            // (a 4x4 quadrant: lane bits 2 and 5) of the 8x8 block - what half-wave / quarter-wave bodies could pair up
            TS_STAT(8, ((unsigned)validm == 0u) != ((unsigned)(validm >> 32) == 0u));
            {
                const mask64 q0 = 0x000000000F0F0F0Full, q1 = 0x00000000F0F0F0F0ull;
                const int quads = ((validm & q0) != 0) + ((validm & q1) != 0) + ((validm & (q0 << 32)) != 0) + ((validm & (q1 << 32)) != 0);
                TS_STAT(9, quads == 1);
                TS_STAT(10, quads == 2);
                TS_STAT(11, quads == 3);
            }
            any = 1;
            // the select takes its mask from an ordinary SGPR pair: the VOP2 form on a vcc that the SCALAR unit
            // wrote (the s_and of the two ballots) costs a wave 19 cycles instead of 5 (tools/micro/lat_bench.hip)
            float am;
            if (TS_BWD_EXEC_MASK) am = a;
            else if (TS_SELECT_SGPR) asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(am) : "v"(a), "s"(validm));
            else am = TS_LANE(validm) ? a : 0.0f;
#if TS_BWD_EXEC_MASK
            // experiment: the rest of the body under EXEC = valid lanes (an invalid lane adds +-0 everywhere: same bits)
            if (TS_LANE(validm)) {
#endif
            const float ra = __builtin_amdgcn_rcpf(1.0f - am);
            const float Tk = T[k] * ra;                 // transmittance in front of the Gaussian
            const float fac = am * Tk;
            float cv = col[0] * vo[k][0];               // colour . v_out of this pixel
#pragma unroll
            for (int c = 1; c < CH; ++c) cv = __builtin_fmaf(col[c], vo[k][c], cv);
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[6 + c] = __builtin_fmaf(fac, vo[k][c], acc[6 + c]);
            // dL/dalpha = Tk (c . v_out) + ra (T_final (v_alpha - bg . v_out) - S_behind fac' c' . v_out)
            const float v_a = __builtin_fmaf(Tk, cv, ra * R[k]);
            R[k] = __builtin_fmaf(-fac, cv, R[k]);
            T[k] = Tk;
            // d alpha / d sigma = -alpha, or 0 where the 0.999 clamp is active
            float v_sig = -am * v_a;
            if (GENERAL && ts::kClampGatesGrad) v_sig = TS_LANE(TS_BALLOT(araw > ts::kAlphaMax)) ? 0.0f : v_sig;
            if (GENERAL && !ts::kClampGatesGrad) v_sig = -(TS_LANE(validm) ? araw : 0.0f) * v_a;   // -opacity vis v_alpha
            const float vdx = v_sig * dx, vdy = v_sig * dy;
            acc[0] += v_sig; acc[1] += vdx; acc[2] += vdy;
            acc[3] = __builtin_fmaf(vdx, dx, acc[3]);
            acc[4] = __builtin_fmaf(vdx, dy, acc[4]);
            acc[5] = __builtin_fmaf(vdy, dy, acc[5]);
#if TS_BWD_EXEC_MASK
            }
#endif
        }
        // `any` is wave-uniform.  It is passed through an empty asm so that the compiler cannot prove
        // "block 3 ran => a flush follows": with that knowledge it specialises the last block body
        // (results in fresh registers) and pays for it with 10-17 register copies per entry on the
        // joining paths; kept opaque, all four bodies accumulate in place and the flush reads acc.
        asm volatile("" : "+s"(any));
        TS_SEG_ADD(ts_seg_, 2, tseg_b);
        TS_SEG_T0(tseg_c);
        constexpr bool kRowsLds = TS_ROWS_LDS && TS_FLUSH_ASM && TS_ABLATE == 0;
        const int row_slot = kRowsLds ? 0 : __builtin_amdgcn_readfirstlane(__float_as_int(r2.w));
        if (TS_BWD_EARLY_RECORD && j + 1 < cnt) {
            r0 = lds[RS * (j + 1)]; r1 = lds[RS * (j + 1) + 1]; r2 = lds[RS * (j + 1) + 2];
            bm_f = CH == 3 ? r2.y : lds[RS * (j + 1) + 3].x;
        }
        if (any && TS_ABLATE != 7) {
            TS_STAT(5, 1);
            TS_WORK(2, 1);
            if (kRowsLds)
                flush_row<CH>(acc, row_slot, num_isects, partials, row_flags, lane,
                              reinterpret_cast<float*>(const_cast<float4*>(lds) + RS * j), row_word);
            else
                flush_row<CH>(acc, row_slot, num_isects, partials, row_flags, lane);
            if (!TS_FLUSH_ZERO_EARLY || TS_ABLATE == 4) {
                // zero the accumulators two at a time (v_mov_b64 on a register pair)
#pragma unroll
                for (int c = 0; c + 1 < 6 + CH; c += 2) {
                    f2 z = (f2)(0.0f);
                    asm volatile("" : "+v"(z));
                    acc[c] = z.x; acc[c + 1] = z.y;
                }
                if ((6 + CH) & 1) acc[5 + CH] = 0.0f;
            }
        }
        TS_SEG_ADD(ts_seg_, 3, tseg_c);
    }
    if (TS_ROWS_LDS && TS_FLUSH_ASM && TS_ABLATE == 0) {
        // the chunk's rows, parked in LDS by the flushes above: lane j stores the row of staged entry j
        TS_WAVE_SYNC();
        if (lane < cnt) {
            float4 p2 = lds[RS * lane + 2];
            if (__float_as_int(p2.z) == kRowMark) {
                const long long slot = (long long)__float_as_int(p2.w);      // < num_isects by construction (pack_splats)
                const float4 p0 = lds[RS * lane], p1 = lds[RS * lane + 1];
                if (CH == 3) p2.y = 0.0f;
                p2.z = 0.0f; p2.w = 0.0f;
                float4* dst = reinterpret_cast<float4*>(partials) + slot * kRowF4;
                dst[0] = p0; dst[1] = p1; dst[2] = p2;
                row_flags[slot] = (unsigned char)((unsigned long long)num_isects >> 56);
            }
        }
        TS_WAVE_SYNC();          // the records are dead now: the caller's next chunk stages over them
    }
}

// Workgroups of the backward launch are ONE wave (TS_BWD_WAVES): its waves never synchronise, and a workgroup's wave
// slots and LDS are only handed to the next workgroup when ALL its waves are done - with four tiles per workgroup a
// wave that finished early left its slot idle until the slowest of the four was through (raster_bwd 532 -> 522 us).
#ifndef TS_BWD_WAVES
#define TS_BWD_WAVES 1
#endif
constexpr int kBwdWaves = TS_BWD_WAVES;
template <int CH, bool SPLIT, int NBX, bool WL>
__global__ __launch_bounds__(64 * kBwdWaves, NBX == 2 ? TS_BWD_MIN_WAVES_16 : TS_BWD_MIN_WAVES) void raster_bwd_kernel(
    const ts_camera cam, const int num_tiles, const long long num_isects,
    const int* __restrict__ tile_bins, const int* __restrict__ ids_sorted,
    const float4* __restrict__ splats, const float* __restrict__ background,
    const float* __restrict__ final_Ts, const int* __restrict__ final_index,
    const float* __restrict__ v_out_img, const float* __restrict__ v_out_depth, const int planes,
    const float* __restrict__ v_out_alpha, const unsigned char* __restrict__ clamp_mask,
    float* __restrict__ partials, unsigned char* __restrict__ row_flags) {
    constexpr int NB = 2 * NBX;
    TS_LDS_PAD_DECL();
    __shared__ float4 lds_all[kBwdWaves][64 * 4];
    __shared__ float4 rect_all[kBwdWaves][NB];
    __shared__ float4 raw_all[TS_LDS_DMA ? kBwdWaves : 1][3 * 64];   // landing zone of the next chunk's records
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // list segments: S work items per cut tile, item `seg` replays the entries [sb, se) of the list; whole tiles (hybrid
    // launch: the first `whole` of every band) are one item.  Work items are handed out band by band: workgroup b
    // runs on XCD b % 8 and takes item b / 8 of band b % 8 - whole tiles first, the small items last
    constexpr bool kSegs = !SPLIT && NBX == 2 && !WL;
    const int S_cam = kSegs ? max(1, min(TS_CAM_SEGS(cam), kSegMax)) : 1;
    int S_seg = S_cam, tile, seg = 0, only = -1, ck_block = -1;
    if (kSegs && S_cam > 1) {
        const CutTiles m = cut_tiles(num_tiles, cam.hints);
        const int j = (int)(blockIdx.x >> 3) * kBwdWaves + wave;
        if (j >= m.items(S_cam)) return;
        int local = j;
        S_seg = 1;
        if (j >= m.whole) {
            S_seg = S_cam;
            local = m.whole + (j - m.whole) / S_cam;
            seg = (j - m.whole) % S_cam;
        }
        tile = (int)(blockIdx.x & 7) * m.band + local;
        if (tile >= num_tiles) return;
        // the segment order rotates from tile to tile: the front segments are the expensive ones (every pixel is still
        // alive there), and one-wave workgroups land on a CU's SIMDs in dispatch order - with seg = unit % S and S = 4 one
        // SIMD of every CU would get all the front segments (raster_bwd 525 -> 856 us, measured)
        if (S_seg > 1) seg = (int)(((unsigned)seg + (((unsigned)tile * 2654435761u) >> 20)) % (unsigned)S_seg);
        ck_block = m.block_of(tile);
    } else {
        const int units = SPLIT ? NB * num_tiles : num_tiles;
        const int unit = xcd_tile_group((units + kBwdWaves - 1) / kBwdWaves) * kBwdWaves + wave;
        if (unit >= units) return;
        tile = SPLIT ? unit / NB : unit;
        only = SPLIT ? unit % NB : -1;            // SPLIT: this wave owns block `only`, row slot*NB + only
    }
    float4* rects = rect_all[wave];
    const int tbx = NBX == 2 ? cam.tile_bounds_x : (cam.tile_bounds_x + 1) >> 1;
    const int tx = tile % tbx, ty = tile / tbx + cam.tile_row0;
    const int list = WL ? (ty - cam.tile_row0) * ((tbx + 1) >> 1) + (tx >> 1) : tile;
    const int2 range = reinterpret_cast<const int2*>(tile_bins)[list];
    if (range.y <= range.x) return;
    int sb = range.x, se = range.y;
    bool front = false;                           // entries lie behind this segment: it starts from a checkpoint
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 n0 = zero4, n1 = zero4, n2 = zero4;
    int id_next = 0;
    if (kSegs && S_seg > 1) {
        const int len = range.y - range.x;
        if (len < kSegMinList) {
            if (seg != 0) return;                  // a short list is one segment
        } else {
            sb = range.x + (seg > 0 ? seg_bound(len, S_seg, seg) : 0);
            if (sb >= range.y) return;
            se = seg + 1 < S_seg ? min(range.x + seg_bound(len, S_seg, seg + 1), range.y) : range.y;
            front = se < range.y;
        }
    }
    TS_WAVE_CLOCK(1, (int)blockIdx.x * kBwdWaves + wave, se - sb);
    float4* lds = lds_all[wave];
    const int px0 = tx * (8 * NBX) + (lane & 7), py0 = ty * 16 + (lane >> 3);
    // sample positions of the lane's pixel in the block columns / the upper and lower block row
    float fpx[NBX];
#pragma unroll
    for (int c = 0; c < NBX; ++c) fpx[c] = (float)(px0 + 8 * c) + ts::kPixOff;
    const float fpy[2] = {(float)py0 + ts::kPixOff, (float)(py0 + 8) + ts::kPixOff};
    const float X0 = (float)(tx * (8 * NBX)) + ts::kPixOff, Y0 = (float)(ty * 16) + ts::kPixOff;
    const int W = cam.img_width, H = cam.img_height;
    const int row_off = cam.tile_row0 * 16;

    float bg[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) bg[c] = background[c];

    // per pixel: T = transmittance behind the Gaussian being replayed; R = T_final*(v_alpha - bg.v_out)
    // - sum over the Gaussians already replayed of fac * (colour . v_out)   (see the inner loop)
    float T[NB], R[NB], vo[NB][CH];
    int fidx[NB], bmax[NB];
    int fmax = -1;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int px = px0 + 8 * (k % NBX), py = py0 + 8 * (k / NBX);
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
            float cb[CH];                          // colour the entries BEHIND this segment contributed (front only)
#pragma unroll
            for (int c = 0; c < CH; ++c) cb[c] = 0.0f;
            float T_start = T[k];
            if (kSegs && front) {                 // records seg + 1 (T) and seg + 2 .. S (colour behind) of the tile's block
                const float* ckp = final_Ts + ckpt_offset(seg_plane_stride(cam)) + ckpt_record<CH>(ck_block, S_cam, 1);
                T_start = ckp[(size_t)seg * ((1 + CH) * 256) + 4 * (64 * k + lane)];
                // all the records are requested before the first addition (one round trip, not one per record); the
                // bounds are wave-uniform, a record that is not wanted stays zero, and the sum runs from the back
                float4 d[kSegMax - 1];
                float d3[kSegMax - 1];
#pragma unroll
                for (int r = 2; r <= kSegMax; ++r) {
                    d[r - 2] = zero4;
                    d3[r - 2] = 0.0f;
                    if (r > S_seg || r < seg + 2) continue;
                    const float* rec = ckp + (size_t)(r - 1) * ((1 + CH) * 256);
                    d[r - 2] = reinterpret_cast<const float4*>(rec)[64 * k + lane];
                    if (CH == 4) d3[r - 2] = rec[1024 + 64 * k + lane];
                }
#pragma unroll
                for (int r = kSegMax; r >= 2; --r) {
                    cb[0] += d[r - 2].y; cb[1] += d[r - 2].z; cb[2] += d[r - 2].w;
                    if (CH == 4) cb[CH - 1] += d3[r - 2];
                }
            }
            const int pass = clamp_mask ? clamp_mask[pix] : 7;   // backward of the fused clamp(max=1)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                float g;
                if (CH == 4 && planes)       // two gradient planes, either may be absent (no part in the loss)
                    g = c < 3 ? (v_out_img ? v_out_img[pix * 3 + c] : 0.0f) : (v_out_depth ? v_out_depth[pix] : 0.0f);
                else
                    g = v_out_img[pix * CH + c];
                vo[k][c] = (c >= 3 || (pass & (1 << c))) ? g : 0.0f;
                dotbg += bg[c] * vo[k][c];
            }
            const float va = v_out_alpha ? v_out_alpha[pix] : 0.0f;
            R[k] = T[k] * (va - dotbg);             // T[k] = T_final here
            if (kSegs && front) {
                float behind = cb[0] * vo[k][0];
#pragma unroll
                for (int c = 1; c < CH; ++c) behind = __builtin_fmaf(cb[c], vo[k][c], behind);
                R[k] -= behind;
                T[k] = T_start;                     // transmittance behind the segment's last entry
            }
        }
        bmax[k] = wave_max_int(fidx[k]);            // last list index any pixel of block k used
        fmax = max(fmax, bmax[k]);
    }
    const int last = min(se - 1, fmax);
    if (last < sb) return;                          // the forward pass stopped every pixel before this segment

    // per-lane sums over its (up to) NB pixels for the Gaussian being replayed, updated in place
    // by the block bodies and zeroed after each row is written:
    // {S v, S v dx, S v dy, S v dx^2, S v dx dy, S v dy^2, v_c0, v_c1, ...}
    float acc[6 + CH];
#pragma unroll
    for (int c = 0; c < 6 + CH; ++c) acc[c] = 0.0f;

    // same software pipeline as the forward kernel, walking the list back to front
    float4* raw = raw_all[TS_LDS_DMA ? wave : 0];
    if (last - lane >= sb) {
        const int g = ids_sorted[last - lane];
#if TS_LDS_DMA
        dma_record(splats, g, raw);
#else
        n0 = splats[3 * (size_t)g]; n1 = splats[3 * (size_t)g + 1]; n2 = splats[3 * (size_t)g + 2];
#endif
    }
    if (last - 64 - lane >= sb) id_next = ids_sorted[last - 64 - lane];

    for (int hi = last; hi >= sb; hi -= 64) {
        TS_SEG_T0(tseg_p);
        const int i = hi - lane;
        const bool have = i >= sb;
#if TS_LDS_DMA
        TS_DMA_WAIT();                               // this chunk's records have landed (and id_next has arrived)
        const float4 q0 = have ? raw[lane] : zero4, q1 = have ? raw[64 + lane] : zero4,
                     q2 = have ? raw[128 + lane] : zero4;
        TS_LDS_WAIT();                               // read out before the next chunk's records may overwrite them
        if (i - 64 >= sb) dma_record(splats, id_next, raw);
#else
        const float4 q0 = n0, q1 = n1, q2 = n2;
        if (i - 64 >= sb) {
            const int g = id_next;
            n0 = splats[3 * (size_t)g]; n1 = splats[3 * (size_t)g + 1]; n2 = splats[3 * (size_t)g + 2];
        }
#endif
        if (i - 128 >= sb) id_next = ids_sorted[i - 128];
        int blocks;
        {   // rectangle of the pixels whose forward list reaches into this chunk (fidx >= chunk low)
            bool sel[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) sel[k] = fidx[k] >= hi - 63;
            blocks = update_rects<NBX>(sel, X0, Y0, rects, lane);
            TS_WAVE_SYNC();
        }
        Staged s = stage_splat<NB>(have, q0, q1, rects, blocks);
        s.mask &= bbox_blocks<NBX, WL>(__float_as_int(q2.w), tx);
#pragma unroll
        for (int k = 0; k < NB; ++k)
            if (i > bmax[k]) s.mask &= ~(1 << k);   // nothing in block k got this far in forward
        const bool keep = (s.mask & ((1 << NB) - 1)) != 0;
        const unsigned long long mask = __ballot(keep);
        const int cnt = __popcll(mask);
        if (keep) {
            const int pos = __popcll(mask & ((1ull << lane) - 1ull));
            // row of this (tile, Gaussian): the slot gsplat's count reserves for the 16x16 tile (column
            // tx16, row ty) of the Gaussian's tile box; a wide tile uses the slot of its left half, or of
            // its right half where the box starts there (record: bbox_w | bbox_minx << 16)
            // SPLIT: one row per (16x16 tile, Gaussian, block of that tile) - block `only` of a wide tile
            // lies in its half (only % NBX) >> 1 and is block lb of that 16x16 tile
            const int wm = __float_as_int(q2.w);
            const int half = SPLIT ? ((only % NBX) >> 1) : 0;
            const int tx16 = NBX == 2 ? tx : (SPLIT ? 2 * tx + half : max(2 * tx, wm >> 16));
            int slot = __float_as_int(q2.z) + ty * (wm & 0xffff) + tx16;
            if (SPLIT) slot = 4 * slot + (NBX == 2 ? only : ((only % NBX) & 1) + 2 * (only / NBX));
            constexpr int RS = CH == 3 ? 3 : 4;           // see bwd_chunk
            lds[RS * pos] = make_float4(s.gx, s.gy, s.hA, s.B);
            lds[RS * pos + 1] = make_float4(s.hC, s.lo, q1.z, q1.w);
            lds[RS * pos + 2] = make_float4(q2.x, CH == 3 ? __int_as_float(s.mask) : q2.y, __int_as_float(i),
                                            __int_as_float(slot));
            if (CH == 4) lds[RS * pos + 3] = make_float4(__int_as_float(s.mask), 0.f, 0.f, 0.f);
        }
        TS_WAVE_SYNC();
        TS_STAT(2, cnt);
        TS_STAT(7, min(64, hi - sb + 1));
        TS_SEG_ADD(ts_wave_clock_.seg, 0, tseg_p);
        // every block that takes part (a pixel's list reaches into the chunk) has ALL its pixels' lists in front of the
        // chunk's last entry: no per-entry `idx <= fidx` test (see STARTED)
        bool started = TS_BWD_STARTED && NBX == 2 && !SPLIT;
        if (started) {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const unsigned long long all = __ballot(fidx[k] >= hi);
                if ((blocks & (1 << k)) && all != ~0ull) started = false;
            }
        }
        const bool general = __ballot(keep && (s.mask & (1 << NB))) != 0ull;
        if (general)
            bwd_chunk<CH, true, NBX>(lds, cnt, fpx, fpy, T, R, vo, fidx, acc, num_isects, partials,
                                     row_flags, lane TS_SEG_ARG);
        else if (started)
            bwd_chunk<CH, false, NBX, true>(lds, cnt, fpx, fpy, T, R, vo, fidx, acc, num_isects, partials,
                                            row_flags, lane TS_SEG_ARG);
        else
            bwd_chunk<CH, false, NBX>(lds, cnt, fpx, fpy, T, R, vo, fidx, acc, num_isects, partials,
                                      row_flags, lane TS_SEG_ARG);
        TS_WAVE_SYNC();
    }
}

// Row layout (raw sums over the pixels of one tile, see raster_bwd_kernel):
//   [ S v_sigma, S v_sigma dx, S v_sigma dy, S v_sigma dx^2, S v_sigma dx dy, S v_sigma dy^2, c0..c3, -, - ]
// with d = xy - pixel.  The conic / opacity factors are applied once per Gaussian here.
// color_mask (may be NULL): clamp mask of the colour stage, applied here (bit c clear -> v_colors[c] = 0)
// when the gradients are summed over ranks before the colour stage's backward runs.
// LANES per Gaussian (TS_REDUCE_LANES, round 6): one lane per Gaussian walked ALL its slots, so a wave took as long as the
// largest of its 64 Gaussians (6 slots on average, 100+ for a large one) and every lane's walk was a chain of dependent
// flag -> row loads.  Now LANES consecutive lanes share a Gaussian: lane j takes the slots j, j + LANES, ... of its
// range in ascending order (the flags of LANES neighbouring slots are neighbouring bytes), sums them in double, and the
// LANES partial sums are combined pairwise (j ^ 1, then j ^ 2, ...) - a fixed order, so the gradients stay
// bit-reproducible run to run.
#ifndef TS_REDUCE_LANES
#define TS_REDUCE_LANES 1          // measured (config 3 / 5, us): 1: 59 / 431, 2: 56 / -, 4: 58 / 452, 8: 87 / 571, 16: 183 / 985 - the walk is bound by its row gathers, not by its longest lane
#endif
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return __hiloint2double(hi, lo);
}
template <int CH>
__global__ __launch_bounds__(256) void reduce_partials_kernel(
    int n, int flags, const int* __restrict__ num_tiles_hit, const int* __restrict__ cum_tiles_hit,
    const float4* __restrict__ partials, const unsigned char* __restrict__ row_flags,
    const float4* __restrict__ splats, float* __restrict__ v_xy, float* __restrict__ v_conic,
    float* __restrict__ v_colors, float* __restrict__ v_opacity, float* __restrict__ v_depth,
    const unsigned char* __restrict__ color_mask, float4* __restrict__ grad_rows) {
    constexpr int L = TS_REDUCE_LANES;
    static_assert(L == 1 || L == 2 || L == 4 || L == 8 || L == 16, "lanes per Gaussian: a power of two within a row of 16");
    const long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
    const int i = (int)(gi / L), j = (int)(gi % L);
    if (i >= n) return;                           // (all L lanes of a Gaussian leave together)
    const int cnt = num_tiles_hit[i];
    const long long end = cum_tiles_hit[i];
    // a row holds data of THIS pass iff its flag equals the pass's value (legacy value 1 on a zeroed array, or
    // the caller's generation on a persistent one: TS_RASTER_FLAG_GEN)
    const unsigned int gen = ((unsigned int)flags >> 8) & 0xffu ? ((unsigned int)flags >> 8) & 0xffu : 1u;
    // The rows of a Gaussian are summed in DOUBLE precision and rounded once (round 6).  A large Gaussian has rows in
    // 100+ tiles whose moments S v dx, S v dx dy, ... carry both signs, and the running float32 sum lost ~sqrt(rows) ulps of
    // the largest partial sum - which the projection's VJP then amplifies wherever its terms cancel (a near-isotropic
    // Gaussian's quaternion gradient: tools/vjp_probe.py, fuzz seed 52).  The kernel waits for memory; the ten
    // conversions and double adds per row are free.
    double s[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) s[c] = 0.0;
    auto add_row = [&](const float4& p0, const float4& p1, const float4& p2) {
        s[0] += (double)p0.x; s[1] += (double)p0.y; s[2] += (double)p0.z; s[3] += (double)p0.w;
        s[4] += (double)p1.x; s[5] += (double)p1.y; s[6] += (double)p1.z; s[7] += (double)p1.w;
        s[8] += (double)p2.x; s[9] += (double)p2.y;
    };
    if (flags & TS_RASTER_SPLIT_BLOCKS) {         // four rows per (tile, Gaussian): one flag word per pair
        const unsigned int* flags4 = reinterpret_cast<const unsigned int*>(row_flags);
        for (long long sl = end - cnt + j; sl < end; sl += L) {
            const unsigned int fw = flags4[sl];
            unsigned int f = 0u;                      // byte k set: row k of the slot was written in this pass
#pragma unroll
            for (int k = 0; k < 4; ++k) f |= (((fw >> (8 * k)) & 0xffu) == gen) ? (0xffu << (8 * k)) : 0u;
            if (f == 0u) continue;
            float4 p0[4], p1[4], p2[4];            // the slot's flagged rows requested together, added in order
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (f & (0xffu << (8 * k))) {
                    p0[k] = partials[kRowF4 * (4 * sl + k)]; p1[k] = partials[kRowF4 * (4 * sl + k) + 1];
                    p2[k] = partials[kRowF4 * (4 * sl + k) + 2];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (f & (0xffu << (8 * k))) add_row(p0[k], p1[k], p2[k]);
            }
        }
    } else {
        // kAhead slots per lane and step: their flags, then the rows of the flagged ones, are all requested before the
        // first addition, which happens in slot order - a lane's walk is otherwise a chain of dependent
        // flag -> row -> flag loads
        constexpr int kAhead = TS_REDUCE_AHEAD;
        for (long long s0 = end - cnt + j; s0 < end; s0 += kAhead * L) {
            bool f[kAhead];
            float4 p0[kAhead], p1[kAhead], p2[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) f[u] = (s0 + u * L < end) && row_flags[s0 + u * L] == gen;   // else: not written in this pass (stale)
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                if (f[u]) {
                    p0[u] = partials[kRowF4 * (s0 + u * L)]; p1[u] = partials[kRowF4 * (s0 + u * L) + 1];
                    p2[u] = partials[kRowF4 * (s0 + u * L) + 2];
                }
            }
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                if (f[u]) add_row(p0[u], p1[u], p2[u]);
            }
        }
    }
#pragma unroll
    for (int m = 1; m < L; m <<= 1) {
#pragma unroll
        for (int c = 0; c < 10; ++c) s[c] += shfl_xor_f64(s[c], m);
    }
    if (j != 0) return;
    float4 a0 = make_float4((float)s[0], (float)s[1], (float)s[2], (float)s[3]);
    float4 a1 = make_float4((float)s[4], (float)s[5], (float)s[6], (float)s[7]);
    float4 a2 = make_float4((float)s[8], (float)s[9], 0.f, 0.f);
    float vx = 0.f, vy = 0.f, vop = 0.f;
    if (cnt > 0) {
        const float4 q0 = splats[3 * (size_t)i], q1 = splats[3 * (size_t)i + 1];
        const float A = q0.w, B = q1.x, C = q1.y, op = q0.z;
        vx = A * a0.y + B * a0.z;
        vy = B * a0.y + C * a0.z;
        vop = op > 0.0f ? -a0.x / op : 0.0f;
        if (flags & TS_RASTER_LOGIT_OPACITY) vop *= op * (1.0f - op);   // through the sigmoid
        if (color_mask) {
            const int m = color_mask[i];
            if (!(m & 1)) a1.z = 0.0f;
            if (!(m & 2)) a1.w = 0.0f;
            if (!(m & 4)) a2.x = 0.0f;
        }
    }
    if (grad_rows) {        // Gaussian-sharded frame (shard.hip): one 48-byte row per imported record
        grad_rows[3 * (size_t)i] = make_float4(vx, vy, 0.5f * a0.w, a1.x);
        grad_rows[3 * (size_t)i + 1] = make_float4(0.5f * a1.y, a1.z, a1.w, a2.x);
        grad_rows[3 * (size_t)i + 2] = make_float4(CH == 4 ? a2.y : 0.0f, vop, 0.0f, 0.0f);
        return;
    }
    reinterpret_cast<float2*>(v_xy)[i] = make_float2(vx, vy);
    v_opacity[i] = vop;
    v_conic[3 * i] = 0.5f * a0.w; v_conic[3 * i + 1] = a1.x; v_conic[3 * i + 2] = 0.5f * a1.y;
    if (CH == 4 && v_depth == nullptr) {
        reinterpret_cast<float4*>(v_colors)[i] = make_float4(a1.z, a1.w, a2.x, a2.y);
    } else if (CH == 4) {                 // RGB + depth pass: v_colors is [n,3], channel 3 goes to v_depth[n]
        v_colors[3 * i] = a1.z; v_colors[3 * i + 1] = a1.w; v_colors[3 * i + 2] = a2.x;
        v_depth[i] = a2.y;
    } else {
        v_colors[3 * i] = a1.z; v_colors[3 * i + 1] = a1.w; v_colors[3 * i + 2] = a2.x;
    }
}

inline int launch_status() { return (int)hipGetLastError(); }

}  // namespace

extern "C" {

#if TS_TIMELINE
int ts_debug_timeline(unsigned long long* out_host, int which, int count) {   // developer builds only (not part of the ABI)
    if (which < 0 || which > 1 || count < 0 || count > kTimelineMax) return TS_E_BADARG;
    return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(ts_timeline), (size_t)count * kTimelineRow * 8,
                                    (size_t)which * kTimelineMax * kTimelineRow * 8);
}
#endif

#if TS_STATS
int ts_debug_stats(unsigned long long* out_host, int reset) {      // developer builds only (not part of the ABI)
    hipError_t e = hipMemcpyFromSymbol(out_host, HIP_SYMBOL(ts_stats), sizeof(ts_stats));
    if (e == hipSuccess && reset) {
        unsigned long long z[12] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(ts_stats), z, sizeof(z));
    }
    return (int)e;
}
#endif

int64_t ts_final_floats(const ts_camera* cam, int32_t channels) {
    if (!cam || (channels != 3 && channels != 4)) return TS_E_BADARG;
    const size_t plane = seg_plane_stride(*cam);
    const int S = min(TS_CAM_SEGS(*cam), kSegMax);
    if (S <= 1 || cam->wide_tiles) return (int64_t)plane;
    const CutTiles m = cut_tiles(ts_num_tiles(cam), cam->hints);
    return (int64_t)(ckpt_offset(plane) + (size_t)8 * (size_t)m.cut() * (size_t)(S * (1 + channels) * 256));
}

int32_t ts_cut_tiles(const ts_camera* cam, int32_t* band, int32_t* whole) {
    if (!cam) return TS_E_BADARG;
    const CutTiles m = cut_tiles(ts_num_tiles(cam), cam->hints);
    if (band) *band = m.band;
    if (whole) *whole = m.whole;
    return 8 * m.cut();
}

int ts_raster_fwd(int32_t channels, int32_t flags, const ts_camera* cam, const int32_t* tile_bins,
                  const int32_t* gaussian_ids_sorted, const float* splats, const float* background,
                  float* out_img, float* final_Ts, int32_t* final_index, uint8_t* clamp_mask,
                  void* stream) {
    return ts_raster_fwd_planes(channels, flags, cam, tile_bins, gaussian_ids_sorted, splats, background, out_img,
                                nullptr, final_Ts, final_index, clamp_mask, stream);
}

int ts_raster_fwd_planes(int32_t channels, int32_t flags, const ts_camera* cam, const int32_t* tile_bins,
                         const int32_t* gaussian_ids_sorted, const float* splats, const float* background,
                         float* out_img, float* out_depth, float* final_Ts, int32_t* final_index,
                         uint8_t* clamp_mask, void* stream) {
    if (!cam || (channels != 3 && channels != 4) || (out_depth && channels != 4)) return TS_E_BADARG;
    const bool narrow = cam->wide_tiles != 0 && (flags & TS_RASTER_NARROW_WAVES) != 0;   // 16x16 waves, wide lists
    const int nt = narrow ? cam->tile_rows * cam->tile_bounds_x : ts_num_tiles(cam);
    if (nt <= 0) return 0;
    if (!tile_bins || !background || !out_img || (!final_Ts != !final_index)) return TS_E_BADARG;
    bool split = (flags & TS_RASTER_SPLIT_BLOCKS) != 0;
    const bool wide = cam->wide_tiles != 0 && !narrow;
    ts_camera kcam = *cam;
    kcam.hints &= ~kHintCoopAll;
    if (split && TS_COOP && !wide && !narrow && (cam->hints & TS_HINT_COOP_SPLIT)) {
        split = false;                          // four waves per tile with SHARED staging instead (COOPERATIVE TILES)
        kcam.hints |= kHintCoopAll;
    }
    const int units = split ? (wide ? 8 : 4) * nt : nt;
    int grid = 8 * (((units + kWaves - 1) / kWaves + 7) / 8);      // see xcd_tile_group
    hipStream_t s = (hipStream_t)stream;
    const float4* sp = reinterpret_cast<const float4*>(splats);
    const int clamp = (flags & TS_RASTER_CLAMP_RGB) ? 1 : 0;
    // list segments (ts_camera.hints bits 8..11): a launch on 16x16 lists also keeps the boundary state of the cut tiles
    const bool segs = !wide && !narrow && final_Ts && TS_CAM_SEGS(*cam) > 1;
    if (!split && !wide && !narrow) grid = fwd_plan(nt, kcam.hints, segs, true).grid();      // COOPERATIVE TILES
#define TS_LAUNCH_FWD(C, S, X, L, G)                                                               \
    hipLaunchKernelGGL((raster_fwd_kernel<C, S, X, L, false, G>), dim3(grid), dim3(kThreads), 0, s, kcam, nt, \
                       tile_bins, gaussian_ids_sorted, (const int*)nullptr, (const float*)nullptr,   \
                       (int*)nullptr, sp, background, out_img, out_depth, final_Ts,                  \
                       final_index, clamp, clamp ? clamp_mask : nullptr)
#define TS_LAUNCH_FWD_X(C, S)                                                                      \
    do {                                                                                           \
        if (wide) TS_LAUNCH_FWD(C, S, 4, false, false);                                            \
        else if (narrow) TS_LAUNCH_FWD(C, S, 2, true, false);                                      \
        else if (segs) TS_LAUNCH_FWD(C, S, 2, false, true);                                        \
        else TS_LAUNCH_FWD(C, S, 2, false, false);                                                 \
    } while (0)
    if (channels == 3) { if (split) TS_LAUNCH_FWD_X(3, true); else TS_LAUNCH_FWD_X(3, false); }
    else { if (split) TS_LAUNCH_FWD_X(4, true); else TS_LAUNCH_FWD_X(4, false); }
#undef TS_LAUNCH_FWD_X
#undef TS_LAUNCH_FWD
    return launch_status();
}

int ts_raster_fwd_sort(int32_t channels, int32_t flags, const ts_camera* cam, const int32_t* tile_bins,
                       const int32_t* bucket_ids, const float* depths, int32_t* gaussian_ids_sorted,
                       const float* splats, const float* background, float* out_img, float* out_depth,
                       float* final_Ts, int32_t* final_index, uint8_t* clamp_mask, void* stream) {
    if (!cam || (channels != 3 && channels != 4) || (out_depth && channels != 4)) return TS_E_BADARG;
    // 16x16 lists, one wave (or, split, one workgroup) per tile: the mappings in which a list has one owner
    if (cam->wide_tiles != 0 || (flags & TS_RASTER_NARROW_WAVES)) return TS_E_BADARG;
    const int nt = ts_num_tiles(cam);
    if (nt <= 0) return 0;
    if (!tile_bins || !background || !out_img || (!final_Ts != !final_index) || !bucket_ids || !depths ||
        !gaussian_ids_sorted)
        return TS_E_BADARG;
    bool split = (flags & TS_RASTER_SPLIT_BLOCKS) != 0;
    ts_camera kcam = *cam;
    kcam.hints &= ~kHintCoopAll;
    if (split && TS_COOP && (cam->hints & TS_HINT_COOP_SPLIT)) {          // see ts_raster_fwd_planes
        split = false;
        kcam.hints |= kHintCoopAll;
    }
    const int units = split ? 4 * nt : nt;
    int grid = 8 * (((units + kWaves - 1) / kWaves + 7) / 8);      // see xcd_tile_group
    hipStream_t s = (hipStream_t)stream;
    const float4* sp = reinterpret_cast<const float4*>(splats);
    const int clamp = (flags & TS_RASTER_CLAMP_RGB) ? 1 : 0;
    const bool segs = final_Ts && TS_CAM_SEGS(*cam) > 1;                // list segments: see ts_raster_fwd_planes
    if (!split) grid = fwd_plan(nt, kcam.hints, segs, true).grid();     // COOPERATIVE TILES
#define TS_LAUNCH_FWD_SORT(C, S, G)                                                                            \
    hipLaunchKernelGGL((raster_fwd_kernel<C, S, 2, false, true, G>), dim3(grid), dim3(kThreads), 0, s, kcam, nt, \
                       tile_bins, (const int*)nullptr, bucket_ids, depths, gaussian_ids_sorted, sp, background, \
                       out_img, out_depth, final_Ts, final_index, clamp, clamp ? clamp_mask : nullptr)
    if (channels == 3) {
        if (split && segs) TS_LAUNCH_FWD_SORT(3, true, true);
        else if (split) TS_LAUNCH_FWD_SORT(3, true, false);
        else if (segs) TS_LAUNCH_FWD_SORT(3, false, true);
        else TS_LAUNCH_FWD_SORT(3, false, false);
    } else {
        if (split && segs) TS_LAUNCH_FWD_SORT(4, true, true);
        else if (split) TS_LAUNCH_FWD_SORT(4, true, false);
        else if (segs) TS_LAUNCH_FWD_SORT(4, false, true);
        else TS_LAUNCH_FWD_SORT(4, false, false);
    }
#undef TS_LAUNCH_FWD_SORT
    return launch_status();
}

int ts_raster_bwd(int32_t channels, int32_t flags, int64_t num_intersects, const ts_camera* cam,
                  const int32_t* tile_bins, const int32_t* gaussian_ids_sorted, const float* splats,
                  const float* background, const float* final_Ts, const int32_t* final_index,
                  const float* v_out_img, const float* v_out_alpha, const uint8_t* clamp_mask,
                  float* partials, uint8_t* row_flags, void* stream) {
    return ts_raster_bwd_planes(channels, flags, num_intersects, cam, tile_bins, gaussian_ids_sorted, splats, background,
                                final_Ts, final_index, v_out_img, nullptr, 0, v_out_alpha, clamp_mask, partials,
                                row_flags, stream);
}

int ts_raster_bwd_planes(int32_t channels, int32_t flags, int64_t num_intersects, const ts_camera* cam,
                         const int32_t* tile_bins, const int32_t* gaussian_ids_sorted, const float* splats,
                         const float* background, const float* final_Ts, const int32_t* final_index,
                         const float* v_out_img, const float* v_out_depth, int32_t planes,
                         const float* v_out_alpha, const uint8_t* clamp_mask,
                         float* partials, uint8_t* row_flags, void* stream) {
    if (!cam || (channels != 3 && channels != 4) || num_intersects < 0) return TS_E_BADARG;
    if (planes ? channels != 4 : v_out_depth != nullptr) return TS_E_BADARG;
    const bool narrow = cam->wide_tiles != 0 && (flags & TS_RASTER_NARROW_WAVES) != 0;
    const int nt = narrow ? cam->tile_rows * cam->tile_bounds_x : ts_num_tiles(cam);
    if (nt <= 0 || num_intersects == 0) return 0;
    if (!tile_bins || !gaussian_ids_sorted || !splats || !background || !final_Ts || !final_index ||
        (!planes && !v_out_img) || !partials || !row_flags)
        return TS_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const bool split = (flags & TS_RASTER_SPLIT_BLOCKS) != 0;
    const bool wide = cam->wide_tiles != 0 && !narrow;
    // row flags: legacy = zero the array, rows written in this pass get 1; with TS_RASTER_FLAG_GEN(g) the caller keeps
    // ONE flag array alive across passes and hands every pass a fresh value g in 1..255 (zeroing the array itself
    // when the values wrap) - no 1-byte-per-pair memset per frame
    const int gen = (flags >> 8) & 0xff;
    if (gen == 0) {
        hipError_t e = hipMemsetAsync(row_flags, 0, (size_t)num_intersects * (split ? 4 : 1), s);
        if (e != hipSuccess) return (int)e;
    }
    const long long isects_tagged = (long long)(((unsigned long long)(gen ? gen : 1) << 56) |
                                                ((unsigned long long)num_intersects & 0x00ffffffffffffffull));
    if (TS_CAM_SEGS(*cam) < 0 || TS_CAM_SEGS(*cam) > kSegMax) return TS_E_BADARG;
    const int segs = (!split && !wide && !narrow && TS_CAM_SEGS(*cam) > 1) ? TS_CAM_SEGS(*cam) : 1;
    int grid;
    if (segs > 1) {                       // band by band, whole tiles first (see raster_bwd_kernel)
        const CutTiles m = cut_tiles(nt, cam->hints);
        grid = 8 * ((m.items(segs) + kBwdWaves - 1) / kBwdWaves);
    } else {
        const int units = split ? (wide ? 8 : 4) * nt : nt;
        grid = 8 * (((units + kBwdWaves - 1) / kBwdWaves + 7) / 8);            // see xcd_tile_group
    }
    const float4* sp = reinterpret_cast<const float4*>(splats);
#define TS_LAUNCH_BWD(C, S, X, L)                                                                  \
    hipLaunchKernelGGL((raster_bwd_kernel<C, S, X, L>), dim3(grid), dim3(64 * kBwdWaves), 0, s, *cam, nt, \
                       isects_tagged, tile_bins, gaussian_ids_sorted, sp, background,              \
                       final_Ts, final_index, v_out_img, v_out_depth, planes ? 1 : 0, v_out_alpha,  \
                       clamp_mask, partials, row_flags)
#define TS_LAUNCH_BWD_X(C, S)                                                                      \
    do {                                                                                           \
        if (wide) TS_LAUNCH_BWD(C, S, 4, false);                                                   \
        else if (narrow) TS_LAUNCH_BWD(C, S, 2, true);                                             \
        else TS_LAUNCH_BWD(C, S, 2, false);                                                        \
    } while (0)
    if (channels == 3) { if (split) TS_LAUNCH_BWD_X(3, true); else TS_LAUNCH_BWD_X(3, false); }
    else { if (split) TS_LAUNCH_BWD_X(4, true); else TS_LAUNCH_BWD_X(4, false); }
#undef TS_LAUNCH_BWD_X
#undef TS_LAUNCH_BWD
    return launch_status();
}

int ts_reduce_partials(int32_t n, int32_t channels, int32_t flags, const int32_t* num_tiles_hit,
                       const int32_t* cum_tiles_hit, const float* partials,
                       const uint8_t* row_flags, const float* splats, float* v_xy, float* v_conic,
                       float* v_colors, float* v_opacity, float* v_depth, const uint8_t* color_mask,
                       void* stream) {
    if (n < 0 || (channels != 3 && channels != 4)) return TS_E_BADARG;
    if (n == 0) return 0;
    if (!num_tiles_hit || !cum_tiles_hit || !row_flags || !splats || !v_xy || !v_conic || !v_colors || !v_opacity)
        return TS_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const float4* pr = reinterpret_cast<const float4*>(partials);
    const float4* sp = reinterpret_cast<const float4*>(splats);
    const int grid = (int)(((long long)n * TS_REDUCE_LANES + 255) / 256);
    if (channels == 3)
        hipLaunchKernelGGL(reduce_partials_kernel<3>, dim3(grid), dim3(256), 0, s, n, (int)flags, num_tiles_hit,
                           cum_tiles_hit, pr, row_flags, sp, v_xy, v_conic, v_colors, v_opacity, v_depth, color_mask,
                           (float4*)nullptr);
    else
        hipLaunchKernelGGL(reduce_partials_kernel<4>, dim3(grid), dim3(256), 0, s, n, (int)flags, num_tiles_hit,
                           cum_tiles_hit, pr, row_flags, sp, v_xy, v_conic, v_colors, v_opacity, v_depth, color_mask,
                           (float4*)nullptr);
    return launch_status();
}

int ts_reduce_partials_rows(int32_t n, int32_t channels, int32_t flags, const int32_t* num_tiles_hit,
                            const int32_t* cum_tiles_hit, const float* partials, const uint8_t* row_flags,
                            const float* splats, float* grad_rows, void* stream) {
    if (n < 0 || (channels != 3 && channels != 4)) return TS_E_BADARG;
    if (n == 0) return 0;
    if (!num_tiles_hit || !cum_tiles_hit || !row_flags || !splats || !grad_rows) return TS_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const float4* pr = reinterpret_cast<const float4*>(partials);
    const float4* sp = reinterpret_cast<const float4*>(splats);
    float4* gr = reinterpret_cast<float4*>(grad_rows);
    const int grid = (int)(((long long)n * TS_REDUCE_LANES + 255) / 256);
    if (channels == 3)
        hipLaunchKernelGGL(reduce_partials_kernel<3>, dim3(grid), dim3(256), 0, s, n, (int)flags, num_tiles_hit,
                           cum_tiles_hit, pr, row_flags, sp, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                           (float*)nullptr, (float*)nullptr, (const unsigned char*)nullptr, gr);
    else
        hipLaunchKernelGGL(reduce_partials_kernel<4>, dim3(grid), dim3(256), 0, s, n, (int)flags, num_tiles_hit,
                           cum_tiles_hit, pr, row_flags, sp, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                           (float*)nullptr, (float*)nullptr, (const unsigned char*)nullptr, gr);
    return launch_status();
}

}  // extern "C"
