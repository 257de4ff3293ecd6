/*
 * tinysplat_hip.h - C ABI of libtinysplat_hip.so (HIP, gfx950 / MI355X).
 *
 * This is the drop-in boundary below tinysplat's render adapter.  tinysplat reaches its rasterizer
 * through three Python callables imported from the third-party gsplat package
 * (/root/reference/tinysplat/splatting/rasterize.py:3-4):
 *
 *     project_gaussians(...)      rasterize.py:32   (13 positional args built at rasterize.py:64-73)
 *     spherical_harmonics(...)    rasterize.py:38   (3 args built at rasterize.py:75-81)
 *     rasterize_gaussians(...)    rasterize.py:44,50 (10 args built at rasterize.py:83-86)
 *
 * gsplat binds those callables to a C++/CUDA extension; the entry points below are what a binding
 * for the same three callables (forward and backward) binds on MI355X.  Each group is labelled with
 * the callable it implements.  tinysplat_amd/_lib.py is the ctypes binding, tinysplat_amd/ops.py the
 * torch.autograd.Function layer that restores the exact Python signatures.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - all arrays are dense, row-major, float32 / int32 / uint64 as typed;
 *   - `stream` is a hipStream_t passed as void*; every entry only enqueues work on it, never
 *     synchronises, never allocates, never frees; workspaces are caller-provided;
 *   - return value: 0 on success, a negative TS_E_* code for an argument error, otherwise the
 *     positive hipError_t of the failed launch;
 *   - tile = 16x16 pixels (rasterize.py:19-20).  `tile_row0/tile_rows` select a stripe of tile rows
 *     [tile_row0, tile_row0 + tile_rows) for multi-GPU tile-stripe sharding; a single GPU passes
 *     (0, tile_bounds_y).  Pixel coordinates stay global; image buffers hold only the stripe's rows.
 */
#ifndef TINYSPLAT_HIP_H
#define TINYSPLAT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TS_ABI_VERSION 8

#define TS_E_BADARG (-1)  /* null pointer / negative size / unsupported channel count */
#define TS_E_DEGREE (-2)  /* SH degree out of range or exceeds stored coefficients */

#define TS_TILE 16
#define TS_SPLAT_RECORD_FLOATS 12   /* packed per-Gaussian record, see ts_pack_splats */
#define TS_PARTIAL_ROW_FLOATS 12    /* per-(tile,Gaussian) gradient row, see ts_raster_bwd */

int ts_abi_version(void);

/* ---- camera block ---------------------------------------------------------------------------
 * Scalars of rasterize.py:64-73 (fx, fy, cx, cy, img_height, img_width, tile_bounds, glob_scale)
 * plus the near-plane threshold of gsplat's project_gaussians (clip_thresh, default 0.01). */
typedef struct ts_camera {
    float fx, fy, cx, cy;
    int32_t img_width, img_height;
    int32_t tile_bounds_x, tile_bounds_y;
    int32_t tile_row0, tile_rows;
    float glob_scale;
    float clip_thresh;
    /* Tile shape of the LISTS (binning, sorting, compositing): 0 = gsplat's 16x16 tiles; 1 = wide 32x16
     * tiles, each the union of two horizontally adjacent 16x16 tiles (column pairs 2j, 2j+1; the last one
     * may be a single column).  Projection, num_tiles_hit / cum_tiles_hit and tile_row0 / tile_rows always
     * count 16x16 tiles; images and gradients do not depend on the shape (a Gaussian is composited only
     * into the 16x16 tiles of its tile box either way), only the number of list entries does. */
    int32_t wide_tiles;
    /* bits 0..7: performance hints (formerly `reserved`, 0 = none); results never depend on them.
     * TS_HINT_BALANCED_WALK: a Gaussian covers many tiles (>= ~10 bounding-box tiles on average): ts_bin_scatter
     * expands (Gaussian, tile row) items over the lanes instead of looping per Gaussian.
     * bits 8..11: LIST SEGMENTS S of the compositing passes (TS_CAM_LIST_SEGMENTS(S), 0 / 1 = off, S <= 8; 16x16
     * lists only).  The backward pass of a tile is a chain over its sorted list; with S > 1 ts_raster_fwd* additionally
     * stores, for every CUT tile, the per-pixel state at up to S - 1 boundaries of its list (lists of >= 65 entries)
     * behind final_Ts - which must then hold ts_final_floats(cam, channels) floats - and ts_raster_bwd* WITHOUT
     * TS_RASTER_SPLIT_BLOCKS replays a cut tile's list as up to S independent one-wave work items (one row per pair,
     * the plain ts_reduce_partials).  Same image; gradients equal to the uncut pass's up to rounding (a segment
     * starts from the forward pass's own transmittance).  Both passes must see the same S and W16.  With
     * TS_RASTER_SPLIT_BLOCKS, wide tiles or narrow waves ts_raster_bwd* ignores the field.
     * bits 12..15: W16 (TS_CAM_WHOLE_TILES(W16), 0..15) - which tiles are cut: 0 = all of them (launches of fewer
     * tiles than the GPU has SIMDs: instead of TS_RASTER_SPLIT_BLOCKS in the backward pass); 1..15 = a HYBRID
     * launch for full frames: the tiles are handed out in eight bands (one per XCD) and the first W16/16 of every
     * band stay whole, only the tiles dispatched last are cut, so that their small work items fill the wave slots
     * the whole tiles leave behind at the end of the launch (ts_cut_tiles tells which).
     * bits 16..19: C16 (TS_CAM_COOP_TILES(C16), 0..15; a pure performance hint of ts_raster_fwd*, one wave per 16x16
     * tile on 16x16 lists) - COOPERATIVE TILES: the first C16/16 of every band, which the forward launch hands out
     * last, are composited by a workgroup of four waves each (shared staging and sort, one 8x8 block per wave), so
     * that the forward launch's tail is filled with items a quarter as long.  Never a cut tile (C16 is capped by
     * W16; with S > 1 and W16 = 0 it is ignored).  Image, final_Ts, final_index: bit for bit the same.
     * TS_HINT_COOP_SPLIT (bit 20; a pure performance hint of ts_raster_fwd*): a launch that asks for
     * TS_RASTER_SPLIT_BLOCKS on 16x16 lists composites every tile as a cooperative workgroup instead - four waves per
     * tile as the split mapping has, but each list is staged and sorted ONCE by the four of them instead of walked by
     * each.  With S > 1 the boundary records are the ones the split launch would leave (W16 = 0: every tile; W16 > 0:
     * the cut tiles are the cooperative ones and the others one wave each).  Same bits in every output. */
    int32_t hints;
} ts_camera;
#define TS_HINT_BALANCED_WALK 1
#define TS_HINT_COOP_SPLIT (1 << 20)
#define TS_CAM_LIST_SEGMENTS(s) (((s) & 15) << 8)
#define TS_CAM_WHOLE_TILES(w16) (((w16) & 15) << 12)
#define TS_CAM_COOP_TILES(c16) (((c16) & 15) << 16)
/* floats final_Ts must hold: the rows*W transmittances, then (S > 1, 16x16 lists) one checkpoint block of
 * S records of (1+channels) 256 floats per cut tile, from a 64-float aligned offset; < 0: bad argument */
int64_t ts_final_floats(const ts_camera* cam_host, int32_t channels);
/* -> number of checkpoint blocks (>= cut tiles) of a launch under cam's S / W16; *band = tiles per band (tile t lies
 * in band t / band), *whole = the first `whole` tiles of every band are composited whole (either may be NULL) */
int32_t ts_cut_tiles(const ts_camera* cam_host, int32_t* band, int32_t* whole);
/* tiles (= lists) of a launch: tile_rows * tile_bounds_x, or tile_rows * ceil(tile_bounds_x / 2) when wide */
int32_t ts_num_tiles(const ts_camera* cam_host);

/* ============================ project_gaussians (rasterize.py:32) ============================ */

/* flags: fold the adapter's argument preparation (rasterize.py:72-73) into the kernels */
#define TS_PROJECT_LOG_SCALES 1  /* `scales` holds log-scales: the kernel uses exp(scales)          */
#define TS_PROJECT_RAW_QUATS 2   /* `quats` is unnormalised: the kernel uses quats / |quats|;        */
                                 /* backward then returns gradients w.r.t. the raw tensors           */

/* Forward.  viewmat: 12 floats (rows of the 3x4 view matrix, rasterize.py:73 passes
 * view_matrix[:3,:]); projmat: 16 floats (P @ V).  Outputs are fully written for every Gaussian;
 * culled ones (z <= clip, singular cov2d, no tile hit) get zeros.  cov3d may be NULL (the adapter
 * discards it, rasterize.py:32; the forward-only viewer path, viewer.py:89-93, passes NULL). */
int ts_project_fwd(int32_t n, const float* means3d, const float* scales, const float* quats,
                   const float* viewmat, const float* projmat, const ts_camera* cam_host,
                   int32_t flags, float* xys, float* depths, int32_t* radii, float* conics,
                   int32_t* num_tiles_hit, float* cov3d, void* stream);

/* Backward.  v_conic uses the true-partial convention for the off-diagonal entry.  v_cov3d may be
 * NULL (tinysplat discards cov3d, rasterize.py:32); v_depth may be NULL (no gradient reaches depths).  Gaussians with radii == 0 receive zeros. */
int ts_project_bwd(int32_t n, const float* means3d, const float* scales, const float* quats,
                   const float* viewmat, const float* projmat, const ts_camera* cam_host,
                   int32_t flags, const int32_t* radii, const float* v_xy, const float* v_depth,
                   const float* v_conic, const float* v_cov3d,
                   float* v_means3d, float* v_scales, float* v_quats, void* stream);

/* ========================= gsplat.sh.spherical_harmonics (rasterize.py:38) ==================== */

/* colors[n,3] = sum over bands <= degrees_to_use of Y_k(normalize(viewdirs)) * coeffs[n,k,:].
 * num_bases = K of the stored coefficients (1,4,9,16,25). */
int ts_sh_fwd(int32_t n, int32_t degrees_to_use, int32_t num_bases, const float* viewdirs,
              const float* coeffs, float* colors, void* stream);

/* v_coeffs[n,K,3]: Y_k * v_colors for active bands, zero for inactive ones (fully written). */
int ts_sh_bwd(int32_t n, int32_t degrees_to_use, int32_t num_bases, const float* viewdirs,
              const float* v_colors, float* v_coeffs, void* stream);

/* Fused colour stage of the render adapter = rasterize.py:75-81 (view directions
 * normalize(means - origin), origin = view_matrix[:3,3]; coefficients given as the two parameter
 * tensors colors_dc[n,3] and colors_rest[n,K-1,3] instead of their torch.cat) + rasterize.py:38-39
 * (SH evaluation, clamp(rgb + 0.5, min=0)).  clamp_mask[n]: bit c set where channel c passes
 * gradient (NULL in forward-only use).  origin: 3 device floats.  colors_rest may be NULL when
 * num_bases == 1. */
int ts_sh_colors_fwd(int32_t n, int32_t degrees_to_use, int32_t num_bases, const float* means3d,
                     const float* origin, const float* colors_dc, const float* colors_rest,
                     float* colors, uint8_t* clamp_mask, const int32_t* live, void* stream);
/* live: NULL, or num_tiles_hit of a tile-row stripe - only Gaussians with live[i] != 0 are evaluated (each
 * reads its own coefficient row) and the colours / masks of the others are left unwritten: the colour stage
 * of one rank of a multi-GPU frame touches the 1/G of the Gaussians that its stripe lists.
 * ts_sh_colors_bwd: clamp_mask may be NULL (all channels pass) when the mask was already applied by
 * ts_reduce_partials(color_mask) - the multi-GPU order, where the 2-D gradients are summed over ranks
 * between the two and a rank holds masks for its own stripe's Gaussians only. */
int ts_sh_colors_bwd(int32_t n, int32_t degrees_to_use, int32_t num_bases, const float* means3d,
                     const float* origin, const uint8_t* clamp_mask, const float* v_colors,
                     float* v_colors_dc, float* v_colors_rest, void* stream);
/* ts_sh_colors_fwd + ts_pack_splats in one launch for the RGB / RGB + depth frame: the colour stage
 * writes the packed record of every listed Gaussian itself (arguments as for ts_pack_splats; the colours
 * stay in registers, no colors[n,3] array exists).  channels == 4 takes channel 3 from depths[]. */
int ts_colors_pack_fwd(int32_t n, int32_t degrees_to_use, int32_t num_bases, const float* means3d,
                       const float* origin, const float* colors_dc, const float* colors_rest,
                       uint8_t* clamp_mask, const int32_t* live, int32_t channels, int32_t flags,
                       const float* xys, const int32_t* radii, const float* conics, const float* opacity,
                       const int32_t* cum_tiles_hit, const ts_camera* cam_host, const float* depths,
                       float* splats, void* stream);

/* ========================= rasterize_gaussians (rasterize.py:44,50) =========================== */
/* Stage order: ts_scan_tiles -> (read total) -> ts_bin_count -> ts_tile_offsets -> ts_bin_scatter
 * -> ts_sort_tiles -> ts_pack_splats -> ts_raster_fwd ; backward: ts_raster_bwd -> ts_reduce_partials */

/* Inclusive int32 prefix sum of num_tiles_hit -> cum_tiles_hit[n]; the grand total (number of
 * tile/Gaussian intersections I) is cum_tiles_hit[n-1].  scan_ws: >= ts_scan_ws_ints(n) int32.
 * total_out: NULL, or a DEVICE-VISIBLE address (the device pointer of mapped pinned host memory, see
 * hipHostGetDevicePointer) that receives the grand total with a system-scope store from the scan itself:
 * the host can poll it there instead of queueing a 4-byte copy behind the scan. */
int64_t ts_scan_ws_ints(int32_t n);
int ts_scan_tiles(int32_t n, const int32_t* num_tiles_hit, int32_t* cum_tiles_hit,
                  int32_t* scan_ws, int32_t* total_out, void* stream);

/* Tile bucketing without global atomics.  The Gaussians are cut into B = ts_bin_chunks(n) contiguous
 * chunks; bin_ws (>= ts_bin_ws_ints(n, num_tiles) int32, num_tiles = tile_rows * tile_bounds_x) holds
 * the B x num_tiles count matrix, num_tiles + 1 tile starts and one spare word.  Stripe-local tile index
 * t = (ty - tile_row0) * tile_bounds_x + tx. */
int64_t ts_bin_ws_ints(int32_t n, int32_t num_tiles);

/* bin_ws[b][t] = number of Gaussians of chunk b whose tile rectangle covers tile t (LDS histograms).
 * splats: NULL -> gsplat's bounding-box lists (what the bit-exact binning checks compare).
 * Non-NULL (the packed records of ts_pack_splats) -> TIGHT lists: a (Gaussian, tile) pair is dropped
 * when the record proves that no pixel of the tile can reach alpha >= 1/255; such pairs contribute
 * exactly nothing, so image and gradients are bit-identical while ~1/3 of the pairs never reach the
 * scatter, the sort and the compositing kernels.  ts_bin_scatter must be given the same pointer.
 * Buffers stay sized by the bounding-box total I = cum_tiles_hit[n-1]. */
int ts_bin_count(int32_t n, const float* xys, const int32_t* radii, const float* splats,
                 const ts_camera* cam_host, int32_t* bin_ws, void* stream);

/* Turns the counts into bases in place (exclusive scan down the chunk axis, then over tiles) and
 * writes tile_bins[t] = {start, end} (both 0 for an empty tile).
 * Capacity guard (cum_tiles_hit non-NULL and capacity >= 0): a caller that sized bucket_ids / gaussian_ids_sorted /
 * partials from an ESTIMATE instead of waiting for the count passes that size; if the frame's total
 * cum_tiles_hit[n-1] exceeds it, every list is left empty and ts_bin_scatter returns without writing, so nothing
 * is touched beyond the buffers - the caller reads the count later, sees the same excess and repeats the frame
 * with exact sizes.  NULL / -1: no guard. */
int ts_tile_offsets(int32_t n, int32_t num_tiles, int32_t* bin_ws, int32_t* tile_bins,
                    const int32_t* cum_tiles_hit, int64_t capacity, void* stream);
/* ... and the length of the frame's LONGEST list stored to *longest_list (device-visible memory - e.g. the word behind
 * the caller's mapped pinned count word; NULL: ts_tile_offsets): a scene statistic for the caller's launch policy
 * (frame.py: list shape and hybrid shares), read a frame later, never waited for. */
int ts_tile_offsets_stats(int32_t n, int32_t num_tiles, int32_t* bin_ws, int32_t* tile_bins,
                          const int32_t* cum_tiles_hit, int64_t capacity, int32_t* longest_list, void* stream);

/* Writes the id of every Gaussian into the bucket of each tile its rectangle covers (bucket_ids[I];
 * order inside a bucket is arbitrary until ts_sort_tiles).  scratch: NULL, or I more int32 (the
 * gaussian_ids_sorted buffer may be passed: it is dead until ts_sort_tiles) - the ids then reach their buckets in
 * two coalesced hops (tile group, then tile) instead of one 4-byte store per cache line; same buckets. */
int ts_bin_scatter(int32_t n, const float* xys, const int32_t* radii, const float* splats,
                   const ts_camera* cam_host, const int32_t* bin_ws, int32_t* bucket_ids, int32_t* scratch,
                   void* stream);

/* Sorts every tile bucket ascending by (depth bits, gaussian id) - i.e. the order of a stable sort
 * of (tile<<32 | depth-bits) keys emitted Gaussian-major - reading bucket_ids[I] and depths[n] and
 * writing gaussian_ids_sorted[I] (a different buffer).  sort_ws: >= num_tiles + 1 int32 of scratch
 * (bin_ws may be passed: its contents are dead once ts_bin_scatter has run). */
int ts_sort_tiles(int32_t num_tiles, const int32_t* tile_bins, const float* depths,
                  const int32_t* bucket_ids, int32_t* gaussian_ids_sorted, int32_t* sort_ws,
                  int32_t* zeroed_counter, void* stream);
/* ABI 5: ts_sort_tiles for the tiles of MORE than 1024 entries only - the companion of ts_raster_fwd_sort, which
 * sorts the shorter lists itself (one wave per tile) right before it composites them. */
int ts_sort_tiles_above(int32_t num_tiles, const int32_t* tile_bins, const float* depths,
                        const int32_t* bucket_ids, int32_t* gaussian_ids_sorted, int32_t* sort_ws,
                        int32_t* zeroed_counter, void* stream);
/* sort_ws, zeroed_counter: DEPRECATED and ignored (pass NULL) - every tile has its own workgroup now, there is
 * no queue of oversized tiles.  ts_sort_tiles still uses them (its zeroed_counter: NULL, or a device word
 * known to be zero; ts_tile_offsets zeroes bin_ws[ts_bin_ws_ints(n, num_tiles) - 1] for this purpose). */

#define TS_RASTER_LOGIT_OPACITY 1 /* `opacity` holds logits: sigmoid (rasterize.py:86) is applied while */
                                  /* packing, and ts_reduce_partials returns the gradient w.r.t. logits */

/* Packs the per-Gaussian operands of the compositing kernels into one 48-byte record:
 *   {x, y, opacity, conic.xx | conic.xy, conic.yy, c0, c1 | c2, c3, slot_base(int), bbox_w | bbox_minx << 16 (int)}
 * channels = 3 (colors[n,3]; c3 = 0) or 4 (colors[n,4]; or, with `depths` non-NULL, colors[n,3] and
 * c3 = depths[i]: the RGB + depth frame of rasterize.py:42-51 in one pass).  slot_base/bbox_w locate the
 * (tile,Gaussian) row of the backward partial buffer: slot = slot_base + ty*bbox_w + tx (16x16 tile column tx;
 * a wide tile uses the column of its left half, or of its right half where the box starts there). */
int ts_pack_splats(int32_t n, int32_t channels, int32_t flags, const float* xys, const int32_t* radii,
                   const float* conics, const float* colors, const float* opacity,
                   const int32_t* cum_tiles_hit, const ts_camera* cam_host, const float* depths,
                   float* splats, void* stream);

/* Front-to-back compositing.  out_img[rows,W,channels], final_Ts[rows,W], final_index[rows,W]
 * where rows = min(16*tile_rows, H - 16*tile_row0).  background: `channels` floats.  final_Ts and
 * final_index exist for the backward pass; both may be NULL together (forward-only rendering). */
/* flags: TS_RASTER_CLAMP_RGB folds the adapter's clamp(rgb, max=1) (rasterize.py:45) into the store of
 * channels 0..2; clamp_mask[rows,W] (bytes, may be NULL when no backward follows) then receives bit c
 * where channel c passes gradient (value <= 1, torch's rule) for ts_raster_bwd. */
#define TS_RASTER_CLAMP_RGB 2
/* TS_RASTER_SPLIT_BLOCKS: four waves per tile, one per 8x8 block (same results per pixel).  For launches
 * with fewer tiles than the GPU has SIMDs (a tile-row stripe of a multi-GPU frame, a small image).  In
 * ts_raster_bwd / ts_reduce_partials the flag makes every (tile, Gaussian) own FOUR partial rows (slot
 * 4 s + block): partials must hold 4 * num_intersects rows and row_flags 4 * num_intersects bytes. */
#define TS_RASTER_SPLIT_BLOCKS 4
/* with ts_camera.wide_tiles: keep one wave per 16x16 tile; each walks the list of the wide tile it lies in */
#define TS_RASTER_NARROW_WAVES 8
/* ts_raster_bwd / ts_reduce_partials(_rows): ROW-FLAG GENERATION.  Without it (g = 0) ts_raster_bwd zeroes
 * row_flags and marks the rows it writes with 1.  A caller that keeps ONE row_flags array alive across passes
 * (all zero at first) can pass a fresh g in 1..255 to both entries of a pass instead: rows are marked with g, only
 * rows marked g count, and nothing is zeroed (the caller zeroes the array itself before reusing a value). */
#define TS_RASTER_FLAG_GEN(g) (((g) & 0xff) << 8)
int ts_raster_fwd(int32_t channels, int32_t flags, const ts_camera* cam_host, const int32_t* tile_bins,
                  const int32_t* gaussian_ids_sorted, const float* splats, const float* background,
                  float* out_img, float* final_Ts, int32_t* final_index, uint8_t* clamp_mask,
                  void* stream);
/* The RGB + depth image as TWO planes (channels = 4 with out_depth != NULL): out_img[P,3] and out_depth[P] - what
 * the adapter hands out as `rgb` and `extras["depth"]` (rasterize.py:45, :51), both contiguous, so that losses on
 * them and their gradients need no slicing of an interleaved image.  out_depth = NULL: ts_raster_fwd. */
int ts_raster_fwd_planes(int32_t channels, int32_t flags, const ts_camera* cam_host, const int32_t* tile_bins,
                         const int32_t* gaussian_ids_sorted, const float* splats, const float* background,
                         float* out_img, float* out_depth, float* final_Ts, int32_t* final_index,
                         uint8_t* clamp_mask, void* stream);

/* ABI 5: ts_raster_fwd_planes with the per-tile sort of the lists of <= 1024 entries inside (bucket_ids[I] ->
 * gaussian_ids_sorted[I], which ts_raster_bwd reads later; ts_sort_tiles_above must have run for the longer lists).
 * Only on 16x16 lists (cam->wide_tiles = 0, no TS_RASTER_NARROW_WAVES: TS_E_BADARG otherwise), with one wave per tile
 * or - TS_RASTER_SPLIT_BLOCKS - one workgroup per tile (its first wave sorts).  Same image, same sorted lists; the latency-bound sort overlaps
 * the VALU-bound compositing of other tiles instead of running as a phase of its own. */
int ts_raster_fwd_sort(int32_t channels, int32_t flags, const ts_camera* cam_host, const int32_t* tile_bins,
                       const int32_t* bucket_ids, const float* depths, int32_t* gaussian_ids_sorted,
                       const float* splats, const float* background, float* out_img, float* out_depth,
                       float* final_Ts, int32_t* final_index, uint8_t* clamp_mask, void* stream);

/* Back-to-front replay.  Writes one TS_PARTIAL_ROW_FLOATS row of raw per-tile sums per contributing
 * (tile,Gaussian), with v_s = dL/dsigma of a pixel and d = xy - pixel:
 *   {S v_s, S v_s dx, S v_s dy, S v_s dx^2, S v_s dx dy, S v_s dy^2, v_c0, v_c1, v_c2, v_c3, -, -}
 * row_flags[I] (bytes) is zeroed by this call and set to 1 for every row written; rows whose flag
 * stays 0 keep stale contents and must be ignored (ts_reduce_partials does).  v_out_alpha may be NULL.
 * clamp_mask: NULL, or the mask ts_raster_fwd wrote under TS_RASTER_CLAMP_RGB (v_out_img is then the
 * gradient w.r.t. the clamped image and is zeroed where the clamp was active). */
int ts_raster_bwd(int32_t channels, int32_t flags, int64_t num_intersects, const ts_camera* cam_host,
                  const int32_t* tile_bins, const int32_t* gaussian_ids_sorted, const float* splats,
                  const float* background, const float* final_Ts, const int32_t* final_index,
                  const float* v_out_img, const float* v_out_alpha, const uint8_t* clamp_mask,
                  float* partials, uint8_t* row_flags, void* stream);
/* planes != 0 (channels = 4): the image gradient as two planes, v_out_img[P,3] and v_out_depth[P]; either may be
 * NULL (that output took no part in the loss: its gradient is zero).  planes = 0: ts_raster_bwd. */
int ts_raster_bwd_planes(int32_t channels, int32_t flags, int64_t num_intersects, const ts_camera* cam_host,
                         const int32_t* tile_bins, const int32_t* gaussian_ids_sorted, const float* splats,
                         const float* background, const float* final_Ts, const int32_t* final_index,
                         const float* v_out_img, const float* v_out_depth, int32_t planes,
                         const float* v_out_alpha, const uint8_t* clamp_mask,
                         float* partials, uint8_t* row_flags, void* stream);

/* Sums each Gaussian's flagged rows (a contiguous range of `partials`, fixed order => run-to-run
 * bit-reproducible gradients), applies the conic / opacity factors read from `splats`, and writes
 * v_xy[n,2], v_conic[n,3] (true partials), v_colors[n,channels], v_opacity[n]. */
int ts_reduce_partials(int32_t n, int32_t channels, int32_t flags, const int32_t* num_tiles_hit,
                       const int32_t* cum_tiles_hit, const float* partials,
                       const uint8_t* row_flags, const float* splats, float* v_xy, float* v_conic,
                       float* v_colors, float* v_opacity, float* v_depth, const uint8_t* color_mask,
                       void* stream);
/* v_depth: NULL, or (channels == 4) the gradient of channel 3 as its own array [n]; v_colors is then
 * [n,3] - the layout the RGB + depth frame hands to ts_sh_colors_bwd / ts_project_bwd.
 * color_mask: NULL, or the colour stage's clamp mask: v_colors[i][c] = 0 where bit c is clear. */

/* ============== whole-frame executor (the adapter's recipe, rasterize.py:26-62, in five calls) ======
 * The reference's GaussianRasterizer.__call__ enqueues ~30 small operations per frame from Python; on
 * scenes of 100 k Gaussians and on multi-GPU tile stripes that host time exceeds the GPU time.  These
 * entries enqueue the same kernels, in the same order, as the per-stage entries above - nothing else -
 * from native code: one ts_frame describes a frame (all buffers caller-allocated; device pointers
 * unless noted), and a frame costs five calls instead of thirty:
 *   ts_frame_fwd_project    project_fwd, scan_tiles; the scan stores the intersection count cum_tiles_hit[n-1]
 *                           into total_host (mapped pinned host memory; a 4-byte copy is queued instead when the
 *                           word is not device-visible) - the caller polls the word, or waits for an event it
 *                           records behind this call, only before step 3
 *   ts_frame_fwd_prepare    colors_pack_fwd (= sh_colors_fwd + pack_splats), bin_count, tile_offsets   (do not need the count)
 *   ts_frame_fwd_composite  bin_scatter, sort_tiles, raster_fwd   (needs bucket_ids / gaussian_ids_sorted
 *                           sized by the count; num_intersects must be set)
 *   ts_frame_bwd_composite  raster_bwd, reduce_partials -> the flat 2-D gradients v_xy | v_conic |
 *                           v_colors | [v_depth] | v_opacity (multi-GPU: all-reduce them after this call)
 *   ts_frame_bwd_params     sh_colors_bwd, project_bwd -> gradients of the six parameter tensors
 * flags: TS_FRAME_TIGHT (tight tile lists: drop (Gaussian, tile) pairs that cannot reach alpha >= 1/255),
 * TS_FRAME_SPLIT (TS_RASTER_SPLIT_BLOCKS for the compositing launches).  channels = 3 (RGB) or 4 (RGB +
 * depth composited in one pass).  Forward-only rendering: final_Ts = final_index = clamp_mask = sh_mask = NULL. */
#define TS_FRAME_TIGHT 1
#define TS_FRAME_SPLIT 2
#define TS_FRAME_NARROW_WAVES 8        /* TS_RASTER_NARROW_WAVES for the compositing launches (cam.wide_tiles) */
#define TS_FRAME_DIRECT_SCATTER 32     /* one-hop ts_bin_scatter (scratch = NULL): A/B timing */
#define TS_FRAME_PLANES 64             /* channels = 4: RGB and depth as two planes (out_img[P,3] + out_depth[P];
                                          v_out_img / v_out_depth, either may be NULL), see ts_raster_fwd_planes */
#define TS_FRAME_LIST_STATS 256        /* total_host points at (at least) TWO mapped words: ts_frame_fwd_prepare stores the
                                        * length of the frame's longest list into word 1 (ts_tile_offsets_stats) */
#define TS_FRAME_SEPARATE_SORT 128     /* ts_sort_tiles + ts_raster_fwd_planes even where ts_raster_fwd_sort applies: A/B */
#define TS_FRAME_STRIPE 16             /* one stripe of a multi-GPU frame: colour stage only for the Gaussians the
                                          stripe lists, clamp mask applied in reduce_partials (before the all-reduce) */
typedef struct ts_frame {
    int32_t n, num_bases, sh_degree, channels, flags;
    int32_t flag_gen;                         /* row-flag generation of the backward pass (0: zero row_flags) */
    ts_camera cam;
    /* parameters and camera (rasterize.py:64-86): log-scales, raw quaternions, opacity logits */
    const float *means, *scales, *quats, *opacities, *colors_dc, *colors_rest;
    const float *view34, *projview, *origin, *background;     /* background: `channels` floats */
    /* per-Gaussian outputs / intermediates */
    float *xys, *depths, *conics, *colors, *splats;
    int32_t *radii, *num_tiles_hit, *cum_tiles_hit;
    uint8_t* sh_mask;
    int32_t *scan_ws, *bin_ws, *tile_bins;
    int32_t* total_host;                      /* pinned HOST int32 */
    /* per-intersection buffers and the count they are sized by; capacity >= 0: the buffers were sized from an
     * estimate before the count was known (ts_tile_offsets' guard), -1: from the count itself */
    int64_t num_intersects;
    int64_t capacity;
    int32_t *bucket_ids, *gaussian_ids_sorted;
    /* image outputs */
    float *out_img, *final_Ts;
    int32_t* final_index;
    uint8_t* clamp_mask;
    /* backward */
    const float* v_out_img;
    float* partials;
    uint8_t* row_flags;
    float *v_xy, *v_conic, *v_colors, *v_depth, *v_opacity;
    float *v_means, *v_scales, *v_quats, *v_colors_dc, *v_colors_rest;
    /* TS_FRAME_PLANES: the depth plane of the image and of its gradient */
    float* out_depth;
    const float* v_out_depth;
} ts_frame;
int32_t ts_frame_struct_bytes(void);       /* sizeof(ts_frame): bindings check their mirror against it */
int ts_frame_fwd_project(const ts_frame* f, void* stream);
int ts_frame_fwd_prepare(const ts_frame* f, void* stream);
int ts_frame_fwd_composite(const ts_frame* f, void* stream);
int ts_frame_bwd_composite(const ts_frame* f, void* stream);
int ts_frame_bwd_params(const ts_frame* f, void* stream);

/* ============== Gaussian-sharded multi-GPU frame (csrc/shard.hip) =====================
 * One rank per GPU owns a contiguous range of Gaussians and one stripe of tile rows.  Per frame it projects and
 * colours its own Gaussians (ts_project_fwd / ts_colors_pack_fwd with the FULL-frame camera), routes a 64-byte
 * export record of each visible one to the rank(s) whose stripe its tile box reaches, and composites the
 * records it receives; backward returns one 48-byte gradient row per record to the owner.  The exchange itself
 * (two all_to_all calls) is the caller's: these entries produce and consume its buffers.
 *
 * Export record (TS_EXPORT_RECORD_FLOATS floats): {x, y, opacity, conic.xx | conic.xy, conic.yy, c0, c1 |
 * c2, c3, depth, radius (int) | global id (int), -, -, -}: the first ten floats are those of the packed
 * compositing record (ts_pack_splats).
 * Gradient row (TS_PARTIAL_ROW_FLOATS floats): {v_x, v_y, v_conic.xx, v_conic.xy | v_conic.yy, v_c0, v_c1, v_c2 |
 * v_c3 (depth channel), v_opacity (w.r.t. the sigmoid's output), -, -}. */
#define TS_MAX_RANKS 16
#define TS_EXPORT_RECORD_FLOATS 16
typedef struct ts_stripes {   /* rank d renders tile rows [row[d], row[d+1]) */
    int32_t num;
    int32_t row[TS_MAX_RANKS + 1];
} ts_stripes;
/* route_ws: >= ts_route_ws_ints(n, num_ranks) int32; written by ts_route_count, read by ts_route_pack and
 * ts_route_accumulate of the same frame.  counts (device, num_ranks int32) <- records per destination. */
int64_t ts_route_ws_ints(int32_t n, int32_t num_ranks);
int ts_route_count(int32_t n, const float* xys, const int32_t* radii, const ts_camera* cam_host,
                   const ts_stripes* stripes_host, int32_t* route_ws, int32_t* counts, void* stream);
/* PADDED groups: group_base_host[num_ranks + 1] (ascending, host memory, read during the call) fixes where every
 * destination's group starts in the send buffer and in the returned gradient rows - capacities the caller chose
 * BEFORE this frame's counts exist (e.g. from the previous frame), so that the exchange can be enqueued without a
 * host read of the counts.  The caller zeroes the send buffer (an all-zero record lists nothing at its destination:
 * radius 0), checks counts[d] <= group_base_host[d+1] - group_base_host[d] when it reads them - e.g. together with
 * the stripe's pair count at the end of the forward pass - and runs the frame again with exact groups if not.
 * group_base_host = NULL: ts_route_count. */
int ts_route_count_padded(int32_t n, const float* xys, const int32_t* radii, const ts_camera* cam_host,
                          const ts_stripes* stripes_host, const int32_t* group_base_host, int32_t* route_ws,
                          int32_t* counts, void* stream);
/* records[sum(counts), 16]: grouped by destination (ascending), ascending Gaussian index inside a group;
 * splats = the owner's packed records (colours, sigmoid opacity), gid_base = global index of Gaussian 0. */
int ts_route_pack(int32_t n, int32_t gid_base, const float* xys, const int32_t* radii, const float* depths,
                  const float* splats, const ts_camera* cam_host, const ts_stripes* stripes_host,
                  const int32_t* route_ws, float* records, void* stream);
/* importing rank, m received records: the arrays binning takes (num_tiles_hit for cam's stripe) ... */
int ts_import_records(int32_t m, const float* records, const ts_camera* cam_host, float* xys, float* depths,
                      int32_t* radii, int32_t* num_tiles_hit, void* stream);
/* ... and, after ts_scan_tiles, the 48-byte compositing records */
int ts_import_pack(int32_t m, const float* records, const int32_t* cum_tiles_hit, const ts_camera* cam_host,
                   float* splats, void* stream);
/* ts_reduce_partials writing one gradient row per (imported) Gaussian instead of the five arrays */
int ts_reduce_partials_rows(int32_t n, int32_t channels, int32_t flags, const int32_t* num_tiles_hit,
                            const int32_t* cum_tiles_hit, const float* partials, const uint8_t* row_flags,
                            const float* splats, float* grad_rows, void* stream);
/* owner: grad_rows[sum(counts), 12] in the order of ts_route_pack's records -> dense 2-D gradients of the owned
 * Gaussians (rows of a Gaussian summed in ascending stripe order; clamp mask of the colour stage and the
 * sigmoid's derivative applied: v_opacity is w.r.t. the logits).  v_depth may be NULL (channels == 3). */
int ts_route_accumulate(int32_t n, int32_t channels, const float* xys, const int32_t* radii, const float* splats,
                        const uint8_t* color_mask, const ts_camera* cam_host, const ts_stripes* stripes_host,
                        const int32_t* route_ws, const float* grad_rows, float* v_xy, float* v_conic,
                        float* v_colors, float* v_depth, float* v_opacity, void* stream);

/* Executor entries of the sharded frame (one native call per group of launches, as ts_frame_* above).  Two
 * ts_frame describe a rank's frame: fo = its OWNED Gaussians with the full-frame camera, fs = the records its
 * stripe imported (n = number of records, cam = the stripe).
 *   ts_shard_owner_fwd          project_fwd, colors_pack_fwd, route_count            (fo)
 *   ts_route_pack               once the caller knows the counts
 *   ts_shard_stripe_fwd_import  import_records, scan_tiles (total -> fs->total_host), import_pack, bin_count,
 *                               tile_offsets                                          (fs)
 *   ts_frame_fwd_composite      bin_scatter, sort_tiles, raster_fwd                   (fs)
 *   ts_shard_stripe_bwd         raster_bwd, reduce_partials_rows -> grad_rows[n, 12]  (fs)
 *   ts_shard_owner_bwd          route_accumulate, sh_colors_bwd, project_bwd          (fo) */
int ts_shard_owner_fwd(const ts_frame* fo, const ts_stripes* stripes_host, int32_t* route_ws, int32_t* counts,
                       void* stream);
int ts_shard_owner_fwd_padded(const ts_frame* fo, const ts_stripes* stripes_host, const int32_t* group_base_host,
                              int32_t* route_ws, int32_t* counts, void* stream);   /* see ts_route_count_padded */
/* The owner stage's forward pass in ONE launch + the route scan (round 5; what ts_shard_owner_fwd* issue when the rank
 * owns at most TS_SMALL_N_FUSED Gaussians - environment, default 262144, 0 = never -: a rank of 8 on a 1 M scene is
 * bound by launch floors there): projection (project_flags as ts_project_fwd), colour stage on the split coefficient
 * tensors, packed records (cum_tiles_hit := num_tiles_hit, as the owner stage packs them) and the destination counts
 * of ts_route_count_padded - the same bits in every array the three launches write. */
int ts_shard_owner_fwd_fused(int32_t n, int32_t degrees_to_use, int32_t num_bases, const float* means3d,
                             const float* scales, const float* quats, const float* view34, const float* projview,
                             const ts_camera* cam_host, int32_t project_flags, const float* origin,
                             const float* colors_dc, const float* colors_rest, const float* opacity, int32_t channels,
                             int32_t raster_flags, float* xys, float* depths, int32_t* radii, float* conics,
                             int32_t* num_tiles_hit, uint8_t* clamp_mask, float* splats, const ts_stripes* stripes_host,
                             const int32_t* group_base_host, int32_t* route_ws, int32_t* counts, void* stream);
/* ... and its backward pass in one launch (what ts_shard_owner_bwd issues for such a shard): ts_route_accumulate,
 * ts_sh_colors_bwd (no mask: it was applied to the sums) and ts_project_bwd (flags 3) in the lane that owns the Gaussian;
 * the same bits in v_xy / v_conic / v_colors / v_depth / v_opacity and the six parameter gradients. */
int ts_shard_owner_bwd_fused(int32_t n, int32_t channels, int32_t degrees_to_use, int32_t num_bases, const float* means3d,
                             const float* scales, const float* quats, const float* view34, const float* projview,
                             const float* origin, const float* xys, const int32_t* radii, const float* splats,
                             const uint8_t* color_mask, const ts_camera* cam_host, const ts_stripes* stripes_host,
                             const int32_t* route_ws, const float* grad_rows, float* v_xy, float* v_conic,
                             float* v_colors, float* v_depth, float* v_opacity, float* v_colors_dc, float* v_colors_rest,
                             float* v_means3d, float* v_scales, float* v_quats, void* stream);
int ts_shard_stripe_fwd_import(const ts_frame* fs, const float* records, void* stream);
int ts_shard_stripe_bwd(const ts_frame* fs, float* grad_rows, void* stream);
int ts_shard_owner_bwd(const ts_frame* fo, const ts_stripes* stripes_host, const int32_t* route_ws,
                       const float* grad_rows, void* stream);

/* ABI 6 - the WHOLE STEP of one rank in four calls, split only where the two collectives sit:
 *   ts_shard_rank_fwd_a   owner stage: project_fwd, colors_pack_fwd, route_count (padded groups), the send buffer zeroed,
 *                         route_pack                                  -> send[send_rows, 16], counts[ranks]
 *      (caller: all_to_all of the records with the groups' capacities as split sizes, all_gather of the counts)
 *   ts_shard_rank_fwd_b   stripe stage: import_records, scan_tiles (total -> fs->total_host), import_pack, bin_count,
 *                         tile_offsets (capacity guard), bin_scatter, sort_tiles, raster_fwd   <- recv[recv_rows, 16]
 *   ts_shard_rank_bwd_a   raster_bwd, reduce_partials_rows            -> grad_rows[recv_rows, 12]
 *      (caller: the reverse all_to_all)
 *   ts_shard_rank_bwd_b   route_accumulate, sh_colors_bwd, project_bwd <- back[send_rows, 12]
 * Nothing of a frame is looked at by the host between the calls: the groups of the exchange have the CAPACITIES the
 * caller derived from an earlier frame (group_base_host: ranks + 1 ascending offsets into send / back), the lists of the
 * stripe the capacity fs->capacity (ts_tile_offsets' guard).  The caller reads the counts and the stripe's pair count
 * once the forward pass is enqueued and runs the frame again with exact sizes (the separate entries above) if a group
 * or the lists outgrew their capacity.  fo / fs as for the entries above; fs->n = recv_rows. */
typedef struct ts_rank_step {
    ts_stripes stripes;
    int32_t group_base[TS_MAX_RANKS + 1];      /* HOST: offsets of the destination groups in send / back */
    int32_t gid_base;                           /* global index of the rank's first Gaussian */
    int32_t send_rows, recv_rows;               /* = group_base[ranks]; sum of the capacities of the groups received */
    int32_t* route_ws;                          /* >= ts_route_ws_ints(fo->n, ranks) */
    int32_t* counts;                            /* device, ranks int32: records per destination of THIS frame */
    float *send, *recv;                         /* [send_rows, 16], [recv_rows, 16] */
    float *grad_rows, *back;                    /* [recv_rows, 12], [send_rows, 12] */
} ts_rank_step;
int32_t ts_rank_step_struct_bytes(void);
int ts_shard_rank_fwd_a(const ts_frame* fo, const ts_rank_step* r, void* stream);
int ts_shard_rank_fwd_b(const ts_frame* fs, const ts_rank_step* r, void* stream);
int ts_shard_rank_bwd_a(const ts_frame* fs, const ts_rank_step* r, void* stream);
int ts_shard_rank_bwd_b(const ts_frame* fo, const ts_rank_step* r, void* stream);

/* ============ training-step ops around the path (scripts/train.py:58-63,97) ===== */

/* Photometric loss of the training step and its gradient w.r.t. the rendered image:
 *   L = c_l1 * mean|X - Y| + c_ssim * (1 - SSIM(X, Y))     (train.py:58-63 with c = 1-lambda, lambda)
 * X = image[H,W,3], Y = target[H,W,3], HWC float32.  SSIM as pytorch_msssim.SSIM(data_range=1,
 * size_average=True, channel=3) (model_gaussian.py:57): 11-tap Gaussian window (sigma 1.5), valid
 * filtering, K = (0.01, 0.03).  The entry writes per-block partial sums to the tail of `ws`
 * (2 floats per 32x32 tile: SSIM-map sum, |X-Y| sum) and, if v_image != NULL,
 *   v_image = w_l1 * sign(X - Y) + w_ssim * dSSIMsum/dX      (caller passes w_l1 = c_l1 / (3 H W),
 *                                                             w_ssim = -c_ssim / (3 (H-10)(W-10))).
 * ws: >= ts_photometric_ws_floats(H, W) floats; the partial sums ({ssim, l1, depth l1} per wave of pass 1; any
 * number of triples: (ts_photometric_ws_floats - 9 (H-10)(W-10)) / 3) start at ws + 9 (H-10)(W-10). */
int64_t ts_photometric_ws_floats(int32_t height, int32_t width);
/* Same loss on the compositing kernels' own output: `image` has pixel_floats (3 or 4) floats per
 * pixel; with 4, channel 3 is the rendered depth and, when depth_target[H,W] != NULL, the depth L1 of
 * train.py:65-69 is evaluated too: per-tile sums {ssim, l1, depth l1} at ws + 9 (H-10)(W-10) (3 per
 * tile) and v_image[..., 3] = w_depth * sign(depth - target) (0 without a target).  v_image has the
 * layout of `image`.  Saves slicing / re-packing a 33 MB image and its gradient per training step. */
int ts_photometric_loss_rgbd(int32_t height, int32_t width, int32_t pixel_floats, const float* image,
                             const float* target, const float* depth_target, float w_l1,
                             float w_ssim, float w_depth, float* ws, float* v_image, void* stream);
int ts_photometric_loss(int32_t height, int32_t width, const float* image, const float* target,
                        float w_l1, float w_ssim, float* ws, float* v_image, void* stream);
/* ABI 5: the same loss on the frame as two planes (ts_raster_fwd_planes): image[H,W,3], depth[H,W] (may be NULL
 * without a depth target), gradients v_image[H,W,3] and v_depth[H,W] (NULL: not wanted; v_depth = 0 everywhere
 * without a depth target).  Sums as ts_photometric_loss_rgbd. */
int ts_photometric_loss_planes(int32_t height, int32_t width, const float* image, const float* depth,
                               const float* target, const float* depth_target, float w_l1, float w_ssim,
                               float w_depth, float* ws, float* v_image, float* v_depth, void* stream);
/* ABI 8: the loss VALUE from the partial sums an entry above left in `ws` (same height, width and weights):
 *   out4 = {loss, mean|X - Y|, SSIM, mean|depth - depth_target|}   (device floats; sums taken in double, fixed order)
 * with loss = (1 - lambda) * out4[1] + lambda * (1 - out4[2]) + lambda_depth * out4[3] as the weights encode it
 * (train.py:58-69).  One small launch instead of a dozen one-element tensor operations on the caller's side. */
int ts_photometric_loss_reduce(int32_t height, int32_t width, float w_l1, float w_ssim, float w_depth,
                               const float* ws, float* out4, void* stream);

/* torch.optim.Adam (defaults: no amsgrad, no weight decay; train.py:26) on up to TS_ADAM_MAX_TENSORS
 * tensors with per-tensor learning rates (model_gaussian.py:112-120) in one launch.  The pointer
 * tables are HOST arrays of device pointers.  steps_host[i] >= 1 is tensor i's own 1-based step count
 * (torch keeps it per parameter: a tensor without a gradient is skipped and does not age). */
#define TS_ADAM_MAX_TENSORS 8
int ts_adam_step(int32_t num_tensors, float* const* params_host, const float* const* grads_host,
                 float* const* exp_avg_host, float* const* exp_avg_sq_host, const int64_t* numel_host,
                 const float* lr_host, const int32_t* steps_host, float beta1, float beta2, float eps,
                 void* stream);

/* FUSED ADAM (round 6; scripts/train.py:93-97): the parameter-stage backward kernels apply the optimiser to the gradient
 * they hold in registers - the very update of ts_adam_step, bit for bit - instead of writing six gradient tensors that
 * the optimiser launch reads once.  ts_adam: the six groups in the order of the model's parameters() (model_gaussian.py:
 * 112-120): means, colors_dc, colors_rest, scales, quats, opacities; step[k] >= 1 is group k's own 1-based step count.
 *   ts_sh_colors_bwd_adam   ts_sh_colors_bwd with colors_dc / colors_rest (and their moments) updated in place
 *   ts_project_bwd_adam     ts_project_bwd (no v_cov3d) with means3d / scales / quats updated in place - the RAW tensors:
 *                           pass the flags the forward pass used (TS_PROJECT_LOG_SCALES | TS_PROJECT_RAW_QUATS) - and,
 *                           when `opacities` is not NULL, the opacity logits from v_opacity (ts_reduce_partials with
 *                           TS_RASTER_LOGIT_OPACITY)
 *   ts_frame_bwd_params_adam  what ts_frame_bwd_params enqueues, with the two entries above: no parameter gradient is
 *                           written (f->v_means ... f->v_colors_rest are not read); f->v_xy - xys.grad - still is */
#define TS_ADAM_MEANS 0
#define TS_ADAM_COLORS_DC 1
#define TS_ADAM_COLORS_REST 2
#define TS_ADAM_SCALES 3
#define TS_ADAM_QUATS 4
#define TS_ADAM_OPACITIES 5
typedef struct ts_adam {
    float* exp_avg[6];
    float* exp_avg_sq[6];
    float lr[6];
    int32_t step[6];
    float beta1, beta2, eps;
} ts_adam;
int ts_sh_colors_bwd_adam(int32_t n, int32_t degrees_to_use, int32_t num_bases, const float* means3d,
                          const float* origin, const uint8_t* clamp_mask, const float* v_colors,
                          float* colors_dc, float* colors_rest, const ts_adam* adam_host, void* stream);
int ts_project_bwd_adam(int32_t n, float* means3d, float* scales, float* quats, const float* viewmat,
                        const float* projmat, const ts_camera* cam_host, int32_t flags, const int32_t* radii,
                        const float* v_xy, const float* v_depth, const float* v_conic, float* opacities,
                        const float* v_opacity, const ts_adam* adam_host, void* stream);
int ts_frame_bwd_params_adam(const ts_frame* f, const ts_adam* adam_host, void* stream);

/* ================== densification hooks (model_gaussian.py:130-242) ========= */
/* Call order of one densify_and_prune (model_gaussian.py:138-195):
 *   ts_densify_classify -> ts_densify_plan -> (host reads counts[4]) -> ts_gather_rows x 3 tables
 *   (parameters: copy_rows = N'; exp_avg, exp_avg_sq: copy_rows = K, the new rows are zero)
 *   -> ts_split_fixup on rows [K + C, N').  update_state(optim, mask) alone (train.py:103-105) is
 *   flags = mask ? TS_DENSIFY_PRUNE : 0 -> ts_densify_plan -> ts_gather_rows x 3.               */
#define TS_DENSIFY_CLONE 1   /* small scale, large mean 2-D gradient (:152-153): row is duplicated       */
#define TS_DENSIFY_SPLIT 2   /* large scale, large gradient (:164-165): two samples replace the row      */
#define TS_DENSIFY_PRUNE 4   /* row is dropped (:181-183; every split row is also pruned)                 */

typedef struct ts_densify_policy {
    float interval_densify;  /* model.interval_densify (:148)                                   */
    float max_dim;           /* max(width, height) of the last rendered camera (:146-148)       */
    float tau_means;         /* --tau-means, train.py:212                                       */
    float scale_thresh;      /* --densify-scale-thresh, train.py:213                            */
} ts_densify_policy;

/* update_grad_accum (:132): accum[i] += ||v_xy[i]||_2.  v_xy [n,2] is extras['xys'].grad. */
int ts_grad_accum(int32_t n, const float* v_xy, float* accum, void* stream);

/* Per-Gaussian policy bits (:148-183).  scales are log-scales [n,3], opacities logits [n]. */
int ts_densify_classify(int32_t n, const float* accum, const float* scales, const float* opacities,
                        const ts_densify_policy* policy, uint8_t* flags, void* stream);

/* Row map of the rebuilt tensors.  counts (device, 4 int32) <- {K kept, C cloned, S split, N' = K + C + 2S};
 * src_of (device, capacity >= 2 n int32; N' <= 2 n always) <- source row of every new row, laid out
 * [kept | cloned | split sample 0 | split sample 1], source order within each part - the order of
 * torch.cat((param[~mask], cat(cloned, split.repeat(2)))) at :187-192, :213-226.
 * ws: >= ts_densify_ws_ints(n) int32. */
int64_t ts_densify_ws_ints(int32_t n);
int ts_densify_plan(int32_t n, const uint8_t* flags, int32_t* ws, int32_t* counts, int32_t* src_of,
                    void* stream);

/* dst_i[r, :] = r < copy_rows ? src_i[src_of[r], :] : 0  for up to TS_GATHER_MAX_TENSORS row-major
 * float32 tensors (row_floats_host[i] floats per row) in one launch.  Pointer tables are HOST arrays. */
#define TS_GATHER_MAX_TENSORS 8
int ts_gather_rows(int32_t num_tensors, const float* const* src_host, float* const* dst_host,
                   const int32_t* row_floats_host, int32_t dst_rows, int32_t copy_rows,
                   const int32_t* src_of, void* stream);

/* GaussianDistribution.sample (:547-557) for the 2S sampled rows: with i = src_of_split[j],
 * means_out[j] = R(quats[i] / |quats[i]|) (z[j] * exp(scales[i])) + means[i],
 * scales_out[j] = log(exp(scales[i]) / 1.6).  z [rows,3] are unit normal draws (the reference's
 * torch.normal(0, std) is z * std).  src_of_split = src_of + K + C; *_out point at row K + C. */
int ts_split_fixup(int32_t rows, const int32_t* src_of_split, const float* means, const float* scales,
                   const float* quats, const float* z, float* means_out, float* scales_out,
                   void* stream);

/* ======================== PLY record layout (model_gaussian.py:330-361) ===== */
/* export_ply's per-Gaussian float32 record: x y z | nx ny nz (zeros) | f_dc_0..2 |
 * f_rest_0..3*k_rest-1 (channel-major: f_rest[c * k_rest + k] = colors_rest[i, k, c], :351) | opacity |
 * scale_0..2 | rot_0..3  ->  ts_ply_row_floats(k_rest) = 17 + 3 k_rest floats (62 at SH degree 3).
 * pack: six tensors -> rows [n, W]; unpack: the inverse (the normals are ignored).  Device pointers. */
int32_t ts_ply_row_floats(int32_t k_rest);
int ts_ply_pack_rows(int32_t n, int32_t k_rest, const float* means, const float* colors_dc,
                     const float* colors_rest, const float* opacities, const float* scales,
                     const float* quats, float* rows, void* stream);
int ts_ply_unpack_rows(int32_t n, int32_t k_rest, const float* rows, float* means, float* colors_dc,
                       float* colors_rest, float* opacities, float* scales, float* quats,
                       void* stream);

/* ======================================= measurement utility ================================== */
/* Streaming read of n_floats float32 (16-byte loads, grid-stride): the read-bandwidth microbenchmark
 * that SURVEY.md 8(d) D1 asks the roofline to be quoted against as well.  sink: >= 1 float. */
int ts_bench_stream_read(const float* src, int64_t n_floats, float* sink, void* stream);
/* Gather of m 48-byte records records[ids[j]] (three 16-byte loads per lane, the compositing kernels' access
 * pattern) on a known record count: calibrates rocprofv3's FETCH_SIZE for gathers.  sink: >= 1 float. */
int ts_bench_gather48(const float* records, const int32_t* ids, int64_t m, float* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TINYSPLAT_HIP_H */
