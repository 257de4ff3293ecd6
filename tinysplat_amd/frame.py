"""Whole-frame autograd function: the render adapter's recipe (rasterize.py:26-62) as ONE
``torch.autograd.Function`` over the C ABI's frame executor.

The three drop-in ops of ``ops.py`` each cost an autograd node, a dozen Python-level tensor
allocations and a binning-cache lookup per frame, and every kernel launch issued from Python costs
~8 us; on small scenes and on multi-GPU stripes that host time (~0.5 ms) exceeds the GPU time.  This
module describes a frame once in a ``ts_frame`` struct (a handful of workspace allocations) and lets
``csrc/frame.hip`` enqueue the kernels (project -> scan -> colour stage -> pack -> binning ->
composite; backward: composite -> reduce -> [all-reduce across stripes] -> colour stage -> project):
three native calls forward, two backward, with the adapter's exp / normalise / sigmoid folded in
(``TS_PROJECT_*`` / ``TS_RASTER_LOGIT_OPACITY``), one 4-channel compositing pass for RGB + depth,
tight tile lists, the adapter's clamp(max=1) inside the compositing kernels, and a single autograd
node.  Results are bitwise those of the op-by-op path
(tests/test_gpu_parity.py: test_one_node_frame_is_bitwise_the_fused_op_recipe).

While ``ops.kernel_timer`` is recording (bench.py's per-entry table) the same kernels are issued one
C-ABI entry at a time from Python on the same buffers, so that each entry can be bracketed by events.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from ._comm import collective_timer
from . import _lib
from ._lib import TsFrame
from .ops import (TileBinning, _call, _camera, _f32c, _need_hip, _ptr, _stream, _stripe_rows, _tile_bounds,
                  deg_from_sh, kernel_timer)

# Tight tile lists (see ts_bin_count): (Gaussian, tile) pairs that provably cannot reach alpha >= 1/255
# anywhere in the tile are dropped at binning time.  Results are bit-identical either way
# (tests/test_gpu_parity.py); TS_TIGHT_BINNING=0 restores gsplat's bounding-box lists for A/B timing.
TIGHT_BINNING = os.environ.get("TS_TIGHT_BINNING", "1") != "0"

# Launches with at most this many tiles (a stripe of a multi-GPU frame, a small image) composite with
# four waves per tile, one per 8x8 block (TS_RASTER_SPLIT_BLOCKS): an MI355X has 1024 SIMDs, and with
# one wave per tile such launches cannot hide any latency.  0 disables.
SPLIT_BLOCKS_BELOW = int(os.environ.get("TS_SPLIT_BLOCKS_BELOW", "2400"))

# LIST SEGMENTS in the backward pass of such a launch (bits 8..11 of ts_camera.hints, csrc/raster.hip: LIST SEGMENTS;
# TS_LIST_SEGMENTS = auto (default) | 1 (off: split backward) | 2..8): the split forward pass also leaves, per pixel,
# the transmittance in front of and the colour behind up to S - 1 boundaries of every list of two chunks or more, and
# the backward pass replays the segments as independent work items of one wave over the whole tile - instead of four
# waves per tile that each walk and stage the WHOLE list for a quarter of the pixels and write a gradient row of
# their own.  "auto": as many segments (2 .. 8) as bring the launch to ~16 384 work items (three rounds of raster_bwd's
# 5 120 wave slots; 8 192 while the kernel ran four waves per SIMD).  A list of one chunk (<= 64 entries) cannot be cut, and a launch of such lists is better off split:
# the BACKWARD pass decides - segments when the frame averages >= LIST_SEGMENTS_FROM bounding-box pairs per tile (the
# frame's own pair count, known by then - the same frame always takes the same path; ~0.65 of them are listed),
# split blocks otherwise.  Measured on MI355X, raster_fwd + raster_bwd + reduce_partials, split -> 8 segments
# (profiles/r04u_list_segments_small_launches.txt): a 1/8 stripe of config 3 (1 020 tiles, ~560 entries per list)
# 83 + 131 + 50 -> 95 + 95 + 30 us, with depth 85 + 137 + 56 -> 97 + 97 + 31; 512x512 / 200 k (295 per list)
# 55 + 93 + 15 -> 65 + 79 + 10; 256x256 / 10 k (lists of one chunk) would be 17 -> 33 us in raster_bwd - hence the
# threshold.  The image is bitwise the split pass's; gradients agree with the uncut pass to rounding (<= 8e-7 of a
# tensor's largest entry) and with the float64 oracle as well as the uncut pass does (the colour behind a boundary is
# summed per segment, not taken as a difference of the running sum: tools/fuzz_frame.py seed 10 keeps 99.9 % of its
# entries within 1e-5 max(1, |ref|), where the difference form dropped to 98.2 %).  Across LIST shapes (tight vs
# bounding-box lists, wide lists) gradients are bitwise equal only with TS_LIST_SEGMENTS=1: the boundaries depend on
# a list's length.  On a full frame segments gain nothing (profiles/HISTORY.md, round 4) and are never used.
_ls = os.environ.get("TS_LIST_SEGMENTS", "auto")
LIST_SEGMENTS = _ls if _ls == "auto" else max(1, min(8, int(_ls)))
LIST_SEGMENTS_FROM = int(os.environ.get("TS_LIST_SEGMENTS_FROM", "192"))
SEGMENT_ITEMS = int(os.environ.get("TS_SEGMENT_ITEMS", "16384"))      # "auto": work items a small launch is cut into


# HYBRID LAUNCHES (bits 12..15 of ts_camera.hints, csrc/raster.hip: HYBRID LAUNCH): a full frame - at least HYBRID_FROM
# 16x16 tiles, one wave per tile - is two rounds of long waves whose last third runs half empty.  Cutting EVERY list
# gains nothing there (above); cutting only the tiles that are dispatched LAST does: the first HYBRID_WHOLE16 / 16 of
# every band of tiles stay whole, the rest become HYBRID_SEGS one-wave items each in the backward pass, and the forward
# pass keeps the boundary state for those tiles only.  TS_HYBRID_SEGS=1 switches it off.
HYBRID_FROM = int(os.environ.get("TS_HYBRID_FROM", "4096"))
HYBRID_SEGS = max(1, min(8, int(os.environ.get("TS_HYBRID_SEGS", "8"))))
HYBRID_WHOLE16 = max(1, min(15, int(os.environ.get("TS_HYBRID_WHOLE16", "13"))))
# ... and a launch between the two regimes (HYBRID_MID_FROM <= tiles < HYBRID_FROM: one rank of 2 or 3 on a 1080p frame) has
# fewer tiles than the GPU has wave slots, so a whole tile's wave IS the critical path: most tiles are cut there
# (tools/midrange_policy.sh: 4 080 tiles 0.79 -> 0.72 ms per step, 2 760 tiles 0.655 -> 0.61)
HYBRID_MID_FROM = int(os.environ.get("TS_HYBRID_MID_FROM", "1537"))
HYBRID_MID_WHOLE16 = max(1, min(15, int(os.environ.get("TS_HYBRID_MID_WHOLE16", "6"))))
HYBRID_MID_COOP16 = max(0, min(15, int(os.environ.get("TS_HYBRID_MID_COOP16", "6"))))
# COOPERATIVE TILES (bits 16..19 of ts_camera.hints, csrc/raster.hip: COOPERATIVE TILES): the FORWARD launch of such a
# frame hands the first HYBRID_COOP16 / 16 of every band - the tiles it dispatches last - to a workgroup of four waves
# each (shared staging and sort, one 8x8 block per wave) instead of one wave, so that the launch's own tail is filled
# with items a quarter as long.  Same image, final_Ts and final_index, bit for bit.  0 switches it off.
HYBRID_COOP16 = max(0, min(15, int(os.environ.get("TS_HYBRID_COOP16", "3"))))
# ... and a SMALL launch (<= SPLIT_BLOCKS_BELOW tiles), whose forward pass used four waves per tile that each walked the
# whole list (TS_RASTER_SPLIT_BLOCKS), composites every tile that way: TS_HINT_COOP_SPLIT, same four waves per tile,
# each list staged and sorted once.  The backward pass (list segments or split blocks) is unchanged, and so is every
# output bit.  TS_COOP_SPLIT=0 switches it off.
COOP_SPLIT = os.environ.get("TS_COOP_SPLIT", "1") != "0"


def set_launch_hints(cam, segs: int, w16: int, mode: int, split: bool, skewed: bool = False) -> None:
    """The compositing launches' fields of ``cam.hints``: list segments / whole-tile share (bits 8..15), cooperative
    tiles of the forward launch (bits 16..20)."""
    cam.hints = (cam.hints & ~0x1FFF00) | (((segs << 8) | (w16 << 12)) if segs > 1 else 0)
    tiles16 = cam.tile_rows * cam.tile_bounds_x
    if mode == 0 and not split and tiles16 >= HYBRID_FROM:
        cam.hints |= (HYBRID_MID_COOP16 if skewed else HYBRID_COOP16) << 16
    elif mode == 0 and not split and tiles16 >= HYBRID_MID_FROM:
        cam.hints |= HYBRID_MID_COOP16 << 16
    if mode == 0 and split and COOP_SPLIT:
        cam.hints |= _lib.HINT_COOP_SPLIT


def _list_segments(tiles16: int, mode: int, split: bool, skewed: bool = False):
    """-> (S, W16): list segments the forward pass prepares for (1 = none) and the whole-tile share of a hybrid launch"""
    if mode != 0 or tiles16 <= 0:
        return 1, 0
    if not split:
        if HYBRID_SEGS > 1 and tiles16 >= HYBRID_FROM:
            return HYBRID_SEGS, (HYBRID_MID_WHOLE16 if skewed else HYBRID_WHOLE16)
        if HYBRID_SEGS > 1 and tiles16 >= HYBRID_MID_FROM:
            return HYBRID_SEGS, HYBRID_MID_WHOLE16
        return 1, 0
    if LIST_SEGMENTS != "auto":
        return int(LIST_SEGMENTS), 0
    return max(2, min(8, SEGMENT_ITEMS // tiles16)), 0


last_segments = {}      # device index -> list segments of the most recent backward pass (1 = none; tests, tools)


def segments_for_count(cam, segs: int, total: int) -> int:
    """The frame's own choice between list segments and the uncut replay, from ITS pair count (the same frame always
    takes the same path): ``segs``, or 1 with the field of ``cam`` cleared.  Called by the FORWARD pass as soon as the
    count is known - a launch of short lists then keeps no boundary records at all (up to 36 planes written
    for nothing) - and again by the backward pass (where the forward pass was enqueued before the count was known)."""
    if segs > 1 and total >= LIST_SEGMENTS_FROM * cam.tile_rows * cam.tile_bounds_x:
        return segs
    cam.hints &= ~0xFF00
    return 1


def backward_segments(cam, segs: int, total: int, dev_index: int = 0) -> int:
    """the backward pass's choice (see above): ``segs``, or 1 with the field of ``cam`` cleared -> split blocks"""
    segs = segments_for_count(cam, segs, total)
    last_segments[dev_index] = segs
    return segs

# WIDE LISTS: the frame path bins, scatters and sorts on 32x16 tiles - two horizontally adjacent 16x16 tiles
# as one list (ts_camera.wide_tiles; 0.73x the list entries on the random scenes, longer lists for the sort
# networks) - and composites with ONE WAVE PER 16x16 TILE as before, each wave walking the list of the wide
# tile it lies in and dropping the Gaussians whose tile box does not contain it (TS_RASTER_NARROW_WAVES).
# Image, depth and every transmittance decision are bitwise those of 16x16 lists, gradients agree to
# rounding (tests/test_gpu_parity.py::test_wide_tiles_change_no_pixel).  Measured on MI355X (mode 0 -> 2):
# config 5 (5 M, 4K) 5.71 -> 5.01 ms, 5.03 -> 4.33 ms in Morton order (sort 0.94 -> 0.55, scatter 0.88 -> 0.70,
# count 0.23 -> 0.13 ms; compositing +1 %); config 3 1.316 -> 1.305 ms (sort 79 -> 64 us, scatter 64 -> 48 us,
# offsets 27 -> 19 us against raster_bwd 545 -> 566 us for the longer lists it stages).
# Mode 1 also gives each wave the whole wide tile (eight pixels per lane: the "super-tile" mapping, one
# butterfly reduction and one gradient row per Gaussian and wide tile): built, bit-identical, and slower -
# raster_fwd 0.25 -> 0.39 ms, raster_bwd 0.55 -> 0.71 ms on config 3, because 150-170 VGPRs leave 3 waves per
# SIMD instead of 4-5.  The drop-in op keeps gsplat's 16x16 lists.
# TS_WIDE_TILES = 0 | 1 | 2 | auto (default): mode 2 where the lists are long - the previous frame on the
# device averaged >= WIDE_LISTS_FROM bounding-box pairs per 16x16 tile (config 5: 2 490; config 3: 766).  The
# threshold was 1 000 until the hybrid backward launch and the cooperative forward tiles (16x16 lists only) existed;
# with them mode 0 wins up to ~2 200 pairs per tile (tools/list_mode_check.sh: 1 M / 2 M at 1280x720 0.96 -> 0.92,
# 1.19 -> 1.11 ms; 2.5 M at 1080p 1.61 -> 1.57) and mode 2 beyond (4 M at 1080p 2.07 -> 2.01, config 5 4.64 -> 3.71) -
# mode 0 otherwise; a first frame goes by its tile count (4K-class frames start wide).  Modes 0 and 2 run the
# same per-tile arithmetic on the same Gaussians in the same order, so switching never changes a result.
_wt = os.environ.get("TS_WIDE_TILES", "auto")
WIDE_TILES = _wt if _wt == "auto" else int(_wt)
WIDE_LISTS_FROM = int(os.environ.get("TS_WIDE_LISTS_FROM", "2300"))
BALANCED_WALK_FROM = float(os.environ.get("TS_BALANCED_WALK_FROM", "10"))     # bounding-box tiles per Gaussian
_pairs_per_tile = {}    # device index -> bounding-box pairs per 16x16 tile of the most recent frame


# ROUND 6 (tools/policy_regret.py, profiles/r06e_policy_regret.txt): on scenes the constants above were not fitted on, the
# policy was within 5 % of the best forced setting at 720p / 1080p, but at 4K (32 400 tiles) with SHORT lists - 500
# pairs per tile - wide lists beat the 16x16 choice by 6 % (uniform scene) and 16 % (opaque scene): with that many lists
# the per-list costs (offsets, sort launches, a wave's prologue / epilogue per tile) weigh more than the longer walks,
# while a CLUSTERED scene (80 % of the Gaussians in 5 % of the image: longest list 5 900 at 567 pairs per tile) loses
# 8 % with wide lists - its few very long lists leave the sorting networks - and wants more of its tiles cut.  The pair
# count alone cannot tell these apart; the LONGEST list can: ts_tile_offsets_stats stores it behind the count word
# (TS_FRAME_LIST_STATS), and the next frame's policy reads it - never waits for it:
#   * many tiles (>= MANY_TILES_FROM) and no long list (< LONG_LIST on 16x16 lists, twice that on wide ones): wide lists;
#   * a skewed scene (longest list >= SKEW_FROM x the bounding-box pairs per tile) on 16x16 lists: the hybrid launch
#     cuts most tiles (the mid-range shares W16 = 6, C16 = 6) instead of the last 3/16.
MANY_TILES_FROM = int(os.environ.get("TS_MANY_TILES_FROM", "20000"))
LONG_LIST = int(os.environ.get("TS_LONG_LIST", "2048"))
SKEW_FROM = float(os.environ.get("TS_SKEW_FROM", "4"))
_stats_mode = {}        # device index -> list mode of the frame that last asked for the statistic (full frames only)
_longest_list = {}      # device index -> (longest list of the most recent frame whose statistic has arrived, its list mode)


def _list_mode(dev_index: int, tiles16: int) -> int:
    if WIDE_TILES != "auto":
        return int(WIDE_TILES)
    prev = _pairs_per_tile.get(dev_index)
    if prev is None:
        return 2 if tiles16 >= MANY_TILES_FROM else 0
    if prev >= WIDE_LISTS_FROM:
        return 2
    st = _longest_list.get(dev_index)
    if tiles16 >= MANY_TILES_FROM and st is not None:
        longest, mode = st
        return 2 if longest < (2 * LONG_LIST if mode == 2 else LONG_LIST) else 0
    return 0


def _skewed(dev_index: int, mode: int) -> bool:
    """a few lists are far longer than the average one (see above); decided from the previous frame's statistics"""
    st, prev = _longest_list.get(dev_index), _pairs_per_tile.get(dev_index)
    if WIDE_TILES != "auto" or mode != 0 or st is None or prev is None or st[1] != 0:
        return False
    return st[0] >= SKEW_FROM * max(prev, 1.0)

# binning of the most recent frame per device index (scene statistics for bench.py / tools)
last_binning = {}

TRACE = None            # developer hook (tools/host_breakdown.py): a list receiving (label, perf_counter) marks


def _mark(label):
    if TRACE is not None:
        import time
        TRACE.append((label, time.perf_counter()))


_layouts = {}           # shape key of a frame -> (section offsets, bytes) of its one workspace allocation
_pinned_total = {}      # device index -> (pinned int32[1], event): the path's one host read
_row_flags = {}         # (device index, stream handle) -> [uint8 tensor, generation], see row_flags_for
_row_flags_lock = threading.Lock()
_row_flags_tick = [0]   # passes served (least-recently-used eviction of _row_flags)

# Row flags of the backward pass (1 byte per partial row: written in THIS pass?).  Instead of zeroing I bytes per
# frame (a 5-10 us fill launch on the critical path), one array per device lives across frames and every pass
# marks its rows with a fresh generation 1..255 (TS_RASTER_FLAG_GEN); the array is zeroed only when it grows or
# the generations wrap.  TS_FLAG_GENERATIONS=0: zero per pass as before.
FLAG_GENERATIONS = os.environ.get("TS_FLAG_GENERATIONS", "1") != "0"


FLAGS_SHRINK_AFTER = 256      # consecutive passes that needed < 1/8 of the array before it is given back


def row_flags_for(dev: torch.device, rows: int):
    """-> (uint8 tensor of >= rows bytes, generation).  generation 0: a fresh array the kernels zero themselves.

    One array per (device, STREAM): two backward passes on different streams of a device (or from two host threads,
    each on its own stream) never share a generation counter or race on the wrap-around zeroing; passes on ONE
    stream are ordered by the stream.  An array much larger than the passes need is given back only after
    FLAGS_SHRINK_AFTER consecutive small passes - a workload that alternates large and small passes on one stream (a
    full frame, then a rank's stripes; multi-resolution training) keeps one array instead of allocating and zeroing
    megabytes at every switch - and the table holds at most 32 (device, stream) entries."""
    if not FLAG_GENERATIONS:
        return torch.empty((rows,), dtype=torch.uint8, device=dev), 0
    key = (dev.index, _stream(dev))
    with _row_flags_lock:
        slot = _row_flags.get(key)
        if slot is not None and slot[0].numel() > 8 * max(rows, 1 << 20):
            slot[2] += 1
            if slot[2] >= FLAGS_SHRINK_AFTER:
                slot = None
        elif slot is not None:
            slot[2] = 0
        if slot is None or slot[0].numel() < rows:
            if key not in _row_flags and len(_row_flags) >= 32:
                # the entry that has gone unused the longest (the OLDEST entry may belong to a stream that is
                # still rendering); its array stays alive through the frames that hold it until their kernels are done
                _row_flags.pop(min(_row_flags, key=lambda k_: _row_flags[k_][3]))
            slot = _row_flags[key] = [torch.zeros((int(rows * 1.25) + 4096,), dtype=torch.uint8, device=dev), 0, 0, 0]
        _row_flags_tick[0] += 1
        slot[3] = _row_flags_tick[0]
        slot[1] += 1
        if slot[1] > 255:
            slot[0].zero_()
            slot[1] = 1
        return slot[0], slot[1]

_bg_cache = {}          # (storage address, version, channels) -> contiguous background with the depth channel

# One stripe of a multi-GPU frame (tile_rows narrower than the frame, training frame): the colour stage
# runs only for the Gaussians the stripe lists and the clamp mask is applied before the all-reduce
# (TS_FRAME_STRIPE, csrc/frame.hip).  Same gradients, bit for bit, as the dense order
# (tests/test_gpu_parity.py, tests/test_gpu_dist.py); a rank's colour stage at 8 stripes: 60 -> 24 us.
# TS_STRIPE_SPARSE=0 for A/B timing.
STRIPE_SPARSE = os.environ.get("TS_STRIPE_SPARSE", "1") != "0"

# ts_bin_scatter in two coalesced hops through the (still unused) sorted-id buffer; TS_TWO_HOP_SCATTER=0: one hop
TWO_HOP_SCATTER = os.environ.get("TS_TWO_HOP_SCATTER", "1") != "0"
# one wave per 16x16 tile on 16x16 lists: the forward compositing kernel sorts the lists of <= 1024 entries itself
# (ts_raster_fwd_sort); TS_INLINE_SORT=0: the sort as a phase of its own (TS_FRAME_SEPARATE_SORT)
INLINE_SORT = os.environ.get("TS_INLINE_SORT", "1") != "0"

DIRECT_GRADS = os.environ.get("TS_DIRECT_GRADS", "1") != "0"      # A/B switch (see _RenderFrame.backward)

# CAPACITY ALLOCATION (option, off by default): the per-intersection buffers (bucket_ids | gaussian_ids_sorted;
# partials in backward) are sized by the bounding-box pair count I, which only the GPU knows (gsplat synchronises
# for it at the same place, rasterize.py:44).  With TS_CAPACITY_ALLOC=1, from the second frame of a (scene size,
# image, stripe) on the buffers are sized by the previous frame's count x 1.25 and the WHOLE forward is enqueued
# before the count is read; a frame that needs more than the estimate leaves every list empty on the device
# (ts_tile_offsets' guard) and is enqueued again with exact sizes (tests/test_gpu_capacity.py).  Measured on
# MI355X it does not pay: the count word is already there when the host has enqueued the three kernels that do not
# need it, config 2 (100 k Gaussians) runs at its kernel time either way (0.305 ms/frame) and in two of three runs
# the early enqueue was SLOWER (0.39 ms), config 3 1.204 vs 1.193 ms - so the default waits for the count, which
# also keeps the buffers exact.  The guard is what a hipGraph capture of the forward frame would build on.
CAPACITY_ALLOC = os.environ.get("TS_CAPACITY_ALLOC", "0") != "0"
CAPACITY_GROWTH = 1.25
_capacity = {}          # (device index, n, width, height, tile rows) -> estimate for the next frame

# How the host waits for the intersection count (the one host read of a frame; the scan kernel stores it
# into the pinned word itself, csrc/frame.hip).  "event": synchronise on an event recorded behind the scan;
# "spin": a sentinel (-2^31) is stored in the word before the scan is issued and the host polls the word -
# no driver call on the critical path (a count is >= 0, or negative only on int32 overflow, which is
# reported either way).  Measured: config 2 (100 k Gaussians, host-bound) 0.400 -> 0.378 ms/frame, config 3
# unchanged.  One word per device, guarded by a lock from the sentinel store to the read of the count.
COUNT_WAIT = os.environ.get("TS_COUNT_WAIT", "spin")
_SPIN_LIMIT = 50_000_000


def alloc_quantum(count: int) -> int:
    """``count`` rounded up to one of eight sizes per octave (at least 65 536): the per-intersection buffers follow the
    frame's pair count, which changes with every camera and every optimiser step - and a caching allocator asked for a
    new size every frame keeps going back to hipMalloc (round 6: the first ~100 steps of a training run took 2.7 ms
    instead of 1.65 until the pools had seen every size).  <= 12.5 % more than asked for, and sizes that repeat."""
    count = max(int(count), 1)
    if count <= 65536:
        return 65536
    g = 1 << (count.bit_length() - 4)
    return (count + g - 1) // g * g


def _total_slot(dev: torch.device):
    slot = _pinned_total.get(dev.index)
    if slot is None:
        # the lock makes the word safe to share when two host threads render on one device (a viewer
        # thread beside the training loop): it is held from the sentinel store to the read of the count
        slot = (torch.zeros((4,), dtype=torch.int32, pin_memory=True), torch.cuda.Event(), threading.Lock())   # [count, longest list, -, -]
        _pinned_total[dev.index] = slot
    return slot


class _Frame:
    """Buffers of one frame plus the ``ts_frame`` struct that points at them."""
    __slots__ = ("fr", "cam", "n", "nb", "ch", "w", "h", "num_tiles", "total", "split", "segs", "keep",
                 "wf", "tile_bins", "ids", "bucket_ids", "out_img", "out_depth", "planes", "xys", "radii", "nth", "cum",
                 "inputs", "bg")


def _forward(means, scales, quats, opacities, colors_dc, colors_rest, view34, projview, origin, background,
             fx, fy, width, height, sh_degree, with_depth, tile_rows, keep: bool, planes: bool = False) -> _Frame:
    """Enqueues the forward frame; ``keep`` = also produce what the backward pass needs; ``planes`` (with
    ``with_depth``) = RGB and depth as two contiguous images instead of one 4-channel image."""
    _mark("fwd:enter")
    dev = _need_hip(means, scales, quats, opacities, colors_dc, colors_rest, view34, projview, origin, background)
    n = means.shape[0]
    nb = colors_rest.shape[1] + 1
    if sh_degree < 0 or sh_degree > deg_from_sh(nb):
        raise ValueError("sh_degree exceeds the stored coefficients")
    means, scales, quats = _f32c(means), _f32c(scales), _f32c(quats)
    opacities, colors_dc, colors_rest = _f32c(opacities), _f32c(colors_dc), _f32c(colors_rest)
    view34, projview, origin = _f32c(view34), _f32c(projview), _f32c(origin)
    w, h = int(width), int(height)
    cam = _camera(fx, fy, w / 2, h / 2, h, w, _tile_bounds(h, w), 1.0, tile_rows=tile_rows)
    slot_ = _pinned_total.get(dev.index)
    if slot_ is not None and dev.index in _stats_mode:
        # the longest list of the most recent frame on this device whose ts_tile_offsets_stats has run (normally the
        # previous frame; read from the mapped word without waiting - a statistic, not a result)
        _longest_list[dev.index] = (int(ctypes.c_int32.from_address(slot_[0].data_ptr() + 4).value), _stats_mode[dev.index])
    mode = _list_mode(dev.index, cam.tile_rows * cam.tile_bounds_x)
    cam.wide_tiles = 1 if mode else 0
    # hint for the scatter: the previous frame's bounding-box tiles per Gaussian (results do not depend on it)
    prev_i = _pairs_per_tile.get(dev.index)
    if prev_i is not None and n > 0 and prev_i * cam.tile_rows * cam.tile_bounds_x >= BALANCED_WALK_FROM * n:
        cam.hints = _lib.HINT_BALANCED_WALK
    ch = 4 if with_depth else 3
    if with_depth:          # channel 3 is composited over background[0], as the reference's depth pass (:86)
        key = (background.data_ptr(), background._version, dev.index)
        hit = _bg_cache.get(key)
        if hit is None:
            if len(_bg_cache) > 16:
                _bg_cache.clear()
            # the source tensor is kept alive with the entry, so that its address cannot be reused
            hit = _bg_cache[key] = (_f32c(torch.cat([background, background[:1]])), background)
        bg = hit[0]
    else:
        bg = _f32c(background)
    lib = _lib.load()
    s = _stream(dev)
    F = _Frame()
    F.cam, F.n, F.nb, F.ch, F.w, F.h, F.keep = cam, n, nb, ch, w, h, keep
    F.inputs = (means, scales, quats, opacities, colors_dc, colors_rest, view34, projview, origin)
    F.bg = bg
    num_tiles = int(lib.ts_num_tiles(ctypes.byref(cam)))            # lists of this launch (wide or 16x16 tiles)
    F.num_tiles = num_tiles
    F.split = 0 < cam.tile_rows * cam.tile_bounds_x <= SPLIT_BLOCKS_BELOW
    rows = _stripe_rows(cam)
    m = max(n, 1)
    skewed = _skewed(dev.index, mode)
    segs, w16 = _list_segments(cam.tile_rows * cam.tile_bounds_x, mode, F.split, skewed) if keep else (1, 0)
    set_launch_hints(cam, segs, w16, mode, F.split, skewed)
    F.segs = segs
    _mark("fwd:inputs checked")
    cur = torch.cuda.current_device()
    if cur != dev.index:
        torch.cuda.set_device(dev)
    try:
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        # ONE allocation for every fixed-size intermediate (a torch.empty costs ~3 us of host time, and a
        # frame used to make a dozen); 256-byte aligned sections:
        #   splats[12n] f32 | xys[2n] | conics[3n] | (unused) | depths[n] | radii[n] i32 | nth[n] | cum[n] |
        #   scan_ws | bin_ws | tile_bins[2T] | sh_mask[n] u8 | final_Ts[P] f32 | final_index[P] i32 | clamp_mask[P] u8
        px = rows * w
        lkey = (n, num_tiles, w, rows, ch, keep, cam.hints & 0xFFFF00, cam.wide_tiles, cam.tile_rows, cam.tile_row0, cam.tile_bounds_x, cam.img_height)
        lay = _layouts.get(lkey)
        if lay is None:                                  # (the same few shapes come back every frame: ~10 us of host time)
            nscan = int(lib.ts_scan_ws_ints(n))
            nbin = int(lib.ts_bin_ws_ints(n, num_tiles))
            fin_floats = int(lib.ts_final_floats(ctypes.byref(cam), ch)) if keep else px      # T_fin + the cut tiles' checkpoints
            # (the colour stage keeps the colours in registers - ts_colors_pack_fwd - so no colors[n,3] section exists)
            sizes = [48 * m, 8 * m, 12 * m, 0, 4 * m, 4 * m, 4 * m, 4 * m, 4 * nscan, 4 * nbin,
                     8 * max(num_tiles, 1)] + ([m, 4 * fin_floats, 4 * px, px] if keep else [])
            offs, off = [], 0
            for sz in sizes:
                offs.append(off)
                off += (sz + 255) & ~255
            if len(_layouts) > 64:
                _layouts.clear()
            lay = _layouts[lkey] = (offs, off)
        offs, off = lay
        _mark("fwd:sizes")
        F.wf = torch.empty((off,), dtype=torch.uint8, device=dev)
        _mark("fwd:workspace allocated")
        base = F.wf.data_ptr()
        ptr = [base + o for o in offs]
        F.planes = bool(planes and with_depth)
        F.out_img = torch.empty((rows, w, 3 if F.planes else ch), **f32)
        F.out_depth = torch.empty((rows, w), **f32) if F.planes else None
        _mark("fwd:image allocated")

        def view(k, dtype, count, shape):
            return F.wf[offs[k]:offs[k] + count * dtype.itemsize].view(dtype).view(shape)
        F.xys = view(1, torch.float32, 2 * n, (n, 2))
        F.radii = view(5, torch.int32, n, (n,))
        F.nth = view(6, torch.int32, n, (n,))
        F.cum = view(7, torch.int32, n, (n,))
        F.tile_bins = view(10, torch.int32, 2 * max(num_tiles, 1), (max(num_tiles, 1), 2))
        host, event, count_lock = _total_slot(dev)
        timed_ = kernel_timer.enabled        # (per-entry timing issues ts_tile_offsets itself: no statistic in those frames)
        fr = TsFrame()
        fr.n, fr.num_bases, fr.sh_degree, fr.channels = n, nb, int(sh_degree), ch
        # a proper stripe of the frame (one rank of a multi-GPU frame): TS_FRAME_STRIPE
        stripe = STRIPE_SPARSE and cam.tile_rows < cam.tile_bounds_y
        full = cam.tile_rows == cam.tile_bounds_y
        fr.flags = ((1 if TIGHT_BINNING else 0) | (2 if F.split else 0) | (8 if mode == 2 else 0) | (16 if stripe else 0)
                    | (0 if TWO_HOP_SCATTER else 32) | (64 if F.planes else 0) | (0 if INLINE_SORT else 128)
                    | (256 if (full and not timed_) else 0))            # TS_FRAME_LIST_STATS (the slot holds four words)
        if full and not timed_:
            _stats_mode[dev.index] = mode
        fr.cam = cam
        fr.means, fr.scales, fr.quats, fr.opacities = means.data_ptr(), scales.data_ptr(), quats.data_ptr(), opacities.data_ptr()
        fr.colors_dc, fr.colors_rest = colors_dc.data_ptr(), colors_rest.data_ptr()
        fr.view34, fr.projview, fr.origin, fr.background = view34.data_ptr(), projview.data_ptr(), origin.data_ptr(), bg.data_ptr()
        fr.splats, fr.xys, fr.conics, fr.depths = ptr[0], ptr[1], ptr[2], ptr[4]
        fr.radii, fr.num_tiles_hit, fr.cum_tiles_hit, fr.scan_ws = ptr[5], ptr[6], ptr[7], ptr[8]
        fr.bin_ws, fr.tile_bins = ptr[9], ptr[10]
        fr.total_host = host.data_ptr()
        fr.out_img = F.out_img.data_ptr()
        if F.planes:
            fr.out_depth = F.out_depth.data_ptr()
        if keep:
            fr.sh_mask, fr.final_Ts, fr.final_index, fr.clamp_mask = ptr[11], ptr[12], ptr[13], ptr[14]
        F.fr = fr
        _mark("fwd:allocated + struct")
        timed = kernel_timer.enabled
        spin = COUNT_WAIT == "spin" and n > 0
        cap_key = (dev.index, n, w, h, cam.tile_row0, cam.tile_rows)
        est = _capacity.get(cap_key) if (CAPACITY_ALLOC and n > 0) else None

        def lists(size):
            cap = alloc_quantum(size)                             # (a multiple of 64; sizes that repeat from frame to frame)
            F.bucket_ids = torch.empty((2 * cap,), **i32)         # bucket_ids | gaussian_ids_sorted
            fr.bucket_ids, fr.gaussian_ids_sorted = F.bucket_ids.data_ptr(), F.bucket_ids.data_ptr() + 4 * cap
            return cap

        def prepare():
            if timed:
                _steps_prepare(lib, fr, s)
            else:
                _lib.check(lib.ts_frame_fwd_prepare(ctypes.byref(fr), s), "ts_frame_fwd_prepare")

        def composite():
            if timed:
                _steps_composite(lib, fr, s)
            else:
                _lib.check(lib.ts_frame_fwd_composite(ctypes.byref(fr), s), "ts_frame_fwd_composite")

        count_lock.acquire()
        issued = consumed = False
        try:
            if spin:
                word = ctypes.c_int32.from_address(host.data_ptr())
                word.value = -(1 << 31)                  # sentinel: no int32 prefix sum ends here (overflow wraps past it)
            if timed:
                _steps_project(lib, fr, s)
                if n > 0:
                    host[:1].copy_(F.cum[-1:], non_blocking=True)
            else:
                _lib.check(lib.ts_frame_fwd_project(ctypes.byref(fr), s), "ts_frame_fwd_project")
            issued = n > 0
            _mark("fwd:call project")
            event.record()                               # (torch's current stream of the current device = dev, set above)
            cap = None
            if est is not None:
                # everything is enqueued against buffers of the estimated size; the device refuses to use them
                # if the frame turns out larger (ts_tile_offsets' guard)
                cap = lists(est)
                fr.capacity, fr.num_intersects = cap, cap
                prepare()
                composite()
            else:
                fr.capacity = -1
                prepare()                                # the kernels that do not need the count run while it travels
            _mark("fwd:call prepare")
            total = 0
            if n > 0:
                if spin:                                          # the one host wait of the path
                    k = 0
                    while word.value == -(1 << 31):
                        k += 1
                        if k > _SPIN_LIMIT:
                            event.synchronize()
                            break
                    total = int(word.value)
                else:
                    event.synchronize()
                    total = int(host[0])
            consumed = True
        finally:
            if issued and not consumed:
                # an exception between the scan and the read of its count: the scan may still be queued and would
                # overwrite the word under the NEXT frame's sentinel - wait for it before the word changes hands
                try:
                    event.synchronize()
                except Exception:
                    pass
            count_lock.release()
        if total < 0:
            raise OverflowError("more than 2^31-1 tile intersections: num_tiles_hit overflows its "
                                "int32 prefix sum (gsplat's cum_tiles_hit is int32 as well)")
        _mark("fwd:waited for count")
        F.total = total
        _pairs_per_tile[dev.index] = total / max(1, cam.tile_rows * cam.tile_bounds_x)
        if n > 0 and CAPACITY_ALLOC:
            if len(_capacity) > 64:
                _capacity.clear()
            _capacity[cap_key] = int(total * CAPACITY_GROWTH) + 4096
        if cap is None or total > cap:
            # first frame of this shape, or the estimate was too small (the device left every list empty)
            redo = cap is not None
            cap = lists(total)
            _mark("fwd:lists allocated")
            fr.capacity, fr.num_intersects = -1, total
            F.segs = segments_for_count(fr.cam, F.segs, total)      # short lists: no boundary records are kept
            if redo:
                prepare()
            composite()
            _mark("fwd:composite enqueued")
        else:
            fr.num_intersects = total
        F.ids = F.bucket_ids[cap:cap + total]
        _mark("fwd:call composite")
    finally:
        if cur != dev.index:
            torch.cuda.set_device(cur)
    b = TileBinning()
    b.cam, b.n, b.num_tiles, b.num_intersects = cam, n, num_tiles, total
    b.tile_bins, b.gaussian_ids_sorted, b.cum_tiles_hit, b.num_tiles_hit = F.tile_bins[:num_tiles], F.ids, F.cum, F.nth
    last_binning[dev.index] = b
    _mark("fwd:exit")
    return F


# ---- the executor's calls, one C-ABI entry at a time (per-entry timing only; mirrors csrc/frame.hip) ----
def _steps_project(lib, fr, s):
    _call("ts_project_fwd", lib.ts_project_fwd, fr.n, fr.means, fr.scales, fr.quats, fr.view34, fr.projview,
          fr.cam, 3, fr.xys, fr.depths, fr.radii, fr.conics, fr.num_tiles_hit, None, s)
    _call("ts_scan_tiles", lib.ts_scan_tiles, fr.n, fr.num_tiles_hit, fr.cum_tiles_hit, fr.scan_ws, None, s)


def _steps_prepare(lib, fr, s):
    _call("ts_colors_pack_fwd", lib.ts_colors_pack_fwd, fr.n, fr.sh_degree, fr.num_bases, fr.means, fr.origin,
          fr.colors_dc, fr.colors_rest if fr.num_bases > 1 else None, fr.sh_mask,
          fr.num_tiles_hit if fr.flags & 16 else None, fr.channels, 1, fr.xys, fr.radii, fr.conics, fr.opacities,
          fr.cum_tiles_hit, fr.cam, fr.depths if fr.channels == 4 else None, fr.splats, s)
    tight = fr.splats if fr.flags & 1 else None
    _call("ts_bin_count", lib.ts_bin_count, fr.n, fr.xys, fr.radii, tight, fr.cam, fr.bin_ws, s)
    _call("ts_tile_offsets", lib.ts_tile_offsets, fr.n, int(lib.ts_num_tiles(ctypes.byref(fr.cam))), fr.bin_ws,
          fr.tile_bins, fr.cum_tiles_hit, fr.capacity, s)


def _steps_composite(lib, fr, s):
    tight = fr.splats if fr.flags & 1 else None
    nt = int(lib.ts_num_tiles(ctypes.byref(fr.cam)))
    fused = fr.num_intersects > 0 and fr.cam.wide_tiles == 0 and not fr.flags & (8 | 128)
    if fr.num_intersects > 0:
        _call("ts_bin_scatter", lib.ts_bin_scatter, fr.n, fr.xys, fr.radii, tight, fr.cam, fr.bin_ws,
              fr.bucket_ids, None if fr.flags & 32 else fr.gaussian_ids_sorted, s)
        _call("ts_sort_tiles", lib.ts_sort_tiles_above if fused else lib.ts_sort_tiles, nt, fr.tile_bins, fr.depths,
              fr.bucket_ids, fr.gaussian_ids_sorted, fr.bin_ws,
              fr.bin_ws + 4 * (int(lib.ts_bin_ws_ints(fr.n, nt)) - 1), s)
    if fused:
        _call("ts_raster_fwd", lib.ts_raster_fwd_sort, fr.channels, 2 | (4 if fr.flags & 2 else 0), fr.cam, fr.tile_bins,
              fr.bucket_ids, fr.depths,
              fr.gaussian_ids_sorted, fr.splats, fr.background, fr.out_img, fr.out_depth if fr.flags & 64 else None,
              fr.final_Ts, fr.final_index, fr.clamp_mask, s)
        return
    _call("ts_raster_fwd", lib.ts_raster_fwd_planes, fr.channels, 2 | (4 if fr.flags & 2 else 0) | (fr.flags & 8), fr.cam,
          fr.tile_bins, fr.gaussian_ids_sorted, fr.splats, fr.background, fr.out_img,
          fr.out_depth if fr.flags & 64 else None, fr.final_Ts, fr.final_index, fr.clamp_mask, s)


def segmented(cam) -> bool:
    """list segments replace the split blocks in the backward pass (csrc/frame.hip: segmented)"""
    return ((cam.hints >> 8) & 15) > 1 and not cam.wide_tiles


def _steps_bwd_composite(lib, fr, s):
    split = 4 if (fr.flags & 2 and not segmented(fr.cam)) else 0
    gen = (fr.flag_gen & 0xff) << 8
    planes = 1 if fr.flags & 64 else 0
    _call("ts_raster_bwd", lib.ts_raster_bwd_planes, fr.channels, split | (fr.flags & 8) | gen, fr.num_intersects, fr.cam,
          fr.tile_bins, fr.gaussian_ids_sorted, fr.splats, fr.background, fr.final_Ts, fr.final_index, fr.v_out_img,
          fr.v_out_depth if planes else None, planes, None, fr.clamp_mask, fr.partials, fr.row_flags, s)
    _call("ts_reduce_partials", lib.ts_reduce_partials, fr.n, fr.channels, 1 | split | gen, fr.num_tiles_hit,
          fr.cum_tiles_hit, fr.partials, fr.row_flags, fr.splats, fr.v_xy, fr.v_conic, fr.v_colors,
          fr.v_opacity, fr.v_depth if fr.channels == 4 else None, fr.sh_mask if fr.flags & 16 else None, s)


def _steps_bwd_params(lib, fr, s):
    _call("ts_sh_colors_bwd", lib.ts_sh_colors_bwd, fr.n, fr.sh_degree, fr.num_bases, fr.means, fr.origin,
          None if fr.flags & 16 else fr.sh_mask, fr.v_colors, fr.v_colors_dc,
          fr.v_colors_rest if fr.num_bases > 1 else None, s)
    _call("ts_project_bwd", lib.ts_project_bwd, fr.n, fr.means, fr.scales, fr.quats, fr.view34, fr.projview,
          fr.cam, 3, fr.radii, fr.v_xy, fr.v_depth, fr.v_conic, None, fr.v_means, fr.v_scales, fr.v_quats, s)


# FUSED ADAM (training.TrainStep; csrc/project.hip: FUSED ADAM): inside ``with fused_adam(optimizer):`` the backward pass of
# a single-GPU frame hands the optimiser's state to ts_frame_bwd_params_adam - the parameter-stage kernels update the
# six tensors from the gradients they hold in registers - and returns no parameter gradients (xys.grad is still set).
class _AdamHook:            # (a plain object, not thread-local: autograd runs a CUDA frame's backward on its device thread)
    opt = None


_adam_hook = _AdamHook()


class fused_adam:
    def __init__(self, optimizer):
        self.optimizer = optimizer

    def __enter__(self):
        self.prev = getattr(_adam_hook, "opt", None)
        _adam_hook.opt = self.optimizer
        return self

    def __exit__(self, *exc):
        _adam_hook.opt = self.prev
        return False


def _steps_bwd_params_adam(lib, fr, adam, s):
    _call("ts_sh_colors_bwd_adam", lib.ts_sh_colors_bwd_adam, fr.n, fr.sh_degree, fr.num_bases, fr.means, fr.origin,
          None if fr.flags & 16 else fr.sh_mask, fr.v_colors, fr.colors_dc,
          fr.colors_rest if fr.num_bases > 1 else None, ctypes.byref(adam), s)
    _call("ts_project_bwd_adam", lib.ts_project_bwd_adam, fr.n, fr.means, fr.scales, fr.quats, fr.view34, fr.projview,
          fr.cam, 3, fr.radii, fr.v_xy, fr.v_depth, fr.v_conic, fr.opacities, fr.v_opacity, ctypes.byref(adam), s)


class _RenderFrame(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, scales, quats, opacities, colors_dc, colors_rest, view34, projview,
                origin, background, fx, fy, width, height, sh_degree, with_depth, tile_rows, group, planes=False):
        F = _forward(means, scales, quats, opacities, colors_dc, colors_rest, view34, projview, origin,
                     background, fx, fy, width, height, sh_degree, with_depth, tile_rows, keep=True, planes=planes)
        ctx.frame, ctx.group = F, group
        ctx.opacity_shape = opacities.shape
        ctx.rest_shape = colors_rest.shape
        xys, radii = F.xys, F.radii
        ctx.xys_out = xys
        ctx.mark_non_differentiable(xys, radii)
        ctx.set_materialize_grads(False)      # no zero tensors for the two index outputs on the way into backward
        # the image must not stay reachable from ctx: it is the differentiable output, its grad_fn owns
        # ctx, and the cycle would keep every buffer of the frame alive until the garbage collector runs
        out, F.out_img = F.out_img, None
        depth, F.out_depth = F.out_depth, None          # None unless the frame was rendered as two planes
        return out, depth, xys, radii

    @staticmethod
    def backward(ctx, v_img, v_depth_img, _v_xys, _v_radii):
        _mark("bwd:enter")
        F = ctx.frame
        if F is None:
            raise RuntimeError("the frame's buffers were released by an earlier backward pass")
        fr, n, ch = F.fr, F.n, F.ch
        dev = F.wf.device
        f32 = dict(dtype=torch.float32, device=dev)
        lib = _lib.load()
        s = _stream(dev)
        if F.planes:                          # an absent plane's gradient is zero inside the kernel: no zero image
            v_img = None if v_img is None else _f32c(v_img)
            v_depth_img = None if v_depth_img is None else _f32c(v_depth_img)
        else:
            if v_img is None:                 # the image did not take part in the loss
                v_img = torch.zeros((F.cam.tile_rows and _stripe_rows(F.cam), F.w, ch), dtype=torch.float32, device=dev)
            v_img = _f32c(v_img)
        _mark("bwd:v_img contiguous")
        with torch.cuda.device(dev):
            _mark("bwd:device ctx")
            # v_xy | v_conic | v_colors | [v_depth] | v_opacity: ONE buffer, so that the multi-GPU path
            # can all-reduce it in place.  Without a collective v_xy and v_opacity are written straight into
            # the tensors handed out (xys.grad, the opacity gradient): no copies behind the kernels.
            single = ctx.group is None and DIRECT_GRADS
            flat = torch.empty((n * ((3 + ch) if single else (6 + ch)),), **f32)
            if single:
                v_xy = torch.empty((n, 2), **f32)
                v_opac = torch.empty(tuple(ctx.opacity_shape), **f32)
            F.segs = backward_segments(fr.cam, F.segs, F.total, dev.index)
            rows = max(F.total, 1) * (4 if (F.split and F.segs <= 1) else 1)
            partials = torch.empty((alloc_quantum(rows), _lib.PARTIAL_ROW_FLOATS), **f32)
            row_flags, fr.flag_gen = row_flags_for(dev, rows)
            _mark("bwd:flat+partials+flags")
            opt = getattr(_adam_hook, "opt", None)
            adam = opt.fused_begin(F.inputs[:6]) if (opt is not None and single) else None     # None: not this model's tensors
            if adam is None:
                v_means = torch.empty((n, 3), **f32)
                v_scales = torch.empty((n, 3), **f32)
                v_quats = torch.empty((n, 4), **f32)
                v_dc = torch.empty((n, 3), **f32)
                v_rest = torch.empty(tuple(ctx.rest_shape), **f32)
            p = flat.data_ptr()
            fr.v_out_img = None if v_img is None else v_img.data_ptr()
            fr.v_out_depth = None if (not F.planes or v_depth_img is None) else v_depth_img.data_ptr()
            fr.partials, fr.row_flags = partials.data_ptr(), row_flags.data_ptr()
            if single:
                fr.v_xy, fr.v_conic, fr.v_colors = v_xy.data_ptr(), p, p + 12 * n
                fr.v_depth = p + 24 * n if ch == 4 else None
                fr.v_opacity = v_opac.data_ptr()
            else:
                fr.v_xy, fr.v_conic, fr.v_colors = p, p + 8 * n, p + 20 * n
                fr.v_depth = p + 32 * n if ch == 4 else None
                fr.v_opacity = p + (20 + 4 * ch) * n
            if adam is None:
                fr.v_means, fr.v_scales, fr.v_quats = v_means.data_ptr(), v_scales.data_ptr(), v_quats.data_ptr()
                fr.v_colors_dc, fr.v_colors_rest = v_dc.data_ptr(), v_rest.data_ptr()
            _mark("bwd:allocated")
            timed = kernel_timer.enabled
            if timed:
                _steps_bwd_composite(lib, fr, s)
            else:
                _lib.check(lib.ts_frame_bwd_composite(ctypes.byref(fr), s), "ts_frame_bwd_composite")
            if ctx.group is not None:                          # tile-stripe sharding: sum over ranks
                with collective_timer.span(on_device=flat.is_cuda and dist.get_backend(ctx.group) != "gloo",
                                           label="all_reduce(2-D gradients)", nbytes=flat.numel() * flat.element_size()):
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=ctx.group)
            if adam is not None:
                if timed:
                    _steps_bwd_params_adam(lib, fr, adam, s)
                else:
                    _lib.check(lib.ts_frame_bwd_params_adam(ctypes.byref(fr), ctypes.byref(adam), s),
                               "ts_frame_bwd_params_adam")
            elif timed:
                _steps_bwd_params(lib, fr, s)
            else:
                _lib.check(lib.ts_frame_bwd_params(ctypes.byref(fr), s), "ts_frame_bwd_params")
        _mark("bwd:calls")
        if not single:      # compact copies: views would pin the n * (6 + ch) buffer
            v_xy = flat[:2 * n].view(n, 2).clone()
            v_opac = flat[(5 + ch) * n:].view(ctx.opacity_shape).clone()
        # what extras['xys'].grad holds in the reference (model_gaussian.py:130-132).  `xys` carries no
        # autograd edge on this fused path (retain_grad() is not needed and would raise); the gradient
        # is a compact copy - not a view that pins the n*(6+ch) buffer - and accumulates like a
        # retained grad when a second backward reaches the same frame.
        xo = ctx.xys_out
        xo.grad = v_xy if xo.grad is None else xo.grad + v_xy
        _mark("bwd:exit")
        if adam is not None:          # the parameters were updated in place: no gradients to hand out
            return (None,) * 19
        return (v_means, v_scales, v_quats, v_opac, v_dc, v_rest) + (None,) * 13


@torch.no_grad()
def render_view(model, view34: Tensor, projview: Tensor, origin: Tensor, fx: float, fy: float,
                width: int, height: int, with_depth: bool = True,
                tile_rows: Optional[Tuple[int, int]] = None, planes: bool = False):
    """Forward-only frame (the viewer's ``with torch.no_grad(): scene.render(camera)``,
    viewer.py:89-93): the kernels of ``render_frame`` without anything kept for a backward pass -
    no cov3d, clamp mask, final_Ts / final_index outputs, no autograd node.
    -> (image[rows, W, 3 or 4] with RGB clamped to <= 1, xys[N,2], radii[N]); with ``planes`` (and
    ``with_depth``) -> (rgb[rows, W, 3], depth[rows, W], xys, radii) as ``render_frame_planes``."""
    F = _forward(model.means.detach(), model.scales.detach(), model.quats.detach(), model.opacities.detach(),
                 model.colors_dc.detach(), model.colors_rest.detach(), view34, projview, origin,
                 model.background, fx, fy, width, height, int(model.active_sh_degree), with_depth, tile_rows,
                 keep=False, planes=planes)
    if F.planes:
        return F.out_img, F.out_depth, F.xys, F.radii
    return F.out_img, F.xys, F.radii


def render_frame(model, view34: Tensor, projview: Tensor, origin: Tensor, fx: float, fy: float,
                 width: int, height: int, with_depth: bool = True,
                 tile_rows: Optional[Tuple[int, int]] = None, group=None):
    """-> (image[rows, W, 3 or 4], xys[N,2], radii[N]).  Channels 0..2 are the RGB image already
    clamped to <= 1 (the adapter's rasterize.py:45, folded into the compositing kernels together with
    its backward); channel 3, when asked for, is the unclamped depth map.

    ``group`` (a process group, or ``dist.group.WORLD``) makes backward sum the 2-D gradients over
    the ranks rendering the other stripes; ``None`` = single GPU, no collective.
    """
    out, _, xys, radii = _RenderFrame.apply(model.means, model.scales, model.quats, model.opacities,
                                            model.colors_dc, model.colors_rest, view34, projview, origin,
                                            model.background, fx, fy, width, height, model.active_sh_degree,
                                            with_depth, tile_rows, group, False)
    return out, xys, radii


def render_frame_planes(model, view34: Tensor, projview: Tensor, origin: Tensor, fx: float, fy: float,
                        width: int, height: int, tile_rows: Optional[Tuple[int, int]] = None):
    """The RGB + depth frame as the adapter hands it out (rasterize.py:45, :51): -> (rgb[rows, W, 3] clamped to
    <= 1, depth[rows, W], xys[N,2], radii[N]), both images contiguous and both differentiable.  One 4-channel
    compositing pass as in ``render_frame``; only the image and its gradient are laid out as two planes, so a
    loss on ``rgb`` and one on ``depth`` cost no slicing of an interleaved image in either direction (at 3840x2160
    autograd's slice / zero-fill / add glue around an interleaved image was 0.3 ms of a 4.5 ms frame)."""
    return _RenderFrame.apply(model.means, model.scales, model.quats, model.opacities,
                              model.colors_dc, model.colors_rest, view34, projview, origin,
                              model.background, fx, fy, width, height, model.active_sh_degree,
                              True, tile_rows, None, True)
