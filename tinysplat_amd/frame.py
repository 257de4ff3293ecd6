"""Whole-frame autograd function: the render adapter's recipe (rasterize.py:26-62) as ONE
``torch.autograd.Function`` over the C ABI.

The three drop-in ops of ``ops.py`` each cost an autograd node, a dozen Python-level tensor
allocations and a binning-cache lookup per frame; on small scenes and on multi-GPU stripes that
host time (~0.5 ms) exceeds the GPU time.  This function enqueues the same kernels in the same order
(project -> scan -> colour stage -> pack -> binning -> composite; backward: composite -> reduce ->
[all-reduce across stripes] -> colour stage -> project) with the adapter's exp / normalise / sigmoid
folded in (``TS_PROJECT_*`` / ``TS_RASTER_LOGIT_OPACITY`` flags), one 4-channel compositing pass for
RGB + depth, tight tile lists, the adapter's clamp(max=1) inside the compositing kernels, and a single
autograd node.  Results are bitwise those of the op-by-op path
(tests/test_gpu_parity.py: test_one_node_frame_is_bitwise_the_fused_op_recipe).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from . import _lib
from .ops import (TileBinning, _IntersectionCount, _call, _camera, _f32c, _need_hip, _ptr, _stream, _stripe_rows,
                  _tile_bounds, deg_from_sh)

# Tight tile lists (see ts_bin_count): (Gaussian, tile) pairs that provably cannot reach alpha >= 1/255
# anywhere in the tile are dropped at binning time.  Results are bit-identical either way
# (tests/test_gpu_parity.py); TS_TIGHT_BINNING=0 restores gsplat's bounding-box lists for A/B timing.
TIGHT_BINNING = os.environ.get("TS_TIGHT_BINNING", "1") != "0"

# Launches with at most this many tiles (a stripe of a multi-GPU frame, a small image) composite with
# four waves per tile, one per 8x8 block (TS_RASTER_SPLIT_BLOCKS): an MI355X has 1024 SIMDs, and with
# one wave per tile such launches cannot hide any latency.  0 disables.
SPLIT_BLOCKS_BELOW = int(os.environ.get("TS_SPLIT_BLOCKS_BELOW", "1536"))

# binning of the most recent frame per device index (scene statistics for bench.py / tools)
last_binning = {}


class _RenderFrame(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, scales, quats, opacities, colors_dc, colors_rest, view34, projview,
                origin, background, fx, fy, width, height, sh_degree, with_depth, tile_rows, group):
        dev = _need_hip(means, scales, quats, opacities, colors_dc, colors_rest, view34, projview,
                        origin, background)
        n = means.shape[0]
        nb = colors_rest.shape[1] + 1
        if sh_degree < 0 or sh_degree > deg_from_sh(nb):
            raise ValueError("sh_degree exceeds the stored coefficients")
        means, scales, quats = _f32c(means), _f32c(scales), _f32c(quats)
        opacities, colors_dc, colors_rest = _f32c(opacities), _f32c(colors_dc), _f32c(colors_rest)
        view34, projview, origin = _f32c(view34), _f32c(projview), _f32c(origin)
        w, h = int(width), int(height)
        tb = _tile_bounds(h, w)
        cam = _camera(fx, fy, w / 2, h / 2, h, w, tb, 1.0, tile_rows=tile_rows)
        ch = 4 if with_depth else 3
        bg = _f32c(torch.cat([background, background[:1]]) if with_depth else background)
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        lib = _lib.load()
        s = _stream(dev)
        with torch.cuda.device(dev):
            xys = torch.empty((n, 2), **f32); depths = torch.empty((n,), **f32)
            radii = torch.empty((n,), **i32); conics = torch.empty((n, 3), **f32)
            nth = torch.empty((n,), **i32)
            _call("ts_project_fwd", lib.ts_project_fwd, n, _ptr(means), _ptr(scales), _ptr(quats),
                  _ptr(view34), _ptr(projview), cam, 3, _ptr(xys), _ptr(depths), _ptr(radii),
                  _ptr(conics), _ptr(nth), None, s)          # cov3d: the adapter discards it (rasterize.py:32)
            # binning, first half; the intersection count travels to the host while the kernels
            # that do not need it (colour stage, per-tile counts, offsets) run
            num_tiles = cam.tile_rows * cam.tile_bounds_x
            cum = torch.empty((n,), **i32)
            ws = torch.empty((int(lib.ts_scan_ws_ints(n)),), **i32)
            _call("ts_scan_tiles", lib.ts_scan_tiles, n, _ptr(nth), _ptr(cum), _ptr(ws), s)
            pending = _IntersectionCount(cum, dev)
            colors = torch.empty((n, 3), **f32)
            mask = torch.empty((n,), dtype=torch.uint8, device=dev)
            _call("ts_sh_colors_fwd", lib.ts_sh_colors_fwd, n, int(sh_degree), nb, _ptr(means),
                  _ptr(origin), _ptr(colors_dc), _ptr(colors_rest) if nb > 1 else None, _ptr(colors),
                  _ptr(mask), s)
            cols = torch.cat([colors, depths[:, None]], dim=1) if with_depth else colors
            bin_ws = torch.empty((int(lib.ts_bin_ws_ints(n, num_tiles)),), **i32)
            tile_bins = torch.empty((max(num_tiles, 1), 2), **i32)
            splats = torch.empty((max(n, 1), 12), **f32)
            _call("ts_pack_splats", lib.ts_pack_splats, n, ch, 1, _ptr(xys), _ptr(radii), _ptr(conics),
                  _ptr(cols), _ptr(opacities), _ptr(cum), cam, _ptr(splats), s)
            tight = _ptr(splats) if TIGHT_BINNING else None      # drop pairs that cannot contribute
            _call("ts_bin_count", lib.ts_bin_count, n, _ptr(xys), _ptr(radii), tight, cam, _ptr(bin_ws), s)
            _call("ts_tile_offsets", lib.ts_tile_offsets, n, num_tiles, _ptr(bin_ws), _ptr(tile_bins), s)
            total = pending.wait()                                # the one host sync of the path
            bucket_ids = torch.empty((max(total, 1),), **i32)
            ids = torch.empty((max(total, 1),), **i32)
            if total > 0:
                _call("ts_bin_scatter", lib.ts_bin_scatter, n, _ptr(xys), _ptr(radii), tight, cam,
                      _ptr(bin_ws), _ptr(bucket_ids), s)
                _call("ts_sort_tiles", lib.ts_sort_tiles, num_tiles, _ptr(tile_bins), _ptr(depths),
                      _ptr(bucket_ids), _ptr(ids), _ptr(bin_ws), s)
            # compositing
            rows = _stripe_rows(cam)
            out_img = torch.empty((rows, w, ch), **f32)
            final_Ts = torch.empty((rows, w), **f32)
            final_idx = torch.empty((rows, w), **i32)
            clamp_mask = torch.empty((rows, w), dtype=torch.uint8, device=dev)
            split = 4 if 0 < num_tiles <= SPLIT_BLOCKS_BELOW else 0      # TS_RASTER_SPLIT_BLOCKS
            _call("ts_raster_fwd", lib.ts_raster_fwd, ch, 2 | split, cam, _ptr(tile_bins), _ptr(ids), _ptr(splats),
                  _ptr(bg), _ptr(out_img), _ptr(final_Ts), _ptr(final_idx), _ptr(clamp_mask), s)
        b = TileBinning()
        b.cam, b.n, b.num_tiles, b.num_intersects = cam, n, num_tiles, total
        b.tile_bins, b.gaussian_ids_sorted, b.cum_tiles_hit, b.num_tiles_hit = tile_bins[:num_tiles], ids[:total], cum, nth
        last_binning[dev.index] = b
        ctx.cam, ctx.ch, ctx.n, ctx.nb, ctx.total, ctx.split = cam, ch, n, nb, total, split
        ctx.sh_degree, ctx.group = int(sh_degree), group
        ctx.opacity_shape = opacities.shape
        ctx.xys_out = xys
        ctx.save_for_backward(means, scales, quats, view34, projview, origin, radii, nth, cum,
                              tile_bins, ids, splats, bg, final_Ts, final_idx, mask, clamp_mask)
        ctx.mark_non_differentiable(xys, radii)
        return out_img, xys, radii

    @staticmethod
    def backward(ctx, v_img, _v_xys, _v_radii):
        (means, scales, quats, view34, projview, origin, radii, nth, cum, tile_bins, ids, splats, bg,
         final_Ts, final_idx, mask, clamp_mask) = ctx.saved_tensors
        dev, n, ch, cam, total = means.device, ctx.n, ctx.ch, ctx.cam, ctx.total
        f32 = dict(dtype=torch.float32, device=dev)
        lib = _lib.load()
        s = _stream(dev)
        v_img = _f32c(v_img)
        with torch.cuda.device(dev):
            flat = torch.empty((n * (6 + ch),), **f32)        # v_xy | v_conic | v_colors | v_opacity
            v_xy = flat[:2 * n].view(n, 2)
            v_conic = flat[2 * n:5 * n].view(n, 3)
            v_cols = flat[5 * n:(5 + ch) * n].view(n, ch)
            v_opac = flat[(5 + ch) * n:]
            rows = max(total, 1) * (4 if ctx.split else 1)
            partials = torch.empty((rows, 12), **f32)
            row_flags = torch.empty((rows,), dtype=torch.uint8, device=dev)
            _call("ts_raster_bwd", lib.ts_raster_bwd, ch, ctx.split, total, cam, _ptr(tile_bins), _ptr(ids),
                  _ptr(splats), _ptr(bg), _ptr(final_Ts), _ptr(final_idx), _ptr(v_img), None,
                  _ptr(clamp_mask), _ptr(partials), _ptr(row_flags), s)
            _call("ts_reduce_partials", lib.ts_reduce_partials, n, ch, 1 | ctx.split, _ptr(nth), _ptr(cum),
                  _ptr(partials), _ptr(row_flags), _ptr(splats), _ptr(v_xy), _ptr(v_conic),
                  _ptr(v_cols), _ptr(v_opac), s)
            if ctx.group is not None:                          # tile-stripe sharding: sum over ranks
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=ctx.group)
            if ch == 4:
                v_colors = v_cols[:, :3].contiguous()
                v_depth = v_cols[:, 3].contiguous()
            else:
                v_colors, v_depth = v_cols, torch.zeros((n,), **f32)
            v_dc = torch.empty((n, 3), **f32)
            v_rest = torch.empty((n, ctx.nb - 1, 3), **f32)
            _call("ts_sh_colors_bwd", lib.ts_sh_colors_bwd, n, ctx.sh_degree, ctx.nb, _ptr(means),
                  _ptr(origin), _ptr(mask), _ptr(v_colors), _ptr(v_dc),
                  _ptr(v_rest) if ctx.nb > 1 else None, s)
            v_means = torch.empty((n, 3), **f32)
            v_scales = torch.empty((n, 3), **f32)
            v_quats = torch.empty((n, 4), **f32)
            _call("ts_project_bwd", lib.ts_project_bwd, n, _ptr(means), _ptr(scales), _ptr(quats),
                  _ptr(view34), _ptr(projview), cam, 3, _ptr(radii), _ptr(v_xy), _ptr(v_depth),
                  _ptr(v_conic), None, _ptr(v_means), _ptr(v_scales), _ptr(v_quats), s)
        # what extras['xys'].grad holds in the reference (model_gaussian.py:130-132).  `xys` carries no
        # autograd edge on this fused path (retain_grad() is not needed and would raise); the gradient
        # is a compact copy - not a view that pins the n*(6+ch) buffer - and accumulates like a
        # retained grad when a second backward reaches the same frame.
        xo = ctx.xys_out
        xo.grad = v_xy.clone() if xo.grad is None else xo.grad + v_xy
        return (v_means, v_scales, v_quats, v_opac.view(ctx.opacity_shape), v_dc, v_rest) + (None,) * 12


@torch.no_grad()
def render_view(model, view34: Tensor, projview: Tensor, origin: Tensor, fx: float, fy: float,
                width: int, height: int, with_depth: bool = True,
                tile_rows: Optional[Tuple[int, int]] = None):
    """Forward-only frame (the viewer's ``with torch.no_grad(): scene.render(camera)``,
    viewer.py:89-93): the kernels of ``render_frame`` without anything kept for a backward pass -
    no cov3d, clamp mask, final_Ts / final_index outputs, no autograd node.
    -> (image[rows, W, 3 or 4] with RGB clamped to <= 1, xys[N,2], radii[N])."""
    ps = [model.means, model.scales, model.quats, model.opacities, model.colors_dc, model.colors_rest,
          view34, projview, origin, model.background]
    dev = _need_hip(*ps)
    means, scales, quats, opacities, colors_dc, colors_rest, view34, projview, origin, background = (
        _f32c(t.detach()) for t in ps)
    n, nb = means.shape[0], colors_rest.shape[1] + 1
    sh_degree = int(model.active_sh_degree)
    if sh_degree < 0 or sh_degree > deg_from_sh(nb):
        raise ValueError("sh_degree exceeds the stored coefficients")
    w, h = int(width), int(height)
    cam = _camera(fx, fy, w / 2, h / 2, h, w, _tile_bounds(h, w), 1.0, tile_rows=tile_rows)
    ch = 4 if with_depth else 3
    bg = _f32c(torch.cat([background, background[:1]]) if with_depth else background)
    f32 = dict(dtype=torch.float32, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    lib = _lib.load()
    s = _stream(dev)
    with torch.cuda.device(dev):
        xys = torch.empty((n, 2), **f32); depths = torch.empty((n,), **f32)
        radii = torch.empty((n,), **i32); conics = torch.empty((n, 3), **f32)
        nth = torch.empty((n,), **i32)
        _call("ts_project_fwd", lib.ts_project_fwd, n, _ptr(means), _ptr(scales), _ptr(quats),
              _ptr(view34), _ptr(projview), cam, 3, _ptr(xys), _ptr(depths), _ptr(radii),
              _ptr(conics), _ptr(nth), None, s)
        num_tiles = cam.tile_rows * cam.tile_bounds_x
        cum = torch.empty((n,), **i32)
        ws = torch.empty((int(lib.ts_scan_ws_ints(n)),), **i32)
        _call("ts_scan_tiles", lib.ts_scan_tiles, n, _ptr(nth), _ptr(cum), _ptr(ws), s)
        pending = _IntersectionCount(cum, dev)
        cols = torch.empty((n, ch), **f32)
        colors = cols if ch == 3 else torch.empty((n, 3), **f32)
        _call("ts_sh_colors_fwd", lib.ts_sh_colors_fwd, n, sh_degree, nb, _ptr(means), _ptr(origin),
              _ptr(colors_dc), _ptr(colors_rest) if nb > 1 else None, _ptr(colors), None, s)
        if ch == 4:
            torch.cat([colors, depths[:, None]], dim=1, out=cols)
        bin_ws = torch.empty((int(lib.ts_bin_ws_ints(n, num_tiles)),), **i32)
        tile_bins = torch.empty((max(num_tiles, 1), 2), **i32)
        splats = torch.empty((max(n, 1), 12), **f32)
        _call("ts_pack_splats", lib.ts_pack_splats, n, ch, 1, _ptr(xys), _ptr(radii), _ptr(conics),
              _ptr(cols), _ptr(opacities), _ptr(cum), cam, _ptr(splats), s)
        tight = _ptr(splats) if TIGHT_BINNING else None
        _call("ts_bin_count", lib.ts_bin_count, n, _ptr(xys), _ptr(radii), tight, cam, _ptr(bin_ws), s)
        _call("ts_tile_offsets", lib.ts_tile_offsets, n, num_tiles, _ptr(bin_ws), _ptr(tile_bins), s)
        total = pending.wait()
        bucket_ids = torch.empty((max(total, 1),), **i32)
        ids = torch.empty((max(total, 1),), **i32)
        if total > 0:
            _call("ts_bin_scatter", lib.ts_bin_scatter, n, _ptr(xys), _ptr(radii), tight, cam,
                  _ptr(bin_ws), _ptr(bucket_ids), s)
            _call("ts_sort_tiles", lib.ts_sort_tiles, num_tiles, _ptr(tile_bins), _ptr(depths),
                  _ptr(bucket_ids), _ptr(ids), _ptr(bin_ws), s)
        out_img = torch.empty((_stripe_rows(cam), w, ch), **f32)
        split = 4 if 0 < num_tiles <= SPLIT_BLOCKS_BELOW else 0
        _call("ts_raster_fwd", lib.ts_raster_fwd, ch, 2 | split, cam, _ptr(tile_bins), _ptr(ids), _ptr(splats),
              _ptr(bg), _ptr(out_img), None, None, None, s)
    b = TileBinning()
    b.cam, b.n, b.num_tiles, b.num_intersects = cam, n, num_tiles, total
    b.tile_bins, b.gaussian_ids_sorted, b.cum_tiles_hit, b.num_tiles_hit = tile_bins[:num_tiles], ids[:total], cum, nth
    last_binning[dev.index] = b
    return out_img, xys, radii


def render_frame(model, view34: Tensor, projview: Tensor, origin: Tensor, fx: float, fy: float,
                 width: int, height: int, with_depth: bool = True,
                 tile_rows: Optional[Tuple[int, int]] = None, group=None):
    """-> (image[rows, W, 3 or 4], xys[N,2], radii[N]).  Channels 0..2 are the RGB image already
    clamped to <= 1 (the adapter's rasterize.py:45, folded into the compositing kernels together with
    its backward); channel 3, when asked for, is the unclamped depth map.

    ``group`` (a process group, or ``dist.group.WORLD``) makes backward sum the 2-D gradients over
    the ranks rendering the other stripes; ``None`` = single GPU, no collective.
    """
    return _RenderFrame.apply(model.means, model.scales, model.quats, model.opacities,
                              model.colors_dc, model.colors_rest, view34, projview, origin,
                              model.background, fx, fy, width, height, model.active_sh_degree,
                              with_depth, tile_rows, group)
