"""Multi-GPU tile-stripe sharding of one frame (new design - the reference is
single-device, tinysplat/splatting/rasterize.py:17 hard-codes "cuda:0").

One process per GPU.  The Gaussian parameters are replicated; the tile grid is cut into
``world_size`` contiguous stripes of tile rows.  Each rank projects all Gaussians (96 N bytes, cheap),
bins / sorts / composites only its stripe, and in backward produces the per-Gaussian 2-D gradients
(v_xy, v_conic, v_colors, v_opacity: 40 N bytes) of its stripe's pixels.  Those are summed over
ranks with ONE all-reduce (RCCL over xGMI when the backend is "nccl") placed between the rasterizer
backward and the projection / SH backward, which then run replicated and leave identical, complete
parameter gradients on every rank.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

from ._comm import collective_timer
from .rasterizer import camera_on_device, project_args, raster_args, sh_args, tile_bounds


def stripe_rows(tile_rows_total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split of tile rows: the first (total % world) ranks get one more row."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world_size")
    q, r = divmod(tile_rows_total, world_size)
    r0 = rank * q + min(rank, r)
    return r0, r0 + q + (1 if rank < r else 0)


def _shared_flat_base(tensors):
    """The common 1-D base tensor if `tensors` are contiguous views that tile it exactly, in order."""
    base = getattr(tensors[0], "_base", None)
    if base is None or base.dim() != 1 or not base.is_contiguous():
        return None
    off = base.storage_offset()
    for t in tensors:
        if getattr(t, "_base", None) is not base or not t.is_contiguous() or t.storage_offset() != off:
            return None
        off += t.numel()
    return base if off == base.storage_offset() + base.numel() else None


class _SumGradsAcrossRanks(torch.autograd.Function):
    """Identity in forward; in backward packs the incoming gradients into one flat buffer, sums it
    over the process group with a single all-reduce, and unpacks."""

    @staticmethod
    def forward(ctx, group, *tensors):
        ctx.group = group
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        live = [g for g in grads if g is not None]
        if not live:
            return (None,) + grads
        base = _shared_flat_base(live)
        if base is not None:          # already one contiguous buffer (ops._RasterizeGaussians.backward)
            with collective_timer.span(on_device=base.is_cuda, label="all_reduce(2-D gradients)",
                                       nbytes=base.numel() * base.element_size()):
                dist.all_reduce(base, op=dist.ReduceOp.SUM, group=ctx.group)
            return (None,) + grads
        flat = torch.cat([g.reshape(-1) for g in live])
        with collective_timer.span(on_device=flat.is_cuda, label="all_reduce(2-D gradients)",
                                   nbytes=flat.numel() * flat.element_size()):
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=ctx.group)
        out, off = [], 0
        for g in grads:
            if g is None:
                out.append(None)
            else:
                out.append(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        return (None,) + tuple(out)


def sum_grads_across_ranks(tensors, group=None):
    return _SumGradsAcrossRanks.apply(group, *tensors)


def _colors(model, camera, ops, device, fused_colors):
    fused = getattr(ops, "sh_colors", None) if fused_colors else None
    if fused is not None:
        origin = camera_on_device(camera, device)[2]
        return fused(model.active_sh_degree, model.means, origin, model.colors_dc, model.colors_rest)
    colors = ops.spherical_harmonics(*sh_args(model, camera, device))
    return torch.clamp(colors + 0.5, min=0.0)


def render_stripe(model, camera, dims, device, rank: int = 0, world_size: int = 1, group=None,
                  tile_rows: Optional[Tuple[int, int]] = None, with_depth: bool = False,
                  collective: Optional[bool] = None):
    """This rank's tile-row stripe of the reference frame (rasterize.py:26-62) through the one-node
    HIP frame (frame.py): RGB (clamped to <= 1 by the kernels) and, with ``with_depth``, the depth
    output of rasterize.py:47-51 as channel 3, composited in the same pass.  In backward the flat
    per-Gaussian 2-D gradient buffer (v_xy | v_conic | v_colors[+v_depth] | v_opacity, (36 + 4 ch) N
    bytes) is summed over the ranks with ONE all-reduce between the compositing backward and the
    replicated SH / projection backward, so every rank ends with the complete parameter gradients.

    -> (out[rows, W, 3 or 4], (row_begin_px, row_end_px), xys[N,2]).  ``world_size == 1`` renders the
    whole frame without a collective; ``collective`` overrides whether the all-reduce runs.
    """
    from . import frame as _frame
    w, h = dims
    tby = tile_bounds(dims)[1]
    if tile_rows is None:
        tile_rows = stripe_rows(tby, world_size, rank)
    sharded = (world_size > 1) if collective is None else collective
    view, projview, origin = camera_on_device(camera, device)
    grp = (group if group is not None else dist.group.WORLD) if sharded else None
    out, xys, _ = _frame.render_frame(model, view[:3, :], projview, origin, camera.f_x, camera.f_y,
                                      w, h, with_depth, tile_rows, grp)
    y0 = 16 * tile_rows[0]
    return out, (y0, y0 + out.shape[0]), xys


def render_rgb_stripe(model, camera, dims, ops, device, rank: int = 0, world_size: int = 1,
                      group=None, tile_rows: Optional[Tuple[int, int]] = None,
                      fused_colors: bool = True, collective: Optional[bool] = None,
                      single_node: bool = True):
    """Steps 1-3 of the reference frame (rasterize.py:30-45: project, SH, clamp, rasterize RGB,
    clamp) for this rank's stripe.  Returns (rgb_stripe[rows,W,3], (row_begin_px, row_end_px), xys).

    world_size == 1 renders the whole frame and involves no collective.
    """
    w, h = dims
    tby = tile_bounds(dims)[1]
    if tile_rows is None:
        tile_rows = stripe_rows(tby, world_size, rank)
    sharded = (world_size > 1) if collective is None else collective   # run the grad all-reduce?
    kw = {"tile_rows": tile_rows} if (sharded or tile_rows != (0, tby)) else {}
    prep = fused_colors and getattr(ops, "fused_prep", False)
    if prep and single_node and getattr(ops, "render_frame", None) is not None:
        # the recipe below as one autograd node (frame.py); its backward all-reduces the same flat
        # 2-D gradient buffer between the compositing and the projection backward
        view, projview, origin = camera_on_device(camera, device)
        grp = (group if group is not None else dist.group.WORLD) if sharded else None
        out, xys, _ = ops.render_frame(model, view[:3, :], projview, origin, camera.f_x, camera.f_y,
                                       w, h, False, tile_rows, grp)
        y0 = 16 * tile_rows[0]
        return out, (y0, y0 + out.shape[0]), xys        # already clamped to <= 1 by the kernels
    if prep:      # exp / normalise / sigmoid folded into the kernels (see GaussianRasterizer)
        view, projview, _ = camera_on_device(camera, device)
        pa = [model.means, model.scales, 1., model.quats, view[:3, :], projview, camera.f_x,
              camera.f_y, w / 2, h / 2, h, w, tile_bounds(dims)]
        xys, depths, radii, conics, num_tiles, _ = ops.project_gaussians(
            *pa, log_scales=True, raw_quats=True, **kw)
    else:
        pa = project_args(model, camera, dims, device)
        xys, depths, radii, conics, num_tiles, _ = ops.project_gaussians(*pa, **kw)
    if xys.requires_grad:
        xys.retain_grad()
    colors = _colors(model, camera, ops, device, fused_colors)
    if prep:      # opacity logits go in as they are: no torch.sigmoid kernel
        ra = [xys, depths, radii, conics, num_tiles, colors, model.opacities, h, w, model.background]
        kw = dict(kw, logit_opacity=True)
    else:
        ra = raster_args(model, xys, depths, radii, conics, num_tiles, colors, dims)
    if sharded:
        ra[0], ra[3], ra[5], ra[6] = sum_grads_across_ranks((ra[0], ra[3], ra[5], ra[6]), group)
    rgb, _ = ops.rasterize_gaussians(*ra, **kw)
    rgb = torch.clamp(rgb, max=1.0)
    y0 = 16 * tile_rows[0]
    return rgb, (y0, y0 + rgb.shape[0]), xys
