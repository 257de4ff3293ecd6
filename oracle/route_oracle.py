"""CPU restatement of the Gaussian-sharded frame's record routing (TEST INFRASTRUCTURE ONLY - nothing under
tinysplat_amd/ may import this; tests and tests' workers do).

What it restates is this build's own multi-GPU design (tinysplat_amd/csrc/shard.hip, tinysplat_amd/sharded.py);
the reference is single-device (/root/reference/tinysplat/splatting/rasterize.py:17), so there is no reference
counterpart and nothing to pin against: the contract is SURVEY.md 8(e) E2-E4 - the stripes of a sharded frame
tile the single-process image, and the gradients equal the single-process ones to rounding - and that is what
the tests built on this file check, on CPU with the oracle ops (tests/test_sharded_cpu.py, gloo world 2) and on
the GPU against the HIP kernels (tests/test_gpu_sharded.py).

Routing rule: a Gaussian with radius > 0 whose tile box (gsplat's get_tile_bbox: trunc then clamp, SURVEY
App. A.1-6) is non-empty sends one record to every rank whose stripe of tile rows intersects the box's rows;
records for one destination are ordered by ascending Gaussian index.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
from torch import Tensor

from . import gsplat_oracle as O


def dest_ranges(xys: Tensor, radii: Tensor, dims: Tuple[int, int], stripes: Sequence[int]):
    """-> (d0[N], d1[N]) destination ranks of every Gaussian (d0 > d1: none)."""
    w, h = dims
    tb = ((w + 15) // 16, (h + 15) // 16, 1)
    minx, miny, maxx, maxy = O.tile_bbox(xys.detach().to(torch.float32), radii.to(torch.float32), tb)
    lo = torch.clamp(miny, min=int(stripes[0]))
    hi = torch.clamp(maxy, max=int(stripes[-1]))
    live = (radii > 0) & (maxx > minx) & (maxy > miny) & (hi > lo)
    edges = torch.tensor(list(stripes[1:-1]), dtype=torch.int32)
    d0 = torch.bucketize(lo, edges, right=True)
    d1 = torch.bucketize(hi - 1, edges, right=True)
    d0 = torch.where(live, d0, torch.ones_like(d0))
    d1 = torch.where(live, d1, torch.zeros_like(d1))
    return d0, d1


def route(xys: Tensor, radii: Tensor, dims: Tuple[int, int], stripes: Sequence[int]) -> List[Tensor]:
    """Ascending index lists, one per destination rank."""
    d0, d1 = dest_ranges(xys, radii, dims, stripes)
    return [torch.nonzero((d0 <= d) & (d <= d1))[:, 0] for d in range(len(stripes) - 1)]


def import_tiles_hit(xys: Tensor, radii: Tensor, dims: Tuple[int, int], tile_rows) -> Tensor:
    w, h = dims
    return O.stripe_tiles_hit(xys, radii, ((w + 15) // 16, (h + 15) // 16, 1), tile_rows)
