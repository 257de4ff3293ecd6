#!/usr/bin/env python
"""Work statistics of the compositing kernels on a sample of tiles (developer tool, GPU box):
how many (Gaussian, 8x8 block) pairs run the per-pixel body and how many lanes are valid there."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tinysplat_amd import ops
from tinysplat_amd.rasterizer import project_args, raster_args, sh_args
from tinysplat_amd.synthetic import make_scene

n, w, h = 1_000_000, 1920, 1080
dev = "cuda:0"
model, cam = make_scene(n, 3, w, h)
model = model.to(dev)
with torch.no_grad():
    xys, depths, radii, conics, nth, _ = ops.project_gaussians(*project_args(model, cam, (w, h), dev))
    col = torch.clamp(ops.spherical_harmonics(*sh_args(model, cam, dev)) + 0.5, min=0)
ra = raster_args(model, xys, depths, radii, conics, nth, col, (w, h))
ra[5] = ra[5].requires_grad_(True)
img, alpha = ops.rasterize_gaussians(*ra)
_, _, fT, fI = img.grad_fn.saved_tensors
from tinysplat_amd import frame as _frame
b = _frame.last_binning.get(0) or ops._bin_cache[0][1]
bins, ids = b.tile_bins.long(), b.gaussian_ids_sorted.long()
op = torch.sigmoid(model.opacities)[:, 0]
tbx = 120
g = torch.Generator().manual_seed(0)
tiles = torch.randint(0, bins.shape[0], (96,), generator=g).tolist()
tot = dict(entries=0, fwd_eval_entries=0, pairs_touch=0, pairs_body_fwd=0, lanes_hit_fwd=0, pairs_body_bwd=0, lanes_valid_bwd=0, surv=0)
for t in tiles:
    s, e = bins[t].tolist()
    if e <= s: continue
    ty, tx = divmod(t, tbx)
    gi = ids[s:e]
    py, px = torch.meshgrid(torch.arange(16, device=dev) + ty * 16, torch.arange(16, device=dev) + tx * 16, indexing="ij")
    inside = (py < h) & (px < w)
    dx = xys[gi, 0][None, None] - px[..., None].float()
    dy = xys[gi, 1][None, None] - py[..., None].float()
    c = conics[gi]
    sig = 0.5 * (c[:, 0] * dx * dx + c[:, 2] * dy * dy) + c[:, 1] * dx * dy
    a = torch.clamp(op[gi] * torch.exp(-sig), max=0.999)
    ok = (sig >= 0) & (a >= 1 / 255) & inside[..., None]            # [16,16,n]
    fidx = torch.full((16, 16), -1, device=dev, dtype=torch.long)
    yy, xx = py.clamp(max=h - 1), px.clamp(max=w - 1)
    fidx = torch.where(inside, fI[yy, xx].long(), fidx)
    idx = torch.arange(s, e, device=dev)
    valid_b = ok & (idx[None, None] <= fidx[..., None])
    # forward "not done" = idx <= stop index; approximate with idx <= fidx+1 region
    live_f = ok & (idx[None, None] <= (fidx[..., None] + 1))
    def blocks(m):   # [16,16,n] -> [4,64,n]
        return m.view(2, 8, 2, 8, -1).permute(0, 2, 1, 3, 4).reshape(4, 64, -1)
    vb, lf, okb = blocks(valid_b), blocks(live_f), blocks(ok)
    tot["entries"] += e - s
    tot["surv"] += int(okb.any(1).any(0).sum())
    tot["pairs_touch"] += int(okb.any(1).sum())
    tot["pairs_body_fwd"] += int(lf.any(1).sum())
    tot["lanes_hit_fwd"] += int(lf.sum())
    tot["pairs_body_bwd"] += int(vb.any(1).sum())
    tot["lanes_valid_bwd"] += int(vb.sum())
print(tot)
print("survivors/entry %.3f" % (tot["surv"] / tot["entries"]))
print("pairs(touch)/survivor %.2f" % (tot["pairs_touch"] / tot["surv"]))
print("fwd: pairs with a live hit / entry %.3f ; lane utilisation in those pairs %.3f" % (tot["pairs_body_fwd"] / tot["entries"], tot["lanes_hit_fwd"] / (64 * tot["pairs_body_fwd"])))
print("bwd: pairs with a valid lane / entry %.3f ; lane utilisation %.3f" % (tot["pairs_body_bwd"] / tot["entries"], tot["lanes_valid_bwd"] / (64 * tot["pairs_body_bwd"])))
