#!/usr/bin/env python
"""Times the C-ABI entries of one config-3 frame in isolation (developer tool, GPU box).
usage: python tools/time_raster.py [n] [width] [height] [scale_mult]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tinysplat_amd import ops
from tinysplat_amd.rasterizer import project_args, raster_args, sh_args
from tinysplat_amd.synthetic import loss_weights, make_scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
mult = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
dev = "cuda:0"
model, cam = make_scene(n, 3, w, h, scale_mult=mult)
model = model.to(dev).requires_grad_(True)
w_rgb = loss_weights(w, h)[0].to(dev)


def frame():
    for p in model.parameters():
        p.grad = None
    xys, depths, radii, conics, nth, _ = ops.project_gaussians(*project_args(model, cam, (w, h), dev))
    col = torch.clamp(ops.spherical_harmonics(*sh_args(model, cam, dev)) + 0.5, min=0)
    img, _ = ops.rasterize_gaussians(*raster_args(model, xys, depths, radii, conics, nth, col, (w, h)))
    (torch.clamp(img, max=1.0) * w_rgb).sum().backward()


for _ in range(3):
    frame()
ops.kernel_timer.start()
for _ in range(10):
    frame()
t = ops.kernel_timer.stop()
tot = sum(v[1] for v in t.values())
print(" ".join(f"{k[3:]}={v[1]*1e3:.0f}us" for k, v in sorted(t.items(), key=lambda kv: -kv[1][1])), f"| sum={tot:.3f}ms")
