#!/usr/bin/env python
"""Times only the binning entries (developer tool). usage: time_binning.py [n] [w] [h]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tinysplat_amd import ops
from tinysplat_amd.rasterizer import project_args
from tinysplat_amd.synthetic import make_scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
dev = "cuda:0"
model, cam = make_scene(n, 3, w, h)
model = model.to(dev)
with torch.no_grad():
    xys, depths, radii, conics, nth, _ = ops.project_gaussians(*project_args(model, cam, (w, h), dev))
for _ in range(2):
    ops.bin_gaussians(xys, depths, radii, nth, h, w, use_cache=False)
ops.kernel_timer.start()
for _ in range(5):
    b = ops.bin_gaussians(xys, depths, radii, nth, h, w, use_cache=False)
t = ops.kernel_timer.stop()
print(" ".join(f"{k[3:]}={v[1]*1e3:.0f}us" for k, v in sorted(t.items(), key=lambda kv: -kv[1][1])), "| I", b.num_intersects)
