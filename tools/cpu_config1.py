#!/usr/bin/env python
"""BASELINE config 1: 10k random Gaussians, SH degree 0, 256x256, forward RGB with the CPU oracle."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oracle import gsplat_oracle as O
from tinysplat_amd.rasterizer import project_args, raster_args, sh_args
from tinysplat_amd.synthetic import make_scene
threads = max(1, min(16, len(os.sched_getaffinity(0))))
torch.set_num_threads(threads)
n, w, h = 10_000, 256, 256
model, cam = make_scene(n, 0, w, h, seed=0)
def fwd():
    with torch.no_grad():
        xys, depths, radii, conics, nth, _ = O.project_gaussians(*project_args(model, cam, (w, h), "cpu"))
        col = torch.clamp(O.spherical_harmonics(*sh_args(model, cam, "cpu")) + 0.5, min=0)
        img, _ = O.rasterize_gaussians(*raster_args(model, xys, depths, radii, conics, nth, col, (w, h)))
    return int(nth.sum())
fwd(); fwd()
ts = []
for _ in range(5):
    t = time.perf_counter(); I = fwd(); ts.append(time.perf_counter() - t)
med = sorted(ts)[2]
print(f"config1 oracle CPU fwd: {med*1e3:.1f} ms/frame, I={I}, {n*w*h/med:.3e} Gaussians*pixels/s, threads={threads}")
