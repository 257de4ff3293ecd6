#!/usr/bin/env python
"""Host (enqueue) time vs GPU time of one frame (developer tool)."""
import sys, time, cProfile, pstats
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tinysplat_amd import ops
from tinysplat_amd.rasterizer import GaussianRasterizer
from tinysplat_amd.sharding import render_stripe
from tinysplat_amd.synthetic import loss_weights, make_scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
w, h, dev = 1920, 1080, torch.device("cuda:0")
model, cam = make_scene(n, 3, w, h)
model = model.to(dev).requires_grad_(True)
w_rgb = loss_weights(w, h)[0].to(dev)
ad = GaussianRasterizer(model, None, device=dev)
def step():
    for p in model.parameters(): p.grad = None
    rgb, (y0, y1), _ = render_stripe(model, cam, (w, h), dev, 0, 1)
    (rgb * w_rgb).sum().backward()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"n={n}: host enqueue {1e3*(t1-t0)/50:.3f} ms/frame, total {1e3*(t2-t0)/50:.3f} ms/frame")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
