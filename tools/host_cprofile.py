#!/usr/bin/env python
"""cProfile of the single-GPU step's host side (developer tool): top functions by own time, per step.
usage: host_cprofile.py [n]"""
import cProfile
import pstats
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
from tinysplat_amd.sharding import render_stripe
from tinysplat_amd.synthetic import loss_weights, make_scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
w, h, dev = 1920, 1080, torch.device("cuda:0")
bench.bind_to_gpu_numa_node(0)
model, cam = make_scene(n, 3, w, h)
model = model.to(dev).requires_grad_(True)
w_rgb = loss_weights(w, h)[0].to(dev)
params = list(model.parameters())


def step():
    for p in params:
        p.grad = None
    rgb, _, _ = render_stripe(model, cam, (w, h), dev, 0, 1)
    torch.autograd.backward([rgb], [w_rgb])


for _ in range(50):
    step()
torch.cuda.synchronize()
K = 300
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:28]
print(f"n = {n}: top functions by own time, us per step (calls per step)")
for (fn, line, name), (cc, nc, tt, ct, _) in rows:
    print(f"{1e6 * tt / K:8.1f} own {1e6 * ct / K:8.1f} cum {nc / K:6.1f}  {name}  ({Path(fn).name}:{line})")
