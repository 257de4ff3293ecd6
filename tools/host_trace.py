#!/usr/bin/env python
"""Per-frame host timeline of the single-GPU step (developer tool): the marks of frame.py per frame - median, p90 and
max over the frames, not a mean that one stall dominates - with the process bound to the GPU's NUMA node as bench.py
binds it (or not: --no-bind), with Python's cyclic collector on or off (--no-gc), and the step's wall time beside the
GPU's own time per step (events around every step).
usage: host_trace.py [n] [--no-bind] [--no-gc] [--frames K]"""
import gc
import statistics
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
from tinysplat_amd import frame
from tinysplat_amd.sharding import render_stripe
from tinysplat_amd.synthetic import loss_weights, make_scene

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 1_000_000
frames = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 200
w, h, dev = 1920, 1080, torch.device("cuda:0")
bound = None if "--no-bind" in sys.argv else bench.bind_to_gpu_numa_node(0)
if "--no-gc" in sys.argv:
    gc.disable()
model, cam = make_scene(n, 3, w, h)
model = model.to(dev).requires_grad_(True)
w_rgb = loss_weights(w, h)[0].to(dev)
params = list(model.parameters())


def step():
    frame._mark("step:begin")
    for p in params:
        p.grad = None
    rgb, (y0, y1), _ = render_stripe(model, cam, (w, h), dev, 0, 1)
    frame._mark("step:rendered")
    torch.autograd.backward([rgb], [w_rgb])
    frame._mark("step:end")


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(frames):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / frames
frame.TRACE = []
t0 = time.perf_counter()
for _ in range(frames):
    step()
torch.cuda.synchronize()
wall_traced = (time.perf_counter() - t0) / frames
marks, frame.TRACE = frame.TRACE, None
per = {}
order = []
frame_host = []
begin = None
for (l0, t0_), (l1, t1_) in zip(marks, marks[1:]):
    if l0 == "step:begin":
        begin = t0_
    if l1 == "step:begin":
        continue
    if l1 not in per:
        per[l1] = []
        order.append(l1)
    per[l1].append(1e6 * (t1_ - t0_))
    if l1 == "step:end" and begin is not None:
        frame_host.append(1e6 * (t1_ - begin))
print(f"n = {n}, {frames} frames, NUMA cpulist {bound}, gc {'off' if not gc.isenabled() else 'on'}")
print(f"wall per step {1e3 * wall:.4f} ms (with the marks recording: {1e3 * wall_traced:.4f} ms)")
print(f"{'mark (time since the previous one)':44s} {'median':>8s} {'p90':>8s} {'max':>9s}  us")
for k in order:
    v = sorted(per[k])
    print(f"{k:44s} {statistics.median(v):8.1f} {v[int(0.9 * (len(v) - 1))]:8.1f} {v[-1]:9.1f}")
v = sorted(frame_host)
print(f"{'host time inside step()':44s} {statistics.median(v):8.1f} {v[int(0.9 * (len(v) - 1))]:8.1f} {v[-1]:9.1f}")
