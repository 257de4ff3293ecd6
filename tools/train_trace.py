#!/usr/bin/env python
"""Host timeline of the TRAINING step (developer tool): frame.py's marks per step over a long run, in windows of 100
steps - where a slow phase of the run spends its host time.  usage: train_trace.py [steps]"""
import statistics
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
from tinysplat_amd import frame
from tinysplat_amd.synthetic import make_scene
from tinysplat_amd.training import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 800
n, w, h, dev = 1_000_000, 1920, 1080, torch.device("cuda:0")
bench.bind_to_gpu_numa_node(0)
model, cam = make_scene(n, 3, w, h)
model = model.to(dev)
g = torch.Generator().manual_seed(2)
tgt = torch.rand(h, w, 3, generator=g).to(dev)
tgt_d = (2.0 + 8.0 * torch.rand(h, w, generator=g)).to(dev)
trainer = TrainStep(model, dev)
for _ in range(5):
    trainer(cam, tgt, tgt_d)
torch.cuda.synchronize()
frame.TRACE = []
walls = []
t_prev = time.perf_counter()
for i in range(steps):
    frame._mark("step:begin")
    trainer(cam, tgt, tgt_d)
    frame._mark("step:end")
    if (i + 1) % 100 == 0:
        torch.cuda.synchronize()
        t = time.perf_counter()
        walls.append((t - t_prev) / 100 * 1e3)
        t_prev = t
marks, frame.TRACE = frame.TRACE, None
print("wall ms/step per window of 100 steps:", " ".join(f"{v:.3f}" for v in walls))
per_win = {}
win = -1
for (l0, t0), (l1, t1) in zip(marks, marks[1:]):
    if l0 == "step:begin":
        win += 1
    if l1 == "step:begin":
        continue
    per_win.setdefault(win // 100, {}).setdefault(l1, []).append(1e6 * (t1 - t0))
labels = list(per_win[0].keys())
print(f"{'mark (median us per step)':36s}" + "".join(f"{'w' + str(k):>8s}" for k in sorted(per_win)))
for lab in labels:
    print(f"{lab:36s}" + "".join(f"{statistics.median(per_win[k].get(lab, [0.0])):8.1f}" for k in sorted(per_win)))
st = torch.cuda.memory_stats()
print("num_device_alloc", st["num_device_alloc"], "num_device_free", st["num_device_free"], "retries", st["num_alloc_retries"],
      "reserved MB", st["reserved_bytes.all.current"] / 1e6)
