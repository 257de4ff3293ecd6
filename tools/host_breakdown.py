#!/usr/bin/env python
"""Where the host time of one frame goes (developer tool): marks inside frame.py + around the step."""
import collections, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tinysplat_amd import frame
from tinysplat_amd.sharding import render_stripe
from tinysplat_amd.synthetic import loss_weights, make_scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
w, h, dev = 1920, 1080, torch.device("cuda:0")
model, cam = make_scene(n, 3, w, h)
model = model.to(dev).requires_grad_(True)
w_rgb = loss_weights(w, h)[0].to(dev)
def step():
    frame._mark("step:begin")
    for p in model.parameters(): p.grad = None
    frame._mark("step:grads cleared")
    rgb, (y0, y1), _ = render_stripe(model, cam, (w, h), dev, 0, 1)
    frame._mark("step:rendered")
    loss = (rgb * w_rgb).sum()
    frame._mark("step:loss")
    loss.backward()
    frame._mark("step:backward done")
for _ in range(10): step()
torch.cuda.synchronize()
frame.TRACE = []
for _ in range(100): step()
torch.cuda.synchronize()
marks, frame.TRACE = frame.TRACE, None
acc = collections.OrderedDict()
for (l0, t0), (l1, t1) in zip(marks, marks[1:]):
    if l1 == "step:begin":
        continue
    acc[l1] = acc.get(l1, 0.0) + (t1 - t0)
tot = sum(acc.values())
for k, v in acc.items():
    print(f"{k:28s} {1e6 * v / 100:8.1f} us")
print(f"{'sum':28s} {1e6 * tot / 100:8.1f} us per frame")
st0 = torch.cuda.memory_stats()
for _ in range(50): step()
torch.cuda.synchronize()
st1 = torch.cuda.memory_stats()
for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "allocation.all.allocated", "segment.all.allocated", "num_sync_all_streams"):
    if k in st0:
        print(k, st1[k] - st0[k], "per 50 frames;", "now", st1[k])
print("reserved MB", st1["reserved_bytes.all.current"] / 1e6, "allocated MB", st1["allocated_bytes.all.current"] / 1e6)
