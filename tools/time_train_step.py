#!/usr/bin/env python
"""Times the full training step (render RGB+depth -> L1+DSSIM (+depth L1) -> backward -> Adam) on the
config-3 scene (developer tool, GPU box)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tinysplat_amd import ops
from tinysplat_amd.synthetic import make_scene
from tinysplat_amd.training import TrainStep
n, w, h = 1_000_000, 1920, 1080
dev = "cuda:0"
model, cam = make_scene(n, 3, w, h)
model = model.to(dev)
g = torch.Generator().manual_seed(1)
tgt = torch.rand(h, w, 3, generator=g).to(dev)
tgt_d = (2 + 8 * torch.rand(h, w, generator=g)).to(dev)
step = TrainStep(model, dev)
for _ in range(3):
    step(cam, tgt, tgt_d)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for _ in range(K):
    out = step(cam, tgt, tgt_d)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
ops.kernel_timer.start()
for _ in range(5):
    step(cam, tgt, tgt_d)
t = ops.kernel_timer.stop()
print(f"train step: {dt*1e3:.3f} ms ({1/dt:.0f} steps/s), loss {out['loss'].item():.4f}")
print(" ".join(f"{k[3:]}={v[1]*1e3:.0f}us" for k, v in sorted(t.items(), key=lambda kv: -kv[1][1])))
