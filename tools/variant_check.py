#!/usr/bin/env python
"""Developer tool (GPU box): times the frame path's C-ABI entries on one scene and compares the six parameter
gradients + image with a reference dump (written by the first run that finds none) - used to A/B library
variants built with TS_EXTRA_HIPCC_FLAGS (tools/ablate_frame.sh).
usage: python tools/variant_check.py <ref.pt> [n] [width] [height] [depth 0/1] [emulate_ranks] [rank]"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tinysplat_amd import ops
from tinysplat_amd.sharding import render_stripe
from tinysplat_amd.synthetic import loss_weights, make_scene

ref_path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
w = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
h = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
depth = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
ranks = int(sys.argv[6]) if len(sys.argv) > 6 else 1
rank = int(sys.argv[7]) if len(sys.argv) > 7 else 0
dev = torch.device("cuda:0")
model, cam = make_scene(n, 3, w, h)
model = model.to(dev).requires_grad_(True)
w_rgb, w_d = (t.to(dev) for t in loss_weights(w, h))


def frame():
    for p in model.parameters():
        p.grad = None
    out, (y0, y1), xys = render_stripe(model, cam, (w, h), dev, rank, ranks, with_depth=depth, collective=False)
    loss = (out[:, :, :3] * w_rgb[y0:y1]).sum()
    if depth:
        loss = loss + (out[:, :, 3] * w_d[y0:y1]).sum()
    loss.backward()
    return out


for _ in range(3):
    out = frame()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    frame()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 20 * 1e3
ops.kernel_timer.start()
for _ in range(5):
    frame()
t = ops.kernel_timer.stop()
cur = {"img": out.detach().cpu()}
for nm, p in zip(("means", "colors_dc", "colors_rest", "scales", "quats", "opacities"), model.parameters()):
    cur[nm] = p.grad.detach().cpu()
msg = ""
if os.path.exists(ref_path):
    ref = torch.load(ref_path)
    worst = 0.0
    for k, v in cur.items():
        d = (v.double() - ref[k].double()).abs().max().item() / max(1.0, ref[k].abs().max().item())
        worst = max(worst, d)
    msg = f"| max rel diff vs ref {worst:.2e}"
else:
    torch.save(cur, ref_path)
    msg = "| reference written"
print(f"frame={ms:.3f}ms " + " ".join(f"{k[3:]}={v[1]*1e3:.0f}" for k, v in sorted(t.items(), key=lambda kv: -kv[1][1])), msg)
