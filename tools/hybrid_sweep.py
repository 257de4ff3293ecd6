#!/usr/bin/env python
"""Hybrid compositing launches (csrc/raster.hip: HYBRID LAUNCH) on the headline frame: for every (S, W16) setting the
frame time, the per-entry times of the compositing stage, and the results against the uncut frame (image bitwise,
gradients to rounding).  Developer tool, GPU box.
usage: python tools/hybrid_sweep.py [--n N --width W --height H] [--settings "S:W16 ..."] [--depth]"""
import argparse
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tinysplat_amd import frame, ops
from tinysplat_amd.frame import render_frame
from tinysplat_amd.synthetic import loss_weights, make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--sh", type=int, default=3)
ap.add_argument("--depth", action="store_true")
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--settings", default="1:0 4:10 4:8 4:12 2:8 2:10 8:12 3:10")
args = ap.parse_args()
dev = torch.device("cuda:0")
n, w, h = args.n, args.width, args.height
model, cam = make_scene(n, args.sh, w, h, seed=0)
model = model.to(dev).requires_grad_(True)
w_rgb, w_d = (t.to(dev) for t in loss_weights(w, h))
params = list(model.parameters())
view34 = cam.view_matrix[:3, :].to(dev).contiguous()
projview = (cam.proj_matrix @ cam.view_matrix).to(dev).contiguous()
origin = cam.view_matrix[:3, 3].to(dev).contiguous()


def step():
    for p in params:
        p.grad = None
    out, xys, _ = render_frame(model, view34, projview, origin, cam.f_x, cam.f_y, w, h, with_depth=args.depth)
    if args.depth:
        loss = (out[:, :, :3] * w_rgb).sum() + (out[:, :, 3] * w_d).sum()
    else:
        loss = torch.dot(out.reshape(-1), w_rgb.reshape(-1))
    loss.backward()
    return out, xys


def run(S, W16):
    frame.HYBRID_SEGS, frame.HYBRID_WHOLE16 = S, max(W16, 1)
    for _ in range(5):
        out, xys = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    ops.kernel_timer.start()
    for _ in range(8):
        step()
    t = ops.kernel_timer.stop()
    out, xys = step()
    res = [out.detach().clone(), xys.grad.detach().clone()] + [p.grad.detach().clone() for p in params]
    ent = {k[3:]: v[1] * 1e3 for k, v in t.items()}
    return ms, ent, res


base = None
for tok in args.settings.split():
    S, W16 = (int(x) for x in tok.split(":"))
    ms, ent, res = run(S, W16)
    line = (f"S={S} W16={W16:2d}  frame {ms:.3f} ms | raster_fwd {ent.get('raster_fwd', 0):.0f} raster_bwd "
            f"{ent.get('raster_bwd', 0):.0f} reduce_partials {ent.get('reduce_partials', 0):.0f} us")
    if base is None:
        base = res
    else:
        img_same = torch.equal(res[0], base[0])
        worst = 0.0
        for a, b in zip(res[1:], base[1:]):
            worst = max(worst, ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item())
        line += f" | image bitwise {img_same}, grads vs first setting: max |d| / max |ref| = {worst:.2e}"
    print(line, flush=True)
