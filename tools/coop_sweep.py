#!/usr/bin/env python
"""Cooperative tiles of the forward compositing launch (csrc/raster.hip: COOPERATIVE TILES) on the headline frame: for
every C16 setting the frame time, the per-entry times of the compositing stage, and the results against C16 = 0
(image AND gradients bitwise: the forward pass leaves the same image, final_Ts, final_index and sorted lists).
Developer tool, GPU box.
usage: python tools/coop_sweep.py [--n N --width W --height H] [--settings "C16 ..."] [--depth] [--forward-only]"""
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
ap.add_argument("--settings", default="0 2 3 4 5 6 8 0")
ap.add_argument("--hybrid-from", type=int, default=None, help="frame.HYBRID_FROM (tiles from which a frame is hybrid)")
args = ap.parse_args()
dev = torch.device("cuda:0")
n, w, h = args.n, args.width, args.height
if args.hybrid_from is not None:
    frame.HYBRID_FROM = args.hybrid_from
model, cam = make_scene(n, args.sh, w, h, seed=0)
model = model.to(dev).requires_grad_(True)
w_rgb, w_d = (t.to(dev) for t in loss_weights(w, h))
w_rgbd = torch.cat([w_rgb, w_d.unsqueeze(-1)], dim=-1).contiguous()
params = list(model.parameters())
view34 = cam.view_matrix[:3, :].to(dev).contiguous()
projview = (cam.proj_matrix @ cam.view_matrix).to(dev).contiguous()
origin = cam.view_matrix[:3, 3].to(dev).contiguous()


def step():
    for p in params:
        p.grad = None
    out, xys, _ = render_frame(model, view34, projview, origin, cam.f_x, cam.f_y, w, h, with_depth=args.depth)
    out.backward(w_rgbd if args.depth else w_rgb)
    return out, xys


def run(c16):
    frame.HYBRID_COOP16 = c16
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
    c16 = int(tok)
    ms, ent, res = run(c16)
    line = (f"C16={c16:2d}  frame {ms:.3f} ms | raster_fwd {ent.get('raster_fwd', 0):.0f} raster_bwd "
            f"{ent.get('raster_bwd', 0):.0f} us")
    if base is None:
        base = res
    else:
        same = [torch.equal(a, b) for a, b in zip(res, base)]
        line += f" | image bitwise {same[0]}, gradients bitwise {all(same[1:])}"
        if not all(same):
            d = (res[0] - base[0]).abs()
            line += f" (image: {int((d > 0).sum())} values differ, max {d.max().item():.3e})"
    print(line, flush=True)
