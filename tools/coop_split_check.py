#!/usr/bin/env python
"""Small launches: four waves per tile with SHARED staging (TS_HINT_COOP_SPLIT, csrc/raster.hip: COOPERATIVE TILES) against
the split-blocks forward pass, on a tile-row stripe of the headline frame (what one rank of G renders) or a small image:
per-entry times, and every output bit compared.  Developer tool, GPU box.
usage: python tools/coop_split_check.py [--ranks G --rank r] [--n N --width W --height H] [--depth]"""
import argparse
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tinysplat_amd import frame, ops
from tinysplat_amd.sharding import render_stripe
from tinysplat_amd.synthetic import loss_weights, make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--sh", type=int, default=3)
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--rank", type=int, default=3)
ap.add_argument("--depth", action="store_true")
ap.add_argument("--steps", type=int, default=30)
args = ap.parse_args()
dev = torch.device("cuda:0")
n, w, h = args.n, args.width, args.height
model, cam = make_scene(n, args.sh, w, h, seed=0)
model = model.to(dev).requires_grad_(True)
w_rgb, w_d = (t.to(dev) for t in loss_weights(w, h))
w_rgbd = torch.cat([w_rgb, w_d.unsqueeze(-1)], dim=-1).contiguous()
params = list(model.parameters())


def step():
    for p in params:
        p.grad = None
    out, (y0, y1), _ = render_stripe(model, cam, (w, h), dev, args.rank, args.ranks, with_depth=args.depth)
    out.backward((w_rgbd if out.shape[2] == 4 else w_rgb)[y0:y1])
    return out


def run(coop):
    frame.COOP_SPLIT = coop
    for _ in range(5):
        step()
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
    out = step()
    res = [out.detach().clone()] + [p.grad.detach().clone() for p in params]
    return ms, {k[3:]: v[1] * 1e3 for k, v in t.items()}, res


base = None
for coop in (False, True, False, True):
    ms, ent, res = run(coop)
    b = frame.last_binning[0]
    line = (f"coop_split={int(coop)} tiles {b.num_tiles} segments {frame.last_segments.get(0)}  step {ms:.3f} ms | raster_fwd "
            f"{ent.get('raster_fwd', 0):.0f} raster_bwd {ent.get('raster_bwd', 0):.0f} reduce_partials "
            f"{ent.get('reduce_partials', 0):.0f} us | kernels {sum(ent.values()):.0f} us")
    if base is None:
        base = res
    else:
        line += f" | all outputs bitwise {all(torch.equal(a, b_) for a, b_ in zip(res, base))}"
    print(line, flush=True)
