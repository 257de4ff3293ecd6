#!/usr/bin/env python
"""Dynamic work counters of the compositing kernels on the frame path (developer tool, GPU box; needs a
library built with TS_EXTRA_HIPCC_FLAGS=-DTS_STATS=1 and TS_ALLOW_VARIANT_LIB=1).
usage: python tools/raster_stats.py [n] [width] [height] [depth 0/1]"""
import ctypes
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tinysplat_amd import _lib, frame
from tinysplat_amd.sharding import render_stripe
from tinysplat_amd.synthetic import loss_weights, make_scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
depth = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
dev = torch.device("cuda:0")
model, cam = make_scene(n, 3, w, h)
model = model.to(dev).requires_grad_(True)
w_rgb, w_d = (t.to(dev) for t in loss_weights(w, h))
lib = _lib.load()
out = (ctypes.c_ulonglong * 12)()


def run():
    o, (y0, y1), _ = render_stripe(model, cam, (w, h), dev, 0, 1, with_depth=depth)
    loss = (o[:, :, :3] * w_rgb).sum() + ((o[:, :, 3] * w_d).sum() if depth else 0.0)
    loss.backward()
    torch.cuda.synchronize()


run()
lib.ts_debug_stats(out, 1)
run()
lib.ts_debug_stats(out, 1)
b = frame.last_binning[0]
listed = int(b.tile_bins[:, 1].max().item())
names = ["fwd staged entries", "fwd block bodies", "bwd staged entries", "bwd block bodies entered",
         "bwd bodies with a valid lane", "bwd rows flushed", "bwd valid lanes", "bwd list entries walked",
         "bwd bodies valid in one half only", "bwd bodies valid in 1 quadrant", "bwd bodies valid in 2 quadrants",
         "bwd bodies valid in 3 quadrants"]
print(f"listed pairs {listed}, bounding-box pairs {int(b.num_intersects)}")
for nm, v in zip(names, out):
    print(f"{nm:32s} {v:12d}   per listed pair {v / listed:.3f}")
print(f"bwd lane utilisation in valid bodies {out[6] / max(1, 64 * out[4]):.3f}")
b = max(1, out[4])
print(f"bwd bodies whose valid pixels lie in ONE half of the 8x8 block (rows 0-3 or 4-7): {out[8] / b:.3f} of the valid bodies")
print(f"bwd bodies whose valid pixels lie in 1 / 2 / 3 / 4 of the four 4x4 quadrants: {out[9] / b:.3f} / {out[10] / b:.3f} / "
      f"{out[11] / b:.3f} / {1 - (out[9] + out[10] + out[11]) / b:.3f}")
print(f"   -> bodies if two half-block jobs shared an instruction stream (perfect pairing): {1 - 0.5 * out[8] / b:.3f} of today's;"
      f" quarter-wave jobs (perfect packing of 16-lane jobs): {(out[9] + 2 * out[10] + 3 * out[11] + 4 * (b - out[9] - out[10] - out[11])) / 4 / b:.3f}")
