#!/usr/bin/env python
"""Forward-only check of the cooperative tiles against the whole-tile waves (developer tool, GPU box)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tinysplat_amd import frame
from tinysplat_amd.frame import render_frame
from tinysplat_amd.synthetic import make_scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
dev = torch.device("cuda:0")
frame.HYBRID_FROM = 1
frame.SPLIT_BLOCKS_BELOW = 0
model, cam = make_scene(n, 3, w, h, seed=0)
model = model.to(dev)
view34 = cam.view_matrix[:3, :].to(dev).contiguous()
projview = (cam.proj_matrix @ cam.view_matrix).to(dev).contiguous()
origin = cam.view_matrix[:3, 3].to(dev).contiguous()
base = None
for c16 in (0, 1, 4, 8, 15):
    frame.HYBRID_COOP16 = c16
    with torch.no_grad():
        out, _, _ = render_frame(model, view34, projview, origin, cam.f_x, cam.f_y, w, h, with_depth=False)
    torch.cuda.synchronize()
    print("C16", c16, "done", flush=True)
    if base is None:
        base = out.clone()
    else:
        d = (out - base).abs()
        print("   image bitwise", torch.equal(out, base), "differ", int((d > 0).sum()), "max", d.max().item(), flush=True)
