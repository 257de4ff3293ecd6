#!/usr/bin/env python
"""Developer probe (GPU box): is the compositing time of config 3 set by the 2-rounds quantisation (8 160 tiles on
4 096 wave slots)?  Renders the SAME per-tile workload - Gaussians of config 3's pixel footprint at config 3's density -
on a frame with four times the tiles (4 M Gaussians, 3840x2160, scale_mult 0.5: ~8 rounds) and prints the kernels'
time per tile for both.  usage: python tools/quantization_probe.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tinysplat_amd import frame, ops
from tinysplat_amd.sharding import render_stripe
from tinysplat_amd.synthetic import loss_weights, make_scene

dev = torch.device("cuda:0")
for n, w, h, mult in ((1_000_000, 1920, 1080, 1.0), (4_000_000, 3840, 2160, 0.5), (2_000_000, 3840, 1080, 0.5)):
    # (the third: twice the tiles - 3840x1080 at the 4K focal length is not the same camera, so only compare per tile)
    model, cam = make_scene(n, 3, w, h, scale_mult=mult)
    model = model.to(dev).requires_grad_(True)
    w_rgb = loss_weights(w, h)[0].to(dev)

    def step():
        for p in model.parameters():
            p.grad = None
        o, _, _ = render_stripe(model, cam, (w, h), dev, 0, 1)
        torch.dot(o.reshape(-1), w_rgb.reshape(-1)).backward()

    for _ in range(3):
        step()
    ops.kernel_timer.start()
    for _ in range(5):
        step()
    t = ops.kernel_timer.stop()
    b = frame.last_binning[0]
    tiles = int(b.num_tiles)
    listed = int(b.tile_bins[:, 1].max().item())
    f, bw = t["ts_raster_fwd"][1] * 1e3, t["ts_raster_bwd"][1] * 1e3
    print(f"n={n} {w}x{h} mult={mult}: tiles {tiles}, listed pairs {listed} ({listed / tiles:.0f} per tile), "
          f"raster_fwd {f:.0f} us = {f / tiles * 1e3:.1f} ns/tile, raster_bwd {bw:.0f} us = {bw / tiles * 1e3:.1f} ns/tile", flush=True)
    del model
    torch.cuda.empty_cache()
