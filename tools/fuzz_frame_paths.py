#!/usr/bin/env python
"""Randomised self-consistency sweep on the GPU (developer tool): for random scenes / image sizes the
one-node frame path must equal the op-by-op fused recipe bitwise (split mapping off), tight lists must
equal bounding-box lists bitwise, the split mapping must reproduce the image bitwise and the gradients
to rounding, and two stripes must tile the frame."""
import random
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from tinysplat_amd import frame
from tinysplat_amd.rasterizer import GaussianRasterizer
from tinysplat_amd.sharding import render_rgb_stripe, stripe_rows
from tinysplat_amd.synthetic import make_scene

DEV = torch.device("cuda:0")


def run(model, cam, w, h, sh, wr, wd, single_node=True):
    md = model.to(DEV).requires_grad_(True)
    r = GaussianRasterizer(md, None, device=DEV)
    r.single_node = single_node
    rgb, ex = r(cam, (w, h), sh)
    ((rgb * wr).sum() + (ex["depth"] * wd).sum()).backward()
    return [rgb.detach(), ex["depth"].detach(), ex["xys"].grad] + [p.grad for p in md.parameters()], md, r


def main(cases=16, seed=0):
    rnd = random.Random(seed)
    keep_split, bad = frame.SPLIT_BLOCKS_BELOW, 0
    for c in range(cases):
        n = rnd.choice([1, 7, 300, 5000, 40000, 150000])
        w, h = rnd.randint(17, 900), rnd.randint(17, 600)
        sh = rnd.randint(0, 3)
        mult = rnd.choice([0.5, 1.0, 3.0, 8.0])
        model, cam = make_scene(n, sh, w, h, seed=100 + c, scale_mult=mult)
        if rnd.random() < 0.5:
            model.opacities = torch.empty(n, 1).uniform_(-6.0, 9.0)
        g = torch.Generator().manual_seed(c)
        wr, wd = torch.rand(h, w, 3, generator=g).to(DEV), torch.rand(h, w, generator=g).to(DEV)
        msgs = []
        frame.SPLIT_BLOCKS_BELOW, frame.TIGHT_BINNING = 0, True
        # (bitwise comparisons across list shapes need the uncut backward pass: a hybrid launch - from 1 537 tiles - cuts
        # lists at boundaries that follow their length, and its gradients equal the uncut pass's to rounding)
        keep_segs, frame.HYBRID_SEGS = frame.HYBRID_SEGS, 1
        base, md, r = run(model, cam, w, h, sh, wr, wd)
        ops_path, _, _ = run(model, cam, w, h, sh, wr, wd, single_node=False)
        if not all(torch.equal(a, b) for a, b in zip(base, ops_path)):
            msgs.append("one-node != op-by-op")
        frame.TIGHT_BINNING = False
        bbox, _, _ = run(model, cam, w, h, sh, wr, wd)
        frame.TIGHT_BINNING = True
        if not all(torch.equal(a, b) for a, b in zip(base, bbox)):
            msgs.append("tight != bbox")
        frame.SPLIT_BLOCKS_BELOW = 1 << 30
        split, _, _ = run(model, cam, w, h, sh, wr, wd)
        if not (torch.equal(split[0], base[0]) and torch.equal(split[1], base[1])):
            msgs.append("split image differs")
        for a, b in zip(split[2:], base[2:]):
            if a.numel() and (a - b).abs().max().item() > 2e-5 * max(1.0, b.abs().max().item()):   # summation order: 4 partial rows per pair
                msgs.append(f"split grads differ: max |diff| {(a - b).abs().max().item():.3e} at scale "
                            f"{b.abs().max().item():.3e}, shape {tuple(a.shape)}")
                break
        frame.HYBRID_SEGS = keep_segs
        # cooperative tiles: in place of the split forward pass (every bit of the split frame) ...
        keep_cs = frame.COOP_SPLIT
        frame.COOP_SPLIT = not keep_cs
        other, _, _ = run(model, cam, w, h, sh, wr, wd)
        frame.COOP_SPLIT = keep_cs
        if not all(torch.equal(a, b) for a, b in zip(split, other)):
            msgs.append("cooperative != split forward pass")
        frame.SPLIT_BLOCKS_BELOW = 0
        # ... and in the tail of a one-wave-per-tile launch, with and without the hybrid backward launch (forced on)
        keep_h = (frame.HYBRID_FROM, frame.HYBRID_MID_FROM, frame.HYBRID_COOP16, frame.HYBRID_SEGS, frame.WIDE_TILES)
        frame.HYBRID_FROM, frame.HYBRID_MID_FROM, frame.WIDE_TILES = 1, 1 << 30, 0
        for segs in (1, 8):
            frame.HYBRID_SEGS = segs
            res = []
            for c16 in (0, rnd.choice([1, 3, 7, 15])):
                frame.HYBRID_COOP16 = c16
                res.append(run(model, cam, w, h, sh, wr, wd)[0])
            if not all(torch.equal(a, b) for a, b in zip(res[0], res[1])):
                msgs.append(f"cooperative tiles change bits (segments {segs})")
            if segs == 1 and not all(torch.equal(a, b) for a, b in zip(res[0], base)):
                msgs.append("forced launch shape without segments != base")
        frame.HYBRID_FROM, frame.HYBRID_MID_FROM, frame.HYBRID_COOP16, frame.HYBRID_SEGS, frame.WIDE_TILES = keep_h
        tby = (h + 15) // 16
        if tby >= 2:
            parts = []
            with torch.no_grad():
                for rank in range(2):
                    part, _, _ = render_rgb_stripe(md, cam, (w, h), r.ops, DEV, rank, 2,
                                                   tile_rows=stripe_rows(tby, 2, rank), collective=False)
                    parts.append(part)
            if not torch.equal(torch.cat(parts, 0), base[0]):
                msgs.append("stripes do not tile the frame")
        finite = all(torch.isfinite(t).all().item() for t in base)
        if not finite:
            msgs.append("non-finite output")
        print(f"case {c}: n={n} {w}x{h} sh={sh} mult={mult}: {'ok' if not msgs else msgs}", flush=True)
        bad += bool(msgs)
    frame.SPLIT_BLOCKS_BELOW = keep_split
    print("FAILED" if bad else "all consistent")
    return bad


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
