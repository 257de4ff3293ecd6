#!/usr/bin/env python
"""Per-wave timeline of the compositing kernels on the frame path (developer tool, GPU box; needs a library
built with TS_EXTRA_HIPCC_FLAGS=-DTS_TIMELINE=1 and TS_ALLOW_VARIANT_LIB=1).  Every wave records its start and
end on the 100 MHz constant clock, its hardware slot and its list length; this prints how well the launch
packs the chip: resident waves per SIMD over time, duration against list length, the tail.
usage: python tools/raster_timeline.py [n] [width] [height] [depth 0/1]"""
import ctypes
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
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


def run():
    o, (y0, y1), _ = render_stripe(model, cam, (w, h), dev, 0, 1, with_depth=depth)
    loss = (o[:, :, :3] * w_rgb).sum() + ((o[:, :, 3] * w_d).sum() if depth else 0.0)
    loss.backward()
    torch.cuda.synchronize()


for _ in range(3):
    run()
tiles = int(frame.last_binning[0].num_tiles)
saved = {}
for which, name in ((0, "raster_fwd"), (1, "raster_bwd")):
    ROW = 12
    rows_ = 1 << 17           # a hybrid launch has more work items than tiles (backward: list segments; forward: cooperative waves)
    buf = (ctypes.c_ulonglong * (ROW * rows_))()
    rc = lib.ts_debug_timeline(buf, which, rows_)
    assert rc == 0, rc
    a = np.frombuffer(buf, dtype=np.uint64).reshape(rows_, ROW)
    is_coop = (np.arange(rows_) >= tiles)[a[:, 1] > 0] if which == 0 else None
    a = a[a[:, 1] > 0]
    seg = a[:, 4:8].astype(np.float64)
    tot = a[:, 8].astype(np.float64)
    t0 = a[:, 0].astype(np.float64) * 1e-2          # us
    t1 = a[:, 1].astype(np.float64) * 1e-2
    hw = a[:, 2]
    ln = a[:, 3].astype(np.float64)
    start = t0.min()
    t0 -= start
    t1 -= start
    span = t1.max()
    dur = t1 - t0
    simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(np.int64)
    cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64)
    sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64)
    se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64)
    xcc = ((hw >> np.uint64(32)) & np.uint64(15)).astype(np.int64)
    slot = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    nsimd = len(np.unique(slot))
    print(f"== {name}: {len(a)} waves, span {span:.1f} us, distinct SIMDs seen {nsimd}")
    print(f"   wave duration us: mean {dur.mean():.1f}  p5 {np.percentile(dur, 5):.1f}  p50 {np.percentile(dur, 50):.1f}"
          f"  p95 {np.percentile(dur, 95):.1f}  max {dur.max():.1f}")
    print(f"   sum of durations / (span x SIMDs) = {dur.sum() / (span * nsimd):.2f} resident waves per SIMD on average")
    per = dur / np.maximum(ln, 1.0)
    print(f"   list length: mean {ln.mean():.0f} max {ln.max():.0f}; us per listed entry: mean {per.mean():.3f} "
          f"p5 {np.percentile(per, 5):.3f} p95 {np.percentile(per, 95):.3f}; corr(duration, length) = {np.corrcoef(dur, ln)[0, 1]:.2f}")
    names = ("chunk prologue (stage, cull, compact)", "record read (LDS -> registers)", "block bodies",
             "row flush" if which == 1 else "per-tile sort inside the launch")
    print(f"   shader clock over a wave's life: {np.median(tot / np.maximum(dur, 1e-9)) * 1e-3:.2f} GHz; cycles per wave {tot.mean():.0f}")
    for i, nm in enumerate(names):
        print(f"   segment {i} {nm:40s} {seg[:, i].sum() / tot.sum() * 100:5.1f} % of the wave cycles"
              f"   ({seg[:, i].sum() / max(1.0, ln.sum()):.0f} cycles per listed entry)")
    print(f"   outside the segments {100 - seg.sum() / tot.sum() * 100:5.1f} %")
    work = a[:, 9:12].astype(np.float64)
    X = np.column_stack([work, ln, np.ones(len(ln))])
    coef, *_ = np.linalg.lstsq(X, tot, rcond=None)
    resid = tot - X @ coef
    print(f"   work per wave: staged {work[:, 0].mean():.0f}, bodies {work[:, 1].mean():.0f}, flushes {work[:, 2].mean():.0f};"
          f" cycles ~ {coef[0]:.0f} staged + {coef[1]:.0f} bodies + {coef[2]:.0f} flushes + {coef[3]:.0f} listed + {coef[4]:.0f};"
          f" residual sigma {resid.std() / tot.mean() * 100:.1f} % of the mean (duration sigma {tot.std() / tot.mean() * 100:.1f} %)")
    saved[name] = (tot, work)
    # resident waves per SIMD over time
    edges = np.linspace(0.0, span, 21)
    line = []
    for lo, hi in zip(edges[:-1], edges[1:]):
        ov = np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0.0, None).sum() / ((hi - lo) * nsimd)
        line.append(f"{ov:.1f}")
    print("   resident waves per SIMD in 20 time slices: " + " ".join(line))
    # duration of a wave against the number of waves started at the same time on its SIMD (first round only)
    first = t0 < 0.05 * span
    print(f"   first-round waves: {first.sum()}, mean duration {dur[first].mean():.1f} us; later waves: mean {dur[~first].mean():.1f} us")
    if which == 0 and is_coop is not None and is_coop.any():
        dc, dw = dur[is_coop], dur[~is_coop]
        print(f"   whole-tile waves: {len(dw)}, mean {dw.mean():.1f} us, sum {dw.sum() / 1e3:.1f} ms | cooperative waves: {len(dc)} "
              f"({len(dc) // 4} tiles), mean {dc.mean():.1f} us p95 {np.percentile(dc, 95):.1f}, sum {dc.sum() / 1e3:.1f} ms; "
              f"first cooperative wave starts at {t0[is_coop].min():.1f} us, last whole-tile wave ends at {t1[~is_coop].max():.1f} us")
        sc, tc = seg[is_coop], tot[is_coop]
        print("   cooperative waves, share of their cycles: staging %.1f %%, barrier waits %.1f %%, bodies + record reads %.1f %%, "
              "shared sort %.1f %%, rest %.1f %%" % tuple(list(sc.sum(axis=0) / tc.sum() * 100) + [100 - sc.sum() / tc.sum() * 100]))
    waves_per_simd = np.bincount(np.unique(slot, return_inverse=True)[1])
    print(f"   waves per SIMD: min {waves_per_simd.min()} mean {waves_per_simd.mean():.2f} max {waves_per_simd.max()}")
    busy_end = np.zeros(nsimd)
    inv = np.unique(slot, return_inverse=True)[1]
    np.maximum.at(busy_end, inv, t1)
    print(f"   last wave end per SIMD, us: p5 {np.percentile(busy_end, 5):.1f} p50 {np.percentile(busy_end, 50):.1f} "
          f"p95 {np.percentile(busy_end, 95):.1f} max {busy_end.max():.1f}")

if len(saved) == 2 and len(saved["raster_fwd"][0]) == len(saved["raster_bwd"][0]):
    f, b = saved["raster_fwd"][0], saved["raster_bwd"][0]
    print(f"corr(fwd cycles, bwd cycles) per tile = {np.corrcoef(f, b)[0, 1]:.3f}")
    X = np.column_stack([saved["raster_fwd"][1][:, :2], np.ones(len(f))])
    coef, *_ = np.linalg.lstsq(X, b, rcond=None)
    r = b - X @ coef
    print(f"bwd cycles predicted from the FORWARD pass's staged / body counts: residual sigma {r.std() / b.mean() * 100:.1f} % of the mean")
