#!/usr/bin/env python
"""Developer probe (GPU box): can an HBM-bound kernel on a SIDE stream hide behind the binning chain (count, column
scan, offsets, scatter, per-tile sort - latency- / VALU-bound kernels) on the main stream?  The colour stage's SH
evaluation (192 MB of coefficients at config 3) does not depend on the binning chain; if the two overlap, a frame
that evaluates the colours beside the chain saves most of the colour stage's ~50 us.  The probe runs the drop-in
binning op (ops.bin_gaussians) with and without a concurrent ts_bench_stream_read of 192 MB.
usage: overlap_probe.py [n] [w] [h]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from tinysplat_amd import _lib, ops
from tinysplat_amd.rasterizer import project_args
from tinysplat_amd.synthetic import make_scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
dev = torch.device("cuda:0")
model, cam = make_scene(n, 3, w, h)
model = model.to(dev)
lib = _lib.load()
with torch.no_grad():
    xys, depths, radii, conics, nth, _ = ops.project_gaussians(*project_args(model, cam, (w, h), dev))
src = torch.rand(48 * n, device=dev)              # 192 MB at 1 M: the SH coefficients' size
sink = torch.zeros(1024, device=dev)
side = torch.cuda.Stream(dev)
main = torch.cuda.current_stream(dev)


def chain():
    ops.bin_gaussians(xys, depths, radii, nth, h, w, use_cache=False)


def read(stream):
    _lib.check(lib.ts_bench_stream_read(src.data_ptr(), src.numel(), sink.data_ptr(), stream.cuda_stream), "stream_read")


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


def both():
    e = torch.cuda.Event()
    e.record(main)
    side.wait_event(e)
    read(side)
    chain()
    e2 = torch.cuda.Event()
    e2.record(side)
    main.wait_event(e2)


def serial():
    read(main)
    chain()


print(f"binning chain alone      {timed(chain):7.1f} us")
print(f"stream read alone        {timed(lambda: read(main)):7.1f} us")
print(f"read then chain (serial) {timed(serial):7.1f} us")
print(f"read on a side stream    {timed(both):7.1f} us")
