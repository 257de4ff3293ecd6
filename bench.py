#!/usr/bin/env python
"""bench.py - Gaussians*pixels/s of the render hot path, forward + backward (BASELINE.json metric).

One "step" = one frame of the hot path over one synthetic scene already resident in HBM:
project_gaussians -> spherical_harmonics -> clamp -> rasterize_gaussians (RGB) -> clamp, then the
backward of L = (rgb * w).sum() down to the six parameter tensors (means, log-scales, quats,
opacity logits, colors_dc, colors_rest).  Default workload = BASELINE.json configs[2]:
1 M Gaussians, SH degree 3, 1920x1080.  With --depth the second (depth) rasterize pass of the
reference adapter is included.

N > 1 (launched by torch.distributed.run, one rank per GPU): the frame is sharded by tile-row
stripes, the 40 N-byte 2-D gradient buffers are summed with one RCCL all-reduce (sharding.py);
the total work is fixed, so scaling is "strong".

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak


def alg_bytes(entry: str, n: int, i: int, p: int, t: int, k: int, ch: int = 3) -> float:
    """Algorithmic (lower-bound) bytes one launch of a C-ABI entry moves; DESIGN.md section 4, which
    splits SURVEY.md 8(d) D5's per-stage figures over the entries."""
    return {
        "ts_project_fwd": 96.0 * n,                       # R 40 N + W 56 N
        "ts_project_bwd": 104.0 * n,                      # R 40+4+24 N, W 40 N  (no v_cov3d)
        "ts_sh_fwd": (24.0 + 12.0 * k) * n,
        "ts_sh_bwd": (24.0 + 12.0 * k) * n,
        "ts_sh_colors_fwd": (24.0 + 12.0 * k) * n,        # same stage, view dirs + clamp fused
        "ts_sh_colors_bwd": (24.0 + 12.0 * k) * n,
        "ts_scan_tiles": 8.0 * n,
        "ts_bin_count": 12.0 * n + 4.0 * t,
        "ts_tile_offsets": 16.0 * t,
        "ts_bin_scatter": 12.0 * n + 4.0 * i,
        "ts_sort_tiles": 8.0 * i + 8.0 * t,
        "ts_pack_splats": (36.0 + 4.0 + 4.0 + 48.0) * n,
        "ts_raster_fwd": 40.0 * i + 20.0 * p + 8.0 * t,   # D5 "raster fwd"
        "ts_raster_bwd": 24.0 * p + 76.0 * i + 36.0 * n,  # D5 "raster bwd" (incl. reduce)
        "ts_reduce_partials": 48.0 * i + 36.0 * n,
        "ts_photometric_loss": 36.0 * p,                  # read X, Y, write gX (3 channels)
        "ts_adam_step": 28.0 * (14.0 + 3.0 * k) * n,      # p, g, m, v read; p, m, v written
    }[entry]


def frame_alg_bytes(n, i, p, t, k, depth):
    if depth:
        return 772.0 * n + 276.0 * i + 88.0 * p
    return 736.0 * n + 160.0 * i + 44.0 * p + 16.0 * t


def measure_read_bandwidth(dev, gib: float = 2.0, reps: int = 10) -> float:
    """Achievable HBM read bandwidth (GB/s) of this GPU: ts_bench_stream_read over a buffer far larger
    than the 256 MiB Infinity Cache, best of `reps` (BW_read_measured)."""
    from tinysplat_amd import _lib
    lib = _lib.load()
    n = int(gib * (1 << 30)) // 4
    buf = torch.ones(n, dtype=torch.float32, device=dev)
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    best = 0.0
    for _ in range(reps + 2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(lib.ts_bench_stream_read(buf.data_ptr(), n, sink.data_ptr(), s), "ts_bench_stream_read")
        b.record()
        b.synchronize()
        best = max(best, 4.0 * n / (a.elapsed_time(b) * 1e-3) / 1e9)
    del buf
    return best


def cpu_baseline(seconds_budget: float = 20.0):
    """Times the oracle (pure-PyTorch CPU restatement, kind "port") on a bounded sample of the
    same workload: same scene generator / camera / SH degree, fwd+bwd of the RGB frame, scaled down
    to N = 100k Gaussians at 480x270 so that it finishes in ~10-30 s of CPU work."""
    from oracle import gsplat_oracle as O
    from tinysplat_amd.rasterizer import project_args, raster_args, sh_args
    from tinysplat_amd.synthetic import loss_weights, make_scene
    n, w, h, sh = 100_000, 480, 270, 3
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))      # many tiny per-tile ops: more threads only add sync cost
    torch.set_num_threads(cores)
    model, cam = make_scene(n, sh, w, h, seed=0)
    model.requires_grad_(True)
    w_rgb, _ = loss_weights(w, h)

    def step():
        for p_ in model.parameters():
            p_.grad = None
        pa = project_args(model, cam, (w, h), "cpu")
        xys, depths, radii, conics, nth, _ = O.project_gaussians(*pa)
        col = torch.clamp(O.spherical_harmonics(*sh_args(model, cam, "cpu")) + 0.5, min=0.0)
        img, _ = O.rasterize_gaussians(*raster_args(model, xys, depths, radii, conics, nth, col, (w, h)))
        (torch.clamp(img, max=1.0) * w_rgb).sum().backward()

    step()                                  # warm-up
    times = []
    t_end = time.perf_counter() + seconds_budget
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 5):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": n * w * h / med, "unit": "Gaussians*pixels/s", "cores": cores, "kind": "port",
            "sample": f"oracle (PyTorch CPU) fwd+bwd RGB frame, N={n} SH{sh} {w}x{h}, "
                      f"median of {len(times)} runs = {med * 1e3:.0f} ms/frame"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scale-mult", type=float, default=1.0)
    ap.add_argument("--depth", action="store_true", help="also run the depth rasterize pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--train-step", action="store_true",
                    help="time the whole training step of SURVEY 8(f) F1 instead: render RGB+depth, "
                         "L1 + DSSIM (+ depth L1) loss, backward, Adam (single GPU)")
    ap.add_argument("--forward-only", action="store_true",
                    help="time the no_grad forward frame only (the viewer path, SURVEY 8(f) F4)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the gradient all-reduce even with one rank")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="single-GPU estimate of the per-rank step at this many tile-row stripes: renders "
                         "only stripe --emulate-rank of that many (with --force-dist the 1-rank all-reduce "
                         "runs too); the printed value is NOT a multi-GPU measurement")
    ap.add_argument("--emulate-rank", type=int, default=0)
    ap.add_argument("--config", type=int, default=None,
                    help="BASELINE.json config shortcut: 2 = 100k/1080p, 3 = 1M/1080p (default), "
                         "5 = 5M/4K with depth")
    args = ap.parse_args()
    if args.config == 2:
        args.n = 100_000
    elif args.config == 5:
        args.n, args.width, args.height, args.depth = 5_000_000, 3840, 2160, True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)

    from tinysplat_amd import ops
    from tinysplat_amd.rasterizer import GaussianRasterizer
    from tinysplat_amd.sharding import render_rgb_stripe
    from tinysplat_amd.synthetic import loss_weights, make_scene

    n, w, h, sh = args.n, args.width, args.height, args.sh_degree
    model, cam = make_scene(n, sh, w, h, seed=0, scale_mult=args.scale_mult)
    model = model.to(dev).requires_grad_(True)
    w_rgb, w_d = loss_weights(w, h)
    w_rgb, w_d = w_rgb.to(dev), w_d.to(dev)
    adapter = GaussianRasterizer(model, None, device=dev)

    trainer = None
    if args.train_step:
        if world > 1:
            raise SystemExit("--train-step is a single-GPU mode")
        from tinysplat_amd.training import TrainStep
        trainer = TrainStep(model, dev)
        g_ = torch.Generator().manual_seed(2)
        tgt_rgb = torch.rand(h, w, 3, generator=g_).to(dev)
        tgt_depth = (2.0 + 8.0 * torch.rand(h, w, generator=g_)).to(dev)

    def step():
        if trainer is not None:
            trainer(cam, tgt_rgb, tgt_depth)
            return
        for p_ in model.parameters():
            p_.grad = None
        if args.forward_only:
            with torch.no_grad():
                adapter(cam, (w, h), sh)
            return
        if args.depth:
            if world > 1:
                raise SystemExit("--depth is a single-GPU mode")
            rgb, extras = adapter(cam, (w, h), sh)
            loss = (rgb * w_rgb).sum() + (extras["depth"] * w_d).sum()
        else:
            if args.emulate_ranks > 1 and world == 1:
                rgb, (y0, y1), _ = render_rgb_stripe(model, cam, (w, h), adapter.ops, dev,
                                                     args.emulate_rank, args.emulate_ranks,
                                                     collective=bool(args.force_dist))
            else:
                rgb, (y0, y1), _ = render_rgb_stripe(model, cam, (w, h), adapter.ops, dev, rank, world,
                                                     collective=True if args.force_dist else None)
            loss = (rgb * w_rgb[y0:y1]).sum()
        loss.backward()

    def barrier():
        if world > 1 or args.force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()

    # scene statistics + per-entry timing (after the timed region)
    ops.kernel_timer.start()
    for _ in range(max(1, args.profile_steps)):
        step()
    per_entry = ops.kernel_timer.stop()
    from tinysplat_amd import frame as _frame
    binning = _frame.last_binning[dev.index] if dev.index in _frame.last_binning else ops._bin_cache.get(dev.index)[1]
    isects = int(binning.num_intersects)
    tiles = int(binning.num_tiles)
    bins = binning.tile_bins
    max_per_tile = int((bins[:, 1] - bins[:, 0]).max().item()) if tiles else 0
    isects_listed = int(bins[:, 1].max().item()) if tiles else 0    # after tight binning (<= isects)
    if world > 1:
        tt = torch.tensor([isects], device=dev, dtype=torch.int64)
        dist.all_reduce(tt)
        isects_total = int(tt.item())
    else:
        isects_total = isects

    if rank == 0:
        p = w * h
        k = (sh + 1) ** 2
        ms = dt / args.steps * 1e3
        value = n * p / (dt / args.steps)
        # dominant kernel = the C-ABI entry with the largest mean duration on this rank
        dom = max(per_entry.items(), key=lambda kv: kv[1][1] * kv[1][0])
        dom_name, (dom_launches, dom_ms) = dom
        p_local = p if world == 1 else binning.cam.tile_rows * 16 * w
        a_bytes = alg_bytes(dom_name, n, isects, p_local, tiles, k)
        achieved = a_bytes / (dom_ms * 1e-3) / 1e9
        frame_bytes = frame_alg_bytes(n, isects_total, p, (w + 15) // 16 * ((h + 15) // 16), k, args.depth)
        traffic = None
        tf = ROOT / "profiles" / "hbm_traffic.json"
        if tf.exists():
            try:
                traffic = json.loads(tf.read_text()).get(dom_name)
            except Exception:
                traffic = None
        bw_meas = measure_read_bandwidth(dev)
        out = {
            "metric": "Gaussians*pixels/s forward only (no_grad, RGB+depth)" if args.forward_only else
                      "Gaussians*pixels/s fwd+bwd" if not args.train_step else
                      "Gaussians*pixels/s of a full training step (render RGB+depth, loss, backward, Adam)",
            "value": value, "unit": "Gaussians*pixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{n} random Gaussians, SH degree {sh}, {w}x{h}, fwd+bwd "
                                   f"({'RGB+depth' if args.depth else 'RGB'}), BASELINE configs[2]"
                                   if (n, sh, w, h) == (1_000_000, 3, 1920, 1080) else
                                   f"{n} random Gaussians, SH degree {sh}, {w}x{h}, fwd+bwd",
                       "intersections": isects_total, "intersections_listed_rank0": isects_listed,
                       "max_per_tile": max_per_tile,
                       "parallelism": f"tile-row stripes x{world}" if world > 1 else "single GPU",
                       "scale_mult": args.scale_mult,
                       **({"emulated_stripe": f"{args.emulate_rank} of {args.emulate_ranks} on ONE GPU "
                                              "(per-rank estimate, not a multi-GPU measurement)"}
                          if args.emulate_ranks > 1 and world == 1 else {})},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "alg_bytes_per_launch": a_bytes,
                         "kernel_ms": dom_ms, "launches_per_step": dom_launches / max(1, args.profile_steps),
                         "peak_read_measured": bw_meas, "frac_of_measured": achieved / bw_meas},
            "frame_roofline": {"alg_bytes": frame_bytes, "achieved": frame_bytes / (ms * 1e-3) / 1e9,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": frame_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "peak_read_measured": bw_meas,
                               "frac_of_measured": frame_bytes / (ms * 1e-3) / 1e9 / bw_meas},
            "entries_ms": {k_: round(v[1] * v[0] / max(1, args.profile_steps), 4)
                           for k_, v in sorted(per_entry.items())},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if world > 1 or args.force_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio, which is flushed at exit: push it out now
        # so that the JSON line is the last line of output
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
