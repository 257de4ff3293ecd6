#!/usr/bin/env python
"""One emulated rank step per rank of G (bench.py --emulate-ranks G --emulate-rank r: the rank's own stages for real,
the other ranks' records replayed - NOT a multi-GPU measurement) -> a table of ms / step over ALL ranks, min / max /
max over mean, for equal-row stripes and for work-balanced ones.  Developer tool, GPU box.
usage: python tools/rank_table.py [--ranks 8] [--config 3|5] [--top-third 0.7] [--steps 50] [--modes equal,balanced]"""
import argparse
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--top-third", type=float, default=0.0)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--modes", default="equal,balanced")
ap.add_argument("--shard-mode", default="gaussians")
args = ap.parse_args()
for mode in args.modes.split(","):
    rows = []
    for r in range(args.ranks):
        cmd = [sys.executable, str(ROOT / "bench.py"), "--config", str(args.config), "--emulate-ranks", str(args.ranks),
               "--emulate-rank", str(r), "--shard-mode", args.shard_mode, "--steps", str(args.steps), "--warmup", "10",
               "--no-cpu-baseline", "--no-pmc", "--no-bandwidth", "--no-rgbd-figure", "--profile-steps", "3"]
        if args.top_third:
            cmd += ["--top-third", str(args.top_third)]
        if mode == "balanced":
            cmd += ["--balance-stripes"]
        out = subprocess.run(cmd, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(f"rank {r}: bench failed\n{out.stderr[-800:]}")
            continue
        d = json.loads(line[-1])
        kern = sum(d["entries_ms"].values())
        rows.append((r, d["ms_per_step"], kern, d["config"]["intersections_listed"], d["config"].get("stripes")))
        print(f"{mode:9s} rank {r}: {d['ms_per_step']:.3f} ms / step, kernels {kern:.3f} ms, listed pairs {rows[-1][3]}", flush=True)
    if rows:
        ms = [x[1] for x in rows]
        ks = [x[2] for x in rows]
        print(f"== {mode}: config {args.config}, top_third {args.top_third}: step min {min(ms):.3f} max {max(ms):.3f} "
              f"max/mean {max(ms) / (sum(ms) / len(ms)):.3f} | kernels min {min(ks):.3f} max {max(ks):.3f} "
              f"max/mean {max(ks) / (sum(ks) / len(ks)):.3f} | stripes {rows[0][4]}", flush=True)
