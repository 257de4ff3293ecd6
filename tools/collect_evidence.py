#!/usr/bin/env python
"""Copies what tools/evidence_round.sh <tag> left under gpurun_out/<tag>/ into profiles/ under <name>_* (the
committed evidence set of a round) and refreshes profiles/hbm_traffic.json and profiles/pmc_latest.json from it.
usage: python tools/collect_evidence.py <tag> [<name>]"""
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else tag
src, dst = ROOT / "gpurun_out" / tag, ROOT / "profiles"


def last_json(p):
    lines = [l for l in p.read_text().splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


for a, b in (("bench.json", "bench.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"),
             ("parity_report.txt", "parity_report.txt"), ("hbm_traffic_raw.json", "hbm_traffic_raw.json")):
    if (src / a).exists():
        shutil.copy(src / a, dst / f"{name}_{b}")
for p in src.glob(f"{tag}_*"):
    shutil.copy(p, dst / p.name.replace(f"{tag}_", f"{name}_", 1))
if (src / "hbm_traffic.json").exists():
    shutil.copy(src / "hbm_traffic.json", dst / "hbm_traffic.json")
others = {}
for p in sorted(src.glob("bench_*.json")):
    if p.name == "bench_under_rocprof.json":
        continue
    d = last_json(p)
    if d:
        others[p.stem[len("bench_"):]] = {k: d[k] for k in ("metric", "value", "ms_per_step", "config", "entries_ms",
                                                             "frame_roofline", "roofline") if k in d}
(dst / f"{name}_other_configs.json").write_text(json.dumps(others, indent=1))
bench = last_json(src / "bench.json")
if bench and isinstance(bench.get("pmc_per_dispatch"), dict) and bench["pmc_per_dispatch"]:
    # bench.py's fallback when its own PMC passes are unavailable: {kernel: {counter: value}}
    (dst / "pmc_latest.json").write_text(json.dumps(bench["pmc_per_dispatch"], indent=1))
print("bench:", bench and round(bench["ms_per_step"], 4), "ms/step; other configurations:", len(others))
