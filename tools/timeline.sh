#!/bin/bash
# Kernel timeline (start, duration, gap before) of the last frame of bench.py's timed loop under rocprofv3
# --kernel-trace.  usage (GPU box): tools/timeline.sh <first-kernel-substring> <bench.py arguments...>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
FIRST=$1; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $REPO/bench.py "$@" --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-pmc --no-bandwidth --no-rgbd-figure > /dev/null 2>&1
FIRST="$FIRST" python - <<'PY'
import csv, glob, os
kt = glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(kt[0])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if os.environ["FIRST"] in r["Kernel_Name"]]
rows = rows[idx[-2]:idx[-1]]
t0 = int(rows[0]["Start_Timestamp"]); prev = t0; busy = 0
print("kernel | start us | duration us | gap before us")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:60]} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {(s - prev) / 1e3:.1f}")
    busy += e - s; prev = e
print(f"frame span {(prev - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, {len(rows)} launches")
PY
