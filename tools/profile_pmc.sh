#!/bin/bash
# rocprofv3 PMC passes (one counter group per run; never combined with tracing) for the bench workload.
# usage: tools/profile_pmc.sh <tag> "<counters...>" [bench args...]
set -u
TAG=$1; CNT=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --output-format csv -d "$OUT/raw" -o "$TAG" -- python "$REPO/bench.py" --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-pmc --no-bandwidth "$@" > "$OUT/bench.log" 2>&1
F=$(find "$OUT/raw" -name "*counter_collection.csv" | head -1)
python - "$F" "$OUT/${TAG}_pmc_summary.csv" <<'PY'
import csv, sys, collections
src, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(src)):
    k = r["Kernel_Name"][:80]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], k)
    if key not in seen:
        seen.add(key); calls[k] += 1
names = sorted({c for v in acc.values() for c in v})
with open(dst, "w") as f:
    w = csv.writer(f); w.writerow(["kernel", "dispatches"] + [n + "_per_dispatch" for n in names])
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
        w.writerow([k, calls[k]] + [f"{v.get(n, 0.0) / max(calls[k], 1):.6g}" for n in names])
print(open(dst).read()[:6000])
PY
rm -rf "$OUT/raw"
