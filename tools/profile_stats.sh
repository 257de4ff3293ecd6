#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench workload (run on the GPU box via gpurun).
# usage: tools/profile_stats.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -o "$TAG" -- python "$REPO/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-bandwidth "$@" > "$OUT/bench.log" 2>&1
tail -1 "$OUT/bench.log"
find "$OUT/raw" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_kernel_stats.csv" \;
find "$OUT/raw" -name "*kernel_trace.csv" -exec sh -c 'head -400 "$1" > "$2"' _ {} "$OUT/${TAG}_kernel_trace_head.csv" \;
rm -rf "$OUT/raw"
column -s, -t < "$OUT/${TAG}_kernel_stats.csv" | cut -c1-200 | head -40
