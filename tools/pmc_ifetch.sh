#!/bin/bash
# Instruction-fetch counters of the compositing kernels on the frame path (round 6): is the I-cache what the waves wait for?
# usage: tools/pmc_ifetch.sh <outfile> [extra args of tools/variant_check.py after the ref path]      (run on the GPU box)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$(realpath -m ${1:-$REPO/gpurun_out/pmc_ifetch.txt}); mkdir -p $(dirname $OUT); shift
cd /tmp && export TMPDIR=/tmp
: > $OUT
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQ_LEVEL_WAVES SQ_ACCUM_PREV"; do
  rm -rf /tmp/pmc_x
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_x -o x -- python $REPO/tools/variant_check.py /tmp/pmc_ref.pt "$@" > /dev/null 2>&1
  python - >> $OUT <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmc_x/**/*counter_collection.csv", recursive=True)
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "raster_" not in k and "reduce_partials" not in k and "colors_pack" not in k: continue
        k = k.replace("(anonymous namespace)::", "").split("(")[0][-60:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in sorted(acc.items()):
        print(k, len(n[k]), {c: f"{x / len(n[k]):.4g}" for c, x in v.items()})
else:
    print("no counter file")
PY
done
cat $OUT
