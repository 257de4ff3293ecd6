#!/bin/bash
# PMC counters of the two compositing kernels (tools/time_raster.py workload), one group per pass.
# usage: tools/pmc_raster.sh <outfile>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$REPO/gpurun_out/pmc_raster.txt}
cd /tmp && export TMPDIR=/tmp
: > $OUT
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA"; do
  rm -rf /tmp/pmc_x
  timeout 150 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_x -o x -- python $REPO/tools/time_raster.py > /dev/null 2>&1
  python - >> $OUT <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmc_x/**/*counter_collection.csv", recursive=True)
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "raster_" not in k: continue
        k = ("raster_bwd" if "raster_bwd" in k else "raster_fwd") + ("<4>" if "<4" in k else "<3>")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        print(k, {c: f"{x / len(n[k]):.4g}" for c, x in v.items()})
else:
    print("no counter file")
PY
done
cat $OUT
