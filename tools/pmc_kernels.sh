#!/bin/bash
# PMC counters (one group per pass, no tracing) of every kernel whose name contains <substring>, averaged per dispatch.
# usage: tools/pmc_kernels.sh <substring> <outfile> <command ...>      (run on the GPU box)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
PAT=$1; OUT=$(realpath -m $2); mkdir -p $(dirname $OUT); shift 2
cd /tmp && export TMPDIR=/tmp
: > $OUT
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_I8"; do
  rm -rf /tmp/pmc_x
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_x -o x -- "$@" > /dev/null 2>&1
  PAT="$PAT" python - >> $OUT <<'PY'
import csv, glob, collections, os
pat = os.environ["PAT"]
f = glob.glob("/tmp/pmc_x/**/*counter_collection.csv", recursive=True)
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if pat not in k: continue
        k = k.replace("(anonymous namespace)::", "").split("(")[0][-60:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        print(k, len(n[k]), {c: f"{x / len(n[k]):.4g}" for c, x in v.items()})
else:
    print("no counter file")
PY
done
cat $OUT
