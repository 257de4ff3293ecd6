#!/bin/bash
# rocprofv3 PMC pass over tools/time_loss.py (the two SSIM kernels): wave / busy cycles, instruction mix, waits.
# usage: tools/pmc_loss.sh <out.txt>      (GPU box; counters only, no tracing)
OUT=${1:-gpurun_out/pmc_loss.txt}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
: > $REPO/$OUT
for CNT in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pl
  rocprofv3 --pmc $CNT --output-format csv -d /tmp/pl -o pl -- python $REPO/tools/time_loss.py > /dev/null 2>&1
  F=$(find /tmp/pl -name "*counter_collection.csv" | head -1)
  python - "$F" >> $REPO/$OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "ssim" not in k: continue
    k = "ssim_fwd" if "ssim_fwd" in k else "ssim_bwd"
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
for k, v in acc.items():
    n = len(calls[k])
    print(k, "dispatches", n, " ".join(f"{c}={x / n:.4g}" for c, x in sorted(v.items())))
PY
done
cat $REPO/$OUT
