#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r02_s3; mkdir -p $OUT
python -m tinysplat_amd._build > $OUT/build.log 2>&1 || tail -5 $OUT/build.log
echo "== time_raster v2b"; python tools/time_raster.py 2>&1 | tail -1
echo "== quick parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3
echo "== microbench"; hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_bench tools/micro/lds_bench.hip 2>/dev/null && /tmp/lds_bench
bash tools/ablate.sh "-DTS_ABLATE=3" "-DTS_ABLATE=5" "-DTS_ABLATE=6" "-DTS_RASTER_WAVES=2" 2>&1 | grep -v "^==  default" | tail -12
echo "== pmc bwd"
cd /tmp && export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  rm -rf /tmp/pmc_x; timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_x -o x -- python $GRAFT_REPO_ROOT/tools/time_raster.py > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmc_x/**/*counter_collection.csv", recursive=True)
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "raster_" not in k: continue
        k = k.split("(")[0][-40:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        print(k, {c: f"{x / len(n[k]):.4g}" for c, x in v.items()})
PY
done
