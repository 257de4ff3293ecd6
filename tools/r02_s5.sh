#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out/r02_s5; mkdir -p $OUT
python -m tinysplat_amd._build > $OUT/build.log 2>&1 || tail -5 $OUT/build.log
echo "== time_raster v2c"; timeout 120 python tools/time_raster.py 2>&1 | tail -1
echo "== quick parity"; timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
echo "== pmc"; timeout 600 bash tools/pmc_raster.sh $OUT/pmc_raster.txt
