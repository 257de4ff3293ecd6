#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out/r02_s6; mkdir -p $OUT
python -m tinysplat_amd._build > $OUT/build.log 2>&1 || tail -5 $OUT/build.log
echo "== time_raster v3"; timeout 120 python tools/time_raster.py 2>&1 | tail -1
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_viewer.py -q -m gpu 2>&1 | tail -70 | cut -c1-200
cp gpurun_out/parity_report.txt $OUT/parity_report.txt
echo "== pmc"; timeout 600 bash tools/pmc_raster.sh $OUT/pmc_raster.txt
echo "== bench"; timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3500 $OUT/bench.json; grep "bench " $OUT/bench.err | tail
