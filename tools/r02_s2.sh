#!/bin/bash
# round-2 GPU session 2: redesigned compositing kernels - timing, parity, new tests, bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r02_s2; mkdir -p $OUT
python -m tinysplat_amd._build > $OUT/build.log 2>&1 || tail -5 $OUT/build.log
echo "== time_raster"; python tools/time_raster.py 2>&1 | tail -1
echo "== parity"; TS_PARITY_REPORT_ONLY=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -m gpu 2>&1 | tail -60 | cut -c1-220
cp gpurun_out/parity_report.txt $OUT/parity_report_1.txt
echo "== fullsize+viewer+training"; timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_viewer.py tests/test_gpu_training.py tests/test_gpu_densify.py tests/test_gpu_formats.py -q -m gpu 2>&1 | tail -15 | cut -c1-220
echo "== new tests"; TS_PARITY_REPORT_ONLY=1 timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_dist.py -q -m gpu -s 2>&1 | tail -60 | cut -c1-250
cp gpurun_out/parity_report.txt $OUT/parity_report_2.txt
echo "== bench"; timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 2500 $OUT/bench.json; grep bench $OUT/bench.err | tail -12
