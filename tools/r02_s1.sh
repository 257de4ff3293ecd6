#!/bin/bash
# round-2 GPU session 1: flush variants, new bench line (PMC + cpu baseline), launcher test, oracle timing
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r02_s1; mkdir -p $OUT
python -m tinysplat_amd._build > $OUT/build.log 2>&1 || tail -5 $OUT/build.log
echo "== default (swap flush)"; python tools/time_raster.py 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
bash tools/ablate.sh "-DTS_FLUSH_SWAP=0" "-DTS_ABLATE=4" 2>&1 | tail -6
echo "== bench"; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== bench --gpus 2 single-device gloo"; timeout 600 python bench.py --gpus 2 --single-device --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/bench2.json 2> $OUT/bench2.err; tail -c 1200 $OUT/bench2.json; tail -5 $OUT/bench2.err
