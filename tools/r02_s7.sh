#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out/r02_s7; mkdir -p $OUT
python -m tinysplat_amd._build > $OUT/build.log 2>&1 || tail -5 $OUT/build.log
echo "== full gpu suite"; timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | cut -c1-200
cp gpurun_out/parity_report.txt $OUT/parity_report.txt
for c in 2 3; do echo "== bench config $c"; timeout 300 python bench.py --config $c --no-cpu-baseline --no-pmc 2>/dev/null | python tools/print_bench.py; done
echo "== bench config 3 depth"; timeout 300 python bench.py --depth --no-cpu-baseline --no-pmc 2>/dev/null | python tools/print_bench.py
echo "== emulate 8 stripes"; timeout 300 python bench.py --emulate-ranks 8 --emulate-rank 4 --no-cpu-baseline --no-pmc 2>/dev/null | python tools/print_bench.py
