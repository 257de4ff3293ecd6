#!/bin/bash
# scratch driver for one gpurun call (developer tool)
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r6/suite_final.txt
cat gpurun_out/r6/suite_final.txt
cp gpurun_out/parity_report.txt gpurun_out/r6/parity_report_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --train-step --no-cpu-baseline --no-pmc 2>/dev/null > gpurun_out/r6/bench_trainstep_final.json
python tools/print_bench.py < gpurun_out/r6/bench_trainstep_final.json | head -1 | cut -c1-120
python bench.py --steps 20 --warmup 5 2>/dev/null > gpurun_out/r6/bench_final.json
python tools/print_bench.py < gpurun_out/r6/bench_final.json | head -1 | cut -c1-160
