#!/bin/bash
# scratch driver for one gpurun call (developer tool)
mkdir -p gpurun_out/r6
python tools/time_densify.py > gpurun_out/r6/densify_gather2.json 2> gpurun_out/r6/densify_gather2.err
python -m pytest tests/test_gpu_densify.py -q -x 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r6/suite_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > gpurun_out/r6/smoke_final.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r6/bench_final.json 2> gpurun_out/r6/bench_final.err
cat gpurun_out/r6/densify_gather2.json gpurun_out/r6/suite_final.txt gpurun_out/r6/smoke_final.txt
cut -c1-400 gpurun_out/r6/bench_final.json
