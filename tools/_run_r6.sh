#!/bin/bash
# scratch driver for one gpurun call (developer tool)
mkdir -p gpurun_out/r6
python tools/vjp_probe.py 578 > gpurun_out/r6/vjp_probe_578.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r6/suite_final.txt
cat gpurun_out/r6/suite_final.txt
cp gpurun_out/parity_report.txt gpurun_out/r6/parity_report_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
