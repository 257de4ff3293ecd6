#!/usr/bin/env python
"""Lists the GPU kernels of ONE step (between two project_fwd launches) of a rocprofv3 --kernel-trace csv: start,
duration, gap to the previous kernel - which launches around the HIP entries does PyTorch add to a training step?

usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o NAME -- python bench.py --train-step ...
       python tools/step_kernels.py DIR        (developer tool)
"""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "project_fwd_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
prev = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("void at::native::", "at::")[:72]
    print(f"{name:72s} start {(s - t0) / 1e3:8.1f} dur {(e - s) / 1e3:7.1f} gap {(s - prev) / 1e3:6.1f} grid {r.get('Grid_Size', '')}")
    prev = e
    busy += e - s
print(f"step span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, launches {b - a}")
