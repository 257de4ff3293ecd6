#!/bin/bash
# Host share of the single-GPU step: ms_per_step beside the sum of the per-entry kernel times, for
# configs 3 and 2, with and without capacity-sized lists (TS_CAPACITY_ALLOC), three runs each on THIS box.
# usage: tools/host_gap.sh [out file]
out=${1:-gpurun_out/host_gap.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
lscpu | grep -E "Model name|^CPU\(s\)" >> "$out"
for cfg in 3 2; do
  for cap in 0 1; do
    for rep in 1 2 3; do
      TS_CAPACITY_ALLOC=$cap python bench.py --config $cfg --steps 50 --warmup 10 --no-pmc --no-cpu-baseline --no-bandwidth --no-rgbd-figure 2>/dev/null \
        | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d.get('entries_ms') or d.get('roofline',{}).get('entries_ms') or {}
print('config $cfg  TS_CAPACITY_ALLOC=$cap  run $rep: ms_per_step %.4f  sum(entries) %.4f  gap %.1f us  with_loss %.4f' % (d['ms_per_step'], sum(e.values()), 1e3*(d['ms_per_step']-sum(e.values())), (d.get('with_loss_kernels') or {}).get('ms_per_step', float('nan'))))" >> "$out"
    done
  done
done
cat "$out"
