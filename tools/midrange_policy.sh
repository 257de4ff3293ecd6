#!/bin/bash
# Launch policies for launches between a rank's stripe and a full frame (1 536 < tiles < 4 096: one rank of 2 or 4 on
# config 3), emulated on one GPU (developer tool, GPU box).  usage: bash tools/midrange_policy.sh
run() {   # G rank env...
  g=$1; r=$2; shift 2
  echo -n "G=$g [$*] "
  env "$@" python bench.py --emulate-ranks $g --emulate-rank $r --shard-mode replicated --steps 30 --no-cpu-baseline --no-pmc --no-bandwidth --no-rgbd-figure 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); e=d['entries_ms']; print(round(d['ms_per_step'],3), 'fwd', round(e['ts_raster_fwd'],3), 'bwd', round(e['ts_raster_bwd'],3), 'reduce', round(e.get('ts_reduce_partials',0),3))"
}
for g in "2 1" "4 2" "3 1"; do set -- $g
  run $1 $2 TS_NOP=1
  run $1 $2 TS_SPLIT_BLOCKS_BELOW=4095
  run $1 $2 TS_SPLIT_BLOCKS_BELOW=4095 TS_LIST_SEGMENTS=8
  run $1 $2 TS_HYBRID_FROM=1537 TS_HYBRID_WHOLE16=4 TS_HYBRID_COOP16=12
  run $1 $2 TS_HYBRID_FROM=1537 TS_HYBRID_WHOLE16=2 TS_HYBRID_COOP16=14
  run $1 $2 TS_HYBRID_FROM=1537 TS_HYBRID_WHOLE16=6 TS_HYBRID_COOP16=10
  run $1 $2 TS_HYBRID_FROM=1537 TS_HYBRID_WHOLE16=6 TS_HYBRID_COOP16=6
done
