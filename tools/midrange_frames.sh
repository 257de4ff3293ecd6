#!/bin/bash
# Single-GPU frames whose tile count falls between a rank's stripe and a full 1080p frame (1280x720: 3 600 tiles): the
# launch policy for 1 537 ... 4 095 tiles against its alternatives (developer tool, GPU box).
# usage: bash tools/midrange_frames.sh
for cfg in "1000000 1280 720" "300000 1280 720" "100000 1280 720" "300000 1600 900"; do set -- $cfg
 for env in "TS_NOP=1" "TS_HYBRID_MID_FROM=100000" "TS_HYBRID_MID_COOP16=0" "TS_HYBRID_MID_COOP16=3" "TS_HYBRID_MID_WHOLE16=10"; do
  echo -n "n=$1 $2x$3 [$env] "
  env $env python bench.py --n $1 --width $2 --height $3 --steps 30 --no-cpu-baseline --no-pmc --no-bandwidth --no-rgbd-figure 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); e=d['entries_ms']; print(round(d['ms_per_step'],3), 'fwd', round(e['ts_raster_fwd'],3), 'bwd', round(e['ts_raster_bwd'],3), d['config']['tile_lists'][:12])"
 done
done
