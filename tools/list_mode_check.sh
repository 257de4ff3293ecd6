#!/bin/bash
# 16x16 lists (hybrid backward launch + cooperative forward tiles) against wide 32x16 lists (one wave per 16x16 tile on
# them) on frames with long lists (developer tool, GPU box).  usage: bash tools/list_mode_check.sh
for cfg in "--config 5 --steps 15 --warmup 4" "--config 5 --spatial-sort --steps 15 --warmup 4" "--n 2000000 --width 1280 --height 720 --steps 20" "--n 4000000 --width 1920 --height 1080 --steps 15"; do
  for e in TS_WIDE_TILES=2 TS_WIDE_TILES=0; do
    echo -n "$e [$cfg]: "
    env $e python bench.py $cfg --no-cpu-baseline --no-pmc --no-bandwidth --no-rgbd-figure 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); e=d['entries_ms']
print(round(d['ms_per_step'],3), {k[3:]:round(v,3) for k,v in e.items() if k[3:] in ('raster_fwd','raster_bwd','sort_tiles','bin_scatter','reduce_partials')}, 'pairs/tile', d['config']['intersections']//max(1,(d['config'].get('tiles',0) or 1)) if False else '', 'max/tile', d['config']['max_per_tile'])"
  done
done
