#!/bin/bash
# SCALE run of the render path on ONE node (BASELINE configs[3] and [4]): exactly what a driver would
# launch - bench.py at 1, 2, 4, 8 ranks, one rank per GPU over RCCL/xGMI, for config 3 (1 M Gaussians, 1080p) and
# config 5 (5 M, 4K, RGB + depth) - one JSON line per run in $OUT/scale_<config>_<N>[_balanced].json.
#
#   tools/scale_run.sh                 the real thing (needs N visible GPUs; N = 1 2 4 8, skipping what the node lacks)
#   tools/scale_run.sh --dry           the same launches with --backend gloo --single-device on ONE GPU: every rank on
#                                      cuda:0, collectives through the host - a FUNCTIONAL rehearsal of the N > 1 path
#                                      (command lines, preflight, both shard designs, the fields of the line), NOT a
#                                      measurement; the scenes are scaled down so that it finishes in minutes
#
# What an N > 1 line carries (bench.py, asserted by tests/test_gpu_dist.py): `value` = whole-job G*px/s from the
# max-over-ranks wall time of K steps between barriers; `multi_gpu.shard_mode` (which design ran: 'gaussians' = every
# rank owns N/G Gaussians and one tile-row stripe, records / gradient rows by all_to_all; N > 1 defaults to timing
# BOTH designs, the north star's 'replicated' one as `multi_gpu.replicated_mode`), `preflight` / `shard_mode_fallback`,
# `rank_ms_min/max`, `collective_ms_per_step`, `collective_calls_per_step`, `collective_bytes_per_step` (by collective).
#
# Environment: HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC: RCCL fails with hipIpcGetMemHandle otherwise on these hosts);
# rendezvous on 127.0.0.1; NCCL_DEBUG=WARN so that a transport problem is visible in the log; nothing else is needed -
# the frame makes 3 small collectives (Gaussian shards: ~1 MB per pair of ranks) or one 40 MB all-reduce (replicated).
set -u
cd "$(dirname "$0")/.."
OUT=${OUT:-gpurun_out/scale}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 NCCL_DEBUG=${NCCL_DEBUG:-WARN}
DRY=0; [ "${1:-}" = "--dry" ] && DRY=1
GPUS=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
STEPS=${STEPS:-20}; WARMUP=${WARMUP:-5}
port=29500
run() {   # run <tag> <ranks> <bench args...>
  local tag=$1 n=$2; shift 2
  port=$((port + 1))
  local extra=""
  if [ $DRY = 1 ]; then extra="--backend gloo --single-device --no-pmc --no-bandwidth --prewarm-ms 0"; fi
  echo "== $tag: $n rank(s): bench.py --gpus $n $* $extra" >&2
  if [ "$n" = 1 ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup $WARMUP --no-cpu-baseline $extra "$@" > "$OUT/$tag.json" 2> "$OUT/$tag.log"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus "$n" --steps $STEPS --warmup $WARMUP --no-cpu-baseline $extra "$@" > "$OUT/$tag.json" 2> "$OUT/$tag.log"
  fi
  local rc=$?
  tail -n 1 "$OUT/$tag.json" | python -c "
import json, sys
try:
    d = json.loads(sys.stdin.read())
except Exception as e:
    print('   no JSON line (rc $rc):', e); sys.exit(0)
m = d.get('multi_gpu') or {}
r = m.get('replicated_mode') or {}
print('   n_gpus %d  %.3f ms/step  %.3e G*px/s | %s: rank step %.3f .. %.3f ms, collectives %s ms in %s calls, bytes %s | replicated: %s ms' % (
    d['n_gpus'], d['ms_per_step'], d['value'], m.get('shard_mode', 'single GPU'), m.get('rank_ms_min', d['ms_per_step']),
    m.get('rank_ms_max', d['ms_per_step']), m.get('collective_ms_per_step'), m.get('collective_calls_per_step'),
    m.get('collective_bytes_per_step'), r.get('ms_per_step')))"
}
for N in 1 2 4 8; do
  if [ $DRY = 0 ] && [ "$N" -gt "$GPUS" ]; then echo "== skipping N = $N: $GPUS GPU(s) visible" >&2; continue; fi
  if [ $DRY = 1 ]; then
    [ "$N" -gt 4 ] && continue                                     # rehearsal: 1, 2, 4 ranks on one GPU
    run "scale_c3_$N" $N --gaussians 200000 --width 960 --height 544
    [ "$N" -gt 1 ] && run "scale_c3_${N}_balanced" $N --gaussians 200000 --width 960 --height 544 --balance-stripes --shard-mode gaussians
    [ "$N" = 2 ] && run "scale_c5_$N" $N --gaussians 400000 --width 1920 --height 1080 --depth
  else
    run "scale_c3_$N" $N --config 3
    [ "$N" -gt 1 ] && run "scale_c3_${N}_balanced" $N --config 3 --balance-stripes --shard-mode gaussians
    run "scale_c5_$N" $N --config 5
    [ "$N" -gt 1 ] && run "scale_c5_${N}_balanced" $N --config 5 --balance-stripes --shard-mode gaussians
  fi
done
echo "lines in $OUT/scale_*.json" >&2
