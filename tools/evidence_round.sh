#!/bin/bash
# One-call evidence run for a round tag (GPU box, via gpurun): full -m gpu suite, smoke, the profiled
# default bench (tools/profile_round.sh) and the bench line of every other configuration.
# usage: tools/evidence_round.sh <tag>
TAG=${1:-r02}
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || tail -5 $OUT/build.log
echo "== gpu tests"; timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tee $OUT/pytest_gpu.log | tail -3 | cut -c1-200
cp gpurun_out/parity_report.txt $OUT/parity_report.txt
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== profile round"; timeout 1500 bash tools/profile_round.sh $TAG 2>&1 | tail -1 | cut -c1-300
echo "== pmc raster"; timeout 600 bash tools/pmc_raster.sh $OUT/${TAG}_pmc_raster.txt > /dev/null 2>&1; cat $OUT/${TAG}_pmc_raster.txt | cut -c1-250
echo "== other configurations"
for a in "--config 2" "--config 5 --steps 20 --warmup 5" "--config 5 --steps 20 --warmup 5 --spatial-sort" "--spatial-sort" "--depth" "--forward-only" "--train-step" \
         "--emulate-ranks 2 --emulate-rank 1" "--emulate-ranks 4 --emulate-rank 2" "--emulate-ranks 8 --emulate-rank 4" "--emulate-ranks 8 --emulate-rank 0" \
         "--emulate-ranks 2 --emulate-rank 1 --shard-mode replicated" "--emulate-ranks 4 --emulate-rank 2 --shard-mode replicated" "--emulate-ranks 8 --emulate-rank 4 --shard-mode replicated" \
         "--emulate-ranks 8 --emulate-rank 4 --shard-mode replicated --force-dist" \
         "--config 5 --steps 20 --warmup 5 --emulate-ranks 8 --emulate-rank 4" "--config 5 --steps 20 --warmup 5 --emulate-ranks 8 --emulate-rank 4 --shard-mode replicated"; do
  timeout 300 python bench.py $a --no-cpu-baseline --no-pmc 2>/dev/null | tee "$OUT/bench_$(echo $a | tr -d ' -').json" | python tools/print_bench.py | head -1 | cut -c1-230
done
echo "== hybrid compositing launches: settings around the default (tools/hybrid_sweep.py)"
timeout 300 python tools/hybrid_sweep.py --settings "1:0 8:13 8:12 8:14 4:13 1:0" 2>&1 | grep "^S=" | tee $OUT/${TAG}_hybrid_sweep.txt
echo "== cooperative tiles of the forward launch: shares around the default (tools/coop_sweep.py), small launches (tools/coop_split_check.py)"
timeout 300 python tools/coop_sweep.py --settings "0 2 3 4 6 0" 2>&1 | grep "^C16" | tee $OUT/${TAG}_coop_sweep.txt
( timeout 300 python tools/coop_split_check.py 2>&1 | grep "^coop_split"; timeout 300 python tools/coop_split_check.py --depth 2>&1 | grep "^coop_split"; \
  timeout 300 python tools/coop_split_check.py --n 200000 --width 512 --height 512 --ranks 1 --rank 0 2>&1 | grep "^coop_split" ) | tee $OUT/${TAG}_coop_split.txt
if [ -f build/tl/libtinysplat_hip.so ]; then
  echo "== per-wave timeline of both compositing launches (-DTS_TIMELINE=1 build under build/tl)"
  TS_LIB_PATH=build/tl/libtinysplat_hip.so TS_ALLOW_VARIANT_LIB=1 timeout 300 python tools/raster_timeline.py 2>&1 | grep -v amdgpu | tee $OUT/${TAG}_wave_timeline.txt | grep "resident waves per SIMD on average"
fi
echo "== launch policies between a rank's stripe and a full frame (tools/midrange_policy.sh)"
timeout 900 bash tools/midrange_policy.sh 2>&1 | grep "^G=" | tee $OUT/${TAG}_midrange_policy.txt | head -3
echo "== one emulated rank step for EVERY rank of 8 (tools/rank_table.py), config 3 and config 5"
timeout 600 python tools/rank_table.py --config 3 --modes equal 2>&1 | grep -v amdgpu | tee $OUT/${TAG}_rank_table_config3.txt | tail -1
timeout 900 python tools/rank_table.py --config 5 --modes equal --steps 20 2>&1 | grep -v amdgpu | tee $OUT/${TAG}_rank_table_config5.txt | tail -1
timeout 600 python tools/rank_table.py --config 3 --modes equal --shard-mode replicated 2>&1 | grep -v amdgpu | tee $OUT/${TAG}_rank_table_config3_replicated.txt | tail -1
echo "== host side of the emulated rank step"; timeout 300 python tools/host_profile_rank.py 200 2>&1 | grep -v amdgpu | head -19 | tee $OUT/${TAG}_rank_host_marks.txt | head -3
echo "== bench --gpus 2 (both ranks on this GPU, gloo: functional)"; timeout 300 python bench.py --gpus 2 --single-device --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | cut -c1-300
echo "== marker + kernel trace of three frames (roctx ranges of the executor calls)"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/mk && timeout 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d /tmp/mk -o mk -- python $OLDPWD/bench.py --steps 3 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-pmc --no-bandwidth --no-rgbd-figure > /dev/null 2>&1; \
  for f in $(find /tmp/mk -name "*marker*stats*.csv" -o -name "*marker_api_stats.csv" | head -2); do cp $f $OUT/${TAG}_marker_stats.csv; done; \
  python - $OUT/${TAG}_timeline.txt <<'PY'
import csv, glob, sys
out = open(sys.argv[1], "w")
kt = glob.glob("/tmp/mk/**/*kernel_trace.csv", recursive=True)
mt = glob.glob("/tmp/mk/**/*marker_api_trace.csv", recursive=True)
if kt:
    rows = sorted(csv.DictReader(open(kt[0])), key=lambda r: int(r["Start_Timestamp"]))
    # the last frame of the TIMED loop (native executor): the frame after it is bench.py's per-entry timing frame,
    # whose launches are issued one by one from Python between event records
    idx = [i for i, r in enumerate(rows) if "project_fwd_kernel" in r["Kernel_Name"]]
    if len(idx) >= 2:
        rows = rows[idx[-2]:idx[-1]]
        t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
        out.write("kernel | start us | duration us | gap before us\n")
        for r in rows:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            out.write(f"{r['Kernel_Name'][:70]} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {(s - prev_end) / 1e3:.1f}\n")
            prev_end = e
if mt:
    out.write("\nroctx ranges (host side) of the trace:\n")
    for r in list(csv.DictReader(open(mt[0])))[-12:]:
        out.write(" | ".join(f"{k}={v}" for k, v in r.items() if k in ("Function", "Start_Timestamp", "End_Timestamp")) + "\n")
out.close()
print(open(sys.argv[1]).read()[:3000])
PY
)
echo "== host time"; python tools/host_time.py 100000 2>&1 | grep "host enqueue"; python tools/host_time.py 1000000 2>&1 | grep "host enqueue"
