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
for a in "--config 2" "--config 5 --steps 20 --warmup 5" "--config 5 --steps 20 --warmup 5 --spatial-sort" "--spatial-sort" "--depth" "--forward-only" "--train-step" "--emulate-ranks 2 --emulate-rank 1" "--emulate-ranks 4 --emulate-rank 2" "--emulate-ranks 8 --emulate-rank 4" "--emulate-ranks 8 --emulate-rank 4 --force-dist"; do
  timeout 300 python bench.py $a --no-cpu-baseline --no-pmc 2>/dev/null | tee "$OUT/bench_$(echo $a | tr -d ' -').json" | python tools/print_bench.py | head -1 | cut -c1-230
done
echo "== bench --gpus 2 (both ranks on this GPU, gloo: functional)"; timeout 300 python bench.py --gpus 2 --single-device --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | cut -c1-300
echo "== host time"; python tools/host_time.py 100000 2>&1 | grep "host enqueue"; python tools/host_time.py 1000000 2>&1 | grep "host enqueue"
