#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r02_s4; mkdir -p $OUT
python -m tinysplat_amd._build > $OUT/build.log 2>&1 || tail -5 $OUT/build.log
echo "== microbench"; hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_bench tools/micro/lds_bench.hip 2>/dev/null && timeout 60 /tmp/lds_bench
echo "== pmc"; timeout 700 bash tools/pmc_raster.sh $OUT/pmc_raster.txt
for f in "-DTS_ABLATE=3" "-DTS_ABLATE=5" "-DTS_ABLATE=6"; do
  echo "== variant $f"; TS_EXTRA_HIPCC_FLAGS="$f" python -m tinysplat_amd._build > /tmp/b.log 2>&1; TS_ALLOW_VARIANT_LIB=1 timeout 120 python tools/time_raster.py 2>&1 | tail -1 | cut -c1-120
done
python -m tinysplat_amd._build > /dev/null 2>&1
