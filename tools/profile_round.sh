#!/bin/bash
# Full evidence run for one round tag (on the GPU box, via gpurun):
#   1. un-profiled default bench (with cpu_baseline)            -> gpurun_out/<tag>/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same workload    -> <tag>_kernel_stats.csv
#   3. PMC passes, one counter group per run, no tracing        -> <tag>_pmc_*.csv, hbm_traffic.json
# usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
python bench.py 2> "$OUT/bench.err" | tail -1 > "$OUT/bench.json"
bash tools/profile_stats.sh "$TAG" > /dev/null 2>&1
cp gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv "$OUT/" 2>/dev/null
grep -o '^{.*' gpurun_out/prof_$TAG/bench.log > "$OUT/bench_under_rocprof.json"
bash tools/profile_pmc.sh ${TAG}_fetch "FETCH_SIZE" > /dev/null 2>&1
bash tools/profile_pmc.sh ${TAG}_write "WRITE_SIZE" > /dev/null 2>&1
bash tools/profile_pmc.sh ${TAG}_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" > /dev/null 2>&1
for g in fetch write sq; do cp gpurun_out/pmc_${TAG}_$g/${TAG}_${g}_pmc_summary.csv "$OUT/" 2>/dev/null; done
python - "$OUT" "$TAG" <<'PY'
import csv, json, sys
out, tag = sys.argv[1], sys.argv[2]
def load(g, col):
    d = {}
    try:
        for r in csv.DictReader(open(f"{out}/{tag}_{g}_pmc_summary.csv")):
            d[r["kernel"]] = float(r[col])
    except Exception as e:
        print("missing", g, e)
    return d
f, w = load("fetch", "FETCH_SIZE_per_dispatch"), load("write", "WRITE_SIZE_per_dispatch")
names = {"raster_bwd_kernel": "ts_raster_bwd", "raster_fwd_kernel": "ts_raster_fwd",
         "sort_tiles_small_kernel": "ts_sort_tiles", "bin_scatter_coarse_kernel": "ts_bin_scatter",
         "sh_fwd_kernel": "ts_sh_fwd", "sh_bwd_kernel": "ts_sh_bwd", "sh_colors_fwd_kernel": "ts_colors_pack_fwd", "sh_colors_fwd_sparse_kernel": "ts_colors_pack_fwd", "sh_colors_bwd_kernel": "ts_sh_colors_bwd", "reduce_partials_kernel": "ts_reduce_partials",
         "project_fwd_kernel": "ts_project_fwd", "project_bwd_kernel": "ts_project_bwd",
         "pack_splats_kernel": "ts_pack_splats", "bin_count_kernel": "ts_bin_count"}
res, raw = {}, {}
for k in f:
    for pat, entry in names.items():
        if pat in k:
            # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE tallies 128-B requests at 64 B
            # (MI355X_MICROARCH.md, HBM section) -> x2 on the read side
            res[entry] = (2.0 * f[k] + w.get(k, 0.0)) * 1024.0
            raw[entry] = {"FETCH_SIZE_KiB": f[k], "WRITE_SIZE_KiB": w.get(k, 0.0)}
json.dump(res, open(f"{out}/hbm_traffic.json", "w"), indent=1)
json.dump(raw, open(f"{out}/hbm_traffic_raw.json", "w"), indent=1)
print(json.dumps(res))
PY
cat "$OUT/bench.json" | cut -c1-1500
