import sys, json
# usage: python tools/print_bench.py bench.json   (or the bench line on stdin)
for line in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    if line.startswith("{"):
        d = json.loads(line)
        print(d["config"]["workload"][:60], "| ms", round(d["ms_per_step"], 3), "| G*px/s %.3e" % d["value"], "| I", d["config"]["intersections"],
              "| max/tile", d["config"]["max_per_tile"], "| B_alg GB %.2f" % (d["frame_roofline"]["alg_bytes"] / 1e9),
              "| frame frac %.3f" % d["frame_roofline"]["frac"], "| dom", d["roofline"]["kernel"], "%.3f ms" % d["roofline"]["kernel_ms"])
        print("   entries:", {k[3:]: round(v, 3) for k, v in d["entries_ms"].items()})
