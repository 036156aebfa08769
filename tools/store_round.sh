#!/bin/bash
# Copies what tools/measure_round.sh left under gpurun_out/<tag>/ (merged back from the GPU box) to where it is kept:
# the counters bench.py reads and the round's summaries under profiles/.
#     bash tools/store_round.sh <tag> [file with the GPU test log]
set -eu
t=$1
cd "$(dirname "$0")/.."
cp gpurun_out/$t/pmc/pmc_counters_dragon.json profiles/pmc_counters_dragon.json
cp gpurun_out/$t/pmc/summary.txt profiles/${t}_pmc.txt
cp gpurun_out/$t/bench.json profiles/${t}_bench.json
cp gpurun_out/$t/bench_one_frame_at_a_time.json profiles/${t}_bench_one_frame_at_a_time.json
cp gpurun_out/$t/stats_1/s_kernel_stats.csv profiles/${t}_one_frame_at_a_time_kernel_stats.csv
cp gpurun_out/$t/stats_3/s_kernel_stats.csv profiles/${t}_three_in_flight_kernel_stats.csv
[ $# -ge 2 ] && cp "$2" profiles/${t}_gpu_tests.log
python3 - "$t" <<'PY'
import json, sys
t = sys.argv[1]
b = json.load(open(f"profiles/{t}_bench.json")); r = b["roofline"]
print(t, "three in flight:", b["value"], "Mrays/s", b["ms_per_step"], "ms; roofline", r["achieved"], r["frac"], "valu", r["valu_issue"].get("frac"), "kernel ms", r.get("kernel_ms"))
b = json.load(open(f"profiles/{t}_bench_one_frame_at_a_time.json")); print(t, "one at a time:", b["value"], b["ms_per_step"])
PY
