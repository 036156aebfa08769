#!/bin/bash
# The round's measurements of the headline workload on one GPU, in the order the bench line needs them:
# counter passes (-> gpurun_out/<tag>/pmc/pmc_counters_dragon.json, which bench.py reads from profiles/),
# kernel durations under rocprofv3 with one frame at a time and with three in flight, then bench.py itself.
#     bash tools/measure_round.sh <tag>            (GPU box, repository root)
set -u
tag=${1:-round}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
PMC_SUMMARY_NAME=${tag}_pmc.txt bash tools/pmc.sh "$out/pmc" dragon || exit 1
cp "$out/pmc/pmc_counters_dragon.json" "$root/profiles/pmc_counters_dragon.json"   # (this box's copy: bench.py below reads it)
cd /tmp && export TMPDIR=/tmp
for k in 1 3; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$k" -o s -- \
    python3 "$root/bench.py" --steps 10 --warmup 3 --no-cpu --in-flight $k > "$out/stats_$k.json" 2> "$out/stats_$k.err" || { echo "stats run $k failed"; exit 1; }
done
cd "$root"
timeout -k 10 300 python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || { echo "bench failed"; exit 1; }
timeout -k 10 120 python3 bench.py --no-cpu --in-flight 1 > "$out/bench_one_frame_at_a_time.json" 2>> "$out/bench.err" || exit 1
find "$out" -name "*kernel_stats.csv" | head
