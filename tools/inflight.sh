#!/bin/bash
export RT_AMD_EXPERIMENTS=1 # the library reads its scheduling knobs from the environment only with this set
# bench.py at 1/4 and 1/8 of the frame's pixels with more frames in flight: what small shards gain from it
for wh in "960 540" "680 382"; do set -- $wh
  for k in 1 3 4 6 8; do
    timeout -k 10 100 python bench.py --steps 40 --warmup 8 --no-cpu --width $1 --height $2 --in-flight $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$wh in flight $k:\", d[\"ms_per_step\"], d[\"value\"])" || exit 1
  done
done
