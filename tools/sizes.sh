#!/bin/bash
export RT_AMD_EXPERIMENTS=1 # the library reads its scheduling knobs from the environment only with this set
# bench.py at the full frame and at 1/2, 1/4, 1/8 of its pixels: what a shard of the frame costs
# (a stand-in for the per-GPU share under tile sharding).  Extra arguments: environment settings.
for wh in "1920 1080" "1360 764" "960 540" "680 382"; do set -- $wh
  env "${EXTRA_ENV[@]}" timeout -k 10 100 python bench.py --steps 20 --warmup 3 --no-cpu --width $1 --height $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$wh\", d[\"ms_per_step\"], d[\"value\"], d[\"roofline\"][\"frame_device_ms_in_timed_region\"])" || exit 1
done
