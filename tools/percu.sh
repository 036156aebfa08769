#!/bin/bash
export RT_AMD_EXPERIMENTS=1 # the library reads its scheduling knobs from the environment only with this set
# fewer persistent blocks per compute unit for the path kernel alone, three frames in flight, several frame sizes
for wh in "680 382" "960 540" "1360 764" "1920 1080"; do set -- $wh
  for b in 0 1 2 3; do
    RT_AMD_PATH_BLOCKS_PER_CU=$b timeout -k 10 100 python bench.py --steps 60 --warmup 8 --no-cpu --width $1 --height $2 --in-flight 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$wh, path blocks per CU $b:\", d[\"ms_per_step\"], d[\"value\"], d[\"roofline\"][\"frame_device_ms_in_timed_region\"])" || exit 1
  done
done
