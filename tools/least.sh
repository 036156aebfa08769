#!/bin/bash
export RT_AMD_EXPERIMENTS=1 # the library reads its scheduling knobs from the environment only with this set
# smallest batch the path kernel's waves take from the queue, at 1/8 and full frame size
for l in 8 16 32 64; do
  for wh in "680 382" "1920 1080"; do set -- $wh
    for k in 1 3; do
      RT_AMD_LEAST_PATH=$l timeout -k 10 100 python bench.py --steps 40 --warmup 6 --no-cpu --width $1 --height $2 --in-flight $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"least $l, $wh, in flight $k:\", d[\"ms_per_step\"], d[\"value\"], d[\"roofline\"][\"frame_device_ms_in_timed_region\"])" || exit 1
    done
  done
done
