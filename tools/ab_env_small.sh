#!/bin/bash
export RT_AMD_EXPERIMENTS=1 # the library reads its scheduling knobs from the environment only with this set
# A/B at 1/8 of the frame's pixels under the environment settings given as arguments
for e in "$@"; do
  for k in 1 3; do
    env $e timeout -k 10 150 python bench.py --steps 60 --warmup 6 --no-cpu --in-flight $k --width 680 --height 382 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d[\"roofline\"]; print(\"$e, in flight $k:\", d[\"ms_per_step\"], d[\"value\"], r[\"frame_device_ms_in_timed_region\"])" || exit 1
  done
done
