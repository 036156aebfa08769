#!/bin/bash
export RT_AMD_EXPERIMENTS=1 # the library reads its scheduling knobs from the environment only with this set
# bench.py at full size and at 1/8 of the pixels for several values of the queues' positional share
for s in 0 8 12 14 16; do
  for wh in "1920 1080" "680 382"; do set -- $wh
    for k in 1 3; do
      RT_AMD_OWN_SHARE=$s timeout -k 10 100 python bench.py --steps 40 --warmup 6 --no-cpu --width $1 --height $2 --in-flight $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"share $s, $wh, in flight $k:\", d[\"ms_per_step\"], d[\"value\"], d[\"roofline\"][\"frame_device_ms_in_timed_region\"])" || exit 1
    done
  done
done
