#!/bin/bash
export RT_AMD_EXPERIMENTS=1 # the library reads its scheduling knobs from the environment only with this set
# more hardware queues per process, so that more than three frames in flight really overlap
for q in 4 8 16; do
  for wh in "680 382" "960 540"; do set -- $wh
    for k in 3 4 6 8; do
      GPU_MAX_HW_QUEUES=$q timeout -k 10 100 python bench.py --steps 60 --warmup 8 --no-cpu --width $1 --height $2 --in-flight $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"queues $q, $wh, in flight $k:\", d[\"ms_per_step\"], d[\"value\"], d[\"roofline\"][\"frame_device_ms_in_timed_region\"][\"all_kernels\"])" || exit 1
    done
  done
done
