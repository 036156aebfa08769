#!/bin/bash
export RT_AMD_EXPERIMENTS=1 # the library reads its scheduling knobs from the environment only with this set
# A/B on another workload: tools/ab_env_wl.sh <workload> <env settings>...
wl=$1; shift
for e in "$@"; do
  for k in 1 3; do
    env $e timeout -k 10 150 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu --in-flight $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d[\"roofline\"]; print(\"$wl $e, in flight $k:\", d[\"ms_per_step\"], d[\"value\"], r[\"frame_device_ms_in_timed_region\"])" || exit 1
  done
done
