#!/bin/bash
export RT_AMD_EXPERIMENTS=1 # the library reads its scheduling knobs from the environment only with this set
# A/B: bench.py (one frame at a time and three in flight) over the library builds given as arguments
for lib in "$@"; do
  for k in 1 3; do
    RT_AMD_LIB=$PWD/$lib timeout -k 10 100 python bench.py --steps 30 --warmup 6 --no-cpu --in-flight $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d[\"roofline\"]; print(\"$lib, in flight $k:\", d[\"ms_per_step\"], d[\"value\"], r[\"frame_device_ms_in_timed_region\"], r[\"inner_nodes_per_ray\"], r[\"tri_tests_per_ray\"])" || exit 1
  done
done
