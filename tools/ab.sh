#!/bin/bash
export RT_AMD_EXPERIMENTS=1 # the library reads its scheduling knobs from the environment only with this set
# A/B helper: bench.py over several library builds / knob settings, one JSON summary line each.
run() { # label, env...
  label=$1; shift
  out=$(env "$@" timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1)
  echo "$label $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "Mrays/s", d["ms_per_step"], "ms", d["roofline"]["frame_device_ms_in_timed_region"])')"
}
