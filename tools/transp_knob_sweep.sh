#!/bin/bash
# Configuration 5's features (transparency, depth of field, area light) at 960x540: the shadow kernel's
# scheduling knobs, which were tuned on the headline frame's any-hit rays.
export RT_AMD_EXPERIMENTS=1
run() {
  label=$1; shift
  env "$@" timeout -k 10 150 python bench.py --workload dragon4k --width 960 --height 540 --steps 2 --warmup 1 --no-cpu --in-flight 1 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" || exit 1
}
run "defaults (min active 32, vote 16/32)" X=1 || exit 1
for m in 16 24 40 48; do run "min active $m" RT_AMD_MIN_ACTIVE_SHADOW=$m || exit 1; done
for a in 8 24 32; do run "vote scale $a" RT_AMD_PARK_A_SHADOW=$a || exit 1; done
for b in 24 40 48; do run "vote bias $b" RT_AMD_PARK_B_SHADOW=$b || exit 1; done
for m in 16 24 40 48; do run "path min active $m" RT_AMD_MIN_ACTIVE=$m RT_AMD_MIN_ACTIVE_SHADOW=32 || exit 1; done
