#!/bin/bash
# A frame of an eighth of the headline frame's pixels (the per-GPU share at 8 GPUs): frames in flight and
# resident blocks per compute unit of the two traversing kernels.
export RT_AMD_EXPERIMENTS=1
run() { # label, environment settings..., then -- bench arguments
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 100 python bench.py --steps 40 --warmup 5 --no-cpu --width 680 --height 382 "$@" 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step'], d['value'])" || exit 1
}
for n in 1 2 3 4 6; do run "in-flight $n" X=1 -- --in-flight $n || exit 1; done
for p in 1 2 3; do for s in 4 5 6; do run "in-flight 3 path/CU $p shadow/CU $s" RT_AMD_PATH_BLOCKS_PER_CU=$p RT_AMD_SHADOW_BLOCKS_PER_CU=$s -- --in-flight 3 || exit 1; done; done
for n in 4 6; do run "in-flight $n path/CU 1 shadow/CU 5" RT_AMD_PATH_BLOCKS_PER_CU=1 RT_AMD_SHADOW_BLOCKS_PER_CU=5 -- --in-flight $n || exit 1; done
