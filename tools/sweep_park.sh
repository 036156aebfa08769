#!/bin/bash
export RT_AMD_EXPERIMENTS=1 # the library reads its scheduling knobs from the environment only with this set
# Sweep of the traversal loop's vote scale / bias on the bench workload (GPU box).
source tools/ab.sh
for a in 8 12 16 24 32; do for b in 16 24 32 40 48; do
  run "scale=$a bias=$b" RT_AMD_PARK_A=$a RT_AMD_PARK_B=$b || exit 1
done; done
