#!/bin/bash
# Every BASELINE configuration that the CPU oracle finishes in a minute, at its full size, default and
# frame-filling camera: oracle against the device under each traversal tree.
for w in "cornell 3.0" "teapot 3.0" "teapot 1.6" "monkey 3.0" "monkey 1.6" "cube 3.0" "dragon 3.0" "dragon 1.6"; do
  timeout -k 10 300 python tools/full_size_vs_oracle.py $w 2>&1 | grep -v "^\[rt_amd\]\|amdgpu.ids" || exit 1
done
