#!/bin/bash
# Headline frame, one frame at a time: records of the tree's top kept in LDS x blocks per compute unit
# of the two traversing kernels.  Usage (GPU box, repo root): bash tools/top_sweep.sh > gpurun_out/top_sweep.txt
for cfg in "0 0" "0 512" "0 -1" "1 0" "1 1024" "1 -1" ${EXTRA_CFGS}; do
  set -- $cfg
  RT_AMD_WF_BLOCKS_PER_CU=$1 RT_AMD_TOP_RECORDS=$2 timeout -k 10 120 python bench.py --no-cpu --in-flight 1 --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('blocks/CU %s top %s: frame %.3f ms  shadow %.3f ms  path+camera %.3f ms' % ('$1','$2',d['ms_per_step'], r['frame_device_ms']['shadowTraceKernel'], r['frame_device_ms']['cameraKernel+pathKernel']))
" || exit 1
done
