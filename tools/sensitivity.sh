#!/bin/bash
# What the shadow traversal kernel is sensitive to: library builds that ADD work of one kind to every
# sibling-pair step (results unused), timed on the headline frame.  A resource the kernel is bound by
# shows the added work in full; one with slack hides it.
#     bash tools/sensitivity.sh build      (here: builds build_ab/sens_*.so)
#     bash tools/sensitivity.sh run        (GPU box: one line per build)
cd "$(dirname "$0")/.." || exit 1
variants=("base:" "load1:-DRT_PROBE_EXTRA_LOADS=1" "load2:-DRT_PROBE_EXTRA_LOADS=2" "fma8:-DRT_PROBE_EXTRA_VALU=8" "fma16:-DRT_PROBE_EXTRA_VALU=16" "max8:-DRT_PROBE_EXTRA_VALU_B=8" "salu8:-DRT_PROBE_EXTRA_SALU=8" "lds1:-DRT_PROBE_EXTRA_LDS=1")
if [ "$1" = build ]; then
  mkdir -p build_ab
  for v in "${variants[@]}"; do
    make -C ray-tracer_amd/csrc -s OUT="$PWD/build_ab/sens_${v%%:*}.so" EXTRA="${v#*:}" 2>&1 | grep -E "error" -A3
  done
  ls -la build_ab/sens_*.so
else
  export RT_AMD_EXPERIMENTS=1
  for v in "${variants[@]}"; do
    lib=build_ab/sens_${v%%:*}.so
    RT_AMD_LIB=$PWD/$lib timeout -k 10 100 python bench.py --steps 20 --warmup 5 --no-cpu --in-flight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d[\"roofline\"]; print(\"${v%%:*} (${v#*:}):\", \"frame\", d[\"ms_per_step\"], \"shadow\", r[\"frame_device_ms_in_timed_region\"][\"shadowTraceKernel\"], \"path\", r[\"frame_device_ms_in_timed_region\"][\"cameraKernel+pathKernel\"], \"pairs/ray\", r[\"inner_nodes_per_ray\"])" || exit 1
  done
fi
