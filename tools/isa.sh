#!/bin/bash
# Device ISA of the queue-based renderer's kernels and their register counts / occupancy.
#     bash tools/isa.sh [extra hipcc flags]   -> build_ab/wf.s, build_ab/shadow.s (shadowTraceKernel<false,false>)
cd "$(dirname "$0")/../ray-tracer_amd/csrc" || exit 1
mkdir -p ../../build_ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Xarch_device -fno-slp-vectorize --cuda-device-only -S -x hip rt_wavefront.hip -o ../../build_ab/wf.s "$@" 2>&1 | grep -v "warning\|^$" | tail -3
cd ../..
awk '/^_ZN2rt17shadowTraceKernelILb0ELb0E/{f=1} f{print} /\.amdhsa_kernel _ZN2rt17shadowTraceKernelILb0ELb0E/{exit}' build_ab/wf.s > build_ab/shadow.s
awk '/^_ZN2rt.*: +; @/{name=$1} /; NumVgprs:/{v=$3} /; ScratchSize:/{s=$3} /; Occupancy:/{print name, "vgprs", v, "scratch", s, "occupancy", $3}' build_ab/wf.s | sed 's/ENS_8DevScene.*: / /; s/EvNS_8.*: / /'
