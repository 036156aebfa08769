#!/bin/bash
# Sweep of queue chunk sizes on the bench workload (GPU box).
source tools/ab.sh
for c in 64 128 256 512; do run "chunkPath=$c" RT_AMD_CHUNK_PATH=$c || exit 1; done
for c in 64 128 256 512 1024; do run "chunkShadow=$c" RT_AMD_CHUNK_SHADOW=$c || exit 1; done
