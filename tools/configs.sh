#!/bin/bash
# One bench line per BASELINE configuration that fits one GPU, at the reference's default camera and,
# for the model scenes, at the frame-filling distance 1.6.  Output: gpurun_out/configs.jsonl
out=gpurun_out/configs.jsonl; : > $out
run() { timeout -k 10 200 python bench.py --no-cpu "$@" 2>/dev/null >> $out || exit 1; echo "done: $*"; }
run --workload cornell --steps 200 --warmup 20
run --workload teapot --steps 200 --warmup 20
run --workload teapot --steps 200 --warmup 20 --camera-distance 1.6
run --workload monkey --steps 200 --warmup 20
run --workload monkey --steps 200 --warmup 20 --camera-distance 1.6
run --workload cube --steps 200 --warmup 20
run --workload dragon --steps 40 --warmup 5
run --workload dragon --steps 10 --warmup 2 --camera-distance 1.6
