#!/bin/bash
# Counter passes over a short bench.py run, one rocprofv3 run per pass (counters are not
# split over passes by the profiler; limits: 8 SQ (SQC included), 4 TCC, 4 TCP, 2 TA, 2 GRBM per pass,
# FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2).  Usage, from the repo root on the GPU box:
#     bash tools/pmc.sh <out-dir> [workload]
# Writes <out-dir>/<pass>/p_counter_collection.csv and <out-dir>/summary.txt.
set -u
out=$(realpath -m "$1"); wl=${2:-dragon}
root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() { # name, counters
  timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d "$out/$1" -o p -- \
    python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu --workload "$wl" ${PMC_BENCH_ARGS:-} > "$out/$1.log" 2>&1 || { echo "pass $1 failed"; return 1; }
}
opt() { pass "$@" || { rm -rf "$out/$1"; true; }; } # diagnostics: a pass the profiler refuses is left out, the rest goes on
pass fetch "FETCH_SIZE GRBM_GUI_ACTIVE" &&
pass write "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" &&
pass sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" &&
pass mix "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA TA_BUSY_avr GRBM_GUI_ACTIVE" &&
pass l1 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum" &&
opt classes "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY" &&
opt levels "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS" &&
opt icache "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" &&
opt stalls "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" &&
python3 "$root/tools/pmc_summarise.py" "$out" > "$out/summary.txt" &&
python3 "$root/tools/pmc_summarise.py" "$out" --counters "$wl" "${PMC_SUMMARY_NAME:-summary.txt}" > "$out/pmc_counters_$wl.json"
