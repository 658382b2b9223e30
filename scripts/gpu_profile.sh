#!/bin/bash
# Runs on the GPU box (through gpurun): bench lines, rocprofv3 kernel-trace stats and the PMC
# passes for k_step, each counter group in its own run (never mixed with other trace domains).
# Usage: scripts/gpu_profile.sh <tag>      -> everything lands under gpurun_out/<tag>/
set -eo pipefail
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python3 bench.py --warmup 40 --no-cpu-baseline --no-extras"

timeout -k 10 300 python3 bench.py --steps 60 --warmup 30 > "$OUT/bench_fused.json" 2> "$OUT/bench_fused.err"
timeout -k 10 300 $B --steps 30 --mode substep > "$OUT/bench_substep.json" 2> "$OUT/bench_substep.err"
timeout -k 10 300 $B --steps 30 --scene mixed-drop --bodies 65536 > "$OUT/bench_mixed65536.json" 2> "$OUT/bench_mixed.err"

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_fused" -- $B --steps 20 > "$OUT/trace_fused.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_substep" -- $B --steps 10 --mode substep > "$OUT/trace_substep.log" 2>&1

pmc() { # name, mode-args, counters...
  local name=$1 extra=$2; shift 2
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- $B --steps 5 $extra > "$OUT/pmc_$name.log" 2>&1
}
pmc valu "" SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
pmc wave "" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
pmc fetch_fused "" FETCH_SIZE
pmc write_fused "" WRITE_SIZE
pmc fetch_substep "--mode substep" FETCH_SIZE
pmc write_substep "--mode substep" WRITE_SIZE
python3 scripts/summarize_profile.py "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
