#!/bin/bash
# per-body kernel time (us per substep inside the timed frames) of three scenes with the current build (or $XPBD_HIP_LIB)
# Usage: scripts/ab_per_body_quick.sh <tag>
set -o pipefail
OUT=gpurun_out/$1; mkdir -p "$OUT"; export TMPDIR=/tmp
for s in boxes_pile_262144_sat stacks_262144_sat boxes_262144_joints_65536; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$s" -- python3 bench.py --steps 20 --warmup 5 --only $s > "$OUT/bench_$s.json" 2> "$OUT/trace_$s.log"
  raw=$(find "$OUT/trace_$s" -name "*kernel_trace.csv" | head -1); python3 scripts/timed_region_kernels.py $raw > "$OUT/timed_$s.json"
done
find "$OUT" -name "*kernel_trace.csv" -delete
