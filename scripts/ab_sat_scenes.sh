#!/bin/bash
# SAT scenes of bench.py with the current build (or $XPBD_HIP_LIB): values and per-kernel time inside the timed frames.
# Usage: scripts/ab_sat_scenes.sh <tag>   -> gpurun_out/<tag>/
set -o pipefail
OUT=gpurun_out/$1; mkdir -p "$OUT"; export TMPDIR=/tmp
for s in boxes_pile_262144_sat mixed_pile_65536_sat stacks_262144_sat; do
  python bench.py --steps 20 --warmup 5 --only $s > "$OUT/bench_$s.json" 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$s" -- python3 bench.py --steps 20 --warmup 5 --only $s > "$OUT/trace_$s.log" 2>&1
  raw=$(find "$OUT/trace_$s" -name "*kernel_trace.csv" | head -1); python3 scripts/timed_region_kernels.py $raw > "$OUT/timed_$s.json"
done
find "$OUT" -name "*kernel_trace.csv" -delete
