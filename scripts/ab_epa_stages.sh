#!/bin/bash
# Usage: scripts/ab_epa_stages.sh <tag> <scene>: kernel statistics of 3 frames from one saved settled scene on the GJK + EPA path, for
# the shipped build and the diagnostic builds lib/variants/libxpbd_hip_epastop{1,2}.so (-DXPBD_EPA_TIMING_STOP: return after the
# first polytope / after the expansion)
set -o pipefail
OUT=gpurun_out/$1; mkdir -p "$OUT"; export TMPDIR=/tmp
S=${2:-stacks}; B=262144; [ "$S" = mixed ] && B=65536
python3 scripts/sat_stage_timing.py --scene $S --bodies $B --narrowphase gjk --save "$OUT/state_$S.npz" || exit 1
L=$PWD/constraint_solver_amd/lib/variants
for v in shipped epastop1 epastop2; do
  if [ $v = shipped ]; then unset XPBD_HIP_LIB; else export XPBD_HIP_LIB=$L/libxpbd_hip_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_${S}_$v" -- python3 scripts/sat_stage_timing.py --scene $S --narrowphase gjk --load "$OUT/state_$S.npz" > "$OUT/${S}_$v.log" 2>&1
  cp $(find "$OUT/trace_${S}_$v" -name "*kernel_stats.csv" | head -1) "$OUT/${S}_${v}_kernel_stats.csv"
done
find "$OUT" -name "*kernel_trace.csv" -delete; rm -f "$OUT"/state_*.npz
