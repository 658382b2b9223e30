#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line with the DRIVER'S arguments, rocprofv3 kernel-trace stats of the
# same command restricted to one part at a time (--only: no extras / CPU leg, so the kernel rows are the timed regime),
# and the PMC passes, each counter group in its own run (never mixed with other trace domains).
# Usage: scripts/gpu_profile.sh <tag> [parts...]   -> everything lands under gpurun_out/<tag>/
#   parts: bench trace trace_big pmc pmc_big contacts_pmc calib  (default: bench trace pmc)
set -eo pipefail
TAG=${1:-prof}; shift || true
PARTS=${*:-bench trace pmc}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python3 bench.py --steps 20 --warmup 5"     # what the driver runs at round end

has() { [[ " $PARTS " == *" $1 "* ]]; }
trace() { # name, extra args
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$name" -- $B "$@" > "$OUT/trace_$name.log" 2>&1
  # --stats averages over the whole process (pre-roll and warm-up too); for the contact scenes also keep the per-kernel
  # time inside the TIMED frames
  local raw; raw=$(find "$OUT/trace_$name" -name '*kernel_trace.csv' | head -1)
  if [ -n "$raw" ]; then
    case $name in pinned*) ;; *) python3 scripts/timed_region_kernels.py "$raw" > "$OUT/timed_$name.json" || true ;; esac
  fi
}
pmc() { # name, "extra args", counters...
  local name=$1 extra=$2; shift 2
  timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- $B $extra > "$OUT/pmc_$name.log" 2>&1
}

pmc_cmd() { # name, counter, command...   (a counter pass over another program: the known-bytes calibration kernels)
  local name=$1 counter=$2; shift 2
  timeout -k 10 400 rocprofv3 --pmc $counter --output-format csv -d "$OUT/pmc_$name" -- "$@" > "$OUT/pmc_$name.log" 2>&1
}

if has bench; then
  timeout -k 10 600 $B > "$OUT/bench.json" 2> "$OUT/bench.err"
fi
if has trace; then
  trace pinned --only pinned
  trace pinned_substep --only pinned --mode substep
  trace stacks --only stacks_262144_sat
  trace mixed_gjk --only mixed_pile_65536_gjk_epa
  trace stacks_gjk --only stacks_262144_gjk_epa
  trace mixed_sat --only mixed_pile_65536_sat
  trace joints --only boxes_262144_joints_65536
  trace boxes_pile --only boxes_pile_262144_sat
fi
if has trace_big; then   # k_step at 2 097 152 bodies, one launch per substep: the HBM-resident roofline line, reproducible from rocprof
  trace pinned_substep_big --only pinned --mode substep --bodies 2097152 --steps 5 --warmup 2
fi
if has calib; then       # FETCH_SIZE / WRITE_SIZE on kernels whose byte counts are known exactly (gathers, streams)
  pmc_cmd calib_fetch FETCH_SIZE python3 scripts/fetch_calibration.py
  pmc_cmd calib_write WRITE_SIZE python3 scripts/fetch_calibration.py
  pmc_cmd calib_rdreq "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" python3 scripts/fetch_calibration.py
  python3 scripts/fetch_calibration.py > "$OUT/calib_known.json" 2> "$OUT/calib_known.err" || true
fi
if has pmc; then
  pmc valu "--only pinned" SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
  pmc wave "--only pinned" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
  pmc fetch_fused "--only pinned" FETCH_SIZE
  pmc write_fused "--only pinned" WRITE_SIZE
  pmc fetch_substep "--only pinned --mode substep" FETCH_SIZE
  pmc write_substep "--only pinned --mode substep" WRITE_SIZE
fi
if has pmc_big; then   # 2 097 152 bodies: 864 MB per launch, far beyond the 256 MiB Infinity Cache
  pmc fetch_substep_big "--only pinned --mode substep --bodies 2097152 --steps 5 --warmup 2" FETCH_SIZE
  pmc write_substep_big "--only pinned --mode substep --bodies 2097152 --steps 5 --warmup 2" WRITE_SIZE
fi
if has contacts_pmc; then
  pmc fetch_stacks "--only stacks_262144_sat" FETCH_SIZE
  pmc write_stacks "--only stacks_262144_sat" WRITE_SIZE
  pmc fetch_mixed_sat "--only mixed_pile_65536_sat" FETCH_SIZE
  pmc write_mixed_sat "--only mixed_pile_65536_sat" WRITE_SIZE
  pmc fetch_boxes_pile "--only boxes_pile_262144_sat" FETCH_SIZE
  pmc write_boxes_pile "--only boxes_pile_262144_sat" WRITE_SIZE
  pmc fetch_stacks_gjk "--only stacks_262144_gjk_epa" FETCH_SIZE
  pmc write_stacks_gjk "--only stacks_262144_gjk_epa" WRITE_SIZE
  pmc fetch_mixed_gjk "--only mixed_pile_65536_gjk_epa" FETCH_SIZE
  pmc write_mixed_gjk "--only mixed_pile_65536_gjk_epa" WRITE_SIZE
  pmc fetch_joints "--only boxes_262144_joints_65536" FETCH_SIZE
  pmc write_joints "--only boxes_262144_joints_65536" WRITE_SIZE
fi
python3 scripts/summarize_profile.py "$OUT" > "$OUT/summary.json"
find "$OUT" -name '*kernel_trace.csv' -delete      # the raw traces (tens of MB each) stay on the box
find "$OUT" -name '*counter_collection.csv' -size +2M -delete   # ... and so do the large raw counter tables (gpurun copies back <= 64 MiB)
cat "$OUT/summary.json"
