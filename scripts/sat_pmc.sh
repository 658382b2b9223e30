#!/bin/bash
# Counter passes over 3 frames of a saved settled boxes pile (scripts/sat_stage_timing.py): what bounds k_sat_survivors?
# Usage: scripts/sat_pmc.sh <tag> [scene]   -> gpurun_out/<tag>/pmc_<pass>.json (per-kernel sums; the raw CSVs are deleted)
set -o pipefail
OUT=gpurun_out/$1; mkdir -p "$OUT"; export TMPDIR=/tmp
S=${2:-boxes}; B=262144; [ "$S" = mixed ] && B=65536
python3 scripts/sat_stage_timing.py --scene $S --bodies $B --save "$OUT/pile_$S.npz" || exit 1
pass() { # name counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 scripts/sat_stage_timing.py --scene $S --load "$OUT/pile_$S.npz" > "$OUT/pmc_$name.log" 2>&1 || return 1
  python3 - "$OUT/pmc_$name" "$OUT/pmc_${S}_$name.json" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
d, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("xpbd::", "").split("(")[0].strip()
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
json.dump({k: {"dispatches": len(calls[k]), **v} for k, v in acc.items() if "sat" in k or "pretest" in k or "pair_solve" in k}, open(out, "w"), indent=1)
PY
  rm -rf "$OUT/pmc_$name"
}
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM &&
pass cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS &&
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64
rm -f "$OUT"/pile_*.npz
