#!/bin/bash
# A/B of two builds of libxpbd_hip.so on the contact pipeline (GPU box): the bench's contacts sub-results one at a time
# with the driver's arguments, then FETCH_SIZE / WRITE_SIZE of the stacks and mixed-pile runs (separate PMC passes).
# Usage: scripts/ab_contacts.sh <tag> <variant .so> [pmc]     -> gpurun_out/<tag>/{after,before}_*.json, pmc_*
set -o pipefail
TAG=$1; VARIANT=$2; PMC=$3
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python3 bench.py --steps 20 --warmup 5"
for which in after before; do
  if [ $which = before ]; then export XPBD_HIP_LIB=$PWD/$VARIANT; else unset XPBD_HIP_LIB; fi
  for part in stacks_262144_sat mixed_pile_65536_sat mixed_pile_65536_gjk_epa boxes_262144_joints_65536; do
    timeout -k 10 200 $B --only $part > "$OUT/${which}_$part.json" 2> "$OUT/${which}_$part.err" || exit 1
  done
  timeout -k 10 200 $B --mode contacts --no-cpu-baseline --scene boxes-drop --pitch 1.8 --layers 4 --bodies 262144 > "$OUT/${which}_boxes_pile_262144.json" 2> "$OUT/${which}_boxes_pile.err" || exit 1
  timeout -k 10 200 $B --mode contacts --no-cpu-baseline --scene stacks --bodies 262144 --narrowphase gjk > "$OUT/${which}_stacks_262144_gjk_epa.json" 2> "$OUT/${which}_stacks_gjk.err" || exit 1
  timeout -k 10 200 $B --mode contacts --no-cpu-baseline --scene boxes-drop --pitch 1.8 --layers 4 --bodies 65536 --narrowphase gjk > "$OUT/${which}_boxes_pile_65536_gjk_epa.json" 2> "$OUT/${which}_boxes_pile_gjk.err" || exit 1
  if [ -n "$PMC" ]; then
    for part in stacks_262144_sat mixed_pile_65536_sat; do
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_${which}_${part}_$c" -- $B --only $part > "$OUT/pmc_${which}_${part}_$c.log" 2>&1 || exit 1
      done
    done
  fi
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    name = os.path.basename(f)[:-5]
    if name == "summary":
        continue
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        res[name] = {"value": d["value"], "ms_per_step": d["ms_per_step"], "substep_us": d["roofline"]["launch_us"] if d.get("roofline") else None}
    except Exception as e:
        res[name] = {"error": str(e)}
pmc = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    if not files:
        continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("xpbd::", "").split("(")[0]
        per[k].append(float(r["Counter_Value"]))
    pmc[os.path.basename(d)[4:]] = {k: {"launches": len(v), "avg_KiB_last_400": sum(v[-400:]) / len(v[-400:])} for k, v in per.items() if len(v) >= 100}
print(json.dumps({"bench": res, "pmc": pmc}, indent=1))
PY
