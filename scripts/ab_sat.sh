#!/bin/bash
# A/B of two builds of libxpbd_hip.so on the SAT path (GPU box): contact tests on the new build, then stacks / piles
# with both, the piles also with the two-pass schedule forced.  Usage: scripts/ab_sat.sh <tag> <variant .so>
set -o pipefail
TAG=$1; VARIANT=$2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python3 -m pytest tests/test_gpu_pairs.py tests/test_gpu_fullsize_contacts.py tests/test_gpu_fuzz.py tests/test_gpu_multi.py -x -q -m gpu > "$OUT/tests.log" 2>&1 || { tail -15 "$OUT/tests.log"; exit 1; }
tail -1 "$OUT/tests.log"
B="python3 bench.py --steps 20 --warmup 5 --mode contacts --no-cpu-baseline"
for which in after before; do
  if [ $which = before ]; then export XPBD_HIP_LIB=$PWD/$VARIANT; else unset XPBD_HIP_LIB; fi
  for sched in auto two-pass; do
    timeout -k 10 200 $B --scene stacks --bodies 262144 --sat-schedule $sched > "$OUT/${which}_stacks_262144_$sched.json" 2> "$OUT/${which}_1.err" || exit 1
    timeout -k 10 200 $B --scene boxes-drop --pitch 1.8 --layers 4 --bodies 262144 --sat-schedule $sched > "$OUT/${which}_boxes_pile_262144_$sched.json" 2> "$OUT/${which}_2.err" || exit 1
    timeout -k 10 200 $B --scene mixed-drop --pitch 1.4 --layers 4 --bodies 65536 --sat-schedule $sched > "$OUT/${which}_mixed_pile_65536_$sched.json" 2> "$OUT/${which}_3.err" || exit 1
    timeout -k 10 200 $B --scene boxes-drop --bodies 262144 --joints 65536 --sat-schedule $sched > "$OUT/${which}_joints_$sched.json" 2> "$OUT/${which}_4.err" || exit 1
  done
done
python3 - "$OUT" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-44s %.4g body-substeps/s  %.1f us per substep" % (os.path.basename(f)[:-5], d["value"], d["roofline"]["launch_us"]))
PY
