#!/bin/bash
# Short A/B of two builds of libxpbd_hip.so on the GJK + EPA path only (GPU box): the narrowphase tests on the new build,
# then mixed pile / boxes pile / stacks with both.  Usage: scripts/ab_gjk.sh <tag> <variant .so>
set -o pipefail
TAG=$1; VARIANT=$2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python3 -m pytest tests/test_gpu_pairs.py tests/test_gpu_fuzz.py -x -q -m gpu > "$OUT/tests.log" 2>&1 || { tail -15 "$OUT/tests.log"; exit 1; }
tail -1 "$OUT/tests.log"
B="python3 bench.py --steps 20 --warmup 5"
for which in after before; do
  if [ $which = before ]; then export XPBD_HIP_LIB=$PWD/$VARIANT; else unset XPBD_HIP_LIB; fi
  timeout -k 10 200 $B --only mixed_pile_65536_gjk_epa > "$OUT/${which}_mixed_pile_65536_gjk_epa.json" 2> "$OUT/${which}_mixed.err" || exit 1
  timeout -k 10 200 $B --mode contacts --no-cpu-baseline --scene stacks --bodies 262144 --narrowphase gjk > "$OUT/${which}_stacks_262144_gjk_epa.json" 2> "$OUT/${which}_stacks_gjk.err" || exit 1
  timeout -k 10 200 $B --mode contacts --no-cpu-baseline --scene boxes-drop --pitch 1.8 --layers 4 --bodies 65536 --narrowphase gjk > "$OUT/${which}_boxes_pile_65536_gjk_epa.json" 2> "$OUT/${which}_boxes_pile_gjk.err" || exit 1
done
python3 - "$OUT" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-40s %.4g body-substeps/s  %.1f us per substep" % (os.path.basename(f)[:-5], d["value"], d["roofline"]["launch_us"]))
PY
