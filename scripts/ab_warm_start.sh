set -o pipefail
mkdir -p gpurun_out/r3k
export TMPDIR=/tmp
for v in sel warm0 warm1; do
  if [ $v = sel ]; then unset XPBD_HIP_LIB; else export XPBD_HIP_LIB=$PWD/constraint_solver_amd/lib/variants/libxpbd_hip_$v.so; fi
  for s in stacks_262144_gjk_epa mixed_pile_65536_gjk_epa; do
    python bench.py --steps 20 --warmup 5 --only $s > gpurun_out/r3k/${v}_$s.json 2>/dev/null
  done
  python bench.py --steps 20 --warmup 5 --mode contacts --no-cpu-baseline --scene boxes-drop --pitch 1.8 --layers 4 --bodies 262144 --narrowphase gjk > gpurun_out/r3k/${v}_boxes_pile_gjk.json 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3k/trace_${v}_mixed -- python3 bench.py --steps 20 --warmup 5 --only mixed_pile_65536_gjk_epa > gpurun_out/r3k/trace_$v.log 2>&1
  raw=$(find gpurun_out/r3k/trace_${v}_mixed -name "*kernel_trace.csv" | head -1); python3 scripts/timed_region_kernels.py $raw > gpurun_out/r3k/timed_${v}_mixed.json
  find gpurun_out/r3k -name "*kernel_trace.csv" -delete
done
