#!/bin/bash
# Runs the CPU-side native code (the C oracle and the C++ host mirror) under AddressSanitizer +
# UndefinedBehaviorSanitizer through the existing pytest suites.  GPU sanitizers are not available
# on the pool, so this covers the CPU build only.  Restores the normal libraries afterwards.
set -eo pipefail
cd "$(dirname "$0")/.."
python -m constraint_solver_amd._build > /dev/null
TMP=$(mktemp -d)
cp oracle/libxpbd_oracle.so "$TMP/oracle.so"
cp constraint_solver_amd/lib/libxpbd_host.so "$TMP/host.so"
restore() {
  cp "$TMP/oracle.so" oracle/libxpbd_oracle.so; cp "$TMP/host.so" constraint_solver_amd/lib/libxpbd_host.so
  touch oracle/libxpbd_oracle.so constraint_solver_amd/lib/libxpbd_host.so; rm -rf "$TMP"
}
trap restore EXIT
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g"
gcc $SAN -ffp-contract=off -fno-fast-math -fPIC -std=c99 -fopenmp -shared -o oracle/libxpbd_oracle.so oracle/*.c -lm
g++ $SAN -ffp-contract=off -std=c++17 -fPIC -shared -o constraint_solver_amd/lib/libxpbd_host.so \
    constraint_solver_amd/host/host_capi.cpp -Lconstraint_solver_amd/lib -lxpbd_hip -Wl,-rpath,'$ORIGIN'
touch oracle/libxpbd_oracle.so constraint_solver_amd/lib/libxpbd_host.so constraint_solver_amd/lib/xpbd_headless
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  python -m pytest -q -p no:cacheprovider tests/test_oracle_kat.py tests/test_pairs_oracle.py tests/test_gjk_oracle.py \
  tests/test_joints_oracle.py tests/test_golden_oracle.py tests/test_host_mirror.py
# ... and the host-only planner of the multi-GPU world (partition, halo plans: csrc/xpbd_multi.cpp compiled by g++ with the
# sanitizers and linked with the other objects of the HIP library) through its CPU tests.
g++ $SAN -std=c++17 -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c constraint_solver_amd/csrc/xpbd_multi.cpp -o "$TMP/multi_san.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$TMP/libxpbd_hip_san.so" \
    $(ls constraint_solver_amd/lib/obj/*.o | grep -v xpbd_multi) "$TMP/multi_san.o" -ldl -lpthread
XPBD_HIP_LIB="$TMP/libxpbd_hip_san.so" LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  python -m pytest -q -p no:cacheprovider tests/test_halo_plan_native.py tests/test_abi.py

