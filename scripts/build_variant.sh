#!/bin/bash
# Builds libxpbd_hip.so of a git revision into constraint_solver_amd/lib/variants/libxpbd_hip_<name>.so (for the A/B runs
# of scripts/ab_contacts.sh: XPBD_HIP_LIB selects the library).  Usage: scripts/build_variant.sh <name> [rev=HEAD] [extra hipcc flags]
set -e
NAME=$1; REV=${2:-HEAD}; shift; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WT=$(mktemp -d /tmp/xpbd_variant.XXXXXX)
git -C "$ROOT" worktree add -f "$WT" "$REV" -q
mkdir -p "$ROOT/constraint_solver_amd/lib/variants"
(cd "$WT" && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -ldl "$@" \
    -o "$ROOT/constraint_solver_amd/lib/variants/libxpbd_hip_$NAME.so" \
    $(python3 -c "import sys; sys.path.insert(0, '$WT'); from constraint_solver_amd import _build; print(' '.join('constraint_solver_amd/csrc/' + s for s in _build.HIP_SOURCES))"))
git -C "$ROOT" worktree remove --force "$WT"
ls -la "$ROOT/constraint_solver_amd/lib/variants/libxpbd_hip_$NAME.so"
