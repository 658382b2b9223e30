#!/usr/bin/env python3
"""Diagnostics (GPU box): where the waves of k_gjk_pairs spend their cycles, on the settled mixed pile.  Needs a build of the
library with -DXPBD_GJK_TIMING (XPBD_HIP_LIB=<that .so>): clock64() at the phase boundaries, summed over the waves."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from constraint_solver_amd import capi  # noqa: E402

L = capi.hip_lib()
n = 65536
b, sid = capi.scene_pile(capi.SCENE_MIXED_DROP, 1, n, 1.4, 4)
out = (C.c_ulonglong * 8)()
with capi.World(mode=capi.MODE_CONTACTS) as w:
    w.set_polytopes(capi.scene_polytopes(capi.SCENE_MIXED_DROP))
    w.set_narrowphase(capi.NARROWPHASE_GJK_EPA)
    w.upload(b, sid)
    for _ in range(180):
        w.step(1 / 60, 20)
    w.synchronize()
    L.xpbd_debug_gjk_timing(out, 1)
    for _ in range(10):
        w.step(1 / 60, 20)
    w.synchronize()
    L.xpbd_debug_gjk_timing(out, 0)
v = np.array(list(out), dtype=np.float64)
waves = v[4]
print("waves that reached the verdict stage: %d (200 launches), GJK iterations of lane 0's pair per wave: %.2f" % (waves, v[5] / waves))
for k, name in enumerate(["input loads + pre-test", "vertex staging", "GJK iterations", "verdicts + hit list"]):
    print("  %-24s %8.0f cycles per wave" % (name, v[k] / waves))
