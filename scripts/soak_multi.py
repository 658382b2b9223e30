#!/usr/bin/env python3
"""Soak of the multi-GPU world: piles from the drop to rest (240-300 frames x 20 substeps) as 3 / 4 shards on ONE device against the
single world, bit for bit, every light plan checked against the full planner (XPBD_MULTI_CHECK_PLANS).  Usage: python3 scripts/soak_multi.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["XPBD_MULTI_CHECK_PLANS"] = "1"
import numpy as np
from constraint_solver_amd import capi
DT = 1/60
for kind, n, pitch, limit, frames, ranks in ((capi.SCENE_MIXED_DROP, 65536, 1.4, 3.0, 300, 3), (capi.SCENE_BOXES_DROP, 131072, 1.8, 0.0, 240, 4)):
    bodies, sid = capi.scene_pile(kind, 1, n, pitch, 4)
    t0 = time.time()
    with capi.World(mode=capi.MODE_CONTACTS) as w:
        w.set_polytopes(capi.scene_polytopes(kind)); w.set_max_depenetration_speed(limit); w.upload(bodies, sid)
        for _ in range(frames): w.step(DT, 20)
        one = w.download()
    t1 = time.time()
    with capi.MultiWorld(ranks, devices=[0]*ranks, transport=capi.TRANSPORT_LOCAL, halo_margin=1.0, auto_replan=True, plan_through_device=True) as mw:
        mw.set_polytopes(capi.scene_polytopes(kind)); mw.set_max_depenetration_speed(limit); mw.upload(bodies, sid, 0, n)
        for _ in range(frames): mw.step(DT, 20)
        stats = mw.plan_stats(); got = mw.download()
    t2 = time.time()
    same = np.array_equal(got.view(np.uint64), one.view(np.uint64))
    print("kind %d n %d ranks %d frames %d: bit-identical %s; plans %d (full %d, light %d) rollbacks %d migrated(last) %d owned %d..%d; single %.1f s, sharded %.1f s"
          % (kind, n, ranks, frames, same, stats["plans"], stats["full_plans"], stats["light_plans"], stats["rollbacks"], stats["migrated"], stats["owned_min"], stats["owned_max"], t1-t0, t2-t1), flush=True)
