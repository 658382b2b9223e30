#!/usr/bin/env python3
"""Randomised differential test of the multi-GPU world: several shards on ONE device (in-process transport) against a single
xpbd_world over the same bodies, bit for bit, on random piles and lines with random joints (some long, some hinges), random
numbering (shuffled or not), 2-4 shards, random halo margins, automatic and explicit re-plans, the plan-time gathers through
the device or not, full plans only or light plans.  With XPBD_MULTI_CHECK_PLANS=1 (set here) every light plan is also
compared with the full planner's lists inside the library.
Usage (on a GPU box): python3 scripts/fuzz_multi.py [--cases 30] [--seed 0]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ["XPBD_MULTI_CHECK_PLANS"] = "1"
from constraint_solver_amd import capi  # noqa: E402

DT = 1.0 / 60.0


def one_case(rng):
    kind = [capi.SCENE_BOXES_DROP, capi.SCENE_MIXED_DROP][int(rng.integers(2))]
    n = int(rng.integers(60, 1500))
    n_ranks = int(rng.integers(2, 5))
    substeps = int(rng.integers(2, 9))
    frames = int(rng.integers(4, 25))
    narrowphase = int(rng.integers(2))
    margin = float(rng.choice([0.5, 0.75, 1.0, 1.5]))
    seed = int(rng.integers(1 << 30))
    if rng.random() < 0.5:
        bodies, sid = capi.scene_pile(kind, seed % 1000, n, float(rng.uniform(1.5, 2.2)), int(rng.integers(2, 5)))
    else:
        bodies, sid = capi.scene_generate(kind, seed, n, grid_w=max(n // int(rng.integers(2, 6)), 1))
        bodies[:, 31:33] *= float(rng.uniform(1.3, 2.0)) / 2.0
        bodies[:, 22:25] *= 0.3
    limit = float(rng.choice([0.0, 3.0])) if kind == capi.SCENE_MIXED_DROP else 0.0
    if kind == capi.SCENE_MIXED_DROP:
        limit = 3.0                                                    # (without it light tetrahedra outrun any margin)
    r2 = np.random.default_rng(seed)
    joints = np.zeros(0, dtype=capi.JOINT_DTYPE)
    if rng.random() < 0.6:
        k = int(rng.integers(1, max(n // 4, 2)))
        a = r2.choice(n, size=k, replace=False).astype(np.uint32)
        far = r2.random(k) < 0.2                                        # a fifth of them between arbitrary bodies (long joints)
        b = np.where(far, r2.integers(0, n, k), (a + 1 + r2.integers(0, 3, k)) % n).astype(np.uint32)
        keep = a != b
        a, b = a[keep], b[keep]
        joints = np.zeros(len(a), dtype=capi.JOINT_DTYPE)
        joints["body_a"], joints["body_b"] = a, b
        joints["anchor_a"], joints["anchor_b"] = [0.5, 0.5, 0.5], [0.5, 0.5, 0.5]
        centre = bodies[:, 31:34] + bodies[:, 28:31]
        joints["distance"] = np.linalg.norm(centre[b] - centre[a], axis=1)   # at rest: the joints pull nobody across the world
        hinge = r2.random(len(a)) < 0.2
        axes = r2.normal(size=(2, len(a), 3))
        axes /= np.linalg.norm(axes, axis=2, keepdims=True)
        joints["kind"][hinge] = capi.JOINT_HINGE
        joints["axis_a"][hinge], joints["axis_b"][hinge] = axes[0][hinge], axes[1][hinge]
    if rng.random() < 0.5:                                             # the caller's numbering: anything
        perm = r2.permutation(n)
        inverse = np.empty_like(perm)
        inverse[perm] = np.arange(n)
        bodies, sid = bodies[perm], sid[perm]
        if len(joints):
            joints["body_a"], joints["body_b"] = inverse[joints["body_a"]], inverse[joints["body_b"]]
    through_device, full_plans = bool(rng.random() < 0.5), bool(rng.random() < 0.2)
    replan_at = set(int(x) for x in r2.integers(0, frames, int(rng.integers(0, 4))))
    what = ("kind %d n %4d ranks %d substeps %d frames %2d narrowphase %d margin %.2f joints %3d limit %.0f through_device %d full_plans %d"
            % (kind, n, n_ranks, substeps, frames, narrowphase, margin, len(joints), limit, through_device, full_plans))
    t0 = time.time()
    with capi.World(mode=capi.MODE_CONTACTS) as w:
        w.set_polytopes(capi.scene_polytopes(kind))
        w.set_narrowphase(narrowphase)
        w.set_max_depenetration_speed(limit)
        w.upload(bodies, sid)
        if len(joints):
            w.set_joints(joints)
        for _ in range(frames):
            w.step(DT, substeps)
        one = w.download()
    try:
        with capi.MultiWorld(n_ranks, devices=[0] * n_ranks, transport=capi.TRANSPORT_LOCAL, halo_margin=margin, narrowphase=narrowphase,
                             auto_replan=True, plan_through_device=through_device, full_plans=full_plans) as mw:
            mw.set_polytopes(capi.scene_polytopes(kind))
            mw.set_max_depenetration_speed(limit)
            mw.upload(bodies, sid, 0, n, joints)
            for f in range(frames):
                if f in replan_at:
                    mw.replan()
                mw.step(DT, substeps)
            stats = mw.plan_stats()
            got = mw.download()
    except capi.XpbdError as e:
        if e.code == capi.E_HALO:                                      # a scene that outruns its margin twice in a frame: not a failure
            return True, what + " -> XPBD_E_HALO (margin too small for this scene; frame undone)"
        return False, what + " -> " + str(e)
    ok = np.array_equal(got.view(np.uint64), one.view(np.uint64)) or bool(np.all((got.view(np.uint64) == one.view(np.uint64)) | (np.isnan(got) & np.isnan(one))))
    return ok, what + " plans %d (%d light) rollbacks %d (%.1f s)" % (stats["plans"], stats["light_plans"], stats["rollbacks"], time.time() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad = 0
    for case in range(args.cases):
        ok, what = one_case(rng)
        print("case %3d %s: %s" % (case, what, "ok" if ok else "MISMATCH"), flush=True)
        bad += 0 if ok else 1
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
