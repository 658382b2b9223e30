#!/usr/bin/env python3
"""Randomised differential test of the contact pipeline: GPU (through the C ABI) against the CPU oracle on random
scenes, sizes, substep counts, narrowphases and pre-test schedules; every body must match bit for bit.
Usage (on a GPU box): python3 scripts/fuzz_contacts.py [--cases 40] [--seed 0]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_binding as ob  # noqa: E402
from constraint_solver_amd import capi  # noqa: E402

POLY = {capi.SCENE_BOXES_DROP: [("cube", 1.0)], capi.SCENE_MIXED_DROP: [("cube", 1.0), ("tetrahedron", 0.5), ("icosahedron", 0.5)]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad = 0
    for case in range(args.cases):
        kind = [capi.SCENE_BOXES_DROP, capi.SCENE_MIXED_DROP][int(rng.integers(2))]
        n = int(rng.integers(40, 2500))
        substeps = int(rng.integers(1, 12))
        frames = int(rng.integers(1, 6))
        narrowphase = int(rng.integers(2))
        schedule = int(rng.integers(3))
        pad = float(rng.choice([0.0, 0.02, 0.1]))
        width = float(rng.uniform(1.5, 12.0))
        seed = int(rng.integers(1 << 30))
        bodies, sid = capi.scene_generate(kind, seed, n)
        r2 = np.random.default_rng(seed)
        bodies[:, 31:33] = r2.uniform(0, width, (n, 2))
        bodies[:, 33] = r2.uniform(0.3, 7.0, n)
        bodies[:, 22:25] *= float(rng.uniform(0.0, 2.0))
        joints = np.zeros(0, dtype=capi.JOINT_DTYPE)
        if rng.random() < 0.3 and n > 10:
            k = int(rng.integers(1, n // 3))
            a = r2.choice(n - 1, size=k, replace=False).astype(np.uint32)
            joints = np.zeros(k, dtype=capi.JOINT_DTYPE)
            joints["body_a"], joints["body_b"] = a, a + 1
            joints["anchor_a"], joints["anchor_b"], joints["distance"] = [0.5, 0.5, 0.5], [0.5, 0.5, 0.5], float(rng.uniform(0.5, 2.0))
        polys = ob.polytopes_array(POLY[kind])
        t0 = time.time()
        want = bodies
        for _ in range(frames):
            want = ob.contacts_step_joints(want, sid, polys, joints, 1 / 60, substeps, pad, narrowphase=narrowphase)
        with capi.World(mode=capi.MODE_CONTACTS) as w:
            w.set_polytopes(capi.scene_polytopes(kind))
            w.set_contact_pad(pad)
            w.set_narrowphase(narrowphase)
            w.set_sat_schedule(schedule)
            w.upload(bodies, sid)
            if len(joints):
                w.set_joints(joints)
            for _ in range(frames):
                w.step(1 / 60, substeps)
            got = w.download()
        same = np.array_equal(got.view(np.uint64), want.view(np.uint64)) or \
            (np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)].view(np.uint64), want[~np.isnan(want)].view(np.uint64)))
        print("case %3d kind %d n %4d substeps %2d frames %d narrowphase %d schedule %d pad %.2f width %5.2f joints %3d: %s (%.1f s)"
              % (case, kind, n, substeps, frames, narrowphase, schedule, pad, width, len(joints), "ok" if same else "MISMATCH", time.time() - t0), flush=True)
        bad += 0 if same else 1
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
