#!/usr/bin/env python3
"""Randomised differential test of the PINNED path (solver::step semantics): GPU through the C ABI against the CPU
oracle on random scenes, body counts, time steps, substep counts, schedules (fused / per substep) and block sizes;
poses and the contact masks of every substep must match bit for bit.
Usage (on a GPU box): python3 scripts/fuzz_pinned.py [--cases 60] [--seed 0]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_binding as ob  # noqa: E402
from constraint_solver_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad = 0
    for case in range(args.cases):
        kind = int(rng.integers(4))
        n = int(rng.integers(1, 6000))
        substeps = int(rng.integers(1, 40))
        frames = int(rng.integers(1, 8))
        dt = float(rng.choice([1 / 60, 1 / 30, 1 / 120, 0.01]))
        mode = [capi.MODE_FUSED, capi.MODE_PER_SUBSTEP][int(rng.integers(2))]
        block = int(rng.choice([0, 64, 128, 256]))
        seed = int(rng.integers(1 << 30))
        bodies, sid = capi.scene_generate(kind, seed, n)
        bodies[:, 22:28] *= float(rng.uniform(0.0, 3.0))
        bodies[:, 33] += float(rng.uniform(-0.4, 0.4))
        verts, off = capi.scene_shapes(kind)
        want, want_masks = bodies, []
        for _ in range(frames):
            want, m = ob.step_bodies(want, sid, verts, off, dt, substeps, want_masks=True)
            want_masks.append(m)
        ok = True
        with capi.World(mode=mode, block_size=block, trace_contacts=True) as w:
            w.set_shapes(verts, off)
            w.upload(bodies, sid)
            for f in range(frames):
                w.step(dt, substeps)
                ok &= np.array_equal(w.contact_masks(substeps), want_masks[f])
            got = w.download()
        ok &= np.array_equal(got.view(np.uint64), want.view(np.uint64))
        print("case %3d kind %d n %4d substeps %2d frames %d dt %.5f mode %d block %3d: %s" % (case, kind, n, substeps, frames, dt, mode, block, "ok" if ok else "MISMATCH"), flush=True)
        bad += 0 if ok else 1
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
