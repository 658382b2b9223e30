#!/usr/bin/env python3
"""Exploration aid (GPU box): how candidate contact scenes evolve -- ground contacts per body, touching pairs and
manifold points per substep, fastest body, frame time -- every 30 frames.  Used to pick the benchmark / test scenes
(a scene with deep initial overlaps flings bodies at > 100 m/s and never reaches a steady contact regime).
Usage: python3 scripts/explore_scenes.py [--bodies 65536] [--frames 240]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from constraint_solver_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", type=int, default=65536)
    ap.add_argument("--frames", type=int, default=240)
    args = ap.parse_args()
    # no initial overlaps: cube radius 0.866, half-size tetrahedron 0.43, icosahedron 0.5 about their centroids; the
    # mixed grid never puts two cubes side by side (shape = index mod 3), layers are 2.5 m apart (1.73 + 0.6 of jitter)
    cases = [("mixed pile pitch 1.4 x 4 layers, SAT", capi.SCENE_MIXED_DROP, 1.4, 4, capi.NARROWPHASE_SAT),
             ("mixed pile pitch 1.4 x 4 layers, GJK/EPA", capi.SCENE_MIXED_DROP, 1.4, 4, capi.NARROWPHASE_GJK_EPA),
             ("mixed single layer pitch 1.4, SAT", capi.SCENE_MIXED_DROP, 1.4, 1, capi.NARROWPHASE_SAT),
             ("boxes pile pitch 1.8 x 4 layers, SAT", capi.SCENE_BOXES_DROP, 1.8, 4, capi.NARROWPHASE_SAT),
             ("boxes pile pitch 1.8 x 4 layers, GJK/EPA", capi.SCENE_BOXES_DROP, 1.8, 4, capi.NARROWPHASE_GJK_EPA)]
    for name, kind, pitch, layers, narrowphase in cases:
        bodies, sid = capi.scene_pile(kind, 1, args.bodies, pitch, layers)
        print("==", name, flush=True)
        with capi.World(mode=capi.MODE_CONTACTS) as w:
            w.set_polytopes(capi.scene_polytopes(kind))
            w.set_narrowphase(narrowphase)
            w.upload(bodies, sid)
            for f0 in range(0, args.frames, 30):
                w.contact_stats()
                w.synchronize()
                t0 = time.perf_counter()
                for _ in range(30):
                    w.step(1 / 60, 20)
                w.synchronize()
                ms = (time.perf_counter() - t0) / 30 * 1e3
                pairs, touching, points = w.contact_stats()
                s = w.download()
                speed = np.linalg.norm(s[:, 22:25], axis=1)
                print("frame %3d: ground %.2f/body, pairs %d, touching %.0f/substep, points %.0f/substep, |v| max %.1f p99 %.2f, "
                      "z mean %.2f max %.1f, nan %d, %.2f ms/frame = %.3g body-substeps/s"
                      % (f0 + 30, len(w.contacts()) / args.bodies, pairs, touching / 600, points / 600, np.nanmax(speed),
                         np.nanpercentile(speed, 99), np.nanmean(s[:, 33]), np.nanmax(s[:, 33]), int(np.isnan(s).any(axis=1).sum()),
                         ms, args.bodies * 20 / ms * 1e3), flush=True)


if __name__ == "__main__":
    main()
