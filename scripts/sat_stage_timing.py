#!/usr/bin/env python3
"""Where does the time of the SAT pass go?  Diagnostic builds of xpbd_pairs.hip that return after the face queries
(-DXPBD_SAT_TIMING_STOP=1) or after the edge axes (=2) give wrong contacts, so they cannot settle a pile themselves:

    python scripts/sat_stage_timing.py --save gpurun_out/x/pile.npz          # the shipped build settles the pile
    XPBD_HIP_LIB=<variant>.so rocprofv3 --kernel-trace --stats ... -- python3 scripts/sat_stage_timing.py --load gpurun_out/x/pile.npz

The second form runs `--frames` frames (default 3) from the saved state; the kernel statistics of that run are the variant's."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from constraint_solver_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--save")
    ap.add_argument("--load")
    ap.add_argument("--scene", default="boxes", choices=["boxes", "mixed"])
    ap.add_argument("--bodies", type=int, default=262144)
    ap.add_argument("--frames", type=int, default=3)
    args = ap.parse_args()
    kind = capi.SCENE_BOXES_DROP if args.scene == "boxes" else capi.SCENE_MIXED_DROP
    with capi.World(mode=capi.MODE_CONTACTS) as w:
        w.set_polytopes(capi.scene_polytopes(kind))
        if args.scene == "mixed":
            w.set_max_depenetration_speed(3.0)
        if args.save:
            bodies, sid = capi.scene_pile(kind, 1, args.bodies, 1.8 if args.scene == "boxes" else 1.4, 4)
            w.upload(bodies, sid)
            for _ in range(180 if args.scene == "boxes" else 240):
                w.step(1 / 60, 20)
            np.savez(args.save, bodies=w.download(), sid=sid)
            return
        z = np.load(args.load)
        w.upload(z["bodies"], z["sid"])
        for _ in range(args.frames):
            w.step(1 / 60, 20)
        w.synchronize()
        print(w.contact_stats())


if __name__ == "__main__":
    main()
