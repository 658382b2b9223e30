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
    ap.add_argument("--scene", default="boxes", choices=["boxes", "mixed", "stacks"])
    ap.add_argument("--narrowphase", default="sat", choices=["sat", "gjk"])
    ap.add_argument("--bodies", type=int, default=262144)
    ap.add_argument("--frames", type=int, default=3)
    args = ap.parse_args()
    kind = {"boxes": capi.SCENE_BOXES_DROP, "mixed": capi.SCENE_MIXED_DROP, "stacks": capi.SCENE_BOX_STACKS}[args.scene]
    with capi.World(mode=capi.MODE_CONTACTS) as w:
        w.set_polytopes(capi.scene_polytopes(kind))
        w.set_narrowphase(capi.NARROWPHASE_GJK_EPA if args.narrowphase == "gjk" else capi.NARROWPHASE_SAT)
        if args.scene == "mixed":
            w.set_max_depenetration_speed(3.0)
        if args.save:
            if args.scene == "stacks":
                bodies, sid = capi.scene_generate(kind, 1, args.bodies, grid_w=capi.default_grid_width(max(args.bodies // 16, 1)))
            else:
                bodies, sid = capi.scene_pile(kind, 1, args.bodies, 1.8 if args.scene == "boxes" else 1.4, 4)
            w.upload(bodies, sid)
            for _ in range({"boxes": 180, "mixed": 240, "stacks": 30}[args.scene]):
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
