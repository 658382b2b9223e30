#!/usr/bin/env python3
"""Host-side time inside xpbd_multi_world_step for a one-device rehearsal of the sharded world (XPBD_TRANSPORT_LOCAL):
`--shards` shards of `--bodies` stacked boxes (in total) on device 0.  Prints one JSON line: wall ms per frame (step +
synchronize), ms the step call itself took, and -- libraries that have xpbd_multi_world_plan_stats -- how that splits into
enqueueing, waiting for the broadphases' pair counts and waiting for the end of the frame.
Run it once with XPBD_HIP_LIB pointing at an older build for the before / after pair (same scene, same calls)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from constraint_solver_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=4)
    ap.add_argument("--bodies", type=int, default=262144)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--substeps", type=int, default=20)
    ap.add_argument("--serial-enqueue", action="store_true", help="XPBD_MULTI_SERIAL_ENQUEUE: one host thread enqueues all shards")
    args = ap.parse_args()
    kind = capi.SCENE_BOX_STACKS
    bodies, sid = capi.scene_generate(kind, 1, args.bodies, grid_w=capi.default_grid_width(max(args.bodies // 16 // args.shards, 1)))
    extra = {"serial_enqueue": True} if args.serial_enqueue else {}
    with capi.MultiWorld(args.shards, devices=[0] * args.shards, transport=capi.TRANSPORT_LOCAL, halo_margin=0.5, auto_replan=True, **extra) as mw:
        mw.set_polytopes(capi.scene_polytopes(kind))
        t0 = time.perf_counter()
        mw.upload(bodies, sid, 0, args.bodies)
        upload_ms = (time.perf_counter() - t0) * 1e3
        for _ in range(args.warmup):
            mw.step(1 / 60, args.substeps)
        mw.synchronize()
        has_stats = hasattr(mw, "plan_stats")
        try:
            before = mw.plan_stats() if has_stats else None
        except AttributeError:                                  # an older library without the entry point
            before, has_stats = None, False
        in_call, t_all = 0.0, time.perf_counter()
        for _ in range(args.frames):
            t0 = time.perf_counter()
            mw.step(1 / 60, args.substeps)
            in_call += time.perf_counter() - t0
        mw.synchronize()
        wall = time.perf_counter() - t_all
        out = {"library": os.environ.get("XPBD_HIP_LIB", "constraint_solver_amd/lib/libxpbd_hip.so"), "shards": args.shards,
               "enqueue": "one host thread" if args.serial_enqueue else "one host thread per shard",
               "bodies": args.bodies, "frames": args.frames, "substeps": args.substeps, "upload_and_plan_ms": upload_ms,
               "wall_ms_per_frame": wall * 1e3 / args.frames, "step_call_ms_per_frame": in_call * 1e3 / args.frames,
               "halo": mw.halo_stats(), "body_substeps_per_s": args.bodies * args.substeps * args.frames / wall}
        if has_stats:
            after = mw.plan_stats()
            f = float(args.frames)
            out["inside_step_ms_per_frame"] = {"enqueue": (after["ns_enqueue"] - before["ns_enqueue"]) / 1e6 / f,
                                               "wait_broadphase_counts": (after["ns_wait_broadphase"] - before["ns_wait_broadphase"]) / 1e6 / f,
                                               "wait_end_of_frame": (after["ns_wait_frame"] - before["ns_wait_frame"]) / 1e6 / f}
            out["plan"] = after
    print(json.dumps(out))


if __name__ == "__main__":
    main()
