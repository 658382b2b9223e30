#!/usr/bin/env python3
"""Who is still fast in a settled pile?  Runs bench.py's mixed pile (65 536 cubes / tetrahedra / icosahedra, 4 layers of a
1.4 m grid) for `--frames` frames with a given xpbd_world_set_max_depenetration_speed and prints speed percentiles over
time plus the fastest bodies at the end (shape, height, linear and angular speed, ground contacts)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from constraint_solver_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--speed", type=float, default=3.0)
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--bodies", type=int, default=65536)
    ap.add_argument("--narrowphase", default="sat")
    args = ap.parse_args()
    kind = capi.SCENE_MIXED_DROP
    bodies, sid = capi.scene_pile(kind, 1, args.bodies, 1.4, 4)
    out = {"speed_limit": args.speed, "history": []}
    with capi.World(mode=capi.MODE_CONTACTS) as w:
        w.set_polytopes(capi.scene_polytopes(kind))
        w.set_narrowphase(capi.NARROWPHASE_GJK_EPA if args.narrowphase == "gjk" else capi.NARROWPHASE_SAT)
        w.set_max_depenetration_speed(args.speed)
        w.upload(bodies, sid)
        for f in range(args.frames):
            w.step(1 / 60, 20)
            if f % 30 == 29:
                s = w.download()
                v = np.linalg.norm(s[:, 22:25], axis=1)
                out["history"].append({"frame": f + 1, "p50": float(np.percentile(v, 50)), "p99": float(np.percentile(v, 99)),
                                       "p999": float(np.percentile(v, 99.9)), "max": float(v.max()), "over_5": int((v > 5).sum())})
        s = w.download()
        contacts = w.contacts()
    v = np.linalg.norm(s[:, 22:25], axis=1)
    ground = np.bincount(contacts[:, 0], minlength=len(s)) if len(contacts) else np.zeros(len(s), dtype=int)
    top = np.argsort(-v)[:12]
    out["fastest"] = [{"body": int(i), "shape": int(sid[i]), "z": float(s[i, 33] + s[i, 30]), "speed": float(v[i]), "v": [round(float(x), 2) for x in s[i, 22:25]],
                       "angular_speed": float(np.linalg.norm(s[i, 25:28])), "ground_contacts": int(ground[i]), "xy": [float(s[i, 31]), float(s[i, 32])]} for i in top]
    out["by_shape_over_5"] = {int(k): int(((v > 5) & (sid == k)).sum()) for k in np.unique(sid)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
