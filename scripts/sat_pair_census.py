#!/usr/bin/env python3
"""What becomes of the candidate pairs of a settled pile in one substep?  (the numbers behind the SAT pass's schedule)

For bench.py's boxes pile (or the mixed pile) after the pre-roll: neighbour pairs, pairs whose tight bounding spheres
overlap (what the pre-test pass lets through apart from cached face axes), and of those: separated by a FACE axis,
separated only by an EDGE axis, touching (by feature).  The face / edge split is taken with the diagnostics of the C ABI
(xpbd_world_narrowphase, xpbd_world_edge_axes_separation): a no-contact pair whose edge query is negative was separated
by a face."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from constraint_solver_amd import capi  # noqa: E402


def quat_rotate(q, v):
    s, u = q[:, :1], q[:, 1:]
    return v + 2.0 * np.cross(u, np.cross(u, v) + s * v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="boxes", choices=["boxes", "mixed"])
    ap.add_argument("--bodies", type=int, default=65536)
    ap.add_argument("--frames", type=int, default=180)
    args = ap.parse_args()
    kind = capi.SCENE_BOXES_DROP if args.scene == "boxes" else capi.SCENE_MIXED_DROP
    pitch = 1.8 if args.scene == "boxes" else 1.4
    bodies, sid = capi.scene_pile(kind, 1, args.bodies, pitch, 4)
    polys = capi.scene_polytopes(kind)
    with capi.World(mode=capi.MODE_CONTACTS) as w:
        w.set_polytopes(polys)
        if args.scene == "mixed":
            w.set_max_depenetration_speed(3.0)
        w.upload(bodies, sid)
        for _ in range(args.frames):
            w.step(1 / 60, 20)
        s = w.download()
        off, nb = w.neighbours(1 / 60)
        i = np.repeat(np.arange(len(s), dtype=np.uint32), np.diff(off))
        keep = i < nb
        pairs = np.stack([i[keep], nb[keep]], 1)
        man = w.narrowphase(pairs)
        edge = w.edge_axes_separation(pairs)
    centroid = np.array([p["centroid"] for p in polys])
    radius = np.array([np.linalg.norm(np.asarray(p["vertices"]) - np.asarray(p["centroid"]), axis=1).max() for p in polys])
    c = s[:, 31:34] + quat_rotate(s[:, 34:38], centroid[sid] - s[:, 28:31]) + 0.0 * s[:, 28:31]
    d = np.linalg.norm(c[pairs[:, 0]] - c[pairs[:, 1]], axis=1)
    overlap = d < radius[sid[pairs[:, 0]]] + radius[sid[pairs[:, 1]]]
    touching = man["n_points"] > 0
    edge_sep = edge["separation"] >= 0.0
    out = {"scene": args.scene, "bodies": args.bodies, "frames": args.frames, "neighbour_pairs": int(len(pairs)),
           "tight_spheres_overlap": int(overlap.sum()), "touching": int(touching.sum()),
           "touching_by_feature": {k: int((touching & (man["feature"] == v)).sum()) for k, v in (("face_a", 0), ("face_b", 1), ("edges", 2))},
           "of the overlapping, not touching": int((overlap & ~touching).sum()),
           "  separated by an edge axis only (edge query >= 0 and no face says so: upper bound = edge query >= 0)": int((overlap & ~touching & edge_sep).sum()),
           "  separated by a face axis (edge query < 0)": int((overlap & ~touching & ~edge_sep).sum()),
           "touching outside overlapping spheres (must be 0)": int((touching & ~overlap).sum())}
    if args.scene == "boxes":
        # face axes of unit cubes in numpy: B's vertices in A's frame (and the other way round) against the slab |x_i| <= 0.5
        verts = np.asarray(polys[0]["vertices"], dtype=np.float64)
        half = np.abs(verts).max(axis=0)

        def conj(q):
            return q * np.array([1.0, -1.0, -1.0, -1.0])

        def face_separates(a, b):
            sep = np.zeros(len(a), dtype=bool)
            for v in verts:
                pass
            world = [s[b, 31:34] + quat_rotate(s[b, 34:38], np.broadcast_to(v - s[b, 28:31], (len(b), 3))) for v in verts]
            local = np.stack([quat_rotate(conj(s[a, 34:38]), wv - s[a, 31:34]) + s[a, 28:31] for wv in world], 1)   # (pairs, 8, 3)
            return ((local.min(axis=1) >= half) | (local.max(axis=1) <= -half)).any(axis=1)

        a, b = pairs[:, 0], pairs[:, 1]
        face = face_separates(a, b) | face_separates(b, a)
        out["numpy: overlapping, not touching, a face axis separates"] = int((overlap & ~touching & face).sum())
        out["numpy: overlapping, not touching, NO face axis separates (edge axis only)"] = int((overlap & ~touching & ~face).sum())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
