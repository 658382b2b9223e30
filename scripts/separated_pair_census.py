#!/usr/bin/env python3
"""How many neighbour pairs of a settled pile does the narrowphase see per substep, and how do they end?  (GPU box.)

Pre-rolls the bench's mixed pile, then for one frame's pair list counts, substep by substep on downloaded poses:
pairs, survivors of the tight-sphere pre-test, SAT verdicts (touching / separated by a face axis of A / of B / by an
edge axis), and how often the axis that separated a pair in one substep still separates it in the next -- the hit
rate a per-pair axis cache would have.  usage: separated_pair_census.py [bodies=65536] [kind=mixed|boxes]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from constraint_solver_amd import capi  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    mixed = (sys.argv[2] if len(sys.argv) > 2 else "mixed") == "mixed"
    kind = capi.SCENE_MIXED_DROP if mixed else capi.SCENE_BOXES_DROP
    bodies, sid = capi.scene_pile(kind, 1, n, 1.4 if mixed else 1.8, 4)
    dt, substeps = 1.0 / 60.0, 20
    with capi.World(mode=capi.MODE_CONTACTS) as w:
        w.set_polytopes(capi.scene_polytopes(kind))
        w.upload(bodies, sid)
        for _ in range(185):
            w.step(dt, substeps)
        w.contact_stats()
        off, nb = w.neighbours(dt)
        i = np.repeat(np.arange(n, dtype=np.uint32), np.diff(off))
        keep = nb > i
        pairs = np.stack([i[keep], nb[keep]], axis=1).astype(np.uint32)
        print("neighbour pairs", len(pairs))
        # oracle SAT (it reports its three queries) on a sample of the pairs, at the poses of consecutive SUBSTEPS
        # (one-substep frames of the same h; the poses after the solve rather than after integrate: close enough)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
        import ctypes as C
        import oracle_binding as ob
        L = ob.load()
        names = [("cube", 1.0), ("tetrahedron", 0.5), ("icosahedron", 0.5)] if mixed else [("cube", 1.0)]
        polys = [ob.polytope(*nm) for nm in names]
        radius = np.array([np.linalg.norm(np.array([[v.x, v.y, v.z] for v in p.vertices[:p.n_vertices]])
                                          - np.array([p.centroid.x, p.centroid.y, p.centroid.z]), axis=1).max() for p in polys])
        rng = np.random.default_rng(3)
        sample = pairs[rng.choice(len(pairs), 4000, replace=False)]
        prev = None
        for sub in range(4):
            state = w.download()
            frames = {}
            for b in np.unique(sample):
                f = L.o_rigid_frame(C.byref(ob.Rigid.from_np(state[b])))
                frames[b] = (f.position.np(), f.rotation.np())
            verdict = np.zeros(len(sample), dtype=np.int64)        # 0 touching, 1 spheres, 2 face A, 3 face B, 4 edges
            axis = np.zeros(len(sample), dtype=np.int64)
            for k, (a, b) in enumerate(sample):
                pa, pb = polys[sid[a]], polys[sid[b]]
                ca = L.o_frame_mulv(ob.frame(*frames[a]), pa.centroid).np()
                cb = L.o_frame_mulv(ob.frame(*frames[b]), pb.centroid).np()
                if np.dot(cb - ca, cb - ca) >= (radius[sid[a]] + radius[sid[b]]) ** 2:
                    verdict[k] = 1                                  # the tight-sphere pre-test answers it
                    continue
                m = ob.sat(frames[a], frames[b], pa, pb)
                if not m.separated:
                    continue
                q = list(m.query)
                if q[0] >= 0:
                    verdict[k], axis[k] = 2, m.index_a
                elif q[1] >= 0:
                    verdict[k], axis[k] = 3, m.index_b
                else:
                    verdict[k] = 4
            print("substep %d of the sample: touching %d, spheres apart %d, face of A %d, face of B %d, edge axis only %d"
                  % (sub, (verdict == 0).sum(), (verdict == 1).sum(), (verdict == 2).sum(), (verdict == 3).sum(), (verdict == 4).sum()))
            if prev is not None:
                was = (prev == 2) | (prev == 3)
                print("   face-separated before: %d; still separated now (any axis): %d; by a face axis: %d"
                      % (was.sum(), (was & (verdict > 0)).sum(), (was & ((verdict == 2) | (verdict == 3))).sum()))
            prev = verdict
            w.step(dt / substeps, 1)


if __name__ == "__main__":
    main()
