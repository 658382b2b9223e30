"""Random convex polytopes (scipy ConvexHull) in the two table formats: the oracle's Polytope struct and the
dict capi.World.set_polytopes takes.  Used to exercise the narrowphases beyond the reference's three shapes."""
import numpy as np
from scipy.spatial import ConvexHull

import oracle_binding as ob


def random_hull(seed, n_points=16, radius=0.5):
    rng = np.random.default_rng(seed)
    while True:
        pts = rng.normal(size=(n_points, 3))
        pts *= radius / np.linalg.norm(pts, axis=1, keepdims=True)      # on a sphere: every point is a hull vertex
        hull = ConvexHull(pts)
        if len(hull.vertices) == n_points:
            break
    faces = hull.simplices.astype(np.uint32)
    edges = sorted({tuple(sorted((int(f[i]), int(f[(i + 1) % 3])))) for f in faces for i in range(3)})
    centroid = pts.mean(axis=0)
    return pts, np.array(edges, dtype=np.uint32), faces, centroid


def as_oracle(pts, edges, faces, centroid):
    p = ob.Polytope()
    p.n_vertices, p.n_edges, p.n_faces = len(pts), len(edges), len(faces)
    assert p.n_vertices <= 32 and p.n_edges <= 64 and p.n_faces <= 32 and 3 * len(faces) <= 128
    for i, v in enumerate(pts):
        p.vertices[i] = ob.vec3(v)
    for i, e in enumerate(edges):
        p.edges[i][0], p.edges[i][1] = int(e[0]), int(e[1])
    for i, f in enumerate(faces):
        p.face_offsets[i] = 3 * i
        for k in range(3):
            p.face_indices[3 * i + k] = int(f[k])
    p.face_offsets[len(faces)] = 3 * len(faces)
    p.centroid = ob.vec3(centroid)
    return p


def as_capi(pts, edges, faces, centroid):
    return {"vertices": pts, "edges": edges, "face_offsets": np.arange(0, 3 * len(faces) + 1, 3, dtype=np.uint32),
            "face_indices": faces.reshape(-1), "centroid": centroid}
