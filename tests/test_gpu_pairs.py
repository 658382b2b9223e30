"""Body-body contact EXTENSION on the GPU: the wave-per-pair SAT narrowphase (through the C ABI)
against its CPU oracle, bit for bit.  Parity vs the reference is unpinned for this extension
(the reference has no body-body contacts); what is asserted is HIP == own oracle + invariants."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from constraint_solver_amd import capi
from golden_util import bits_equal

pytestmark = pytest.mark.gpu

ORACLE_POLYS = {0: ("cube", 1.0), 1: ("tetrahedron", 0.5), 2: ("icosahedron", 0.5)}


def cluster(kind, n, seed, spread):
    """n bodies of a scene kind squeezed into a small cloud so that many pairs overlap."""
    rng = np.random.default_rng(seed)
    bodies, sid = capi.scene_generate(kind, seed, n)
    bodies[:, 31:34] = rng.uniform(-spread, spread, (n, 3))
    return bodies, sid


def oracle_manifolds(bodies, sid, pairs):
    L = ob.load()
    polys = {k: ob.polytope(*v) for k, v in ORACLE_POLYS.items()}
    frames = []
    for b in bodies:
        f = L.o_rigid_frame(C.byref(ob.Rigid.from_np(b)))
        frames.append((f.position.np(), f.rotation.np()))
    return [ob.sat(frames[i], frames[j], polys[int(sid[i])], polys[int(sid[j])]) for i, j in pairs]


def assert_same(got, want_list):
    feats = set()
    for g, w in zip(got, want_list):
        if w.separated or w.n_points == 0:
            assert g["n_points"] == 0
            continue
        feats.add(int(w.feature))
        assert (g["n_points"], g["feature"], g["index_a"], g["index_b"]) == (w.n_points, w.feature, w.index_a, w.index_b)
        assert bits_equal(np.array([g["separation"]]), np.array([w.separation]))
        ref, inc = w.points()
        assert bits_equal(g["p_ref"][: w.n_points], ref) and bits_equal(g["p_inc"][: w.n_points], inc)
    return feats


@pytest.mark.parametrize("kind,n,spread", [(capi.SCENE_BOXES, 160, 1.6), (capi.SCENE_MIXED, 150, 1.0)])
def test_narrowphase_matches_oracle(kind, n, spread):
    bodies, sid = cluster(kind, n, 21, spread)
    rng = np.random.default_rng(3)
    pairs = rng.integers(0, n, (4000, 2)).astype(np.uint32)
    pairs = pairs[pairs[:, 0] != pairs[:, 1]]
    with capi.World() as w:
        w.set_polytopes(capi.scene_polytopes(kind))
        w.upload(bodies, sid)
        got = w.narrowphase(pairs)
    want = oracle_manifolds(bodies, sid, pairs)
    feats = assert_same(got, want)
    assert feats == {0, 1, 2}                                       # face-A, face-B and edge-edge all occurred
    touching = sum(1 for m in want if not m.separated)
    assert 0.05 * len(want) < touching < 0.95 * len(want)           # and so did separated pairs


def test_stacked_boxes_give_four_point_manifolds():
    n = 64
    bodies, sid = capi.scene_generate(capi.SCENE_BOXES, 1, n)
    bodies[:, 34:38] = [1.0, 0.0, 0.0, 0.0]
    bodies[:, 31] = 0.0
    bodies[:, 32] = 0.0
    bodies[:, 33] = np.arange(n) * 0.999                            # a column, 1 mm interpenetration per level
    pairs = np.stack([np.arange(n - 1), np.arange(1, n)], axis=1).astype(np.uint32)
    with capi.World() as w:
        w.set_polytopes(capi.scene_polytopes(capi.SCENE_BOXES))
        w.upload(bodies, sid)
        got = w.narrowphase(pairs)
        far = w.narrowphase(np.array([[0, 2], [5, 40]], dtype=np.uint32))
    assert (got["n_points"] == 4).all() and (got["feature"] == capi.FEATURE_FACE_A).all()
    assert (got["index_a"] == 1).all() and (got["index_b"] == 0).all()          # top face of the lower box, bottom of the upper
    np.testing.assert_allclose(got["separation"], -0.001, rtol=1e-9)
    assert (far["n_points"] == 0).all()
    assert_same(got, oracle_manifolds(bodies, sid, pairs))


def test_narrowphase_argument_errors():
    bodies, sid = capi.scene_generate(capi.SCENE_BOXES, 1, 4)
    with capi.World() as w:
        verts, off = capi.scene_shapes(capi.SCENE_BOXES)
        w.set_shapes(verts, off)
        w.upload(bodies, sid)
        with pytest.raises(capi.XpbdError):
            w.narrowphase(np.array([[0, 1]], dtype=np.uint32))      # topology not set
        w.set_polytopes(capi.scene_polytopes(capi.SCENE_BOXES))
        with pytest.raises(capi.XpbdError):
            w.narrowphase(np.array([[0, 9]], dtype=np.uint32))      # body out of range
        assert w.narrowphase(np.zeros((0, 2), dtype=np.uint32)).shape == (0,)
        # the stepper still works on a world configured through set_polytopes
        w.step(1 / 60, 20)
        want, _ = ob.step_bodies(bodies, sid, verts, off, 1 / 60, 20)
        assert bits_equal(w.download(), want)
