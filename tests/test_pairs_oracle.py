"""CPU checks of the body-body contact EXTENSION's oracle (oracle/xpbd_pairs_oracle.c).

Parity for this extension is unpinned: the reference's `sat` is an uncalled stub
(src/collision.rs:37-121).  These tests pin the geometry by construction (known overlaps,
invariants under symmetry and translation) instead."""
import math

import numpy as np
import pytest

import oracle_binding as ob

I4 = [1.0, 0.0, 0.0, 0.0]
CUBE = ob.polytope("cube")


def rand_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def test_axis_aligned_face_contact_full_overlap():
    m = ob.sat(([0, 0, 0], I4), ([0.9, 0, 0], I4), CUBE, CUBE)
    assert not m.separated and m.feature == ob.FEATURE_FACE_A            # a == b == e: A's face wins the tie
    assert (m.index_a, m.index_b) == (3, 5)                              # A's +x face, B's -x face
    assert m.separation == pytest.approx(-0.1, abs=1e-15) and m.n_points == 4
    ref, inc = m.points()
    np.testing.assert_allclose(ref[:, 0], 1.0, atol=1e-15)
    np.testing.assert_allclose(inc[:, 0], 0.9, atol=1e-15)
    assert sorted(map(tuple, inc[:, 1:].round(12))) == [(0, 0), (0, 1), (1, 0), (1, 1)]
    np.testing.assert_allclose(ref[:, 1:], inc[:, 1:], atol=1e-15)       # projection along the normal only


def test_partial_overlap_is_clipped_to_the_reference_face():
    m = ob.sat(([0, 0, 0], I4), ([0.2, 0.1, 0.99], I4), CUBE, CUBE)     # B stacked on A, shifted
    assert m.feature == ob.FEATURE_FACE_A and (m.index_a, m.index_b) == (1, 0) and m.n_points == 4
    ref, inc = m.points()
    assert sorted(map(tuple, inc[:, :2].round(12))) == [(0.2, 0.1), (0.2, 1.0), (1.0, 0.1), (1.0, 1.0)]
    np.testing.assert_allclose(inc[:, 2], 0.99, atol=1e-15)
    np.testing.assert_allclose(ref[:, 2], 1.0, atol=1e-15)


def test_separated_and_touching_pairs_produce_no_contacts():
    assert ob.sat(([0, 0, 0], I4), ([1.1, 0, 0], I4), CUBE, CUBE).separated
    assert ob.sat(([0, 0, 0], I4), ([1.0, 0, 0], I4), CUBE, CUBE).separated      # distance 0 is `>= 0`: separated
    assert ob.sat(([0, 0, 0], I4), ([0, 0, -1.5], I4), CUBE, CUBE).separated
    empty = ob.Polytope()
    assert ob.sat(([0, 0, 0], I4), ([0.5, 0, 0], I4), CUBE, empty).separated


def test_edge_against_face_gives_two_points():
    h = math.radians(45) / 2
    m = ob.sat(([0, 0, 0], I4), ([1.3, 0.5, 0.2], [math.cos(h), 0, 0, math.sin(h)]), CUBE, CUBE)
    assert not m.separated and m.n_points == 2 and m.feature in (ob.FEATURE_FACE_A, ob.FEATURE_FACE_B)
    ref, inc = m.points()
    np.testing.assert_allclose(np.linalg.norm(ref - inc, axis=1), -m.separation, rtol=1e-9)


def test_random_pairs_invariants():
    rng = np.random.default_rng(11)
    polys = [CUBE, ob.polytope("tetrahedron", 0.5), ob.polytope("icosahedron", 0.5)]
    seen = set()
    for _ in range(1500):
        pa, pb = polys[rng.integers(3)], polys[rng.integers(3)]
        fa = (rng.uniform(-0.3, 0.3, 3), rand_quat(rng))
        fb = (rng.uniform(-0.9, 0.9, 3), rand_quat(rng))
        m = ob.sat(fa, fb, pa, pb)
        if m.separated:
            assert m.n_points == 0
            continue
        seen.add(m.feature)
        assert m.separation < 0 and m.n_points <= 8
        ref, inc = m.points()
        if m.n_points and m.feature != ob.FEATURE_EDGES:
            depth = np.linalg.norm(ref - inc, axis=1)
            assert (depth <= -m.separation * (1 + 1e-9) + 1e-12).all()       # no point deeper than the SAT depth
        if m.feature == ob.FEATURE_EDGES:
            # both edges are supporting features, so the closest points are at least the SAT depth apart
            assert m.n_points == 1 and np.linalg.norm(ref[0] - inc[0]) >= -m.separation * (1 - 1e-9)
        # moving B out along the contact normal by more than the depth separates the pair
        if m.n_points and m.feature != ob.FEATURE_EDGES:
            n = (ref[0] - inc[0]) / np.linalg.norm(ref[0] - inc[0])          # from the incident point to the surface
            push = n * (-m.separation + 1e-6)
            if m.feature == ob.FEATURE_FACE_A:                                # incident body is B: move B outward
                assert ob.sat(fa, (fb[0] + push, fb[1]), pa, pb).separated
            else:                                                             # incident body is A
                assert ob.sat((fa[0] + push, fa[1]), fb, pa, pb).separated
    assert seen == {ob.FEATURE_FACE_A, ob.FEATURE_FACE_B, ob.FEATURE_EDGES}


def test_swapping_the_bodies_swaps_the_roles():
    rng = np.random.default_rng(12)
    for _ in range(300):
        fa = (rng.uniform(-0.2, 0.2, 3), rand_quat(rng))
        fb = (rng.uniform(-0.8, 0.8, 3), rand_quat(rng))
        m, w = ob.sat(fa, fb, CUBE, CUBE), ob.sat(fb, fa, CUBE, CUBE)
        assert m.separated == w.separated
        if not m.separated and m.query[0] != m.query[1]:
            assert m.query[0] == w.query[1] and m.query[1] == w.query[0]      # the two face queries trade places


def test_translation_invariance_of_the_manifold():
    rng = np.random.default_rng(13)
    shift = np.array([64.0, -32.0, 16.0])                                     # exact in binary at these magnitudes
    for _ in range(200):
        fa = (rng.uniform(-0.2, 0.2, 3).round(3), rand_quat(rng))
        fb = (rng.uniform(-0.8, 0.8, 3).round(3), rand_quat(rng))
        m = ob.sat(fa, fb, CUBE, CUBE)
        w = ob.sat((fa[0] + shift, fa[1]), (fb[0] + shift, fb[1]), CUBE, CUBE)
        assert m.separated == w.separated
        if not m.separated and m.feature == w.feature and m.n_points == w.n_points:
            np.testing.assert_allclose(w.points()[1] - shift, m.points()[1], atol=1e-9)
