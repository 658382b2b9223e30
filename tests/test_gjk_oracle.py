"""GJK + EPA oracle (oracle/xpbd_gjk_oracle.c; extension, the reference has no gjk/epa) checked on CPU against
the exact SAT of oracle/xpbd_pairs_oracle.c: same verdict, same penetration depth, consistent witnesses."""
import math

import numpy as np

import oracle_binding as ob

I4 = [1.0, 0.0, 0.0, 0.0]
CUBE = ob.polytope("cube")
POLYS = [CUBE, ob.polytope("tetrahedron", 0.5), ob.polytope("icosahedron", 0.5)]


def rand_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def test_axis_aligned_overlap():
    r = ob.gjk_epa(([0, 0, 0], I4), ([0.9, 0.1, 0.2], I4), CUBE, CUBE)
    assert r.status == ob.GJK_PENETRATING and abs(r.depth - 0.1) < 1e-12
    assert np.allclose(r.normal.np(), [1, 0, 0], atol=1e-12)          # from A towards B
    assert abs(r.point_a.x - 1.0) < 1e-12 and abs(r.point_b.x - 0.9) < 1e-12
    assert ob.gjk_epa(([0, 0, 0], I4), ([1.5, 0, 0], I4), CUBE, CUBE).status == ob.GJK_SEPARATED
    empty = ob.Polytope()
    assert ob.gjk_epa(([0, 0, 0], I4), ([0.5, 0, 0], I4), CUBE, empty).status == ob.GJK_SEPARATED


def test_exactly_aligned_boxes_are_not_degenerate():
    """Boxes stacked exactly on top of each other: the first two support points and the origin are collinear, the origin lies
    on a face of the first tetrahedron and the Minkowski difference is a 3 x 3 x 3 grid of points (collinear triples,
    coplanar quadruples).  Every such query must still answer; a dropped contact lets a stack sink into itself."""
    rng = np.random.default_rng(11)
    for depth in (0.3, 1e-2, 1e-4, 1e-6):
        for axis in range(3):
            for shift in ([0.0, 0.0, 0.0], [0.3, -0.2, 0.1], [0.5, 0.5, 0.5]):     # the last: corner over centre
                at = np.array([0.25, -1.5, 3.0])
                offset = np.zeros(3)
                offset[axis] = 1.0 - depth
                lateral = np.array(shift) * (np.arange(3) != axis)
                for sign in (1.0, -1.0):
                    r = ob.gjk_epa((at, I4), (at + sign * offset + lateral, I4), CUBE, CUBE)
                    assert r.status == ob.GJK_PENETRATING
                    assert abs(r.depth - depth) < 1e-12
                    want = np.zeros(3)
                    want[axis] = sign
                    np.testing.assert_allclose(r.normal.np(), want, atol=1e-12)
    # and with rounding-sized perturbations of the pose
    for noise in (1e-16, 1e-14, 1e-12):
        for _ in range(300):
            q = np.array(I4) + rng.normal(size=4) * noise
            fb = (np.array([rng.normal() * noise, rng.normal() * noise, 1.0 - 1e-4]), q / np.linalg.norm(q))
            r = ob.gjk_epa(([0.0, 0.0, 0.0], I4), fb, CUBE, CUBE)
            assert r.status == ob.GJK_PENETRATING and abs(r.depth - 1e-4) < 1e-9 and r.normal.z > 1.0 - 1e-9


def test_agrees_with_exact_sat_on_random_pairs():
    rng = np.random.default_rng(5)
    penetrating = separated = 0
    for _ in range(4000):
        pa, pb = POLYS[rng.integers(3)], POLYS[rng.integers(3)]
        fa = (rng.uniform(-0.3, 0.3, 3), rand_quat(rng))
        fb = (rng.uniform(-1.0, 1.0, 3), rand_quat(rng))
        m, r = ob.sat(fa, fb, pa, pb), ob.gjk_epa(fa, fb, pa, pb)
        assert r.status != ob.GJK_DEGENERATE
        assert (r.status == ob.GJK_PENETRATING) == (not m.separated)
        if r.status == ob.GJK_PENETRATING:
            penetrating += 1
            # SAT is exact for polytopes and EPA converges to it.  Compare with the largest of the three SAT
            # queries, not m.separation: the SAT prefers a face axis unless the edge axis wins by more than 1 um.
            assert abs(r.depth + max(m.query)) < 1e-9
            n = r.normal.np()
            assert abs(np.linalg.norm(n) - 1) < 1e-12
            np.testing.assert_allclose(r.point_a.np() - r.point_b.np(), r.depth * n, atol=1e-9)
            # pushing B out along the normal by a bit more than the depth separates the pair
            assert ob.gjk_epa(fa, (fb[0] + n * (r.depth + 1e-6), fb[1]), pa, pb).status == ob.GJK_SEPARATED
            assert r.gjk_iterations <= 32 and r.epa_iterations <= 48
        else:
            separated += 1
    assert penetrating > 800 and separated > 800


def test_swapping_bodies_mirrors_the_result():
    rng = np.random.default_rng(6)
    for _ in range(300):
        fa = (rng.uniform(-0.2, 0.2, 3), rand_quat(rng))
        fb = (rng.uniform(-0.7, 0.7, 3), rand_quat(rng))
        r, w = ob.gjk_epa(fa, fb, CUBE, POLYS[2]), ob.gjk_epa(fb, fa, POLYS[2], CUBE)
        assert r.status == w.status
        if r.status == ob.GJK_PENETRATING:
            assert abs(r.depth - w.depth) < 1e-9


def test_exactly_touching_faces_are_not_penetrating():
    r = ob.gjk_epa(([0, 0, 0], I4), ([1.0, 0, 0], I4), CUBE, CUBE)
    assert r.status in (ob.GJK_SEPARATED, ob.GJK_DEGENERATE)
    h = math.radians(45) / 2
    r = ob.gjk_epa(([0, 0, 0], I4), ([1.3, 0.5, 0.2], [math.cos(h), 0, 0, math.sin(h)]), CUBE, CUBE)
    assert r.status == ob.GJK_PENETRATING and abs(r.depth - 0.1414213562373) < 1e-9


def test_random_convex_hulls_sat_equals_epa():
    """Shapes beyond the reference's three: random 16-vertex hulls (28 faces, 42 edges, all directions distinct)."""
    import hull_util as hu
    hulls = [hu.as_oracle(*hu.random_hull(s)) for s in (1, 2, 3)]
    rng = np.random.default_rng(9)
    hits = 0
    for _ in range(400):
        pa, pb = hulls[rng.integers(3)], hulls[rng.integers(3)]
        fa = (rng.uniform(-0.2, 0.2, 3), rand_quat(rng))
        fb = (rng.uniform(-0.7, 0.7, 3), rand_quat(rng))
        m, r = ob.sat(fa, fb, pa, pb), ob.gjk_epa(fa, fb, pa, pb)
        assert r.status != ob.GJK_DEGENERATE
        assert (r.status == ob.GJK_PENETRATING) == (not m.separated)
        if r.status == ob.GJK_PENETRATING:
            hits += 1
            assert abs(r.depth + max(m.query)) < 1e-9
    assert hits > 100


def test_cached_separating_direction():
    """The pipeline's cache (og_gjk_epa_cached): a separated query leaves the direction that proved it, the next query of
    the pair tries that direction first.  It may only ever answer 'separated' for pairs that are separated (up to a
    rounding-sized overlap); a penetrating query leaves its NORMAL, which warm-starts the next expansion (same depth and
    normal as the cold query to rounding, see test_warm_started_expansion); following a pair along a path the cached
    direction answers most queries."""
    rng = np.random.default_rng(21)
    hits = queries = warm = 0
    for _ in range(300):
        pa, pb = POLYS[rng.integers(3)], POLYS[rng.integers(3)]
        fa = (rng.uniform(-0.3, 0.3, 3), rand_quat(rng))
        qb, start = rand_quat(rng), rng.uniform(-1.0, 1.0, 3)
        start *= 1.6 / np.linalg.norm(start)                          # far enough to start separated ...
        drift, spin = -start / 40 + rng.normal(size=3) * 0.004, rng.normal(size=4) * 0.01
        axis = np.zeros(3)
        for step in range(60):                                        # ... then B drifts through A, tumbling slowly
            q = qb + spin * step
            fb = (start + drift * step, q / np.linalg.norm(q))
            plain = ob.gjk_epa(fa, fb, pa, pb)
            had = axis.any()
            cached, axis = ob.gjk_epa_cached(fa, fb, pa, pb, axis)
            queries += 1
            if cached.gjk_iterations == 0 and cached.status == ob.GJK_SEPARATED and had:
                hits += 1                                             # answered by the cached direction
                m = ob.sat(fa, fb, pa, pb)
                assert m.separated or max(m.query) > -1e-12
                assert plain.status != ob.GJK_PENETRATING or plain.depth < 1e-12
            elif cached.gjk_iterations:                               # the full query from GJK: exactly the uncached result
                assert (cached.status, cached.gjk_iterations, cached.epa_iterations) == (plain.status, plain.gjk_iterations, plain.epa_iterations)
                assert cached.depth == plain.depth
                assert axis.any() == (plain.status != ob.GJK_DEGENERATE)
                if plain.status == ob.GJK_PENETRATING:
                    assert np.array_equal(axis, plain.normal.np())
            else:                                                     # the expansion was warm-started: the same answer to rounding
                assert cached.status == ob.GJK_PENETRATING and had
                assert plain.status == ob.GJK_PENETRATING and abs(cached.depth - plain.depth) < 1e-12
                assert np.linalg.norm(cached.normal.np() - plain.normal.np()) < 1e-9
                warm += 1
    assert hits > 0.3 * queries and warm > 0


def test_warm_started_expansion():
    """seed_polytope: the pair's last penetration normal seeds the polytope with the face of the Minkowski difference that has
    that normal.  (1) Boxes resting exactly on each other -- EPA's worst case: 3 GJK + 7 EPA iterations cold -- are answered
    without GJK in ONE expansion step, with the same depth and normal.  (2) For random penetrating pairs followed over a small
    motion the warm answer equals the cold one to rounding, whatever the cached direction was worth.  (3) A useless cached
    direction (orthogonal to the contact) costs nothing but the attempt: the query falls back to GJK and is the cold one."""
    cube = POLYS[0]
    ident = np.array([1.0, 0.0, 0.0, 0.0])
    for depth in (1e-5, 1e-3, 5e-2):
        fa, fb = (np.zeros(3), ident), (np.array([0.0, 0.0, 1.0 - depth]), ident)
        cold = ob.gjk_epa(fa, fb, cube, cube)
        warm, axis = ob.gjk_epa_cached(fa, fb, cube, cube, np.array([0.0, 0.0, 1.0]))
        assert (cold.status, cold.gjk_iterations, cold.epa_iterations) == (ob.GJK_PENETRATING, 3, 7)
        assert (warm.status, warm.gjk_iterations, warm.epa_iterations) == (ob.GJK_PENETRATING, 0, 1)
        assert abs(warm.depth - cold.depth) < 1e-15 and np.allclose(warm.normal.np(), [0.0, 0.0, 1.0], atol=1e-15)
        assert np.array_equal(axis, warm.normal.np())                                       # one step: keep warm-starting
        sideways, _ = ob.gjk_epa_cached(fa, fb, cube, cube, np.array([1.0, 0.0, 0.0]))       # (3): the pyramid does not hold the origin
        assert (sideways.gjk_iterations, sideways.epa_iterations, sideways.depth) == (cold.gjk_iterations, cold.epa_iterations, cold.depth)
    rng = np.random.default_rng(5)
    seeded = total = saved = 0
    for _ in range(1500):
        pa, pb = POLYS[rng.integers(3)], POLYS[rng.integers(3)]
        fa = (rng.uniform(-0.1, 0.1, 3), rand_quat(rng))
        fb = (rng.uniform(-0.6, 0.6, 3), rand_quat(rng))
        first = ob.gjk_epa(fa, fb, pa, pb)
        if first.status != ob.GJK_PENETRATING:
            continue
        q = fb[1] + rng.normal(scale=1e-3, size=4)
        fb = (fb[0] + rng.normal(scale=1e-3, size=3), q / np.linalg.norm(q))
        cold = ob.gjk_epa(fa, fb, pa, pb)
        warm, axis = ob.gjk_epa_cached(fa, fb, pa, pb, first.normal.np())
        assert warm.status == cold.status
        if cold.status != ob.GJK_PENETRATING:
            continue
        total += 1
        seeded += warm.gjk_iterations == 0
        saved += (cold.gjk_iterations + cold.epa_iterations) - (warm.gjk_iterations + warm.epa_iterations)
        assert abs(warm.depth - cold.depth) < 1e-12 and np.linalg.norm(warm.normal.np() - cold.normal.np()) < 1e-9
        m = ob.sat(fa, fb, pa, pb)                                                           # ... and both equal the exact SAT
        assert not m.separated and abs(-max(m.query) - warm.depth) < 1e-6 + 1e-9
    assert total > 500 and seeded > 0.2 * total and saved > 0
