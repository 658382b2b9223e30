"""Known-answer tests that pin the CPU oracle (oracle/xpbd_oracle.c).

The reference has no tests or fixtures, so the oracle is pinned
by closed-form consequences of the reference's formulas
plus algebraic identities of the cgmath primitives it restates.
"""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_binding as ob
from oracle_binding import Constraint, Frame, Metrics, Rigid, Vec3, frame, quat, vec3


def make_rigid(L, shape="cube", scale=1.0, density=1.0):
    p = ob.polytope(shape, scale)
    m = Metrics()
    L.o_rigid_metrics(C.byref(p), density, C.byref(m))
    r = Rigid()
    assert L.o_rigid_new(C.byref(m), C.byref(r)) == 1
    return p, m, r


# ---------------------------------------------------------------- cgmath primitives
def rot_matrix(q):
    s, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - s * z), 2 * (x * z + s * y)],
                     [2 * (x * y + s * z), 1 - 2 * (x * x + z * z), 2 * (y * z - s * x)],
                     [2 * (x * z - s * y), 2 * (y * z + s * x), 1 - 2 * (x * x + y * y)]])


def test_quaternion_rotation_matches_matrix(oracle):
    rng = np.random.default_rng(0)
    for _ in range(50):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        v = rng.normal(size=3)
        got = oracle.o_qrot(quat(q), vec3(v)).np()
        np.testing.assert_allclose(got, rot_matrix(q) @ v, rtol=0, atol=1e-14)


def test_quaternion_product_is_hamilton(oracle):
    i, j, k = quat([0, 1, 0, 0]), quat([0, 0, 1, 0]), quat([0, 0, 0, 1])
    assert list(oracle.o_qmul(i, j).np()) == [0, 0, 0, 1]      # ij = k
    assert list(oracle.o_qmul(j, k).np()) == [0, 1, 0, 0]      # jk = i
    assert list(oracle.o_qmul(k, i).np()) == [0, 0, 1, 0]      # ki = j
    assert list(oracle.o_qmul(i, i).np()) == [-1, 0, 0, 0]


def test_vector_ops_operation_order(oracle):
    # dot = (x*x' + y*y') + z*z' -- distinguishable from other orders in floating point
    a, b = vec3([1e16, 1.0, -1e16]), vec3([1.0, 1.0, 1.0])
    assert oracle.o_dot(a, b) == (1e16 * 1.0 + 1.0 * 1.0) + -1e16 * 1.0
    # normalize = v * (1/|v|), not v / |v|
    v = np.array([3.0, 1.0, 7.0])
    k = 1.0 / math.sqrt((3.0 * 3.0 + 1.0 * 1.0) + 7.0 * 7.0)
    assert list(oracle.o_normalize(vec3(v)).np()) == [3.0 * k, 1.0 * k, 7.0 * k]
    # project_on(o) = o * (a.o / o.o)
    o = np.array([0.3, -2.0, 0.7])
    f = ((3.0 * 0.3 + 1.0 * -2.0) + 7.0 * 0.7) / ((0.3 * 0.3 + -2.0 * -2.0) + 0.7 * 0.7)
    assert list(oracle.o_project_on(vec3(v), vec3(o)).np()) == [0.3 * f, -2.0 * f, 0.7 * f]
    c = oracle.o_cross(vec3([1, 0, 0]), vec3([0, 1, 0])).np()
    assert list(c) == [0, 0, 1]


def test_matrix_invert_and_product(oracle):
    m = ob.Mat3(vec3([2, 0, 0]), vec3([0, 4, 0]), vec3([0, 0, 8]))
    inv = ob.Mat3()
    assert oracle.o_mat3_invert(m, C.byref(inv)) == 1
    assert np.array_equal(inv.np(), np.diag([0.5, 0.25, 0.125]))
    sing = ob.Mat3(vec3([1, 2, 3]), vec3([2, 4, 6]), vec3([0, 0, 1]))
    assert oracle.o_mat3_invert(sing, C.byref(inv)) == 0      # None -> the reference panics (rigid.rs:59)
    rng = np.random.default_rng(1)
    a = rng.normal(size=(3, 3))
    m = ob.Mat3(vec3(a[:, 0]), vec3(a[:, 1]), vec3(a[:, 2]))  # columns
    assert oracle.o_mat3_invert(m, C.byref(inv)) == 1
    np.testing.assert_allclose(inv.np().T, np.linalg.inv(a), rtol=1e-12, atol=1e-13)
    v = rng.normal(size=3)
    np.testing.assert_allclose(oracle.o_mat3_mulv(m, vec3(v)).np(), a @ v, rtol=1e-14, atol=1e-15)


def test_euler_xyz_to_quaternion(oracle):
    # cgmath From<Euler> is q = qx * qy * qz (world.rs:28 uses Deg(10), Deg(15), Deg(5))
    def axis_q(axis, deg):
        h = math.radians(deg) / 2
        q = [math.cos(h), 0, 0, 0]
        q[1 + axis] = math.sin(h)
        return quat(q)
    want = oracle.o_qmul(oracle.o_qmul(axis_q(0, 10.0), axis_q(1, 15.0)), axis_q(2, 5.0)).np()
    got = oracle.o_quat_from_euler_deg(10.0, 15.0, 5.0).np()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-16)
    assert abs(np.linalg.norm(got) - 1) < 1e-15


# ---------------------------------------------------------------- frame.rs
def test_frame_algebra(oracle):
    rng = np.random.default_rng(2)
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    f = frame(rng.normal(size=3), q)
    v = rng.normal(size=3)
    back = oracle.o_frame_mulv(oracle.o_frame_inverse(f), oracle.o_frame_mulv(f, vec3(v))).np()
    np.testing.assert_allclose(back, v, atol=1e-14)
    # delta against itself is zero to rounding; against a translated past it is the translation
    g = oracle.o_frame_mulv(f, vec3(v))
    np.testing.assert_allclose(oracle.o_frame_delta(f, f, g).np(), 0, atol=1e-14)
    past = frame(np.array([f.position.x - 0.25, f.position.y, f.position.z + 1.0]), q)
    np.testing.assert_allclose(oracle.o_frame_delta(f, past, g).np(), [0.25, 0, -1.0], atol=1e-14)
    # (a*b)*v == a*(b*v)
    q2 = rng.normal(size=4)
    q2 /= np.linalg.norm(q2)
    b = frame(rng.normal(size=3), q2)
    lhs = oracle.o_frame_mulv(oracle.o_frame_mul(f, b), vec3(v)).np()
    rhs = oracle.o_frame_mulv(f, oracle.o_frame_mulv(b, vec3(v))).np()
    np.testing.assert_allclose(lhs, rhs, atol=1e-14)


# ---------------------------------------------------------------- K1 mass properties
def test_k1_cube_metrics(oracle):
    _, m, r = make_rigid(oracle, "cube", 1.0, 0.1)
    assert m.volume == pytest.approx(1.0, rel=1e-14)
    assert m.mass == pytest.approx(0.1, rel=1e-14)
    np.testing.assert_allclose(m.center_of_mass.np(), 0.5, rtol=1e-14)
    np.testing.assert_allclose(m.inertia_tensor.np(), np.eye(3) / 60.0, rtol=1e-13, atol=1e-17)
    assert r.inverse_mass == pytest.approx(10.0, rel=1e-14)
    np.testing.assert_allclose(r.inverse_inertia.np(), np.eye(3) * 60.0, rtol=1e-13, atol=1e-12)


def test_k1_half_tetrahedron_metrics(oracle):
    _, m, _ = make_rigid(oracle, "tetrahedron", 0.5, 5.0)
    assert m.volume == pytest.approx(1.0 / 48.0, rel=1e-14)
    assert m.mass == pytest.approx(5.0 / 48.0, rel=1e-14)
    np.testing.assert_allclose(m.center_of_mass.np(), 0.125, rtol=1e-14)
    j = m.inertia_tensor.np()
    np.testing.assert_allclose(np.diag(j), 1.0 / 512.0, rtol=1e-13)
    off = j[~np.eye(3, dtype=bool)]
    np.testing.assert_allclose(off, 1.0 / 3072.0, rtol=1e-12)   # PLUS sign, as the reference computes it
    assert np.array_equal(j, j.T)


def test_k1_icosahedron_metrics(oracle):
    _, m, _ = make_rigid(oracle, "icosahedron", 1.0, 1.0)
    assert m.volume == pytest.approx(2.5361507101204093, rel=1e-13)
    assert m.mass == pytest.approx(2.5361507101204093, rel=1e-13)
    np.testing.assert_allclose(m.center_of_mass.np(), 0.0, atol=1e-15)
    np.testing.assert_allclose(m.inertia_tensor.np(), np.eye(3) * 0.73407035758462, rtol=1e-12, atol=1e-15)


def test_cube_planes_point_outward(oracle):
    p = ob.polytope("cube")
    want = {0: ([0, 0, -1], 0.0), 1: ([0, 0, 1], 1.0), 2: ([0, -1, 0], 0.0), 3: ([1, 0, 0], 1.0),
            4: ([0, 1, 0], 1.0), 5: ([-1, 0, 0], 0.0)}
    for i, (n, d) in want.items():
        pl = oracle.o_polytope_plane(C.byref(p), i)
        assert list(pl.normal.np()) == n and pl.displacement == d
        assert oracle.o_plane_distance(pl, vec3([0.5, 0.5, 0.5])) == -0.5


# ---------------------------------------------------------------- K2 free flight
def test_k2_free_flight_closed_form(oracle):
    p, _, r = make_rigid(oracle, "cube", 1.0, 0.1)
    r.position = vec3([0.0, 0.0, 50.0])
    r.velocity = vec3([0.3, 2.5, -1.0])
    r.external_force = vec3([0.0, 0.0, -2.0])
    n, dt = 1000, 1.0
    h = dt / n
    x0, v0 = r.position.np(), r.velocity.np()
    a = np.array([0.0, 0.0, -2.0 * 10.0])                     # F * inv_m, inv_m = 10 (world.rs body a)
    oracle.o_step(C.byref(r), p.vertices, p.n_vertices, dt, n, None)
    k = np.arange(1, n + 1)
    v_want = v0 + n * h * a
    x_want = x0 + h * (n * v0 + a * h * k.sum())
    # derive() recomputes v = (x - x0)/h every substep, so each substep injects ~ulp(x)/h of
    # velocity noise (reference behaviour, not an oracle artefact): tolerances sized for |x| ~ 50.
    np.testing.assert_allclose(r.position.np(), x_want, rtol=1e-11)
    np.testing.assert_allclose(r.velocity.np(), v_want, rtol=0, atol=1e-9)
    assert list(r.angular_velocity.np()) == [0, 0, 0]
    assert list(r.rotation.np()) == [1, 0, 0, 0]


# ---------------------------------------------------------------- K3 angular shrink
def test_k3_angular_velocity_shrink(oracle):
    p, _, r = make_rigid(oracle, "cube", 1.0, 0.1)
    r.position = vec3([0, 0, 100.0])
    w0 = np.array([-4.0, 1.0, 0.0])                           # world.rs body a
    r.angular_velocity = vec3(w0)
    h = 1.0 / 1500.0
    oracle.o_step(C.byref(r), p.vertices, p.n_vertices, h, 1, None)
    factor = 1.0 / math.sqrt(1.0 + (h * np.linalg.norm(w0) / 2.0) ** 2)
    np.testing.assert_allclose(r.angular_velocity.np(), w0 * factor, rtol=1e-12, atol=1e-15)
    assert factor == pytest.approx(1 - 9.44e-7, abs=1e-9)
    w = r.angular_velocity.np()
    for _ in range(10):
        wn = np.linalg.norm(w)
        oracle.o_step(C.byref(r), p.vertices, p.n_vertices, h, 1, None)
        w = r.angular_velocity.np()
        assert np.linalg.norm(w) / wn == pytest.approx(1.0 / math.sqrt(1.0 + (h * wn / 2.0) ** 2), rel=1e-12)


# ---------------------------------------------------------------- K4 single flat contact
def test_k4_flat_contact_first_lambda(oracle):
    rho, depth, h = 0.1, 2.0 ** -7, 1.0 / 1500.0   # depth exact in binary: frame() = (z + 0.5) - 0.5 rounds otherwise
    p, _, r = make_rigid(oracle, "cube", 1.0, rho)
    r.position = vec3([0.0, 0.0, -depth])
    past = oracle.o_rigid_frame(C.byref(r))
    cs = (Constraint * 32)()
    idx = (C.c_uint32 * 32)()
    n = oracle.o_ground(C.byref(r), past, p.vertices, p.n_vertices, cs, idx)
    assert n == 4 and list(idx[:4]) == [0, 1, 2, 3]
    c = cs[0]
    assert list(c.contact0.np()) == [0.0, 0.0, -depth]
    assert list(c.contact1.np()) == [0.0, 0.0, 0.0]           # at rest: no tangential correction
    assert c.distance == 0.0 and c.rigid == 0
    assert oracle.o_constraint_current_distance(C.byref(c)) == depth
    rp = (C.POINTER(Rigid) * 1)(C.pointer(r))
    w = oracle.o_constraint_inverse_resistance(C.byref(c), rp)
    assert w == pytest.approx(4.0 / rho, rel=1e-13)           # 1/rho * (1 + 6 * 0.5)
    alpha = 1e-6 / (h * h)
    lam = depth / (w + alpha)
    z0 = r.position.z
    oracle.o_constraint_act(C.byref(c), rp, lam)
    assert r.position.z == pytest.approx(z0 + lam * (1.0 / rho), rel=1e-14)
    assert r.position.x == 0.0 and r.position.y == 0.0


def test_solve_is_sequential_gauss_seidel(oracle):
    # After the first constraint moved the body, inverse_resistance of the second must see the NEW pose.
    p, _, r = make_rigid(oracle, "cube", 1.0, 1.0)
    r.position = vec3([0.0, 0.0, -0.05])
    r.rotation = oracle.o_qnormalize(quat([1.0, 0.02, -0.03, 0.0]))
    past = oracle.o_rigid_frame(C.byref(r))
    cs = (Constraint * 32)()
    n = oracle.o_ground(C.byref(r), past, p.vertices, p.n_vertices, cs, None)
    assert n >= 2
    h = 1.0 / 240.0
    whole = Rigid.from_np(r.np())
    oracle.o_solve(C.byref(whole), cs, n, h)
    manual = Rigid.from_np(r.np())
    rp = (C.POINTER(Rigid) * 1)(C.pointer(manual))
    for k in range(n):
        d = oracle.o_constraint_current_distance(C.byref(cs[k])) - cs[k].distance
        lam = d / (oracle.o_constraint_inverse_resistance(C.byref(cs[k]), rp) + 1e-6 / (h * h))
        oracle.o_constraint_act(C.byref(cs[k]), rp, lam)
    assert np.array_equal(whole.np(), manual.np())


# ---------------------------------------------------------------- K5 ordering and the >= test
def test_k5_zero_height_vertex_is_not_a_contact(oracle):
    p, _, r = make_rigid(oracle, "cube", 1.0, 1.0)
    past = oracle.o_rigid_frame(C.byref(r))
    cs = (Constraint * 32)()
    assert oracle.o_ground(C.byref(r), past, p.vertices, p.n_vertices, cs, None) == 0    # z == +0.0
    r.position = vec3([0.0, 0.0, -0.0])
    assert oracle.o_ground(C.byref(r), past, p.vertices, p.n_vertices, cs, None) == 0    # -0.0 >= 0.0
    r.position = vec3([0.0, 0.0, -2.0 ** -53])   # smallest drop that survives frame()'s (z + 0.5) - 0.5
    idx = (C.c_uint32 * 32)()
    assert oracle.o_ground(C.byref(r), past, p.vertices, p.n_vertices, cs, idx) == 4
    assert list(idx[:4]) == [0, 1, 2, 3]                                                  # ascending vertex order


def test_k5_contact_mask_matches_ground_order(oracle):
    p, _, r = make_rigid(oracle, "icosahedron", 0.5, 1.0)
    r.position = vec3([0.0, 0.0, 0.2])
    r.rotation = oracle.o_qnormalize(quat([0.9, 0.1, 0.4, -0.2]))
    r.external_force = vec3([0, 0, -9.81 * 0.3])
    masks = (C.c_uint32 * 8)()
    oracle.o_step(C.byref(r), p.vertices, p.n_vertices, 1 / 60, 8, masks)
    assert any(masks)
    assert all(m < (1 << 12) for m in masks)


# ---------------------------------------------------------------- K6 NaN propagates
def test_k6_coincident_contacts_give_nan(oracle):
    _, _, r = make_rigid(oracle, "cube", 1.0, 1.0)
    c = Constraint(0, vec3([0.1, 0.2, 0.0]), vec3([0.1, 0.2, 0.0]), 0.0)
    cs = (Constraint * 1)(c)
    oracle.o_solve(C.byref(r), cs, 1, 1e-3)
    assert np.isnan(r.position.np()).all() and np.isnan(r.rotation.np()).all()


# ---------------------------------------------------------------- world.rs
def test_world_new_initial_conditions(oracle):
    p1, p2 = ob.polytope("cube"), ob.polytope("tetrahedron", 0.5)
    a, b = Rigid(), Rigid()
    assert oracle.o_world_new(C.byref(p1), C.byref(p2), C.byref(a), C.byref(b)) == 1
    assert a.inverse_mass == pytest.approx(10.0) and b.inverse_mass == pytest.approx(9.6)
    assert list(a.position.np()) == [0, 0, 4] and list(b.position.np()) == [4, 0, 4]
    assert list(a.velocity.np()) == [0, 2.5, 0] and list(b.velocity.np()) == [0, 0, 7]
    assert list(a.angular_velocity.np()) == [-4, 1, 0] and list(b.angular_velocity.np()) == [-5, 5, 0]
    assert a.external_force.z == -2.0 and b.external_force.z == -2.0
    # K2 accelerations quoted in SURVEY 8c: a_z = -20, b_z = -19.2
    assert a.external_force.z * a.inverse_mass == pytest.approx(-20.0)
    assert b.external_force.z * b.inverse_mass == pytest.approx(-19.2)
    # body b keeps the tetrahedron's (non-diagonal) inverse inertia
    assert abs(b.inverse_inertia.np()[0, 1]) > 1.0
    # World::integrate steps both bodies against p1 with 25 substeps
    a2, b2 = Rigid.from_np(a.np()), Rigid.from_np(b.np())
    oracle.o_world_integrate(C.byref(a), C.byref(b), 1 / 60, C.byref(p1))
    oracle.o_step(C.byref(a2), p1.vertices, p1.n_vertices, 1 / 60, 25, None)
    oracle.o_step(C.byref(b2), p1.vertices, p1.n_vertices, 1 / 60, 25, None)
    assert np.array_equal(a.np(), a2.np()) and np.array_equal(b.np(), b2.np())


# ---------------------------------------------------------------- SAT conventions (dead code in the reference)
def test_support_takes_last_maximum(oracle):
    p = ob.polytope("cube")
    ident = frame([0, 0, 0], [1, 0, 0, 0])
    # direction +z: vertices 4..7 tie at z = 1; Iterator::max_by keeps the last -> vertex 7
    assert list(oracle.o_polytope_support(C.byref(p), ident, vec3([0, 0, 1])).np()) == [1, 1, 1]
    assert list(oracle.o_polytope_support(C.byref(p), ident, vec3([0, 0, -1])).np()) == [1, 1, 0]
    ms = oracle.o_polytope_minkowski_support(C.byref(p), ident, frame([3, 0, 0], [1, 0, 0, 0]), vec3([1, 0, 0]))
    assert list(ms.np()) == [1 - 3, 1 - 1, 1 - 1]


def test_face_axes_separation_axis_aligned_cubes(oracle):
    p = ob.polytope("cube")
    fa = frame([0, 0, 0], [1, 0, 0, 0])
    idx = C.c_uint64()
    d = oracle.o_face_axes_separation(fa, frame([1.25, 0, 0], [1, 0, 0, 0]), C.byref(p), C.byref(p), C.byref(idx))
    assert d == 0.25 and idx.value == 3                        # A's +x face (5,1,3,7)
    d = oracle.o_face_axes_separation(fa, frame([0.5, 0.25, 0], [1, 0, 0, 0]), C.byref(p), C.byref(p), C.byref(idx))
    assert d == -0.5 and idx.value == 3                        # deepest-first: '>' keeps the FIRST maximum
    empty = ob.Polytope()
    d = oracle.o_face_axes_separation(fa, fa, C.byref(empty), C.byref(p), C.byref(idx))
    assert d == -np.finfo(np.float64).max and idx.value == 2 ** 64 - 1


def test_edge_axes_separation_skips_parallel_edges(oracle):
    p = ob.polytope("cube")
    fa = frame([0, 0, 0], [1, 0, 0, 0])
    ea, eb = C.c_uint64(), C.c_uint64()
    # B rotated 45 degrees about z and pushed away along x+y: the separating axis is an edge-edge cross product
    h = math.radians(45) / 2
    fb = frame([2.2, 2.2, 0.3], [math.cos(h), 0, 0, math.sin(h)])
    d = oracle.o_edge_axes_separation(fa, fb, C.byref(p), C.byref(p), C.byref(ea), C.byref(eb))
    assert math.isfinite(d) and ea.value < 12 and eb.value < 12
    idx = C.c_uint64()
    df = oracle.o_face_axes_separation(fa, fb, C.byref(p), C.byref(p), C.byref(idx))
    assert d > 0 and df > 0                                    # both queries see the gap
