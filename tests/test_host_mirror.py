"""Host-side mirror of the reference API (constraint_solver_amd/host/constraint_solver.hpp):
set-up math and scene generation, checked on the CPU against the oracle and against an
independent Python splitmix64."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from constraint_solver_amd import capi
from constraint_solver_amd.sharding import shard_range
from golden_util import bits_equal

SHAPES = [(capi.SHAPE_CUBE, "cube"), (capi.SHAPE_TETRAHEDRON, "tetrahedron"), (capi.SHAPE_ICOSAHEDRON, "icosahedron")]


@pytest.mark.parametrize("code,name", SHAPES)
@pytest.mark.parametrize("scale,density", [(1.0, 1.0), (0.5, 5.0), (1.0, 0.1), (2.75, 0.37)])
def test_rigid_metrics_bit_exact_vs_oracle(oracle, code, name, scale, density):
    p = ob.polytope(name, scale)
    m = ob.Metrics()
    oracle.o_rigid_metrics(C.byref(p), density, C.byref(m))
    want = np.concatenate([[m.mass, m.volume], m.center_of_mass.np(), m.inertia_tensor.np().reshape(-1)])
    got = capi.rigid_metrics(code, scale, density)
    assert bits_equal(got, want)
    r = ob.Rigid()
    assert oracle.o_rigid_new(C.byref(m), C.byref(r)) == 1
    assert bits_equal(capi.rigid_new(got), r.np())


def test_rigid_new_singular_inertia_is_an_error():
    m = np.zeros(14)
    m[0] = 1.0
    with pytest.raises(capi.XpbdError) as e:
        capi.rigid_new(m)
    assert e.value.code == capi.E_SINGULAR_INERTIA


def test_rigid_frame_bit_exact_vs_oracle(oracle):
    bodies, _ = capi.scene_generate(capi.SCENE_MIXED, 7, 12)
    for b in bodies:
        f = oracle.o_rigid_frame(C.byref(ob.Rigid.from_np(b)))
        assert bits_equal(capi.rigid_frame(b), np.concatenate([f.position.np(), f.rotation.np()]))


def test_world_new_bit_exact_vs_oracle(oracle):
    p1, p2 = ob.polytope("cube"), ob.polytope("tetrahedron", 0.5)
    a, b = ob.Rigid(), ob.Rigid()
    assert oracle.o_world_new(C.byref(p1), C.byref(p2), C.byref(a), C.byref(b)) == 1
    ha, hb = capi.world_new()
    assert bits_equal(ha, a.np()) and bits_equal(hb, b.np())


@pytest.mark.parametrize("code,name", SHAPES)
def test_planes_bit_exact_vs_oracle(oracle, code, name):
    p = ob.polytope(name, 0.5)
    planes = capi.shape_planes(code, 0.5)
    assert planes.shape[0] == p.n_faces
    for i in range(p.n_faces):
        pl = oracle.o_polytope_plane(C.byref(p), i)
        assert bits_equal(planes[i], np.concatenate([pl.normal.np(), [pl.displacement]]))


def test_scene_shape_tables_match_oracle_polytopes():
    verts, off = capi.scene_shapes(capi.SCENE_MIXED)
    assert list(off) == [0, 8, 12, 24]
    want = np.concatenate([ob.polytope("cube").verts(), ob.polytope("tetrahedron", 0.5).verts(),
                           ob.polytope("icosahedron", 0.5).verts()])
    assert bits_equal(verts, want)
    verts, off = capi.scene_shapes(capi.SCENE_BOXES)
    assert list(off) == [0, 8] and bits_equal(verts, ob.polytope("cube").verts())


# ---- scene generator --------------------------------------------------------------------------
M64 = (1 << 64) - 1


def splitmix_uniforms(seed, body, count):
    state = (seed + (body + 1) * 0x9E3779B97F4A7C15) & M64
    out = []
    for _ in range(count):
        state = (state + 0x9E3779B97F4A7C15) & M64
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        z ^= z >> 31
        out.append((z >> 11) * 2.0 ** -53)
    return out


def test_splitmix64_reference_vector():
    # Vigna's splitmix64.c, seed 1234567: first outputs (state advanced before mixing).
    state, outs = 1234567, []
    for _ in range(3):
        state = (state + 0x9E3779B97F4A7C15) & M64
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        outs.append(z ^ (z >> 31))
    assert outs == [6457827717110365317, 3203168211198807973, 9817491932198370423]


@pytest.mark.parametrize("kind", [capi.SCENE_BOXES, capi.SCENE_MIXED, capi.SCENE_BOXES_DROP, capi.SCENE_MIXED_DROP])
def test_scene_matches_specification(kind):
    n, seed = 50, 3
    bodies, sid = capi.scene_generate(kind, seed, n)
    w = capi.default_grid_width(n)
    assert w == 8
    mixed, drop = bool(kind & 1), bool(kind & 2)
    for i in range(n):
        u = splitmix_uniforms(seed, i, 11)
        b = bodies[i]
        assert sid[i] == (i % 3 if mixed else 0)
        assert list(b[31:33]) == [2.0 * (i % w), 2.0 * (i // w)]
        assert b[33] == (0.4 if drop else -0.05) + 0.6 * u[0]
        q = np.array([-1.0 + 2.0 * x for x in u[1:5]])
        np.testing.assert_allclose(b[34:38], q / np.linalg.norm(q), rtol=1e-15)
        assert list(b[22:25]) == [-1.0 + 2.0 * x for x in u[5:8]]
        assert list(b[25:28]) == [-4.0 + 8.0 * x for x in u[8:11]]
        mass = 1.0 / b[0]
        assert b[12] == -9.81 * mass and b[10] == 0 and b[11] == 0
        assert not b[13:22].any()                                  # internal force, torques: zero
    if drop:
        # lowest possible vertex is 0.366 below `position`, so nobody penetrates at t = 0
        assert bodies[:, 33].min() >= 0.4


def test_scene_is_deterministic_and_shards_consistently():
    n = 1000
    full, sid = capi.scene_generate(capi.SCENE_MIXED, 1, n)
    again, _ = capi.scene_generate(capi.SCENE_MIXED, 1, n)
    assert bits_equal(full, again)
    other, _ = capi.scene_generate(capi.SCENE_MIXED, 2, n)
    assert not bits_equal(full, other)
    for world_size in (1, 2, 3, 8):
        covered = 0
        for rank in range(world_size):
            first, count = shard_range(n, rank, world_size)
            assert first == covered
            part, psid = capi.scene_generate(capi.SCENE_MIXED, 1, n, first=first, count=count)
            assert bits_equal(part, full[first:first + count]) and np.array_equal(psid, sid[first:first + count])
            covered += count
        assert covered == n


def test_shard_range_edge_cases():
    assert shard_range(0, 0, 4) == (0, 0)
    assert [shard_range(5, r, 8)[1] for r in range(8)] == [1, 1, 1, 1, 1, 0, 0, 0]
    assert shard_range(262144, 7, 8) == (229376, 32768)
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


def test_default_grid_width():
    assert capi.default_grid_width(32) == 8
    assert capi.default_grid_width(4096) == 64
    assert capi.default_grid_width(65536) == 256
    assert capi.default_grid_width(262144) == 512
    assert capi.default_grid_width(262145) == 513
