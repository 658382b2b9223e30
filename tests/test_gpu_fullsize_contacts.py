"""Contact pipeline (EXTENSION, parity unpinned) at BASELINE.json's sizes: configs[2] = 65 536 mixed convex polyhedra
(SAT and GJK/EPA), configs[4] = 262 144 bodies + 65 536 distance joints.  The oracle cannot run these sizes whole, so:
  * an ISLAND -- a block of bodies moved out of reach of all the others -- inside the full-size world must come out
    exactly as the oracle steps that block alone (the pipeline's results do not depend on the rest of the world, on the
    hash-table size or on launch order);
  * two half-worlds separated by a gap compose to the whole, bit for bit;
  * the one-pass and two-pass pre-test schedules give the same bits;
  * at the settled state the SAT and GJK/EPA narrowphases agree on every neighbour pair (verdict; depth to the SAT's
    1 um face-preference bias);
  * joints: no NaN, every joint length within a bound of its rest length."""
import numpy as np
import pytest

import oracle_binding as ob
from constraint_solver_amd import capi
from golden_util import bits_equal

pytestmark = pytest.mark.gpu

DT = 1.0 / 60.0
MIXED = [("cube", 1.0), ("tetrahedron", 0.5), ("icosahedron", 0.5)]


def run(bodies, sid, kind, frames, substeps, narrowphase=capi.NARROWPHASE_SAT, schedule=capi.SAT_SCHEDULE_AUTO, joints=None,
        pairs_out=None):
    with capi.World(mode=capi.MODE_CONTACTS) as w:
        w.set_polytopes(capi.scene_polytopes(kind))
        w.set_narrowphase(narrowphase)
        w.set_sat_schedule(schedule)
        w.upload(bodies, sid)
        if joints is not None:
            w.set_joints(joints)
        for _ in range(frames):
            w.step(DT, substeps)
        stats = w.contact_stats()
        state = w.download()
        if pairs_out is not None:
            off, nb = w.neighbours(DT)
            i = np.repeat(np.arange(len(off) - 1), np.diff(off))
            keep = nb > i
            pairs_out.append(np.stack([i[keep], nb[keep]], axis=1).astype(np.uint32))
            pairs_out.append(w.narrowphase(pairs_out[0]))
            pairs_out.append(w.narrowphase_gjk(pairs_out[0]))
    return state, stats


LAYERS = 4


def mixed_world(n, island_columns):
    """configs[2]: n mixed polyhedra dropped as a pile (capi.scene_pile: 4 layers of a 1.4 m grid, nothing overlaps at
    t = 0).  The island = the first `island_columns` grid points of EVERY layer, moved 60 m away in x and y so that
    nothing else can reach it.  Returns (bodies, shape ids, ascending indices of the island's bodies)."""
    bodies, sid = capi.scene_pile(capi.SCENE_MIXED_DROP, 1, n, 1.4, LAYERS)
    per_layer = n // LAYERS
    island = np.concatenate([np.arange(l * per_layer, l * per_layer + island_columns) for l in range(LAYERS)])
    bodies[island, 31:33] -= 60.0
    return bodies, sid, island


@pytest.mark.parametrize("narrowphase", [capi.NARROWPHASE_SAT, capi.NARROWPHASE_GJK_EPA])
def test_config3_island_in_the_full_size_world_equals_the_oracle(narrowphase):
    n, frames, substeps = 65536, 75, 20
    bodies, sid, island = mixed_world(n, 128)                   # one grid row of every layer: 512 bodies
    got, stats = run(bodies, sid, capi.SCENE_MIXED_DROP, frames, substeps, narrowphase)
    assert not np.isnan(got).any() and stats[1] > 0
    polys = ob.polytopes_array(MIXED)
    want = bodies[island]
    none = np.zeros(0, dtype=capi.JOINT_DTYPE)
    touching = ob.ContactStats()
    for _ in range(frames):
        want = ob.contacts_step_joints(want, sid[island], polys, none, DT, substeps, 0.02, narrowphase=narrowphase, stats=touching)
    assert touching.n_touching > 1000                           # the island itself piles up: body-body contacts
    assert bits_equal(got[island], want)


def test_config3_half_worlds_compose_schedules_agree_and_sat_agrees_with_epa():
    n, frames, substeps = 65536, 75, 20
    bodies, sid, _ = mixed_world(n, 0)
    # two halves that cannot reach each other: the first 64 rows of every layer and the last 64, 60 m apart
    per_layer, w = n // LAYERS, 128
    row = (np.arange(n) % per_layer) // w
    first = row < w // 2
    bodies[~first, 32] += 60.0
    order = np.concatenate([np.nonzero(first)[0], np.nonzero(~first)[0]])      # halves contiguous, ascending inside each
    bodies, sid = bodies[order], sid[order]
    half = int(first.sum())
    extras = []
    whole, stats = run(bodies, sid, capi.SCENE_MIXED_DROP, frames, substeps, schedule=capi.SAT_SCHEDULE_ONE_PASS, pairs_out=extras)
    two_pass, stats2 = run(bodies, sid, capi.SCENE_MIXED_DROP, frames, substeps, schedule=capi.SAT_SCHEDULE_TWO_PASS)
    assert bits_equal(whole, two_pass) and stats == stats2 and stats[1] > 1000
    lo, _ = run(bodies[:half], sid[:half], capi.SCENE_MIXED_DROP, frames, substeps)
    hi, _ = run(bodies[half:], sid[half:], capi.SCENE_MIXED_DROP, frames, substeps)
    assert bits_equal(np.concatenate([lo, hi]), whole)
    # SAT vs GJK/EPA on every neighbour pair of the settled state
    pairs, sat, gjk = extras
    assert len(pairs) > 20000
    usable = gjk["status"] != capi.GJK_DEGENERATE
    assert usable.mean() > 0.99
    sat_hit, gjk_hit = sat["n_points"] > 0, gjk["status"] == capi.GJK_PENETRATING
    depth_sat, depth_gjk = -sat["separation"], gjk["depth"]
    clear = usable & ~((sat_hit & (depth_sat < 1e-8)) | (gjk_hit & (depth_gjk < 1e-8)))    # not within rounding of touching
    assert np.array_equal(sat_hit[clear], gjk_hit[clear])
    both = clear & sat_hit & gjk_hit
    assert both.sum() > 500
    # the SAT prefers a face axis unless an edge axis beats it by 1 um, so it may report up to 1 um more depth
    diff = depth_sat[both] - depth_gjk[both]
    assert (diff > -1e-9).all() and (diff < 1e-6 + 1e-9).all()


def joints_world(n, n_joints, island):
    """configs[4]: n dropped boxes on the 2 m grid, chains of 5 along x linked by 4 distance joints each at the pitch;
    the first `island` bodies (whole chains, whole rows) moved 30 m away in y."""
    bodies, sid = capi.scene_generate(capi.SCENE_BOXES_DROP, 1, n)
    k = np.arange(n_joints)
    a = (k // 4) * 5 + (k % 4)
    joints = np.zeros(n_joints, dtype=capi.JOINT_DTYPE)
    joints["body_a"], joints["body_b"] = a, a + 1
    joints["anchor_a"], joints["anchor_b"], joints["distance"] = [0.5, 0.5, 0.5], [0.5, 0.5, 0.5], 2.0
    # a chain must not wrap around the end of a grid row (its joint would span the whole row)
    w = capi.default_grid_width(n)
    same_row = (joints["body_a"] // w) == (joints["body_b"] // w)
    joints = joints[same_row]
    bodies[:island, 32] -= 30.0
    return bodies, sid, joints


def joint_lengths(state, joints):
    def anchor(idx, local):
        out = np.empty((len(idx), 3))
        for k, (i, a) in enumerate(zip(idx, local)):
            out[k] = capi.rigid_frame(state[i])[:3] + rotate(capi.rigid_frame(state[i])[3:], a)
        return out

    def rotate(q, v):                                           # q = (s, x, y, z)
        s, u = q[0], q[1:]
        t = np.cross(u, v) + v * s
        return np.cross(u, t) * 2 + v
    pa = anchor(joints["body_a"], joints["anchor_a"])
    pb = anchor(joints["body_b"], joints["anchor_b"])
    return np.linalg.norm(pb - pa, axis=1)


def test_config5_joints_at_full_size():
    n, n_joints, frames, substeps = 262144, 65536, 30, 20
    island = 2 * 512                                            # two grid rows of the 512-wide grid
    bodies, sid, joints = joints_world(n, n_joints, island)
    assert len(joints) > 60000
    got, stats = run(bodies, sid, capi.SCENE_BOXES_DROP, frames, substeps, schedule=capi.SAT_SCHEDULE_ONE_PASS, joints=joints)
    two, stats2 = run(bodies, sid, capi.SCENE_BOXES_DROP, frames, substeps, schedule=capi.SAT_SCHEDULE_TWO_PASS, joints=joints)
    assert bits_equal(got, two) and stats == stats2
    assert not np.isnan(got).any()
    sample = joints[:: 257]                                     # joint lengths stay near the rest length (compliant XPBD)
    lengths = joint_lengths(got, sample)
    assert np.abs(lengths - 2.0).max() < 0.05
    # the island (two rows with their chains) against the oracle
    inside = joints[(joints["body_a"] < island) & (joints["body_b"] < island)]
    assert len(inside) > 500
    polys = ob.polytopes_array([("cube", 1.0)])
    want = bodies[:island]
    for _ in range(frames):
        want = ob.contacts_step_joints(want, sid[:island], polys, inside, DT, substeps, 0.02)
    assert bits_equal(got[:island], want)
