"""The native multi-GPU world (xpbd_multi_world_*, csrc/xpbd_multi.cpp) on the one GPU of the box: several shards on
device 0 with the in-process transport (XPBD_TRANSPORT_LOCAL) must reproduce a single xpbd_world over the same bodies BIT FOR
BIT -- shards cut by the library from the spatial-hash cell order whatever the caller's numbering, halo plan built by the
library, one all-gather per substep, joints crossing shard boundaries, re-plans that re-balance -- and the oracle; a frame in
which a body outruns halo_margin is UNDONE and reported (XPBD_E_HALO) or re-run after a re-plan, never a silently lost
contact; the RCCL transport is exercised with a one-rank communicator (RCCL refuses two ranks on one device).
EXTENSION: parity unpinned."""
import numpy as np
import pytest

import oracle_binding as ob
from constraint_solver_amd import capi
from golden_util import bits_equal
from halo_common import POLY_NAMES, chain_joints, expected, line_scene

pytestmark = pytest.mark.gpu

DT = 1.0 / 60.0


def single(bodies, sid, kind, frames, substeps, joints=None, narrowphase=capi.NARROWPHASE_SAT, pad=0.02):
    with capi.World(mode=capi.MODE_CONTACTS) as w:
        w.set_polytopes(capi.scene_polytopes(kind))
        w.set_contact_pad(pad)
        w.set_narrowphase(narrowphase)
        w.upload(bodies, sid)
        if joints is not None:
            w.set_joints(joints)
        for _ in range(frames):
            w.step(DT, substeps)
        return w.download(), w.contact_stats()


def sharded(bodies, sid, kind, n_ranks, frames, substeps, joints=None, narrowphase=capi.NARROWPHASE_SAT, margin=0.75, replan_at=(),
            auto_replan=False, pad=0.02, plan_through_device=False):
    with capi.MultiWorld(n_ranks, devices=[0] * n_ranks, transport=capi.TRANSPORT_LOCAL, halo_margin=margin, narrowphase=narrowphase,
                         auto_replan=auto_replan, pad=pad, plan_through_device=plan_through_device) as mw:
        mw.set_polytopes(capi.scene_polytopes(kind))
        mw.upload(bodies, sid, 0, len(bodies), joints)
        stats0 = mw.halo_stats()
        for f in range(frames):
            if f in replan_at:
                mw.replan()
            mw.step(DT, substeps)
        return mw.download(), stats0, mw.halo_stats(), mw.contact_stats()


@pytest.mark.parametrize("n_ranks,through_device", [(2, False), (3, False), (3, True)])
def test_sharded_equals_single_device_and_oracle(n_ranks, through_device):
    """(through_device: the plan-time all-gathers -- cell keys, boundary lists, boundary records -- take the staged device path
    of a one-process-per-GPU run instead of the in-process shortcut)"""
    kind, n, substeps, frames = capi.SCENE_BOXES_DROP, 96, 6, 40
    bodies, sid = line_scene(capi, kind, n, 11, 1.3)
    got, s0, s1, cstats = sharded(bodies, sid, kind, n_ranks, frames, substeps, replan_at=(25,), auto_replan=True,
                                  plan_through_device=through_device)
    assert s0["ghosts"] > 0 and s0["boundary"] > 0 and s1["plans"] >= 3      # the explicit one and the automatic ones
    one, one_stats = single(bodies, sid, kind, frames, substeps)
    assert one_stats[1] > 0                                              # body-body contacts did happen
    assert bits_equal(got, one)
    assert bits_equal(got, expected(ob, bodies, sid, kind, substeps, frames, 0.02))


@pytest.mark.parametrize("which", ["every third body", "one shard's side only"])
def test_sharded_bodies_of_one_shape_with_different_masses(which):
    """Bodies of ONE shape with different mass properties: a world keeps the mass properties per shape while all bodies of a
    shape share them bit for bit (xpbd_world.cpp: stat_shared) and per body otherwise.  The re-plans re-pack the shards on the
    device and must notice when an ARRIVING body (a ghost, a migrated body) breaks a shard's sharing: heavy bodies everywhere,
    and heavy bodies at one end of the line only (so that one shard starts with shared properties and learns better from
    its first ghosts)."""
    kind, n, substeps, frames = capi.SCENE_BOXES_DROP, 96, 6, 40
    bodies, sid = line_scene(capi, kind, n, 13, 1.3)
    order = np.argsort(bodies[:, 31])                                     # along the line the shards are cut across
    heavy = order[::3] if which == "every third body" else order[: n // 2 - 2]
    bodies[heavy, 0] *= 0.25                                              # inverse mass / 4 ...
    bodies[heavy, 1:10] *= 0.25                                           # ... and the inverse inertia with it
    got, s0, s1, _ = sharded(bodies, sid, kind, 2, frames, substeps, replan_at=(10, 25), auto_replan=True)
    assert s0["ghosts"] > 0 and s1["plans"] >= 3
    one, one_stats = single(bodies, sid, kind, frames, substeps)
    assert one_stats[1] > 0
    assert bits_equal(got, one)
    assert bits_equal(got, expected(ob, bodies, sid, kind, substeps, frames, 0.02))


def test_sharded_mixed_shapes_carry_their_shape_ids_into_the_ghosts():
    kind, n, substeps, frames = capi.SCENE_MIXED_DROP, 150, 6, 30
    bodies, sid = line_scene(capi, kind, n, 4, 1.4)
    bodies[:, 33] += 0.6                                                 # (the scene's icosahedra start inside the ground)
    got, s0, _, _ = sharded(bodies, sid, kind, 3, frames, substeps, margin=20.0)
    assert s0["ghosts"] > 0
    one, one_stats = single(bodies, sid, kind, frames, substeps)
    assert one_stats[1] > 0 and bits_equal(got, one)


def test_sharded_pile_with_joints_across_shards_and_gjk():
    kind, n, substeps, frames = capi.SCENE_BOXES_DROP, 600, 8, 45
    bodies, sid = capi.scene_pile(kind, 5, n, 1.8, 3)
    per_layer = n // 3
    # order the bodies by grid row so that index ranges are slabs of space (thin halos), as the ABI asks of its caller
    w = capi.default_grid_width(per_layer)
    row = (np.arange(n) % per_layer) // w
    order = np.lexsort((np.arange(n), row))
    bodies, sid = bodies[order], sid[order]
    joints = chain_joints(capi, n, every=7)                             # partners 7 apart: many joints cross shard boundaries
    centre = bodies[:, 31:34] + bodies[:, 28:31]                        # ... at rest at t = 0 (anchors are the centres of mass)
    joints["distance"] = np.linalg.norm(centre[joints["body_b"]] - centre[joints["body_a"]], axis=1)
    for narrowphase in (capi.NARROWPHASE_SAT, capi.NARROWPHASE_GJK_EPA):
        got, s0, s1, _ = sharded(bodies, sid, kind, 3, frames, substeps, joints=joints, narrowphase=narrowphase, margin=1.0,
                                 auto_replan=True)
        assert 0 < s0["ghosts"] < 2 * n
        one, one_stats = single(bodies, sid, kind, frames, substeps, joints=joints, narrowphase=narrowphase)
        assert one_stats[1] > 100 and not np.isnan(one).any()
        assert bits_equal(got, one)


def boundary_body(kind, bodies, sid, n_ranks, margin):
    """Index of a body the plan puts next to a shard boundary (travel allowance = halo_margin, not the larger one of a
    body deep inside its slab)."""
    with capi.MultiWorld(n_ranks, devices=[0] * n_ranks, transport=capi.TRANSPORT_LOCAL, halo_margin=margin) as mw:
        mw.set_polytopes(capi.scene_polytopes(kind))
        mw.upload(bodies, sid, 0, len(bodies))
        owner = mw.owners()
    x = bodies[:, 31]
    edge_x = 0.5 * (x[owner == 0].max() + x[owner == 1].min())
    return int(np.argmin(np.abs(x - edge_x) + 1e3 * (owner != 0)))


def test_a_frame_in_which_a_body_outruns_the_halo_margin_is_undone_and_reported():
    """The validity check sits at the END of the frame: the frame that used up the allowance returns XPBD_E_HALO with the
    state of its START in place, so no state with possibly missed contacts ever reaches the caller; after a re-plan the
    same frame goes through and the run equals the single device."""
    kind, n, substeps, margin = capi.SCENE_BOXES_DROP, 64, 4, 0.5
    bodies, sid = line_scene(capi, kind, n, 3, 1.5)
    fast = boundary_body(kind, bodies, sid, 2, margin)
    bodies[fast, 22:25] = [0.0, 0.0, 20.0]                               # 20 m/s upwards: 0.33 m per frame against a 0.5 m margin
    with capi.MultiWorld(2, devices=[0, 0], transport=capi.TRANSPORT_LOCAL, halo_margin=margin) as mw:
        mw.set_polytopes(capi.scene_polytopes(kind))
        mw.upload(bodies, sid, 0, n)
        mw.step(DT, substeps)                                            # 0.33 m: within the margin
        after_one = mw.download()
        with pytest.raises(capi.XpbdError) as e:                         # 0.66 m since the plan: this frame is undone
            mw.step(DT, substeps)
        assert e.value.code == capi.E_HALO and "halo_margin" in str(e.value) and "undone" in str(e.value)
        assert mw.halo_stats()["max_displacement"] > margin and mw.plan_stats()["rollbacks"] == 1
        with pytest.raises(capi.XpbdError) as e:                         # sticky until the halos are re-planned
            mw.step(DT, substeps)
        assert e.value.code == capi.E_HALO
        assert bits_equal(mw.download(), after_one)                      # the state of the failed frame's start
        mw.replan()
        mw.step(DT, substeps)                                            # the same frame again, from a fresh plan
        mw.replan()                                                      # (0.33 m per frame: a plan holds for one frame)
        mw.step(DT, substeps)
        got = mw.download()
    one, _ = single(bodies, sid, kind, 3, substeps)
    assert bits_equal(got, one)
    assert bits_equal(after_one, single(bodies, sid, kind, 1, substeps)[0])
    # a body that leaves the margin within ONE frame cannot be stepped at all with this margin: error, state untouched
    bodies[fast, 22:25] = [0.0, 0.0, 40.0]                               # 0.67 m per frame
    for auto in (False, True):
        with capi.MultiWorld(2, devices=[0, 0], transport=capi.TRANSPORT_LOCAL, halo_margin=margin, auto_replan=auto) as mw:
            mw.set_polytopes(capi.scene_polytopes(kind))
            mw.upload(bodies, sid, 0, n)
            with pytest.raises(capi.XpbdError) as e:
                mw.step(DT, substeps)
            assert e.value.code == capi.E_HALO
            assert bits_equal(mw.download(), bodies)
    # with a margin that holds for one frame and automatic re-planning the same scene runs, and equals the single device
    got, _, s1, _ = sharded(bodies, sid, kind, 2, 12, substeps, margin=1.5, auto_replan=True)
    assert s1["plans"] > 2
    one, _ = single(bodies, sid, kind, 12, substeps)
    assert bits_equal(got, one)


def test_automatic_replanning_undoes_and_reruns_a_frame_that_outran_its_halos():
    """A body that accelerates from rest: 0.19 m in the first frame (no pre-emptive re-plan: two and a half times that
    still fits the 0.5 m margin), 0.49 m more in the second -- 0.68 m since the plan.  With XPBD_MULTI_AUTO_REPLAN the second
    frame is undone, the halos are re-planned from its start state and the frame runs again: the caller sees nothing but
    the single-device result."""
    kind, n, substeps, margin = capi.SCENE_BOXES_DROP, 64, 4, 0.5
    bodies, sid = line_scene(capi, kind, n, 3, 1.5)
    fast = boundary_body(kind, bodies, sid, 2, margin)
    bodies[fast, 22:25] = 0.0
    bodies[fast, 10:13] = [0.0, 0.0, 1080.0 / bodies[fast, 0]]          # 1080 m/s^2 upwards: +18 m/s per frame
    with capi.MultiWorld(2, devices=[0, 0], transport=capi.TRANSPORT_LOCAL, halo_margin=margin, auto_replan=True) as mw:
        mw.set_polytopes(capi.scene_polytopes(kind))
        mw.upload(bodies, sid, 0, n)
        mw.step(DT, substeps)
        assert mw.plan_stats()["rollbacks"] == 0 and mw.halo_stats()["plans"] == 1
        mw.step(DT, substeps)
        stats = mw.plan_stats()
        assert stats["rollbacks"] == 1 and stats["plans"] >= 2
        got = mw.download()
    one, _ = single(bodies, sid, kind, 2, substeps)
    assert bits_equal(got, one)


@pytest.mark.parametrize("n_ranks", [2, 4])
def test_the_library_cuts_the_shards_whatever_the_callers_numbering(n_ranks):
    """The same pile numbered at random and numbered row by row: both sharded runs equal the single device over the same
    numbering, the shards are slabs of space in both (as few ghosts for the shuffled scene as for the ordered one) and
    near-equal in size; the re-plans on the way re-balance them."""
    kind, n, substeps, frames = capi.SCENE_BOXES_DROP, 1200, 8, 30
    bodies, sid = capi.scene_pile(kind, 5, n, 1.8, 3)
    joints = chain_joints(capi, n, every=7)
    centre = bodies[:, 31:34] + bodies[:, 28:31]
    joints["distance"] = np.linalg.norm(centre[joints["body_b"]] - centre[joints["body_a"]], axis=1)
    perm = np.random.default_rng(9).permutation(n)                       # caller's index k holds pile body perm[k]
    inverse = np.empty_like(perm)
    inverse[perm] = np.arange(n)
    shuffled_joints = joints.copy()
    shuffled_joints["body_a"], shuffled_joints["body_b"] = inverse[joints["body_a"]], inverse[joints["body_b"]]
    results = {}
    for name, (b, s, j) in {"ordered": (bodies, sid, joints), "shuffled": (bodies[perm], sid[perm], shuffled_joints)}.items():
        with capi.MultiWorld(n_ranks, devices=[0] * n_ranks, transport=capi.TRANSPORT_LOCAL, halo_margin=1.0, auto_replan=True) as mw:
            mw.set_polytopes(capi.scene_polytopes(kind))
            mw.upload(b, s, 0, n, j)
            halo0, plan0, owner0 = mw.halo_stats(), mw.plan_stats(), mw.owners()
            for _ in range(frames):
                mw.step(DT, substeps)
            got, plan1 = mw.download(), mw.plan_stats()
            ids, owned = mw.download_owned()
        one, one_stats = single(b, s, kind, frames, substeps, joints=j)
        assert one_stats[1] > 100 and bits_equal(got, one)
        assert np.array_equal(np.sort(ids), np.arange(n)) and bits_equal(owned, one[ids])
        assert plan0["owned_max"] - plan0["owned_min"] <= n // n_ranks // 4 and plan1["owned_max"] - plan1["owned_min"] <= n // n_ranks // 4
        assert plan1["plans"] > 1
        results[name] = (halo0["ghosts"], owner0)
    g_ordered, g_shuffled = results["ordered"][0], results["shuffled"][0]
    assert 0 < g_shuffled <= 1.1 * g_ordered + 8                        # the caller's numbering does not thicken the halos
    # the same body has the same owner under both numberings (up to the few bodies of a cell split between two ranks)
    assert np.mean(results["shuffled"][1] == results["ordered"][1][perm]) > 0.97


def test_light_plans_keep_the_cuts_and_equal_the_full_planner(monkeypatch):
    """After the first plan a re-plan keeps the cuts and exchanges only the RIMS of the shards (bodies near a cut, bodies that
    change owner, ends of joints that leave their holder).  With XPBD_MULTI_CHECK_PLANS the library also gathers the keys of
    the whole world at every light plan and fails unless the full planner arrives at the same lists (owned, ghosts, boundary,
    far) -- here for a pile with joints across the cuts, three shards, a re-plan every few frames; and the result equals the single
    world and a world that makes full plans only (XPBD_MULTI_FULL_PLANS)."""
    monkeypatch.setenv("XPBD_MULTI_CHECK_PLANS", "1")
    kind, n, substeps, frames = capi.SCENE_BOXES_DROP, 1200, 8, 30
    bodies, sid = capi.scene_pile(kind, 5, n, 1.8, 3)
    joints = chain_joints(capi, n, every=7)
    centre = bodies[:, 31:34] + bodies[:, 28:31]
    joints["distance"] = np.linalg.norm(centre[joints["body_b"]] - centre[joints["body_a"]], axis=1)
    results = {}
    for full in (False, True):
        with capi.MultiWorld(3, devices=[0] * 3, transport=capi.TRANSPORT_LOCAL, halo_margin=1.0, auto_replan=True, full_plans=full,
                             plan_through_device=not full) as mw:
            mw.set_polytopes(capi.scene_polytopes(kind))
            mw.upload(bodies, sid, 0, n, joints)
            for f in range(frames):
                if f % 3 == 1:
                    mw.replan()
                mw.step(DT, substeps)
            stats = mw.plan_stats()
            owners = mw.owners()
            results[full] = mw.download()
        assert stats["plans"] == stats["full_plans"] + stats["light_plans"] and stats["plans"] >= 11
        assert (stats["light_plans"] == 0) if full else (stats["full_plans"] >= 1 and stats["light_plans"] >= 8)
        assert np.bincount(owners, minlength=3).min() == stats["owned_min"] and np.bincount(owners, minlength=3).max() == stats["owned_max"]
    one, _ = single(bodies, sid, kind, frames, substeps, joints)
    assert bits_equal(results[False], one) and bits_equal(results[True], one)


def test_long_joints_between_bodies_far_from_any_cut(monkeypatch):
    """Joints between bodies at opposite ends of the world: neither end is anywhere near a cut, both are mirrored by the other
    end's owner all the same (a light plan learns about them because their holders publish the ends of joints that leave them).
    Four shards, re-plans every other frame, checked against the full planner and the single world."""
    monkeypatch.setenv("XPBD_MULTI_CHECK_PLANS", "1")
    kind, n, substeps, frames = capi.SCENE_BOXES_DROP, 160, 6, 24
    bodies, sid = line_scene(capi, kind, n, 17, 1.4)
    order = np.argsort(bodies[:, 31], kind="stable")                      # along the line
    ends = 24
    joints = np.zeros(ends, dtype=capi.JOINT_DTYPE)
    joints["body_a"], joints["body_b"] = order[:ends], order[::-1][:ends]
    joints["anchor_a"], joints["anchor_b"] = [0.5, 0.5, 0.5], [0.5, 0.5, 0.5]
    centre = bodies[:, 31:34] + bodies[:, 28:31]
    joints["distance"] = np.linalg.norm(centre[joints["body_b"]] - centre[joints["body_a"]], axis=1)
    with capi.MultiWorld(4, devices=[0] * 4, transport=capi.TRANSPORT_LOCAL, halo_margin=0.75, auto_replan=True, plan_through_device=True) as mw:
        mw.set_polytopes(capi.scene_polytopes(kind))
        mw.upload(bodies, sid, 0, n, joints)
        halo = mw.halo_stats()
        for f in range(frames):
            if f % 2:
                mw.replan()
            mw.step(DT, substeps)
        stats = mw.plan_stats()
        got = mw.download()
    assert halo["ghosts"] >= 2 * ends and stats["light_plans"] >= frames // 2
    one, _ = single(bodies, sid, kind, frames, substeps, joints)
    assert bits_equal(got, one)


def test_a_drifting_world_gets_new_cuts_when_the_shards_are_out_of_balance():
    """All bodies drift along the axis the shards are cut across: with the cuts kept, one shard would fill up and the other
    empty.  A light plan that finds a shard a tenth of a share off balance gives way to a full plan (new cuts)."""
    kind, n, substeps, frames = capi.SCENE_BOXES_DROP, 256, 4, 90
    bodies, sid = line_scene(capi, kind, n, 5, 1.3)
    bodies[:, 22] += 12.0                                                 # 0.2 m per frame along the line: 18 m in 90 frames, 28 bodies across the cut
    bodies[:, 10:13] = 0.0
    with capi.MultiWorld(2, devices=[0, 0], transport=capi.TRANSPORT_LOCAL, halo_margin=0.5, auto_replan=True) as mw:
        mw.set_polytopes(capi.scene_polytopes(kind))
        mw.upload(bodies, sid, 0, n)
        for _ in range(frames):
            mw.step(DT, substeps)
        stats = mw.plan_stats()
        got = mw.download()
    assert stats["light_plans"] >= 3 and stats["full_plans"] >= 2 and stats["rollbacks"] == 0
    assert stats["owned_max"] - stats["owned_min"] <= n // 2 // 10 + 16    # never far out of balance
    one, _ = single(bodies, sid, kind, frames, substeps)
    assert bits_equal(got, one)


def test_sharded_world_with_hinges_across_shards_and_the_depenetration_limit():
    """XPBD_JOINT_HINGE joints whose bodies live on different shards and xpbd_multi_world_set_max_depenetration_speed: the
    sharded world equals the single one bit for bit (the angular term reads the partner's rotation from its ghost)."""
    kind, n, substeps, frames = capi.SCENE_BOXES_DROP, 96, 6, 25
    bodies, sid = line_scene(capi, kind, n, 11, 1.3)
    bodies[:, 34:38] = [1.0, 0.0, 0.0, 0.0]                              # upright, so that the hinges' anchor points coincide at t = 0
    joints = chain_joints(capi, n, every=1, distance=1.3, limit=n // 2)
    with capi.MultiWorld(3, devices=[0] * 3, transport=capi.TRANSPORT_LOCAL, halo_margin=0.75) as mw:   # who will own what?
        mw.set_polytopes(capi.scene_polytopes(kind))
        mw.upload(bodies, sid, 0, n)
        owner = mw.owners()
    hinges = (np.arange(len(joints)) % 4 == 3) | (owner[joints["body_a"]] != owner[joints["body_b"]])   # ... every joint across a cut
    joints["kind"][hinges], joints["distance"][hinges] = capi.JOINT_HINGE, 0.0
    joints["anchor_a"][hinges], joints["anchor_b"][hinges] = [1.15, 0.5, 0.5], [-0.15, 0.5, 0.5]
    joints["axis_a"][hinges] = joints["axis_b"][hinges] = [0.0, 1.0, 0.0]
    with capi.World(mode=capi.MODE_CONTACTS) as w:
        w.set_polytopes(capi.scene_polytopes(kind))
        w.set_max_depenetration_speed(2.0)
        w.upload(bodies, sid)
        w.set_joints(joints)
        for _ in range(frames):
            w.step(DT, substeps)
        one = w.download()
    with capi.MultiWorld(3, devices=[0] * 3, transport=capi.TRANSPORT_LOCAL, halo_margin=0.75, auto_replan=True) as mw:
        mw.set_polytopes(capi.scene_polytopes(kind))
        mw.set_max_depenetration_speed(2.0)
        mw.upload(bodies, sid, 0, n, joints)
        owner = mw.owners()
        for _ in range(frames):
            mw.step(DT, substeps)
        got = mw.download()
    crossing = owner[joints["body_a"]] != owner[joints["body_b"]]
    assert (crossing & hinges).any()
    assert not np.isnan(one).any() and bits_equal(got, one)


def test_rccl_transport_with_a_one_rank_communicator():
    """RCCL refuses two ranks on one device, so the collective path is exercised with one rank: unique id, communicator,
    the per-frame displacement all-gather and the per-substep halo all-gather run through ncclAllGather."""
    kind, n, substeps, frames = capi.SCENE_BOXES_DROP, 80, 5, 10
    bodies, sid = line_scene(capi, kind, n, 2, 1.3)
    cid = capi.comm_unique_id()
    assert len(cid) == capi.COMM_ID_BYTES and any(cid)
    with capi.MultiWorld(1, devices=[0], transport=capi.TRANSPORT_RCCL, comm_id=cid, plan_through_device=True) as mw:
        mw.set_polytopes(capi.scene_polytopes(kind))
        mw.upload(bodies, sid, 0, n)                        # the plan's all-gathers go through ncclAllGather as well
        for _ in range(frames):
            mw.step(DT, substeps)
        mw.replan()
        mw.step(DT, substeps)
        got, stats = mw.download(), mw.halo_stats()
    frames += 1
    assert stats["ghosts"] == 0 and stats["owned"] == n
    one, _ = single(bodies, sid, kind, frames, substeps)
    assert bits_equal(got, one)


def test_multi_world_argument_errors():
    kind = capi.SCENE_BOXES_DROP
    bodies, sid = line_scene(capi, kind, 10, 1, 1.5)
    with pytest.raises(capi.XpbdError):
        capi.MultiWorld(2, devices=[0], transport=capi.TRANSPORT_LOCAL)          # LOCAL needs every rank in the process
    with pytest.raises(capi.XpbdError):
        capi.MultiWorld(2, devices=[0, 0], transport=capi.TRANSPORT_RCCL)        # RCCL needs a communicator id
    with capi.MultiWorld(2, devices=[0, 0], transport=capi.TRANSPORT_LOCAL) as mw:
        with pytest.raises(capi.XpbdError):
            mw.upload(bodies, sid, 0, 10)                                        # shapes not set
        mw.set_polytopes(capi.scene_polytopes(kind))
        with pytest.raises(capi.XpbdError):
            mw.upload(bodies[:5], sid[:5], 0, 10)                                # this process owns all 10 bodies
        with pytest.raises(capi.XpbdError):
            mw.step(DT, 4)                                                       # nothing uploaded
        mw.upload(bodies, sid, 0, 10)
        with pytest.raises(capi.XpbdError):
            mw.step(DT, 0)
        mw.step(DT, 4)
        assert mw.download().shape == (10, 38)


def test_rccl_world_beside_torch_distributed_nccl(tmp_path):
    """bench.py --gpus N runs the native world inside a process that has torch.distributed's nccl (= RCCL) group up and
    hands the communicator id around with broadcast_object_list.  The same here with a one-rank group (one GPU), in a
    child process: the library must bind to the RCCL copy the process already carries and work beside it."""
    import subprocess
    import sys
    code = r'''
import os, sys, json
sys.path[:0] = [%r, %r]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", RANK="0", WORLD_SIZE="1")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from constraint_solver_amd import capi
from halo_common import line_scene
cid = [capi.comm_unique_id()]
dist.broadcast_object_list(cid, src=0)
bodies, sid = line_scene(capi, capi.SCENE_BOXES_DROP, 64, 2, 1.3)
with capi.MultiWorld(1, devices=[0], transport=capi.TRANSPORT_RCCL, comm_id=cid[0]) as mw:
    mw.set_polytopes(capi.scene_polytopes(capi.SCENE_BOXES_DROP))
    mw.upload(bodies, sid, 0, 64)
    for _ in range(5):
        mw.step(1 / 60, 4)
    got = mw.download()
t = torch.ones(4, device="cuda")
dist.all_reduce(t)
with capi.World(mode=capi.MODE_CONTACTS) as w:
    w.set_polytopes(capi.scene_polytopes(capi.SCENE_BOXES_DROP))
    w.upload(bodies, sid)
    for _ in range(5):
        w.step(1 / 60, 4)
    one = w.download()
print(json.dumps({"same": bool(np.array_equal(got.view(np.uint64), one.view(np.uint64))), "lib": capi.comm_library(), "sum": float(t.sum())}))
dist.destroy_process_group()
''' % (str(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))),
       str(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    script = tmp_path / "child.py"
    script.write_text(code)
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["same"] and res["sum"] == 4.0 and res["lib"]
    print("RCCL bound from", res["lib"])


def test_multi_world_edge_cases():
    """More ranks than bodies (empty shards), a one-body world, a world without any contact, re-uploading a different scene
    into the same multi world, and a NaN position (rejected at upload)."""
    kind = capi.SCENE_BOXES_DROP
    bodies, sid = line_scene(capi, kind, 3, 5, 1.3)
    with capi.MultiWorld(4, devices=[0] * 4, transport=capi.TRANSPORT_LOCAL, auto_replan=True) as mw:
        mw.set_polytopes(capi.scene_polytopes(kind))
        mw.upload(bodies, sid, 0, 3)                                   # fewer bodies than ranks: some shards stay empty
        for _ in range(20):
            mw.step(DT, 5)
        got = mw.download()
        one, _ = single(bodies, sid, kind, 20, 5)
        assert bits_equal(got, one)
        mw.upload(bodies[:1], sid[:1], 0, 1)                           # a one-body world
        mw.step(DT, 5)
        one, _ = single(bodies[:1], sid[:1], kind, 1, 5)
        assert bits_equal(mw.download(), one)
        big, big_sid = line_scene(capi, kind, 90, 6, 3.0)              # 3 m pitch: nobody ever touches anybody
        mw.upload(big, big_sid, 0, 90)
        for _ in range(10):
            mw.step(DT, 5)
        one, stats = single(big, big_sid, kind, 10, 5)
        assert stats[1] == 0 and bits_equal(mw.download(), one)
        verts, off = capi.scene_shapes(kind)                           # ... which is the reference path itself
        want = big
        for _ in range(10):
            want, _ = ob.step_bodies(want, big_sid, verts, off, DT, 5)
        assert bits_equal(one, want)
        bad = big.copy()
        bad[7, 31] = np.nan
        with pytest.raises(capi.XpbdError) as e:                       # a body without a position cannot be given to a shard
            mw.upload(bad, big_sid, 0, 90)
        assert e.value.code == capi.E_INVALID and "non-finite" in str(e.value)
