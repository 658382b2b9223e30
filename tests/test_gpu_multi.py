"""The native multi-GPU world (xpbd_multi_world_*, csrc/xpbd_multi.cpp) on the one GPU of the box: several shards on
device 0 with the in-process transport (XPBD_TRANSPORT_LOCAL) must reproduce a single xpbd_world over the same bodies BIT FOR
BIT -- halo plan built by the library, one all-gather per substep, joints crossing shard boundaries, re-plans -- and the
oracle; a body that outruns halo_margin must be an error (XPBD_E_HALO), not a silently lost contact; the RCCL transport is
exercised with a one-rank communicator (RCCL refuses two ranks on one device).  EXTENSION: parity unpinned."""
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


def test_a_body_that_outruns_the_halo_margin_is_an_error_not_a_lost_contact():
    kind, n, substeps = capi.SCENE_BOXES_DROP, 64, 4
    bodies, sid = line_scene(capi, kind, n, 3, 1.5)
    bodies[5, 22:25] = [0.0, 0.0, 40.0]                                  # 40 m/s upwards: 0.67 m per frame against a 0.5 m margin
    with capi.MultiWorld(2, devices=[0, 0], transport=capi.TRANSPORT_LOCAL, halo_margin=0.5) as mw:
        mw.set_polytopes(capi.scene_polytopes(kind))
        mw.upload(bodies, sid, 0, n)
        mw.step(DT, substeps)                                            # nothing has moved yet when this frame is checked
        with pytest.raises(capi.XpbdError) as e:
            mw.step(DT, substeps)
        assert e.value.code == capi.E_HALO and "halo_margin" in str(e.value)
        assert mw.halo_stats()["max_displacement"] > 0.5
        with pytest.raises(capi.XpbdError) as e:                         # sticky until the halos are re-planned
            mw.step(DT, substeps)
        assert e.value.code == capi.E_HALO
        mw.replan()
        mw.step(DT, substeps)
    # with automatic re-planning and a margin that holds for one frame the same scene runs, and equals the single device
    got, _, s1, _ = sharded(bodies, sid, kind, 2, 12, substeps, margin=1.5, auto_replan=True)
    assert s1["plans"] > 2
    one, _ = single(bodies, sid, kind, 12, substeps)
    assert bits_equal(got, one)


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
    into the same multi world, and NaN bodies (which count as having outrun every margin)."""
    kind = capi.SCENE_BOXES_DROP
    bodies, sid = line_scene(capi, kind, 3, 5, 1.3)
    with capi.MultiWorld(4, devices=[0] * 4, transport=capi.TRANSPORT_LOCAL, auto_replan=True) as mw:
        mw.set_polytopes(capi.scene_polytopes(kind))
        mw.upload(bodies, sid, 0, 3)                                   # ranks 0..2 own one body each, rank 3 none
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
        mw.upload(bad, big_sid, 0, 90)                                 # the plan itself copes (NaN cells are clamped) ...
        with pytest.raises(capi.XpbdError) as e:                       # ... but a NaN position has left every margin
            mw.step(DT, 5)
        assert e.value.code == capi.E_HALO
