"""The native sharded world (xpbd_multi_world_*, csrc/xpbd_multi.cpp) at BASELINE.json's sizes, on the one GPU of the box:
configs[3] = 262 144 stacked boxes and configs[4] = 262 144 boxes + 65 536 chain joints, as 2 and 4 local shards on device 0
with the in-process transport, must equal the single xpbd_world over the same bodies BIT FOR BIT -- the same property
tests/test_gpu_multi.py checks at 64-1 200 bodies -- with non-empty halos, near-equal shards and (joints scene) joints that
cross shard boundaries.  The bodies are handed over in the generator's own order and, for the stacks, also numbered at
random: the library cuts the shards from the spatial-hash cell order either way.  EXTENSION: parity unpinned; the RCCL
transport itself cannot run with more than one rank on a one-GPU box."""
import numpy as np
import pytest

from constraint_solver_amd import capi
from golden_util import bits_equal

pytestmark = pytest.mark.gpu

DT = 1.0 / 60.0
N = 262144


def single(bodies, sid, kind, frames, substeps, joints=None):
    with capi.World(mode=capi.MODE_CONTACTS) as w:
        w.set_polytopes(capi.scene_polytopes(kind))
        w.upload(bodies, sid)
        if joints is not None:
            w.set_joints(joints)
        for _ in range(frames):
            w.step(DT, substeps)
        return w.download(), w.contact_stats()


def sharded(bodies, sid, kind, n_ranks, frames, substeps, joints=None, margin=0.5):
    with capi.MultiWorld(n_ranks, devices=[0] * n_ranks, transport=capi.TRANSPORT_LOCAL, halo_margin=margin, auto_replan=True) as mw:
        mw.set_polytopes(capi.scene_polytopes(kind))
        mw.upload(bodies, sid, 0, len(bodies), joints)
        halo, plan, owner = mw.halo_stats(), mw.plan_stats(), mw.owners()
        for _ in range(frames):
            mw.step(DT, substeps)
        return mw.download(), halo, plan, mw.plan_stats(), owner


@pytest.fixture(scope="module")
def stacks():
    """configs[3]: 16 384 columns of 16 unit boxes on a 128 x 128 grid of columns, every pair of neighbours in contact."""
    kind = capi.SCENE_BOX_STACKS
    bodies, sid = capi.scene_generate(kind, 1, N, grid_w=128)
    frames, substeps = 4, 20
    one, stats = single(bodies, sid, kind, frames, substeps)
    assert stats[1] > 1000000 and not np.isnan(one).any()      # the 1 mm gaps close within the first frame
    return kind, bodies, sid, frames, substeps, one


@pytest.mark.parametrize("n_ranks", [2, 4])
def test_config4_sharded_stacks_at_full_size_equal_the_single_world(stacks, n_ranks):
    kind, bodies, sid, frames, substeps, one = stacks
    got, halo, plan, plan_end, owner = sharded(bodies, sid, kind, n_ranks, frames, substeps)
    assert bits_equal(got, one)
    assert halo["ghosts"] > 1000 and halo["boundary"] > 1000
    assert halo["ghosts"] < 0.2 * N                                     # slabs of space: a few columns of ghosts per cut
    assert plan["owned_max"] - plan["owned_min"] <= N // n_ranks // 4
    # whole columns: the 16 boxes of a stack share a grid column, hence a cell column, hence (up to a split cell) an owner
    assert np.mean(owner.reshape(-1, 16).min(axis=1) == owner.reshape(-1, 16).max(axis=1)) > 0.99


def test_config4_sharded_stacks_numbered_at_random_equal_the_single_world(stacks):
    kind, bodies, sid, frames, substeps, _ = stacks
    perm = np.random.default_rng(4).permutation(N)
    b, s = bodies[perm], sid[perm]
    one, _ = single(b, s, kind, frames, substeps)
    got, halo, plan, _, owner = sharded(b, s, kind, 4, frames, substeps)
    assert bits_equal(got, one)
    _, halo_ordered, _, _, owner_ordered = sharded(bodies, sid, kind, 4, 0, substeps)
    assert 0 < halo["ghosts"] <= 1.05 * halo_ordered["ghosts"] + 64     # the caller's numbering does not thicken the halos
    assert np.mean(owner == owner_ordered[perm]) > 0.99


@pytest.mark.parametrize("n_ranks", [2, 4])
def test_config5_sharded_boxes_with_chain_joints_at_full_size_equal_the_single_world(n_ranks):
    """configs[4]: 262 144 dropped boxes on the 2 m grid, chains of 5 along x linked by 4 distance joints each (65 536 joints
    less the ones that would wrap around a grid row); the shards are slabs along x, so thousands of chains cross a cut."""
    kind = capi.SCENE_BOXES_DROP
    n_joints, frames, substeps = 65536, 30, 20
    bodies, sid = capi.scene_generate(kind, 1, N)
    k = np.arange(n_joints)
    a = (k // 4) * 5 + (k % 4)
    w = capi.default_grid_width(N)
    a = a[a // w == (a + 1) // w]
    joints = np.zeros(len(a), dtype=capi.JOINT_DTYPE)
    joints["body_a"], joints["body_b"] = a, a + 1
    joints["anchor_a"], joints["anchor_b"], joints["distance"] = [0.5, 0.5, 0.5], [0.5, 0.5, 0.5], 2.0
    one, stats = single(bodies, sid, kind, frames, substeps, joints=joints)
    assert not np.isnan(one).any()
    got, halo, plan, plan_end, owner = sharded(bodies, sid, kind, n_ranks, frames, substeps, joints=joints)
    assert bits_equal(got, one)
    crossing = int((owner[joints["body_a"]] != owner[joints["body_b"]]).sum())
    assert crossing > 100 and halo["ghosts"] > 1000
    assert plan["owned_max"] - plan["owned_min"] <= N // n_ranks // 4
