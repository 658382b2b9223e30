"""Halo exchange of the body-body contact extension, rehearsed on CPU: world_size 2 and 3 over gloo,
every rank stepping owned + ghost bodies with the CPU oracle and exchanging boundary bodies after
each substep.  The sharded result must equal the single-process op_contacts_step bit for bit."""
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import halo_common as hc
import oracle_binding as ob
from constraint_solver_amd import capi
from constraint_solver_amd.distributed import HaloPlan
from golden_util import bits_equal


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world_size,kind,n,replan_at,joints", [(2, capi.SCENE_BOXES_DROP, 90, -1, False),
                                                                (3, capi.SCENE_MIXED_DROP, 120, 3, False),
                                                                (2, capi.SCENE_BOXES_DROP, 80, 2, True)])
def test_sharded_oracle_run_equals_single_process(tmp_path, world_size, kind, n, replan_at, joints):
    seed, width, substeps, frames, pad = 8, 3.0, 8, 6, 0.02
    mp.spawn(hc.worker, args=(world_size, free_port(), str(tmp_path), "oracle", kind, n, seed, width, substeps, frames, pad,
                              replan_at, joints), nprocs=world_size, join=True)
    bodies, sid = hc.pile(capi, kind, n, seed, width, 6.0)
    want = hc.expected(ob, bodies, sid, kind, substeps, frames, pad, hc.chain_joints(capi, n) if joints else None)
    got = np.load(tmp_path / "sharded.npy")
    assert bits_equal(got, want)


def test_spatial_ordering_shrinks_halos_and_matches_the_renumbered_single_run(tmp_path):
    """order="spatial": bodies are renumbered by grid cell before the index-range split, so each rank owns a slab of
    space and mirrors only the bodies next to it.  The result equals the single-process run over the same renumbered
    bodies.  (A negative `width` selects halo_common.line_scene with that pitch: a scene whose speeds respect the
    halo margin -- the random piles of the other cases fling bodies at > 100 m/s out of their initial overlaps.)"""
    kind, n, seed, substeps, frames, pad, pitch = capi.SCENE_BOXES_DROP, 96, 3, 8, 12, 0.02, 1.15
    chain = dict(every=1, distance=pitch, limit=n // 2)     # links along the first row, at their rest length
    args = (2, free_port(), str(tmp_path), "oracle", kind, n, seed, -pitch, substeps, frames, pad, 6, chain, "spatial")
    mp.spawn(hc.worker, args=args, nprocs=2, join=True)
    got, perm, ghosts = np.load(tmp_path / "sharded.npy"), np.load(tmp_path / "perm.npy"), np.load(tmp_path / "ghosts.npy")
    bodies, sid = hc.line_scene(capi, kind, n, seed, pitch)
    joints = hc.chain_joints(capi, n, **chain)
    inverse = np.empty_like(perm)
    inverse[perm] = np.arange(n)
    jr = joints.copy()
    jr["body_a"], jr["body_b"] = inverse[joints["body_a"]], inverse[joints["body_b"]]
    want_internal = hc.expected(ob, bodies[perm], sid[perm], kind, substeps, frames, pad, jr)
    want = np.empty_like(want_internal)
    want[perm] = want_internal
    assert bits_equal(got, want)
    assert not np.array_equal(perm, np.arange(n))                    # the renumbering did happen
    assert ghosts.max() < n // 4                                     # thin halos: a few columns, not half the world
    assert np.abs(want[:, 22:25]).max() < 30.0                       # the scene stayed within the halo contract


def test_halo_plan_is_conservative_and_consistent():
    rng = np.random.default_rng(1)
    n, world_size = 400, 4
    centre = rng.uniform(0, 12, (n, 3))
    radius = rng.uniform(0.3, 0.9, n)
    plan = HaloPlan(centre, radius, world_size, halo_margin=0.25, pad=0.02)
    reach = radius[:, None] + radius[None, :] + 2 * (0.25 + 0.02)
    close = np.linalg.norm(centre[:, None, :] - centre[None, :, :], axis=2) < reach
    for r, (first, count) in enumerate(plan.owned):
        ids, owned_mask, boundary_slots, ghost_slots, rows = plan.rank_view(r)
        assert np.array_equal(ids, np.sort(ids)) and owned_mask.sum() == count
        local = set(ids.tolist())
        for i in range(first, first + count):                      # every body that can reach an owned one is present
            assert set(np.nonzero(close[i])[0].tolist()) <= local
        # the buffer rows of my ghosts point at the right global ids
        flat = np.full(world_size * plan.capacity, -1)
        for o in range(world_size):
            flat[o * plan.capacity: o * plan.capacity + len(plan.boundary[o])] = plan.boundary[o]
        assert np.array_equal(flat[rows], ids[ghost_slots])
        assert np.array_equal(ids[boundary_slots], plan.boundary[r])


def _outrun_worker(rank, world_size, port, out_dir):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from constraint_solver_amd.distributed import HaloMarginExceeded, ShardedContactWorld
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        kind, n = capi.SCENE_BOXES_DROP, 64
        bodies, sid = hc.line_scene(capi, kind, n, 3, 1.5)
        bodies[5, 22:25] = [0.0, 0.0, 40.0]                         # 0.67 m per frame against a 0.5 m margin
        polys, radius, centroid = hc.shape_tables(capi, kind)
        world = ShardedContactWorld(hc.OracleBackend(ob, hc.POLY_NAMES[kind], 0.02), rank, world_size, bodies, sid, radius, centroid,
                                    pad=0.02, halo_margin=0.5)
        world.step(hc.DT, 4)
        raised = False
        try:
            world.step(hc.DT, 4)
        except HaloMarginExceeded:
            raised = True                                           # on EVERY rank: the displacement is reduced over all of them
        world.replan()
        world.step(hc.DT, 4)
        np.save(os.path.join(out_dir, "raised%d.npy" % rank), np.array([raised]))
    finally:
        dist.destroy_process_group()


def test_python_halo_loop_detects_a_body_that_outruns_the_margin(tmp_path):
    mp.spawn(_outrun_worker, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)
    assert np.load(tmp_path / "raised0.npy")[0] and np.load(tmp_path / "raised1.npy")[0]


def _library_worker(rank, world_size, port, out_dir, n, seed, substeps, frames, pitch):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from constraint_solver_amd.distributed import ShardedContactWorld
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        kind = capi.SCENE_BOXES_DROP
        bodies, sid, joints = _shuffled_line(n, seed, pitch)
        polys, radius, centroid = hc.shape_tables(capi, kind)
        world = ShardedContactWorld(hc.OracleBackend(ob, hc.POLY_NAMES[kind], 0.02), rank, world_size, bodies, sid, radius, centroid,
                                    pad=0.02, halo_margin=0.75, joints_global=joints, order="library", replan_every=1)
        ghosts0 = [len(g) for g in world.plan.ghosts]
        owners0 = world.plan._owner.copy()
        for _ in range(frames):
            world.step(hc.DT, substeps)
        state = world.gather_global_in_caller_order()
        if rank == 0:
            np.save(os.path.join(out_dir, "sharded.npy"), state)
            np.save(os.path.join(out_dir, "ghosts.npy"), np.array(ghosts0))
            np.save(os.path.join(out_dir, "owners.npy"), np.stack([owners0, world.plan._owner]))
    finally:
        dist.destroy_process_group()


def _shuffled_line(n, seed, pitch):
    """halo_common.line_scene with its bodies numbered at random (and the chain joints re-indexed accordingly)."""
    kind = capi.SCENE_BOXES_DROP
    bodies, sid = hc.line_scene(capi, kind, n, seed, pitch)
    joints = hc.chain_joints(capi, n, every=1, distance=pitch, limit=n // 2)
    perm = np.random.default_rng(seed).permutation(n)              # caller's index k holds grid body perm[k]
    inverse = np.empty_like(perm)
    inverse[perm] = np.arange(n)
    joints["body_a"], joints["body_b"] = inverse[joints["body_a"]], inverse[joints["body_b"]]
    return bodies[perm], sid[perm], joints


@pytest.mark.parametrize("world_size", [2, 3])
def test_library_owned_partition_of_a_shuffled_scene_equals_the_single_run_in_the_callers_order(tmp_path, world_size):
    """order="library" (what xpbd_multi_world_upload does): the caller's numbering is random, the shards are cut from the
    spatial-hash cell order and re-cut at every re-plan; bodies keep their numbers, so the result is the single-process
    run over the caller's bodies, bit for bit -- no renumbering -- and the halos are as thin as for a pre-ordered scene."""
    n, seed, substeps, frames, pitch = 96, 3, 8, 12, 1.15
    mp.spawn(_library_worker, args=(world_size, free_port(), str(tmp_path), n, seed, substeps, frames, pitch), nprocs=world_size, join=True)
    bodies, sid, joints = _shuffled_line(n, seed, pitch)
    want = hc.expected(ob, bodies, sid, capi.SCENE_BOXES_DROP, substeps, frames, 0.02, joints)
    got, ghosts, owners = np.load(tmp_path / "sharded.npy"), np.load(tmp_path / "ghosts.npy"), np.load(tmp_path / "owners.npy")
    assert bits_equal(got, want)
    assert 0 < ghosts.max() < n // 4                                 # slabs of space although the indices are shuffled
    counts = np.bincount(owners[0], minlength=world_size)
    assert counts.max() - counts.min() <= n // world_size // 2       # near-equal shares
    assert np.abs(want[:, 22:25]).max() < 30.0                       # the scene stayed within the halo contract
