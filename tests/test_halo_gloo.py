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
