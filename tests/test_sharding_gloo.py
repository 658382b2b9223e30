"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): each rank builds and steps only
its index shard (here with the CPU oracle standing in for the per-rank device), results are
gathered, and must equal the single-process run of the whole world bit for bit.  Also covers
the barrier + MAX-over-ranks timing reduction bench.py uses."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_binding as ob
from constraint_solver_amd import capi
from constraint_solver_amd.sharding import shard_range
from golden_util import bits_equal

N, SEED, SUBSTEPS, FRAMES = 301, 4, 20, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world_size, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from bench import reduce_max_seconds   # the reduction the benchmark uses
        verts, off = capi.scene_shapes(capi.SCENE_MIXED)
        first, count = shard_range(N, rank, world_size)
        bodies, sid = capi.scene_generate(capi.SCENE_MIXED, SEED, N, first=first, count=count)
        dist.barrier()
        for _ in range(FRAMES):
            bodies, _ = ob.step_bodies(bodies, sid, verts, off, 1 / 60, SUBSTEPS)
        slowest = reduce_max_seconds(0.25 * (rank + 1))
        assert slowest == pytest.approx(0.25 * world_size)
        # gather the shards on rank 0 (padding to the largest shard)
        cap = shard_range(N, 0, world_size)[1]
        mine = torch.zeros(cap, 38, dtype=torch.float64)
        mine[:count] = torch.from_numpy(bodies)
        parts = [torch.zeros_like(mine) for _ in range(world_size)] if rank == 0 else None
        dist.gather(mine, parts, dst=0)
        if rank == 0:
            whole = np.concatenate([parts[r][: shard_range(N, r, world_size)[1]].numpy() for r in range(world_size)])
            np.save(os.path.join(out_dir, "gathered.npy"), whole)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_run_equals_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    gathered = np.load(tmp_path / "gathered.npy")
    verts, off = capi.scene_shapes(capi.SCENE_MIXED)
    bodies, sid = capi.scene_generate(capi.SCENE_MIXED, SEED, N)
    for _ in range(FRAMES):
        bodies, _ = ob.step_bodies(bodies, sid, verts, off, 1 / 60, SUBSTEPS)
    assert gathered.shape == bodies.shape and bits_equal(gathered, bodies)
