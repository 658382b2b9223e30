"""Shared by the halo-exchange tests: scene, per-rank worker, single-process expectation."""
import os

import numpy as np

DT = 1.0 / 60.0
POLY_NAMES = {0: [("cube", 1.0)], 2: [("cube", 1.0)],
              1: [("cube", 1.0), ("tetrahedron", 0.5), ("icosahedron", 0.5)],
              3: [("cube", 1.0), ("tetrahedron", 0.5), ("icosahedron", 0.5)]}


def pile(capi, kind, n, seed, width, height):
    rng = np.random.default_rng(seed)
    bodies, sid = capi.scene_generate(kind, seed, n)
    bodies[:, 31:33] = rng.uniform(0, width, (n, 2))
    bodies[:, 33] = rng.uniform(0.5, height, n)
    bodies[:, 22:25] *= 0.3
    return bodies, sid


def shape_tables(capi, kind):
    polys = capi.scene_polytopes(kind)
    radius = np.array([np.linalg.norm(p["vertices"] - p["centroid"], axis=1).max() for p in polys])
    centroid = np.array([p["centroid"] for p in polys])
    return polys, radius, centroid


def expected(ob, bodies, sid, kind, substeps, frames, pad):
    polys = ob.polytopes_array(POLY_NAMES[kind])
    want = bodies
    for _ in range(frames):
        want, _, _ = ob.contacts_step(want, sid, polys, DT, substeps, pad)
    return want


def worker(rank, world_size, port, out_dir, backend_name, kind, n, seed, width, substeps, frames, pad, replan_at):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [here, os.path.dirname(here)]
    import torch.distributed as dist
    from constraint_solver_amd import capi
    from constraint_solver_amd.distributed import GpuBackend, OracleBackend, ShardedContactWorld
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        bodies, sid = pile(capi, kind, n, seed, width, 6.0)
        polys, radius, centroid = shape_tables(capi, kind)
        if backend_name == "gpu":
            backend = GpuBackend(capi, polys, pad, device=0)     # rehearsal: every rank on the one GPU of the box
        else:
            import oracle_binding as ob
            backend = OracleBackend(ob, POLY_NAMES[kind], pad)
        world = ShardedContactWorld(backend, rank, world_size, bodies, sid, radius, centroid, pad=pad, halo_margin=0.75)
        assert sum(len(g) for g in world.plan.ghosts) > 0         # the case does have halos
        for f in range(frames):
            if f == replan_at:
                world.replan()
            world.step(DT, substeps)
        state = world.gather_global()
        if rank == 0:
            np.save(os.path.join(out_dir, "sharded.npy"), state)
        backend.close()
    finally:
        dist.destroy_process_group()
