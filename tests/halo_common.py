"""Shared by the halo-exchange tests: scene, per-rank worker, single-process expectation."""
import os

import numpy as np

DT = 1.0 / 60.0
POLY_NAMES = {0: [("cube", 1.0)], 2: [("cube", 1.0)],
              1: [("cube", 1.0), ("tetrahedron", 0.5), ("icosahedron", 0.5)],
              3: [("cube", 1.0), ("tetrahedron", 0.5), ("icosahedron", 0.5)]}


DYN_FIELDS = 13
DYN_AOS_COLUMNS = np.array([31, 32, 33, 34, 35, 36, 37, 22, 23, 24, 25, 26, 27])


class OracleBackend:
    """CPU stand-in for distributed.GpuBackend (same interface) on oracle/xpbd_pairs_oracle.c."""

    def __init__(self, oracle_binding, poly_names, pad):
        import ctypes
        import torch
        self.torch, self.C, self.ob = torch, ctypes, oracle_binding
        self.L = oracle_binding._contacts_api()
        P = ctypes.POINTER
        self.L.op_contacts_begin.restype = ctypes.c_void_p
        self.L.op_contacts_begin.argtypes = [ctypes.c_void_p, P(ctypes.c_uint32), ctypes.c_uint32,
                                             P(oracle_binding.Polytope), ctypes.c_double, ctypes.c_double]
        self.L.op_contacts_substep.restype = None
        self.L.op_contacts_substep.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
        self.L.op_contacts_end.restype = None
        self.L.op_contacts_end.argtypes = [ctypes.c_void_p]
        self.polys = oracle_binding.polytopes_array(poly_names)
        self.pad = pad
        self.frame = None
        self.joints = None

    def upload(self, bodies, shape_id, joints=None):
        self.bodies = np.array(bodies, dtype=np.float64).reshape(-1, 38).copy()
        self.sid = np.ascontiguousarray(shape_id, dtype=np.uint32)
        self.joints = None if joints is None or not len(joints) else np.ascontiguousarray(joints)

    def begin(self, dt):
        self._end()
        self.frame = self.L.op_contacts_begin(self.bodies.ctypes.data, self.sid.ctypes.data_as(self.C.POINTER(self.C.c_uint32)),
                                              self.bodies.shape[0], self.polys, dt, self.pad)
        if self.joints is not None:
            self.L.op_contacts_attach_joints.restype = None
            self.L.op_contacts_attach_joints.argtypes = [self.C.c_void_p, self.C.c_void_p, self.C.c_uint32]
            self.L.op_contacts_attach_joints(self.frame, self.joints.ctypes.data, self.joints.size)

    def substep(self, h):
        self.L.op_contacts_substep(self.frame, self.bodies.ctypes.data, h, None, None)

    def index_tensor(self, slots):
        return self.torch.as_tensor(np.asarray(slots, dtype=np.int64))

    def export(self, idx, out):
        out.copy_(self.torch.from_numpy(self.bodies[idx.numpy()][:, DYN_AOS_COLUMNS]))

    def import_rows(self, idx, rows, buf):
        slots = idx.numpy()
        self.bodies[slots[:, None], DYN_AOS_COLUMNS[None, :]] = buf.numpy()[rows.numpy()]

    def empty(self, rows):
        return self.torch.empty((rows, DYN_FIELDS), dtype=self.torch.float64)

    def download(self):
        return self.bodies.copy()

    def _end(self):
        if self.frame:
            self.L.op_contacts_end(self.frame)
            self.frame = None

    def close(self):
        self._end()


def pile(capi, kind, n, seed, width, height):
    rng = np.random.default_rng(seed)
    bodies, sid = capi.scene_generate(kind, seed, n)
    bodies[:, 31:33] = rng.uniform(0, width, (n, 2))
    bodies[:, 33] = rng.uniform(0.5, height, n)
    bodies[:, 22:25] *= 0.3
    return bodies, sid


def line_scene(capi, kind, n, seed, pitch):
    """A long, two-row grid of dropped bodies at `pitch` metres: no initial overlaps, bodies land, tumble and
    bump into their neighbours at a few m/s -- a scene that respects a sub-metre halo margin."""
    bodies, sid = capi.scene_generate(kind, seed, n, grid_w=max(n // 2, 1))
    bodies[:, 31:33] *= pitch / 2.0
    bodies[:, 22:25] *= 0.3
    return bodies, sid


def shape_tables(capi, kind):
    polys = capi.scene_polytopes(kind)
    radius = np.array([np.linalg.norm(p["vertices"] - p["centroid"], axis=1).max() for p in polys])
    centroid = np.array([p["centroid"] for p in polys])
    return polys, radius, centroid


def chain_joints(capi, n, every=3, distance=1.5, limit=None):
    """Distance joints between body k and k + every (so they cross shard boundaries), centre anchors;
    `limit` keeps both ends below that index (one row of line_scene)."""
    a = np.arange(0, (limit or n) - every, 2)
    j = np.zeros(len(a), dtype=capi.JOINT_DTYPE)
    j["body_a"], j["body_b"] = a, a + every
    j["anchor_a"], j["anchor_b"], j["distance"] = [0.5, 0.5, 0.5], [0.5, 0.5, 0.5], distance
    return j


def expected(ob, bodies, sid, kind, substeps, frames, pad, joints=None):
    polys = ob.polytopes_array(POLY_NAMES[kind])
    want = bodies
    for _ in range(frames):
        if joints is None:
            want, _, _ = ob.contacts_step(want, sid, polys, DT, substeps, pad)
        else:
            want = ob.contacts_step_joints(want, sid, polys, joints, DT, substeps, pad)
    return want


def worker(rank, world_size, port, out_dir, backend_name, kind, n, seed, width, substeps, frames, pad, replan_at,
           with_joints=False, order="index"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [here, os.path.dirname(here)]
    import torch.distributed as dist
    from constraint_solver_amd import capi
    from constraint_solver_amd.distributed import GpuBackend, ShardedContactWorld
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        bodies, sid = line_scene(capi, kind, n, seed, -width) if width < 0 else pile(capi, kind, n, seed, width, 6.0)
        polys, radius, centroid = shape_tables(capi, kind)
        if backend_name == "gpu":
            backend = GpuBackend(capi, polys, pad, device=0)     # rehearsal: every rank on the one GPU of the box
        else:
            import oracle_binding as ob
            backend = OracleBackend(ob, POLY_NAMES[kind], pad)
        # with_joints: False, True (the default chain) or the keyword arguments of chain_joints
        joints = chain_joints(capi, n, **(with_joints if isinstance(with_joints, dict) else {})) if with_joints else None
        # (the random piles fling bodies at > 100 m/s out of their initial overlaps: their halos hold everybody anyway and the
        # margin check is off; the line scene respects its margin and keeps the check on)
        world = ShardedContactWorld(backend, rank, world_size, bodies, sid, radius, centroid, pad=pad, halo_margin=0.75,
                                    joints_global=joints, order=order, check_margin=width < 0, replan_every=1 if width < 0 else 0)
        assert sum(len(g) for g in world.plan.ghosts) > 0         # the case does have halos
        for f in range(frames):
            if f == replan_at:
                world.replan()
            world.step(DT, substeps)
        state = world.gather_global_in_caller_order()
        if rank == 0:
            np.save(os.path.join(out_dir, "sharded.npy"), state)
            np.save(os.path.join(out_dir, "perm.npy"), world.perm)
            np.save(os.path.join(out_dir, "ghosts.npy"), np.array([len(g) for g in world.plan.ghosts]))
        backend.close()
    finally:
        dist.destroy_process_group()
