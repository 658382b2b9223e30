"""Multi-GPU form of the body-body contact EXTENSION: one process per GPU, halo bodies exchanged
after every substep with one all-gather (RCCL over xGMI when the backend is nccl).

Not part of the reference (which has no body-body contacts and no multi-device path); parity for
it is "sharded run == single-device run, bit for bit", checked by tests/test_halo_gloo.py (CPU, gloo,
with tests/halo_common.OracleBackend standing in for the device) and tests/test_gpu_pairs.py (GpuBackend).

Scheme
------
* Ownership: order="library" (what the native world does, csrc/xpbd_multi.cpp): the x-major sequence of
  spatial-hash cells is cut into world_size runs of near-equal body count (partition_owner), so every rank
  owns a slab of space (across the world's longest axis) whatever the caller's numbering, and a re-plan re-balances the slabs; bodies keep
  the caller's numbering and results equal a single-device run over the caller's bodies, bit for bit.
  Older forms, kept: contiguous index ranges (sharding.shard_range) of the caller's body order
  (order="index") or of the bodies renumbered by cell (order="spatial": results equal a single-device
  run over the renumbered bodies).
* Every rank's world holds OWNED bodies plus GHOST copies of remote bodies that can reach an
  owned body before the next re-plan; the local order is ascending global id, so every
  neighbour list and every floating-point sum has the same order as on a single device.
* A substep runs unchanged on owned + ghost bodies: a ghost's integrate + ground-contact stage
  depends only on its own state, so it is recomputed locally (no exchange needed before the
  SAT / pair solve); only the ghost's END-of-substep state is wrong locally (its neighbour set
  is incomplete), and that is overwritten with the owner's result:
      export owned boundary bodies (13 doubles each) -> all_gather -> import into the ghosts.
* The halo is static between re-plans: `halo_margin` is how far any body may travel before
  `replan()` must be called (`replan_every=k` does it every k frames; it re-gathers the whole state
  on the host, O(N), not on the per-substep path).  A body that outruns the margin -- e.g. one
  flung out of a deep initial overlap at 100 m/s -- could miss contacts with remote bodies, so
  every step() first reduces, over all ranks, the largest distance any owned body has travelled
  since the plan and raises HaloMarginExceeded beyond the margin (check_margin=False skips it).

This Python loop is round 1's orchestration, kept as the CPU (gloo) test vehicle; the product path is the
native multi-GPU world behind the C ABI (csrc/xpbd_multi.cpp, capi.MultiWorld), which does the same inside
the library with one ncclAllGather per substep.
"""
import numpy as np

from .sharding import shard_range

DYN_FIELDS = 13
# AoS (xpbd_rigid) columns of the 13 dynamic doubles in SoA field order: position, rotation, velocity, angular velocity
DYN_AOS_COLUMNS = np.array([31, 32, 33, 34, 35, 36, 37, 22, 23, 24, 25, 26, 27])


def bounding_spheres(bodies, shape_id, shape_radius, shape_centroid):
    """Conservative world-space bounding spheres: centre = position + com, radius = r_shape + |centroid - com|."""
    bodies = np.asarray(bodies)
    com = bodies[:, 28:31]
    centre = bodies[:, 31:34] + com
    sid = np.asarray(shape_id, dtype=np.int64)
    radius = np.asarray(shape_radius)[sid] + np.linalg.norm(np.asarray(shape_centroid)[sid] - com, axis=1)
    return centre, radius


def spatial_order(bodies, shape_id, shape_radius, shape_centroid, pad, halo_margin):
    """Permutation that sorts the bodies by spatial-hash cell, x-major (then y, z, then index): contiguous index
    ranges of the sorted bodies are slabs of space.  Same cell edge as HaloPlan."""
    centre, radius = bounding_spheres(bodies, shape_id, shape_radius, shape_centroid)
    edge = 2.0 * (float(radius.max()) + pad + halo_margin) if len(bodies) else 1.0
    cell = np.floor(centre / edge).astype(np.int64)
    return np.lexsort((np.arange(len(bodies)), cell[:, 2], cell[:, 1], cell[:, 0]))


def cell_keys(centre, edge):
    """Spatial-hash cell key of every centre (x-major: ascending keys are slabs along x), as xpbd_halo_cell_key."""
    lim = (1 << 20) - 4
    cell = np.clip(np.nan_to_num(np.floor(np.asarray(centre) / edge), nan=-lim, posinf=lim, neginf=-lim), -lim, lim).astype(np.int64)
    return HaloPlan._key(cell)


def partition_owner(key, world_size):
    """owner[g] from the cell keys of ALL bodies, as xpbd_halo_partition (csrc/xpbd_multi.cpp partition_cuts): the bodies
    sorted by (key, index) are cut into world_size runs at the positions of sharding.shard_range, each cut moved to the
    nearer boundary of the cell it falls into -- unless that leaves more than a quarter of a share on the wrong side, in
    which case the cell is split by body index."""
    key = np.asarray(key, dtype=np.int64)
    n = len(key)
    owner = np.zeros(n, dtype=np.uint8)
    if n == 0 or world_size < 2:
        return owner
    # slabs across the LONGEST axis of the world's box of cells: re-pack the keys with the axes by falling extent (ties: x, y, z)
    bias, mask = 1 << 20, (1 << 21) - 1
    cell = np.stack([(key >> 42) - bias, ((key >> 21) & mask) - bias, (key & mask) - bias], axis=1)
    extent = cell.max(axis=0) - cell.min(axis=0)
    axes = sorted(range(3), key=lambda a: -int(extent[a]))          # (stable)
    key = HaloPlan._key(cell[:, axes])
    order = np.lexsort((np.arange(n), key))
    sk = key[order]
    share = max(1, n // world_size)
    big = (np.iinfo(np.int64).max, 2 ** 32 - 1)
    cuts = [(np.iinfo(np.int64).min, 0)]
    for r in range(1, world_size):
        t = shard_range(n, r, world_size)[0]
        if t >= n:
            cut = big
        else:
            K = int(sk[t])
            less, leq = int(np.searchsorted(sk, K, "left")), int(np.searchsorted(sk, K, "right"))
            before, after = t - less, leq - t
            if min(before, after) * 4 <= share:
                cut = (K, 0) if before <= after else (K + 1, 0)
            else:
                cut = (K, int(order[t]))
        cuts.append(max(cut, cuts[-1]))
    pairs = list(zip(key.tolist(), range(n)))
    for g, pair in enumerate(pairs):
        owner[g] = sum(1 for c in cuts if c <= pair) - 1
    return owner


class HaloMarginExceeded(RuntimeError):
    """A body travelled farther than halo_margin since the halos were planned: remote contacts may have been missed."""


class HaloPlan:
    """Which remote bodies each rank mirrors, and where they sit in the all-gather buffer.
    Deterministic function of the global sphere table, so every rank computes the same plan."""

    def __init__(self, centre, radius, world_size, halo_margin, pad, joint_pairs=None, owner=None):
        """owner: None = contiguous index ranges; "library" = partition_owner of the cell keys; or an array."""
        n = centre.shape[0]
        self.n, self.world_size = n, world_size
        edge = 2.0 * (float(radius.max()) + pad + halo_margin) if n else 1.0
        cell = np.floor(centre / edge).astype(np.int64)
        key = self._key(cell)
        if owner is None:
            owner = np.zeros(n, dtype=np.int64)
            for r in range(world_size):
                first, count = shard_range(n, r, world_size)
                owner[first:first + count] = r
        elif isinstance(owner, str):
            owner = partition_owner(key, world_size).astype(np.int64)
        else:
            owner = np.asarray(owner, dtype=np.int64)
        self.own_ids = [np.nonzero(owner == r)[0] for r in range(world_size)]
        self.owned = [(int(ids[0]) if len(ids) else 0, len(ids)) for ids in self.own_ids]   # (first, count): ranges when owner is None
        offsets = np.array([(dx, dy, dz) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)], dtype=np.int64)
        self.local_ids, ghost_sets = [], []
        for r, own in enumerate(self.own_ids):
            own_cells = np.unique(cell[own], axis=0) if len(own) else np.zeros((0, 3), dtype=np.int64)
            reach = np.unique(self._key((own_cells[:, None, :] + offsets[None, :, :]).reshape(-1, 3)))
            remote = np.nonzero((owner != r) & np.isin(key, reach))[0]
            if joint_pairs is not None and len(joint_pairs):
                # a joint with an owned body needs its partner locally, however far away it is
                a, b = joint_pairs[:, 0], joint_pairs[:, 1]
                partners = np.concatenate([b[(owner[a] == r) & (owner[b] != r)], a[(owner[b] == r) & (owner[a] != r)]])
                remote = np.union1d(remote, partners)
            ghost_sets.append(remote)
            self.local_ids.append(np.sort(np.concatenate([own, remote])))
        needed = np.zeros(n, dtype=bool)
        for g in ghost_sets:
            needed[g] = True
        # boundary bodies of each rank = its owned bodies that some other rank mirrors, ascending
        self.boundary = [own[needed[own]] for own in self.own_ids]
        self.capacity = max([len(b) for b in self.boundary] + [1])
        self.ghosts = ghost_sets
        self._owner = owner

    @staticmethod
    def _key(cell):
        c = cell + (1 << 20)
        return (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]

    def rank_view(self, rank):
        """Index arrays of one rank: local ids (global id per local slot), owned mask, local slots of its
        boundary bodies, local slots of its ghosts and their rows in the gathered [W * capacity] buffer."""
        ids = self.local_ids[rank]
        owned_mask = self._owner[ids] == rank
        boundary_slots = np.searchsorted(ids, self.boundary[rank])
        ghost_ids = self.ghosts[rank]
        ghost_slots = np.searchsorted(ids, ghost_ids)
        owners = self._owner[ghost_ids]
        rows = np.empty(len(ghost_ids), dtype=np.int64)
        for o in np.unique(owners):                      # one vectorised search per owning rank
            mine = owners == o
            rows[mine] = o * self.capacity + np.searchsorted(self.boundary[o], ghost_ids[mine])
        return ids, owned_mask, boundary_slots, ghost_slots, rows


class GpuBackend:
    """capi.World in XPBD_MODE_CONTACTS; exchange buffers are torch CUDA tensors."""

    def __init__(self, capi, polytopes, pad, device=0):
        import torch
        self.torch, self.capi = torch, capi
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(device)
        self.stream = torch.cuda.Stream()          # non-default: handle 0 would mean "the world's own stream"
        torch.cuda.set_stream(self.stream)
        self.world = capi.World(device=device, mode=capi.MODE_CONTACTS)
        self.world.set_polytopes(polytopes)
        self.world.set_contact_pad(pad)
        self.world.set_stream(self.stream.cuda_stream)

    def upload(self, bodies, shape_id, joints=None):
        self.world.upload(bodies, shape_id)
        if joints is not None and len(joints):
            self.world.set_joints(joints)

    def begin(self, dt):
        self.world.contacts_begin(dt)

    def substep(self, h):
        self.world.contacts_substep(h)

    def index_tensor(self, slots):
        return self.torch.as_tensor(np.asarray(slots, dtype=np.int32), device=self.device)

    def export(self, idx, out):
        self.world.export_dynamic(idx.data_ptr(), idx.numel(), out.data_ptr())

    def import_rows(self, idx, rows, buf):
        """Local slots idx[k] take row rows[k] of buf (the gathered halo buffer, imported as it is)."""
        self.world.import_dynamic_rows(idx.data_ptr(), rows.data_ptr(), idx.numel(), buf.data_ptr())

    def empty(self, rows):
        return self.torch.empty((rows, DYN_FIELDS), dtype=self.torch.float64, device=self.device)

    def download(self):
        return self.world.download()

    def close(self):
        self.world.close()


class ShardedContactWorld:
    """One rank of an N-body world with body-body contacts sharded over `world_size` processes."""

    def __init__(self, backend, rank, world_size, bodies_global, shape_id_global, shape_radius, shape_centroid,
                 pad=0.02, halo_margin=0.5, group=None, joints_global=None, order="index", replan_every=0, check_margin=True):
        self.backend, self.rank, self.world_size, self.group = backend, rank, world_size, group
        self.check_margin = check_margin
        self.replan_every, self._frames = replan_every, 0   # > 0: re-select the halos every that many step() calls
        self.pad, self.halo_margin = pad, halo_margin
        self.shape_radius, self.shape_centroid = shape_radius, shape_centroid
        bodies_global = np.asarray(bodies_global, dtype=np.float64)
        shape_id_global = np.asarray(shape_id_global, dtype=np.uint32)
        # `perm[k]` = caller's index of the body that is number k internally
        self.library_owner = order == "library"         # ownership by partition_owner, re-cut at every plan; no renumbering
        self.perm = spatial_order(bodies_global, shape_id_global, shape_radius, shape_centroid, pad, halo_margin) \
            if order == "spatial" else np.arange(bodies_global.shape[0])
        inverse = np.empty_like(self.perm)
        inverse[self.perm] = np.arange(len(self.perm))
        self.shape_id_global = shape_id_global[self.perm]
        if joints_global is not None and len(joints_global):
            joints_global = joints_global.copy()    # records with body_a / body_b as ids (capi.JOINT_DTYPE)
            joints_global["body_a"] = inverse[joints_global["body_a"]]
            joints_global["body_b"] = inverse[joints_global["body_b"]]
        self.joints_global = joints_global
        self._plan(bodies_global[self.perm])

    def _plan(self, bodies_global):
        centre, radius = bounding_spheres(bodies_global, self.shape_id_global, self.shape_radius, self.shape_centroid)
        jg = self.joints_global
        pairs = None if jg is None or not len(jg) else np.stack([jg["body_a"], jg["body_b"]], axis=1).astype(np.int64)
        self.plan = HaloPlan(centre, radius, self.world_size, self.halo_margin, self.pad, joint_pairs=pairs,
                             owner="library" if self.library_owner else None)
        ids, self.owned_mask, boundary_slots, ghost_slots, rows = self.plan.rank_view(self.rank)
        self.local_ids = ids
        b = self.backend
        local_joints = None
        if pairs is not None:
            # joints whose two bodies are both present here, in global joint order, re-indexed to local slots
            present = np.isin(pairs[:, 0], ids) & np.isin(pairs[:, 1], ids)
            local_joints = jg[present].copy()
            local_joints["body_a"] = np.searchsorted(ids, pairs[present, 0])
            local_joints["body_b"] = np.searchsorted(ids, pairs[present, 1])
        b.upload(bodies_global[ids], self.shape_id_global[ids], local_joints)
        self.boundary_idx = b.index_tensor(boundary_slots)
        self.ghost_idx = b.index_tensor(ghost_slots)
        self.ghost_rows = b.index_tensor(rows)
        self.send = b.empty(self.plan.capacity)
        self.recv = b.empty(self.plan.capacity * self.world_size)
        self.n_boundary = len(boundary_slots)
        self._plan_positions = bodies_global[ids][self.owned_mask][:, 31:34].copy()    # of the owned bodies, at plan time
        if self.world_size > 1:
            import torch.distributed as dist
            self._on_gloo = dist.get_backend(self.group) == "gloo"

    def _exchange(self):
        if self.world_size == 1:
            return
        import torch.distributed as dist
        b = self.backend
        if self.n_boundary:
            b.export(self.boundary_idx, self.send[: self.n_boundary])
        # ONE collective per substep into a buffer allocated at plan time (RCCL over xGMI when the backend is nccl;
        # the collective is ordered after the export kernel and before the import kernel on the world's stream)
        if self._on_gloo and self.send.is_cuda:                      # single-GPU rehearsal: gloo moves host memory
            host = self.recv.cpu()
            dist.all_gather_into_tensor(host, self.send.cpu(), group=self.group)
            self.recv.copy_(host)
        else:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        if len(self.ghost_rows):
            b.import_rows(self.ghost_idx, self.ghost_rows, self.recv)

    def max_displacement(self):
        """Largest distance any owned body of ANY rank has travelled since the plan (a NaN position counts as inf)."""
        moved = np.linalg.norm(self.backend.download()[self.owned_mask][:, 31:34] - self._plan_positions, axis=1)
        worst = float(np.nan_to_num(moved, nan=np.inf).max()) if len(moved) else 0.0
        if self.world_size > 1:
            import torch
            import torch.distributed as dist
            t = torch.tensor([worst], dtype=torch.float64)
            if dist.get_backend(self.group) == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            worst = float(t.item())
        return worst

    def step(self, dt, substeps):
        """xpbd_world_step(dt, substeps) of the whole sharded world (lock step over ranks)."""
        if self.replan_every and self._frames and self._frames % self.replan_every == 0:
            self.replan()
        if self.check_margin and self.world_size > 1 and self._frames:
            moved = self.max_displacement()
            if not moved <= self.halo_margin:
                raise HaloMarginExceeded("a body has travelled %.3g m since the halos were planned, beyond halo_margin %.3g m: "
                                         "call replan() more often or raise the margin" % (moved, self.halo_margin))
        self._frames += 1
        h = dt / float(substeps)
        self.backend.begin(dt)
        for _ in range(substeps):
            self.backend.substep(h)
            self._exchange()

    def owned_state(self):
        """(global ids, (k, 38) states) of the bodies this rank owns."""
        state = self.backend.download()
        return self.local_ids[self.owned_mask], state[self.owned_mask]

    def gather_global(self):
        """Full (N, 38) state on every rank (host-side; for tests, check-pointing and replan)."""
        import torch
        import torch.distributed as dist
        ids, state = self.owned_state()
        if self.world_size == 1:
            return state
        cap = max(len(own) for own in self.plan.own_ids)
        mine = torch.zeros(cap, 38, dtype=torch.float64)
        mine[: len(ids)] = torch.from_numpy(state)
        on_nccl = dist.get_backend(self.group) == "nccl"
        if on_nccl:
            mine = mine.cuda()
        parts = [torch.zeros_like(mine) for _ in range(self.world_size)]
        dist.all_gather(parts, mine, group=self.group)
        out = np.empty((self.plan.n, 38))
        for r, own in enumerate(self.plan.own_ids):          # every rank knows every rank's owned ids: the plan is global
            out[own] = parts[r][: len(own)].cpu().numpy()
        return out

    def gather_global_in_caller_order(self):
        """gather_global() mapped back from the internal (possibly spatial) numbering to the caller's body order."""
        state = self.gather_global()
        out = np.empty_like(state)
        out[self.perm] = state
        return out

    def replan(self):
        """Re-select the halos from the current positions (call before any body has moved halo_margin).
        The numbering is kept; with order="library" the shards are re-cut (re-balanced) as well, else only the ghost sets change."""
        self._plan(self.gather_global())
