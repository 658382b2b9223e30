#!/usr/bin/env python3
"""bench.py -- body·substeps/s of the XPBD stepper hot path on N MI355X GPUs of one node.

A "step" is one frame: xpbd_world_step(dt = 1/60, substeps) over the rank's resident bodies, i.e. for every
body `solver::step(body, shape, dt, substeps)` (reference src/solver.rs:3-17).  Workload at N = 1:
BASELINE.json's metric configuration, 262 144 rigid bodies (unit boxes) x 20 substeps/frame, bodies already
resident in HBM (SoA).

The timed state does not depend on --warmup / --steps: every scene is first PRE-ROLLED a fixed number of frames
(PREROLL) into its steady regime -- for `boxes-drop` the bodies have landed and rest / rock on the ground plane with
~3-4 ground contacts each, the expensive regime -- and only then come the W warm-up and the K timed frames.  The line
prints the ground contacts per body at the start and at the end of the timed region.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL).  On the pinned path bodies never interact
(reference semantics), so the world is sharded by contiguous body-index range with NO data-path collective; scaling
is weak (every rank steps --bodies bodies).  The `contacts` sub-results are the north-star workload (body-body
contacts: the EXTENSION, parity unpinned).

ONE JSON line on rank 0:
  roofline      the dominant kernel (k_step) at the algorithmic 412 B per body per launch against the
                8 TB/s HBM peak, plus `peak_measured` = a device-to-device copy by the library's own streaming kernel
                in this run, and `frac_of_measured`
  cpu_baseline  the CPU oracle (C restatement of the reference, single thread like the reference) on a bounded sample
                of the SAME pre-rolled state, median of 3 repeats
  contacts      {name: {...}} box stacks with SAT contacts (262 144 x 20) and 65 536 mixed polyhedra on the GJK/EPA
                path, each with its own value, roofline (whole-substep byte model) and cpu_baseline (op_contacts_step)
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json's metric, verbatim (the file travels with the repo; the constant is the fallback)
METRIC = "body·substeps/sec at 262k rigid bodies, 20 substeps/frame; 1/2/4/8 GPU"
try:
    with open(os.path.join(ROOT, "BASELINE.json")) as _f:
        METRIC = json.load(_f).get("metric", METRIC)
except (OSError, ValueError):
    pass
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FRAME_TIME = 1.0 / 60.0  # reference src/app.rs:15
UNIT = "body·substeps/s"
# Frames every scene is stepped before anything is timed (independent of --warmup): `*-drop` bodies land after ~0.4 s
# and have stopped tumbling after ~2 s; the other scenes start in (or next to) contact and settle faster.
PREROLL = {"boxes-drop": 120, "mixed-drop": 120, "boxes": 60, "mixed": 60, "stacks": 30}
PREROLL_PILE = 180      # a pile (capi.scene_pile) has fallen and come to rest after ~3 s


def reduce_max_seconds(seconds):
    """MAX over ranks of a wall time (identity when torch.distributed is not initialised)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def timed_frames(step, stream, steps):
    """EXACTLY `steps` calls of step() bracketed by barrier + synchronize on both sides.
    Returns (MAX-over-ranks wall seconds, this rank's HIP-event milliseconds on the kernels' stream)."""
    import torch
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    barrier()
    wall = reduce_max_seconds(time.perf_counter() - t0)
    return wall, ev0.elapsed_time(ev1)


def median_rate(run, units, repeats=3):
    """run() -> seconds; median over `repeats` of units / seconds, plus the individual rates."""
    rates = [units / run() for _ in range(repeats)]
    return statistics.median(rates), rates


def cpu_baseline_pinned(state, shape_id, verts, offsets, substeps, budget_s=4.0, sample=8192, repeats=3):
    """The CPU oracle on the first `sample` bodies of the GPU's pre-rolled state (the state the GPU line is timed on):
    every repeat starts from that same state, median of `repeats`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    n = min(sample, state.shape[0])
    start, sid = state[:n].copy(), shape_id[:n].copy()
    t0 = time.perf_counter()
    ob.step_bodies(start, sid, verts, offsets, FRAME_TIME, substeps)            # calibration frame
    one = time.perf_counter() - t0
    frames = int(max(2, min(200, budget_s / max(one, 1e-6))))

    def run(threads=1):
        bodies = start
        t0 = time.perf_counter()
        for _ in range(frames):
            bodies, _ = ob.step_bodies(bodies, sid, verts, offsets, FRAME_TIME, substeps, threads=threads)
        return time.perf_counter() - t0

    value, rates = median_rate(run, n * substeps * frames, repeats)
    out = {"value": value, "unit": UNIT, "cores": 1, "kind": "port", "repeats": rates,
           "sample": "first %d bodies of the pre-rolled benchmark state, %d frames x %d substeps per repeat, median of %d, "
                     "oracle/xpbd_oracle.c (C restatement of the reference, gcc -O2 -ffp-contract=off), 1 thread "
                     "like the single-threaded reference" % (n, frames, substeps, repeats)}
    cores = min(len(os.sched_getaffinity(0)), 16)        # a 1-GPU box's CPU share is 16 cores
    if cores > 1:                                        # for information: the reference itself is single-threaded
        out["value_all_cores"], _ = median_rate(lambda: run(cores), n * substeps * frames, repeats)
        out["cores_all"] = cores
    return out


def cpu_baseline_contacts(state, shape_id, poly_names, substeps, pad, narrowphase, joints, pick, what, budget_s=4.0, repeats=3,
                          depenetration=0.0):
    """op_contacts_* of the oracle (body-body contact EXTENSION, single thread) on the bodies `pick` (ascending indices)
    of the GPU's pre-rolled state; joints with both bodies inside the sample are kept."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_binding as ob
    n = len(pick)
    start, sid = state[pick].copy(), shape_id[pick].copy()
    polys = ob.polytopes_array(poly_names)
    if joints is None:
        from constraint_solver_amd import capi
        joints = np.zeros(0, dtype=capi.JOINT_DTYPE)
    inside = np.isin(joints["body_a"], pick) & np.isin(joints["body_b"], pick)
    joints = joints[inside].copy()
    joints["body_a"], joints["body_b"] = np.searchsorted(pick, joints["body_a"]), np.searchsorted(pick, joints["body_b"])
    t0 = time.perf_counter()
    ob.contacts_step_joints(start, sid, polys, joints, FRAME_TIME, substeps, pad, narrowphase=narrowphase, max_depenetration_speed=depenetration)
    one = time.perf_counter() - t0
    frames = int(max(1, min(50, budget_s / max(one, 1e-6))))

    def run():
        bodies = start
        t0 = time.perf_counter()
        for _ in range(frames):
            bodies = ob.contacts_step_joints(bodies, sid, polys, joints, FRAME_TIME, substeps, pad, narrowphase=narrowphase,
                                             max_depenetration_speed=depenetration)
        return time.perf_counter() - t0

    value, rates = median_rate(run, n * substeps * frames, repeats)
    return {"value": value, "unit": UNIT, "cores": 1, "kind": "port", "repeats": rates,
            "sample": "%d bodies of the pre-rolled state (%s; %d joints), %d frames x %d substeps per repeat, median of %d, "
                      "oracle/xpbd_pairs_oracle.c op_contacts_* (the build's own CPU definition of the EXTENSION: the "
                      "reference has no body-body contacts), 1 thread" % (n, what, len(joints), frames, substeps, repeats)}


def qrot(np, q, v):
    """cgmath's q * v for arrays of quaternions (s, x, y, z) and vectors."""
    s, u = q[:, :1], q[:, 1:]
    t = np.cross(u, v) + v * s
    return np.cross(u, t) * 2 + v


def chain_joints(capi, np, n_joints, count, pitch, grid_w, state=None, first=0):
    """BASELINE configs[4] (extension): chains of 5 bodies with 4 joints each, "every 4th a hinge":
    three distance joints centre to centre at the pitch, the fourth an XPBD_JOINT_HINGE (ball joint + angular term) at the
    point midway between the two bodies about the vertical.  A chain that would wrap around the end of a grid row loses
    the joint across the wrap.  state: the (count, 38) bodies [first, first + count) at t = 0 -- a hinge's anchors and axes
    live in the object space of its (randomly turned) bodies, so they are taken from the initial poses: every hinge
    starts exactly satisfied.  Without `state` (sharded runs, which do not hold all bodies) all joints are distance joints."""
    k = np.arange(n_joints)
    a = (k // 4) * 5 + (k % 4)
    keep = (a + 1 < count) & (a // grid_w == (a + 1) // grid_w)
    a, k = a[keep], k[keep]
    joints = np.zeros(len(a), dtype=capi.JOINT_DTYPE)
    joints["body_a"], joints["body_b"] = a, a + 1
    joints["anchor_a"], joints["anchor_b"], joints["distance"] = [0.5, 0.5, 0.5], [0.5, 0.5, 0.5], pitch
    if state is not None:
        h = np.nonzero(k % 4 == 3)[0]
        ba, bb = state[a[h] - first], state[a[h] + 1 - first]

        def pose(b):                                     # Rigid::frame (src/rigid.rs:75-80): origin, rotation; world centre of mass
            com, q = b[:, 28:31], b[:, 34:38]
            return (b[:, 31:34] + com) + qrot(np, q, -com), q, b[:, 31:34] + com
        oa, qa, ca = pose(ba)
        ob_, qb, cb = pose(bb)
        mid = 0.5 * (ca + cb)
        conj = lambda q: q * np.array([1.0, -1.0, -1.0, -1.0])
        up = np.tile([0.0, 0.0, 1.0], (len(h), 1))
        joints["kind"][h], joints["distance"][h] = capi.JOINT_HINGE, 0.0
        joints["anchor_a"][h], joints["anchor_b"][h] = qrot(np, conj(qa), mid - oa), qrot(np, conj(qb), mid - ob_)
        joints["axis_a"][h], joints["axis_b"][h] = qrot(np, conj(qa), up), qrot(np, conj(qb), up)
    return joints


SCENE_KIND = {"boxes": "SCENE_BOXES", "mixed": "SCENE_MIXED", "boxes-drop": "SCENE_BOXES_DROP", "mixed-drop": "SCENE_MIXED_DROP",
              "stacks": "SCENE_BOX_STACKS"}
POLY_NAMES = {False: [("cube", 1.0)], True: [("cube", 1.0), ("tetrahedron", 0.5), ("icosahedron", 0.5)]}


def contacts_workload(scene, bodies_per_gpu, substeps, narrowphase, joints, pitch, layers):
    what = "mixed convex polyhedra (cube / tetrahedron / icosahedron)" if "mixed" in scene else "unit boxes"
    layout = ("dropped as a pile: %d layers of a %.2f m grid, pre-rolled %d frames to rest" % (layers, pitch, PREROLL_PILE)) if layers \
        else "scene '%s'%s, pre-rolled %d frames" % (scene, "" if pitch == 2.0 else " at %.2f m pitch" % pitch, PREROLL[scene])
    return ("EXTENSION body-body contacts: %d %s per GPU x %d substeps/frame, dt=1/60, %s, %s narrowphase%s; ground "
            "contacts as in the reference, pair contacts Jacobi-averaged"
            % (bodies_per_gpu, what, substeps, layout, "GJK + EPA" if narrowphase == "gjk" else "SAT",
               ", %d distance joints" % joints if joints else ""))


def scene_grid_width(capi, kind, bodies_per_rank, total):
    """Grid width of a contact scene.  Index ranges are the shards, so the grid is laid out for ONE rank's bodies to be
    about square and the ranks' slabs to follow each other along y: box stacks (16 bodies per grid point) 128 columns wide
    for 262 144 bodies per rank -- the generator's default (the square root of the BODY count) would make that world 16 times
    wider than deep and a rank's slab a dozen rows of which four are halo."""
    if kind == capi.SCENE_BOX_STACKS:
        return capi.default_grid_width(max(bodies_per_rank // 16, 1))
    return capi.default_grid_width(total)


def contacts_scene(capi, args, kind, total, pitch, layers):
    """All `total` bodies of a contact scene: the seeded grid scene at `pitch`, or (layers > 0) the same bodies as a pile."""
    if layers:
        return capi.scene_pile(kind, args.seed, total, pitch, layers)
    state, shape_id = capi.scene_generate(kind, args.seed, total, grid_w=scene_grid_width(capi, kind, total, total))
    if pitch != 2.0:
        state[:, 31:33] *= pitch / 2.0
    return state, shape_id


def sample_of(np, total, layers, sample):
    """Indices of a bounded, self-contained sample of a scene: the first `sample` bodies, or -- for a pile -- the same
    block of grid points in EVERY layer (a pile's bottom layer alone is not a pile)."""
    if not layers:
        return np.arange(min(sample, total)), "the first %d bodies" % min(sample, total)
    per_layer = (total + layers - 1) // layers
    block = min(sample // layers, per_layer)
    pick = np.concatenate([np.arange(l * per_layer, min(l * per_layer + block, total)) for l in range(layers)])
    return pick, "the first %d grid points of each of the %d layers" % (block, layers)


def run_contacts(capi, np, torch, args, scene, bodies, narrowphase, joints_n, pitch, rank, local_rank, world_size, steps, warmup,
                 with_cpu, layers=0, name=None, depenetration=0.0, preroll=None):
    """One contact-pipeline measurement on a world of its own: pre-roll, warm-up, K timed frames.  Returns the result
    object (rank 0) -- value, roofline of a whole substep, cpu_baseline -- or None."""
    kind = getattr(capi, SCENE_KIND[scene])
    total = bodies * world_size
    pad = 0.02
    preroll = preroll or (PREROLL_PILE if layers else PREROLL[scene])
    if world_size > 1 or args.local_shards:
        return run_contacts_sharded(capi, np, torch, args, scene, kind, bodies, narrowphase, joints_n, pitch, layers, preroll, rank,
                                    local_rank, world_size, steps, warmup, local_shards=args.local_shards, depenetration=depenetration)
    count = total
    state, shape_id = contacts_scene(capi, args, kind, total, pitch, layers)
    world = capi.World(device=local_rank, mode=capi.MODE_CONTACTS)
    world.set_polytopes(capi.scene_polytopes(kind))
    world.set_contact_pad(pad)
    world.set_narrowphase(capi.NARROWPHASE_GJK_EPA if narrowphase == "gjk" else capi.NARROWPHASE_SAT)
    world.set_max_depenetration_speed(depenetration)
    world.set_sat_schedule({"auto": capi.SAT_SCHEDULE_AUTO, "one-pass": capi.SAT_SCHEDULE_ONE_PASS,
                            "two-pass": capi.SAT_SCHEDULE_TWO_PASS}[args.sat_schedule])
    world.upload(state, shape_id)
    joints = chain_joints(capi, np, joints_n, count, pitch, scene_grid_width(capi, kind, total, total), state=state) if joints_n else None
    if joints is not None:
        world.set_joints(joints)
    stream = torch.cuda.current_stream()
    world.set_stream(stream.cuda_stream)
    for _ in range(preroll):
        world.step(FRAME_TIME, args.substeps)
    start_state = world.download() if with_cpu else None
    ground_start = len(world.contacts()) / max(count, 1)
    for _ in range(warmup):
        world.step(FRAME_TIME, args.substeps)
    world.contact_stats()                                      # reset the counters: the stats below cover the timed frames only
    wall, device_ms = timed_frames(lambda: world.step(FRAME_TIME, args.substeps), stream, steps)
    result = None
    if rank == 0:
        pairs, touching, points = world.contact_stats()
        n_sub = max(steps * args.substeps, 1)
        touching_ps, points_ps = touching / n_sub, points / n_sub
        substep_s = device_ms * 1e-3 / n_sub
        # Algorithmic bytes of ONE SUBSTEP of the pipeline (narrowphase + the fused per-body kernel), DESIGN.md 8: per body
        # 716 B (state 13 + 25 doubles, frames in 14 + out 17 doubles, state out 13, shape id), per listed pair 120 B (two
        # frames, the pair), per touching pair its 24-byte header written once and read by both bodies, per contact point
        # 48 B written once and read twice.
        substep_bytes = count * 716 + pairs * 120 + touching_ps * 72 + points_ps * 144
        achieved = substep_bytes / substep_s / 1e9
        # ... and the ALGORITHMIC bytes of SURVEY.md 8(d): 412 B per body.substep (the pinned path) + per candidate pair two
        # 56-byte pose reads and one manifold write of up to 4 x 56 B = 336 B
        survey_bytes = count * 412 + pairs * 336
        traffic = traffic_note = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile) and name:                # PMC counters cannot be read from inside the run: the committed pass
            entry = json.load(open(tfile)).get("contacts", {}).get(name, {})
            traffic, traffic_note = entry.get("bytes_per_substep"), entry.get("source")
        end_state = world.download()
        speed = np.linalg.norm(end_state[:, 22:25], axis=1)
        result = {
            "value": total * args.substeps * steps / wall, "unit": UNIT, "ms_per_step": wall * 1e3 / steps,
            "steps": steps, "warmup": warmup, "preroll_frames": preroll,
            "config": {"workload": contacts_workload(scene, bodies, args.substeps, narrowphase, joints_n, pitch, layers),
                       "bodies_per_gpu": bodies, "substeps": args.substeps, "scene": scene, "pitch": pitch, "layers": layers,
                       "narrowphase": narrowphase, "joints": 0 if joints is None else int(len(joints)),
                       "hinges": 0 if joints is None else int((joints["kind"] == capi.JOINT_HINGE).sum()),
                       "max_depenetration_speed": depenetration,
                       "neighbour_pairs": pairs, "touching_pairs_per_substep": touching_ps,
                       "manifold_points_per_substep": points_ps,
                       "ground_contacts_per_body_at_start": ground_start,
                       "ground_contacts_per_body_at_end": len(world.contacts()) / max(count, 1),
                       "speed_p50_p99_max_at_end": [float(np.nanpercentile(speed, 50)), float(np.nanpercentile(speed, 99)),
                                                    float(np.nanmax(speed))],
                       "extension": "body-body contacts: NOT in the reference (parity unpinned)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": "one substep: narrowphase + k_pair_solve_integrate_ground", "launch_us": substep_s * 1e6,
                         "bytes_per_launch": substep_bytes,
                         "achieved_survey_8d": survey_bytes / substep_s / 1e9, "frac_survey_8d": survey_bytes / substep_s / 1e9 / HBM_PEAK_GBPS,
                         "bytes_per_launch_survey_8d": survey_bytes,
                         "note": "whole-substep figure from the byte model of DESIGN.md 8 (the narrowphase is f64-VALU / latency "
                                 "bound, the per-body kernel bandwidth bound; per-kernel times in profiles/)"},
            "cpu_baseline": None,
        }
        if with_cpu:
            pick, what = sample_of(np, total, layers, 4096)
            result["cpu_baseline"] = cpu_baseline_contacts(start_state, shape_id, POLY_NAMES["mixed" in scene], args.substeps, pad,
                                                           1 if narrowphase == "gjk" else 0, joints, pick, what, depenetration=depenetration)
    world.close()
    return result


def run_contacts_sharded(capi, np, torch, args, scene, kind, bodies, narrowphase, joints_n, pitch, layers, preroll, rank, local_rank,
                         world_size, steps, warmup, local_shards=0, depenetration=0.0):
    """EXTENSION, N > 1: body-body contacts through the NATIVE multi-GPU world (xpbd_multi_world_*, csrc/xpbd_multi.cpp):
    every rank hands over its index slice of the scene (generating ONLY those bodies); the library cuts the shards -- slabs
    of space across the world's longest axis, near-equal body counts, re-balanced at every re-plan -- moves each body to
    its owner, steps owned + ghost bodies, and the boundary bodies travel in one ncclAllGather per substep (RCCL over
    xGMI); a frame in which a body outruns halo_margin is undone, re-planned and run again.  local_shards > 0: rehearsal on one GPU -- that many shards in THIS process on device 0
    with the in-process transport.  Not in the reference; parity = sharded == single device (tests/test_gpu_multi.py)."""
    import torch.distributed as dist
    from constraint_solver_amd.sharding import shard_range
    n_ranks = local_shards or world_size
    total = bodies * n_ranks
    parts, sids, first_global = [], [], None
    my_ranks = range(n_ranks) if local_shards else [rank]
    for r in my_ranks:
        first, count = shard_range(total, r, n_ranks)
        first_global = first if first_global is None else first_global
        if layers:
            part, sid = capi.scene_pile(kind, args.seed, total, pitch, layers, first=first, count=count,
                                        y_offset=r * capi.pile_depth(count, pitch, layers))
        else:
            part, sid = capi.scene_generate(kind, args.seed, total, first=first, count=count, grid_w=scene_grid_width(capi, kind, bodies, total))
            if pitch != 2.0:
                part[:, 31:33] *= pitch / 2.0
        parts.append(part)
        sids.append(sid)
    state, shape_id = np.concatenate(parts), np.concatenate(sids)
    joints = None
    if joints_n:
        # ALL joints of the world on every rank (the ABI asks for that).  A hinge's anchors come from its two bodies' initial
        # poses, and the scene is a pure function of (seed, index): generate it in chunks of whole chains, keep only the joints.
        grid_w, chunk, parts_j = scene_grid_width(capi, kind, bodies, total), 5 * 65536, []
        for c0 in range(0, total, chunk):
            cn = min(chunk, total - c0)
            part, _ = capi.scene_generate(kind, args.seed, total, first=c0, count=cn, grid_w=grid_w)
            if pitch != 2.0:
                part[:, 31:33] *= pitch / 2.0
            j = chain_joints(capi, np, (cn // 5) * 4, cn, pitch, grid_w, state=part)      # (chunk-local indices; c0 is a multiple of 5)
            keep = (j["body_a"] + c0) // grid_w == (j["body_b"] + c0) // grid_w             # the row test, in global indices
            j = j[keep]
            j["body_a"] += c0
            j["body_b"] += c0
            parts_j.append(j)
        joints = np.concatenate(parts_j)[: joints_n * n_ranks]
    if local_shards:
        world = capi.MultiWorld(n_ranks, devices=[local_rank] * n_ranks, transport=capi.TRANSPORT_LOCAL, halo_margin=args.halo_margin,
                                narrowphase=capi.NARROWPHASE_GJK_EPA if narrowphase == "gjk" else capi.NARROWPHASE_SAT, auto_replan=True)
        transport = "in-process peer copies (rehearsal: %d shards on one device)" % n_ranks
    else:
        cid = [capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(cid, src=0)
        world = capi.MultiWorld(n_ranks, first_rank=rank, devices=[local_rank], transport=capi.TRANSPORT_RCCL, comm_id=cid[0],
                                halo_margin=args.halo_margin, narrowphase=capi.NARROWPHASE_GJK_EPA if narrowphase == "gjk" else capi.NARROWPHASE_SAT,
                                auto_replan=True)
        transport = "ncclAllGather, RCCL bound from %s" % capi.comm_library()
    world.set_polytopes(capi.scene_polytopes(kind))
    world.set_max_depenetration_speed(depenetration)
    world.upload(state, shape_id, first_global, total, joints)
    for _ in range(preroll + warmup):
        world.step(FRAME_TIME, args.substeps)
    world.contact_stats()
    plans_before = world.halo_stats()["plans"]
    wall, _ = timed_frames(lambda: world.step(FRAME_TIME, args.substeps), torch.cuda.current_stream(), steps)
    result = None
    if rank == 0:
        halo = world.halo_stats()
        halo["plans_during_the_timed_frames"] = halo["plans"] - plans_before
        pairs, touching, points = world.contact_stats()            # of this process's shards (owned + ghost bodies)
        n_sub = max(steps * args.substeps, 1)
        result = {
            "value": total * args.substeps * steps / wall, "unit": UNIT, "ms_per_step": wall * 1e3 / steps,
            "steps": steps, "warmup": warmup, "preroll_frames": preroll,
            "config": {"workload": contacts_workload(scene, bodies, args.substeps, narrowphase, joints_n, pitch, layers),
                       "bodies_per_gpu": bodies, "bodies_total": total, "substeps": args.substeps, "scene": scene,
                       "max_depenetration_speed": depenetration,
                       "sharding": "shards cut by the library from the spatial-hash cell order (slabs across the world's longest axis, "
                                   "near-equal body counts, re-balanced at re-plans), owned + ghost bodies per rank; halo plan, end-of-frame "
                                   "halo-validity check (a violating frame is undone and re-run) and one all-gather per substep inside "
                                   "xpbd_multi_world_step (%s)" % transport,
                       "halo": halo, "halo_margin": args.halo_margin, "plan": world.plan_stats(), "neighbour_pairs_here": pairs, "touching_pairs_per_substep_here": touching / n_sub,
                       "manifold_points_per_substep_here": points / n_sub,
                       "extension": "not in the reference (parity unpinned; sharded == single device bit for bit)"},
            "roofline": None, "cpu_baseline": None,
        }
        # the same byte model as on one GPU, for THIS rank's shard (owned + ghost bodies are all stepped here)
        local_bodies = halo["owned"] + halo["ghosts"]
        substep_bytes = local_bodies * 716 + pairs * 120 + touching / n_sub * 72 + points / n_sub * 144
        substep_s = wall / n_sub
        result["roofline"] = {"bound": "hbm", "achieved": substep_bytes / substep_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": substep_bytes / substep_s / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                              "kernel": "one substep of one rank: integrate + ground, narrowphase, pair solve, halo export / "
                                        "all-gather / import", "launch_us": substep_s * 1e6, "bytes_per_launch": substep_bytes,
                              "note": "per-GPU figure from wall time (the exchange is inside it); byte model of DESIGN.md 8"}
    world.close()
    return result


def contacts_in_child_processes(args, rank):
    """N > 1: the contacts sub-results run the native multi-GPU world (RCCL all-gathers of its own).  Whatever happens in
    there -- a communicator that never forms, a rank lost, a crash inside the library -- the headline line must still come
    out, so every rank runs them in a CHILD process (same ranks, a rendezvous port of its own) and waits for it with a
    time limit; rank 0 takes the sub-results from its child's stdout.  The parent's own process group stays idle meanwhile."""
    import subprocess
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 1)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)        # rank 0 of the children hosts their store itself
    cmd = [sys.executable, os.path.abspath(__file__), "--contacts-child", "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--substeps", str(args.substeps), "--seed", str(args.seed), "--backend", args.backend,
           "--halo-margin", str(args.halo_margin)]
    if args.single_device:
        cmd.append("--single-device")
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL,
                             stderr=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, text=True)
    try:
        out, err = child.communicate(timeout=args.contacts_timeout)
    except subprocess.TimeoutExpired:
        child.kill()
        child.communicate()
        return {"error": "timed out after %d s" % args.contacts_timeout}
    if rank != 0:
        return {}
    for line in reversed((out or "").strip().splitlines()):
        try:
            return json.loads(line)["contacts"]
        except (ValueError, KeyError):
            continue
    return {"error": "child exited with code %d: %s" % (child.returncode, (err or "").strip()[-400:])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--bodies", type=int, default=262144, help="bodies per GPU (weak scaling)")
    ap.add_argument("--substeps", type=int, default=20)
    ap.add_argument("--scene", default="boxes-drop", choices=sorted(SCENE_KIND))
    ap.add_argument("--mode", default="fused", choices=["fused", "substep", "contacts"],
                    help="contacts = EXTENSION (body-body contacts; not in the reference, parity unpinned)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--block-size", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the informational extra measurements (unfused roofline, HBM-resident size, PCIe-inclusive)")
    ap.add_argument("--no-contacts", action="store_true",
                    help="skip the `contacts` sub-results of the default run (box stacks with SAT contacts, mixed polyhedra on GJK/EPA)")
    ap.add_argument("--halo-margin", type=float, default=1.0, help="sharded contact scenes: metres a body may travel between re-plans")
    ap.add_argument("--contacts-timeout", type=int, default=420,
                    help="N > 1: seconds after which the contacts sub-results are abandoned and the line is printed without them")
    ap.add_argument("--only", default="", help="profiling aid: run ONLY this part ('pinned', or a contacts sub-result name) "
                                               "with no extras and no CPU leg")
    ap.add_argument("--narrowphase", default="sat", choices=["sat", "gjk"],
                    help="contacts mode: SAT (up to 8 points per pair) or GJK + EPA (face manifolds where the normal is a face normal, else one point)")
    ap.add_argument("--sat-schedule", default="auto", choices=["auto", "one-pass", "two-pass"],
                    help="contacts mode, SAT: pre-test inside the SAT kernel, or as a pass of its own (with the cached separating "
                         "face axes) and the SAT over the survivors; auto = the library's choice.  Same results either way")
    ap.add_argument("--joints", type=int, default=0,
                    help="contacts mode: link bodies into chains of 5 along x with this many distance joints (4 per chain)")
    ap.add_argument("--local-shards", type=int, default=0,
                    help="contacts: rehearsal of the multi-GPU path on ONE GPU -- this many shards in this process on one device, "
                         "exchanged by the in-process transport (bodies per shard = --bodies)")
    ap.add_argument("--pitch", type=float, default=2.0,
                    help="grid pitch of the scene in metres (generator default 2.0); < 2 packs bodies so that they collide")
    ap.add_argument("--max-depenetration-speed", type=float, default=0.0,
                    help="contacts mode: xpbd_world_set_max_depenetration_speed in m/s (0 = off = the reference's solver loop)")
    ap.add_argument("--layers", type=int, default=0,
                    help="contacts mode: drop the scene's bodies as a pile of this many grid layers (capi.scene_pile)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the barrier/MAX reduction (gloo: rehearsal only)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--contacts-child", action="store_true", help=argparse.SUPPRESS)   # N > 1: see contacts_in_child_processes
    ap.add_argument("--force-contacts", action="store_true", help=argparse.SUPPRESS)   # rehearsal of the child's failure path
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world_size, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.single_device:
        if args.backend != "gloo":
            raise SystemExit("--single-device shares one GPU between ranks; RCCL refuses that, use --backend gloo")
        local_rank = 0
    if args.single_device and world_size > 1 and not (args.force_contacts or args.contacts_child):
        args.no_contacts = True          # the contacts sub-results exchange halos over RCCL, which refuses two ranks on one device
    torch.cuda.set_device(local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)

    from constraint_solver_amd import capi
    from constraint_solver_amd.sharding import shard_range

    # Everything runs on an explicit torch stream so torch.cuda.Event (HIP events) brackets OUR launches; the
    # default stream's handle is 0, which the ABI reads as "use the world's own stream".
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    with_cpu = world_size == 1 and not args.no_cpu_baseline and not args.only

    # the contacts sub-results of the default line: the north-star workloads (BASELINE.json configs[3] size with SAT
    # box stacks; configs[2]: 65 536 mixed convex polyhedra on the GJK/EPA path)
    sub_runs = {
        "stacks_262144_sat": dict(scene="stacks", bodies=262144, narrowphase="sat", joints_n=0, pitch=2.0),
        "stacks_262144_gjk_epa": dict(scene="stacks", bodies=262144, narrowphase="gjk", joints_n=0, pitch=2.0),
        "boxes_pile_262144_sat": dict(scene="boxes-drop", bodies=262144, narrowphase="sat", joints_n=0, pitch=1.8, layers=4),
        # (the mixed piles with xpbd_world_set_max_depenetration_speed = 3 m/s: without a limit a light tetrahedron -- 1/48 of a
        #  cube's mass -- squeezed between heavy bodies leaves the pile at > 100 m/s, frame after frame, and the pile never rests;
        #  with it nobody moves faster than 5 m/s after 210 frames, hence the longer pre-roll: profiles/r03_d_mixed_pile_speed_census.json)
        "mixed_pile_65536_gjk_epa": dict(scene="mixed-drop", bodies=65536, narrowphase="gjk", joints_n=0, pitch=1.4, layers=4, depenetration=3.0,
                                         preroll=240),
        "mixed_pile_65536_sat": dict(scene="mixed-drop", bodies=65536, narrowphase="sat", joints_n=0, pitch=1.4, layers=4, depenetration=3.0,
                                     preroll=240),
        "boxes_262144_joints_65536": dict(scene="boxes-drop", bodies=262144, narrowphase="sat", joints_n=65536, pitch=2.0),
    }
    if args.contacts_child:
        # N > 1: the sub-results of the default line, in a process group of their own (see contacts_in_child_processes)
        subs = {}
        for name, cfg in sub_runs.items():
            try:
                r = run_contacts(capi, np, torch, args, rank=rank, local_rank=local_rank, world_size=world_size,
                                 steps=min(args.steps, 30), warmup=min(args.warmup, 10), with_cpu=False, name=name, **cfg)
            except capi.XpbdError as e:                     # e.g. XPBD_E_HALO in a scene that outruns its margin
                r = {"error": str(e)}
            subs[name] = r
        if rank == 0:
            print(json.dumps({"contacts": subs}), flush=True)
        barrier()
        dist.destroy_process_group()
        return subs
    if args.only and args.only != "pinned":
        if args.only not in sub_runs:
            raise SystemExit("--only: one of pinned, %s" % ", ".join(sub_runs))
        r = run_contacts(capi, np, torch, args, rank=rank, local_rank=local_rank, world_size=world_size, steps=args.steps,
                         warmup=args.warmup, with_cpu=False, name=args.only, **sub_runs[args.only])
        if rank == 0:
            print(json.dumps({"metric": METRIC, "n_gpus": world_size, **r}), flush=True)
        barrier()
        if world_size > 1:
            dist.destroy_process_group()
        return r

    if args.mode == "contacts":
        r = run_contacts(capi, np, torch, args, args.scene, args.bodies, args.narrowphase, args.joints, args.pitch, rank, local_rank,
                         world_size, args.steps, args.warmup, with_cpu, layers=args.layers, depenetration=args.max_depenetration_speed)
        result = None
        if rank == 0:
            result = {"metric": METRIC, "value": r["value"], "unit": UNIT, "n_gpus": world_size, "steps": args.steps,
                      "warmup": args.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                      "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": r["config"], "roofline": r["roofline"],
                      "cpu_baseline": r["cpu_baseline"], "preroll_frames": r["preroll_frames"]}
            print(json.dumps(result), flush=True)
        barrier()
        if world_size > 1:
            dist.destroy_process_group()
        return result

    # ---------------------------------------------------------------- the pinned path (reference semantics)
    kind = getattr(capi, SCENE_KIND[args.scene])
    mode = {"fused": capi.MODE_FUSED, "substep": capi.MODE_PER_SUBSTEP}[args.mode]
    total = args.bodies * world_size
    first, count = shard_range(total, rank, world_size)
    verts, offsets = capi.scene_shapes(kind)
    bodies, shape_id = capi.scene_generate(kind, args.seed, total, first=first, count=count)
    if args.pitch != 2.0:
        bodies[:, 31:33] *= args.pitch / 2.0

    # Pre-roll into the steady regime on a world of its own, created with the contact trace on: that instantiates the
    # stepper as k_step<true>, so in a rocprofv3 kernel trace of this command the k_step<false> rows are exactly the
    # warm-up and timed launches below (same arithmetic, same bits; the trace only adds the mask stores).
    with capi.World(device=local_rank, mode=capi.MODE_FUSED, trace_contacts=True) as pre:
        pre.set_shapes(verts, offsets)
        pre.upload(bodies, shape_id)
        for _ in range(PREROLL[args.scene]):
            pre.step(FRAME_TIME, args.substeps)
        start_state = pre.download()
    world = capi.World(device=local_rank, mode=mode, block_size=args.block_size)
    world.set_shapes(verts, offsets)
    world.upload(start_state, shape_id)                  # inputs resident in HBM before any timing
    world.set_stream(stream.cuda_stream)
    world.step(FRAME_TIME, args.substeps)                # (one untimed frame so that the contact list below exists)
    contacts_start = len(world.contacts()) / max(count, 1)
    for _ in range(args.warmup):
        world.step(FRAME_TIME, args.substeps)
    wall, device_ms = timed_frames(lambda: world.step(FRAME_TIME, args.substeps), stream, args.steps)

    result = None
    if rank == 0:
        launches_per_step = 1 if mode == capi.MODE_FUSED else args.substeps
        launch_s = device_ms * 1e-3 / (args.steps * launches_per_step)
        bytes_per_launch = capi.BYTES_PER_BODY_SUBSTEP * count   # 412 B x bodies, fused or not
        achieved = bytes_per_launch / launch_s / 1e9
        traffic, traffic_note = None, None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):                        # PMC counters cannot be read from inside the run: the figure is the
            entry = json.load(open(tfile)).get("%s_%d" % (args.mode, count), {})   # committed rocprofv3 --pmc pass of this
            traffic, traffic_note = entry.get("bytes_per_launch"), entry.get("source")        # same configuration
        result = {
            "metric": METRIC,
            "value": total * args.substeps * args.steps / wall,
            "unit": UNIT,
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d %s per GPU x %d substeps/frame, dt=1/60, ground contacts (reference semantics: bodies "
                                   "do not interact), seeded scene '%s' pre-rolled %d frames into resting contact"
                                   % (args.bodies, "mixed convex polyhedra" if "mixed" in args.scene else "unit boxes",
                                      args.substeps, args.scene, PREROLL[args.scene]),
                       "bodies_per_gpu": args.bodies, "bodies_total": total, "substeps": args.substeps,
                       "mode": args.mode, "preroll_frames": PREROLL[args.scene],
                       "ground_contacts_per_body_at_start": contacts_start,
                       "ground_contacts_per_body_at_end": len(world.contacts()) / max(count, 1),
                       "sharding": "contiguous body-index ranges, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": "k_step", "launch_us": launch_s * 1e6, "bytes_per_launch": bytes_per_launch,
                         "note": ("fused: all %d substeps of a body run in registers, so one launch moves 412 B/body "
                                  "once and the kernel is f64-VALU bound, not HBM bound" % args.substeps)
                         if mode == capi.MODE_FUSED else "one launch per substep: state round-trips HBM every substep"},
        }
        if mode == capi.MODE_FUSED and count == 262144 and args.scene == "boxes-drop":
            # What actually bounds the fused kernel: the f64 issue rate (no FMA under -ffp-contract=off).  PMC counters
            # cannot be read from inside the run: these come from the newest committed rocprofv3 --pmc pass of this
            # same configuration (scripts/gpu_profile.sh ... pmc -> profiles/*_summary.json, derived_k_step_fused).
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")), reverse=True):
                try:
                    d = json.load(open(f)).get("derived_k_step_fused")
                except (OSError, ValueError):
                    d = None
                if d:
                    result["roofline"]["valu"] = {"bound": "f64 VALU issue", "pipe_busy": d.get("f64_valu_pipe_busy"),
                                                  "lane_utilisation": d.get("lane_utilisation"),
                                                  "insts_per_wave_substep": d.get("valu_insts_per_wave_substep"),
                                                  "source": os.path.relpath(f, ROOT)}
                    break
        extras = world_size == 1 and not args.no_extras and not args.only
        if extras:
            # The HBM roof as measured in THIS run: device-to-device copy of 2 GiB (8x the 256 MiB Infinity Cache) by the
            # library's own streaming kernel ("HBM_peak both nominal and measured").
            measured = capi.selftest_hbm_copy(1 << 31, 10, device=local_rank)
            result["roofline"]["peak_measured"] = measured
            result["roofline"]["frac_of_measured"] = achieved / measured
        if extras and mode == capi.MODE_FUSED:
            # The same kernel scheduled one launch per substep (state round-trips HBM every substep):
            # the HBM-roofline-comparable form, measured in the same run on the same resident state.
            frames = max(3, min(10, args.steps))
            world.set_mode(capi.MODE_PER_SUBSTEP)
            for _ in range(2):
                world.step(FRAME_TIME, args.substeps)
            _, ms = timed_frames(lambda: world.step(FRAME_TIME, args.substeps), stream, frames)
            world.set_mode(capi.MODE_FUSED)
            sub_s = ms * 1e-3 / (frames * args.substeps)
            tr = None
            if os.path.exists(tfile):
                tr = json.load(open(tfile)).get("substep_%d" % count, {}).get("bytes_per_launch")
            ach = bytes_per_launch / sub_s / 1e9
            result["roofline_unfused"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                          "frac": ach / HBM_PEAK_GBPS, "peak_measured": measured, "frac_of_measured": ach / measured,
                                          "traffic": tr, "kernel": "k_step", "launch_us": sub_s * 1e6,
                                          "body_substeps_per_s": count / sub_s,
                                          "note": "XPBD_MODE_PER_SUBSTEP: one launch per substep, 412 B per body per launch; the "
                                                  "108 MB working set fits the 256 MiB Infinity Cache (see roofline_hbm_resident)"}
            # ... and at a size that does NOT fit the Infinity Cache: 2 097 152 bodies = 864 MB of state per launch
            big_n = 1 << 21
            big_bodies, big_sid = capi.scene_generate(kind, args.seed, big_n)
            with capi.World(device=local_rank, mode=capi.MODE_FUSED) as big:
                big.set_shapes(verts, offsets)
                big.upload(big_bodies, big_sid)
                del big_bodies
                big.set_stream(stream.cuda_stream)
                for _ in range(PREROLL[args.scene]):
                    big.step(FRAME_TIME, args.substeps)
                big.set_mode(capi.MODE_PER_SUBSTEP)
                big.step(FRAME_TIME, args.substeps)
                _, ms = timed_frames(lambda: big.step(FRAME_TIME, args.substeps), stream, 5)
                big_s = ms * 1e-3 / (5 * args.substeps)
                big_bytes = capi.BYTES_PER_BODY_SUBSTEP * big_n
                big_contacts = len(big.contacts()) / big_n
            ach = big_bytes / big_s / 1e9
            big_tr = json.load(open(tfile)).get("substep_%d" % big_n, {}).get("bytes_per_launch") if os.path.exists(tfile) else None
            # ... and what the memory system gives this very ACCESS PATTERN (38 doubles in, 13 out per body: 51 concurrent
            # streams in the world's field-major layout) with no arithmetic at all -- the kernel's real roof -- and what a
            # tile-major layout of the same bytes would give
            pattern = capi.selftest_field_streams(big_n, tile_major=False, repeats=10, device=local_rank)
            pattern_tiles = capi.selftest_field_streams(big_n, tile_major=True, repeats=10, device=local_rank)
            result["roofline_hbm_resident"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                               "frac": ach / HBM_PEAK_GBPS, "peak_measured": measured, "frac_of_measured": ach / measured,
                                               "peak_access_pattern": pattern, "frac_of_access_pattern": ach / pattern,
                                               "peak_access_pattern_tile_major": pattern_tiles,
                                               "traffic": big_tr, "kernel": "k_step", "launch_us": big_s * 1e6, "bodies": big_n,
                                               "bytes_per_launch": big_bytes, "body_substeps_per_s": big_n / big_s,
                                               "ground_contacts_per_body": big_contacts,
                                               "note": "XPBD_MODE_PER_SUBSTEP at 2 097 152 bodies: 864 MB per launch, 3.4x the "
                                                       "Infinity Cache, so this one IS an HBM number"}
        if extras:
            # Informational, never `value`: the same frame when the boundary hands over HOST buffers
            # (AoS upload over PCIe -> step -> AoS download), as a literal per-frame drop-in would.
            state = world.download()
            t0 = time.perf_counter()
            for _ in range(3):
                world.upload(state, shape_id)
                world.step(FRAME_TIME, args.substeps)
                state = world.download()
            per_frame = (time.perf_counter() - t0) / 3
            result["pcie_inclusive"] = {"value": count * args.substeps / per_frame, "unit": UNIT,
                                        "ms_per_frame": per_frame * 1e3,
                                        "what": "pageable host AoS upload + step + download every frame"}
            # What the reference's app actually needs per frame: state stays resident, only Rigid::frame() of every
            # body comes back for rendering (src/app.rs:227-232).
            t0 = time.perf_counter()
            for _ in range(5):
                world.step(FRAME_TIME, args.substeps)
                world.frames()
            per_frame = (time.perf_counter() - t0) / 5
            result["render_readback"] = {"value": count * args.substeps / per_frame, "unit": UNIT,
                                         "ms_per_frame": per_frame * 1e3,
                                         "what": "step + download of Rigid::frame() (56 B/body) every frame"}
        if with_cpu:
            result["cpu_baseline"] = cpu_baseline_pinned(start_state, shape_id, verts, offsets, args.substeps)
    world.close()

    # ---------------------------------------------------------------- the north-star workloads (extension)
    if not args.no_contacts and not args.only and mode == capi.MODE_FUSED:
        if world_size > 1:
            subs = contacts_in_child_processes(args, rank)
        else:
            subs = {}
            for name, cfg in sub_runs.items():
                try:
                    subs[name] = run_contacts(capi, np, torch, args, rank=rank, local_rank=local_rank, world_size=world_size,
                                              steps=min(args.steps, 30), warmup=min(args.warmup, 10), with_cpu=with_cpu, name=name, **cfg)
                except capi.XpbdError as e:
                    subs[name] = {"error": str(e)}
        if rank == 0:
            result["contacts"] = subs
    if rank == 0:
        # The north-star numbers once more, short and LAST on the line (a long line is kept from its tail): body.substeps/s of
        # the workloads BASELINE.json names -- box stacks with SAT contacts, mixed convex polyhedra on GJK/EPA, bodies + joints --
        # with the fraction of the 8 TB/s HBM roof at SURVEY.md 8(d)'s algorithmic bytes, next to the pinned headline.
        subs = result.get("contacts") or {}

        def brief(key):
            r = subs.get(key) or {}
            if "value" not in r:
                return {"error": r.get("error", "not run")} if r else None
            roof = r.get("roofline") or {}
            return {"value": r["value"], "us_per_substep": roof.get("launch_us"), "frac_hbm_survey_8d": roof.get("frac_survey_8d"),
                    "frac_hbm_byte_model": roof.get("frac")}
        hbm = result.get("roofline_hbm_resident") or {}
        result["north_star"] = {
            "unit": UNIT, "n_gpus": world_size,
            "pinned_262144_fused": {"value": result["value"], "frac_hbm": result["roofline"]["frac"],
                                    "f64_pipe_busy": (result["roofline"].get("valu") or {}).get("pipe_busy")},
            "pinned_hbm_resident": {"bodies": hbm.get("bodies"), "frac_hbm": hbm.get("frac"), "frac_of_access_pattern": hbm.get("frac_of_access_pattern")}
            if hbm else None,
            "stacks_sat": brief("stacks_262144_sat"), "stacks_gjk_epa": brief("stacks_262144_gjk_epa"),
            "boxes_pile_sat": brief("boxes_pile_262144_sat"), "mixed_gjk_epa": brief("mixed_pile_65536_gjk_epa"),
            "mixed_sat": brief("mixed_pile_65536_sat"), "joints": brief("boxes_262144_joints_65536"),
            "cpu_1_thread": (result.get("cpu_baseline") or {}).get("value"),
        }
        print(json.dumps(result), flush=True)
    barrier()
    if world_size > 1:
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
