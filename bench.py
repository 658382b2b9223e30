#!/usr/bin/env python3
"""bench.py -- body\u00b7substeps/s of the XPBD stepper hot path on N MI355X GPUs of one node.

A "step" is one frame: xpbd_world_step(dt = 1/60, substeps) over the rank's resident
bodies, i.e. for every body `solver::step(body, shape, dt, substeps)` (reference
src/solver.rs:3-17).  Workload at N = 1: BASELINE.json's metric configuration, 262 144
rigid bodies (unit boxes) x 20 substeps/frame, bodies already resident in HBM (SoA).

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL, used only for the
barrier and the MAX-over-ranks of the wall time).  Bodies never interact in the
reference, so the world is sharded by contiguous body-index range with NO data-path
collective; scaling is weak (every rank steps --bodies bodies).

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel (k_step) at the
algorithmic 412 B per body per launch against the 8 TB/s HBM peak;
`cpu_baseline` times the CPU oracle (C restatement of the reference, single thread like
the reference) on a bounded sample of the same bodies on this node's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json's metric, verbatim (the file travels with the repo; the constant is the fallback)
METRIC = "body\u00b7substeps/sec at 262k rigid bodies, 20 substeps/frame; 1/2/4/8 GPU"
try:
    with open(os.path.join(ROOT, "BASELINE.json")) as _f:
        METRIC = json.load(_f).get("metric", METRIC)
except (OSError, ValueError):
    pass
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FRAME_TIME = 1.0 / 60.0  # reference src/app.rs:15


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


def cpu_baseline(state, shape_id, verts, offsets, substeps, budget_s=12.0, sample=8192):
    """Times the CPU oracle on the first `sample` bodies of the GPU's current state."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    n = min(sample, state.shape[0])
    bodies, sid = state[:n].copy(), shape_id[:n].copy()
    t0 = time.perf_counter()
    bodies, _ = ob.step_bodies(bodies, sid, verts, offsets, FRAME_TIME, substeps)   # calibration frame (also timed)
    one = time.perf_counter() - t0
    frames = int(max(2, min(400, budget_s / max(one, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(frames):
        bodies, _ = ob.step_bodies(bodies, sid, verts, offsets, FRAME_TIME, substeps)
    sec = time.perf_counter() - t0
    out = {"value": n * substeps * frames / sec, "unit": "body\u00b7substeps/s", "cores": 1, "kind": "port",
           "sample": "first %d bodies of the benchmark state after warmup, %d frames x %d substeps, %.1f s, "
                     "oracle/xpbd_oracle.c (C restatement of the reference, gcc -O2 -ffp-contract=off), 1 thread "
                     "like the single-threaded reference" % (n, frames, substeps, sec)}
    # all host cores, for information (the reference itself is single-threaded)
    cores = min(len(os.sched_getaffinity(0)), 16)        # a 1-GPU box's CPU share is 16 cores
    if cores > 1:
        f2 = max(2, frames // 2)
        t0 = time.perf_counter()
        for _ in range(f2):
            bodies, _ = ob.step_bodies(bodies, sid, verts, offsets, FRAME_TIME, substeps, threads=cores)
        out["value_all_cores"] = n * substeps * f2 / (time.perf_counter() - t0)
        out["cores_all"] = cores
    return out


def run_contacts_sharded(args, capi, kind, rank, local_rank, world_size):
    """EXTENSION, N > 1: body-body contacts with the world sharded by body-index range; every rank steps
    owned + ghost bodies and the boundary bodies are exchanged after EVERY substep with one all-gather
    (RCCL over xGMI under the nccl backend).  Not in the reference; parity = sharded == single device."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from constraint_solver_amd.distributed import GpuBackend, ShardedContactWorld
    total = args.bodies * world_size
    bodies, shape_id = capi.scene_generate(kind, args.seed, total)
    polys = capi.scene_polytopes(kind)
    radius = np.array([np.linalg.norm(p["vertices"] - p["centroid"], axis=1).max() for p in polys])
    centroid = np.array([p["centroid"] for p in polys])
    backend = GpuBackend(capi, polys, 0.02, device=local_rank)
    world = ShardedContactWorld(backend, rank, world_size, bodies, shape_id, radius, centroid, pad=0.02, halo_margin=0.5,
                                order=args.order)
    for _ in range(args.warmup):
        world.step(FRAME_TIME, args.substeps)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        world.step(FRAME_TIME, args.substeps)
    torch.cuda.synchronize()
    barrier()
    wall = reduce_max_seconds(time.perf_counter() - t0)
    result = None
    if rank == 0:
        result = {
            "metric": METRIC,
            "value": total * args.substeps * args.steps / wall, "unit": "body\u00b7substeps/s", "n_gpus": world_size,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "EXTENSION body-body contacts: %d unit boxes per GPU x %d substeps/frame, scene '%s'"
                                   % (args.bodies, args.substeps, args.scene),
                       "bodies_per_gpu": args.bodies, "bodies_total": total, "substeps": args.substeps, "mode": "contacts",
                       "order": args.order,
                       "sharding": "body-index ranges + ghost bodies; halo all-gather after every substep (%s)"
                                   % dist.get_backend(),
                       "halo_bodies_rank0": int(len(world.plan.ghosts[0])), "boundary_capacity": int(world.plan.capacity),
                       "extension": "not in the reference (parity unpinned; sharded == single device bit for bit)"},
            "roofline": None, "cpu_baseline": None,
        }
        print(json.dumps(result), flush=True)
    barrier()
    backend.close()
    dist.destroy_process_group()
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--bodies", type=int, default=262144, help="bodies per GPU (weak scaling)")
    ap.add_argument("--substeps", type=int, default=20)
    ap.add_argument("--scene", default="boxes-drop", choices=["boxes", "mixed", "boxes-drop", "mixed-drop", "stacks"])
    ap.add_argument("--mode", default="fused", choices=["fused", "substep", "contacts"],
                    help="contacts = EXTENSION (body-body contacts; not in the reference, parity unpinned)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--block-size", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the informational extra measurements (unfused roofline, PCIe-inclusive); for profiling runs")
    ap.add_argument("--narrowphase", default="sat", choices=["sat", "gjk"],
                    help="contacts mode: SAT (up to 8 points per pair) or GJK + EPA (one point per pair)")
    ap.add_argument("--joints", type=int, default=0,
                    help="contacts mode: link bodies into chains of 5 along x with this many distance joints (4 per chain)")
    ap.add_argument("--order", default="index", choices=["index", "spatial"],
                    help="contacts mode, N > 1: shard the caller's index ranges, or renumber bodies by grid cell first")
    ap.add_argument("--pitch", type=float, default=2.0,
                    help="grid pitch of the scene in metres (generator default 2.0); < 2 packs bodies so that they collide")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the barrier/MAX reduction (gloo: rehearsal only)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)")
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

    kind = {"boxes": capi.SCENE_BOXES, "mixed": capi.SCENE_MIXED, "boxes-drop": capi.SCENE_BOXES_DROP,
            "mixed-drop": capi.SCENE_MIXED_DROP, "stacks": capi.SCENE_BOX_STACKS}[args.scene]
    mode = {"fused": capi.MODE_FUSED, "substep": capi.MODE_PER_SUBSTEP, "contacts": capi.MODE_CONTACTS}[args.mode]
    total = args.bodies * world_size
    first, count = shard_range(total, rank, world_size)
    verts, offsets = capi.scene_shapes(kind)
    bodies, shape_id = capi.scene_generate(kind, args.seed, total, first=first, count=count)
    if args.pitch != 2.0:
        bodies[:, 31:33] *= args.pitch / 2.0

    if mode == capi.MODE_CONTACTS and world_size > 1:
        return run_contacts_sharded(args, capi, kind, rank, local_rank, world_size)

    world = capi.World(device=local_rank, mode=mode, block_size=args.block_size)
    if mode == capi.MODE_CONTACTS:
        world.set_polytopes(capi.scene_polytopes(kind))
        world.set_narrowphase(capi.NARROWPHASE_GJK_EPA if args.narrowphase == "gjk" else capi.NARROWPHASE_SAT)
    else:
        world.set_shapes(verts, offsets)
    world.upload(bodies, shape_id)                       # inputs resident in HBM before any timing
    if args.joints and mode == capi.MODE_CONTACTS:
        # BASELINE configs[4] (extension): chains of 5 bodies, 4 distance joints each, centre to centre at the pitch
        k = np.arange(args.joints)
        a = (k // 4) * 5 + (k % 4)
        a = a[a + 1 < count]
        joints = np.zeros(len(a), dtype=capi.JOINT_DTYPE)
        joints["body_a"], joints["body_b"] = a, a + 1
        joints["anchor_a"], joints["anchor_b"], joints["distance"] = [0.5, 0.5, 0.5], [0.5, 0.5, 0.5], args.pitch
        world.set_joints(joints)
    # Run on an explicit torch stream so torch.cuda.Event (HIP events) brackets OUR launches; the
    # default stream's handle is 0, which the ABI reads as "use the world's own stream".
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    world.set_stream(stream.cuda_stream)

    for _ in range(args.warmup):
        world.step(FRAME_TIME, args.substeps)
    torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        world.step(FRAME_TIME, args.substeps)
    ev1.record(stream)
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    wall = reduce_max_seconds(wall)
    device_ms = ev0.elapsed_time(ev1)                    # HIP events on the kernels' stream

    result = None
    if rank == 0:
        launches_per_step = 1 if mode == capi.MODE_FUSED else args.substeps   # contacts: 3 kernels per substep
        launch_s = device_ms * 1e-3 / (args.steps * launches_per_step)
        bytes_per_launch = capi.BYTES_PER_BODY_SUBSTEP * count   # 412 B x bodies, fused or not
        achieved = bytes_per_launch / launch_s / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):
            key = "%s_%d" % (args.mode, count)
            traffic = json.load(open(tfile)).get(key, {}).get("bytes_per_launch")
        result = {
            "metric": METRIC,
            "value": total * args.substeps * args.steps / wall,
            "unit": "body\u00b7substeps/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d %s per GPU x %d substeps/frame, dt=1/60, ground contacts "
                                   "(reference semantics), seeded scene '%s'"
                                   % (args.bodies, "mixed convex polyhedra" if "mixed" in args.scene else "unit boxes",
                                      args.substeps, args.scene),
                       "bodies_per_gpu": args.bodies, "bodies_total": total, "substeps": args.substeps,
                       "mode": args.mode, "sharding": "contiguous body-index ranges, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "k_step", "launch_us": launch_s * 1e6, "bytes_per_launch": bytes_per_launch,
                         "note": ("fused: all %d substeps of a body run in registers, so one launch moves 412 B/body "
                                  "once and the kernel is f64-VALU bound, not HBM bound" % args.substeps)
                         if mode == capi.MODE_FUSED else "one launch per substep: state round-trips HBM every substep"},
        }
        result["config"]["ground_contacts_per_body_after_run"] = len(world.contacts()) / max(count, 1)
        if mode == capi.MODE_CONTACTS:
            pairs, touching, points = world.contact_stats()
            all_substeps = max((args.steps + args.warmup) * args.substeps, 1)
            touching_per_substep, points_per_substep = touching / all_substeps, points / all_substeps
            result["config"]["extension"] = "body-body contacts: NOT in the reference (parity unpinned)"
            result["config"]["joints"] = args.joints
            result["config"]["narrowphase"] = args.narrowphase
            result["config"]["neighbour_pairs"] = pairs
            result["config"]["touching_pairs_per_substep"] = touching_per_substep
            result["config"]["manifold_points_per_substep"] = points_per_substep
            # Algorithmic bytes of ONE SUBSTEP of the pipeline (narrowphase + the fused per-body kernel), DESIGN.md 8:
            # per body 716 B (state 13 + 25 doubles, frames in 14 + out 17 doubles, state out 13, shape id), per listed
            # pair 120 B (two frames, the pair), per touching pair its 24-byte header written once and read by both
            # bodies, per contact point 48 B written once and read twice.
            substep_bytes = count * 716 + pairs * 120 + touching_per_substep * 72 + points_per_substep * 144
            result["roofline"].update({
                "achieved": substep_bytes / launch_s / 1e9, "frac": substep_bytes / launch_s / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                "kernel": "one substep: narrowphase + k_pair_solve_integrate_ground", "bytes_per_launch": substep_bytes,
                "note": "whole-substep figure (the SAT is f64-VALU / latency bound, the per-body kernel bandwidth bound; "
                        "per-kernel times in profiles/r01_o_contacts_*_kernel_stats.csv)"})
        if world_size == 1 and mode == capi.MODE_FUSED and not args.no_extras:
            # The same kernel scheduled one launch per substep (state round-trips HBM every substep):
            # the HBM-roofline-comparable form, measured in the same run on the same resident state.
            frames = max(3, min(10, args.steps))
            world.set_mode(capi.MODE_PER_SUBSTEP)
            for _ in range(2):
                world.step(FRAME_TIME, args.substeps)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(frames):
                world.step(FRAME_TIME, args.substeps)
            e1.record(stream)
            torch.cuda.synchronize()
            world.set_mode(capi.MODE_FUSED)
            sub_s = e0.elapsed_time(e1) * 1e-3 / (frames * args.substeps)
            tr = None
            if os.path.exists(tfile):
                tr = json.load(open(tfile)).get("substep_%d" % count, {}).get("bytes_per_launch")
            result["roofline_unfused"] = {"bound": "hbm", "achieved": bytes_per_launch / sub_s / 1e9, "peak": HBM_PEAK_GBPS,
                                          "unit": "GB/s", "frac": bytes_per_launch / sub_s / 1e9 / HBM_PEAK_GBPS,
                                          "traffic": tr, "kernel": "k_step", "launch_us": sub_s * 1e6,
                                          "body_substeps_per_s": count / sub_s,
                                          "note": "XPBD_MODE_PER_SUBSTEP: one launch per substep, 412 B per body per launch"}
        if world_size == 1 and not args.no_extras:
            # Informational, never `value`: the same frame when the boundary hands over HOST buffers
            # (AoS upload over PCIe -> step -> AoS download), as a literal per-frame drop-in would.
            state = world.download()
            t0 = time.perf_counter()
            for _ in range(3):
                world.upload(state, shape_id)
                world.step(FRAME_TIME, args.substeps)
                state = world.download()
            per_frame = (time.perf_counter() - t0) / 3
            result["pcie_inclusive"] = {"value": count * args.substeps / per_frame, "unit": "body\u00b7substeps/s",
                                        "ms_per_frame": per_frame * 1e3,
                                        "what": "pageable host AoS upload + step + download every frame"}
            # What the reference's app actually needs per frame: state stays resident, only Rigid::frame() of every
            # body comes back for rendering (src/app.rs:227-232).
            t0 = time.perf_counter()
            for _ in range(5):
                world.step(FRAME_TIME, args.substeps)
                world.frames()
            per_frame = (time.perf_counter() - t0) / 5
            result["render_readback"] = {"value": count * args.substeps / per_frame, "unit": "body\u00b7substeps/s",
                                         "ms_per_frame": per_frame * 1e3,
                                         "what": "step + download of Rigid::frame() (56 B/body) every frame"}
        if world_size == 1 and not args.no_cpu_baseline and mode != capi.MODE_CONTACTS:
            state = world.download()
            result["cpu_baseline"] = cpu_baseline(state, shape_id, verts, offsets, args.substeps)
        print(json.dumps(result), flush=True)
    barrier()
    world.close()
    if world_size > 1:
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
