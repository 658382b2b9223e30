"""Randomised differential cases shared by scripts/fuzz_pinned.py, scripts/fuzz_contacts.py (long runs on a GPU box) and
tests/test_gpu_fuzz.py (a reduced, fixed-seed selection inside `pytest -m gpu`): the GPU through the C ABI against the
CPU oracle, every bit of every body (and, on the pinned path, the contact masks of every substep)."""
import time

import numpy as np

import oracle_binding as ob
from constraint_solver_amd import capi

POLY = {capi.SCENE_BOXES_DROP: [("cube", 1.0)],
        capi.SCENE_MIXED_DROP: [("cube", 1.0), ("tetrahedron", 0.5), ("icosahedron", 0.5)]}


def bits_or_nan_equal(got, want):
    if np.array_equal(got.view(np.uint64), want.view(np.uint64)):
        return True
    return np.array_equal(np.isnan(got), np.isnan(want)) and \
        np.array_equal(got[~np.isnan(got)].view(np.uint64), want[~np.isnan(want)].view(np.uint64))


def pinned_case(rng, max_bodies=6000):
    """One random case of the PINNED path (solver::step semantics).  Returns (ok, description)."""
    kind = int(rng.integers(4))
    n = int(rng.integers(1, max_bodies))
    substeps = int(rng.integers(1, 40))
    frames = int(rng.integers(1, 8))
    dt = float(rng.choice([1 / 60, 1 / 30, 1 / 120, 0.01]))
    mode = [capi.MODE_FUSED, capi.MODE_PER_SUBSTEP][int(rng.integers(2))]
    block = int(rng.choice([0, 64, 128, 256]))
    seed = int(rng.integers(1 << 30))
    bodies, sid = capi.scene_generate(kind, seed, n)
    bodies[:, 22:28] *= float(rng.uniform(0.0, 3.0))
    bodies[:, 33] += float(rng.uniform(-0.4, 0.4))
    verts, off = capi.scene_shapes(kind)
    want, want_masks = bodies, []
    for _ in range(frames):
        want, m = ob.step_bodies(want, sid, verts, off, dt, substeps, want_masks=True, threads=8)
        want_masks.append(m)
    ok, contacts = True, 0
    with capi.World(mode=mode, block_size=block, trace_contacts=True) as w:
        w.set_shapes(verts, off)
        w.upload(bodies, sid)
        for f in range(frames):
            w.step(dt, substeps)
            ok &= np.array_equal(w.contact_masks(substeps), want_masks[f])
            contacts += int(np.count_nonzero(want_masks[f]))
        got = w.download()
    ok &= np.array_equal(got.view(np.uint64), want.view(np.uint64))
    return bool(ok), ("kind %d n %4d substeps %2d frames %d dt %.5f mode %d block %3d contacts %d"
                      % (kind, n, substeps, frames, dt, mode, block, contacts))


def contacts_case(rng, max_bodies=2500):
    """One random case of the contact pipeline (EXTENSION).  Returns (ok, description)."""
    kind = [capi.SCENE_BOXES_DROP, capi.SCENE_MIXED_DROP][int(rng.integers(2))]
    n = int(rng.integers(40, max_bodies))
    substeps = int(rng.integers(1, 12))
    frames = int(rng.integers(1, 6))
    narrowphase = int(rng.integers(2))
    schedule = int(rng.integers(3))
    pad = float(rng.choice([0.0, 0.02, 0.1]))
    width = float(rng.uniform(1.5, 12.0))
    seed = int(rng.integers(1 << 30))
    bodies, sid = capi.scene_generate(kind, seed, n)
    r2 = np.random.default_rng(seed)
    bodies[:, 31:33] = r2.uniform(0, width, (n, 2))
    bodies[:, 33] = r2.uniform(0.3, 7.0, n)
    bodies[:, 22:25] *= float(rng.uniform(0.0, 2.0))
    joints = np.zeros(0, dtype=capi.JOINT_DTYPE)
    if rng.random() < 0.3 and n > 10:
        k = int(rng.integers(1, n // 3))
        a = r2.choice(n - 1, size=k, replace=False).astype(np.uint32)
        joints = np.zeros(k, dtype=capi.JOINT_DTYPE)
        joints["body_a"], joints["body_b"] = a, a + 1
        joints["anchor_a"], joints["anchor_b"], joints["distance"] = [0.5, 0.5, 0.5], [0.5, 0.5, 0.5], float(rng.uniform(0.5, 2.0))
        if rng.random() < 0.5:                                         # round 3: some of them hinges (ball joint + axis alignment)
            hinge = r2.random(k) < 0.5
            axes = r2.normal(size=(2, k, 3))
            axes /= np.linalg.norm(axes, axis=2, keepdims=True)
            joints["kind"][hinge], joints["distance"][hinge] = capi.JOINT_HINGE, 0.0
            joints["axis_a"][hinge], joints["axis_b"][hinge] = axes[0][hinge], axes[1][hinge]
    limit = float(rng.choice([0.0, 0.0, 1.0, 3.0]))                      # round 3: xpbd_world_set_max_depenetration_speed
    polys = ob.polytopes_array(POLY[kind])
    t0 = time.time()
    want = bodies
    for _ in range(frames):
        want = ob.contacts_step_joints(want, sid, polys, joints, 1 / 60, substeps, pad, narrowphase=narrowphase, max_depenetration_speed=limit)
    with capi.World(mode=capi.MODE_CONTACTS) as w:
        w.set_polytopes(capi.scene_polytopes(kind))
        w.set_contact_pad(pad)
        w.set_narrowphase(narrowphase)
        w.set_sat_schedule(schedule)
        w.set_max_depenetration_speed(limit)
        w.upload(bodies, sid)
        if len(joints):
            w.set_joints(joints)
        for _ in range(frames):
            w.step(1 / 60, substeps)
        got = w.download()
        _, touching, _ = w.contact_stats()
    hinges = int((joints["kind"] == capi.JOINT_HINGE).sum()) if len(joints) else 0
    return bits_or_nan_equal(got, want), ("kind %d n %4d substeps %2d frames %d narrowphase %d schedule %d pad %.2f width %5.2f "
                                          "joints %3d (%d hinges) limit %.0f touching %d (%.1f s)"
                                          % (kind, n, substeps, frames, narrowphase, schedule, pad, width, len(joints), hinges, limit, touching,
                                             time.time() - t0))
