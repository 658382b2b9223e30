"""Joints of the extension (oracle/xpbd_pairs_oracle.c; the reference has no joint type): physical sanity
of the oracle on CPU.  GPU == oracle and sharded == single are asserted in test_gpu_pairs.py / test_halo_gloo.py."""
import numpy as np

import oracle_binding as ob
from constraint_solver_amd import capi

DT = 1.0 / 60.0
POLYS = ob.polytopes_array([("cube", 1.0)])


def make_bodies(n):
    bodies, sid = capi.scene_generate(capi.SCENE_BOXES, 1, n)
    bodies[:, 34:38] = [1.0, 0.0, 0.0, 0.0]
    bodies[:, 22:28] = 0.0
    return bodies, sid


def joint(a, b, anchor_a, anchor_b, distance):
    j = np.zeros(1, dtype=capi.JOINT_DTYPE)
    j["body_a"], j["body_b"], j["anchor_a"], j["anchor_b"], j["distance"] = a, b, anchor_a, anchor_b, distance
    return j


def test_pendulum_keeps_its_length_and_swings_down():
    bodies, sid = make_bodies(2)
    bodies[0, 0] = 0.0                       # static anchor body: infinite mass and inertia, no gravity
    bodies[0, 1:10] = 0.0
    bodies[0, 10:13] = 0.0
    bodies[0, 31:34] = [0.0, 0.0, 10.0]
    bodies[1, 31:34] = [2.0, 0.0, 10.0]      # hangs 2 m to the side: centre to centre
    j = joint(0, 1, [0.5, 0.5, 0.5], [0.5, 0.5, 0.5], 2.0)
    lowest = 10.0
    for _ in range(120):
        bodies = ob.contacts_step_joints(bodies, sid, POLYS, j, DT, 20, 0.02)
        d = np.linalg.norm((bodies[1, 31:34] + 0.5) - (bodies[0, 31:34] + 0.5))
        assert abs(d - 2.0) < 2e-2            # XPBD compliance 1e-6/h^2 lets it stretch a little under load
        lowest = min(lowest, bodies[1, 33])
    assert np.array_equal(bodies[0, 31:34], [0.0, 0.0, 10.0])       # the anchor never moves
    assert lowest < 8.3                                             # it swung through the bottom (10 - 2 + slack)


def test_ball_joint_holds_two_free_boxes_together():
    bodies, sid = make_bodies(2)
    bodies[0, 31:34] = [0.0, 0.0, 5.0]
    bodies[1, 31:34] = [1.2, 0.0, 5.0]
    bodies[0, 22] = -1.0                     # flying apart
    bodies[1, 22] = +1.0
    bodies[:, 10:13] = 0.0                   # no gravity: momentum must be conserved by the joint
    j = joint(0, 1, [1.0, 0.5, 0.5], [0.0, 0.5, 0.5], 0.2)          # face centres, 0.2 m apart
    for _ in range(60):
        bodies = ob.contacts_step_joints(bodies, sid, POLYS, j, DT, 20, 0.02)
    pa = bodies[0, 31:34] + [1.0, 0.5, 0.5]
    pb = bodies[1, 31:34] + [0.0, 0.5, 0.5]
    assert abs(np.linalg.norm(pb - pa) - 0.2) < 5e-3
    np.testing.assert_allclose(bodies[0, 22:25] + bodies[1, 22:25], 0.0, atol=1e-9)   # equal masses: total momentum 0


def test_coincident_ball_joint_is_skipped_not_nan():
    bodies, sid = make_bodies(2)
    bodies[:, 10:13] = 0.0
    bodies[0, 31:34] = [0.0, 0.0, 5.0]
    bodies[1, 31:34] = [1.0, 0.0, 5.0]
    j = joint(0, 1, [1.0, 0.5, 0.5], [0.0, 0.5, 0.5], 0.0)          # exactly satisfied ball joint
    out = ob.contacts_step_joints(bodies, sid, POLYS, j, DT, 20, 0.0)
    assert not np.isnan(out).any()


def hinge(a, b, anchor_a, anchor_b, axis_a, axis_b):
    j = joint(a, b, anchor_a, anchor_b, 0.0)
    j["axis_a"], j["axis_b"], j["kind"] = axis_a, axis_b, capi.JOINT_HINGE
    return j


def rotate(q, v):                                                    # q = (s, x, y, z), cgmath's q * v
    s, u = q[0], np.asarray(q[1:])
    t = np.cross(u, v) + np.asarray(v) * s
    return np.cross(u, t) * 2 + np.asarray(v)


def test_hinge_keeps_the_axes_aligned_and_leaves_the_rotation_about_them_free():
    """A door on a static post: ball joint on the hinge line + the angular term (XPBD_JOINT_HINGE).  The door is spun about the
    hinge axis (free) AND about a perpendicular axis (resisted): afterwards its axis is still the post's, and it has turned
    about it.  The same door on a plain ball joint tumbles."""
    bodies, sid = make_bodies(2)
    bodies[:, 10:13] = 0.0                                           # no gravity
    bodies[0, 0:10] = 0.0                                            # the post: infinite mass and inertia
    bodies[0, 31:34] = [0.0, 0.0, 5.0]
    bodies[1, 31:34] = [1.6, 0.0, 5.0]                               # the door: a cube 0.6 m away (no contact), hinged on the line x = 1.3, y = 0
    bodies[1, 25:28] = [1.5, 0.0, 3.0]                               # spin: 3 rad/s about the hinge axis z, 1.5 rad/s about x
    z = [0.0, 0.0, 1.0]
    results = {}
    for name, j in (("hinge", hinge(0, 1, [1.3, 0.0, 0.5], [-0.3, 0.0, 0.5], z, z)), ("ball", joint(0, 1, [1.3, 0.0, 0.5], [-0.3, 0.0, 0.5], 0.0))):
        b = bodies.copy()
        for _ in range(60):
            b = ob.contacts_step_joints(b, sid, POLYS, j, DT, 20, 0.02)
        assert not np.isnan(b).any()
        results[name] = b
    door = results["hinge"][1]
    axis = rotate(door[34:38], z)
    assert np.degrees(np.arccos(np.clip(axis[2], -1, 1))) < 2.0      # still (nearly) the post's axis
    x_axis = rotate(door[34:38], [1.0, 0.0, 0.0])
    assert abs(np.degrees(np.arctan2(x_axis[1], x_axis[0]))) > 5.0   # ... and it did swing about it
    frame_origin = np.array(capi.rigid_frame(door)[:3])
    anchor = frame_origin + rotate(door[34:38], [-0.3, 0.0, 0.5])
    assert np.linalg.norm(anchor - [1.3, 0.0, 5.5]) < 2e-2           # the hinge point stays where the post holds it
    tumbled = rotate(results["ball"][1][34:38], z)
    assert np.degrees(np.arccos(np.clip(tumbled[2], -1, 1))) > 20.0  # without the angular term the axis wanders off
    assert np.array_equal(results["hinge"][0], bodies[0])            # the post never moves


def test_hinge_between_two_free_bodies_conserves_angular_momentum_about_the_axis_direction():
    bodies, sid = make_bodies(2)
    bodies[:, 10:13] = 0.0
    bodies[0, 31:34] = [0.0, 0.0, 5.0]
    bodies[1, 31:34] = [1.0, 0.0, 5.0]
    bodies[1, 34:38] = [np.cos(0.15), np.sin(0.15), 0.0, 0.0]        # body 1 tilted 0.3 rad about x: axes misaligned
    z = [0.0, 0.0, 1.0]
    j = hinge(0, 1, [1.0, 0.5, 0.5], [0.0, 0.5, 0.5], z, z)
    b = bodies
    for _ in range(90):
        b = ob.contacts_step_joints(b, sid, POLYS, j, DT, 20, 0.02)
    a0, a1 = rotate(b[0, 34:38], z), rotate(b[1, 34:38], z)
    assert np.degrees(np.arccos(np.clip(np.dot(a0, a1), -1, 1))) < 1.0                     # the two axes found each other
    np.testing.assert_allclose(b[0, 22:25] + b[1, 22:25], 0.0, atol=1e-9)                  # equal masses: no net momentum appears


def test_max_depenetration_speed_limits_how_fast_overlapping_boxes_fly_apart():
    """Two cubes overlapping by 0.2 m, at rest, no gravity: the reference's loop (src/solver.rs:19-27) resolves the overlap in
    ONE substep, i.e. at depth / h; with the limit the bodies part at the limit."""
    bodies, sid = make_bodies(2)
    bodies[:, 10:13] = 0.0
    bodies[0, 31:34] = [0.0, 0.0, 5.0]
    bodies[1, 31:34] = [0.8, 0.0, 5.0]
    none = np.zeros(0, dtype=capi.JOINT_DTYPE)
    free = ob.contacts_step_joints(bodies, sid, POLYS, none, DT, 20, 0.02)
    limited = ob.contacts_step_joints(bodies, sid, POLYS, none, DT, 20, 0.02, max_depenetration_speed=2.0)
    v_free = np.abs(free[:, 22]).max()
    assert v_free > 20.0                                             # 0.1 m each within one substep of 1/1200 s: ~100 m/s
    b, speeds = bodies, []
    for _ in range(10):
        b = ob.contacts_step_joints(b, sid, POLYS, none, DT, 20, 0.02, max_depenetration_speed=2.0)
        speeds.append(np.abs(b[:, 22]).max())
    assert max(speeds) < 2.0 * 1.05 and speeds[0] > 0.5              # each body recedes at <= half of 2 x the limit ... they share it
    assert b[1, 31] - b[0, 31] > 0.85                                # ... and the overlap does close, frame by frame
    assert np.array_equal(ob.contacts_step_joints(bodies, sid, POLYS, none, DT, 20, 0.02, max_depenetration_speed=0.0), free)
    assert limited[1, 31] - limited[0, 31] < free[1, 31] - free[0, 31]
