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
