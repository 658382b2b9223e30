/*
 * xpbd_pairs_oracle.h -- CPU ORACLE for the body-body contact EXTENSION
 * (test infrastructure, NOT product code).
 *
 * PARITY UNPINNED: the reference has no body-body contact path.  Its `sat`
 * (src/collision.rs:37-121) is an uncalled stub that stops after the A-face
 * query; the B-face query, the edge query, the feature choice and the contact
 * generation exist only as commented-out lines.  This file finishes that
 * sketch in the most literal way and is the build's own definition of the
 * extension's semantics; there is no reference output to compare with.
 * What IS taken from the reference (and pinned by tests/test_oracle_kat.py):
 *   face_axes_separation   src/collision.rs:123-149  (last-max support, first-max face)
 *   edge_axes_separation   src/collision.rs:151-197  (NaN for parallel edges, first maximum; the pair enumeration is
 *                          replaced by the classic test over unique edge directions, see op_edge_query)
 *   feature choice         src/collision.rs:47-59,89-92 (comments): separated if any query >= 0;
 *                          reference face on A if a == max(a, b, e), else on B if b == max, else edge-edge
 *   reference plane        src/collision.rs:66       frames.0 * polytopes.0.plane(face)
 *   incident face          src/collision.rs:76-85    least normal . ref_normal, strict '<' (first minimum)
 *   Plane::project         src/geometry.rs:45-47
 */
#ifndef XPBD_PAIRS_ORACLE_H
#define XPBD_PAIRS_ORACLE_H

#include "xpbd_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

#define OP_MAX_POINTS 8
/* The edge-edge axis is used only if it separates better than both face axes by this much (m). */
#define OP_EDGE_BIAS 1e-6

#define OP_FEATURE_FACE_A 0
#define OP_FEATURE_FACE_B 1
#define OP_FEATURE_EDGES  2

typedef struct {
    int32_t  separated;    /* 1: some axis separates the pair (or a query was NaN): no contacts */
    int32_t  feature;      /* OP_FEATURE_* */
    uint32_t index_a;      /* FACE_A: reference face on A | FACE_B: incident face on A | EDGES: edge of A */
    uint32_t index_b;      /* FACE_A: incident face on B  | FACE_B: reference face on B | EDGES: edge of B */
    double   separation;   /* max(a, b, e) < 0 when touching */
    double   query[3];     /* the three query values a, b, e (e is -DBL_MAX if no edge pair qualified) */
    uint32_t n_points;     /* <= OP_MAX_POINTS */
    o_vec3   p_ref[OP_MAX_POINTS]; /* point on the REFERENCE body's surface (world) */
    o_vec3   p_inc[OP_MAX_POINTS]; /* the penetrating point of the INCIDENT body (world) */
} op_manifold;
/* Reference body: A for FACE_A and EDGES, B for FACE_B.  Incident body: the other one. */

/* unique edge directions of a polytope (up to sign) and the direction index of every edge; returns their number */
uint32_t op_edge_directions(const o_polytope *p, o_vec3 *dirs, uint32_t *dir_of_edge);
/* separating-axis test over (unique direction of A) x (unique direction of B); returns the largest separation */
double op_edge_query(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb, uint64_t *dir_a,
                     uint64_t *dir_b);
void op_sat(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb, op_manifold *out);

/* ---------------------------------------------------------------------------
 * N-body step with body-body contacts (extension; parity unpinned).
 *
 * Once per call (broadphase):  bounding sphere of body b = (frame_b * centroid,
 *   r_shape + min(|v_b| * dt, r_shape) + pad);  j is a neighbour of b iff the spheres overlap
 *   (strict '<' on squared distances).  Neighbour lists are sorted by index.
 * Per substep (h = dt / substeps), for all bodies in lock step:
 *   1. past = pose; Rigid::integrate(h); P1 = Rigid::frame()            (src/solver.rs:7-10)
 *   2. manifold(i,j) = op_sat(P1_i, P1_j) for every neighbour pair i < j
 *   3. ground contacts from P1, solved sequentially per body              (src/solver.rs:12-13, unchanged)
 *   4. pair contacts, Jacobi: every contact point computes its XPBD correction from the
 *      poses after step 3; a body adds the corrections of all its points in (neighbour
 *      index, point index) order and applies their AVERAGE.  Each point is a two-body
 *      version of the reference Constraint (src/constraint.rs:6-37): c0 = the incident
 *      body's point, c1 = the reference body's surface point minus the relative
 *      tangential motion of the two material points during the substep (the positional
 *      friction of collision::ground, src/collision.rs:24-29), lambda = |c1 - c0| /
 *      (w_inc + w_ref + compliance), +lambda*dir on the incident body at c0 and
 *      -lambda*dir on the reference body at its surface point.
 *   5. Rigid::derive(past, h)                                              (src/solver.rs:15)
 * With no overlapping spheres this is exactly o_step_bodies.
 * ------------------------------------------------------------------------- */
typedef struct {
    uint64_t n_pairs;          /* neighbour pairs i < j of the call */
    uint64_t n_touching;       /* sum over substeps of pairs with n_points > 0 */
    uint64_t n_points;         /* sum over substeps of manifold points */
} op_contact_stats;

/* CSR neighbour lists (sorted ascending), malloc'ed; caller frees both. */
void op_broadphase(const o_rigid *bodies, const uint32_t *shape_id, uint32_t n, const o_polytope *shapes,
                   double dt, double pad, uint32_t **offsets_out, uint32_t **neighbours_out);

/* Split form for hosts that overwrite halo bodies between substeps:
 *   f = begin(bodies, .., dt, pad);  substeps x { substep(f, bodies, dt / substeps, ..); <exchange> };  end(f) */
typedef struct op_frame op_frame;
op_frame *op_contacts_begin(const o_rigid *bodies, const uint32_t *shape_id, uint32_t n, const o_polytope *shapes,
                            double dt, double pad);
uint64_t op_contacts_pair_count(const op_frame *f);
void op_contacts_substep(op_frame *f, o_rigid *bodies, double h, uint32_t *masks_row, op_contact_stats *stats);
void op_contacts_end(op_frame *f);

/* Joints (extension; the reference has no joint type, only the unused `distance` field of
 * Constraint, src/constraint.rs:9).  A joint keeps |frame_b * anchor_b - frame_a * anchor_a| at
 * `distance`; anchors are in object space (the space of the shape vertices).  distance = 0 is a ball
 * joint, a hinge is two ball joints on the axis.  Joints are projected in the Jacobi pass of step 4,
 * after the body's pair contacts, in ascending joint index, with the reference's constraint math:
 * contacts = (p_a, p_b), lambda = (|p_b - p_a| - distance) / (w_a + w_b + compliance), +lambda*dir on
 * a at p_a, -lambda*dir on b at p_b.  A joint whose points coincide exactly is skipped (the
 * reference's direction() would be NaN, cf. K6). */
/* kind OP_JOINT_HINGE adds an ANGULAR term after the positional one (its own entry in the Jacobi average): the unit axes
 * axis_a / axis_b (object space of a / b) are kept aligned.  With a_w = rot_a * axis_a, b_w = rot_b * axis_b (poses after the
 * ground contacts): delta = a_w x b_w, n = delta / |delta| (skipped when |delta| = 0: aligned, or exactly opposed),
 * w = sum over the two bodies of (I^-1 (q^-1 n)) . (q^-1 n)  -- the angular half of Constraint::inverse_resitance,
 * src/constraint.rs:25-32 --, lambda = |delta| / (w + compliance); body a turns by +lambda n, body b by -lambda n, applied as
 * Rigid::apply_impulse applies an angular displacement (src/rigid.rs:118-122): rotation += (0.5 * Quat(0, I^-1 (lambda n))) * rotation.
 * A hinge = a ball joint (distance 0) at a point of the axis + this term: one rotational degree of freedom left. */
#define OP_JOINT_DISTANCE 0u
#define OP_JOINT_HINGE    1u
typedef struct {
    uint32_t body_a, body_b;
    double anchor_a[3], anchor_b[3];
    double distance;
    double axis_a[3], axis_b[3]; /* OP_JOINT_HINGE only */
    uint32_t kind, reserved;
} op_joint;
/* Must be called between begin and the first substep; `joints` must outlive the frame. */
void op_contacts_attach_joints(op_frame *f, const op_joint *joints, uint32_t n_joints);

/* Optional limit on how fast a BODY-BODY contact may push its bodies apart (extension knob; 0 = off = the reference's solver
 * loop, src/solver.rs:19-27, which resolves any penetration within one substep, i.e. at depth / h): the length of a contact
 * point's positional correction (the `current_distance` of its constraint, friction part included) is limited to
 * max(0, speed * h - closing), closing = what the incident point has already moved towards the reference surface in this
 * substep (delta_rel . correction / |correction|), before lambda is formed: the bodies part at `speed`, they do not
 * accelerate by it.  With the knob on, the ground contacts of the contact pipeline (step 3) are limited the same way -- a
 * light body pressed into the plane z = 0 by a heavy one is otherwise shot out of the pile by the plane -- ; joints never are.
 * With the knob off (the default) nothing changes anywhere. */
void op_contacts_set_max_depenetration_speed(op_frame *f, double speed);

/* Narrowphase of step 2: the SAT above (default) or GJK + EPA (xpbd_gjk_oracle.h), which yields ONE contact
 * point per touching pair (reference body A, incident body B; a degenerate query yields no contact). */
#define OP_NARROWPHASE_SAT     0
#define OP_NARROWPHASE_GJK_EPA 1
void op_contacts_set_narrowphase(op_frame *f, int narrowphase);

void op_contacts_step(o_rigid *bodies, const uint32_t *shape_id, uint32_t n, const o_polytope *shapes,
                      double dt, uint32_t substeps, double pad, uint32_t *ground_masks, op_contact_stats *stats);

#ifdef __cplusplus
}
#endif
#endif
