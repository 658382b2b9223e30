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
 *   edge_axes_separation   src/collision.rs:151-197  (axis orientation, skip rule, NaN for parallel edges)
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
/* Slack of the "is this edge a supporting feature" tests of the edge query (m). */
#define OP_SUPPORT_TOL 1e-9

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

double op_edge_query(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb, uint64_t *edge_a,
                     uint64_t *edge_b);
void op_sat(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb, op_manifold *out);

#ifdef __cplusplus
}
#endif
#endif
