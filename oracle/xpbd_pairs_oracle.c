/*
 * xpbd_pairs_oracle.c -- CPU ORACLE for the body-body contact extension.
 * Test infrastructure only; PARITY UNPINNED (see xpbd_pairs_oracle.h).
 */
#include "xpbd_pairs_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

static o_vec3 madd(o_vec3 a, o_vec3 d, double t) { return o_add(a, o_scale(d, t)); }

static double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

/* World-space polygon of face `f` of polytope p under frame fr. */
static uint32_t face_polygon(const o_polytope *p, o_frame fr, uint32_t f, o_vec3 *out)
{
    uint32_t n = p->face_offsets[f + 1] - p->face_offsets[f];
    for (uint32_t k = 0; k < n; k++)
        out[k] = o_frame_mulv(fr, p->vertices[p->face_indices[p->face_offsets[f] + k]]);
    return n;
}

/*
 * Face contact: reference face `rf` of body R (frame fr), incident body I (frame fi).
 *  - reference plane in world space: fr * plane(rf)                    (src/collision.rs:66)
 *  - incident face: least normal . ref_normal, first minimum          (src/collision.rs:76-85)
 *  - clip the incident polygon against the side planes of the reference face
 *    (Sutherland-Hodgman; side normal = edge x n, flipped so the face's own far vertex is inside)
 *  - every clipped point strictly below the reference plane is a contact; its partner on the
 *    reference body is Plane::project (src/geometry.rs:45-47).
 */
static void face_contact(o_frame fr, const o_polytope *pr, uint32_t rf, o_frame fi, const o_polytope *pi,
                         uint32_t *incident_face, op_manifold *out)
{
    o_plane ref_plane = o_frame_mulplane(fr, o_polytope_plane(pr, rf));

    double least_dot = DBL_MAX;
    uint32_t inc = 0;
    for (uint32_t i = 0; i < pi->n_faces; i++) {
        o_plane pl = o_frame_mulplane(fi, o_polytope_plane(pi, i));
        double d = o_dot(pl.normal, ref_plane.normal);
        if (d < least_dot) {
            least_dot = d;
            inc = i;
        }
    }
    *incident_face = inc;

    o_vec3 ref_poly[O_MAX_VERTS], poly[2][16];
    uint32_t nr = face_polygon(pr, fr, rf, ref_poly);
    uint32_t np = face_polygon(pi, fi, inc, poly[0]);
    int cur = 0;
    for (uint32_t k = 0; k < nr && np > 0; k++) {
        o_vec3 a = ref_poly[k], b = ref_poly[(k + 1) % nr], c = ref_poly[(k + 2) % nr];
        o_vec3 s = o_cross(o_sub(b, a), ref_plane.normal);
        if (o_dot(s, o_sub(c, a)) > 0.0)
            s = o_neg(s);
        const o_vec3 *in = poly[cur];
        o_vec3 *dst = poly[cur ^ 1];
        uint32_t nd = 0;
        for (uint32_t m = 0; m < np; m++) {
            o_vec3 p0 = in[m], p1 = in[(m + 1) % np];
            double d0 = o_dot(s, o_sub(p0, a)), d1 = o_dot(s, o_sub(p1, a));
            int in0 = d0 <= 0.0, in1 = d1 <= 0.0;
            if (in0 && nd < 16)
                dst[nd++] = p0;
            if (in0 != in1 && nd < 16)
                dst[nd++] = madd(p0, o_sub(p1, p0), d0 / (d0 - d1));
        }
        np = nd;
        cur ^= 1;
    }

    out->n_points = 0;
    for (uint32_t m = 0; m < np && out->n_points < OP_MAX_POINTS; m++) {
        o_vec3 p = poly[cur][m];
        double d = o_plane_distance(ref_plane, p);
        if (d >= 0.0)
            continue;
        out->p_inc[out->n_points] = p;
        out->p_ref[out->n_points] = o_sub(p, o_lscale(d, ref_plane.normal)); /* Plane::project */
        out->n_points++;
    }
}

/* Edge-edge contact: closest points of the two (non-parallel) world-space segments. */
static void edge_contact(o_frame fa, const o_polytope *pa, uint32_t ea, o_frame fb, const o_polytope *pb,
                         uint32_t eb, op_manifold *out)
{
    o_vec3 a0 = o_frame_mulv(fa, pa->vertices[pa->edges[ea][0]]), a1 = o_frame_mulv(fa, pa->vertices[pa->edges[ea][1]]);
    o_vec3 b0 = o_frame_mulv(fb, pb->vertices[pb->edges[eb][0]]), b1 = o_frame_mulv(fb, pb->vertices[pb->edges[eb][1]]);
    o_vec3 d1 = o_sub(a1, a0), d2 = o_sub(b1, b0), r = o_sub(a0, b0);
    double a = o_dot(d1, d1), e = o_dot(d2, d2), f = o_dot(d2, r), c = o_dot(d1, r), b = o_dot(d1, d2);
    double denom = a * e - b * b;
    double s = clamp01((b * f - c * e) / denom);
    double t = (b * s + f) / e;
    if (t < 0.0) {
        t = 0.0;
        s = clamp01(-c / a);
    } else if (t > 1.0) {
        t = 1.0;
        s = clamp01((b - c) / a);
    }
    out->n_points = 1;
    out->p_ref[0] = madd(a0, d1, s); /* on A (reference body of the edge case) */
    out->p_inc[0] = madd(b0, d2, t); /* on B */
}

/*
 * Edge query of the extension.  Same structure, axis orientation, NaN behaviour and first-maximum
 * rule as edge_axes_separation (src/collision.rs:151-197, restated literally in
 * o_edge_axes_separation), with two robustness changes the dead reference code lacks:
 *   - its "another point on `a` is further" test gets a tolerance (OP_SUPPORT_TOL): without one the
 *     edge's own second endpoint beats the foot by rounding noise about half of the time;
 *   - the mirrored test is applied to B's edge, so both edges are supporting features and the
 *     closest points of the two segments really are the contact (the reference only constrains A).
 */
double op_edge_query(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb, uint64_t *edge_a,
                     uint64_t *edge_b)
{
    double max_distance = -DBL_MAX;
    *edge_a = UINT64_MAX;
    *edge_b = UINT64_MAX;
    o_vec3 wa[O_MAX_VERTS], wb[O_MAX_VERTS];
    for (uint32_t k = 0; k < pa->n_vertices; k++)
        wa[k] = o_frame_mulv(fa, pa->vertices[k]);
    for (uint32_t k = 0; k < pb->n_vertices; k++)
        wb[k] = o_frame_mulv(fb, pb->vertices[k]);
    o_vec3 centroid_a = o_frame_mulv(fa, pa->centroid);
    for (uint32_t ie = 0; ie < pa->n_edges; ie++) {
        for (uint32_t je = 0; je < pb->n_edges; je++) {
            o_vec3 foot = wa[pa->edges[ie][0]];
            o_vec3 e0 = o_sub(wa[pa->edges[ie][1]], foot);
            o_vec3 b0 = wb[pb->edges[je][0]];
            o_vec3 e1 = o_sub(wb[pb->edges[je][1]], b0);
            o_vec3 axis = o_normalize(o_cross(e0, e1));
            if (!(fabs(axis.x) <= DBL_MAX && fabs(axis.y) <= DBL_MAX && fabs(axis.z) <= DBL_MAX))
                continue; /* parallel edges: NaN axis, contributes nothing (as in the reference) */
            if (o_dot(axis, o_sub(foot, centroid_a)) < 0.0)
                axis = o_neg(axis);
            double reach = o_dot(wa[0], axis);
            for (uint32_t k = 1; k < pa->n_vertices; k++) {
                double r = o_dot(wa[k], axis);
                if (r > reach)
                    reach = r;
            }
            if (reach > o_dot(foot, axis) + OP_SUPPORT_TOL)
                continue;
            o_vec3 nax = o_neg(axis);
            o_vec3 sup = wb[0];
            double breach = o_dot(sup, nax);
            for (uint32_t k = 1; k < pb->n_vertices; k++) {
                double r = o_dot(wb[k], nax);
                if (r >= breach) { /* last maximum, as Polytope::support */
                    breach = r;
                    sup = wb[k];
                }
            }
            if (breach > o_dot(b0, nax) + OP_SUPPORT_TOL)
                continue;
            o_plane plane = o_plane_from_point_normal(foot, axis);
            double distance = o_plane_distance(plane, sup);
            if (distance > max_distance) {
                max_distance = distance;
                *edge_a = ie;
                *edge_b = je;
            }
        }
    }
    return max_distance;
}

void op_sat(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb, op_manifold *out)
{
    uint64_t face_a, face_b, edge_a, edge_b;
    memset(out, 0, sizeof *out);
    out->separated = 1;
    if (pa->n_vertices == 0 || pb->n_vertices == 0 || pa->n_faces == 0 || pb->n_faces == 0)
        return; /* the reference's .unwrap() / face index would panic; the extension reports "no contact" */

    /* src/collision.rs:42-45 */
    double qa = o_face_axes_separation(fa, fb, pa, pb, &face_a);
    out->query[0] = qa;
    if (qa >= 0.0 || qa != qa)
        return;
    /* src/collision.rs:47-50 (commented there): the same query with the roles swapped */
    double qb = o_face_axes_separation(fb, fa, pb, pa, &face_b);
    out->query[1] = qb;
    if (qb >= 0.0 || qb != qb)
        return;
    /* src/collision.rs:52-55 (commented there) */
    double qe = op_edge_query(fa, fb, pa, pb, &edge_a, &edge_b);
    out->query[2] = qe;
    if (qe >= 0.0)
        return;
    if (face_a == UINT64_MAX || face_b == UINT64_MAX)
        return; /* a body without faces: the reference would index out of bounds */

    /* src/collision.rs:57-59, 89-92 (commented there): reference face on A if a is the maximum, else
     * on B, else the edge pair.  Extension decision: the edge pair must beat both faces by
     * OP_EDGE_BIAS, otherwise two stacked boxes flip between a 4-point face manifold and a single
     * edge point on rounding noise (their edge axes coincide with face normals). */
    double m = qa > qb ? qa : qb;
    int use_edges = edge_a != UINT64_MAX && qe > m + OP_EDGE_BIAS;
    if (use_edges)
        m = qe;
    out->separated = 0;
    out->separation = m;
    if (use_edges) {
        out->feature = OP_FEATURE_EDGES;
        out->index_a = (uint32_t)edge_a;
        out->index_b = (uint32_t)edge_b;
        edge_contact(fa, pa, (uint32_t)edge_a, fb, pb, (uint32_t)edge_b, out);
    } else if (qa == m) {
        out->feature = OP_FEATURE_FACE_A;
        out->index_a = (uint32_t)face_a;
        face_contact(fa, pa, (uint32_t)face_a, fb, pb, &out->index_b, out);
    } else {
        out->feature = OP_FEATURE_FACE_B;
        out->index_b = (uint32_t)face_b;
        face_contact(fb, pb, (uint32_t)face_b, fa, pa, &out->index_a, out);
    }
}
