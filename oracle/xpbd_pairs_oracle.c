/*
 * xpbd_pairs_oracle.c -- CPU ORACLE for the body-body contact extension.
 * Test infrastructure only; PARITY UNPINNED (see xpbd_pairs_oracle.h).
 */
#include "xpbd_pairs_oracle.h"
#include "xpbd_gjk_oracle.h"

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
 * Edge axes of the extension: the classic separating-axis test over the UNIQUE edge directions.
 * The reference's dead edge_axes_separation (src/collision.rs:151-197, restated literally in
 * o_edge_axes_separation and pinned by the KATs) enumerates all E_A x E_B edge pairs and re-derives a
 * support for each; parallel edges give the same axis again and again (a box has 12 edges but 3
 * directions).  Here every pair (unique direction of A) x (unique direction of B) gives one axis
 *     n = normalize(dA x dB), flipped to point from A's centroid towards B's,
 * and its separation is  min_B n.b - max_A n.a  (negative = overlap).  Kept from the reference: the
 * NaN behaviour for parallel directions (contributes nothing) and the first-maximum rule, now over
 * (direction of A, direction of B) in ascending order.
 */
#define OP_MAX_DIRS O_MAX_EDGES

/* Unique edge directions (up to sign): dirs[k] = v[e.1] - v[e.0] of the first edge with that direction. */
uint32_t op_edge_directions(const o_polytope *p, o_vec3 *dirs, uint32_t *dir_of_edge)
{
    uint32_t nd = 0;
    for (uint32_t e = 0; e < p->n_edges; e++) {
        o_vec3 d = o_sub(p->vertices[p->edges[e][1]], p->vertices[p->edges[e][0]]);
        uint32_t found = nd;
        for (uint32_t k = 0; k < nd; k++) {
            o_vec3 c = o_cross(d, dirs[k]);
            if (o_dot(c, c) <= 1e-12 * (o_dot(d, d) * o_dot(dirs[k], dirs[k]))) {
                found = k;
                break;
            }
        }
        if (found == nd)
            dirs[nd++] = d;
        dir_of_edge[e] = found;
    }
    return nd;
}

static int edge_axis(o_frame fa, o_frame fb, o_vec3 dir_a, o_vec3 dir_b, o_vec3 a_to_b, o_vec3 *axis)
{
    o_vec3 n = o_normalize(o_cross(o_qrot(fa.rotation, dir_a), o_qrot(fb.rotation, dir_b)));
    if (!(fabs(n.x) <= DBL_MAX && fabs(n.y) <= DBL_MAX && fabs(n.z) <= DBL_MAX))
        return 0; /* parallel directions: NaN axis, contributes nothing (as in the reference) */
    if (o_dot(n, a_to_b) < 0.0)
        n = o_neg(n);
    *axis = n;
    return 1;
}

double op_edge_query(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb, uint64_t *dir_a,
                     uint64_t *dir_b)
{
    double max_distance = -DBL_MAX;
    *dir_a = UINT64_MAX;
    *dir_b = UINT64_MAX;
    o_vec3 wa[O_MAX_VERTS], wb[O_MAX_VERTS], da[OP_MAX_DIRS], db[OP_MAX_DIRS];
    uint32_t ea[O_MAX_EDGES], eb[O_MAX_EDGES];
    for (uint32_t k = 0; k < pa->n_vertices; k++)
        wa[k] = o_frame_mulv(fa, pa->vertices[k]);
    for (uint32_t k = 0; k < pb->n_vertices; k++)
        wb[k] = o_frame_mulv(fb, pb->vertices[k]);
    uint32_t nda = op_edge_directions(pa, da, ea), ndb = op_edge_directions(pb, db, eb);
    o_vec3 a_to_b = o_sub(o_frame_mulv(fb, pb->centroid), o_frame_mulv(fa, pa->centroid));
    for (uint32_t i = 0; i < nda; i++) {
        for (uint32_t j = 0; j < ndb; j++) {
            o_vec3 axis;
            if (!edge_axis(fa, fb, da[i], db[j], a_to_b, &axis))
                continue;
            double reach_a = o_dot(wa[0], axis), reach_b = o_dot(wb[0], axis);
            for (uint32_t k = 1; k < pa->n_vertices; k++) {
                double r = o_dot(wa[k], axis);
                if (r > reach_a)
                    reach_a = r;
            }
            for (uint32_t k = 1; k < pb->n_vertices; k++) {
                double r = o_dot(wb[k], axis);
                if (r < reach_b)
                    reach_b = r;
            }
            double distance = reach_b - reach_a;
            if (distance > max_distance) {
                max_distance = distance;
                *dir_a = i;
                *dir_b = j;
            }
        }
    }
    return max_distance;
}

/* Supporting edges of the chosen edge axis: A's edge of direction dir_a furthest along the axis, B's edge of
 * direction dir_b furthest against it (sum of the two endpoint projections; first extremum). */
static void supporting_edges(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb, uint32_t dir_a,
                             uint32_t dir_b, uint32_t *edge_a, uint32_t *edge_b)
{
    o_vec3 da[OP_MAX_DIRS], db[OP_MAX_DIRS], axis;
    uint32_t ea[O_MAX_EDGES], eb[O_MAX_EDGES];
    op_edge_directions(pa, da, ea);
    op_edge_directions(pb, db, eb);
    o_vec3 a_to_b = o_sub(o_frame_mulv(fb, pb->centroid), o_frame_mulv(fa, pa->centroid));
    edge_axis(fa, fb, da[dir_a], db[dir_b], a_to_b, &axis);
    double best = -DBL_MAX;
    *edge_a = 0;
    for (uint32_t e = 0; e < pa->n_edges; e++) {
        if (ea[e] != dir_a)
            continue;
        double sproj = o_dot(o_frame_mulv(fa, pa->vertices[pa->edges[e][0]]), axis)
                     + o_dot(o_frame_mulv(fa, pa->vertices[pa->edges[e][1]]), axis);
        if (sproj > best) {
            best = sproj;
            *edge_a = e;
        }
    }
    best = DBL_MAX;
    *edge_b = 0;
    for (uint32_t e = 0; e < pb->n_edges; e++) {
        if (eb[e] != dir_b)
            continue;
        double sproj = o_dot(o_frame_mulv(fb, pb->vertices[pb->edges[e][0]]), axis)
                     + o_dot(o_frame_mulv(fb, pb->vertices[pb->edges[e][1]]), axis);
        if (sproj < best) {
            best = sproj;
            *edge_b = e;
        }
    }
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
        uint32_t sup_a, sup_b;
        supporting_edges(fa, fb, pa, pb, (uint32_t)edge_a, (uint32_t)edge_b, &sup_a, &sup_b);
        out->feature = OP_FEATURE_EDGES;
        out->index_a = sup_a;
        out->index_b = sup_b;
        edge_contact(fa, pa, sup_a, fb, pb, sup_b, out);
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

/* ======================================================================
 * N-body step with body-body contacts (semantics: xpbd_pairs_oracle.h)
 * ====================================================================== */
#include <stdlib.h>

static double shape_radius(const o_polytope *p)
{
    double r = 0.0;
    for (uint32_t k = 0; k < p->n_vertices; k++) {
        double d = o_magnitude(o_sub(p->vertices[k], p->centroid));
        if (d > r)
            r = d;
    }
    return r;
}

void op_broadphase(const o_rigid *bodies, const uint32_t *shape_id, uint32_t n, const o_polytope *shapes,
                   double dt, double pad, uint32_t **offsets_out, uint32_t **neighbours_out)
{
    o_vec3 *c = (o_vec3 *)malloc(sizeof(o_vec3) * (n ? n : 1));
    double *r = (double *)malloc(sizeof(double) * (n ? n : 1));
    for (uint32_t i = 0; i < n; i++) {
        const o_polytope *p = &shapes[shape_id ? shape_id[i] : 0];
        c[i] = o_frame_mulv(o_rigid_frame(&bodies[i]), p->centroid);
        /* the velocity inflation is clamped at the shape's own radius: a body that travels further than
         * its own size in one frame tunnels under discrete detection anyway, and an unclamped outlier
         * would blow up every grid cell of the device broadphase */
        double rs = shape_radius(p), travel = o_magnitude(bodies[i].velocity) * dt;
        r[i] = rs + (travel < rs ? travel : rs) + pad;
    }
    uint32_t *off = (uint32_t *)calloc((size_t)n + 1, sizeof(uint32_t));
    size_t cap = 16, cnt = 0;
    uint32_t *nb = (uint32_t *)malloc(cap * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; i++) {
        off[i] = (uint32_t)cnt;
        for (uint32_t j = 0; j < n; j++) {
            if (j == i)
                continue;
            o_vec3 d = o_sub(c[i], c[j]);
            double reach = r[i] + r[j];
            if (o_dot(d, d) < reach * reach) {
                if (cnt == cap) {
                    cap *= 2;
                    nb = (uint32_t *)realloc(nb, cap * sizeof(uint32_t));
                }
                nb[cnt++] = j;
            }
        }
    }
    off[n] = (uint32_t)cnt;
    free(c);
    free(r);
    *offsets_out = off;
    *neighbours_out = nb;
}

/* Generalized inverse mass of `body` for an impulse along dir at `point` (src/constraint.rs:25-32). */
static double generalized_inverse_mass(const o_rigid *body, o_vec3 point, o_vec3 dir)
{
    o_vec3 angular_impulse = o_qrot(o_qconj(body->rotation),
                                    o_cross(o_sub(point, o_add(body->position, body->center_of_mass)), dir));
    return body->inverse_mass + o_dot(o_mat3_mulv(body->inverse_inertia, angular_impulse), angular_impulse);
}

typedef struct {
    o_vec3 dpos;
    o_quat drot;
    uint32_t count;
} pair_accum;

/* One manifold point seen from body `self_is_inc ? inc : ref`. */
static void accumulate_point(int self_is_inc, const o_rigid *inc, const o_rigid *ref, o_frame inc_p1, o_frame inc_past,
                             o_frame ref_p1, o_frame ref_past, o_vec3 p_inc, o_vec3 p_ref, double compliance, double limit,
                             pair_accum *acc)
{
    o_vec3 correction = o_sub(p_ref, p_inc);
    o_vec3 delta_rel = o_sub(o_frame_delta(inc_p1, inc_past, p_inc), o_frame_delta(ref_p1, ref_past, p_ref));
    o_vec3 delta_tangential = o_sub(delta_rel, o_project_on(delta_rel, correction));
    o_vec3 c0 = p_inc;
    o_vec3 c1 = o_sub(p_ref, o_lscale(1.0, delta_tangential));
    o_vec3 difference = o_sub(c1, c0);
    double distance = o_magnitude(difference);
    o_vec3 dir = o_scale(difference, 1.0 / distance);
    double w = generalized_inverse_mass(inc, c0, dir) + generalized_inverse_mass(ref, p_ref, dir);
    double error = distance;
    if (limit > 0.0) { /* op_contacts_set_max_depenetration_speed: limit = speed * h, 0 when off */
        /* what the incident point has ALREADY moved towards the reference surface in this substep (inertia, earlier
         * corrections) counts against the allowance: the bodies part at the limit, they do not accelerate by it */
        double len = o_magnitude(correction);
        double closing = len > 0.0 ? o_dot(delta_rel, correction) / len : 0.0;
        double allowed = limit - closing;
        if (!(allowed > 0.0))
            allowed = 0.0;
        if (distance > allowed)
            error = allowed;
    }
    double lambda = (error - 0.0) / (w + compliance);

    const o_rigid *self = self_is_inc ? inc : ref;
    o_vec3 point = self_is_inc ? c0 : p_ref;
    o_vec3 impulse = self_is_inc ? o_lscale(lambda, dir) : o_lscale(-lambda, dir);
    acc->dpos = o_add(acc->dpos, o_scale(impulse, self->inverse_mass));
    o_vec3 arm = o_sub(point, o_add(self->position, self->center_of_mass));
    o_quat spin = { 0.0, o_cross(o_mat3_mulv(self->inverse_inertia, arm), impulse) };
    acc->drot = o_qadd(acc->drot, o_qmul(o_qlscale(0.5, spin), self->rotation));
    acc->count++;
}

/* Broadphase result + scratch of one frame (split form of op_contacts_step for sharded hosts). */
struct op_frame {
    uint32_t n;
    const uint32_t *shape_id;
    const o_polytope *shapes;
    uint32_t *off, *nb, *pair_first;
    uint32_t n_pairs;
    op_manifold *manifolds;
    o_vec3 *gjk_axis; /* per pair: the direction that separated it in its last GJK query (zero = none), see og_gjk_epa_cached */
    o_frame *past, *p1;
    o_vec3 *past_pos;
    o_rigid *next;
    const op_joint *joints;
    uint32_t n_joints;
    int narrowphase; /* OP_NARROWPHASE_* */
    double max_depenetration_speed; /* 0 = off */
};

void op_contacts_set_narrowphase(op_frame *f, int narrowphase) { f->narrowphase = narrowphase; }
void op_contacts_set_max_depenetration_speed(op_frame *f, double speed) { f->max_depenetration_speed = speed; }

/*
 * GJK + EPA result as a contact manifold (extension decision; nothing of this exists in the reference).
 * EPA yields one point per pair; a box resting on a face would rock about it.  Where the penetration normal n (from A
 * towards B) is a face normal of one of the bodies -- the face of A most aligned with n, or the face of B most aligned
 * with -n, cosine >= OP_FACE_ALIGN, first maximum, A on ties -- that face is the reference face of a clipped face contact
 * exactly as in the SAT.  Any other normal (an edge-edge contact), and a clip that leaves no point below the reference
 * plane, keep the one EPA point with A as the reference body.  The separation is -depth either way.
 * `axis`: the pair's cached separating direction (og_gjk_epa_cached), kept from substep to substep of a frame.
 */
#define OP_FACE_ALIGN 0.999
static void gjk_manifold(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb, o_vec3 *axis, op_manifold *m)
{
    og_result r;
    og_gjk_epa_cached(fa, fb, pa, pb, axis, &r);
    memset(m, 0, sizeof *m);
    m->separated = r.status != OG_PENETRATING; /* a degenerate query yields no contact in this substep */
    if (m->separated)
        return;
    double align[2] = { -DBL_MAX, -DBL_MAX };
    uint32_t face[2] = { UINT32_MAX, UINT32_MAX };
    for (int side = 0; side < 2; side++) {
        const o_polytope *p = side ? pb : pa;
        o_frame f = side ? fb : fa;
        o_vec3 n = side ? o_neg(r.normal) : r.normal;
        for (uint32_t k = 0; k < p->n_faces; k++) {
            double a = o_dot(o_frame_mulplane(f, o_polytope_plane(p, k)).normal, n);
            if (a > align[side]) {
                align[side] = a;
                face[side] = k;
            }
        }
    }
    if ((align[0] > align[1] ? align[0] : align[1]) >= OP_FACE_ALIGN) {
        if (align[0] >= align[1]) {
            m->feature = OP_FEATURE_FACE_A;
            m->index_a = face[0];
            face_contact(fa, pa, face[0], fb, pb, &m->index_b, m);
        } else {
            m->feature = OP_FEATURE_FACE_B;
            m->index_b = face[1];
            face_contact(fb, pb, face[1], fa, pa, &m->index_a, m);
        }
    }
    m->separation = -r.depth;
    if (m->n_points == 0) {
        m->feature = OP_FEATURE_EDGES;
        m->index_a = m->index_b = 0;
        m->n_points = 1;
        m->p_ref[0] = r.point_a;
        m->p_inc[0] = r.point_b;
    }
}

void op_contacts_attach_joints(op_frame *f, const op_joint *joints, uint32_t n_joints)
{
    f->joints = joints;
    f->n_joints = n_joints;
}

/* One joint seen from body `self_is_a ? a : b` (poses after the ground contacts). */
static void accumulate_joint(int self_is_a, const o_rigid *a, const o_rigid *b, const op_joint *j, double compliance,
                             pair_accum *acc)
{
    o_vec3 p_a = o_frame_mulv(o_rigid_frame(a), (o_vec3){ j->anchor_a[0], j->anchor_a[1], j->anchor_a[2] });
    o_vec3 p_b = o_frame_mulv(o_rigid_frame(b), (o_vec3){ j->anchor_b[0], j->anchor_b[1], j->anchor_b[2] });
    o_vec3 difference = o_sub(p_b, p_a);
    double distance = o_magnitude(difference);
    if (distance == 0.0)
        return;
    o_vec3 dir = o_scale(difference, 1.0 / distance);
    double w = generalized_inverse_mass(a, p_a, dir) + generalized_inverse_mass(b, p_b, dir);
    double lambda = (distance - j->distance) / (w + compliance);

    const o_rigid *self = self_is_a ? a : b;
    o_vec3 point = self_is_a ? p_a : p_b;
    o_vec3 impulse = self_is_a ? o_lscale(lambda, dir) : o_lscale(-lambda, dir);
    acc->dpos = o_add(acc->dpos, o_scale(impulse, self->inverse_mass));
    o_vec3 arm = o_sub(point, o_add(self->position, self->center_of_mass));
    o_quat spin = { 0.0, o_cross(o_mat3_mulv(self->inverse_inertia, arm), impulse) };
    acc->drot = o_qadd(acc->drot, o_qmul(o_qlscale(0.5, spin), self->rotation));
    acc->count++;
}

/* The angular half of Constraint::inverse_resitance (src/constraint.rs:25-32) for a unit rotation axis n. */
static double angular_inverse_mass(const o_rigid *body, o_vec3 n)
{
    o_vec3 local = o_qrot(o_qconj(body->rotation), n);
    return o_dot(o_mat3_mulv(body->inverse_inertia, local), local);
}

/* The angular term of a hinge seen from body `self_is_a ? a : b` (op_joint in the header). */
static void accumulate_hinge(int self_is_a, const o_rigid *a, const o_rigid *b, const op_joint *j, double compliance, pair_accum *acc)
{
    o_vec3 a_w = o_qrot(a->rotation, (o_vec3){ j->axis_a[0], j->axis_a[1], j->axis_a[2] });
    o_vec3 b_w = o_qrot(b->rotation, (o_vec3){ j->axis_b[0], j->axis_b[1], j->axis_b[2] });
    o_vec3 delta = o_cross(a_w, b_w);
    double mag = o_magnitude(delta);
    if (mag == 0.0)
        return;
    o_vec3 n = o_scale(delta, 1.0 / mag);
    double w = angular_inverse_mass(a, n) + angular_inverse_mass(b, n);
    double lambda = mag / (w + compliance);
    const o_rigid *self = self_is_a ? a : b;
    o_vec3 turn = self_is_a ? o_lscale(lambda, n) : o_lscale(-lambda, n);
    o_quat spin = { 0.0, o_mat3_mulv(self->inverse_inertia, turn) };
    acc->drot = o_qadd(acc->drot, o_qmul(o_qlscale(0.5, spin), self->rotation));
    acc->count++;
}

/* solver::solve (src/solver.rs:19-27) over the ground constraints of one body with op_contacts_set_max_depenetration_speed
 * on: the length of a constraint's correction is limited to max(0, limit - closing), closing = what the vertex has already
 * moved towards its target in this substep (collision::ground's delta_position along its correction, src/collision.rs:22-24;
 * taken from the body's pose BEFORE the first constraint acts, like the constraints themselves). */
static void solve_ground_limited(o_rigid *rigid, o_frame past, const o_constraint *cs, uint32_t n, double dt, double limit)
{
    double compliance = 1e-6 / (dt * dt);
    double closing[O_MAX_VERTS];
    o_frame cur = o_rigid_frame(rigid);
    for (uint32_t k = 0; k < n; k++) {
        o_vec3 position = cs[k].contact0;
        o_vec3 correction = o_sub((o_vec3){ position.x, position.y, 0.0 }, position);
        o_vec3 delta = o_frame_delta(cur, past, position);
        double len = o_magnitude(correction);
        closing[k] = len > 0.0 ? o_dot(delta, correction) / len : 0.0;
    }
    for (uint32_t k = 0; k < n; k++) {
        const o_constraint *c = &cs[k];
        const o_rigid *ro[1] = { rigid };
        o_rigid *rw[1] = { rigid };
        double distance = o_constraint_current_distance(c);
        double allowed = limit - closing[k];
        if (!(allowed > 0.0))
            allowed = 0.0;
        double error = distance > allowed ? allowed : distance;
        double lagrange_factor = (error - c->distance) / (o_constraint_inverse_resistance(c, ro) + compliance);
        o_constraint_act(c, rw, lagrange_factor);
    }
}

op_frame *op_contacts_begin(const o_rigid *bodies, const uint32_t *shape_id, uint32_t n, const o_polytope *shapes,
                            double dt, double pad)
{
    op_frame *f = (op_frame *)calloc(1, sizeof *f);
    f->n = n;
    f->shape_id = shape_id;
    f->shapes = shapes;
    op_broadphase(bodies, shape_id, n, shapes, dt, pad, &f->off, &f->nb);
    /* pair index of (i, j), i < j: pair_first[i] + rank of j among i's neighbours > i */
    f->pair_first = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)n + 1));
    uint32_t n_pairs = 0;
    for (uint32_t i = 0; i < n; i++) {
        f->pair_first[i] = n_pairs;
        for (uint32_t k = f->off[i]; k < f->off[i + 1]; k++)
            n_pairs += f->nb[k] > i;
    }
    f->pair_first[n] = n_pairs;
    f->n_pairs = n_pairs;
    f->manifolds = (op_manifold *)malloc(sizeof(op_manifold) * (n_pairs ? n_pairs : 1));
    f->gjk_axis = (o_vec3 *)calloc(n_pairs ? n_pairs : 1, sizeof(o_vec3)); /* a new pair list: nothing cached */
    f->past = (o_frame *)malloc(sizeof(o_frame) * (n ? n : 1));
    f->p1 = (o_frame *)malloc(sizeof(o_frame) * (n ? n : 1));
    f->past_pos = (o_vec3 *)malloc(sizeof(o_vec3) * (n ? n : 1));
    f->next = (o_rigid *)malloc(sizeof(o_rigid) * (n ? n : 1));
    return f;
}

uint64_t op_contacts_pair_count(const op_frame *f) { return f->n_pairs; }

void op_contacts_end(op_frame *f)
{
    if (!f)
        return;
    free(f->off);
    free(f->nb);
    free(f->pair_first);
    free(f->manifolds);
    free(f->gjk_axis);
    free(f->past);
    free(f->p1);
    free(f->past_pos);
    free(f->next);
    free(f);
}

/* One substep (steps 1-5 of the header comment).  masks_row: n entries or NULL. */
void op_contacts_substep(op_frame *f, o_rigid *bodies, double h, uint32_t *masks_row, op_contact_stats *stats)
{
    const uint32_t n = f->n;
    const uint32_t *shape_id = f->shape_id;
    const o_polytope *shapes = f->shapes;
    uint32_t *off = f->off, *nb = f->nb, *pair_first = f->pair_first;
    op_manifold *manifolds = f->manifolds;
    o_frame *past = f->past, *p1 = f->p1;
    o_vec3 *past_pos = f->past_pos;
    o_rigid *next = f->next;
    const double compliance = 1e-6 / (h * h);
    const double limit = f->max_depenetration_speed > 0.0 ? f->max_depenetration_speed * h : 0.0;
    {
        /* 1. integrate */
        for (uint32_t i = 0; i < n; i++) {
            past[i] = o_rigid_frame(&bodies[i]);
            past_pos[i] = bodies[i].position;
            o_rigid_integrate(&bodies[i], h);
            p1[i] = o_rigid_frame(&bodies[i]);
        }
        /* 2. narrowphase on the post-integrate frames */
        for (uint32_t i = 0; i < n; i++) {
            uint32_t q = pair_first[i];
            for (uint32_t k = off[i]; k < off[i + 1]; k++) {
                uint32_t j = nb[k];
                if (j <= i)
                    continue;
                const o_polytope *pa = &shapes[shape_id ? shape_id[i] : 0], *pb = &shapes[shape_id ? shape_id[j] : 0];
                /* Pre-test with the TIGHT bounding spheres (centroid, largest vertex distance; no velocity term, no
                 * pad): the neighbour lists are built once per frame from spheres inflated by the travel of a whole
                 * frame, so in most substeps most pairs are nowhere near each other.  Disjoint spheres cannot touch. */
                o_vec3 between = o_sub(o_frame_mulv(p1[j], pb->centroid), o_frame_mulv(p1[i], pa->centroid));
                double reach = shape_radius(pa) + shape_radius(pb);
                if (!(o_dot(between, between) < reach * reach)) {
                    memset(&manifolds[q], 0, sizeof manifolds[q]);
                    manifolds[q].separated = 1;
                } else if (f->narrowphase == OP_NARROWPHASE_GJK_EPA)
                    gjk_manifold(p1[i], p1[j], pa, pb, &f->gjk_axis[q], &manifolds[q]);
                else
                    op_sat(p1[i], p1[j], pa, pb, &manifolds[q]);
                if (manifolds[q].separated)
                    manifolds[q].n_points = 0;
                if (stats && manifolds[q].n_points) {
                    stats->n_touching++;
                    stats->n_points += manifolds[q].n_points;
                }
                q++;
            }
        }
        /* 3. ground contacts, sequential per body (reference path) */
        for (uint32_t i = 0; i < n; i++) {
            const o_polytope *p = &shapes[shape_id ? shape_id[i] : 0];
            o_constraint cs[O_MAX_VERTS];
            uint32_t cv[O_MAX_VERTS];
            uint32_t nc = o_ground(&bodies[i], past[i], p->vertices, p->n_vertices, cs, cv);
            if (limit > 0.0)
                solve_ground_limited(&bodies[i], past[i], cs, nc, h, limit);
            else
                o_solve(&bodies[i], cs, nc, h);
            if (masks_row) {
                uint32_t mask = 0;
                for (uint32_t c = 0; c < nc; c++)
                    mask |= 1u << cv[c];
                masks_row[i] = mask;
            }
        }
        /* 4. pair contacts, Jacobi with averaging; reads bodies[], writes next[] */
        for (uint32_t b = 0; b < n; b++) {
            pair_accum acc;
            memset(&acc, 0, sizeof acc);
            for (uint32_t k = off[b]; k < off[b + 1]; k++) {
                uint32_t j = nb[k];
                uint32_t a_body = b < j ? b : j, b_body = b < j ? j : b;
                /* pair index: rank of b_body among a_body's upper neighbours */
                uint32_t q = pair_first[a_body];
                for (uint32_t t = off[a_body]; nb[t] != b_body; t++)
                    q += nb[t] > a_body;
                const op_manifold *m = &manifolds[q];
                if (m->n_points == 0)
                    continue;
                int ref_is_a = m->feature != OP_FEATURE_FACE_B;
                uint32_t ref = ref_is_a ? a_body : b_body, inc = ref_is_a ? b_body : a_body;
                for (uint32_t pt = 0; pt < m->n_points; pt++)
                    accumulate_point(inc == b, &bodies[inc], &bodies[ref], p1[inc], past[inc], p1[ref], past[ref],
                                     m->p_inc[pt], m->p_ref[pt], compliance, limit, &acc);
            }
            for (uint32_t jn = 0; jn < f->n_joints; jn++) { /* ascending joint index */
                const op_joint *j = &f->joints[jn];
                if (j->body_a == b || j->body_b == b) {
                    accumulate_joint(j->body_a == b, &bodies[j->body_a], &bodies[j->body_b], j, compliance, &acc);
                    if (j->kind == OP_JOINT_HINGE)
                        accumulate_hinge(j->body_a == b, &bodies[j->body_a], &bodies[j->body_b], j, compliance, &acc);
                }
            }
            next[b] = bodies[b];
            if (acc.count) {
                double cnt = (double)acc.count;
                next[b].position = o_add(bodies[b].position, o_divs(acc.dpos, cnt));
                o_quat avg = { acc.drot.s / cnt, o_divs(acc.drot.v, cnt) };
                next[b].rotation = o_qnormalize(o_qadd(bodies[b].rotation, avg));
            }
        }
        /* 5. derive */
        for (uint32_t i = 0; i < n; i++) {
            bodies[i] = next[i];
            o_rigid_derive(&bodies[i], past_pos[i], past[i].rotation, h);
        }
    }
}

void op_contacts_step(o_rigid *bodies, const uint32_t *shape_id, uint32_t n, const o_polytope *shapes,
                      double dt, uint32_t substeps, double pad, uint32_t *ground_masks, op_contact_stats *stats)
{
    op_frame *f = op_contacts_begin(bodies, shape_id, n, shapes, dt, pad);
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->n_pairs = f->n_pairs;
    }
    const double h = dt / (double)substeps;
    for (uint32_t step = 0; step < substeps; step++)
        op_contacts_substep(f, bodies, h, ground_masks ? ground_masks + (size_t)step * n : NULL, stats);
    op_contacts_end(f);
}
