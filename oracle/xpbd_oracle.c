/*
 * xpbd_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See xpbd_oracle.h for scope and parity status ("parity unpinned": no
 * reference fixtures exist; pinned by analytic KATs only).
 *
 * Every function names the reference lines it restates.  Expressions keep the
 * reference's operator order token for token; C and Rust agree on precedence
 * and left-associativity for + - * /, and this file is compiled with
 * -ffp-contract=off so no a*b+c is fused.
 */
#include "xpbd_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

/* ======================================================================
 * cgmath 0.18.0 (non-SIMD generic paths)
 * ====================================================================== */

static o_vec3 v3(double x, double y, double z) { o_vec3 r = { x, y, z }; return r; }
static o_quat q4(double s, double x, double y, double z) { o_quat r = { s, { x, y, z } }; return r; }

o_vec3 o_add(o_vec3 a, o_vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
o_vec3 o_sub(o_vec3 a, o_vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
o_vec3 o_neg(o_vec3 a) { return v3(-a.x, -a.y, -a.z); }
o_vec3 o_scale(o_vec3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
o_vec3 o_lscale(double s, o_vec3 a) { return v3(s * a.x, s * a.y, s * a.z); }
o_vec3 o_divs(o_vec3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }

/* cgmath vector.rs: dot = mul_element_wise(..).sum() = (x*x' + y*y') + z*z' */
double o_dot(o_vec3 a, o_vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

/* cgmath vector.rs Vector3::cross */
o_vec3 o_cross(o_vec3 a, o_vec3 b)
{
    return v3((a.y * b.z) - (a.z * b.y),
              (a.z * b.x) - (a.x * b.z),
              (a.x * b.y) - (a.y * b.x));
}

/* cgmath structure.rs InnerSpace: magnitude2 = dot(self,self); magnitude = sqrt;
 * normalize = normalize_to(1) = self * (1 / magnitude). */
double o_magnitude2(o_vec3 a) { return o_dot(a, a); }
double o_magnitude(o_vec3 a) { return sqrt(o_magnitude2(a)); }
o_vec3 o_normalize(o_vec3 a) { return o_scale(a, 1.0 / o_magnitude(a)); }
/* project_on(other) = other * (self.dot(other) / other.magnitude2()) */
o_vec3 o_project_on(o_vec3 a, o_vec3 onto)
{
    return o_scale(onto, o_dot(a, onto) / o_magnitude2(onto));
}

/* cgmath quaternion.rs Mul<Quaternion> */
o_quat o_qmul(o_quat a, o_quat b)
{
    return q4(a.s * b.s - a.v.x * b.v.x - a.v.y * b.v.y - a.v.z * b.v.z,
              a.s * b.v.x + a.v.x * b.s + a.v.y * b.v.z - a.v.z * b.v.y,
              a.s * b.v.y + a.v.y * b.s + a.v.z * b.v.x - a.v.x * b.v.z,
              a.s * b.v.z + a.v.z * b.s + a.v.x * b.v.y - a.v.y * b.v.x);
}

/* cgmath quaternion.rs Mul<Vector3>: tmp = v x rhs + rhs*s; (v x tmp)*2 + rhs */
o_vec3 o_qrot(o_quat q, o_vec3 rhs)
{
    o_vec3 tmp = o_add(o_cross(q.v, rhs), o_scale(rhs, q.s));
    return o_add(o_scale(o_cross(q.v, tmp), 2.0), rhs);
}

o_quat o_qconj(o_quat q) { o_quat r = { q.s, o_neg(q.v) }; return r; }
o_quat o_qadd(o_quat a, o_quat b) { o_quat r = { a.s + b.s, o_add(a.v, b.v) }; return r; }
o_quat o_qlscale(double s, o_quat q) { o_quat r = { s * q.s, o_lscale(s, q.v) }; return r; }
o_quat o_qneg(o_quat q) { o_quat r = { -q.s, o_neg(q.v) }; return r; }
/* quaternion InnerSpace: dot = s*s' + v.dot(v'); normalize via the default */
o_quat o_qnormalize(o_quat q)
{
    double mag = sqrt(q.s * q.s + o_dot(q.v, q.v));
    double k = 1.0 / mag;
    o_quat r = { q.s * k, o_scale(q.v, k) };
    return r;
}

/* cgmath quaternion.rs From<Euler<A>> with Deg -> Rad = deg * (PI/180). */
o_quat o_quat_from_euler_deg(double x_deg, double y_deg, double z_deg)
{
    const double k = 3.14159265358979323846264338327950288 / 180.0;
    double hx = (x_deg * k) * 0.5, hy = (y_deg * k) * 0.5, hz = (z_deg * k) * 0.5;
    double s_x = sin(hx), c_x = cos(hx);
    double s_y = sin(hy), c_y = cos(hy);
    double s_z = sin(hz), c_z = cos(hz);
    return q4(-s_x * s_y * s_z + c_x * c_y * c_z,
              s_x * c_y * c_z + s_y * s_z * c_x,
              -s_x * s_z * c_y + s_y * c_x * c_z,
              s_x * s_y * c_z + s_z * c_x * c_y);
}

/* cgmath matrix.rs Mul<Vector3> for Matrix3: m[0]*v[0] + m[1]*v[1] + m[2]*v[2] */
o_vec3 o_mat3_mulv(o_mat3 m, o_vec3 v)
{
    return o_add(o_add(o_scale(m.x, v.x), o_scale(m.y, v.y)), o_scale(m.z, v.z));
}

o_mat3 o_mat3_lscale(double s, o_mat3 m)
{
    o_mat3 r = { o_lscale(s, m.x), o_lscale(s, m.y), o_lscale(s, m.z) };
    return r;
}

/* cgmath matrix.rs SquareMatrix for Matrix3: determinant + invert */
int o_mat3_invert(o_mat3 m, o_mat3 *out)
{
    /* self[c][r]: c = column */
    double det = m.x.x * (m.y.y * m.z.z - m.z.y * m.y.z)
               - m.y.x * (m.x.y * m.z.z - m.z.y * m.x.z)
               + m.z.x * (m.x.y * m.y.z - m.y.y * m.x.z);
    if (det == 0.0)
        return 0;
    o_vec3 c0 = o_divs(o_cross(m.y, m.z), det);
    o_vec3 c1 = o_divs(o_cross(m.z, m.x), det);
    o_vec3 c2 = o_divs(o_cross(m.x, m.y), det);
    /* from_cols(c0,c1,c2).transpose() */
    out->x = v3(c0.x, c1.x, c2.x);
    out->y = v3(c0.y, c1.y, c2.y);
    out->z = v3(c0.z, c1.z, c2.z);
    return 1;
}

/* ======================================================================
 * frame.rs
 * ====================================================================== */

/* frame.rs:30-37 */
o_frame o_frame_inverse(o_frame f)
{
    o_quat inverse_orientation = o_qconj(f.rotation);
    o_vec3 inverse_position = o_qrot(inverse_orientation, o_neg(f.position));
    o_frame r = { inverse_position, inverse_orientation };
    return r;
}

/* frame.rs:47-53  Frame * Vector3 = rotation * rhs + position */
o_vec3 o_frame_mulv(o_frame f, o_vec3 v) { return o_add(o_qrot(f.rotation, v), f.position); }

/* frame.rs:40-44 */
o_vec3 o_frame_delta(o_frame f, o_frame past, o_vec3 global)
{
    o_vec3 local = o_frame_mulv(o_frame_inverse(f), global);
    o_vec3 past_global = o_frame_mulv(past, local);
    return o_sub(global, past_global);
}

/* frame.rs:55-64 */
o_plane o_frame_mulplane(o_frame f, o_plane p)
{
    o_vec3 support = o_lscale(p.displacement, p.normal);
    support = o_frame_mulv(f, support);
    o_vec3 normal = o_qrot(f.rotation, p.normal);
    return o_plane_from_point_normal(support, normal);
}

/* frame.rs:67-76 */
o_frame o_frame_mul(o_frame a, o_frame b)
{
    o_frame r = { o_add(a.position, o_qrot(a.rotation, b.position)), o_qmul(a.rotation, b.rotation) };
    return r;
}

/* ======================================================================
 * geometry.rs
 * ====================================================================== */

/* geometry.rs:16-24 */
o_plane o_plane_from_points(o_vec3 p0, o_vec3 p1, o_vec3 p2)
{
    o_vec3 normal = o_normalize(o_cross(o_sub(p1, p0), o_sub(p2, p0)));
    o_plane r = { normal, o_dot(normal, p0) };
    return r;
}

/* geometry.rs:27-36 */
o_plane o_plane_from_point_normal(o_vec3 point, o_vec3 normal)
{
    double displacement = o_magnitude(o_project_on(point, normal));
    if (o_dot(point, normal) < 0.0)
        displacement *= -1.0;
    o_plane r = { normal, displacement };
    return r;
}

/* geometry.rs:39-41 */
double o_plane_distance(o_plane pl, o_vec3 p) { return o_dot(pl.normal, p) - pl.displacement; }

/* geometry.rs:62-64, :66-68, :55-60 */
static o_vec3 plane_support(o_plane pl) { return o_lscale(pl.displacement, pl.normal); }
static int plane_facing(o_plane pl, o_vec3 p) { return o_dot(pl.normal, o_sub(p, plane_support(pl))) >= 0.0; }
static o_plane plane_flip(o_plane pl) { o_plane r = { o_neg(pl.normal), -pl.displacement }; return r; }

static void set_faces(o_polytope *p, const uint32_t *counts, const uint32_t *flat, uint32_t n_faces)
{
    uint32_t off = 0;
    p->n_faces = n_faces;
    for (uint32_t f = 0; f < n_faces; f++) {
        p->face_offsets[f] = off;
        for (uint32_t k = 0; k < counts[f]; k++)
            p->face_indices[off + k] = flat[off + k];
        off += counts[f];
    }
    p->face_offsets[n_faces] = off;
}

/* geometry.rs:97-109 */
void o_polytope_tetrahedron(o_polytope *p)
{
    static const uint32_t e[6][2] = { {0,1},{0,2},{0,3},{1,2},{1,3},{2,3} };
    static const uint32_t fc[4] = { 3, 3, 3, 3 };
    static const uint32_t ff[12] = { 0,3,2, 3,0,1, 2,1,0, 1,2,3 };
    memset(p, 0, sizeof *p);
    p->centroid = v3(0.25, 0.25, 0.25);
    p->n_vertices = 4;
    p->vertices[0] = v3(0.0, 0.0, 0.0);
    p->vertices[1] = v3(1.0, 0.0, 0.0);
    p->vertices[2] = v3(0.0, 1.0, 0.0);
    p->vertices[3] = v3(0.0, 0.0, 1.0);
    p->n_edges = 6;
    memcpy(p->edges, e, sizeof e);
    set_faces(p, fc, ff, 4);
}

/* geometry.rs:113-149 */
void o_polytope_cube(o_polytope *p)
{
    static const uint32_t e[12][2] = { {0,1},{1,3},{3,2},{2,0},{4,5},{5,7},{7,6},{6,4},{0,4},{1,5},{3,7},{2,6} };
    static const uint32_t fc[6] = { 4, 4, 4, 4, 4, 4 };
    static const uint32_t ff[24] = { 0,2,3,1, 4,5,7,6, 4,0,1,5, 5,1,3,7, 7,3,2,6, 6,2,0,4 };
    memset(p, 0, sizeof *p);
    p->centroid = v3(0.5, 0.5, 0.5);
    p->n_vertices = 8;
    for (uint32_t i = 0; i < 8; i++)
        p->vertices[i] = v3((i & 1) ? 1.0 : 0.0, (i & 2) ? 1.0 : 0.0, (i & 4) ? 1.0 : 0.0);
    p->n_edges = 12;
    memcpy(p->edges, e, sizeof e);
    set_faces(p, fc, ff, 6);
}

/* geometry.rs:153-230 */
void o_polytope_icosahedron(o_polytope *p)
{
    static const uint32_t e[30][2] = {
        {8,9},{8,0},{8,1},{1,0},{9,2},{9,3},{2,3},{2,5},{2,11},{5,11},
        {3,7},{3,11},{7,11},{0,5},{1,7},{4,5},{4,0},{4,8},{4,9},{4,2},
        {6,9},{6,8},{6,1},{6,7},{6,3},{10,1},{10,0},{10,5},{10,11},{10,7} };
    static const uint32_t ff[60] = {
        0,5,4, 2,4,5, 1,6,7, 3,7,6, 1,0,8, 0,1,10, 2,3,9, 3,2,11, 4,9,8, 6,8,9,
        5,10,11, 7,11,10, 0,4,8, 0,10,5, 2,9,4, 2,5,11, 1,8,6, 1,7,10, 3,6,9, 3,11,7 };
    uint32_t fc[20];
    double phi = (1.0 + sqrt(5.0)) / 2.0;
    double mag = sqrt(phi * phi + 1.0);
    double a = phi / mag;
    double b = 1.0 / mag;
    memset(p, 0, sizeof *p);
    p->centroid = v3(0.0, 0.0, 0.0);
    p->n_vertices = 12;
    p->vertices[0] = v3(a, b, 0.0);
    p->vertices[1] = v3(a, -b, 0.0);
    p->vertices[2] = v3(-a, b, 0.0);
    p->vertices[3] = v3(-a, -b, 0.0);
    p->vertices[4] = v3(0.0, a, b);
    p->vertices[5] = v3(0.0, a, -b);
    p->vertices[6] = v3(0.0, -a, b);
    p->vertices[7] = v3(0.0, -a, -b);
    p->vertices[8] = v3(b, 0.0, a);
    p->vertices[9] = v3(-b, 0.0, a);
    p->vertices[10] = v3(b, 0.0, -a);
    p->vertices[11] = v3(-b, 0.0, -a);
    p->n_edges = 30;
    memcpy(p->edges, e, sizeof e);
    for (int i = 0; i < 20; i++) fc[i] = 3;
    set_faces(p, fc, ff, 20);
}

/* geometry.rs:296-307  f64 * Polytope */
void o_polytope_scale(double s, o_polytope *p)
{
    for (uint32_t i = 0; i < p->n_vertices; i++)
        p->vertices[i] = o_lscale(s, p->vertices[i]);
    p->centroid = o_lscale(s, p->centroid);
}

/* geometry.rs:262-271 */
o_plane o_polytope_plane(const o_polytope *p, uint32_t i)
{
    const uint32_t *face = &p->face_indices[p->face_offsets[i]];
    o_plane plane = o_plane_from_points(p->vertices[face[0]], p->vertices[face[1]], p->vertices[face[2]]);
    if (!plane_facing(plane, p->centroid))
        return plane;
    return plane_flip(plane);
}

/* f64::total_cmp (IEEE totalOrder) as Rust implements it. */
static int total_cmp(double a, double b)
{
    int64_t l, r;
    memcpy(&l, &a, 8);
    memcpy(&r, &b, 8);
    l ^= (int64_t)((uint64_t)(l >> 63) >> 1);
    r ^= (int64_t)((uint64_t)(r >> 63) >> 1);
    return (l > r) - (l < r);
}

/* geometry.rs:274-281 + :309-319.  Iterator::max_by keeps the LAST maximum. */
o_vec3 o_polytope_support(const o_polytope *p, o_frame f, o_vec3 direction)
{
    o_vec3 best = o_frame_mulv(f, p->vertices[0]);
    for (uint32_t i = 1; i < p->n_vertices; i++) {
        o_vec3 x = o_frame_mulv(f, p->vertices[i]);
        if (total_cmp(o_dot(best, direction), o_dot(x, direction)) <= 0)
            best = x;
    }
    return best;
}

/* geometry.rs:283-289 (same polytope under both frames, as written) */
o_vec3 o_polytope_minkowski_support(const o_polytope *p, o_frame f0, o_frame f1, o_vec3 d)
{
    return o_sub(o_polytope_support(p, f0, d), o_polytope_support(p, f1, o_neg(d)));
}

/* ======================================================================
 * geometry/integrate.rs  (Mirtich 1996)
 * ====================================================================== */

typedef struct { double e, a, b, aa, ab, bb, aaa, aab, abb, bbb; } proj_integrals;
typedef struct { double a, b, c, aa, bb, cc, aaa, bbb, ccc, aab, bbc, cca; } face_integrals_t;
typedef struct { o_vec3 normal; double displacement; const uint32_t *vertices; uint32_t n; } face_t;

static double comp(o_vec3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
static double *compp(o_vec3 *v, int i) { return i == 0 ? &v->x : (i == 1 ? &v->y : &v->z); }
/* integrate.rs:290-296: powi(2) = x*x, powi(3) = (x*x)*x */
static double sq(double x) { return x * x; }
static double cb(double x) { return x * x * x; }

/* integrate.rs:223-288 */
static proj_integrals projection_integrals(const o_vec3 *verts, const face_t *f, int alpha, int beta)
{
    proj_integrals in;
    memset(&in, 0, sizeof in);
    for (uint32_t i = 0; i < f->n; i++) {
        double a0 = comp(verts[f->vertices[i]], alpha);
        double b0 = comp(verts[f->vertices[i]], beta);
        double a1 = comp(verts[f->vertices[(i + 1) % f->n]], alpha);
        double b1 = comp(verts[f->vertices[(i + 1) % f->n]], beta);
        double da = a1 - a0;
        double db = b1 - b0;
        double a0_2 = a0 * a0;
        double a0_3 = a0_2 * a0;
        double a0_4 = a0_3 * a0;
        double b0_2 = b0 * b0;
        double b0_3 = b0_2 * b0;
        double b0_4 = b0_3 * b0;
        double a1_2 = a1 * a1;
        double a1_3 = a1_2 * a1;
        double b1_2 = b1 * b1;
        double b1_3 = b1_2 * b1;

        double c_1 = a1 + a0;
        double c_a = a1 * c_1 + a0_2;
        double c_aa = a1 * c_a + a0_3;
        double c_aaa = a1 * c_aa + a0_4;
        double c_b = b1 * (b1 + b0) + b0_2;
        double c_bb = b1 * c_b + b0_3;
        double c_bbb = b1 * c_bb + b0_4;
        double c_ab = 3.0 * a1_2 + 2.0 * a1 * a0 + a0_2;
        double k_ab = a1_2 + 2.0 * a1 * a0 + 3.0 * a0_2;
        double c_aab = a0 * c_ab + 4.0 * a1_3;
        double k_aab = a1 * k_ab + 4.0 * a0_3;
        double c_abb = 4.0 * b1_3 + 3.0 * b1_2 * b0 + 2.0 * b1 * b0_2 + b0_3;
        double k_abb = b1_3 + 2.0 * b1_2 * b0 + 3.0 * b1 * b0_2 + 4.0 * b0_3;

        in.e += db * c_1;
        in.a += db * c_a;
        in.aa += db * c_aa;
        in.aaa += db * c_aaa;
        in.b += da * c_b;
        in.bb += da * c_bb;
        in.bbb += da * c_bbb;
        in.ab += db * (b1 * c_ab + b0 * k_ab);
        in.aab += db * (b1 * c_aab + b0 * k_aab);
        in.abb += da * (a1 * c_abb + a0 * k_abb);
    }
    in.e /= 2.0;
    in.a /= 6.0;
    in.aa /= 12.0;
    in.aaa /= 20.0;
    in.b /= -6.0;
    in.bb /= -12.0;
    in.bbb /= -20.0;
    in.ab /= 24.0;
    in.aab /= 60.0;
    in.abb /= -60.0;
    return in;
}

/* integrate.rs:171-220 */
static face_integrals_t face_integrals(const o_vec3 *verts, const face_t *f, int alpha, int beta, int gamma)
{
    proj_integrals p = projection_integrals(verts, f, alpha, beta);
    double w = f->displacement;
    double na = comp(f->normal, alpha), nb = comp(f->normal, beta);
    double k1 = 1.0 / comp(f->normal, gamma);
    double k2 = k1 * k1;
    double k3 = k2 * k1;
    double k4 = k3 * k1;
    face_integrals_t r;
    r.a = k1 * p.a;
    r.b = k1 * p.b;
    r.c = -k2 * (na * p.a + nb * p.b + w * p.e);
    r.aa = k1 * p.aa;
    r.bb = k1 * p.bb;
    r.cc = k3 * (sq(na) * p.aa + 2.0 * na * nb * p.ab + sq(nb) * p.bb
                 + w * (2.0 * (na * p.a + nb * p.b) + w * p.e));
    r.aaa = k1 * p.aaa;
    r.bbb = k1 * p.bbb;
    r.ccc = -k4 * (cb(na) * p.aaa + 3.0 * sq(na) * nb * p.aab + 3.0 * na * sq(nb) * p.abb
                   + cb(nb) * p.bbb
                   + 3.0 * w * (sq(na) * p.aa + 2.0 * na * nb * p.ab + sq(nb) * p.bb)
                   + w * w * (3.0 * (na * p.a + nb * p.b) + w * p.e));
    r.aab = k1 * p.aab;
    r.bbc = -k2 * (na * p.abb + nb * p.bbb + w * p.bb);
    r.cca = k3 * (sq(na) * p.aaa + 2.0 * na * nb * p.aab + sq(nb) * p.abb
                  + w * (2.0 * (na * p.aa + nb * p.ab) + w * p.a));
    return r;
}

/* integrate.rs:26-75 (rigid_metrics) with :126-169 (volume_integrals) inlined */
void o_rigid_metrics(const o_polytope *poly, double density, o_metrics *out)
{
    double t0 = 0.0;
    o_vec3 t1 = v3(0, 0, 0), t2 = v3(0, 0, 0), tp = v3(0, 0, 0);

    for (uint32_t i = 0; i < poly->n_faces; i++) {
        o_plane plane = o_polytope_plane(poly, i);
        face_t face;
        face.normal = plane.normal;
        face.displacement = -plane.displacement; /* Plane::constant(), integrate.rs:35 */
        face.vertices = &poly->face_indices[poly->face_offsets[i]];
        face.n = poly->face_offsets[i + 1] - poly->face_offsets[i];

        double nx = fabs(face.normal.x), ny = fabs(face.normal.y), nz = fabs(face.normal.z);
        int gamma = (nx > ny && nx > nz) ? 0 : (ny > nz ? 1 : 2);
        int alpha = (gamma + 1) % 3;
        int beta = (alpha + 1) % 3;

        face_integrals_t f = face_integrals(poly->vertices, &face, alpha, beta, gamma);

        t0 += face.normal.x * (alpha == 0 ? f.a : (beta == 0 ? f.b : f.c));

        *compp(&t1, alpha) += comp(face.normal, alpha) * f.aa;
        *compp(&t1, beta) += comp(face.normal, beta) * f.bb;
        *compp(&t1, gamma) += comp(face.normal, gamma) * f.cc;
        *compp(&t2, alpha) += comp(face.normal, alpha) * f.aaa;
        *compp(&t2, beta) += comp(face.normal, beta) * f.bbb;
        *compp(&t2, gamma) += comp(face.normal, gamma) * f.ccc;
        *compp(&tp, alpha) += comp(face.normal, alpha) * f.aab;
        *compp(&tp, beta) += comp(face.normal, beta) * f.bbc;
        *compp(&tp, gamma) += comp(face.normal, gamma) * f.cca;
    }
    t1 = o_divs(t1, 2.0);
    t2 = o_divs(t2, 3.0);
    tp = o_divs(tp, 2.0);

    double m = density * t0;
    o_vec3 r = o_divs(t1, t0);
    o_mat3 j;
    memset(&j, 0, sizeof j);

    j.x.x = density * (t2.y + t2.z);
    j.y.y = density * (t2.z + t2.x);
    j.z.z = density * (t2.x + t2.y);
    j.x.y = -density * tp.x;
    j.y.z = -density * tp.y;
    j.z.x = -density * tp.z;
    j.y.x = j.x.y;
    j.z.y = j.y.z;
    j.x.z = j.z.x;

    j.x.x -= m * (r.y * r.y + r.z * r.z);
    j.y.y -= m * (r.z * r.z + r.x * r.x);
    j.z.z -= m * (r.x * r.x + r.y * r.y);
    j.x.y += m * r.x * r.y;
    j.y.z += m * r.y * r.z;
    j.z.x += m * r.z * r.x;
    j.y.x = j.x.y;
    j.z.y = j.y.z;
    j.x.z = j.z.x;

    out->mass = m;
    out->volume = t0;
    out->center_of_mass = r;
    out->inertia_tensor = j;
}

/* ======================================================================
 * rigid.rs
 * ====================================================================== */

/* rigid.rs:53-71 */
int o_rigid_new(const o_metrics *m, o_rigid *out)
{
    memset(out, 0, sizeof *out);
    out->inverse_mass = 1.0 / m->mass;
    if (!o_mat3_invert(m->inertia_tensor, &out->inverse_inertia))
        return 0; /* .expect("Inertia tensor is not invertible") */
    out->center_of_mass = m->center_of_mass;
    out->rotation = q4(1.0, 0.0, 0.0, 0.0);
    return 1;
}

/* rigid.rs:75-80 */
o_frame o_rigid_frame(const o_rigid *r)
{
    o_frame f;
    f.position = o_add(o_add(r->position, r->center_of_mass), o_qrot(r->rotation, o_neg(r->center_of_mass)));
    f.rotation = r->rotation;
    return f;
}

/* rigid.rs:82-99 */
void o_rigid_integrate(o_rigid *r, double dt)
{
    o_vec3 force = o_add(r->external_force, o_qrot(r->rotation, r->internal_force));
    r->velocity = o_add(r->velocity, o_scale(o_lscale(dt, force), r->inverse_mass));
    r->position = o_add(r->position, o_lscale(dt, r->velocity));

    o_vec3 torque = o_add(r->external_torque, o_qrot(r->rotation, r->internal_torque));
    r->angular_velocity = o_add(r->angular_velocity, o_mat3_mulv(o_mat3_lscale(dt, r->inverse_inertia), torque));
    o_quat delta_rotation = o_qmul(
        o_qlscale(dt * 0.5, q4(0.0, r->angular_velocity.x, r->angular_velocity.y, r->angular_velocity.z)),
        r->rotation);
    r->rotation = o_qnormalize(o_qadd(r->rotation, delta_rotation));
}

/* rigid.rs:101-109 */
void o_rigid_derive(o_rigid *r, o_vec3 position, o_quat rotation, double dt)
{
    r->velocity = o_divs(o_sub(r->position, position), dt);
    o_quat delta = o_qmul(r->rotation, o_qconj(rotation));
    if (delta.s < 0.0)
        delta = o_qneg(delta);
    r->angular_velocity = o_divs(o_lscale(2.0, delta.v), dt);
}

/* rigid.rs:113-123 */
void o_rigid_apply_impulse(o_rigid *r, o_vec3 impulse, o_vec3 point)
{
    r->position = o_add(r->position, o_scale(impulse, r->inverse_mass));
    o_vec3 arm = o_sub(point, o_add(r->position, r->center_of_mass));
    o_quat w = { 0.0, o_cross(o_mat3_mulv(r->inverse_inertia, arm), impulse) };
    r->rotation = o_qadd(r->rotation, o_qmul(o_qlscale(0.5, w), r->rotation));
    r->rotation = o_qnormalize(r->rotation);
}

/* ======================================================================
 * constraint.rs
 * ====================================================================== */

static o_vec3 constraint_difference(const o_constraint *c) { return o_sub(c->contact1, c->contact0); } /* :13-15 */
static o_vec3 constraint_direction(const o_constraint *c) { return o_normalize(constraint_difference(c)); } /* :17-19 */

/* constraint.rs:21-23 */
double o_constraint_current_distance(const o_constraint *c) { return o_magnitude(constraint_difference(c)); }

/* constraint.rs:25-32 */
double o_constraint_inverse_resistance(const o_constraint *c, const o_rigid *const *rigids)
{
    const o_rigid *rigid = rigids[c->rigid];
    o_vec3 angular_impulse = o_qrot(
        o_qconj(rigid->rotation),
        o_cross(o_sub(c->contact0, o_add(rigid->position, rigid->center_of_mass)), constraint_direction(c)));
    return rigid->inverse_mass + o_dot(o_mat3_mulv(rigid->inverse_inertia, angular_impulse), angular_impulse);
}

/* constraint.rs:34-37 */
void o_constraint_act(const o_constraint *c, o_rigid *const *rigids, double factor)
{
    o_vec3 impulse = o_lscale(factor, constraint_direction(c));
    o_rigid_apply_impulse(rigids[c->rigid], impulse, c->contact0);
}

/* ======================================================================
 * collision.rs
 * ====================================================================== */

/* collision.rs:13-35 */
uint32_t o_ground(const o_rigid *rigid, o_frame past, const o_vec3 *vertices, uint32_t n_vertices,
                  o_constraint *out, uint32_t *out_vertex)
{
    uint32_t n = 0;
    for (uint32_t i = 0; i < n_vertices; i++) {
        o_vec3 position = o_frame_mulv(o_rigid_frame(rigid), vertices[i]);
        if (position.z >= 0.0)
            continue;

        o_vec3 target_position = v3(position.x, position.y, 0.0);
        o_vec3 correction = o_sub(target_position, position);
        o_vec3 delta_position = o_frame_delta(o_rigid_frame(rigid), past, position);
        o_vec3 delta_tangential_position = o_sub(delta_position, o_project_on(delta_position, correction));

        out[n].rigid = 0;
        out[n].contact0 = position;
        out[n].contact1 = o_sub(target_position, o_lscale(1.0, delta_tangential_position));
        out[n].distance = 0.0;
        if (out_vertex)
            out_vertex[n] = i;
        n++;
    }
    return n;
}

/* collision.rs:123-149 */
double o_face_axes_separation(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb,
                              uint64_t *face_index)
{
    double max_distance = -DBL_MAX; /* f64::MIN */
    *face_index = UINT64_MAX;       /* usize::MAX */
    o_frame fa_inv = o_frame_inverse(fa);
    for (uint32_t i = 0; i < pa->n_faces; i++) {
        o_plane plane = o_polytope_plane(pa, i);
        o_vec3 nn = o_neg(plane.normal);
        o_vec3 support = o_frame_mulv(fa_inv, o_frame_mulv(fb, pb->vertices[0]));
        for (uint32_t k = 1; k < pb->n_vertices; k++) {
            o_vec3 x = o_frame_mulv(fa_inv, o_frame_mulv(fb, pb->vertices[k]));
            if (total_cmp(o_dot(support, nn), o_dot(x, nn)) <= 0)
                support = x; /* max_by: last maximum wins */
        }
        double distance = o_plane_distance(plane, support);
        if (distance > max_distance) {
            max_distance = distance;
            *face_index = i;
        }
    }
    return max_distance;
}

/* collision.rs:151-197 */
double o_edge_axes_separation(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb,
                              uint64_t *edge_a, uint64_t *edge_b)
{
    double max_distance = -DBL_MAX;
    *edge_a = UINT64_MAX;
    *edge_b = UINT64_MAX;
    for (uint32_t ie = 0; ie < pa->n_edges; ie++) {
        for (uint32_t je = 0; je < pb->n_edges; je++) {
            const uint32_t *i = pa->edges[ie];
            const uint32_t *j = pb->edges[je];
            o_vec3 foot = o_frame_mulv(fa, pa->vertices[i[0]]);
            o_vec3 e0 = o_sub(o_frame_mulv(fa, pa->vertices[i[1]]), foot);
            o_vec3 e1 = o_sub(o_frame_mulv(fb, pb->vertices[j[1]]), o_frame_mulv(fb, pb->vertices[j[0]]));
            o_vec3 axis = o_normalize(o_cross(e0, e1));
            if (o_dot(axis, o_sub(foot, o_frame_mulv(fa, pa->centroid))) < 0.0)
                axis = o_neg(axis);
            if (o_dot(o_polytope_support(pa, fa, axis), axis) > o_dot(foot, axis))
                continue;
            o_plane plane = o_plane_from_point_normal(foot, axis);
            double distance = o_plane_distance(plane, o_polytope_support(pb, fb, o_neg(axis)));
            if (distance > max_distance) {
                max_distance = distance;
                *edge_a = ie;
                *edge_b = je;
            }
        }
    }
    return max_distance;
}

/* ======================================================================
 * solver.rs
 * ====================================================================== */

/* solver.rs:19-27 */
void o_solve(o_rigid *rigid, const o_constraint *cs, uint32_t n, double dt)
{
    double compliance = 1e-6 / (dt * dt);
    for (uint32_t k = 0; k < n; k++) {
        const o_constraint *c = &cs[k];
        const o_rigid *ro[1] = { rigid };
        o_rigid *rw[1] = { rigid };
        double difference = o_constraint_current_distance(c) - c->distance;
        double lagrange_factor = difference / (o_constraint_inverse_resistance(c, ro) + compliance);
        o_constraint_act(c, rw, lagrange_factor);
    }
}

/* solver.rs:3-17 */
static void step_impl(o_rigid *rigid, const o_vec3 *vertices, uint32_t n_vertices, double dt,
                      size_t substep_count, uint32_t *masks, size_t mask_stride)
{
    o_constraint cs[O_MAX_VERTS];
    uint32_t cv[O_MAX_VERTS];
    dt = dt / (double)substep_count;

    for (size_t k = 0; k < substep_count; k++) {
        o_vec3 past_position = rigid->position;
        o_quat past_rotation = rigid->rotation;
        o_frame past_frame = o_rigid_frame(rigid);
        o_rigid_integrate(rigid, dt);

        uint32_t n = o_ground(rigid, past_frame, vertices, n_vertices, cs, cv);
        o_solve(rigid, cs, n, dt);

        o_rigid_derive(rigid, past_position, past_rotation, dt);

        if (masks) {
            uint32_t m = 0;
            for (uint32_t c = 0; c < n; c++)
                m |= 1u << cv[c];
            masks[k * mask_stride] = m;
        }
    }
}

void o_step(o_rigid *rigid, const o_vec3 *vertices, uint32_t n_vertices, double dt,
            size_t substep_count, uint32_t *masks)
{
    step_impl(rigid, vertices, n_vertices, dt, substep_count, masks, 1);
}

/* ======================================================================
 * world.rs
 * ====================================================================== */

/* world.rs:12-31 */
int o_world_new(const o_polytope *p1, const o_polytope *p2, o_rigid *a, o_rigid *b)
{
    o_metrics ma, mb;
    o_rigid_metrics(p1, 0.1, &ma);
    o_rigid_metrics(p2, 5.0, &mb);
    if (!o_rigid_new(&ma, a) || !o_rigid_new(&mb, b))
        return 0;

    a->position.z = 4.0;
    a->velocity.y = 2.5;
    a->angular_velocity.x = -4.0;
    a->angular_velocity.y = 1.0;
    a->external_force.z = -2.0;

    b->position.x = 4.0;
    b->position.z = 4.0;
    b->velocity.z = 7.0;
    b->angular_velocity.x = -5.0;
    b->angular_velocity.y = 5.0;
    b->external_force.z = -2.0;
    b->rotation = o_quat_from_euler_deg(10.0, 15.0, 5.0);
    return 1;
}

/* world.rs:34-43 -- p1 for BOTH bodies, 25 substeps, as written. */
void o_world_integrate(o_rigid *a, o_rigid *b, double dt, const o_polytope *p1)
{
    o_step(a, p1->vertices, p1->n_vertices, dt, 25, NULL);
    o_step(b, p1->vertices, p1->n_vertices, dt, 25, NULL);
}

/* ======================================================================
 * batch helper (parity tests, cpu_baseline)
 * ====================================================================== */

void o_step_bodies(o_rigid *bodies, const uint32_t *shape_id, uint32_t n,
                   const double *verts_xyz, const uint32_t *vert_offsets,
                   double dt, uint32_t substeps, uint32_t *masks, int threads)
{
    const o_vec3 *verts = (const o_vec3 *)verts_xyz;
    long i;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 1 ? threads : 1)
#endif
    for (i = 0; i < (long)n; i++) {
        uint32_t sid = shape_id ? shape_id[i] : 0;
        uint32_t v0 = vert_offsets[sid], nv = vert_offsets[sid + 1] - v0;
        step_impl(&bodies[i], verts + v0, nv, dt, substeps, masks ? masks + i : NULL, n);
    }
}
