/*
 * xpbd_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A literal plain-C restatement of the reference's per-substep hot path
 * (jim-ec/constraint_solver, /root/reference/src/{solver,rigid,constraint,
 * collision,frame,geometry,geometry/integrate,world}.rs) and of the cgmath
 * 0.18.0 arithmetic it executes (Cargo.lock:372-375; crate not vendored).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (constraint_solver_amd/) never links it.
 *
 * PARITY STATUS: "parity unpinned" at the Rust-binary level.  The reference
 * ships no tests, fixtures or golden vectors and no Rust
 * toolchain exists in the build image, so this restatement cannot be checked
 * against reference outputs.  It is pinned only by analytic known-answer
 * tests derived from the reference's formulas (tests/test_oracle_kat.py,
 * SURVEY.md section 8c K1-K6).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (Rust never contracts a*b+c
 * into an FMA, so neither may the restatement).
 */
#ifndef XPBD_ORACLE_H
#define XPBD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double x, y, z; } o_vec3;
/* cgmath Quaternion::new(w, xi, yj, zk): scalar first in this mirror. */
typedef struct { double s; o_vec3 v; } o_quat;
/* cgmath Matrix3 {x, y, z}: three COLUMNS; m.x.y = column x, row y. */
typedef struct { o_vec3 x, y, z; } o_mat3;

/* frame.rs:8-11 */
typedef struct { o_vec3 position; o_quat rotation; } o_frame;
/* geometry.rs:9-12 */
typedef struct { o_vec3 normal; double displacement; } o_plane;

/* rigid.rs:6-50, field order kept, `color` dropped: 38 doubles. */
typedef struct {
    double inverse_mass;
    o_mat3 inverse_inertia;
    o_vec3 external_force;
    o_vec3 internal_force;
    o_vec3 external_torque;
    o_vec3 internal_torque;
    o_vec3 velocity;
    o_vec3 angular_velocity;
    o_vec3 center_of_mass;
    o_vec3 position;
    o_quat rotation;
} o_rigid;

/* constraint.rs:6-10 */
typedef struct {
    size_t rigid;
    o_vec3 contact0, contact1;
    double distance;
} o_constraint;

/* geometry/integrate.rs:18-24 */
typedef struct {
    double mass, volume;
    o_vec3 center_of_mass;
    o_mat3 inertia_tensor;
} o_metrics;

/* geometry.rs:82-93.  Faces are CSR (face_offsets has n_faces+1 entries). */
#define O_MAX_VERTS 32
#define O_MAX_EDGES 64
#define O_MAX_FACES 32
#define O_MAX_FACE_IDX 128
typedef struct {
    uint32_t n_vertices, n_edges, n_faces;
    o_vec3 vertices[O_MAX_VERTS];
    uint32_t edges[O_MAX_EDGES][2];
    uint32_t face_offsets[O_MAX_FACES + 1];
    uint32_t face_indices[O_MAX_FACE_IDX];
    o_vec3 centroid;
} o_polytope;

/* ---- cgmath 0.18 primitives (exported so the KATs can pin them) ---- */
o_vec3 o_add(o_vec3 a, o_vec3 b);
o_vec3 o_sub(o_vec3 a, o_vec3 b);
o_vec3 o_neg(o_vec3 a);
o_vec3 o_scale(o_vec3 a, double s);      /* Vector3 * s  */
o_vec3 o_lscale(double s, o_vec3 a);     /* s * Vector3  */
o_vec3 o_divs(o_vec3 a, double s);       /* Vector3 / s  */
double o_dot(o_vec3 a, o_vec3 b);
o_vec3 o_cross(o_vec3 a, o_vec3 b);
double o_magnitude2(o_vec3 a);
double o_magnitude(o_vec3 a);
o_vec3 o_normalize(o_vec3 a);
o_vec3 o_project_on(o_vec3 a, o_vec3 onto);
o_quat o_qmul(o_quat a, o_quat b);
o_vec3 o_qrot(o_quat q, o_vec3 v);       /* Quaternion * Vector3 */
o_quat o_qconj(o_quat q);
o_quat o_qadd(o_quat a, o_quat b);
o_quat o_qlscale(double s, o_quat q);
o_quat o_qneg(o_quat q);
o_quat o_qnormalize(o_quat q);
o_quat o_quat_from_euler_deg(double x_deg, double y_deg, double z_deg);
o_vec3 o_mat3_mulv(o_mat3 m, o_vec3 v);
o_mat3 o_mat3_lscale(double s, o_mat3 m);
int    o_mat3_invert(o_mat3 m, o_mat3 *out);   /* 0 = singular (None) */

/* ---- frame.rs ---- */
o_frame o_frame_inverse(o_frame f);                       /* :30-37 */
o_vec3  o_frame_delta(o_frame f, o_frame past, o_vec3 g); /* :40-44 */
o_vec3  o_frame_mulv(o_frame f, o_vec3 v);                /* :47-53 */
o_plane o_frame_mulplane(o_frame f, o_plane p);           /* :55-64 */
o_frame o_frame_mul(o_frame a, o_frame b);                /* :67-76 */

/* ---- geometry.rs ---- */
o_plane o_plane_from_points(o_vec3 p0, o_vec3 p1, o_vec3 p2); /* :16-24 */
o_plane o_plane_from_point_normal(o_vec3 p, o_vec3 n);        /* :27-36 */
double  o_plane_distance(o_plane pl, o_vec3 p);               /* :39-41 */
void    o_polytope_tetrahedron(o_polytope *p);                /* :97-109 */
void    o_polytope_cube(o_polytope *p);                       /* :113-149 */
void    o_polytope_icosahedron(o_polytope *p);                /* :153-230 */
void    o_polytope_scale(double s, o_polytope *p);            /* :296-307 */
o_plane o_polytope_plane(const o_polytope *p, uint32_t i);    /* :262-271 */
o_vec3  o_polytope_support(const o_polytope *p, o_frame f, o_vec3 d);           /* :274-281 */
o_vec3  o_polytope_minkowski_support(const o_polytope *p, o_frame f0, o_frame f1, o_vec3 d); /* :283-289 */
/* geometry/integrate.rs:26-75 */
void    o_rigid_metrics(const o_polytope *p, double density, o_metrics *out);

/* ---- rigid.rs ---- */
int     o_rigid_new(const o_metrics *m, o_rigid *out);        /* :53-71, 0 = singular inertia (panic) */
o_frame o_rigid_frame(const o_rigid *r);                      /* :75-80 */
void    o_rigid_integrate(o_rigid *r, double dt);             /* :82-99 */
void    o_rigid_derive(o_rigid *r, o_vec3 p0, o_quat q0, double dt); /* :101-109 */
void    o_rigid_apply_impulse(o_rigid *r, o_vec3 impulse, o_vec3 point); /* :113-123 */

/* ---- constraint.rs ---- */
double  o_constraint_current_distance(const o_constraint *c);            /* :21-23 */
double  o_constraint_inverse_resistance(const o_constraint *c, const o_rigid *const *rigids); /* :25-32 */
void    o_constraint_act(const o_constraint *c, o_rigid *const *rigids, double factor);       /* :34-37 */

/* ---- collision.rs ---- */
/* ground (:13-35). Writes up to n_vertices constraints to out[], the shape
 * vertex index of each to out_vertex[] (may be NULL); returns the count. */
uint32_t o_ground(const o_rigid *r, o_frame past, const o_vec3 *vertices,
                  uint32_t n_vertices, o_constraint *out, uint32_t *out_vertex);
/* face_axes_separation (:123-149): returns distance, writes face index. */
double  o_face_axes_separation(o_frame fa, o_frame fb, const o_polytope *pa,
                               const o_polytope *pb, uint64_t *face_index);
/* edge_axes_separation (:151-197) */
double  o_edge_axes_separation(o_frame fa, o_frame fb, const o_polytope *pa,
                               const o_polytope *pb, uint64_t *edge_a, uint64_t *edge_b);

/* ---- solver.rs ---- */
void o_solve(o_rigid *r, const o_constraint *cs, uint32_t n, double dt);  /* :19-27 */
/* step (:3-17).  If masks != NULL, masks[k] receives for substep k the bitmask
 * of shape vertices that produced a ground constraint (bit v = vertex v), i.e.
 * the reference's constraint push order read as a set. */
void o_step(o_rigid *r, const o_vec3 *vertices, uint32_t n_vertices, double dt,
            size_t substep_count, uint32_t *masks);

/* ---- world.rs ---- */
/* World::new (:12-31): a from p1@0.1, b from p2@5.0 + hard-coded state. */
int  o_world_new(const o_polytope *p1, const o_polytope *p2, o_rigid *a, o_rigid *b);
/* World::integrate (:34-43): both bodies collide as p1, 25 substeps. */
void o_world_integrate(o_rigid *a, o_rigid *b, double dt, const o_polytope *p1);

/* ---- batch helper used by parity tests and bench.py cpu_baseline ---- */
/* For each body i: solver::step(bodies[i], shape[shape_id[i]], dt, substeps).
 * verts: all shapes' vertices back to back; vert_offsets: CSR, n_shapes+1.
 * masks (optional): [substeps][n] row-major contact masks.
 * threads <= 1: plain serial loop (faithful to the single-threaded reference).
 * threads > 1: OpenMP static partition over bodies (bodies are independent). */
void o_step_bodies(o_rigid *bodies, const uint32_t *shape_id, uint32_t n,
                   const double *verts_xyz, const uint32_t *vert_offsets,
                   double dt, uint32_t substeps, uint32_t *masks, int threads);

#ifdef __cplusplus
}
#endif
#endif
