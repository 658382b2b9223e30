// xpbd_step.hpp -- the per-body substep of solver::step (reference src/solver.rs:6-16) as device
// functions, shared by the fused stepper (xpbd_kernels.hip) and the contact pipeline
// (xpbd_contacts.hip).  All arithmetic keeps the reference's operation order.
#pragma once

#include "xpbd_device.hpp"

namespace xpbd {

// Per-body state held in registers.
struct BodyStatic {
    double inv_mass;
    Mat3 inv_inertia;
    Vec3 ext_force, int_force, ext_torque, int_torque;
    Vec3 com;
};

struct BodyDynamic {
    Vec3 pos;
    Quat rot;
    Vec3 vel;
    Vec3 ang;
};

__device__ __forceinline__ BodyStatic load_static(const BodyArrays &b, uint32_t i)
{
    const uint32_t st = b.stride;
    BodyStatic s;
    s.inv_mass = b.stat[(size_t)S_INV_MASS * st + i];
    s.inv_inertia.cx = load3(b.stat, S_INV_INERTIA + 0, st, i);
    s.inv_inertia.cy = load3(b.stat, S_INV_INERTIA + 3, st, i);
    s.inv_inertia.cz = load3(b.stat, S_INV_INERTIA + 6, st, i);
    s.ext_force = load3(b.stat, S_EXT_FORCE, st, i);
    s.int_force = load3(b.stat, S_INT_FORCE, st, i);
    s.ext_torque = load3(b.stat, S_EXT_TORQUE, st, i);
    s.int_torque = load3(b.stat, S_INT_TORQUE, st, i);
    s.com = load3(b.stat, S_COM, st, i);
    return s;
}

__device__ __forceinline__ BodyDynamic load_dynamic(const double *dyn, uint32_t st, uint32_t i)
{
    BodyDynamic d;
    d.pos = load3(dyn, D_POS, st, i);
    d.rot = load_quat(dyn, D_ROT, st, i);
    d.vel = load3(dyn, D_VEL, st, i);
    d.ang = load3(dyn, D_ANG, st, i);
    return d;
}

__device__ __forceinline__ void store_quat(double *base, uint32_t field, uint32_t st, uint32_t i, Quat q)
{
    base[(size_t)(field + 0) * st + i] = q.s;
    base[(size_t)(field + 1) * st + i] = q.x;
    base[(size_t)(field + 2) * st + i] = q.y;
    base[(size_t)(field + 3) * st + i] = q.z;
}

__device__ __forceinline__ void store_dynamic(double *dyn, uint32_t st, uint32_t i, const BodyDynamic &d)
{
    store3(dyn, D_POS, st, i, d.pos);
    store_quat(dyn, D_ROT, st, i, d.rot);
    store3(dyn, D_VEL, st, i, d.vel);
    store3(dyn, D_ANG, st, i, d.ang);
}

// What one substep keeps from before/after the integration.
struct SubstepFrames {
    Vec3 past_pos;  // src/solver.rs:7
    Quat past_rot;  // src/solver.rs:8
    Frame past;     // src/solver.rs:9  rigid.frame() before integrate
    Frame cur;      // rigid.frame() after integrate: frozen for the whole of ground() (src/collision.rs:17,24)
};

// src/solver.rs:7-10: remember the past pose, then Rigid::integrate (src/rigid.rs:82-99).
__device__ __forceinline__ SubstepFrames integrate_body(BodyDynamic &d, const BodyStatic &s, double h)
{
    SubstepFrames f;
    f.past_pos = d.pos;
    f.past_rot = d.rot;
    f.past = Frame{frame_origin(d.pos, d.rot, s.com), d.rot};

    const Vec3 force = s.ext_force + d.rot * s.int_force;
    d.vel = d.vel + (h * force) * s.inv_mass;
    d.pos = d.pos + h * d.vel;

    const Vec3 torque = s.ext_torque + d.rot * s.int_torque;
    d.ang = d.ang + (h * s.inv_inertia) * torque;
    const Quat dq = ((h * 0.5) * Quat{0.0, d.ang.x, d.ang.y, d.ang.z}) * d.rot;
    d.rot = normalized(d.rot + dq);

    f.cur = Frame{frame_origin(d.pos, d.rot, s.com), d.rot};
    return f;
}

// collision::ground + solver::solve for one body (src/collision.rs:13-35, src/solver.rs:19-27), in two passes.
//
// ground() only reads the post-integrate pose and the past frame, and solve() consumes the
// constraints in push order, so no constraint list is stored: constraint v is rebuilt from the
// frozen post-integrate frame `cur` and immediately projected onto the live pose (pos, rot).
// The arithmetic and its order per constraint are exactly the reference's.

// Pass 1 -- the `position.z >= 0.0` test of every vertex (src/collision.rs:17-18): bit v set <=> shape vertex v
// produces a constraint.  Only the z component of frame * vertex is live here, so the compiler drops the x/y
// arithmetic.  (A NaN height fails `>=` and therefore IS a contact, as in the reference.)
__device__ __forceinline__ uint32_t ground_mask(const Frame &cur, const double *verts, uint32_t n_verts)
{
    uint32_t mask = 0;
    for (uint32_t v = 0; v < n_verts; ++v) {
        const Vec3 vertex{verts[3 * v + 0], verts[3 * v + 1], verts[3 * v + 2]};
        const Vec3 x = cur * vertex;
        if (!(x.z >= 0.0))
            mask |= 1u << v;
    }
    return mask;
}

// Pass 2 -- the lane walks the penetrating vertices of `mask` in ascending index order (the reference's push
// order).  Lane-compacting the contact work this way makes a wave run the expensive body max-over-lanes(contact
// count) times instead of once per shape vertex with most lanes masked off.  Recomputing x for the chosen vertex
// repeats the pass-1 arithmetic exactly, so the bits match.  (pos, rot) is the live pose the impulses act on.
// `limit` (> 0 only in XPBD_MODE_CONTACTS with xpbd_world_set_max_depenetration_speed; the pinned path passes the literal 0
// and compiles to the reference's arithmetic alone): the length of a ground constraint's correction is limited to
// max(0, limit - what the vertex has already moved towards its target in this substep), limit = speed * h.
__device__ __forceinline__ void solve_masked(Vec3 &pos, Quat &rot, double inv_mass, const Mat3 &inv_inertia, const Vec3 &com,
                                             const Frame &cur, const Frame &past, double compliance, const double *verts,
                                             uint32_t mask, double limit = 0.0)
{
    const Frame cur_inv = inverse(cur); // src/frame.rs:30-37, shared by every penetrating vertex
    for (uint32_t todo = mask; todo != 0; todo &= todo - 1) {
        const uint32_t v = __ffs(todo) - 1;
        const Vec3 vertex{verts[3 * v + 0], verts[3 * v + 1], verts[3 * v + 2]};
        const Vec3 x = cur * vertex; // src/collision.rs:17

        // src/collision.rs:22-29
        const Vec3 target{x.x, x.y, 0.0};
        const Vec3 correction = target - x;
        const Vec3 local = cur_inv * x;      // src/frame.rs:41
        const Vec3 delta = x - past * local; // src/frame.rs:42-43
        const Vec3 delta_tangential = delta - project_on(delta, correction);
        const Vec3 c0 = x;
        const Vec3 c1 = target - 1.0 * delta_tangential;

        // solver::solve body, src/solver.rs:23-25 (distance == 0.0, src/collision.rs:30)
        const Vec3 difference = c1 - c0;                              // src/constraint.rs:13-15
        const double current_distance = length(difference);           // src/constraint.rs:21-23
        const Vec3 direction = difference * (1.0 / current_distance); // src/constraint.rs:17-19
        // inverse_resitance, src/constraint.rs:25-32 (reads the LIVE pose)
        const Vec3 angular_impulse = conjugate(rot) * cross(c0 - (pos + com), direction);
        const double w = inv_mass + dot(inv_inertia * angular_impulse, angular_impulse);
        double error = current_distance;
        if (limit > 0.0) {
            const double len = length(correction);
            const double closing = len > 0.0 ? dot(delta, correction) / len : 0.0;
            double allowed = limit - closing;
            if (!(allowed > 0.0))
                allowed = 0.0;
            if (current_distance > allowed)
                error = allowed;
        }
        const double lagrange = (error - 0.0) / (w + compliance);
        // act -> apply_impulse, src/constraint.rs:34-37, src/rigid.rs:113-123
        const Vec3 impulse = lagrange * direction;
        pos = pos + impulse * inv_mass;
        const Vec3 arm = c0 - (pos + com);
        const Quat spin = quat_sv(0.0, cross(inv_inertia * arm, impulse));
        rot = rot + (0.5 * spin) * rot;
        rot = normalized(rot);
    }
}

// Both passes for the lane's own body.  Returns the contact mask.
__device__ __forceinline__ uint32_t solve_ground(BodyDynamic &d, const BodyStatic &s, const SubstepFrames &f,
                                                 double compliance, const double *verts, uint32_t n_verts, double limit = 0.0)
{
    const uint32_t mask = ground_mask(f.cur, verts, n_verts);
    solve_masked(d.pos, d.rot, s.inv_mass, s.inv_inertia, s.com, f.cur, f.past, compliance, verts, mask, limit);
    return mask;
}

// Rigid::derive, src/rigid.rs:101-109
__device__ __forceinline__ void derive_body(BodyDynamic &d, Vec3 past_pos, Quat past_rot, double h)
{
    d.vel = (d.pos - past_pos) / h;
    Quat dr = d.rot * conjugate(past_rot);
    if (dr.s < 0.0)
        dr = -dr;
    d.ang = (2.0 * vec_of(dr)) / h;
}

// One whole substep of solver::step for one body (src/solver.rs:7-15).
__device__ __forceinline__ uint32_t substep(BodyDynamic &d, const BodyStatic &s, double h, double compliance,
                                            const double *verts, uint32_t n_verts)
{
    const SubstepFrames f = integrate_body(d, s, h);
    const uint32_t mask = solve_ground(d, s, f, compliance, verts, n_verts);
    derive_body(d, f.past_pos, f.past_rot, h);
    return mask;
}

} // namespace xpbd
