// constraint_solver.hpp -- host-side mirror of the reference's World / Rigid /
// Polytope / solver API, sitting above the C ABI (include/xpbd.h).
//
// The reference's host is Rust; no Rust toolchain exists in the build image,
// so this C++17 header plays the role the Rust host would: it keeps the
// reference's module and function names and argument meaning, does the
// set-up math on the CPU (shape tables, Mirtich mass properties, Rigid::new,
// World::new) and hands the per-substep hot path to the HIP library through
// the same extern "C" entry points a Rust `extern` block would bind
// (INTEGRATION.md shows that binding).
//
//   geometry::Plane / Polytope      src/geometry.rs:9-78, 82-307
//   geometry::rigid_metrics         src/geometry/integrate.rs:26-288
//   rigid::Rigid                    src/rigid.rs:6-80
//   solver::step                    src/solver.rs:3      -> xpbd_step_one
//   world::World                    src/world.rs:6-43    -> xpbd_step_one x2
//   world::BatchWorld               N-body generalisation -> xpbd_world_*
//
// Compile with -ffp-contract=off (see xpbd_math.hpp).
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/xpbd.h"
#include "../csrc/xpbd_math.hpp"

namespace constraint_solver {

using xpbd::Frame;
using xpbd::Mat3;
using xpbd::Quat;
using xpbd::Vec3;

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

inline void check(int rc)
{
    if (rc != XPBD_OK)
        throw Error(rc, xpbd_last_error());
}

// ===========================================================================
namespace geometry {

// src/geometry.rs:9-78
struct Plane {
    Vec3 normal{0.0, 0.0, 0.0};
    double displacement = 0.0;

    static Plane from_points(Vec3 p0, Vec3 p1, Vec3 p2)
    {
        const Vec3 n = xpbd::normalized(xpbd::cross(p1 - p0, p2 - p0));
        return Plane{n, xpbd::dot(n, p0)};
    }
    static Plane from_point_normal(Vec3 point, Vec3 normal)
    {
        double d = xpbd::length(xpbd::project_on(point, normal));
        if (xpbd::dot(point, normal) < 0.0)
            d *= -1.0;
        return Plane{normal, d};
    }
    double distance(Vec3 p) const { return xpbd::dot(normal, p) - displacement; }
    Vec3 project(Vec3 p) const { return p - distance(p) * normal; }
    double constant() const { return -displacement; }
    Plane flip() const { return Plane{-normal, -displacement}; }
    Vec3 support() const { return displacement * normal; }
    bool facing(Vec3 p) const { return xpbd::dot(normal, p - support()) >= 0.0; }
};

// src/frame.rs:55-64
inline Plane operator*(const Frame &f, const Plane &p)
{
    return Plane::from_point_normal(f * (p.displacement * p.normal), f.rotation * p.normal);
}

// src/geometry/integrate.rs:18-24
struct RigidMetrics {
    double mass, volume;
    Vec3 center_of_mass;
    Mat3 inertia_tensor;
};

// IEEE totalOrder, as f64::total_cmp.
inline bool total_le(double a, double b)
{
    auto key = [](double v) {
        int64_t i;
        static_assert(sizeof i == sizeof v, "");
        __builtin_memcpy(&i, &v, 8);
        return i ^ (int64_t)((uint64_t)(i >> 63) >> 1);
    };
    return key(a) <= key(b);
}

// src/geometry.rs:82-93
struct Polytope {
    std::vector<Vec3> vertices;
    std::vector<std::array<uint32_t, 2>> edges;
    std::vector<std::vector<uint32_t>> faces;
    Vec3 centroid{0.0, 0.0, 0.0};

    // src/geometry.rs:97-109
    static Polytope new_tetrahedron()
    {
        Polytope p;
        p.centroid = Vec3{0.25, 0.25, 0.25};
        p.vertices = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        p.edges = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
        p.faces = {{0, 3, 2}, {3, 0, 1}, {2, 1, 0}, {1, 2, 3}};
        return p;
    }
    // src/geometry.rs:113-149
    static Polytope new_cube()
    {
        Polytope p;
        p.centroid = Vec3{0.5, 0.5, 0.5};
        p.vertices = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {1, 1, 0}, {0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
        p.edges = {{0, 1}, {1, 3}, {3, 2}, {2, 0}, {4, 5}, {5, 7}, {7, 6}, {6, 4}, {0, 4}, {1, 5}, {3, 7}, {2, 6}};
        p.faces = {{0, 2, 3, 1}, {4, 5, 7, 6}, {4, 0, 1, 5}, {5, 1, 3, 7}, {7, 3, 2, 6}, {6, 2, 0, 4}};
        return p;
    }
    // src/geometry.rs:153-230
    static Polytope new_icosahedron()
    {
        const double phi = (1.0 + std::sqrt(5.0)) / 2.0;
        const double mag = std::sqrt(phi * phi + 1.0);
        const double a = phi / mag, b = 1.0 / mag;
        Polytope p;
        p.centroid = Vec3{0, 0, 0};
        p.vertices = {{a, b, 0},  {a, -b, 0}, {-a, b, 0}, {-a, -b, 0}, {0, a, b},  {0, a, -b},
                      {0, -a, b}, {0, -a, -b}, {b, 0, a}, {-b, 0, a},  {b, 0, -a}, {-b, 0, -a}};
        p.edges = {{8, 9}, {8, 0},  {8, 1},  {1, 0}, {9, 2}, {9, 3}, {2, 3}, {2, 5}, {2, 11}, {5, 11},
                   {3, 7}, {3, 11}, {7, 11}, {0, 5}, {1, 7}, {4, 5}, {4, 0}, {4, 8}, {4, 9},  {4, 2},
                   {6, 9}, {6, 8},  {6, 1},  {6, 7}, {6, 3}, {10, 1}, {10, 0}, {10, 5}, {10, 11}, {10, 7}};
        p.faces = {{0, 5, 4},  {2, 4, 5},   {1, 6, 7},   {3, 7, 6},  {1, 0, 8},  {0, 1, 10}, {2, 3, 9},
                   {3, 2, 11}, {4, 9, 8},   {6, 8, 9},   {5, 10, 11}, {7, 11, 10}, {0, 4, 8},  {0, 10, 5},
                   {2, 9, 4},  {2, 5, 11},  {1, 8, 6},   {1, 7, 10}, {3, 6, 9},  {3, 11, 7}};
        return p;
    }

    // src/geometry.rs:262-271: outward plane of face i.
    Plane plane(size_t i) const
    {
        const auto &f = faces.at(i);
        const Plane pl = Plane::from_points(vertices[f[0]], vertices[f[1]], vertices[f[2]]);
        return !pl.facing(centroid) ? pl : pl.flip();
    }

    // src/geometry.rs:274-281: world-space support point; last maximum wins (Iterator::max_by).
    Vec3 support(const Frame &frame, Vec3 direction) const
    {
        if (vertices.empty())
            throw Error(XPBD_E_INVALID, "support of an empty polytope"); // .unwrap() panic
        Vec3 best = frame * vertices[0];
        for (size_t i = 1; i < vertices.size(); ++i) {
            const Vec3 x = frame * vertices[i];
            if (total_le(xpbd::dot(best, direction), xpbd::dot(x, direction)))
                best = x;
        }
        return best;
    }

    // src/geometry.rs:283-289 (same polytope under both frames, as written)
    Vec3 minkowski_support(const Frame &f0, const Frame &f1, Vec3 direction) const
    {
        return support(f0, direction) - support(f1, -direction);
    }

    RigidMetrics rigid_metrics(double density) const; // src/geometry.rs:291-293
};

// src/geometry.rs:296-307
inline Polytope operator*(double k, Polytope p)
{
    for (Vec3 &v : p.vertices)
        v = k * v;
    p.centroid = k * p.centroid;
    return p;
}

// --- Mirtich 1996 polyhedral mass properties, src/geometry/integrate.rs ------
namespace detail {

inline double axis(const Vec3 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
inline double &axis_ref(Vec3 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

struct Projection { double one = 0, a = 0, b = 0, aa = 0, ab = 0, bb = 0, aaa = 0, aab = 0, abb = 0, bbb = 0; };

// src/geometry/integrate.rs:223-288
inline Projection projection_integrals(const Polytope &poly, const std::vector<uint32_t> &loop, int A, int B)
{
    Projection P;
    const size_t n = loop.size();
    for (size_t i = 0; i < n; ++i) {
        const double a0 = axis(poly.vertices[loop[i]], A), b0 = axis(poly.vertices[loop[i]], B);
        const double a1 = axis(poly.vertices[loop[(i + 1) % n]], A), b1 = axis(poly.vertices[loop[(i + 1) % n]], B);
        const double da = a1 - a0, db = b1 - b0;
        const double a0_2 = a0 * a0, a0_3 = a0_2 * a0, a0_4 = a0_3 * a0;
        const double b0_2 = b0 * b0, b0_3 = b0_2 * b0, b0_4 = b0_3 * b0;
        const double a1_2 = a1 * a1, a1_3 = a1_2 * a1;
        const double b1_2 = b1 * b1, b1_3 = b1_2 * b1;

        const double C1 = a1 + a0;
        const double Ca = a1 * C1 + a0_2, Caa = a1 * Ca + a0_3, Caaa = a1 * Caa + a0_4;
        const double Cb = b1 * (b1 + b0) + b0_2, Cbb = b1 * Cb + b0_3, Cbbb = b1 * Cbb + b0_4;
        const double Cab = 3.0 * a1_2 + 2.0 * a1 * a0 + a0_2, Kab = a1_2 + 2.0 * a1 * a0 + 3.0 * a0_2;
        const double Caab = a0 * Cab + 4.0 * a1_3, Kaab = a1 * Kab + 4.0 * a0_3;
        const double Cabb = 4.0 * b1_3 + 3.0 * b1_2 * b0 + 2.0 * b1 * b0_2 + b0_3;
        const double Kabb = b1_3 + 2.0 * b1_2 * b0 + 3.0 * b1 * b0_2 + 4.0 * b0_3;

        P.one += db * C1;
        P.a += db * Ca;
        P.aa += db * Caa;
        P.aaa += db * Caaa;
        P.b += da * Cb;
        P.bb += da * Cbb;
        P.bbb += da * Cbbb;
        P.ab += db * (b1 * Cab + b0 * Kab);
        P.aab += db * (b1 * Caab + b0 * Kaab);
        P.abb += da * (a1 * Cabb + a0 * Kabb);
    }
    P.one /= 2.0;
    P.a /= 6.0;
    P.aa /= 12.0;
    P.aaa /= 20.0;
    P.b /= -6.0;
    P.bb /= -12.0;
    P.bbb /= -20.0;
    P.ab /= 24.0;
    P.aab /= 60.0;
    P.abb /= -60.0;
    return P;
}

struct FaceIntegrals { double a, b, c, aa, bb, cc, aaa, bbb, ccc, aab, bbc, cca; };

// src/geometry/integrate.rs:171-220;  sq = powi(2) = x*x, cb = powi(3) = x*x*x
inline FaceIntegrals face_integrals(const Polytope &poly, const std::vector<uint32_t> &loop, Vec3 n, double w,
                                    int A, int B, int C)
{
    const Projection P = projection_integrals(poly, loop, A, B);
    auto sq = [](double x) { return x * x; };
    auto cb = [](double x) { return x * x * x; };
    const double na = axis(n, A), nb = axis(n, B);
    const double k1 = 1.0 / axis(n, C), k2 = k1 * k1, k3 = k2 * k1, k4 = k3 * k1;
    FaceIntegrals F;
    F.a = k1 * P.a;
    F.b = k1 * P.b;
    F.c = -k2 * (na * P.a + nb * P.b + w * P.one);
    F.aa = k1 * P.aa;
    F.bb = k1 * P.bb;
    F.cc = k3 * (sq(na) * P.aa + 2.0 * na * nb * P.ab + sq(nb) * P.bb + w * (2.0 * (na * P.a + nb * P.b) + w * P.one));
    F.aaa = k1 * P.aaa;
    F.bbb = k1 * P.bbb;
    F.ccc = -k4 * (cb(na) * P.aaa + 3.0 * sq(na) * nb * P.aab + 3.0 * na * sq(nb) * P.abb + cb(nb) * P.bbb
                   + 3.0 * w * (sq(na) * P.aa + 2.0 * na * nb * P.ab + sq(nb) * P.bb)
                   + w * w * (3.0 * (na * P.a + nb * P.b) + w * P.one));
    F.aab = k1 * P.aab;
    F.bbc = -k2 * (na * P.abb + nb * P.bbb + w * P.bb);
    F.cca = k3 * (sq(na) * P.aaa + 2.0 * na * nb * P.aab + sq(nb) * P.abb + w * (2.0 * (na * P.aa + nb * P.ab) + w * P.a));
    return F;
}

} // namespace detail

// src/geometry/integrate.rs:26-75 with volume_integrals (:126-169)
inline RigidMetrics rigid_metrics(const Polytope &poly, double density)
{
    double T0 = 0.0;
    Vec3 T1{0, 0, 0}, T2{0, 0, 0}, TP{0, 0, 0};
    for (size_t i = 0; i < poly.faces.size(); ++i) {
        const Plane pl = poly.plane(i);
        const Vec3 n = pl.normal;
        const double w = pl.constant(); // Face.displacement = plane.constant(), :35
        const double ax = std::fabs(n.x), ay = std::fabs(n.y), az = std::fabs(n.z);
        const int C = (ax > ay && ax > az) ? 0 : (ay > az ? 1 : 2);
        const int A = (C + 1) % 3, B = (A + 1) % 3;
        const detail::FaceIntegrals F = detail::face_integrals(poly, poly.faces[i], n, w, A, B, C);

        T0 += n.x * (A == 0 ? F.a : (B == 0 ? F.b : F.c));
        detail::axis_ref(T1, A) += detail::axis(n, A) * F.aa;
        detail::axis_ref(T1, B) += detail::axis(n, B) * F.bb;
        detail::axis_ref(T1, C) += detail::axis(n, C) * F.cc;
        detail::axis_ref(T2, A) += detail::axis(n, A) * F.aaa;
        detail::axis_ref(T2, B) += detail::axis(n, B) * F.bbb;
        detail::axis_ref(T2, C) += detail::axis(n, C) * F.ccc;
        detail::axis_ref(TP, A) += detail::axis(n, A) * F.aab;
        detail::axis_ref(TP, B) += detail::axis(n, B) * F.bbc;
        detail::axis_ref(TP, C) += detail::axis(n, C) * F.cca;
    }
    T1 = T1 / 2.0;
    T2 = T2 / 3.0;
    TP = TP / 2.0;

    const double m = density * T0;
    const Vec3 r = T1 / T0;
    Mat3 J{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    J.cx.x = density * (T2.y + T2.z);
    J.cy.y = density * (T2.z + T2.x);
    J.cz.z = density * (T2.x + T2.y);
    J.cx.y = -density * TP.x;
    J.cy.z = -density * TP.y;
    J.cz.x = -density * TP.z;
    J.cy.x = J.cx.y;
    J.cz.y = J.cy.z;
    J.cx.z = J.cz.x;

    J.cx.x -= m * (r.y * r.y + r.z * r.z);
    J.cy.y -= m * (r.z * r.z + r.x * r.x);
    J.cz.z -= m * (r.x * r.x + r.y * r.y);
    J.cx.y += m * r.x * r.y;
    J.cy.z += m * r.y * r.z;
    J.cz.x += m * r.z * r.x;
    J.cy.x = J.cx.y;
    J.cz.y = J.cy.z;
    J.cx.z = J.cz.x;
    return RigidMetrics{m, T0, r, J};
}

inline RigidMetrics Polytope::rigid_metrics(double density) const { return geometry::rigid_metrics(*this, density); }

} // namespace geometry

// ===========================================================================
namespace rigid {

// src/rigid.rs:6-50, `color` dropped.  Same memory layout as xpbd_rigid.
struct Rigid {
    double inverse_mass;
    Mat3 inverse_inertia;
    Vec3 external_force, internal_force, external_torque, internal_torque;
    Vec3 velocity, angular_velocity, center_of_mass, position;
    Quat rotation;

    // src/rigid.rs:53-71; throws where the reference panics on a singular tensor.
    static Rigid make(const geometry::RigidMetrics &m)
    {
        Rigid r{};
        r.inverse_mass = 1.0 / m.mass;
        if (!xpbd::invert(m.inertia_tensor, r.inverse_inertia))
            throw Error(XPBD_E_SINGULAR_INERTIA, "Inertia tensor is not invertible");
        r.center_of_mass = m.center_of_mass;
        r.rotation = Quat{1.0, 0.0, 0.0, 0.0};
        return r;
    }
    // src/rigid.rs:75-80
    Frame frame() const
    {
        return Frame{position + center_of_mass + rotation * -center_of_mass, rotation};
    }
    xpbd_rigid *c() { return reinterpret_cast<xpbd_rigid *>(this); }
    const xpbd_rigid *c() const { return reinterpret_cast<const xpbd_rigid *>(this); }
};
static_assert(sizeof(Rigid) == sizeof(xpbd_rigid), "Rigid must mirror xpbd_rigid");
static_assert(sizeof(xpbd_rigid) == 38 * sizeof(double), "xpbd_rigid is 38 doubles");

} // namespace rigid

// ===========================================================================
namespace solver {

// src/solver.rs:3 -- the drop-in: the substep loop runs on the GPU.
inline void step(rigid::Rigid &r, const geometry::Polytope &polytope, double dt, size_t substep_count)
{
    check(xpbd_step_one(r.c(), polytope.vertices.empty() ? nullptr : &polytope.vertices[0].x,
                        (uint32_t)polytope.vertices.size(), dt, (uint32_t)substep_count));
}

} // namespace solver

// ===========================================================================
namespace world {

// cgmath From<Euler<Deg>> for Quaternion (XYZ), used at src/world.rs:28.
inline Quat quat_from_euler_deg(double x, double y, double z)
{
    const double k = 3.14159265358979323846264338327950288 / 180.0;
    const double hx = (x * k) * 0.5, hy = (y * k) * 0.5, hz = (z * k) * 0.5;
    const double sx = std::sin(hx), cx = std::cos(hx), sy = std::sin(hy), cy = std::cos(hy);
    const double sz = std::sin(hz), cz = std::cos(hz);
    return Quat{-sx * sy * sz + cx * cy * cz, sx * cy * cz + sy * sz * cx, -sx * sz * cy + sy * cx * cz,
                sx * sy * cz + sz * cx * cy};
}

// src/world.rs:6-43
struct World {
    rigid::Rigid a, b;

    World(const geometry::Polytope &p1, const geometry::Polytope &p2)
        : a(rigid::Rigid::make(p1.rigid_metrics(0.1))), b(rigid::Rigid::make(p2.rigid_metrics(5.0)))
    {
        a.position.z = 4.0;
        a.velocity.y = 2.5;
        a.angular_velocity.x = -4.0;
        a.angular_velocity.y = 1.0;
        a.external_force.z = -2.0;

        b.position.x = 4.0;
        b.position.z = 4.0;
        b.velocity.z = 7.0;
        b.angular_velocity.x = -5.0;
        b.angular_velocity.y = 5.0;
        b.external_force.z = -2.0;
        b.rotation = quat_from_euler_deg(10.0, 15.0, 5.0);
    }

    // src/world.rs:34-43: both bodies collide as p1, 25 substeps (as written).
    void integrate(double dt, const geometry::Polytope &p1, const geometry::Polytope & /*p2*/)
    {
        solver::step(a, p1, dt, 25);
        solver::step(b, p1, dt, 25);
    }
};

// xpbd_polytope descriptors of a list of Polytopes (the descriptors point into the object's own arrays).
struct PolytopeDescs {
    std::vector<std::vector<double>> verts;
    std::vector<std::vector<uint32_t>> edges, foff, fidx;
    std::vector<xpbd_polytope> desc;
    explicit PolytopeDescs(const std::vector<geometry::Polytope> &shapes)
        : verts(shapes.size()), edges(shapes.size()), foff(shapes.size()), fidx(shapes.size()), desc(shapes.size())
    {
        for (size_t s = 0; s < shapes.size(); ++s) {
            const geometry::Polytope &p = shapes[s];
            for (const Vec3 &x : p.vertices)
                verts[s].insert(verts[s].end(), {x.x, x.y, x.z});
            for (const auto &e : p.edges)
                edges[s].insert(edges[s].end(), {e[0], e[1]});
            foff[s].push_back(0);
            for (const auto &f : p.faces) {
                fidx[s].insert(fidx[s].end(), f.begin(), f.end());
                foff[s].push_back((uint32_t)fidx[s].size());
            }
            desc[s] = xpbd_polytope{verts[s].data(), edges[s].data(), foff[s].data(), fidx[s].data(),
                                    (uint32_t)p.vertices.size(), (uint32_t)p.edges.size(), (uint32_t)p.faces.size(), 0,
                                    {p.centroid.x, p.centroid.y, p.centroid.z}};
        }
    }
};

// N-body generalisation of World: one batched xpbd_world per GPU.
class BatchWorld {
  public:
    explicit BatchWorld(const xpbd_config *cfg = nullptr) { check(xpbd_world_create(&w_, cfg)); }
    ~BatchWorld() { xpbd_world_destroy(w_); }
    BatchWorld(const BatchWorld &) = delete;
    BatchWorld &operator=(const BatchWorld &) = delete;

    void set_shapes(const std::vector<geometry::Polytope> &shapes)
    {
        std::vector<double> v;
        std::vector<uint32_t> off{0};
        for (const auto &p : shapes) {
            for (const Vec3 &x : p.vertices) {
                v.push_back(x.x);
                v.push_back(x.y);
                v.push_back(x.z);
            }
            off.push_back((uint32_t)(v.size() / 3));
        }
        static const double none[3] = {0, 0, 0};
        check(xpbd_world_set_shapes(w_, v.empty() ? none : v.data(), off.data(), (uint32_t)shapes.size()));
    }
    // EXTENSION: full topology, needed by XPBD_MODE_CONTACTS (body-body contacts; not in the reference).
    void set_polytopes(const std::vector<geometry::Polytope> &shapes)
    {
        const PolytopeDescs d(shapes);
        check(xpbd_world_set_polytopes(w_, d.desc.data(), (uint32_t)d.desc.size()));
    }
    void upload(const std::vector<rigid::Rigid> &bodies, const std::vector<uint32_t> &shape_id)
    {
        if (!shape_id.empty() && shape_id.size() != bodies.size())
            throw Error(XPBD_E_INVALID, "shape_id size mismatch");
        check(xpbd_world_upload_bodies(w_, bodies.empty() ? nullptr : bodies[0].c(),
                                       shape_id.empty() ? nullptr : shape_id.data(), (uint32_t)bodies.size()));
    }
    // for each body: solver::step(body, shape, dt, substeps)
    void integrate(double dt, uint32_t substeps) { check(xpbd_world_step(w_, dt, substeps)); }
    void synchronize() { check(xpbd_world_synchronize(w_)); }
    void download(std::vector<rigid::Rigid> &bodies)
    {
        bodies.resize(xpbd_world_body_count(w_));
        check(xpbd_world_download_bodies(w_, bodies.empty() ? nullptr : bodies[0].c(), (uint32_t)bodies.size()));
    }
    // state history on the device (the reference app's `states` vector, src/app.rs:48)
    uint32_t history_push()
    {
        uint32_t index = 0;
        check(xpbd_world_history_push(w_, &index));
        return index;
    }
    void history_restore(uint32_t index) { check(xpbd_world_history_restore(w_, index)); }
    void history_truncate(uint32_t length) { check(xpbd_world_history_truncate(w_, length)); }
    uint32_t history_length() const { return xpbd_world_history_length(w_); }

    std::vector<xpbd_contact> contacts()
    {
        uint32_t n = 0;
        int rc = xpbd_world_download_contacts(w_, nullptr, 0, &n);
        if (rc != XPBD_OK && rc != XPBD_E_CAPACITY)
            check(rc);
        std::vector<xpbd_contact> out(n);
        if (n)
            check(xpbd_world_download_contacts(w_, out.data(), n, &n));
        return out;
    }
    xpbd_world *handle() { return w_; }

  private:
    xpbd_world *w_ = nullptr;
};

// EXTENSION: the N-body contact world sharded over the GPUs of one node, all shards driven by THIS process -- what a
// Rust `World::integrate` (src/world.rs:34-43) would hold in place of its two `solver::step` calls.  One call per frame;
// the library owns a stream and an RCCL communicator per device, plans the halos, exchanges them after every substep and
// throws XPBD_E_HALO when a body outruns the halo margin (automatic re-plans by default).  With fewer visible devices than
// shards the shards share `first_device` and exchange by peer copies (XPBD_TRANSPORT_LOCAL): the one-GPU rehearsal.
class ShardedWorld {
  public:
    ShardedWorld(uint32_t n_shards, int first_device = 0, double halo_margin = 0.5, bool auto_replan = true, double contact_pad = 0.02,
                 uint32_t narrowphase = XPBD_NARROWPHASE_SAT)
    {
        const int visible = xpbd_device_count();
        const bool one_each = visible >= first_device + (int)n_shards;
        std::vector<int32_t> devices(n_shards);
        for (uint32_t k = 0; k < n_shards; ++k)
            devices[k] = one_each ? first_device + (int)k : first_device;
        uint8_t id[XPBD_COMM_ID_BYTES] = {0};
        xpbd_multi_config cfg;
        xpbd_multi_config_default(&cfg);
        cfg.n_ranks = cfg.n_local = n_shards;
        cfg.devices = devices.data();
        cfg.transport = one_each && n_shards > 1 ? XPBD_TRANSPORT_RCCL : XPBD_TRANSPORT_LOCAL;
        if (cfg.transport == XPBD_TRANSPORT_RCCL) {
            check(xpbd_comm_unique_id(id));
            cfg.comm_id = id;
        }
        cfg.flags = auto_replan ? XPBD_MULTI_AUTO_REPLAN : 0u;
        cfg.contact_pad = contact_pad;
        cfg.halo_margin = halo_margin;
        cfg.narrowphase = narrowphase;
        check(xpbd_multi_world_create(&w_, &cfg));
        rccl_ = cfg.transport == XPBD_TRANSPORT_RCCL;
    }
    ~ShardedWorld() { xpbd_multi_world_destroy(w_); }
    ShardedWorld(const ShardedWorld &) = delete;
    ShardedWorld &operator=(const ShardedWorld &) = delete;

    void set_polytopes(const std::vector<geometry::Polytope> &shapes)
    {
        const PolytopeDescs d(shapes);
        check(xpbd_multi_world_set_polytopes(w_, d.desc.data(), (uint32_t)d.desc.size()));
    }
    // bodies in an order whose contiguous index ranges are compact in space (grid rows, spatial-hash cell order)
    void upload(const std::vector<rigid::Rigid> &bodies, const std::vector<uint32_t> &shape_id, const std::vector<xpbd_joint> &joints = {})
    {
        if (!shape_id.empty() && shape_id.size() != bodies.size())
            throw Error(XPBD_E_INVALID, "shape_id size mismatch");
        n_ = (uint32_t)bodies.size();
        check(xpbd_multi_world_upload(w_, bodies.empty() ? nullptr : bodies[0].c(), shape_id.empty() ? nullptr : shape_id.data(), 0, n_, n_,
                                      joints.empty() ? nullptr : joints.data(), (uint32_t)joints.size()));
    }
    void integrate(double dt, uint32_t substeps) { check(xpbd_multi_world_step(w_, dt, substeps)); }
    void replan() { check(xpbd_multi_world_replan(w_)); }
    void synchronize() { check(xpbd_multi_world_synchronize(w_)); }
    void download(std::vector<rigid::Rigid> &bodies)
    {
        bodies.resize(n_);
        check(xpbd_multi_world_download(w_, bodies.empty() ? nullptr : bodies[0].c(), n_));
    }
    // {bodies of the world, owned here, ghosts, boundary bodies, rows per rank of the all-gather, plans made}
    std::vector<uint64_t> halo_stats(double *max_displacement = nullptr)
    {
        std::vector<uint64_t> out(6);
        check(xpbd_multi_world_halo_stats(w_, out.data(), max_displacement));
        return out;
    }
    bool over_rccl() const { return rccl_; }

  private:
    xpbd_multi_world *w_ = nullptr;
    uint32_t n_ = 0;
    bool rccl_ = false;
};

// The reference app's state history (src/app.rs:48, 206-212): `states` lives on the device as the world's history,
// `current_state` is the host-side cursor; advance() is the body of the RedrawRequested loop.
class Timeline {
  public:
    explicit Timeline(BatchWorld &world) : world_(world) { resident_ = world_.history_push(); } // states = vec![World::new(..)]
    void advance(double dt, uint32_t substeps, uint32_t time_speed = 1)
    {
        for (uint32_t k = 0; k < time_speed; ++k) {
            const uint32_t len = world_.history_length();
            if (current_state + 1 >= len) {
                if (resident_ != len - 1)
                    world_.history_restore(len - 1); // `let mut world = states[current_state].0` (the newest one here)
                world_.integrate(dt, substeps);
                resident_ = world_.history_push();
            }
            ++current_state;
        }
    }
    // scrubbing: make state `index` the current one (read it back with BatchWorld::download / frames)
    void seek(uint32_t index)
    {
        world_.history_restore(index);
        current_state = resident_ = index;
    }
    uint32_t current_state = 0;

  private:
    BatchWorld &world_;
    uint32_t resident_ = 0; // which state the device currently holds
};

} // namespace world

// ===========================================================================
// Seeded synthetic scenes.  Body i depends only on (seed, i),
// so any index shard of a scene equals the same slice of the whole scene.
namespace scene {

struct SplitMix64 {
    uint64_t state;
    SplitMix64(uint64_t seed, uint64_t body_index) : state(seed + (body_index + 1) * 0x9E3779B97F4A7C15ull) {}
    uint64_t next()
    {
        state += 0x9E3779B97F4A7C15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return (double)(next() >> 11) * 0x1.0p-53; }                  // [0,1)
    double range(double lo, double hi) { return lo + (hi - lo) * uniform(); }
};

enum Kind : uint32_t {
    BOXES = 0,      // config 1/2/4: unit cubes, density 1, z in [-0.05, 0.55): many start penetrating
    MIXED = 1,      // config 3: shape = i mod 3 over {cube, 0.5*tetrahedron, 0.5*icosahedron}, density 1
    BOXES_DROP = 2, // as BOXES but z in [0.4, 1.0): nobody penetrates at t=0, all land within 0.5 s and settle
    MIXED_DROP = 3, // as MIXED with the DROP heights
    // EXTENSION scene (body-body contacts, SURVEY 8d config 2/4 extension variant): columns of
    // kStackHeight unit boxes, 1 mm gaps, upright and at rest; columns on a 2 m pitch grid.
    BOX_STACKS = 4
};
constexpr uint32_t kStackHeight = 16;
inline bool is_mixed(Kind k) { return (k & 1u) != 0; }
inline bool is_drop(Kind k) { return (k & 2u) != 0; }

inline std::vector<geometry::Polytope> shapes_of(Kind kind)
{
    using geometry::Polytope;
    if (!is_mixed(kind))
        return {Polytope::new_cube()};
    return {Polytope::new_cube(), 0.5 * Polytope::new_tetrahedron(), 0.5 * Polytope::new_icosahedron()};
}

// Bodies [first, first+count) of a scene laid out on a `grid_w`-wide, 2 m pitch grid.
inline void generate(Kind kind, uint64_t seed, uint32_t grid_w, uint32_t first, uint32_t count,
                     std::vector<rigid::Rigid> &bodies, std::vector<uint32_t> &shape_id)
{
    const auto shapes = shapes_of(kind);
    std::vector<rigid::Rigid> proto;
    std::vector<double> mass;
    for (const auto &p : shapes) {
        const geometry::RigidMetrics m = p.rigid_metrics(1.0);
        proto.push_back(rigid::Rigid::make(m));
        mass.push_back(m.mass);
    }
    bodies.resize(count);
    shape_id.resize(count);
    for (uint32_t k = 0; k < count; ++k) {
        const uint32_t i = first + k;
        const uint32_t sid = is_mixed(kind) ? i % 3u : 0u;
        SplitMix64 rng(seed, i);
        rigid::Rigid r = proto[sid];
        const double z_base = is_drop(kind) ? 0.4 : -0.05;
        r.position = Vec3{2.0 * (double)(i % grid_w), 2.0 * (double)(i / grid_w), z_base + 0.6 * rng.uniform()};
        Quat q;
        q.s = rng.range(-1.0, 1.0);
        q.x = rng.range(-1.0, 1.0);
        q.y = rng.range(-1.0, 1.0);
        q.z = rng.range(-1.0, 1.0);
        r.rotation = xpbd::normalized(q);
        r.velocity = Vec3{rng.range(-1.0, 1.0), rng.range(-1.0, 1.0), rng.range(-1.0, 1.0)};
        r.angular_velocity = Vec3{rng.range(-4.0, 4.0), rng.range(-4.0, 4.0), rng.range(-4.0, 4.0)};
        r.external_force = Vec3{0.0, 0.0, -9.81 * mass[sid]};
        if (kind == BOX_STACKS) {
            const uint32_t column = i / kStackHeight, level = i % kStackHeight;
            r.position = Vec3{2.0 * (double)(column % grid_w), 2.0 * (double)(column / grid_w), 1.001 * (double)level};
            r.rotation = Quat{1.0, 0.0, 0.0, 0.0};
            r.velocity = Vec3{0.0, 0.0, 0.0};
            r.angular_velocity = Vec3{0.0, 0.0, 0.0};
        }
        bodies[k] = r;
        shape_id[k] = sid;
    }
}

// Smallest square grid holding n bodies (32 -> 8 wide is fixed by config 1).
inline uint32_t default_grid_width(uint32_t n)
{
    if (n <= 32)
        return 8;
    uint32_t w = 1;
    while ((uint64_t)w * w < n)
        ++w;
    return w;
}

} // namespace scene

} // namespace constraint_solver
