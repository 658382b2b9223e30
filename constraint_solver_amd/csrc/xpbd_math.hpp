// xpbd_math.hpp -- f64 3-vector / quaternion / 3x3 algebra for the XPBD stepper.
//
// The reference does all arithmetic through cgmath 0.18.0 (Cargo.lock:372-375).
// Contact index lists only come out bit-exact if every expression here is
// evaluated in cgmath's order, so each operator documents the order it keeps.
// Compile every translation unit that includes this with -ffp-contract=off
// (hipcc defaults to fast contraction; Rust never fuses a*b+c).
#pragma once

#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define XPBD_HD __host__ __device__ __forceinline__
#else
#define XPBD_HD inline
#endif

namespace xpbd {

struct Vec3 {
    double x, y, z;
};

// Scalar part first: Quaternion::new(w, xi, yj, zk).
struct Quat {
    double s, x, y, z;
};

// Three columns, as cgmath's Matrix3 {x, y, z}.
struct Mat3 {
    Vec3 cx, cy, cz;
};

XPBD_HD Vec3 operator+(Vec3 a, Vec3 b) { return Vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
XPBD_HD Vec3 operator-(Vec3 a, Vec3 b) { return Vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
XPBD_HD Vec3 operator-(Vec3 a) { return Vec3{-a.x, -a.y, -a.z}; }
XPBD_HD Vec3 operator*(Vec3 a, double k) { return Vec3{a.x * k, a.y * k, a.z * k}; }
XPBD_HD Vec3 operator*(double k, Vec3 a) { return Vec3{k * a.x, k * a.y, k * a.z}; }
XPBD_HD Vec3 operator/(Vec3 a, double k) { return Vec3{a.x / k, a.y / k, a.z / k}; }

// (x*x' + y*y') + z*z'
XPBD_HD double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

XPBD_HD Vec3 cross(Vec3 a, Vec3 b)
{
    return Vec3{(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)};
}

XPBD_HD double length2(Vec3 a) { return dot(a, a); }
XPBD_HD double length(Vec3 a) { return sqrt(length2(a)); }
// v * (1 / |v|): one divide, three multiplies.
XPBD_HD Vec3 normalized(Vec3 a) { return a * (1.0 / length(a)); }
// onto * (a.onto / |onto|^2)
XPBD_HD Vec3 project_on(Vec3 a, Vec3 onto) { return onto * (dot(a, onto) / length2(onto)); }

XPBD_HD Vec3 vec_of(Quat q) { return Vec3{q.x, q.y, q.z}; }
XPBD_HD Quat quat_sv(double s, Vec3 v) { return Quat{s, v.x, v.y, v.z}; }

// Hamilton product, each component summed left to right.
XPBD_HD Quat operator*(Quat a, Quat b)
{
    return Quat{a.s * b.s - a.x * b.x - a.y * b.y - a.z * b.z,
                a.s * b.x + a.x * b.s + a.y * b.z - a.z * b.y,
                a.s * b.y + a.y * b.s + a.z * b.x - a.x * b.z,
                a.s * b.z + a.z * b.s + a.x * b.y - a.y * b.x};
}

// Rotation of a vector: t = qv x v + v*s;  (qv x t)*2 + v
XPBD_HD Vec3 operator*(Quat q, Vec3 v)
{
    const Vec3 qv = vec_of(q);
    const Vec3 t = cross(qv, v) + v * q.s;
    return cross(qv, t) * 2.0 + v;
}

XPBD_HD Quat conjugate(Quat q) { return Quat{q.s, -q.x, -q.y, -q.z}; }
XPBD_HD Quat operator+(Quat a, Quat b) { return Quat{a.s + b.s, a.x + b.x, a.y + b.y, a.z + b.z}; }
XPBD_HD Quat operator*(double k, Quat q) { return Quat{k * q.s, k * q.x, k * q.y, k * q.z}; }
XPBD_HD Quat operator-(Quat q) { return Quat{-q.s, -q.x, -q.y, -q.z}; }

// |q|^2 = s*s + ((x*x + y*y) + z*z);  q * (1/|q|)
XPBD_HD Quat normalized(Quat q)
{
    const double m2 = q.s * q.s + (q.x * q.x + q.y * q.y + q.z * q.z);
    const double k = 1.0 / sqrt(m2);
    return Quat{q.s * k, q.x * k, q.y * k, q.z * k};
}

// (cx*v.x + cy*v.y) + cz*v.z
XPBD_HD Vec3 operator*(const Mat3 &m, Vec3 v) { return m.cx * v.x + m.cy * v.y + m.cz * v.z; }
XPBD_HD Mat3 operator*(double k, const Mat3 &m) { return Mat3{k * m.cx, k * m.cy, k * m.cz}; }

// cgmath SquareMatrix::invert for Matrix3; false when det == 0 (None).
XPBD_HD bool invert(const Mat3 &m, Mat3 &out)
{
    const double det = m.cx.x * (m.cy.y * m.cz.z - m.cz.y * m.cy.z)
                     - m.cy.x * (m.cx.y * m.cz.z - m.cz.y * m.cx.z)
                     + m.cz.x * (m.cx.y * m.cy.z - m.cy.y * m.cx.z);
    if (det == 0.0)
        return false;
    const Vec3 r0 = cross(m.cy, m.cz) / det;
    const Vec3 r1 = cross(m.cz, m.cx) / det;
    const Vec3 r2 = cross(m.cx, m.cy) / det;
    out.cx = Vec3{r0.x, r1.x, r2.x};
    out.cy = Vec3{r0.y, r1.y, r2.y};
    out.cz = Vec3{r0.z, r1.z, r2.z};
    return true;
}

// Rigid transform (reference src/frame.rs:8-11).
struct Frame {
    Vec3 position;
    Quat rotation;
};

// src/frame.rs:47-53
XPBD_HD Vec3 operator*(const Frame &f, Vec3 v) { return f.rotation * v + f.position; }
// src/frame.rs:30-37
XPBD_HD Frame inverse(const Frame &f)
{
    const Quat qi = conjugate(f.rotation);
    return Frame{qi * (-f.position), qi};
}
// src/frame.rs:67-76
XPBD_HD Frame operator*(const Frame &a, const Frame &b)
{
    return Frame{a.position + a.rotation * b.position, a.rotation * b.rotation};
}

// Plane in Hessian form (reference src/geometry.rs:9-12).
struct Plane {
    Vec3 normal;
    double displacement;
};

// src/geometry.rs:27-36
XPBD_HD Plane plane_from_point_normal(Vec3 point, Vec3 normal)
{
    double displacement = length(project_on(point, normal));
    if (dot(point, normal) < 0.0)
        displacement *= -1.0;
    return Plane{normal, displacement};
}

// src/geometry.rs:39-41
XPBD_HD double distance(const Plane &pl, Vec3 p) { return dot(pl.normal, p) - pl.displacement; }

// Frame * Plane, src/frame.rs:55-64
XPBD_HD Plane operator*(const Frame &f, const Plane &p)
{
    const Vec3 support = f * (p.displacement * p.normal);
    return plane_from_point_normal(support, f.rotation * p.normal);
}

} // namespace xpbd
