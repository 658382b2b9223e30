// xpbd_device.hpp -- device helpers shared by the kernels: SoA field access and Rigid::frame.
#pragma once

#include "xpbd_kernels.h"
#include "xpbd_math.hpp"

namespace xpbd {

__device__ __forceinline__ Vec3 load3(const double *base, uint32_t field, uint32_t stride, uint32_t i)
{
    return Vec3{base[(size_t)(field + 0) * stride + i], base[(size_t)(field + 1) * stride + i],
                base[(size_t)(field + 2) * stride + i]};
}

__device__ __forceinline__ void store3(double *base, uint32_t field, uint32_t stride, uint32_t i, Vec3 v)
{
    base[(size_t)(field + 0) * stride + i] = v.x;
    base[(size_t)(field + 1) * stride + i] = v.y;
    base[(size_t)(field + 2) * stride + i] = v.z;
}

__device__ __forceinline__ Quat load_quat(const double *base, uint32_t field, uint32_t stride, uint32_t i)
{
    return Quat{base[(size_t)(field + 0) * stride + i], base[(size_t)(field + 1) * stride + i],
                base[(size_t)(field + 2) * stride + i], base[(size_t)(field + 3) * stride + i]};
}

// Rigid::frame().position  (src/rigid.rs:77): (position + com) + rotation * (-com)
__device__ __forceinline__ Vec3 frame_origin(Vec3 pos, Quat rot, Vec3 com)
{
    return (pos + com) + rot * (-com);
}

// Rigid::frame() of body i from the SoA arrays (src/rigid.rs:75-80).
__device__ __forceinline__ Frame body_frame(const BodyArrays &b, uint32_t i)
{
    const Vec3 pos = load3(b.dyn, D_POS, b.stride, i);
    const Quat rot = load_quat(b.dyn, D_ROT, b.stride, i);
    const Vec3 com = load3(b.stat, S_COM, b.stride, i);
    return Frame{frame_origin(pos, rot, com), rot};
}

// ---- body-major records of the contact pipeline --------------------------------------------------------------------
// What a body's NEIGHBOURS read of it sits in one record per body instead of one SoA array per scalar: a neighbour is a
// gather, and 34 scalars from 34 arrays are 34 cache lines where two records are three (round 1, PMC on a packed scene:
// 502 MB fetched per pair-solve launch against ~200 MB needed).  Records are 16-byte aligned and read / written with
// 16-byte accesses.
//   BodyRecord  24 doubles = 192 B (two 128-byte lines whatever k: a record starts at offset 0 or 64 of a line), rewritten
//               every substep by the integrate + ground stage:
//                 [0..6] frame after integrate (origin xyz, rotation s x y z) -- all the narrowphase reads: one line
//                 [7..13] frame before integrate   [14..16] position, [17..20] rotation after the ground contacts
//                 [21..23] position before integrate (derive)
//   StatRecord  16 doubles = 128 B (one line), written when the bodies are uploaded:
//                 [0] inverse_mass  [1..9] inverse_inertia (column-major)  [10..12] center_of_mass
struct BodyRecord {
    Frame p1, past;
    Vec3 pos;
    Quat rot;
    Vec3 past_pos;
};

__device__ __forceinline__ Frame load_record_p1(const double *__restrict__ rec, uint32_t i)
{
    const double2 *r = reinterpret_cast<const double2 *>(rec + (size_t)i * kRecDoubles);
    const double2 a = r[0], b = r[1], c = r[2], d = r[3];
    return Frame{Vec3{a.x, a.y, b.x}, Quat{b.y, c.x, c.y, d.x}};
}

__device__ __forceinline__ BodyRecord load_record(const double *__restrict__ rec, uint32_t i)
{
    const double2 *r = reinterpret_cast<const double2 *>(rec + (size_t)i * kRecDoubles);
    double2 v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k)
        v[k] = r[k];
    BodyRecord o;
    o.p1 = Frame{Vec3{v[0].x, v[0].y, v[1].x}, Quat{v[1].y, v[2].x, v[2].y, v[3].x}};
    o.past = Frame{Vec3{v[3].y, v[4].x, v[4].y}, Quat{v[5].x, v[5].y, v[6].x, v[6].y}};
    o.pos = Vec3{v[7].x, v[7].y, v[8].x};
    o.rot = Quat{v[8].y, v[9].x, v[9].y, v[10].x};
    o.past_pos = Vec3{v[10].y, v[11].x, v[11].y};
    return o;
}

__device__ __forceinline__ void store_record(double *__restrict__ rec, uint32_t i, const Frame &p1, const Frame &past, Vec3 pos, Quat rot,
                                             Vec3 past_pos)
{
    double2 *r = reinterpret_cast<double2 *>(rec + (size_t)i * kRecDoubles);
    r[0] = double2{p1.position.x, p1.position.y};
    r[1] = double2{p1.position.z, p1.rotation.s};
    r[2] = double2{p1.rotation.x, p1.rotation.y};
    r[3] = double2{p1.rotation.z, past.position.x};
    r[4] = double2{past.position.y, past.position.z};
    r[5] = double2{past.rotation.s, past.rotation.x};
    r[6] = double2{past.rotation.y, past.rotation.z};
    r[7] = double2{pos.x, pos.y};
    r[8] = double2{pos.z, rot.s};
    r[9] = double2{rot.x, rot.y};
    r[10] = double2{rot.z, past_pos.x};
    r[11] = double2{past_pos.y, past_pos.z};
}

// Only the post-integrate frame of a record (the diagnostic narrowphase entry points fill nothing else).
__device__ __forceinline__ void store_record_p1(double *__restrict__ rec, uint32_t i, const Frame &p1)
{
    double2 *r = reinterpret_cast<double2 *>(rec + (size_t)i * kRecDoubles);
    r[0] = double2{p1.position.x, p1.position.y};
    r[1] = double2{p1.position.z, p1.rotation.s};
    r[2] = double2{p1.rotation.x, p1.rotation.y};
    r[3] = double2{p1.rotation.z, 0.0};
}

struct StatRecord {
    double inv_mass;
    Mat3 inv_inertia;
    Vec3 com;
};

__device__ __forceinline__ StatRecord load_stat_record(const double *__restrict__ stat_rec, uint32_t i)
{
    const double2 *r = reinterpret_cast<const double2 *>(stat_rec + (size_t)i * kStatRecDoubles);
    double2 v[7];
#pragma unroll
    for (int k = 0; k < 7; ++k)
        v[k] = r[k];
    StatRecord o;
    o.inv_mass = v[0].x;
    o.inv_inertia.cx = Vec3{v[0].y, v[1].x, v[1].y};
    o.inv_inertia.cy = Vec3{v[2].x, v[2].y, v[3].x};
    o.inv_inertia.cz = Vec3{v[3].y, v[4].x, v[4].y};
    o.com = Vec3{v[5].x, v[5].y, v[6].x};
    return o;
}

} // namespace xpbd
