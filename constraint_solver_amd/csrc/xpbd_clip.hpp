// xpbd_clip.hpp -- the face contact of the body-body contact EXTENSION, shared by the SAT (xpbd_pairs.hip) and the
// GJK + EPA narrowphase (xpbd_gjk.hip): incident face choice, Sutherland-Hodgman clipping with one polygon vertex per
// lane, and the small LDS / reduction helpers both files use.  Device-only; the block is ONE wave everywhere this is used.
//
// Reference lines the conventions come from (the reference's `sat` is an uncalled stub, parity UNPINNED; the checker is
// face_contact of oracle/xpbd_pairs_oracle.c):
//   reference plane = frames.0 * polytopes.0.plane(face)          src/collision.rs:66
//   incident face   = least normal . ref_normal, first minimum     src/collision.rs:76-85
//   partner point   = Plane::project                               src/geometry.rs:45-47
#pragma once
#include <cfloat>

#include "xpbd_device.hpp"
#include "xpbd_pairs.h"

namespace xpbd {

__device__ __forceinline__ Vec3 ld3(const double (*a)[3], uint32_t k) { return Vec3{a[k][0], a[k][1], a[k][2]}; }
__device__ __forceinline__ void st3(double (*a)[3], uint32_t k, Vec3 v)
{
    a[k][0] = v.x;
    a[k][1] = v.y;
    a[k][2] = v.z;
}

// Key whose signed-integer order is IEEE totalOrder (Rust f64::total_cmp).
__device__ __forceinline__ long long total_key(double v)
{
    const long long i = __double_as_longlong(v);
    return i ^ (long long)((unsigned long long)(i >> 63) >> 1);
}

// Value of `v` on the calling lane's PARTNER of reduction level `off` (a power of two).  Levels 1 and 2 are the xor
// partners (DPP quad_perm), 4 and 8 the MIRROR partners inside the aligned 8 / 16 lanes (DPP row_half_mirror / row_mirror:
// lane i <-> 7 - i, 15 - i).  Run over off = W/2 ... 1 the levels still connect all W lanes of an aligned group -- a
// mirror flips the level's bit and all lower ones, which the lower levels flip back -- and a DPP move is one VALU
// instruction where __shfl_xor is an address computation plus a ds_bpermute round trip through the LDS crossbar
// (the narrowphase kernels are chains of such reductions).  Levels 16 and 32 cross the DPP rows: ds_bpermute.
// Only for all-reductions whose combine does not depend on the pairing: min / max with a total-order tie-break.
__device__ __forceinline__ int partner_i32(int v, uint32_t off)
{
    switch (off) {
    case 1:
        return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false); // quad_perm [1, 0, 3, 2]
    case 2:
        return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false); // quad_perm [2, 3, 0, 1]
    case 4:
        return __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false); // row_half_mirror
    case 8:
        return __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false); // row_mirror
    default:
        return __shfl_xor(v, (int)off, 64);
    }
}
__device__ __forceinline__ uint32_t partner(uint32_t v, uint32_t off) { return (uint32_t)partner_i32((int)v, off); }
__device__ __forceinline__ long long partner(long long v, uint32_t off)
{
    const int lo = partner_i32((int)(unsigned long long)v, off), hi = partner_i32((int)((unsigned long long)v >> 32), off);
    return (long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo);
}
__device__ __forceinline__ double partner(double v, uint32_t off) { return __longlong_as_double(partner(__double_as_longlong(v), off)); }

// (value, index) all-reductions over `width` consecutive lanes (a power of two, aligned): every lane ends with the result.
__device__ __forceinline__ void reduce_max_first(double &v, uint32_t &idx, uint32_t width)
{
    for (uint32_t off = width >> 1; off; off >>= 1) {
        const double ov = partner(v, off);
        const uint32_t oi = partner(idx, off);
        if (ov > v || (ov == v && oi < idx)) {
            v = ov;
            idx = oi;
        }
    }
}

__device__ __forceinline__ void reduce_min_first(double &v, uint32_t &idx, uint32_t width)
{
    for (uint32_t off = width >> 1; off; off >>= 1) {
        const double ov = partner(v, off);
        const uint32_t oi = partner(idx, off);
        if (ov < v || (ov == v && oi < idx)) {
            v = ov;
            idx = oi;
        }
    }
}

// The bits of a wave-wide ballot that belong to the calling lane's group of L lanes, moved down to bit 0.
template <uint32_t L>
__device__ __forceinline__ uint64_t group_bits(unsigned long long wave_mask)
{
    const uint32_t first = (threadIdx.x & 63u) / L * L; // the block is one wave
    return L == 64 ? wave_mask : (wave_mask >> first) & ((1ull << (L & 63u)) - 1ull);
}

// The pieces of a result in either layout (Manifold: the public one; ContactManifold: the pipeline's, xpbd_pairs.h).
// Point k of a face contact: on the incident body, and its projection onto the reference plane.
__device__ __forceinline__ void set_point(Manifold &m, uint32_t k, Vec3 inc, Vec3 ref)
{
    m.p_inc[k][0] = inc.x, m.p_inc[k][1] = inc.y, m.p_inc[k][2] = inc.z;
    m.p_ref[k][0] = ref.x, m.p_ref[k][1] = ref.y, m.p_ref[k][2] = ref.z;
}
__device__ __forceinline__ void set_point(ContactManifold &m, uint32_t k, Vec3 inc, Vec3)
{
    m.point[k][0] = inc.x, m.point[k][1] = inc.y, m.point[k][2] = inc.z; // (the reader projects it: see ContactManifold)
}
// The reference plane of a face contact.
__device__ __forceinline__ void set_plane(Manifold &, const Plane &) {}
__device__ __forceinline__ void set_plane(ContactManifold &m, const Plane &pl)
{
    m.plane[0] = pl.normal.x, m.plane[1] = pl.normal.y, m.plane[2] = pl.normal.z, m.plane[3] = pl.displacement;
}
// The single contact of feature 2 (an edge pair; an EPA query without a face normal).
__device__ __forceinline__ void set_single_contact(Manifold &m, Vec3 inc, Vec3 ref) { set_point(m, 0, inc, ref); }
__device__ __forceinline__ void set_single_contact(ContactManifold &m, Vec3 inc, Vec3 ref)
{
    m.point[0][0] = inc.x, m.point[0][1] = inc.y, m.point[0][2] = inc.z;
    m.point[1][0] = ref.x, m.point[1][1] = ref.y, m.point[1][2] = ref.z;
}
// The header of the public layout; the pipeline keeps n_points and feature in its per-pair code byte instead.
__device__ __forceinline__ void set_header(Manifold &m, uint32_t n_points, uint32_t feature, uint32_t index_a, uint32_t index_b, double separation)
{
    m.n_points = n_points, m.feature = feature, m.index_a = index_a, m.index_b = index_b, m.separation = separation;
}
__device__ __forceinline__ void set_header(ContactManifold &, uint32_t, uint32_t, uint32_t, uint32_t, double) {}

// Face contact of one pair by a group of L lanes: reference face `ref_face` of body R (frame fr, shape dr, world-space
// vertices world_r), incident body I (fi, di, world_i).  Writes the contact points of *m and returns their number;
// `iface` = the incident face.  The caller fills in the other fields of the manifold.
//
// Scratch (LDS, all of it owned by the group): `poly0` / `poly1` = the clipper's ping-pong polygons, P rows each
// (P = polygon capacity = one vertex per lane, at most 16); `ref` = kMaxFaceVerts rows for the reference face.  The
// polygons may alias anything the caller no longer needs, `ref` may alias world_i: the block is one wave and its LDS
// accesses execute in program order -- every lane's read of an instruction has been issued before any lane's write of a
// later one -- so the incident face is copied out before the reference face overwrites the incident body's vertices.
template <uint32_t L, uint32_t P, class M>
__device__ __forceinline__ uint32_t face_contact_group(const PolytopeTables &t, const ShapeDesc &dr, const ShapeDesc &di,
                                                       const Frame &fr, const Frame &fi, uint32_t ref_face,
                                                       const double (*world_r)[3], const double (*world_i)[3],
                                                       double (*poly0)[3], double (*poly1)[3], double (*ref)[3], M *m,
                                                       uint32_t lane, uint32_t &iface)
{
    // VPL polygon vertices per lane: vertex k of a polygon belongs to lane k % L, turn k / L.  One per lane where the group
    // is at least as wide as the polygon capacity; the 4-lane groups of box pairs take two turns.
    constexpr uint32_t VPL = (P + L - 1) / L;
    static_assert(P <= 16 && VPL * L >= P && VPL * L >= kMaxFaceVerts, "every polygon and face vertex has a lane and a turn");
    const double *rp = t.planes + 4 * (size_t)(dr.face0 + ref_face);
    const Plane ref_plane = fr * Plane{Vec3{rp[0], rp[1], rp[2]}, rp[3]}; // frames.0 * polytopes.0.plane(face), :66

    // incident face: least normal . ref_normal, first minimum (:76-85); faces strided over the group
    double idot = DBL_MAX;
    iface = 0xFFFFFFFFu;
    for (uint32_t f = lane; f < di.n_faces; f += L) {
        const double *pl = t.planes + 4 * (size_t)(di.face0 + f);
        const Plane w = fi * Plane{Vec3{pl[0], pl[1], pl[2]}, pl[3]};
        const double d = dot(w.normal, ref_plane.normal);
        if (d < idot) { // ascending f on this lane: first minimum
            idot = d;
            iface = f;
        }
    }
    reduce_min_first(idot, iface, L);
    if (iface == 0xFFFFFFFFu)
        iface = 0;

    // Sutherland-Hodgman with the polygon's vertices spread over the lanes (polygons have <= 16 vertices): every vertex
    // tests its edge (p0 -> p1) against the side plane, a prefix sum of the 0/1/2 points it emits gives their slots, so
    // the output order is exactly that of the sequential algorithm.
    const uint32_t *rv = t.face_verts + t.face_start[dr.face0 + ref_face];
    const uint32_t nr = t.face_start[dr.face0 + ref_face + 1] - t.face_start[dr.face0 + ref_face];
    const uint32_t *iv = t.face_verts + t.face_start[di.face0 + iface];
    uint32_t np = t.face_start[di.face0 + iface + 1] - t.face_start[di.face0 + iface];
    Vec3 inc_vertex[VPL], ref_vertex[VPL];
#pragma unroll
    for (uint32_t v = 0; v < VPL; ++v) {
        const uint32_t k = v * L + lane;
        inc_vertex[v] = ref_vertex[v] = Vec3{0.0, 0.0, 0.0};
        if (k < np)
            inc_vertex[v] = ld3(world_i, iv[k]);
        if (k < nr && k < kMaxFaceVerts)
            ref_vertex[v] = ld3(world_r, rv[k]);
    }
    __syncthreads();
    // the reference face's vertices by index once, lane-parallel: the clipping loop below then reads LDS only
    // (phase timing: the three dependent global index loads per side plane were ~40 % of the loop)
#pragma unroll
    for (uint32_t v = 0; v < VPL; ++v) {
        const uint32_t k = v * L + lane;
        if (k < np)
            st3(poly0, k, inc_vertex[v]);
        if (k < nr && k < kMaxFaceVerts)
            st3(ref, k, ref_vertex[v]);
    }
    __syncthreads();
    double(*cur)[3] = poly0, (*nxt)[3] = poly1;
    const uint64_t below = (1ull << lane) - 1ull;
    for (uint32_t e = 0; e < nr && np > 0; ++e) {
        const Vec3 a = ld3(ref, e), bnext = ld3(ref, (e + 1) % nr);
        const Vec3 c = ld3(ref, (e + 2) % nr);
        Vec3 side = cross(bnext - a, ref_plane.normal);
        if (dot(side, c - a) > 0.0)
            side = -side;
        Vec3 p0[VPL], p1[VPL];
        double d0[VPL], d1[VPL];
        bool in0[VPL], crossing[VPL];
        uint64_t in_bits[VPL], cross_bits[VPL];
        // slots of the 0 / 1 / 2 points a vertex emits = points emitted by the vertices before it: two ballots and popcounts
        // per turn over the group's bits of the wave mask (a shuffle scan would be log2(P) round trips)
        uint32_t total = 0;
#pragma unroll
        for (uint32_t v = 0; v < VPL; ++v) {
            const uint32_t k = v * L + lane;
            p0[v] = p1[v] = Vec3{0.0, 0.0, 0.0};
            d0[v] = d1[v] = 0.0;
            in0[v] = crossing[v] = false;
            if (k < np) {
                p0[v] = ld3(cur, k);
                p1[v] = ld3(cur, k + 1 == np ? 0u : k + 1);
                d0[v] = dot(side, p0[v] - a);
                d1[v] = dot(side, p1[v] - a);
                in0[v] = d0[v] <= 0.0;
                crossing[v] = in0[v] != (d1[v] <= 0.0);
            }
            in_bits[v] = group_bits<L>(__ballot(in0[v]));
            cross_bits[v] = group_bits<L>(__ballot(crossing[v]));
        }
#pragma unroll
        for (uint32_t v = 0; v < VPL; ++v) {
            uint32_t slot = total + (uint32_t)(__popcll(in_bits[v] & below) + __popcll(cross_bits[v] & below));
            if (in0[v] && slot < P)
                st3(nxt, slot++, p0[v]);
            if (crossing[v] && slot < P)
                st3(nxt, slot, p0[v] + (p1[v] - p0[v]) * (d0[v] / (d0[v] - d1[v])));
            total += (uint32_t)(__popcll(in_bits[v]) + __popcll(cross_bits[v]));
        }
        np = total < P ? total : P;
        double(*const swap)[3] = cur;
        cur = nxt;
        nxt = swap;
        __syncthreads();
    }
    // every clipped point strictly below the reference plane is a contact, in polygon order
    uint32_t kept = 0;
#pragma unroll
    for (uint32_t v = 0; v < VPL; ++v) {
        const uint32_t k = v * L + lane;
        Vec3 pt{0.0, 0.0, 0.0};
        double depth = 0.0;
        bool keep = false;
        if (k < np) {
            pt = ld3(cur, k);
            depth = distance(ref_plane, pt);
            keep = !(depth >= 0.0);
        }
        const uint64_t keep_bits = group_bits<L>(__ballot(keep));
        const uint32_t at = kept + (uint32_t)__popcll(keep_bits & below);
        if (keep && at < kMaxManifoldPoints) {
            const Vec3 on_ref = pt - depth * ref_plane.normal; // Plane::project, src/geometry.rs:45-47
            set_point(*m, at, pt, on_ref);
        }
        kept += (uint32_t)__popcll(keep_bits);
    }
    if (lane == 0)
        set_plane(*m, ref_plane);
    return kept < kMaxManifoldPoints ? kept : kMaxManifoldPoints;
}

} // namespace xpbd
