// xpbd_gjk.hip -- GJK + EPA narrowphase for gfx950, one wave per body pair (extension, SURVEY 8f
// rank 3; the reference has neither gjk nor epa -- parity UNPINNED, checker oracle/xpbd_gjk_oracle.c).
//
// Kept from the reference: the support convention of Polytope::support / minkowski_support
// (src/geometry.rs:274-289): world-space vertex with the LAST maximal dot under f64::total_cmp, and
// support(frame_a, d) - support(frame_b, -d) -- here with one polytope per frame.
//
// Mapping: both vertex sets live in LDS (world space).  A support query is one wave instruction
// stream: lanes 0-31 evaluate A's vertices, lanes 32-63 B's, a __shfl_xor (key, index) reduction per
// half picks the last maximum.  The GJK simplex (<= 4 points with their witnesses) stays in
// registers, identical on every lane.  The EPA polytope (<= 52 vertices, <= 128 faces) is staged in
// LDS: one face per lane for the closest-face search, the visibility test, the horizon test and the
// construction of the new faces; wave prefix sums give the surviving and the new faces the same
// canonical slots the sequential oracle uses, so the results are bit-identical.
#include <cfloat>
#include <climits>

#include "xpbd_device.hpp"
#include "xpbd_gjk.h"

namespace xpbd {
namespace {

constexpr uint32_t kMaxV = XPBD_MAX_SHAPE_VERTS;
constexpr uint32_t kNone = 0xFFFFFFFFu;

struct GjkLds {
    double wa[kMaxV][3], wb[kMaxV][3];                                        // world-space vertices
    double vw[kMaxEpaVerts][3], va[kMaxEpaVerts][3], vb[kMaxEpaVerts][3];      // polytope vertices + witnesses
    uint32_t fi[kMaxEpaFaces][3];                                              // faces: vertex indices (outward winding)
    double fn[kMaxEpaFaces][3];                                                //        unit normal
    double fd[kMaxEpaFaces];                                                   //        distance of the plane from the origin
};

struct MVert {
    Vec3 w, a, b; // w = a - b
};

__device__ __forceinline__ Vec3 ld3(const double (*a)[3], uint32_t k) { return Vec3{a[k][0], a[k][1], a[k][2]}; }
__device__ __forceinline__ void st3(double (*a)[3], uint32_t k, Vec3 v)
{
    a[k][0] = v.x;
    a[k][1] = v.y;
    a[k][2] = v.z;
}

__device__ __forceinline__ long long total_key(double v)
{
    const long long i = __double_as_longlong(v);
    return i ^ (long long)((unsigned long long)(i >> 63) >> 1);
}

// support(A, d) - support(B, -d); all lanes return the same value.
__device__ __forceinline__ MVert minkowski_support(const GjkLds &s, uint32_t na, uint32_t nb, Vec3 d, uint32_t lane)
{
    const uint32_t half = lane >> 5, k = lane & 31u;
    long long key = LLONG_MIN;
    uint32_t idx = 0;
    if (k < (half ? nb : na)) {
        key = total_key(half ? dot(ld3(s.wb, k), -d) : dot(ld3(s.wa, k), d));
        idx = k;
    }
    for (uint32_t off = 16; off; off >>= 1) { // maximum, HIGHEST index on ties (Iterator::max_by keeps the last)
        const long long ok = __shfl_xor(key, off, 64);
        const uint32_t oi = __shfl_xor(idx, off, 64);
        if (ok > key || (ok == key && oi > idx)) {
            key = ok;
            idx = oi;
        }
    }
    const Vec3 a = ld3(s.wa, __shfl(idx, 0, 64)), b = ld3(s.wb, __shfl(idx, 32, 64));
    return MVert{a - b, a, b};
}

__device__ __forceinline__ Vec3 triple(Vec3 a, Vec3 b, Vec3 c) { return cross(cross(a, b), c); }
__device__ __forceinline__ bool same_dir(Vec3 a, Vec3 b) { return dot(a, b) > 0.0; }

// Triangle case of the boolean GJK: A newest, then B, C.  Rewrites (s0, s1, s2, n) and the direction.
__device__ __forceinline__ void simplex3(const MVert &A, const MVert &B, const MVert &C, MVert &s0, MVert &s1, MVert &s2,
                                         uint32_t &n, Vec3 &d)
{
    const Vec3 a = A.w, ab = B.w - a, ac = C.w - a, ao = -a, abc = cross(ab, ac);
    if (same_dir(cross(abc, ac), ao)) {
        if (same_dir(ac, ao)) {
            s0 = C, s1 = A, n = 2;
            d = triple(ac, ao, ac);
        } else if (same_dir(ab, ao)) {
            s0 = B, s1 = A, n = 2;
            d = triple(ab, ao, ab);
        } else {
            s0 = A, n = 1;
            d = ao;
        }
    } else if (same_dir(cross(ab, abc), ao)) {
        if (same_dir(ab, ao)) {
            s0 = B, s1 = A, n = 2;
            d = triple(ab, ao, ab);
        } else {
            s0 = A, n = 1;
            d = ao;
        }
    } else if (same_dir(abc, ao)) {
        s0 = C, s1 = B, s2 = A, n = 3; // above the triangle
        d = abc;
    } else {
        s0 = B, s1 = C, s2 = A, n = 3; // below: flip the winding
        d = -abc;
    }
}

struct Face {
    uint32_t i0, i1, i2;
    Vec3 n;
    double dist;
    bool ok;
};

// Face (i0, i1, i2) of the polytope with its normal turned away from the origin (which is inside).
__device__ __forceinline__ Face make_face(const GjkLds &s, uint32_t i0, uint32_t i1, uint32_t i2)
{
    Face f;
    const Vec3 p0 = ld3(s.vw, i0);
    Vec3 n = cross(ld3(s.vw, i1) - p0, ld3(s.vw, i2) - p0);
    const double len = length(n);
    f.ok = len > 0.0;
    n = n * (1.0 / len);
    double dist = dot(n, p0);
    if (dist < 0.0) {
        const uint32_t t = i1;
        i1 = i2;
        i2 = t;
        n = -n;
        dist = -dist;
    }
    f.i0 = i0, f.i1 = i1, f.i2 = i2;
    f.n = n;
    f.dist = dist;
    return f;
}

__device__ __forceinline__ void store_face(GjkLds &s, uint32_t slot, const Face &f)
{
    s.fi[slot][0] = f.i0, s.fi[slot][1] = f.i1, s.fi[slot][2] = f.i2;
    st3(s.fn, slot, f.n);
    s.fd[slot] = f.dist;
}

__device__ __forceinline__ uint32_t wave_exclusive_scan(uint32_t v, uint32_t lane, uint32_t *total)
{
    uint32_t inc = v;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(inc, d, 64);
        if (lane >= d)
            inc += up;
    }
    *total = __shfl(inc, 63, 64);
    return inc - v;
}

// First face with the smallest plane distance (all lanes return the same index).
__device__ __forceinline__ uint32_t closest_face(const GjkLds &s, uint32_t nf, uint32_t lane, double *dist_out)
{
    double bd = DBL_MAX;
    uint32_t bi = kNone;
    for (uint32_t f = lane; f < nf; f += 64)
        if (s.fd[f] < bd) {
            bd = s.fd[f];
            bi = f;
        }
    for (uint32_t off = 32; off; off >>= 1) {
        const double od = __shfl_xor(bd, off, 64);
        const uint32_t oi = __shfl_xor(bi, off, 64);
        if (od < bd || (od == bd && oi < bi)) {
            bd = od;
            bi = oi;
        }
    }
    *dist_out = bd;
    return bi;
}

// out: per-pair GjkResult (diagnostic entry point) and/or manifolds: one-point Manifold for the contact pipeline
// (A = reference body, B = incident body; a degenerate query yields no contact).
__global__ void __launch_bounds__(64) k_gjk_epa_pairs(BodyArrays b, PolytopeTables t, const double *__restrict__ frames,
                                                      const uint32_t *__restrict__ pairs, uint32_t n_pairs,
                                                      GjkResult *__restrict__ out, Manifold *__restrict__ manifolds)
{
    __shared__ GjkLds s;
    const uint32_t p = blockIdx.x, lane = threadIdx.x;
    if (p >= n_pairs)
        return;
    const uint32_t ia = pairs[2 * p], ib = pairs[2 * p + 1];
    const Frame fa = load_frame(frames, b.stride, ia), fb = load_frame(frames, b.stride, ib);
    const uint32_t sa = b.shape_id[ia], sb = b.shape_id[ib];
    const ShapeDesc da = t.desc[sa], db = t.desc[sb];
    GjkResult *r = out ? out + p : nullptr;
    Manifold *mf = manifolds ? manifolds + p : nullptr;

    int32_t status = 0; // separated
    uint32_t gjk_iters = 0, epa_iters = 0;
    auto finish = [&](int32_t st) {
        if (lane == 0) {
            if (r) {
                r->status = st;
                r->gjk_iterations = gjk_iters;
                r->epa_iterations = epa_iters;
            }
            if (mf && st != 1)
                mf->n_points = 0;
        }
    };
    if (da.n_verts == 0 || db.n_verts == 0) {
        finish(0);
        return;
    }
    {
        const uint32_t half = lane >> 5, k = lane & 31u;
        const ShapeDesc dm = half ? db : da;
        if (k < dm.n_verts) {
            const double *v = t.verts + 3 * (size_t)(dm.vert0 + k);
            st3(half ? s.wb : s.wa, k, (half ? fb : fa) * Vec3{v[0], v[1], v[2]});
        }
    }
    __syncthreads();
    const uint32_t na = da.n_verts, nb = db.n_verts;

    // ---- boolean GJK: the simplex lives in registers, identical on every lane ----
    MVert s0, s1, s2, s3;
    uint32_t n = 1;
    Vec3 d;
    {
        const double *ca = t.centroids + 3 * (size_t)sa, *cb = t.centroids + 3 * (size_t)sb;
        d = fb * Vec3{cb[0], cb[1], cb[2]} - fa * Vec3{ca[0], ca[1], ca[2]};
        if (!(dot(d, d) > 0.0))
            d = Vec3{1.0, 0.0, 0.0};
    }
    s0 = minkowski_support(s, na, nb, d, lane);
    s1 = s2 = s3 = s0;
    d = -s0.w;
    bool hit = false;
    for (uint32_t it = 0; it < kMaxGjkIters; ++it) {
        gjk_iters = it + 1;
        if (!(dot(d, d) > 0.0)) { // origin on the simplex: touching / degenerate
            finish(2);
            return;
        }
        const MVert pnt = minkowski_support(s, na, nb, d, lane);
        if (!(dot(pnt.w, d) > 0.0)) { // separated (or just touching)
            finish(0);
            return;
        }
        if (n == 1) {
            s1 = pnt;
            n = 2;
            const Vec3 a = s1.w, ab = s0.w - a, ao = -a;
            if (same_dir(ab, ao)) {
                d = triple(ab, ao, ab);
            } else {
                s0 = s1;
                n = 1;
                d = ao;
            }
        } else if (n == 2) {
            const MVert A = pnt, B = s1, C = s0;
            simplex3(A, B, C, s0, s1, s2, n, d);
        } else {
            const MVert A = pnt, B = s2, C = s1, D = s0;
            const Vec3 a = A.w, ao = -a, ab = B.w - a, ac = C.w - a, ad = D.w - a;
            const Vec3 abc = cross(ab, ac), acd = cross(ac, ad), adb = cross(ad, ab);
            if (same_dir(abc, ao)) {
                simplex3(A, B, C, s0, s1, s2, n, d);
            } else if (same_dir(acd, ao)) {
                simplex3(A, C, D, s0, s1, s2, n, d);
            } else if (same_dir(adb, ao)) {
                simplex3(A, D, B, s0, s1, s2, n, d);
            } else {
                s3 = A; // D, C, B, A enclose the origin
                hit = true;
                break;
            }
        }
    }
    if (!hit) {
        finish(2);
        return;
    }

    // ---- EPA: the polytope is staged in LDS ----
    if (lane < 4) {
        const MVert &m = lane == 0 ? s0 : (lane == 1 ? s1 : (lane == 2 ? s2 : s3));
        st3(s.vw, lane, m.w);
        st3(s.va, lane, m.a);
        st3(s.vb, lane, m.b);
    }
    __syncthreads();
    uint32_t nv = 4, nf = 4;
    {
        bool bad = false;
        if (lane < 4) {
            const uint32_t t0 = lane == 3 ? 1u : 0u;
            const uint32_t t1 = lane == 0 ? 1u : (lane == 1 ? 3u : (lane == 2 ? 2u : 3u));
            const uint32_t t2 = lane == 0 ? 2u : (lane == 1 ? 1u : (lane == 2 ? 3u : 2u));
            const Face f = make_face(s, t0, t1, t2); // {0,1,2} {0,3,1} {0,2,3} {1,3,2}
            store_face(s, lane, f);
            bad = !f.ok;
        }
        if (__ballot(bad)) {
            finish(2);
            return;
        }
    }
    __syncthreads();

    for (uint32_t it = 0; it < kMaxEpaIters; ++it) {
        epa_iters = it + 1;
        double best_dist;
        const uint32_t best = closest_face(s, nf, lane, &best_dist);
        const Vec3 bn = ld3(s.fn, best);
        const MVert pnt = minkowski_support(s, na, nb, bn, lane);
        if (dot(pnt.w, bn) - best_dist < kEpaTolerance || nv == kMaxEpaVerts)
            break;

        // visibility of "my" two faces (lane, lane + 64); the masks are wave-uniform
        const uint32_t f0 = lane, f1 = lane + 64;
        const bool vis0 = f0 < nf && dot(ld3(s.fn, f0), pnt.w - ld3(s.vw, s.fi[f0][0])) > 0.0;
        const bool vis1 = f1 < nf && dot(ld3(s.fn, f1), pnt.w - ld3(s.vw, s.fi[f1][0])) > 0.0;
        const unsigned long long mask0 = __ballot(vis0), mask1 = __ballot(vis1);

        // horizon test of my faces' edges: a->b is on the horizon iff no other VISIBLE face holds b->a
        uint32_t hz[2] = {0, 0};
        for (uint32_t w = 0; w < 2; ++w) {
            const uint32_t k = w ? f1 : f0;
            if (!(w ? vis1 : vis0))
                continue;
            for (uint32_t e = 0; e < 3; ++e) {
                const uint32_t ea = s.fi[k][e], eb = s.fi[k][e == 2 ? 0 : e + 1];
                bool interior = false;
                for (uint32_t q = 0; q < nf && !interior; ++q) {
                    if (q == k || !(((q < 64 ? mask0 : mask1) >> (q & 63u)) & 1ull))
                        continue;
                    const uint32_t q0 = s.fi[q][0], q1 = s.fi[q][1], q2 = s.fi[q][2];
                    interior = (q0 == eb && q1 == ea) || (q1 == eb && q2 == ea) || (q2 == eb && q0 == ea);
                }
                if (!interior)
                    hz[w] |= 1u << e;
            }
        }
        // canonical slots: surviving faces keep their order; horizon edges ordered by (face, edge)
        uint32_t keep_a, keep_b, ne_a, ne_b;
        const uint32_t kpos0 = wave_exclusive_scan((f0 < nf && !vis0) ? 1u : 0u, lane, &keep_a);
        const uint32_t kpos1 = keep_a + wave_exclusive_scan((f1 < nf && !vis1) ? 1u : 0u, lane, &keep_b);
        const uint32_t epos0 = wave_exclusive_scan(__popc(hz[0]), lane, &ne_a);
        const uint32_t epos1 = ne_a + wave_exclusive_scan(__popc(hz[1]), lane, &ne_b);
        const uint32_t keep = keep_a + keep_b, ne = ne_a + ne_b;
        if (ne == 0 || keep + ne > kMaxEpaFaces)
            break; // numerical dead end or out of room: report the best face found so far

        // pull my faces into registers, then rewrite the face table
        Face mine[2];
        uint32_t edge_a[2][3], edge_b[2][3];
        for (uint32_t w = 0; w < 2; ++w) {
            const uint32_t k = w ? f1 : f0;
            if (k < nf) {
                mine[w].i0 = s.fi[k][0], mine[w].i1 = s.fi[k][1], mine[w].i2 = s.fi[k][2];
                mine[w].n = ld3(s.fn, k);
                mine[w].dist = s.fd[k];
                edge_a[w][0] = mine[w].i0, edge_b[w][0] = mine[w].i1;
                edge_a[w][1] = mine[w].i1, edge_b[w][1] = mine[w].i2;
                edge_a[w][2] = mine[w].i2, edge_b[w][2] = mine[w].i0;
            }
        }
        if (lane == 0) {
            st3(s.vw, nv, pnt.w);
            st3(s.va, nv, pnt.a);
            st3(s.vb, nv, pnt.b);
        }
        __syncthreads();
        if (f0 < nf && !vis0)
            store_face(s, kpos0, mine[0]);
        if (f1 < nf && !vis1)
            store_face(s, kpos1, mine[1]);
        bool bad = false;
        for (uint32_t w = 0; w < 2; ++w) {
            uint32_t slot = keep + (w ? epos1 : epos0);
            for (uint32_t e = 0; e < 3; ++e)
                if (hz[w] & (1u << e)) {
                    const Face f = make_face(s, edge_a[w][e], edge_b[w][e], nv);
                    store_face(s, slot++, f);
                    bad |= !f.ok;
                }
        }
        nf = keep + ne;
        ++nv;
        __syncthreads();
        if (__ballot(bad)) {
            finish(2);
            return;
        }
    }

    double best_dist;
    const uint32_t best = closest_face(s, nf, lane, &best_dist);
    if (lane == 0) {
        const uint32_t i0 = s.fi[best][0], i1 = s.fi[best][1], i2 = s.fi[best][2];
        const Vec3 nrm = ld3(s.fn, best);
        const Vec3 aw = ld3(s.vw, i0), proj = nrm * best_dist;
        const Vec3 v0 = ld3(s.vw, i1) - aw, v1 = ld3(s.vw, i2) - aw, v2 = proj - aw;
        const double d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
        const double denom = d00 * d11 - d01 * d01;
        const double bv = (d11 * d20 - d01 * d21) / denom, bw = (d00 * d21 - d01 * d20) / denom, bu = 1.0 - bv - bw;
        const Vec3 pa = ld3(s.va, i0) * bu + ld3(s.va, i1) * bv + ld3(s.va, i2) * bw;
        const Vec3 pb = ld3(s.vb, i0) * bu + ld3(s.vb, i1) * bv + ld3(s.vb, i2) * bw;
        if (r) {
            r->depth = best_dist;
            r->normal[0] = nrm.x, r->normal[1] = nrm.y, r->normal[2] = nrm.z;
            r->point_a[0] = pa.x, r->point_a[1] = pa.y, r->point_a[2] = pa.z;
            r->point_b[0] = pb.x, r->point_b[1] = pb.y, r->point_b[2] = pb.z;
        }
        if (mf) {
            mf->n_points = 1;
            mf->feature = 2; // reference body A, incident body B
            mf->index_a = mf->index_b = 0;
            mf->separation = -best_dist;
            mf->p_ref[0][0] = pa.x, mf->p_ref[0][1] = pa.y, mf->p_ref[0][2] = pa.z;
            mf->p_inc[0][0] = pb.x, mf->p_inc[0][1] = pb.y, mf->p_inc[0][2] = pb.z;
        }
    }
    (void)status;
    finish(1);
}

} // namespace

hipError_t launch_gjk_epa_pairs(const BodyArrays &b, const PolytopeTables &t, const double *frames, const uint32_t *pairs,
                                uint32_t n_pairs, GjkResult *out, Manifold *manifolds, hipStream_t stream)
{
    if (n_pairs)
        hipLaunchKernelGGL(k_gjk_epa_pairs, dim3(n_pairs), dim3(64), 0, stream, b, t, frames, pairs, n_pairs, out,
                           manifolds);
    return hipGetLastError();
}

} // namespace xpbd
