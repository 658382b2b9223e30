// xpbd_gjk.hip -- GJK + EPA narrowphase for gfx950 (extension; the reference has
// neither gjk nor epa -- parity UNPINNED, checker oracle/xpbd_gjk_oracle.c).
//
// Kept from the reference: the support convention of Polytope::support / minkowski_support
// (src/geometry.rs:274-289): world-space vertex with the LAST maximal dot under f64::total_cmp, and
// support(frame_a, d) - support(frame_b, -d) -- here with one polytope per frame.
//
// Two kernels, because most broadphase pairs are separated and never need a polytope:
//  * k_gjk_pairs<L>: the boolean GJK, L = 8 or 32 lanes per pair (8 or 2 pairs per wave).  Both vertex
//    sets live in LDS (world space).  A support query: the first half of the group strides over A's
//    vertices, the second half over B's, a __shfl_xor (key, index) reduction per half picks the last
//    maximum.  The simplex (<= 4 points with their witnesses) stays in registers, identical on every
//    lane of the group.  A pair whose simplex encloses the origin is appended to a hit list (one
//    wave-aggregated atomic per wave) together with the vertex indices of its four simplex points.
//  * k_epa_pairs: one wave per HIT, grid-stride over the hit list.  The polytope (<= 52 vertices,
//    <= 128 faces) is staged in LDS: one face per lane for the closest-face search, the visibility
//    test, the horizon test and the construction of the new faces; wave prefix sums give the
//    surviving and the new faces the same canonical slots the sequential oracle uses.
// Every pair's arithmetic is that of the sequential oracle, so the results are bit-identical whatever
// the order of the hit list.
#include <cfloat>
#include <climits>
#include <type_traits>

#include "xpbd_clip.hpp"
#include "xpbd_device.hpp"
#include "xpbd_gjk.h"

#ifndef XPBD_EPA_TIMING_STOP
#define XPBD_EPA_TIMING_STOP 0
#endif
namespace xpbd {
namespace {

#ifndef XPBD_GJK_SMALL_LANES
#define XPBD_GJK_SMALL_LANES 8 // A/B on 262 144 mixed polyhedra: 16 lanes 1.06e9, 8 lanes 1.12e9 body-substeps/s
#endif
constexpr uint32_t kMaxV = XPBD_MAX_SHAPE_VERTS;
constexpr uint32_t kNone = 0xFFFFFFFFu;

template <uint32_t V>
struct GjkVertsT {
    double wa[V][3], wb[V][3];                                                // world-space vertices (V = capacity per body)
};
using GjkVerts = GjkVertsT<kMaxV>;

// ... and, in k_gjk_pairs, the (at most five) vertices of a warm-start polytope, where make_face reads them
template <uint32_t V>
struct GjkPairLds : GjkVertsT<V> {
    double vw[5][3];
};

struct GjkLds : GjkVerts {
    double vw[kMaxEpaVerts][3];                                                // polytope vertices w = wa[via] - wb[vib]
    uint8_t via[kMaxEpaVerts], vib[kMaxEpaVerts];                              // their witnesses, as vertex indices
    uint32_t fi[kMaxEpaFaces];                                                 // faces: vertex indices (outward winding), a byte each
    double fn[kMaxEpaFaces][3];                                                //        unit normal
    double fd[kMaxEpaFaces];                                                   //        distance of the plane from the origin
    // ve[a][b] != 0: some face that SEES the new point holds the directed edge a -> b.  Set and cleared again inside every
    // EPA iteration (all zero in between, zeroed once per workgroup): an edge a -> b of a visible face is on the horizon iff
    // ve[b][a] == 0 -- one LDS read instead of a search through all faces (which made an iteration O(faces^2) per lane).
    uint8_t ve[kMaxEpaVerts][kMaxEpaVerts + 4];
    // the horizon of the current iteration: directed edges a -> b in canonical order ((visible face, edge) ascending); lane q
    // then builds the new face over edge q -- one make_face (a cross product, a square root, a division) per lane instead
    // of up to three per visible face one after the other
    uint8_t he[kMaxEpaFaces][2];
};

#ifdef XPBD_GJK_TIMING
// Diagnostics build only: cycles spent by the waves of k_gjk_pairs in its phases, summed over waves (lane 0 of each wave).
__device__ unsigned long long g_gjk_timing[8];
#define GJK_TICK(slot)                                                                       \
    do {                                                                                     \
        const unsigned long long now_ = clock64();                                           \
        if (threadIdx.x == 0)                                                                \
            atomicAdd(&g_gjk_timing[slot], now_ - tick_);                                    \
        tick_ = now_;                                                                        \
    } while (0)
#else
#define GJK_TICK(slot) do { } while (0)
#endif

struct MVert {
    Vec3 w, a, b;     // w = a - b
    uint32_t ia, ib;  // a = A's world vertex ia, b = B's world vertex ib
};

// support(A, d) - support(B, -d) by a group of L lanes (`lane` = lane inside the group); all lanes of the
// group return the same value.
template <uint32_t L, class Verts>
__device__ __forceinline__ MVert minkowski_support(const Verts &s, uint32_t na, uint32_t nb, Vec3 d, uint32_t lane)
{
    constexpr uint32_t H = L / 2;
    const uint32_t half = lane / H, k = lane % H;
    const uint32_t n = half ? nb : na;
    const double(*w)[3] = half ? s.wb : s.wa;
    const Vec3 dir = half ? -d : d;
    long long key = LLONG_MIN;
    uint32_t idx = 0;
    for (uint32_t v = k; v < n; v += H) { // ascending v: >= keeps the LAST maximum (Iterator::max_by)
        const long long kv = total_key(dot(ld3(w, v), dir));
        if (kv >= key) {
            key = kv;
            idx = v;
        }
    }
    for (uint32_t off = H / 2; off; off >>= 1) { // maximum, HIGHEST index on ties
        const long long ok = partner(key, off);
        const uint32_t oi = partner(idx, off);
        if (ok > key || (ok == key && oi > idx)) {
            key = ok;
            idx = oi;
        }
    }
    // every lane of a half holds its half's result; the level-H partner is in the other half
    const uint32_t other = partner(idx, H);
    const uint32_t ia = half ? other : idx, ib = half ? idx : other;
    const Vec3 a = ld3(s.wa, ia), b = ld3(s.wb, ib);
    return MVert{a - b, a, b, ia, ib};
}

__device__ __forceinline__ Vec3 triple(Vec3 a, Vec3 b, Vec3 c) { return cross(cross(a, b), c); }
__device__ __forceinline__ bool same_dir(Vec3 a, Vec3 b) { return dot(a, b) > 0.0; }

// Search direction from the segment (direction ab) towards the origin (ao = origin - a).  With the origin ON the segment's
// line the triple product vanishes -- two boxes stacked exactly on top of each other start like this -- and any
// perpendicular of ab will do: ab x the coordinate axis ab has the least extent along (first minimum).
__device__ __forceinline__ Vec3 edge_direction(Vec3 ab, Vec3 ao)
{
    const Vec3 d = triple(ab, ao, ab);
    if (dot(d, d) > 0.0)
        return d;
    const double ax = fabs(ab.x), ay = fabs(ab.y), az = fabs(ab.z);
    const Vec3 e = (ax <= ay && ax <= az) ? Vec3{1.0, 0.0, 0.0} : (ay <= az ? Vec3{0.0, 1.0, 0.0} : Vec3{0.0, 0.0, 1.0});
    return cross(ab, e);
}

// Triangle case of the boolean GJK: A newest, then B, C.  Rewrites (s0, s1, s2, n) and the direction.
__device__ __forceinline__ void simplex3(const MVert &A, const MVert &B, const MVert &C, MVert &s0, MVert &s1, MVert &s2,
                                         uint32_t &n, Vec3 &d)
{
    const Vec3 a = A.w, ab = B.w - a, ac = C.w - a, ao = -a, abc = cross(ab, ac);
    if (same_dir(cross(abc, ac), ao)) {
        if (same_dir(ac, ao)) {
            s0 = C, s1 = A, n = 2;
            d = edge_direction(ac, ao);
        } else if (same_dir(ab, ao)) {
            s0 = B, s1 = A, n = 2;
            d = edge_direction(ab, ao);
        } else {
            s0 = A, n = 1;
            d = ao;
        }
    } else if (same_dir(cross(ab, abc), ao)) {
        if (same_dir(ab, ao)) {
            s0 = B, s1 = A, n = 2;
            d = edge_direction(ab, ao);
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
// vertex e (0..2; 3 = vertex 0 again) of a face's packed index word
__device__ __forceinline__ uint32_t face_vertex(uint32_t packed, uint32_t e) { return (packed >> (8 * (e == 3 ? 0 : e))) & 0xFFu; }

// Face (i0, i1, i2) of the polytope.  `opposite` = a vertex known to lie behind the face (the fourth vertex of the first
// tetrahedron): the winding is flipped so that the normal points away from it.  kNone = trust the winding: a new face
// (a, b, p) over a horizon edge a -> b inherits the outward winding of the visible face the edge came from.  (The sign of
// the origin's distance is rounding noise exactly where it matters: origin ON a face of the first tetrahedron.)
template <class S>
__device__ __forceinline__ Face make_face(const S &s, uint32_t i0, uint32_t i1, uint32_t i2, uint32_t opposite)
{
    Face f;
    const Vec3 p0 = ld3(s.vw, i0);
    Vec3 n = cross(ld3(s.vw, i1) - p0, ld3(s.vw, i2) - p0);
    const double len = length(n);
    f.ok = len > 0.0;
    n = n * (1.0 / len);
    if (opposite != kNone) {
        const double side = dot(n, ld3(s.vw, opposite) - p0);
        f.ok = f.ok && side != 0.0; // flat tetrahedron
        if (side > 0.0) {
            const uint32_t t = i1;
            i1 = i2;
            i2 = t;
            n = -n;
        }
    }
    f.i0 = i0, f.i1 = i1, f.i2 = i2;
    f.n = n;
    f.dist = dot(n, p0);
    return f;
}

template <class S>
__device__ __forceinline__ void store_face(S &s, uint32_t slot, const Face &f)
{
    s.fi[slot] = f.i0 | (f.i1 << 8) | (f.i2 << 16);
    st3(s.fn, slot, f.n);
    s.fd[slot] = f.dist;
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
        const double od = partner(bd, off);
        const uint32_t oi = partner(bi, off);
        if (od < bd || (od == bd && oi < bi)) {
            bd = od;
            bi = oi;
        }
    }
    *dist_out = bd;
    return bi;
}

// World-space vertices of both bodies of a pair into LDS, `group` lanes cooperating (first half A, second half B).
template <class Verts>
__device__ __forceinline__ void stage_world_vertices(Verts &s, const PolytopeTables &t, const ShapeDesc &da, const ShapeDesc &db,
                                                     const Frame &fa, const Frame &fb, uint32_t lane, uint32_t group)
{
    const uint32_t H = group / 2, half = lane / H, k = lane % H;
    const ShapeDesc dm = half ? db : da;
    for (uint32_t vtx = k; vtx < dm.n_verts; vtx += H) {
        const double *v = t.verts + 3 * (size_t)(dm.vert0 + vtx);
        st3(half ? s.wb : s.wa, vtx, (half ? fb : fa) * Vec3{v[0], v[1], v[2]});
    }
}

// The shape tables the GJK / EPA kernels read per pair -- descriptor, centroid, bounding radius, vertices -- staged into LDS
// once per workgroup when the world's tables are small (the usual case: a handful of shapes).  Phase timing of k_gjk_pairs
// on the settled mixed pile (scripts/gjk_phase_timing.py): the GJK iterations were a fifth of a wave's time, the rest the
// chain of dependent global loads in front of them (pair -> record and shape id -> descriptor, centroid, radius ->
// vertices) and the stores behind; with the tables in LDS the chain ends at the shape id.
constexpr uint32_t kStageShapes = 8, kStageVerts = 64;
struct StagedTables {
    double verts[kStageVerts * 3];
    double centroids[kStageShapes * 3];
    double radii[kStageShapes];
    ShapeDesc desc[kStageShapes];
};

// all lanes of the workgroup; returns the tables to use (pointing into `sh` when staged)
template <bool STAGED>
__device__ __forceinline__ PolytopeTables stage_tables(const PolytopeTables &t, StagedTables &sh)
{
    if (!STAGED)
        return t;
    for (uint32_t k = threadIdx.x; k < 3 * t.total_verts; k += blockDim.x)
        sh.verts[k] = t.verts[k];
    for (uint32_t k = threadIdx.x; k < t.n_shapes; k += blockDim.x) {
        sh.desc[k] = t.desc[k];
        sh.radii[k] = t.radii[k];
        for (uint32_t a = 0; a < 3; ++a)
            sh.centroids[3 * k + a] = t.centroids[3 * k + a];
    }
    __syncthreads();
    PolytopeTables l = t;
    l.verts = sh.verts;
    l.centroids = sh.centroids;
    l.radii = sh.radii;
    l.desc = sh.desc;
    return l;
}

// What k_gjk_pairs hands to the EPA kernels, two words per pair.  Word 0: vertex indices of the first four polytope points
// (byte k = A's index of point k, byte 4 + k = B's).  Word 1: byte 0 / 1 = A's / B's index of a fifth point, byte 2 = the
// MODE: 0 = the four points are GJK's final tetrahedron; m = 3 or 4 = a WARM-START pyramid (below): points 0 .. m - 1 are its
// top, point m its apex.
__device__ __forceinline__ unsigned long long pack_seed(const MVert &s0, const MVert &s1, const MVert &s2, const MVert &s3)
{
    const unsigned long long a = s0.ia | (s1.ia << 8) | (s2.ia << 16) | (s3.ia << 24);
    const unsigned long long b = s0.ib | (s1.ib << 8) | (s2.ib << 16) | (s3.ib << 24);
    return a | (b << 32);
}

// WARM START of the expansion (oracle: seed_polytope of xpbd_gjk_oracle.c, which is the normative text).  `warm` is the pair's
// cached direction: its last penetration normal, or the direction that separated it before it came into contact.  The polytope
// is seeded with the face of the Minkowski difference that has (nearly) that normal: four support points in directions tilted
// by kWarmTilt towards the diagonals of a tangent frame, duplicates dropped, plus the support point of -n as the apex.  Face f
// of the pyramid over m = 3 or 4 top points: the top fan (0, f + 1, f + 2) for f < m - 2 (opposite vertex: the apex m), then the
// sides ((k + 1) % m, k, m), k = f - (m - 2) (opposite vertex: (k + 2) % m).
constexpr double kWarmTilt = 1e-3, kWarmConvex = 1e-10;
#ifndef XPBD_GJK_WARM_MODE
#define XPBD_GJK_WARM_MODE 1 // 0: an A/B build that never warm-starts (it does NOT match the oracle)
#endif
// (i0 | i1 << 8 | i2 << 16 | opposite << 24)
__device__ __forceinline__ uint32_t warm_face_indices(uint32_t f, uint32_t m)
{
    if (f + 2 < m)
        return 0u | ((f + 1) << 8) | ((f + 2) << 16) | (m << 24);
    const uint32_t k = f - (m - 2);
    const uint32_t k1 = k + 1 == m ? 0u : k + 1, k2 = k1 + 1 == m ? 0u : k1 + 1; // (k + 1) % m, (k + 2) % m
    return k1 | (k << 8) | (m << 16) | (k2 << 24);
}

// By a group of L lanes (all lanes return the same verdict and, when it is true, the same points v[0 .. m] and m): is the
// pyramid sound, convex and around the origin?  Then the pair is penetrating and GJK is not needed.
template <uint32_t L, class Lds>
__device__ __forceinline__ bool warm_seed(Lds &s, uint32_t na, uint32_t nb, Vec3 warm, uint32_t lane, MVert &v0, MVert &v1, MVert &v2,
                                          MVert &v3, MVert &v4, uint32_t &m_out)
{
    const double len = length(warm);
    if (!(len > 0.0) || !(len <= DBL_MAX)) // group-uniform
        return false;
    const Vec3 n = warm * (1.0 / len);
    const double ax = fabs(n.x), ay = fabs(n.y), az = fabs(n.z);
    const bool use_x = ax <= ay && ax <= az, use_y = !use_x && ay <= az;
    const Vec3 e{use_x ? 1.0 : 0.0, use_y ? 1.0 : 0.0, (use_x || use_y) ? 0.0 : 1.0};
    const Vec3 t1 = normalized(cross(n, e)), t2 = cross(n, t1);
    // the four tilted support points, in the order (+t1 +t2), (-t1 +t2), (-t1 -t2), (+t1 -t2)
    const MVert q0 = minkowski_support<L>(s, na, nb, n + ((1.0 * kWarmTilt) * t1 + (1.0 * kWarmTilt) * t2), lane);
    const MVert q1 = minkowski_support<L>(s, na, nb, n + ((-1.0 * kWarmTilt) * t1 + (1.0 * kWarmTilt) * t2), lane);
    const MVert q2 = minkowski_support<L>(s, na, nb, n + ((-1.0 * kWarmTilt) * t1 + (-1.0 * kWarmTilt) * t2), lane);
    const MVert q3 = minkowski_support<L>(s, na, nb, n + ((1.0 * kWarmTilt) * t1 + (-1.0 * kWarmTilt) * t2), lane);
    auto same = [](const MVert &x, const MVert &y) { return x.ia == y.ia && x.ib == y.ib; };
    // duplicates dropped (a point equal to an earlier one; everything below is group-uniform: every lane holds the same points)
    const bool k1 = !same(q1, q0), k2 = !same(q2, q0) && !same(q2, q1), k3 = !same(q3, q0) && !same(q3, q1) && !same(q3, q2);
    const uint32_t m = 1u + (k1 ? 1u : 0u) + (k2 ? 1u : 0u) + (k3 ? 1u : 0u);
    if (m < 3)
        return false;
    v0 = q0;
    v1 = k1 ? q1 : q2;                  // (m >= 3: at most one of the three was dropped)
    v2 = (k1 && k2) ? q2 : q3;
    const MVert apex = minkowski_support<L>(s, na, nb, -n, lane);
    if (same(apex, v0) || same(apex, v1) || same(apex, v2) || (m == 4 && same(apex, q3)))
        return false;
    v3 = m == 4 ? q3 : apex;
    v4 = apex;
    // the faces, one per lane, from the points in LDS
    __syncthreads();
    if (lane == 0) {
        st3(s.vw, 0, v0.w);
        st3(s.vw, 1, v1.w);
        st3(s.vw, 2, v2.w);
        st3(s.vw, 3, v3.w);
        st3(s.vw, 4, v4.w);
    }
    __syncthreads();
    const uint32_t nf = (m - 2) + m;
    bool ok = true;
    if (lane < nf) {
        const uint32_t idx = warm_face_indices(lane, m);
        const Face f = make_face(s, idx & 0xFFu, (idx >> 8) & 0xFFu, (idx >> 16) & 0xFFu, idx >> 24);
        ok = f.ok && f.dist >= 0.0; // sound, and the origin inside or on it
        const Vec3 p0 = ld3(s.vw, f.i0);
        for (uint32_t q = 0; q <= m; ++q)
            ok = ok && dot(f.n, ld3(s.vw, q) - p0) <= kWarmConvex; // convex with exactly this face structure
    }
    static_assert(L >= 8, "six faces, one per lane");
    const unsigned long long wave_bad = __ballot(!ok);
    const uint32_t first = (threadIdx.x & 63u) / L * L;
    const unsigned long long group_bad = L == 64 ? wave_bad : (wave_bad >> first) & ((1ull << (L & 63u)) - 1ull);
    m_out = m;
    return group_bad == 0;
}

// Boolean GJK, L lanes per pair.  out / manifolds (each optional) receive the verdict of every pair that is NOT
// penetrating; a penetrating pair goes to the hit list with its simplex and is finished by k_epa_pairs.
// PRETEST (contact pipeline only): as in k_sat_pairs, disjoint tight bounding spheres mean "separated" at once.
// With `survivors` (the two-pass form, after k_pair_pretest of xpbd_pairs.hip) the groups take their pairs from that
// list, n_pairs is read from *survivor_count, and block 0 zeroes the counter of the next launch.
// V: vertex capacity per body of a pair's LDS record (the launcher picks 16 when no shape has more: 768 bytes per pair
// instead of 1 536, i.e. twice as many waves per CU for a kernel that waits on LDS round trips and shuffles).
template <uint32_t L, uint32_t V, bool PRETEST, bool STAGED>
__global__ void __launch_bounds__(64) k_gjk_pairs(BodyArrays b, PolytopeTables t_global, const double *__restrict__ frames,
                                                  const uint32_t *__restrict__ pairs, uint32_t n_pairs,
                                                  const uint32_t *__restrict__ survivors, const uint32_t *__restrict__ survivor_count,
                                                  uint32_t *__restrict__ next_survivor_count,
                                                  GjkResult *__restrict__ out, ContactManifold *__restrict__ manifolds,
                                                  uint32_t *__restrict__ hit_counts, uint32_t *__restrict__ hits, uint32_t segment_capacity,
                                                  unsigned long long *__restrict__ seeds, double *__restrict__ axis_cache,
                                                  uint8_t *__restrict__ codes, uint32_t *__restrict__ overflow_count)
{
#ifdef XPBD_GJK_TIMING
    unsigned long long tick_ = clock64();
#endif
    // (the overflow list of the expansion that follows starts empty: zeroed here instead of by a 4-byte fill of its own,
    // which cost 5.8 us of stream time per substep)
    if (overflow_count && blockIdx.x == 0 && threadIdx.x == 0)
        *overflow_count = 0;
    constexpr uint32_t PW = 64 / L; // pairs per wave
    __shared__ GjkPairLds<V> s_all[PW];
    __shared__ StagedTables staged;
    const PolytopeTables t = stage_tables<STAGED>(t_global, staged);
    GjkPairLds<V> &s = s_all[threadIdx.x / L];
    const uint32_t slot = blockIdx.x * PW + threadIdx.x / L;
    const uint32_t lane = threadIdx.x % L; // lane inside this pair's group
    if (survivors) {
        n_pairs = *survivor_count;
        if (blockIdx.x == 0 && threadIdx.x < kSurvivorCounters)
            next_survivor_count[threadIdx.x] = 0; // the counters (one per pair class) of the next launch, see SatScratch
    }
    const bool live = slot < n_pairs;
    const uint32_t p = (survivors && live) ? survivors[slot] : slot;

    uint32_t ia = 0, ib = 0, sa = 0, sb = 0;
    ShapeDesc da{}, db{};
    Frame fa{}, fb{};
    if (live) {
        ia = pairs[2 * (size_t)p], ib = pairs[2 * (size_t)p + 1];
        fa = load_record_p1(frames, ia), fb = load_record_p1(frames, ib);
        sa = b.shape_id[ia], sb = b.shape_id[ib];
        da = t.desc[sa], db = t.desc[sb];
    }
    bool usable = live && da.n_verts != 0 && db.n_verts != 0;
    bool queried = live; // a pair the sphere pre-test rejects is not a query: its cached direction stays
    if (PRETEST && usable) {
        const double *ca = t.centroids + 3 * (size_t)sa, *cb = t.centroids + 3 * (size_t)sb;
        const Vec3 between = fb * Vec3{cb[0], cb[1], cb[2]} - fa * Vec3{ca[0], ca[1], ca[2]};
        const double reach = t.radii[sa] + t.radii[sb];
        usable = dot(between, between) < reach * reach;
        queried = usable;
    }
    GJK_TICK(0); // input loads + pre-test
    if (usable)
        stage_world_vertices(s, t, da, db, fa, fb, lane, L);
    __syncthreads();
    GJK_TICK(1); // vertex staging

    // status: 0 separated, 1 penetrating (goes to EPA), 2 degenerate; the simplex is identical on every lane of the group
    int32_t status = 0;
    uint32_t gjk_iters = 0;
    MVert s0{}, s1{}, s2{}, s3{};
    Vec3 separating{0.0, 0.0, 0.0}; // the direction whose support plane proves the pair separated (GjkScratch::axis_cache)
    // warm start: the pair's cached direction seeds the polytope and, if the seed holds, replaces the boolean GJK
    unsigned long long seed_word1 = 0;
    bool seeded = false;
    if (XPBD_GJK_WARM_MODE != 0 && usable && axis_cache) {
        const double *c = axis_cache + 3 * (size_t)p;
        const Vec3 warm{c[0], c[1], c[2]};
        if (dot(warm, warm) > 0.0) { // group-uniform
            MVert v0{}, v1{}, v2{}, v3{}, v4{};
            uint32_t m = 0;
            if (warm_seed<L>(s, da.n_verts, db.n_verts, warm, lane, v0, v1, v2, v3, v4, m)) {
                seeded = true;
                status = 1;
                s0 = v0, s1 = v1, s2 = v2, s3 = v3;
                seed_word1 = (m == 4 ? (unsigned long long)(v4.ia | (v4.ib << 8)) : 0ull) | ((unsigned long long)m << 16);
            }
        }
    }
    if (usable && !seeded) {
        const uint32_t na = da.n_verts, nb = db.n_verts;
        uint32_t n = 1;
        Vec3 d;
        {
            const double *ca = t.centroids + 3 * (size_t)sa, *cb = t.centroids + 3 * (size_t)sb;
            d = fb * Vec3{cb[0], cb[1], cb[2]} - fa * Vec3{ca[0], ca[1], ca[2]};
            if (!(dot(d, d) > 0.0))
                d = Vec3{1.0, 0.0, 0.0};
        }
        s0 = minkowski_support<L>(s, na, nb, d, lane);
        s1 = s2 = s3 = s0;
        d = -s0.w;
        status = 2; // running out of iterations is a degenerate query
        for (uint32_t it = 0; it < kMaxGjkIters; ++it) {
            gjk_iters = it + 1;
            if (!(dot(d, d) > 0.0)) // origin on the simplex: touching / degenerate
                break;
            const MVert pnt = minkowski_support<L>(s, na, nb, d, lane);
            if (!(dot(pnt.w, d) > 0.0)) { // separated (or just touching)
                status = 0;
                separating = d;
                break;
            }
            if (n == 1) {
                s1 = pnt;
                n = 2;
                const Vec3 a = s1.w, ab = s0.w - a, ao = -a;
                if (same_dir(ab, ao)) {
                    d = edge_direction(ab, ao);
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
                    status = 1;
                    break;
                }
            }
        }
    }

    GJK_TICK(2); // the GJK iterations
#ifdef XPBD_GJK_TIMING
    if (threadIdx.x == 0) {
        atomicAdd(&g_gjk_timing[4], 1ull);
        atomicAdd(&g_gjk_timing[5], (unsigned long long)gjk_iters);
    }
#endif
    // verdicts; the wave appends its penetrating pairs to the hit list with ONE atomic
    const bool writer = live && lane == 0;
    if (writer) {
        if (out) {
            out[p].status = status;
            out[p].gjk_iterations = gjk_iters;
            out[p].epa_iterations = 0;
        }
        if (manifolds && status != 1)
            codes[p] = 0; // "no contact" lives in the pair's code byte only (xpbd_pairs.h)
        if (axis_cache && queried) {
            double *c = axis_cache + 3 * (size_t)p;
            c[0] = separating.x, c[1] = separating.y, c[2] = separating.z;
        }
    }
    const bool hit = writer && status == 1;
    const unsigned long long hit_mask = __ballot(hit);
    if (hit_mask) {
        const uint32_t leader = (uint32_t)__ffsll((long long)hit_mask) - 1u;
        uint32_t base = 0;
        const uint32_t segment = blockIdx.x % kHitSegments; // this workgroup's segment of the hit list, with a counter of its own
        if (threadIdx.x == leader)
            base = atomicAdd(hit_counts + segment * kHitCounterStride, (uint32_t)__popcll(hit_mask));
        base = __shfl(base, leader, 64);
        if (hit) {
            const uint32_t slot = base + (uint32_t)__popcll(hit_mask & ((1ull << threadIdx.x) - 1ull));
            hits[(size_t)segment * segment_capacity + slot] = p;
            seeds[2 * (size_t)p] = pack_seed(s0, s1, s2, s3);
            seeds[2 * (size_t)p + 1] = seed_word1;
        }
    }
    GJK_TICK(3); // verdicts and hit list
}

// Result of a finished expansion: the witness points of the closest face `best` (barycentric coordinates of the origin's
// projection), and the contact manifold of the pair.
//
// The manifold (extension decision; checker gjk_manifold of oracle/xpbd_pairs_oracle.c): EPA yields ONE point per pair,
// which lets a box resting on a face rock about that point from substep to substep.  Where the penetration normal is a
// face normal of one of the bodies (the face of A most aligned with n, or the face of B most aligned with -n, within
// kFaceAlign) that face becomes the reference face of a clipped face contact exactly as in the SAT (A on ties); any
// other normal -- an edge-edge contact -- and a clip that leaves no point below the reference plane keep the one point.
// The clipper's polygons and reference face reuse the polytope's vertex rows and face normals: the expansion is over.
constexpr double kFaceAlign = 0.999; // cosine: 2.6 degrees

template <uint32_t L, class S>
__device__ __forceinline__ void epa_emit(S &s, const PolytopeTables &t, const ShapeDesc &da, const ShapeDesc &db, const Frame &fa,
                                         const Frame &fb, uint32_t best, double best_dist, GjkResult *__restrict__ r,
                                         ContactManifold *__restrict__ mf, uint8_t *__restrict__ code, double *__restrict__ axis, uint32_t lane)
{
    constexpr uint32_t P = L < 16 ? L : 16;
    static_assert(sizeof(s.vw) >= P * 3 * sizeof(double) && sizeof(s.fn) >= (P + kMaxFaceVerts) * 3 * sizeof(double),
                  "the clipper reuses the polytope's vertex rows and face normals");
    const Vec3 nrm = ld3(s.fn, best);
    if (lane == 0 && axis) { // the penetration normal warm-starts the pair's next expansion (GjkScratch::axis_cache)
        axis[0] = nrm.x, axis[1] = nrm.y, axis[2] = nrm.z;
    }
    // The witness points (barycentric coordinates of the origin's projection in the closest face: two divisions on one lane)
    // are wanted by the diagnostic result and by the single-point contact only; a face contact -- nearly every hit of a
    // stack or a settled pile -- never looks at them.  They must be taken before the clipper reuses the polytope's rows.
    const uint32_t best_fi = s.fi[best];
    const Vec3 w0 = ld3(s.vw, face_vertex(best_fi, 0)), w1 = ld3(s.vw, face_vertex(best_fi, 1)), w2 = ld3(s.vw, face_vertex(best_fi, 2));
    const uint32_t ia0 = s.via[face_vertex(best_fi, 0)], ia1 = s.via[face_vertex(best_fi, 1)], ia2 = s.via[face_vertex(best_fi, 2)];
    const uint32_t ib0 = s.vib[face_vertex(best_fi, 0)], ib1 = s.vib[face_vertex(best_fi, 1)], ib2 = s.vib[face_vertex(best_fi, 2)];
    auto witness = [&](Vec3 &pa, Vec3 &pb) {
        const Vec3 proj = nrm * best_dist;
        const Vec3 v0 = w1 - w0, v1 = w2 - w0, v2 = proj - w0;
        const double d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
        const double denom = d00 * d11 - d01 * d01;
        const double bv = (d11 * d20 - d01 * d21) / denom, bw = (d00 * d21 - d01 * d20) / denom, bu = 1.0 - bv - bw;
        pa = ld3(s.wa, ia0) * bu + ld3(s.wa, ia1) * bv + ld3(s.wa, ia2) * bw;
        pb = ld3(s.wb, ib0) * bu + ld3(s.wb, ib1) * bv + ld3(s.wb, ib2) * bw;
    };
    if (lane == 0 && r) {
        Vec3 pa, pb;
        witness(pa, pb);
        r->depth = best_dist;
        r->normal[0] = nrm.x, r->normal[1] = nrm.y, r->normal[2] = nrm.z;
        r->point_a[0] = pa.x, r->point_a[1] = pa.y, r->point_a[2] = pa.z;
        r->point_b[0] = pb.x, r->point_b[1] = pb.y, r->point_b[2] = pb.z;
    }
    if (!mf) // group-uniform
        return;
    // face of A most aligned with n, face of B most aligned with -n (first maximum)
    double align[2];
    uint32_t face[2];
    for (uint32_t side = 0; side < 2; ++side) {
        const ShapeDesc &d = side ? db : da;
        const Frame &f = side ? fb : fa;
        const Vec3 n = side ? -nrm : nrm;
        double top = -DBL_MAX;
        uint32_t idx = kNone;
        for (uint32_t k = lane; k < d.n_faces; k += L) {
            const double *pl = t.planes + 4 * (size_t)(d.face0 + k);
            const Plane w = f * Plane{Vec3{pl[0], pl[1], pl[2]}, pl[3]};
            const double a = dot(w.normal, n);
            if (a > top) {
                top = a;
                idx = k;
            }
        }
        reduce_max_first(top, idx, L);
        align[side] = top;
        face[side] = idx;
    }
    uint32_t n_out = 0, iface = 0;
    const uint32_t ref = align[0] >= align[1] ? 0u : 1u; // reference body: 0 = A, 1 = B
    if ((align[0] > align[1] ? align[0] : align[1]) >= kFaceAlign) // group-uniform
        n_out = face_contact_group<L, P>(t, ref ? db : da, ref ? da : db, ref ? fb : fa, ref ? fa : fb, face[ref], ref ? s.wb : s.wa,
                                         ref ? s.wa : s.wb, s.vw, s.fn, s.fn + P, mf, lane, iface);
    if (lane != 0)
        return;
    // (the incident / reference face indices and the separation -best_dist exist in the public SAT layout only)
    (void)iface;
    if (n_out) {
        *code = (uint8_t)(n_out | (ref << kPairCodeFeatureShift));
    } else {
        Vec3 pa, pb;
        witness(pa, pb);
        *code = (uint8_t)(1u | (2u << kPairCodeFeatureShift)); // one point, reference body A, incident body B
        set_single_contact(*mf, pb, pa);                      // pb on the incident body B, pa on the reference body A
    }
}

// EPA of one penetrating pair by one wave; the polytope starts from the simplex k_gjk_pairs left.
__device__ __forceinline__ void epa_pair(GjkLds &s, const BodyArrays &b, const PolytopeTables &t, const double *__restrict__ frames,
                                         const uint32_t *__restrict__ pairs, uint32_t p, unsigned long long seed, unsigned long long seed1,
                                         GjkResult *__restrict__ out, ContactManifold *__restrict__ manifolds, uint8_t *__restrict__ codes,
                                         double *__restrict__ axis_cache, uint32_t lane)
{
    const uint32_t ia = pairs[2 * (size_t)p], ib = pairs[2 * (size_t)p + 1];
    const Frame fa = load_record_p1(frames, ia), fb = load_record_p1(frames, ib);
    const ShapeDesc da = t.desc[b.shape_id[ia]], db = t.desc[b.shape_id[ib]];
    const uint32_t na = da.n_verts, nb = db.n_verts;
    GjkResult *r = out ? out + p : nullptr;
    ContactManifold *mf = manifolds ? manifolds + p : nullptr;
    uint8_t *const code = manifolds ? codes + p : nullptr;
    uint32_t epa_iters = 0;
    auto finish = [&](int32_t st) {
        if (lane == 0) {
            if (r) {
                r->status = st;
                r->epa_iterations = epa_iters;
            }
            if (mf && st != 1)
                *code = 0;
        }
    };

    stage_world_vertices(s, t, da, db, fa, fb, lane, 64);
    __syncthreads();
    // the first polytope: GJK's tetrahedron, or the warm-start pyramid over m top points (pack_seed)
    const uint32_t warm_m = (uint32_t)(seed1 >> 16) & 0xFFu;
    uint32_t nv = warm_m ? warm_m + 1u : 4u, nf = warm_m ? 2u * warm_m - 2u : 4u;
    if (lane < nv) {
        const uint32_t va = lane < 4 ? (uint32_t)(seed >> (8 * lane)) & 0xFFu : (uint32_t)seed1 & 0xFFu;
        const uint32_t vb = lane < 4 ? (uint32_t)(seed >> (32 + 8 * lane)) & 0xFFu : (uint32_t)(seed1 >> 8) & 0xFFu;
        const Vec3 a = ld3(s.wa, va), bb = ld3(s.wb, vb);
        st3(s.vw, lane, a - bb);
        s.via[lane] = (uint8_t)va, s.vib[lane] = (uint8_t)vb;
    }
    __syncthreads();
    {
        bool bad = false;
        if (lane < nf) {
            uint32_t t0, t1, t2, opposite;
            if (warm_m) {
                const uint32_t idx = warm_face_indices(lane, warm_m);
                t0 = idx & 0xFFu, t1 = (idx >> 8) & 0xFFu, t2 = (idx >> 16) & 0xFFu, opposite = idx >> 24;
            } else { // {0,1,2} {0,3,1} {0,2,3} {1,3,2}: face k lacks vertex 3 - k
                t0 = lane == 3 ? 1u : 0u;
                t1 = lane == 0 ? 1u : (lane == 1 ? 3u : (lane == 2 ? 2u : 3u));
                t2 = lane == 0 ? 2u : (lane == 1 ? 1u : (lane == 2 ? 3u : 2u));
                opposite = 3u - lane;
            }
            const Face f = make_face(s, t0, t1, t2, opposite);
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
        const MVert pnt = minkowski_support<64>(s, na, nb, bn, lane);
        if (dot(pnt.w, bn) - best_dist < kEpaTolerance || nv == kMaxEpaVerts)
            break;

        // visibility of "my" two faces (lane, lane + 64)
        const uint32_t f0 = lane, f1 = lane + 64;
        const uint32_t pk[2] = {f0 < nf ? s.fi[f0] : 0u, f1 < nf ? s.fi[f1] : 0u}; // my faces' vertex indices, read once
        const bool vis0 = f0 < nf && dot(ld3(s.fn, f0), pnt.w - ld3(s.vw, face_vertex(pk[0], 0))) > -kEpaCoplanar;
        const bool vis1 = f1 < nf && dot(ld3(s.fn, f1), pnt.w - ld3(s.vw, face_vertex(pk[1], 0))) > -kEpaCoplanar;

        // horizon test of my faces' edges: a->b is on the horizon iff no other VISIBLE face holds b->a (a face never holds
        // the reverse of its own edge: its three vertices are distinct, or make_face would have failed).  The visible faces
        // mark their directed edges in `ve`, every visible face's edge then looks its reverse up, and the marks are
        // cleared again before anything can leave the loop.
        uint32_t hz[2] = {0, 0};
        for (uint32_t w = 0; w < 2; ++w)
            if (w ? vis1 : vis0)
                for (uint32_t e = 0; e < 3; ++e)
                    s.ve[face_vertex(pk[w], e)][face_vertex(pk[w], e + 1)] = 1;
        __syncthreads();
        for (uint32_t w = 0; w < 2; ++w) {
            if (!(w ? vis1 : vis0))
                continue;
            for (uint32_t e = 0; e < 3; ++e)
                if (!s.ve[face_vertex(pk[w], e + 1)][face_vertex(pk[w], e)])
                    hz[w] |= 1u << e;
        }
        __syncthreads();
        for (uint32_t w = 0; w < 2; ++w)
            if (w ? vis1 : vis0)
                for (uint32_t e = 0; e < 3; ++e)
                    s.ve[face_vertex(pk[w], e)][face_vertex(pk[w], e + 1)] = 0;
        // canonical slots: surviving faces keep their order; horizon edges ordered by (face, edge)
        // (exclusive prefix sums over the lanes of 0/1 flags and of 3-bit edge masks: a ballot per bit and a popcount
        // of the lanes below, instead of a six-step shuffle scan each)
        const unsigned long long below = (1ull << lane) - 1ull;
        auto flag_scan = [&](bool flag, uint32_t *total) {
            const unsigned long long m = __ballot(flag);
            *total = (uint32_t)__popcll(m);
            return (uint32_t)__popcll(m & below);
        };
        auto edge_scan = [&](uint32_t bits, uint32_t *total) {
            uint32_t before = 0;
            *total = 0;
            for (uint32_t e = 0; e < 3; ++e) {
                const unsigned long long m = __ballot((bits >> e) & 1u);
                before += (uint32_t)__popcll(m & below);
                *total += (uint32_t)__popcll(m);
            }
            return before;
        };
        uint32_t keep_a, keep_b, ne_a, ne_b;
        const uint32_t kpos0 = flag_scan(f0 < nf && !vis0, &keep_a);
        const uint32_t kpos1 = keep_a + flag_scan(f1 < nf && !vis1, &keep_b);
        const uint32_t epos0 = edge_scan(hz[0], &ne_a);
        const uint32_t epos1 = ne_a + edge_scan(hz[1], &ne_b);
        const uint32_t keep = keep_a + keep_b, ne = ne_a + ne_b;
        if (ne == 0 || keep + ne > kMaxEpaFaces)
            break; // numerical dead end or out of room: report the best face found so far

        // pull my surviving faces into registers, list the horizon, then rewrite the face table
        Face mine[2];
        for (uint32_t w = 0; w < 2; ++w) {
            const uint32_t k = w ? f1 : f0;
            if (k >= nf)
                continue;
            mine[w].i0 = face_vertex(pk[w], 0), mine[w].i1 = face_vertex(pk[w], 1), mine[w].i2 = face_vertex(pk[w], 2);
            if (w ? vis1 : vis0) {
                const uint32_t ea[3] = {mine[w].i0, mine[w].i1, mine[w].i2}, eb[3] = {mine[w].i1, mine[w].i2, mine[w].i0};
                uint32_t q = w ? epos1 : epos0;
                for (uint32_t e = 0; e < 3; ++e)
                    if (hz[w] & (1u << e)) {
                        s.he[q][0] = (uint8_t)ea[e], s.he[q][1] = (uint8_t)eb[e];
                        ++q;
                    }
            } else {
                mine[w].n = ld3(s.fn, k);
                mine[w].dist = s.fd[k];
            }
        }
        if (lane == 0) {
            st3(s.vw, nv, pnt.w);
            s.via[nv] = (uint8_t)pnt.ia, s.vib[nv] = (uint8_t)pnt.ib;
        }
        __syncthreads();
        if (f0 < nf && !vis0)
            store_face(s, kpos0, mine[0]);
        if (f1 < nf && !vis1)
            store_face(s, kpos1, mine[1]);
        bool bad = false;
        for (uint32_t q = lane; q < ne; q += 64) { // the new faces over the horizon edges, one per lane
            const Face f = make_face(s, s.he[q][0], s.he[q][1], nv, kNone);
            store_face(s, keep + q, f);
            bad |= !f.ok;
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
    epa_emit<64>(s, t, da, db, fa, fb, best, best_dist, r, mf, code, axis_cache ? axis_cache + 3 * (size_t)p : nullptr, lane);
    finish(1);
}

// The segmented hit list as its consumers see it: `prefix` (LDS, n_segments + 1 entries) = exclusive prefix sums of the
// segment counts; entry h of the concatenation is hits[segment * capacity + (h - prefix[segment])].  First wave of the block
// fills it (all threads must call: it has a barrier); returns the total.  Also zeroes the NEXT launch's counters (block 0).
__device__ __forceinline__ uint32_t hit_list_open(uint32_t *prefix, const uint32_t *__restrict__ counts, uint32_t n_segments,
                                                  uint32_t *__restrict__ next_counts, uint32_t n_next)
{
    static_assert(kHitSegments <= 64, "one segment counter per lane of the first wave");
    if (threadIdx.x < 64) {
        // one counter per lane and a shuffle scan: ONE round trip to memory (a loop on thread 0 paid one dependent load per segment
        // before a block could start)
        const uint32_t lane = threadIdx.x;
        const uint32_t mine = lane < n_segments ? counts[lane * kHitCounterStride] : 0u;
        uint32_t run = mine;
        for (uint32_t off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(run, off, 64);
            if (lane >= off)
                run += up;
        }
        if (lane < n_segments)
            prefix[lane] = run - mine; // exclusive
        if (lane == 63)
            prefix[n_segments] = run;
    }
    if (blockIdx.x == 0 && threadIdx.x < n_next)
        next_counts[threadIdx.x * kHitCounterStride] = 0;
    __syncthreads();
    return prefix[n_segments];
}

__device__ __forceinline__ uint32_t hit_list_entry(const uint32_t *prefix, uint32_t n_segments, const uint32_t *__restrict__ hits,
                                                   uint32_t capacity, uint32_t h)
{
    uint32_t lo = 0, hi = n_segments; // the segment with prefix[seg] <= h < prefix[seg + 1]
    while (lo + 1 < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (prefix[mid] <= h)
            lo = mid;
        else
            hi = mid;
    }
    return hits[(size_t)lo * capacity + (h - prefix[lo])];
}

// ---------------------------------------------------------------------------------------------------------------------
// EPA by a GROUP of L lanes (L = 16: four hits per wave) for shapes of at most kSubVerts vertices.  A hit of the wave-per-hit
// kernel above spends ~20 k cycles on ~900 VALU instructions: it is a chain of LDS round trips, reductions and fences, and
// what bounds the kernel is how many hits a CU has in flight (11 with one wave and 13.8 KB of LDS per hit).  A group keeps
// a polytope of at most kSubPolyVerts vertices / kSubPolyFaces faces (F = 2 V - 4) in 4.3 KB, so a CU holds four times as
// many hits.  Face f belongs to lane f % L; with at most 64 faces every per-face flag of an iteration (visible, horizon
// edge e) is one bit of a 64-bit mask assembled from per-round ballots, and the canonical slots of the sequential oracle
// -- surviving faces keep their order, new faces follow in (visible face, edge) order -- are popcounts of those masks.
// Same arithmetic per face and the same order as epa_pair, so the same bits.  A hit that would outgrow the small polytope
// is not finished here: it goes to the overflow list and the wave-per-hit kernel redoes it from its simplex.
constexpr uint32_t kSubVerts = 16;      // shape vertices per body
#ifndef XPBD_EPA_SUB_POLY_VERTS
#define XPBD_EPA_SUB_POLY_VERTS 18      // 4 + 14 expansions, 32 faces (a build with 6 sends most hits through the overflow path: used to test it)
#endif
constexpr uint32_t kSubPolyVerts = XPBD_EPA_SUB_POLY_VERTS;
constexpr uint32_t kSubRows = kSubPolyVerts < 16 ? 16 : kSubPolyVerts; // the clipper needs 16 rows
constexpr uint32_t kSubPolyFaces = 2 * kSubPolyVerts - 4; // Euler: a closed triangulated polytope with V vertices has 2 V - 4 faces

struct EpaSubLds : GjkVertsT<kSubVerts> {
    double vw[kSubRows][3];                  // polytope vertices w = wa[via] - wb[vib] (then a clipper polygon)
    uint8_t via[kSubRows], vib[kSubRows];    // their witnesses, as vertex indices
    uint32_t fi[kSubPolyFaces];
    double fn[kSubPolyFaces][3];
    double fd[kSubPolyFaces];
    uint32_t ve[kSubRows];        // the directed-edge marks of GjkLds as bit rows: bit b of ve[a] = edge a -> b (18 <= 32 vertices)
    uint8_t he[kSubPolyFaces][2]; // the horizon of the current iteration (see GjkLds)
};

template <uint32_t L>
__device__ __forceinline__ unsigned long long group_ballot(bool flag)
{
    const unsigned long long wave_mask = __ballot(flag);
    const uint32_t first = (threadIdx.x & 63u) / L * L; // the block is one wave
    return L == 64 ? wave_mask : (wave_mask >> first) & ((1ull << (L & 63u)) - 1ull);
}

// returns false when the hit outgrew the small polytope (nothing has been written for it then)
template <uint32_t L>
__device__ __forceinline__ bool epa_pair_sub(EpaSubLds &s, const BodyArrays &b, const PolytopeTables &t, const double *__restrict__ frames,
                                             const uint32_t *__restrict__ pairs, uint32_t p, unsigned long long seed, unsigned long long seed1,
                                             GjkResult *__restrict__ out, ContactManifold *__restrict__ manifolds, uint8_t *__restrict__ codes,
                                             double *__restrict__ axis_cache, uint32_t lane)
{
    static_assert(kSubPolyFaces <= 64 && L * 4 >= kSubPolyFaces, "face flags live in one 64-bit mask; at most four faces per lane");
    static_assert(kSubRows <= 32, "the directed-edge marks are 32-bit rows");
    constexpr uint32_t R = (kSubPolyFaces + L - 1) / L; // faces per lane (rounds)
    const uint32_t ia = pairs[2 * (size_t)p], ib = pairs[2 * (size_t)p + 1];
    const Frame fa = load_record_p1(frames, ia), fb = load_record_p1(frames, ib);
    const ShapeDesc da = t.desc[b.shape_id[ia]], db = t.desc[b.shape_id[ib]];
    const uint32_t na = da.n_verts, nb = db.n_verts;
    GjkResult *r = out ? out + p : nullptr;
    ContactManifold *mf = manifolds ? manifolds + p : nullptr;
    uint8_t *const code = manifolds ? codes + p : nullptr;
    uint32_t epa_iters = 0;
    auto finish = [&](int32_t st) {
        if (lane == 0) {
            if (r) {
                r->status = st;
                r->epa_iterations = epa_iters;
            }
            if (mf && st != 1)
                *code = 0;
        }
    };
    // first / lowest face with the smallest plane distance, over the group
    auto closest = [&](uint32_t nf, double *dist_out) {
        double bd = DBL_MAX;
        uint32_t bi = kNone;
        for (uint32_t f = lane; f < nf; f += L)
            if (s.fd[f] < bd) {
                bd = s.fd[f];
                bi = f;
            }
        for (uint32_t off = L / 2; off; off >>= 1) {
            const double od = partner(bd, off);
            const uint32_t oi = partner(bi, off);
            if (od < bd || (od == bd && oi < bi)) {
                bd = od;
                bi = oi;
            }
        }
        *dist_out = bd;
        return bi;
    };

    stage_world_vertices(s, t, da, db, fa, fb, lane, L);
    __syncthreads();
    // the first polytope: GJK's tetrahedron, or the warm-start pyramid over m top points (pack_seed)
    const uint32_t warm_m = (uint32_t)(seed1 >> 16) & 0xFFu;
    uint32_t nv = warm_m ? warm_m + 1u : 4u, nf = warm_m ? 2u * warm_m - 2u : 4u;
    if (lane < nv) {
        const uint32_t va = lane < 4 ? (uint32_t)(seed >> (8 * lane)) & 0xFFu : (uint32_t)seed1 & 0xFFu;
        const uint32_t vb = lane < 4 ? (uint32_t)(seed >> (32 + 8 * lane)) & 0xFFu : (uint32_t)(seed1 >> 8) & 0xFFu;
        const Vec3 a = ld3(s.wa, va), bb = ld3(s.wb, vb);
        st3(s.vw, lane, a - bb);
        s.via[lane] = (uint8_t)va, s.vib[lane] = (uint8_t)vb;
    }
    __syncthreads();
    {
        bool bad = false;
        if (lane < nf) {
            uint32_t t0, t1, t2, opposite;
            if (warm_m) {
                const uint32_t idx = warm_face_indices(lane, warm_m);
                t0 = idx & 0xFFu, t1 = (idx >> 8) & 0xFFu, t2 = (idx >> 16) & 0xFFu, opposite = idx >> 24;
            } else { // {0,1,2} {0,3,1} {0,2,3} {1,3,2}: face k lacks vertex 3 - k
                t0 = lane == 3 ? 1u : 0u;
                t1 = lane == 0 ? 1u : (lane == 1 ? 3u : (lane == 2 ? 2u : 3u));
                t2 = lane == 0 ? 2u : (lane == 1 ? 1u : (lane == 2 ? 3u : 2u));
                opposite = 3u - lane;
            }
            const Face f = make_face(s, t0, t1, t2, opposite);
            store_face(s, lane, f);
            bad = !f.ok;
        }
        if (group_ballot<L>(bad)) {
            finish(2);
            return true;
        }
    }
    __syncthreads();
#if XPBD_EPA_TIMING_STOP == 1 // (diagnostic builds: where does the time of a hit go?  wrong results)
    finish(2);
    return true;
#endif

    for (uint32_t it = 0; it < kMaxEpaIters; ++it) {
        epa_iters = it + 1;
        double best_dist;
        const uint32_t best = closest(nf, &best_dist);
        const Vec3 bn = ld3(s.fn, best);
        const MVert pnt = minkowski_support<L>(s, na, nb, bn, lane);
        if (dot(pnt.w, bn) - best_dist < kEpaTolerance)
            break;
        if (nv == kSubPolyVerts)
            return false; // (epa_pair stops at kMaxEpaVerts: this hit needs the large polytope)

        // visibility of my faces f = lane + L * j; bit f of `vis` for the whole group
        bool my_vis[R];
        uint32_t pk[R]; // my faces' vertex indices, read once
        unsigned long long vis = 0, valid = 0;
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) {
            const uint32_t f = lane + L * j;
            pk[j] = f < nf ? s.fi[f] : 0u;
            my_vis[j] = f < nf && dot(ld3(s.fn, f), pnt.w - ld3(s.vw, face_vertex(pk[j], 0))) > -kEpaCoplanar;
            vis |= group_ballot<L>(my_vis[j]) << (L * j);
            valid |= group_ballot<L>(f < nf) << (L * j);
        }
        // horizon: the visible faces mark their directed edges, every edge of a visible face looks its reverse up, the
        // marks are cleared again (see epa_pair)
#pragma unroll
        for (uint32_t j = 0; j < R; ++j)
            if (my_vis[j])
                for (uint32_t e = 0; e < 3; ++e)
                    atomicOr(&s.ve[face_vertex(pk[j], e)], 1u << face_vertex(pk[j], e + 1));
        __syncthreads();
        uint32_t hz[R];
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) {
            hz[j] = 0;
            if (my_vis[j])
                for (uint32_t e = 0; e < 3; ++e)
                    if (!((s.ve[face_vertex(pk[j], e + 1)] >> face_vertex(pk[j], e)) & 1u))
                        hz[j] |= 1u << e;
        }
        __syncthreads();
        unsigned long long edge_mask[3] = {0, 0, 0}; // bit f: edge e of face f is on the horizon
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) {
            if (my_vis[j])
                for (uint32_t e = 0; e < 3; ++e)
                    s.ve[face_vertex(pk[j], e)] = 0; // (every mark of a row belongs to a visible face: all of them go)
#pragma unroll
            for (uint32_t e = 0; e < 3; ++e)
                edge_mask[e] |= group_ballot<L>((hz[j] >> e) & 1u) << (L * j);
        }
        const unsigned long long kept = valid & ~vis;
        const uint32_t keep = (uint32_t)__popcll(kept);
        const uint32_t ne = (uint32_t)(__popcll(edge_mask[0]) + __popcll(edge_mask[1]) + __popcll(edge_mask[2]));
        if (ne == 0)
            break; // numerical dead end: report the best face found so far (as epa_pair)
        if (keep + ne > kSubPolyFaces)
            return false;

        // pull my surviving faces into registers, list the horizon (my horizon edges come after those of all lower faces,
        // in edge order), then rewrite the face table in the canonical order
        Face mine[R];
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) {
            const uint32_t f = lane + L * j;
            if (f >= nf)
                continue;
            mine[j].i0 = face_vertex(pk[j], 0), mine[j].i1 = face_vertex(pk[j], 1), mine[j].i2 = face_vertex(pk[j], 2);
            if (my_vis[j]) {
                const unsigned long long below = (1ull << f) - 1ull;
                uint32_t q = (uint32_t)(__popcll(edge_mask[0] & below) + __popcll(edge_mask[1] & below) + __popcll(edge_mask[2] & below));
                const uint32_t ea[3] = {mine[j].i0, mine[j].i1, mine[j].i2}, eb[3] = {mine[j].i1, mine[j].i2, mine[j].i0};
                for (uint32_t e = 0; e < 3; ++e)
                    if (hz[j] & (1u << e)) {
                        s.he[q][0] = (uint8_t)ea[e], s.he[q][1] = (uint8_t)eb[e];
                        ++q;
                    }
            } else {
                mine[j].n = ld3(s.fn, f);
                mine[j].dist = s.fd[f];
            }
        }
        if (lane == 0) {
            st3(s.vw, nv, pnt.w);
            s.via[nv] = (uint8_t)pnt.ia, s.vib[nv] = (uint8_t)pnt.ib;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) {
            const uint32_t f = lane + L * j;
            if (f < nf && !my_vis[j])
                store_face(s, (uint32_t)__popcll(kept & ((1ull << f) - 1ull)), mine[j]);
        }
        bool bad = false;
        for (uint32_t q = lane; q < ne; q += L) { // the new faces over the horizon edges, one per lane
            const Face nfce = make_face(s, s.he[q][0], s.he[q][1], nv, kNone);
            store_face(s, keep + q, nfce);
            bad |= !nfce.ok;
        }
        nf = keep + ne;
        ++nv;
        __syncthreads();
        if (group_ballot<L>(bad)) {
            finish(2);
            return true;
        }
    }

#if XPBD_EPA_TIMING_STOP == 2
    {
        double bd;
        const uint32_t bf = closest(nf, &bd);
        const Vec3 nrm = ld3(s.fn, bf);
        if (lane == 0 && axis_cache) // (keep the warm start of the next substep, as epa_emit would)
            axis_cache[3 * (size_t)p] = nrm.x, axis_cache[3 * (size_t)p + 1] = nrm.y, axis_cache[3 * (size_t)p + 2] = nrm.z;
    }
    finish(2);
    return true;
#endif
    double best_dist;
    const uint32_t best = closest(nf, &best_dist);
    epa_emit<L>(s, t, da, db, fa, fb, best, best_dist, r, mf, code, axis_cache ? axis_cache + 3 * (size_t)p : nullptr, lane);
    finish(1);
    return true;
}

// Four hits per wave (16 lanes each), grid-stride over the hit list; hits that outgrow the small polytope are appended to
// `overflow` (one atomic each: rare) for k_epa_pairs.  Block 0 zeroes the counter the NEXT k_gjk_pairs launch appends
// through -- the overflow pass (launched after this kernel with the same pointer) only zeroes it again.
// The kernel is a chain of LDS round trips and cross-lane reductions per hit, so what counts is how many hits a CU has in
// flight: 168 VGPRs (3 waves per SIMD; the compiler takes 180 when left alone) and 12.7 KB of LDS per workgroup -- witness
// points as vertex indices, a face's three vertex indices in one word, and no staged shape tables (they cost 2 KB and
// one workgroup per CU: A/B on the mixed pile 1.78e8 -> 1.88e8 body-substeps/s without them) -- give 12 workgroups per CU.
template <uint32_t L>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(L == 8 ? 2 : 3, L == 8 ? 2 : 3)))
k_epa_pairs_sub(BodyArrays b, PolytopeTables t, const double *__restrict__ frames,
                                                      const uint32_t *__restrict__ pairs, GjkResult *__restrict__ out,
                                                      ContactManifold *__restrict__ manifolds, const uint32_t *__restrict__ hit_counts,
                                                      uint32_t *__restrict__ next_hit_counts, const uint32_t *__restrict__ hits,
                                                      uint32_t segment_capacity, const unsigned long long *__restrict__ seeds,
                                                      uint32_t *__restrict__ overflow_count, uint32_t *__restrict__ overflow,
                                                      uint8_t *__restrict__ codes, double *__restrict__ axis_cache)
{
    constexpr uint32_t PW = 64 / L;
    __shared__ EpaSubLds s_all[PW];
    EpaSubLds &s = s_all[threadIdx.x / L];
    const uint32_t lane = threadIdx.x % L;
    for (uint32_t k = lane; k < kSubRows; k += L)
        s.ve[k] = 0;
    __shared__ uint32_t prefix[kHitSegments + 1];
    const uint32_t n_hits = hit_list_open(prefix, hit_counts, kHitSegments, next_hit_counts, kHitSegments);
    for (uint32_t h = blockIdx.x * PW + threadIdx.x / L; h < n_hits; h += gridDim.x * PW) {
        const uint32_t p = hit_list_entry(prefix, kHitSegments, hits, segment_capacity, h);
        const bool done = epa_pair_sub<L>(s, b, t, frames, pairs, p, seeds[2 * (size_t)p], seeds[2 * (size_t)p + 1], out, manifolds, codes, axis_cache, lane);
        if (!done) {
            // the marks of the interrupted iteration were cleared before the exits; hand the hit over
            if (lane == 0)
                overflow[atomicAdd(overflow_count, 1u)] = p;
        }
        __syncthreads(); // the next hit reuses the LDS
    }
}

// One wave per block, grid-stride over a hit list (n_segments segments of `segment_capacity` entries; the overflow list of
// k_epa_pairs_sub is one segment).  Block 0 also zeroes the counters the NEXT k_gjk_pairs launch appends through (the two
// sets alternate, see GjkScratch).
__global__ void __launch_bounds__(64) k_epa_pairs(BodyArrays b, PolytopeTables t, const double *__restrict__ frames,
                                                  const uint32_t *__restrict__ pairs, GjkResult *__restrict__ out,
                                                  ContactManifold *__restrict__ manifolds, const uint32_t *__restrict__ hit_counts,
                                                  uint32_t n_segments, uint32_t *__restrict__ next_hit_counts, const uint32_t *__restrict__ hits,
                                                  uint32_t segment_capacity, const unsigned long long *__restrict__ seeds,
                                                  uint8_t *__restrict__ codes, double *__restrict__ axis_cache)
{
    __shared__ GjkLds s;
    for (uint32_t k = threadIdx.x; k < sizeof(s.ve) / 4; k += 64)
        reinterpret_cast<uint32_t *>(&s.ve[0][0])[k] = 0;
    __shared__ uint32_t prefix[kHitSegments + 1];
    const uint32_t n_hits = hit_list_open(prefix, hit_counts, n_segments, next_hit_counts, kHitSegments);
    for (uint32_t h = blockIdx.x; h < n_hits; h += gridDim.x) {
        const uint32_t p = hit_list_entry(prefix, n_segments, hits, segment_capacity, h);
        epa_pair(s, b, t, frames, pairs, p, seeds[2 * (size_t)p], seeds[2 * (size_t)p + 1], out, manifolds, codes, axis_cache, threadIdx.x);
        __syncthreads(); // the next hit reuses the LDS
    }
}

} // namespace

// seeds (16 bytes per pair), the segmented hit list (4 bytes per pair + one wave's worth of slack per segment), the overflow
// list (4 bytes per pair) and its counter
#ifdef XPBD_GJK_TIMING
extern "C" int xpbd_debug_gjk_timing(unsigned long long out[8], int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gjk_timing), sizeof(unsigned long long) * 8) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_gjk_timing), zero, sizeof zero) != hipSuccess)
            return -1;
    }
    return 0;
}
#endif

size_t gjk_scratch_bytes(uint32_t n_pairs) { return (size_t)n_pairs * 24 + (size_t)kHitSegments * 64 * 4 + 16; }

hipError_t launch_gjk_epa_pairs(const BodyArrays &b, const PolytopeTables &t, const double *frames, const uint32_t *pairs,
                                uint32_t n_pairs, GjkResult *out, ContactManifold *manifolds, GjkScratch &scratch, bool sphere_pretest,
                                SatScratch *list, hipStream_t stream)
{
    if (n_pairs == 0)
        return hipSuccess;
    unsigned long long *seeds = static_cast<unsigned long long *>(scratch.pairs_scratch);
    uint32_t *hits = reinterpret_cast<uint32_t *>(seeds + 2 * (size_t)n_pairs);
    uint32_t *count = scratch.counters + (scratch.calls & 1u) * kHitSegments * kHitCounterStride;
    uint32_t *next = scratch.counters + ((scratch.calls + 1u) & 1u) * kHitSegments * kHitCounterStride;
    ++scratch.calls;
    const uint32_t *survivors = nullptr;
    uint32_t *survivor_count = nullptr, *next_survivor_count = nullptr;
    if (sphere_pretest && list && manifolds) { // two-pass form: the pre-test pass answers the rejected pairs
        if (hipError_t e = launch_pair_pretest(b, t, frames, pairs, n_pairs, scratch.codes, *list, &survivor_count, &next_survivor_count, stream,
                                               false, scratch.axis_cache))
            return e;
        survivors = list->survivors;
        sphere_pretest = false; // the survivors have passed it
    }
    const bool staged = t.n_shapes <= kStageShapes && t.total_verts <= kStageVerts;
    // the cache is part of the pipeline's semantics, and the pre-test pass is where it is consulted: without that pass
    // (the diagnostic entry point, or a caller that wants the pre-test inside the kernel) it is neither read nor written
    double *const axis = survivors ? scratch.axis_cache : nullptr;
    uint8_t *const codes = manifolds ? scratch.codes : nullptr;
    uint32_t segment_capacity = 0;
    // the overflow list of the small-polytope expansion (below) and its counter, behind the hit list
    uint32_t *const overflow_count = t.max_verts <= kSubVerts ? hits + n_pairs + kHitSegments * 64 + n_pairs : nullptr;
    auto launch = [&](auto lanes, auto pretest) {
        constexpr uint32_t L = decltype(lanes)::value, V = L == 32 ? kMaxV : 16;
        constexpr bool P = decltype(pretest)::value;
        const dim3 grid((n_pairs + 64 / L - 1) / (64 / L));
        segment_capacity = (grid.x + kHitSegments - 1) / kHitSegments * (64 / L); // what the workgroups of one segment can append
        if (staged)
            hipLaunchKernelGGL((k_gjk_pairs<L, V, P, true>), grid, dim3(64), 0, stream, b, t, frames, pairs, n_pairs, survivors, survivor_count,
                               next_survivor_count, out, manifolds, count, hits, segment_capacity, seeds, axis, codes, overflow_count);
        else
            hipLaunchKernelGGL((k_gjk_pairs<L, V, P, false>), grid, dim3(64), 0, stream, b, t, frames, pairs, n_pairs, survivors, survivor_count,
                               next_survivor_count, out, manifolds, count, hits, segment_capacity, seeds, axis, codes, overflow_count);
    };
    using std::integral_constant;
    // lanes per pair: XPBD_GJK_SMALL_LANES for shapes of at most 16 vertices (the simplex logic is replicated on every
    // lane of a group, so narrow groups amortise it over more pairs), 32 above
    if (t.max_verts <= 16 && sphere_pretest)
        launch(integral_constant<uint32_t, XPBD_GJK_SMALL_LANES>{}, std::true_type{});
    else if (t.max_verts <= 16)
        launch(integral_constant<uint32_t, XPBD_GJK_SMALL_LANES>{}, std::false_type{});
    else if (sphere_pretest)
        launch(integral_constant<uint32_t, 32>{}, std::true_type{});
    else
        launch(integral_constant<uint32_t, 32>{}, std::false_type{});
    if (t.max_verts <= kSubVerts) {
        // small shapes: four hits per wave; the rare hit that outgrows the small polytope is redone by the wave-per-hit kernel
        uint32_t *overflow = hits + n_pairs + kHitSegments * 64; // (overflow_count = overflow + n_pairs: zeroed by k_gjk_pairs above)
        // 8 lanes per hit where no face has more than four vertices (the manifold's clipper runs one polygon vertex per
        // lane, and a quadrilateral clipped against a quadrilateral has at most eight), 16 otherwise
        if (t.max_face_verts <= 4) {
            const uint32_t groups = (n_pairs + 7) / 8;
            hipLaunchKernelGGL(k_epa_pairs_sub<8>, dim3(groups < kEpaBlocks ? groups : kEpaBlocks), dim3(64), 0, stream, b, t, frames, pairs, out,
                               manifolds, count, next, hits, segment_capacity, seeds, overflow_count, overflow, codes, axis);
        } else {
            const uint32_t groups = (n_pairs + 3) / 4;
            hipLaunchKernelGGL(k_epa_pairs_sub<16>, dim3(groups < kEpaBlocks ? groups : kEpaBlocks), dim3(64), 0, stream, b, t, frames, pairs, out,
                               manifolds, count, next, hits, segment_capacity, seeds, overflow_count, overflow, codes, axis);
        }
        hipLaunchKernelGGL(k_epa_pairs, dim3(n_pairs < 256 ? n_pairs : 256), dim3(64), 0, stream, b, t, frames, pairs, out, manifolds,
                           overflow_count, 1u, next, overflow, n_pairs, seeds, codes, axis);
        return hipGetLastError();
    }
    const uint32_t blocks = n_pairs < kEpaBlocks ? n_pairs : kEpaBlocks;
    hipLaunchKernelGGL(k_epa_pairs, dim3(blocks), dim3(64), 0, stream, b, t, frames, pairs, out, manifolds, count, kHitSegments, next, hits,
                       segment_capacity, seeds, codes, axis);
    return hipGetLastError();
}

} // namespace xpbd
