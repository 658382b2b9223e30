// xpbd_pairs.hip -- body-body contact EXTENSION: SAT narrowphase for gfx950, one group of 16 / 32 / 64 lanes per pair.
//
// The reference stops after the A-face query of `sat` (src/collision.rs:37-121, an uncalled
// stub); this finishes the commented-out sketch there.  Parity is UNPINNED (no reference result
// exists); the checker is oracle/xpbd_pairs_oracle.c.  Conventions kept from the reference:
//   face_axes_separation  src/collision.rs:123-149  support = LAST maximum under f64::total_cmp,
//                                                   face   = FIRST maximum ('>')
//   edge_axes_separation  src/collision.rs:151-197  parallel edges give a NaN axis and contribute nothing,
//                                                   first maximum; its E_A x E_B pair enumeration is replaced
//                                                   by the classic test over UNIQUE edge directions
//   feature choice        src/collision.rs:47-59,89-92 (comments there)
//   reference plane / incident face  src/collision.rs:66, 76-85 (first minimum of n . n_ref)
//
// Mapping: ONE GROUP OF LANES = ONE CANDIDATE PAIR (sat_pair below; 4, 2 or 1 pairs per wave by the largest shape).
// Both bodies' vertices are transformed once into LDS (world space, and each into the other's local space); the
// lanes of the group then run in parallel over faces x vertices, over the pairs of unique edge directions and over the
// incident body's faces, and combine with __shfl_xor reductions that carry (value, index) so the reference's
// first/last tie-breaks survive the parallel order.  Clipping runs one polygon vertex per lane with prefix sums.
// Kernels: k_sat_pairs (every listed pair; optionally with the contact pipeline's tight-sphere pre-test),
// k_pair_pretest + k_sat_survivors (the pre-test as a pass of its own, the SAT over the survivors only).
#include <cfloat>
#include <cstdlib>

#include <type_traits>

#include "xpbd_clip.hpp"
#include "xpbd_device.hpp"
#include "xpbd_pairs.h"

namespace xpbd {
namespace {

constexpr uint32_t kMaxV = XPBD_MAX_SHAPE_VERTS; // 32
#ifndef XPBD_SAT_MIN_WAVES_PER_SIMD
#define XPBD_SAT_MIN_WAVES_PER_SIMD 4 // <= 128 VGPRs
#endif
#ifndef XPBD_SAT_BOX_LANES
#define XPBD_SAT_BOX_LANES 8 // shapes of <= 8 vertices and faces with <= 4 vertices per face: clipped polygons have <= 8 vertices
#endif
#ifndef XPBD_SAT_DENSE_BOX_LANES
#define XPBD_SAT_DENSE_BOX_LANES 4 // ... in dense scenes with at least 65 536 pairs (for_shape_maxima)
#endif
#ifndef XPBD_SAT_SMALL_LANES
#define XPBD_SAT_SMALL_LANES 16 // shapes of <= 8 vertices and faces otherwise
#endif
#ifndef XPBD_SAT_LS_LANES
#define XPBD_SAT_LS_LANES 16 // a large (<= 16 vertices) against a small shape; A/B on the mixed pile: 8 lanes 2.31e8, 16 lanes 2.44e8, 32 lanes 2.32e8
                             // (and 64 instead of 32 lanes for large-large pairs: 2.19e8)
#endif
#ifndef XPBD_SAT_MID_LANES
#define XPBD_SAT_MID_LANES 32 // A/B on 65 536 mixed polyhedra: 16 lanes 5.19e8, 32 lanes 5.59e8 body-substeps/s
#endif
#ifndef XPBD_SAT_TIMING_STOP
#define XPBD_SAT_TIMING_STOP 0
#endif
#ifndef XPBD_SAT_EDGE_CACHE_GAP
#define XPBD_SAT_EDGE_CACHE_GAP 1.0e-5
#endif
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kAxisCacheEdge = 0x8000u; // SatScratch::axis_cache: 0 none, 1 + 2 f / 2 + 2 f face f of A / B, kAxisCacheEdge | q edge axis q
constexpr uint32_t kWidePairCount = 32768; // 8 pairs per wave x 4 096 wave slots (1 024 SIMDs x 4)

// Working set of one pair.  V = vertex capacity per body, P = polygon capacity of the clipper (the launcher picks the
// group width and both from the largest shape).  Two halves: the world-space vertices, and a scratch area that is, in
// turn, the vertices in the other body's local space (face queries), the world-space edge directions (edge axes) and
// the clipper's ping-pong polygons; the reference face's vertices take the place of the INCIDENT body's world-space
// vertices once the incident face has been copied out.  The kernel is bound by latency (LDS round trips, shuffles), so
// what counts is how many waves a CU holds: with everything side by side (1 376 bytes per box pair, eight pairs per
// wave) the LDS allowed 14 waves per CU, now (800 bytes) 25.
// The tail pads the record so that its stride staggers the records of the PW pairs of a wave over the LDS banks -- all
// groups read "vertex k of my pair" in the same instruction, and records a multiple of 64 dwords apart would put those
// reads on the same banks (measured: SQ_LDS_BANK_CONFLICT = 86 % of the LDS cycles).
template <uint32_t V, uint32_t P, uint32_t PW>
struct PairLds {
    static constexpr uint32_t kVerts = V;
    static_assert(kMaxFaceVerts <= V, "the reference face reuses vertex rows");
    static constexpr uint32_t kScratchRows = P > V ? P : V;
    static constexpr uint32_t kBaseDwords = 2 * (2 * V * 3 + 2 * kScratchRows * 3);
    // record stride = an ODD multiple of 64 / PW dwords (mod 64): the PW records then start on PW different bank
    // groups; of the candidates take the one that needs the least padding
    static constexpr uint32_t pad_dwords()
    {
        if (PW == 1)
            return 0;
        uint32_t best = 64;
        for (uint32_t odd = 1; odd < 2 * PW; odd += 2) {
            const uint32_t pad = (odd * (64 / PW) + 64 - kBaseDwords % 64) % 64;
            best = pad < best ? pad : best;
        }
        return best;
    }
    static constexpr uint32_t kPadDwords = pad_dwords();
    double world[2][V][3]; // world-space vertices of A (0) and B (1)
    union {
        double local[2][V][3]; // [0]: A's vertices in B-local space, [1]: B's vertices in A-local space; then the edge directions
        double poly[2][P][3];  // clipping ping-pong
    };
    uint32_t pad[kPadDwords ? kPadDwords : 2];
};

__device__ __forceinline__ bool finite3(Vec3 v)
{
    return fabs(v.x) <= DBL_MAX && fabs(v.y) <= DBL_MAX && fabs(v.z) <= DBL_MAX;
}

// Vertex of `verts[0..n)` with the LAST maximal dot(v, dir) under the total order (Iterator::max_by).
__device__ __forceinline__ Vec3 support_last_max(const double (*verts)[3], uint32_t n, Vec3 dir)
{
    Vec3 best = ld3(verts, 0);
    long long best_key = total_key(dot(best, dir));
    for (uint32_t k = 1; k < n; ++k) {
        const Vec3 x = ld3(verts, k);
        const long long key = total_key(dot(x, dir));
        if (best_key <= key) {
            best_key = key;
            best = x;
        }
    }
    return best;
}

__device__ __forceinline__ double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

// The SAT of ONE pair by a group of L lanes (`lane` = lane inside the group, `s` = the group's LDS record).
// L = 64 (any shapes), 32, or 16 (for boxes every stage -- 8 + 8 vertices, 6 + 6 faces, 3 x 3 edge axes, <= 16 polygon
// points -- fits in 16 lanes, so a whole wave per pair would idle 3/4 of its lanes).  All loops stride by the group
// width and all shuffles stay inside the group, so every instantiation produces the same bits.  The block is ONE
// wave, hence __syncthreads() is a wave-local fence and the groups of a wave may diverge freely (one pair separated,
// the next one clipping).
template <uint32_t L, class Lds, class M>
__device__ __forceinline__ void sat_pair(Lds &s, const BodyArrays &b, const PolytopeTables &t, const double *__restrict__ frames,
                                         const uint32_t *__restrict__ pairs, uint32_t p, M *__restrict__ out, uint32_t lane,
                                         uint16_t *__restrict__ axis_cache = nullptr, uint8_t *__restrict__ codes = nullptr, bool cache_edge_axes = false)
{
    constexpr uint32_t H = L / 2;             // lanes per body in the two-sided stages
    constexpr uint32_t P = (Lds::kVerts <= 8 && L <= 8) ? 8 : 16; // polygon capacity of the clipper: 8 in the box classes (the launcher
                                                                  // takes them only where no polygon can exceed 8 vertices), else 16;
                                                                  // groups narrower than that take several turns (xpbd_clip.hpp)
    // ---- group-uniform inputs ---------------------------------------------------------------------
    const uint32_t ia = pairs[2 * (size_t)p], ib = pairs[2 * (size_t)p + 1];
    const Frame fa = load_record_p1(frames, ia), fb = load_record_p1(frames, ib);
    const Frame fa_inv = inverse(fa), fb_inv = inverse(fb);
    const uint32_t sa = b.shape_id[ia], sb = b.shape_id[ib];
    const ShapeDesc da = t.desc[sa], db = t.desc[sb];

    M *m = out + p;
    // the pair's verdict (lane 0): the header of the public layout, the pipeline's code byte (xpbd_pairs.h)
    auto answer = [&](uint32_t n_points, uint32_t feature, uint32_t index_a, uint32_t index_b, double separation) {
        if (codes)
            codes[p] = (uint8_t)(n_points | (feature << kPairCodeFeatureShift));
        set_header(*m, n_points, feature, index_a, index_b, separation);
    };
    auto no_contact = [&]() {
        if (codes)
            codes[p] = 0;
        else
            set_header(*m, 0, 0, 0, 0, 0.0);
    };
    if (da.n_verts == 0 || db.n_verts == 0 || da.n_faces == 0 || db.n_faces == 0) {
        if (lane == 0)
            no_contact(); // the reference's .unwrap() / index would panic; the extension reports "no contact"
        return;
    }

    // ---- both vertex sets into LDS: world space, and the other body's local space -----------------
    const uint32_t half = lane / H, k = lane % H; // first half of the group works for A, second for B
    {
        const ShapeDesc dm = half ? db : da;
        for (uint32_t vtx = k; vtx < dm.n_verts; vtx += H) {
            const double *v = t.verts + 3 * (size_t)(dm.vert0 + vtx);
            const Vec3 w = (half ? fb : fa) * Vec3{v[0], v[1], v[2]};
            st3(s.world[half], vtx, w);
            st3(s.local[half], vtx, (half ? fa_inv : fb_inv) * w); // frames.0.inverse() * (frames.1 * p), :136-137
        }
    }
    __syncthreads();

    // ---- face queries (src/collision.rs:123-149): A's faces on the first half, B's on the second -------
    double fdist = -DBL_MAX;
    uint32_t fidx = kNone;
    {
        const ShapeDesc dm = half ? db : da;
        const uint32_t n_other = half ? da.n_verts : db.n_verts;
        for (uint32_t f = k; f < dm.n_faces; f += H) {
            const double *pl = t.planes + 4 * (size_t)(dm.face0 + f);
            const Vec3 n{pl[0], pl[1], pl[2]};
            const Vec3 sup = support_last_max(s.local[half ^ 1u], n_other, -n);
            const double dist = dot(n, sup) - pl[3];
            if (dist > fdist) { // ascending f on this lane: first maximum; a NaN never beats f64::MIN
                fdist = dist;
                fidx = f;
            }
        }
    }
    reduce_max_first(fdist, fidx, H);
    // every lane of a half holds its half's result; the level-H partner is in the other half
    const double fdist_other = partner(fdist, H);
    const uint32_t fidx_other = partner(fidx, H);
    const double qa = half ? fdist_other : fdist, qb = half ? fdist : fdist_other;
    const uint32_t face_a = half ? fidx_other : fidx, face_b = half ? fidx : fidx_other;
    if (qa >= 0.0 || qb >= 0.0 || face_a == kNone || face_b == kNone) {
        if (lane == 0) {
            no_contact();
            if (axis_cache) // the face that separates the pair: the pre-test pass tries it first in the next substep
                axis_cache[p] = (uint16_t)(qa >= 0.0 ? 1u + 2u * face_a : (qb >= 0.0 ? 2u + 2u * face_b : 0u));
        }
        return;
    }
    if (axis_cache && lane == 0)
        axis_cache[p] = 0; // not separated by a face axis (any more)
#if XPBD_SAT_TIMING_STOP == 1 // (diagnostic builds: where does the time of a pair go?  wrong results)
    if (lane == 0)
        no_contact();
    return;
#endif

    // ---- edge axes: (unique edge direction of A) x (unique edge direction of B), strided over the group ----
    // n = normalize(dA x dB) pointing from A's centroid to B's; separation = min_B n.b - max_A n.a.
    // Parallel directions give a NaN axis and contribute nothing; first maximum wins (ascending pair index).
    double ebest = -DBL_MAX;
    uint32_t eq = kNone;
    const double *cca = t.centroids + 3 * (size_t)sa, *ccb = t.centroids + 3 * (size_t)sb;
    const Vec3 a_to_b = fb * Vec3{ccb[0], ccb[1], ccb[2]} - fa * Vec3{cca[0], cca[1], cca[2]};
    // The world-space edge directions of both bodies, once per pair: every axis below is the cross product of one of
    // A's with one of B's, and rotating them inside the axis loop cost two quaternion rotations and six global loads
    // per axis.  They take the place of `local` (the face queries above were its last readers) when they fit.
    const bool dirs_staged = da.n_dirs <= Lds::kVerts && db.n_dirs <= Lds::kVerts;
    __syncthreads();
    if (dirs_staged) {
        for (uint32_t d = lane; d < da.n_dirs + db.n_dirs; d += L) {
            const bool of_b = d >= da.n_dirs;
            const uint32_t kd = of_b ? d - da.n_dirs : d;
            const double *dd = t.edge_dirs + 3 * (size_t)((of_b ? db.dir0 : da.dir0) + kd);
            st3(s.local[of_b ? 1 : 0], kd, (of_b ? fb : fa).rotation * Vec3{dd[0], dd[1], dd[2]});
        }
        __syncthreads();
    }
    auto edge_axis = [&](uint32_t i, uint32_t j, Vec3 &axis) -> bool {
        Vec3 n;
        if (dirs_staged) {
            n = normalized(cross(ld3(s.local[0], i), ld3(s.local[1], j)));
        } else {
            const double *da_ = t.edge_dirs + 3 * (size_t)(da.dir0 + i), *db_ = t.edge_dirs + 3 * (size_t)(db.dir0 + j);
            n = normalized(cross(fa.rotation * Vec3{da_[0], da_[1], da_[2]}, fb.rotation * Vec3{db_[0], db_[1], db_[2]}));
        }
        if (!finite3(n))
            return false;
        if (dot(n, a_to_b) < 0.0)
            n = -n;
        axis = n;
        return true;
    };
    {
        const uint32_t total = da.n_dirs * db.n_dirs;
        for (uint32_t q = lane; q < total; q += L) {
            const uint32_t i = q / db.n_dirs, j = q - i * db.n_dirs;
            Vec3 axis;
            if (!edge_axis(i, j, axis))
                continue;
            double reach_a = dot(ld3(s.world[0], 0), axis), reach_b = dot(ld3(s.world[1], 0), axis);
            for (uint32_t v = 1; v < da.n_verts; ++v) {
                const double rr = dot(ld3(s.world[0], v), axis);
                if (rr > reach_a)
                    reach_a = rr;
            }
            for (uint32_t v = 1; v < db.n_verts; ++v) {
                const double rr = dot(ld3(s.world[1], v), axis);
                if (rr < reach_b)
                    reach_b = rr;
            }
            const double dist = reach_b - reach_a;
            if (dist > ebest) { // ascending q on this lane: first maximum
                ebest = dist;
                eq = q;
            }
        }
    }
    reduce_max_first(ebest, eq, L);
    if (ebest >= 0.0) {
        if (lane == 0) {
            no_contact();
            // the edge axis that separates the pair: the pre-test pass tries it first in the next substep -- if the gap is
            // worth it: a resting contact is "separated" by a few nanometres after one solve and touches again after the next
            // integration, and trying its axis first would only be paid for (XPBD_SAT_EDGE_CACHE_GAP, metres)
            if (axis_cache && cache_edge_axes && ebest > XPBD_SAT_EDGE_CACHE_GAP)
                axis_cache[p] = (uint16_t)(kAxisCacheEdge | eq);
        }
        return;
    }

#if XPBD_SAT_TIMING_STOP == 2
    if (lane == 0)
        no_contact();
    return;
#endif
    // ---- feature choice (src/collision.rs:57-59, 89-92 as comments; kEdgeBias is this extension's) --
    const double face_best = qa > qb ? qa : qb;
    const bool use_edges = eq != kNone && ebest > face_best + kEdgeBias;

    if (use_edges) {
        // supporting edges of the winning axis: A's edge of that direction furthest along the axis, B's edge
        // furthest against it (sum of the endpoint projections, first extremum); edges strided over the group
        const uint32_t di = eq / db.n_dirs, dj = eq - di * db.n_dirs;
        Vec3 axis;
        (void)edge_axis(di, dj, axis);
        double sa_best = -DBL_MAX, sb_best = DBL_MAX;
        uint32_t edge_i = kNone, edge_j = kNone;
        for (uint32_t e = lane; e < da.n_edges; e += L) {
            if (t.edge_dir_id[da.edge0 + e] != di)
                continue;
            const uint32_t *ev = t.edges + 2 * (size_t)(da.edge0 + e);
            const double sp = dot(ld3(s.world[0], ev[0]), axis) + dot(ld3(s.world[0], ev[1]), axis);
            if (sp > sa_best) {
                sa_best = sp;
                edge_i = e;
            }
        }
        for (uint32_t e = lane; e < db.n_edges; e += L) {
            if (t.edge_dir_id[db.edge0 + e] != dj)
                continue;
            const uint32_t *ev = t.edges + 2 * (size_t)(db.edge0 + e);
            const double sp = dot(ld3(s.world[1], ev[0]), axis) + dot(ld3(s.world[1], ev[1]), axis);
            if (sp < sb_best) {
                sb_best = sp;
                edge_j = e;
            }
        }
        reduce_max_first(sa_best, edge_i, L);
        reduce_min_first(sb_best, edge_j, L);
        if (edge_i == kNone)
            edge_i = 0;
        if (edge_j == kNone)
            edge_j = 0;
        if (lane == 0) {
            const uint32_t i = edge_i, j = edge_j;
            const uint32_t *ea = t.edges + 2 * (size_t)(da.edge0 + i), *eb = t.edges + 2 * (size_t)(db.edge0 + j);
            const Vec3 a0 = ld3(s.world[0], ea[0]), a1 = ld3(s.world[0], ea[1]);
            const Vec3 b0 = ld3(s.world[1], eb[0]), b1 = ld3(s.world[1], eb[1]);
            const Vec3 d1 = a1 - a0, d2 = b1 - b0, r = a0 - b0;
            const double a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), c = dot(d1, r), bb = dot(d1, d2);
            const double denom = a * e - bb * bb;
            double sp = clamp01((bb * f - c * e) / denom);
            double tp = (bb * sp + f) / e;
            if (tp < 0.0) {
                tp = 0.0;
                sp = clamp01(-c / a);
            } else if (tp > 1.0) {
                tp = 1.0;
                sp = clamp01((bb - c) / a);
            }
            const Vec3 pa = a0 + d1 * sp, pb = b0 + d2 * tp;
            answer(1, 2, i, j, ebest);
            set_single_contact(*m, pb, pa);
        }
        return;
    }

    // ---- face contact: reference body R (A if qa is the maximum, else B), incident body I ----------
    const uint32_t r = (qa == face_best) ? 0u : 1u; // index into s.world: 0 = A, 1 = B
    const uint32_t ref_face = r ? face_b : face_a;
    const Frame fr = r ? fb : fa, fi = r ? fa : fb;
    const ShapeDesc dr = r ? db : da, di = r ? da : db;
    // clipping scratch: the polygons take the place of the edge directions, the reference face that of the INCIDENT
    // body's world-space vertices (see face_contact_group)
    uint32_t iface;
    const uint32_t n_out = face_contact_group<L, P>(t, dr, di, fr, fi, ref_face, s.world[r], s.world[r ^ 1u], s.poly[0], s.poly[1],
                                                    s.world[r ^ 1u], m, lane, iface);
    if (lane != 0)
        return;
    answer(n_out, r, r ? iface : face_a, r ? face_b : iface, face_best);
}

// Tight bounding spheres (centroid, largest vertex distance; no velocity term, no pad) of pair p overlap?  The neighbour
// lists come from spheres inflated by a whole frame of travel, so in a given substep most pairs of a loose scene are
// nowhere near each other (mixed scene: three in four).  Disjoint spheres cannot touch: the contact pipeline answers
// "no contact" for them without running the SAT (semantics: op_contacts_substep of the oracle).
__device__ __forceinline__ bool tight_spheres_overlap(const BodyArrays &b, const PolytopeTables &t, const double *__restrict__ frames,
                                                      const uint32_t *__restrict__ pairs, uint32_t p)
{
    const uint32_t ia = pairs[2 * (size_t)p], ib = pairs[2 * (size_t)p + 1];
    const uint32_t sa = b.shape_id[ia], sb = b.shape_id[ib];
    const double *ca = t.centroids + 3 * (size_t)sa, *cb = t.centroids + 3 * (size_t)sb;
    const Vec3 between = load_record_p1(frames, ib) * Vec3{cb[0], cb[1], cb[2]} - load_record_p1(frames, ia) * Vec3{ca[0], ca[1], ca[2]};
    const double reach = t.radii[sa] + t.radii[sb];
    return dot(between, between) < reach * reach;
}

template <uint32_t L, uint32_t V>
struct SatLds {
    static constexpr uint32_t PW = 64 / L;            // pairs per wave
    static constexpr uint32_t P = (V <= 8 && L <= 8) ? 8 : 16;   // polygon capacity (sat_pair)
    static_assert(V == 8 || V == 16 || V == kMaxV, "vertex capacity per body");
    using Record = PairLds<V, P, PW>;
    static_assert(sizeof(Record) % 8 == 0, "pair records must stay 8-byte aligned");
};

// One group of L lanes per pair, 64 / L pairs per wave, pairs in list order.  PRETEST: the contact pipeline's form.
template <uint32_t L, uint32_t V, bool PRETEST, class M>
__global__ void __launch_bounds__(64, XPBD_SAT_MIN_WAVES_PER_SIMD) k_sat_pairs(BodyArrays b, PolytopeTables t,
                                                                               const double *__restrict__ frames,
                                                                               const uint32_t *__restrict__ pairs, uint32_t n_pairs,
                                                                               M *__restrict__ out, uint8_t *__restrict__ codes)
{
    using Lds = typename SatLds<L, V>::Record;
    __shared__ Lds s_all[SatLds<L, V>::PW];
    const uint32_t group = threadIdx.x / L, lane = threadIdx.x % L;
    const uint32_t p = blockIdx.x * SatLds<L, V>::PW + group;
    if (p >= n_pairs)
        return;
    if (PRETEST && !tight_spheres_overlap(b, t, frames, pairs, p)) {
        if (lane == 0)
            codes[p] = 0; // (PRETEST is the pipeline's form: it always has the code bytes)
        return;
    }
    sat_pair<L>(s_all[group], b, t, frames, pairs, p, out, lane, nullptr, codes);
}

// The pre-test as a pass of its own, one LANE per pair: rejected pairs are answered, the others are appended to a
// survivor list -- with CLASSES to one list per pair class (class 0 from the front of `survivors`, class 1 from the back).
// ONE atomic per list and 1024-pair workgroup (same-address atomics serialise at ~10-25 ns each: one per wave made this
// pass 25 us for 50 000 pairs); the order of a list is irrelevant, results go to out[p].
constexpr uint32_t kPretestBlock = 1024;

// GjkScratch::axis_cache, one thread per pair: does the support plane of the cached direction d still separate the
// pair?  og_direction_separates of the oracle: the support vertices are picked in each body's LOCAL space (d turned back
// by the conjugate rotation, last maximum under the total order), only the two winners go to world space.
__device__ __forceinline__ bool cached_direction_separates(const BodyArrays &b, const PolytopeTables &t, const double *__restrict__ frames,
                                                           const uint32_t *__restrict__ pairs, uint32_t p, Vec3 d)
{
    const uint32_t ia = pairs[2 * (size_t)p], ib = pairs[2 * (size_t)p + 1];
    const Frame fa = load_record_p1(frames, ia), fb = load_record_p1(frames, ib);
    const ShapeDesc da = t.desc[b.shape_id[ia]], db = t.desc[b.shape_id[ib]];
    if (da.n_verts == 0 || db.n_verts == 0)
        return false;
    auto support = [&](const ShapeDesc &ds, Vec3 dir) { // support_index: last maximum under the total order
        Vec3 best{0.0, 0.0, 0.0};
        long long best_key = 0;
        for (uint32_t k = 0; k < ds.n_verts; ++k) {
            const double *v = t.verts + 3 * (size_t)(ds.vert0 + k);
            const Vec3 x{v[0], v[1], v[2]};
            const long long key = total_key(dot(x, dir));
            if (k == 0 || best_key <= key) {
                best_key = key;
                best = x;
            }
        }
        return best;
    };
    const Vec3 a = fa * support(da, conjugate(fa.rotation) * d), bb = fb * support(db, conjugate(fb.rotation) * (-d));
    return !(dot(a - bb, d) > 0.0);
}

// The face query of ONE cached face (SatScratch::axis_cache) by one thread: does face `face` of body X (0 = A, 1 = B)
// still separate pair p?  The arithmetic is that of sat_pair for this face -- the other body's vertices through its own
// frame into world space and through the inverse of X's frame into X-local space, the LAST maximum of -n . v under the
// total order, n . support - displacement -- so a "yes" here is the "separated" the full SAT would return.
__device__ __forceinline__ bool cached_face_separates(const BodyArrays &b, const PolytopeTables &t, const double *__restrict__ frames,
                                                      const uint32_t *__restrict__ pairs, uint32_t p, uint32_t code)
{
    const uint32_t owner = (code - 1u) & 1u, face = (code - 1u) >> 1;
    const uint32_t ia = pairs[2 * (size_t)p], ib = pairs[2 * (size_t)p + 1];
    const Frame fa = load_record_p1(frames, ia), fb = load_record_p1(frames, ib);
    const ShapeDesc da = t.desc[b.shape_id[ia]], db = t.desc[b.shape_id[ib]];
    const ShapeDesc dx = owner ? db : da, dy = owner ? da : db;
    if (face >= dx.n_faces || dy.n_verts == 0)
        return false;
    const Frame fy = owner ? fa : fb, fx_inv = inverse(owner ? fb : fa);
    const double *pl = t.planes + 4 * (size_t)(dx.face0 + face);
    const Vec3 n{pl[0], pl[1], pl[2]}, dir = -n;
    Vec3 best{0.0, 0.0, 0.0};
    long long best_key = 0;
    for (uint32_t k = 0; k < dy.n_verts; ++k) { // support_last_max over the other body's vertices in X-local space
        const double *v = t.verts + 3 * (size_t)(dy.vert0 + k);
        const Vec3 w = fy * Vec3{v[0], v[1], v[2]};
        const Vec3 local = fx_inv * w; // frames.0.inverse() * (frames.1 * p), :136-137
        const long long key = total_key(dot(local, dir));
        if (k == 0 || best_key <= key) {
            best_key = key;
            best = local;
        }
    }
    return dot(n, best) - pl[3] >= 0.0;
}

// ... and of ONE cached edge axis (code = kAxisCacheEdge | q, q = direction of A x direction of B as sat_pair numbers them):
// the axis, its orientation and the two reaches over the world-space vertices exactly as sat_pair computes them for this q,
// so a distance >= 0 here means the full SAT's best edge distance is >= 0 as well: "no contact" either way.  In a settled
// pile of boxes 27 % of the pairs whose tight spheres overlap are separated by an edge axis and by no face (31 % by a face,
// 42 % touch: scripts/sat_pair_census.py), and without this they ran both face queries and all edge axes in every substep.
__device__ __forceinline__ bool cached_edge_axis_separates(const BodyArrays &b, const PolytopeTables &t, const double *__restrict__ frames,
                                                           const uint32_t *__restrict__ pairs, uint32_t p, uint32_t q)
{
    const uint32_t ia = pairs[2 * (size_t)p], ib = pairs[2 * (size_t)p + 1];
    const Frame fa = load_record_p1(frames, ia), fb = load_record_p1(frames, ib);
    const uint32_t sa = b.shape_id[ia], sb = b.shape_id[ib];
    const ShapeDesc da = t.desc[sa], db = t.desc[sb];
    if (da.n_verts == 0 || db.n_verts == 0 || db.n_dirs == 0)
        return false;
    const uint32_t i = q / db.n_dirs, j = q - i * db.n_dirs;
    if (i >= da.n_dirs)
        return false;
    const double *da_ = t.edge_dirs + 3 * (size_t)(da.dir0 + i), *db_ = t.edge_dirs + 3 * (size_t)(db.dir0 + j);
    Vec3 n = normalized(cross(fa.rotation * Vec3{da_[0], da_[1], da_[2]}, fb.rotation * Vec3{db_[0], db_[1], db_[2]}));
    if (!finite3(n))
        return false;
    const double *cca = t.centroids + 3 * (size_t)sa, *ccb = t.centroids + 3 * (size_t)sb;
    const Vec3 a_to_b = fb * Vec3{ccb[0], ccb[1], ccb[2]} - fa * Vec3{cca[0], cca[1], cca[2]};
    if (dot(n, a_to_b) < 0.0)
        n = -n;
    double reach_a = 0.0, reach_b = 0.0;
    for (uint32_t v = 0; v < da.n_verts; ++v) {
        const double *x = t.verts + 3 * (size_t)(da.vert0 + v);
        const double rr = dot(fa * Vec3{x[0], x[1], x[2]}, n);
        if (v == 0 || rr > reach_a)
            reach_a = rr;
    }
    for (uint32_t v = 0; v < db.n_verts; ++v) {
        const double *x = t.verts + 3 * (size_t)(db.vert0 + v);
        const double rr = dot(fb * Vec3{x[0], x[1], x[2]}, n);
        if (v == 0 || rr < reach_b)
            reach_b = rr;
    }
    return reach_b - reach_a >= 0.0;
}

template <bool CLASSES>
__global__ void __launch_bounds__(kPretestBlock) k_pair_pretest(BodyArrays b, PolytopeTables t, const double *__restrict__ frames,
                                                                const uint32_t *__restrict__ pairs, uint32_t n_pairs,
                                                                uint8_t *__restrict__ codes, uint32_t *__restrict__ survivor_count,
                                                                uint32_t *__restrict__ survivors, const uint16_t *__restrict__ axis_cache,
                                                                const double *__restrict__ gjk_axis_cache)
{
    constexpr uint32_t NC = CLASSES ? 3 : 1; // small-small, large-small, large-large pairs (SatScratch)
    // Inside the slice of a list that this workgroup appends, the survivors are grouped by the path they are EXPECTED to
    // take through the narrowphase -- the one their last verdict (the code byte of the previous substep) stands for: no
    // contact (they leave after the axis tests), a face contact (clipping), an edge contact (closest points of two edges).
    // A wave of the SAT runs the union of its groups' paths with the other groups' lanes idle (PMC on the boxes pile: 3 400
    // VALU instructions per wave of eight pairs at 14 % of the lanes, the SIMDs saturated); with pairs of one kind side by
    // side most waves run one path only.  Only the ORDER of a list changes, and results go to out[p]: the same bits.
    constexpr uint32_t NP = 3, NW = kPretestBlock / 64;
    __shared__ uint32_t wave_base[NC][NP][NW];
    __shared__ uint32_t list_base[NC];
    const uint32_t p = blockIdx.x * kPretestBlock + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    bool keep = false;
    uint32_t cls = 0, path = 0;
    if (p < n_pairs) {
        const uint32_t last = codes[p];
        path = last == 0 ? 0u : ((last >> kPairCodeFeatureShift) == 2u ? 2u : 1u);
        keep = tight_spheres_overlap(b, t, frames, pairs, p);
        if (keep && axis_cache) {
            const uint32_t code = axis_cache[p];
            if (code & kAxisCacheEdge ? cached_edge_axis_separates(b, t, frames, pairs, p, code & (kAxisCacheEdge - 1u))
                                      : (code && cached_face_separates(b, t, frames, pairs, p, code)))
                keep = false; // (the entry stays as it is)
        }
        if (keep && gjk_axis_cache) {
            const double *c = gjk_axis_cache + 3 * (size_t)p;
            const Vec3 d{c[0], c[1], c[2]};
            if (dot(d, d) > 0.0 && cached_direction_separates(b, t, frames, pairs, p, d))
                keep = false; // (the entry stays as it is)
        }
        if (!keep)
            codes[p] = 0;
        else if (CLASSES) {
            const uint32_t ca = t.shape_class[b.shape_id[pairs[2 * (size_t)p]]], cb = t.shape_class[b.shape_id[pairs[2 * (size_t)p + 1]]];
            cls = ca + cb;
        }
    }
    unsigned long long mine = 0;
#pragma unroll
    for (uint32_t c = 0; c < NC; ++c)
#pragma unroll
        for (uint32_t q = 0; q < NP; ++q) {
            const unsigned long long mask = __ballot(keep && cls == c && path == q);
            if (lane == 0)
                wave_base[c][q][wave] = (uint32_t)__popcll(mask);
            if (cls == c && path == q)
                mine = mask;
        }
    __syncthreads();
    if (threadIdx.x < NC) { // exclusive scan of the (path, wave) counts of a list, then one atomic for the whole workgroup and list
        const uint32_t c = threadIdx.x;
        uint32_t run = 0;
        for (uint32_t q = 0; q < NP; ++q)
            for (uint32_t w = 0; w < NW; ++w) {
                const uint32_t n = wave_base[c][q][w];
                wave_base[c][q][w] = run;
                run += n;
            }
        list_base[c] = run ? atomicAdd(survivor_count + c, run) : 0u;
    }
    __syncthreads();
    if (keep) {
        const uint32_t at = list_base[cls] + wave_base[cls][path][wave] + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
        survivors[cls == 0 ? at : (cls == 1 ? n_pairs - 1u - at : n_pairs + at)] = p;
    }
}

// ... and the SAT over the survivors only, so that every group of every wave has a pair that needs it.  The grid is
// sized for all pairs (the count lives on the device); blocks past the survivors leave at once.  Block 0 zeroes the
// counters of the NEXT launch (the two pairs of counters alternate, as in xpbd_gjk.hip).  back_n != 0: the list that grows
// from the back of `survivors` (entry k sits at back_n - 1 - k).
template <uint32_t L, uint32_t V>
__global__ void __launch_bounds__(64, XPBD_SAT_MIN_WAVES_PER_SIMD) k_sat_survivors(BodyArrays b, PolytopeTables t,
                                                                                   const double *__restrict__ frames,
                                                                                   const uint32_t *__restrict__ pairs,
                                                                                   const uint32_t *__restrict__ survivor_count,
                                                                                   uint32_t *__restrict__ next_survivor_counts,
                                                                                   const uint32_t *__restrict__ survivors, uint32_t back_n,
                                                                                   ContactManifold *__restrict__ out, uint16_t *__restrict__ axis_cache,
                                                                                   uint8_t *__restrict__ codes, bool cache_edge_axes)
{
    using Lds = typename SatLds<L, V>::Record;
    __shared__ Lds s_all[SatLds<L, V>::PW];
    const uint32_t group = threadIdx.x / L, lane = threadIdx.x % L;
    const uint32_t k = blockIdx.x * SatLds<L, V>::PW + group;
    const uint32_t n = *survivor_count;
    if (blockIdx.x == 0 && threadIdx.x < kSurvivorCounters)
        next_survivor_counts[threadIdx.x] = 0;
    if (k >= n)
        return;
    sat_pair<L>(s_all[group], b, t, frames, pairs, survivors[back_n ? back_n - 1u - k : k], out, lane, axis_cache, codes, cache_edge_axes);
}

// ---------------------------------------------------------------------------------------------------------------------
// edge_axes_separation AS THE REFERENCE WROTE IT (src/collision.rs:151-197; uncalled there, even by `sat`): every edge of A
// against every edge of B (144 pairs for boxes), axis = normalize(eA x eB) turned away from A's centroid, the pair is
// skipped when A has a vertex further along the axis than the edge's foot, distance of B's support along -axis from the
// plane through the foot; the FIRST maximum in the (edge of A, edge of B) enumeration wins; parallel edges give a NaN axis
// and contribute nothing.  A diagnostic counterpart of that function, checked against the oracle's literal
// o_edge_axes_separation: the contact pipeline's SAT uses the unique-edge-direction test instead (sat_pair above).
// One wave per pair; the edge pairs are strided over the lanes.
struct EdgeLds {
    double world[2][kMaxV][3];
};

__device__ __forceinline__ Vec3 support_last_max_position(const double (*verts)[3], uint32_t n, Vec3 dir)
{
    // Polytope::support (src/geometry.rs:274-281): max_by over the world-space vertices, last maximum under total_cmp
    return support_last_max(verts, n, dir);
}

__global__ void __launch_bounds__(64) k_edge_axes_reference(BodyArrays b, PolytopeTables t, const double *__restrict__ frames,
                                                            const uint32_t *__restrict__ pairs, uint32_t n_pairs, EdgeQuery *__restrict__ out)
{
    __shared__ EdgeLds s;
    const uint32_t p = blockIdx.x, lane = threadIdx.x;
    if (p >= n_pairs)
        return;
    const uint32_t ia = pairs[2 * (size_t)p], ib = pairs[2 * (size_t)p + 1];
    const Frame fa = load_record_p1(frames, ia), fb = load_record_p1(frames, ib);
    const uint32_t sa = b.shape_id[ia], sb = b.shape_id[ib];
    const ShapeDesc da = t.desc[sa], db = t.desc[sb];
    for (uint32_t v = lane; v < da.n_verts + db.n_verts; v += 64) {
        const bool of_b = v >= da.n_verts;
        const uint32_t k = of_b ? v - da.n_verts : v;
        const double *x = t.verts + 3 * (size_t)((of_b ? db.vert0 : da.vert0) + k);
        st3(s.world[of_b ? 1 : 0], k, (of_b ? fb : fa) * Vec3{x[0], x[1], x[2]});
    }
    __syncthreads();
    const double *cca = t.centroids + 3 * (size_t)sa;
    const Vec3 centroid_a = fa * Vec3{cca[0], cca[1], cca[2]};
    double best = -DBL_MAX; // f64::MIN
    uint32_t best_q = kNone;
    const uint32_t total = da.n_edges * db.n_edges;
    for (uint32_t q = lane; q < total; q += 64) {
        const uint32_t ie = q / db.n_edges, je = q - ie * db.n_edges;
        const uint32_t *ea = t.edges + 2 * (size_t)(da.edge0 + ie), *eb = t.edges + 2 * (size_t)(db.edge0 + je);
        const Vec3 foot = ld3(s.world[0], ea[0]);
        const Vec3 e0 = ld3(s.world[0], ea[1]) - foot;
        const Vec3 e1 = ld3(s.world[1], eb[1]) - ld3(s.world[1], eb[0]);
        Vec3 axis = normalized(cross(e0, e1));
        if (dot(axis, foot - centroid_a) < 0.0)
            axis = -axis;
        if (da.n_verts == 0 || db.n_verts == 0)
            continue; // the reference's .unwrap() would panic on a polytope without vertices
        if (dot(support_last_max_position(s.world[0], da.n_verts, axis), axis) > dot(foot, axis))
            continue;
        const Plane plane = plane_from_point_normal(foot, axis);
        const double dist = distance(plane, support_last_max_position(s.world[1], db.n_verts, -axis));
        if (dist > best) { // ascending q on this lane: first maximum; a NaN never wins
            best = dist;
            best_q = q;
        }
    }
    reduce_max_first(best, best_q, 64);
    if (lane == 0) {
        out[p].separation = best;
        out[p].edge_a = best_q == kNone ? kNone : best_q / db.n_edges;
        out[p].edge_b = best_q == kNone ? kNone : best_q % db.n_edges;
    }
}

} // namespace

namespace {
// Lanes per pair and vertex capacity by the largest shape: boxes and tetrahedra (<= 8 vertices and faces) run four
// pairs per wave with 8-vertex records -- eight when no face has more than 4 vertices (a clipped polygon then has at
// most 8, one per lane) and the launch is large; up to 16 vertices (icosahedra) XPBD_SAT_MID_LANES lanes with
// 16-vertex records; anything larger gets a whole wave.
// (XPBD_SAT_WIDE_PAIRS in the environment overrides the pair count from which the narrow groups are taken -- the tests set it
// to 1 so that small random scenes run through the 8- and 4-lane classes too; read at every launch, so a test can set it)
uint32_t wide_pair_count()
{
    const char *e = std::getenv("XPBD_SAT_WIDE_PAIRS");
    return e ? (uint32_t)std::strtoul(e, nullptr, 10) : kWidePairCount;
}

template <class Launch>
void for_shape_maxima(uint32_t max_verts, uint32_t max_faces, uint32_t max_face_verts, uint32_t n_pairs, bool dense, Launch launch)
{
    const uint32_t wide = wide_pair_count();
    // 8 lanes per pair halve the instructions per pair of the clipping half of the SAT (208 -> 148 us on 245 760 box
    // pairs) but lengthen the chain of a wave: only when there are enough pairs to fill the GPU with 8-pair waves.
    // 4 lanes per pair (round 3; the clipper takes two polygon vertices per lane then) halve again what a group computes
    // redundantly on all its lanes -- frames, feature choice, side planes -- and fill the rounds of the face and edge
    // loops: another 13 % off the SAT of a boxes pile and of box stacks; only in DENSE scenes (the caller's judgement: many
    // of the pairs touch) with twice as many pairs, where the longer chain of a 16-pair wave is hidden.
    if (max_verts <= 8 && max_faces <= 8 && max_face_verts <= 4 && dense && n_pairs >= 2 * (size_t)wide)
        launch(std::integral_constant<uint32_t, XPBD_SAT_DENSE_BOX_LANES>{}, std::integral_constant<uint32_t, 8>{});
    else if (max_verts <= 8 && max_faces <= 8 && max_face_verts <= 4 && n_pairs >= wide)
        launch(std::integral_constant<uint32_t, XPBD_SAT_BOX_LANES>{}, std::integral_constant<uint32_t, 8>{});
    else if (max_verts <= 8 && max_faces <= 8)
        launch(std::integral_constant<uint32_t, XPBD_SAT_SMALL_LANES>{}, std::integral_constant<uint32_t, 8>{});
    else if (max_verts <= 16)
        launch(std::integral_constant<uint32_t, XPBD_SAT_MID_LANES>{}, std::integral_constant<uint32_t, 16>{});
    else
        launch(std::integral_constant<uint32_t, 64>{}, std::integral_constant<uint32_t, kMaxV>{});
}

template <class Launch>
void for_shape_class(const PolytopeTables &t, uint32_t n_pairs, bool dense, Launch launch)
{
    for_shape_maxima(t.max_verts, t.max_faces, t.max_face_verts, n_pairs, dense, launch);
}
} // namespace

hipError_t launch_sat_pairs(const BodyArrays &b, const PolytopeTables &t, const double *frames, const uint32_t *pairs,
                            uint32_t n_pairs, Manifold *out, hipStream_t stream)
{
    if (n_pairs)
        for_shape_class(t, n_pairs, false, [&](auto lanes, auto verts) {
            constexpr uint32_t L = decltype(lanes)::value, V = decltype(verts)::value;
            hipLaunchKernelGGL((k_sat_pairs<L, V, false, Manifold>), dim3((n_pairs + 64 / L - 1) / (64 / L)), dim3(64), 0, stream, b, t,
                               frames, pairs, n_pairs, out, nullptr);
        });
    return hipGetLastError();
}

hipError_t launch_edge_axes_reference(const BodyArrays &b, const PolytopeTables &t, const double *frames, const uint32_t *pairs,
                                      uint32_t n_pairs, EdgeQuery *out, hipStream_t stream)
{
    if (n_pairs)
        hipLaunchKernelGGL(k_edge_axes_reference, dim3(n_pairs), dim3(64), 0, stream, b, t, frames, pairs, n_pairs, out);
    return hipGetLastError();
}

hipError_t launch_pair_pretest(const BodyArrays &b, const PolytopeTables &t, const double *frames, const uint32_t *pairs,
                               uint32_t n_pairs, uint8_t *codes, SatScratch &list, uint32_t **count, uint32_t **next_count,
                               hipStream_t stream, bool use_axis_cache, const double *gjk_axis_cache)
{
    *count = list.counters + kSurvivorCounters * (list.calls & 1u);
    *next_count = list.counters + kSurvivorCounters * ((list.calls + 1u) & 1u);
    ++list.calls;
    hipLaunchKernelGGL(k_pair_pretest<false>, dim3((n_pairs + kPretestBlock - 1) / kPretestBlock), dim3(kPretestBlock), 0, stream, b, t, frames,
                       pairs, n_pairs, codes, *count, list.survivors, use_axis_cache ? list.axis_cache : nullptr, gjk_axis_cache);
    return hipGetLastError();
}

hipError_t launch_sat_contacts(const BodyArrays &b, const PolytopeTables &t, const double *frames, const uint32_t *pairs,
                               uint32_t n_pairs, ContactManifold *out, uint8_t *codes, SatScratch *list, hipStream_t stream, bool dense)
{
    if (n_pairs == 0)
        return hipSuccess;
    if (list && t.two_classes) {
        // pre-test pass into one survivor list per pair class, then every class in groups of its own width: small-small
        // pairs in narrow groups, large-large pairs in the widest shape's, large-small pairs in half of that (an
        // icosahedron against a cube has 45 edge axes, against another icosahedron 225: sharing waves, the light pair
        // waits for the heavy one) -- every wave is full of pairs of its own kind
        uint32_t *count = list->counters + kSurvivorCounters * (list->calls & 1u);
        uint32_t *next = list->counters + kSurvivorCounters * ((list->calls + 1u) & 1u);
        ++list->calls;
        hipLaunchKernelGGL(k_pair_pretest<true>, dim3((n_pairs + kPretestBlock - 1) / kPretestBlock), dim3(kPretestBlock), 0, stream, b, t,
                           frames, pairs, n_pairs, codes, count, list->survivors, list->axis_cache, nullptr);
        for_shape_maxima(8, 8, t.small_max_face_verts, n_pairs, dense, [&](auto lanes, auto verts) {
            constexpr uint32_t L = decltype(lanes)::value, V = decltype(verts)::value;
            hipLaunchKernelGGL((k_sat_survivors<L, V>), dim3((n_pairs + 64 / L - 1) / (64 / L)), dim3(64), 0, stream, b, t, frames, pairs, count,
                               next, list->survivors, 0u, out, list->axis_cache, codes, list->cache_edge_axes);
        });
        if (t.max_verts <= 16)
            hipLaunchKernelGGL((k_sat_survivors<XPBD_SAT_LS_LANES, 16>), dim3((n_pairs + 64 / XPBD_SAT_LS_LANES - 1) / (64 / XPBD_SAT_LS_LANES)), dim3(64), 0,
                               stream, b, t, frames, pairs, count + 1, next, list->survivors, n_pairs, out, list->axis_cache, codes, list->cache_edge_axes);
        else
            hipLaunchKernelGGL((k_sat_survivors<32, kMaxV>), dim3((n_pairs + 1) / 2), dim3(64), 0, stream, b, t, frames, pairs, count + 1, next,
                               list->survivors, n_pairs, out, list->axis_cache, codes, list->cache_edge_axes);
        for_shape_class(t, n_pairs, dense, [&](auto lanes, auto verts) {
            constexpr uint32_t L = decltype(lanes)::value, V = decltype(verts)::value;
            hipLaunchKernelGGL((k_sat_survivors<L, V>), dim3((n_pairs + 64 / L - 1) / (64 / L)), dim3(64), 0, stream, b, t, frames, pairs,
                               count + 2, next, list->survivors + n_pairs, 0u, out, list->axis_cache, codes, list->cache_edge_axes);
        });
        return hipGetLastError();
    }
    for_shape_class(t, n_pairs, dense, [&](auto lanes, auto verts) {
        constexpr uint32_t L = decltype(lanes)::value, V = decltype(verts)::value;
        const dim3 grid((n_pairs + 64 / L - 1) / (64 / L));
        if (list) { // pre-test pass, then the SAT over the survivors
            uint32_t *count = nullptr, *next = nullptr;
            (void)launch_pair_pretest(b, t, frames, pairs, n_pairs, codes, *list, &count, &next, stream, true);
            hipLaunchKernelGGL((k_sat_survivors<L, V>), grid, dim3(64), 0, stream, b, t, frames, pairs, count, next, list->survivors, 0u, out,
                               list->axis_cache, codes, list->cache_edge_axes);
        } else {
            hipLaunchKernelGGL((k_sat_pairs<L, V, true, ContactManifold>), grid, dim3(64), 0, stream, b, t, frames, pairs, n_pairs, out, codes);
        }
    });
    return hipGetLastError();
}

} // namespace xpbd
