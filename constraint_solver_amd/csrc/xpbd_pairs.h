// xpbd_pairs.h -- body-body contact EXTENSION: launchers for the
// wave-per-pair SAT narrowphase in xpbd_pairs.hip.
//
// The reference has no body-body contacts: its `sat` (src/collision.rs:37-121) is an uncalled stub.
// This extension finishes that sketch; conventions kept from the reference are listed next to the
// code.  Parity for it is UNPINNED (own CPU oracle: oracle/xpbd_pairs_oracle.c).
#pragma once

#include <cstdint>
#include <hip/hip_runtime_api.h>

#include "xpbd_kernels.h"

namespace xpbd {

// Per-shape topology tables in device memory (all shapes back to back).
struct ShapeDesc {
    uint32_t vert0, n_verts;   // into verts (shared with ShapeTable::verts)
    uint32_t face0, n_faces;   // into planes / face_start
    uint32_t edge0, n_edges;   // into edges / edge_dir_id
    uint32_t dir0, n_dirs;     // into edge_dirs: unique edge directions (up to sign)
};

struct PolytopeTables {
    const double *verts;         // [total_verts][3]
    const double *planes;        // [total_faces][4]  outward local plane (normal, displacement), Polytope::plane
    const double *centroids;     // [n_shapes][3]
    const ShapeDesc *desc;       // [n_shapes]
    const uint32_t *face_start;  // [total_faces + 1] offsets into face_verts
    const uint32_t *face_verts;  // shape-local vertex indices of every face, back to back
    const uint32_t *edges;       // [total_edges][2] shape-local vertex indices
    const double *radii;         // [n_shapes] max |vertex - centroid|
    const double *edge_dirs;     // [total_dirs][3] shape-local direction v[e.1] - v[e.0] of the first edge with it
    const uint32_t *edge_dir_id; // [total_edges] shape-local index of every edge's direction
    uint32_t n_shapes;
    uint32_t total_verts;        // vertices of all shapes together
    uint32_t max_verts, max_faces, max_face_verts; // over all shapes: the narrowphase launchers pick their sub-wave width by them
    // Mixed worlds: shapes of at most 8 vertices and 8 faces are the SMALL class (0), the others class 1; a pair's class is
    // the larger of its bodies'.  With both classes present the two-pass SAT keeps one survivor list per class and runs the
    // small pairs in narrow groups (8 or 16 lanes, four or eight pairs per wave) instead of the widest shape's 32 or 64.
    const uint8_t *shape_class;    // [n_shapes]
    uint32_t two_classes;          // both classes occur
    uint32_t small_max_face_verts; // over the small shapes
};

constexpr uint32_t kMaxManifoldPoints = 8;
constexpr uint32_t kMaxFaceVerts = 8;      // vertices per face accepted by the clipper
constexpr double kEdgeBias = 1e-6;         // an edge axis must beat both face axes by this much (metres)

// Result of one pair, 16-byte header + 8 x 2 points (xpbd_manifold in include/xpbd.h has this layout).
struct Manifold {
    uint32_t n_points;   // 0 = separated / not touching
    uint32_t feature;    // 0 face of A, 1 face of B, 2 edge-edge
    uint32_t index_a, index_b;
    double separation;
    double p_ref[kMaxManifoldPoints][3];
    double p_inc[kMaxManifoldPoints][3];
};

// pairs[2*p], pairs[2*p+1] are body indices (A, B); 8, 16, 32 or 64 lanes per pair.  frames: the bodies' BodyRecords
// (xpbd_device.hpp), of which the narrowphase reads the post-integrate frame (the first 56 bytes: one cache line).
// The same result as the contact pipeline stores it (the pair solve reads every manifold twice, and on box stacks those
// reads are half of its traffic), in two places:
//  * `codes[p]` -- one byte per pair: n_points | feature << 4.  The per-body solve looks at ALL neighbours of a body and most
//    of them do not touch: it reads the verdicts from this array (a megabyte, cache resident) and opens a manifold only
//    when there are points in it; "no contact" is written here only.
//  * ContactManifold -- the points.  A FACE contact stores its reference plane and the points on the incident body only;
//    the reader projects them onto the plane with the clipper's own expression (Plane::project: p - distance(plane, p) * n),
//    so it gets the reference points bit for bit.  Plane + four points = 128 bytes: the usual manifold is ONE cache line
//    (the public layout above: four; round 1's interleaved 512-byte record: two).  The single contact of an edge pair
//    or of an EPA query without a face normal (feature 2) stores both of its points: point[0] on the incident body B,
//    point[1] on the reference body A.
struct alignas(256) ContactManifold {
    double plane[4];                     // face contact: reference plane (normal, displacement), world space
    double point[kMaxManifoldPoints][3]; // see above
};
static_assert(sizeof(ContactManifold) == 256, "plane + four points = the first 128-byte line");
constexpr uint32_t kPairCodeFeatureShift = 4;
constexpr uint32_t kSurvivorCounters = 4; // per set: three pair classes + one of padding (SatScratch)

// Result of the reference's edge_axes_separation for one pair (xpbd_edge_query in include/xpbd.h has this layout).
struct EdgeQuery {
    double separation;      // f64::MIN (-DBL_MAX) when no edge pair qualified
    uint32_t edge_a, edge_b; // usize::MAX there, 0xFFFFFFFF here, when none
};
// edge_axes_separation literally as in src/collision.rs:151-197 (diagnostic; one wave per pair).
hipError_t launch_edge_axes_reference(const BodyArrays &b, const PolytopeTables &t, const double *frames, const uint32_t *pairs,
                                      uint32_t n_pairs, EdgeQuery *out, hipStream_t stream);

// (The pipeline's statistics -- touching pairs, contact points -- are summed by the pair solve, xpbd_contacts.hip.)

// Device scratch of the two-pass form (pre-test pass + SAT over the survivors).  `counters`: two SETS of
// kSurvivorCounters uint32 (one counter per pair class), zero when idle; launch k appends through set k & 1 and its
// consumer kernels zero set (k + 1) & 1 for the next launch (all launches of one world are stream-ordered).
// `survivors`: 2 * n_pairs uint32.  Mixed worlds sort the pairs into three classes by the SMALL / large class of their
// shapes (small-small, large-small, large-large): class 0 fills the first n_pairs entries from the front, class 1 from
// the back, class 2 the second n_pairs entries from the front.
//
// `axis_cache` (n_pairs uint16, or null): the axis that separated a pair the last time the SAT looked at it --
// 0 = none, 1 + 2 * face + (0: a face of A, 1: of B), or 0x8000 | q for edge axis q (direction of A x direction of B in the
// SAT's numbering; round 3).  A pair of a settled pile that is separated by a face axis in
// one substep is separated by the SAME axis in the next 99 times in 100 (scripts/separated_pair_census.py), and more
// than half of a pile's neighbour pairs are such pairs.  The pre-test pass therefore evaluates the cached face query
// first -- exactly the arithmetic the full SAT would do for that face, so the verdict "separated" is the SAT's own --
// and answers the pair without sending it to the SAT at all; the SAT kernels refresh the entry on every exit.
// The owner zeroes the cache whenever the pair list changes.
struct SatScratch {
    uint32_t *counters;
    uint32_t *survivors;
    uint32_t calls;
    uint16_t *axis_cache;
    bool cache_edge_axes; // the SAT kernels also leave separating EDGE axes in the cache (worth it in dense scenes only: the owner decides per frame)
};

// The contact pipeline answers "no contact" for pairs whose tight bounding spheres are disjoint (the diagnostic entry
// point xpbd_world_narrowphase runs the full query on every pair).  With `list` the pre-test runs as a pass of its own
// and the SAT only over the surviving pairs -- same results, worth it when most pairs fail the test (a scene of loose
// bodies), a few per cent slower when most pass (stacks).
hipError_t launch_sat_pairs(const BodyArrays &b, const PolytopeTables &t, const double *frames, const uint32_t *pairs,
                            uint32_t n_pairs, Manifold *out, hipStream_t stream); // diagnostic: every pair, full query
// The pre-test pass alone: answers the rejected pairs in `out`, appends the others to list.survivors and returns the
// counter this launch appends through and the one the consumer kernel must zero for the next launch.  use_axis_cache:
// also answer the pairs whose cached SAT face axis still separates them (only for a SAT consumer: "separated" is then the
// SAT's own verdict; GJK's is not defined by face axes).  gjk_axis_cache: GjkScratch::axis_cache of a GJK consumer.
hipError_t launch_pair_pretest(const BodyArrays &b, const PolytopeTables &t, const double *frames, const uint32_t *pairs,
                               uint32_t n_pairs, uint8_t *codes, SatScratch &list, uint32_t **count, uint32_t **next_count,
                               hipStream_t stream, bool use_axis_cache = false, const double *gjk_axis_cache = nullptr);
hipError_t launch_sat_contacts(const BodyArrays &b, const PolytopeTables &t, const double *frames, const uint32_t *pairs,
                               uint32_t n_pairs, ContactManifold *out, uint8_t *codes, SatScratch *list,
                               hipStream_t stream, bool dense = false); // contact pipeline: sphere pre-test, `list` = two-pass form;
                                                                        // dense: many of the pairs touch (narrower groups for boxes)

} // namespace xpbd
