// xpbd_gjk.h -- GJK + EPA narrowphase launcher (extension; parity unpinned).
#pragma once

#include "xpbd_pairs.h"

namespace xpbd {

constexpr uint32_t kMaxGjkIters = 32;
constexpr uint32_t kMaxEpaIters = 48;
constexpr uint32_t kMaxEpaVerts = 52;   // 4 + kMaxEpaIters
constexpr uint32_t kMaxEpaFaces = 128;
constexpr double kEpaTolerance = 1e-10;
constexpr double kEpaCoplanar = 1e-12;  // a face this close to the new point's plane counts as seeing it (re-triangulated with it)

// xpbd_gjk_result in include/xpbd.h has this layout (96 bytes).
struct GjkResult {
    int32_t status;            // 0 separated, 1 penetrating, 2 degenerate (no answer: use SAT)
    uint32_t gjk_iterations, epa_iterations, reserved;
    double depth;
    double normal[3];          // unit, from A towards B
    double point_a[3], point_b[3];
};

constexpr uint32_t kEpaBlocks = 8192;   // grid of the EPA kernel (one wave per block, grid-stride over the hits)

// Device scratch of the two-kernel narrowphase.  The hit list is SEGMENTED: workgroup g of the GJK kernel appends to segment
// g % kHitSegments through that segment's own counter, each counter on a cache line of its own -- one list behind one
// counter meant one same-address atomic per wave, which serialise at ~10 ns each: 30 000 waves = the whole 300 us of the
// kernel, whatever it computed.  `counters`: two sets (launch k appends through set k & 1, its EPA kernel zeroes set
// (k + 1) & 1 for the next launch; all launches of one world are stream-ordered) of kHitSegments counters,
// kHitCounterStride uint32 apart, zero when idle: gjk_counter_bytes() bytes.  `pairs_scratch`: gjk_scratch_bytes(n_pairs).
constexpr uint32_t kHitSegments = 32;
constexpr uint32_t kHitCounterStride = 32; // uint32: 128 bytes
constexpr size_t gjk_counter_bytes() { return (size_t)2 * kHitSegments * kHitCounterStride * 4; }
// `axis_cache` (3 doubles per pair, or null; needs the pre-test as a pass of its own): the direction whose support plane
// proved a pair separated in its last query, or its last PENETRATION NORMAL; zero = none.  The pre-test pass tries it as a
// separating plane before the pair is sent to GJK again (in a settled pile most separated pairs stay separated by the same
// plane from one substep to the next), and the full query WARM-STARTS its expansion with it: the polytope is seeded with
// the face of the Minkowski difference that has that normal, which makes the boolean GJK unnecessary and lets the
// expansion of two bodies resting face on face end in its first iteration instead of the seventh (semantics:
// og_gjk_epa_cached / seed_polytope of the oracle).  Every full query refreshes it, the owner zeroes it with every new
// pair list.
struct GjkScratch {
    uint32_t *counters;
    void *pairs_scratch;
    uint32_t calls;
    double *axis_cache;
    uint8_t *codes; // with `manifolds`: the pipeline's per-pair code bytes (n_points | feature << 4, xpbd_pairs.h)
};
size_t gjk_scratch_bytes(uint32_t n_pairs);

// Boolean GJK over all pairs (16 or 32 lanes per pair), then EPA over the penetrating ones (one wave per pair).
// `out` and `manifolds` may each be NULL: manifolds (with scratch.codes) receives the pipeline's form of the result -- a
// clipped face contact where the penetration normal is a face normal of one of the bodies, else the single EPA contact
// (reference body A, incident body B).  sphere_pretest / list: the pipeline's pre-test inside the GJK kernel
// (list == NULL) or as a pass of its own with the GJK over the survivors (see launch_sat_contacts); the contact pipeline
// always uses the latter, which is also where the cached separating directions are consulted.
hipError_t launch_gjk_epa_pairs(const BodyArrays &b, const PolytopeTables &t, const double *frames, const uint32_t *pairs,
                                uint32_t n_pairs, GjkResult *out, ContactManifold *manifolds, GjkScratch &scratch, bool sphere_pretest,
                                SatScratch *list, hipStream_t stream);

} // namespace xpbd
