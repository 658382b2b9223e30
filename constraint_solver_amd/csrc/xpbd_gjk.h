// xpbd_gjk.h -- GJK + EPA narrowphase launcher (extension; parity unpinned).
#pragma once

#include "xpbd_pairs.h"

namespace xpbd {

constexpr uint32_t kMaxGjkIters = 32;
constexpr uint32_t kMaxEpaIters = 48;
constexpr uint32_t kMaxEpaVerts = 52;   // 4 + kMaxEpaIters
constexpr uint32_t kMaxEpaFaces = 128;
constexpr double kEpaTolerance = 1e-10;

// xpbd_gjk_result in include/xpbd.h has this layout (96 bytes).
struct GjkResult {
    int32_t status;            // 0 separated, 1 penetrating, 2 degenerate (no answer: use SAT)
    uint32_t gjk_iterations, epa_iterations, reserved;
    double depth;
    double normal[3];          // unit, from A towards B
    double point_a[3], point_b[3];
};

// One wave per pair, as launch_sat_pairs.  `out` and `manifolds` may each be NULL: manifolds receives the result as
// a one-point Manifold (reference body A, incident body B) for the contact pipeline.
hipError_t launch_gjk_epa_pairs(const BodyArrays &b, const PolytopeTables &t, const double *frames, const uint32_t *pairs,
                                uint32_t n_pairs, GjkResult *out, Manifold *manifolds, hipStream_t stream);

} // namespace xpbd
