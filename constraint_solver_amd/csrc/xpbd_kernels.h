// xpbd_kernels.h -- host-callable launchers for the gfx950 kernels in xpbd_kernels.hip.
#pragma once

#include <cstdint>
#include <hip/hip_runtime_api.h>

#include "../../include/xpbd.h"

namespace xpbd {

// ---- SoA layout of the bodies in HBM --------------------------------------
// Each scalar field of `Rigid` (reference src/rigid.rs:6-50) is its own array
// of `stride` doubles, so lane i of a wave reads element i of every field:
// every load/store of the stepper is one contiguous 512-byte wave access.
//
//   dyn  : 13 fields  position[3] rotation{s,x,y,z} velocity[3] angular_velocity[3]
//   stat : 25 fields  inverse_mass inverse_inertia[9] external_force[3]
//                     internal_force[3] external_torque[3] internal_torque[3]
//                     center_of_mass[3]
// Field f of body i lives at  base[f * stride + i].
constexpr uint32_t kDynFields = 13;
constexpr uint32_t kStatFields = 25;
constexpr uint32_t kRigidDoubles = 38; // sizeof(xpbd_rigid) / 8

enum DynField : uint32_t { D_POS = 0, D_ROT = 3, D_VEL = 7, D_ANG = 10 };
enum StatField : uint32_t {
    S_INV_MASS = 0, S_INV_INERTIA = 1, S_EXT_FORCE = 10, S_INT_FORCE = 13,
    S_EXT_TORQUE = 16, S_INT_TORQUE = 19, S_COM = 22
};

// Body-major records of the contact pipeline (layout and rationale: xpbd_device.hpp), in doubles per body.
constexpr uint32_t kRecDoubles = 24;
constexpr uint32_t kStatRecDoubles = 16;

struct BodyArrays {
    double *dyn;        // kDynFields  * stride doubles
    double *stat;       // kStatFields * stride doubles
    uint32_t *shape_id; // stride
    uint32_t stride;    // >= n, multiple of 64
    uint32_t n;
};

struct ShapeTable {
    const double *verts;      // device, total_verts * 3 doubles
    const uint32_t *offsets;  // device, n_shapes + 1
    uint32_t n_shapes;
    uint32_t total_verts;
};

// Algorithmic HBM traffic of one unfused substep of one body:
// read 13 dyn + 25 stat doubles + 4 B shape id, write 13 dyn doubles.
constexpr uint32_t kBytesPerBodySubstep = (13 + 25) * 8 + 4 + 13 * 8; // 412

// Largest workgroup launch_step accepts (k_step's __launch_bounds__).
constexpr uint32_t kMaxStepBlock = 256;

// Runs `substeps` substeps of solver::step (src/solver.rs:6-16) for every body,
// in one launch.  h = dt / substep_count (src/solver.rs:4) is computed by the caller.
// last_mask[i]      <- contact mask of the final substep (always written).
// trace_masks       <- optional [substeps][stride_trace] masks of every substep,
//                      written starting at row `trace_row0`.
hipError_t launch_step(const BodyArrays &b, const ShapeTable &s, double h, uint32_t substeps,
                       uint32_t *last_mask, uint32_t *trace_masks, uint32_t trace_row0,
                       uint32_t block_size, hipStream_t stream);

// AoS (xpbd_rigid[n], device staging copy) <-> SoA.
hipError_t launch_aos_to_soa(const double *aos, const BodyArrays &b, hipStream_t stream);
hipError_t launch_soa_to_aos(const BodyArrays &b, double *aos, hipStream_t stream);

// Contact list of the last substep from the per-body masks, sorted by body then
// vertex.  block_counts: scratch of ceil(n/256)+1 uint32.  Phase 1 leaves the
// total in block_counts[ceil(n/256)]; phase 2 scatters into `out`.
hipError_t launch_contacts_count(const uint32_t *mask, uint32_t n, uint32_t *block_counts,
                                 hipStream_t stream);
hipError_t launch_contacts_emit(const uint32_t *mask, uint32_t n, const uint32_t *block_counts,
                                xpbd_contact *out, uint32_t cap, hipStream_t stream);

// Device-to-device copy of `bytes` (multiple of 16, < 2^40) with one of the library's streaming kernels:
// variant 0 / 1 one 16-byte element per lane (plain / non-temporal), 2 / 3 grid-stride with four loads in flight.
constexpr uint32_t kCopyVariants = 4;
hipError_t launch_copy16(const void *src, void *dst, size_t bytes, uint32_t variant, hipStream_t stream);
// Diagnostics: the per-substep traffic of the pinned path and nothing else -- kDynFields + kStatFields doubles read, kDynFields
// written per body, in the world's field-major layout (base[f * stride + i]) or tile-major (64 bodies x all fields contiguous).
hipError_t launch_field_streams(const double *in, double *out, size_t bodies, bool tile_major, hipStream_t stream);

// Diagnostics: lane i reads the first read_bytes of record perm(i) (records: a power of two >= 256; record_bytes / read_bytes:
// one of the pairs instantiated in xpbd_kernels.hip) with 16-byte loads and writes one double.
hipError_t launch_gather_records(const void *in, double *out, uint32_t records, uint32_t record_bytes, uint32_t read_bytes, hipStream_t stream);

// Diagnostics: q = a / b, r = sqrt(a), element-wise, all device pointers.
hipError_t launch_selftest_div_sqrt(const double *a, const double *b, double *q, double *r, uint32_t n,
                                    hipStream_t stream);

} // namespace xpbd
