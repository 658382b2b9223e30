// xpbd_contacts.h -- body-body contact EXTENSION: broadphase + per-substep contact pipeline.
// Semantics are defined by oracle/xpbd_pairs_oracle.h (op_contacts_step); parity unpinned.
#pragma once

#include "xpbd_pairs.h"

namespace xpbd {

// xpbd_joint, device copy.
struct Joint {
    uint32_t body_a, body_b;
    double anchor_a[3], anchor_b[3];
    double distance;
    double axis_a[3], axis_b[3]; // XPBD_JOINT_HINGE
    uint32_t kind, reserved;
};

// The uniform grid of one broadphase (device, written by k_grid).
struct GridInfo {
    double edge;        // cell edge = 2 * largest bounding radius
    int32_t origin[3];  // dense mode: cell of the box's minimum corner, minus the margin cell
    uint32_t dims[3];   //             cells per axis including the margins
    uint32_t dense;     // 1: bucket key = linear cell index (no two cells share a bucket); 0: hashed cell
};

// Device buffers of the contact pipeline (owned by the world, sized by the host).
struct ContactBuffers {
    // broadphase, per body
    double *centers;        // [3][stride] bounding-sphere centre (frame * centroid)
    double *radius;         // [stride]    r_shape + min(|v| dt, r_shape) + pad
    int32_t *cell;          // [3][stride] grid cell of the centre
    uint32_t *key;          // [stride]    hash bucket of that cell
    GridInfo *grid;         // [1]
    double *grid_partials;  // [workgroups of k_bounds][7] largest radius, min xyz, max xyz of the centres
    // hash table, table_size = power of two
    uint32_t *bucket_start; // [table_size + 1] counts -> exclusive scan
    uint32_t *bucket_cursor;// [table_size]
    uint32_t *items;        // [n] body ids grouped by bucket, ascending inside a bucket
    uint32_t *items_unsorted; // [n] the same in scatter (arrival) order
    uint32_t table_size;
    double *slot_sphere;    // [4][stride] centre xyz and radius of items[s], i.e. in bucket order
    int32_t *slot_cell;     // [3][stride] cell of items[s]
    // neighbour lists (CSR, ascending) and the pair list i < j
    uint32_t *nbr_off;      // [n + 1]
    uint32_t *pair_first;   // [n + 1]
    uint32_t *upper_start;  // [n]  index in nbr of b's first neighbour > b
    uint32_t *nbr;          // [entries]
    uint32_t *nbr_pair;     // [entries] pair index of (min, max)
    uint32_t *pairs;        // [n_pairs][2]
    // per substep
    double *rec;            // [stride][kRecDoubles] BodyRecord of this substep (xpbd_device.hpp): frames before / after
                            //   integrate, pose after the ground contacts, position before integrate
    const double *stat_rec; // [stride][kStatRecDoubles] StatRecord: inverse mass, inverse inertia, centre of mass
    const uint32_t *stat_index; // NULL: stat_rec is indexed by body; else stat_rec[stat_index[body]] (mass properties shared per shape)
    ContactManifold *manifolds; // [n_pairs]
    uint8_t *pair_codes;    // [n_pairs] n_points | feature << 4 of every pair (xpbd_pairs.h)
    unsigned long long *stats; // [2] touching pairs, manifold points (summed over substeps)
    uint32_t *scan_scratch; // block totals of the scans
    // joints: CSR body -> incident joints (ascending joint index); joint_off is NULL when there are none
    const Joint *joints;
    const uint32_t *joint_off;   // [n + 1]
    const uint32_t *joint_list;  // [2 * n_joints]
    double max_depenetration_speed; // 0 = off: xpbd_world_set_max_depenetration_speed
};

// Which bodies a per-body kernel of the pipeline works on.  Default: all of them.  The multi-GPU world (xpbd_multi.cpp) runs
// the BOUNDARY bodies first, with their end-of-substep state exported straight into the halo send buffer, starts the
// exchange, runs the interior bodies meanwhile (`skip` marks boundary and ghost bodies), and finally gives the ghosts their
// owners' state from the gathered buffer.
struct BodySubset {
    const uint32_t *list = nullptr;        // NULL: bodies 0..n (minus `skip`); else the bodies list[0..count)
    uint32_t count = 0;
    const uint8_t *skip = nullptr;         // with list == NULL: bodies with skip[i] != 0 are left out
    double *export_rows = nullptr;         // with list (pair-solve kernels): the end-of-substep state of list[k] -> row k (13 doubles)
    const double *import_buf = nullptr;    // with list (integrate kernel): the state of list[k] comes from row import_rows[k]
    const uint32_t *import_rows = nullptr;
};

// ---- broadphase (once per step call) -------------------------------------------------------------
hipError_t launch_bounds_and_cells(const BodyArrays &b, const PolytopeTables &t, const double *shape_radius,
                                   double dt, double pad, const ContactBuffers &c, hipStream_t stream);
hipError_t launch_build_buckets(const BodyArrays &b, const ContactBuffers &c, hipStream_t stream);
// counts into nbr_off / pair_first (then scanned in place; totals at [n])
hipError_t launch_neighbour_count(const BodyArrays &b, const ContactBuffers &c, hipStream_t stream);
hipError_t launch_neighbour_fill(const BodyArrays &b, const ContactBuffers &c, hipStream_t stream);

// ---- per substep -----------------------------------------------------------------------------------
// integrate + ground contacts (sequential per body): reads b.dyn, writes the body's record (frames, pose') into c.rec
hipError_t launch_integrate_ground(const BodyArrays &b, const ShapeTable &s, double h, const ContactBuffers &c,
                                   uint32_t *last_mask, uint32_t *trace_masks, uint32_t trace_row, hipStream_t stream,
                                   const BodySubset &subset = BodySubset());
// SAT of every neighbour pair on the post-integrate frames
// (list: NULL = pre-test inside the SAT kernel, else the two-pass form of launch_sat_pairs; dense: many of the pairs touch)
hipError_t launch_sat_contact_pairs(const BodyArrays &b, const PolytopeTables &t, const ContactBuffers &c,
                                    uint32_t n_pairs, SatScratch *list, hipStream_t stream, bool dense = false);
// Jacobi pair solve + derive: reads the records, writes all 13 dynamic fields to dyn_out (b.dyn itself is fine: nobody
// reads another body's SoA state here)
hipError_t launch_pair_solve_derive(const BodyArrays &b, double *dyn_out, double h, const ContactBuffers &c,
                                    hipStream_t stream, const BodySubset &subset = BodySubset());

// launch_pair_solve_derive of this substep and launch_integrate_ground of the next one in a single kernel: the next
// substep's records go to next_rec (a second set: the records in `c` are still being read by the other bodies); the SoA
// state is NOT written (the records carry everything from substep to substep; the step's last launch_pair_solve_derive
// writes it).
hipError_t launch_pair_solve_integrate_ground(const BodyArrays &b, const ShapeTable &s, double h, const ContactBuffers &c,
                                              double *next_rec, uint32_t *last_mask, uint32_t *trace_masks, uint32_t trace_row,
                                              hipStream_t stream, const BodySubset &subset = BodySubset());

// Rigid::frame() of all bodies from the SoA state into the post-integrate frame of their records (all the diagnostic
// narrowphase entry points need); and the StatRecords from the SoA static fields (after an upload).
hipError_t launch_body_frames(const BodyArrays &b, double *rec, hipStream_t stream);
hipError_t launch_stat_records(const BodyArrays &b, double *stat_rec, hipStream_t stream);
// ... and body-major, frames[7 * i + f], for the host read-back.
hipError_t launch_body_frames_aos(const BodyArrays &b, double *frames, hipStream_t stream);

// Halo exchange: gather / scatter the 13 dynamic fields of the listed bodies, body-major buffer.
hipError_t launch_export_dynamic(const BodyArrays &b, const uint32_t *indices, uint32_t n, double *buf, hipStream_t stream);
hipError_t launch_import_dynamic(const BodyArrays &b, const uint32_t *indices, const uint32_t *rows, uint32_t n, const double *buf,
                                 hipStream_t stream); // rows: NULL = row k of buf for entry k

// Halo validity: snapshot[3k..3k+2] = position of body indices[k]; *out = max(*out, max_k scale[k] * |position - snapshot|^2)
// (scale == NULL: 1).
hipError_t launch_snapshot_positions(const BodyArrays &b, const uint32_t *indices, uint32_t n, double *snapshot, hipStream_t stream);
hipError_t launch_max_displacement2(const BodyArrays &b, const uint32_t *indices, uint32_t n, const double *snapshot, const double *scale,
                                    double *out, hipStream_t stream);

// Re-planning the multi-GPU world on the device.  keys[k] = grid cell (xpbd_halo_cell_key, same bits) of body indices[k]
// (indices == NULL: body k), *bad = min(*bad, first list index with a non-finite centre);  out[k] = the 38 doubles of
// xpbd_rigid of body indices[k] + its shape id as a double;  new world (AoS + shape ids) from src[s] >= 0: slot of the old
// world's AoS copy, < 0: incoming record -src[s] - 1 (39 doubles each).
hipError_t launch_cell_keys(const BodyArrays &b, const uint32_t *indices, uint32_t n, double edge, int64_t *keys, uint32_t *bad, hipStream_t stream);
hipError_t launch_gather_records(const BodyArrays &b, const uint32_t *indices, uint32_t n, double *out, hipStream_t stream);
hipError_t launch_repack_bodies(const double *old_aos, const uint32_t *old_shape, const int32_t *src, uint32_t n_new, const double *incoming,
                                double *new_aos, uint32_t *new_shape, hipStream_t stream);

// Exclusive scan of data[0..n) in place; data[n] receives the total.  scratch: >= n/1024 + 2 uint32.
hipError_t launch_exclusive_scan(uint32_t *data, uint32_t n, uint32_t *scratch, hipStream_t stream);

} // namespace xpbd
