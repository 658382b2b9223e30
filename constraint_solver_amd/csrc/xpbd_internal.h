// xpbd_internal.h -- shared by the translation units behind the C ABI (not installed).
#pragma once

#include <cstdint>

namespace xpbd {

// Records the thread's last error message (xpbd_last_error) and returns `code`.
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// ---- one frame of a shard of the multi-GPU world (xpbd_multi.cpp), split at the halo exchange ---------------------------
// All device pointers; slots are local body indices of the shard's world.
struct HaloLists {
    const uint32_t *boundary;   // owned bodies that other ranks mirror (ascending)
    uint32_t n_boundary;
    const uint32_t *ghosts;     // local copies of remote bodies (ascending)
    const uint32_t *ghost_rows; // their rows in the gathered buffer
    uint32_t n_ghosts;
    const uint8_t *skip;        // [n] 1 for boundary and ghost bodies: what the interior launch leaves out (NULL: none)
    double *send;               // n_boundary rows of 13 doubles: this shard's contribution to the all-gather
    const double *recv;         // the gathered buffer
};

} // namespace xpbd

struct xpbd_world;

namespace xpbd {
// frame:   halo_frame_begin; substeps x { halo_substep_boundary; <all-gather send -> recv, overlapping:> halo_substep_interior;
//          <wait for the gather> halo_substep_ghosts }
// begin:    the broadphase of the frame and the integrate + ground stage of substep 0 for every local body
// boundary: the narrowphase of substep k, then the pair solve (+ the integrate + ground stage of substep k + 1 unless `last`)
//           of the BOUNDARY bodies, whose end-of-substep state goes straight into `send`
// interior: the same for the owned bodies nobody mirrors
// ghosts:   the ghosts take their owners' end-of-substep state from `recv` (and run their own integrate + ground stage of
//           substep k + 1 unless `last`)
// Same arithmetic per body as xpbd_world_step, so the same bits.
int halo_frame_begin(xpbd_world *w, double dt, double h);
// ... in two halves, so that a host driving several shards enqueues the broadphase of ALL of them before it waits for any:
// enqueue = bounding spheres, buckets, neighbour counts, the totals on their way to pinned host memory (no wait);
// collect = wait for the totals, size the pair buffers, fill the lists, integrate + ground stage of substep 0.
int halo_frame_begin_enqueue(xpbd_world *w, double dt);
int halo_frame_begin_collect(xpbd_world *w, double h);
// The state a frame starts from (13 dynamic fields per body + last contact masks) kept aside on the device / put back:
// a frame whose halos turn out to have been too thin is undone, re-planned and run again (xpbd_multi.cpp).
int frame_snapshot_save(xpbd_world *w);
int frame_snapshot_restore(xpbd_world *w);
int halo_substep_boundary(xpbd_world *w, double h, uint32_t k, bool last, const HaloLists &l);
int halo_substep_interior(xpbd_world *w, double h, uint32_t k, bool last, const HaloLists &l);
int halo_substep_ghosts(xpbd_world *w, double h, uint32_t k, bool last, const HaloLists &l);
// ---- a plan of the multi-GPU world with the bodies staying on the device -----------------------------------------------------
// host_keys[k] = grid cell (xpbd_halo_cell_key: same bits) of the bounding-sphere centre of body dev_slots[k] (a device array;
// NULL: body k); *bad_index = first k whose centre is not finite (UINT32_MAX: none).  Synchronous.
int halo_cell_keys(xpbd_world *w, const uint32_t *dev_slots, uint32_t n, double edge, int64_t *host_keys, uint32_t *bad_index);
// out39[k] = the xpbd_rigid of body host_slots[k] (38 doubles) and its shape id as a double.  Synchronous.
int download_records(xpbd_world *w, const uint32_t *host_slots, uint32_t n, double *out39);
// The world's bodies become: body s = the present body host_src[s] (>= 0) or incoming record -host_src[s] - 1 (39 doubles
// each).  Only the incoming records cross the bus; otherwise as xpbd_world_upload_bodies (joints and neighbour lists dropped).
int repack_bodies(xpbd_world *w, const int32_t *host_src, uint32_t n_new, const double *incoming39, uint32_t n_incoming);
} // namespace xpbd
