// xpbd_world.cpp -- implementation of the C ABI in include/xpbd.h.
//
// Owns the device-side SoA body arrays, the shape tables and the HIP stream of
// one world, and turns each ABI call into kernel launches from xpbd_kernels.hip.
// There is no CPU fallback: without a usable HIP device every compute entry
// point fails with XPBD_E_NO_DEVICE / XPBD_E_HIP.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "xpbd_internal.h"
#include "xpbd_kernels.h"
#include "xpbd_math.hpp"
#include "xpbd_contacts.h"
#include "xpbd_gjk.h"
#include "xpbd_pairs.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define XPBD_HIP_TRY(expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? XPBD_E_OOM : XPBD_E_HIP, "%s failed: %s",      \
                        #expr, hipGetErrorString(e_));                                             \
    } while (0)

uint32_t round_up(uint32_t v, uint32_t to) { return (v + to - 1) / to * to; }

} // namespace

namespace xpbd {
int set_error(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
} // namespace xpbd

namespace {

// A device allocation that only ever grows.  The first request is served exactly (most buffers are sized by the body
// count and never change); a buffer that has to GROW takes a quarter more than asked: the pair, neighbour and manifold
// buffers follow the pair count of the frame, which creeps up frame after frame while a pile settles, and every
// hipFree + hipMalloc of a block of ~100 MB stalls the stream for several hundred microseconds.
struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;

    hipError_t reserve(size_t want)
    {
        if (want <= bytes)
            return hipSuccess;
        if (ptr) {
            want += want / 4;
            hipError_t e = hipFree(ptr);
            ptr = nullptr;
            bytes = 0;
            if (e != hipSuccess)
                return e;
        }
        hipError_t e = hipMalloc(&ptr, want);
        if (e == hipSuccess)
            bytes = want;
        return e;
    }
    void release()
    {
        if (ptr)
            (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
    template <class T> T *as() const { return static_cast<T *>(ptr); }
};

} // namespace

struct xpbd_world {
    int device = 0;
    uint32_t mode = XPBD_MODE_FUSED;
    uint32_t flags = 0;
    uint32_t block_size = 0;

    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;

    // bodies
    uint32_t n = 0;
    uint32_t stride = 0;
    uint32_t max_shape_id = 0; // largest shape id among the uploaded bodies (xpbd_world_set_shapes re-validates against it)
    DeviceBuffer dyn, stat, shape_id, aos_staging, last_mask, trace, block_counts, contacts;
    uint32_t trace_rows = 0; // substeps recorded by the last step()
    bool stepped = false;

    // shapes
    DeviceBuffer shape_verts, shape_offsets;
    uint32_t n_shapes = 0, total_verts = 0;

    // extension: polytope topology for the body-body narrowphase
    DeviceBuffer planes, centroids, shape_desc, face_start, face_verts, edges, pair_buf, manifold_buf;
    DeviceBuffer shape_radii, edge_dirs, edge_dir_id, shape_class;
    uint32_t two_classes = 0, small_max_face_verts = 0;
    bool has_topology = false;
    uint32_t max_verts = 0, max_faces = 0, max_face_verts = 0;
    xpbd::PolytopeTables tables() const
    {
        return xpbd::PolytopeTables{shape_verts.as<double>(), planes.as<double>(), centroids.as<double>(),
                                    shape_desc.as<xpbd::ShapeDesc>(), face_start.as<uint32_t>(),
                                    face_verts.as<uint32_t>(), edges.as<uint32_t>(), shape_radii.as<double>(),
                                    edge_dirs.as<double>(), edge_dir_id.as<uint32_t>(), n_shapes, total_verts, max_verts, max_faces, max_face_verts,
                                    shape_class.as<uint8_t>(), two_classes, small_max_face_verts};
    }

    // extension: contact pipeline (XPBD_MODE_CONTACTS)
    double contact_pad = 0.02;
    double max_depenetration_speed = 0.0; // 0 = off (the reference's solver loop)
    uint32_t narrowphase = XPBD_NARROWPHASE_SAT;
    DeviceBuffer dyn_alt, cb_centers, cb_radius, cb_cell, cb_key, cb_maxr, cb_bucket_start, cb_bucket_cursor, cb_items,
        cb_nbr_off, cb_pair_first, cb_upper_start, cb_nbr, cb_nbr_pair, cb_pairs, cb_rec, cb_stat_rec,
        cb_manifolds, cb_stats, cb_scan, cb_slot_sphere, cb_slot_cell;
    // second set of per-substep records for the fused "end of substep k + start of substep k + 1" kernel (step_contacts)
    DeviceBuffer cb_rec_b;
    bool stat_rec_valid = false; // the StatRecords mirror the SoA static fields of the bodies uploaded last
    // Mass properties shared per shape: when every body of a shape has bit-identical inverse mass, inverse inertia and
    // centre of mass (checked on the host at upload -- the usual case: bodies are instances of a few shapes), the contact
    // kernels read them from a table of n_shapes StatRecords indexed by shape id (cache resident) instead of gathering a
    // 128-byte record per body and per neighbour.  Same values, so the same bits.
    bool stat_shared = false;
    std::vector<double> stat_shape_host; // [n_shapes][kStatRecDoubles]
    std::vector<uint8_t> stat_shape_seen; // [n_shapes] a body of the shape has been uploaded: its row of stat_shape_host is set
    // re-packing the bodies on the device (xpbd::repack_bodies): the new world in AoS, its shape ids, the source map, incoming records
    DeviceBuffer repack_aos, repack_shape, repack_src, repack_incoming, halo_keys, halo_records;
    DeviceBuffer cb_stat_shape;
    DeviceBuffer cb_grid_partials, cb_items_unsorted;
    uint32_t table_size = 0, n_entries = 0, n_pairs = 0;
    bool have_neighbours = false;
    // The broadphase in two halves (build_neighbours_enqueue / _collect): the counting kernels leave the totals the host
    // needs to size the pair buffers in PINNED memory and record an event; only the second half waits for it.  A host that
    // drives several worlds (xpbd_multi.cpp: one per GPU) enqueues all of them before it waits for any.
    struct BroadphaseTotals {
        uint32_t entries, pairs;
        unsigned long long stats[2];
    };
    BroadphaseTotals *bp_totals = nullptr; // hipHostMalloc
    hipEvent_t bp_event = nullptr;
    bool bp_pending = false;
    // state at the start of a frame (xpbd::frame_snapshot_save / _restore): the 13 dynamic fields and the contact masks
    DeviceBuffer frame_snapshot;
    bool frame_snapshot_valid = false, frame_snapshot_stepped = false;
    DeviceBuffer jt_joints, jt_off, jt_list;
    uint32_t n_joints = 0;
    // state history (xpbd_world_history_*): `history_length` slots of history_slot_bytes() in one growing block
    DeviceBuffer history;
    uint32_t history_length = 0;
    std::vector<uint8_t> history_stepped;
    size_t history_slot_bytes() const { return ((size_t)xpbd::kDynFields * stride * 8 + (size_t)stride * 4 + 255) / 256 * 256; }
    // SAT in two passes (pre-test pass + survivor list, xpbd_pairs.h): chosen per frame from the share of touching
    // pairs in the previous frame, read back at the broadphase's synchronisation point
    DeviceBuffer sat_counters, sat_survivors, sat_axis_cache, gjk_axis_cache, cb_pair_codes;
    xpbd::SatScratch sat_scratch{nullptr, nullptr, 0, nullptr, true};
    bool sat_two_pass = false;
    uint32_t sat_schedule = XPBD_SAT_SCHEDULE_AUTO;
    unsigned long long stats_touching_seen = 0, stats_pair_substeps = 0, stats_pair_substeps_seen = 0;
    DeviceBuffer gjk_counters, gjk_pairs_scratch;   // hit list of the two-kernel GJK/EPA narrowphase (xpbd_gjk.h)
    xpbd::GjkScratch gjk_scratch{nullptr, nullptr, 0, nullptr, nullptr};
    // frame_set: which of the two frame sets the substep reads (always 0 outside step_contacts)
    xpbd::ContactBuffers contact_buffers(uint32_t frame_set = 0) const
    {
        xpbd::ContactBuffers c{};
        c.centers = cb_centers.as<double>();
        c.radius = cb_radius.as<double>();
        c.cell = cb_cell.as<int32_t>();
        c.key = cb_key.as<uint32_t>();
        c.grid = cb_maxr.as<xpbd::GridInfo>();
        c.grid_partials = cb_grid_partials.as<double>();
        c.bucket_start = cb_bucket_start.as<uint32_t>();
        c.bucket_cursor = cb_bucket_cursor.as<uint32_t>();
        c.items = cb_items.as<uint32_t>();
        c.items_unsorted = cb_items_unsorted.as<uint32_t>();
        c.table_size = table_size;
        c.slot_sphere = cb_slot_sphere.as<double>();
        c.slot_cell = cb_slot_cell.as<int32_t>();
        c.nbr_off = cb_nbr_off.as<uint32_t>();
        c.pair_first = cb_pair_first.as<uint32_t>();
        c.upper_start = cb_upper_start.as<uint32_t>();
        c.nbr = cb_nbr.as<uint32_t>();
        c.nbr_pair = cb_nbr_pair.as<uint32_t>();
        c.pairs = cb_pairs.as<uint32_t>();
        c.rec = (frame_set ? cb_rec_b : cb_rec).as<double>();
        c.stat_rec = stat_shared ? cb_stat_shape.as<double>() : cb_stat_rec.as<double>();
        c.stat_index = stat_shared ? shape_id.as<uint32_t>() : nullptr;
        c.manifolds = cb_manifolds.as<xpbd::ContactManifold>();
        c.pair_codes = cb_pair_codes.as<uint8_t>();
        c.stats = cb_stats.as<unsigned long long>();
        c.scan_scratch = cb_scan.as<uint32_t>();
        c.joints = n_joints ? jt_joints.as<xpbd::Joint>() : nullptr;
        c.joint_off = n_joints ? jt_off.as<uint32_t>() : nullptr;
        c.joint_list = n_joints ? jt_list.as<uint32_t>() : nullptr;
        c.max_depenetration_speed = max_depenetration_speed;
        return c;
    }

    xpbd::BodyArrays arrays() const
    {
        return xpbd::BodyArrays{dyn.as<double>(), stat.as<double>(), shape_id.as<uint32_t>(), stride, n};
    }
    xpbd::ShapeTable shapes() const
    {
        return xpbd::ShapeTable{shape_verts.as<double>(), shape_offsets.as<uint32_t>(), n_shapes, total_verts};
    }
};

namespace {

constexpr uint32_t kDefaultBlock = 64;

int bind_device(const xpbd_world *w)
{
    XPBD_HIP_TRY(hipSetDevice(w->device));
    return XPBD_OK;
}

uint32_t next_pow2(uint32_t v)
{
    uint32_t p = 1;
    while (p < v)
        p <<= 1;
    return p;
}

// Scratch of launch_gjk_epa_pairs for `n_pairs` pairs (growing it frees the old block, which waits for the device).
int ensure_gjk_scratch(xpbd_world *w, uint32_t n_pairs)
{
    if (!w->gjk_counters.ptr) {
        XPBD_HIP_TRY(w->gjk_counters.reserve(xpbd::gjk_counter_bytes()));
        XPBD_HIP_TRY(hipMemsetAsync(w->gjk_counters.ptr, 0, xpbd::gjk_counter_bytes(), w->stream));
        w->gjk_scratch.calls = 0;
    }
    XPBD_HIP_TRY(w->gjk_pairs_scratch.reserve(xpbd::gjk_scratch_bytes(n_pairs ? n_pairs : 1)));
    w->gjk_scratch.counters = w->gjk_counters.as<uint32_t>();
    w->gjk_scratch.pairs_scratch = w->gjk_pairs_scratch.ptr;
    return XPBD_OK;
}

// Sphere broadphase of the contact pipeline: neighbour lists + pair list for the coming frame, in two halves.
// First half: bounding spheres, buckets and the neighbour COUNT of every body are enqueued, the totals travel to pinned host
// memory behind them, an event marks their arrival.  Nothing here waits for the device.
int build_neighbours_enqueue(xpbd_world *w, double dt)
{
    if (!w->bp_totals) {
        XPBD_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&w->bp_totals), sizeof *w->bp_totals, hipHostMallocDefault));
        XPBD_HIP_TRY(hipEventCreateWithFlags(&w->bp_event, hipEventDisableTiming));
    }
    const uint32_t n = w->n, st = w->stride;
    w->table_size = next_pow2(n < 512 ? 1024 : 2 * n);
    XPBD_HIP_TRY(w->cb_centers.reserve((size_t)3 * st * 8));
    XPBD_HIP_TRY(w->cb_radius.reserve((size_t)st * 8));
    XPBD_HIP_TRY(w->cb_cell.reserve((size_t)3 * st * 4));
    XPBD_HIP_TRY(w->cb_key.reserve((size_t)st * 4));
    XPBD_HIP_TRY(w->cb_maxr.reserve(sizeof(xpbd::GridInfo)));
    XPBD_HIP_TRY(w->cb_grid_partials.reserve(((size_t)st / 256 + 1) * 7 * 8));
    XPBD_HIP_TRY(w->cb_bucket_start.reserve((size_t)(w->table_size + 1) * 4));
    XPBD_HIP_TRY(w->cb_bucket_cursor.reserve((size_t)w->table_size * 4));
    XPBD_HIP_TRY(w->cb_items.reserve((size_t)st * 4));
    XPBD_HIP_TRY(w->cb_items_unsorted.reserve((size_t)st * 4));
    XPBD_HIP_TRY(w->cb_slot_sphere.reserve((size_t)4 * st * 8));
    XPBD_HIP_TRY(w->cb_slot_cell.reserve((size_t)3 * st * 4));
    XPBD_HIP_TRY(w->cb_nbr_off.reserve((size_t)(st + 1) * 4));
    XPBD_HIP_TRY(w->cb_pair_first.reserve((size_t)(st + 1) * 4));
    XPBD_HIP_TRY(w->cb_upper_start.reserve((size_t)st * 4));
    XPBD_HIP_TRY(w->cb_rec.reserve((size_t)xpbd::kRecDoubles * st * 8));
    XPBD_HIP_TRY(w->cb_rec_b.reserve((size_t)xpbd::kRecDoubles * st * 8));
    if (w->stat_shared) {
        if (!w->stat_rec_valid) {
            XPBD_HIP_TRY(w->cb_stat_shape.reserve(w->stat_shape_host.size() * 8));
            XPBD_HIP_TRY(hipMemcpyAsync(w->cb_stat_shape.ptr, w->stat_shape_host.data(), w->stat_shape_host.size() * 8,
                                        hipMemcpyHostToDevice, w->stream));
            XPBD_HIP_TRY(hipStreamSynchronize(w->stream)); // (pageable source)
            w->stat_rec_valid = true;
        }
    } else {
        if (w->cb_stat_rec.bytes < (size_t)xpbd::kStatRecDoubles * st * 8)
            w->stat_rec_valid = false;
        XPBD_HIP_TRY(w->cb_stat_rec.reserve((size_t)xpbd::kStatRecDoubles * st * 8));
        if (!w->stat_rec_valid) {
            XPBD_HIP_TRY(xpbd::launch_stat_records(w->arrays(), w->cb_stat_rec.as<double>(), w->stream));
            w->stat_rec_valid = true;
        }
    }
    XPBD_HIP_TRY(w->cb_scan.reserve(((size_t)(w->table_size > st ? w->table_size : st) / 1024 + 8) * 4));
    if (!w->cb_stats.ptr) {
        XPBD_HIP_TRY(w->cb_stats.reserve(16));
        XPBD_HIP_TRY(hipMemsetAsync(w->cb_stats.ptr, 0, 16, w->stream));
        w->stats_touching_seen = 0;
        w->stats_pair_substeps_seen = w->stats_pair_substeps;
    }
    const xpbd::BodyArrays b = w->arrays();
    xpbd::ContactBuffers c = w->contact_buffers();
    XPBD_HIP_TRY(xpbd::launch_bounds_and_cells(b, w->tables(), w->shape_radii.as<double>(), dt, w->contact_pad, c,
                                               w->stream));
    XPBD_HIP_TRY(xpbd::launch_build_buckets(b, c, w->stream));
    XPBD_HIP_TRY(xpbd::launch_neighbour_count(b, c, w->stream));
    XPBD_HIP_TRY(hipMemcpyAsync(&w->bp_totals->entries, c.nbr_off + n, 4, hipMemcpyDeviceToHost, w->stream));
    XPBD_HIP_TRY(hipMemcpyAsync(&w->bp_totals->pairs, c.pair_first + n, 4, hipMemcpyDeviceToHost, w->stream));
    XPBD_HIP_TRY(hipMemcpyAsync(w->bp_totals->stats, c.stats, 16, hipMemcpyDeviceToHost, w->stream));
    XPBD_HIP_TRY(hipEventRecord(w->bp_event, w->stream));
    w->bp_pending = true;
    w->have_neighbours = false;
    return XPBD_OK;
}

// Second half: waits for the totals (the one host synchronisation of a frame in XPBD_MODE_CONTACTS), sizes the pair
// buffers and fills the neighbour and pair lists.
int build_neighbours_collect(xpbd_world *w)
{
    if (!w->bp_pending)
        return fail(XPBD_E_INVALID, "build_neighbours_collect without build_neighbours_enqueue");
    XPBD_HIP_TRY(hipEventSynchronize(w->bp_event));
    w->bp_pending = false;
    const xpbd::BodyArrays b = w->arrays();
    xpbd::ContactBuffers c = w->contact_buffers();
    const unsigned long long stats_now[2] = {w->bp_totals->stats[0], w->bp_totals->stats[1]};
    w->n_entries = w->bp_totals->entries;
    w->n_pairs = w->bp_totals->pairs;
    {
        // Pre-test as a pass of its own for the coming frame?  Either way the results are the same bits; this only
        // picks the cheaper schedule from how many of the pairs examined since the last broadphase were touching.
        // SAT: the pre-test pass also answers the pairs whose cached face axis still separates them (SatScratch), so it
        // pays unless nearly every pair touches (box stacks: 99 % touching, one pass 5 % faster; a pile of boxes with
        // 40 % touching: two passes 35 % faster).  (GJK + EPA always runs the pre-test as a pass of its own: see
        // narrowphase_contacts.)
        const unsigned long long touching = stats_now[0] - w->stats_touching_seen;
        const unsigned long long examined = w->stats_pair_substeps - w->stats_pair_substeps_seen;
        if (w->sat_schedule != XPBD_SAT_SCHEDULE_AUTO)
            w->sat_two_pass = w->sat_schedule == XPBD_SAT_SCHEDULE_TWO_PASS;
        else if (examined)
            w->sat_two_pass = touching * 5 < examined * 4;
        if (w->sat_schedule == XPBD_SAT_SCHEDULE_AUTO && w->two_classes)
            w->sat_two_pass = true; // small and large shapes: the two-pass form sorts the pairs by class (xpbd_pairs.h)
        // A DENSE scene (>= 15 % of the pairs examined in the last frame touched)?  Then (a) separating EDGE axes go into the
        // axis cache (SatScratch): trying one costs the pre-test a whole edge query for every wave that holds such a pair; it
        // pays where many pairs are close (piles: 35-40 % of the pairs touch, +4 % / +9 %), not where a few are (chains of
        // spaced boxes: 5 % touch, -4 %); and (b) box pairs run in groups of four lanes instead of eight (for_shape_maxima:
        // piles and stacks +7 %, the chains -3 %).  Same bits either way.
        if (examined)
            w->sat_scratch.cache_edge_axes = touching * 100 >= examined * 15;
        w->stats_touching_seen = stats_now[0];
        w->stats_pair_substeps_seen = w->stats_pair_substeps;
    }
    if (!w->sat_counters.ptr) {
        XPBD_HIP_TRY(w->sat_counters.reserve(2 * xpbd::kSurvivorCounters * 4));
        XPBD_HIP_TRY(hipMemsetAsync(w->sat_counters.ptr, 0, 2 * xpbd::kSurvivorCounters * 4, w->stream));
        w->sat_scratch.calls = 0;
    }
    XPBD_HIP_TRY(w->sat_survivors.reserve((size_t)(w->n_pairs ? w->n_pairs : 1) * 2 * 4)); // 2 * n_pairs entries: SatScratch
    // the pair list is new: nothing is known about which axis separates which pair
    XPBD_HIP_TRY(w->sat_axis_cache.reserve((size_t)(w->n_pairs ? w->n_pairs : 1) * 2));
    XPBD_HIP_TRY(hipMemsetAsync(w->sat_axis_cache.ptr, 0, (size_t)(w->n_pairs ? w->n_pairs : 1) * 2, w->stream));
    w->sat_scratch.counters = w->sat_counters.as<uint32_t>();
    w->sat_scratch.survivors = w->sat_survivors.as<uint32_t>();
    w->sat_scratch.axis_cache = w->sat_axis_cache.as<uint16_t>();
    if (w->narrowphase == XPBD_NARROWPHASE_GJK_EPA) {
        XPBD_HIP_TRY(w->gjk_axis_cache.reserve((size_t)(w->n_pairs ? w->n_pairs : 1) * 24));
        XPBD_HIP_TRY(hipMemsetAsync(w->gjk_axis_cache.ptr, 0, (size_t)(w->n_pairs ? w->n_pairs : 1) * 24, w->stream));
    }
    XPBD_HIP_TRY(w->cb_nbr.reserve((size_t)(w->n_entries ? w->n_entries : 1) * 4));
    XPBD_HIP_TRY(w->cb_nbr_pair.reserve((size_t)(w->n_entries ? w->n_entries : 1) * 4));
    XPBD_HIP_TRY(w->cb_pairs.reserve((size_t)(w->n_pairs ? w->n_pairs : 1) * 8));
    XPBD_HIP_TRY(w->cb_manifolds.reserve((size_t)(w->n_pairs ? w->n_pairs : 1) * sizeof(xpbd::ContactManifold)));
    XPBD_HIP_TRY(w->cb_pair_codes.reserve((size_t)(w->n_pairs ? w->n_pairs : 1)));
    c = w->contact_buffers();
    XPBD_HIP_TRY(xpbd::launch_neighbour_fill(b, c, w->stream));
    w->have_neighbours = true;
    return XPBD_OK;
}

int build_neighbours(xpbd_world *w, double dt)
{
    if (int rc = build_neighbours_enqueue(w, dt))
        return rc;
    return build_neighbours_collect(w);
}

// Narrowphase of the current substep on the post-integrate frames of `c`.
int narrowphase_contacts(xpbd_world *w, const xpbd::BodyArrays &b, const xpbd::ContactBuffers &c)
{
    if (w->narrowphase == XPBD_NARROWPHASE_GJK_EPA) {
        if (int rc = ensure_gjk_scratch(w, w->n_pairs))
            return rc;
        // always with the pre-test as a pass of its own: that pass consults the cached separating directions, which are
        // part of the narrowphase's semantics (og_gjk_epa_cached of the oracle), not a schedule
        w->gjk_scratch.axis_cache = w->gjk_axis_cache.as<double>();
        w->gjk_scratch.codes = c.pair_codes;
        XPBD_HIP_TRY(xpbd::launch_gjk_epa_pairs(b, w->tables(), c.rec, c.pairs, w->n_pairs, nullptr, c.manifolds,
                                                w->gjk_scratch, true, &w->sat_scratch, w->stream));
    } else {
        XPBD_HIP_TRY(xpbd::launch_sat_contact_pairs(b, w->tables(), c, w->n_pairs, w->sat_two_pass ? &w->sat_scratch : nullptr,
                                                    w->stream, w->sat_scratch.cache_edge_axes /* = a dense scene, see below */));
    }
    w->stats_pair_substeps += w->n_pairs;
    return XPBD_OK;
}

// One substep of the contact pipeline (neighbour lists must be current): the form the split API
// (xpbd_world_contacts_substep) exposes, with a seam for the halo exchange after it.
int substep_contacts(xpbd_world *w, double h, uint32_t *trace, uint32_t trace_row)
{
    const xpbd::BodyArrays b = w->arrays();
    const xpbd::ContactBuffers c = w->contact_buffers();
    XPBD_HIP_TRY(xpbd::launch_integrate_ground(b, w->shapes(), h, c, w->last_mask.as<uint32_t>(), trace, trace_row, w->stream));
    if (int rc = narrowphase_contacts(w, b, c))
        return rc;
    XPBD_HIP_TRY(xpbd::launch_pair_solve_derive(b, b.dyn, h, c, w->stream));
    return XPBD_OK;
}

// One xpbd_world_step in XPBD_MODE_CONTACTS (semantics: oracle/xpbd_pairs_oracle.h).  All substeps run here, so the
// pair solve of substep k and the integrate + ground stage of substep k + 1 are one kernel; the body records alternate
// between two sets because the bodies still read each other's records of substep k while those of k + 1 are written.
// Between the substeps the state lives in the records only; the SoA arrays are read by the first kernel and written by
// the last.
int step_contacts(xpbd_world *w, double dt, double h, uint32_t substeps, uint32_t *trace)
{
    if (!w->has_topology)
        return fail(XPBD_E_INVALID, "XPBD_MODE_CONTACTS needs xpbd_world_set_polytopes");
    if (int rc = build_neighbours(w, dt))
        return rc;
    if (substeps == 0)
        return XPBD_OK;
    XPBD_HIP_TRY(xpbd::launch_integrate_ground(w->arrays(), w->shapes(), h, w->contact_buffers(0), w->last_mask.as<uint32_t>(), trace, 0,
                                               w->stream));
    for (uint32_t k = 0; k < substeps; ++k) {
        const xpbd::BodyArrays b = w->arrays();
        const xpbd::ContactBuffers c = w->contact_buffers(k & 1u);
        if (int rc = narrowphase_contacts(w, b, c))
            return rc;
        if (k + 1 < substeps) {
            const xpbd::ContactBuffers next = w->contact_buffers((k + 1u) & 1u);
            XPBD_HIP_TRY(xpbd::launch_pair_solve_integrate_ground(b, w->shapes(), h, c, next.rec, w->last_mask.as<uint32_t>(), trace,
                                                                  k + 1, w->stream));
        } else {
            XPBD_HIP_TRY(xpbd::launch_pair_solve_derive(b, b.dyn, h, c, w->stream));
        }
    }
    return XPBD_OK;
}

} // namespace

// ---- the frame of a multi-GPU shard, split at the halo exchange (xpbd_internal.h) -------------------------------------------
namespace xpbd {

int halo_frame_begin_enqueue(xpbd_world *w, double dt)
{
    if (!w || w->mode != XPBD_MODE_CONTACTS || !w->has_topology)
        return fail(XPBD_E_INVALID, "halo_frame_begin: needs XPBD_MODE_CONTACTS and xpbd_world_set_polytopes");
    if (w->n == 0)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    return build_neighbours_enqueue(w, dt);
}

int halo_frame_begin_collect(xpbd_world *w, double h)
{
    if (!w || w->mode != XPBD_MODE_CONTACTS || !w->has_topology)
        return fail(XPBD_E_INVALID, "halo_frame_begin: needs XPBD_MODE_CONTACTS and xpbd_world_set_polytopes");
    if (w->n == 0)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    if (int rc = build_neighbours_collect(w))
        return rc;
    XPBD_HIP_TRY(launch_integrate_ground(w->arrays(), w->shapes(), h, w->contact_buffers(0), w->last_mask.as<uint32_t>(), nullptr, 0, w->stream));
    w->stepped = true;
    return XPBD_OK;
}

int halo_frame_begin(xpbd_world *w, double dt, double h)
{
    if (int rc = halo_frame_begin_enqueue(w, dt))
        return rc;
    return halo_frame_begin_collect(w, h);
}

// The state a frame starts from -- the 13 dynamic fields of every body and the contact masks of the last substep -- kept
// aside (device to device, on the world's stream) so that a frame whose halos turn out to have been too thin can be undone.
int frame_snapshot_save(xpbd_world *w)
{
    if (!w)
        return fail(XPBD_E_INVALID, "frame_snapshot_save: NULL world");
    if (w->n == 0)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    const size_t dyn_bytes = (size_t)kDynFields * w->stride * 8, mask_bytes = (size_t)w->stride * 4;
    if (w->frame_snapshot.bytes < dyn_bytes + mask_bytes) {
        XPBD_HIP_TRY(hipStreamSynchronize(w->stream)); // reserve() frees the old block
        XPBD_HIP_TRY(w->frame_snapshot.reserve(dyn_bytes + mask_bytes));
    }
    char *dst = static_cast<char *>(w->frame_snapshot.ptr);
    XPBD_HIP_TRY(hipMemcpyAsync(dst, w->dyn.ptr, dyn_bytes, hipMemcpyDeviceToDevice, w->stream));
    XPBD_HIP_TRY(hipMemcpyAsync(dst + dyn_bytes, w->last_mask.ptr, mask_bytes, hipMemcpyDeviceToDevice, w->stream));
    w->frame_snapshot_valid = true;
    w->frame_snapshot_stepped = w->stepped;
    return XPBD_OK;
}

int frame_snapshot_restore(xpbd_world *w)
{
    if (!w)
        return fail(XPBD_E_INVALID, "frame_snapshot_restore: NULL world");
    if (w->n == 0)
        return XPBD_OK;
    if (!w->frame_snapshot_valid)
        return fail(XPBD_E_INVALID, "frame_snapshot_restore: no snapshot");
    if (int rc = bind_device(w))
        return rc;
    const size_t dyn_bytes = (size_t)kDynFields * w->stride * 8, mask_bytes = (size_t)w->stride * 4;
    const char *src = static_cast<const char *>(w->frame_snapshot.ptr);
    XPBD_HIP_TRY(hipMemcpyAsync(w->dyn.ptr, src, dyn_bytes, hipMemcpyDeviceToDevice, w->stream));
    XPBD_HIP_TRY(hipMemcpyAsync(w->last_mask.ptr, src + dyn_bytes, mask_bytes, hipMemcpyDeviceToDevice, w->stream));
    w->stepped = w->frame_snapshot_stepped;
    w->have_neighbours = false;
    w->bp_pending = false;
    w->trace_rows = 0;
    return XPBD_OK;
}

namespace {
// pair solve (+ next substep's integrate + ground unless `last`) of the bodies of `subset`
int halo_pair_solve(xpbd_world *w, double h, uint32_t k, bool last, const BodySubset &subset)
{
    const BodyArrays b = w->arrays();
    const ContactBuffers c = w->contact_buffers(k & 1u);
    if (last)
        XPBD_HIP_TRY(launch_pair_solve_derive(b, b.dyn, h, c, w->stream, subset));
    else
        XPBD_HIP_TRY(launch_pair_solve_integrate_ground(b, w->shapes(), h, c, w->contact_buffers((k + 1u) & 1u).rec, w->last_mask.as<uint32_t>(),
                                                        nullptr, k + 1, w->stream, subset));
    return XPBD_OK;
}
} // namespace

int halo_substep_boundary(xpbd_world *w, double h, uint32_t k, bool last, const HaloLists &l)
{
    if (w->n == 0)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    if (int rc = narrowphase_contacts(w, w->arrays(), w->contact_buffers(k & 1u)))
        return rc;
    BodySubset subset;
    subset.list = l.boundary;
    subset.count = l.n_boundary;
    subset.export_rows = l.send;
    return halo_pair_solve(w, h, k, last, subset);
}

int halo_substep_interior(xpbd_world *w, double h, uint32_t k, bool last, const HaloLists &l)
{
    if (w->n == 0)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    BodySubset subset;
    subset.skip = l.skip;
    return halo_pair_solve(w, h, k, last, subset);
}

int halo_substep_ghosts(xpbd_world *w, double h, uint32_t k, bool last, const HaloLists &l)
{
    if (w->n == 0 || l.n_ghosts == 0)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    const BodyArrays b = w->arrays();
    if (last) {
        XPBD_HIP_TRY(launch_import_dynamic(b, l.ghosts, l.ghost_rows, l.n_ghosts, l.recv, w->stream));
        return XPBD_OK;
    }
    BodySubset subset;
    subset.list = l.ghosts;
    subset.count = l.n_ghosts;
    subset.import_buf = l.recv;
    subset.import_rows = l.ghost_rows;
    XPBD_HIP_TRY(launch_integrate_ground(b, w->shapes(), h, w->contact_buffers((k + 1u) & 1u), w->last_mask.as<uint32_t>(), nullptr, k + 1, w->stream,
                                         subset));
    return XPBD_OK;
}

} // namespace xpbd

namespace {
// Do all bodies of a shape share their mass properties bit for bit?  (see stat_shared)  One more body of shape `sid`.
void absorb_stat_record(xpbd_world *w, const xpbd_rigid &body, uint32_t sid)
{
    double v[xpbd::kStatRecDoubles] = {};
    v[0] = body.inverse_mass;
    std::memcpy(v + 1, body.inverse_inertia, 9 * sizeof(double));
    std::memcpy(v + 10, body.center_of_mass, 3 * sizeof(double));
    double *slot = w->stat_shape_host.data() + (size_t)sid * xpbd::kStatRecDoubles;
    if (!w->stat_shape_seen[sid]) {
        std::memcpy(slot, v, sizeof v);
        w->stat_shape_seen[sid] = 1;
    } else if (std::memcmp(slot, v, sizeof v) != 0) {
        w->stat_shared = false;
    }
}
} // namespace

// ---- re-planning a shard of the multi-GPU world on the device (xpbd_multi.cpp) ---------------------------------------------
namespace xpbd {

int halo_cell_keys(xpbd_world *w, const uint32_t *dev_slots, uint32_t n, double edge, int64_t *host_keys, uint32_t *bad_index)
{
    *bad_index = UINT32_MAX;
    if (n == 0)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream)); // reserve() may free a block
    XPBD_HIP_TRY(w->halo_keys.reserve((size_t)n * 8 + 8));
    uint32_t *bad = reinterpret_cast<uint32_t *>(w->halo_keys.as<int64_t>() + n);
    XPBD_HIP_TRY(hipMemsetAsync(bad, 0xFF, 4, w->stream));
    XPBD_HIP_TRY(launch_cell_keys(w->arrays(), dev_slots, n, edge, w->halo_keys.as<int64_t>(), bad, w->stream));
    XPBD_HIP_TRY(hipMemcpyAsync(host_keys, w->halo_keys.ptr, (size_t)n * 8, hipMemcpyDeviceToHost, w->stream));
    XPBD_HIP_TRY(hipMemcpyAsync(bad_index, bad, 4, hipMemcpyDeviceToHost, w->stream));
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    return XPBD_OK;
}

int download_records(xpbd_world *w, const uint32_t *host_slots, uint32_t n, double *out39)
{
    if (n == 0)
        return XPBD_OK;
    for (uint32_t k = 0; k < n; ++k)
        if (host_slots[k] >= w->n)
            return fail(XPBD_E_INVALID, "xpbd::download_records: slot %u of a world of %u bodies", host_slots[k], w->n);
    if (int rc = bind_device(w))
        return rc;
    constexpr size_t rec = (size_t)(kRigidDoubles + 1) * 8;
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    XPBD_HIP_TRY(w->halo_records.reserve((size_t)n * rec));
    XPBD_HIP_TRY(w->repack_src.reserve((size_t)n * 4));
    XPBD_HIP_TRY(hipMemcpyAsync(w->repack_src.ptr, host_slots, (size_t)n * 4, hipMemcpyHostToDevice, w->stream));
    XPBD_HIP_TRY(launch_gather_records(w->arrays(), w->repack_src.as<uint32_t>(), n, w->halo_records.as<double>(), w->stream));
    XPBD_HIP_TRY(hipMemcpyAsync(out39, w->halo_records.ptr, (size_t)n * rec, hipMemcpyDeviceToHost, w->stream));
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    return XPBD_OK;
}

// The world's bodies become: body s = the present body src[s] (src[s] >= 0) or incoming record -src[s] - 1 (39 doubles: an
// xpbd_rigid and its shape id).  Everything stays on the device but the incoming records; otherwise as xpbd_world_upload_bodies
// (joints, history, contact masks and the neighbour lists are dropped).
int repack_bodies(xpbd_world *w, const int32_t *host_src, uint32_t n_new, const double *incoming39, uint32_t n_incoming)
{
    if (!w || (n_new && !host_src) || (n_incoming && !incoming39))
        return fail(XPBD_E_INVALID, "xpbd::repack_bodies: NULL argument");
    constexpr uint32_t rec = kRigidDoubles + 1;
    uint32_t max_shape_id = w->max_shape_id;
    for (uint32_t s = 0; s < n_new; ++s) {
        const int32_t from = host_src[s];
        if (from >= 0 ? (uint32_t)from >= w->n : (uint32_t)(-(from + 1)) >= n_incoming)
            return fail(XPBD_E_INVALID, "xpbd::repack_bodies: body %u comes from %d (%u bodies present, %u incoming)", s, from, w->n, n_incoming);
    }
    for (uint32_t k = 0; k < n_incoming; ++k) {
        const double sid = incoming39[(size_t)k * rec + kRigidDoubles];
        if (!(sid >= 0.0) || sid >= (double)w->n_shapes)
            return fail(XPBD_E_INVALID, "xpbd::repack_bodies: incoming record %u has shape id %g of %u", k, sid, w->n_shapes);
        max_shape_id = std::max(max_shape_id, (uint32_t)sid);
    }
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    // 1. the present bodies in AoS, the new ones gathered from them and from the incoming records
    XPBD_HIP_TRY(w->aos_staging.reserve((size_t)std::max(w->n, 1u) * sizeof(xpbd_rigid)));
    XPBD_HIP_TRY(w->repack_aos.reserve((size_t)std::max(n_new, 1u) * sizeof(xpbd_rigid)));
    XPBD_HIP_TRY(w->repack_shape.reserve((size_t)std::max(n_new, 1u) * 4));
    XPBD_HIP_TRY(w->repack_src.reserve((size_t)std::max(n_new, 1u) * 4));
    XPBD_HIP_TRY(w->repack_incoming.reserve((size_t)std::max(n_incoming, 1u) * rec * 8));
    XPBD_HIP_TRY(launch_soa_to_aos(w->arrays(), w->aos_staging.as<double>(), w->stream));
    if (n_new)
        XPBD_HIP_TRY(hipMemcpyAsync(w->repack_src.ptr, host_src, (size_t)n_new * 4, hipMemcpyHostToDevice, w->stream));
    if (n_incoming)
        XPBD_HIP_TRY(hipMemcpyAsync(w->repack_incoming.ptr, incoming39, (size_t)n_incoming * rec * 8, hipMemcpyHostToDevice, w->stream));
    XPBD_HIP_TRY(launch_repack_bodies(w->aos_staging.as<double>(), w->shape_id.as<uint32_t>(), w->repack_src.as<int32_t>(), n_new,
                                      w->repack_incoming.as<double>(), w->repack_aos.as<double>(), w->repack_shape.as<uint32_t>(), w->stream));
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream)); // the arrays below may move
    // 2. the world takes the new size (as xpbd_world_upload_bodies)
    const uint32_t stride = round_up(n_new ? n_new : 1, 256);
    XPBD_HIP_TRY(w->dyn.reserve((size_t)kDynFields * stride * 8));
    XPBD_HIP_TRY(w->stat.reserve((size_t)kStatFields * stride * 8));
    XPBD_HIP_TRY(w->shape_id.reserve((size_t)stride * 4));
    XPBD_HIP_TRY(w->last_mask.reserve((size_t)stride * 4));
    XPBD_HIP_TRY(w->aos_staging.reserve((size_t)(n_new ? n_new : 1) * sizeof(xpbd_rigid)));
    w->have_neighbours = false;
    w->n_joints = 0;
    w->history_length = 0;
    w->history_stepped.clear();
    w->n = n_new;
    w->stride = stride;
    w->stat_rec_valid = false;
    w->max_shape_id = max_shape_id;
    w->stepped = false;
    w->trace_rows = 0;
    w->frame_snapshot_valid = false;
    w->bp_pending = false;
    // mass properties shared per shape: the bodies that stay kept the property, the incoming ones are checked
    if (w->stat_shape_seen.size() != w->n_shapes) {
        w->stat_shape_host.assign((size_t)w->n_shapes * kStatRecDoubles, 0.0);
        w->stat_shape_seen.assign(w->n_shapes, 0);
        w->stat_shared = false;
    }
    for (uint32_t k = 0; k < n_incoming && w->stat_shared; ++k) {
        xpbd_rigid body;
        std::memcpy(&body, incoming39 + (size_t)k * rec, sizeof body);
        absorb_stat_record(w, body, (uint32_t)incoming39[(size_t)k * rec + kRigidDoubles]);
    }
    if (n_new == 0)
        return XPBD_OK;
    XPBD_HIP_TRY(hipMemsetAsync(w->shape_id.ptr, 0, (size_t)stride * 4, w->stream));
    XPBD_HIP_TRY(hipMemcpyAsync(w->shape_id.ptr, w->repack_shape.ptr, (size_t)n_new * 4, hipMemcpyDeviceToDevice, w->stream));
    XPBD_HIP_TRY(hipMemsetAsync(w->last_mask.ptr, 0, (size_t)stride * 4, w->stream));
    XPBD_HIP_TRY(launch_aos_to_soa(w->repack_aos.as<double>(), w->arrays(), w->stream));
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream)); // the caller's buffers are only borrowed
    return XPBD_OK;
}

} // namespace xpbd

extern "C" {

uint32_t xpbd_abi_version(void) { return XPBD_ABI_VERSION; }

const char *xpbd_last_error(void) { return g_last_error.c_str(); }

void xpbd_config_default(xpbd_config *cfg)
{
    if (!cfg)
        return;
    std::memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = sizeof *cfg;
    cfg->device = 0;
    cfg->mode = XPBD_MODE_FUSED;
}

int xpbd_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess)
        return fail(XPBD_E_NO_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return n;
}

int xpbd_world_create(xpbd_world **out, const xpbd_config *cfg)
{
    if (!out)
        return fail(XPBD_E_INVALID, "xpbd_world_create: out is NULL");
    *out = nullptr;
    xpbd_config c;
    xpbd_config_default(&c);
    if (cfg) {
        if (cfg->struct_size != sizeof(xpbd_config))
            return fail(XPBD_E_INVALID, "xpbd_world_create: struct_size %u != %zu", cfg->struct_size,
                        sizeof(xpbd_config));
        c = *cfg;
    }
    if (c.mode != XPBD_MODE_FUSED && c.mode != XPBD_MODE_PER_SUBSTEP && c.mode != XPBD_MODE_CONTACTS)
        return fail(XPBD_E_INVALID, "xpbd_world_create: unknown mode %u", c.mode);
    if (c.flags & ~XPBD_FLAG_TRACE_CONTACTS)
        return fail(XPBD_E_INVALID, "xpbd_world_create: unknown flags 0x%x", c.flags);
    // k_step is compiled with __launch_bounds__(xpbd::kMaxStepBlock): a larger workgroup is a launch failure
    if (c.block_size != 0 && (c.block_size % 64 != 0 || c.block_size > xpbd::kMaxStepBlock))
        return fail(XPBD_E_INVALID, "xpbd_world_create: block_size %u must be a multiple of 64, <= %u",
                    c.block_size, xpbd::kMaxStepBlock);
    if (c.reserved[0] || c.reserved[1] || c.reserved[2])
        return fail(XPBD_E_INVALID, "xpbd_world_create: reserved fields must be 0");

    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(XPBD_E_NO_DEVICE, "no HIP device available (%s)",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (c.device < 0 || c.device >= count)
        return fail(XPBD_E_INVALID, "xpbd_world_create: device %d out of range [0,%d)", c.device, count);

    xpbd_world *w = new (std::nothrow) xpbd_world;
    if (!w)
        return fail(XPBD_E_OOM, "xpbd_world_create: host allocation failed");
    w->device = c.device;
    w->mode = c.mode;
    w->flags = c.flags;
    w->block_size = c.block_size ? c.block_size : kDefaultBlock;
    e = hipSetDevice(w->device);
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&w->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete w;
        return fail(XPBD_E_HIP, "xpbd_world_create: stream creation failed: %s", hipGetErrorString(e));
    }
    w->stream = w->own_stream;
    *out = w;
    return XPBD_OK;
}

void xpbd_world_destroy(xpbd_world *w)
{
    if (!w)
        return;
    (void)hipSetDevice(w->device);
    if (w->stream)
        (void)hipStreamSynchronize(w->stream);
    for (DeviceBuffer *b : {&w->dyn, &w->stat, &w->shape_id, &w->aos_staging, &w->last_mask, &w->trace,
                            &w->block_counts, &w->contacts, &w->shape_verts, &w->shape_offsets, &w->planes,
                            &w->centroids, &w->shape_desc, &w->face_start, &w->face_verts, &w->edges, &w->pair_buf,
                            &w->manifold_buf, &w->shape_radii, &w->edge_dirs, &w->edge_dir_id, &w->shape_class, &w->dyn_alt, &w->cb_centers, &w->cb_radius, &w->cb_cell,
                            &w->cb_key, &w->cb_maxr, &w->cb_bucket_start, &w->cb_bucket_cursor, &w->cb_items,
                            &w->cb_nbr_off, &w->cb_pair_first, &w->cb_upper_start, &w->cb_nbr, &w->cb_nbr_pair,
                            &w->cb_pairs, &w->cb_rec, &w->cb_stat_rec, &w->cb_manifolds,
                            &w->cb_stats, &w->cb_scan, &w->jt_joints, &w->jt_off, &w->jt_list, &w->gjk_counters,
                            &w->gjk_pairs_scratch, &w->cb_slot_sphere, &w->cb_slot_cell, &w->history,
                            &w->sat_counters, &w->sat_survivors, &w->sat_axis_cache, &w->gjk_axis_cache, &w->cb_stat_shape, &w->cb_pair_codes, &w->cb_rec_b,
                            &w->cb_grid_partials, &w->cb_items_unsorted})
        b->release();
    for (DeviceBuffer *b : {&w->frame_snapshot, &w->repack_aos, &w->repack_shape, &w->repack_src, &w->repack_incoming, &w->halo_keys, &w->halo_records})
        b->release();
    if (w->bp_totals)
        (void)hipHostFree(w->bp_totals);
    if (w->bp_event)
        (void)hipEventDestroy(w->bp_event);
    if (w->own_stream)
        (void)hipStreamDestroy(w->own_stream);
    delete w;
}

int xpbd_world_set_shapes(xpbd_world *w, const double *verts_xyz, const uint32_t *vert_offsets,
                          uint32_t n_shapes)
{
    if (!w || !verts_xyz || !vert_offsets || n_shapes == 0)
        return fail(XPBD_E_INVALID, "xpbd_world_set_shapes: NULL argument or no shapes");
    if (vert_offsets[0] != 0)
        return fail(XPBD_E_INVALID, "xpbd_world_set_shapes: vert_offsets[0] must be 0");
    for (uint32_t s = 0; s < n_shapes; ++s) {
        if (vert_offsets[s + 1] < vert_offsets[s])
            return fail(XPBD_E_INVALID, "xpbd_world_set_shapes: vert_offsets not monotone at %u", s);
        if (vert_offsets[s + 1] - vert_offsets[s] > XPBD_MAX_SHAPE_VERTS)
            return fail(XPBD_E_INVALID, "xpbd_world_set_shapes: shape %u has %u vertices (max %u)", s,
                        vert_offsets[s + 1] - vert_offsets[s], XPBD_MAX_SHAPE_VERTS);
    }
    const uint32_t total = vert_offsets[n_shapes];
    // The tables are staged into LDS by every block; keep them far below the 160 KiB/CU.
    if ((size_t)total * 24 + (size_t)(n_shapes + 1) * 4 > 48 * 1024)
        return fail(XPBD_E_INVALID, "xpbd_world_set_shapes: shape tables exceed 48 KiB");
    // the resident bodies keep their shape ids: the kernels index the staged table with them unchecked
    if (w->n && n_shapes <= w->max_shape_id)
        return fail(XPBD_E_INVALID, "xpbd_world_set_shapes: %u shapes, but the uploaded bodies use shape id %u (upload bodies "
                                    "again after shrinking the table)", n_shapes, w->max_shape_id);
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    XPBD_HIP_TRY(w->shape_verts.reserve(total ? (size_t)total * 24 : 8));
    XPBD_HIP_TRY(w->shape_offsets.reserve((size_t)(n_shapes + 1) * 4));
    if (total)
        XPBD_HIP_TRY(hipMemcpy(w->shape_verts.ptr, verts_xyz, (size_t)total * 24, hipMemcpyHostToDevice));
    XPBD_HIP_TRY(hipMemcpy(w->shape_offsets.ptr, vert_offsets, (size_t)(n_shapes + 1) * 4, hipMemcpyHostToDevice));
    w->n_shapes = n_shapes;
    w->total_verts = total;
    w->has_topology = false;
    return XPBD_OK;
}

int xpbd_world_set_polytopes(xpbd_world *w, const xpbd_polytope *shapes, uint32_t n_shapes)
{
    if (!w || !shapes || n_shapes == 0)
        return fail(XPBD_E_INVALID, "xpbd_world_set_polytopes: NULL argument or no shapes");
    std::vector<double> verts, planes, centroids, radii, dirs;
    std::vector<uint32_t> vert_offsets{0}, face_start{0}, face_verts, edges, dir_id;
    std::vector<xpbd::ShapeDesc> desc;
    for (uint32_t s = 0; s < n_shapes; ++s) {
        const xpbd_polytope &p = shapes[s];
        if ((p.n_vertices && !p.vertices_xyz) || (p.n_edges && !p.edges) ||
            (p.n_faces && (!p.face_offsets || !p.face_indices)))
            return fail(XPBD_E_INVALID, "xpbd_world_set_polytopes: shape %u has a NULL table", s);
        if (p.n_vertices > XPBD_MAX_SHAPE_VERTS || p.n_faces > 64 || p.n_edges > 4096)
            return fail(XPBD_E_INVALID, "xpbd_world_set_polytopes: shape %u too large (%u vertices, %u faces, %u edges)",
                        s, p.n_vertices, p.n_faces, p.n_edges);
        xpbd::ShapeDesc d{};
        d.vert0 = (uint32_t)(verts.size() / 3);
        d.n_verts = p.n_vertices;
        d.face0 = (uint32_t)(planes.size() / 4);
        d.n_faces = p.n_faces;
        d.edge0 = (uint32_t)(edges.size() / 2);
        d.n_edges = p.n_edges;
        verts.insert(verts.end(), p.vertices_xyz, p.vertices_xyz + 3 * (size_t)p.n_vertices);
        vert_offsets.push_back((uint32_t)(verts.size() / 3));
        auto vertex = [&](uint32_t i) {
            return xpbd::Vec3{p.vertices_xyz[3 * i], p.vertices_xyz[3 * i + 1], p.vertices_xyz[3 * i + 2]};
        };
        for (uint32_t e = 0; e < 2 * p.n_edges; ++e) {
            if (p.edges[e] >= p.n_vertices)
                return fail(XPBD_E_INVALID, "xpbd_world_set_polytopes: shape %u edge vertex out of range", s);
            edges.push_back(p.edges[e]);
        }
        // unique edge directions (up to sign), first edge with a direction represents it
        d.dir0 = (uint32_t)(dirs.size() / 3);
        for (uint32_t e = 0; e < p.n_edges; ++e) {
            const xpbd::Vec3 dv = vertex(p.edges[2 * e + 1]) - vertex(p.edges[2 * e]);
            const uint32_t nd = (uint32_t)(dirs.size() / 3) - d.dir0;
            uint32_t found = nd;
            for (uint32_t k = 0; k < nd; ++k) {
                const double *u = &dirs[3 * (size_t)(d.dir0 + k)];
                const xpbd::Vec3 uv{u[0], u[1], u[2]};
                const xpbd::Vec3 c = xpbd::cross(dv, uv);
                if (xpbd::dot(c, c) <= 1e-12 * (xpbd::dot(dv, dv) * xpbd::dot(uv, uv))) {
                    found = k;
                    break;
                }
            }
            if (found == nd)
                dirs.insert(dirs.end(), {dv.x, dv.y, dv.z});
            dir_id.push_back(found);
        }
        d.n_dirs = (uint32_t)(dirs.size() / 3) - d.dir0;
        const xpbd::Vec3 centroid{p.centroid[0], p.centroid[1], p.centroid[2]};
        for (uint32_t f = 0; f < p.n_faces; ++f) {
            const uint32_t f0 = p.face_offsets[f], f1 = p.face_offsets[f + 1];
            if (f1 < f0 || f1 - f0 < 3 || f1 - f0 > xpbd::kMaxFaceVerts)
                return fail(XPBD_E_INVALID, "xpbd_world_set_polytopes: shape %u face %u needs 3..%u vertices", s, f,
                            xpbd::kMaxFaceVerts);
            for (uint32_t q = f0; q < f1; ++q) {
                if (p.face_indices[q] >= p.n_vertices)
                    return fail(XPBD_E_INVALID, "xpbd_world_set_polytopes: shape %u face vertex out of range", s);
                face_verts.push_back(p.face_indices[q]);
            }
            face_start.push_back((uint32_t)face_verts.size());
            // Polytope::plane (src/geometry.rs:262-271) over Plane::from_points / facing / flip (:16-24, :55-68)
            const xpbd::Vec3 p0 = vertex(p.face_indices[f0]), p1 = vertex(p.face_indices[f0 + 1]),
                             p2 = vertex(p.face_indices[f0 + 2]);
            xpbd::Vec3 n = xpbd::normalized(xpbd::cross(p1 - p0, p2 - p0));
            double disp = xpbd::dot(n, p0);
            const bool facing = xpbd::dot(n, centroid - disp * n) >= 0.0;
            if (facing) {
                n = -n;
                disp = -disp;
            }
            planes.insert(planes.end(), {n.x, n.y, n.z, disp});
        }
        centroids.insert(centroids.end(), {centroid.x, centroid.y, centroid.z});
        double radius = 0.0; // bounding sphere about the centroid
        for (uint32_t k = 0; k < p.n_vertices; ++k) {
            const double dist = xpbd::length(vertex(k) - centroid);
            if (dist > radius)
                radius = dist;
        }
        radii.push_back(radius);
        desc.push_back(d);
    }
    static const double zero3[3] = {0, 0, 0};
    if (int rc = xpbd_world_set_shapes(w, verts.empty() ? zero3 : verts.data(), vert_offsets.data(), n_shapes))
        return rc;
    auto upload = [&](DeviceBuffer &buf, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = buf.reserve(bytes ? bytes : 8);
        if (e == hipSuccess && bytes)
            e = hipMemcpy(buf.ptr, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    XPBD_HIP_TRY(upload(w->planes, planes.data(), planes.size() * 8));
    XPBD_HIP_TRY(upload(w->centroids, centroids.data(), centroids.size() * 8));
    XPBD_HIP_TRY(upload(w->shape_desc, desc.data(), desc.size() * sizeof(xpbd::ShapeDesc)));
    XPBD_HIP_TRY(upload(w->face_start, face_start.data(), face_start.size() * 4));
    XPBD_HIP_TRY(upload(w->face_verts, face_verts.data(), face_verts.size() * 4));
    XPBD_HIP_TRY(upload(w->edges, edges.data(), edges.size() * 4));
    XPBD_HIP_TRY(upload(w->shape_radii, radii.data(), radii.size() * 8));
    XPBD_HIP_TRY(upload(w->edge_dirs, dirs.data(), dirs.size() * 8));
    XPBD_HIP_TRY(upload(w->edge_dir_id, dir_id.data(), dir_id.size() * 4));
    w->has_topology = true;
    uint32_t max_verts = 0, max_faces = 0, max_face_verts = 0, small_max_face_verts = 0, n_small = 0;
    std::vector<uint8_t> shape_class(n_shapes, 0);
    for (uint32_t k = 0; k < n_shapes; ++k) {
        max_verts = shapes[k].n_vertices > max_verts ? shapes[k].n_vertices : max_verts;
        max_faces = shapes[k].n_faces > max_faces ? shapes[k].n_faces : max_faces;
        const bool small = shapes[k].n_vertices <= 8 && shapes[k].n_faces <= 8;
        shape_class[k] = small ? 0 : 1;
        n_small += small;
        for (uint32_t f = 0; f < shapes[k].n_faces; ++f) {
            const uint32_t nfv = shapes[k].face_offsets[f + 1] - shapes[k].face_offsets[f];
            max_face_verts = nfv > max_face_verts ? nfv : max_face_verts;
            if (small)
                small_max_face_verts = nfv > small_max_face_verts ? nfv : small_max_face_verts;
        }
    }
    XPBD_HIP_TRY(upload(w->shape_class, shape_class.data(), shape_class.size()));
    w->two_classes = (n_small != 0 && n_small != n_shapes) ? 1u : 0u;
    w->small_max_face_verts = small_max_face_verts;
    w->max_verts = max_verts;
    w->max_faces = max_faces;
    w->max_face_verts = max_face_verts;
    return XPBD_OK;
}

int xpbd_world_narrowphase(xpbd_world *w, const uint32_t *pairs, uint32_t n_pairs, xpbd_manifold *out)
{
    static_assert(sizeof(xpbd_manifold) == sizeof(xpbd::Manifold), "xpbd_manifold must mirror xpbd::Manifold");
    if (!w || (n_pairs && (!pairs || !out)))
        return fail(XPBD_E_INVALID, "xpbd_world_narrowphase: NULL argument");
    if (!w->has_topology)
        return fail(XPBD_E_INVALID, "xpbd_world_narrowphase: call xpbd_world_set_polytopes first");
    for (uint32_t k = 0; k < 2 * n_pairs; ++k)
        if (pairs[k] >= w->n)
            return fail(XPBD_E_INVALID, "xpbd_world_narrowphase: pair %u names body %u of %u", k / 2, pairs[k], w->n);
    if (n_pairs == 0)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    XPBD_HIP_TRY(w->pair_buf.reserve((size_t)n_pairs * 8));
    XPBD_HIP_TRY(w->manifold_buf.reserve((size_t)n_pairs * sizeof(xpbd::Manifold)));
    XPBD_HIP_TRY(hipMemcpyAsync(w->pair_buf.ptr, pairs, (size_t)n_pairs * 8, hipMemcpyHostToDevice, w->stream));
    XPBD_HIP_TRY(hipMemsetAsync(w->manifold_buf.ptr, 0, (size_t)n_pairs * sizeof(xpbd::Manifold), w->stream));
    XPBD_HIP_TRY(w->cb_rec.reserve((size_t)xpbd::kRecDoubles * w->stride * 8));
    XPBD_HIP_TRY(xpbd::launch_body_frames(w->arrays(), w->cb_rec.as<double>(), w->stream));
    XPBD_HIP_TRY(xpbd::launch_sat_pairs(w->arrays(), w->tables(), w->cb_rec.as<double>(), w->pair_buf.as<uint32_t>(),
                                        n_pairs, w->manifold_buf.as<xpbd::Manifold>(), w->stream));
    XPBD_HIP_TRY(hipMemcpyAsync(out, w->manifold_buf.ptr, (size_t)n_pairs * sizeof(xpbd::Manifold),
                                hipMemcpyDeviceToHost, w->stream));
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    return XPBD_OK;
}

int xpbd_world_edge_axes_separation(xpbd_world *w, const uint32_t *pairs, uint32_t n_pairs, xpbd_edge_query *out)
{
    static_assert(sizeof(xpbd_edge_query) == sizeof(xpbd::EdgeQuery), "xpbd_edge_query must mirror xpbd::EdgeQuery");
    if (!w || (n_pairs && (!pairs || !out)))
        return fail(XPBD_E_INVALID, "xpbd_world_edge_axes_separation: NULL argument");
    if (!w->has_topology)
        return fail(XPBD_E_INVALID, "xpbd_world_edge_axes_separation: call xpbd_world_set_polytopes first");
    for (uint32_t k = 0; k < 2 * n_pairs; ++k)
        if (pairs[k] >= w->n)
            return fail(XPBD_E_INVALID, "xpbd_world_edge_axes_separation: pair %u names body %u of %u", k / 2, pairs[k], w->n);
    if (n_pairs == 0)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    XPBD_HIP_TRY(w->pair_buf.reserve((size_t)n_pairs * 8));
    XPBD_HIP_TRY(w->manifold_buf.reserve((size_t)n_pairs * sizeof(xpbd::EdgeQuery)));
    XPBD_HIP_TRY(w->cb_rec.reserve((size_t)xpbd::kRecDoubles * w->stride * 8));
    XPBD_HIP_TRY(hipMemcpyAsync(w->pair_buf.ptr, pairs, (size_t)n_pairs * 8, hipMemcpyHostToDevice, w->stream));
    XPBD_HIP_TRY(xpbd::launch_body_frames(w->arrays(), w->cb_rec.as<double>(), w->stream));
    XPBD_HIP_TRY(xpbd::launch_edge_axes_reference(w->arrays(), w->tables(), w->cb_rec.as<double>(), w->pair_buf.as<uint32_t>(), n_pairs,
                                                  w->manifold_buf.as<xpbd::EdgeQuery>(), w->stream));
    XPBD_HIP_TRY(hipMemcpyAsync(out, w->manifold_buf.ptr, (size_t)n_pairs * sizeof(xpbd::EdgeQuery), hipMemcpyDeviceToHost, w->stream));
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    return XPBD_OK;
}

int xpbd_world_upload_bodies(xpbd_world *w, const xpbd_rigid *aos, const uint32_t *shape_id, uint32_t n)
{
    if (!w || (!aos && n))
        return fail(XPBD_E_INVALID, "xpbd_world_upload_bodies: NULL argument");
    if (w->n_shapes == 0)
        return fail(XPBD_E_INVALID, "xpbd_world_upload_bodies: call xpbd_world_set_shapes first");
    uint32_t max_shape_id = 0;
    if (shape_id)
        for (uint32_t i = 0; i < n; ++i) {
            if (shape_id[i] >= w->n_shapes)
                return fail(XPBD_E_INVALID, "xpbd_world_upload_bodies: shape_id[%u] = %u >= n_shapes %u", i,
                            shape_id[i], w->n_shapes);
            max_shape_id = shape_id[i] > max_shape_id ? shape_id[i] : max_shape_id;
        }
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    const uint32_t stride = round_up(n ? n : 1, 256);
    XPBD_HIP_TRY(w->dyn.reserve((size_t)xpbd::kDynFields * stride * 8));
    w->have_neighbours = false;
    w->n_joints = 0; // joints name bodies by index: a new upload invalidates them
    w->history_length = 0;
    w->history_stepped.clear();
    XPBD_HIP_TRY(w->stat.reserve((size_t)xpbd::kStatFields * stride * 8));
    XPBD_HIP_TRY(w->shape_id.reserve((size_t)stride * 4));
    XPBD_HIP_TRY(w->last_mask.reserve((size_t)stride * 4));
    XPBD_HIP_TRY(w->aos_staging.reserve((size_t)(n ? n : 1) * sizeof(xpbd_rigid)));
    w->n = n;
    w->stride = stride;
    w->stat_rec_valid = false;
    w->max_shape_id = max_shape_id;
    w->stat_shape_host.assign((size_t)w->n_shapes * xpbd::kStatRecDoubles, 0.0);
    w->stat_shape_seen.assign(w->n_shapes, 0);
    w->stat_shared = n != 0;
    for (uint32_t i = 0; i < n && w->stat_shared; ++i)
        absorb_stat_record(w, aos[i], shape_id ? shape_id[i] : 0u);
    w->stepped = false;
    w->trace_rows = 0;
    w->frame_snapshot_valid = false;
    w->bp_pending = false;
    if (n == 0)
        return XPBD_OK;
    XPBD_HIP_TRY(hipMemcpyAsync(w->aos_staging.ptr, aos, (size_t)n * sizeof(xpbd_rigid), hipMemcpyHostToDevice,
                                w->stream));
    if (shape_id)
        XPBD_HIP_TRY(hipMemcpyAsync(w->shape_id.ptr, shape_id, (size_t)n * 4, hipMemcpyHostToDevice, w->stream));
    else
        XPBD_HIP_TRY(hipMemsetAsync(w->shape_id.ptr, 0, (size_t)stride * 4, w->stream));
    XPBD_HIP_TRY(hipMemsetAsync(w->last_mask.ptr, 0, (size_t)stride * 4, w->stream));
    XPBD_HIP_TRY(xpbd::launch_aos_to_soa(w->aos_staging.as<double>(), w->arrays(), w->stream));
    // The caller's buffers are only borrowed for the duration of the call.
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    return XPBD_OK;
}

int xpbd_world_download_bodies(xpbd_world *w, xpbd_rigid *aos, uint32_t n)
{
    if (!w || (!aos && n))
        return fail(XPBD_E_INVALID, "xpbd_world_download_bodies: NULL argument");
    if (n != w->n)
        return fail(XPBD_E_INVALID, "xpbd_world_download_bodies: n = %u but the world holds %u bodies", n, w->n);
    if (n == 0)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(xpbd::launch_soa_to_aos(w->arrays(), w->aos_staging.as<double>(), w->stream));
    XPBD_HIP_TRY(hipMemcpyAsync(aos, w->aos_staging.ptr, (size_t)n * sizeof(xpbd_rigid), hipMemcpyDeviceToHost,
                                w->stream));
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    return XPBD_OK;
}

uint32_t xpbd_world_body_count(const xpbd_world *w) { return w ? w->n : 0; }

int xpbd_world_download_frames(xpbd_world *w, double *frames, uint32_t n)
{
    if (!w || (!frames && n))
        return fail(XPBD_E_INVALID, "xpbd_world_download_frames: NULL argument");
    if (n != w->n)
        return fail(XPBD_E_INVALID, "xpbd_world_download_frames: n = %u but the world holds %u bodies", n, w->n);
    if (n == 0)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    // aos_staging holds n * 38 doubles: room for the n * 7 frame doubles
    XPBD_HIP_TRY(xpbd::launch_body_frames_aos(w->arrays(), w->aos_staging.as<double>(), w->stream));
    XPBD_HIP_TRY(hipMemcpyAsync(frames, w->aos_staging.ptr, (size_t)n * 7 * 8, hipMemcpyDeviceToHost, w->stream));
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    return XPBD_OK;
}

int xpbd_world_step(xpbd_world *w, double dt, uint32_t substeps)
{
    if (!w)
        return fail(XPBD_E_INVALID, "xpbd_world_step: NULL world");
    if (substeps == 0) // the reference divides by zero and runs no substep; treat as an argument error
        return fail(XPBD_E_INVALID, "xpbd_world_step: substeps must be > 0");
    if (w->n_shapes == 0)
        return fail(XPBD_E_INVALID, "xpbd_world_step: no shapes set");
    if (w->n == 0)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    const double h = dt / (double)substeps; // src/solver.rs:4
    uint32_t *trace = nullptr;
    if (w->flags & XPBD_FLAG_TRACE_CONTACTS) {
        XPBD_HIP_TRY(hipStreamSynchronize(w->stream)); // reserve() may free the previous buffer
        XPBD_HIP_TRY(w->trace.reserve((size_t)substeps * w->stride * 4));
        trace = w->trace.as<uint32_t>();
        w->trace_rows = substeps;
    }
    if (w->mode == XPBD_MODE_CONTACTS) {
        if (int rc = step_contacts(w, dt, h, substeps, trace))
            return rc;
    } else if (w->mode == XPBD_MODE_FUSED) {
        XPBD_HIP_TRY(xpbd::launch_step(w->arrays(), w->shapes(), h, substeps, w->last_mask.as<uint32_t>(), trace, 0,
                                       w->block_size, w->stream));
    } else {
        for (uint32_t k = 0; k < substeps; ++k)
            XPBD_HIP_TRY(xpbd::launch_step(w->arrays(), w->shapes(), h, 1, w->last_mask.as<uint32_t>(), trace, k,
                                           w->block_size, w->stream));
    }
    w->stepped = true;
    return XPBD_OK;
}

int xpbd_world_synchronize(xpbd_world *w)
{
    if (!w)
        return fail(XPBD_E_INVALID, "xpbd_world_synchronize: NULL world");
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    return XPBD_OK;
}

int xpbd_world_download_contacts(xpbd_world *w, xpbd_contact *out, uint32_t cap, uint32_t *n_out)
{
    if (!w || !n_out || (!out && cap))
        return fail(XPBD_E_INVALID, "xpbd_world_download_contacts: NULL argument");
    *n_out = 0;
    if (w->n == 0 || !w->stepped)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    const uint32_t nb = (w->n + 255) / 256;
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    XPBD_HIP_TRY(w->block_counts.reserve((size_t)(nb + 1) * 4));
    XPBD_HIP_TRY(xpbd::launch_contacts_count(w->last_mask.as<uint32_t>(), w->n, w->block_counts.as<uint32_t>(),
                                             w->stream));
    uint32_t total = 0;
    XPBD_HIP_TRY(hipMemcpyAsync(&total, w->block_counts.as<uint32_t>() + nb, 4, hipMemcpyDeviceToHost, w->stream));
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    *n_out = total;
    const uint32_t take = total < cap ? total : cap;
    if (take) {
        XPBD_HIP_TRY(w->contacts.reserve((size_t)take * sizeof(xpbd_contact)));
        XPBD_HIP_TRY(xpbd::launch_contacts_emit(w->last_mask.as<uint32_t>(), w->n, w->block_counts.as<uint32_t>(),
                                                w->contacts.as<xpbd_contact>(), take, w->stream));
        XPBD_HIP_TRY(hipMemcpyAsync(out, w->contacts.ptr, (size_t)take * sizeof(xpbd_contact), hipMemcpyDeviceToHost,
                                    w->stream));
        XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    }
    if (total > cap)
        return fail(XPBD_E_CAPACITY, "xpbd_world_download_contacts: %u contacts, capacity %u", total, cap);
    return XPBD_OK;
}

int xpbd_world_download_contact_masks(xpbd_world *w, uint32_t *masks, uint32_t substeps, uint32_t n)
{
    if (!w || !masks)
        return fail(XPBD_E_INVALID, "xpbd_world_download_contact_masks: NULL argument");
    if (!(w->flags & XPBD_FLAG_TRACE_CONTACTS))
        return fail(XPBD_E_INVALID, "xpbd_world_download_contact_masks: world created without "
                                    "XPBD_FLAG_TRACE_CONTACTS");
    if (n != w->n || substeps != w->trace_rows)
        return fail(XPBD_E_INVALID, "xpbd_world_download_contact_masks: asked for %u x %u, last step recorded %u x %u",
                    substeps, n, w->trace_rows, w->n);
    if (n == 0 || substeps == 0)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(hipMemcpy2DAsync(masks, (size_t)n * 4, w->trace.ptr, (size_t)w->stride * 4, (size_t)n * 4, substeps,
                                  hipMemcpyDeviceToHost, w->stream));
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    return XPBD_OK;
}

int xpbd_world_set_stream(xpbd_world *w, void *hip_stream)
{
    if (!w)
        return fail(XPBD_E_INVALID, "xpbd_world_set_stream: NULL world");
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    w->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : w->own_stream;
    return XPBD_OK;
}

void *xpbd_world_get_stream(const xpbd_world *w) { return w ? static_cast<void *>(w->stream) : nullptr; }

int xpbd_world_set_mode(xpbd_world *w, uint32_t mode)
{
    if (!w || (mode != XPBD_MODE_FUSED && mode != XPBD_MODE_PER_SUBSTEP && mode != XPBD_MODE_CONTACTS))
        return fail(XPBD_E_INVALID, "xpbd_world_set_mode: bad argument");
    w->mode = mode;
    return XPBD_OK;
}

int xpbd_step_one(xpbd_rigid *rigid, const double *verts_xyz, uint32_t nverts, double dt, uint32_t substeps)
{
    if (!rigid || !verts_xyz)
        return fail(XPBD_E_INVALID, "xpbd_step_one: NULL argument");
    // One cached single-body world per host thread (the reference's step is re-entrant and
    // stateless; so is this apart from the cache).
    struct Cache {
        xpbd_world *w = nullptr;
        ~Cache() { xpbd_world_destroy(w); }
    };
    thread_local Cache cache;
    if (!cache.w) {
        xpbd_config cfg;
        xpbd_config_default(&cfg);
        if (int rc = xpbd_world_create(&cache.w, &cfg))
            return rc;
    }
    const uint32_t offsets[2] = {0, nverts};
    if (int rc = xpbd_world_set_shapes(cache.w, verts_xyz, offsets, 1))
        return rc;
    if (int rc = xpbd_world_upload_bodies(cache.w, rigid, nullptr, 1))
        return rc;
    if (int rc = xpbd_world_step(cache.w, dt, substeps))
        return rc;
    return xpbd_world_download_bodies(cache.w, rigid, 1);
}

int xpbd_world_narrowphase_gjk(xpbd_world *w, const uint32_t *pairs, uint32_t n_pairs, xpbd_gjk_result *out)
{
    static_assert(sizeof(xpbd_gjk_result) == sizeof(xpbd::GjkResult), "xpbd_gjk_result must mirror xpbd::GjkResult");
    if (!w || (n_pairs && (!pairs || !out)))
        return fail(XPBD_E_INVALID, "xpbd_world_narrowphase_gjk: NULL argument");
    if (!w->has_topology)
        return fail(XPBD_E_INVALID, "xpbd_world_narrowphase_gjk: call xpbd_world_set_polytopes first");
    for (uint32_t k = 0; k < 2 * n_pairs; ++k)
        if (pairs[k] >= w->n)
            return fail(XPBD_E_INVALID, "xpbd_world_narrowphase_gjk: pair %u names body %u of %u", k / 2, pairs[k], w->n);
    if (n_pairs == 0)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    XPBD_HIP_TRY(w->pair_buf.reserve((size_t)n_pairs * 8));
    XPBD_HIP_TRY(w->manifold_buf.reserve((size_t)n_pairs * sizeof(xpbd::GjkResult)));
    XPBD_HIP_TRY(w->cb_rec.reserve((size_t)xpbd::kRecDoubles * w->stride * 8));
    XPBD_HIP_TRY(hipMemcpyAsync(w->pair_buf.ptr, pairs, (size_t)n_pairs * 8, hipMemcpyHostToDevice, w->stream));
    XPBD_HIP_TRY(hipMemsetAsync(w->manifold_buf.ptr, 0, (size_t)n_pairs * sizeof(xpbd::GjkResult), w->stream));
    XPBD_HIP_TRY(xpbd::launch_body_frames(w->arrays(), w->cb_rec.as<double>(), w->stream));
    if (int rc = ensure_gjk_scratch(w, n_pairs))
        return rc;
    XPBD_HIP_TRY(xpbd::launch_gjk_epa_pairs(w->arrays(), w->tables(), w->cb_rec.as<double>(),
                                            w->pair_buf.as<uint32_t>(), n_pairs, w->manifold_buf.as<xpbd::GjkResult>(),
                                            nullptr, w->gjk_scratch, false, nullptr, w->stream));
    XPBD_HIP_TRY(hipMemcpyAsync(out, w->manifold_buf.ptr, (size_t)n_pairs * sizeof(xpbd::GjkResult), hipMemcpyDeviceToHost,
                                w->stream));
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    return XPBD_OK;
}

int xpbd_world_set_joints(xpbd_world *w, const xpbd_joint *joints, uint32_t n_joints)
{
    static_assert(sizeof(xpbd_joint) == sizeof(xpbd::Joint), "xpbd_joint must mirror xpbd::Joint");
    if (!w || (n_joints && !joints))
        return fail(XPBD_E_INVALID, "xpbd_world_set_joints: NULL argument");
    if (n_joints && w->mode != XPBD_MODE_CONTACTS) // only the contact pipeline projects joints: do not accept and ignore them
        return fail(XPBD_E_INVALID, "xpbd_world_set_joints: joints need XPBD_MODE_CONTACTS (world is in mode %u)", w->mode);
    std::vector<uint32_t> off((size_t)w->n + 2, 0), list((size_t)2 * n_joints);
    for (uint32_t k = 0; k < n_joints; ++k) {
        const xpbd_joint &j = joints[k];
        if (j.body_a >= w->n || j.body_b >= w->n || j.body_a == j.body_b)
            return fail(XPBD_E_INVALID, "xpbd_world_set_joints: joint %u links bodies %u and %u of %u", k, j.body_a,
                        j.body_b, w->n);
        if (!(j.distance >= 0.0) || !(j.distance <= 1.0e300))
            return fail(XPBD_E_INVALID, "xpbd_world_set_joints: joint %u has distance %g", k, j.distance);
        if (j.kind != XPBD_JOINT_DISTANCE && j.kind != XPBD_JOINT_HINGE)
            return fail(XPBD_E_INVALID, "xpbd_world_set_joints: joint %u has unknown kind %u", k, j.kind);
        if (j.reserved != 0)
            return fail(XPBD_E_INVALID, "xpbd_world_set_joints: joint %u: reserved must be 0", k);
        if (j.kind == XPBD_JOINT_HINGE)
            for (const double *axis : {j.axis_a, j.axis_b}) {
                const double len2 = axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2];
                if (!(len2 > 0.999 && len2 < 1.001))
                    return fail(XPBD_E_INVALID, "xpbd_world_set_joints: hinge %u needs unit axes (|axis|^2 = %g)", k, len2);
            }
        ++off[j.body_a + 1];
        ++off[j.body_b + 1];
    }
    for (uint32_t i = 0; i < w->n; ++i)
        off[i + 1] += off[i];
    std::vector<uint32_t> cursor(off.begin(), off.end() - 1);
    for (uint32_t k = 0; k < n_joints; ++k) { // ascending joint index inside every body's list
        list[cursor[joints[k].body_a]++] = k;
        list[cursor[joints[k].body_b]++] = k;
    }
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    w->n_joints = 0;
    if (n_joints == 0)
        return XPBD_OK;
    XPBD_HIP_TRY(w->jt_joints.reserve((size_t)n_joints * sizeof(xpbd::Joint)));
    XPBD_HIP_TRY(w->jt_off.reserve((size_t)(w->n + 1) * 4));
    XPBD_HIP_TRY(w->jt_list.reserve((size_t)2 * n_joints * 4));
    XPBD_HIP_TRY(hipMemcpy(w->jt_joints.ptr, joints, (size_t)n_joints * sizeof(xpbd::Joint), hipMemcpyHostToDevice));
    XPBD_HIP_TRY(hipMemcpy(w->jt_off.ptr, off.data(), (size_t)(w->n + 1) * 4, hipMemcpyHostToDevice));
    XPBD_HIP_TRY(hipMemcpy(w->jt_list.ptr, list.data(), (size_t)2 * n_joints * 4, hipMemcpyHostToDevice));
    w->n_joints = n_joints;
    return XPBD_OK;
}

int xpbd_world_contacts_begin(xpbd_world *w, double dt)
{
    if (!w)
        return fail(XPBD_E_INVALID, "xpbd_world_contacts_begin: NULL world");
    if (w->mode != XPBD_MODE_CONTACTS || !w->has_topology)
        return fail(XPBD_E_INVALID, "xpbd_world_contacts_begin: needs XPBD_MODE_CONTACTS and xpbd_world_set_polytopes");
    if (int rc = bind_device(w))
        return rc;
    w->stepped = true;
    return w->n ? build_neighbours(w, dt) : XPBD_OK;
}

int xpbd_world_contacts_substep(xpbd_world *w, double h)
{
    if (!w)
        return fail(XPBD_E_INVALID, "xpbd_world_contacts_substep: NULL world");
    if (w->mode != XPBD_MODE_CONTACTS || (w->n && !w->have_neighbours))
        return fail(XPBD_E_INVALID, "xpbd_world_contacts_substep: call xpbd_world_contacts_begin first");
    if (w->n == 0)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    return substep_contacts(w, h, nullptr, 0);
}

int xpbd_world_export_dynamic(xpbd_world *w, const uint32_t *dev_indices, uint32_t n, double *dev_buf)
{
    if (!w || (n && (!dev_indices || !dev_buf)))
        return fail(XPBD_E_INVALID, "xpbd_world_export_dynamic: NULL argument");
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(xpbd::launch_export_dynamic(w->arrays(), dev_indices, n, dev_buf, w->stream));
    return XPBD_OK;
}

int xpbd_world_import_dynamic(xpbd_world *w, const uint32_t *dev_indices, uint32_t n, const double *dev_buf)
{
    if (!w || (n && (!dev_indices || !dev_buf)))
        return fail(XPBD_E_INVALID, "xpbd_world_import_dynamic: NULL argument");
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(xpbd::launch_import_dynamic(w->arrays(), dev_indices, nullptr, n, dev_buf, w->stream));
    return XPBD_OK;
}

int xpbd_world_import_dynamic_rows(xpbd_world *w, const uint32_t *dev_indices, const uint32_t *dev_rows, uint32_t n,
                                   const double *dev_buf)
{
    if (!w || (n && (!dev_indices || !dev_rows || !dev_buf)))
        return fail(XPBD_E_INVALID, "xpbd_world_import_dynamic_rows: NULL argument");
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(xpbd::launch_import_dynamic(w->arrays(), dev_indices, dev_rows, n, dev_buf, w->stream));
    return XPBD_OK;
}

int xpbd_world_snapshot_positions(xpbd_world *w, const uint32_t *dev_indices, uint32_t n, double *dev_snapshot)
{
    if (!w || (n && (!dev_indices || !dev_snapshot)))
        return fail(XPBD_E_INVALID, "xpbd_world_snapshot_positions: NULL argument");
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(xpbd::launch_snapshot_positions(w->arrays(), dev_indices, n, dev_snapshot, w->stream));
    return XPBD_OK;
}

int xpbd_world_max_displacement2(xpbd_world *w, const uint32_t *dev_indices, uint32_t n, const double *dev_snapshot, const double *dev_scale,
                                 double *dev_max)
{
    if (!w || !dev_max || (n && (!dev_indices || !dev_snapshot)))
        return fail(XPBD_E_INVALID, "xpbd_world_max_displacement2: NULL argument");
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(xpbd::launch_max_displacement2(w->arrays(), dev_indices, n, dev_snapshot, dev_scale, dev_max, w->stream));
    return XPBD_OK;
}

int xpbd_world_set_sat_schedule(xpbd_world *w, uint32_t schedule)
{
    if (!w)
        return fail(XPBD_E_INVALID, "xpbd_world_set_sat_schedule: NULL world");
    if (schedule > XPBD_SAT_SCHEDULE_TWO_PASS)
        return fail(XPBD_E_INVALID, "xpbd_world_set_sat_schedule: unknown schedule %u", schedule);
    w->sat_schedule = schedule;
    return XPBD_OK;
}

int xpbd_world_set_narrowphase(xpbd_world *w, uint32_t narrowphase)
{
    if (!w || (narrowphase != XPBD_NARROWPHASE_SAT && narrowphase != XPBD_NARROWPHASE_GJK_EPA))
        return fail(XPBD_E_INVALID, "xpbd_world_set_narrowphase: bad argument");
    w->narrowphase = narrowphase;
    return XPBD_OK;
}

int xpbd_world_set_contact_pad(xpbd_world *w, double pad)
{
    if (!w || !(pad >= 0.0) || pad > 1.0e6)
        return fail(XPBD_E_INVALID, "xpbd_world_set_contact_pad: bad argument");
    w->contact_pad = pad;
    return XPBD_OK;
}

int xpbd_world_set_max_depenetration_speed(xpbd_world *w, double speed)
{
    if (!w || !(speed >= 0.0) || speed > 1.0e300)
        return fail(XPBD_E_INVALID, "xpbd_world_set_max_depenetration_speed: bad argument");
    w->max_depenetration_speed = speed;
    return XPBD_OK;
}

int xpbd_world_contact_stats(xpbd_world *w, uint64_t out[3])
{
    if (!w || !out)
        return fail(XPBD_E_INVALID, "xpbd_world_contact_stats: NULL argument");
    out[0] = w->n_pairs;
    out[1] = out[2] = 0;
    if (!w->cb_stats.ptr)
        return XPBD_OK;
    if (int rc = bind_device(w))
        return rc;
    unsigned long long host[2] = {0, 0};
    XPBD_HIP_TRY(hipMemcpyAsync(host, w->cb_stats.ptr, 16, hipMemcpyDeviceToHost, w->stream));
    XPBD_HIP_TRY(hipMemsetAsync(w->cb_stats.ptr, 0, 16, w->stream));
    w->stats_touching_seen = 0; // the schedule heuristic of build_neighbours starts counting afresh
    w->stats_pair_substeps_seen = w->stats_pair_substeps;
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    out[1] = host[0];
    out[2] = host[1];
    return XPBD_OK;
}

int xpbd_world_build_neighbours(xpbd_world *w, double dt, uint32_t *n_entries_out)
{
    if (!w || !n_entries_out)
        return fail(XPBD_E_INVALID, "xpbd_world_build_neighbours: NULL argument");
    if (!w->has_topology)
        return fail(XPBD_E_INVALID, "xpbd_world_build_neighbours: call xpbd_world_set_polytopes first");
    if (int rc = bind_device(w))
        return rc;
    if (int rc = build_neighbours(w, dt))
        return rc;
    *n_entries_out = w->n_entries;
    return XPBD_OK;
}

int xpbd_world_download_neighbours(xpbd_world *w, uint32_t *offsets, uint32_t *neighbours, uint32_t cap)
{
    if (!w || !offsets || (!neighbours && cap))
        return fail(XPBD_E_INVALID, "xpbd_world_download_neighbours: NULL argument");
    if (!w->have_neighbours)
        return fail(XPBD_E_INVALID, "xpbd_world_download_neighbours: no neighbour lists built yet");
    if (cap < w->n_entries)
        return fail(XPBD_E_CAPACITY, "xpbd_world_download_neighbours: %u entries, capacity %u", w->n_entries, cap);
    if (int rc = bind_device(w))
        return rc;
    XPBD_HIP_TRY(hipMemcpyAsync(offsets, w->cb_nbr_off.ptr, (size_t)(w->n + 1) * 4, hipMemcpyDeviceToHost, w->stream));
    if (w->n_entries)
        XPBD_HIP_TRY(hipMemcpyAsync(neighbours, w->cb_nbr.ptr, (size_t)w->n_entries * 4, hipMemcpyDeviceToHost, w->stream));
    XPBD_HIP_TRY(hipStreamSynchronize(w->stream));
    return XPBD_OK;
}

int xpbd_world_history_push(xpbd_world *w, uint32_t *index_out)
{
    if (!w)
        return fail(XPBD_E_INVALID, "xpbd_world_history_push: NULL world");
    if (w->n == 0)
        return fail(XPBD_E_INVALID, "xpbd_world_history_push: no bodies uploaded");
    if (int rc = bind_device(w))
        return rc;
    const size_t slot = w->history_slot_bytes();
    if ((size_t)(w->history_length + 1) * slot > w->history.bytes) {
        // grow geometrically into a new block, carrying the old states over (device to device)
        uint32_t capacity = (uint32_t)(w->history.bytes / slot);
        capacity = capacity < 8 ? 8 : capacity * 2;
        DeviceBuffer bigger;
        hipError_t e = bigger.reserve((size_t)capacity * slot);
        if (e != hipSuccess)
            return fail(e == hipErrorOutOfMemory ? XPBD_E_OOM : XPBD_E_HIP, "xpbd_world_history_push: %u states of %zu bytes: %s",
                        capacity, slot, hipGetErrorString(e));
        e = hipStreamSynchronize(w->stream);
        if (e == hipSuccess && w->history_length)
            e = hipMemcpy(bigger.ptr, w->history.ptr, (size_t)w->history_length * slot, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            bigger.release();
            return fail(XPBD_E_HIP, "xpbd_world_history_push: carrying %u states over failed: %s", w->history_length, hipGetErrorString(e));
        }
        w->history.release();
        w->history = bigger;
    }
    char *dst = static_cast<char *>(w->history.ptr) + (size_t)w->history_length * slot;
    const size_t dyn_bytes = (size_t)xpbd::kDynFields * w->stride * 8;
    XPBD_HIP_TRY(hipMemcpyAsync(dst, w->dyn.ptr, dyn_bytes, hipMemcpyDeviceToDevice, w->stream));
    XPBD_HIP_TRY(hipMemcpyAsync(dst + dyn_bytes, w->last_mask.ptr, (size_t)w->stride * 4, hipMemcpyDeviceToDevice, w->stream));
    w->history_stepped.push_back(w->stepped ? 1 : 0);
    if (index_out)
        *index_out = w->history_length;
    ++w->history_length;
    return XPBD_OK;
}

int xpbd_world_history_restore(xpbd_world *w, uint32_t index)
{
    if (!w)
        return fail(XPBD_E_INVALID, "xpbd_world_history_restore: NULL world");
    if (index >= w->history_length)
        return fail(XPBD_E_INVALID, "xpbd_world_history_restore: state %u of %u", index, w->history_length);
    if (int rc = bind_device(w))
        return rc;
    const size_t slot = w->history_slot_bytes();
    const char *src = static_cast<const char *>(w->history.ptr) + (size_t)index * slot;
    const size_t dyn_bytes = (size_t)xpbd::kDynFields * w->stride * 8;
    XPBD_HIP_TRY(hipMemcpyAsync(w->dyn.ptr, src, dyn_bytes, hipMemcpyDeviceToDevice, w->stream));
    XPBD_HIP_TRY(hipMemcpyAsync(w->last_mask.ptr, src + dyn_bytes, (size_t)w->stride * 4, hipMemcpyDeviceToDevice, w->stream));
    w->stepped = w->history_stepped[index] != 0;
    w->have_neighbours = false;
    w->trace_rows = 0; // the per-substep trace belongs to the step call that was overwritten
    return XPBD_OK;
}

int xpbd_world_history_truncate(xpbd_world *w, uint32_t length)
{
    if (!w)
        return fail(XPBD_E_INVALID, "xpbd_world_history_truncate: NULL world");
    if (length > w->history_length)
        return fail(XPBD_E_INVALID, "xpbd_world_history_truncate: length %u > %u", length, w->history_length);
    w->history_length = length;
    w->history_stepped.resize(length);
    return XPBD_OK;
}

uint32_t xpbd_world_history_length(const xpbd_world *w) { return w ? w->history_length : 0; }

int xpbd_selftest_div_sqrt(int32_t device, const double *a, const double *b, double *quotient, double *root,
                           uint32_t n)
{
    if (n && (!a || !b || !quotient || !root))
        return fail(XPBD_E_INVALID, "xpbd_selftest_div_sqrt: NULL argument");
    if (n == 0)
        return XPBD_OK;
    XPBD_HIP_TRY(hipSetDevice(device));
    DeviceBuffer buf;
    const size_t bytes = (size_t)n * 8;
    hipError_t e = buf.reserve(4 * bytes);
    if (e != hipSuccess)
        return fail(XPBD_E_OOM, "xpbd_selftest_div_sqrt: %s", hipGetErrorString(e));
    double *d = buf.as<double>();
    e = hipMemcpy(d, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = xpbd::launch_selftest_div_sqrt(d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n, n, nullptr);
    if (e == hipSuccess) e = hipMemcpy(quotient, d + 2 * (size_t)n, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(root, d + 3 * (size_t)n, bytes, hipMemcpyDeviceToHost);
    buf.release();
    if (e != hipSuccess)
        return fail(XPBD_E_HIP, "xpbd_selftest_div_sqrt: %s", hipGetErrorString(e));
    return XPBD_OK;
}

int xpbd_selftest_hbm_copy(int32_t device, uint64_t bytes, uint32_t repeats, double *gbytes_per_s)
{
    if (!gbytes_per_s || bytes < 16 || bytes >= (1ull << 40) || repeats == 0)
        return fail(XPBD_E_INVALID, "xpbd_selftest_hbm_copy: bad argument");
    *gbytes_per_s = 0.0;
    XPBD_HIP_TRY(hipSetDevice(device));
    bytes &= ~(uint64_t)15;
    DeviceBuffer src, dst;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = src.reserve(bytes);
    if (e == hipSuccess) e = dst.reserve(bytes);
    if (e == hipSuccess) e = hipMemset(src.ptr, 0x3c, bytes);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float best_ms = 0.0f;
    for (uint32_t variant = 0; variant < xpbd::kCopyVariants && e == hipSuccess; ++variant) { // report the fastest kernel
        const uint32_t nt = variant;
        e = xpbd::launch_copy16(src.ptr, dst.ptr, bytes, nt, nullptr); // warm-up (page tables, clocks)
        if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
        for (uint32_t k = 0; k < repeats && e == hipSuccess; ++k)
            e = xpbd::launch_copy16(src.ptr, dst.ptr, bytes, nt, nullptr);
        if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && (variant == 0 || ms < best_ms))
            best_ms = ms;
    }
    const float ms = best_ms;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    src.release();
    dst.release();
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? XPBD_E_OOM : XPBD_E_HIP, "xpbd_selftest_hbm_copy: %s", hipGetErrorString(e));
    *gbytes_per_s = 2.0 * (double)bytes * repeats / ((double)ms * 1e-3) / 1e9; // bytes read + bytes written
    return XPBD_OK;
}

int xpbd_selftest_gather(int32_t device, uint32_t records, uint32_t record_bytes, uint32_t read_bytes, uint32_t repeats, double *gbytes_per_s)
{
    if (!gbytes_per_s || records < 256 || (records & (records - 1)) != 0 || repeats == 0 || read_bytes > record_bytes)
        return fail(XPBD_E_INVALID, "xpbd_selftest_gather: records must be a power of two >= 256, read_bytes <= record_bytes");
    *gbytes_per_s = 0.0;
    XPBD_HIP_TRY(hipSetDevice(device));
    const size_t in_bytes = (size_t)records * record_bytes, out_bytes = (size_t)records * 8;
    DeviceBuffer in, out;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = in.reserve(in_bytes);
    if (e == hipSuccess) e = out.reserve(out_bytes);
    if (e == hipSuccess) e = hipMemset(in.ptr, 0, in_bytes);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float best_ms = 0.0f;
    for (uint32_t k = 0; k < repeats + 1 && e == hipSuccess; ++k) { // (the first launch warms up and is not counted)
        e = hipEventRecord(e0, nullptr);
        if (e == hipSuccess) e = xpbd::launch_gather_records(in.ptr, out.as<double>(), records, record_bytes, read_bytes, nullptr);
        if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && k > 0 && (k == 1 || ms < best_ms))
            best_ms = ms;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    in.release();
    out.release();
    if (e == hipErrorInvalidValue)
        return fail(XPBD_E_INVALID, "xpbd_selftest_gather: no kernel for %u bytes read of %u-byte records", read_bytes, record_bytes);
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? XPBD_E_OOM : XPBD_E_HIP, "xpbd_selftest_gather: %s", hipGetErrorString(e));
    *gbytes_per_s = ((double)records * read_bytes + (double)out_bytes) / ((double)best_ms * 1e-3) / 1e9;
    return XPBD_OK;
}

int xpbd_selftest_field_streams(int32_t device, uint64_t bodies, uint32_t tile_major, uint32_t repeats, double *gbytes_per_s)
{
    if (!gbytes_per_s || bodies < 64 || bodies > (1ull << 28) || repeats == 0)
        return fail(XPBD_E_INVALID, "xpbd_selftest_field_streams: bad argument");
    *gbytes_per_s = 0.0;
    XPBD_HIP_TRY(hipSetDevice(device));
    bodies = (bodies + 63) / 64 * 64;
    const size_t in_bytes = (size_t)bodies * (xpbd::kDynFields + xpbd::kStatFields) * 8, out_bytes = (size_t)bodies * xpbd::kDynFields * 8;
    DeviceBuffer in, out;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = in.reserve(in_bytes);
    if (e == hipSuccess) e = out.reserve(out_bytes);
    if (e == hipSuccess) e = hipMemset(in.ptr, 0, in_bytes);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float best_ms = 0.0f;
    for (uint32_t k = 0; k < repeats + 1 && e == hipSuccess; ++k) { // (the first launch warms up and is not counted)
        e = hipEventRecord(e0, nullptr);
        if (e == hipSuccess) e = xpbd::launch_field_streams(in.as<double>(), out.as<double>(), bodies, tile_major != 0, nullptr);
        if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && k > 0 && (k == 1 || ms < best_ms))
            best_ms = ms;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    in.release();
    out.release();
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? XPBD_E_OOM : XPBD_E_HIP, "xpbd_selftest_field_streams: %s", hipGetErrorString(e));
    *gbytes_per_s = (double)(in_bytes + out_bytes) / ((double)best_ms * 1e-3) / 1e9;
    return XPBD_OK;
}

} // extern "C"
