// xpbd_multi.cpp -- the multi-GPU world behind the C ABI (xpbd_multi_world_* in include/xpbd.h).
//
// EXTENSION: the reference is single-threaded and has neither body-body contacts nor any
// multi-device path; its caller is World::integrate (src/world.rs:34-43), which this replaces for an N-body world whose
// bodies are sharded over the GPUs of one node.  Parity: sharded == single device, bit for bit (tests).
//
// One xpbd_multi_world drives the LOCAL shards of a world of n_ranks shards -- all of them (a single process that owns
// every GPU of the node: what a Rust host would do) or one each (one process per GPU, the ranks of a launcher).  A shard is
// an ordinary xpbd_world in XPBD_MODE_CONTACTS that holds its OWNED bodies (a contiguous global index range) plus GHOST
// copies of the remote bodies that can reach an owned body before the next plan, in ascending global id -- so every
// neighbour list and every floating-point sum has the order of the single-device run.  Per substep:
//     substep on owned + ghost bodies  ->  export the owned boundary bodies' 13 dynamic doubles  ->  ONE all-gather
//     (RCCL over xGMI: ncclAllGather, all local shards in one group call)  ->  import into the ghosts.
// The halo plan (who mirrors whom) is built HERE, in C++, from the shards' own bodies plus three small all-gathers (cell keys,
// boundary lists, boundary records); no rank ever holds the global scene.  Once per frame the largest distance any owned
// body has travelled since the plan is reduced over all ranks: beyond halo_margin the step fails with XPBD_E_HALO (a
// remote contact may have been missed) instead of silently losing contacts, and with XPBD_MULTI_AUTO_REPLAN the halos are
// re-planned at half the margin.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/xpbd.h"
#include "xpbd_internal.h"
#include "xpbd_rccl.h"

namespace {

using xpbd::set_error;

#define MW_HIP_TRY(expr)                                                                                      \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return set_error(e_ == hipErrorOutOfMemory ? XPBD_E_OOM : XPBD_E_HIP, "%s failed: %s", #expr,     \
                             hipGetErrorString(e_));                                                          \
    } while (0)
#define MW_TRY(expr)            \
    do {                        \
        if (int rc_ = (expr))   \
            return rc_;         \
    } while (0)

constexpr uint32_t kDyn = 13;    // dynamic doubles per body: position, rotation, velocity, angular velocity
constexpr uint32_t kRigid = 38;  // sizeof(xpbd_rigid) / 8
constexpr uint32_t kRecord = 39; // a boundary body's plan-time record: its xpbd_rigid + the shape id
constexpr int64_t kCellBias = 1 << 20;

struct Range {
    uint32_t first, count;
};

// Ownership: contiguous index ranges, the first n % w ranks one body longer (constraint_solver_amd/sharding.py).
Range shard_range(uint32_t n, uint32_t rank, uint32_t w)
{
    const uint32_t base = n / w, extra = n % w;
    return Range{rank * base + std::min(rank, extra), base + (rank < extra ? 1u : 0u)};
}

uint32_t owner_of(uint32_t g, uint32_t n, uint32_t w)
{
    const uint32_t base = n / w, extra = n % w;
    const uint64_t long_part = (uint64_t)(base + 1) * extra;
    if (g < long_part)
        return g / (base + 1);
    return base ? extra + (uint32_t)((g - long_part) / base) : w - 1;
}

int64_t clamp_cell(double q)
{
    const double lim = (double)(kCellBias - 2);
    if (!(q >= -lim)) // NaN or far negative
        return -(kCellBias - 2);
    return q > lim ? kCellBias - 2 : (int64_t)q;
}

int64_t cell_key(int64_t x, int64_t y, int64_t z) { return ((x + kCellBias) << 42) | ((y + kCellBias) << 21) | (z + kCellBias); }

void cell_of_key(int64_t key, int64_t c[3])
{
    c[0] = (key >> 42) - kCellBias;
    c[1] = ((key >> 21) & ((1 << 21) - 1)) - kCellBias;
    c[2] = (key & ((1 << 21) - 1)) - kCellBias;
}

struct DevBuf {
    void *ptr = nullptr;
    size_t bytes = 0;
    hipError_t reserve(size_t want)
    {
        if (want <= bytes)
            return hipSuccess;
        if (ptr)
            want += want / 4; // a buffer that grows once (ghost lists after a re-plan) will grow again: see DeviceBuffer
        release();
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

// Which remote bodies a rank mirrors and which of its own bodies the others mirror; a pure function of the global cell
// keys and the joints, so every rank computes consistent plans.
struct HaloPlanner {
    uint32_t n = 0, w = 0;
    const int64_t *keys = nullptr;

    // ghosts: remote bodies in a cell within one cell of a cell this rank owns a body in (ascending);
    // boundary: this rank's bodies in a cell within one cell of a cell another rank owns a body in (ascending).
    // A joint between an owned and a remote body puts the remote one among the ghosts and the owned one on the boundary.
    // far (optional, one flag per owned body): no cell within two cells of the body's holds a foreign body.  Such a body
    // may travel halo_margin + edge / 2 before it can meet a body this rank does not mirror (any foreign body starts more
    // than two cell edges away, and 2 * (margin + edge / 2) = edge + 2 * margin is less than the 2 * edge - 2 r - pad the
    // two would have to close), the others halo_margin.
    // Cost: one pass over ALL cell keys that only decodes them; everything hashed lies within two cells of this rank's
    // bounding box (the foreign cells next to the slab) or belongs to the rank itself.
    void plan_rank(uint32_t rank, const xpbd_joint *joints, uint32_t n_joints, std::vector<uint32_t> &ghosts,
                   std::vector<uint32_t> &boundary, std::vector<uint8_t> *far = nullptr) const
    {
        const Range own = shard_range(n, rank, w);
        const uint32_t own_end = own.first + own.count;
        // this rank's unique cells and their bounding box
        std::unordered_map<int64_t, uint8_t> own_cells; // cell -> some other rank owns a body within one cell of it
        own_cells.reserve((size_t)own.count / 2 + 16);
        int64_t lo[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, hi[3] = {INT64_MIN, INT64_MIN, INT64_MIN};
        for (uint32_t g = own.first; g < own_end; ++g)
            if (own_cells.emplace(keys[g], 0).second) {
                int64_t c[3];
                cell_of_key(keys[g], c);
                for (int a = 0; a < 3; ++a) {
                    lo[a] = std::min(lo[a], c[a]);
                    hi[a] = std::max(hi[a], c[a]);
                }
            }
        // the foreign bodies within two cells of that box: their cells, and (within one cell of an own cell) the ghosts
        std::unordered_set<int64_t> foreign_cells;
        std::vector<uint32_t> candidates; // foreign bodies inside the box grown by one cell: the possible ghosts
        for (uint32_t g = 0; g < n; ++g) {
            if (own.count && g == own.first) {
                g = own_end - 1; // skip the own range
                continue;
            }
            int64_t c[3];
            cell_of_key(keys[g], c);
            bool in2 = true, in1 = true;
            for (int a = 0; a < 3; ++a) {
                in2 = in2 && c[a] >= lo[a] - 2 && c[a] <= hi[a] + 2;
                in1 = in1 && c[a] >= lo[a] - 1 && c[a] <= hi[a] + 1;
            }
            if (!in2 || own.count == 0)
                continue;
            foreign_cells.insert(keys[g]);
            if (in1)
                candidates.push_back(g);
        }
        // per own cell: a foreign body within one cell?  (then all its bodies are boundary bodies)
        for (auto &cell : own_cells) {
            int64_t c[3];
            cell_of_key(cell.first, c);
            bool seen = false;
            for (int dx = -1; dx <= 1 && !seen; ++dx)
                for (int dy = -1; dy <= 1 && !seen; ++dy)
                    for (int dz = -1; dz <= 1 && !seen; ++dz)
                        seen = foreign_cells.count(cell_key(c[0] + dx, c[1] + dy, c[2] + dz)) != 0;
            cell.second = seen;
        }
        std::vector<uint8_t> is_boundary(own.count, 0);
        for (uint32_t k = 0; k < own.count; ++k)
            is_boundary[k] = own_cells.find(keys[own.first + k])->second;
        // a candidate is a ghost iff an own cell lies within one cell of its cell
        std::unordered_map<int64_t, uint8_t> reached; // foreign cell -> within one cell of an own cell (memoised)
        std::vector<uint32_t> ghost_list;
        for (uint32_t g : candidates) {
            auto it = reached.find(keys[g]);
            if (it == reached.end()) {
                int64_t c[3];
                cell_of_key(keys[g], c);
                bool near = false;
                for (int dx = -1; dx <= 1 && !near; ++dx)
                    for (int dy = -1; dy <= 1 && !near; ++dy)
                        for (int dz = -1; dz <= 1 && !near; ++dz)
                            near = own_cells.count(cell_key(c[0] + dx, c[1] + dy, c[2] + dz)) != 0;
                it = reached.emplace(keys[g], near).first;
            }
            if (it->second)
                ghost_list.push_back(g);
        }
        for (uint32_t j = 0; j < n_joints; ++j) {
            const uint32_t a = joints[j].body_a, b = joints[j].body_b;
            const bool own_a = a >= own.first && a < own_end, own_b = b >= own.first && b < own_end;
            if (own_a && !own_b) {
                ghost_list.push_back(b);
                is_boundary[a - own.first] = 1;
            } else if (own_b && !own_a) {
                ghost_list.push_back(a);
                is_boundary[b - own.first] = 1;
            }
        }
        std::sort(ghost_list.begin(), ghost_list.end());
        ghost_list.erase(std::unique(ghost_list.begin(), ghost_list.end()), ghost_list.end());
        ghosts.swap(ghost_list);
        boundary.clear();
        for (uint32_t k = 0; k < own.count; ++k)
            if (is_boundary[k])
                boundary.push_back(own.first + k);
        if (far) {
            // dilate the nearby foreign cells by two cells; an own body outside that set (and not a boundary body) is far
            far->assign(own.count, 0);
            const size_t limit = 20000; // beyond that many foreign cells around the slab the dilation is not worth it: nobody is far
            if (own.count && foreign_cells.size() <= limit) {
                std::unordered_set<int64_t> near_cells;
                near_cells.reserve(foreign_cells.size() * 40 + 16);
                for (int64_t key : foreign_cells) {
                    int64_t c[3];
                    cell_of_key(key, c);
                    for (int dx = -2; dx <= 2; ++dx)
                        for (int dy = -2; dy <= 2; ++dy)
                            for (int dz = -2; dz <= 2; ++dz)
                                near_cells.insert(cell_key(c[0] + dx, c[1] + dy, c[2] + dz));
                }
                for (uint32_t k = 0; k < own.count; ++k)
                    (*far)[k] = !near_cells.count(keys[own.first + k]) && !is_boundary[k];
            }
        }
    }
};

struct Shard {
    int device = 0;
    uint32_t rank = 0;
    Range own{0, 0};
    xpbd_world *world = nullptr;
    hipStream_t stream = nullptr;      // the shard's world stream: every kernel of the shard
    hipStream_t comm_stream = nullptr; // the per-substep halo all-gather, overlapping the interior bodies' kernels
    ncclComm_t comm = nullptr;
    hipEvent_t ev_send = nullptr, ev_recv = nullptr; // in-process transport
    hipEvent_t ev_ready = nullptr, ev_gathered = nullptr; // world stream -> communication stream -> world stream
    // the owned bodies as last uploaded / re-planned (host): kRigid doubles each, shape ids
    std::vector<double> owned;
    std::vector<uint32_t> owned_sid;
    // plan
    std::vector<uint32_t> local_ids, ghosts, boundary;
    std::vector<uint8_t> far; // per owned body: more than two cells away from every foreign body (larger travel allowance)
    uint32_t own_slot0 = 0; // local slot of the first owned body (the owned bodies are contiguous in the local order)
    DevBuf boundary_slots, ghost_slots, ghost_rows, owned_slots, skip_flags, disp_scale, send, recv, snapshot, disp, disp_all, stage_send, stage_recv;
    double *disp_host = nullptr; // pinned, n_ranks doubles
};

} // namespace

struct xpbd_multi_world {
    uint32_t n_ranks = 1, first_rank = 0, transport = XPBD_TRANSPORT_RCCL, flags = 0, narrowphase = XPBD_NARROWPHASE_SAT;
    double pad = 0.02, margin = 0.5;
    std::vector<Shard> shards;
    const xpbd::RcclApi *rccl = nullptr;
    bool have_shapes = false, planned = false, violated = false;
    std::vector<double> shape_radius, shape_centroid; // per shape: max |vertex - centroid|, centroid xyz
    uint32_t n_global = 0, first_global = 0, n_bodies = 0, capacity = 1;
    std::vector<xpbd_joint> joints;
    uint64_t plans = 0;
    double cell_edge = 0.0;
    double last_displacement = 0.0; // the largest fraction of its travel allowance any body had used at the last check, times halo_margin
    bool all_local() const { return shards.size() == n_ranks; }
    uint32_t rows_per_rank() const { return capacity; }
};

namespace {

int bind(const Shard &s)
{
    MW_HIP_TRY(hipSetDevice(s.device));
    return XPBD_OK;
}

int nccl_fail(const xpbd_multi_world *mw, ncclResult_t r, const char *what)
{
    return set_error(XPBD_E_HIP, "%s failed: %s (RCCL from %s)", what, mw->rccl ? mw->rccl->GetErrorString(r) : "?", mw->rccl ? mw->rccl->path : "?");
}

// One all-gather over all ranks: every local shard contributes `bytes` from its `send` and receives n_ranks x bytes into its
// `recv` (device pointers, picked per shard by the callbacks), ordered on the shards' streams.
// `on_comm_stream`: enqueue on the shards' communication streams (the caller orders them against the world streams with
// events) instead of the world streams.
template <class Send, class Recv>
int all_gather_device(xpbd_multi_world *mw, size_t bytes, Send send_of, Recv recv_of, bool on_comm_stream = false)
{
    auto stream_of = [on_comm_stream](Shard &s) { return on_comm_stream ? s.comm_stream : s.stream; };
    if (mw->transport == XPBD_TRANSPORT_RCCL) {
        // RCCL reads the thread's last HIP error after its own calls: a stale, harmless one left by somebody else in the
        // process (hipErrorNotReady from an event query, say) would be reported as "unhandled cuda error"
        (void)hipGetLastError();
        ncclResult_t r = mw->rccl->GroupStart();
        if (r != ncclSuccess)
            return nccl_fail(mw, r, "ncclGroupStart");
        for (Shard &s : mw->shards) {
            MW_TRY(bind(s));
            r = mw->rccl->AllGather(send_of(s), recv_of(s), bytes, ncclChar, s.comm, stream_of(s));
            if (r != ncclSuccess) {
                (void)mw->rccl->GroupEnd();
                return nccl_fail(mw, r, "ncclAllGather");
            }
        }
        r = mw->rccl->GroupEnd();
        if (r != ncclSuccess)
            return nccl_fail(mw, r, "ncclGroupEnd");
        return XPBD_OK;
    }
    // XPBD_TRANSPORT_LOCAL: every rank lives in this process; peer copies ordered by events
    for (Shard &p : mw->shards) {
        MW_TRY(bind(p));
        MW_HIP_TRY(hipEventRecord(p.ev_send, stream_of(p)));
    }
    for (Shard &r : mw->shards) {
        MW_TRY(bind(r));
        for (Shard &p : mw->shards) {
            if (&p != &r)
                MW_HIP_TRY(hipStreamWaitEvent(stream_of(r), p.ev_send, 0));
            MW_HIP_TRY(hipMemcpyAsync(static_cast<char *>(recv_of(r)) + (size_t)p.rank * bytes, send_of(p), bytes, hipMemcpyDefault, stream_of(r)));
        }
        MW_HIP_TRY(hipEventRecord(r.ev_recv, stream_of(r)));
    }
    for (Shard &p : mw->shards) { // nobody overwrites its send buffer before every peer has read it
        MW_TRY(bind(p));
        for (Shard &r : mw->shards)
            if (&p != &r)
                MW_HIP_TRY(hipStreamWaitEvent(stream_of(p), r.ev_recv, 0));
    }
    return XPBD_OK;
}

// Plan-time all-gather of host data: send[k] = `bytes` of local shard k; out = the n_ranks x bytes everybody ends up with.
int all_gather_host(xpbd_multi_world *mw, const std::vector<const void *> &send, size_t bytes, std::vector<uint8_t> &out)
{
    out.assign((size_t)mw->n_ranks * bytes, 0);
    if (bytes == 0)
        return XPBD_OK;
    if (mw->all_local() && !(mw->flags & XPBD_MULTI_PLAN_THROUGH_DEVICE)) { // every rank is here: no device round trip needed
        for (size_t k = 0; k < mw->shards.size(); ++k)
            std::memcpy(out.data() + (size_t)mw->shards[k].rank * bytes, send[k], bytes);
        return XPBD_OK;
    }
    for (size_t k = 0; k < mw->shards.size(); ++k) {
        Shard &s = mw->shards[k];
        MW_TRY(bind(s));
        MW_HIP_TRY(hipStreamSynchronize(s.stream)); // reserve() may free a block that is still in use
        MW_HIP_TRY(s.stage_send.reserve(bytes));
        MW_HIP_TRY(s.stage_recv.reserve((size_t)mw->n_ranks * bytes));
        MW_HIP_TRY(hipMemcpyAsync(s.stage_send.ptr, send[k], bytes, hipMemcpyHostToDevice, s.stream));
    }
    MW_TRY(all_gather_device(mw, bytes, [](Shard &s) { return s.stage_send.ptr; }, [](Shard &s) { return s.stage_recv.ptr; }));
    for (Shard &s : mw->shards) { // every shard takes part in the collective; the content is the same everywhere
        MW_TRY(bind(s));
        if (&s == &mw->shards[0])
            MW_HIP_TRY(hipMemcpyAsync(out.data(), s.stage_recv.ptr, out.size(), hipMemcpyDeviceToHost, s.stream));
        MW_HIP_TRY(hipStreamSynchronize(s.stream));
    }
    return XPBD_OK;
}

template <class T>
int upload_vector(DevBuf &buf, const std::vector<T> &v, hipStream_t stream)
{
    MW_HIP_TRY(buf.reserve(std::max<size_t>(v.size() * sizeof(T), 8)));
    if (!v.empty())
        MW_HIP_TRY(hipMemcpyAsync(buf.ptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, stream));
    return XPBD_OK;
}

// Builds the halos from the shards' owned bodies (Shard::owned) and uploads every shard's local world.
int make_plan(xpbd_multi_world *mw)
{
    const uint32_t n = mw->n_global, w = mw->n_ranks;
    const size_t n_local = mw->shards.size();
    // 1. bounding spheres of the owned bodies (centre = position + center_of_mass, radius = r_shape + |centroid - com|:
    //    conservative whatever the rotation) and the largest radius of the whole world
    std::vector<std::vector<double>> centre(n_local);
    std::vector<double> rmax_local(n_local, 0.0);
    for (size_t k = 0; k < n_local; ++k) {
        const Shard &s = mw->shards[k];
        centre[k].resize((size_t)3 * s.own.count);
        for (uint32_t i = 0; i < s.own.count; ++i) {
            const double *b = &s.owned[(size_t)i * kRigid];
            const uint32_t sid = s.owned_sid[i];
            double off2 = 0.0;
            for (int a = 0; a < 3; ++a) {
                centre[k][3 * (size_t)i + a] = b[31 + a] + b[28 + a];
                const double d = mw->shape_centroid[3 * (size_t)sid + a] - b[28 + a];
                off2 += d * d;
            }
            const double r = mw->shape_radius[sid] + std::sqrt(off2);
            if (r > rmax_local[k])
                rmax_local[k] = r;
        }
    }
    std::vector<uint8_t> gathered;
    {
        std::vector<const void *> send(n_local);
        for (size_t k = 0; k < n_local; ++k)
            send[k] = &rmax_local[k];
        MW_TRY(all_gather_host(mw, send, sizeof(double), gathered));
    }
    double rmax = 0.0;
    for (uint32_t r = 0; r < w; ++r)
        rmax = std::max(rmax, reinterpret_cast<const double *>(gathered.data())[r]);
    const double edge = 2.0 * (rmax + mw->pad + mw->margin);
    if (!(edge > 0.0) || !(edge <= 1.0e300))
        return set_error(XPBD_E_INVALID, "xpbd_multi_world: cell edge %g from radius %g, pad %g, halo_margin %g", edge, rmax, mw->pad, mw->margin);

    // 2. grid cell of every body of the world (8 bytes per body, one all-gather)
    const uint32_t max_count = shard_range(n, 0, w).count;
    std::vector<std::vector<int64_t>> keys_local(n_local);
    {
        std::vector<const void *> send(n_local);
        for (size_t k = 0; k < n_local; ++k) {
            const Shard &s = mw->shards[k];
            keys_local[k].assign(max_count, 0);
            for (uint32_t i = 0; i < s.own.count; ++i)
                keys_local[k][i] = cell_key(clamp_cell(std::floor(centre[k][3 * (size_t)i] / edge)), clamp_cell(std::floor(centre[k][3 * (size_t)i + 1] / edge)),
                                            clamp_cell(std::floor(centre[k][3 * (size_t)i + 2] / edge)));
            send[k] = keys_local[k].data();
        }
        MW_TRY(all_gather_host(mw, send, (size_t)max_count * 8, gathered));
    }
    std::vector<int64_t> keys(n);
    for (uint32_t r = 0; r < w; ++r) {
        const Range rr = shard_range(n, r, w);
        std::memcpy(keys.data() + rr.first, gathered.data() + (size_t)r * max_count * 8, (size_t)rr.count * 8);
    }

    // 3. who mirrors whom
    HaloPlanner planner;
    planner.n = n, planner.w = w, planner.keys = keys.data();
    for (Shard &s : mw->shards)
        planner.plan_rank(s.rank, mw->joints.data(), (uint32_t)mw->joints.size(), s.ghosts, s.boundary, &s.far);
    mw->cell_edge = edge;

    // 4. the boundary lists of all ranks (ascending global ids) fix the rows of the per-substep all-gather
    std::vector<uint32_t> counts(w);
    {
        std::vector<uint32_t> mine(n_local);
        std::vector<const void *> send(n_local);
        for (size_t k = 0; k < n_local; ++k) {
            mine[k] = (uint32_t)mw->shards[k].boundary.size();
            send[k] = &mine[k];
        }
        MW_TRY(all_gather_host(mw, send, 4, gathered));
        std::memcpy(counts.data(), gathered.data(), (size_t)w * 4);
    }
    mw->capacity = std::max(1u, *std::max_element(counts.begin(), counts.end()));
    const uint32_t cap = mw->capacity;
    std::vector<uint32_t> lists((size_t)w * cap);
    std::vector<double> records((size_t)w * cap * kRecord);
    {
        std::vector<std::vector<uint32_t>> pad_list(n_local);
        std::vector<std::vector<double>> pad_rec(n_local);
        std::vector<const void *> send(n_local);
        for (size_t k = 0; k < n_local; ++k) {
            const Shard &s = mw->shards[k];
            pad_list[k].assign(cap, 0xFFFFFFFFu);
            std::copy(s.boundary.begin(), s.boundary.end(), pad_list[k].begin());
            send[k] = pad_list[k].data();
        }
        MW_TRY(all_gather_host(mw, send, (size_t)cap * 4, gathered));
        std::memcpy(lists.data(), gathered.data(), lists.size() * 4);
        for (size_t k = 0; k < n_local; ++k) {
            const Shard &s = mw->shards[k];
            pad_rec[k].assign((size_t)cap * kRecord, 0.0);
            for (size_t q = 0; q < s.boundary.size(); ++q) {
                const uint32_t i = s.boundary[q] - s.own.first;
                std::memcpy(&pad_rec[k][q * kRecord], &s.owned[(size_t)i * kRigid], kRigid * 8);
                pad_rec[k][q * kRecord + kRigid] = (double)s.owned_sid[i];
            }
            send[k] = pad_rec[k].data();
        }
        MW_TRY(all_gather_host(mw, send, (size_t)cap * kRecord * 8, gathered));
        std::memcpy(records.data(), gathered.data(), records.size() * 8);
    }

    // 5. every shard's local world: owned + ghost bodies in ascending global id
    const uint32_t rows = mw->rows_per_rank();
    for (Shard &s : mw->shards) {
        MW_TRY(bind(s));
        const uint32_t n_ghost = (uint32_t)s.ghosts.size(), n_loc = s.own.count + n_ghost;
        std::vector<double> aos((size_t)n_loc * kRigid);
        std::vector<uint32_t> sid(n_loc), ghost_slots(n_ghost), ghost_rows(n_ghost), boundary_slots(s.boundary.size()), owned_slots(s.own.count);
        s.local_ids.resize(n_loc);
        uint32_t slot = 0, gq = 0;
        auto put_ghost = [&](uint32_t g) -> int {
            const uint32_t o = owner_of(g, n, w);
            const uint32_t *lo = &lists[(size_t)o * cap], *hi = lo + counts[o];
            const uint32_t *at = std::lower_bound(lo, hi, g);
            if (at == hi || *at != g)
                return set_error(XPBD_E_HIP, "xpbd_multi_world: body %u is mirrored by rank %u but not exported by its owner %u (inconsistent plans)", g, s.rank, o);
            const size_t row = (size_t)o * cap + (size_t)(at - lo);
            std::memcpy(&aos[(size_t)slot * kRigid], &records[row * kRecord], kRigid * 8);
            sid[slot] = (uint32_t)records[row * kRecord + kRigid];
            ghost_slots[gq] = slot;
            ghost_rows[gq] = o * rows + (uint32_t)(at - lo);
            ++gq;
            s.local_ids[slot++] = g;
            return XPBD_OK;
        };
        size_t gi = 0;
        for (; gi < s.ghosts.size() && s.ghosts[gi] < s.own.first; ++gi)
            MW_TRY(put_ghost(s.ghosts[gi]));
        s.own_slot0 = slot;
        if (s.own.count)
            std::memcpy(&aos[(size_t)slot * kRigid], s.owned.data(), (size_t)s.own.count * kRigid * 8);
        for (uint32_t i = 0; i < s.own.count; ++i) {
            sid[slot] = s.owned_sid[i];
            owned_slots[i] = slot;
            s.local_ids[slot++] = s.own.first + i;
        }
        for (; gi < s.ghosts.size(); ++gi)
            MW_TRY(put_ghost(s.ghosts[gi]));
        for (size_t q = 0; q < s.boundary.size(); ++q)
            boundary_slots[q] = s.own_slot0 + (s.boundary[q] - s.own.first);
        if (int rc = xpbd_world_upload_bodies(s.world, reinterpret_cast<const xpbd_rigid *>(aos.data()), sid.data(), n_loc))
            return rc;
        // joints whose two bodies are both present here, in global joint order, re-indexed to local slots
        std::vector<xpbd_joint> local_joints;
        for (const xpbd_joint &j : mw->joints) {
            const auto a = std::lower_bound(s.local_ids.begin(), s.local_ids.end(), j.body_a), b = std::lower_bound(s.local_ids.begin(), s.local_ids.end(), j.body_b);
            if (a == s.local_ids.end() || *a != j.body_a || b == s.local_ids.end() || *b != j.body_b)
                continue;
            xpbd_joint l = j;
            l.body_a = (uint32_t)(a - s.local_ids.begin());
            l.body_b = (uint32_t)(b - s.local_ids.begin());
            local_joints.push_back(l);
        }
        if (int rc = xpbd_world_set_joints(s.world, local_joints.data(), (uint32_t)local_joints.size()))
            return rc;
        MW_HIP_TRY(hipStreamSynchronize(s.stream));
        MW_TRY(upload_vector(s.boundary_slots, boundary_slots, s.stream));
        MW_TRY(upload_vector(s.ghost_slots, ghost_slots, s.stream));
        MW_TRY(upload_vector(s.ghost_rows, ghost_rows, s.stream));
        MW_TRY(upload_vector(s.owned_slots, owned_slots, s.stream));
        std::vector<uint8_t> skip(n_loc, 0); // what the interior launch leaves out: boundary bodies (done first) and ghosts (done last)
        for (uint32_t q : boundary_slots)
            skip[q] = 1;
        for (uint32_t q : ghost_slots)
            skip[q] = 1;
        MW_TRY(upload_vector(s.skip_flags, skip, s.stream));
        // 1 / allowance^2 per owned body: the displacement check then yields the largest FRACTION of its allowance any body has used
        std::vector<double> scale(s.own.count);
        const double near_allow = mw->margin, far_allow = mw->margin + 0.5 * edge;
        for (uint32_t i = 0; i < s.own.count; ++i) {
            const double allow = s.far[i] ? far_allow : near_allow;
            scale[i] = 1.0 / (allow * allow);
        }
        MW_TRY(upload_vector(s.disp_scale, scale, s.stream));
        MW_HIP_TRY(s.send.reserve((size_t)rows * kDyn * 8));
        MW_HIP_TRY(s.recv.reserve((size_t)w * rows * kDyn * 8));
        MW_HIP_TRY(hipMemsetAsync(s.send.ptr, 0, (size_t)rows * kDyn * 8, s.stream));
        MW_HIP_TRY(s.snapshot.reserve(std::max<size_t>((size_t)3 * s.own.count * 8, 8)));
        MW_HIP_TRY(s.disp.reserve(8));
        MW_HIP_TRY(s.disp_all.reserve((size_t)w * 8));
        if (int rc = xpbd_world_snapshot_positions(s.world, s.owned_slots.as<uint32_t>(), s.own.count, s.snapshot.as<double>()))
            return rc;
        MW_HIP_TRY(hipStreamSynchronize(s.stream)); // the host vectors above go out of scope
    }
    mw->planned = true;
    mw->violated = false;
    mw->last_displacement = 0.0;
    ++mw->plans;
    return XPBD_OK;
}

// The owned bodies' current state back into Shard::owned (for a re-plan or a download).
int fetch_owned(xpbd_multi_world *mw)
{
    for (Shard &s : mw->shards) {
        MW_TRY(bind(s));
        const uint32_t n_loc = (uint32_t)s.local_ids.size();
        std::vector<double> aos((size_t)n_loc * kRigid);
        if (int rc = xpbd_world_download_bodies(s.world, reinterpret_cast<xpbd_rigid *>(aos.data()), n_loc))
            return rc;
        if (s.own.count)
            std::memcpy(s.owned.data(), &aos[(size_t)s.own_slot0 * kRigid], (size_t)s.own.count * kRigid * 8);
    }
    return XPBD_OK;
}

// Largest fraction of its travel allowance any owned body of any rank has used since the plan, agreed on by all ranks and
// expressed in margin-equivalent metres (x halo_margin): a body next to a shard boundary may travel halo_margin, one more
// than two cells away from every foreign body halo_margin + half a cell edge (HaloPlanner::plan_rank).
int measure_displacement(xpbd_multi_world *mw, double *out)
{
    for (Shard &s : mw->shards) {
        MW_TRY(bind(s));
        MW_HIP_TRY(hipMemsetAsync(s.disp.ptr, 0, 8, s.stream));
        if (int rc = xpbd_world_max_displacement2(s.world, s.owned_slots.as<uint32_t>(), s.own.count, s.snapshot.as<double>(), s.disp_scale.as<double>(),
                                                  s.disp.as<double>()))
            return rc;
    }
    MW_TRY(all_gather_device(mw, 8, [](Shard &s) { return s.disp.ptr; }, [](Shard &s) { return s.disp_all.ptr; })); // 8 bytes per rank
    double worst = 0.0;
    for (Shard &s : mw->shards) {
        MW_TRY(bind(s));
        MW_HIP_TRY(hipMemcpyAsync(s.disp_host, s.disp_all.ptr, (size_t)mw->n_ranks * 8, hipMemcpyDeviceToHost, s.stream));
        MW_HIP_TRY(hipStreamSynchronize(s.stream));
        for (uint32_t r = 0; r < mw->n_ranks; ++r)
            worst = std::max(worst, s.disp_host[r]);
    }
    *out = std::sqrt(worst) * mw->margin; // "margin-equivalent" metres: halo_margin means the allowance is used up
    return XPBD_OK;
}

void destroy(xpbd_multi_world *mw)
{
    if (!mw)
        return;
    for (Shard &s : mw->shards) {
        (void)hipSetDevice(s.device);
        if (s.stream)
            (void)hipStreamSynchronize(s.stream);
        if (s.comm && mw->rccl)
            (void)mw->rccl->CommDestroy(s.comm);
        for (DevBuf *b : {&s.boundary_slots, &s.ghost_slots, &s.ghost_rows, &s.owned_slots, &s.skip_flags, &s.disp_scale, &s.send, &s.recv, &s.snapshot, &s.disp, &s.disp_all, &s.stage_send, &s.stage_recv})
            b->release();
        if (s.disp_host)
            (void)hipHostFree(s.disp_host);
        if (s.comm_stream)
            (void)hipStreamDestroy(s.comm_stream);
        if (s.ev_send)
            (void)hipEventDestroy(s.ev_send);
        if (s.ev_ready)
            (void)hipEventDestroy(s.ev_ready);
        if (s.ev_gathered)
            (void)hipEventDestroy(s.ev_gathered);
        if (s.ev_recv)
            (void)hipEventDestroy(s.ev_recv);
        xpbd_world_destroy(s.world);
    }
    delete mw;
}

} // namespace

extern "C" {

int xpbd_comm_unique_id(uint8_t id[XPBD_COMM_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == XPBD_COMM_ID_BYTES, "XPBD_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
    if (!id)
        return set_error(XPBD_E_INVALID, "xpbd_comm_unique_id: NULL argument");
    const char *why = nullptr;
    const xpbd::RcclApi *api = xpbd::rccl_api(&why);
    if (!api)
        return set_error(XPBD_E_NO_DEVICE, "xpbd_comm_unique_id: RCCL is not available (%s)", why);
    ncclUniqueId u;
    (void)hipGetLastError(); // see all_gather_device
    const ncclResult_t r = api->GetUniqueId(&u);
    if (r != ncclSuccess)
        return set_error(XPBD_E_HIP, "ncclGetUniqueId failed: %s", api->GetErrorString(r));
    std::memcpy(id, &u, XPBD_COMM_ID_BYTES);
    return XPBD_OK;
}

const char *xpbd_comm_library(void)
{
    const xpbd::RcclApi *api = xpbd::rccl_api(nullptr);
    return api ? api->path : nullptr;
}

void xpbd_multi_config_default(xpbd_multi_config *cfg)
{
    if (!cfg)
        return;
    std::memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = sizeof *cfg;
    cfg->n_ranks = 1;
    cfg->n_local = 1;
    cfg->transport = XPBD_TRANSPORT_RCCL;
    cfg->contact_pad = 0.02;
    cfg->halo_margin = 0.5;
    cfg->narrowphase = XPBD_NARROWPHASE_SAT;
}

int xpbd_multi_world_create(xpbd_multi_world **out, const xpbd_multi_config *cfg)
{
    if (!out || !cfg)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_create: NULL argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(xpbd_multi_config))
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_create: struct_size %u != %zu", cfg->struct_size, sizeof(xpbd_multi_config));
    if (cfg->n_ranks == 0 || cfg->n_ranks > 64 || cfg->n_local == 0 || cfg->first_rank + cfg->n_local > cfg->n_ranks || !cfg->devices)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_create: ranks [%u, %u) of %u (at most 64) and a device list are needed", cfg->first_rank,
                         cfg->first_rank + cfg->n_local, cfg->n_ranks);
    if (cfg->transport != XPBD_TRANSPORT_RCCL && cfg->transport != XPBD_TRANSPORT_LOCAL)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_create: unknown transport %u", cfg->transport);
    if (cfg->transport == XPBD_TRANSPORT_LOCAL && cfg->n_local != cfg->n_ranks)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_create: XPBD_TRANSPORT_LOCAL needs every rank in this process (n_local == n_ranks)");
    if (cfg->transport == XPBD_TRANSPORT_RCCL && !cfg->comm_id)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_create: XPBD_TRANSPORT_RCCL needs comm_id (xpbd_comm_unique_id on one rank, handed to all)");
    if (!(cfg->contact_pad >= 0.0) || !(cfg->halo_margin > 0.0) || cfg->contact_pad > 1.0e6 || cfg->halo_margin > 1.0e6)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_create: contact_pad %g / halo_margin %g", cfg->contact_pad, cfg->halo_margin);
    if (cfg->flags & ~(XPBD_MULTI_AUTO_REPLAN | XPBD_MULTI_PLAN_THROUGH_DEVICE))
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_create: unknown flags 0x%x", cfg->flags);
    if (cfg->narrowphase != XPBD_NARROWPHASE_SAT && cfg->narrowphase != XPBD_NARROWPHASE_GJK_EPA)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_create: unknown narrowphase %u", cfg->narrowphase);

    xpbd_multi_world *mw = new (std::nothrow) xpbd_multi_world;
    if (!mw)
        return set_error(XPBD_E_OOM, "xpbd_multi_world_create: host allocation failed");
    mw->n_ranks = cfg->n_ranks, mw->first_rank = cfg->first_rank, mw->transport = cfg->transport, mw->flags = cfg->flags;
    mw->pad = cfg->contact_pad, mw->margin = cfg->halo_margin, mw->narrowphase = cfg->narrowphase;
    mw->shards.resize(cfg->n_local);
    auto bail = [&](int rc) {
        destroy(mw);
        return rc;
    };
    if (mw->transport == XPBD_TRANSPORT_RCCL) {
        const char *why = nullptr;
        mw->rccl = xpbd::rccl_api(&why);
        if (!mw->rccl)
            return bail(set_error(XPBD_E_NO_DEVICE, "xpbd_multi_world_create: RCCL is not available (%s)", why));
    }
    for (uint32_t k = 0; k < cfg->n_local; ++k) {
        Shard &s = mw->shards[k];
        s.device = cfg->devices[k];
        s.rank = cfg->first_rank + k;
        xpbd_config wc;
        xpbd_config_default(&wc);
        wc.device = s.device;
        wc.mode = XPBD_MODE_CONTACTS;
        if (int rc = xpbd_world_create(&s.world, &wc))
            return bail(rc);
        if (int rc = xpbd_world_set_contact_pad(s.world, mw->pad))
            return bail(rc);
        if (int rc = xpbd_world_set_narrowphase(s.world, mw->narrowphase))
            return bail(rc);
        s.stream = static_cast<hipStream_t>(xpbd_world_get_stream(s.world));
        hipError_t e = hipSetDevice(s.device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_send, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_recv, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_gathered, hipEventDisableTiming);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.comm_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&s.disp_host), (size_t)cfg->n_ranks * 8, hipHostMallocDefault);
        if (e != hipSuccess)
            return bail(set_error(XPBD_E_HIP, "xpbd_multi_world_create: %s", hipGetErrorString(e)));
    }
    if (mw->transport == XPBD_TRANSPORT_RCCL) {
        ncclUniqueId id;
        std::memcpy(&id, cfg->comm_id, sizeof id);
        (void)hipGetLastError(); // see all_gather_device
        ncclResult_t r = mw->rccl->GroupStart();
        for (Shard &s : mw->shards) {
            if (r != ncclSuccess)
                break;
            if (hipSetDevice(s.device) != hipSuccess) {
                (void)mw->rccl->GroupEnd();
                return bail(set_error(XPBD_E_HIP, "xpbd_multi_world_create: hipSetDevice(%d) failed", s.device));
            }
            r = mw->rccl->CommInitRank(&s.comm, (int)mw->n_ranks, id, (int)s.rank);
        }
        const ncclResult_t r_end = mw->rccl->GroupEnd();
        if (r == ncclSuccess)
            r = r_end;
        if (r != ncclSuccess)
            return bail(nccl_fail(mw, r, "ncclCommInitRank"));
    }
    *out = mw;
    return XPBD_OK;
}

void xpbd_multi_world_destroy(xpbd_multi_world *mw) { destroy(mw); }

int xpbd_multi_world_set_polytopes(xpbd_multi_world *mw, const xpbd_polytope *shapes, uint32_t n_shapes)
{
    if (!mw || !shapes || n_shapes == 0)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_set_polytopes: NULL argument or no shapes");
    for (Shard &s : mw->shards)
        if (int rc = xpbd_world_set_polytopes(s.world, shapes, n_shapes))
            return rc;
    mw->shape_radius.assign(n_shapes, 0.0);
    mw->shape_centroid.assign((size_t)3 * n_shapes, 0.0);
    for (uint32_t k = 0; k < n_shapes; ++k) {
        const xpbd_polytope &p = shapes[k];
        for (int a = 0; a < 3; ++a)
            mw->shape_centroid[3 * (size_t)k + a] = p.centroid[a];
        for (uint32_t v = 0; v < p.n_vertices; ++v) {
            double d2 = 0.0;
            for (int a = 0; a < 3; ++a) {
                const double d = p.vertices_xyz[3 * (size_t)v + a] - p.centroid[a];
                d2 += d * d;
            }
            mw->shape_radius[k] = std::max(mw->shape_radius[k], std::sqrt(d2));
        }
    }
    mw->have_shapes = true;
    mw->planned = false;
    return XPBD_OK;
}

int xpbd_multi_world_upload(xpbd_multi_world *mw, const xpbd_rigid *bodies, const uint32_t *shape_id, uint32_t first_global, uint32_t n_bodies,
                            uint32_t n_global, const xpbd_joint *joints, uint32_t n_joints)
{
    if (!mw || (n_bodies && !bodies) || (n_joints && !joints))
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_upload: NULL argument");
    if (!mw->have_shapes)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_upload: call xpbd_multi_world_set_polytopes first");
    const Range lo = shard_range(n_global, mw->first_rank, mw->n_ranks), hi = shard_range(n_global, mw->first_rank + (uint32_t)mw->shards.size() - 1, mw->n_ranks);
    if (first_global != lo.first || n_bodies != hi.first + hi.count - lo.first)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_upload: ranks [%u, %u) of %u own bodies [%u, %u) of %u, got [%u, %u)", mw->first_rank,
                         mw->first_rank + (uint32_t)mw->shards.size(), mw->n_ranks, lo.first, hi.first + hi.count, n_global, first_global,
                         first_global + n_bodies);
    const size_t n_shapes = mw->shape_radius.size();
    for (uint32_t i = 0; shape_id && i < n_bodies; ++i)
        if (shape_id[i] >= n_shapes)
            return set_error(XPBD_E_INVALID, "xpbd_multi_world_upload: shape_id[%u] = %u >= n_shapes %zu", i, shape_id[i], n_shapes);
    for (uint32_t j = 0; j < n_joints; ++j)
        if (joints[j].body_a >= n_global || joints[j].body_b >= n_global || joints[j].body_a == joints[j].body_b)
            return set_error(XPBD_E_INVALID, "xpbd_multi_world_upload: joint %u links bodies %u and %u of %u", j, joints[j].body_a, joints[j].body_b, n_global);
    mw->n_global = n_global, mw->first_global = first_global, mw->n_bodies = n_bodies;
    mw->joints.assign(joints, joints + n_joints);
    for (Shard &s : mw->shards) {
        s.own = shard_range(n_global, s.rank, mw->n_ranks);
        s.owned.resize((size_t)s.own.count * kRigid);
        s.owned_sid.assign(s.own.count, 0);
        if (s.own.count) {
            std::memcpy(s.owned.data(), bodies + (s.own.first - first_global), (size_t)s.own.count * sizeof(xpbd_rigid));
            if (shape_id)
                std::memcpy(s.owned_sid.data(), shape_id + (s.own.first - first_global), (size_t)s.own.count * 4);
        }
    }
    mw->plans = 0;
    return make_plan(mw);
}

int xpbd_multi_world_replan(xpbd_multi_world *mw)
{
    if (!mw || !mw->planned)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_replan: no bodies uploaded");
    MW_TRY(fetch_owned(mw));
    return make_plan(mw);
}

int xpbd_multi_world_step(xpbd_multi_world *mw, double dt, uint32_t substeps)
{
    if (!mw)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_step: NULL world");
    if (substeps == 0)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_step: substeps must be > 0");
    if (!mw->planned)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_step: no bodies uploaded");
    if (mw->violated)
        return set_error(XPBD_E_HALO, "xpbd_multi_world_step: a body has used up its travel allowance since the halos were planned (%.3g m in "
                                      "margin-equivalent metres, halo_margin %.3g m): remote contacts may have been missed -- call "
                                      "xpbd_multi_world_replan (and re-plan more often or raise the margin)",
                         mw->last_displacement, mw->margin);
    // halo validity, agreed on by all ranks, before anything is stepped
    if (mw->n_ranks > 1) {
        double moved = 0.0;
        MW_TRY(measure_displacement(mw, &moved));
        mw->last_displacement = moved;
        if (!(moved <= mw->margin)) {
            mw->violated = true;
            return set_error(XPBD_E_HALO, "xpbd_multi_world_step: a body has used up its travel allowance since the halos were planned (%.3g m in "
                                          "margin-equivalent metres, beyond halo_margin %.3g m): remote contacts may have been missed in the "
                                          "last frame -- call xpbd_multi_world_replan",
                             moved, mw->margin);
        }
        if ((mw->flags & XPBD_MULTI_AUTO_REPLAN) && moved > 0.5 * mw->margin)
            MW_TRY(xpbd_multi_world_replan(mw));
    }
    const double h = dt / (double)substeps; // src/solver.rs:4
    for (Shard &s : mw->shards)
        if (int rc = xpbd::halo_frame_begin(s.world, dt, h))
            return rc;
    const size_t bytes = (size_t)mw->rows_per_rank() * kDyn * 8;
    auto lists_of = [](Shard &s) {
        return xpbd::HaloLists{s.boundary_slots.as<uint32_t>(), (uint32_t)s.boundary.size(), s.ghost_slots.as<uint32_t>(), s.ghost_rows.as<uint32_t>(),
                               (uint32_t)s.ghosts.size(), s.skip_flags.as<uint8_t>(), s.send.as<double>(), s.recv.as<double>()};
    };
    for (uint32_t k = 0; k < substeps; ++k) {
        const bool last = k + 1 == substeps;
        // 1. the narrowphase, then the boundary bodies: their end-of-substep state lands in the send buffer
        for (Shard &s : mw->shards) {
            if (int rc = xpbd::halo_substep_boundary(s.world, h, k, last, lists_of(s)))
                return rc;
            if (mw->n_ranks > 1) {
                MW_TRY(bind(s));
                MW_HIP_TRY(hipEventRecord(s.ev_ready, s.stream));
                MW_HIP_TRY(hipStreamWaitEvent(s.comm_stream, s.ev_ready, 0));
            }
        }
        // 2. ONE all-gather per substep on the communication streams ...
        if (mw->n_ranks > 1) {
            MW_TRY(all_gather_device(mw, bytes, [](Shard &s) { return s.send.ptr; }, [](Shard &s) { return s.recv.ptr; }, true));
            for (Shard &s : mw->shards) {
                MW_TRY(bind(s));
                MW_HIP_TRY(hipEventRecord(s.ev_gathered, s.comm_stream));
            }
        }
        // 3. ... while the interior bodies (nobody mirrors them) run on the world streams
        for (Shard &s : mw->shards)
            if (int rc = xpbd::halo_substep_interior(s.world, h, k, last, lists_of(s)))
                return rc;
        // 4. the ghosts take their owners' state from the gathered buffer
        if (mw->n_ranks > 1)
            for (Shard &s : mw->shards) {
                MW_TRY(bind(s));
                MW_HIP_TRY(hipStreamWaitEvent(s.stream, s.ev_gathered, 0));
                if (int rc = xpbd::halo_substep_ghosts(s.world, h, k, last, lists_of(s)))
                    return rc;
                // the next substep's boundary launch rewrites the send buffer: not before this exchange has read it
                // (the communication stream is in order, so waiting for ev_gathered above covers the own copy; the peers'
                // reads of OUR buffer are ordered by the transport: RCCL completes the collective, the in-process
                // transport makes the communication streams wait for every peer's ev_recv)
            }
    }
    return XPBD_OK;
}

int xpbd_multi_world_synchronize(xpbd_multi_world *mw)
{
    if (!mw)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_synchronize: NULL world");
    for (Shard &s : mw->shards)
        if (int rc = xpbd_world_synchronize(s.world))
            return rc;
    return XPBD_OK;
}

int xpbd_multi_world_download(xpbd_multi_world *mw, xpbd_rigid *out, uint32_t n)
{
    if (!mw || (n && !out))
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_download: NULL argument");
    if (!mw->planned || n != mw->n_bodies)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_download: n = %u but this process owns %u bodies", n, mw->planned ? mw->n_bodies : 0);
    MW_TRY(fetch_owned(mw));
    for (const Shard &s : mw->shards)
        if (s.own.count)
            std::memcpy(out + (s.own.first - mw->first_global), s.owned.data(), (size_t)s.own.count * sizeof(xpbd_rigid));
    return XPBD_OK;
}

int xpbd_multi_world_halo_stats(xpbd_multi_world *mw, uint64_t out[6], double *max_displacement)
{
    if (!mw || !out)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_halo_stats: NULL argument");
    uint64_t owned = 0, ghosts = 0, boundary = 0;
    for (const Shard &s : mw->shards) {
        owned += s.own.count;
        ghosts += s.ghosts.size();
        boundary += s.boundary.size();
    }
    out[0] = mw->n_global, out[1] = owned, out[2] = ghosts, out[3] = boundary, out[4] = mw->capacity, out[5] = mw->plans;
    if (max_displacement)
        *max_displacement = mw->last_displacement;
    return XPBD_OK;
}

int xpbd_multi_world_contact_stats(xpbd_multi_world *mw, uint64_t out[3])
{
    if (!mw || !out)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_contact_stats: NULL argument");
    out[0] = out[1] = out[2] = 0;
    for (Shard &s : mw->shards) {
        uint64_t one[3] = {0, 0, 0};
        if (int rc = xpbd_world_contact_stats(s.world, one))
            return rc;
        for (int k = 0; k < 3; ++k)
            out[k] += one[k];
    }
    return XPBD_OK;
}

// Diagnostics (host only, no device): the halo plan of one rank from the global cell keys, as make_plan computes it.
int xpbd_halo_plan(const int64_t *cell_keys, uint32_t n_global, uint32_t n_ranks, uint32_t rank, const xpbd_joint *joints, uint32_t n_joints,
                   uint32_t *ghosts, uint32_t *n_ghosts, uint32_t *boundary, uint32_t *n_boundary, uint32_t cap)
{
    if (!cell_keys || !n_ghosts || !n_boundary || n_ranks == 0 || n_ranks > 64 || rank >= n_ranks || (n_joints && !joints) || (cap && (!ghosts || !boundary)))
        return set_error(XPBD_E_INVALID, "xpbd_halo_plan: bad argument");
    HaloPlanner planner;
    planner.n = n_global, planner.w = n_ranks, planner.keys = cell_keys;
    std::vector<uint32_t> g, b;
    planner.plan_rank(rank, joints, n_joints, g, b);
    *n_ghosts = (uint32_t)g.size(), *n_boundary = (uint32_t)b.size();
    if (g.size() > cap || b.size() > cap)
        return set_error(XPBD_E_CAPACITY, "xpbd_halo_plan: %zu ghosts, %zu boundary bodies, capacity %u", g.size(), b.size(), cap);
    std::copy(g.begin(), g.end(), ghosts);
    std::copy(b.begin(), b.end(), boundary);
    return XPBD_OK;
}

int xpbd_halo_plan_far(const int64_t *cell_keys, uint32_t n_global, uint32_t n_ranks, uint32_t rank, uint8_t *far, uint32_t cap, uint32_t *n_owned)
{
    if (!cell_keys || !n_owned || n_ranks == 0 || n_ranks > 64 || rank >= n_ranks || (cap && !far))
        return set_error(XPBD_E_INVALID, "xpbd_halo_plan_far: bad argument");
    HaloPlanner planner;
    planner.n = n_global, planner.w = n_ranks, planner.keys = cell_keys;
    std::vector<uint32_t> g, b;
    std::vector<uint8_t> f;
    planner.plan_rank(rank, nullptr, 0, g, b, &f);
    *n_owned = (uint32_t)f.size();
    if (f.size() > cap)
        return set_error(XPBD_E_CAPACITY, "xpbd_halo_plan_far: %zu owned bodies, capacity %u", f.size(), cap);
    std::copy(f.begin(), f.end(), far);
    return XPBD_OK;
}

int64_t xpbd_halo_cell_key(const double centre[3], double cell_edge)
{
    if (!centre || !(cell_edge > 0.0))
        return 0;
    return cell_key(clamp_cell(std::floor(centre[0] / cell_edge)), clamp_cell(std::floor(centre[1] / cell_edge)), clamp_cell(std::floor(centre[2] / cell_edge)));
}

} // extern "C"
