// xpbd_multi.cpp -- the multi-GPU world behind the C ABI (xpbd_multi_world_* in include/xpbd.h).
//
// EXTENSION: the reference is single-threaded and has neither body-body contacts nor any
// multi-device path; its caller is World::integrate (src/world.rs:34-43), which this replaces for an N-body world whose
// bodies are sharded over the GPUs of one node.  Parity: sharded == single device, bit for bit (tests).
//
// One xpbd_multi_world drives the LOCAL shards of a world of n_ranks shards -- all of them (a single process that owns
// every GPU of the node: what a Rust host would do) or one each (one process per GPU, the ranks of a launcher).
//
// Ownership is the LIBRARY's: the bodies are binned into the cells of a uniform grid (edge = 2 * (largest bounding radius +
// pad + halo_margin)), the cells are ordered by their spatial-hash cell key taken along the LONGEST axis of the world first,
// and that sequence is cut into n_ranks runs of near-equal body count -- every rank owns a slab of space across the world's
// longest axis, whatever order the caller numbered its bodies in.  The first plan cuts the slabs (a FULL plan: every rank
// sees 16 bytes per body of the world); the re-plans keep the cuts, move the bodies that crossed one to their new owner and
// exchange only the RIMS of the shards (LIGHT plans: make_plan_light), until a shard is a tenth of a share out of balance --
// then the slabs are cut anew.  The bodies themselves stay on the devices through every plan.  A shard is an ordinary
// xpbd_world in XPBD_MODE_CONTACTS holding its OWNED bodies plus GHOST copies of the remote bodies that can reach an owned
// body before the next plan, in ascending GLOBAL id (the caller's numbering) -- so every neighbour list and every
// floating-point sum has the order of the single-device run and the result is that run's, bit for bit.  Per substep:
//     narrowphase -> boundary bodies (their end-of-substep state straight into the send buffer) -> ONE all-gather (RCCL over
//     xGMI: ncclAllGather, all local shards in one group call) on a communication stream, overlapping the interior bodies ->
//     the ghosts take their owners' state from the gathered buffer.
// The plan (who owns, who mirrors whom) is built HERE, in C++, from the shards' own bodies plus a handful of small
// all-gathers (cell keys or rims, boundary lists, the records of migrating and boundary bodies); no rank holds the global scene.
// Every plan-time all-gather carries a status word per rank and so does the frame's last one, so a rank that fails locally
// (out of memory, say) still takes part in the collectives and EVERY rank returns an error instead of the others hanging.
//
// Halo validity is checked at the END of every frame, over all ranks: if a body has used up its travel allowance since the
// plan, a remote contact may have been missed IN THAT FRAME -- the frame is undone (the state it started from is kept aside
// on the device), and either re-run after a re-plan (XPBD_MULTI_AUTO_REPLAN) or reported as XPBD_E_HALO with the state of
// the frame's start in place.  A state with possibly missed contacts never reaches the caller.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
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
constexpr uint32_t kRecord = 39; // a body's plan-time record: its xpbd_rigid + the shape id
constexpr int64_t kCellBias = 1 << 20;
constexpr int64_t kCellLimit = kCellBias - 4; // |cell| <= this: the +-2 dilations of the planner stay inside the 21-bit fields

struct Range {
    uint32_t first, count;
};

// The index slice of the caller's bodies a rank HANDS OVER at upload: contiguous ranges, the first n % w ranks one body
// longer (constraint_solver_amd/sharding.py).  It says nothing about ownership.
Range shard_range(uint32_t n, uint32_t rank, uint32_t w)
{
    const uint32_t base = n / w, extra = n % w;
    return Range{rank * base + std::min(rank, extra), base + (rank < extra ? 1u : 0u)};
}

int64_t clamp_cell(double q)
{
    const double lim = (double)kCellLimit;
    if (!(q >= -lim)) // NaN or far negative
        return -kCellLimit;
    return q > lim ? kCellLimit : (int64_t)q;
}

// x-major: ascending keys are slabs along x, inside a slab rows along y, inside a row columns along z
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

// ---- ownership: the x-major sequence of grid cells cut into n_ranks runs of near-equal body count ----------------------------
// A cut is a (cell key, body id) pair; rank r owns the bodies whose (key, id) lies in [cut[r], cut[r + 1]).  Cuts fall on
// cell boundaries (whole cells stay together) unless that would leave a rank more than a quarter of its share off balance
// (many bodies in one cell: a tiny world), in which case the cell is split by body id.
struct Cut {
    int64_t key;
    uint32_t id;
    bool operator<=(const Cut &o) const { return key < o.key || (key == o.key && id <= o.id); }
};

// The planner's passes over ALL bodies of the world (every rank makes them at every re-plan) in a few host threads.
// fn(thread, begin, end) for contiguous chunks in thread order; small inputs stay on the calling thread.
constexpr unsigned kPlanThreads = 8;

unsigned plan_threads(size_t n)
{
    static const unsigned hw = [] { // XPBD_PLAN_THREADS=<1..8> overrides (1: everything on the calling thread)
        const char *e = std::getenv("XPBD_PLAN_THREADS");
        const unsigned want = e ? (unsigned)std::atoi(e) : std::thread::hardware_concurrency();
        return std::max(1u, std::min(kPlanThreads, want));
    }();
    return n < ((size_t)1 << 16) ? 1u : hw;
}

template <class F>
void parallel_chunks(size_t n, F fn)
{
    const unsigned t_count = plan_threads(n);
    if (t_count == 1) {
        fn(0u, (size_t)0, n);
        return;
    }
    std::vector<std::thread> threads;
    threads.reserve(t_count - 1);
    for (unsigned t = 1; t < t_count; ++t)
        threads.emplace_back([&fn, t, t_count, n] { fn(t, n * t / t_count, n * (t + 1) / t_count); });
    fn(0u, (size_t)0, n / t_count);
    for (std::thread &th : threads)
        th.join();
}

// The slabs are cut ACROSS THE LONGEST AXIS of the world's box of cells (a world 64 cells by 256 gets four slabs of 64 x 64,
// not of 16 x 256: a quarter of the boundary): `order` = the axes by falling extent (ties: x, y, z), and the bodies are
// sequenced by their cell key re-packed with the axes in that order.  lo / hi: the box.
void slab_axes(const int64_t *keys, uint32_t n, int order[3], int64_t lo[3], int64_t hi[3])
{
    int64_t tlo[kPlanThreads][3], thi[kPlanThreads][3];
    for (unsigned t = 0; t < kPlanThreads; ++t)
        for (int a = 0; a < 3; ++a)
            tlo[t][a] = INT64_MAX, thi[t][a] = INT64_MIN;
    parallel_chunks(n, [&](unsigned t, size_t begin, size_t end) {
        int64_t l[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, h[3] = {INT64_MIN, INT64_MIN, INT64_MIN};
        for (size_t g = begin; g < end; ++g) {
            int64_t c[3];
            cell_of_key(keys[g], c);
            for (int a = 0; a < 3; ++a) {
                l[a] = std::min(l[a], c[a]);
                h[a] = std::max(h[a], c[a]);
            }
        }
        for (int a = 0; a < 3; ++a)
            tlo[t][a] = l[a], thi[t][a] = h[a];
    });
    for (int a = 0; a < 3; ++a) {
        lo[a] = INT64_MAX, hi[a] = INT64_MIN;
        for (unsigned t = 0; t < kPlanThreads; ++t) {
            lo[a] = std::min(lo[a], tlo[t][a]);
            hi[a] = std::max(hi[a], thi[t][a]);
        }
    }
    order[0] = 0, order[1] = 1, order[2] = 2;
    if (n == 0)
        return;
    std::stable_sort(order, order + 3, [&](int a, int b) { return hi[a] - lo[a] > hi[b] - lo[b]; });
}

int64_t slab_key(int64_t key, const int order[3])
{
    int64_t c[3];
    cell_of_key(key, c);
    return cell_key(c[order[0]], c[order[1]], c[order[2]]);
}

// Ownership of all bodies from their cell keys: owner[g], and the cuts (cuts[r] for r = 1 .. w - 1 over the SLAB keys,
// cuts[0] = the smallest possible pair); a deterministic function of the keys alone.
//   The body sequence (slab key, id) is cut at positions t_r = the start of rank r's equal share.  K_r = the slab key at
// position t_r of the sorted sequence is found without sorting it: a histogram of the bodies per LAYER (the slab key's
// leading coordinate, the one the slabs are cut across) locates the layer position t_r falls in, and only that layer's
// bodies are gathered and sorted (a 1024-layer world: a thousandth of the bodies per cut).
void compute_owners(const int64_t *keys, uint32_t n, uint32_t w, uint8_t *owner, std::vector<Cut> &cuts, int axes[3])
{
    cuts.assign(w, Cut{INT64_MIN, 0});
    axes[0] = 0, axes[1] = 1, axes[2] = 2;
    if (n == 0)
        return;
    int64_t lo[3], hi[3];
    slab_axes(keys, n, axes, lo, hi);
    std::vector<int64_t> slab(n);
    const int64_t layer_lo = lo[axes[0]];
    const size_t n_layers = (size_t)(hi[axes[0]] - layer_lo + 1);
    auto layer_of = [layer_lo](int64_t slab_key_) { return (size_t)(((slab_key_ >> 42) - kCellBias) - layer_lo); };
    const unsigned t_count = plan_threads(n);
    std::vector<std::vector<uint32_t>> hist(t_count, std::vector<uint32_t>(n_layers, 0));
    parallel_chunks(n, [&](unsigned t, size_t begin, size_t end) {
        uint32_t *h = hist[t].data();
        for (size_t g = begin; g < end; ++g) {
            slab[g] = slab_key(keys[g], axes);
            ++h[layer_of(slab[g])];
        }
    });
    if (w >= 2) {
        std::vector<uint64_t> first(n_layers + 1, 0); // bodies in the layers before layer x
        for (size_t x = 0; x < n_layers; ++x) {
            uint64_t c = 0;
            for (unsigned t = 0; t < t_count; ++t)
                c += hist[t][x];
            first[x + 1] = first[x] + c;
        }
        // the layer every cut position falls in; the bodies of those layers
        std::vector<size_t> cut_layer(w, SIZE_MAX);
        std::vector<int32_t> slot_of_layer(n_layers, -1);
        std::vector<std::vector<Cut>> members;
        for (uint32_t r = 1; r < w; ++r) {
            const size_t t = shard_range(n, r, w).first;
            if (t >= n)
                continue;
            const size_t x = (size_t)(std::upper_bound(first.begin(), first.end(), (uint64_t)t) - first.begin()) - 1;
            cut_layer[r] = x;
            if (slot_of_layer[x] < 0) {
                slot_of_layer[x] = (int32_t)members.size();
                members.emplace_back();
            }
        }
        std::vector<std::vector<std::vector<Cut>>> found(t_count, std::vector<std::vector<Cut>>(members.size()));
        parallel_chunks(n, [&](unsigned t, size_t begin, size_t end) {
            for (size_t g = begin; g < end; ++g) {
                const int32_t m = slot_of_layer[layer_of(slab[g])];
                if (m >= 0)
                    found[t][(size_t)m].push_back(Cut{slab[g], (uint32_t)g});
            }
        });
        for (size_t m = 0; m < members.size(); ++m) {
            for (unsigned t = 0; t < t_count; ++t)
                members[m].insert(members[m].end(), found[t][m].begin(), found[t][m].end());
            std::sort(members[m].begin(), members[m].end(), [](const Cut &a, const Cut &b) { return a.key < b.key || (a.key == b.key && a.id < b.id); });
        }
        const uint32_t share = std::max(1u, n / w);
        for (uint32_t r = 1; r < w; ++r) {
            const size_t t = shard_range(n, r, w).first; // bodies the ranks before r should own
            if (t >= n) {
                cuts[r] = Cut{INT64_MAX, UINT32_MAX};
                continue;
            }
            const std::vector<Cut> &layer = members[(size_t)slot_of_layer[cut_layer[r]]];
            const size_t base = (size_t)first[cut_layer[r]];
            const int64_t K = layer[t - base].key;
            const auto key_less = [](const Cut &c, int64_t k) { return c.key < k; };
            const size_t i_less = (size_t)(std::lower_bound(layer.begin(), layer.end(), K, key_less) - layer.begin());
            const size_t i_leq = (size_t)(std::lower_bound(layer.begin(), layer.end(), K + 1, key_less) - layer.begin());
            const size_t less = base + i_less, leq = base + i_leq;
            const size_t before = t - less, after = leq - t; // bodies of cell K on the wrong side if the cut goes before / after it
            if (std::min(before, after) * 4 <= share)
                cuts[r] = before <= after ? Cut{K, 0} : Cut{K + 1, 0};
            else // split cell K: its `before` lowest ids stay with the ranks before r
                cuts[r] = Cut{K, layer[i_less + before].id};
        }
        for (uint32_t r = 1; r < w; ++r) // monotone whatever the snapping did
            if (!(cuts[r - 1] <= cuts[r]))
                cuts[r] = cuts[r - 1];
    }
    const std::vector<Cut> &cuts_ref = cuts;
    parallel_chunks(n, [&](unsigned, size_t begin, size_t end) {
        for (size_t g = begin; g < end; ++g) {
            // number of cuts <= (key, id), minus one; cuts[0] is the smallest pair
            uint32_t l = 0, h = (uint32_t)cuts_ref.size(); // cuts[l] <= pair < cuts[h]
            const Cut me{slab[g], (uint32_t)g};
            while (h - l > 1) {
                const uint32_t mid = (l + h) / 2;
                if (cuts_ref[mid] <= me)
                    l = mid;
                else
                    h = mid;
            }
            owner[g] = (uint8_t)l;
        }
    });
}

uint32_t owner_of(const std::vector<Cut> &cuts, int64_t slab, uint32_t id)
{
    // number of cuts <= (key, id), minus one; cuts[0] is the smallest pair
    uint32_t lo = 0, hi = (uint32_t)cuts.size(); // cuts[lo] <= pair < cuts[hi]
    const Cut me{slab, id};
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (cuts[mid] <= me)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// Which remote bodies a rank mirrors and which of its own bodies the others mirror; a pure function of the cell keys, the
// owners and the joints, so every rank computes consistent plans.
//   ghosts:   remote bodies in a cell within one cell of a cell this rank owns a body in (ascending);
//   boundary: this rank's bodies in a cell within one cell of a cell another rank owns a body in (ascending).
//   A joint between an owned and a remote body puts the remote one among the ghosts and the owned one on the boundary.
//   far (optional, one flag per owned body): no cell within two cells of the body's holds a foreign body.  Such a body
//   may travel halo_margin + edge / 2 before it can meet a body this rank does not mirror (any foreign body starts more
//   than two cell edges away, and 2 * (margin + edge / 2) = edge + 2 * margin is less than the 2 * edge - 2 r - pad the
//   two would have to close), the others halo_margin.
struct HaloPlanner {
    struct Foreign {
        uint32_t id;
        int64_t key;
    };
    struct CrossJoint { // a joint between one of the rank's bodies and a remote one
        uint32_t own_id, remote_id;
    };

    // The plan from LISTS: the rank's bodies (ascending ids) with their cell keys, the foreign bodies that may matter (any
    // superset of those within two cells of the box of the rank's cells), the joints that leave the rank.  Everything hashed
    // lies in the RIM of the rank's box: a per-axis occupancy of foreign cells picks the axis along which the fewest of the
    // rank's layers have a foreign cell within two layers (the axis the slabs are cut across), and own bodies outside those
    // layers -- nearly all of a slab -- are classified without a hash lookup.
    static void plan_lists(const std::vector<uint32_t> &own, const std::vector<int64_t> &own_keys, const std::vector<Foreign> &foreign,
                           const std::vector<CrossJoint> &cross, std::vector<uint32_t> &ghosts, std::vector<uint32_t> &boundary,
                           std::vector<uint8_t> *far)
    {
        const uint32_t n_own = (uint32_t)own.size();
        int64_t lo[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, hi[3] = {INT64_MIN, INT64_MIN, INT64_MIN};
        const unsigned t_count = plan_threads(n_own);
        {
            std::vector<int64_t> box((size_t)t_count * 6);
            parallel_chunks(n_own, [&](unsigned t, size_t begin, size_t end) {
                int64_t l[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, h[3] = {INT64_MIN, INT64_MIN, INT64_MIN};
                for (size_t k = begin; k < end; ++k) {
                    int64_t c[3];
                    cell_of_key(own_keys[k], c);
                    for (int a = 0; a < 3; ++a) {
                        l[a] = std::min(l[a], c[a]);
                        h[a] = std::max(h[a], c[a]);
                    }
                }
                for (int a = 0; a < 3; ++a)
                    box[(size_t)t * 6 + a] = l[a], box[(size_t)t * 6 + 3 + a] = h[a];
            });
            for (unsigned t = 0; t < t_count; ++t)
                for (int a = 0; a < 3; ++a) {
                    lo[a] = std::min(lo[a], box[(size_t)t * 6 + a]);
                    hi[a] = std::max(hi[a], box[(size_t)t * 6 + 3 + a]);
                }
        }
        // `layers[a]`: per coordinate of the box along axis a, is there a foreign cell?
        std::vector<uint8_t> layers[3];
        int64_t layer0[3] = {0, 0, 0};
        for (int a = 0; a < 3 && n_own; ++a) {
            layer0[a] = lo[a] - 4;
            layers[a].assign((size_t)(hi[a] - lo[a] + 9), 0);
        }
        std::unordered_set<int64_t> foreign_cells;
        std::vector<Foreign> candidates; // foreign bodies inside the box grown by one cell: the possible ghosts
        for (const Foreign &f : foreign) {
            if (!n_own)
                break;
            int64_t c[3];
            cell_of_key(f.key, c);
            bool in2 = true, in1 = true;
            for (int a = 0; a < 3; ++a) {
                in2 = in2 && c[a] >= lo[a] - 2 && c[a] <= hi[a] + 2;
                in1 = in1 && c[a] >= lo[a] - 1 && c[a] <= hi[a] + 1;
            }
            if (!in2)
                continue;
            foreign_cells.insert(f.key);
            for (int a = 0; a < 3; ++a)
                layers[a][(size_t)(c[a] - layer0[a])] = 1;
            if (in1)
                candidates.push_back(f);
        }
        // near1 / near2 [x]: a foreign cell within one / two layers of layer x, along the axis that leaves the smallest share
        // of the box near foreign layers
        int major = 0;
        std::vector<uint8_t> near1, near2;
        double best = 2.0;
        for (int a = 0; a < 3 && n_own; ++a) {
            std::vector<uint8_t> n1(layers[a].size(), 0), n2(layers[a].size(), 0);
            size_t marked = 0;
            for (size_t x = 0; x < layers[a].size(); ++x) {
                for (int d = -2; d <= 2; ++d) {
                    const size_t y = x + (size_t)(d + 2);
                    if (y < 2 || y - 2 >= layers[a].size() || !layers[a][y - 2])
                        continue;
                    n2[x] = 1;
                    if (d >= -1 && d <= 1)
                        n1[x] = 1;
                }
                marked += n2[x] && x >= 4 && x < layers[a].size() - 4;
            }
            const double share = (double)marked / (double)(hi[a] - lo[a] + 1);
            if (share < best) {
                best = share;
                major = a;
                near1.swap(n1);
                near2.swap(n2);
            }
        }
        const int64_t layer_base = layer0[major];
        // this rank's cells next to foreign layers (the rim): per cell, is a foreign body within one cell?  (then all its
        // bodies are boundary bodies)
        std::unordered_map<int64_t, uint8_t> rim_cells;
        std::vector<uint8_t> is_boundary(n_own, 0);
        // (which own bodies lie in layers near foreign ones: a pass over all of them, in a few threads; the hashing below is
        // for those only)
        std::vector<std::vector<uint32_t>> rim_part(t_count), near2_part(t_count);
        parallel_chunks(n_own, [&](unsigned t, size_t begin, size_t end) {
            for (size_t k = begin; k < end; ++k) {
                int64_t c[3];
                cell_of_key(own_keys[k], c);
                const size_t x = (size_t)(c[major] - layer_base);
                if (near1[x])
                    rim_part[t].push_back((uint32_t)k);
                if (near2[x])
                    near2_part[t].push_back((uint32_t)k);
            }
        });
        std::vector<uint32_t> rim_list, near2_list;
        for (unsigned t = 0; t < t_count; ++t) {
            rim_list.insert(rim_list.end(), rim_part[t].begin(), rim_part[t].end());
            near2_list.insert(near2_list.end(), near2_part[t].begin(), near2_part[t].end());
        }
        for (uint32_t k : rim_list) {
            int64_t c[3];
            cell_of_key(own_keys[k], c);
            auto it = rim_cells.find(own_keys[k]);
            if (it == rim_cells.end()) {
                bool seen = false;
                for (int dx = -1; dx <= 1 && !seen; ++dx)
                    for (int dy = -1; dy <= 1 && !seen; ++dy)
                        for (int dz = -1; dz <= 1 && !seen; ++dz)
                            seen = foreign_cells.count(cell_key(c[0] + dx, c[1] + dy, c[2] + dz)) != 0;
                it = rim_cells.emplace(own_keys[k], seen).first;
            }
            is_boundary[k] = it->second;
        }
        // a candidate is a ghost iff an own cell lies within one cell of its cell (such an own cell is a rim cell)
        std::unordered_map<int64_t, uint8_t> reached; // foreign cell -> within one cell of an own cell (memoised)
        std::vector<uint32_t> ghost_list;
        for (const Foreign &f : candidates) {
            auto it = reached.find(f.key);
            if (it == reached.end()) {
                int64_t c[3];
                cell_of_key(f.key, c);
                bool near = false;
                for (int dx = -1; dx <= 1 && !near; ++dx)
                    for (int dy = -1; dy <= 1 && !near; ++dy)
                        for (int dz = -1; dz <= 1 && !near; ++dz)
                            near = rim_cells.count(cell_key(c[0] + dx, c[1] + dy, c[2] + dz)) != 0;
                it = reached.emplace(f.key, near).first;
            }
            if (it->second)
                ghost_list.push_back(f.id);
        }
        for (const CrossJoint &j : cross) {
            ghost_list.push_back(j.remote_id);
            is_boundary[std::lower_bound(own.begin(), own.end(), j.own_id) - own.begin()] = 1;
        }
        std::sort(ghost_list.begin(), ghost_list.end());
        ghost_list.erase(std::unique(ghost_list.begin(), ghost_list.end()), ghost_list.end());
        ghosts.swap(ghost_list);
        boundary.clear();
        for (uint32_t k = 0; k < n_own; ++k)
            if (is_boundary[k])
                boundary.push_back(own[k]);
        if (far) {
            // an own body with a foreign cell within two cells of its own (or a boundary body) is not far
            far->assign(n_own, 0);
            const size_t limit = 20000; // beyond that many foreign cells around the slab the test is not worth it: nobody is far
            if (n_own && foreign_cells.size() <= limit) {
                std::unordered_map<int64_t, uint8_t> near_cells; // own cell in a layer near foreign ones -> a foreign cell within two cells
                for (uint32_t k = 0; k < n_own; ++k) // (outside those layers: far unless a joint made it a boundary body)
                    (*far)[k] = !is_boundary[k];
                for (uint32_t k : near2_list) {
                    int64_t c[3];
                    cell_of_key(own_keys[k], c);
                    auto it = near_cells.find(own_keys[k]);
                    if (it == near_cells.end()) {
                        bool seen = false;
                        for (int dx = -2; dx <= 2 && !seen; ++dx)
                            for (int dy = -2; dy <= 2 && !seen; ++dy)
                                for (int dz = -2; dz <= 2 && !seen; ++dz)
                                    seen = foreign_cells.count(cell_key(c[0] + dx, c[1] + dy, c[2] + dz)) != 0;
                        it = near_cells.emplace(own_keys[k], seen).first;
                    }
                    (*far)[k] = !it->second && !is_boundary[k];
                }
            }
        }
    }

    // ... and from the cell keys and owners of ALL bodies (a full plan, the host-only diagnostics): two passes over the world
    // (in a few threads, see parallel_chunks) collect the rank's bodies and the foreign bodies within two cells of their box.
    uint32_t n = 0, w = 0;
    const int64_t *keys = nullptr;
    const uint8_t *owner = nullptr; // [n] rank owning body g

    void plan_rank(uint32_t rank, const xpbd_joint *joints, uint32_t n_joints, std::vector<uint32_t> &own, std::vector<uint32_t> &ghosts,
                   std::vector<uint32_t> &boundary, std::vector<uint8_t> *far = nullptr) const
    {
        own.clear();
        std::vector<int64_t> own_keys;
        int64_t lo[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, hi[3] = {INT64_MIN, INT64_MIN, INT64_MIN};
        const unsigned t_count = plan_threads(n);
        {
            std::vector<std::vector<uint32_t>> part(t_count);
            std::vector<int64_t> box((size_t)t_count * 6);
            parallel_chunks(n, [&](unsigned t, size_t begin, size_t end) {
                int64_t l[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, h[3] = {INT64_MIN, INT64_MIN, INT64_MIN};
                std::vector<uint32_t> &mine = part[t];
                for (size_t g = begin; g < end; ++g)
                    if (owner[g] == rank) {
                        mine.push_back((uint32_t)g);
                        int64_t c[3];
                        cell_of_key(keys[g], c);
                        for (int a = 0; a < 3; ++a) {
                            l[a] = std::min(l[a], c[a]);
                            h[a] = std::max(h[a], c[a]);
                        }
                    }
                for (int a = 0; a < 3; ++a)
                    box[(size_t)t * 6 + a] = l[a], box[(size_t)t * 6 + 3 + a] = h[a];
            });
            for (unsigned t = 0; t < t_count; ++t) {
                own.insert(own.end(), part[t].begin(), part[t].end());
                for (int a = 0; a < 3; ++a) {
                    lo[a] = std::min(lo[a], box[(size_t)t * 6 + a]);
                    hi[a] = std::max(hi[a], box[(size_t)t * 6 + 3 + a]);
                }
            }
        }
        own_keys.resize(own.size());
        for (size_t k = 0; k < own.size(); ++k)
            own_keys[k] = keys[own[k]];
        std::vector<Foreign> foreign;
        if (!own.empty()) {
            std::vector<std::vector<Foreign>> part(t_count);
            parallel_chunks(n, [&](unsigned t, size_t begin, size_t end) {
                for (size_t g = begin; g < end; ++g) {
                    if (owner[g] == rank)
                        continue;
                    int64_t c[3];
                    cell_of_key(keys[g], c);
                    bool in2 = true;
                    for (int a = 0; a < 3; ++a)
                        in2 = in2 && c[a] >= lo[a] - 2 && c[a] <= hi[a] + 2;
                    if (in2)
                        part[t].push_back(Foreign{(uint32_t)g, keys[g]});
                }
            });
            for (unsigned t = 0; t < t_count; ++t)
                foreign.insert(foreign.end(), part[t].begin(), part[t].end());
        }
        std::vector<CrossJoint> cross;
        for (uint32_t j = 0; j < n_joints; ++j) {
            const uint32_t a = joints[j].body_a, b = joints[j].body_b;
            const bool own_a = owner[a] == rank, own_b = owner[b] == rank;
            if (own_a && !own_b)
                cross.push_back(CrossJoint{a, b});
            else if (own_b && !own_a)
                cross.push_back(CrossJoint{b, a});
        }
        plan_lists(own, own_keys, foreign, cross, ghosts, boundary, far);
    }
};

struct Shard {
    int device = 0;
    uint32_t rank = 0;
    xpbd_world *world = nullptr;
    hipStream_t stream = nullptr;      // the shard's world stream: every kernel of the shard
    hipStream_t comm_stream = nullptr; // the per-substep halo all-gather, overlapping the interior bodies' kernels
    ncclComm_t comm = nullptr;
    hipEvent_t ev_send = nullptr, ev_recv = nullptr; // in-process transport
    hipEvent_t ev_ready = nullptr, ev_gathered = nullptr; // world stream -> communication stream -> world stream
    // The bodies this shard HOLDS (global ids, ascending): before the first plan the index slice the caller handed over,
    // after a plan the bodies it owns.  Their state lives in the shard's world on the device (slot owned_slots_h[i]) and
    // stays there through re-plans.
    std::vector<uint32_t> held_ids;
    // plan
    std::vector<uint32_t> local_ids, ghosts, boundary; // global ids (ascending): owned + ghost bodies; ghosts; mirrored owned bodies
    std::vector<uint32_t> owned_slots_h;               // local slot of held_ids[i]
    std::vector<uint8_t> far; // per owned body: more than two cells away from every foreign body (larger travel allowance)
    DevBuf boundary_slots, ghost_slots, ghost_rows, owned_slots, skip_flags, disp_scale, send, recv, snapshot, disp, disp_all, stage_send, stage_recv;
    double *disp_host = nullptr;   // pinned, n_ranks x {largest squared fraction of an allowance used, status}
    double *status_host = nullptr; // pinned, this process's status of the frame
};

// One host thread per local shard (a process that owns several GPUs): a frame is ~20 runtime calls per shard and substep, and
// a single thread enqueueing them for 8 GPUs takes longer than the GPUs need to run them (measured on one device with four
// shards: 5.5 ms of enqueueing per 4.4 ms frame, profiles/r03_a_multi_host_time.json).  The threads meet at a barrier only
// where the in-process transport needs every shard's event to have been RECORDED before anybody waits for it.
struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    uint32_t n = 1, waiting = 0;
    uint64_t generation = 0;
    void arrive_and_wait()
    {
        std::unique_lock<std::mutex> lock(m);
        const uint64_t g = generation;
        if (++waiting == n) {
            waiting = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lock, [&] { return generation != g; });
        }
    }
};

struct ShardJob {
    int rc = XPBD_OK;           // first local error of the shard's frame (LocalStatus)
    std::string message;
    int fatal = XPBD_OK;        // a collective could not be enqueued
    std::string fatal_message;
    uint64_t ns_wait_broadphase = 0;
};

struct Workers {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    uint64_t job = 0;
    uint32_t done = 0;
    bool quit = false;
    double dt = 0.0;
    uint32_t substeps = 0;
    std::vector<ShardJob> result;
    Barrier barrier;
};

} // namespace

struct xpbd_multi_world {
    uint32_t n_ranks = 1, first_rank = 0, transport = XPBD_TRANSPORT_RCCL, flags = 0, narrowphase = XPBD_NARROWPHASE_SAT;
    double pad = 0.02, margin = 0.5;
    std::vector<Shard> shards;
    const xpbd::RcclApi *rccl = nullptr;
    bool have_shapes = false, planned = false, violated = false, broken = false;
    std::vector<double> shape_radius, shape_centroid; // per shape: max |vertex - centroid|, centroid xyz
    uint32_t n_global = 0, first_global = 0, n_bodies = 0, capacity = 1;
    std::vector<xpbd_joint> joints;
    std::vector<uint8_t> owner;        // [n_global] as of the last plan
    std::vector<uint32_t> owned_count; // [n_ranks]
    uint64_t plans = 0, rollbacks = 0, migrated = 0, steps = 0, ns_enqueue = 0, ns_wait_broadphase = 0, ns_wait_frame = 0, ns_plan = 0;
    double cell_edge = 0.0;
    double rmax_local = 0.0;          // largest bounding radius (r_shape + |centroid - com|) among the bodies this process handed over
    std::vector<int32_t> slot_of;     // [n_global] scratch of a plan: local slot of a body, -1 outside the shard being built
    std::vector<uint32_t> joint_off, joint_adj; // [n_global + 1], [2 * joints]: the joints at every body (indices into `joints`, ascending)
    std::vector<Cut> cuts;            // the cuts of the last full plan, kept by the light plans between (cuts_valid)
    int cut_axes[3] = {0, 1, 2};      // ... over slab keys packed in this order of the axes
    bool cuts_valid = false, check_plans = false, plan_torn = false;
    uint32_t balance_most = 0, balance_least = 0; // most / fewest bodies of a rank right after the last full plan
    uint64_t full_plans = 0, light_plans = 0;
    double last_displacement = 0.0; // the largest fraction of its travel allowance any body had used at the last check, times halo_margin
    Workers *workers = nullptr;     // one enqueueing thread per local shard (n_local > 1), started by the first step
    bool all_local() const { return shards.size() == n_ranks; }
    bool shortcut() const { return all_local() && !(flags & XPBD_MULTI_PLAN_THROUGH_DEVICE); } // plan-time gathers are memcpys
    uint32_t rows_per_rank() const { return capacity; }
    Shard *local_shard(uint32_t rank) { return rank >= first_rank && rank < first_rank + shards.size() ? &shards[rank - first_rank] : nullptr; }
};

namespace {

uint64_t now_ns()
{
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int bind(const Shard &s)
{
    MW_HIP_TRY(hipSetDevice(s.device));
    return XPBD_OK;
}

int nccl_fail(const xpbd_multi_world *mw, ncclResult_t r, const char *what)
{
    return set_error(XPBD_E_HIP, "%s failed: %s (RCCL from %s)", what, mw->rccl ? mw->rccl->GetErrorString(r) : "?", mw->rccl ? mw->rccl->path : "?");
}

// A collective that could not be enqueued leaves the ranks out of step for good: the communicators are aborted (peers
// blocked in the collective return with an error instead of hanging) and every later call on this world fails.
int transport_broken(xpbd_multi_world *mw, int rc)
{
    const std::string msg = xpbd_last_error();
    mw->broken = true;
    if (mw->rccl && mw->rccl->CommAbort)
        for (Shard &s : mw->shards)
            if (s.comm) {
                (void)hipSetDevice(s.device);
                (void)mw->rccl->CommAbort(s.comm);
                s.comm = nullptr;
            }
    return set_error(rc, "%s -- the communicator is unusable now: destroy this xpbd_multi_world on every rank", msg.c_str());
}

// One all-gather over all ranks: every local shard contributes `bytes` from its `send` and receives n_ranks x bytes into its
// `recv` (device pointers, picked per shard by the callbacks), ordered on the shards' streams.
// `on_comm_stream`: enqueue on the shards' communication streams (the caller orders them against the world streams with
// events) instead of the world streams.
template <class Send, class Recv>
int all_gather_device_raw(xpbd_multi_world *mw, size_t bytes, Send send_of, Recv recv_of, bool on_comm_stream)
{
    auto stream_of = [on_comm_stream](Shard &s) { return on_comm_stream ? s.comm_stream : s.stream; };
    if (mw->transport == XPBD_TRANSPORT_RCCL) {
        // RCCL reads the thread's last HIP error after its own calls: a stale, harmless one left by somebody else in the
        // process (hipErrorNotReady from an event query, say) would be reported as "unhandled cuda error"
        (void)hipGetLastError();
        ncclResult_t r = mw->rccl->GroupStart();
        if (r != ncclSuccess)
            return nccl_fail(mw, r, "ncclGroupStart");
        int rc = XPBD_OK;
        for (Shard &s : mw->shards) {
            if ((rc = bind(s)) != XPBD_OK)
                break;
            r = mw->rccl->AllGather(send_of(s), recv_of(s), bytes, ncclChar, s.comm, stream_of(s));
            if (r != ncclSuccess) {
                rc = nccl_fail(mw, r, "ncclAllGather");
                break;
            }
        }
        r = mw->rccl->GroupEnd(); // the group is closed whatever happened inside it
        if (rc != XPBD_OK)
            return rc;
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

template <class Send, class Recv>
int all_gather_device(xpbd_multi_world *mw, size_t bytes, Send send_of, Recv recv_of, bool on_comm_stream = false)
{
    if (int rc = all_gather_device_raw(mw, bytes, send_of, recv_of, on_comm_stream))
        return transport_broken(mw, rc);
    return XPBD_OK;
}

// The first local error of a collective operation (a plan, a frame).  The operation goes on taking part in its collectives
// -- with whatever payload -- so that the ranks stay in step, and every collective carries each rank's status: all ranks
// leave the operation with an error at the same point.
struct LocalStatus {
    int rc = XPBD_OK;
    std::string message;
    void keep(int r)
    {
        if (rc == XPBD_OK && r != XPBD_OK) {
            rc = r;
            message = xpbd_last_error();
        }
    }
    bool ok() const { return rc == XPBD_OK; }
    int report() const { return set_error(rc, "%s", message.c_str()); }
};

// Plan-time all-gather of host data: send[k] = `bytes` of local shard k; out = the n_ranks x bytes everybody ends up with.
// Every row is preceded by the sender's status; if any rank reports an error, every rank returns one.
int all_gather_host(xpbd_multi_world *mw, const std::vector<const void *> &send, size_t bytes, std::vector<uint8_t> &out, LocalStatus &status)
{
    const size_t row = bytes + 8;
    std::vector<uint8_t> all((size_t)mw->n_ranks * row, 0);
    const int64_t mine = status.rc;
    if (mw->shortcut()) { // every rank is here: no device round trip needed
        for (size_t k = 0; k < mw->shards.size(); ++k) {
            uint8_t *dst = all.data() + (size_t)mw->shards[k].rank * row;
            std::memcpy(dst, &mine, 8);
            if (bytes && status.ok())
                std::memcpy(dst + 8, send[k], bytes);
        }
    } else {
        std::vector<std::vector<uint8_t>> staged(mw->shards.size());
        int rc = XPBD_OK;
        auto stage = [&]() -> int {
            for (size_t k = 0; k < mw->shards.size(); ++k) {
                Shard &s = mw->shards[k];
                staged[k].assign(row, 0);
                std::memcpy(staged[k].data(), &mine, 8);
                if (bytes && status.ok())
                    std::memcpy(staged[k].data() + 8, send[k], bytes);
                MW_TRY(bind(s));
                MW_HIP_TRY(hipStreamSynchronize(s.stream)); // reserve() may free a block that is still in use
                MW_HIP_TRY(s.stage_send.reserve(row));
                MW_HIP_TRY(s.stage_recv.reserve((size_t)mw->n_ranks * row));
                MW_HIP_TRY(hipMemcpyAsync(s.stage_send.ptr, staged[k].data(), row, hipMemcpyHostToDevice, s.stream));
            }
            return XPBD_OK;
        };
        if ((rc = stage()) != XPBD_OK) // the staging buffers are the transport's: without them this rank cannot take part
            return transport_broken(mw, rc);
        MW_TRY(all_gather_device(mw, row, [](Shard &s) { return s.stage_send.ptr; }, [](Shard &s) { return s.stage_recv.ptr; }));
        auto collect = [&]() -> int {
            for (Shard &s : mw->shards) { // every shard takes part in the collective; the content is the same everywhere
                MW_TRY(bind(s));
                if (&s == &mw->shards[0])
                    MW_HIP_TRY(hipMemcpyAsync(all.data(), s.stage_recv.ptr, all.size(), hipMemcpyDeviceToHost, s.stream));
                MW_HIP_TRY(hipStreamSynchronize(s.stream));
            }
            return XPBD_OK;
        };
        if ((rc = collect()) != XPBD_OK)
            return transport_broken(mw, rc);
    }
    out.assign((size_t)mw->n_ranks * bytes, 0);
    int64_t peer_rc = 0;
    uint32_t peer = 0;
    for (uint32_t r = 0; r < mw->n_ranks; ++r) {
        int64_t st = 0;
        std::memcpy(&st, all.data() + (size_t)r * row, 8);
        if (st != 0 && peer_rc == 0) {
            peer_rc = st;
            peer = r;
        }
        if (bytes)
            std::memcpy(out.data() + (size_t)r * bytes, all.data() + (size_t)r * row + 8, bytes);
    }
    if (!status.ok())
        return status.report();
    if (peer_rc != 0)
        return set_error((int)peer_rc, "xpbd_multi_world: rank %u failed with error %d in this collective call (see its xpbd_last_error); "
                                       "every rank leaves the call with that error", peer, (int)peer_rc);
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

struct KeyRow {
    int64_t key;
    uint32_t id, unused;
};

// XPBD_MULTI_TRACE_PLAN=1: the host time of every phase of a plan on stderr
struct PlanTrace {
    uint64_t plan = 0, t_last = 0;
    bool on = false;
    explicit PlanTrace(uint64_t plan_index) : plan(plan_index), t_last(now_ns())
    {
        static const bool trace = std::getenv("XPBD_MULTI_TRACE_PLAN") != nullptr;
        on = trace;
    }
    void lap(const char *what)
    {
        if (on) {
            const uint64_t t = now_ns();
            std::fprintf(stderr, "[xpbd plan %llu] %-28s %8.3f ms\n", (unsigned long long)plan, what, (double)(t - t_last) * 1e-6);
            t_last = t;
        }
    }
};

// The new plan of one local shard, besides Shard::ghosts / boundary / far: what it will own (ascending ids) and who holds
// those bodies now, who owns and who holds its ghosts, what it hands out (bodies that change owner, bodies others mirror).
struct ShardPlan {
    std::vector<uint32_t> own, exports;
    std::vector<uint8_t> own_holder, ghost_owner, ghost_holder;
    std::vector<uint32_t> ghosts, boundary; // (they replace Shard::ghosts / boundary / far only when the shard has been re-packed:
    std::vector<uint8_t> far;               //  a plan that fails before that leaves the old plan as it was)
};

// The cell keys of the bodies every local shard holds, computed where the bodies are (8 bytes per body come back).
int held_cell_keys(xpbd_multi_world *mw, LocalStatus &st, double edge, std::vector<std::vector<int64_t>> &held_keys)
{
    held_keys.assign(mw->shards.size(), std::vector<int64_t>());
    for (size_t k = 0; k < mw->shards.size(); ++k) {
        Shard &s = mw->shards[k];
        const uint32_t cnt = (uint32_t)s.held_ids.size();
        held_keys[k].assign(cnt, 0);
        uint32_t bad = UINT32_MAX;
        if (st.ok())
            st.keep(xpbd::halo_cell_keys(s.world, s.owned_slots.as<uint32_t>(), cnt, edge, held_keys[k].data(), &bad));
        if (st.ok() && bad != UINT32_MAX)
            st.keep(set_error(XPBD_E_INVALID, "xpbd_multi_world: body %u has a non-finite position: it cannot be placed in the grid the shards "
                                              "are cut from", s.held_ids[bad]));
    }
    return XPBD_OK;
}

// One all-gather of 16 bytes per body: the cell key of every body of the world and who holds it.
int gather_world_keys(xpbd_multi_world *mw, LocalStatus &st, const std::vector<std::vector<int64_t>> &held_keys, uint64_t max_held,
                      std::vector<int64_t> &keys, std::vector<uint8_t> &holder)
{
    const uint32_t n = mw->n_global, w = mw->n_ranks;
    const size_t n_local = mw->shards.size();
    std::vector<std::vector<KeyRow>> key_rows(n_local);
    std::vector<const void *> send(n_local);
    std::vector<uint8_t> gathered;
    for (size_t k = 0; k < n_local; ++k) {
        const Shard &s = mw->shards[k];
        key_rows[k].assign(max_held, KeyRow{0, UINT32_MAX, 0});
        for (size_t i = 0; i < s.held_ids.size() && st.ok(); ++i)
            key_rows[k][i] = KeyRow{held_keys[k][i], s.held_ids[i], 0};
        send[k] = key_rows[k].data();
    }
    MW_TRY(all_gather_host(mw, send, (size_t)max_held * sizeof(KeyRow), gathered, st));
    keys.assign(n, 0);
    holder.assign(n, 0xFF);
    for (uint32_t r = 0; r < w && st.ok(); ++r) {
        const KeyRow *rows = reinterpret_cast<const KeyRow *>(gathered.data() + (size_t)r * max_held * sizeof(KeyRow));
        for (uint64_t i = 0; i < max_held; ++i) {
            if (rows[i].id == UINT32_MAX)
                break;
            if (rows[i].id >= n || holder[rows[i].id] != 0xFF) {
                st.keep(set_error(XPBD_E_INVALID, "xpbd_multi_world: body %u is held twice or out of range (rank %u)", rows[i].id, r));
                break;
            }
            keys[rows[i].id] = rows[i].key;
            holder[rows[i].id] = (uint8_t)r;
        }
    }
    return XPBD_OK;
}

// The second half of every plan: the boundary lists of all ranks fix the rows of the per-substep all-gather, the records of
// the bodies that change hands or are mirrored travel, and every shard's local world is re-packed on its device.  Collective.
int finish_plan(xpbd_multi_world *mw, LocalStatus &st, std::vector<ShardPlan> &plans, double edge, PlanTrace &trace)
{
    const uint32_t n = mw->n_global, w = mw->n_ranks;
    const size_t n_local = mw->shards.size();
    std::vector<uint8_t> gathered;
    std::vector<const void *> send(n_local);
    // 4. the boundary lists of all ranks (ascending global ids) fix the rows of the per-substep all-gather; the export lists
    //    those of the plan-time record exchange
    struct Counts {
        uint32_t boundary, exports;
    };
    std::vector<Counts> mine(n_local), counts(w);
    for (size_t k = 0; k < n_local; ++k) {
        mine[k] = Counts{(uint32_t)plans[k].boundary.size(), (uint32_t)plans[k].exports.size()};
        send[k] = &mine[k];
    }
    MW_TRY(all_gather_host(mw, send, sizeof(Counts), gathered, st));
    std::memcpy(counts.data(), gathered.data(), (size_t)w * sizeof(Counts));
    uint32_t cap = 1, cap_exp = 0;
    for (uint32_t r = 0; r < w; ++r) {
        cap = std::max(cap, counts[r].boundary);
        cap_exp = std::max(cap_exp, counts[r].exports);
    }
    std::vector<uint32_t> lists((size_t)w * cap);
    {
        std::vector<std::vector<uint32_t>> pad_list(n_local);
        for (size_t k = 0; k < n_local; ++k) {
            pad_list[k].assign(cap, 0xFFFFFFFFu);
            if (st.ok())
                std::copy(plans[k].boundary.begin(), plans[k].boundary.end(), pad_list[k].begin());
            send[k] = pad_list[k].data();
        }
        MW_TRY(all_gather_host(mw, send, (size_t)cap * 4, gathered, st));
        std::memcpy(lists.data(), gathered.data(), lists.size() * 4);
    }
    // the records of the exported bodies: gathered on the device that holds them, 39 doubles each
    std::vector<uint32_t> exp_lists;
    std::vector<double> exp_records;
    if (cap_exp > 0) {
        std::vector<std::vector<uint32_t>> pad_list(n_local);
        std::vector<std::vector<double>> pad_rec(n_local);
        for (size_t k = 0; k < n_local; ++k) {
            pad_list[k].assign(cap_exp, 0xFFFFFFFFu);
            if (st.ok())
                std::copy(plans[k].exports.begin(), plans[k].exports.end(), pad_list[k].begin());
            send[k] = pad_list[k].data();
        }
        MW_TRY(all_gather_host(mw, send, (size_t)cap_exp * 4, gathered, st));
        exp_lists.resize((size_t)w * cap_exp);
        std::memcpy(exp_lists.data(), gathered.data(), exp_lists.size() * 4);
        std::vector<uint32_t> slots;
        for (size_t k = 0; k < n_local; ++k) {
            Shard &s = mw->shards[k];
            pad_rec[k].assign((size_t)cap_exp * kRecord, 0.0);
            slots.clear();
            if (st.ok())
                for (uint32_t g : plans[k].exports)
                    slots.push_back(s.owned_slots_h[std::lower_bound(s.held_ids.begin(), s.held_ids.end(), g) - s.held_ids.begin()]);
            if (st.ok())
                st.keep(xpbd::download_records(s.world, slots.data(), (uint32_t)slots.size(), pad_rec[k].data()));
            send[k] = pad_rec[k].data();
        }
        MW_TRY(all_gather_host(mw, send, (size_t)cap_exp * kRecord * 8, gathered, st));
        exp_records.resize((size_t)w * cap_exp * kRecord);
        std::memcpy(exp_records.data(), gathered.data(), exp_records.size() * 8);
    }
    trace.lap("lists, exported records");

    // 5. every shard's local world: owned + ghost bodies in ascending global id.  A body the shard owned before and still
    //    owns moves on the device; a body that arrives (a new owner, a ghost) comes from its holder's exported record.
    const uint32_t rows = cap; // (rows per rank of the per-substep all-gather from now on: mw->capacity, set when the plan has succeeded)
    auto exported = [&](const Shard &me, uint32_t g, uint32_t h) -> const double * {
        const uint32_t *lo = exp_lists.data() + (size_t)h * cap_exp, *hi = h < w ? lo + counts[h].exports : lo;
        const uint32_t *at = h < w ? std::lower_bound(lo, hi, g) : hi;
        if (at == hi || *at != g) {
            (void)set_error(XPBD_E_HIP, "xpbd_multi_world: body %u is needed by rank %u but not exported by its holder %u (inconsistent plans)", g, me.rank, h);
            return nullptr;
        }
        return &exp_records[((size_t)h * cap_exp + (size_t)(at - lo)) * kRecord];
    };
    if (mw->slot_of.size() != n)
        mw->slot_of.assign(n, -1);
    auto build_shard = [&](size_t k) -> int {
        Shard &s = mw->shards[k];
        ShardPlan &pl = plans[k];
        MW_TRY(bind(s));
        const uint32_t n_own = (uint32_t)pl.own.size(), n_ghost = (uint32_t)pl.ghosts.size(), n_loc = n_own + n_ghost;
        std::vector<int32_t> src(n_loc);
        std::vector<double> incoming;
        std::vector<uint32_t> ghost_slots(n_ghost), ghost_rows(n_ghost), boundary_slots(pl.boundary.size()), owned_slots(n_own);
        std::vector<uint32_t> local_ids(n_loc);
        uint32_t slot = 0, oi = 0, gi = 0, n_in = 0;
        size_t old = 0; // walks the bodies owned so far (ascending, like pl.own)
        while (oi < n_own || gi < n_ghost) {
            const bool take_own = gi >= n_ghost || (oi < n_own && pl.own[oi] < pl.ghosts[gi]);
            const uint32_t g = take_own ? pl.own[oi] : pl.ghosts[gi];
            local_ids[slot] = g;
            bool here = false;
            if (take_own && pl.own_holder[oi] == s.rank) {
                while (old < s.held_ids.size() && s.held_ids[old] < g)
                    ++old;
                if (old == s.held_ids.size() || s.held_ids[old] != g)
                    return set_error(XPBD_E_HIP, "xpbd_multi_world: body %u is not among the bodies rank %u holds (inconsistent plans)", g, s.rank);
                src[slot] = (int32_t)s.owned_slots_h[old];
                here = true;
            }
            if (!here) {
                const double *rec = exported(s, g, take_own ? pl.own_holder[oi] : pl.ghost_holder[gi]);
                if (!rec)
                    return XPBD_E_HIP;
                incoming.insert(incoming.end(), rec, rec + kRecord);
                src[slot] = -(int32_t)(++n_in);
            }
            if (take_own) {
                owned_slots[oi++] = slot;
            } else {
                const uint32_t o = pl.ghost_owner[gi];
                const uint32_t *lo = &lists[(size_t)(o < w ? o : 0) * cap], *hi = o < w ? lo + counts[o].boundary : lo;
                const uint32_t *at = std::lower_bound(lo, hi, g);
                if (at == hi || *at != g)
                    return set_error(XPBD_E_HIP, "xpbd_multi_world: body %u is mirrored by rank %u but not exported by its owner %u (inconsistent plans)", g, s.rank, o);
                ghost_slots[gi] = slot;
                ghost_rows[gi] = o * rows + (uint32_t)(at - lo);
                ++gi;
            }
            ++slot;
        }
        for (size_t q = 0; q < pl.boundary.size(); ++q)
            boundary_slots[q] = owned_slots[std::lower_bound(pl.own.begin(), pl.own.end(), pl.boundary[q]) - pl.own.begin()];
        trace.lap("  source map of a shard");
        if (int rc = xpbd::repack_bodies(s.world, src.data(), n_loc, incoming.data(), n_in))
            return rc;
        trace.lap("  re-pack on the device");
        // joints whose two bodies are both present here, in global joint order, re-indexed to local slots: the joints of the
        // local bodies through the per-body joint lists (a joint is taken at its lower-numbered end)
        for (uint32_t q = 0; q < n_loc; ++q)
            mw->slot_of[local_ids[q]] = (int32_t)q;
        std::vector<uint32_t> joint_ids;
        for (uint32_t q = 0; q < n_loc; ++q) {
            const uint32_t g = local_ids[q];
            for (uint32_t e = mw->joint_off[g]; e < mw->joint_off[g + 1]; ++e) {
                const xpbd_joint &j = mw->joints[mw->joint_adj[e]];
                const uint32_t other = j.body_a == g ? j.body_b : j.body_a;
                if (mw->slot_of[other] >= 0 && (g < other || (g == other && j.body_a == g)))
                    joint_ids.push_back(mw->joint_adj[e]);
            }
        }
        std::sort(joint_ids.begin(), joint_ids.end());
        joint_ids.erase(std::unique(joint_ids.begin(), joint_ids.end()), joint_ids.end());
        std::vector<xpbd_joint> local_joints(joint_ids.size());
        for (size_t q = 0; q < joint_ids.size(); ++q) {
            xpbd_joint l = mw->joints[joint_ids[q]];
            l.body_a = (uint32_t)mw->slot_of[l.body_a];
            l.body_b = (uint32_t)mw->slot_of[l.body_b];
            local_joints[q] = l;
        }
        for (uint32_t q = 0; q < n_loc; ++q)
            mw->slot_of[local_ids[q]] = -1;
        if (int rc = xpbd_world_set_joints(s.world, local_joints.data(), (uint32_t)local_joints.size()))
            return rc;
        MW_HIP_TRY(hipStreamSynchronize(s.stream));
        trace.lap("  joints");
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
        std::vector<double> scale(n_own);
        const double near_allow = mw->margin, far_allow = mw->margin + 0.5 * edge;
        const double near_scale = 1.0 / (near_allow * near_allow), far_scale = 1.0 / (far_allow * far_allow);
        for (uint32_t i = 0; i < n_own; ++i)
            scale[i] = pl.far[i] ? far_scale : near_scale;
        MW_TRY(upload_vector(s.disp_scale, scale, s.stream));
        MW_HIP_TRY(s.send.reserve((size_t)rows * kDyn * 8));
        MW_HIP_TRY(s.recv.reserve((size_t)w * rows * kDyn * 8));
        MW_HIP_TRY(hipMemsetAsync(s.send.ptr, 0, (size_t)rows * kDyn * 8, s.stream));
        MW_HIP_TRY(s.snapshot.reserve(std::max<size_t>((size_t)3 * n_own * 8, 8)));
        MW_HIP_TRY(s.disp.reserve(16));
        MW_HIP_TRY(s.disp_all.reserve((size_t)w * 16));
        if (int rc = xpbd_world_snapshot_positions(s.world, s.owned_slots.as<uint32_t>(), n_own, s.snapshot.as<double>()))
            return rc;
        MW_HIP_TRY(hipStreamSynchronize(s.stream)); // the host vectors above go out of scope
        s.owned_slots_h.swap(owned_slots);
        s.local_ids.swap(local_ids);
        s.held_ids.swap(pl.own); // from now on the shard holds what it owns
        s.ghosts.swap(pl.ghosts);
        s.boundary.swap(pl.boundary);
        s.far.swap(pl.far);
        trace.lap("  index lists of a shard");
        return XPBD_OK;
    };
    for (size_t k = 0; k < n_local && st.ok(); ++k) {
        st.keep(build_shard(k));
        if (!st.ok())
            mw->plan_torn = true; // some shards re-packed, this one half-way: there is no plan to go back to (see make_plan)
    }
    // all ranks leave the plan together: a last status-only exchange (a failed re-pack on one rank fails the plan everywhere)
    for (size_t k = 0; k < n_local; ++k)
        send[k] = nullptr;
    if (int rc = all_gather_host(mw, send, 0, gathered, st)) {
        mw->plan_torn = true; // (a shard of some rank failed while being re-packed)
        return rc;
    }
    mw->capacity = cap;
    mw->cell_edge = edge;
    mw->planned = true;
    mw->violated = false;
    mw->last_displacement = 0.0;
    ++mw->plans;
    return XPBD_OK;
}

// What a shard holds and somebody else needs: bodies that change owner, and its (remaining) bodies that others mirror.
// held_owner[i] = new owner of held body i; boundary = the shard's NEW boundary list (ascending).
void exports_of(const Shard &s, const std::vector<uint8_t> &held_owner, const std::vector<uint32_t> &boundary, std::vector<uint32_t> &exports)
{
    exports.clear();
    size_t b = 0;
    for (size_t i = 0; i < s.held_ids.size(); ++i) {
        const uint32_t g = s.held_ids[i];
        while (b < boundary.size() && boundary[b] < g)
            ++b;
        if (held_owner[i] != s.rank || (b < boundary.size() && boundary[b] == g))
            exports.push_back(g);
    }
}

double plan_cell_edge(const xpbd_multi_world *mw, double rmax) { return 2.0 * (rmax + mw->pad + mw->margin); }

// (rmax of the world, bodies held per rank): the first collective of every plan
int gather_heads(xpbd_multi_world *mw, LocalStatus &st, double &rmax, uint64_t &max_held)
{
    struct Head {
        double rmax;
        uint64_t count;
    };
    const size_t n_local = mw->shards.size();
    std::vector<Head> head(n_local);
    std::vector<const void *> send(n_local);
    std::vector<uint8_t> gathered;
    for (size_t k = 0; k < n_local; ++k) {
        head[k] = Head{mw->rmax_local, mw->shards[k].held_ids.size()};
        send[k] = &head[k];
    }
    MW_TRY(all_gather_host(mw, send, sizeof(Head), gathered, st));
    rmax = 0.0, max_held = 0;
    uint64_t total_held = 0;
    for (uint32_t r = 0; r < mw->n_ranks; ++r) {
        Head h;
        std::memcpy(&h, gathered.data() + (size_t)r * sizeof(Head), sizeof h);
        rmax = std::max(rmax, h.rmax);
        max_held = std::max(max_held, h.count);
        total_held += h.count;
    }
    const double edge = plan_cell_edge(mw, rmax);
    if (!(edge > 0.0) || !(edge <= 1.0e300))
        st.keep(set_error(XPBD_E_INVALID, "xpbd_multi_world: cell edge %g from radius %g, pad %g, halo_margin %g", edge, rmax, mw->pad, mw->margin));
    if (total_held != mw->n_global)
        st.keep(set_error(XPBD_E_INVALID, "xpbd_multi_world: the ranks hold %llu bodies together, the world has %u", (unsigned long long)total_held,
                          mw->n_global));
    return XPBD_OK;
}

// A FULL plan: ownership re-cut from the cell keys of the whole world (every rank passes over 16 bytes per body of it), halos
// from the global keys and owners.  The first plan, and whenever the sticky cuts of the light plans have drifted out of balance.
// The bodies stay on the device: what crosses the bus is 8 bytes of cell key per owned body, the records of the bodies that
// change hands or are mirrored (a few per cent) and the index lists of the new plan.  Collective.  Local failures are
// carried through the collectives (LocalStatus: `st` may already hold one), so all ranks fail together.
int make_plan_full(xpbd_multi_world *mw, LocalStatus &st, PlanTrace &trace)
{
    const uint32_t n = mw->n_global, w = mw->n_ranks;
    const size_t n_local = mw->shards.size();
    // 1. the largest bounding radius of the whole world (r_shape + |centroid - com|: conservative whatever the rotation;
    //    a property of the bodies, known since the upload) and how many bodies every rank owns
    double rmax = 0.0;
    uint64_t max_held = 0;
    MW_TRY(gather_heads(mw, st, rmax, max_held));
    const double edge = plan_cell_edge(mw, rmax);
    // 2. grid cell of every body of the world (centre = position + center_of_mass)
    std::vector<std::vector<int64_t>> held_keys;
    MW_TRY(held_cell_keys(mw, st, edge, held_keys));
    trace.lap("cell keys (device)");
    std::vector<int64_t> keys;
    std::vector<uint8_t> holder;
    MW_TRY(gather_world_keys(mw, st, held_keys, max_held, keys, holder));
    trace.lap("keys of the world");

    // 3. ownership: the cell sequence (longest axis first) cut into runs of near-equal body count; then who mirrors whom
    std::vector<uint8_t> owner(n, 0);
    std::vector<uint32_t> owned_count(w, 0);
    std::vector<ShardPlan> plans(n_local);
    uint64_t migrated = 0;
    if (st.ok()) {
        compute_owners(keys.data(), n, w, owner.data(), mw->cuts, mw->cut_axes);
        for (uint32_t g = 0; g < n; ++g) {
            ++owned_count[owner[g]];
            migrated += owner[g] != holder[g];
        }
        trace.lap("cuts, owners");
        HaloPlanner planner;
        planner.n = n, planner.w = w, planner.keys = keys.data(), planner.owner = owner.data();
        std::vector<uint8_t> held_owner;
        for (size_t k = 0; k < n_local; ++k) {
            Shard &s = mw->shards[k];
            ShardPlan &pl = plans[k];
            planner.plan_rank(s.rank, mw->joints.data(), (uint32_t)mw->joints.size(), pl.own, pl.ghosts, pl.boundary, &pl.far);
            pl.own_holder.resize(pl.own.size());
            for (size_t i = 0; i < pl.own.size(); ++i)
                pl.own_holder[i] = holder[pl.own[i]];
            pl.ghost_owner.resize(pl.ghosts.size());
            pl.ghost_holder.resize(pl.ghosts.size());
            for (size_t i = 0; i < pl.ghosts.size(); ++i)
                pl.ghost_owner[i] = owner[pl.ghosts[i]], pl.ghost_holder[i] = holder[pl.ghosts[i]];
            held_owner.resize(s.held_ids.size());
            for (size_t i = 0; i < s.held_ids.size(); ++i)
                held_owner[i] = owner[s.held_ids[i]];
            exports_of(s, held_owner, pl.boundary, pl.exports);
        }
    }
    trace.lap("halo plans");
    MW_TRY(finish_plan(mw, st, plans, edge, trace));
    mw->owner.swap(owner);
    mw->owned_count.swap(owned_count);
    mw->migrated = migrated;
    mw->cuts_valid = true;
    mw->balance_most = 0, mw->balance_least = UINT32_MAX;
    for (uint32_t c : mw->owned_count) {
        mw->balance_most = std::max(mw->balance_most, c);
        mw->balance_least = std::min(mw->balance_least, c);
    }
    ++mw->full_plans;
    return XPBD_OK;
}

// ---- the pieces of a light plan that are pure host logic (also behind the diagnostic xpbd_halo_plan_light) -----------------
struct RimRow { // what a holder publishes of a body: its cell key, its id, its new owner
    int64_t key;
    uint32_t id;
    uint8_t owner, pad[3];
};
struct Known { // ... and what everybody then knows of it
    int64_t key;
    uint8_t owner, holder;
};
using KnownMap = std::unordered_map<uint32_t, Known>;
struct JointIndex { // the joints of the world and, per body, the joints it is an end of (ascending joint index)
    const xpbd_joint *joints;
    const uint32_t *off, *adj;
    bool any;
};

std::vector<int64_t> cut_layers_of(const std::vector<Cut> &cuts)
{
    std::vector<int64_t> layers;
    for (size_t r = 1; r < cuts.size(); ++r)
        if (cuts[r].key != INT64_MAX)
            layers.push_back((cuts[r].key >> 42) - kCellBias);
    std::sort(layers.begin(), layers.end());
    layers.erase(std::unique(layers.begin(), layers.end()), layers.end());
    return layers;
}

// The RIM a holder publishes: its bodies within two layers of a cut (no body further from every cut can lie within two cells
// of a foreign cell: a rank's bodies and a foreign body near them sit on opposite sides of a cut layer), the bodies that
// change owner, and the ends of joints that leave the shard (the other end is held elsewhere, or the two ends get different
// owners).  slot_of: [n_global] scratch, -1 everywhere on entry and on return.
void rim_rows_of(uint32_t rank, const std::vector<uint32_t> &held_ids, const std::vector<int64_t> &held_keys, const std::vector<uint8_t> &held_owner,
                 const std::vector<int64_t> &held_slab, const std::vector<int64_t> &cut_layers, const JointIndex &ji, std::vector<int32_t> &slot_of,
                 std::vector<RimRow> &rows)
{
    auto near_a_cut = [&](int64_t slab) {
        const int64_t layer = (slab >> 42) - kCellBias;
        const auto at = std::lower_bound(cut_layers.begin(), cut_layers.end(), layer - 2);
        return at != cut_layers.end() && *at <= layer + 2;
    };
    if (ji.any) // (scratch: where in the held lists a body of this shard sits)
        for (size_t i = 0; i < held_ids.size(); ++i)
            slot_of[held_ids[i]] = (int32_t)i;
    for (size_t i = 0; i < held_ids.size(); ++i) {
        const uint32_t g = held_ids[i];
        bool publish = held_owner[i] != rank || near_a_cut(held_slab[i]);
        for (uint32_t e = ji.any ? ji.off[g] : 0u; ji.any && e < ji.off[g + 1] && !publish; ++e) {
            const xpbd_joint &j = ji.joints[ji.adj[e]];
            const int32_t at = slot_of[j.body_a == g ? j.body_b : j.body_a];
            // the other end lives elsewhere now, or will: this end's owner (or mirror) must learn about both
            publish = at < 0 || held_owner[(size_t)at] != held_owner[i];
        }
        if (publish)
            rows.push_back(RimRow{held_keys[i], g, held_owner[i], {0, 0, 0}});
    }
    if (ji.any)
        for (size_t i = 0; i < held_ids.size(); ++i)
            slot_of[held_ids[i]] = -1;
}

// One rank's new plan from the bodies it holds and everybody's rims: what it will own (the held bodies that stay and the
// published bodies that come to it), its ghosts / boundary / far lists, who owns and holds the ghosts.
int light_rank_plan(uint32_t rank, const std::vector<uint32_t> &held_ids, const std::vector<int64_t> &held_keys, const std::vector<uint8_t> &held_owner,
                    KnownMap &known, const JointIndex &ji, std::vector<int32_t> &slot_of, ShardPlan &pl)
{
    std::vector<std::pair<uint32_t, int64_t>> arriving;
    std::vector<HaloPlanner::Foreign> foreign;
    for (const auto &kv : known) {
        if (kv.second.owner == rank) {
            if (kv.second.holder != rank)
                arriving.emplace_back(kv.first, kv.second.key);
        } else {
            foreign.push_back(HaloPlanner::Foreign{kv.first, kv.second.key});
        }
    }
    std::sort(arriving.begin(), arriving.end());
    std::vector<int64_t> own_keys;
    pl.own.reserve(held_ids.size() + arriving.size());
    own_keys.reserve(held_ids.size() + arriving.size());
    size_t ai = 0;
    for (size_t i = 0; i <= held_ids.size(); ++i) {
        const uint32_t g = i < held_ids.size() ? held_ids[i] : UINT32_MAX;
        for (; ai < arriving.size() && arriving[ai].first < g; ++ai) {
            pl.own.push_back(arriving[ai].first);
            own_keys.push_back(arriving[ai].second);
            pl.own_holder.push_back(known[arriving[ai].first].holder);
        }
        if (i < held_ids.size() && held_owner[i] == rank) {
            pl.own.push_back(g);
            own_keys.push_back(held_keys[i]);
            pl.own_holder.push_back((uint8_t)rank);
        }
    }
    // joints that leave the rank: the other end was published by its holder (or is held here and goes elsewhere)
    std::vector<HaloPlanner::CrossJoint> cross;
    int rc = XPBD_OK;
    if (ji.any) {
        for (uint32_t g : pl.own)
            slot_of[g] = 0; // (scratch: the bodies the shard will own)
        for (uint32_t g : pl.own) {
            for (uint32_t e = ji.off[g]; e < ji.off[g + 1] && rc == XPBD_OK; ++e) {
                const xpbd_joint &j = ji.joints[ji.adj[e]];
                const uint32_t other = j.body_a == g ? j.body_b : j.body_a;
                if (slot_of[other] >= 0)
                    continue;
                if (!known.count(other)) {
                    rc = set_error(XPBD_E_HIP, "xpbd_multi_world: body %u (joint %u) is in nobody's rim (inconsistent plans)", other, ji.adj[e]);
                    break;
                }
                cross.push_back(HaloPlanner::CrossJoint{g, other});
            }
        }
        for (uint32_t g : pl.own)
            slot_of[g] = -1;
    }
    if (rc != XPBD_OK)
        return rc;
    HaloPlanner::plan_lists(pl.own, own_keys, foreign, cross, pl.ghosts, pl.boundary, &pl.far);
    pl.ghost_owner.resize(pl.ghosts.size());
    pl.ghost_holder.resize(pl.ghosts.size());
    for (size_t i = 0; i < pl.ghosts.size(); ++i) {
        const Known &kn = known[pl.ghosts[i]];
        pl.ghost_owner[i] = kn.owner, pl.ghost_holder[i] = kn.holder;
    }
    return XPBD_OK;
}

// A LIGHT plan: the cuts stay where the last full plan put them, so a body's owner follows from its own cell key, and what a
// rank must know of the others is the RIM -- the bodies within two layers of a cut (nobody else can lie within two cells of a
// foreign cell), the bodies that change owner, and the ends of joints that leave their holder.  Host work and traffic are
// proportional to the rank's own bodies plus the rims, not to the world.  Falls back to a full plan (same collectives on
// every rank: the decision is taken from gathered data) when the shards have drifted out of balance.
int make_plan_light(xpbd_multi_world *mw, LocalStatus &st, PlanTrace &trace, bool &done)
{
    done = false;
    const uint32_t n = mw->n_global, w = mw->n_ranks;
    const size_t n_local = mw->shards.size();
    std::vector<uint8_t> gathered;
    std::vector<const void *> send(n_local);
    double rmax = 0.0;
    uint64_t max_held = 0;
    MW_TRY(gather_heads(mw, st, rmax, max_held));
    const double edge = plan_cell_edge(mw, rmax);
    std::vector<std::vector<int64_t>> held_keys;
    MW_TRY(held_cell_keys(mw, st, edge, held_keys));
    trace.lap("cell keys (device)");
    // the new owner of every held body from the sticky cuts; how many bodies every rank sends to every rank
    std::vector<std::vector<uint8_t>> held_owner(n_local);
    std::vector<std::vector<int64_t>> held_slab(n_local);
    std::vector<std::vector<uint32_t>> tally(n_local, std::vector<uint32_t>(w, 0));
    for (size_t k = 0; k < n_local; ++k) {
        const Shard &s = mw->shards[k];
        held_owner[k].assign(s.held_ids.size(), (uint8_t)s.rank);
        held_slab[k].assign(s.held_ids.size(), 0);
        for (size_t i = 0; i < s.held_ids.size() && st.ok(); ++i) {
            held_slab[k][i] = slab_key(held_keys[k][i], mw->cut_axes);
            held_owner[k][i] = (uint8_t)owner_of(mw->cuts, held_slab[k][i], s.held_ids[i]);
            ++tally[k][held_owner[k][i]];
        }
        send[k] = tally[k].data();
    }
    MW_TRY(all_gather_host(mw, send, (size_t)w * 4, gathered, st));
    std::vector<uint32_t> owned_count(w, 0);
    uint64_t migrated = 0;
    for (uint32_t h = 0; h < w; ++h)
        for (uint32_t r = 0; r < w; ++r) {
            uint32_t c;
            std::memcpy(&c, gathered.data() + ((size_t)h * w + r) * 4, 4);
            owned_count[r] += c;
            if (h != r)
                migrated += c;
        }
    {
        // out of balance (a tenth of a share off): time for new cuts
        const uint32_t share = std::max(1u, n / w);
        uint32_t most = 0, least = UINT32_MAX;
        for (uint32_t c : owned_count) {
            most = std::max(most, c);
            least = std::min(least, c);
        }
        // (against what the last full plan achieved: cuts snap to cell boundaries, a world of few cells is never even)
        if (most > mw->balance_most + share / 10 + 8 || least + share / 10 + 8 < mw->balance_least)
            return XPBD_OK; // (done stays false: the caller makes a full plan; every rank decides alike)
    }
    trace.lap("owners from the cuts");
    // the rims: (key, id, new owner) of the held bodies near a cut, changing owner, or at the end of a joint that leaves the shard
    const std::vector<int64_t> cut_layers = cut_layers_of(mw->cuts);
    const JointIndex ji{mw->joints.data(), mw->joint_off.data(), mw->joint_adj.data(), !mw->joints.empty()};
    std::vector<std::vector<RimRow>> rim(n_local);
    if (mw->slot_of.size() != n)
        mw->slot_of.assign(n, -1);
    for (size_t k = 0; k < n_local && st.ok(); ++k)
        rim_rows_of(mw->shards[k].rank, mw->shards[k].held_ids, held_keys[k], held_owner[k], held_slab[k], cut_layers, ji, mw->slot_of, rim[k]);
    std::vector<uint32_t> rim_count(n_local), rim_counts(w);
    for (size_t k = 0; k < n_local; ++k) {
        rim_count[k] = (uint32_t)rim[k].size();
        send[k] = &rim_count[k];
    }
    MW_TRY(all_gather_host(mw, send, 4, gathered, st));
    std::memcpy(rim_counts.data(), gathered.data(), (size_t)w * 4);
    uint32_t cap_rim = 1;
    for (uint32_t c : rim_counts)
        cap_rim = std::max(cap_rim, c);
    for (size_t k = 0; k < n_local; ++k) {
        rim[k].resize(cap_rim, RimRow{0, UINT32_MAX, 0xFF, {0, 0, 0}});
        send[k] = rim[k].data();
    }
    MW_TRY(all_gather_host(mw, send, (size_t)cap_rim * sizeof(RimRow), gathered, st));
    if (trace.on)
        std::fprintf(stderr, "[xpbd plan %llu] rim rows per rank: up to %u\n", (unsigned long long)mw->plans, cap_rim);
    trace.lap("rims of the world");
    // everybody's rim by body id: (key, owner, holder)
    KnownMap known;
    if (st.ok()) {
        size_t total = 0;
        for (uint32_t c : rim_counts)
            total += c;
        known.reserve(total * 2 + 16);
        for (uint32_t h = 0; h < w && st.ok(); ++h) {
            const RimRow *rows = reinterpret_cast<const RimRow *>(gathered.data() + (size_t)h * cap_rim * sizeof(RimRow));
            for (uint32_t i = 0; i < rim_counts[h]; ++i) {
                if (rows[i].id >= n || rows[i].owner >= w || !known.emplace(rows[i].id, Known{rows[i].key, rows[i].owner, (uint8_t)h}).second) {
                    st.keep(set_error(XPBD_E_INVALID, "xpbd_multi_world: body %u is held twice or out of range (rank %u)", rows[i].id, h));
                    break;
                }
            }
        }
    }
    std::vector<ShardPlan> plans(n_local);
    for (size_t k = 0; k < n_local && st.ok(); ++k) {
        Shard &s = mw->shards[k];
        ShardPlan &pl = plans[k];
        st.keep(light_rank_plan(s.rank, s.held_ids, held_keys[k], held_owner[k], known, ji, mw->slot_of, pl));
        if (!st.ok())
            break;
        if (pl.own.size() != owned_count[s.rank]) {
            st.keep(set_error(XPBD_E_HIP, "xpbd_multi_world: rank %u is to own %u bodies but finds %zu (inconsistent plans)", s.rank, owned_count[s.rank],
                              pl.own.size()));
            break;
        }
        exports_of(s, held_owner[k], pl.boundary, pl.exports);
    }
    trace.lap("halo plans (light)");
    if (mw->check_plans) { // XPBD_MULTI_CHECK_PLANS=1: the same lists from the keys of the whole world (the full planner, same cuts)
        std::vector<int64_t> keys;
        std::vector<uint8_t> holder;
        MW_TRY(gather_world_keys(mw, st, held_keys, max_held, keys, holder));
        if (st.ok()) {
            std::vector<uint8_t> owner(n);
            for (uint32_t g = 0; g < n; ++g)
                owner[g] = (uint8_t)owner_of(mw->cuts, slab_key(keys[g], mw->cut_axes), g);
            HaloPlanner planner;
            planner.n = n, planner.w = w, planner.keys = keys.data(), planner.owner = owner.data();
            for (size_t k = 0; k < n_local && st.ok(); ++k) {
                std::vector<uint32_t> own, ghosts, boundary;
                std::vector<uint8_t> far;
                planner.plan_rank(mw->shards[k].rank, mw->joints.data(), (uint32_t)mw->joints.size(), own, ghosts, boundary, &far);
                const ShardPlan &pl = plans[k];
                if (own != pl.own || ghosts != pl.ghosts || boundary != pl.boundary || far != pl.far)
                    st.keep(set_error(XPBD_E_HIP, "xpbd_multi_world: the light plan of rank %u differs from the full planner's (own %zu / %zu, ghosts %zu / %zu, "
                                                  "boundary %zu / %zu)", mw->shards[k].rank, pl.own.size(), own.size(), pl.ghosts.size(), ghosts.size(),
                                      pl.boundary.size(), boundary.size()));
            }
        }
        trace.lap("checked against the full planner");
    }
    MW_TRY(finish_plan(mw, st, plans, edge, trace));
    mw->owner.clear(); // (rebuilt on demand: xpbd_multi_world_owners)
    mw->owned_count.swap(owned_count);
    mw->migrated = migrated;
    ++mw->light_plans;
    done = true;
    return XPBD_OK;
}

// Builds ownership and halos from the bodies the shards own AT THE MOMENT (the first time: the index slices the caller handed
// over, uploaded as they are) and re-packs every shard's local world.  Collective.
int make_plan(xpbd_multi_world *mw, LocalStatus &st)
{
    const uint64_t t_plan = now_ns();
    PlanTrace trace(mw->plans);
    bool done = false;
    int rc = XPBD_OK;
    if (mw->cuts_valid && mw->n_ranks > 1 && !(mw->flags & XPBD_MULTI_FULL_PLANS))
        rc = make_plan_light(mw, st, trace, done);
    if (rc == XPBD_OK && !done)
        rc = make_plan_full(mw, st, trace);
    mw->ns_plan += now_ns() - t_plan;
    if (rc != XPBD_OK && mw->plan_torn && !mw->broken) {
        // A plan that fails BEFORE any shard is re-packed leaves the old plan and the state as they were (every rank returns
        // the error); one that fails in the middle of the re-packing does not: the world cannot be used any more.
        const std::string msg = xpbd_last_error();
        mw->broken = true;
        return set_error(rc, "%s -- the shards were being re-packed: destroy this xpbd_multi_world on every rank", msg.c_str());
    }
    return rc;
}

// The owned bodies' current state (ascending global id, like Shard::held_ids), for a download.
int fetch_owned(xpbd_multi_world *mw, std::vector<std::vector<double>> &held)
{
    held.assign(mw->shards.size(), std::vector<double>());
    for (size_t k = 0; k < mw->shards.size(); ++k) {
        Shard &s = mw->shards[k];
        MW_TRY(bind(s));
        const uint32_t n_loc = (uint32_t)s.local_ids.size();
        std::vector<double> aos((size_t)n_loc * kRigid);
        if (int rc = xpbd_world_download_bodies(s.world, reinterpret_cast<xpbd_rigid *>(aos.data()), n_loc))
            return rc;
        held[k].resize(s.held_ids.size() * kRigid);
        for (size_t i = 0; i < s.held_ids.size(); ++i)
            std::memcpy(&held[k][i * kRigid], &aos[(size_t)s.owned_slots_h[i] * kRigid], kRigid * 8);
    }
    return XPBD_OK;
}

int replan(xpbd_multi_world *mw)
{
    LocalStatus st;
    return make_plan(mw, st);
}

// Enqueues one whole frame on every local shard and, behind it, the end-of-frame exchange: per rank the largest fraction of
// its travel allowance any owned body has used since the plan (squared) and the rank's status.  Local errors are kept in
// `st` (the remaining local launches are skipped, the collectives still run); only a failing collective returns at once.
int enqueue_frame(xpbd_multi_world *mw, double dt, uint32_t substeps, LocalStatus &st)
{
    const bool multi = mw->n_ranks > 1;
    const double h = dt / (double)substeps; // src/solver.rs:4
    const uint64_t t0 = now_ns();
    if (multi)
        for (Shard &s : mw->shards)
            if (st.ok())
                st.keep(xpbd::frame_snapshot_save(s.world));
    // the broadphase of ALL shards is on its way before the host waits for any of them
    for (Shard &s : mw->shards)
        if (st.ok())
            st.keep(xpbd::halo_frame_begin_enqueue(s.world, dt));
    const uint64_t t1 = now_ns();
    for (Shard &s : mw->shards)
        if (st.ok())
            st.keep(xpbd::halo_frame_begin_collect(s.world, h));
    const uint64_t t2 = now_ns();
    mw->ns_wait_broadphase += t2 - t1;
    const size_t bytes = (size_t)mw->rows_per_rank() * kDyn * 8;
    auto lists_of = [](Shard &s) {
        return xpbd::HaloLists{s.boundary_slots.as<uint32_t>(), (uint32_t)s.boundary.size(), s.ghost_slots.as<uint32_t>(), s.ghost_rows.as<uint32_t>(),
                               (uint32_t)s.ghosts.size(), s.skip_flags.as<uint8_t>(), s.send.as<double>(), s.recv.as<double>()};
    };
    auto hip_keep = [&st](hipError_t e, const char *what) {
        if (e != hipSuccess)
            st.keep(set_error(XPBD_E_HIP, "%s failed: %s", what, hipGetErrorString(e)));
    };
    for (uint32_t k = 0; k < substeps; ++k) {
        const bool last = k + 1 == substeps;
        // 1. the narrowphase, then the boundary bodies: their end-of-substep state lands in the send buffer
        for (Shard &s : mw->shards) {
            if (st.ok())
                st.keep(xpbd::halo_substep_boundary(s.world, h, k, last, lists_of(s)));
            if (multi) {
                st.keep(bind(s));
                hip_keep(hipEventRecord(s.ev_ready, s.stream), "hipEventRecord");
                hip_keep(hipStreamWaitEvent(s.comm_stream, s.ev_ready, 0), "hipStreamWaitEvent");
            }
        }
        // 2. ONE all-gather per substep on the communication streams ...
        if (multi) {
            MW_TRY(all_gather_device(mw, bytes, [](Shard &s) { return s.send.ptr; }, [](Shard &s) { return s.recv.ptr; }, true));
            for (Shard &s : mw->shards) {
                st.keep(bind(s));
                hip_keep(hipEventRecord(s.ev_gathered, s.comm_stream), "hipEventRecord");
            }
        }
        // 3. ... while the interior bodies (nobody mirrors them) run on the world streams
        for (Shard &s : mw->shards)
            if (st.ok())
                st.keep(xpbd::halo_substep_interior(s.world, h, k, last, lists_of(s)));
        // 4. the ghosts take their owners' state from the gathered buffer
        if (multi)
            for (Shard &s : mw->shards) {
                st.keep(bind(s));
                hip_keep(hipStreamWaitEvent(s.stream, s.ev_gathered, 0), "hipStreamWaitEvent");
                if (st.ok())
                    st.keep(xpbd::halo_substep_ghosts(s.world, h, k, last, lists_of(s)));
                // the next substep's boundary launch rewrites the send buffer: not before this exchange has read it
                // (the communication stream is in order, so waiting for ev_gathered above covers the own copy; the peers'
                // reads of OUR buffer are ordered by the transport: RCCL completes the collective, the in-process
                // transport makes the communication streams wait for every peer's ev_recv)
            }
    }
    if (multi) {
        // halo validity of THIS frame and the ranks' status, agreed on by all ranks
        for (Shard &s : mw->shards) {
            st.keep(bind(s));
            hip_keep(hipMemsetAsync(s.disp.ptr, 0, 16, s.stream), "hipMemsetAsync");
            if (st.ok())
                st.keep(xpbd_world_max_displacement2(s.world, s.owned_slots.as<uint32_t>(), (uint32_t)s.held_ids.size(), s.snapshot.as<double>(),
                                                     s.disp_scale.as<double>(), s.disp.as<double>()));
        }
        for (Shard &s : mw->shards) { // (after the last local launch that could still fail)
            (void)hipSetDevice(s.device);
            *s.status_host = (double)st.rc;
            hip_keep(hipMemcpyAsync(s.disp.as<double>() + 1, s.status_host, 8, hipMemcpyHostToDevice, s.stream), "hipMemcpyAsync");
        }
        MW_TRY(all_gather_device(mw, 16, [](Shard &s) { return s.disp.ptr; }, [](Shard &s) { return s.disp_all.ptr; }));
        for (Shard &s : mw->shards) {
            (void)hipSetDevice(s.device);
            hip_keep(hipMemcpyAsync(s.disp_host, s.disp_all.ptr, (size_t)mw->n_ranks * 16, hipMemcpyDeviceToHost, s.stream), "hipMemcpyAsync");
        }
    }
    mw->ns_enqueue += (now_ns() - t0) - (t2 - t1);
    return XPBD_OK;
}

// ---- the same frame with one enqueueing thread per local shard -----------------------------------------------------------------
// What thread k does for shard k is what enqueue_frame does for it, in the same order on the same streams; the collectives:
//   RCCL   each thread calls ncclAllGather on its own communicator (one thread per device, no group call);
//   local  record my send event | BARRIER | wait for the peers' send events, copy their rows, record my receive event |
//          BARRIER | wait for the peers' receive events (nobody overwrites a send buffer that is still being read).
// Every thread passes every barrier whatever went wrong (errors only skip the runtime calls), so nobody is left waiting.
void shard_frame(xpbd_multi_world *mw, size_t k, double dt, uint32_t substeps, ShardJob &job, Barrier &bar)
{
    Shard &s = mw->shards[k];
    const bool multi = mw->n_ranks > 1;
    const double h = dt / (double)substeps; // src/solver.rs:4
    LocalStatus st;
    auto hip_keep = [&st](hipError_t e, const char *what) {
        if (e != hipSuccess)
            st.keep(set_error(XPBD_E_HIP, "%s failed: %s", what, hipGetErrorString(e)));
    };
    auto fatal = [&job](int rc) {
        if (job.fatal == XPBD_OK && rc != XPBD_OK) {
            job.fatal = rc;
            job.fatal_message = xpbd_last_error();
        }
    };
    auto fatal_hip = [&](hipError_t e, const char *what) {
        if (e != hipSuccess)
            fatal(set_error(XPBD_E_HIP, "%s failed: %s", what, hipGetErrorString(e)));
    };
    // one all-gather: `bytes` from send_of(me) into my recv at row rank, for every rank
    auto gather = [&](size_t bytes, void *(*send_of)(Shard &), void *recv, hipStream_t stream, hipStream_t (*stream_of)(Shard &)) {
        (void)stream_of;
        if (mw->transport == XPBD_TRANSPORT_RCCL) {
            if (job.fatal == XPBD_OK) {
                (void)hipGetLastError(); // see all_gather_device_raw
                const ncclResult_t r = mw->rccl->AllGather(send_of(s), recv, bytes, ncclChar, s.comm, stream);
                if (r != ncclSuccess)
                    fatal(nccl_fail(mw, r, "ncclAllGather"));
            }
            return;
        }
        if (job.fatal == XPBD_OK)
            fatal_hip(hipEventRecord(s.ev_send, stream), "hipEventRecord");
        bar.arrive_and_wait(); // every shard's send event has been recorded
        if (job.fatal == XPBD_OK) {
            for (Shard &p : mw->shards) {
                if (&p != &s)
                    fatal_hip(hipStreamWaitEvent(stream, p.ev_send, 0), "hipStreamWaitEvent");
                fatal_hip(hipMemcpyAsync(static_cast<char *>(recv) + (size_t)p.rank * bytes, send_of(p), bytes, hipMemcpyDefault, stream), "hipMemcpyAsync");
            }
            fatal_hip(hipEventRecord(s.ev_recv, stream), "hipEventRecord");
        }
        bar.arrive_and_wait(); // every shard's receive event has been recorded
        if (job.fatal == XPBD_OK)
            for (Shard &r : mw->shards)
                if (&r != &s)
                    fatal_hip(hipStreamWaitEvent(stream, r.ev_recv, 0), "hipStreamWaitEvent");
    };
    st.keep(bind(s));
    if (multi && st.ok())
        st.keep(xpbd::frame_snapshot_save(s.world));
    if (st.ok())
        st.keep(xpbd::halo_frame_begin_enqueue(s.world, dt));
    const uint64_t t0 = now_ns();
    if (st.ok())
        st.keep(xpbd::halo_frame_begin_collect(s.world, h));
    job.ns_wait_broadphase = now_ns() - t0;
    const xpbd::HaloLists lists{s.boundary_slots.as<uint32_t>(), (uint32_t)s.boundary.size(), s.ghost_slots.as<uint32_t>(), s.ghost_rows.as<uint32_t>(),
                                (uint32_t)s.ghosts.size(), s.skip_flags.as<uint8_t>(), s.send.as<double>(), s.recv.as<double>()};
    const size_t bytes = (size_t)mw->rows_per_rank() * kDyn * 8;
    for (uint32_t q = 0; q < substeps; ++q) {
        const bool last = q + 1 == substeps;
        if (st.ok())
            st.keep(xpbd::halo_substep_boundary(s.world, h, q, last, lists));
        if (multi) {
            hip_keep(hipEventRecord(s.ev_ready, s.stream), "hipEventRecord");
            hip_keep(hipStreamWaitEvent(s.comm_stream, s.ev_ready, 0), "hipStreamWaitEvent");
            gather(bytes, [](Shard &x) -> void * { return x.send.ptr; }, s.recv.ptr, s.comm_stream, nullptr);
            hip_keep(hipEventRecord(s.ev_gathered, s.comm_stream), "hipEventRecord");
        }
        if (st.ok())
            st.keep(xpbd::halo_substep_interior(s.world, h, q, last, lists));
        if (multi) {
            hip_keep(hipStreamWaitEvent(s.stream, s.ev_gathered, 0), "hipStreamWaitEvent");
            if (st.ok())
                st.keep(xpbd::halo_substep_ghosts(s.world, h, q, last, lists));
        }
    }
    if (multi) {
        hip_keep(hipMemsetAsync(s.disp.ptr, 0, 16, s.stream), "hipMemsetAsync");
        if (st.ok())
            st.keep(xpbd_world_max_displacement2(s.world, s.owned_slots.as<uint32_t>(), (uint32_t)s.held_ids.size(), s.snapshot.as<double>(),
                                                 s.disp_scale.as<double>(), s.disp.as<double>()));
        *s.status_host = (double)st.rc;
        hip_keep(hipMemcpyAsync(s.disp.as<double>() + 1, s.status_host, 8, hipMemcpyHostToDevice, s.stream), "hipMemcpyAsync");
        gather(16, [](Shard &x) -> void * { return x.disp.ptr; }, s.disp_all.ptr, s.stream, nullptr);
        hip_keep(hipMemcpyAsync(s.disp_host, s.disp_all.ptr, (size_t)mw->n_ranks * 16, hipMemcpyDeviceToHost, s.stream), "hipMemcpyAsync");
    }
    job.rc = st.rc;
    job.message = st.message;
}

void worker_main(xpbd_multi_world *mw, size_t k)
{
    Workers &w = *mw->workers;
    uint64_t seen = 0;
    for (;;) {
        double dt;
        uint32_t substeps;
        {
            std::unique_lock<std::mutex> lock(w.m);
            w.cv_job.wait(lock, [&] { return w.quit || w.job != seen; });
            if (w.quit)
                return;
            seen = w.job;
            dt = w.dt, substeps = w.substeps;
        }
        w.result[k] = ShardJob();
        shard_frame(mw, k, dt, substeps, w.result[k], w.barrier);
        {
            std::lock_guard<std::mutex> lock(w.m);
            ++w.done;
        }
        w.cv_done.notify_one();
    }
}

int enqueue_frame_threaded(xpbd_multi_world *mw, double dt, uint32_t substeps, LocalStatus &st)
{
    const uint64_t t0 = now_ns();
    if (!mw->workers) {
        mw->workers = new (std::nothrow) Workers;
        if (!mw->workers)
            return set_error(XPBD_E_OOM, "xpbd_multi_world_step: host allocation failed");
        Workers &w = *mw->workers;
        w.result.resize(mw->shards.size());
        w.barrier.n = (uint32_t)mw->shards.size();
        for (size_t k = 0; k < mw->shards.size(); ++k)
            w.threads.emplace_back(worker_main, mw, k);
    }
    Workers &w = *mw->workers;
    {
        std::unique_lock<std::mutex> lock(w.m);
        w.dt = dt, w.substeps = substeps, w.done = 0;
        ++w.job;
        w.cv_job.notify_all();
        w.cv_done.wait(lock, [&] { return w.done == (uint32_t)w.threads.size(); });
    }
    uint64_t wait = 0;
    for (const ShardJob &j : w.result)
        wait = std::max(wait, j.ns_wait_broadphase);
    mw->ns_wait_broadphase += wait;
    mw->ns_enqueue += (now_ns() - t0) - wait;
    for (const ShardJob &j : w.result)
        if (j.fatal != XPBD_OK)
            return transport_broken(mw, set_error(j.fatal, "%s", j.fatal_message.c_str()));
    for (const ShardJob &j : w.result)
        if (j.rc != XPBD_OK && st.ok()) {
            st.rc = j.rc;
            st.message = j.message;
        }
    return XPBD_OK;
}

void stop_workers(xpbd_multi_world *mw)
{
    if (!mw->workers)
        return;
    {
        std::lock_guard<std::mutex> lock(mw->workers->m);
        mw->workers->quit = true;
    }
    mw->workers->cv_job.notify_all();
    for (std::thread &t : mw->workers->threads)
        t.join();
    delete mw->workers;
    mw->workers = nullptr;
}

// Waits for the frame's last exchange.  *moved = the largest fraction of its travel allowance any owned body of any rank
// has used since the plan, in margin-equivalent metres (x halo_margin): a body next to a shard boundary may travel
// halo_margin, one more than two cells away from every foreign body halo_margin + half a cell edge (HaloPlanner::plan_rank).
// Returns the error all ranks agree on, if any rank had one.
int finish_frame(xpbd_multi_world *mw, LocalStatus &st, double *moved)
{
    const uint64_t t0 = now_ns();
    double worst = 0.0;
    int64_t peer_rc = 0;
    uint32_t peer = 0;
    for (Shard &s : mw->shards) {
        (void)hipSetDevice(s.device);
        const hipError_t e = hipStreamSynchronize(s.stream);
        if (e != hipSuccess)
            return transport_broken(mw, set_error(XPBD_E_HIP, "xpbd_multi_world_step: hipStreamSynchronize failed: %s", hipGetErrorString(e)));
        for (uint32_t r = 0; r < mw->n_ranks; ++r) {
            const double d = s.disp_host[2 * r], code = s.disp_host[2 * r + 1];
            if (d != d) // (the kernel already counts a NaN position as +inf)
                worst = INFINITY;
            else if (d > worst)
                worst = d;
            if (code != 0.0 && peer_rc == 0) {
                peer_rc = (int64_t)code;
                peer = r;
            }
        }
    }
    mw->ns_wait_frame += now_ns() - t0;
    *moved = std::sqrt(worst) * mw->margin; // "margin-equivalent" metres: halo_margin means the allowance is used up
    if (!st.ok())
        return st.report();
    if (peer_rc != 0)
        return set_error((int)peer_rc, "xpbd_multi_world_step: rank %u failed with error %d in this frame (see its xpbd_last_error); the frame "
                                       "is undone on every rank", peer, (int)peer_rc);
    return XPBD_OK;
}

int restore_frame(xpbd_multi_world *mw)
{
    for (Shard &s : mw->shards)
        MW_TRY(xpbd::frame_snapshot_restore(s.world));
    return XPBD_OK;
}

void destroy(xpbd_multi_world *mw)
{
    if (!mw)
        return;
    stop_workers(mw);
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
        if (s.status_host)
            (void)hipHostFree(s.status_host);
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

int check_usable(const xpbd_multi_world *mw, const char *who)
{
    if (!mw)
        return set_error(XPBD_E_INVALID, "%s: NULL world", who);
    if (mw->broken)
        return set_error(XPBD_E_HIP, "%s: a collective of this world failed earlier; destroy it on every rank", who);
    return XPBD_OK;
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
    if (cfg->reserved != 0)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_create: reserved must be 0");
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
    if (cfg->flags & ~(XPBD_MULTI_AUTO_REPLAN | XPBD_MULTI_PLAN_THROUGH_DEVICE | XPBD_MULTI_SERIAL_ENQUEUE | XPBD_MULTI_FULL_PLANS))
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
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&s.disp_host), (size_t)cfg->n_ranks * 16, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&s.status_host), 8, hipHostMallocDefault);
        if (e != hipSuccess)
            return bail(set_error(XPBD_E_HIP, "xpbd_multi_world_create: %s", hipGetErrorString(e)));
    }
    if (mw->transport == XPBD_TRANSPORT_RCCL) {
        ncclUniqueId id;
        std::memcpy(&id, cfg->comm_id, sizeof id);
        (void)hipGetLastError(); // see all_gather_device
        ncclResult_t r = mw->rccl->GroupStart();
        bool device_failed = false;
        int failed_device = 0;
        for (Shard &s : mw->shards) {
            if (r != ncclSuccess)
                break;
            if (hipSetDevice(s.device) != hipSuccess) {
                device_failed = true;
                failed_device = s.device;
                break;
            }
            r = mw->rccl->CommInitRank(&s.comm, (int)mw->n_ranks, id, (int)s.rank);
        }
        const ncclResult_t r_end = mw->rccl->GroupEnd();
        if (device_failed)
            return bail(set_error(XPBD_E_HIP, "xpbd_multi_world_create: hipSetDevice(%d) failed", failed_device));
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
    // validated by the first shard before any shard changes; a later failure leaves the world without shapes (and says so)
    mw->have_shapes = false;
    mw->planned = false;
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
    return XPBD_OK;
}

int xpbd_multi_world_set_max_depenetration_speed(xpbd_multi_world *mw, double speed)
{
    if (!mw)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_set_max_depenetration_speed: NULL world");
    for (Shard &s : mw->shards)
        if (int rc = xpbd_world_set_max_depenetration_speed(s.world, speed))
            return rc;
    return XPBD_OK;
}

int xpbd_multi_world_upload(xpbd_multi_world *mw, const xpbd_rigid *bodies, const uint32_t *shape_id, uint32_t first_global, uint32_t n_bodies,
                            uint32_t n_global, const xpbd_joint *joints, uint32_t n_joints)
{
    MW_TRY(check_usable(mw, "xpbd_multi_world_upload"));
    // Argument errors are found by every rank alike (or are the caller's to agree on): they return before any collective.
    if ((n_bodies && !bodies) || (n_joints && !joints))
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_upload: NULL argument");
    if (!mw->have_shapes)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_upload: call xpbd_multi_world_set_polytopes first");
    const Range lo = shard_range(n_global, mw->first_rank, mw->n_ranks), hi = shard_range(n_global, mw->first_rank + (uint32_t)mw->shards.size() - 1, mw->n_ranks);
    if (first_global != lo.first || n_bodies != hi.first + hi.count - lo.first)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_upload: ranks [%u, %u) of %u hand over bodies [%u, %u) of %u, got [%u, %u)", mw->first_rank,
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
    mw->planned = false;
    mw->cuts_valid = false; // the first plan is a full one
    mw->check_plans = std::getenv("XPBD_MULTI_CHECK_PLANS") != nullptr;
    // the joints at every body (ascending joint index per body): the plans walk the joints of a rank's own bodies only
    mw->joint_off.assign((size_t)n_global + 1, 0);
    for (uint32_t j = 0; j < n_joints; ++j)
        ++mw->joint_off[joints[j].body_a + 1], ++mw->joint_off[joints[j].body_b + 1];
    for (uint32_t g = 0; g < n_global; ++g)
        mw->joint_off[g + 1] += mw->joint_off[g];
    mw->joint_adj.assign((size_t)2 * n_joints, 0);
    {
        std::vector<uint32_t> cursor(mw->joint_off.begin(), mw->joint_off.end() - 1);
        for (uint32_t j = 0; j < n_joints; ++j) {
            mw->joint_adj[cursor[joints[j].body_a]++] = j;
            mw->joint_adj[cursor[joints[j].body_b]++] = j;
        }
    }
    // the largest bounding radius among the bodies handed over here (r_shape + |centroid - com|: a property of the bodies)
    mw->rmax_local = 0.0;
    for (uint32_t i = 0; i < n_bodies; ++i) {
        const uint32_t sid = shape_id ? shape_id[i] : 0u;
        double off2 = 0.0;
        for (int a = 0; a < 3; ++a) {
            const double d = mw->shape_centroid[3 * (size_t)sid + a] - bodies[i].center_of_mass[a];
            off2 += d * d;
        }
        mw->rmax_local = std::max(mw->rmax_local, mw->shape_radius[sid] + std::sqrt(off2));
    }
    // every shard's slice goes to its device as it is: a world of owned bodies only, which the first plan re-packs
    LocalStatus st;
    for (Shard &s : mw->shards) {
        const Range slice = shard_range(n_global, s.rank, mw->n_ranks);
        s.held_ids.resize(slice.count);
        s.owned_slots_h.resize(slice.count);
        for (uint32_t i = 0; i < slice.count; ++i)
            s.held_ids[i] = slice.first + i, s.owned_slots_h[i] = i;
        s.local_ids = s.held_ids;
        s.ghosts.clear(), s.boundary.clear();
        auto upload = [&]() -> int {
            MW_TRY(bind(s));
            if (int rc = xpbd_world_upload_bodies(s.world, bodies + (slice.first - first_global), shape_id ? shape_id + (slice.first - first_global) : nullptr,
                                                  slice.count))
                return rc;
            MW_TRY(upload_vector(s.owned_slots, s.owned_slots_h, s.stream));
            MW_HIP_TRY(hipStreamSynchronize(s.stream));
            return XPBD_OK;
        };
        if (st.ok())
            st.keep(upload());
    }
    mw->plans = 0;
    mw->rollbacks = 0;
    mw->full_plans = mw->light_plans = 0;
    return make_plan(mw, st);
}

int xpbd_multi_world_replan(xpbd_multi_world *mw)
{
    MW_TRY(check_usable(mw, "xpbd_multi_world_replan"));
    if (!mw->planned)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_replan: no bodies uploaded");
    return replan(mw);
}

int xpbd_multi_world_step(xpbd_multi_world *mw, double dt, uint32_t substeps)
{
    MW_TRY(check_usable(mw, "xpbd_multi_world_step"));
    if (substeps == 0)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_step: substeps must be > 0");
    if (!mw->planned)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_step: no bodies uploaded");
    if (mw->violated)
        return set_error(XPBD_E_HALO, "xpbd_multi_world_step: a body used up its travel allowance within the last frame (%.3g m in margin-equivalent "
                                      "metres, halo_margin %.3g m); that frame was undone -- call xpbd_multi_world_replan (and re-plan more often "
                                      "or raise the margin)", mw->last_displacement, mw->margin);
    ++mw->steps;
    for (int attempt = 0;; ++attempt) {
        LocalStatus st;
        if (mw->shards.size() > 1 && !(mw->flags & XPBD_MULTI_SERIAL_ENQUEUE))
            MW_TRY(enqueue_frame_threaded(mw, dt, substeps, st));
        else
            MW_TRY(enqueue_frame(mw, dt, substeps, st));
        if (mw->n_ranks == 1)
            return st.ok() ? XPBD_OK : st.report(); // no ghosts, nothing to validate: asynchronous after the broadphase
        double moved = 0.0;
        if (int rc = finish_frame(mw, st, &moved)) {
            if (mw->broken)
                return rc;
            const std::string msg = xpbd_last_error();
            (void)restore_frame(mw); // best effort: the state of the frame's start, on every rank
            return set_error(rc, "%s", msg.c_str());
        }
        const double gain = std::max(0.0, moved - mw->last_displacement); // what this frame used up of the allowance
        mw->last_displacement = moved;
        if (moved <= mw->margin) {
            // pre-emptive: another frame like this one (and half as much again) would outrun the allowance, so re-plan (and
            // re-balance) from the state just reached.  A wrong guess costs a frame: it is undone and run again below.
            if ((mw->flags & XPBD_MULTI_AUTO_REPLAN) && moved + 1.5 * gain > mw->margin)
                MW_TRY(replan(mw));
            return XPBD_OK;
        }
        // A body outran its allowance during THIS frame: a remote contact may have been missed in it.  Undo the frame.
        MW_TRY(restore_frame(mw));
        ++mw->rollbacks;
        if (!(mw->flags & XPBD_MULTI_AUTO_REPLAN) || attempt == 1) {
            mw->violated = true;
            return set_error(XPBD_E_HALO, "xpbd_multi_world_step: a body used up its travel allowance during this frame (%.3g m in "
                                          "margin-equivalent metres, beyond halo_margin %.3g m%s): remote contacts may have been missed, so the "
                                          "frame was undone -- call xpbd_multi_world_replan and step again%s",
                             moved, mw->margin, attempt ? ", even right after a re-plan" : "",
                             attempt ? " with a larger halo_margin or a shorter frame" : "");
        }
        MW_TRY(replan(mw)); // from the restored state; then the same frame again
    }
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

int xpbd_multi_world_download_owned(xpbd_multi_world *mw, uint32_t *ids, xpbd_rigid *out, uint32_t cap, uint32_t *n_out)
{
    MW_TRY(check_usable(mw, "xpbd_multi_world_download_owned"));
    if (!n_out || (cap && (!ids || !out)))
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_download_owned: NULL argument");
    if (!mw->planned)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_download_owned: no bodies uploaded");
    size_t total = 0;
    for (const Shard &s : mw->shards)
        total += s.held_ids.size();
    *n_out = (uint32_t)total;
    if (total > cap)
        return set_error(XPBD_E_CAPACITY, "xpbd_multi_world_download_owned: %zu owned bodies, capacity %u", total, cap);
    std::vector<std::vector<double>> held;
    MW_TRY(fetch_owned(mw, held));
    size_t at = 0;
    for (size_t k = 0; k < mw->shards.size(); ++k) {
        const Shard &s = mw->shards[k];
        if (s.held_ids.empty())
            continue;
        std::memcpy(ids + at, s.held_ids.data(), s.held_ids.size() * 4);
        std::memcpy(out + at, held[k].data(), s.held_ids.size() * sizeof(xpbd_rigid));
        at += s.held_ids.size();
    }
    return XPBD_OK;
}

int xpbd_multi_world_download(xpbd_multi_world *mw, xpbd_rigid *out, uint32_t n)
{
    MW_TRY(check_usable(mw, "xpbd_multi_world_download"));
    if (n && !out)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_download: NULL argument");
    if (!mw->planned || n != mw->n_bodies)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_download: n = %u but this process handed over %u bodies", n, mw->planned ? mw->n_bodies : 0);
    LocalStatus st;
    std::vector<std::vector<double>> held;
    st.keep(fetch_owned(mw, held));
    const uint32_t lo = mw->first_global, hi = mw->first_global + mw->n_bodies;
    if (mw->shortcut()) { // every owner is here
        if (!st.ok())
            return st.report();
        for (size_t k = 0; k < mw->shards.size(); ++k) {
            const Shard &s = mw->shards[k];
            for (size_t i = 0; i < s.held_ids.size(); ++i)
                std::memcpy(out + (s.held_ids[i] - lo), &held[k][i * kRigid], sizeof(xpbd_rigid));
        }
        return XPBD_OK;
    }
    // one all-gather of every rank's owned bodies (id + state); each process keeps the slice it handed over
    uint32_t cap = 0;
    for (uint32_t c : mw->owned_count)
        cap = std::max(cap, c);
    const size_t row = 4 + (size_t)kRigid * 8, n_local = mw->shards.size();
    std::vector<std::vector<uint8_t>> payload(n_local);
    std::vector<const void *> send(n_local);
    for (size_t k = 0; k < n_local; ++k) {
        const Shard &s = mw->shards[k];
        payload[k].assign((size_t)cap * row, 0xFF);
        for (size_t i = 0; i < s.held_ids.size() && st.ok(); ++i) {
            std::memcpy(&payload[k][i * row], &s.held_ids[i], 4);
            std::memcpy(&payload[k][i * row + 4], &held[k][i * kRigid], kRigid * 8);
        }
        send[k] = payload[k].data();
    }
    std::vector<uint8_t> gathered;
    MW_TRY(all_gather_host(mw, send, (size_t)cap * row, gathered, st));
    uint32_t found = 0;
    for (uint32_t r = 0; r < mw->n_ranks; ++r)
        for (uint32_t i = 0; i < mw->owned_count[r]; ++i) {
            const uint8_t *p = gathered.data() + ((size_t)r * cap + i) * row;
            uint32_t g;
            std::memcpy(&g, p, 4);
            if (g >= lo && g < hi) {
                std::memcpy(out + (g - lo), p + 4, sizeof(xpbd_rigid));
                ++found;
            }
        }
    if (found != mw->n_bodies)
        return set_error(XPBD_E_HIP, "xpbd_multi_world_download: %u of the %u bodies of this process came back", found, mw->n_bodies);
    return XPBD_OK;
}

int xpbd_multi_world_halo_stats(xpbd_multi_world *mw, uint64_t out[6], double *max_displacement)
{
    if (!mw || !out)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_halo_stats: NULL argument");
    uint64_t owned = 0, ghosts = 0, boundary = 0;
    for (const Shard &s : mw->shards) {
        owned += s.held_ids.size();
        ghosts += s.ghosts.size();
        boundary += s.boundary.size();
    }
    out[0] = mw->n_global, out[1] = owned, out[2] = ghosts, out[3] = boundary, out[4] = mw->capacity, out[5] = mw->plans;
    if (max_displacement)
        *max_displacement = mw->last_displacement;
    return XPBD_OK;
}

int xpbd_multi_world_plan_stats(xpbd_multi_world *mw, uint64_t out[12])
{
    if (!mw || !out)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_plan_stats: NULL argument");
    uint32_t lo = UINT32_MAX, hi = 0;
    for (uint32_t c : mw->owned_count) {
        lo = std::min(lo, c);
        hi = std::max(hi, c);
    }
    out[0] = mw->plans, out[1] = mw->rollbacks, out[2] = mw->migrated, out[3] = mw->owned_count.empty() ? 0 : lo, out[4] = hi;
    out[5] = mw->steps, out[6] = mw->ns_enqueue, out[7] = mw->ns_wait_broadphase, out[8] = mw->ns_wait_frame, out[9] = mw->ns_plan;
    out[10] = mw->full_plans, out[11] = mw->light_plans;
    return XPBD_OK;
}

int xpbd_multi_world_owners(xpbd_multi_world *mw, uint8_t *owner, uint32_t n_global)
{
    if (!mw || !owner)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_owners: NULL argument");
    if (!mw->planned || n_global != mw->n_global)
        return set_error(XPBD_E_INVALID, "xpbd_multi_world_owners: n_global = %u but the world has %u bodies", n_global, mw->planned ? mw->n_global : 0);
    if (mw->owner.size() != n_global) {
        // a light plan does not know the owner of every body of the world: ask the ranks what they own (COLLECTIVE then)
        LocalStatus st;
        uint32_t cap = 1;
        for (uint32_t c : mw->owned_count)
            cap = std::max(cap, c);
        const size_t n_local = mw->shards.size();
        std::vector<std::vector<uint32_t>> ids(n_local);
        std::vector<const void *> send(n_local);
        for (size_t k = 0; k < n_local; ++k) {
            ids[k].assign(cap, UINT32_MAX);
            std::copy(mw->shards[k].held_ids.begin(), mw->shards[k].held_ids.end(), ids[k].begin());
            send[k] = ids[k].data();
        }
        std::vector<uint8_t> gathered;
        MW_TRY(all_gather_host(mw, send, (size_t)cap * 4, gathered, st));
        std::vector<uint8_t> all(n_global, 0xFF);
        for (uint32_t r = 0; r < mw->n_ranks; ++r)
            for (uint32_t i = 0; i < mw->owned_count[r]; ++i) {
                uint32_t g;
                std::memcpy(&g, gathered.data() + ((size_t)r * cap + i) * 4, 4);
                if (g < n_global)
                    all[g] = (uint8_t)r;
            }
        mw->owner.swap(all);
    }
    std::memcpy(owner, mw->owner.data(), n_global);
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

// Diagnostics (host only, no device): ownership and halo plans from the global cell keys, as make_plan computes them.
int xpbd_halo_partition(const int64_t *cell_keys, uint32_t n_global, uint32_t n_ranks, uint8_t *owner)
{
    if ((n_global && (!cell_keys || !owner)) || n_ranks == 0 || n_ranks > 64)
        return set_error(XPBD_E_INVALID, "xpbd_halo_partition: bad argument");
    std::vector<Cut> cuts;
    int axes[3];
    compute_owners(cell_keys, n_global, n_ranks, owner, cuts, axes);
    return XPBD_OK;
}

int xpbd_halo_plan_light(const int64_t *keys_at_cut, const int64_t *cell_keys, uint32_t n_global, uint32_t n_ranks, uint32_t rank,
                         const xpbd_joint *joints, uint32_t n_joints, uint8_t *owner_now, uint32_t *own, uint32_t *n_own, uint32_t *ghosts,
                         uint32_t *n_ghosts, uint32_t *boundary, uint32_t *n_boundary, uint8_t *far, uint32_t cap)
{
    if (!keys_at_cut || !cell_keys || !owner_now || !n_own || !n_ghosts || !n_boundary || n_ranks == 0 || n_ranks > 64 || rank >= n_ranks ||
        (n_joints && !joints) || (cap && (!own || !ghosts || !boundary)))
        return set_error(XPBD_E_INVALID, "xpbd_halo_plan_light: bad argument");
    for (uint32_t j = 0; j < n_joints; ++j)
        if (joints[j].body_a >= n_global || joints[j].body_b >= n_global)
            return set_error(XPBD_E_INVALID, "xpbd_halo_plan_light: joint %u names a body out of range", j);
    // the cuts of the last full plan; who holds what since
    std::vector<Cut> cuts;
    int axes[3];
    std::vector<uint8_t> holder(n_global);
    compute_owners(keys_at_cut, n_global, n_ranks, holder.data(), cuts, axes);
    // the joints at every body
    std::vector<uint32_t> joint_off((size_t)n_global + 1, 0), joint_adj((size_t)2 * n_joints, 0);
    for (uint32_t j = 0; j < n_joints; ++j)
        ++joint_off[joints[j].body_a + 1], ++joint_off[joints[j].body_b + 1];
    for (uint32_t g = 0; g < n_global; ++g)
        joint_off[g + 1] += joint_off[g];
    {
        std::vector<uint32_t> cursor(joint_off.begin(), joint_off.end() - 1);
        for (uint32_t j = 0; j < n_joints; ++j) {
            joint_adj[cursor[joints[j].body_a]++] = j;
            joint_adj[cursor[joints[j].body_b]++] = j;
        }
    }
    const JointIndex ji{joints, joint_off.data(), joint_adj.data(), n_joints != 0};
    // every rank: the bodies it holds, their owners from the kept cuts, its rim
    const std::vector<int64_t> cut_layers = cut_layers_of(cuts);
    std::vector<int32_t> slot_of(n_global, -1);
    std::vector<std::vector<uint32_t>> held_ids(n_ranks);
    std::vector<std::vector<int64_t>> held_keys(n_ranks), held_slab(n_ranks);
    std::vector<std::vector<uint8_t>> held_owner(n_ranks);
    for (uint32_t g = 0; g < n_global; ++g) {
        const uint32_t h = holder[g];
        const int64_t slab = slab_key(cell_keys[g], axes);
        owner_now[g] = (uint8_t)owner_of(cuts, slab, g);
        held_ids[h].push_back(g);
        held_keys[h].push_back(cell_keys[g]);
        held_slab[h].push_back(slab);
        held_owner[h].push_back(owner_now[g]);
    }
    KnownMap known;
    for (uint32_t h = 0; h < n_ranks; ++h) {
        std::vector<RimRow> rows;
        rim_rows_of(h, held_ids[h], held_keys[h], held_owner[h], held_slab[h], cut_layers, ji, slot_of, rows);
        for (const RimRow &r : rows)
            known.emplace(r.id, Known{r.key, r.owner, (uint8_t)h});
    }
    ShardPlan pl;
    if (int rc = light_rank_plan(rank, held_ids[rank], held_keys[rank], held_owner[rank], known, ji, slot_of, pl))
        return rc;
    *n_own = (uint32_t)pl.own.size(), *n_ghosts = (uint32_t)pl.ghosts.size(), *n_boundary = (uint32_t)pl.boundary.size();
    if (pl.own.size() > cap || pl.ghosts.size() > cap || pl.boundary.size() > cap)
        return set_error(XPBD_E_CAPACITY, "xpbd_halo_plan_light: %zu owned, %zu ghosts, %zu boundary bodies, capacity %u", pl.own.size(), pl.ghosts.size(),
                         pl.boundary.size(), cap);
    std::copy(pl.own.begin(), pl.own.end(), own);
    std::copy(pl.ghosts.begin(), pl.ghosts.end(), ghosts);
    std::copy(pl.boundary.begin(), pl.boundary.end(), boundary);
    if (far)
        std::copy(pl.far.begin(), pl.far.end(), far);
    return XPBD_OK;
}

int xpbd_halo_plan_owned(const int64_t *cell_keys, const uint8_t *owner, uint32_t n_global, uint32_t n_ranks, uint32_t rank, const xpbd_joint *joints,
                         uint32_t n_joints, uint32_t *ghosts, uint32_t *n_ghosts, uint32_t *boundary, uint32_t *n_boundary, uint8_t *far, uint32_t cap)
{
    if (!cell_keys || !owner || !n_ghosts || !n_boundary || n_ranks == 0 || n_ranks > 64 || rank >= n_ranks || (n_joints && !joints) ||
        (cap && (!ghosts || !boundary)))
        return set_error(XPBD_E_INVALID, "xpbd_halo_plan_owned: bad argument");
    for (uint32_t g = 0; g < n_global; ++g)
        if (owner[g] >= n_ranks)
            return set_error(XPBD_E_INVALID, "xpbd_halo_plan_owned: owner[%u] = %u of %u ranks", g, owner[g], n_ranks);
    for (uint32_t j = 0; j < n_joints; ++j)
        if (joints[j].body_a >= n_global || joints[j].body_b >= n_global)
            return set_error(XPBD_E_INVALID, "xpbd_halo_plan_owned: joint %u names a body out of range", j);
    HaloPlanner planner;
    planner.n = n_global, planner.w = n_ranks, planner.keys = cell_keys, planner.owner = owner;
    std::vector<uint32_t> own, g, b;
    std::vector<uint8_t> f;
    planner.plan_rank(rank, joints, n_joints, own, g, b, far ? &f : nullptr);
    *n_ghosts = (uint32_t)g.size(), *n_boundary = (uint32_t)b.size();
    if (g.size() > cap || b.size() > cap || (far && f.size() > cap))
        return set_error(XPBD_E_CAPACITY, "xpbd_halo_plan_owned: %zu ghosts, %zu boundary bodies, %zu owned, capacity %u", g.size(), b.size(), own.size(), cap);
    std::copy(g.begin(), g.end(), ghosts);
    std::copy(b.begin(), b.end(), boundary);
    if (far)
        std::copy(f.begin(), f.end(), far); // one flag per owned body, in ascending id
    return XPBD_OK;
}

namespace {
void range_owners(uint32_t n_global, uint32_t n_ranks, std::vector<uint8_t> &owner)
{
    owner.resize(n_global);
    for (uint32_t r = 0; r < n_ranks; ++r) {
        const Range rr = shard_range(n_global, r, n_ranks);
        std::fill(owner.begin() + rr.first, owner.begin() + rr.first + rr.count, (uint8_t)r);
    }
}
} // namespace

// ... with ownership by contiguous index ranges (what a caller that orders its bodies itself would get)
int xpbd_halo_plan(const int64_t *cell_keys, uint32_t n_global, uint32_t n_ranks, uint32_t rank, const xpbd_joint *joints, uint32_t n_joints,
                   uint32_t *ghosts, uint32_t *n_ghosts, uint32_t *boundary, uint32_t *n_boundary, uint32_t cap)
{
    if (n_ranks == 0 || n_ranks > 64)
        return set_error(XPBD_E_INVALID, "xpbd_halo_plan: bad argument");
    std::vector<uint8_t> owner;
    range_owners(n_global, n_ranks, owner);
    return xpbd_halo_plan_owned(cell_keys, owner.data(), n_global, n_ranks, rank, joints, n_joints, ghosts, n_ghosts, boundary, n_boundary, nullptr, cap);
}

int xpbd_halo_plan_far(const int64_t *cell_keys, uint32_t n_global, uint32_t n_ranks, uint32_t rank, uint8_t *far, uint32_t cap, uint32_t *n_owned)
{
    if (!cell_keys || !n_owned || n_ranks == 0 || n_ranks > 64 || rank >= n_ranks || (cap && !far))
        return set_error(XPBD_E_INVALID, "xpbd_halo_plan_far: bad argument");
    std::vector<uint8_t> owner;
    range_owners(n_global, n_ranks, owner);
    HaloPlanner planner;
    planner.n = n_global, planner.w = n_ranks, planner.keys = cell_keys, planner.owner = owner.data();
    std::vector<uint32_t> own, g, b;
    std::vector<uint8_t> f;
    planner.plan_rank(rank, nullptr, 0, own, g, b, &f);
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
