// xpbd_contacts.hip -- body-body contact EXTENSION for gfx950: sphere broadphase on a hashed
// uniform grid and the per-substep contact pipeline
//     integrate + ground contacts  ->  wave-per-pair SAT  ->  Jacobi pair solve + derive.
// Semantics are defined by oracle/xpbd_pairs_oracle.h (op_contacts_step); the reference has no
// body-body contacts, so parity is UNPINNED.  The ground-contact part reuses the reference path
// (xpbd_step.hpp) unchanged.
//
// Determinism: every floating-point sum is formed by ONE lane in a fixed order (neighbour index,
// then manifold point index); atomics are used on integers only (bucket counts, cursors, stats)
// and every order they leave open is removed by a sort.  Results do not depend on launch order,
// on the hash-table size or on how the world is sharded.
#include <cfloat>
#include <type_traits>

#include "xpbd_contacts.h"
#include "xpbd_step.hpp"

namespace xpbd {
namespace {

constexpr uint32_t kBlock = 256;
// Worlds up to this many bodies run the pair solve with one lane per manifold POINT (eight lanes per body): below it
// the GPU is not full and a substep costs the dependent chain of one wave, which this shortens.
#ifndef XPBD_SMALL_WORLD
#define XPBD_SMALL_WORLD 16384
#endif
constexpr uint32_t kSmallWorld = XPBD_SMALL_WORLD;
#ifndef XPBD_PAIR_SOLVE_MIN_WAVES
#define XPBD_PAIR_SOLVE_MIN_WAVES 2
#endif

// Bucket key of a grid cell.  (Tried: a Morton interleave of the coordinates, so that neighbouring cells share cache
// lines.  With equal bits per axis a flat scene -- 144 x 144 x 9 cells -- folds six cells onto one bucket and the
// search slowed down, 178 -> 196 us at 262 144 stacked boxes; the multiplicative hash spreads any extent evenly.)
__device__ __forceinline__ uint32_t cell_hash(int32_t x, int32_t y, int32_t z)
{
    return ((uint32_t)x * 73856093u) ^ ((uint32_t)y * 19349663u) ^ ((uint32_t)z * 83492791u);
}

// ---------------------------------------------------------------------------------------------------
// Exclusive scan of uint32 (three small kernels; 1024 elements per block).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_scan_exclusive(uint32_t v, uint32_t *total)
{
    __shared__ uint32_t wave_sum[kBlock / 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(inc, d, 64);
        if (lane >= d)
            inc += up;
    }
    if (lane == 63)
        wave_sum[wave] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t k = 0; k < kBlock / 64; ++k) {
        const uint32_t ws = wave_sum[k];
        if (k < wave)
            before += ws;
        all += ws;
    }
    __syncthreads();
    *total = all;
    return before + inc - v;
}

__global__ void k_scan_blocks(uint32_t *__restrict__ data, uint32_t n, uint32_t *__restrict__ block_totals)
{
    const uint32_t base = blockIdx.x * (kBlock * 4) + threadIdx.x * 4;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        v[k] = base + k < n ? data[base + k] : 0u;
        sum += v[k];
    }
    uint32_t total;
    uint32_t run = block_scan_exclusive(sum, &total);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        if (base + k < n)
            data[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 0)
        block_totals[blockIdx.x] = total;
}

__global__ void k_scan_totals(uint32_t *__restrict__ block_totals, uint32_t nb)
{
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0)
        carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += kBlock) {
        const uint32_t k = base + threadIdx.x;
        const uint32_t v = k < nb ? block_totals[k] : 0u;
        uint32_t total;
        const uint32_t ex = block_scan_exclusive(v, &total);
        const uint32_t carry = carry_s;
        if (k < nb)
            block_totals[k] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0)
            carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        block_totals[nb] = carry_s;
}

__global__ void k_scan_add(uint32_t *__restrict__ data, uint32_t n, const uint32_t *__restrict__ block_totals, uint32_t nb)
{
    const uint32_t base = blockIdx.x * (kBlock * 4) + threadIdx.x * 4;
    const uint32_t add = block_totals[blockIdx.x];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (base + k < n)
            data[base + k] += add;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        data[n] = block_totals[nb];
}

// ---------------------------------------------------------------------------------------------------
// Broadphase
// ---------------------------------------------------------------------------------------------------
// Bounding sphere of every body: centre = frame * centroid, radius = r_shape + min(|v| dt, r_shape) + pad.
// The clamp keeps one runaway body from inflating the grid cell of everybody (cell edge = 2 * max radius).
// Every workgroup also leaves the largest radius and the bounding box of its centres in c.grid_partials (7 doubles
// per workgroup: no atomics); k_grid reduces them.
__global__ void __launch_bounds__(kBlock) k_bounds(BodyArrays b, PolytopeTables t, const double *__restrict__ shape_radius, double dt,
                                                   double pad, ContactBuffers c)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[7] = {0.0, DBL_MAX, DBL_MAX, DBL_MAX, -DBL_MAX, -DBL_MAX, -DBL_MAX}; // rmax, min xyz, max xyz
    if (i < b.n) {
        const uint32_t sid = b.shape_id[i];
        const double *cc = t.centroids + 3 * (size_t)sid;
        const Vec3 centre = body_frame(b, i) * Vec3{cc[0], cc[1], cc[2]};
        const Vec3 vel = load3(b.dyn, D_VEL, b.stride, i);
        const double rs = shape_radius[sid], travel = length(vel) * dt;
        const double r = rs + (travel < rs ? travel : rs) + pad;
        store3(c.centers, 0, b.stride, i, centre);
        c.radius[i] = r;
        v[0] = (r > 0.0 && r <= DBL_MAX) ? r : 0.0;
        const double xyz[3] = {centre.x, centre.y, centre.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) { // a NaN coordinate opens the box completely: the grid then falls back to hashing
            v[1 + a] = xyz[a] == xyz[a] ? xyz[a] : -DBL_MAX;
            v[4 + a] = xyz[a] == xyz[a] ? xyz[a] : DBL_MAX;
        }
    }
    __shared__ double part[kBlock / 64][7];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        double x = v[q];
        for (uint32_t off = 32; off; off >>= 1) {
            const double o = __shfl_xor(x, off, 64);
            x = (q >= 1 && q <= 3) ? (o < x ? o : x) : (o > x ? o : x);
        }
        if ((threadIdx.x & 63u) == 0)
            part[threadIdx.x >> 6][q] = x;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const uint32_t q = threadIdx.x;
        double x = part[0][q];
        for (uint32_t w = 1; w < kBlock / 64; ++w) {
            const double o = part[w][q];
            x = (q >= 1 && q <= 3) ? (o < x ? o : x) : (o > x ? o : x);
        }
        c.grid_partials[(size_t)blockIdx.x * 7 + q] = x;
    }
}

__device__ __forceinline__ double clamp_cell(double q)
{
    if (!(q >= -1.0e9)) // NaN or far negative
        q = -1.0e9;
    return q > 1.0e9 ? 1.0e9 : q;
}

// One workgroup: reduces the partials of k_bounds to the grid of this broadphase.  Cell edge = 2 * largest radius, so
// overlapping spheres sit in adjacent cells.  When the box of all centres (plus one cell of margin on every side)
// has no more cells than the table has buckets, the bucket key is the LINEAR cell index (`dense`): neighbouring
// cells are neighbouring buckets, the 27 lookups of a body become 9 runs of 3 consecutive entries and bodies next to
// each other share them.  Otherwise the key is a hash of the cell and buckets may hold several cells.
__global__ void __launch_bounds__(kBlock) k_grid(ContactBuffers c, uint32_t n_partials)
{
    __shared__ double part[kBlock / 64][7];
    double v[7] = {0.0, DBL_MAX, DBL_MAX, DBL_MAX, -DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (uint32_t k = threadIdx.x; k < n_partials; k += kBlock)
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const double o = c.grid_partials[(size_t)k * 7 + q];
            v[q] = (q >= 1 && q <= 3) ? (o < v[q] ? o : v[q]) : (o > v[q] ? o : v[q]);
        }
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        double x = v[q];
        for (uint32_t off = 32; off; off >>= 1) {
            const double o = __shfl_xor(x, off, 64);
            x = (q >= 1 && q <= 3) ? (o < x ? o : x) : (o > x ? o : x);
        }
        if ((threadIdx.x & 63u) == 0)
            part[threadIdx.x >> 6][q] = x;
    }
    __syncthreads();
    if (threadIdx.x != 0)
        return;
    for (int q = 0; q < 7; ++q)
        for (uint32_t w = 1; w < kBlock / 64; ++w) {
            const double o = part[w][q];
            part[0][q] = (q >= 1 && q <= 3) ? (o < part[0][q] ? o : part[0][q]) : (o > part[0][q] ? o : part[0][q]);
        }
    GridInfo g;
    g.edge = 2.0 * part[0][0];
    double cells = 1.0;
    for (int a = 0; a < 3; ++a) {
        const double lo = g.edge > 0.0 ? clamp_cell(floor(part[0][1 + a] / g.edge)) : 0.0;
        const double hi = g.edge > 0.0 ? clamp_cell(floor(part[0][4 + a] / g.edge)) : 0.0;
        const double dim = hi >= lo ? hi - lo + 3.0 : 3.0; // one cell of margin on either side
        g.origin[a] = (int32_t)lo - 1;
        g.dims[a] = dim <= 2147483647.0 ? (uint32_t)dim : 0x7FFFFFFFu;
        cells *= dim;
    }
    g.dense = (g.edge > 0.0 && cells <= (double)c.table_size) ? 1u : 0u;
    *c.grid = g;
}

__device__ __forceinline__ uint32_t cell_key(const GridInfo &g, uint32_t table_size, int32_t x, int32_t y, int32_t z)
{
    if (g.dense)
        return (uint32_t)(x - g.origin[0]) + g.dims[0] * ((uint32_t)(y - g.origin[1]) + g.dims[1] * (uint32_t)(z - g.origin[2]));
    return cell_hash(x, y, z) & (table_size - 1);
}

// Grid cell of every body and the population count of its bucket.
__global__ void k_cells(BodyArrays b, ContactBuffers c)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n)
        return;
    const GridInfo g = *c.grid;
    const Vec3 centre = load3(c.centers, 0, b.stride, i);
    int32_t cell[3];
    const double coord[3] = {centre.x, centre.y, centre.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        cell[a] = (int32_t)(g.edge > 0.0 ? clamp_cell(floor(coord[a] / g.edge)) : 0.0);
        c.cell[(size_t)a * b.stride + i] = cell[a];
    }
    const uint32_t key = cell_key(g, c.table_size, cell[0], cell[1], cell[2]);
    c.key[i] = key;
    atomicAdd(&c.bucket_start[key], 1u);
}

// The scatter order inside a bucket depends on timing; the bucket's place of body i is instead its RANK among the
// bodies of the bucket (ids are distinct), which every body works out for itself: one lane per body, O(bucket) reads
// each.  (The first version sorted every bucket with one lane and an insertion sort: 55 us per broadphase on the
// mixed scene, whose dense-grid buckets hold ~19 bodies.)
__global__ void k_scatter(BodyArrays b, ContactBuffers c)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n)
        return;
    const uint32_t key = c.key[i];
    c.items_unsorted[c.bucket_start[key] + atomicAdd(&c.bucket_cursor[key], 1u)] = i;
}

__global__ void k_rank_items(BodyArrays b, ContactBuffers c)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n)
        return;
    const uint32_t key = c.key[i];
    const uint32_t lo = c.bucket_start[key], hi = c.bucket_start[key + 1];
    uint32_t rank = 0;
    for (uint32_t s = lo; s < hi; ++s)
        rank += c.items_unsorted[s] < i;
    c.items[lo + rank] = i;
}

// Bounding spheres and cells once more, in bucket (slot) order: a bucket scan then reads contiguous memory.
__global__ void k_gather_slots(BodyArrays b, ContactBuffers c)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= b.n)
        return;
    const uint32_t j = c.items[s];
    const size_t st = b.stride;
#pragma unroll
    for (uint32_t a = 0; a < 3; ++a) {
        c.slot_sphere[a * st + s] = c.centers[a * st + j];
        c.slot_cell[a * st + s] = c.cell[a * st + j];
    }
    c.slot_sphere[3 * st + s] = c.radius[j];
}

constexpr uint32_t kCellLanes = 8;    // lanes per body in the neighbour kernels
constexpr uint32_t kNbrStage = 128;   // neighbours of one body staged in LDS for the rank sort
constexpr uint32_t kBodiesPerBlock = kBlock / kCellLanes;

// Bucket ranges of the 27 cells around one body, in LDS (one row per body of the block).
struct CellRanges {
    uint32_t start[kBodiesPerBlock][27];
    uint32_t len[kBodiesPerBlock][27];
};

// Neighbour search of body i by its group of kCellLanes lanes (`cl` = lane inside the group, `g` = the group's row in
// the LDS tables).  First the 27 bucket ranges are fetched side by side; then the lanes stride over the CONCATENATION
// of the 27 ranges, so every lane has independent loads in flight whatever the occupancy of the single cells.
// Several cells can share a bucket, so a candidate counts only when it sits in the cell being visited: every
// overlapping j != i is visited exactly once, by exactly one lane.  Must be called by all lanes of the block
// (`live` = false for the groups past the last body): it contains a barrier.
template <class Visit>
__device__ __forceinline__ void for_each_neighbour(const BodyArrays &b, const ContactBuffers &c, CellRanges &r, uint32_t i, bool live,
                                                   uint32_t g, uint32_t cl, Visit visit)
{
    const size_t st = b.stride;
    int32_t cx = 0, cy = 0, cz = 0;
    if (live) {
        cx = c.cell[i], cy = c.cell[st + i], cz = c.cell[2 * st + i];
        const GridInfo grid = *c.grid;
        for (uint32_t k = cl; k < 27; k += kCellLanes) {
            const int32_t nx = cx + (int32_t)(k % 3) - 1, ny = cy + (int32_t)((k / 3) % 3) - 1, nz = cz + (int32_t)(k / 9) - 1;
            const uint32_t key = cell_key(grid, c.table_size, nx, ny, nz);
            const uint32_t lo = c.bucket_start[key];
            r.start[g][k] = lo;
            r.len[g][k] = c.bucket_start[key + 1] - lo;
        }
    }
    __syncthreads();
    if (!live)
        return;
    const Vec3 ci = load3(c.centers, 0, b.stride, i);
    const double ri = c.radius[i];
    uint32_t k = 0, base = 0; // cell being walked and the position of its first candidate in the concatenation
    for (uint32_t q = cl;; q += kCellLanes) {
        while (k < 27 && q >= base + r.len[g][k])
            base += r.len[g][k++];
        if (k == 27)
            break;
        const uint32_t s = r.start[g][k] + (q - base);
        const int32_t nx = cx + (int32_t)(k % 3) - 1, ny = cy + (int32_t)((k / 3) % 3) - 1, nz = cz + (int32_t)(k / 9) - 1;
        if (c.slot_cell[s] != nx || c.slot_cell[st + s] != ny || c.slot_cell[2 * st + s] != nz)
            continue;
        const uint32_t j = c.items[s];
        if (j == i)
            continue;
        const Vec3 d = ci - load3(c.slot_sphere, 0, b.stride, s);
        const double reach = ri + c.slot_sphere[3 * st + s];
        if (dot(d, d) < reach * reach)
            visit(j);
    }
}

__device__ __forceinline__ uint32_t cell_group_sum(uint32_t v)
{
    for (uint32_t off = kCellLanes / 2; off; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

// Counts of all neighbours of every body and of those with a larger id.
__global__ void __launch_bounds__(kBlock) k_neighbour_count(BodyArrays b, ContactBuffers c)
{
    __shared__ CellRanges ranges;
    const uint32_t g = threadIdx.x / kCellLanes, cl = threadIdx.x % kCellLanes;
    // bodies in BUCKET order when the grid is dense (= cell order: the groups of a workgroup then search the same few
    // cells), in index order under the scattering hash (bucket order is a random order there: 178 -> 335 us on `stacks`)
    const uint32_t slot = blockIdx.x * kBodiesPerBlock + g;
    const bool live = slot < b.n;
    const uint32_t i = (live && c.grid->dense) ? c.items[slot] : slot;
    uint32_t all = 0, upper = 0;
    for_each_neighbour(b, c, ranges, i, live, g, cl, [&](uint32_t j) {
        ++all;
        upper += j > i;
    });
    if (!live)
        return;
    all = cell_group_sum(all);
    upper = cell_group_sum(upper);
    if (cl == 0) {
        c.nbr_off[i] = all;
        c.pair_first[i] = upper;
    }
}

// Neighbour list of every body, ASCENDING, the pair list i < j in (i, j) order, and upper_start.  The lanes append
// their finds to an LDS list in arrival order; a rank sort (ids are distinct) then gives every entry its final
// place, so the result does not depend on that order.  Lists longer than kNbrStage take a one-lane path.
__global__ void __launch_bounds__(kBlock) k_neighbour_fill(BodyArrays b, ContactBuffers c)
{
    __shared__ CellRanges ranges;
    __shared__ uint32_t stage[kBodiesPerBlock][kNbrStage];
    __shared__ uint32_t cursor[kBodiesPerBlock];
    const uint32_t g = threadIdx.x / kCellLanes, cl = threadIdx.x % kCellLanes;
    const uint32_t slot = blockIdx.x * kBodiesPerBlock + g;
    const bool live = slot < b.n;
    const uint32_t i = (live && c.grid->dense) ? c.items[slot] : slot;
    uint32_t lo = 0, total = 0;
    if (live) {
        lo = c.nbr_off[i];
        total = c.nbr_off[i + 1] - lo;
    }
    const bool staged = total <= kNbrStage;
    if (cl == 0)
        cursor[g] = 0; // for_each_neighbour's barrier comes before the first visit
    for_each_neighbour(b, c, ranges, i, live, g, cl, [&](uint32_t j) {
        const uint32_t pos = atomicAdd(&cursor[g], 1u);
        if (staged)
            stage[g][pos] = j;
        else
            c.nbr[lo + pos] = j;
    });
    __syncthreads();
    if (!live)
        return;
    const uint32_t first = c.pair_first[i];
    if (staged) {
        // entry e of the arrival list goes to place rank(e); lane cl takes e = cl, cl + kCellLanes, ...
        uint32_t below = 0;
        for (uint32_t e = cl; e < total; e += kCellLanes)
            below += stage[g][e] < i;
        below = cell_group_sum(below);
        if (cl == 0)
            c.upper_start[i] = lo + below;
        for (uint32_t e = cl; e < total; e += kCellLanes) {
            const uint32_t v = stage[g][e];
            uint32_t rank = 0;
            for (uint32_t k = 0; k < total; ++k)
                rank += stage[g][k] < v;
            c.nbr[lo + rank] = v;
            if (v > i) {
                const size_t pair = (size_t)first + (rank - below);
                c.pairs[2 * pair] = i;
                c.pairs[2 * pair + 1] = v;
            }
        }
    } else if (cl == 0) {
        for (uint32_t a = lo + 1; a < lo + total; ++a) {
            const uint32_t v = c.nbr[a];
            uint32_t q = a;
            while (q > lo && c.nbr[q - 1] > v) {
                c.nbr[q] = c.nbr[q - 1];
                --q;
            }
            c.nbr[q] = v;
        }
        uint32_t below = 0;
        while (below < total && c.nbr[lo + below] < i)
            ++below;
        c.upper_start[i] = lo + below;
        for (uint32_t k = below; k < total; ++k) {
            c.pairs[2 * (size_t)(first + k - below)] = i;
            c.pairs[2 * (size_t)(first + k - below) + 1] = c.nbr[lo + k];
        }
    }
}

__global__ void k_neighbour_pair_index(BodyArrays b, ContactBuffers c)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n)
        return;
    for (uint32_t k = c.nbr_off[i]; k < c.nbr_off[i + 1]; ++k) {
        const uint32_t j = c.nbr[k];
        if (j > i) {
            c.nbr_pair[k] = c.pair_first[i] + (k - c.upper_start[i]);
        } else {
            // i sits in the upper part of j's ascending list: binary search
            uint32_t lo = c.upper_start[j], hi = c.nbr_off[j + 1];
            while (lo + 1 < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (c.nbr[mid] <= i)
                    lo = mid;
                else
                    hi = mid;
            }
            c.nbr_pair[k] = c.pair_first[j] + (lo - c.upper_start[j]);
        }
    }
}

// Body of work item `slot` under a BodySubset (false: nothing to do).
__device__ __forceinline__ bool subset_body(const BodySubset &ss, uint32_t n, uint32_t slot, uint32_t &i)
{
    if (ss.list) {
        if (slot >= ss.count)
            return false;
        i = ss.list[slot];
        return true;
    }
    i = slot;
    return slot < n && !(ss.skip && ss.skip[slot]);
}

// The 13 dynamic doubles of a body as one body-major row (halo buffers): position, rotation s x y z, velocity, angular velocity.
__device__ __forceinline__ void store_row(double *__restrict__ rows, uint32_t r, const BodyDynamic &d)
{
    double *o = rows + (size_t)r * kDynFields;
    o[0] = d.pos.x, o[1] = d.pos.y, o[2] = d.pos.z;
    o[3] = d.rot.s, o[4] = d.rot.x, o[5] = d.rot.y, o[6] = d.rot.z;
    o[7] = d.vel.x, o[8] = d.vel.y, o[9] = d.vel.z;
    o[10] = d.ang.x, o[11] = d.ang.y, o[12] = d.ang.z;
}

__device__ __forceinline__ BodyDynamic load_row(const double *__restrict__ rows, uint32_t r)
{
    const double *o = rows + (size_t)r * kDynFields;
    return BodyDynamic{Vec3{o[0], o[1], o[2]}, Quat{o[3], o[4], o[5], o[6]}, Vec3{o[7], o[8], o[9]}, Vec3{o[10], o[11], o[12]}};
}

// ---------------------------------------------------------------------------------------------------
// Per substep, per body: integrate, remember the frames, ground contacts (reference path).
// ---------------------------------------------------------------------------------------------------
template <bool TRACE>
__global__ void __launch_bounds__(kBlock) k_integrate_ground(BodyArrays b, ShapeTable shapes, double h, ContactBuffers c, BodySubset subset,
                                                             uint32_t *__restrict__ last_mask,
                                                             uint32_t *__restrict__ trace_masks, uint32_t trace_row)
{
    extern __shared__ double lds[]; // shape vertex tables, as in k_step
    uint32_t *lds_off = reinterpret_cast<uint32_t *>(lds + 3 * shapes.total_verts);
    for (uint32_t k = threadIdx.x; k < 3 * shapes.total_verts; k += blockDim.x)
        lds[k] = shapes.verts[k];
    for (uint32_t k = threadIdx.x; k <= shapes.n_shapes; k += blockDim.x)
        lds_off[k] = shapes.offsets[k];
    __syncthreads();

    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i;
    if (!subset_body(subset, b.n, slot, i))
        return;
    const uint32_t st = b.stride;
    const BodyStatic s = load_static(b, i);
    // (a ghost body starts the substep from its owner's end-of-substep state, straight from the gathered halo buffer)
    BodyDynamic d = subset.import_buf ? load_row(subset.import_buf, subset.import_rows[slot]) : load_dynamic(b.dyn, st, i);
    const uint32_t sid = b.shape_id[i];
    const uint32_t v0 = lds_off[sid];
    const double compliance = 1e-6 / (h * h);

    const SubstepFrames f = integrate_body(d, s, h);
    const uint32_t mask = solve_ground(d, s, f, compliance, lds + 3 * v0, lds_off[sid + 1] - v0,
                                       c.max_depenetration_speed > 0.0 ? c.max_depenetration_speed * h : 0.0);

    // the pose after the ground contacts lives in the record only: the pair solve (the one reader) takes it from there and
    // rewrites the whole SoA state (velocities come from derive)
    store_record(c.rec, i, f.cur, f.past, d.pos, d.rot, f.past_pos);
    last_mask[i] = mask;
    if (TRACE)
        trace_masks[(size_t)trace_row * st + i] = mask;
}

__global__ void k_body_frames(BodyArrays b, double *__restrict__ rec)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < b.n)
        store_record_p1(rec, i, body_frame(b, i));
}

__global__ void k_stat_records(BodyArrays b, double *__restrict__ stat_rec)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n)
        return;
    const uint32_t st = b.stride;
    double v[kStatRecDoubles] = {};
    v[0] = b.stat[(size_t)S_INV_MASS * st + i];
    for (uint32_t k = 0; k < 9; ++k)
        v[1 + k] = b.stat[(size_t)(S_INV_INERTIA + k) * st + i];
    for (uint32_t k = 0; k < 3; ++k)
        v[10 + k] = b.stat[(size_t)(S_COM + k) * st + i];
    double2 *r = reinterpret_cast<double2 *>(stat_rec + (size_t)i * kStatRecDoubles);
#pragma unroll
    for (uint32_t k = 0; k < kStatRecDoubles / 2; ++k)
        r[k] = double2{v[2 * k], v[2 * k + 1]};
}

// Rigid::frame() of every body, body-major (7 doubles each), staged through LDS so that the global
// stores are contiguous.
__global__ void __launch_bounds__(kBlock) k_body_frames_aos(BodyArrays b, double *__restrict__ frames)
{
    __shared__ double tile[kBlock * 7];
    const uint32_t base = blockIdx.x * kBlock, i = base + threadIdx.x;
    if (i < b.n) {
        const Frame f = body_frame(b, i);
        double *t = tile + threadIdx.x * 7;
        t[0] = f.position.x, t[1] = f.position.y, t[2] = f.position.z;
        t[3] = f.rotation.s, t[4] = f.rotation.x, t[5] = f.rotation.y, t[6] = f.rotation.z;
    }
    __syncthreads();
    const uint32_t count = min(kBlock, b.n - base);
    for (uint32_t k = threadIdx.x; k < count * 7; k += kBlock)
        frames[(size_t)base * 7 + k] = tile[k];
}

// ---------------------------------------------------------------------------------------------------
// Per substep, per body: Jacobi over the pair contacts, then derive.
// ---------------------------------------------------------------------------------------------------
// What the pair solve needs of a body: pose after the ground contacts, mass properties, and its
// frames before / after this substep's integration.
struct PairBody {
    Vec3 pos;
    Quat rot;
    double inv_mass;
    Mat3 inv_inertia;
    Vec3 com;
    Frame p1, past;
};

// The static fields of body i for its own integrate stage: mass properties from the StatRecord it has loaded anyway, the
// forces and torques (which nobody else reads) from the SoA arrays.
__device__ __forceinline__ BodyStatic static_of(const BodyArrays &b, uint32_t i, double inv_mass, const Mat3 &inv_inertia, Vec3 com)
{
    const uint32_t st = b.stride;
    BodyStatic s;
    s.inv_mass = inv_mass;
    s.inv_inertia = inv_inertia;
    s.com = com;
    s.ext_force = load3(b.stat, S_EXT_FORCE, st, i);
    s.int_force = load3(b.stat, S_INT_FORCE, st, i);
    s.ext_torque = load3(b.stat, S_EXT_TORQUE, st, i);
    s.int_torque = load3(b.stat, S_INT_TORQUE, st, i);
    return s;
}

// The BodyRecords of a workgroup's OWN bodies, staged in LDS (the per-body kernels with one lane per body and no body list):
// a body's neighbours are gathers of whole records -- 2 x 128-byte lines each -- and in an index-coherent scene (box stacks: a
// column is 16 consecutive bodies) most of them are bodies of the same workgroup, whose records the workgroup has loaded once
// already.  PMC on `stacks` (profiles/r02_c): 373 MB read per launch for ~160 MB of distinct data, the kernel at the gather
// rate of the memory system (5.3 TB/s of lines).  Word k of the record of the workgroup's body t sits at words[k * kBlock + t]:
// consecutive lanes read consecutive 16-byte words (no bank conflicts).  Same values from another place: same bits.
#ifndef XPBD_PAIR_SOLVE_STAGE_RECORDS
#define XPBD_PAIR_SOLVE_STAGE_RECORDS 0 // measured and rejected, see DESIGN.md 8 (the fetched bytes of the kernel fall by 29 % on box stacks, its time does not)
#endif
struct RecordStage {
    const double2 *words = nullptr; // LDS; null = not staged
    uint32_t first = 0, count = 0;  // bodies [first, first + count) are staged
};

__device__ __forceinline__ BodyRecord load_record_staged(const double *__restrict__ rec, uint32_t i, const RecordStage &stage)
{
    const uint32_t t = i - stage.first;
    if (stage.words && t < stage.count) {
        double2 v[12];
#pragma unroll
        for (int k = 0; k < 12; ++k)
            v[k] = stage.words[k * kBlock + t];
        BodyRecord o;
        o.p1 = Frame{Vec3{v[0].x, v[0].y, v[1].x}, Quat{v[1].y, v[2].x, v[2].y, v[3].x}};
        o.past = Frame{Vec3{v[3].y, v[4].x, v[4].y}, Quat{v[5].x, v[5].y, v[6].x, v[6].y}};
        o.pos = Vec3{v[7].x, v[7].y, v[8].x};
        o.rot = Quat{v[8].y, v[9].x, v[9].y, v[10].x};
        o.past_pos = Vec3{v[10].y, v[11].x, v[11].y};
        return o;
    }
    return load_record(rec, i);
}

// Every thread of the workgroup calls this (it ends in a barrier): thread t stages the record of body `first + t`.
__device__ __forceinline__ RecordStage stage_records(double2 *__restrict__ words, const double *__restrict__ rec, uint32_t first, uint32_t n)
{
    const uint32_t t = threadIdx.x, i = first + t;
    if (i < n) {
        const double2 *r = reinterpret_cast<const double2 *>(rec + (size_t)i * kRecDoubles);
        double2 v[12];
#pragma unroll
        for (int k = 0; k < 12; ++k)
            v[k] = r[k];
#pragma unroll
        for (int k = 0; k < 12; ++k)
            words[k * kBlock + t] = v[k];
    }
    __syncthreads();
    RecordStage s;
    s.words = words;
    s.first = first;
    s.count = n > first ? (n - first < kBlock ? n - first : kBlock) : 0u;
    return s;
}

// (past_pos: optional, the position before integrate -- only a body's own derive needs it)
__device__ __forceinline__ PairBody load_pair_body(const ContactBuffers &c, uint32_t i, Vec3 *past_pos = nullptr,
                                                   const RecordStage &stage = RecordStage())
{
    const BodyRecord r = load_record_staged(c.rec, i, stage); // two cache lines, or LDS
    const StatRecord s = load_stat_record(c.stat_rec, c.stat_index ? c.stat_index[i] : i); // one, or a cached table entry
    PairBody p;
    p.pos = r.pos;
    p.rot = r.rot;
    p.inv_mass = s.inv_mass;
    p.inv_inertia = s.inv_inertia;
    p.com = s.com;
    p.p1 = r.p1;
    p.past = r.past;
    if (past_pos)
        *past_pos = r.past_pos;
    return p;
}

// Frame::delta, src/frame.rs:40-44
__device__ __forceinline__ Vec3 frame_delta(const Frame &cur, const Frame &past, Vec3 global)
{
    const Vec3 local = inverse(cur) * global;
    return global - past * local;
}

// Constraint::inverse_resitance for one body, src/constraint.rs:25-32
__device__ __forceinline__ double generalized_inverse_mass(const PairBody &p, Vec3 point, Vec3 dir)
{
    const Vec3 angular_impulse = conjugate(p.rot) * cross(point - (p.pos + p.com), dir);
    return p.inv_mass + dot(p.inv_inertia * angular_impulse, angular_impulse);
}

// Jacobi pair solve + joints + derive of body i: its state at the end of the substep.
// (touching, points: += the manifolds with contact points among this body's pairs (i, j > i) and their points -- every
// pair is counted by its smaller body, which gives the pipeline's statistics without a pass of their own)
//
// G = lanes per body.  G = 1: the lane walks the points of a manifold one after the other.  G = 8 (small worlds, where
// a wave's dependent chain is the whole cost): lane `sub` evaluates point `sub` of the manifold, and the terms are then
// added on every lane in point order -- the same values in the same order, so the same bits.
#ifndef XPBD_PAIR_SOLVE_ROUND_POINTS
#define XPBD_PAIR_SOLVE_ROUND_POINTS 2 // points of a manifold per round, see pass 2 below; A/B (boxes pile / mixed pile SAT / GJK+EPA,
                                       // 1e8 body-substeps/s): 1 point 5.24 / 2.56 / 2.76, 2 points 5.35 / 2.57 / 2.80, 3 points 5.30 / 2.54 / 2.75
#endif
template <uint32_t G>
__device__ __forceinline__ BodyDynamic pair_solve_derive_body(const ContactBuffers &c, uint32_t i, double h,
                                                              const PairBody &self, Vec3 self_past_pos, uint32_t sub, uint32_t &touching,
                                                              uint32_t &points, const RecordStage &stage = RecordStage())
{
    static_assert(G == 1 || G == kMaxManifoldPoints, "one lane per body or one lane per manifold point");
    const double compliance = 1e-6 / (h * h);
    const double limit = c.max_depenetration_speed > 0.0 ? c.max_depenetration_speed * h : 0.0;

    Vec3 dpos{0.0, 0.0, 0.0};
    Quat drot{0.0, 0.0, 0.0, 0.0};
    uint32_t count = 0;
    // One neighbour with contact points: its points' terms added in point order.
    auto neighbour_terms = [&](uint32_t j, const ContactManifold *m, uint32_t n_points, uint32_t feature, uint32_t pt_begin, uint32_t pt_end) {
        if (j > i && pt_begin == 0) {
            ++touching;
            points += n_points;
        }
        const PairBody other = load_pair_body(c, j, nullptr, stage);
        // pair (A, B) = (min, max); the reference body is A unless the reference face is on B.  The formulas are
        // written in terms of the incident and the reference body; here every term is evaluated for `self` and
        // `other` with their own point and only 3-vectors are selected by role -- selecting whole bodies by a
        // per-lane condition made the compiler keep four body records live (255 VGPRs + AGPRs, one wave per SIMD).
        const bool self_is_a = i < j;
        const bool ref_is_a = feature != 1u;
        const bool self_is_inc = self_is_a != ref_is_a;
        // a face contact stores its reference plane and the points on the incident body; their partners on the reference
        // body are the clipper's projections, re-evaluated with its own expression (xpbd_clip.hpp: Plane::project) -- the
        // same inputs through the same operations, so the same bits.  Feature 2: the single contact's two points.
        const bool face = feature != 2u;
        const Plane ref_plane{Vec3{m->plane[0], m->plane[1], m->plane[2]}, m->plane[3]};
        // one contact point: what it adds to dpos and drot
        auto point_term = [&](uint32_t pt, Vec3 &term_pos, Quat &term_rot) {
            const Vec3 p_inc{m->point[pt][0], m->point[pt][1], m->point[pt][2]};
            Vec3 p_ref;
            if (face) {
                const double depth = distance(ref_plane, p_inc);
                p_ref = p_inc - depth * ref_plane.normal;
            } else {
                p_ref = Vec3{m->point[1][0], m->point[1][1], m->point[1][2]};
            }
            const Vec3 p_self = self_is_inc ? p_inc : p_ref, p_other = self_is_inc ? p_ref : p_inc;
            const Vec3 correction = p_ref - p_inc;
            const Vec3 moved_self = frame_delta(self.p1, self.past, p_self), moved_other = frame_delta(other.p1, other.past, p_other);
            const Vec3 delta_rel = self_is_inc ? moved_self - moved_other : moved_other - moved_self; // incident - reference
            const Vec3 delta_tangential = delta_rel - project_on(delta_rel, correction);
            const Vec3 c0 = p_inc;
            const Vec3 c1 = p_ref - 1.0 * delta_tangential;
            const Vec3 difference = c1 - c0;
            const double dist = length(difference);
            const Vec3 dir = difference * (1.0 / dist);
            // w = w_incident + w_reference; IEEE addition commutes, so the order of the two bodies does not matter
            const double w = generalized_inverse_mass(self, p_self, dir) + generalized_inverse_mass(other, p_other, dir);
            double error = dist;
            if (limit > 0.0) { // xpbd_world_set_max_depenetration_speed (uniform over the launch; 0 = off); oracle: accumulate_point
                const double len = length(correction);
                const double closing = len > 0.0 ? dot(delta_rel, correction) / len : 0.0;
                double allowed = limit - closing;
                if (!(allowed > 0.0))
                    allowed = 0.0;
                if (dist > allowed)
                    error = allowed;
            }
            const double lambda = (error - 0.0) / (w + compliance);

            const Vec3 impulse = self_is_inc ? lambda * dir : (-lambda) * dir;
            term_pos = impulse * self.inv_mass;
            const Vec3 arm = p_self - (self.pos + self.com);
            const Quat spin = quat_sv(0.0, cross(self.inv_inertia * arm, impulse));
            term_rot = (0.5 * spin) * self.rot;
        };
        if (G == 1) {
            for (uint32_t pt = pt_begin; pt < pt_end; ++pt) {
                Vec3 term_pos;
                Quat term_rot;
                point_term(pt, term_pos, term_rot);
                dpos = dpos + term_pos;
                drot = drot + term_rot;
                ++count;
            }
        } else {
            Vec3 term_pos{0.0, 0.0, 0.0};
            Quat term_rot{0.0, 0.0, 0.0, 0.0};
            if (sub < n_points)
                point_term(sub, term_pos, term_rot);
            for (uint32_t pt = 0; pt < n_points; ++pt) { // every lane of the group adds the terms in point order
                dpos = dpos + Vec3{__shfl(term_pos.x, pt, G), __shfl(term_pos.y, pt, G), __shfl(term_pos.z, pt, G)};
                drot = drot + Quat{__shfl(term_rot.s, pt, G), __shfl(term_rot.x, pt, G), __shfl(term_rot.y, pt, G),
                                   __shfl(term_rot.z, pt, G)};
                ++count;
            }
        }
    };
    // Pass 1 -- which of the body's neighbour slots have contact points: bit u of `touch` = slot k_begin + u.  The
    // neighbours in batches of four: their pair indices, then their point counts, are fetched side by side, so a body
    // pays two dependent round trips per batch instead of two per neighbour (most neighbours of a loose scene do not
    // touch, and the whole cost of looking at them is that latency).
    // Pass 2 -- the lanes of a wave then walk THEIR OWN touching neighbours, round by round (ascending, so every body
    // still adds its terms in the oracle's order): a wave runs as many rounds as its busiest body has touching
    // neighbours.  Walking the slots in lockstep instead made it run every slot that any of its 64 bodies had a
    // contact in -- in a settled pile of boxes (7.8 neighbours a body, 3.2 of them touching) 14 slots with four lanes in
    // ten idle in each: 17 % lane utilisation (PMC), 19 000 VALU instructions per wave.
    const uint32_t k_end = c.nbr_off[i + 1];
    for (uint32_t k_base = c.nbr_off[i]; k_base < k_end; k_base += 64) { // (64 slots per mask: one trip for all but monsters)
        const uint32_t k_stop = k_end - k_base > 64u ? k_base + 64u : k_end;
        unsigned long long touch = 0;
        // (a wave whose bodies all have one or two neighbours -- box stacks -- skips pass 1: nothing to filter there)
        const bool few = __all(k_stop - k_base <= 2u);
        if (few)
            touch = k_stop - k_base == 2u ? 3ull : 1ull;
        for (uint32_t k0 = k_base; !few && k0 < k_stop; k0 += 4) {
            uint32_t pair_of[4], points_of[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u)
                pair_of[u] = k0 + u < k_stop ? c.nbr_pair[k0 + u] : 0u;
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u)
                points_of[u] = k0 + u < k_stop ? (uint32_t)c.pair_codes[pair_of[u]] : 0u;
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u)
                if (points_of[u])
                    touch |= 1ull << (k0 - k_base + u);
        }
        // (the next neighbour and its pair index are fetched while the current one's points are evaluated)
        uint32_t j_next = 0, pair_next = 0;
        if (touch) {
            const uint32_t k = k_base + (uint32_t)__ffsll((long long)touch) - 1u;
            j_next = c.nbr[k], pair_next = c.nbr_pair[k];
        }
        // A round takes at most `chunk` points of a manifold; a longer manifold goes on in the lane's next round.  The
        // rounds of a wave are as long as their longest share, and in a pile the manifolds have 1.4 points on average but
        // four in the longest: with whole manifolds per round the lanes idled through two thirds of every round (PMC:
        // 26 % lane utilisation).  Stacks (every manifold four points) take them whole.  G lanes per body: one point per lane.
        const uint32_t chunk = (G > 1 || few) ? kMaxManifoldPoints : (uint32_t)XPBD_PAIR_SOLVE_ROUND_POINTS;
        uint32_t j_cur = 0, pair_cur = 0, code_cur = 0, n_cur = 0, pt = 0;
        while (touch || pt < n_cur) {
            if (pt == n_cur) { // this lane's next touching neighbour
                j_cur = j_next, pair_cur = pair_next;
                touch &= touch - 1ull;
                if (touch) {
                    const uint32_t k = k_base + (uint32_t)__ffsll((long long)touch) - 1u;
                    j_next = c.nbr[k], pair_next = c.nbr_pair[k];
                }
                code_cur = c.pair_codes[pair_cur];
                n_cur = code_cur & ((1u << kPairCodeFeatureShift) - 1u);
                pt = 0;
            }
            if (n_cur) { // (zero only after the shortcut above: a listed neighbour that does not touch)
                const uint32_t stop = n_cur - pt > chunk ? pt + chunk : n_cur;
                neighbour_terms(j_cur, c.manifolds + pair_cur, n_cur, code_cur >> kPairCodeFeatureShift, pt, stop);
                pt = stop;
            }
        }
    }

    // joints of this body, ascending joint index (same accumulator: the "mixed-constraint" pass)
    if (c.joint_off) {
        for (uint32_t k = c.joint_off[i]; k < c.joint_off[i + 1]; ++k) {
            const Joint &jt = c.joints[c.joint_list[k]];
            const bool self_is_a = jt.body_a == i;
            const PairBody other = load_pair_body(c, self_is_a ? jt.body_b : jt.body_a, nullptr, stage);
            // as above: evaluate per body, select 3-vectors by role (a / b)
            const Vec3 anchor_self = self_is_a ? Vec3{jt.anchor_a[0], jt.anchor_a[1], jt.anchor_a[2]} : Vec3{jt.anchor_b[0], jt.anchor_b[1], jt.anchor_b[2]};
            const Vec3 anchor_other = self_is_a ? Vec3{jt.anchor_b[0], jt.anchor_b[1], jt.anchor_b[2]} : Vec3{jt.anchor_a[0], jt.anchor_a[1], jt.anchor_a[2]};
            const Vec3 p_self = Frame{frame_origin(self.pos, self.rot, self.com), self.rot} * anchor_self;
            const Vec3 p_other = Frame{frame_origin(other.pos, other.rot, other.com), other.rot} * anchor_other;
            const Vec3 difference = self_is_a ? p_other - p_self : p_self - p_other; // p_b - p_a
            const double dist = length(difference);
            if (dist != 0.0) { // (coincident points: the reference's direction() would be NaN (K6); nothing to correct)
                const Vec3 dir = difference * (1.0 / dist);
                const double w = generalized_inverse_mass(self, p_self, dir) + generalized_inverse_mass(other, p_other, dir);
                const double lambda = (dist - jt.distance) / (w + compliance);
                const Vec3 impulse = self_is_a ? lambda * dir : (-lambda) * dir;
                dpos = dpos + impulse * self.inv_mass;
                const Vec3 arm = p_self - (self.pos + self.com);
                const Quat spin = quat_sv(0.0, cross(self.inv_inertia * arm, impulse));
                drot = drot + (0.5 * spin) * self.rot;
                ++count;
            }
            if (jt.kind == XPBD_JOINT_HINGE) {
                // the angular term (oracle: accumulate_hinge): the joint's axes, unit vectors in the object space of a and b,
                // are kept aligned; evaluated per body, 3-vectors selected by role as above
                const Vec3 axis_self = self_is_a ? Vec3{jt.axis_a[0], jt.axis_a[1], jt.axis_a[2]} : Vec3{jt.axis_b[0], jt.axis_b[1], jt.axis_b[2]};
                const Vec3 axis_other = self_is_a ? Vec3{jt.axis_b[0], jt.axis_b[1], jt.axis_b[2]} : Vec3{jt.axis_a[0], jt.axis_a[1], jt.axis_a[2]};
                const Vec3 w_self = self.rot * axis_self, w_other = other.rot * axis_other;
                const Vec3 delta = self_is_a ? cross(w_self, w_other) : cross(w_other, w_self); // a_w x b_w
                const double mag = length(delta);
                if (mag != 0.0) {
                    const Vec3 n = delta * (1.0 / mag);
                    const Vec3 n_self = conjugate(self.rot) * n, n_other = conjugate(other.rot) * n;
                    const double w_s = dot(self.inv_inertia * n_self, n_self), w_o = dot(other.inv_inertia * n_other, n_other);
                    const double w = self_is_a ? w_s + w_o : w_o + w_s; // w_a + w_b (IEEE addition commutes; written out for the reader)
                    const double lambda = mag / (w + compliance);
                    const Vec3 turn = self_is_a ? lambda * n : (-lambda) * n;
                    const Quat spin = quat_sv(0.0, self.inv_inertia * turn);
                    drot = drot + (0.5 * spin) * self.rot;
                    ++count;
                }
            }
        }
    }

    BodyDynamic d;
    d.pos = self.pos;
    d.rot = self.rot;
    if (count) {
        const double cnt = (double)count;
        d.pos = self.pos + dpos / cnt;
        d.rot = normalized(self.rot + Quat{drot.s / cnt, drot.x / cnt, drot.y / cnt, drot.z / cnt});
    }
    derive_body(d, self_past_pos, self.past.rotation, h);
    return d;
}

// stats[0] += touching, stats[1] += points, summed over the workgroup: TWO atomics per workgroup (same-address atomics
// serialise at ~10-25 ns each -- never one per pair, see xpbd_pairs.hip).  Every thread of the block must call this.
__device__ __forceinline__ void block_add_stats(uint32_t touching, uint32_t points, unsigned long long *__restrict__ stats)
{
    for (uint32_t off = 32; off; off >>= 1) {
        touching += __shfl_xor(touching, off, 64);
        points += __shfl_xor(points, off, 64);
    }
    __shared__ uint32_t part[2][kBlock / 64];
    if ((threadIdx.x & 63u) == 0) {
        part[0][threadIdx.x >> 6] = touching;
        part[1][threadIdx.x >> 6] = points;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0, p = 0;
        for (uint32_t w = 0; w < kBlock / 64; ++w) {
            t += part[0][w];
            p += part[1][w];
        }
        if (t)
            atomicAdd(&stats[0], (unsigned long long)t);
        if (p)
            atomicAdd(&stats[1], (unsigned long long)p);
    }
}

template <uint32_t G>
__global__ void __launch_bounds__(kBlock, XPBD_PAIR_SOLVE_MIN_WAVES) k_pair_solve_derive(BodyArrays b, double *__restrict__ dyn_out, double h,
                                                                                        ContactBuffers c, BodySubset subset)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, slot = gid / G, sub = gid % G;
    // one lane per body, bodies in index order: the workgroup's own records go through LDS (load_record_staged)
    constexpr bool kStage = G == 1 && XPBD_PAIR_SOLVE_STAGE_RECORDS;
    __shared__ double2 stage_words[kStage ? 12 * kBlock : 1];
    RecordStage stage;
    if (kStage && !subset.list)
        stage = stage_records(stage_words, c.rec, blockIdx.x * kBlock, b.n);
    uint32_t touching = 0, points = 0, i;
    if (subset_body(subset, b.n, slot, i)) {
        Vec3 past_pos;
        const PairBody self = load_pair_body(c, i, &past_pos, stage);
        const BodyDynamic d = pair_solve_derive_body<G>(c, i, h, self, past_pos, sub, touching, points, stage);
        if (sub == 0) {
            store_dynamic(dyn_out, b.stride, i, d);
            if (subset.export_rows)
                store_row(subset.export_rows, slot, d);
        }
    }
    block_add_stats(sub == 0 ? touching : 0u, sub == 0 ? points : 0u, c.stats);
}

// The end of substep k and the beginning of substep k + 1 of one body in one kernel: pair solve + derive, then
// integrate + ground contacts straight from registers.  Nothing a body needs from the others changes in between
// (they read each other's state of substep k: dyn and the frames of `c`), so the new state goes to the other dyn
// buffer and the frames of substep k + 1 to the other frame set (`next`).  Saves the store + reload of the dynamic
// state, half of the static loads and one launch per substep; same arithmetic, same bits.  Only for step calls
// that run all their substeps on one device: a halo exchange sits exactly at this seam.
// (G lanes per body as in pair_solve_derive_body; with G = 8 the integrate + ground part runs redundantly on the eight
// lanes and lane 0 stores.)
template <bool TRACE, uint32_t G>
__global__ void __launch_bounds__(kBlock, XPBD_PAIR_SOLVE_MIN_WAVES) k_pair_solve_integrate_ground(
    BodyArrays b, ShapeTable shapes, double h, ContactBuffers c, BodySubset subset, double *__restrict__ next_rec,
    uint32_t *__restrict__ last_mask, uint32_t *__restrict__ trace_masks, uint32_t trace_row)
{
    extern __shared__ double lds[]; // shape vertex tables, as in k_integrate_ground
    uint32_t *lds_off = reinterpret_cast<uint32_t *>(lds + 3 * shapes.total_verts);
    for (uint32_t k = threadIdx.x; k < 3 * shapes.total_verts; k += blockDim.x)
        lds[k] = shapes.verts[k];
    for (uint32_t k = threadIdx.x; k <= shapes.n_shapes; k += blockDim.x)
        lds_off[k] = shapes.offsets[k];
    __syncthreads();

    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, slot = gid / G, sub = gid % G;
    constexpr bool kStage = G == 1 && XPBD_PAIR_SOLVE_STAGE_RECORDS; // see k_pair_solve_derive
    __shared__ double2 stage_words[kStage ? 12 * kBlock : 1];
    RecordStage stage;
    if (kStage && !subset.list)
        stage = stage_records(stage_words, c.rec, blockIdx.x * kBlock, b.n);
    uint32_t touching = 0, points = 0, i;
    if (subset_body(subset, b.n, slot, i)) {
        const uint32_t st = b.stride;
        BodyDynamic d;
        BodyStatic s;
        {
            Vec3 past_pos;
            const PairBody self = load_pair_body(c, i, &past_pos, stage);
            d = pair_solve_derive_body<G>(c, i, h, self, past_pos, sub, touching, points, stage);
            s = static_of(b, i, self.inv_mass, self.inv_inertia, self.com);
        }
        if (subset.export_rows && sub == 0)
            store_row(subset.export_rows, slot, d); // the end-of-substep state, before the next substep's integration
        const uint32_t sid = b.shape_id[i];
        const uint32_t v0 = lds_off[sid];
        const double compliance = 1e-6 / (h * h);
        const SubstepFrames f = integrate_body(d, s, h);
        const uint32_t mask = solve_ground(d, s, f, compliance, lds + 3 * v0, lds_off[sid + 1] - v0,
                                       c.max_depenetration_speed > 0.0 ? c.max_depenetration_speed * h : 0.0);
        if (sub == 0) {
            // everything the next kernel needs of this body is its record: no SoA state is written between the substeps of
            // a step call (the last substep's k_pair_solve_derive writes all 13 dynamic fields)
            store_record(next_rec, i, f.cur, f.past, d.pos, d.rot, f.past_pos);
            last_mask[i] = mask;
            if (TRACE)
                trace_masks[(size_t)trace_row * st + i] = mask;
        }
    }
    block_add_stats(sub == 0 ? touching : 0u, sub == 0 ? points : 0u, c.stats);
}

// Halo exchange: one lane per (body, field); the buffer side is contiguous, the SoA side is a gather.
__global__ void k_export_dynamic(BodyArrays b, const uint32_t *__restrict__ indices, uint32_t n, double *__restrict__ buf)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * kDynFields)
        return;
    const uint32_t k = t / kDynFields, f = t - k * kDynFields;
    buf[t] = b.dyn[(size_t)f * b.stride + indices[k]];
}

// (rows: optional; entry k of the list then comes from row rows[k] of buf instead of row k -- the gathered halo
// buffer of an all-gather can be imported as it is)
__global__ void k_import_dynamic(BodyArrays b, const uint32_t *__restrict__ indices, const uint32_t *__restrict__ rows, uint32_t n,
                                 const double *__restrict__ buf)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * kDynFields)
        return;
    const uint32_t k = t / kDynFields, f = t - k * kDynFields;
    b.dyn[(size_t)f * b.stride + indices[k]] = buf[(size_t)(rows ? rows[k] : k) * kDynFields + f];
}

// Halo validity (multi-GPU): positions of the listed bodies when the halos were chosen ...
__global__ void k_snapshot_positions(BodyArrays b, const uint32_t *__restrict__ indices, uint32_t n, double *__restrict__ snapshot)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    const Vec3 p = load3(b.dyn, D_POS, b.stride, indices[k]);
    snapshot[3 * (size_t)k + 0] = p.x, snapshot[3 * (size_t)k + 1] = p.y, snapshot[3 * (size_t)k + 2] = p.z;
}

// ... and the largest squared distance any of them has travelled since: *out = max(*out, max_k |pos_k - snapshot_k|^2).
// Non-negative doubles order like their bit patterns, so the maximum is an integer atomicMax (one per workgroup); a NaN
// position counts as +inf.
// (scale: optional per-entry factor on the squared distance, e.g. 1 / allowance^2: the result is then a ratio)
__global__ void __launch_bounds__(kBlock) k_max_displacement2(BodyArrays b, const uint32_t *__restrict__ indices, uint32_t n,
                                                              const double *__restrict__ snapshot, const double *__restrict__ scale,
                                                              double *__restrict__ out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    double d2 = 0.0;
    if (k < n) {
        const Vec3 p = load3(b.dyn, D_POS, b.stride, indices[k]);
        const Vec3 d = p - Vec3{snapshot[3 * (size_t)k + 0], snapshot[3 * (size_t)k + 1], snapshot[3 * (size_t)k + 2]};
        d2 = dot(d, d);
        if (scale)
            d2 *= scale[k];
        if (!(d2 <= DBL_MAX))
            d2 = __longlong_as_double(0x7FF0000000000000ll);
    }
    for (uint32_t off = 32; off; off >>= 1) {
        const double o = __shfl_xor(d2, off, 64);
        d2 = o > d2 ? o : d2;
    }
    __shared__ double part[kBlock / 64];
    if ((threadIdx.x & 63u) == 0)
        part[threadIdx.x >> 6] = d2;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < kBlock / 64; ++w)
            d2 = part[w] > d2 ? part[w] : d2;
        if (d2 > 0.0)
            atomicMax(reinterpret_cast<unsigned long long *>(out), (unsigned long long)__double_as_longlong(d2));
    }
}

// ---- re-planning the multi-GPU world without a host round trip of the bodies (xpbd_multi.cpp) --------------------------------
// The grid cell of the bounding-sphere centre (position + center_of_mass, as the host planner adds them) of the listed
// bodies: the same floor(c / edge), clamp and packing as xpbd_halo_cell_key, so the same bits.  *bad = the smallest list
// index with a non-finite centre (UINT32_MAX: none).
constexpr long long kHaloCellBias = 1ll << 20, kHaloCellLimit = kHaloCellBias - 4;

__device__ __forceinline__ long long halo_clamp_cell(double q)
{
    const double lim = (double)kHaloCellLimit;
    if (!(q >= -lim)) // NaN or far negative
        return -kHaloCellLimit;
    return q > lim ? kHaloCellLimit : (long long)q;
}

__global__ void k_cell_keys(BodyArrays b, const uint32_t *__restrict__ indices, uint32_t n, double edge, long long *__restrict__ keys,
                            uint32_t *__restrict__ bad)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    const uint32_t i = indices ? indices[k] : k;
    const Vec3 p = load3(b.dyn, D_POS, b.stride, i), com = load3(b.stat, S_COM, b.stride, i);
    const double c[3] = {p.x + com.x, p.y + com.y, p.z + com.z};
    long long cell[3];
    bool finite = true;
    for (int a = 0; a < 3; ++a) {
        finite = finite && fabs(c[a]) <= DBL_MAX;
        cell[a] = halo_clamp_cell(floor(c[a] / edge));
    }
    if (!finite)
        atomicMin(bad, k);
    keys[k] = ((cell[0] + kHaloCellBias) << 42) | ((cell[1] + kHaloCellBias) << 21) | (cell[2] + kHaloCellBias);
}

__device__ __forceinline__ uint32_t aos_slot_of_field(uint32_t f) // xpbd_rigid double index of SoA field f (dyn first, then stat)
{
    if (f < kDynFields) // pos 31-33, rot 34-37, vel 22-24, ang 25-27
        return f < 7 ? 31 + f : (f < 10 ? 22 + (f - 7) : 25 + (f - 10));
    const uint32_t g = f - kDynFields; // 0..21 identical; com 28-30
    return g < 22 ? g : 28 + (g - 22);
}

// out[k] = {xpbd_rigid of body indices[k] (38 doubles), its shape id as a double}: the plan-time record of a body
__global__ void k_gather_records(BodyArrays b, const uint32_t *__restrict__ indices, uint32_t n, double *__restrict__ out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr uint32_t kFields = kDynFields + kStatFields;
    if (t >= n * (kFields + 1))
        return;
    const uint32_t k = t / (kFields + 1), f = t - k * (kFields + 1), i = indices[k];
    double *rec = out + (size_t)k * (kFields + 1);
    if (f == kFields)
        rec[kFields] = (double)b.shape_id[i];
    else
        rec[aos_slot_of_field(f)] = f < kDynFields ? b.dyn[(size_t)f * b.stride + i] : b.stat[(size_t)(f - kDynFields) * b.stride + i];
}

// The bodies of a new local world in AoS: body s comes from slot src[s] of the old world's AoS copy (src[s] >= 0) or is
// incoming record -src[s] - 1 (39 doubles each: xpbd_rigid + shape id).
__global__ void k_repack_bodies(const double *__restrict__ old_aos, const uint32_t *__restrict__ old_shape, const int32_t *__restrict__ src,
                                uint32_t n_new, const double *__restrict__ incoming, double *__restrict__ new_aos,
                                uint32_t *__restrict__ new_shape)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr uint32_t kFields = kDynFields + kStatFields;
    if (t >= (size_t)n_new * kFields)
        return;
    const uint32_t s = (uint32_t)(t / kFields), f = (uint32_t)(t - (size_t)s * kFields);
    const int32_t from = src[s];
    if (from >= 0) {
        new_aos[t] = old_aos[(size_t)from * kFields + f];
        if (f == 0)
            new_shape[s] = old_shape[from];
    } else {
        const double *rec = incoming + (size_t)(-(from + 1)) * (kFields + 1);
        new_aos[t] = rec[f];
        if (f == 0)
            new_shape[s] = (uint32_t)rec[kFields];
    }
}

uint32_t blocks_for(uint32_t n) { return (n + kBlock - 1) / kBlock; }

} // namespace

// ---------------------------------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------------------------------
hipError_t launch_exclusive_scan(uint32_t *data, uint32_t n, uint32_t *scratch, hipStream_t stream)
{
    const uint32_t nb = (n + kBlock * 4 - 1) / (kBlock * 4);
    if (nb == 0) {
        return hipMemsetAsync(data, 0, sizeof(uint32_t), stream);
    }
    hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(kBlock), 0, stream, data, n, scratch);
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(kBlock), 0, stream, scratch, nb);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(kBlock), 0, stream, data, n, scratch, nb);
    return hipGetLastError();
}

hipError_t launch_bounds_and_cells(const BodyArrays &b, const PolytopeTables &t, const double *shape_radius,
                                   double dt, double pad, const ContactBuffers &c, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(c.bucket_start, 0, (size_t)(c.table_size + 1) * 4, stream);
    if (e == hipSuccess)
        e = hipMemsetAsync(c.bucket_cursor, 0, (size_t)c.table_size * 4, stream);
    if (e != hipSuccess || b.n == 0)
        return e;
    hipLaunchKernelGGL(k_bounds, dim3(blocks_for(b.n)), dim3(kBlock), 0, stream, b, t, shape_radius, dt, pad, c);
    hipLaunchKernelGGL(k_grid, dim3(1), dim3(kBlock), 0, stream, c, blocks_for(b.n));
    hipLaunchKernelGGL(k_cells, dim3(blocks_for(b.n)), dim3(kBlock), 0, stream, b, c);
    return hipGetLastError();
}

hipError_t launch_build_buckets(const BodyArrays &b, const ContactBuffers &c, hipStream_t stream)
{
    hipError_t e = launch_exclusive_scan(c.bucket_start, c.table_size, c.scan_scratch, stream);
    if (e != hipSuccess || b.n == 0)
        return e;
    hipLaunchKernelGGL(k_scatter, dim3(blocks_for(b.n)), dim3(kBlock), 0, stream, b, c);
    hipLaunchKernelGGL(k_rank_items, dim3(blocks_for(b.n)), dim3(kBlock), 0, stream, b, c);
    hipLaunchKernelGGL(k_gather_slots, dim3(blocks_for(b.n)), dim3(kBlock), 0, stream, b, c);
    return hipGetLastError();
}

hipError_t launch_neighbour_count(const BodyArrays &b, const ContactBuffers &c, hipStream_t stream)
{
    if (b.n)
        hipLaunchKernelGGL(k_neighbour_count, dim3((b.n + kBodiesPerBlock - 1) / kBodiesPerBlock), dim3(kBlock), 0, stream, b, c);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = launch_exclusive_scan(c.nbr_off, b.n, c.scan_scratch, stream);
    if (e == hipSuccess)
        e = launch_exclusive_scan(c.pair_first, b.n, c.scan_scratch, stream);
    return e;
}

hipError_t launch_neighbour_fill(const BodyArrays &b, const ContactBuffers &c, hipStream_t stream)
{
    if (b.n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_neighbour_fill, dim3((b.n + kBodiesPerBlock - 1) / kBodiesPerBlock), dim3(kBlock), 0, stream, b, c);
    hipLaunchKernelGGL(k_neighbour_pair_index, dim3(blocks_for(b.n)), dim3(kBlock), 0, stream, b, c);
    return hipGetLastError();
}

hipError_t launch_integrate_ground(const BodyArrays &b, const ShapeTable &s, double h, const ContactBuffers &c,
                                   uint32_t *last_mask, uint32_t *trace_masks, uint32_t trace_row, hipStream_t stream,
                                   const BodySubset &subset)
{
    const uint32_t items = subset.list ? subset.count : b.n;
    if (items == 0)
        return hipSuccess;
    const size_t lds_bytes = (size_t)s.total_verts * 3 * sizeof(double) + (size_t)(s.n_shapes + 1) * sizeof(uint32_t);
    if (trace_masks)
        hipLaunchKernelGGL(k_integrate_ground<true>, dim3(blocks_for(items)), dim3(kBlock), lds_bytes, stream, b, s, h, c, subset,
                           last_mask, trace_masks, trace_row);
    else
        hipLaunchKernelGGL(k_integrate_ground<false>, dim3(blocks_for(items)), dim3(kBlock), lds_bytes, stream, b, s, h, c, subset,
                           last_mask, trace_masks, trace_row);
    return hipGetLastError();
}

hipError_t launch_sat_contact_pairs(const BodyArrays &b, const PolytopeTables &t, const ContactBuffers &c,
                                    uint32_t n_pairs, SatScratch *list, hipStream_t stream, bool dense)
{
    return launch_sat_contacts(b, t, c.rec, c.pairs, n_pairs, c.manifolds, c.pair_codes, list, stream, dense);
}

hipError_t launch_pair_solve_derive(const BodyArrays &b, double *dyn_out, double h, const ContactBuffers &c,
                                    hipStream_t stream, const BodySubset &subset)
{
    const uint32_t items = subset.list ? subset.count : b.n;
    if (items == 0)
        return hipSuccess;
    if (b.n <= kSmallWorld)
        hipLaunchKernelGGL(k_pair_solve_derive<kMaxManifoldPoints>, dim3(blocks_for(items * kMaxManifoldPoints)), dim3(kBlock), 0, stream, b,
                           dyn_out, h, c, subset);
    else
        hipLaunchKernelGGL(k_pair_solve_derive<1>, dim3(blocks_for(items)), dim3(kBlock), 0, stream, b, dyn_out, h, c, subset);
    return hipGetLastError();
}

hipError_t launch_pair_solve_integrate_ground(const BodyArrays &b, const ShapeTable &s, double h, const ContactBuffers &c,
                                              double *next_rec, uint32_t *last_mask, uint32_t *trace_masks, uint32_t trace_row,
                                              hipStream_t stream, const BodySubset &subset)
{
    const uint32_t items = subset.list ? subset.count : b.n;
    if (items == 0)
        return hipSuccess;
    const size_t lds_bytes = (size_t)s.total_verts * 3 * sizeof(double) + (size_t)(s.n_shapes + 1) * sizeof(uint32_t);
    auto launch = [&](auto trace, auto lanes) {
        constexpr uint32_t G = decltype(lanes)::value;
        hipLaunchKernelGGL((k_pair_solve_integrate_ground<decltype(trace)::value, G>), dim3(blocks_for(items * G)), dim3(kBlock), lds_bytes,
                           stream, b, s, h, c, subset, next_rec, last_mask, trace_masks, trace_row);
    };
    using Wide = std::integral_constant<uint32_t, kMaxManifoldPoints>;
    using One = std::integral_constant<uint32_t, 1>;
    const bool small = b.n <= kSmallWorld;
    if (trace_masks && small)
        launch(std::true_type{}, Wide{});
    else if (trace_masks)
        launch(std::true_type{}, One{});
    else if (small)
        launch(std::false_type{}, Wide{});
    else
        launch(std::false_type{}, One{});
    return hipGetLastError();
}

hipError_t launch_body_frames_aos(const BodyArrays &b, double *frames, hipStream_t stream)
{
    if (b.n)
        hipLaunchKernelGGL(k_body_frames_aos, dim3(blocks_for(b.n)), dim3(kBlock), 0, stream, b, frames);
    return hipGetLastError();
}

hipError_t launch_export_dynamic(const BodyArrays &b, const uint32_t *indices, uint32_t n, double *buf, hipStream_t stream)
{
    if (n)
        hipLaunchKernelGGL(k_export_dynamic, dim3(blocks_for(n * kDynFields)), dim3(kBlock), 0, stream, b, indices, n, buf);
    return hipGetLastError();
}

hipError_t launch_import_dynamic(const BodyArrays &b, const uint32_t *indices, const uint32_t *rows, uint32_t n, const double *buf,
                                 hipStream_t stream)
{
    if (n)
        hipLaunchKernelGGL(k_import_dynamic, dim3(blocks_for(n * kDynFields)), dim3(kBlock), 0, stream, b, indices, rows, n, buf);
    return hipGetLastError();
}

hipError_t launch_snapshot_positions(const BodyArrays &b, const uint32_t *indices, uint32_t n, double *snapshot, hipStream_t stream)
{
    if (n)
        hipLaunchKernelGGL(k_snapshot_positions, dim3(blocks_for(n)), dim3(kBlock), 0, stream, b, indices, n, snapshot);
    return hipGetLastError();
}

hipError_t launch_max_displacement2(const BodyArrays &b, const uint32_t *indices, uint32_t n, const double *snapshot, const double *scale,
                                    double *out, hipStream_t stream)
{
    if (n)
        hipLaunchKernelGGL(k_max_displacement2, dim3(blocks_for(n)), dim3(kBlock), 0, stream, b, indices, n, snapshot, scale, out);
    return hipGetLastError();
}

hipError_t launch_cell_keys(const BodyArrays &b, const uint32_t *indices, uint32_t n, double edge, int64_t *keys, uint32_t *bad, hipStream_t stream)
{
    if (n)
        hipLaunchKernelGGL(k_cell_keys, dim3(blocks_for(n)), dim3(kBlock), 0, stream, b, indices, n, edge, reinterpret_cast<long long *>(keys), bad);
    return hipGetLastError();
}

hipError_t launch_gather_records(const BodyArrays &b, const uint32_t *indices, uint32_t n, double *out, hipStream_t stream)
{
    if (n)
        hipLaunchKernelGGL(k_gather_records, dim3(blocks_for(n * (kDynFields + kStatFields + 1))), dim3(kBlock), 0, stream, b, indices, n, out);
    return hipGetLastError();
}

hipError_t launch_repack_bodies(const double *old_aos, const uint32_t *old_shape, const int32_t *src, uint32_t n_new, const double *incoming,
                                double *new_aos, uint32_t *new_shape, hipStream_t stream)
{
    if (n_new) {
        const size_t threads = (size_t)n_new * (kDynFields + kStatFields);
        hipLaunchKernelGGL(k_repack_bodies, dim3((uint32_t)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, old_aos, old_shape, src, n_new,
                           incoming, new_aos, new_shape);
    }
    return hipGetLastError();
}

hipError_t launch_body_frames(const BodyArrays &b, double *rec, hipStream_t stream)
{
    if (b.n)
        hipLaunchKernelGGL(k_body_frames, dim3(blocks_for(b.n)), dim3(kBlock), 0, stream, b, rec);
    return hipGetLastError();
}

hipError_t launch_stat_records(const BodyArrays &b, double *stat_rec, hipStream_t stream)
{
    if (b.n)
        hipLaunchKernelGGL(k_stat_records, dim3(blocks_for(b.n)), dim3(kBlock), 0, stream, b, stat_rec);
    return hipGetLastError();
}

} // namespace xpbd
