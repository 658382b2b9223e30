// xpbd_kernels.hip -- gfx950 (MI355X, CDNA4, wave64) kernels of the XPBD stepper.
//
// Hot path rebuilt here (reference file:line):
//   solver::step substep loop        src/solver.rs:6-16
//   Rigid::frame                     src/rigid.rs:75-80
//   Rigid::integrate                 src/rigid.rs:82-99
//   collision::ground                src/collision.rs:13-35   (+ Frame ops src/frame.rs:30-53)
//   solver::solve                    src/solver.rs:19-27
//   Constraint::{current_distance, inverse_resitance, act}    src/constraint.rs:21-37
//   Rigid::apply_impulse             src/rigid.rs:113-123
//   Rigid::derive                    src/rigid.rs:101-109
//
// Mapping: one lane = one body, all arithmetic IEEE f64 in the reference's
// operation order (compile with -ffp-contract=off).  No MFMA: this is 3-vector
// math, not a contraction.  State lives in registers across substeps; shape
// vertex tables sit in LDS; every global access is a contiguous 512-byte wave
// access on the SoA arrays (xpbd_kernels.h).
#include "xpbd_step.hpp"

namespace xpbd {
namespace {

// ---------------------------------------------------------------------------
// k_step: `substeps` substeps for every body in one launch.
// LDS: the shape vertex tables (<= a few hundred bytes), staged once per block.
// ---------------------------------------------------------------------------
#ifndef XPBD_STEP_MIN_WAVES_PER_SIMD
#define XPBD_STEP_MIN_WAVES_PER_SIMD 1
#endif

template <bool TRACE>
__global__ void __launch_bounds__(kMaxStepBlock, XPBD_STEP_MIN_WAVES_PER_SIMD) k_step(BodyArrays b, ShapeTable shapes, double h, uint32_t substeps,
                       uint32_t *__restrict__ last_mask, uint32_t *__restrict__ trace_masks,
                       uint32_t trace_row0)
{
    extern __shared__ double lds[]; // [total_verts*3] doubles, then [n_shapes+1] uint32
    uint32_t *lds_off = reinterpret_cast<uint32_t *>(lds + 3 * shapes.total_verts);
    for (uint32_t k = threadIdx.x; k < 3 * shapes.total_verts; k += blockDim.x)
        lds[k] = shapes.verts[k];
    for (uint32_t k = threadIdx.x; k <= shapes.n_shapes; k += blockDim.x)
        lds_off[k] = shapes.offsets[k];
    __syncthreads();

    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n)
        return;
    const uint32_t st = b.stride;

    const BodyStatic s = load_static(b, i);
    BodyDynamic d = load_dynamic(b.dyn, st, i);

    const uint32_t sid = b.shape_id[i];
    const uint32_t v0 = lds_off[sid];
    const uint32_t nv = lds_off[sid + 1] - v0;
    const double *verts = lds + 3 * v0;

    const double compliance = 1e-6 / (h * h); // src/solver.rs:20

    uint32_t mask = 0;
    for (uint32_t k = 0; k < substeps; ++k) {
        mask = substep(d, s, h, compliance, verts, nv);
        if (TRACE)
            trace_masks[(size_t)(trace_row0 + k) * st + i] = mask;
    }

    store_dynamic(b.dyn, st, i, d);
    last_mask[i] = mask;
}

// ---------------------------------------------------------------------------
// AoS <-> SoA.  A block moves 64 bodies: the 64*38-double AoS chunk is one
// contiguous range (coalesced) and is transposed through LDS into 38
// contiguous 512-byte field segments.
// ---------------------------------------------------------------------------
constexpr uint32_t kTile = 64;

// AoS double index of SoA field f (dyn first, then stat); see xpbd.h xpbd_rigid.
__device__ __forceinline__ uint32_t aos_index_of_dyn(uint32_t f)
{
    // pos 31-33, rot 34-37, vel 22-24, ang 25-27
    return f < 7 ? 31 + f : (f < 10 ? 22 + (f - 7) : 25 + (f - 10));
}
__device__ __forceinline__ uint32_t aos_index_of_stat(uint32_t f)
{
    // 0..21 identical; com 28-30
    return f < 22 ? f : 28 + (f - 22);
}

__global__ void k_aos_to_soa(const double *__restrict__ aos, BodyArrays b)
{
    __shared__ double tile[kTile * kRigidDoubles];
    const uint32_t base = blockIdx.x * kTile;
    const uint32_t count = min(kTile, b.n - base);
    const size_t src0 = (size_t)base * kRigidDoubles;
    for (uint32_t k = threadIdx.x; k < count * kRigidDoubles; k += blockDim.x)
        tile[k] = aos[src0 + k];
    __syncthreads();
    const uint32_t t = threadIdx.x;
    if (t >= count)
        return;
    for (uint32_t f = 0; f < kDynFields; ++f)
        b.dyn[(size_t)f * b.stride + base + t] = tile[t * kRigidDoubles + aos_index_of_dyn(f)];
    for (uint32_t f = 0; f < kStatFields; ++f)
        b.stat[(size_t)f * b.stride + base + t] = tile[t * kRigidDoubles + aos_index_of_stat(f)];
}

__global__ void k_soa_to_aos(BodyArrays b, double *__restrict__ aos)
{
    __shared__ double tile[kTile * kRigidDoubles];
    const uint32_t base = blockIdx.x * kTile;
    const uint32_t count = min(kTile, b.n - base);
    const uint32_t t = threadIdx.x;
    if (t < count) {
        for (uint32_t f = 0; f < kDynFields; ++f)
            tile[t * kRigidDoubles + aos_index_of_dyn(f)] = b.dyn[(size_t)f * b.stride + base + t];
        for (uint32_t f = 0; f < kStatFields; ++f)
            tile[t * kRigidDoubles + aos_index_of_stat(f)] = b.stat[(size_t)f * b.stride + base + t];
    }
    __syncthreads();
    const size_t dst0 = (size_t)base * kRigidDoubles;
    for (uint32_t k = threadIdx.x; k < count * kRigidDoubles; k += blockDim.x)
        aos[dst0 + k] = tile[k];
}

// ---------------------------------------------------------------------------
// Contact list: per-body masks -> (body, vertex) pairs sorted by body, vertex.
// Two passes over the masks with a block-level scan in between.
// ---------------------------------------------------------------------------
constexpr uint32_t kScanBlock = 256;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(v, d, 64);
        if (lane >= d)
            v += up;
    }
    return v;
}

// Exclusive scan across a 256-thread block; returns the block total via *total.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *total)
{
    __shared__ uint32_t wave_sum[kScanBlock / 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63)
        wave_sum[wave] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanBlock / 64; ++k) {
        const uint32_t ws = wave_sum[k];
        if (k < wave)
            before += ws;
        all += ws;
    }
    *total = all;
    return before + inc - v;
}

__global__ void k_contacts_block_count(const uint32_t *__restrict__ mask, uint32_t n,
                                       uint32_t *__restrict__ block_counts)
{
    const uint32_t i = blockIdx.x * kScanBlock + threadIdx.x;
    const uint32_t c = i < n ? __popc(mask[i]) : 0u;
    uint32_t total;
    (void)block_exclusive_scan(c, &total);
    if (threadIdx.x == 0)
        block_counts[blockIdx.x] = total;
}

// Single block: in-place exclusive scan of block_counts[0..nb), total -> [nb].
__global__ void k_contacts_scan_blocks(uint32_t *__restrict__ block_counts, uint32_t nb)
{
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0)
        carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += kScanBlock) {
        const uint32_t k = base + threadIdx.x;
        const uint32_t v = k < nb ? block_counts[k] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, &total);
        const uint32_t carry = carry_s;
        if (k < nb)
            block_counts[k] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0)
            carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        block_counts[nb] = carry_s;
}

__global__ void k_contacts_emit(const uint32_t *__restrict__ mask, uint32_t n,
                                const uint32_t *__restrict__ block_offsets,
                                xpbd_contact *__restrict__ out, uint32_t cap)
{
    const uint32_t i = blockIdx.x * kScanBlock + threadIdx.x;
    uint32_t m = i < n ? mask[i] : 0u;
    uint32_t total;
    uint32_t at = block_offsets[blockIdx.x] + block_exclusive_scan(__popc(m), &total);
    while (m) {
        const uint32_t v = __ffs(m) - 1;
        m &= m - 1;
        if (at < cap)
            out[at] = xpbd_contact{i, v};
        ++at;
    }
}

__global__ void k_selftest_div_sqrt(const double *__restrict__ a, const double *__restrict__ b,
                                    double *__restrict__ q, double *__restrict__ r, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        q[i] = a[i] / b[i];
        r[i] = sqrt(a[i]);
    }
}

// Device-to-device copy kernels: the measured HBM roof the stepper's achieved bandwidth is priced against
// (xpbd_selftest_hbm_copy reports the fastest variant).  16 bytes per lane and access; ONE element per lane with a
// grid as large as the buffer, or grid-stride with four independent loads in flight per lane; plain or non-temporal
// accesses (every byte is touched once).
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ v4f copy_load(const v4f *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void copy_store(v4f *p, v4f v)
{
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

template <bool NT>
__global__ void __launch_bounds__(256) k_copy16_flat(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n16)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n16)
        copy_store<NT>(dst + k, copy_load<NT>(src + k));
}

template <bool NT>
__global__ void __launch_bounds__(256) k_copy16_strided(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; k + 3 * stride < n16; k += 4 * stride) {
        v4f v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = copy_load<NT>(src + k + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            copy_store<NT>(dst + k + u * stride, v[u]);
    }
    for (; k < n16; k += stride)
        dst[k] = src[k];
}

} // namespace

// ---------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------
hipError_t launch_step(const BodyArrays &b, const ShapeTable &s, double h, uint32_t substeps,
                       uint32_t *last_mask, uint32_t *trace_masks, uint32_t trace_row0,
                       uint32_t block_size, hipStream_t stream)
{
    if (b.n == 0)
        return hipSuccess;
    const uint32_t grid = (b.n + block_size - 1) / block_size;
    const size_t lds_bytes = (size_t)s.total_verts * 3 * sizeof(double) + (size_t)(s.n_shapes + 1) * sizeof(uint32_t);
    if (trace_masks)
        hipLaunchKernelGGL(k_step<true>, dim3(grid), dim3(block_size), lds_bytes, stream, b, s, h, substeps,
                           last_mask, trace_masks, trace_row0);
    else
        hipLaunchKernelGGL(k_step<false>, dim3(grid), dim3(block_size), lds_bytes, stream, b, s, h, substeps,
                           last_mask, trace_masks, trace_row0);
    return hipGetLastError();
}

hipError_t launch_aos_to_soa(const double *aos, const BodyArrays &b, hipStream_t stream)
{
    if (b.n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_aos_to_soa, dim3((b.n + kTile - 1) / kTile), dim3(kTile), 0, stream, aos, b);
    return hipGetLastError();
}

hipError_t launch_soa_to_aos(const BodyArrays &b, double *aos, hipStream_t stream)
{
    if (b.n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_soa_to_aos, dim3((b.n + kTile - 1) / kTile), dim3(kTile), 0, stream, b, aos);
    return hipGetLastError();
}

hipError_t launch_contacts_count(const uint32_t *mask, uint32_t n, uint32_t *block_counts, hipStream_t stream)
{
    const uint32_t nb = (n + kScanBlock - 1) / kScanBlock;
    if (nb)
        hipLaunchKernelGGL(k_contacts_block_count, dim3(nb), dim3(kScanBlock), 0, stream, mask, n, block_counts);
    hipLaunchKernelGGL(k_contacts_scan_blocks, dim3(1), dim3(kScanBlock), 0, stream, block_counts, nb);
    return hipGetLastError();
}

hipError_t launch_contacts_emit(const uint32_t *mask, uint32_t n, const uint32_t *block_counts,
                                xpbd_contact *out, uint32_t cap, hipStream_t stream)
{
    const uint32_t nb = (n + kScanBlock - 1) / kScanBlock;
    if (nb)
        hipLaunchKernelGGL(k_contacts_emit, dim3(nb), dim3(kScanBlock), 0, stream, mask, n, block_counts, out, cap);
    return hipGetLastError();
}

namespace {
// The memory system alone on one substep's traffic (xpbd_selftest_field_streams): what XPBD_MODE_PER_SUBSTEP can reach at most
// in this layout -- 38 + 13 concurrent streams of 512 bytes per wave, which HBM serves well below a two-stream copy.
template <bool TILE_MAJOR>
__global__ void __launch_bounds__(64) k_field_streams(const double *__restrict__ in, double *__restrict__ out, size_t n)
{
    constexpr uint32_t kIn = kDynFields + kStatFields, kOut = kDynFields;
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n)
        return;
    const double *src = TILE_MAJOR ? in + (i / 64) * kIn * 64 + i % 64 : in + i;
    double *dst = TILE_MAJOR ? out + (i / 64) * kOut * 64 + i % 64 : out + i;
    const size_t step = TILE_MAJOR ? 64 : n;
    double acc = 0.0;
#pragma unroll
    for (uint32_t f = 0; f < kIn; ++f)
        acc += src[f * step];
#pragma unroll
    for (uint32_t f = 0; f < kOut; ++f)
        dst[f * step] = acc + (double)f;
}
} // namespace

hipError_t launch_field_streams(const double *in, double *out, size_t bodies, bool tile_major, hipStream_t stream)
{
    const dim3 grid((unsigned)((bodies + 63) / 64));
    if (tile_major)
        hipLaunchKernelGGL(k_field_streams<true>, grid, dim3(64), 0, stream, in, out, bodies);
    else
        hipLaunchKernelGGL(k_field_streams<false>, grid, dim3(64), 0, stream, in, out, bodies);
    return hipGetLastError();
}

namespace {
// A GATHER of known size (xpbd_selftest_gather): lane i reads the first READ 16-byte words of record perm(i) of `records`
// records of REC words each -- every record exactly once (perm = multiplication by an odd constant modulo a power of two),
// so no byte is served twice -- and writes one double.  The contact kernels read their body records, mass properties and
// manifolds this way; running this under `rocprofv3 --pmc FETCH_SIZE` calibrates that counter for gathered 16-byte loads
// (the guide calibrates it for wide coalesced streams only).
template <uint32_t READ, uint32_t REC>
__global__ void __launch_bounds__(256) k_gather_records(const double2 *__restrict__ in, double *__restrict__ out, uint32_t mask)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t rec = (i * 2654435761u) & mask;
    const double2 *r = in + (size_t)rec * REC;
    double2 v[READ];
#pragma unroll
    for (uint32_t k = 0; k < READ; ++k)
        v[k] = r[k];
    double acc = 0.0;
#pragma unroll
    for (uint32_t k = 0; k < READ; ++k)
        acc += v[k].x + v[k].y;
    out[i] = acc;
}
} // namespace

hipError_t launch_gather_records(const void *in, double *out, uint32_t records, uint32_t record_bytes, uint32_t read_bytes, hipStream_t stream)
{
    const dim3 grid(records / 256), block(256);
    const double2 *src = static_cast<const double2 *>(in);
    const uint32_t mask = records - 1;
#define XPBD_GATHER_CASE(READ, REC)                                                                      \
    if (read_bytes == (READ) * 16 && record_bytes == (REC) * 16) {                                        \
        hipLaunchKernelGGL((k_gather_records<READ, REC>), grid, block, 0, stream, src, out, mask);       \
        return hipGetLastError();                                                                         \
    }
    XPBD_GATHER_CASE(8, 8)    // a whole 128-byte record (StatRecord)
    XPBD_GATHER_CASE(4, 8)    // the first 64 bytes of a 128-byte record
    XPBD_GATHER_CASE(1, 8)    // 16 bytes of a 128-byte record
    XPBD_GATHER_CASE(12, 12)  // a whole 192-byte record (BodyRecord: every other one straddles two 128-byte lines)
    XPBD_GATHER_CASE(4, 12)   // the first 56..64 bytes of a BodyRecord (what the narrowphase reads)
    XPBD_GATHER_CASE(8, 16)   // the first line of a 256-byte record (a manifold of up to four points)
#undef XPBD_GATHER_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_copy16(const void *src, void *dst, size_t bytes, uint32_t variant, hipStream_t stream)
{
    const size_t n16 = bytes / 16;
    if (n16 == 0)
        return hipSuccess;
    const v4f *s = static_cast<const v4f *>(src);
    v4f *d = static_cast<v4f *>(dst);
    const dim3 flat((uint32_t)((n16 + 255) / 256)), strided(2048); // strided: 256 CUs x 8 workgroups of 4 waves
    switch (variant % kCopyVariants) {
    case 0: hipLaunchKernelGGL(k_copy16_flat<false>, flat, dim3(256), 0, stream, s, d, n16); break;
    case 1: hipLaunchKernelGGL(k_copy16_flat<true>, flat, dim3(256), 0, stream, s, d, n16); break;
    case 2: hipLaunchKernelGGL(k_copy16_strided<false>, strided, dim3(256), 0, stream, s, d, n16); break;
    default: hipLaunchKernelGGL(k_copy16_strided<true>, strided, dim3(256), 0, stream, s, d, n16); break;
    }
    return hipGetLastError();
}

hipError_t launch_selftest_div_sqrt(const double *a, const double *b, double *q, double *r, uint32_t n,
                                    hipStream_t stream)
{
    if (n)
        hipLaunchKernelGGL(k_selftest_div_sqrt, dim3((n + 255) / 256), dim3(256), 0, stream, a, b, q, r, n);
    return hipGetLastError();
}

} // namespace xpbd
