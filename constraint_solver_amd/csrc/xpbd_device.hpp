// xpbd_device.hpp -- device helpers shared by the kernels: SoA field access and Rigid::frame.
#pragma once

#include "xpbd_kernels.h"
#include "xpbd_math.hpp"

namespace xpbd {

__device__ __forceinline__ Vec3 load3(const double *base, uint32_t field, uint32_t stride, uint32_t i)
{
    return Vec3{base[(size_t)(field + 0) * stride + i], base[(size_t)(field + 1) * stride + i],
                base[(size_t)(field + 2) * stride + i]};
}

__device__ __forceinline__ void store3(double *base, uint32_t field, uint32_t stride, uint32_t i, Vec3 v)
{
    base[(size_t)(field + 0) * stride + i] = v.x;
    base[(size_t)(field + 1) * stride + i] = v.y;
    base[(size_t)(field + 2) * stride + i] = v.z;
}

__device__ __forceinline__ Quat load_quat(const double *base, uint32_t field, uint32_t stride, uint32_t i)
{
    return Quat{base[(size_t)(field + 0) * stride + i], base[(size_t)(field + 1) * stride + i],
                base[(size_t)(field + 2) * stride + i], base[(size_t)(field + 3) * stride + i]};
}

// Rigid::frame().position  (src/rigid.rs:77): (position + com) + rotation * (-com)
__device__ __forceinline__ Vec3 frame_origin(Vec3 pos, Quat rot, Vec3 com)
{
    return (pos + com) + rot * (-com);
}

// Rigid::frame() of body i from the SoA arrays (src/rigid.rs:75-80).
__device__ __forceinline__ Frame body_frame(const BodyArrays &b, uint32_t i)
{
    const Vec3 pos = load3(b.dyn, D_POS, b.stride, i);
    const Quat rot = load_quat(b.dyn, D_ROT, b.stride, i);
    const Vec3 com = load3(b.stat, S_COM, b.stride, i);
    return Frame{frame_origin(pos, rot, com), rot};
}

// A frame stored as 7 SoA fields (origin xyz, rotation s x y z).
__device__ __forceinline__ Frame load_frame(const double *frames, uint32_t stride, uint32_t i)
{
    return Frame{load3(frames, 0, stride, i), load_quat(frames, 3, stride, i)};
}

__device__ __forceinline__ void store_frame(double *frames, uint32_t stride, uint32_t i, const Frame &f)
{
    store3(frames, 0, stride, i, f.position);
    frames[(size_t)3 * stride + i] = f.rotation.s;
    frames[(size_t)4 * stride + i] = f.rotation.x;
    frames[(size_t)5 * stride + i] = f.rotation.y;
    frames[(size_t)6 * stride + i] = f.rotation.z;
}

} // namespace xpbd
