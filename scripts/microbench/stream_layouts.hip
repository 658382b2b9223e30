// Micro-benchmark behind DESIGN.md 9: how fast does MI355X stream the pinned path's per-substep traffic (38 doubles in,
// 13 doubles out per body) when the bodies are laid out FIELD-MAJOR (today: 38 + 13 concurrent streams, 512 bytes per
// wave and stream) and when they are laid out TILE-MAJOR (64 bodies x all fields contiguous: one stream in, one out)?
// No arithmetic worth mentioning: this is the memory system alone.
// build: hipcc --offload-arch=gfx950 -O3 -o stream_layouts stream_layouts.hip      run: ./stream_layouts [bodies]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

constexpr int kIn = 38, kOut = 13, kTile = 64;

// field-major: field f of body i at base[f * stride + i]
__global__ void k_field_major(const double *__restrict__ in, double *__restrict__ out, size_t stride, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    double acc = 0.0;
#pragma unroll
    for (int f = 0; f < kIn; ++f)
        acc += in[(size_t)f * stride + i];
#pragma unroll
    for (int f = 0; f < kOut; ++f)
        out[(size_t)f * stride + i] = acc + f;
}

// tile-major: field f of body i at base[(i / 64) * fields * 64 + f * 64 + i % 64]
__global__ void k_tile_major(const double *__restrict__ in, double *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const size_t tile = i / kTile, lane = i % kTile;
    const double *src = in + tile * kIn * kTile + lane;
    double *dst = out + tile * kOut * kTile + lane;
    double acc = 0.0;
#pragma unroll
    for (int f = 0; f < kIn; ++f)
        acc += src[f * kTile];
#pragma unroll
    for (int f = 0; f < kOut; ++f)
        dst[f * kTile] = acc + f;
}

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 2097152;
    double *in = nullptr, *out = nullptr;
    if (hipMalloc(&in, n * kIn * 8) != hipSuccess || hipMalloc(&out, n * kOut * 8) != hipSuccess)
        return 1;
    hipMemset(in, 0, n * kIn * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double bytes = (double)n * (kIn + kOut) * 8;
    // field-major again with the stride between two fields padded off the power of two (stride = n: all 51 streams of a wave sit
    // at the same offset modulo every power of two up to n * 8 bytes, i.e. on the same channel and bank at the same time)
    for (size_t pad : {(size_t)0, (size_t)64, (size_t)320, (size_t)4160, (size_t)65600}) {
        double *pin = nullptr, *pout = nullptr;
        const size_t stride = n + pad;
        if (hipMalloc(&pin, stride * kIn * 8) != hipSuccess || hipMalloc(&pout, stride * kOut * 8) != hipSuccess)
            return 1;
        hipMemset(pin, 0, stride * kIn * 8);
        const int block = 64;
        const dim3 grid((unsigned)((n + block - 1) / block));
        float best = 1e30f;
        for (int rep = 0; rep < 12; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_field_major, grid, dim3(block), 0, 0, pin, pout, stride, n);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2 && ms < best)
                best = ms;
        }
        printf("field-major stride n + %6zu: %8.1f us  %6.2f TB/s\n", pad, best * 1e3, bytes / (best * 1e-3) / 1e12);
        hipFree(pin);
        hipFree(pout);
    }
    for (int layout = 0; layout < 2; ++layout)
        for (int block : {64, 256}) {
            const dim3 grid((unsigned)((n + block - 1) / block));
            float best = 1e30f;
            for (int rep = 0; rep < 12; ++rep) {
                hipEventRecord(e0);
                if (layout == 0)
                    hipLaunchKernelGGL(k_field_major, grid, dim3(block), 0, 0, in, out, n, n);
                else
                    hipLaunchKernelGGL(k_tile_major, grid, dim3(block), 0, 0, in, out, n);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2 && ms < best)
                    best = ms;
            }
            printf("%-11s block %3d: %8.1f us  %6.2f TB/s (%zu bodies, %.0f MB per launch)\n", layout ? "tile-major" : "field-major", block,
                   best * 1e3, bytes / (best * 1e-3) / 1e12, n, bytes / 1e6);
        }
    return 0;
}
