// kernels_resolve.h - per-pixel resolve: sum a pixel's sample radiances in sample order, / spp, w = 1 (main.cpp:242, 262-263).
#pragma once

#include "dev_scene.h"

namespace prt {

// One lane per pixel: sum the spp sample colours in order, divide, w = 1 (main.cpp:235-263).
// FIXED: the samples are the fixed-point accumulators of the wavefront / pool pipelines (dev_scene.h Accum), else float4.
template <bool FIXED>
PRT_D f3 load_sample_rgb(const void * sample_rgb, size_t i) {
    if (FIXED) return accum_read(reinterpret_cast<const Accum *>(sample_rgb) + i);
    const float4 c = reinterpret_cast<const float4 *>(sample_rgb)[i];
    return mk3(c.x, c.y, c.z);
}

// Work item p of the pass that starts at work item `base` -> its place in the call's output (dev_scene.h local_of_work).
struct ResolveMap { unsigned int base, width, tile_pixels, reverse_n, scatter_n, scatter_mul; };
PRT_D size_t resolve_out_index(const ResolveMap & m, unsigned int p) { return local_of_work(m.base + p, m.width, m.tile_pixels, m.reverse_n, m.scatter_n, m.scatter_mul); }

template <bool FIXED>
__global__ void k_resolve(const void * sample_rgb, float4 * out_rgba, unsigned int n_pixels, unsigned int spp, ResolveMap map) {
    const unsigned int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    f3 color = mk3(0.0f, 0.0f, 0.0f);
    for (unsigned int s = 0; s < spp; ++s) color = color + load_sample_rgb<FIXED>(sample_rgb, (size_t)p * spp + s);
    color = color / (float)spp;
    out_rgba[resolve_out_index(map, p)] = make_float4(color.x, color.y, color.z, 1.0f);
}

// The same for spp = 2, 4, ... 64 with coalesced loads: lane l of a wave reads sample (wave base + l), the first lane of
// every group of SPP lanes then adds its neighbours' values one after the other - the reference's summation order - with
// wave shuffles.  (k_resolve's per-lane runs of spp x 16 B make every load instruction touch 64 different cache lines:
// 0.28 ms per 1080p x 8 spp frame; this one: 0.10 ms.)
template <int SPP, bool FIXED>
__global__ __launch_bounds__(256) void k_resolve_pow2(const void * sample_rgb, float4 * out_rgba, unsigned int n_pixels, ResolveMap map) {
    const unsigned long long sid = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    const unsigned long long n_samples = (unsigned long long)n_pixels * SPP;
    f3 c = mk3(0.0f, 0.0f, 0.0f);
    if (sid < n_samples) c = load_sample_rgb<FIXED>(sample_rgb, (size_t)sid);
    const int lane = (int)(threadIdx.x & 63u);
    const int leader = lane & ~(SPP - 1);
    f3 color = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < SPP; ++k) {
        const f3 v = mk3(__shfl(c.x, leader + k), __shfl(c.y, leader + k), __shfl(c.z, leader + k));
        color = color + v;                                      // 0 + s0, + s1, ...: main.cpp:242
    }
    if (lane == leader && sid < n_samples) {
        color = color / (float)SPP;
        out_rgba[resolve_out_index(map, (unsigned int)(sid / SPP))] = make_float4(color.x, color.y, color.z, 1.0f);
    }
}

}  // namespace prt
