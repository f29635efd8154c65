// kernels_mega.h - "one lane per (pixel, sample)" kernel: ray generation, traversal and shading in-lane.
//
// First slice of the build plan and the cross-check for the wavefront pipeline:
// each lane owns one sample, runs sample_advance() until it asks for a ray, traces it through the common
// traversal loop and feeds the hit back.  All lanes of a wave re-converge at the single trace_ray() call
// site every iteration, whatever recursion level each is at.  Sample colours go to a per-sample buffer;
// a second kernel sums them per pixel in sample order (the reference's `color += scratch[samp]`,
// main.cpp:242) and divides by spp (main.cpp:262-263).
#pragma once

#include "../../include/prt_key.h"
#include "dev_shade.h"

namespace prt {

template <int MAXLEV>
struct PrivateFrameStore {
    Frame slots[MAXLEV];
    PRT_D void save(int level, const Frame & f) { slots[level] = f; }
    PRT_D void load(int level, Frame & f) { f = slots[level]; }
};

PRT_D void flush_counters(DevCounters * ctr, unsigned int rays, unsigned int shaded, const TraceStats & st, bool count) {
    // one atomic per wave and counter: the compiler's atomic optimiser turns these into a DPP reduction
    atomicAdd(&ctr->ray_count, (unsigned long long)rays);
    atomicAdd(&ctr->shaded_hits, (unsigned long long)shaded);
    if (count) {
        atomicAdd(&ctr->node_visits, (unsigned long long)st.nodes);
        atomicAdd(&ctr->tri_tests, (unsigned long long)st.tris);
    }
}

// grid: ceil(n_samples / BLOCK); dynamic LDS: stack_entries * BLOCK * 4 bytes.
template <int BLOCK, int MAXLEV, bool RING, bool COUNT>
__global__ __launch_bounds__(BLOCK) void k_render_mega(DevScene sc, DevCamera cam, DevParams P, unsigned int n_samples,
                                                       float4 * sample_rgb, DevCounters * ctr, u64 * ring_ws) {
    extern __shared__ int s_stack[];
    const unsigned int sid = blockIdx.x * BLOCK + threadIdx.x;
    if (sid >= n_samples) return;
    LdsSpillStack<BLOCK> stack;
    stack.col = s_stack + threadIdx.x;
    stack.cap = P.stack_lds_entries;
    stack.spill = P.stack_spill;
    stack.spill_stride = P.stack_spill_stride;
    const unsigned int pixel = pixel_of_local(P, sid / P.spp);
    const unsigned int samp = sid % P.spp;
    u64 * ring = RING ? ring_ws + sid : nullptr;
    const size_t ring_stride = n_samples;

    SampleState S;
    Frame cur;
    PrivateFrameStore<MAXLEV> store;
    sample_begin<RING>(cam, P, pixel, samp, S, cur, ring, ring_stride);

    HitRec hit;
    hit.t = 0.0f; hit.v = hit.w = 0.0f; hit.tri = -1;
    RayReq req;
    TraceStats st;
    st.nodes = st.tris = st.wnodes = st.wleaves = st.wtris = st.wrefills = st.wrays = st.max_sp = st.culled = 0;
    unsigned int rays = 0, shaded = 0;
    while (sample_advance<RING>(sc, P, S, cur, store, hit, req, shaded, ring, ring_stride)) {
        rays++;                                                     // debug->ray_count++  raytracer.cpp:161
        const f3 ob = req.o + req.d * P.ray_bias;                   // raytracer.cpp:163
        hit = trace_ray<LdsSpillStack<BLOCK>, COUNT>(sc, ob, req.d, req.kind, P.box_pad, stack, st);
    }
    sample_rgb[sid] = make_float4(S.ret.x, S.ret.y, S.ret.z, 0.0f);
    flush_counters(ctr, rays, shaded, st, COUNT);
}

// One lane per pixel: sum the spp sample colours in order, divide, w = 1 (main.cpp:235-263).
// FIXED: the samples are the fixed-point accumulators of the wavefront / pool pipelines (dev_scene.h Accum), else float4.
template <bool FIXED>
PRT_D f3 load_sample_rgb(const void * sample_rgb, size_t i) {
    if (FIXED) return accum_read(reinterpret_cast<const Accum *>(sample_rgb) + i);
    const float4 c = reinterpret_cast<const float4 *>(sample_rgb)[i];
    return mk3(c.x, c.y, c.z);
}

// Work item p of the pass that starts at work item `base` -> its place in the call's output (dev_scene.h local_of_work).
struct ResolveMap { unsigned int base, width, tile_pixels; };
PRT_D size_t resolve_out_index(const ResolveMap & m, unsigned int p) { return local_of_work(m.base + p, m.width, m.tile_pixels); }

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
