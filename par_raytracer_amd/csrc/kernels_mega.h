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
    stack.attach(s_stack, threadIdx.x);
    stack.cap = P.stack_lds_entries;
    stack.set_spill(P.stack_spill, P.stack_spill_stride);
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

}  // namespace prt
