// kernels_persist.h - the persistent-threads pipeline: one launch per frame, no ray queues in HBM.
//
// Measured on the wavefront pipeline (profiles/r01_wavefront_*): its shade kernel moves 16-23 GB of ray /
// hit / frame records per frame through HBM (5.4 ms) and every bounce round costs a host round trip.  This
// kernel keeps each sample's state in the lane that owns it and only batches the CONTROL FLOW:
//
//   every lane is EMPTY (needs a sample), ADVANCE (its ray came back: run the bounce-tree state machine of
//   dev_shade.h until the next ray) or TRAVERSE (walking the BVH).  A wave alternates between
//     phase A  all non-traversing lanes fetch samples from the global sample counter (one atomic per chunk,
//              slots by ballot / mbcnt rank) and run sample_advance() - executed only when at least
//              64 - keep_min lanes need it, so the shading code runs at >= ~40 % lane utilisation instead of
//              once per finished ray;
//   and
//     phase B  the traversal loop of k_trace (node phase until fewer than node_min lanes walk, then leaf phase)
//              until fewer than keep_min lanes are still traversing.
//   Traversal state survives phase A in registers + the LDS stack column; sample state survives phase B in
//   registers (parked frames in per-lane private memory, <= 400 B, L2 resident for the ~400k persistent lanes).
//
// Same sample_advance() as the megakernel, so colours keep the reference's exact float association (only
// device powf differs); same trav_* code as k_trace.  Shadow and closest-hit rays of a sample are traced by
// its lane one after the other, in the reference's order.
#pragma once

#include "kernels_mega.h"
#include "kernels_wave.h"

namespace prt {

enum { LANE_EMPTY = 0, LANE_ADVANCE = 1, LANE_TRAVERSE = 2 };

template <int BLOCK, int MAXLEV, bool RING, bool COUNT>
__global__ __launch_bounds__(BLOCK, 4) void k_render_persistent(DevScene sc, DevCamera cam, DevParams P, unsigned int n_samples,
                                                              float4 * sample_rgb, DevCounters * ctr, u64 * ring_ws,
                                                              unsigned int * head, int keep_min, int node_min,
                                                              unsigned int chunk) {
    extern __shared__ int s_stack[];
    const unsigned int slot = blockIdx.x * BLOCK + threadIdx.x;          // persistent lane id
    const unsigned int lane = lane_id();
    LdsSpillStack<BLOCK> stack;
    stack.attach(s_stack, threadIdx.x);
    stack.cap = P.stack_lds_entries;
    stack.set_spill(P.stack_spill, P.stack_spill_stride);
    u64 * ring = RING ? ring_ws + slot : nullptr;
    const size_t ring_stride = (size_t)gridDim.x * BLOCK;

    int state = LANE_EMPTY;
    unsigned int sid = 0;
    SampleState S;
    S.level = 0; S.phase = PH_START; S.ret = mk3(0, 0, 0);
    S.rng.chain = S.rng.prev = S.rng.seed0 = 0; S.rng.k = 0;
    Frame cur;
    PrivateFrameStore<MAXLEV> store;
    HitRec hit;
    hit.t = 0.0f; hit.v = hit.w = 0.0f; hit.tri = -1;
    TravRay r;
    trav_idle(r);
    TraceStats st;
    st.nodes = st.tris = st.wnodes = st.wleaves = st.wtris = st.wrefills = st.wrays = st.max_sp = st.culled = 0;
    unsigned int rays = 0, shaded = 0;
    unsigned int chunk_next = 0, chunk_end = 0;      // wave-uniform: samples reserved for this wave
    bool exhausted = false;                          // wave-uniform: the sample counter ran past n_samples

    for (;;) {
        // ---- phase A.1: EMPTY lanes take the next samples
        const unsigned long long empty = __ballot(state == LANE_EMPTY);
        if (empty != 0ull && !(exhausted && chunk_next == chunk_end)) {
            if (chunk_next == chunk_end) {
                unsigned int base = 0;
                if (lane == 0) base = atomicAdd(head, chunk);
                base = (unsigned int)__shfl((int)base, 0);
                if (base >= n_samples) {
                    exhausted = true;
                } else {
                    chunk_next = base;
                    chunk_end = base + chunk < n_samples ? base + chunk : n_samples;
                }
            }
            const unsigned int avail = chunk_end - chunk_next;
            if (avail) {
                if (COUNT && lane == 0) st.wrefills++;
                const unsigned int prefix = __builtin_amdgcn_mbcnt_hi((unsigned int)(empty >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)empty, 0u));
                const unsigned int n_empty = (unsigned int)__popcll(empty);
                const unsigned int take = n_empty < avail ? n_empty : avail;
                if (state == LANE_EMPTY && prefix < take) {
                    sid = chunk_next + prefix;
                    sample_begin<RING>(cam, P, pixel_of_local(P, sid / P.spp), sid % P.spp, S, cur, ring, ring_stride);
                    state = LANE_ADVANCE;
                }
                chunk_next += take;
            }
        }
        // ---- phase A.2: lanes whose ray came back (or that just started) run the bounce tree to the next ray
        if (state == LANE_ADVANCE) {
            RayReq req;
            if (sample_advance<RING>(sc, P, S, cur, store, hit, req, shaded, ring, ring_stride)) {
                rays++;                                                     // debug->ray_count++  raytracer.cpp:161
                const f3 ob = req.o + req.d * P.ray_bias;                   // raytracer.cpp:163
                trav_init(r, ob, req.d, req.kind, P.box_pad, stack);
                state = LANE_TRAVERSE;
            } else {
                sample_rgb[sid] = make_float4(S.ret.x, S.ret.y, S.ret.z, 0.0f);
                state = LANE_EMPTY;
            }
        }
        const unsigned long long live = __ballot(state != LANE_EMPTY);
        if (live == 0ull) {
            if (exhausted && chunk_next == chunk_end) break;                // nothing left anywhere
            continue;                                                       // samples remain: fetch again
        }

        // ---- phase B: traverse until too few of the wave's live lanes are still walking
        const int n_live = __popcll(live);
        int leave_below = (n_live * keep_min) >> 6;
        if (leave_below < 1) leave_below = 1;
        while (state == LANE_TRAVERSE) {
            const int walkers = __popcll(__ballot(trav_walking(r)));
            const int nmin = node_min < (walkers >> 1) ? node_min : (walkers >> 1);
            while (trav_walking(r)) {
                trav_node_step<LdsSpillStack<BLOCK>, COUNT>(sc, r, stack, st, P.box_pad);
                if (__popcll(__ballot(trav_walking(r))) < nmin) break;
            }
            bool fin = trav_done(r);
            if (!fin && !trav_walking(r)) fin = trav_leaf<LdsSpillStack<BLOCK>, COUNT>(sc, r, stack, st);
            if (fin) {
                if (trav_wants_resolve(r, stack)) r.best = resolve_near_ties<LdsSpillStack<BLOCK>, COUNT>(sc, r.o, r.d, P.box_pad, r.best.t, stack, st);
                hit = r.best;
                state = LANE_ADVANCE;
                break;
            }
            if (__popcll(__ballot(true)) < leave_below) break;
        }
    }
    flush_counters(ctr, rays, shaded, st, COUNT);
    if (COUNT) {
        atomicAdd(&ctr->wave_node_steps, (unsigned long long)st.wnodes);
        atomicAdd(&ctr->wave_leaf_steps, (unsigned long long)st.wleaves);
        atomicAdd(&ctr->wave_tri_steps, (unsigned long long)st.wtris);
        atomicAdd(&ctr->wave_refills, (unsigned long long)st.wrefills);
    }
}

}  // namespace prt
