// kernels_flow.h - the pool pipeline WITHOUT ROUNDS (round 4, experimental: option POOL_FLOW).
//
// What the round-based pool kernel (kernels_pool.h) loses, it loses at the end of every round: once a round's ray list is dry
// the wave walks on with the rays still in flight, fewer and fewer, until the last one is done - 13 % of a C4 frame's
// wave-level node steps at 30 % lane fill, 38 % of a 1/8 shard's, 49 % of the adaptive mode's (profiles/r04_round_drains.txt) -
// and a sample's chain of bounces advances one ROUND per bounce.  Here a workgroup has no rounds:
//
//   three TRACER waves  pull rays - closest-hit ray slots from a ring of slot ids, shadow rays from a ring of entries - as their
//                       lanes fall idle (the persistent traversal loop of k_trace / k_pool, refilled below keep_min busy lanes),
//                       write a finished closest-hit ray's hit record to its slot and the slot id to their own "done" ring,
//                       add a finished shadow ray's radiance;
//   one SHADING wave    (a different wave of every workgroup, so that every SIMD of a compute unit hosts tracers and shaders
//                       alike) takes finished slots from the three done rings, up to 64 at a time, runs shade_entry() on them -
//                       the bounce-tree state machine of kernels_wave.h, unchanged -, writes each sample's next ray into the SAME
//                       slot and pushes the slot id to the tracers' ring, pushes the hit's shadow rays to theirs, frees the slot
//                       when the sample has ended, and tops the slots up with fresh samples from the global sample counter.
//
// Rings and slots live in HBM (structure of arrays, as the pool kernel's lists); their head / tail words live in LDS.  Every
// ring has ONE producer: the shading wave for the two ray rings, each tracer wave for its own done ring - so a tail is a plain
// word published after the entries (workgroup-scope release; the waves of a workgroup share their compute unit's L1).  The ray
// rings have three consumers, which claim ranges with a compare-and-swap on the head word; an entry a consumer has read is marked
// (slot id ring: 0xFFFFFFFF; shadow ring: sample = -1) and the producer writes an entry only where it finds the mark, so a ring
// can never overwrite what has been claimed but not yet read, whatever its size.  Nothing crosses a workgroup: per-sample
// state (RNG, pending frames) is touched by the shading wave alone; radiance is added with atomics by everyone (a sample's
// shadow rays and its next hit are no longer a phase apart: Emit::ATOMIC_RADIANCE).
//
// Rays whose hit has company within a few ulp, or whose stack column overflowed, are parked exactly as in k_pool (global park
// lists) and finished by the same follow-up launches (k_pool_parked_shadows, the adopting EXACT k_pool).  Results are the pool
// kernel's bit for bit (same functions, same keys, fixed-point radiance sums).
//
// Every wait has a watchdog: a wave that spins longer than FLOW_SPIN_LIMIT polls raises the kernel's error flag and every
// wave leaves; the render call then fails instead of hanging the device.
#pragma once

#include "kernels_pool.h"

namespace prt {

struct FlowBuffers {
    unsigned int * crq;          // [blocks][rc]      ring of slot ids to trace (0xFFFFFFFF = read)
    unsigned int * done;         // [blocks][3][rd]   per tracer wave: ring of slot ids whose hit record is written
    unsigned int * freelist;     // [blocks][slots]   the shading wave's stack of free slots
    unsigned int slots;          // ray slots per workgroup = samples in flight per workgroup (multiple of 64)
    unsigned int rc_mask, rd_mask, rs_mask;       // ring sizes - 1 (powers of two); rs: entries of the shadow ring (PoolBuffers::sq, [blocks][3][rs])
    unsigned int topup_min, topup_max;            // fresh samples are fetched when this many slots are free / at most so many at once
    unsigned int low_water;      // the shading wave shades partial batches (and tops up first) while the tracers' ring holds fewer ids than this
    unsigned int * error;        // set by a watchdog
};

enum { FLOW_C_HEAD = 0, FLOW_C_TAIL, FLOW_S_HEAD, FLOW_S_TAIL, FLOW_D_TAIL0, FLOW_D_TAIL1, FLOW_D_TAIL2, FLOW_QUIT, FLOW_WORDS };
enum { FLOW_INVALID = 0xFFFFFFFFu, FLOW_SHADOW_RAY = 0x40000000, FLOW_SPIN_LIMIT = 1 << 22 };

PRT_D unsigned int flow_load(const unsigned int * w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
PRT_D void flow_store(unsigned int * w, unsigned int v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
PRT_D void flow_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
PRT_D void flow_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
PRT_D unsigned int wave_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
}

// Emitter of the shading wave.  One producer: the fill positions are wave-uniform registers, published by the caller after the batch.
struct FlowEmit {
    enum { KEEPS_RNG = 0, ATOMIC_RADIANCE = 1 };
    float4 * co, * cd, * ct;     // the workgroup's ray slots
    float4 * so, * sc, * sd;     // its shadow ring
    unsigned int * crq, * fs;    // its slot id ring, its free-slot stack
    unsigned int * s_f;          // LDS control words
    unsigned int * error;
    unsigned int rc_mask, rs_mask;
    unsigned int c_tail, s_tail, free_count;     // wave-uniform
    unsigned int m_elided, n_pushed;             // wave-uniform: shadow rays counted, not traced; rays pushed (closest + shadow)
    unsigned int my_slot;        // per lane: the slot of the hit this lane shades
    bool release_slot;           // per lane: the slot is to be freed even though no sample ended here (a parked ray's)
    bool failed;                 // wave-uniform: a watchdog fired
    PRT_D void elided(bool dead) { m_elided += (unsigned int)__popcll(__ballot(dead)); }
    PRT_D void shadow(bool want, unsigned int s, f3 o, f3 d, f3 contrib, float w, int kind) {
        const unsigned long long mask = __ballot(want);
        if (want) {
            const unsigned int pos = (s_tail + wave_rank(mask)) & rs_mask;
            // the entry this one replaces must have been READ by the tracer that claimed it (sample = -1)
            unsigned int spins = 0;
            while (__hip_atomic_load(reinterpret_cast<const int *>(&so[pos].w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != -1) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (unsigned int)FLOW_SPIN_LIMIT) { atomicOr(error, 1u); break; }
            }
            if (kind == WF_KIND_SHADOW_DIST) sd[pos] = make_float4(d.x, d.y, d.z, 0.0f);
            sc[pos] = make_float4(contrib.x, contrib.y, contrib.z, w);
            so[pos] = make_float4(o.x, o.y, o.z, as_f((int)s));
        }
        const unsigned int n = (unsigned int)__popcll(mask);
        s_tail += n;
        n_pushed += n;
    }
    PRT_D unsigned int closest(bool want, unsigned int s, f3 o, f3 d, f3 T, int level, unsigned int pending, bool sample_ended) {
        const unsigned long long mask = __ballot(want);
        if (want) {
            co[my_slot] = make_float4(o.x, o.y, o.z, as_f((int)s));
            cd[my_slot] = make_float4(d.x, d.y, d.z, as_f(level | (int)(pending << 8)));
            ct[my_slot] = make_float4(T.x, T.y, T.z, 0.0f);
            const unsigned int pos = (c_tail + wave_rank(mask)) & rc_mask;
            unsigned int spins = 0;
            while (flow_load(&crq[pos]) != (unsigned int)FLOW_INVALID) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (unsigned int)FLOW_SPIN_LIMIT) { atomicOr(error, 2u); break; }
            }
            crq[pos] = my_slot;
        }
        const unsigned int n = (unsigned int)__popcll(mask);
        c_tail += n;
        n_pushed += n;
        // slots whose sample has ended (or whose ray was parked: the sample continues in the follow-up launches) are free again
        const bool give = sample_ended || release_slot;
        const unsigned long long fm = __ballot(give);
        if (give) fs[free_count + wave_rank(fm)] = my_slot;
        free_count += (unsigned int)__popcll(fm);
        return my_slot;
    }
};

struct FlowArgs {
    PoolArgs pool;               // scene, camera, parameters, per-sample buffers, park lists, slots (Q.cq, Q.hits, Q.sq), sample counter
    FlowBuffers F;
};

PRT_D FlowBuffers flow_buffers(const FlowArgs * args) {
    asm volatile("" : "+s"(args));
    FlowBuffers F;
    typedef const FlowBuffers __attribute__((address_space(4))) * Ptr;
    __builtin_memcpy(&F, (Ptr)&args->F, sizeof(FlowBuffers));
    F.crq = as_global(F.crq); F.done = as_global(F.done); F.freelist = as_global(F.freelist); F.error = as_global(F.error);
    return F;
}

// `follow`: the PoolArgs of the launches behind k_flow (k_pool_parked_shadows, the adopting EXACT k_pool) - the pool kernel's own,
// with ITS wave-private list layout over the same buffers (k_flow's slots and rings are laid out differently: FlowArgs::pool)
__global__ void k_flow_store_args(FlowArgs a, PoolArgs follow, FlowArgs * dst, PoolArgs * pool_dst, unsigned int * zero, unsigned int n_zero,
                                  unsigned int * adopt_head) {
    if (threadIdx.x < n_zero) zero[threadIdx.x] = 0u;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        dst[0] = a;
        pool_dst[0] = follow;
        follow.Q.adopt = 1;
        follow.Q.head = adopt_head;
        pool_dst[1] = follow;
    }
}

// grid = resident workgroups of 256 threads; dynamic LDS = the traversal stack columns (the shading wave's double as its frames).
template <int BLOCK, int WAVES, bool RING, bool COUNT, bool TEX, int RINGMEM>
__global__ __launch_bounds__(BLOCK, WAVES) void k_flow(const FlowArgs * fargs, DevCounters * ctr) {
    static_assert(BLOCK == 256, "three tracer waves and one shading wave");
    extern __shared__ int s_stack[];
    constexpr int LDS_MATS = 32, LDS_LIGHTS = 4;
    __shared__ DevMaterial s_mats[LDS_MATS];
    __shared__ DevLight s_lights[LDS_LIGHTS];
    __shared__ unsigned long long s_red[2];
    __shared__ unsigned int s_f[FLOW_WORDS];
    const PoolArgs * const args = &fargs->pool;
    {
        const PoolArgs A0 = pool_args(args);
        const DevScene & sc = A0.sc;
        if (sc.material_count <= (unsigned int)LDS_MATS) {
            const float4 * src = reinterpret_cast<const float4 *>(sc.materials);
            float4 * dst = reinterpret_cast<float4 *>(s_mats);
            for (unsigned int k = threadIdx.x; k < sc.material_count * 4u; k += BLOCK) dst[k] = src[k];
        }
        if (sc.light_count <= (unsigned int)LDS_LIGHTS) {
            const float4 * src = reinterpret_cast<const float4 *>(sc.lights);
            float4 * dst = reinterpret_cast<float4 *>(s_lights);
            for (unsigned int k = threadIdx.x; k < sc.light_count * 3u; k += BLOCK) dst[k] = src[k];
        }
    }
    if (threadIdx.x < 2) s_red[threadIdx.x] = 0ull;
    if (threadIdx.x < (unsigned int)FLOW_WORDS) s_f[threadIdx.x] = 0u;
    const unsigned int block = blockIdx.x;
    {
        // every entry of the two ray rings starts out "read": the producer may write it
        const PoolArgs A0 = pool_args(args);
        const FlowBuffers F0 = flow_buffers(fargs);
        unsigned int * const crq0 = F0.crq + (size_t)block * (F0.rc_mask + 1u);
        float4 * const so0 = A0.Q.sq + (size_t)block * 3u * (F0.rs_mask + 1u);
        for (unsigned int k = threadIdx.x; k <= F0.rc_mask; k += BLOCK) crq0[k] = (unsigned int)FLOW_INVALID;
        for (unsigned int k = threadIdx.x; k <= F0.rs_mask; k += BLOCK) so0[k].w = as_f(-1);
    }
    __syncthreads();

    const unsigned int lane = lane_id();
    const unsigned int wid = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned int shader_wave = block & 3u;                    // every SIMD of a compute unit hosts tracers and shaders of its workgroups alike
    LdsStack<BLOCK> stack;
    stack.attach(s_stack, threadIdx.x);
    stack.cap = ((PoolArgsPtr)args)->P.stack_lds_entries;

    unsigned long long rays = 0ull;            // wave-uniform (shading wave): rays produced = TraceRay calls (raytracer.cpp:161)
    unsigned int shaded_w = 0, elided_w = 0;
    TraceStats st;
    st.nodes = st.tris = st.wnodes = st.wleaves = st.wtris = st.wrefills = st.wrays = st.max_sp = st.culled = 0;
    // COUNT: where the waves' time goes (wave cycles): tracers [0] waiting for rays, [1] whole loop; shading wave [2] topping up,
    // [3] shading, [4] waiting, [5] whole loop; [6] shade batches, [7] hits in them
    unsigned long long fl[8] = { 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull };
    const unsigned long long fl_begin = COUNT ? __builtin_readcyclecounter() : 0ull;

// this workgroup's slots and rings, from the arguments as re-read at this point
#define PRT_FLOW_BUFFERS(A, F)                                                                         \
    const unsigned int slots = (F).slots, rs = (F).rs_mask + 1u;                                      \
    float4 * const co = (A).Q.cq + (size_t)block * 3u * slots;                                        \
    float4 * const cd = co + slots;                                                                   \
    float4 * const ct = cd + slots;                                                                   \
    float4 * const hits = (A).Q.hits + (size_t)block * slots;                                         \
    float4 * const sq_o = (A).Q.sq + (size_t)block * 3u * rs;                                         \
    float4 * const sq_c = sq_o + rs;                                                                  \
    float4 * const sq_d = sq_c + rs;                                                                  \
    unsigned int * const crq = (F).crq + (size_t)block * ((F).rc_mask + 1u);                          \
    unsigned int * const done_base = (F).done + (size_t)block * 3u * ((F).rd_mask + 1u);              \
    (void)ct; (void)hits; (void)sq_d; (void)crq; (void)done_base

    if (wid != shader_wave) {
        // =================================================================================================== tracer waves
        const unsigned int ti = wid < shader_wave ? wid : wid - 1u;      // 0 .. 2: which done ring is this wave's
        const PoolArgs A = pool_args(args);
        const FlowBuffers F = flow_buffers(fargs);
        const DevScene & sc = A.sc;
        const DevParams & P = A.P;
        const WaveBuffers & B = A.B;
        const int keep_min = A.keep_min, node_min = A.node_min, node_frac = A.node_frac;
        PRT_FLOW_BUFFERS(A, F);
        unsigned int * const done = done_base + (size_t)ti * (F.rd_mask + 1u);
        const DevLight * lights = sc.light_count <= (unsigned int)LDS_LIGHTS ? s_lights : sc.lights;
        TravRay r;
        trav_idle(r);
        int ray = -1;                                  // -1 idle; a slot id: closest-hit ray; FLOW_SHADOW_RAY: shadow ray
        int done_id = -1;                              // slot whose hit record this lane has just written
        float4 payload = make_float4(0, 0, 0, 0);
        float4 shadow_o = make_float4(0, 0, 0, 0);     // a shadow ray's list entry (origin, sample), kept for the park list
        int sample = 0;
        unsigned int d_tail = 0;                       // wave-uniform: entries of this wave's done ring
        unsigned int idle_spins = 0;
        for (;;) {
            const unsigned long long idle = __ballot(ray < 0);
            bool more = false;                         // wave-uniform: the rings still held rays after this wave's claim
            if (idle != 0ull) {
                const unsigned int n_idle = (unsigned int)__popcll(idle);
                unsigned int base_c = 0, take_c = 0, base_s = 0, take_s = 0, left = 0;
                if (lane == 0) {
                    for (;;) {
                        const unsigned int h = flow_load(&s_f[FLOW_C_HEAD]), t = flow_load(&s_f[FLOW_C_TAIL]);
                        const unsigned int av = t - h, tk = n_idle < av ? n_idle : av;
                        if (tk == 0u) break;
                        if (atomicCAS(&s_f[FLOW_C_HEAD], h, h + tk) == h) { base_c = h; take_c = tk; left = av - tk; break; }
                    }
                    const unsigned int rem = n_idle - take_c;
                    for (; rem != 0u;) {
                        const unsigned int h = flow_load(&s_f[FLOW_S_HEAD]), t = flow_load(&s_f[FLOW_S_TAIL]);
                        const unsigned int av = t - h, tk = rem < av ? rem : av;
                        if (tk == 0u) break;
                        if (atomicCAS(&s_f[FLOW_S_HEAD], h, h + tk) == h) { base_s = h; take_s = tk; left += av - tk; break; }
                    }
                }
                base_c = (unsigned int)__builtin_amdgcn_readfirstlane((int)base_c); take_c = (unsigned int)__builtin_amdgcn_readfirstlane((int)take_c);
                base_s = (unsigned int)__builtin_amdgcn_readfirstlane((int)base_s); take_s = (unsigned int)__builtin_amdgcn_readfirstlane((int)take_s);
                more = __builtin_amdgcn_readfirstlane((int)left) != 0;
                if (take_c + take_s) {
                    if (COUNT && lane == 0) st.wrefills++;
                    flow_acquire();
                    const unsigned int prefix = wave_rank(idle);
                    if (ray < 0 && prefix < take_c + take_s) {
                        float4 ro, rd;
                        int kind;
                        if (prefix < take_c) {
                            const unsigned int pos = (base_c + prefix) & F.rc_mask;
                            const unsigned int id = crq[pos];
                            flow_store(&crq[pos], (unsigned int)FLOW_INVALID);                      // read: the ring may reuse the entry
                            ro = co[id];
                            rd = cd[id];
                            kind = WF_KIND_CLOSEST;
                            ray = (int)id;
                        } else {
                            const unsigned int pos = (base_s + prefix - take_c) & F.rs_mask;
                            ro = sq_o[pos];
                            payload = sq_c[pos];
                            if (payload.w < 0.0f) {          // directional light: the direction is a per-light constant
                                const DevLight & L = lights[(unsigned int)(-payload.w) - 1u];
                                const f3 lv = mk3(L.facing[0], L.facing[1], L.facing[2]) * -1.0f;   // raytracer.cpp:240
                                rd = make_float4(lv.x, lv.y, lv.z, 0.0f);
                                kind = WF_KIND_SHADOW_ANY;
                            } else {
                                rd = sq_d[pos];
                                kind = WF_KIND_SHADOW_DIST;
                            }
                            shadow_o = ro;
                            __hip_atomic_store(reinterpret_cast<int *>(&sq_o[pos].w), -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // read
                            ray = (int)FLOW_SHADOW_RAY;
                        }
                        sample = as_i(ro.w);
                        const f3 d = mk3(rd.x, rd.y, rd.z);
                        const f3 ob = mk3(ro.x, ro.y, ro.z) + d * P.ray_bias;          // raytracer.cpp:163
                        trav_init(r, ob, d, kind == WF_KIND_SHADOW_ANY ? TRACE_ANY : TRACE_CLOSEST, P.box_pad, stack);
                    }
                }
            }
            const unsigned int busy = (unsigned int)__popcll(__ballot(ray >= 0));
            if (busy == 0u) {
                if (flow_load(&s_f[FLOW_QUIT]) != 0u) break;
                const unsigned long long w0 = COUNT ? __builtin_readcyclecounter() : 0ull;
                __builtin_amdgcn_s_sleep(4);
                if (COUNT) fl[0] += __builtin_readcyclecounter() - w0 + 40ull;
                if (++idle_spins > (unsigned int)FLOW_SPIN_LIMIT) { if (lane == 0) { atomicOr(F.error, 4u); flow_store(&s_f[FLOW_QUIT], 1u); } break; }
                continue;
            }
            idle_spins = 0;

            // come back for more rays when fewer than keep_min lanes are busy and the rings had more; if they had none, when half
            // of what is in flight is done (the shading wave may have produced more by then)
            const int leave_below = more ? keep_min : (int)((busy + 1u) >> 1);
            while (ray >= 0) {
                const int walkers = __popcll(__ballot(trav_walking(r)));
                const int wfrac = (walkers * node_frac) >> 3;
                const int nmin = node_min < wfrac ? node_min : wfrac;
                const unsigned int with_ray = COUNT ? (unsigned int)__popcll(__ballot(true)) : 0u;
                while (trav_walking(r)) {
                    trav_node_step<LdsStack<BLOCK>, COUNT>(sc, r, stack, st, P.box_pad);
                    if (COUNT && first_active_lane()) st.wrays += with_ray;
                    if (__popcll(__ballot(trav_walking(r))) < nmin) break;
                }
                bool fin = trav_done(r);
                if (!fin && !trav_walking(r)) fin = trav_leaf<LdsStack<BLOCK>, COUNT>(sc, r, stack, st);
                if (fin) {
                    const bool is_closest = ray != (int)FLOW_SHADOW_RAY;
                    if (trav_needs_slow_path(r, stack)) {
                        // the hit has company within a few ulp, or a push did not fit the LDS column: parked for the follow-up launches
                        if (is_closest) {
                            const unsigned int slot = atomicAdd(A.Q.park_count, 1u);
                            if (slot < A.Q.park_cap) {
                                const float4 t4 = ct[ray];
                                A.Q.park[slot] = co[ray];
                                A.Q.park[(size_t)A.Q.park_cap + slot] = cd[ray];
                                A.Q.park[2u * (size_t)A.Q.park_cap + slot] = make_float4(t4.x, t4.y, t4.z, as_f(POOL_PARK_CLOSEST));
                            }
                            hits[ray] = make_float4(0.0f, 0.0f, 0.0f, as_f(POOL_PARKED_MARK));     // its sample leaves this workgroup
                        } else {
                            const unsigned int slot = atomicAdd(A.Q.park_count + 1, 1u);
                            if (slot < A.Q.spark_cap) {
                                A.Q.spark[slot] = shadow_o;
                                A.Q.spark[(size_t)A.Q.spark_cap + slot] = payload;
                                A.Q.spark[2u * (size_t)A.Q.spark_cap + slot] = make_float4(r.d.x, r.d.y, r.d.z, 0.0f);
                            }
                        }
                    } else if (is_closest) {
                        hits[ray] = make_float4(r.best.t, r.best.v, r.best.w, as_f(r.best.tri));
                    } else {
                        const bool lit = r.best.tri < 0 || (payload.w >= 0.0f && r.best.t * r.best.t <= payload.w);
                        if (lit) accum_add(B.accum + sample, mk3(payload.x, payload.y, payload.z));
                    }
                    if (is_closest) done_id = ray;
                    ray = -1;
                    break;
                }
                if (__popcll(__ballot(true)) < leave_below) break;
            }
            // ---- finished closest-hit rays: their slots go to this wave's done ring (one producer: a plain tail)
            const unsigned long long dm = __ballot(done_id >= 0);
            if (dm != 0ull) {
                if (done_id >= 0) done[(d_tail + wave_rank(dm)) & F.rd_mask] = (unsigned int)done_id;
                done_id = -1;
                d_tail += (unsigned int)__popcll(dm);
                flow_release();
                if (lane == 0) flow_store(&s_f[FLOW_D_TAIL0 + ti], d_tail);
            }
        }
        if (COUNT) fl[1] = __builtin_readcyclecounter() - fl_begin;
    } else {
        // =================================================================================================== the shading wave
        unsigned int d_head[3] = { 0u, 0u, 0u };       // wave-uniform: what it has taken from the tracers' done rings
        unsigned int c_tail = 0, s_tail = 0;           // what it has pushed to the ray rings
        unsigned int free_count;
        bool fetch_done = false;
        unsigned int spins = 0;
        {
            const FlowBuffers F = flow_buffers(fargs);
            unsigned int * const fs = F.freelist + (size_t)block * F.slots;
            for (unsigned int k = lane; k < F.slots; k += 64u) fs[k] = F.slots - 1u - k;        // slot 0 on top
            free_count = F.slots;
        }
        for (;;) {
            const PoolArgs A = pool_args(args);
            const FlowBuffers F = flow_buffers(fargs);
            const DevScene & sc = A.sc;
            const DevParams & P = A.P;
            const WaveBuffers & B = A.B;
            PRT_FLOW_BUFFERS(A, F);
            unsigned int * const fs = F.freelist + (size_t)block * slots;
            if (flow_load(&s_f[FLOW_QUIT]) != 0u) break;                       // a tracer's watchdog
            const unsigned int t0 = flow_load(&s_f[FLOW_D_TAIL0]), t1 = flow_load(&s_f[FLOW_D_TAIL1]), t2 = flow_load(&s_f[FLOW_D_TAIL2]);
            const unsigned int av0 = t0 - d_head[0], av1 = t1 - d_head[1], av2 = t2 - d_head[2];
            const unsigned int avail = av0 + av1 + av2;
            const unsigned int c_fill = c_tail - flow_load(&s_f[FLOW_C_HEAD]);          // slot ids the tracers have not claimed yet
            const bool hungry = c_fill < F.low_water;

            const unsigned long long s0 = COUNT ? __builtin_readcyclecounter() : 0ull;
            if (!fetch_done && free_count >= F.topup_min && (hungry || avail == 0u)) {
                // ---- top up: fresh samples into free slots --------------------------------------------------
                unsigned int want = free_count < F.topup_max ? free_count : F.topup_max;
                unsigned int base = 0;
                if (lane == 0) base = atomicAdd(A.Q.head, want);
                base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                if (base >= B.n_samples) {
                    fetch_done = true;
                } else {
                    unsigned int cnt = B.n_samples - base;
                    if (cnt <= want) fetch_done = true; else cnt = want;
                    const DevCamera cam = A.cam;
                    bool failed = false;
                    for (unsigned int k = lane; k < cnt; k += 64u) {
                        const unsigned int sid = base + k;
                        const unsigned int gsid = B.sample_base + sid;
                        const unsigned int slot = fs[free_count - 1u - k];
                        SampleState S;
                        Frame fr;
                        u64 * ring = RING && RINGMEM != 0 ? B.ring + (size_t)sid * B.ring_step : nullptr;
                        sample_begin<RING>(cam, P, pixel_of_local(P, gsid / P.spp), gsid % P.spp, S, fr, ring, B.ring_stride);
                        B.rng[sid] = make_ulonglong2(S.rng.chain, S.rng.prev);
                        if (RING && RINGMEM != 0) B.rng_aux[sid] = make_ulonglong2(S.rng.seed0, (u64)S.rng.k);
                        co[slot] = make_float4(fr.ray_o.x, fr.ray_o.y, fr.ray_o.z, as_f((int)sid));
                        cd[slot] = make_float4(fr.ray_d.x, fr.ray_d.y, fr.ray_d.z, as_f((int)WF_PENDING_FRESH_BIT << 8));
                        ct[slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
                        const unsigned int pos = (c_tail + k) & F.rc_mask;
                        unsigned int sp = 0;
                        while (flow_load(&crq[pos]) != (unsigned int)FLOW_INVALID) {
                            __builtin_amdgcn_s_sleep(2);
                            if (++sp > (unsigned int)FLOW_SPIN_LIMIT) { failed = true; break; }
                        }
                        crq[pos] = slot;
                    }
                    if (__ballot(failed) != 0ull) { if (lane == 0) { atomicOr(F.error, 8u); flow_store(&s_f[FLOW_QUIT], 1u); } break; }
                    free_count -= cnt;
                    c_tail += cnt;
                    rays += cnt;
                    flow_release();
                    if (lane == 0) flow_store(&s_f[FLOW_C_TAIL], c_tail);
                }
                spins = 0;
                if (COUNT) fl[2] += __builtin_readcyclecounter() - s0;
                continue;
            }

            if (avail >= 64u || (avail != 0u && (hungry || fetch_done))) {
                // ---- shade: up to 64 finished slots, from the three done rings in turn ------------------------
                const unsigned int n = avail < 64u ? avail : 64u;
                const unsigned int a0 = av0 < n ? av0 : n;
                const unsigned int a1 = av1 < n - a0 ? av1 : n - a0;
                const unsigned int a2 = n - a0 - a1;
                flow_acquire();
                const bool have = lane < n;
                unsigned int id = 0;
                if (have) {
                    const unsigned int rdn = F.rd_mask + 1u;
                    if (lane < a0) id = done_base[(d_head[0] + lane) & F.rd_mask];
                    else if (lane < a0 + a1) id = done_base[rdn + ((d_head[1] + lane - a0) & F.rd_mask)];
                    else id = done_base[2u * rdn + ((d_head[2] + lane - a0 - a1) & F.rd_mask)];
                }
                d_head[0] += a0; d_head[1] += a1; d_head[2] += a2;
                unsigned int s = 0;
                int level = 0;
                unsigned int pending = 0;
                f3 ray_o = mk3(0, 0, 0), ray_d = mk3(0, 0, 1), T = mk3(0, 0, 0);
                HitRec hit;
                hit.t = 0.0f; hit.v = hit.w = 0.0f; hit.tri = -1;
                if (have) {
                    const float4 ro = co[id], rd = cd[id], rt = ct[id], h = hits[id];
                    s = (unsigned int)as_i(ro.w);
                    level = as_i(rd.w) & 0xFF;
                    pending = ((unsigned int)as_i(rd.w)) >> 8;
                    ray_o = mk3(ro.x, ro.y, ro.z);
                    ray_d = mk3(rd.x, rd.y, rd.z);
                    T = mk3(rt.x, rt.y, rt.z);
                    hit.t = h.x; hit.v = h.y; hit.w = h.z; hit.tri = as_i(h.w);
                }
                const bool parked = have && hit.tri == POOL_PARKED_MARK;      // continues in the follow-up launches; its slot is free
                ShadeTables tb;
                tb.diffuse = sc.diffuse_dirs;
                tb.materials = sc.material_count <= (unsigned int)LDS_MATS ? s_mats : sc.materials;
                tb.lights = sc.light_count <= (unsigned int)LDS_LIGHTS ? s_lights : sc.lights;
                FlowEmit emit;
                emit.co = co; emit.cd = cd; emit.ct = ct;
                emit.so = sq_o; emit.sc = sq_c; emit.sd = sq_d;
                emit.crq = crq; emit.fs = fs; emit.s_f = s_f; emit.error = F.error;
                emit.rc_mask = F.rc_mask; emit.rs_mask = F.rs_mask;
                emit.c_tail = c_tail; emit.s_tail = s_tail; emit.free_count = free_count;
                emit.m_elided = 0; emit.n_pushed = 0;
                emit.my_slot = id; emit.release_slot = parked; emit.failed = false;
                unsigned int shaded = 0;
                shade_entry_lds<RING, TEX, BLOCK, RINGMEM>(sc, P, B, tb, have && !parked, s, level, pending, ray_o, ray_d, T, hit, emit, shaded, stack.frame_col());
                shaded_w += (unsigned int)__popcll(__ballot(shaded != 0));
                rays += emit.n_pushed + emit.m_elided;                          // counted as the reference counts them (raytracer.cpp:161)
                elided_w += emit.m_elided;
                free_count = emit.free_count;
                flow_release();
                if (lane == 0) {
                    if (emit.c_tail != c_tail) flow_store(&s_f[FLOW_C_TAIL], emit.c_tail);
                    if (emit.s_tail != s_tail) flow_store(&s_f[FLOW_S_TAIL], emit.s_tail);
                }
                c_tail = emit.c_tail;
                s_tail = emit.s_tail;
                spins = 0;
                if (COUNT) { fl[3] += __builtin_readcyclecounter() - s0; fl[6] += 1ull; fl[7] += n; }
                continue;
            }

            if (fetch_done && avail == 0u && free_count == slots) {
                // every sample has ended (or left for the follow-up launches); the tracers finish the shadow rays they hold
                if (flow_load(&s_f[FLOW_S_HEAD]) == s_tail) { if (lane == 0) flow_store(&s_f[FLOW_QUIT], 1u); break; }
            }
            __builtin_amdgcn_s_sleep(4);
            if (COUNT) fl[4] += __builtin_readcyclecounter() - s0;
            if (++spins > (unsigned int)FLOW_SPIN_LIMIT) { if (lane == 0) { atomicOr(F.error, 16u); flow_store(&s_f[FLOW_QUIT], 1u); } break; }
        }
        if (COUNT) fl[5] = __builtin_readcyclecounter() - fl_begin;
    }
#undef PRT_FLOW_BUFFERS

    // ---- counters: one atomic per workgroup and counter ----------------------------------------------------
    if (lane == 0) {
        if (rays) atomicAdd(&s_red[0], rays);
        if (shaded_w) atomicAdd(&s_red[1], (unsigned long long)shaded_w);
        if (elided_w) atomicAdd(&ctr->elided_shadow_rays, (unsigned long long)elided_w);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_red[0]) atomicAdd(&ctr->ray_count, s_red[0]);
        if (s_red[1]) atomicAdd(&ctr->shaded_hits, s_red[1]);
    }
    if (COUNT) {
        atomicAdd(&ctr->node_visits, (unsigned long long)st.nodes);
        atomicAdd(&ctr->tri_tests, (unsigned long long)st.tris);
        atomicAdd(&ctr->wave_node_steps, (unsigned long long)st.wnodes);
        atomicAdd(&ctr->wave_leaf_steps, (unsigned long long)st.wleaves);
        atomicAdd(&ctr->wave_tri_steps, (unsigned long long)st.wtris);
        atomicAdd(&ctr->wave_refills, (unsigned long long)st.wrefills);
        atomicMax(&ctr->max_sp, (unsigned long long)st.max_sp);
        atomicAdd(&ctr->culled, (unsigned long long)st.culled);
        atomicAdd(&ctr->wave_node_step_rays, (unsigned long long)st.wrays);
        if (lane == 0) for (int k = 0; k < 8; ++k) if (fl[k]) atomicAdd(&ctr->flow_cycles[k], fl[k]);
    }
}

}  // namespace prt
