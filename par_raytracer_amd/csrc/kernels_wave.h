// kernels_wave.h - the wavefront (ray-batched) pipeline: ray generation / persistent traversal / shading.
//
// Why: in the one-lane-per-sample kernel a wave lives as long as its unluckiest sample (1..14 rays, each
// 5..100 traversal steps) and rocprof shows ~10x more VALU wave-instructions than one lane's average
// work.  Here rays are independent work items in structure-of-arrays queues in HBM:
//
//   k_raygen   one lane per (pixel, sample): seed the RNG, jitter, camera ray -> closest-hit queue
//   k_trace    PERSISTENT waves pull rays (closest-hit and shadow rays mixed) from the queues through one
//              atomic head; a lane that finishes its ray goes idle, and when fewer than KEEP_MIN lanes of a
//              wave are still traversing, the wave leaves the loop, compacts its idle lanes with
//              ballot / mbcnt and refills them with the next rays.  Traversal state (node, stack pointer,
//              best hit) stays in registers + the per-lane LDS stack column across refills.
//   k_shade    one lane per closest-hit result: runs the sample's bounce-tree state machine until the next
//              ray, appends it (and the shadow rays of the hit) to the next round's queues with one
//              wave-aggregated atomic per queue.
//
// The reference consumes its RNG in depth-first order of the recursion (raytracer.cpp:413-577), so a
// sample has at most one closest-hit ray in flight; shadow rays use no RNG and ride along in the same
// round.  Radiance is accumulated in throughput form: a frame contributes T * (ambient*0.1 + direct*Kd*w_d
// + spec*Ks) and scales its children by T * Kd*w_d*weight / T * Ks*weight - the same polynomial as
// raytracer.cpp:543-552 with the products distributed (differences ~1e-7 relative, tolerance 1e-4;
// control flow - hits, roulette, directions - is bit exact and ray counts are equal).  Frames that still
// have children to spawn wait in a per-sample, per-level slot (64 B) and a bitmask of pending levels
// travels with the ray.
#pragma once

#include "../../include/prt_key.h"
#include "dev_shade.h"
#include "dev_texture.h"

namespace prt {

enum { WF_KIND_CLOSEST = 0, WF_KIND_SHADOW_ANY = 1, WF_KIND_SHADOW_DIST = 2 };
enum { WF_STAGE_REFL = 0, WF_STAGE_SPEC = 1, WF_STAGE_ALPHA = 2, WF_STAGE_DONE = 3 };
// In a closest-hit list entry's `pending` field (24 bits; the low 17 are the levels with a parked frame): this is the FIRST
// closest-hit ray of its sample, so the sample's radiance record holds nothing yet - the shading lane does not read it and
// writes it whatever it adds (no zero fill when the sample is fetched, no read at its first hit: 56 bytes per sample).
enum { WF_PENDING_FRESH_BIT = 0x400000 };

struct WaveBuffers {
    Accum * accum;               // [N] per-sample radiance (xyz), fixed point (dev_scene.h)
    ulonglong2 * rng;            // [N] (chain, prev)
    ulonglong2 * rng_aux;        // [N] (seed0, k)            RING only
    u64 * ring;                  // RING only: a sample's 16 slots at ring[sample * ring_step + slot * ring_stride].  Fixed spp:
                                 // [16][N] (step 1, stride N) - the lanes of a wave are consecutive samples making the same
                                 // draw, one run of 8-byte words.  Adaptive mode: [N][16] (step 16, stride 1) - the lanes are
                                 // pixels at different draws, and a pixel's slots share one 128-byte line
                                 // (profiles/r02_experiments.txt item 17)
    unsigned int ring_step, ring_stride;
    float4 * frames;             // [levels][FR4][N] pending frames
    float4 * rq_o[2];            // closest-hit queues, double buffered: (o.xyz, sample)
    float4 * rq_d[2];            //                                      (d.xyz, level | pending_mask << 8)
    float4 * rq_t[2];            //                                      (throughput.xyz, -)
    float4 * hits;               // [N] (t, v, w, tri) by queue position
    float4 * sq_o;               // shadow queue: (o.xyz, sample)
    float4 * sq_c;               //               (radiance if unoccluded .xyz, w): w >= 0: point light, w = distance^2 to
                                 //               it (raytracer.cpp:393-396); w < 0: directional light number -w - 1, whose
                                 //               direction comes from the light table - no per-ray copy of a constant
    float4 * sq_d;               //               (d.xyz, -)   written and read for point lights only
    unsigned int * counts;       // [0] next closest count, [1] next shadow count, [2] trace fetch head, [3] rays handed to k_trace_exact, [4] shadow rays of this round's hits that were counted, not traced
    unsigned int * overflow;     // ray indices whose hit has a near tie or whose stack overflowed (traced again, exactly, by k_trace_exact)
    unsigned int n_samples;      // samples of THIS chain (all per-sample arrays are indexed 0 .. n_samples)
    unsigned int sample_base;    // global id of its first sample (pixel / key derivation only)
};

PRT_D unsigned int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Wave-aggregated queue append: one atomic per wave, slots handed out by rank among the appending lanes.
PRT_D unsigned int wave_append(unsigned int * counter, bool pred) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return 0u;
    const unsigned int n = (unsigned int)__popcll(mask);
    const unsigned int prefix = __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
    const int leader = __ffsll((long long)mask) - 1;
    unsigned int base = 0;
    if ((int)lane_id() == leader) base = atomicAdd(counter, n);
    base = (unsigned int)__shfl((int)base, leader);
    return base + prefix;
}

// Workgroup-aggregated append: ONE atomic per workgroup and queue.  Same-address atomics retire at ~90 per
// microsecond chip-wide (MI355X_MICROARCH.md "dequeue"), so per-wave appends from 260k waves cost
// milliseconds; per-workgroup appends from 1024-thread groups cost ~0.2 ms.  Every thread of the block must
// call this (no early exits).  NW = waves per block.
template <int NW>
PRT_D unsigned int block_append(unsigned int * counter, bool pred, unsigned int * s_cnt /* [NW + 1] LDS */) {
    const unsigned long long mask = __ballot(pred);
    const unsigned int n = (unsigned int)__popcll(mask);
    const unsigned int prefix = __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
    const unsigned int wave = threadIdx.x >> 6;
    if (lane_id() == 0) s_cnt[wave] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int total = 0;
        for (int w = 0; w < NW; ++w) { const unsigned int c = s_cnt[w]; s_cnt[w] = total; total += c; }
        s_cnt[NW] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    const unsigned int slot = s_cnt[NW] + s_cnt[wave] + prefix;
    __syncthreads();
    return slot;
}

// ---------------------------------------------------------------------------------------------------------
template <bool RING>
__global__ __launch_bounds__(256) void k_raygen(DevCamera cam, DevParams P, WaveBuffers B) {
    const unsigned int sid = blockIdx.x * blockDim.x + threadIdx.x;
    if (sid >= B.n_samples) return;
    const unsigned int gsid = B.sample_base + sid;
    const unsigned int pixel = pixel_of_local(P, gsid / P.spp);
    const unsigned int samp = gsid % P.spp;
    SampleState S;
    Frame cur;
    u64 * ring = RING ? B.ring + (size_t)sid * B.ring_step : nullptr;
    sample_begin<RING>(cam, P, pixel, samp, S, cur, ring, B.ring_stride);
    B.rng[sid] = make_ulonglong2(S.rng.chain, S.rng.prev);
    if (RING) B.rng_aux[sid] = make_ulonglong2(S.rng.seed0, (u64)S.rng.k);
    B.rq_o[0][sid] = make_float4(cur.ray_o.x, cur.ray_o.y, cur.ray_o.z, as_f((int)sid));
    B.rq_d[0][sid] = make_float4(cur.ray_d.x, cur.ray_d.y, cur.ray_d.z, as_f((int)WF_PENDING_FRESH_BIT << 8));
    B.rq_t[0][sid] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
}

// ---------------------------------------------------------------------------------------------------------
// Persistent traversal.  grid = resident blocks; dynamic LDS = stack_entries * BLOCK * 4.
template <int BLOCK, bool COUNT>
__global__ __launch_bounds__(BLOCK, 6) void k_trace(DevScene sc, DevParams P, WaveBuffers B, int cur, unsigned int n_closest,
                                                  unsigned int n_shadow, int keep_min, int node_min, unsigned int chunk, int multi_light,
                                                  DevCounters * ctr) {
    extern __shared__ int s_stack[];
    LdsStack<BLOCK> stack;
    stack.attach(s_stack, threadIdx.x);
    stack.cap = P.stack_lds_entries;
    const unsigned int total = n_closest + n_shadow;
    const unsigned int lane = lane_id();
    const float4 * rq_o = B.rq_o[cur];
    const float4 * rq_d = B.rq_d[cur];
    unsigned int * head = B.counts + 2;

    TravRay r;
    trav_idle(r);
    int ray = -1;                        // index into the combined [closest | shadow] ray space, -1 = idle
    float4 payload = make_float4(0, 0, 0, 0);
    int sample = 0;
    bool exhausted = false;              // wave-uniform: the queue has no more rays to hand out
    TraceStats st;
    st.nodes = st.tris = st.wnodes = st.wleaves = st.wtris = st.wrefills = st.wrays = st.max_sp = st.culled = 0;

    unsigned int chunk_next = 0, chunk_end = 0;      // wave-uniform: rays reserved for this wave, not yet handed out
    for (;;) {
        // ---- refill idle lanes from the wave's reserved chunk; reserve a new chunk when it runs dry.
        // One atomic per CHUNK rays, not per refill: the head word is a single address (~90 atomics/us chip-wide).
        const unsigned long long idle = __ballot(ray < 0);
        if (idle != 0ull && !(exhausted && chunk_next == chunk_end)) {
            if (chunk_next == chunk_end) {
                unsigned int base = 0;
                if (lane == 0) base = atomicAdd(head, chunk);
                base = (unsigned int)__shfl((int)base, 0);
                if (base >= total) {
                    exhausted = true;
                } else {
                    chunk_next = base;
                    chunk_end = base + chunk < total ? base + chunk : total;
                }
            }
            const unsigned int avail = chunk_end - chunk_next;
            if (COUNT && avail && lane == 0) st.wrefills++;
            if (avail) {
                const unsigned int prefix = __builtin_amdgcn_mbcnt_hi((unsigned int)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)idle, 0u));
                const unsigned int n_idle = (unsigned int)__popcll(idle);
                const unsigned int take = n_idle < avail ? n_idle : avail;
                if (ray < 0 && prefix < take) {
                    const unsigned int idx = chunk_next + prefix;
                    float4 ro, rd;
                    int kind;
                    if (idx < n_closest) {
                        ro = rq_o[idx];
                        rd = rq_d[idx];
                        kind = WF_KIND_CLOSEST;
                    } else {
                        const unsigned int j = idx - n_closest;
                        ro = B.sq_o[j];
                        payload = B.sq_c[j];
                        if (payload.w < 0.0f) {          // directional light: the direction is a per-light constant
                            const DevLight & L = sc.lights[(unsigned int)(-payload.w) - 1u];
                            const f3 lv = mk3(L.facing[0], L.facing[1], L.facing[2]) * -1.0f;   // raytracer.cpp:240
                            rd = make_float4(lv.x, lv.y, lv.z, 0.0f);
                            kind = WF_KIND_SHADOW_ANY;
                        } else {
                            rd = B.sq_d[j];
                            kind = WF_KIND_SHADOW_DIST;
                        }
                    }
                    sample = as_i(ro.w);
                    const f3 d = mk3(rd.x, rd.y, rd.z);
                    const f3 ob = mk3(ro.x, ro.y, ro.z) + d * P.ray_bias;          // raytracer.cpp:163
                    trav_init(r, ob, d, kind == WF_KIND_SHADOW_ANY ? TRACE_ANY : TRACE_CLOSEST, P.box_pad, stack);
                    ray = (int)idx;
                }
                chunk_next += take;
            }
        }
        if (__ballot(ray >= 0) == 0ull) break;

        // ---- traverse until fewer than `leave_below` lanes of the wave are still busy
        const int leave_below = (exhausted && chunk_next == chunk_end) ? 1 : keep_min;
        while (ray >= 0) {
            // Node phase.  Lanes drop out as they reach a leaf; once fewer than `node_min` lanes are still
            // walking, the stragglers are suspended too (they keep their node) so the wave can run the leaf
            // phase for the majority instead of idling behind the longest walk.
            const int walkers = __popcll(__ballot(trav_walking(r)));
            const int nmin = node_min < (walkers >> 1) ? node_min : (walkers >> 1);
            while (trav_walking(r)) {
                trav_node_step<LdsStack<BLOCK>, COUNT>(sc, r, stack, st, P.box_pad);
                if (__popcll(__ballot(trav_walking(r))) < nmin) break;
            }
            bool fin = trav_done(r);
            if (!fin && !trav_walking(r)) fin = trav_leaf<LdsStack<BLOCK>, COUNT>(sc, r, stack, st);
            if (fin) {
                if (trav_needs_slow_path(r, stack)) {
                    // rare: the hit has company within a few ulp and the reference's visit order decides (dev_trace.h), or
                    // a push did not fit the LDS column (never observed on real scenes): hand the ray to k_trace_exact, which
                    // traces it again on a full-height stack and replays that order, before k_shade runs
                    B.overflow[atomicAdd(B.counts + 3, 1u)] = (unsigned int)ray;
                } else if ((unsigned int)ray < n_closest) {
                    B.hits[ray] = make_float4(r.best.t, r.best.v, r.best.w, as_f(r.best.tri));
                } else {
                    // shadow ray: add the precomputed radiance when unoccluded.  payload.w < 0: directional light
                    // (boolean only, raytracer.cpp:385); otherwise the point light's inverted distance test (:396)
                    const bool lit = r.best.tri < 0 || (payload.w >= 0.0f && r.best.t * r.best.t <= payload.w);
                    if (lit) {
                        accum_add(B.accum + sample, mk3(payload.x, payload.y, payload.z));
                    }
                }
                ray = -1;
                break;
            }
            if (__popcll(__ballot(true)) < leave_below) break;
        }
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
    }
}

// ---------------------------------------------------------------------------------------------------------
// Slow path of k_trace: rays whose hit has a near tie or whose LDS stack column overflowed are traced again by trace_ray,
// which replays the reference's visit order over near-tied candidates (dev_trace.h resolve_near_ties), on a per-lane global
// stack that holds the full worst-case bound.  A small fixed grid launched after every k_trace; it reads the list length on the device (no host round trip)
// and normally finds 0.
template <bool COUNT>
__global__ __launch_bounds__(256) void k_trace_exact(DevScene sc, DevParams P, WaveBuffers B, int cur, unsigned int n_closest,
                                                         int multi_light, DevCounters * ctr) {
    const unsigned int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int n_overflow = B.counts[3];
    TraceStats st;
    st.nodes = st.tris = st.wnodes = st.wleaves = st.wtris = st.wrefills = st.wrays = st.max_sp = st.culled = 0;
    GlobalStack slow;
    slow.attach(P.exact_stack, gid, P.exact_stack_stride);
    for (unsigned int i = gid; i < n_overflow; i += gridDim.x * blockDim.x) {
        const unsigned int idx = B.overflow[i];
        float4 ro, rd, payload = make_float4(0, 0, 0, 0);
        int kind;
        if (idx < n_closest) {
            ro = B.rq_o[cur][idx];
            rd = B.rq_d[cur][idx];
            kind = WF_KIND_CLOSEST;
        } else {
            const unsigned int j = idx - n_closest;
            ro = B.sq_o[j];
            payload = B.sq_c[j];
            if (payload.w < 0.0f) {
                const DevLight & L = sc.lights[(unsigned int)(-payload.w) - 1u];
                const f3 lv = mk3(L.facing[0], L.facing[1], L.facing[2]) * -1.0f;
                rd = make_float4(lv.x, lv.y, lv.z, 0.0f);
                kind = WF_KIND_SHADOW_ANY;
            } else {
                rd = B.sq_d[j];
                kind = WF_KIND_SHADOW_DIST;
            }
        }
        const int sample = as_i(ro.w);
        const f3 d = mk3(rd.x, rd.y, rd.z);
        const f3 ob = mk3(ro.x, ro.y, ro.z) + d * P.ray_bias;
        const HitRec best = trace_ray<GlobalStack, COUNT>(sc, ob, d, kind == WF_KIND_SHADOW_ANY ? TRACE_ANY : TRACE_CLOSEST, P.box_pad, slow, st);
        if (idx < n_closest) {
            B.hits[idx] = make_float4(best.t, best.v, best.w, as_f(best.tri));
        } else {
            const float dist_sq = kind == WF_KIND_SHADOW_ANY ? -1.0f : payload.w;
            const bool lit = best.tri < 0 || (dist_sq >= 0.0f && best.t * best.t <= dist_sq);
            if (lit) {
                accum_add(B.accum + sample, mk3(payload.x, payload.y, payload.z));
            }
        }
    }
    if (COUNT) {
        atomicAdd(&ctr->node_visits, (unsigned long long)st.nodes);
        atomicAdd(&ctr->tri_tests, (unsigned long long)st.tris);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Pending frame record: FR4 float4 per (level, sample).
//   f0 = (hit_p.xyz, mat)  f1 = (hit_n.xyz, stage | idx << 8)  f2 = (ray_d.xyz, alpha)  f3 = (T_in.xyz, w_diffuse)
//   f4 = (hit_pos.xyz, -)   only when translucent materials can occur (alpha continuation, raytracer.cpp:547-552)
//   f5 = (Kd.xyz, -)  f6 = (Ks.xyz, -)   only for textured scenes: the hit's own colours (raytracer.cpp:455-462);
//                                        untextured scenes re-read them from the material table instead
struct WFrame {
    f3 hit_p, hit_n, ray_d, T_in, hit_pos, kd, ks;
    float alpha, w_diffuse;
    int mat, stage, idx;
};

// The same frame held in LDS: field k of lane l at col[k * STRIDE] (col = the lane's own traversal stack column, which
// is idle while the wave shades).  Proxies make `f.hit_n`, `f.idx++`, `f.stage = x` read and write LDS in place, so
// the state machine of shade_entry keeps no frame in registers across its loop (97 -> 81 VGPRs).
template <int STRIDE>
struct LdsF {
    int * p;
    PRT_D operator float() const { return as_f(*p); }
    PRT_D LdsF & operator=(float v) { *p = as_i(v); return *this; }
    PRT_D LdsF & operator=(const LdsF & o) { *p = *o.p; return *this; }
};
template <int STRIDE>
struct LdsI {
    int * p;
    PRT_D operator int() const { return *p; }
    PRT_D LdsI & operator=(int v) { *p = v; return *this; }
    PRT_D LdsI & operator=(const LdsI & o) { *p = *o.p; return *this; }
    PRT_D int operator++(int) { const int v = *p; *p = v + 1; return v; }
};
template <int STRIDE>
struct LdsF3 {
    int * p;
    PRT_D operator f3() const { return mk3(as_f(p[0]), as_f(p[STRIDE]), as_f(p[2 * STRIDE])); }
    PRT_D LdsF3 & operator=(f3 v) { p[0] = as_i(v.x); p[STRIDE] = as_i(v.y); p[2 * STRIDE] = as_i(v.z); return *this; }
    PRT_D LdsF3 & operator=(const LdsF3 & o) { return *this = (f3)o; }
};
// hit_pos - where an alpha continuation ray starts (raytracer.cpp:547-552) - is only ever read when a material can be
// translucent; a render that cannot meet one (POS = false) has no such field: writes vanish, 23 dwords per lane instead of 26.
struct NoF3 {
    PRT_D NoF3 & operator=(f3) { return *this; }
    PRT_D operator f3() const { return mk3(0.0f, 0.0f, 0.0f); }
};
template <int STRIDE, bool POS> struct WFramePos { LdsF3<STRIDE> hit_pos; PRT_D void place(int * col) { hit_pos.p = col + 23 * STRIDE; } };
template <int STRIDE> struct WFramePos<STRIDE, false> { NoF3 hit_pos; PRT_D void place(int *) {} };
template <int STRIDE, bool POS>
struct WFrameLds : WFramePos<STRIDE, POS> {
    LdsF3<STRIDE> hit_p, hit_n, ray_d, T_in, kd, ks;
    LdsF<STRIDE> alpha, w_diffuse;
    LdsI<STRIDE> mat, stage, idx;
    PRT_D explicit WFrameLds(int * col) {
        hit_p.p = col; hit_n.p = col + 3 * STRIDE; ray_d.p = col + 6 * STRIDE; T_in.p = col + 9 * STRIDE;
        kd.p = col + 12 * STRIDE; ks.p = col + 15 * STRIDE;
        alpha.p = col + 18 * STRIDE; w_diffuse.p = col + 19 * STRIDE;
        mat.p = col + 20 * STRIDE; stage.p = col + 21 * STRIDE; idx.p = col + 22 * STRIDE;
        this->place(col);
    }
};
enum { WFRAME_LDS_DWORDS = 26, WFRAME_LDS_DWORDS_NOPOS = 23 };

// POS: the frame carries hit_pos (alpha continuation rays start there, raytracer.cpp:547-552): only renders that can meet a
// translucent material (RING && RINGMEM in the pool pipeline, RING in the wavefront pipeline) pay for that fifth float4.
template <bool POS, bool TEX, class FrameT>
PRT_D void wframe_save(const WaveBuffers & B, int level, unsigned int s, const FrameT & fr) {
    WFrame f;
    f.hit_p = fr.hit_p; f.hit_n = fr.hit_n; f.ray_d = fr.ray_d; f.T_in = fr.T_in; f.hit_pos = fr.hit_pos; f.kd = fr.kd; f.ks = fr.ks;
    f.alpha = fr.alpha; f.w_diffuse = fr.w_diffuse; f.mat = fr.mat; f.stage = fr.stage; f.idx = fr.idx;
    constexpr int FR4 = TEX ? 7 : POS ? 5 : 4;
    float4 * p = B.frames + ((size_t)level * FR4) * B.n_samples + s;
    p[0] = make_float4(f.hit_p.x, f.hit_p.y, f.hit_p.z, as_f(f.mat));
    p[(size_t)B.n_samples] = make_float4(f.hit_n.x, f.hit_n.y, f.hit_n.z, as_f(f.stage | (f.idx << 8)));
    p[(size_t)B.n_samples * 2] = make_float4(f.ray_d.x, f.ray_d.y, f.ray_d.z, f.alpha);
    p[(size_t)B.n_samples * 3] = make_float4(f.T_in.x, f.T_in.y, f.T_in.z, f.w_diffuse);
    if (POS) p[(size_t)B.n_samples * 4] = make_float4(f.hit_pos.x, f.hit_pos.y, f.hit_pos.z, 0.0f);
    if (TEX) {
        p[(size_t)B.n_samples * 5] = make_float4(f.kd.x, f.kd.y, f.kd.z, 0.0f);
        p[(size_t)B.n_samples * 6] = make_float4(f.ks.x, f.ks.y, f.ks.z, 0.0f);
    }
}

template <bool POS, bool TEX, class FrameT>
PRT_D void wframe_load(const WaveBuffers & B, int level, unsigned int s, FrameT & fr) {
    WFrame f;
    constexpr int FR4 = TEX ? 7 : POS ? 5 : 4;
    const float4 * p = B.frames + ((size_t)level * FR4) * B.n_samples + s;
    const float4 a = p[0], b = p[(size_t)B.n_samples], c = p[(size_t)B.n_samples * 2], d = p[(size_t)B.n_samples * 3];
    f.hit_p = mk3(a.x, a.y, a.z); f.mat = as_i(a.w);
    f.hit_n = mk3(b.x, b.y, b.z); f.stage = as_i(b.w) & 0xFF; f.idx = as_i(b.w) >> 8;
    f.ray_d = mk3(c.x, c.y, c.z); f.alpha = c.w;
    f.T_in = mk3(d.x, d.y, d.z); f.w_diffuse = d.w;
    if (POS) {
        const float4 e = p[(size_t)B.n_samples * 4];
        f.hit_pos = mk3(e.x, e.y, e.z);
    } else {
        f.hit_pos = f.hit_p;
    }
    if (TEX) {
        const float4 g = p[(size_t)B.n_samples * 5], h = p[(size_t)B.n_samples * 6];
        f.kd = mk3(g.x, g.y, g.z);
        f.ks = mk3(h.x, h.y, h.z);
    } else {
        f.kd = f.ks = mk3(0, 0, 0);
    }
    fr.hit_p = f.hit_p; fr.hit_n = f.hit_n; fr.ray_d = f.ray_d; fr.T_in = f.T_in; fr.hit_pos = f.hit_pos;
    if (TEX) { fr.kd = f.kd; fr.ks = f.ks; }
    fr.alpha = f.alpha; fr.w_diffuse = f.w_diffuse; fr.mat = f.mat; fr.stage = f.stage; fr.idx = f.idx;
}

// What k_shade reads besides the queues: small read-only tables (global, or staged in LDS by the caller).
struct ShadeTables {
    const float4 * diffuse;          // 1024 Hammersley cosine-lobe directions (tangent space)
    const DevMaterial * materials;
    const DevLight * lights;
};

// One closest-hit result -> the sample's next rays.  `live` lanes process (s, level, pending, ray, T, hit); every
// lane of the calling group must call it (the emitter aggregates appends).  emit.shadow(...) is called once per
// light by every lane, emit.closest(...) once at the end (its last argument tells the emitter that this lane's
// sample has no ray left - the adaptive mode of k_pool starts the pixel's next sample from there).
template <bool RING, bool TEX, int RINGMEM, class Emit, class FrameT>
PRT_D void shade_entry_on(const DevScene & sc, const DevParams & P, const WaveBuffers & B, const ShadeTables & tb, bool live,
                          unsigned int s, int level, unsigned int pending, f3 ray_o, f3 ray_d, f3 T, const HitRec & hit, Emit & emit,
                          unsigned int & shaded, FrameT & f) {
    const int depth = (int)P.bounce_depth;
    // RINGMEM: 0 = no draw ring in memory (and an opaque scene); 1 = draw ring, materials may be translucent; 2 = draw ring,
    // opaque scene (kernels_pool.h k_pool).  TRANS: the translucency paths exist.  POS: frames carry the hit position.
    constexpr bool RM = RINGMEM != 0, TRANS = RING && RINGMEM != 2, POS = RING && RINGMEM == 1;
    Rng rng;
    rng.chain = rng.prev = rng.seed0 = 0; rng.k = 0;
    // the sample's radiance record, requested with the rest of its state: this lane owns the sample for the phase (a sample has
    // one closest-hit ray in flight), so what the hit adds is a plain read-modify-write - no atomic -, and the read rides in the
    // memory round trip the pass makes anyway (dev_scene.h accum_add_owner; at the end of the pass it was a round trip of its own)
    long long acc_x = 0, acc_y = 0, acc_z = 0;
    const bool fresh = (pending & (unsigned int)WF_PENDING_FRESH_BIT) != 0u;     // the sample's first closest-hit ray
    pending &= ~(unsigned int)WF_PENDING_FRESH_BIT;
    if (live) {
        const ulonglong2 rs = B.rng[s];
        rng.chain = rs.x; rng.prev = rs.y;
        // (seed word 0, draws so far) matter from the 15th draw on: a render whose samples provably stop before that
        // (RINGMEM = false) neither reads nor writes them - 32 bytes less per shaded hit
        if (RING && RM) { const ulonglong2 ra = B.rng_aux[s]; rng.seed0 = ra.x; rng.k = (u32)ra.y; }
        // (Emit::ATOMIC_RADIANCE - the continuous-flow kernel, kernels_flow.h: a sample's shadow rays are no longer a phase apart
        // from its next hit, so the hit's radiance is added with atomics like theirs and nothing is read here)
        if (!fresh && !Emit::ATOMIC_RADIANCE) accum_load_owner(B.accum + s, acc_x, acc_y, acc_z);
    }
    // RINGMEM = false: the general-RNG code without its draw ring in memory, for renders whose samples provably make at
    // most 15 draws (dev_rng.h) - a compile-time NULL, so the ring code folds away
    u64 * ring = RING && RM ? B.ring + (size_t)s * B.ring_step : nullptr;
    const size_t ring_stride = B.ring_stride;

    f3 add = mk3(0.0f, 0.0f, 0.0f);          // radiance this invocation adds to the sample
    bool emit_closest = false;
    f3 next_o = ray_o, next_d = ray_d, next_T = T;
    int next_level = 0;

    // ---- step 1: the hit (or miss) of the ray that just came back ----------------------------------------
    enum { M_NEXT_CHILD, M_ENTER, M_RETURN_UP, M_DONE };
    int mode = M_DONE;
    f.hit_p = f.hit_n = f.ray_d = f.T_in = f.hit_pos = f.kd = f.ks = mk3(0, 0, 0);
    f.alpha = 1.0f; f.w_diffuse = 0.0f; f.mat = 0; f.stage = WF_STAGE_DONE; f.idx = 0;
    bool want_shadow = false;
    bool f_held = false;
    f3 T_own = T;
    DevMaterial mat = tb.materials[0];
    if (live) {
        if (hit.tri < 0) {                                                         // raytracer.cpp:573-575
            add = add + T * P.background;
            mode = M_RETURN_UP;
        } else {
            shaded = 1;
            const f3 ob = ray_o + ray_d * P.ray_bias;                               // raytracer.cpp:163
            const f3 pos = ob + ray_d * hit.t;                                      // raytracer.cpp:121
            const float4 * sp = sc.shade + 4 * (size_t)hit.tri;
            const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
            const f3 gn = normalize3(mk3(s2.y, s2.z, s2.w));                        // raytracer.cpp:122 (n = Cross(ab, ac))
            const int m = as_i(s3.w);
            mat = tb.materials[m];
            float alpha = mat.alpha;
            const float bwy = hit.v, bwz = hit.w;
            const float bwx = 1.0f - bwy - bwz;                                     // raytracer.cpp:120
            f3 ka = mk3(mat.ambient[0], mat.ambient[1], mat.ambient[2]);
            f.kd = mk3(mat.diffuse[0], mat.diffuse[1], mat.diffuse[2]);
            f.ks = mk3(mat.specular[0], mat.specular[1], mat.specular[2]);
            float tu = 0.0f, tv = 0.0f;
            bool alpha_tested = mat.alpha <= 1.0f;                                  // raytracer.cpp:443
            const bool any_tex = TEX && (mat.tex[0] & mat.tex[1] & mat.tex[2]) != 0xFFFFFFFFu;
            if (TEX && any_tex) {
                const float4 * up = sc.tri_uv + 2 * (size_t)hit.tri;
                const float4 u01 = up[0], u2 = up[1];
                tu = 0.0f + u01.x * bwx; tv = 0.0f + u01.y * bwx;                   // raytracer.cpp:439-442
                tu = tu + u01.z * bwy; tv = tv + u01.w * bwy;
                tu = tu + u2.x * bwz; tv = tv + u2.y * bwz;
                const unsigned int alpha_tex = mat.tex[1] >> 16;
                if (alpha_tex != DEV_TEX_NONE) {                                    // raytracer.cpp:444-446
                    alpha *= tex_sample_r(sc, alpha_tex, tu, tv);
                    alpha_tested = true;
                }
            }
            // (the non-RING variants only run scenes whose materials all have alpha >= 1: the translucency paths fold away)
            if (TRANS && alpha_tested && alpha <= 0.05f) {                           // raytracer.cpp:443-453
                next_o = pos + ray_d * P.ray_bias * 2.0f;
                next_d = ray_d;
                next_T = T;
                next_level = level;
                mode = M_ENTER;
            } else {
                if (TEX && any_tex) {                                               // raytracer.cpp:455-462
                    const unsigned int t_ka = mat.tex[0] & 0xFFFFu, t_kd = mat.tex[0] >> 16, t_ks = mat.tex[1] & 0xFFFFu;
                    if (t_ka != DEV_TEX_NONE) ka = ka * tex_sample_rgb(sc, t_ka, tu, tv);
                    if (t_kd != DEV_TEX_NONE) f.kd = f.kd * tex_sample_rgb(sc, t_kd, tu, tv);
                    if (t_ks != DEV_TEX_NONE) f.ks = tex_sample_rgb(sc, t_ks, tu, tv);      // replaces Ks
                }
                f3 interp = mk3(0.0f, 0.0f, 0.0f);                                  // raytracer.cpp:464-467
                interp = interp + mk3(s0.x, s0.y, s0.z) * bwx;
                interp = interp + mk3(s0.w, s1.x, s1.y) * bwy;
                interp = interp + mk3(s1.z, s1.w, s2.x) * bwz;
                f.hit_n = normalize3(interp);
                if (TEX && (mat.tex[2] & 0xFFFFu) != DEV_TEX_NONE) {                // raytracer.cpp:468-502
                    const float4 * tp = sc.tri_tan + 3 * (size_t)hit.tri;
                    const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                    f3 tangent = mk3(0.0f, 0.0f, 0.0f);
                    tangent = tangent + mk3(t0.x, t0.y, t0.z) * bwx;
                    tangent = tangent + mk3(t0.w, t1.x, t1.y) * bwy;
                    tangent = tangent + mk3(t1.z, t1.w, t2.x) * bwz;
                    tangent = normalize3(tangent);
                    const f3 bitangent = normalize3(cross3(f.hit_n, tangent));
                    const f3 smp = tex_sample_rgb(sc, mat.tex[2] & 0xFFFFu, tu, tv);
                    const f3 sn = smp * 2.0f - mk3(1.0f, 1.0f, 1.0f);
                    // world_from_tangent_space * sn, columns (tangent, bitangent, normal), mathlib.h:697-709; not renormalised
                    const f3 nn = f.hit_n;
                    f.hit_n = mk3((tangent.x * sn.x + bitangent.x * sn.y) + nn.x * sn.z,
                                  (tangent.y * sn.x + bitangent.y * sn.y) + nn.y * sn.z,
                                  (tangent.z * sn.x + bitangent.z * sn.y) + nn.z * sn.z);
                }
                f.hit_pos = pos;
                f.hit_p = pos + gn * P.ray_bias;                                    // raytracer.cpp:425
                f.ray_d = ray_d;
                f.T_in = T;
                f.alpha = alpha;
                f.mat = m;
                const float object_reflectivity = 0.04f;                            // raytracer.cpp:538-541
                const float fresnel = fresnel_amount(1.0f, mat.index_of_refraction, f.hit_n, ray_d);
                const float w_reflect = (object_reflectivity + (1.0f - object_reflectivity) * fresnel);
                f.w_diffuse = 1.0f - w_reflect;
                f.stage = WF_STAGE_REFL;
                f.idx = 0;
                T_own = TRANS && alpha < 1.0f ? T * alpha : T;                       // raytracer.cpp:551
                add = add + T_own * (ka * 0.1f);                                    // raytracer.cpp:543
                want_shadow = true;
                mode = M_NEXT_CHILD;
            }
        }
    }

    // everything this invocation adds to the sample is known now (the bounce walk below adds nothing): write the record back
    if (live && (fresh || add.x != 0.0f || add.y != 0.0f || add.z != 0.0f)) {
        if (Emit::ATOMIC_RADIANCE && !fresh) accum_add(B.accum + s, add);       // (a fresh record holds nothing and nobody else adds to it yet: a plain store)
        else accum_store_owner(B.accum + s, acc_x, acc_y, acc_z, add);
    }

    // ---- shadow rays of this hit (raytracer.cpp:507-511, 378-411): radiance-if-unoccluded rides with the ray
    for (unsigned int li = 0; li < sc.light_count; ++li) {
        f3 so = mk3(0, 0, 0), sd = mk3(0, 0, 1), contrib = mk3(0, 0, 0);
        float dist_sq = -1.0f;
        int kind = WF_KIND_SHADOW_ANY;
        if (want_shadow) {
            const DevLight L = tb.lights[li];
            f3 light_color = mk3(L.color[0], L.color[1], L.color[2]);
            f3 light_vector;
            if (L.type == 0) {
                light_vector = mk3(L.facing[0], L.facing[1], L.facing[2]) * -1.0f;
            } else {
                const f3 lp = mk3(L.position[0], L.position[1], L.position[2]);
                light_vector = normalize3(lp - f.hit_p);
                const f3 dv = lp - f.hit_p;
                dist_sq = dot3(dv, dv);
                const float falloff_denom = (sqrtf(dist_sq) / L.falloff) + 1.0f;
                light_color = light_color * (1.0f / (falloff_denom * falloff_denom));
                kind = WF_KIND_SHADOW_DIST;
            }
            const float spec_cos = dot3(f.ray_d * -1.0f, reflect3(light_vector, f.hit_n));
            const f3 dd = light_color * 2.0f * ref_max(0.0f, dot3(f.hit_n, light_vector));
            const f3 ds = light_color * powf(ref_max(0.0f, spec_cos), mat.specular_intensity);
            contrib = T_own * (dd * f.kd * f.w_diffuse + ds * f.ks);
            so = f.hit_p;
            sd = light_vector;
        }
        // A shadow ray whose radiance-if-unoccluded is exactly zero - the surface faces away from the light and the highlight
        // term is zero too - cannot change the image whatever it hits.  The reference casts it all the same (raytracer.cpp:385,
        // and counts it, :161); here it is COUNTED (ray_count stays the reference's) and not traced.
        const bool dead = want_shadow && P.elide_dead_shadow_rays && contrib.x == 0.0f && contrib.y == 0.0f && contrib.z == 0.0f;
        emit.elided(dead);
        emit.shadow(want_shadow && !dead, s, so, sd, contrib, kind == WF_KIND_SHADOW_DIST ? dist_sq : -(float)(li + 1u), kind);
    }

    // ---- step 2: walk the bounce tree in depth-first order until the next ray or the end of the sample ----
    while (mode != M_DONE) {
        if (mode == M_NEXT_CHILD) {                       // frame f at `level` spawns its next child, if any
            const int iters = depth - level;
            const DevMaterial fm = tb.materials[f.mat];
            const f3 own = TRANS && f.alpha < 1.0f ? f.T_in * f.alpha : f.T_in;
            // which child comes next (cheap), then ONE copy of the expensive direction code for both lobes
            int kind = -1;                                                          // 0 diffuse, 1 specular, 2 alpha continuation
            if (f.stage == WF_STAGE_REFL) {                                         // raytracer.cpp:516-526
                if (iters > 0 && (unsigned int)f.idx < P.reflection_samples) kind = 0;
                else { f.stage = WF_STAGE_SPEC; f.idx = 0; }
            }
            if (kind < 0 && f.stage == WF_STAGE_SPEC) {                             // raytracer.cpp:528-535
                if (iters > 0 && (unsigned int)f.idx < P.spec_samples) kind = 1;
                else f.stage = WF_STAGE_ALPHA;
            }
            if (kind < 0 && f.stage == WF_STAGE_ALPHA) {                            // raytracer.cpp:547-552
                f.stage = WF_STAGE_DONE;
                if (TRANS && f.alpha < 1.0f) kind = 2;
            }
            const bool spawned = kind >= 0;
            if (kind == 2) {
                next_o = f.hit_pos + f.ray_d * P.ray_bias * 2.0f;
                next_d = f.ray_d;
                next_T = f.T_in * (1.0f - f.alpha);
            } else if (kind >= 0) {
                float4 ts;
                if (kind == 0) {
                    const unsigned int series_i = (unsigned int)(rng_next<RING>(rng, ring, ring_stride) % 1024ull);
                    ts = tb.diffuse[series_i];
                } else {
                    ts = sc.spec_dirs[(size_t)f.mat * sc.spec_samples + (unsigned int)f.idx];
                }
                next_d = tangent_to_world(f.hit_n, mk3(ts.x, ts.y, ts.z));
                const float cw = kind == 0 ? ref_max(0.0f, dot3(f.hit_n, next_d)) : ref_max(0.0f, dot3(next_d, f.ray_d * -1.0f));
                // the frame's colours: carried in the frame for textured scenes, else re-read from the material table
                const f3 fkd = TEX ? f.kd : mk3(fm.diffuse[0], fm.diffuse[1], fm.diffuse[2]);
                const f3 fks = TEX ? f.ks : mk3(fm.specular[0], fm.specular[1], fm.specular[2]);
                const f3 lobe = kind == 0 ? fkd * f.w_diffuse : fks;
                next_T = own * (lobe * cw);
                next_o = f.hit_p;
                f.idx++;
                // look ahead: if this was the frame's last child, retire the frame now so it is neither parked
                // (64 B written, 64 B read back later) nor revisited
                const bool more_refl = kind == 0 && (unsigned int)f.idx < P.reflection_samples;
                const bool more_spec = kind == 0 ? P.spec_samples > 0u : (unsigned int)f.idx < P.spec_samples;
                if (!more_refl && !more_spec && !(TRANS && f.alpha < 1.0f)) f.stage = WF_STAGE_DONE;
            }
            if (!spawned) { f_held = false; mode = M_RETURN_UP; continue; }
            f_held = true;                                // f (at `level`) stays in registers until the child's fate is known
            next_level = level + 1;
            mode = M_ENTER;
        } else if (mode == M_ENTER) {                     // TraceRayColor entry (raytracer.cpp:415-420) for (next_*, next_level)
            const int iters = depth - next_level;
            bool dead = iters < 0;
            if (!dead && next_level != 0) dead = rng_float01<RING>(rng, ring, ring_stride) < 0.5f;
            if (!dead) {
                // the child flies: park its parent frame if that still has children to spawn afterwards
                if (f_held && f.stage != WF_STAGE_DONE) {
                    wframe_save<POS, TEX>(B, level, s, f);
                    pending |= 1u << level;
                }
                emit_closest = true;
                mode = M_DONE;
            } else if (f_held) {
                mode = M_NEXT_CHILD;                      // that invocation returned black; same frame, next child
            } else {
                mode = M_RETURN_UP;
            }
        } else {                                          // M_RETURN_UP: resume the deepest parked frame
            if (pending == 0u) { mode = M_DONE; break; }
            level = 31 - __clz((int)pending);
            pending &= ~(1u << level);
            wframe_load<POS, TEX>(B, level, s, f);
            mode = M_NEXT_CHILD;
        }
    }

    // ---- outputs ------------------------------------------------------------------------------------------
    // the RNG state goes back to memory for the sample's next shade event; a sample that has no ray left (every level returned)
    // will never draw again - except in adaptive mode, where the pixel's next sample continues the stream (Emit::KEEPS_RNG)
    if (live && (emit_closest || Emit::KEEPS_RNG)) {
        B.rng[s] = make_ulonglong2(rng.chain, rng.prev);
        if (RING && RM) B.rng_aux[s] = make_ulonglong2(rng.seed0, (u64)rng.k);
    }
    emit.closest(emit_closest, s, next_o, next_d, next_T, next_level, pending, live && !emit_closest);
}

// The frame in registers (k_shade) ...
template <bool RING, bool TEX, class Emit>
PRT_D void shade_entry(const DevScene & sc, const DevParams & P, const WaveBuffers & B, const ShadeTables & tb, bool live,
                       unsigned int s, int level, unsigned int pending, f3 ray_o, f3 ray_d, f3 T, const HitRec & hit, Emit & emit,
                       unsigned int & shaded) {
    WFrame f;
    shade_entry_on<RING, TEX, 1>(sc, P, B, tb, live, s, level, pending, ray_o, ray_d, T, hit, emit, shaded, f);
}

// ... or in the lane's LDS column `col` (stride STRIDE dwords between fields; WFRAME_LDS_DWORDS fields).
template <bool RING, bool TEX, int STRIDE, int RINGMEM, class Emit>
PRT_D void shade_entry_lds(const DevScene & sc, const DevParams & P, const WaveBuffers & B, const ShadeTables & tb, bool live,
                           unsigned int s, int level, unsigned int pending, f3 ray_o, f3 ray_d, f3 T, const HitRec & hit, Emit & emit,
                           unsigned int & shaded, int * col) {
    WFrameLds<STRIDE, RING && RINGMEM == 1> f(col);
    shade_entry_on<RING, TEX, RINGMEM>(sc, P, B, tb, live, s, level, pending, ray_o, ray_d, T, hit, emit, shaded, f);
}

// Emitter of k_shade: workgroup-aggregated appends to the global next-round queues.
template <int BLOCK>
struct QueueEmit {
    enum { KEEPS_RNG = 0, ATOMIC_RADIANCE = 0 };           // fixed spp: a sample that has ended never draws again; a hit's radiance by the owner's read-modify-write
    const WaveBuffers & B;
    int nxt;
    unsigned int * s_cnt;
    unsigned int * n_elided;          // this thread's count of shadow rays that were counted, not traced
    PRT_D void elided(bool dead) const { if (dead) ++*n_elided; }
    PRT_D void shadow(bool want, unsigned int s, f3 o, f3 d, f3 contrib, float w, int kind) const {
        const unsigned int slot = block_append<BLOCK / 64>(B.counts + 1, want, s_cnt);
        if (want) {
            B.sq_o[slot] = make_float4(o.x, o.y, o.z, as_f((int)s));
            if (kind == WF_KIND_SHADOW_DIST) B.sq_d[slot] = make_float4(d.x, d.y, d.z, 0.0f);
            B.sq_c[slot] = make_float4(contrib.x, contrib.y, contrib.z, w);      // w >= 0: point light distance^2; < 0: -(light + 1)
        }
    }
    PRT_D void closest(bool want, unsigned int s, f3 o, f3 d, f3 T, int level, unsigned int pending, bool /*sample_ended*/) const {
        const unsigned int slot = block_append<BLOCK / 64>(B.counts + 0, want, s_cnt);
        if (want) {
            B.rq_o[nxt][slot] = make_float4(o.x, o.y, o.z, as_f((int)s));
            B.rq_d[nxt][slot] = make_float4(d.x, d.y, d.z, as_f(level | (int)(pending << 8)));
            B.rq_t[nxt][slot] = make_float4(T.x, T.y, T.z, 0.0f);
        }
    }
};

// One lane per closest-hit result of queue `cur`; appends to queue `cur ^ 1` and to the shadow queue.
template <bool RING, int BLOCK, bool TEX>
__global__ __launch_bounds__(BLOCK) void k_shade(DevScene sc, DevParams P, WaveBuffers B, int cur, unsigned int n_closest,
                                                 DevCounters * ctr) {
    __shared__ unsigned int s_cnt[BLOCK / 64 + 1];
    // Small read-only tables staged in LDS once per workgroup: every dependent global load removed from the
    // per-hit chain (triangle -> material -> light -> direction table) is a memory round trip less per wave.
    constexpr int LDS_MATS = 32, LDS_LIGHTS = 4;
    __shared__ float4 s_diffuse[1024];                         // 16 KB: the Hammersley cosine-lobe directions
    __shared__ DevMaterial s_mats[LDS_MATS];                    // 2 KB
    __shared__ DevLight s_lights[LDS_LIGHTS];
    const bool lds_mats = sc.material_count <= (unsigned int)LDS_MATS;
    const bool lds_lights = sc.light_count <= (unsigned int)LDS_LIGHTS;
    for (unsigned int k = threadIdx.x; k < 1024u; k += BLOCK) s_diffuse[k] = sc.diffuse_dirs[k];
    if (lds_mats) {
        const float4 * src = reinterpret_cast<const float4 *>(sc.materials);
        float4 * dst = reinterpret_cast<float4 *>(s_mats);
        for (unsigned int k = threadIdx.x; k < sc.material_count * 4u; k += BLOCK) dst[k] = src[k];
    }
    if (lds_lights) {
        const float4 * src = reinterpret_cast<const float4 *>(sc.lights);
        float4 * dst = reinterpret_cast<float4 *>(s_lights);
        for (unsigned int k = threadIdx.x; k < sc.light_count * 3u; k += BLOCK) dst[k] = src[k];
    }
    __syncthreads();
    ShadeTables tb;
    tb.diffuse = s_diffuse;
    tb.materials = lds_mats ? s_mats : sc.materials;
    tb.lights = lds_lights ? s_lights : sc.lights;

    const unsigned int i = blockIdx.x * BLOCK + threadIdx.x;
    const bool live = i < n_closest;
    unsigned int s = 0;
    int level = 0;
    unsigned int pending = 0;
    f3 ray_o = mk3(0, 0, 0), ray_d = mk3(0, 0, 1), T = mk3(0, 0, 0);
    HitRec hit;
    hit.t = 0.0f; hit.v = hit.w = 0.0f; hit.tri = -1;
    if (live) {
        const float4 ro = B.rq_o[cur][i], rd = B.rq_d[cur][i], rt = B.rq_t[cur][i], h = B.hits[i];
        s = (unsigned int)as_i(ro.w);
        level = as_i(rd.w) & 0xFF;
        pending = ((unsigned int)as_i(rd.w)) >> 8;
        ray_o = mk3(ro.x, ro.y, ro.z);
        ray_d = mk3(rd.x, rd.y, rd.z);
        T = mk3(rt.x, rt.y, rt.z);
        hit.t = h.x; hit.v = h.y; hit.w = h.z; hit.tri = as_i(h.w);
    }
    unsigned int n_elided = 0;
    QueueEmit<BLOCK> emit = { B, cur ^ 1, s_cnt, &n_elided };
    unsigned int shaded = 0;
    shade_entry<RING, TEX>(sc, P, B, tb, live, s, level, pending, ray_o, ray_d, T, hit, emit, shaded);

    // shaded-hit count: one atomic per workgroup
    {
        const unsigned int n = (unsigned int)__popcll(__ballot(shaded != 0));
        if (lane_id() == 0) s_cnt[threadIdx.x >> 6] = n;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int total = 0;
            for (int w = 0; w < BLOCK / 64; ++w) total += s_cnt[w];
            if (total) atomicAdd(&ctr->shaded_hits, (unsigned long long)total);
        }
    }
    // shadow rays counted but not traced: one atomic per workgroup on the round's counter (the host adds it to the ray count)
    {
        unsigned int n = n_elided;
        for (int off = 32; off > 0; off >>= 1) n += (unsigned int)__shfl_down((int)n, off);
        __syncthreads();
        if (lane_id() == 0) s_cnt[threadIdx.x >> 6] = n;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int total = 0;
            for (int w = 0; w < BLOCK / 64; ++w) total += s_cnt[w];
            if (total) { atomicAdd(B.counts + 4, total); atomicAdd(&ctr->elided_shadow_rays, (unsigned long long)total); }
        }
    }
}

}  // namespace prt
