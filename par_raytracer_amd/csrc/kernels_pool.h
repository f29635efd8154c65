// kernels_pool.h - the wave-pool pipeline: ONE launch per frame, every wave runs the whole wavefront loop on a
// private pool of rays.
//
// Measured on the wavefront pipeline (profiles/r01_wavefront_*): each of its ~8 bounce rounds pays a launch ramp,
// a drain tail (the longest rays of the round with the rest of the chip idle), a host round trip for the queue
// sizes and chip-wide append atomics.  At full-frame size that is ~10 % of the frame; on a 1/8 frame shard
// (multi-GPU) it is half of it.  Here the ROUND is private to a wave:
//
//   every wave owns `cap` closest-hit ray slots and `cap * lights` shadow ray slots in HBM (structure of arrays,
//   so a wave's 64 lanes read / write 1 KB runs) and loops
//     top up   while samples remain: reserve `cap - n` of them from the global sample counter (one atomic per
//              top-up, the only inter-wave communication of the frame) and generate their camera rays in place
//     trace    all rays of the pool, closest-hit and shadow mixed, with the same traversal loop as k_trace: lanes
//              are refilled from the wave's own list by ballot / mbcnt rank, no atomics
//     shade    all closest hits of the pool, 64 per pass at full lane utilisation, with shade_entry() of
//              kernels_wave.h; the next rays are appended to the pool's other list by ballot prefix, no atomics
//   until the sample counter is exhausted and its pool is empty.  Per-sample state (RNG, accumulated radiance,
//   pending frames) lives in the same per-sample arrays as in the wavefront pipeline and is only ever touched by
//   the wave that fetched the sample, so plain loads and stores are enough (a workgroup-scope fence between the
//   phases orders them; the vector L1 is coherent within a CU).  Waves drift out of phase with each other, so the
//   HBM-bound shading of some waves overlaps the issue-bound traversal of others on the same SIMD.
//
// Results are those of the wavefront pipeline bit for bit: same shade_entry(), same traversal, same per-sample
// RNG keys; only the order in which independent samples are processed differs.
//
// Round 4, measured and NOT built in (profiles/r04_round_drains.txt): a round ends with a DRAIN - once its list is dry the wave
// walks on with the rays in flight, fewer and fewer, until the last is done: 13 % of a C4 frame's wave-level node steps at 30 %
// lane fill, 38 % of a 1/8 shard's, 49 % of the adaptive mode's.  Setting the last rays aside and resuming them in the next
// round removes half of that and gains nothing: the carried ray's sample is shaded a round later, and pool slots, not lanes,
// are what thin rounds are short of.  Handing the last rays to the other waves of the block (PRT_POOL_EXCHANGE_BUILD below)
// shrinks the drains by a third and makes everything slower (profiles/r04_ray_exchange.txt): a drain's thin wave steps cost
// next to nothing - the SIMD's other waves fill the issue slots - and the round ends with its longest ray, which no hand-over
// shortens.  A kernel without rounds (kernels_flow.h) loses for another reason: its one shading wave per block is the bottleneck.
//
// Round 4: BLOCK-SHARED pools (template parameter SHARED).  A round that is private to a wave is thin where there is little
// work per wave - a 1/8-frame shard of a multi-GPU run, the adaptive mode's one ray per pixel - and every thin round ends with
// a wave that issues its node steps for a handful of lanes (a 1/8 shard: 32 % of the node loop's lane slots hold no ray).
// With SHARED the UNIT that owns a pool is the workgroup: its four waves top up one list together, pull the rays of the
// trace phase from it by an LDS counter (ds_add_rtn per refill, slots by ballot rank as before), shade its hits 64 at a
// time and append to the next list through LDS counters; the phases are separated by workgroup barriers.  A round is then
// four times as wide at the same state in flight, its tail is shared by four waves - the waves that run dry first wait at
// the barrier, where they issue nothing, instead of walking half-empty - and everything else (lists in HBM, per-sample
// state, parking, the adopting EXACT launch, which keeps wave-private pools) is unchanged.
#pragma once

#include "kernels_wave.h"

namespace prt {

struct PoolBuffers {
    float4 * cq;                 // [units][2][3][cap]   closest-hit lists, double buffered: (o, sample) (d, level | pending << 8) (T, -)
    float4 * hits;               // [units][cap]         (t, v, w, tri) by list position
    float4 * sq;                 // [units][3][scap]     shadow list: (o, sample) (radiance, w) (d, -), see WaveBuffers::sq_*
    unsigned int * head;         // global sample counter (adopting launch: counter into the park list)
    unsigned int cap, scap;      // slots per unit (a wave; a workgroup when the pools are block-shared); multiples of 64
    unsigned int topup_min;      // top up when at least this many slots are free
    unsigned int topup_max;      // ... and take at most this many samples at once (experiment POOL_FAIR: one fair share per unit)
    // Rays the fast kernel (EXACT = false) cannot finish - the hit has company within a few ulp and the reference's visit order
    // decides (dev_trace.h resolve_near_ties), or a push did not fit the LDS stack column - are PARKED, list entry and all,
    // and dropped from the pool.  After the fast kernel: k_pool_parked_shadows traces the parked shadow rays exactly and adds
    // their radiance; then a second launch of k_pool, the EXACT variant (spill-capable stack, near ties decided in its loop),
    // adopts the parked closest-hit rays instead of fresh samples and runs their samples to the end.  The slow code thus never
    // sits in the fast kernel.  Normally nothing is parked and both follow-up launches find empty lists.
    float4 * park;               // [3][park_cap]  kind 0: (o, sample) (d, level | pending << 8) (T, kind) of a closest-hit ray
                                 //                kind 2 (ADAPT): (-, pixel) (-) (-, kind): a pixel whose finalise step was deferred
    float4 * spark;              // [3][spark_cap] (o, sample) (radiance, w as in sq_c) (d, -) of a parked shadow ray
    unsigned int * park_count;   // [0] closest / finalise entries parked so far, [1] shadow entries
    unsigned int park_cap, spark_cap;
    unsigned int adopt;          // EXACT kernels: 1 = top up from the park list, not from the sample counter
    // adaptive mode (k_pool<ADAPT>): the unit in the pool is a PIXEL that runs its samples one after the other
    unsigned int * fin;          // [units][cap]         pixels whose current sample has no ray left; finalised after the next trace phase
    float4 * scratch;            // [n_pixels][max_spp]  every sample's colour (RenderPixel's scratch_buffer, main.cpp:232)
    float4 * jobsum;             // [n_pixels]           (running colour sum .xyz, samples finished so far as int bits)
    float4 * final_rgb;          // [n_pixels]           the pixel's colour when it is done
    // diagnostics (COUNT kernels, option DEBUG_UTIL): per wave of the fast kernel, the 100 MHz wall clock at (start of its main
    // loop, the moment the sample counter ran dry for it, its exit); NULL otherwise
    unsigned long long * wave_times;
    // block-shared pools, experiment POOL_EXCHANGE: rays in flight that a wave hands to the other waves of its block when the
    // round's list is dry and it is down to xchg_max of them or fewer (see k_pool's trace phase)
    unsigned int guided, guided_min;     // guided self-scheduling of the top-ups (wave-private pools): see the top-up
    float * xchg;                // [units][4][POOL_XCHG_FIELDS][xchg_max]
    unsigned int xchg_max;       // 0: no exchange
};

// Emitter of k_pool: appends to the unit's lists, slots by rank among the appending lanes.  Per-wave pools (SHARED = false):
// the fill counts are wave-uniform registers.  Block-shared pools: they are LDS words that every wave of the block adds to -
// one ds_add_rtn per wave and call, the lanes' slots follow from the returned base and the ballot rank as before.
template <bool ADAPT, bool SHARED>
struct PoolEmit {
    enum { KEEPS_RNG = ADAPT ? 1 : 0,        // adaptive mode: the pixel's next sample continues the RNG stream of the one that ended
           ATOMIC_RADIANCE = 0 };            // a hit's radiance by the shading lane's own read-modify-write (the phases order it against the shadow rays' atomics)
    float4 * co, * cd, * ct;     // next closest list
    float4 * so, * sc, * sd;     // next shadow list
    unsigned int * fin;          // ADAPT: pixels to finalise after the next trace phase
    unsigned int m_c, m_s, m_f;  // wave-uniform fill counts (SHARED: unused, the counts are s_cnt[0..2])
    unsigned int * s_cnt;        // SHARED: LDS, [0] closest [1] shadow [2] finalise entries of the lists being filled
    unsigned int m_elided;       // wave-uniform: shadow rays counted, not traced (their radiance-if-unoccluded is zero)
    // first slot of this wave's `n` new entries of list `which` (0 closest, 1 shadow, 2 finalise); called wave-converged
    PRT_D unsigned int reserve(int which, unsigned int n) {
        if (SHARED) {
            if (n == 0u) return 0u;
            unsigned int base = 0;
            if (lane_id() == 0u) base = atomicAdd(&s_cnt[which], n);
            return (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
        }
        unsigned int & m = which == 0 ? m_c : which == 1 ? m_s : m_f;
        const unsigned int base = m;
        m += n;
        return base;
    }
    PRT_D void elided(bool dead) { m_elided += (unsigned int)__popcll(__ballot(dead)); }
    PRT_D void shadow(bool want, unsigned int s, f3 o, f3 d, f3 contrib, float w, int kind) {
        const unsigned long long mask = __ballot(want);
        const unsigned int prefix = __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
        const unsigned int base = reserve(1, (unsigned int)__popcll(mask));
        if (want) {
            const unsigned int slot = base + prefix;
            so[slot] = make_float4(o.x, o.y, o.z, as_f((int)s));
            if (kind == WF_KIND_SHADOW_DIST) sd[slot] = make_float4(d.x, d.y, d.z, 0.0f);
            sc[slot] = make_float4(contrib.x, contrib.y, contrib.z, w);
        }
    }
    // returns the list slot of this lane's ray (meaningful where `want`)
    PRT_D unsigned int closest(bool want, unsigned int s, f3 o, f3 d, f3 T, int level, unsigned int pending, bool sample_ended) {
        if (ADAPT) {
            const unsigned long long ended = __ballot(sample_ended);
            const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(ended >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)ended, 0u));
            const unsigned int fbase = reserve(2, (unsigned int)__popcll(ended));
            if (sample_ended) fin[fbase + rank] = s;
        }
        const unsigned long long mask = __ballot(want);
        const unsigned int prefix = __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
        const unsigned int slot = reserve(0, (unsigned int)__popcll(mask)) + prefix;
        if (want) {
            co[slot] = make_float4(o.x, o.y, o.z, as_f((int)s));
            cd[slot] = make_float4(d.x, d.y, d.z, as_f(level | (int)(pending << 8)));
            ct[slot] = make_float4(T.x, T.y, T.z, 0.0f);
        }
        return slot;
    }
};

enum { POOL_PARKED_MARK = 0x7FFFFFFE };        // hits[].w of a closest-hit ray that was parked: not shaded here (never a triangle index)
enum { POOL_PARK_CLOSEST = 0, POOL_PARK_FINALISE = 2 };
// Adaptive mode, per-pixel state word jobsum[].w: the index of the pixel's current sample in the low bits, and
enum { POOL_JOB_SAMPLE_MASK = 0x00FFFFFF,
       POOL_JOB_DEFERRED = (int)0x80000000,    // a shadow ray of the pixel is parked: its finalise step waits for the EXACT launch
       POOL_JOB_SPEC_FLYING = 0x40000000,      // the NEXT sample's camera ray is already in the pool (started speculatively)
       POOL_JOB_SPEC_DRAWN = 0x20000000 };     // ... its jitter is drawn (RNG advanced, offsets in final_rgb[].xy) but no ray is in flight
enum { POOL_CANCELLED_MARK = 0x7FFFFFFD };     // hits[].w of a speculative camera ray that must not be shaded (nor counted)
enum { POOL_SPEC_PENDING_BIT = 0x800000 };     // in a list entry's `pending` field: this camera ray was started speculatively
enum { POOL_PARKED_SHADOW_BLOCKS = 64 };         // k_pool_parked_shadows' fixed grid

PRT_D void pool_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
// Between the phases of a round: what one phase wrote to the unit's lists (global memory, through the compute unit's own L1)
// the next phase reads.  Wave-private pools: program order plus a fence.  Block-shared pools: a workgroup barrier as well -
// and every condition around a call of this must be block-uniform.
template <bool SHARED> PRT_D void pool_sync() {
    if (SHARED) __syncthreads(); else pool_fence();
}
// LDS control words of a block-shared pool
enum { PCTL_TRACE_NEXT = 0,      // rays of the trace phase handed out so far
       PCTL_TOPUP_BASE = 1,      // the sample counter's value this top-up got
       PCTL_CNT = 4,             // [2][4]: (closest, shadow, finalise, started-ahead) entries of the lists being filled, by parity of `cur`
       // the ray exchange of a trace phase (PoolBuffers::xchg), zeroed / set again behind the barrier that ends the phase:
       PCTL_X_TRACING = 12,      // waves of the block still tracing (a wave that hands its rays over, or has none left, leaves)
       PCTL_X_PENDING = 13,      // waves that have left and are still writing the rays they hand over
       PCTL_X_PUB = 14,          // [4]: rays wave w has handed over this round (published once, when all of them are written)
       PCTL_X_TAKEN = 18,        // [4]: of those, rays other waves have taken
       PCTL_WORDS = 22 };
// Measured and NOT built in by default (profiles/r04_ray_exchange.txt: images identical, every workload 1 - 7 % slower - a drain
// is bound by its longest ray, not by the vector issue its thin waves take): -DPRT_POOL_EXCHANGE_BUILD=1 compiles it, the
// option POOL_EXCHANGE = n turns it on.
#ifndef PRT_POOL_EXCHANGE_BUILD
#define PRT_POOL_EXCHANGE_BUILD 0
#endif
enum { POOL_XCHG_MAX = 32 };     // most rays a wave hands over
// a handed-over ray: its traversal registers, then (list index, sample, the thread whose LDS column holds its stack, payload)
enum { POOL_XCHG_FIELDS = TRAV_STATE_DWORDS + 7 };

// Everything the kernel is told, in device memory.  Passed by value these ~130 dwords are loaded into SGPRs at kernel
// entry and stay live through every loop; the kernel then spills ~170 SGPRs into VGPR lanes and reloads them with
// v_readlane inside the traversal loop (45 of its 250 VALU instructions, measured).  Instead each phase of the main loop
// reads what it needs through a pointer the compiler cannot see through (pool_args), as scalar loads from the
// constant address space, and the values die with the phase.
struct PoolArgs {
    DevScene sc;
    DevCamera cam;
    DevParams P;
    WaveBuffers B;
    PoolBuffers Q;
    int keep_min, node_min, multi_light, node_frac;      // node_frac / 8 of the walkers must still be walking, or the node loop ends
};
typedef const PoolArgs __attribute__((address_space(4))) * PoolArgsPtr;

// Pointers read from memory have no known address space: tell the compiler they are global, or every access through them
// becomes a FLAT instruction with a 64-bit VGPR address.
template <class T>
PRT_D T * as_global(T * p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_assume(!__builtin_amdgcn_is_shared((const void *)p));
    __builtin_assume(!__builtin_amdgcn_is_private((const void *)p));
#endif
    return p;
}

PRT_D PoolArgs pool_args(const PoolArgs * args) {
    asm volatile("" : "+s"(args));                 // a fresh value as far as the optimiser knows: no load is hoisted above this point
    PoolArgs A;
    __builtin_memcpy(&A, (PoolArgsPtr)args, sizeof(PoolArgs));      // constant address space: uniform scalar loads
    A.sc.nodes = as_global(A.sc.nodes); A.sc.tris = as_global(A.sc.tris); A.sc.shade = as_global(A.sc.shade);
    A.sc.tri_rank = as_global(A.sc.tri_rank); A.sc.materials = as_global(A.sc.materials); A.sc.lights = as_global(A.sc.lights);
    A.sc.diffuse_dirs = as_global(A.sc.diffuse_dirs); A.sc.spec_dirs = as_global(A.sc.spec_dirs);
    A.sc.textures = as_global(A.sc.textures); A.sc.texels = as_global(A.sc.texels); A.sc.srgb_lut = as_global(A.sc.srgb_lut);
    A.sc.tri_uv = as_global(A.sc.tri_uv); A.sc.tri_tan = as_global(A.sc.tri_tan);
    A.P.pixel_list = as_global(A.P.pixel_list); A.P.stack_spill = as_global(A.P.stack_spill);
    A.B.accum = as_global(A.B.accum); A.B.rng = as_global(A.B.rng); A.B.rng_aux = as_global(A.B.rng_aux); A.B.ring = as_global(A.B.ring);
    A.B.frames = as_global(A.B.frames);
    A.Q.cq = as_global(A.Q.cq); A.Q.hits = as_global(A.Q.hits); A.Q.sq = as_global(A.Q.sq); A.Q.head = as_global(A.Q.head);
    A.Q.park = as_global(A.Q.park); A.Q.spark = as_global(A.Q.spark); A.Q.park_count = as_global(A.Q.park_count);
    A.Q.fin = as_global(A.Q.fin); A.Q.scratch = as_global(A.Q.scratch); A.Q.jobsum = as_global(A.Q.jobsum); A.Q.final_rgb = as_global(A.Q.final_rgb);
    A.Q.wave_times = as_global(A.Q.wave_times);
    return A;
}

// The fast kernel's stack drops what does not fit its LDS column (and parks the ray); the EXACT kernel's continues in memory.
template <int BLOCK, bool EXACT> struct PoolStack { typedef LdsStack<BLOCK> type; };
template <int BLOCK> struct PoolStack<BLOCK, true> { typedef LdsSpillStack<BLOCK> type; };
template <int BLOCK> PRT_D void pool_stack_spill(LdsStack<BLOCK> &, PoolArgsPtr) {}
template <int BLOCK> PRT_D void pool_stack_spill(LdsSpillStack<BLOCK> & stack, PoolArgsPtr args) {
    stack.set_spill(as_global(args->P.stack_spill), args->P.stack_spill_stride);
}

// Puts the arguments where k_pool reads them.  A kernel rather than a hipMemcpyAsync: kernel arguments are captured at
// launch, whatever the runtime does with asynchronous copies from pageable memory.
// One launch prepares a whole render: dst[0] for the fast kernel, dst[1] - the same arguments with the park list as the
// source of work - for the adopting EXACT launch behind it, and the render's counters (sample counter, park counts, the
// adopting launch's counter) zeroed.
__global__ void k_pool_store_args(PoolArgs a, PoolArgs * dst, unsigned int * zero, unsigned int n_zero, unsigned int * adopt_head,
                                  unsigned int adopt_cap_div) {
    if (threadIdx.x < n_zero) zero[threadIdx.x] = 0u;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        dst[0] = a;
        if (adopt_head) {
            a.Q.adopt = 1;
            a.Q.head = adopt_head;
            a.Q.cap /= adopt_cap_div;             // the fast kernel's pools were block-shared: the adopting launch's are a wave's share of one
            a.Q.scap /= adopt_cap_div;
            dst[1] = a;
        }
    }
}

// grid = resident blocks; dynamic LDS = max(stack_entries, WFRAME_LDS_DWORDS) * BLOCK * 4 (traversal stack columns).
// EXACT: the slow variant - stack columns that continue in global memory, near ties decided inside the traversal loop
// (resolve_near_ties).  It runs as the adopting second launch of a render (PoolBuffers::park); the fast variant parks what
// it cannot finish.
// SHARED: the workgroup, not the wave, owns a pool (see the head of this file); never together with EXACT.
template <int BLOCK, int WAVES, bool LDSTAB, bool RING, bool COUNT, bool TEX, bool ADAPT, int RINGMEM, bool EXACT, bool SHARED>
__global__ __launch_bounds__(BLOCK, WAVES) void k_pool(const PoolArgs * args, DevCounters * ctr) {
    static_assert(!(SHARED && EXACT), "the adopting EXACT launch keeps wave-private pools");
    extern __shared__ int s_stack[];
    constexpr int LDS_MATS = 32, LDS_LIGHTS = 4;
    __shared__ float4 s_diffuse[LDSTAB ? 1024 : 1];
    __shared__ DevMaterial s_mats[LDS_MATS];
    __shared__ DevLight s_lights[LDS_LIGHTS];
    __shared__ unsigned long long s_red[2];
    __shared__ unsigned int s_ctl[SHARED ? PCTL_WORDS : 1];
    // experiment PRT_TOP_LDS (dev_trace4.h): the top of the tree in LDS, for the fast kernel's traversal loop only
    constexpr int TOPN = EXACT ? 0 : (int)TRAV_TOP_LDS_NODES;
    __shared__ uint4 s_top[TOPN > 0 ? 4 * TOPN : 1];
    {
        const PoolArgs A0 = pool_args(args);
        const DevScene & sc = A0.sc;
        if (LDSTAB) for (unsigned int k = threadIdx.x; k < 1024u; k += BLOCK) s_diffuse[k] = sc.diffuse_dirs[k];
        if (TOPN > 0) {
            const unsigned int n4 = 4u * (sc.node_count < (unsigned int)TOPN ? sc.node_count : (unsigned int)TOPN);
            for (unsigned int k = threadIdx.x; k < n4; k += BLOCK) s_top[k] = reinterpret_cast<const uint4 *>(sc.nodes)[k];
        }
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
    if (SHARED && threadIdx.x < (unsigned int)PCTL_WORDS) s_ctl[threadIdx.x] = threadIdx.x == (unsigned int)PCTL_X_TRACING ? (unsigned int)(BLOCK / 64) : 0u;
    __syncthreads();

    const unsigned int slot_id = blockIdx.x * BLOCK + threadIdx.x;
    const unsigned int lane = lane_id();
    // the unit that owns the lists: this wave, or this workgroup.  Scalar: the list pointers stay in SGPRs
    const unsigned int wave = SHARED ? blockIdx.x : (unsigned int)__builtin_amdgcn_readfirstlane((int)(slot_id >> 6));
    constexpr unsigned int ULANES = SHARED ? (unsigned int)BLOCK : 64u;                            // lanes of a unit
    const unsigned int ulane = SHARED ? threadIdx.x : lane;                                        // this lane's index in its unit
    const bool lead_wave = !SHARED || (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0u;   // adds the unit's counts
    typedef typename PoolStack<BLOCK, EXACT>::type Stack;
    Stack stack;
    stack.attach(s_stack, threadIdx.x);
    stack.cap = ((PoolArgsPtr)args)->P.stack_lds_entries;
    pool_stack_spill(stack, (PoolArgsPtr)args);

    // the unit's state: wave-uniform, and with SHARED block-uniform - every wave of the block derives the same values from the
    // same LDS words behind the same barriers, so that the barriers below are reached by all of them
    unsigned int n_f = 0;                      // (ADAPT) pixels waiting to be finalised
    unsigned int n_spec = 0;                   // (ADAPT) of those, pixels whose next camera ray is already in the closest list
    int cur = 0;
    unsigned int n_c = 0, n_s = 0;             // rays in the current closest / shadow list
    bool fetch_done = false;                   // the sample counter ran past n_samples
    unsigned long long rays = 0ull;            // wave-uniform
    unsigned int shaded_w = 0;                 // wave-uniform
    unsigned int elided_w = 0;                 // wave-uniform: shadow rays counted, not traced
    TraceStats st;
    st.nodes = st.tris = st.wnodes = st.wleaves = st.wtris = st.wrefills = st.wrays = st.max_sp = st.culled = 0;

// this wave's lists, from the arguments as re-read in the current phase
#define PRT_POOL_LISTS(A)                                                                              \
    const unsigned int cap = (A).Q.cap, scap = (A).Q.scap;                                            \
    float4 * const cq_base = (A).Q.cq + (size_t)wave * 6u * cap;                                      \
    float4 * const hits = (A).Q.hits + (size_t)wave * cap;                                            \
    float4 * const sq_o = (A).Q.sq + (size_t)wave * 3u * scap;                                        \
    float4 * const sq_c = sq_o + scap;                                                                \
    float4 * const sq_d = sq_c + scap;                                                                \
    float4 * const co = cq_base + (size_t)cur * 3u * cap;                                             \
    float4 * const cd = co + cap;                                                                     \
    float4 * const ct = cd + cap;                                                                     \
    (void)hits; (void)sq_o; (void)sq_c; (void)sq_d; (void)ct

    unsigned long long ph_topup = 0ull, ph_trace = 0ull, ph_shade = 0ull, ph_final = 0ull;      // COUNT: wave-cycles per phase
    unsigned long long dry_steps = 0ull, dry_rays = 0ull;      // COUNT: wave-level node steps taken after the round's list ran dry (the round's drain), lanes with a ray in them
    const unsigned long long ph_begin = COUNT ? __builtin_readcyclecounter() : 0ull;
    const unsigned long long wt_begin = COUNT ? wall_clock64() : 0ull;
    unsigned long long wt_dry = 0ull;
    for (;;) {
        const unsigned long long ph_t0 = COUNT ? __builtin_readcyclecounter() : 0ull;
        // ---- top up: fresh samples into the free closest-hit slots ------------------------------------------
        if (!fetch_done) {
          const PoolArgs A = pool_args(args);
          const DevParams & P = A.P;
          const WaveBuffers & B = A.B;
          const PoolBuffers & Q = A.Q;
          PRT_POOL_LISTS(A);
          if (EXACT && Q.adopt) {
            // second launch: the work is the park list.  Only into an EMPTY pool, so that an adopted ray never meets rays of
            // its own sample's earlier life.
            if (n_c + n_s + n_f == 0u) {
                const unsigned int n_parked = *Q.park_count < Q.park_cap ? *Q.park_count : Q.park_cap;
                // spread thinly: every parked sample still has its remaining bounces to go, one round after the other, and this
                // launch is the tail of the frame - a few rays per wave on many waves, not all of them on the first wave
                const unsigned int n_waves = gridDim.x * (BLOCK / 64);
                unsigned int chunk = (n_parked + n_waves - 1u) / n_waves;
                chunk = chunk < 1u ? 1u : chunk > cap ? cap : chunk;
                unsigned int base = 0;
                if (lane == 0) base = atomicAdd(Q.head, chunk);
                base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                if (base >= n_parked) {
                    fetch_done = true;
                } else {
                    unsigned int cnt = n_parked - base;
                    if (cnt <= chunk) fetch_done = true; else cnt = chunk;
                    // closest-hit rays go to the list; a deferred finalise step goes to the finalise list (its pixel's parked
                    // shadow rays have been traced by k_pool_parked_shadows in the meantime)
                    unsigned int * const fin_list = ADAPT ? Q.fin + (size_t)wave * cap : nullptr;
                    unsigned int m_c = 0, m_f = 0;
                    for (unsigned int k0 = 0; k0 < cnt; k0 += 64u) {
                        const unsigned int k = k0 + lane;
                        const bool have = k < cnt;
                        float4 e0 = make_float4(0, 0, 0, 0), e1 = e0, e2 = e0;
                        if (have) { e0 = Q.park[base + k]; e1 = Q.park[(size_t)Q.park_cap + base + k]; e2 = Q.park[2u * (size_t)Q.park_cap + base + k]; }
                        const bool is_fin = ADAPT && have && as_i(e2.w) == POOL_PARK_FINALISE;
                        const bool is_ray = have && !is_fin;
                        const unsigned long long mr = __ballot(is_ray), mf = __ballot(is_fin);
                        const unsigned int pr = __builtin_amdgcn_mbcnt_hi((unsigned int)(mr >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mr, 0u));
                        const unsigned int pf = __builtin_amdgcn_mbcnt_hi((unsigned int)(mf >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mf, 0u));
                        if (is_ray) { co[m_c + pr] = e0; cd[m_c + pr] = e1; ct[m_c + pr] = make_float4(e2.x, e2.y, e2.z, 0.0f); }
                        if (is_fin) {
                            const unsigned int j = (unsigned int)as_i(e0.w);
                            fin_list[m_f + pf] = j;
                            float4 js = Q.jobsum[j];
                            js.w = as_f(as_i(js.w) & 0x7FFFFFFF);                  // the deferral mark
                            Q.jobsum[j] = js;
                        }
                        m_c += (unsigned int)__popcll(mr);
                        m_f += (unsigned int)__popcll(mf);
                    }
                    cnt = m_c;
                    n_f = m_f;
                    n_c = cnt;
                    rays -= cnt;                  // these rays were counted when the first launch traced them (raytracer.cpp:161: one TraceRay each)
                }
            }
          } else if (n_c + n_f - n_spec + (!SHARED && Q.guided ? (Q.guided_min < Q.topup_min ? Q.guided_min : Q.topup_min) : Q.topup_min) <= cap) {
            const unsigned int room = cap - (n_c + n_f - n_spec);
            unsigned int want = room < Q.topup_max ? room : Q.topup_max;
            if (!SHARED && Q.guided) {
                // guided self-scheduling (option POOL_GUIDED; adaptive mode's default): towards the end of the call no wave takes
                // more than its share of what is left (x guided / 8), so that the last units - in adaptive mode pixels with up
                // to max_spp samples to run one after the other - are spread over all waves instead of filling the pools of the
                // waves that asked last (C4 adaptive 125.7 -> 123.2 ms, fixed-spp frames and shards unchanged:
                // profiles/r04_ray_exchange.txt).  The counter is read without claiming anything; the value may be stale, the
                // claim below is not.
                const unsigned int taken = __hip_atomic_load(Q.head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int left = taken < B.n_samples ? B.n_samples - taken : 0u;
                const unsigned int n_units = gridDim.x * (unsigned int)(BLOCK / 64);
                unsigned int lim = (unsigned int)(((unsigned long long)left * Q.guided) / (8ull * n_units));
                if (lim < Q.guided_min) lim = Q.guided_min;
                // not worth a top-up yet?  Judged by the ROOM, not by what another clamp left of it, and never true of an empty
                // pool (room = cap >= topup_min): a wave that could wait here for ever would never see the counter run dry
                if (room < (lim < Q.topup_min ? lim : Q.topup_min)) want = 0u;
                else if (want > lim) want = lim;
            }
            if (want != 0u) {
            unsigned int base = 0;
            if (SHARED) {
                if (threadIdx.x == 0) s_ctl[PCTL_TOPUP_BASE] = atomicAdd(Q.head, want);
                __syncthreads();
                base = s_ctl[PCTL_TOPUP_BASE];
                __syncthreads();                                   // ... and nobody overwrites the word before every wave has read it
            } else if (lane == 0) base = atomicAdd(Q.head, want);
            base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
            if (base >= B.n_samples) {
                fetch_done = true;
            } else {
                unsigned int cnt = B.n_samples - base;
                if (cnt <= want) fetch_done = true; else cnt = want;
                const DevCamera cam = A.cam;
                for (unsigned int k = ulane; k < cnt; k += ULANES) {
                    const unsigned int sid = base + k;
                    const unsigned int gsid = B.sample_base + sid;
                    SampleState S;
                    Frame fr;
                    u64 * ring = RING && RINGMEM != 0 ? B.ring + (size_t)sid * B.ring_step : nullptr;
                    if (ADAPT) {
                        // the unit is the pixel: one RNG stream, key of sample 0 (include/prt.h prt_params::max_spp)
                        sample_begin<RING>(cam, P, pixel_of_local(P, gsid), 0u, S, fr, ring, B.ring_stride);
                        Q.jobsum[sid] = make_float4(0.0f, 0.0f, 0.0f, as_f(0));
                    } else {
                        sample_begin<RING>(cam, P, pixel_of_local(P, gsid / P.spp), gsid % P.spp, S, fr, ring, B.ring_stride);
                    }
                    B.rng[sid] = make_ulonglong2(S.rng.chain, S.rng.prev);
                    if (RING && RINGMEM != 0) B.rng_aux[sid] = make_ulonglong2(S.rng.seed0, (u64)S.rng.k);     // (seed word 0, draw count): only a sample that can pass 15 draws needs them
                    // the radiance record is first written by the sample's first shade event (WF_PENDING_FRESH_BIT) - in adaptive
                    // mode too: the finalise step has read the previous sample's sum before that event (same wave, earlier in
                    // the phase), so nothing zeroes the record between the samples of a pixel
                    co[n_c + k] = make_float4(fr.ray_o.x, fr.ray_o.y, fr.ray_o.z, as_f((int)sid));
                    cd[n_c + k] = make_float4(fr.ray_d.x, fr.ray_d.y, fr.ray_d.z, as_f((int)WF_PENDING_FRESH_BIT << 8));
                    ct[n_c + k] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
                }
                n_c += cnt;
            }
            }
          }
        }
        if (COUNT && fetch_done && wt_dry == 0ull) wt_dry = wall_clock64();
        const unsigned int total = n_c + n_s;
        if (total == 0u && n_f == 0u) {
            if (fetch_done) break;
            continue;                      // (never twice in a row: an empty pool has room for any top-up, and a top-up that gets nothing sets fetch_done)
        }
        pool_sync<SHARED>();
        if (lead_wave) rays += total;                                      // debug->ray_count++  raytracer.cpp:161
        const unsigned long long ph_t1 = COUNT ? __builtin_readcyclecounter() : 0ull;

        // ---- trace: every ray of the pool ------------------------------------------------------------------
        {
            const PoolArgs A = pool_args(args);
            const DevScene & sc = A.sc;
            const DevParams & P = A.P;
            const WaveBuffers & B = A.B;
            const int keep_min = A.keep_min, node_min = A.node_min, node_frac = A.node_frac;
            PRT_POOL_LISTS(A);
            const DevLight * lights = sc.light_count <= (unsigned int)LDS_LIGHTS ? s_lights : sc.lights;
            TravRay r;
            trav_idle(r);
            int ray = -1;
            float4 payload = make_float4(0, 0, 0, 0);
            int sample = 0;
            unsigned int next = 0;                                         // wave-uniform: rays handed out so far (SHARED: where this wave's last refill began)
            bool dry = total == 0u;                                        // wave-uniform: the list has no ray left to hand out
            // the exchange (block-shared pools): see the end of this loop's head
            const unsigned int xmax = SHARED && PRT_POOL_EXCHANGE_BUILD ? A.Q.xchg_max : 0u;      // (0 at compile time: the code below is not built)
            const unsigned int xwave = (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            float * const xbase = SHARED ? A.Q.xchg + (size_t)wave * (4u * (unsigned int)POOL_XCHG_FIELDS) * xmax : nullptr;
            unsigned int x_spins = 0;
            bool x_stays = false;                                           // wave-uniform: this wave found itself the last one tracing
            for (;;) {
                const unsigned long long idle = __ballot(ray < 0);
                if (idle != 0ull && !dry) {
                    if (COUNT && lane == 0) st.wrefills++;
                    const unsigned int prefix = __builtin_amdgcn_mbcnt_hi((unsigned int)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)idle, 0u));
                    const unsigned int n_idle = (unsigned int)__popcll(idle);
                    if (SHARED) {
                        // the block's list: this wave's idle lanes claim the next n_idle entries (the counter may run past the end)
                        unsigned int b = 0;
                        if (lane == 0) b = atomicAdd(&s_ctl[PCTL_TRACE_NEXT], n_idle);
                        next = (unsigned int)__builtin_amdgcn_readfirstlane((int)b);
                        if (next > total) next = total;
                    }
                    const unsigned int avail = total - next;
                    const unsigned int take = n_idle < avail ? n_idle : avail;
                    if (ray < 0 && prefix < take) {
                        const unsigned int idx = next + prefix;
                        float4 ro, rd;
                        int kind;
                        if (idx < n_c) {
                            ro = co[idx];
                            rd = cd[idx];
                            kind = WF_KIND_CLOSEST;
                        } else {
                            const unsigned int j = idx - n_c;
                            ro = sq_o[j];
                            payload = sq_c[j];
                            if (payload.w < 0.0f) {          // directional light: the direction is a per-light constant
                                const DevLight & L = lights[(unsigned int)(-payload.w) - 1u];
                                const f3 lv = mk3(L.facing[0], L.facing[1], L.facing[2]) * -1.0f;   // raytracer.cpp:240
                                rd = make_float4(lv.x, lv.y, lv.z, 0.0f);
                                kind = WF_KIND_SHADOW_ANY;
                            } else {
                                rd = sq_d[j];
                                kind = WF_KIND_SHADOW_DIST;
                            }
                        }
                        sample = as_i(ro.w);
                        const f3 d = mk3(rd.x, rd.y, rd.z);
                        const f3 ob = mk3(ro.x, ro.y, ro.z) + d * P.ray_bias;          // raytracer.cpp:163
                        trav_init(r, ob, d, kind == WF_KIND_SHADOW_ANY ? TRACE_ANY : TRACE_CLOSEST, P.box_pad, stack);
                        ray = (int)idx;
                    }
                    next += take;
                    dry = next == total;
                }
                int leave_below = dry ? 1 : keep_min;
                if (SHARED && xmax != 0u && dry) {
                    // ---- the exchange: a round ends with a drain - the list is dry and every wave walks its last rays down at a
                    // falling lane fill (profiles/r04_round_drains.txt).  A wave that is down to xmax rays hands them to the
                    // waves of its block that are still tracing and goes to the barrier; those take them into their idle lanes.
                    // The registers travel through memory, the stack entries are copied from the giver's LDS column (it waits at
                    // the barrier and does not touch it).  The last wave tracing cannot leave.
                    // take: idle lanes <- rays handed over by waves that left
                    if (__ballot(ray < 0) != 0ull) {
                        for (unsigned int w2 = 0; w2 < (unsigned int)(BLOCK / 64); ++w2) {
                            const unsigned int pub = __hip_atomic_load(&s_ctl[PCTL_X_PUB + w2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (pub == 0u) continue;
                            const unsigned long long idle2 = __ballot(ray < 0);
                            if (idle2 == 0ull) break;
                            const unsigned int n_idle2 = (unsigned int)__popcll(idle2);
                            unsigned int b = 0;
                            if (lane == 0) b = atomicAdd(&s_ctl[PCTL_X_TAKEN + w2], n_idle2);       // (may run past `pub`: nobody adds rays to a published set)
                            b = (unsigned int)__builtin_amdgcn_readfirstlane((int)b);
                            if (b >= pub) continue;
                            pool_fence();
                            const unsigned int k = pub - b < n_idle2 ? pub - b : n_idle2;
                            const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(idle2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)idle2, 0u));
                            if (ray < 0 && rank < k) {
                                const float * rec = xbase + (size_t)w2 * (unsigned int)POOL_XCHG_FIELDS * xmax + (b + rank);
                                trav_restore_regs(r, rec, xmax);
                                rec += (size_t)TRAV_STATE_DWORDS * xmax;
                                ray = as_i(rec[0]);
                                sample = as_i(rec[xmax]);
                                Stack from = stack;
                                from.attach(s_stack, (unsigned int)as_i(rec[2u * xmax]));
                                payload = make_float4(rec[3u * xmax], rec[4u * xmax], rec[5u * xmax], rec[6u * xmax]);
                                trav_copy_stack(stack, from, r.sp);
                            }
                        }
                    }
                    const unsigned int busy = (unsigned int)__popcll(__ballot(ray >= 0));
                    if (!x_stays && busy <= xmax) {
                        // leave: announce (pending first, so that the last wave waits for the records), step out, hand over
                        unsigned int was = 0;
                        if (lane == 0) {
                            if (busy != 0u) atomicAdd(&s_ctl[PCTL_X_PENDING], 1u);
                            was = atomicSub(&s_ctl[PCTL_X_TRACING], 1u);
                        }
                        was = (unsigned int)__builtin_amdgcn_readfirstlane((int)was);
                        if (was != 1u) {
                            if (busy != 0u) {
                                const unsigned long long act = __ballot(ray >= 0);
                                const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)act, 0u));
                                if (ray >= 0) {
                                    float * rec = xbase + (size_t)xwave * (unsigned int)POOL_XCHG_FIELDS * xmax + rank;
                                    trav_save_regs(r, rec, xmax);
                                    rec += (size_t)TRAV_STATE_DWORDS * xmax;
                                    rec[0] = as_f(ray);
                                    rec[xmax] = as_f(sample);
                                    rec[2u * xmax] = as_f((int)threadIdx.x);
                                    rec[3u * xmax] = payload.x; rec[4u * xmax] = payload.y; rec[5u * xmax] = payload.z; rec[6u * xmax] = payload.w;
                                }
                                pool_fence();
                                if (lane == 0) {
                                    __hip_atomic_store(&s_ctl[PCTL_X_PUB + xwave], busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    pool_fence();
                                    atomicSub(&s_ctl[PCTL_X_PENDING], 1u);
                                }
                            }
                            break;
                        }
                        // the last wave tracing stays - until every wave that left has written its records and all are taken
                        if (lane == 0) {
                            atomicAdd(&s_ctl[PCTL_X_TRACING], 1u);
                            if (busy != 0u) atomicSub(&s_ctl[PCTL_X_PENDING], 1u);
                        }
                        x_stays = true;
                    }
                    if (x_stays) {
                        bool more = false;
                        if (__hip_atomic_load(&s_ctl[PCTL_X_PENDING], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) more = true;
                        for (unsigned int w2 = 0; w2 < (unsigned int)(BLOCK / 64); ++w2)
                            if (__hip_atomic_load(&s_ctl[PCTL_X_TAKEN + w2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <
                                __hip_atomic_load(&s_ctl[PCTL_X_PUB + w2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) more = true;
                        if (busy == 0u) {
                            if (!more) break;
                            if (++x_spins > (1u << 22)) { if (lane == 0) ctr->flow_error = 0x7801u; break; }      // (never seen; a hang would take the GPU with it)
                            __builtin_amdgcn_s_sleep(2);
                            continue;
                        }
                        // with rays of its own: come back for what is handed over every time a few lanes have fallen idle
                        leave_below = more ? (int)busy - 3 : 1;
                        if (leave_below < 1) leave_below = 1;
                    } else {
                        // (busy > xmax) come back to hand over, and on the way for what others have handed over
                        leave_below = (int)busy - 7 > (int)xmax + 1 ? (int)busy - 7 : (int)xmax + 1;
                    }
                } else if (__ballot(ray >= 0) == 0ull) break;

                while (ray >= 0) {
                    const int walkers = __popcll(__ballot(trav_walking(r)));
                    const int wfrac = (walkers * node_frac) >> 3;
                    const int nmin = node_min < wfrac ? node_min : wfrac;
                    const unsigned int with_ray = COUNT ? (unsigned int)__popcll(__ballot(true)) : 0u;
                    while (trav_walking(r)) {
                        trav_node_step<Stack, COUNT>(sc, r, stack, st, P.box_pad, TOPN > 0 ? s_top : nullptr);
                        if (COUNT && first_active_lane()) { st.wrays += with_ray; if (dry) { dry_steps += 1ull; dry_rays += with_ray; } }
                        if (__popcll(__ballot(trav_walking(r))) < nmin) break;
                    }
                    bool fin = trav_done(r);
                    if (!fin && !trav_walking(r)) fin = trav_leaf<Stack, COUNT>(sc, r, stack, st);
                    if (fin) {
                        if (EXACT && trav_wants_resolve(r, stack)) r.best = resolve_near_ties<Stack, COUNT>(sc, r.o, r.d, P.box_pad, r.best.t, stack, st);
                        if (!EXACT && trav_needs_slow_path(r, stack)) {
                            // rare: the hit has company within a few ulp and the reference's visit order decides, or a push did
                            // not fit the LDS column (dev_trace.h).  Not here: the ray is parked for the launches that follow.
                            if (ADAPT && (unsigned int)ray < n_c && ((unsigned int)as_i(cd[ray].w) >> 8 & POOL_SPEC_PENDING_BIT)) {
                                // a speculative camera ray (see the end of the shade phase) is not worth parking: call it off; the
                                // finalise step notices and starts the sample again, in the open
                                hits[ray] = make_float4(0.0f, 0.0f, 0.0f, as_f(POOL_CANCELLED_MARK));
                            } else if ((unsigned int)ray < n_c) {
                                const unsigned int slot = atomicAdd(A.Q.park_count, 1u);
                                if (slot < A.Q.park_cap) {
                                    const float4 t4 = ct[ray];
                                    A.Q.park[slot] = co[ray];
                                    A.Q.park[(size_t)A.Q.park_cap + slot] = cd[ray];
                                    A.Q.park[2u * (size_t)A.Q.park_cap + slot] = make_float4(t4.x, t4.y, t4.z, as_f(POOL_PARK_CLOSEST));
                                }
                                hits[ray] = make_float4(0.0f, 0.0f, 0.0f, as_f(POOL_PARKED_MARK));     // its sample leaves this pool
                            } else {
                                const unsigned int slot = atomicAdd(A.Q.park_count + 1, 1u);
                                if (slot < A.Q.spark_cap) {
                                    A.Q.spark[slot] = sq_o[(unsigned int)ray - n_c];
                                    A.Q.spark[(size_t)A.Q.spark_cap + slot] = payload;
                                    A.Q.spark[2u * (size_t)A.Q.spark_cap + slot] = make_float4(r.d.x, r.d.y, r.d.z, 0.0f);
                                }
                                if (ADAPT) {
                                    // the pixel's sample must not be finalised before this ray's radiance has landed: mark the
                                    // pixel, the finalise step parks itself when it sees the mark
                                    float * const w = &A.Q.jobsum[sample].w;
                                    *w = as_f(as_i(*w) | (int)0x80000000);
                                }
                            }
                        } else if ((unsigned int)ray < n_c) {
                            hits[ray] = make_float4(r.best.t, r.best.v, r.best.w, as_f(r.best.tri));
                        } else {
                            // shadow ray: add the precomputed radiance when unoccluded (k_trace has the commentary)
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
        }
        pool_sync<SHARED>();
        const unsigned long long ph_t2 = COUNT ? __builtin_readcyclecounter() : 0ull;

        // ---- shade: every closest hit of the pool, 64 per pass ----------------------------------------------
        {
            const PoolArgs A = pool_args(args);
            const DevScene & sc = A.sc;
            const DevParams & P = A.P;
            const WaveBuffers & B = A.B;
            const PoolBuffers & Q = A.Q;
            PRT_POOL_LISTS(A);
            unsigned int * const fin = ADAPT ? Q.fin + (size_t)wave * cap : nullptr;
            ShadeTables tb;
            tb.diffuse = LDSTAB ? s_diffuse : sc.diffuse_dirs;
            tb.materials = sc.material_count <= (unsigned int)LDS_MATS ? s_mats : sc.materials;
            tb.lights = sc.light_count <= (unsigned int)LDS_LIGHTS ? s_lights : sc.lights;
            const int nxt = cur ^ 1;
            PoolEmit<ADAPT, SHARED> emit;
            emit.co = cq_base + (size_t)nxt * 3u * cap;
            emit.cd = emit.co + cap;
            emit.ct = emit.cd + cap;
            emit.so = sq_o; emit.sc = sq_c; emit.sd = sq_d;
            emit.fin = fin;
            emit.m_c = 0; emit.m_s = 0; emit.m_f = 0; emit.m_elided = 0;
            // block-shared: the fill counts of the lists this phase writes live in LDS, one set per parity of `cur`; this set was
            // zeroed a round ago (below), the other one - read by every wave at the end of the previous round - is zeroed now,
            // and so is the trace phase's hand-out counter (the barrier behind the trace phase has passed)
            emit.s_cnt = SHARED ? &s_ctl[PCTL_CNT + 4 * nxt] : nullptr;
            if (SHARED && threadIdx.x < 4u) {
                s_ctl[PCTL_CNT + 4 * cur + threadIdx.x] = 0u;
                s_ctl[PCTL_X_PUB + threadIdx.x] = 0u;
                s_ctl[PCTL_X_TAKEN + threadIdx.x] = 0u;
                if (threadIdx.x == 0) { s_ctl[PCTL_TRACE_NEXT] = 0u; s_ctl[PCTL_X_TRACING] = (unsigned int)(BLOCK / 64); s_ctl[PCTL_X_PENDING] = 0u; }
            }
            // the unit's list entries are shaded 64 at a time; a block-shared list by the block's waves in turn
            const unsigned int b_first = SHARED ? (threadIdx.x >> 6) * 64u : 0u;
            if (ADAPT) {
                // ---- finalise: pixels whose sample ended one round ago; its last shadow rays have landed by now.
                // RenderPixel's loops (main.cpp:236-258) one step at a time: store the sample, apply the stopping rule,
                // start the next sample on the same RNG stream.
                // the pixel's sample colours, pixel-major: the variance loop of a lane walks 16-byte neighbours (four samples per
                // 64-byte line) instead of one line per sample (12-byte entries measured no better: profiles/r02_experiments.txt)
                float4 * const scratch = Q.scratch;
#define POOL_SCRATCH_AT(k, j) ((size_t)(j) * P.max_spp + (k))
                const unsigned long long ph_f0 = COUNT ? __builtin_readcyclecounter() : 0ull;
                for (unsigned int b0 = b_first; b0 < n_f; b0 += ULANES) {
                    const unsigned int i = b0 + lane;
                    const bool live = i < n_f;
                    bool go_on = false;
                    unsigned int j = 0;
                    f3 ray_o = mk3(0, 0, 0), ray_d = mk3(0, 0, 1);
                    if (live) {
                        j = fin[i];
                    }
                    int job = 0;                                                    // the pixel's state word (POOL_JOB_*)
                    unsigned int spec_idx = 0;
                    float4 spec_off = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (live) {
                        job = as_i(Q.jobsum[j].w);
                        if (job & (POOL_JOB_SPEC_FLYING | POOL_JOB_SPEC_DRAWN)) {
                            spec_off = Q.final_rgb[j];                              // (jitter x, jitter y, list index of the ray, -)
                            spec_idx = (unsigned int)as_i(spec_off.z);
                            // called off in the trace phase (it would have had to be parked)?
                            if ((job & POOL_JOB_SPEC_FLYING) && as_i(hits[spec_idx].w) == POOL_CANCELLED_MARK)
                                job = (job & ~POOL_JOB_SPEC_FLYING) | POOL_JOB_SPEC_DRAWN;
                        }
                    }
                    bool defer = false;
                    if (live && !EXACT) {
                        // a shadow ray of this pixel is parked (see the trace phase): the step waits for the EXACT launch
                        defer = (job & POOL_JOB_DEFERRED) != 0;
                        if (defer) {
                            const unsigned int slot = atomicAdd(Q.park_count, 1u);
                            if (slot < Q.park_cap) {
                                Q.park[slot] = make_float4(0.0f, 0.0f, 0.0f, as_f((int)j));
                                Q.park[(size_t)Q.park_cap + slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                                Q.park[2u * (size_t)Q.park_cap + slot] = make_float4(0.0f, 0.0f, 0.0f, as_f(POOL_PARK_FINALISE));
                            }
                            if (job & POOL_JOB_SPEC_FLYING) {
                                // its next camera ray, started ahead, cannot be shaded before the step has run: call it off (the
                                // jitter stays drawn) - the EXACT launch starts the sample again from the saved offsets
                                hits[spec_idx].w = as_f(POOL_CANCELLED_MARK);
                                job = (job & ~POOL_JOB_SPEC_FLYING) | POOL_JOB_SPEC_DRAWN;
                            }
                            Q.jobsum[j].w = as_f(job);
                        }
                    }
                    if (live && !defer) {
                        const f3 a = accum_read(B.accum + j);
                        const float4 st = Q.jobsum[j];
                        unsigned int samp = (unsigned int)(job & POOL_JOB_SAMPLE_MASK);  // index of the sample that just ended
                        const f3 sum_prev = mk3(st.x, st.y, st.z);
                        const f3 c = a;
                        // .w of a stored sample: the sum of squares of all channels of the samples up to and including it
                        const float sq_prev = samp ? scratch[POOL_SCRATCH_AT(samp - 1u, j)].w : 0.0f;
                        scratch[POOL_SCRATCH_AT(samp, j)] = make_float4(c.x, c.y, c.z, sq_prev + ((c.x * c.x + c.y * c.y) + c.z * c.z));
                        const f3 sum = sum_prev + c;                                // color += scratch_buffer[samp]
                        bool stop = false;
                        // The rule's sum of squared L1 distances to the mean is at least the sum of squared Euclidean distances,
                        // and that is at least sq - |sum|^2 / n whatever value the mean was rounded to.  When this bound clears
                        // the threshold by more than any rounding of either side (float sums of <= 50 terms: 6e-6 relative;
                        // 2e-5 of sq and 1e-3 of the threshold are allowed) the verdict is "go on" without the loop over the
                        // stored samples - which is the verdict of nearly every pixel that is not sky.  NaN fails the test.
                        bool decided = false;
                        if (samp >= P.spp) {
                            const float n = (float)samp;
                            const float bound = sq_prev - ((sum_prev.x * sum_prev.x + sum_prev.y * sum_prev.y) + sum_prev.z * sum_prev.z) / n;
                            decided = bound - 2e-5f * sq_prev > P.variance_threshold * (n - 1.0f) * 1.001f + 1e-30f;
                        }
                        if (samp >= P.spp && !decided) {                            // second loop: CalculateVariance(scratch, samp), main.cpp:253
                            const f3 mean = sum_prev / (float)samp;                 // the mean's running sum IS the colour sum so far
                            float variance = 0.0f;
                            // the sum runs in sample order, as in the reference; the LOADS do not have to: eight in flight at a
                            // time (one dependent load per iteration made this loop a chain of memory round trips)
                            unsigned int k = 0;
                            for (; k + 8u <= samp; k += 8u) {
                                float4 v[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) v[u] = scratch[POOL_SCRATCH_AT(k + (unsigned int)u, j)];
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    const float d = (fabsf(v[u].x - mean.x) + fabsf(v[u].y - mean.y)) + fabsf(v[u].z - mean.z);   // main.cpp:179-186
                                    variance += d * d;
                                }
                            }
                            for (; k < samp; ++k) {
                                const float4 v = scratch[POOL_SCRATCH_AT(k, j)];
                                const float d = (fabsf(v.x - mean.x) + fabsf(v.y - mean.y)) + fabsf(v.z - mean.z);
                                variance += d * d;
                            }
                            variance /= (float)(samp - 1u);
                            stop = variance <= P.variance_threshold;                // break BEFORE ++samp: the divisor misses this sample
                            // a verdict this close to the threshold is the one place where the device's last bits (its powf, the
                            // throughput form of the colour polynomial) could stop a pixel a sample away from the reference's:
                            // counted, so that "no close call" = "the reference's sample counts" can be checked (include/prt.h)
                            if (fabsf(variance - P.variance_threshold) <= 1e-3f * P.variance_threshold + 1e-7f) atomicAdd(&ctr->variance_close_calls, 1ull);
                        }
                        if (!stop) {
                            ++samp;
                            stop = samp >= P.max_spp;
                        }
                        if (stop) {
                            // the pixel is done; a camera ray started ahead for a sample that does not exist is called off
                            if (job & POOL_JOB_SPEC_FLYING) hits[spec_idx].w = as_f(POOL_CANCELLED_MARK);
                            const f3 out = sum / (float)samp;                       // color /= samp, main.cpp:262
                            Q.final_rgb[j] = make_float4(out.x, out.y, out.z, 1.0f);
                        } else {
                            Q.jobsum[j] = make_float4(sum.x, sum.y, sum.z, as_f((int)samp));
                            if (!(job & POOL_JOB_SPEC_FLYING)) {
                                // the sample's camera ray is not in the pool yet (the common case is that it is: see below)
                                float off_x, off_y;
                                if (job & POOL_JOB_SPEC_DRAWN) {
                                    off_x = spec_off.x; off_y = spec_off.y;              // drawn ahead, the ray was called off
                                } else {
                                    Rng rng;
                                    const ulonglong2 rs = B.rng[j], ra = B.rng_aux[j];
                                    rng.chain = rs.x; rng.prev = rs.y; rng.seed0 = ra.x; rng.k = (u32)ra.y;
                                    u64 * ring = B.ring + (size_t)j * B.ring_step;
                                    off_y = rng_float11<true>(rng, ring, B.ring_stride);          // first draw -> .y (main.cpp:238, 247)
                                    off_x = rng_float11<true>(rng, ring, B.ring_stride);
                                    B.rng[j] = make_ulonglong2(rng.chain, rng.prev);
                                    B.rng_aux[j] = make_ulonglong2(rng.seed0, (u64)rng.k);
                                }
                                const float jitter = samp < P.spp ? 0.5f : 1.0f;        // main.cpp:240 vs :249
                                const unsigned int pixel = pixel_of_local(P, B.sample_base + j);
                                const unsigned int x = pixel % P.width, y = pixel / P.width;
                                const DevCamera cam = A.cam;
                                ray_o = cam.position;
                                ray_d = make_camera_dir(cam, (float)x + off_x * jitter, (float)y + off_y * jitter);
                                go_on = true;
                            }
                        }
                    }
                    emit.closest(go_on, j, ray_o, ray_d, mk3(1.0f, 1.0f, 1.0f), 0, (unsigned int)WF_PENDING_FRESH_BIT, false);
                }
                emit.m_f = 0;                                                       // the list is consumed; shading refills it from 0
                pool_sync<SHARED>();                                                // (block-shared: by every wave, before any wave refills it)
                if (COUNT) ph_final += __builtin_readcyclecounter() - ph_f0;
            }
            for (unsigned int b0 = b_first; b0 < n_c; b0 += ULANES) {
                const unsigned int i = b0 + lane;
                const bool live = i < n_c;
                unsigned int s = 0;
                int level = 0;
                unsigned int pending = 0;
                f3 ray_o = mk3(0, 0, 0), ray_d = mk3(0, 0, 1), T = mk3(0, 0, 0);
                HitRec hit;
                hit.t = 0.0f; hit.v = hit.w = 0.0f; hit.tri = -1;
                if (live) {
                    const float4 ro = co[i], rd = cd[i], rt = ct[i], h = hits[i];
                    s = (unsigned int)as_i(ro.w);
                    level = as_i(rd.w) & 0xFF;
                    pending = ((unsigned int)as_i(rd.w)) >> 8;
                    ray_o = mk3(ro.x, ro.y, ro.z);
                    ray_d = mk3(rd.x, rd.y, rd.z);
                    T = mk3(rt.x, rt.y, rt.z);
                    hit.t = h.x; hit.v = h.y; hit.w = h.z; hit.tri = as_i(h.w);
                }
                const bool parked = !EXACT && live && hit.tri == POOL_PARKED_MARK;      // continues in the EXACT launch
                // adaptive mode: a camera ray started ahead of its pixel's finalise step, which then ended the pixel (or had to
                // wait): as if it had never been traced
                const bool cancelled = ADAPT && live && hit.tri == POOL_CANCELLED_MARK;
                if (ADAPT) {
                    pending &= ~(unsigned int)POOL_SPEC_PENDING_BIT;
                    rays -= (unsigned long long)__popcll(__ballot(cancelled));
                }
                unsigned int shaded = 0;
                // the frame under construction lives in this lane's (idle) traversal stack column
                shade_entry_lds<RING, TEX, BLOCK, RINGMEM>(sc, P, B, tb, live && !parked && !cancelled, s, level, pending, ray_o, ray_d, T, hit, emit, shaded, stack.frame_col());
                shaded_w += (unsigned int)__popcll(__ballot(shaded != 0));
            }
            if (ADAPT) {
                // ---- start ahead: a pixel whose sample just ended would sit out the next trace phase - its finalise step has to
                // wait for the sample's last shadow rays - and that is one round in three or four with no ray from the pixel.
                // The next sample's camera ray needs nothing from that step but the fact that there IS a next sample: below
                // max_spp - 1 only the variance rule can end the pixel, and it rarely does.  So the jitter is drawn now (the
                // RNG state is final when a sample ends: shadow rays draw nothing) and the ray joins the list, marked; the
                // finalise step calls it off if the pixel ends, and a called-off ray is not shaded and not counted.
                pool_sync<SHARED>();
                const unsigned int ended = SHARED ? emit.s_cnt[2] : emit.m_f;
                unsigned int started = 0;
                for (unsigned int b0 = b_first; b0 < ended; b0 += ULANES) {
                    const unsigned int i = b0 + lane;
                    bool go = false;
                    unsigned int j = 0;
                    f3 ray_o = mk3(0, 0, 0), ray_d = mk3(0, 0, 1);
                    float off_x = 0.0f, off_y = 0.0f;
                    if (i < ended) {
                        j = fin[i];
                        const int job = as_i(Q.jobsum[j].w);
                        const unsigned int next = (unsigned int)(job & POOL_JOB_SAMPLE_MASK) + 1u;
                        if (next < P.max_spp && !(job & (POOL_JOB_SPEC_FLYING | POOL_JOB_SPEC_DRAWN))) {
                            Rng rng;
                            const ulonglong2 rs = B.rng[j], ra = B.rng_aux[j];
                            rng.chain = rs.x; rng.prev = rs.y; rng.seed0 = ra.x; rng.k = (u32)ra.y;
                            u64 * ring = B.ring + (size_t)j * B.ring_step;
                            off_y = rng_float11<true>(rng, ring, B.ring_stride);              // first draw -> .y (main.cpp:238, 247)
                            off_x = rng_float11<true>(rng, ring, B.ring_stride);
                            B.rng[j] = make_ulonglong2(rng.chain, rng.prev);
                            B.rng_aux[j] = make_ulonglong2(rng.seed0, (u64)rng.k);
                            const float jitter = next < P.spp ? 0.5f : 1.0f;        // main.cpp:240 vs :249
                            const unsigned int pixel = pixel_of_local(P, B.sample_base + j);
                            const unsigned int x = pixel % P.width, y = pixel / P.width;
                            const DevCamera cam = A.cam;
                            ray_o = cam.position;
                            ray_d = make_camera_dir(cam, (float)x + off_x * jitter, (float)y + off_y * jitter);
                            go = true;
                        }
                    }
                    const unsigned long long m = __ballot(go);
                    const unsigned int slot = emit.closest(go, j, ray_o, ray_d, mk3(1.0f, 1.0f, 1.0f), 0, (unsigned int)POOL_SPEC_PENDING_BIT | (unsigned int)WF_PENDING_FRESH_BIT, false);
                    if (go) {
                        Q.final_rgb[j] = make_float4(off_x, off_y, as_f((int)slot), 0.0f);
                        Q.jobsum[j].w = as_f(as_i(Q.jobsum[j].w) | POOL_JOB_SPEC_FLYING);
                    }
                    started += (unsigned int)__popcll(m);
                }
                if (SHARED) { if (started && lane == 0) atomicAdd(&emit.s_cnt[3], started); }
                else n_spec = started;
            }
            rays += emit.m_elided;                                             // counted as the reference counts them (raytracer.cpp:161)
            elided_w += emit.m_elided;
            if (SHARED) {
                __syncthreads();                                               // every wave's appends are counted
                n_c = emit.s_cnt[0]; n_s = emit.s_cnt[1]; n_f = emit.s_cnt[2];
                if (ADAPT) n_spec = emit.s_cnt[3];
            } else {
                n_c = emit.m_c;
                n_s = emit.m_s;
                n_f = emit.m_f;
            }
            cur = nxt;
        }
        if (!SHARED) pool_fence();
        if (COUNT) {
            const unsigned long long ph_t3 = __builtin_readcyclecounter();
            ph_topup += ph_t1 - ph_t0; ph_trace += ph_t2 - ph_t1; ph_shade += ph_t3 - ph_t2;
        }
    }

    // ---- counters: one atomic per workgroup and counter ----------------------------------------------------
    if (lane == 0) {
        atomicAdd(&s_red[0], rays);
        atomicAdd(&s_red[1], (unsigned long long)shaded_w);
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
        if (dry_steps) { atomicAdd(&ctr->drain_node_steps, dry_steps); atomicAdd(&ctr->drain_node_step_rays, dry_rays); }
        if (lane == 0) {
            atomicAdd(&ctr->phase_cycles[0], ph_topup);
            atomicAdd(&ctr->phase_cycles[1], ph_trace);
            atomicAdd(&ctr->phase_cycles[2], ph_shade);
            const unsigned long long ph_all = __builtin_readcyclecounter() - ph_begin;
            atomicAdd(&ctr->phase_cycles[3], ph_all);
            if (!EXACT) {
                unsigned long long * const wt = ((PoolArgsPtr)args)->Q.wave_times;
                if (wt) {
                    unsigned long long * const w3 = as_global(wt) + 3u * (size_t)(slot_id >> 6);
                    w3[0] = wt_begin; w3[1] = wt_dry; w3[2] = wall_clock64();
                }
                atomicMax(&ctr->wave_cycles_max, ph_all);
                atomicAdd(&ctr->wave_cycles_sum, ph_all);
                atomicAdd(&ctr->wave_count, 1ull);
            }
            if (ADAPT) atomicAdd(&ctr->phase_cycles[4], ph_final);
        }
    }
}

#undef PRT_POOL_LISTS

// Between the fast k_pool and its adopting EXACT launch: the parked shadow rays, traced exactly (trace_ray on a full-height
// global stack; a point light's shadow ray, which uses its hit distance, gets its near ties resolved) and their radiance
// added.  A small fixed grid that reads the list length on the device and normally finds 0.
template <bool COUNT>
__global__ __launch_bounds__(256) void k_pool_parked_shadows(const PoolArgs * args, DevCounters * ctr) {
    const PoolArgs A = pool_args(args);
    const DevScene & sc = A.sc;
    const DevParams & P = A.P;
    const unsigned int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int n = A.Q.park_count[1] < A.Q.spark_cap ? A.Q.park_count[1] : A.Q.spark_cap;
    if (gid == 0) {
        atomicMax(&ctr->park_peak[0], (unsigned long long)A.Q.park_count[0]);
        atomicMax(&ctr->park_peak[1], (unsigned long long)A.Q.park_count[1]);
        // decided HERE, per launch, against the capacities this launch's lists really have (a later pass of the same call may
        // have shorter ones: its clamp is its own worst case)
        if (A.Q.park_count[0] > A.Q.park_cap) atomicMax(&ctr->park_over[0], (unsigned long long)A.Q.park_count[0]);
        if (A.Q.park_count[1] > A.Q.spark_cap) atomicMax(&ctr->park_over[1], (unsigned long long)A.Q.park_count[1]);
    }
    TraceStats st;
    st.nodes = st.tris = st.wnodes = st.wleaves = st.wtris = st.wrefills = st.wrays = st.max_sp = st.culled = 0;
    GlobalStack slow;
    slow.attach(P.exact_stack, gid, P.exact_stack_stride);
    for (unsigned int i = gid; i < n; i += gridDim.x * blockDim.x) {
        const float4 ro = A.Q.spark[i], pay = A.Q.spark[(size_t)A.Q.spark_cap + i], rd = A.Q.spark[2u * (size_t)A.Q.spark_cap + i];
        const int sample = as_i(ro.w);
        const f3 d = mk3(rd.x, rd.y, rd.z);
        const f3 ob = mk3(ro.x, ro.y, ro.z) + d * P.ray_bias;                   // raytracer.cpp:163
        const HitRec h = trace_ray<GlobalStack, COUNT>(sc, ob, d, pay.w < 0.0f ? TRACE_ANY : TRACE_CLOSEST, P.box_pad, slow, st);
        const bool lit = h.tri < 0 || (pay.w >= 0.0f && h.t * h.t <= pay.w);
        if (lit) accum_add(A.B.accum + sample, mk3(pay.x, pay.y, pay.z));
    }
    if (COUNT) {
        atomicAdd(&ctr->node_visits, (unsigned long long)st.nodes);
        atomicAdd(&ctr->tri_tests, (unsigned long long)st.tris);
    }
}

}  // namespace prt
