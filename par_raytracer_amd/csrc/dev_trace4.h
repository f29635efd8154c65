// dev_trace4.h - traversal of the quantised 4-wide BVH (64 B nodes, children sorted by entry distance at every step).
// The default.  Round 3 built the 8-wide alternative (dev_trace8.h, -DPRT_BVH8) and measured the two against each other
// (profiles/r03_ab_bvh8.txt): level on the headline frame until the shading phase got lighter, then this one 1 - 2 % ahead.
#pragma once

#include "dev_trace_common.h"

namespace prt {

// Per-lane traversal registers.  A ray can be suspended and resumed at any node boundary (the persistent
// kernel does so when it leaves the traversal loop to refill idle lanes).
struct TravRay {
    f3 o, d;                          // origin ALREADY biased by direction * ray_bias (raytracer.cpp:163), direction
    float ix, iy, iz;                 // 1 / direction, components clamped away from 0
    float pnx, pny, pnz;              // (o +- pad) / direction for the plane the ray ENTERS through on each axis
#if defined(PRT_BVH4_SIX_PLANE_OFFSETS)
    float pfx, pfy, pfz;              // ... and for the plane it LEAVES through (pad always widens the box)
#endif                                // (default: the exit side's offset is the entry side's plus 2 pad |1 / d|, formed per step - three registers fewer)
    HitRec best;
    int node, sp, kind;               // kind: TRACE_CLOSEST / TRACE_ANY
};


// Bottom-of-stack marker.  A traversal ends when it pops it.  The two rare things a traversal has to report - a candidate
// within 2^-19 of the best hit (resolve_near_ties() must decide), a push that did not fit the stack - are recorded IN the
// marker, i.e. in slot 0 of the lane's own stack column, so that they cost the loop no register: the marker that comes back
// tells the story.  None of the four values is a valid leaf link (their first_tri would be 2^29 - 1; upload caps the
// triangle count below that).
enum { TRAV_SENTINEL = (int)0x80000000, TRAV_FLAG_NEAR = 1, TRAV_FLAG_OVERFLOW = 2, TRAV_SENTINEL_LAST = (int)0x80000003 };
PRT_D bool trav_done(int node) { return node <= TRAV_SENTINEL_LAST; }                              // the marker was popped
PRT_D bool trav_flagged(int node) { return node <= TRAV_SENTINEL_LAST && node != TRAV_SENTINEL; }

// Traversal stacks.  LdsStack: this lane's column of a workgroup LDS array (entry e of lane l at col[e*BLOCK + l]: a wave's
// push/pop of one level is one conflict-free ds_write/ds_read_b32).  Its height bounds occupancy, so it is sized for what
// rays really use (<= 24 entries; the deepest ever observed on the 1M triangle scene is 16) and not for the worst case (3
// pushes per 4-wide level).  A push that does not fit is DROPPED and the marker gets TRAV_FLAG_OVERFLOW: the result of such
// a ray is not trustworthy (a found any-hit occluder still is) and the kernel hands the ray to a slow path that traces it
// again on a stack that holds the whole bound.  The hot loop therefore carries one compare per push and no spill code.
// LdsSpillStack: the same column, continued in a per-lane global column behind it; never overflows (slow paths only: the
// extra branch per push and pop costs the fast kernels 5 %).  GlobalStack: a whole column in global memory.
typedef int StackEntry;                  // one link per entry
enum { STACK_ENTRY_INTS = 1, BVH_NODE_BYTES = 64, STACK_LDS_CAP_DEFAULT = 24 };

template <int BLOCK>
struct LdsStack {
    int * col;
    unsigned int cap;
    PRT_D void attach(int * lds, unsigned int tid) { col = lds + tid; }
    PRT_D int * frame_col() const { return col; }      // the lane's column as plain dwords, stride BLOCK (the shading frame lives there)
    PRT_D bool push(int sp, int v) const {
        if ((unsigned int)sp < cap) { col[sp * BLOCK] = v; return true; }
        return false;
    }
    PRT_D int pop(int sp) const { return col[sp * BLOCK]; }
    PRT_D void flag(int bit) const { col[0] |= bit; }
};

template <int BLOCK>
struct LdsSpillStack {
    int * col;
    unsigned int cap;
    PRT_D void attach(int * lds, unsigned int tid) { col = lds + tid; }
    PRT_D int * frame_col() const { return col; }
    PRT_D void set_spill(int * base, unsigned int lanes) { spill = base; spill_stride = lanes; }
    int * spill;                      // WAVE-UNIFORM base of the spill area (null when cap covers the bound): entry cap + k of
    unsigned int spill_stride;        // the lane with global thread id g at spill[k * spill_stride + g] - no per-lane pointer is kept
    PRT_D size_t spill_index(int sp) const {
        return (size_t)((unsigned int)sp - cap) * spill_stride + (blockIdx.x * (unsigned int)BLOCK + threadIdx.x);
    }
    PRT_D bool push(int sp, int v) const {
        if ((unsigned int)sp < cap) col[sp * BLOCK] = v;
        else spill[spill_index(sp)] = v;
        return true;
    }
    PRT_D int pop(int sp) const {
        if ((unsigned int)sp < cap) return col[sp * BLOCK];
        return spill[spill_index(sp)];
    }
    PRT_D void flag(int bit) const { col[0] |= bit; }
};

struct GlobalStack {
    int * col;
    size_t stride;
    PRT_D void attach(int * base, size_t lane, size_t lanes) { col = base + lane; stride = lanes; }
    PRT_D bool push(int sp, int v) const { col[(size_t)sp * stride] = v; return true; }
    PRT_D int pop(int sp) const { return col[(size_t)sp * stride]; }
    PRT_D void flag(int bit) const { col[0] |= bit; }
};

// The states of a lane's ray, as the kernels' loops ask for them
PRT_D void trav_idle(TravRay & r) { r.node = TRAV_SENTINEL; r.sp = 0; r.kind = 0; }          // no ray
PRT_D bool trav_walking(const TravRay & r) { return r.node >= 0; }                            // wants a node step (else: holds a leaf, or is done)
PRT_D bool trav_done(const TravRay & r) { return trav_done(r.node); }

// After a traversal ended: what its marker says.  0 for a ray that ended on an any-hit occluder (it never pops the marker,
// and a found occluder is final whatever happened before).
template <class STK> PRT_D int trav_end_flags(const TravRay & r, const STK &) { return trav_done(r.node) ? (r.node & 3) : 0; }
// the hit of a closest-hit ray has company within a few ulp: the reference's visit order decides (resolve_near_ties)
template <class STK> PRT_D bool trav_wants_resolve(const TravRay & r, const STK & stk) { return r.kind == TRACE_CLOSEST && (trav_end_flags(r, stk) & TRAV_FLAG_NEAR) != 0 && r.best.tri >= 0; }
// fast kernels: the ray cannot be finished here (near tie, or a dropped push): it goes to the slow path
template <class STK> PRT_D bool trav_needs_slow_path(const TravRay & r, const STK & stk) { return (trav_end_flags(r, stk) & TRAV_FLAG_OVERFLOW) != 0 || trav_wants_resolve(r, stk); }

// Pop and push through the ray.  (The top entry mirrored in a register, so that the entry a pop hands out was read one step
// earlier and the LDS latency is off the node-to-node chain, measured 0.8 % slower: profiles/r02_experiments.txt item 19.)
template <class STK>
PRT_D void trav_pop(TravRay & r, const STK & stk) {
    r.sp--;
    r.node = stk.pop(r.sp);
}
template <class STK>
PRT_D void trav_push(TravRay & r, const STK & stk, int link) {
    if (stk.push(r.sp, link)) r.sp++;
    else stk.flag(TRAV_FLAG_OVERFLOW);
}

template <class STK>
PRT_D void trav_init(TravRay & r, f3 o, f3 d, int kind, float pad, const STK & stk) {
    r.o = o;
    r.d = d;
    // direction components are clamped away from 0 so no inf/NaN enters the box test
    const float tiny = 1e-30f;
    float dx = fabsf(d.x) < tiny ? (d.x < 0.0f ? -tiny : tiny) : d.x;
    float dy = fabsf(d.y) < tiny ? (d.y < 0.0f ? -tiny : tiny) : d.y;
    float dz = fabsf(d.z) < tiny ? (d.z < 0.0f ? -tiny : tiny) : d.z;
    r.ix = 1.0f / dx; r.iy = 1.0f / dy; r.iz = 1.0f / dz;
    // direction >= 0: enters through the lo plane (seen from o + pad), leaves through hi (from o - pad); else swapped
    r.pnx = (dx < 0.0f ? o.x - pad : o.x + pad) * r.ix;
    r.pny = (dy < 0.0f ? o.y - pad : o.y + pad) * r.iy;
    r.pnz = (dz < 0.0f ? o.z - pad : o.z + pad) * r.iz;
#if defined(PRT_BVH4_SIX_PLANE_OFFSETS)
    r.pfx = (dx < 0.0f ? o.x + pad : o.x - pad) * r.ix;
    r.pfy = (dy < 0.0f ? o.y + pad : o.y - pad) * r.iy;
    r.pfz = (dz < 0.0f ? o.z + pad : o.z - pad) * r.iz;
#endif
    r.best.t = 3.402823466e+38f;
    r.best.v = r.best.w = 0.0f;
    r.best.tri = -1;
    r.kind = kind;
    stk.push(0, TRAV_SENTINEL);
    r.sp = 1;
    r.node = 0;
}

// A ray's traversal registers as dwords, and its stack column copied from another lane's: what a wave needs to take over a ray
// another wave of its workgroup was tracing (kernels_pool.h: rays handed over at the end of a round).  Field k at dst[k * stride].
#if defined(PRT_BVH4_SIX_PLANE_OFFSETS)
#error "trav_save_regs / trav_restore_regs cover the default ray state (three plane offsets)"
#endif
enum { TRAV_STATE_DWORDS = 19 };
PRT_D void trav_save_regs(const TravRay & r, float * dst, unsigned int stride) {
    const float f[TRAV_STATE_DWORDS] = { r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, r.ix, r.iy, r.iz, r.pnx, r.pny, r.pnz,
                                         r.best.t, r.best.v, r.best.w, as_f(r.best.tri), as_f(r.node), as_f(r.sp), as_f(r.kind) };
#pragma unroll
    for (int k = 0; k < TRAV_STATE_DWORDS; ++k) dst[(size_t)k * stride] = f[k];
}
PRT_D void trav_restore_regs(TravRay & r, const float * src, unsigned int stride) {
    float f[TRAV_STATE_DWORDS];
#pragma unroll
    for (int k = 0; k < TRAV_STATE_DWORDS; ++k) f[k] = src[(size_t)k * stride];
    r.o = mk3(f[0], f[1], f[2]); r.d = mk3(f[3], f[4], f[5]);
    r.ix = f[6]; r.iy = f[7]; r.iz = f[8]; r.pnx = f[9]; r.pny = f[10]; r.pnz = f[11];
    r.best.t = f[12]; r.best.v = f[13]; r.best.w = f[14]; r.best.tri = as_i(f[15]);
    r.node = as_i(f[16]); r.sp = as_i(f[17]); r.kind = as_i(f[18]);
}
template <class STK>
PRT_D void trav_copy_stack(const STK & dst, const STK & src, int sp) {
    for (int e = 0; e < sp; ++e) dst.push(e, src.pop(e));
}

PRT_D void cswap(float & ka, float & kb, int & la, int & lb) {
    const bool sw = kb < ka;
    const float k0 = sw ? kb : ka, k1 = sw ? ka : kb;
    const int l0 = sw ? lb : la, l1 = sw ? la : lb;
    ka = k0; kb = k1; la = l0; lb = l1;
}

// One 4-wide node: fetch 64 B, dequantise + slab-test four child boxes, sort the hit children by entry
// distance, descend into the nearest and push the others (farthest first).
//   plane = origin + q * 2^e  =>  t = (plane - o -+ pad) / d = q * (2^e / d) + (origin / d - (o +- pad) / d)
// so after 3 scale products and 6 FMAs per node every plane costs one byte->float convert and one FMA.
// The slab test may use FMA: it only has to be conservative, and the boxes are widened by `pad`.
// Experiment PRT_TOP_LDS = n (round 4, the north star's "BVH nodes staged through LDS" on the production kernel): the first n
// nodes of the breadth-first array - the top levels every ray walks - live in a workgroup LDS table `top` and a step at one of
// them reads LDS instead of the vector L1.  Measured in profiles/r04_ab_top_levels_in_lds.txt; not in the shipped build.
#if defined(PRT_TOP_LDS)
enum { TRAV_TOP_LDS_NODES = PRT_TOP_LDS };
#else
enum { TRAV_TOP_LDS_NODES = 0 };
#endif

template <class STK, bool COUNT>
PRT_D void trav_node_step(const DevScene & sc, TravRay & r, const STK & stk, TraceStats & st, float pad, const uint4 * top = nullptr) {
    // 32-bit byte offset from the (scalar) array base: the loads take the SGPR-base + VGPR-offset form and no 64-bit address is
    // built per lane (-0.7 % frame time; upload caps the scene at 2^26 triangles, so nodes * 64 and triangles * 48 fit)
    const uint4 * np = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(sc.nodes) + ((unsigned int)r.node << 6));
    uint4 w0, w1, w2, w3;
    if (TRAV_TOP_LDS_NODES > 0 && top && r.node < (int)TRAV_TOP_LDS_NODES) {
        const uint4 * tp = top + 4 * r.node;
        w0 = tp[0]; w1 = tp[1]; w2 = tp[2]; w3 = tp[3];
    } else {
        w0 = np[0]; w1 = np[1]; w2 = np[2]; w3 = np[3];
    }
    if (COUNT) { st.nodes++; if (first_active_lane()) st.wnodes++; if ((unsigned int)r.sp > st.max_sp) st.max_sp = (unsigned int)r.sp; }
#ifdef PRT_PROBE_EXTRA_LOAD
    // sensitivity probe (tools/ab_probe.sh): one more divergent vector-memory instruction per node step
    { unsigned int e; asm volatile("global_load_dword %0, %1, off offset:32\n\ts_waitcnt vmcnt(0)" : "=v"(e) : "v"(np) : "memory"); }
#endif
#ifdef PRT_PROBE_EXTRA_VALU
    // sensitivity probe: PRT_PROBE_EXTRA_VALU more vector ALU instructions per node step, on a value the step needs
    { unsigned int x = w0.w;
#pragma unroll
      for (int i = 0; i < PRT_PROBE_EXTRA_VALU; ++i) asm volatile("v_mov_b32 %0, %0" : "+v"(x));
      const_cast<uint4 &>(w0).w = x; }
#endif
    const float kx = __uint_as_float(w0.w) * r.ix;
    const float ky = __uint_as_float(w3.z) * r.iy;
    const float kz = __uint_as_float(w3.w) * r.iz;
    const float ox = __uint_as_float(w0.x), oy = __uint_as_float(w0.y), oz = __uint_as_float(w0.z);
    // entry / exit parameter of the node origin on each axis; the ray's direction signs pick, per axis, which
    // quantised plane set (lo or hi bytes) is the entry side - no per-plane min/max, and an empty child slot
    // (lo = 255 > hi = 0 on every axis) can never satisfy entry <= exit.
#if defined(PRT_BVH4_SIX_PLANE_OFFSETS)
    const float cnx = __builtin_fmaf(ox, r.ix, -r.pnx), cfx = __builtin_fmaf(ox, r.ix, -r.pfx);
    const float cny = __builtin_fmaf(oy, r.iy, -r.pny), cfy = __builtin_fmaf(oy, r.iy, -r.pfy);
    const float cnz = __builtin_fmaf(oz, r.iz, -r.pnz), cfz = __builtin_fmaf(oz, r.iz, -r.pfz);
#else
    // (o -+ pad) / d on the exit side = the entry side's value + 2 pad |1 / d|: same six FMAs, three ray registers fewer
    const float pad2 = pad + pad;
    const float cnx = __builtin_fmaf(ox, r.ix, -r.pnx), cfx = __builtin_fmaf(pad2, fabsf(r.ix), cnx);
    const float cny = __builtin_fmaf(oy, r.iy, -r.pny), cfy = __builtin_fmaf(pad2, fabsf(r.iy), cny);
    const float cnz = __builtin_fmaf(oz, r.iz, -r.pnz), cfz = __builtin_fmaf(pad2, fabsf(r.iz), cnz);
#endif
    const bool sx = r.ix < 0.0f, sy = r.iy < 0.0f, sz = r.iz < 0.0f;
    const unsigned int qnx = sx ? w1.w : w1.x, qfx = sx ? w1.x : w1.w;
    const unsigned int qny = sy ? w2.x : w1.y, qfy = sy ? w1.y : w2.x;
    const unsigned int qnz = sz ? w2.y : w1.z, qfz = sz ? w1.z : w2.y;
    float key[4];
    int link[4] = { (int)w2.z, (int)w2.w, (int)w3.x, (int)w3.y };
    const float inf = __uint_as_float(0x7F800000u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float nx = __builtin_fmaf((float)((qnx >> (8 * k)) & 0xFFu), kx, cnx);
        const float fx = __builtin_fmaf((float)((qfx >> (8 * k)) & 0xFFu), kx, cfx);
        const float ny = __builtin_fmaf((float)((qny >> (8 * k)) & 0xFFu), ky, cny);
        const float fy = __builtin_fmaf((float)((qfy >> (8 * k)) & 0xFFu), ky, cfy);
        const float nz = __builtin_fmaf((float)((qnz >> (8 * k)) & 0xFFu), kz, cnz);
        const float fz = __builtin_fmaf((float)((qfz >> (8 * k)) & 0xFFu), kz, cfz);
        const float tmin = fmaxf(fmaxf(fmaxf(nx, ny), nz), 0.0f);
        const float tmax = fminf(fminf(fminf(fx, fy), fz), r.best.t);
        key[k] = tmin <= tmax ? tmin : inf;
    }
    // sorting network for 4 keys, ascending; misses (inf) sink to the end
    cswap(key[0], key[1], link[0], link[1]);
    cswap(key[2], key[3], link[2], link[3]);
    cswap(key[0], key[2], link[0], link[2]);
    cswap(key[1], key[3], link[1], link[3]);
    cswap(key[1], key[2], link[1], link[2]);
    if (key[0] < inf) {
        if (key[3] < inf) trav_push(r, stk, link[3]);
        if (key[2] < inf) trav_push(r, stk, link[2]);
        if (key[1] < inf) trav_push(r, stk, link[1]);
        r.node = link[0];
    } else {
        if (COUNT && r.best.tri >= 0) st.culled++;
        trav_pop(r, stk);
    }
}

// The leaf in r.node: test its triangles, then pop.  Returns true when an any-hit ray found its hit.
template <class STK, bool COUNT>
PRT_D bool trav_leaf(const DevScene & sc, TravRay & r, const STK & stk, TraceStats & st) {
    const f3 qp = r.o - (r.o + r.d);                 // raytracer.cpp:88-89, not bitwise -d
    const unsigned int leaf = (unsigned int)~r.node;
    const unsigned int first = leaf >> 2, count = (leaf & 3u) + 1u;
    if (COUNT) { if (first_active_lane()) st.wleaves++; }
    // the entry this leaf will pop, read before the triangle tests (nothing pushes while a leaf is tested): its LDS latency
    // runs beside the triangle fetches instead of between the last test and the next node's fetch (-1.1 % frame time;
    // requesting the next triangle before testing this one gains as much alone and loses with this: registers)
    int next_node = stk.pop(r.sp - 1);
    bool flagged = false;
    for (unsigned int i = 0; i < count; ++i) {
        const unsigned int ti = first + i;
        const float4 * tp = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sc.tris) + ti * 48u);
        const float4 r0 = tp[0], r1 = tp[1], r2 = tp[2];
        // all 48 bytes at once: left alone, the compiler sinks the load of `a` below the facing test - a second memory round
        // trip inside every test of a front-facing triangle, 3.8 % of the frame (profiles/r02_experiments.txt item 26)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" :: "v"(r0.x), "v"(r0.y), "v"(r0.z));
#endif
        if (COUNT) { st.tris++; if (first_active_lane()) st.wtris++; }
        float t, v, w;
        bool near;
        const bool hit = tri_test(r.o, r.d, qp, mk3(r0.x, r0.y, r0.z), mk3(r0.w, r1.x, r1.y), mk3(r1.z, r1.w, r2.x),
                                  mk3(r2.y, r2.z, r2.w), r.best.t, t, v, w, near);
        if (near) {
            stk.flag(TRAV_FLAG_NEAR);
            flagged = true;
        }
        if (hit) {
            r.best.t = t;
            r.best.v = v;
            r.best.w = w;
            r.best.tri = (int)ti;
            if (r.kind & TRACE_ANY) return true;
        }
    }
    if (flagged && r.sp == 1) next_node |= TRAV_FLAG_NEAR;      // the entry read ahead was the marker itself
    r.sp--;
    r.node = next_node;
    return false;
}

// ---------------------------------------------------------------------------------------------------------------------
// Near ties: the reference's answer for a ray whose closest hit has company within a few ulp.
//
// The reference keeps `best` (FLT_MAX at first) and offers it every triangle in ITS visit order - sphere tree depth first,
// c1 before c0, groups in leaf order, triangles in index order (raytracer.cpp:136, 208-209); a triangle replaces best iff
// !(t > best * d) and t / d < best (:104, :149, :220).  Far from best both comparisons say the same; within an ulp or two
// they need not, so which of several near-coincident hits survives depends on the order.  What cannot depend on it:
// let N be the candidates with t <= bound, where no candidate lies in the "moat" (bound, bound * (1 + 2^-20)].  Then
//   - every member of N beats any best that is not in N on both comparisons with room to spare, so the first member the
//     reference meets is accepted whatever came before it;
//   - from then on best <= bound, and nothing outside N can pass `t / d < best`.
// Hence the reference's final hit is its own filter run over N alone, in its visit order, from FLT_MAX.  That is what this
// function does: N's members are fetched one by one in visit order (tri_rank) - each fetch a traversal bounded by `bound`,
// no storage needed - and put through tri_test_ref.  If the moat turns out to be occupied the bound is widened and the
// replay starts over; after sc.tie_widen_max widenings the replay is finished over the set as it stands and the event is
// counted (DevScene::near_tie_unresolved: the render call then fails).  The reference also skips a whole GROUP whose bounding
// sphere it enters later than its best hit so far (raytracer.cpp:176-181): RefSphereWalk (dev_trace_common.h) replays that.
template <class STK, bool COUNT>
PRT_D HitRec resolve_near_ties(const DevScene & sc, f3 o, f3 d, float pad, float min_t, const STK & stk, TraceStats & st) {
    TravRay r;
    HitRec result;
    result.t = 3.402823466e+38f; result.v = result.w = 0.0f; result.tri = -1;
    const f3 qp = o - (o + d);
    float bound = min_t * PRT_TIE_NEAR;
    for (unsigned int widen = 0; ; ++widen) {
        const bool last = widen >= sc.tie_widen_max;
        const float moat = bound * 1.00000095367431640625f;          // 1 + 2^-20
        bool occupied = false;
        float best = 3.402823466e+38f;                              // the reference's best_hit.t, replayed
        result.tri = -1;
        unsigned int next_rank = 0;                                 // candidates of rank >= next_rank are still to come
        RefSphereWalk walk;
        walk.reset();
        for (;;) {
            // the member of N with the smallest rank >= next_rank
            unsigned int c_rank = 0xFFFFFFFFu;
            int c_tri = -1;
            trav_init(r, o, d, TRACE_CLOSEST, pad, stk);
            r.best.t = moat;                                        // the boxes are culled against the moat's far side
            for (;;) {
                while (r.node >= 0) trav_node_step<STK, COUNT>(sc, r, stk, st, pad);
                if (trav_done(r.node)) break;
                const unsigned int leaf = (unsigned int)~r.node;
                const unsigned int first = leaf >> 2, count = (leaf & 3u) + 1u;
                for (unsigned int i = 0; i < count; ++i) {
                    const unsigned int ti = first + i;
                    const float4 * tp = sc.tris + 3 * (size_t)ti;
                    const float4 r0 = tp[0], r1 = tp[1], r2 = tp[2];
                    float t, dd, v, w;
                    if (!tri_geom(o, qp, mk3(r0.x, r0.y, r0.z), mk3(r0.w, r1.x, r1.y), mk3(r1.z, r1.w, r2.x), mk3(r2.y, r2.z, r2.w), t, dd, v, w)) continue;
                    const float th = t * (1.0f / dd);
                    if (th > bound) { if (th <= moat) occupied = true; continue; }
                    const unsigned int rk = sc.tri_rank[ti];
                    if (rk >= next_rank && rk < c_rank) { c_rank = rk; c_tri = (int)ti; }
                }
                trav_pop(r, stk);
            }
            if (c_tri < 0 || (occupied && !last)) break;
            const float4 * tp = sc.tris + 3 * (size_t)c_tri;
            const float4 r0 = tp[0], r1 = tp[1], r2 = tp[2];
            float t, v, w;
            if (walk.offers(sc, o, d, c_rank, best) &&
                tri_test_ref(o, qp, mk3(r0.x, r0.y, r0.z), mk3(r0.w, r1.x, r1.y), mk3(r1.z, r1.w, r2.x), mk3(r2.y, r2.z, r2.w), best, t, v, w)) {
                best = t;
                result.t = t; result.v = v; result.w = w; result.tri = c_tri;
            }
            next_rank = c_rank + 1u;
        }
        if (!occupied) break;
        if (last) {                                                 // gave up widening: the replay ran over the set as it stood
            if (sc.near_tie_unresolved) atomicAdd(sc.near_tie_unresolved, 1ull);
            break;
        }
        bound = moat * PRT_TIE_NEAR;                                // take the moat's occupants in and try again
    }
    return result;
}

// Whole-ray traversal, "while-while" (Aila & Laine): every lane first walks internal nodes until it holds
// a leaf (or runs out of work), and only then does the wave run the triangle code.  With 64 lanes a fused
// node-or-leaf loop would execute the (4x longer) leaf body in almost every iteration.  Near ties are decided on the spot:
// this is the form for the slow paths and the experimental kernels, on a stack that cannot overflow.
template <class STK, bool COUNT>
PRT_D HitRec trace_ray(const DevScene & sc, f3 o, f3 d, int kind, float pad, const STK & stk, TraceStats & st) {
    TravRay r;
    trav_init(r, o, d, kind, pad, stk);
    for (;;) {
        while (r.node >= 0) trav_node_step<STK, COUNT>(sc, r, stk, st, pad);
        if (trav_done(r.node)) break;
        if (trav_leaf<STK, COUNT>(sc, r, stk, st)) return r.best;           // any-hit ray: found its occluder
    }
    if (kind == TRACE_CLOSEST && (r.node & TRAV_FLAG_NEAR) && r.best.tri >= 0) return resolve_near_ties<STK, COUNT>(sc, o, d, pad, r.best.t, stk, st);
    return r.best;
}

}  // namespace prt
