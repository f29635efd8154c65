// dev_trace8.h - traversal of the 8-wide compressed BVH (80 B nodes: bvh_build.h Bvh8Result).  A build option (-DPRT_BVH8,
// make hip-bvh8); the default library uses the 4-wide sorted tree of dev_trace4.h - see the last paragraph.
//
// Replaces TraceRay / IntersectRaySphere / IntersectRayMesh (raytracer.cpp:32-60, 127-232); the triangle test and what has to
// be preserved about the reference's result are in dev_trace_common.h.
//
// Why 8-wide.  An 8-wide node decides about eight subtrees per memory round trip: 9.6 node steps + 2.2 leaf visits per traced
// ray on the 1M-triangle scene where the 4-wide tree (dev_trace4.h) takes 14.0 + 1.9 (tools/bvh_price.cpp; the frame's counters
// agree).  What makes it affordable:
//   * no sort.  The builder stores a node's children in ascending order of their centres along the axis on which those centres
//     spread most (2 bits per node); a ray visits the hit slots in ascending or descending slot order by the sign of its
//     direction on that axis, so "which child next" is a find-first-bit on the 8-bit hit mask, not a 5-comparator network with
//     30 selects.  (-DPRT_BVH8_OCTANT: the slot-per-octant order of Ylitie, Karras, Laine 2017, three conditional bit swaps
//     per pick; measured 1.7 % more node visits and 0.9 % more triangle tests on the frame, profiles/r03_ab_bvh8.txt.)
//   * no links.  Internal children are consecutive from child_base and the triangles of the node's leaves consecutive from
//     tri_base, both in slot order: a child's address is a base plus a population count, and the node is 5 dwordx4 loads
//     for eight children where the 4-wide node is 4 loads for four;
//   * one stack entry per NODE, not per child: (child_base, imask | unvisited hit slots << 8 | axis << 16).  The stack is as
//     deep as the tree (9 levels for a million triangles), so the 8-byte entries need no more LDS than the 4-wide tree's 24 links.
// A node's hit LEAF slots are tested first, in slot order, then its internal children in ray order (a leaf that lies behind
// an internal child is tested a little early: +0.5 triangle tests per ray).
// What it buys on MI355X (profiles/r03_ab_bvh8.txt, same box): 25 % fewer wave-level node steps for 17 % more vector
// instructions (205 per step for eight boxes against 140 for four; 15 % more triangle tests), and the node array halves (8.2 MB
// instead of 16.6 MB).  On the headline frame that was level with the 4-wide tree while the shading phase still carried its
// atomics, and is 2.3 % slower since (12.28 - 12.32 against 12.00 - 12.02 ms, same call): this kernel's vector pipe is full.
// Deep bounce trees (C5) render 1.4 % faster, the 12-triangle C2 6 %.  Hence an option, not the default.
#pragma once

#include "dev_trace_common.h"

namespace prt {

typedef int2 StackEntry;                 // (child_base, imask | unvisited hit slots << 8); the marker: (TRAV_SENTINEL | flags, 0)
enum { STACK_ENTRY_INTS = 2, BVH_NODE_BYTES = 80, STACK_LDS_CAP_DEFAULT = 13 };
// distance between nodes in the device array: 80 = packed.  (Experiment -DPRT_BVH8_STRIDE=128: one node per 128-byte cache line,
// so that no node straddles two lines - half of the packed ones do; no gain: profiles/r03_ab_bvh8.txt item 7.)
#ifndef PRT_BVH8_STRIDE
#define PRT_BVH8_STRIDE 80
#endif
enum { BVH_NODE_STRIDE = PRT_BVH8_STRIDE };
enum { TRAV_TOP_LDS_NODES = 0 };          // the top-levels-in-LDS experiment exists for the 4-wide tree only (dev_trace4.h)

// Per-lane traversal registers.  A ray can be suspended and resumed at any step boundary.
struct TravRay {
    f3 o, d;                          // origin ALREADY biased by direction * ray_bias (raytracer.cpp:163), direction
    float ix, iy, iz;                 // 1 / direction, components clamped away from 0
    float pnx, pny, pnz;              // (o +- pad) / direction for the plane the ray ENTERS through on each axis; the plane it
                                      // leaves through is 2 pad |1 / direction| further on (pad always widens the box)
    HitRec best;
    int node;                         // >= 0: the node to fetch next; < 0: no node left (the marker was popped)
    unsigned int tbase;               // pending leaves of the last node: first triangle of its first leaf slot ...
    unsigned int tbits;               // ... hit leaf slots (bits 0-7) | lmask << 8 | c0 << 16 | c1 << 24 (bvh_build.h)
    int sp, kind;                     // kind: TRACE_CLOSEST / TRACE_ANY | octant << 4
};

// Bottom-of-stack marker: word 0 of entry 0 of the lane's own stack column.  The two rare things a traversal has to report
// - a candidate within 2^-19 of the best hit (resolve_near_ties() must decide), a push that did not fit the stack - are
// recorded IN the marker, so that they cost the loop no register.
enum { TRAV_SENTINEL = (int)0x80000000, TRAV_FLAG_NEAR = 1, TRAV_FLAG_OVERFLOW = 2, TRAV_SENTINEL_LAST = (int)0x80000003 };

// Traversal stacks of 8-byte entries.  LdsStack: this lane's column of a workgroup LDS array, the two words of entry e of lane l
// at col[2 e * BLOCK + l] and col[(2 e + 1) * BLOCK + l] - two dword planes, NOT an int2 per lane: every word a lane ever
// touches lies in its own dword column (index = l mod BLOCK), as do the fields of the frame the same lane shades in the same
// LDS (kernels_wave.h WFrameLds, stride BLOCK).  The waves of a block are not phase-synchronised - one shades while another
// traverses - so a lane's stack words must never fall into another lane's column (an interleaved int2 layout does exactly
// that: the first version of this file faulted on it).  A wave's push / pop of one level is one conflict-free
// ds_write2st64_b32 / ds_read2st64_b32.  A push that does not fit is DROPPED and the marker gets TRAV_FLAG_OVERFLOW: the ray
// is traced again by a slow path on a stack that holds the whole bound.  LdsSpillStack: the same column, continued in a
// per-lane global column behind it; never overflows (slow paths only).  GlobalStack: a whole column in global memory.
template <int BLOCK>
struct LdsStack {
    int * col;
    unsigned int cap;
    PRT_D void attach(int * lds, unsigned int tid) { col = lds + tid; }
    PRT_D int * frame_col() const { return col; }      // the lane's column as plain dwords, stride BLOCK (the shading frame lives there)
    PRT_D bool put(int sp, int2 v) const {
        if ((unsigned int)sp < cap) { col[(2 * sp) * BLOCK] = v.x; col[(2 * sp + 1) * BLOCK] = v.y; return true; }
        return false;
    }
    PRT_D int2 get(int sp) const { return make_int2(col[(2 * sp) * BLOCK], col[(2 * sp + 1) * BLOCK]); }
    PRT_D void flag(int bit) const { col[0] |= bit; }
    PRT_D int marker() const { return col[0]; }
};

template <int BLOCK>
struct LdsSpillStack {
    int * col;
    unsigned int cap;
    int2 * spill;                     // WAVE-UNIFORM base of the spill area (null when cap covers the bound): entry cap + k of
    unsigned int spill_stride;        // the lane with global thread id g at spill[k * spill_stride + g]
    PRT_D void attach(int * lds, unsigned int tid) { col = lds + tid; }
    PRT_D int * frame_col() const { return col; }
    PRT_D void set_spill(int * base, unsigned int lanes) { spill = reinterpret_cast<int2 *>(base); spill_stride = lanes; }
    PRT_D size_t spill_index(int sp) const {
        return (size_t)((unsigned int)sp - cap) * spill_stride + (blockIdx.x * (unsigned int)BLOCK + threadIdx.x);
    }
    PRT_D bool put(int sp, int2 v) const {
        if ((unsigned int)sp < cap) { col[(2 * sp) * BLOCK] = v.x; col[(2 * sp + 1) * BLOCK] = v.y; }
        else spill[spill_index(sp)] = v;
        return true;
    }
    PRT_D int2 get(int sp) const {
        if ((unsigned int)sp < cap) return make_int2(col[(2 * sp) * BLOCK], col[(2 * sp + 1) * BLOCK]);
        return spill[spill_index(sp)];
    }
    PRT_D void flag(int bit) const { col[0] |= bit; }
    PRT_D int marker() const { return col[0]; }
};

struct GlobalStack {
    int2 * col;
    size_t stride;
    PRT_D void attach(int * base, size_t lane, size_t lanes) { col = reinterpret_cast<int2 *>(base) + lane; stride = lanes; }
    PRT_D bool put(int sp, int2 v) const { col[(size_t)sp * stride] = v; return true; }
    PRT_D int2 get(int sp) const { return col[(size_t)sp * stride]; }
    PRT_D void flag(int bit) const { col[0].x |= bit; }
    PRT_D int marker() const { return col[0].x; }
};

// The states of a lane's ray, as the kernels' loops ask for them
PRT_D void trav_idle(TravRay & r) { r.node = TRAV_SENTINEL; r.tbits = 0u; r.tbase = 0u; r.sp = 0; r.kind = 0; }      // no ray
PRT_D bool trav_at_leaf(const TravRay & r) { return (r.tbits & 0xFFu) != 0u; }
PRT_D bool trav_walking(const TravRay & r) { return (r.tbits & 0xFFu) == 0u && r.node >= 0; }    // wants a node step
PRT_D bool trav_done(const TravRay & r) { return (r.tbits & 0xFFu) == 0u && r.node < 0; }

// A ray's traversal registers as dwords, and its stack column copied from another lane's (dev_trace4.h has the commentary).
enum { TRAV_STATE_DWORDS = 21 };
PRT_D void trav_save_regs(const TravRay & r, float * dst, unsigned int stride) {
    const float f[TRAV_STATE_DWORDS] = { r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, r.ix, r.iy, r.iz, r.pnx, r.pny, r.pnz,
                                         r.best.t, r.best.v, r.best.w, as_f(r.best.tri), as_f(r.node), as_f((int)r.tbase), as_f((int)r.tbits),
                                         as_f(r.sp), as_f(r.kind) };
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
    r.node = as_i(f[16]); r.tbase = (unsigned int)as_i(f[17]); r.tbits = (unsigned int)as_i(f[18]); r.sp = as_i(f[19]); r.kind = as_i(f[20]);
}
template <class STK>
PRT_D void trav_copy_stack(const STK & dst, const STK & src, int sp) {
    for (int e = 0; e < sp; ++e) dst.put(e, src.get(e));
}

// After a traversal ended: what its marker says.  0 for a ray that ended on an any-hit occluder (a found occluder is final
// whatever happened before).
template <class STK> PRT_D int trav_end_flags(const TravRay & r, const STK & stk) {
    return ((r.kind & TRACE_ANY) && r.best.tri >= 0) ? 0 : (stk.marker() & 3);
}
// the hit of a closest-hit ray has company within a few ulp: the reference's visit order decides (resolve_near_ties)
template <class STK> PRT_D bool trav_wants_resolve(const TravRay & r, const STK & stk) {
    return !(r.kind & TRACE_ANY) && r.best.tri >= 0 && (stk.marker() & TRAV_FLAG_NEAR) != 0;
}
// fast kernels: the ray cannot be finished here (near tie, or a dropped push): it goes to the slow path
template <class STK> PRT_D bool trav_needs_slow_path(const TravRay & r, const STK & stk) {
    const int f = trav_end_flags(r, stk);
    return (f & TRAV_FLAG_OVERFLOW) != 0 || (!(r.kind & TRACE_ANY) && r.best.tri >= 0 && (f & TRAV_FLAG_NEAR) != 0);
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
    // direction >= 0: enters through the lo plane (seen from o + pad); else through the hi plane (seen from o - pad)
    r.pnx = (dx < 0.0f ? o.x - pad : o.x + pad) * r.ix;
    r.pny = (dy < 0.0f ? o.y - pad : o.y + pad) * r.iy;
    r.pnz = (dz < 0.0f ? o.z - pad : o.z + pad) * r.iz;
    r.best.t = 3.402823466e+38f;
    r.best.v = r.best.w = 0.0f;
    r.best.tri = -1;
    const int oct = (dx < 0.0f ? 1 : 0) | (dy < 0.0f ? 2 : 0) | (dz < 0.0f ? 4 : 0);
    r.kind = kind | oct << 4;
    stk.put(0, make_int2(TRAV_SENTINEL, 0));
    r.sp = 1;
    r.node = 0;
    r.tbase = 0u;
    r.tbits = 0u;
}

// Of the hit slots `rest` (non-zero), the one a ray of octant `oct` visits first: smallest (slot XOR oct).
// Bit j of the mask moves to bit (j XOR oct) - three conditional swaps of neighbours, pairs, halves - and the lowest set bit wins.
PRT_D unsigned int trav_pick_slot(unsigned int rest, unsigned int oct) {
    unsigned int p = rest;
    const unsigned int s1 = ((p & 0x55u) << 1) | ((p >> 1) & 0x55u);
    p = (oct & 1u) ? s1 : p;
    const unsigned int s2 = ((p & 0x33u) << 2) | ((p >> 2) & 0x33u);
    p = (oct & 2u) ? s2 : p;
    const unsigned int s4 = ((p & 0x0Fu) << 4) | ((p >> 4) & 0x0Fu);
    p = (oct & 4u) ? s4 : p;
    return (unsigned int)(__ffs((int)p) - 1) ^ oct;
}

// One 8-wide node: fetch 80 B, dequantise + slab-test eight child boxes, note the hit leaf slots for the leaf phase, descend
// into the first hit internal child in octant order and leave the others, as one entry, on the stack - or, if no internal
// child was hit, take the next child from the entry on top of the stack.
//   plane = origin + q * 2^e  =>  t = (plane - o -+ pad) / d = q * (2^e / d) + (origin / d - (o +- pad) / d)
// The slab test may use FMA: it only has to be conservative, and the boxes are widened by `pad`.
template <class STK, bool COUNT>
PRT_D void trav_node_step(const DevScene & sc, TravRay & r, const STK & stk, TraceStats & st, float pad, const uint4 * /*top: dev_trace4.h's PRT_TOP_LDS experiment*/ = nullptr) {
    // 32-bit byte offset from the (scalar) array base (upload caps the scene at 2^26 triangles)
    const uint4 * np = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(sc.nodes) + (unsigned int)r.node * (unsigned int)BVH_NODE_STRIDE);
    const uint4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3], q4 = np[4];
    if (COUNT) { st.nodes++; if (first_active_lane()) st.wnodes++; if ((unsigned int)r.sp > st.max_sp) st.max_sp = (unsigned int)r.sp; }
#if defined(PRT_PROBE_EXTRA_VALU) && defined(__HIP_DEVICE_COMPILE__)
    // sensitivity probe (tools/ab_probe.sh): PRT_PROBE_EXTRA_VALU more vector ALU instructions per node step, on a value it needs
    { unsigned int x = q0.w;
#pragma unroll
      for (int i = 0; i < PRT_PROBE_EXTRA_VALU; ++i) asm volatile("v_mov_b32 %0, %0" : "+v"(x));
      const_cast<uint4 &>(q0).w = x; }
#endif
    const float kx = __uint_as_float(q0.w & 0x7F800000u) * r.ix;
    const float ky = __uint_as_float(q1.z & 0x7F800000u) * r.iy;
    const float kz = __uint_as_float(q1.w & 0x7F800000u) * r.iz;
    // entry / exit parameter of the node origin on each axis; the exit side is 2 pad |1 / d| beyond the entry side's offset
    const float pad2 = pad + pad;
    const float cnx = __builtin_fmaf(__uint_as_float(q0.x), r.ix, -r.pnx), cfx = __builtin_fmaf(pad2, fabsf(r.ix), cnx);
    const float cny = __builtin_fmaf(__uint_as_float(q0.y), r.iy, -r.pny), cfy = __builtin_fmaf(pad2, fabsf(r.iy), cny);
    const float cnz = __builtin_fmaf(__uint_as_float(q0.z), r.iz, -r.pnz), cfz = __builtin_fmaf(pad2, fabsf(r.iz), cnz);
    // the ray's direction signs pick, per axis, which quantised plane set (lo or hi bytes) is the entry side - no per-plane
    // min / max, and an empty slot (lo = 255 > hi = 0 on every axis) can never satisfy entry <= exit
    const bool sx = r.ix < 0.0f, sy = r.iy < 0.0f, sz = r.iz < 0.0f;
    const unsigned int qnx[2] = { sx ? q3.z : q2.x, sx ? q3.w : q2.y }, qfx[2] = { sx ? q2.x : q3.z, sx ? q2.y : q3.w };
    const unsigned int qny[2] = { sy ? q4.x : q2.z, sy ? q4.y : q2.w }, qfy[2] = { sy ? q2.z : q4.x, sy ? q2.w : q4.y };
    const unsigned int qnz[2] = { sz ? q4.z : q3.x, sz ? q4.w : q3.y }, qfz[2] = { sz ? q3.x : q4.z, sz ? q3.y : q4.w };
    unsigned int m = 0u;
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        const int h = j >> 2, k = j & 3;
        const float nx = __builtin_fmaf((float)((qnx[h] >> (8 * k)) & 0xFFu), kx, cnx);
        const float fx = __builtin_fmaf((float)((qfx[h] >> (8 * k)) & 0xFFu), kx, cfx);
        const float ny = __builtin_fmaf((float)((qny[h] >> (8 * k)) & 0xFFu), ky, cny);
        const float fy = __builtin_fmaf((float)((qfy[h] >> (8 * k)) & 0xFFu), ky, cfy);
        const float nz = __builtin_fmaf((float)((qnz[h] >> (8 * k)) & 0xFFu), kz, cnz);
        const float fz = __builtin_fmaf((float)((qfz[h] >> (8 * k)) & 0xFFu), kz, cfz);
        const float tmin = fmaxf(fmaxf(fmaxf(nx, ny), nz), 0.0f);
        const float tmax = fminf(fminf(fminf(fx, fy), fz), r.best.t);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PRT_BVH8_NO_ADDC)
        // m = 2 m + (tmin <= tmax): the compare's lane mask goes straight into an add-with-carry, one instruction per child
        // where a select and an or would be two
        asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(tmin), "v"(tmax) : "vcc");
#else
        m = (m << 1) | (tmin <= tmax ? 1u : 0u);
#endif
    }
    const unsigned int imask = q0.w & 0xFFu;
    // the node's hit leaves wait for the leaf phase
    r.tbase = q1.y;
    r.tbits = (m & ~imask) | (q0.w & 0xFF00u) | (q1.z << 16);
    // the next node: from this node's hit internal children, else from the group on top of the stack
    const unsigned int mi = m & imask;
#if defined(PRT_BVH8_OCTANT)
    unsigned int gbase = q1.x, gbits = imask | mi << 8;
#else
    unsigned int gbase = q1.x, gbits = imask | mi << 8 | (q1.w & 3u) << 16;
#endif
    int at = r.sp;                                      // where the group's remainder goes: a new entry, or back where it came from
    if (mi == 0u) {
        if (COUNT && r.best.tri >= 0 && (m & ~imask) == 0u) st.culled++;
        at = r.sp - 1;
        const int2 e = stk.get(at);
        if (e.x <= TRAV_SENTINEL_LAST) {                // the marker: no node is left (the pending leaves still are)
            r.node = TRAV_SENTINEL;
            r.sp = at;
            return;
        }
        gbase = (unsigned int)e.x;
        gbits = (unsigned int)e.y;
    }
#if defined(PRT_BVH8_OCTANT)
    const unsigned int s = trav_pick_slot(gbits >> 8, ((unsigned int)r.kind >> 4) & 7u);
#else
    // the slots of a node are sorted along its ordering axis (bits 16-17 of the group word): a ray takes the hit ones in
    // ascending or descending slot order by its direction sign on that axis
    const unsigned int rest8 = (gbits >> 8) & 0xFFu;
    const bool backwards = (((unsigned int)r.kind >> 4) >> ((gbits >> 16) & 3u)) & 1u;
    const unsigned int s = backwards ? 31u - (unsigned int)__clz((int)rest8) : (unsigned int)(__ffs((int)rest8) - 1);
#endif
    r.node = (int)(gbase + (unsigned int)__popc(gbits & ((1u << s) - 1u) & 0xFFu));
    gbits &= ~(0x100u << s);
    if ((gbits >> 8) & 0xFFu) {
        if (stk.put(at, make_int2((int)gbase, (int)gbits))) r.sp = at + 1;
        else { stk.flag(TRAV_FLAG_OVERFLOW); r.sp = at; }
    } else {
        r.sp = at;
    }
}

// First triangle and triangle count of the lowest pending leaf slot; trav_leaf_consume() retires that slot.
PRT_D void trav_leaf_range(const TravRay & r, unsigned int & first, unsigned int & count) {
    const unsigned int s = (unsigned int)(__ffs((int)(r.tbits & 0xFFu)) - 1);
    const unsigned int below = (1u << s) - 1u;
    // lmask, c0 and c1 each add a triangle for every leaf slot below s that has their bit set (c1 adds two)
    const unsigned int ones = (unsigned int)__popc(r.tbits & (below << 8 | below << 16));
    const unsigned int twos = (unsigned int)__popc(r.tbits & (below << 24));
    first = r.tbase + ones + 2u * twos;
    count = 1u + ((r.tbits >> (16u + s)) & 1u) + 2u * ((r.tbits >> (24u + s)) & 1u);
}
PRT_D void trav_leaf_consume(TravRay & r) { r.tbits &= r.tbits - 1u; }     // the lowest set bit is a pending slot (bits 0-7)

// The lowest pending leaf slot: test its triangles.  Returns true when an any-hit ray found its hit.
template <class STK, bool COUNT>
PRT_D bool trav_leaf(const DevScene & sc, TravRay & r, const STK & stk, TraceStats & st) {
    const f3 qp = r.o - (r.o + r.d);                 // raytracer.cpp:88-89, not bitwise -d
    unsigned int first, count;
    trav_leaf_range(r, first, count);
    if (COUNT) { if (first_active_lane()) st.wleaves++; }
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
        if (near) stk.flag(TRAV_FLAG_NEAR);
        if (hit) {
            r.best.t = t;
            r.best.v = v;
            r.best.w = w;
            r.best.tri = (int)ti;
            if (r.kind & TRACE_ANY) return true;
        }
    }
    trav_leaf_consume(r);
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
// counted (DevScene::near_tie_unresolved: the render call then fails - it has never been seen to happen).
// The reference also skips a whole GROUP whose bounding sphere it enters later than its best hit so far
// (raytracer.cpp:176-181): RefSphereWalk (dev_trace_common.h) replays that on the uploaded sphere tree.
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
                while (trav_walking(r)) trav_node_step<STK, COUNT>(sc, r, stk, st, pad);
                if (trav_done(r)) break;
                unsigned int first, count;
                trav_leaf_range(r, first, count);
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
                trav_leaf_consume(r);
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

// Whole-ray traversal, "while-while" (Aila & Laine): every lane first walks internal nodes until it holds a leaf (or runs out
// of work), and only then does the wave run the triangle code.  Near ties are decided on the spot: this is the form for the
// slow paths and the experimental kernels, on a stack that cannot overflow.
template <class STK, bool COUNT>
PRT_D HitRec trace_ray(const DevScene & sc, f3 o, f3 d, int kind, float pad, const STK & stk, TraceStats & st) {
    TravRay r;
    trav_init(r, o, d, kind, pad, stk);
    for (;;) {
        while (trav_walking(r)) trav_node_step<STK, COUNT>(sc, r, stk, st, pad);
        if (trav_done(r)) break;
        if (trav_leaf<STK, COUNT>(sc, r, stk, st)) return r.best;           // any-hit ray: found its occluder
    }
    if (trav_wants_resolve(r, stk)) return resolve_near_ties<STK, COUNT>(sc, o, d, pad, r.best.t, stk, st);
    return r.best;
}

}  // namespace prt
