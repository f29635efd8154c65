// dev_trace.h - TraceRay for the device: quantised 4-wide BVH traversal + the reference's triangle test.
//
// Replaces TraceRay / IntersectRaySphere / IntersectRayMesh / IntersectRayTriangle
// (raytracer.cpp:32-60, 82-232).  What must be preserved is the RESULT of the reference's traversal:
// the closest hit over all front-facing triangles under ITS float arithmetic, first-visited wins on
// equal t.  So
//   * the triangle test is the reference's, operation for operation (no FMA: -ffp-contract=off), on
//     host-precomputed ab / ac / n which are the same bits the CPU computes per call;
//   * the culling structure only has to be conservative: every BVH box is widened by `pad` (world units,
//     2^-16 of the scene + camera extent, >= 100x the rounding of the slab arithmetic and of the triangle
//     test's acceptance region) by shifting the ray origin per plane side, so a slab test with plain
//     float rounding can never cull a triangle the reference would accept;
//   * ties in t are broken by the triangle's rank in the reference's visit order.
// The early reject `t > best*d` (raytracer.cpp:104) is kept in the same form; like in the reference it
// can differ by an ulp from the final `t*ood < best` test for two nearly coincident hits, the one place
// where visit order is observable (SURVEY.md §7.2 "Tie-breaking").
//
// Traversal stack: per-lane LDS column, with a flag-and-retrace fallback on a global column (see LdsStack).
#pragma once

#include "dev_scene.h"

namespace prt {

struct HitRec {
    float t;        // distance along the (biased-origin) ray; FLT_MAX when nothing was hit
    float v, w;     // bw.y, bw.z (raytracer.cpp:118-119)
    int tri;        // leaf-order triangle index, -1 = miss
};

enum { TRACE_CLOSEST = 0, TRACE_ANY = 1 };

struct TraceStats {
    unsigned int nodes, tris;
    unsigned int wnodes, wleaves, wtris, wrefills;   // counted by the first active lane only (wave-level steps)
    unsigned int max_sp, culled;                     // deepest stack use; popped nodes whose entry distance was already beyond the hit
};

PRT_D bool first_active_lane() {
    const unsigned long long m = __ballot(true);
    return (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))) == (__ffsll((long long)m) - 1);
}

PRT_D float as_f(int v) { return __int_as_float(v); }
PRT_D int as_i(float v) { return __float_as_int(v); }

// Reference triangle test on pre-differenced data.  Returns true and updates (best_t, v, w) when the
// reference's IntersectRayTriangle would return true AND IntersectRayMesh would keep it (strict <).
// `equal_t` reports a bit-equal t so the caller can consult the visit rank.
PRT_D bool tri_test(f3 o, f3 d, f3 qp, f3 a, f3 ab, f3 ac, f3 n, float best_t, float & out_t, float & out_v, float & out_w,
                    bool & equal_t) {
    equal_t = false;
    float dd = dot3(qp, n);
    if (dd <= 0.0f) return false;
    f3 ap = o - a;
    float t = dot3(ap, n);
    if (t < 0.0f) return false;
    if (t > best_t * dd) return false;
    f3 e = cross3(qp, ap);
    float v = dot3(ac, e);
    if (v < 0.0f || v > dd) return false;
    float w = -dot3(ab, e);
    if (w < 0.0f || (v + w) > dd) return false;
    float ood = 1.0f / dd;
    float th = t * ood;
    if (th < best_t) {
        out_t = th;
        out_v = v * ood;
        out_w = w * ood;
        return true;
    }
    equal_t = (th == best_t);
    if (equal_t) {
        out_t = th;
        out_v = v * ood;
        out_w = w * ood;
    }
    return false;
}

// Per-lane traversal registers.  A ray can be suspended and resumed at any node boundary (the persistent
// kernel does so when it leaves the traversal loop to refill idle lanes).
struct TravRay {
    f3 o, d;                          // origin ALREADY biased by direction * ray_bias (raytracer.cpp:163), direction
    float ix, iy, iz;                 // 1 / direction, components clamped away from 0
    float pnx, pny, pnz;              // (o +- pad) / direction for the plane the ray ENTERS through on each axis
    float pfx, pfy, pfz;              // ... and for the plane it LEAVES through (pad always widens the box)
    HitRec best;
    unsigned int best_rank;
    int node, sp, kind;
    bool overflow;                    // a push was dropped: the result is not trustworthy, re-trace on the slow stack
};

// Traversal stacks.  The fast one is this lane's column of a workgroup LDS array (entry e of lane l at
// col[e*BLOCK + l]: a wave's push/pop of one level is one conflict-free ds_write/ds_read_b32).  Its height
// bounds occupancy, so it is sized for what rays really use (<= 24 entries; the deepest ever observed on the 1M
// triangle scene is 16) and not for the worst case (3 pushes per 4-wide level).  A push that does not fit is
// DROPPED and the ray is flagged; a flagged ray is re-traced from scratch on the slow stack, a per-lane column in
// global memory that holds the full bound.  The hot loop therefore carries one compare per push and no spill code.
template <int BLOCK>
struct LdsStack {
    int * col;
    unsigned int cap;
    PRT_D bool push(int sp, int v) const {
        if ((unsigned int)sp < cap) { col[sp * BLOCK] = v; return true; }
        return false;
    }
    PRT_D int pop(int sp) const { return col[sp * BLOCK]; }
};

struct GlobalStack {
    int * col;
    size_t stride;
    PRT_D bool push(int sp, int v) const { col[(size_t)sp * stride] = v; return true; }
    PRT_D int pop(int sp) const { return col[(size_t)sp * stride]; }
};

enum { TRAV_SENTINEL = (int)0x80000000 };   // bottom-of-stack marker; never a valid leaf link (first_tri < 2^29)

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
    r.pnx = (dx < 0.0f ? o.x - pad : o.x + pad) * r.ix; r.pfx = (dx < 0.0f ? o.x + pad : o.x - pad) * r.ix;
    r.pny = (dy < 0.0f ? o.y - pad : o.y + pad) * r.iy; r.pfy = (dy < 0.0f ? o.y + pad : o.y - pad) * r.iy;
    r.pnz = (dz < 0.0f ? o.z - pad : o.z + pad) * r.iz; r.pfz = (dz < 0.0f ? o.z + pad : o.z - pad) * r.iz;
    r.best.t = 3.402823466e+38f;
    r.best.v = r.best.w = 0.0f;
    r.best.tri = -1;
    r.best_rank = 0xFFFFFFFFu;
    r.kind = kind;
    stk.push(0, TRAV_SENTINEL);
    r.sp = 1;
    r.overflow = false;
    r.node = 0;
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
template <class STK, bool COUNT>
PRT_D void trav_node_step(const DevScene & sc, TravRay & r, const STK & stk, TraceStats & st) {
    const uint4 * np = reinterpret_cast<const uint4 *>(sc.nodes) + 4 * (size_t)r.node;
    const uint4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
    if (COUNT) { st.nodes++; if (first_active_lane()) st.wnodes++; if ((unsigned int)r.sp > st.max_sp) st.max_sp = (unsigned int)r.sp; }
    const float kx = __uint_as_float(w0.w) * r.ix;
    const float ky = __uint_as_float(w3.z) * r.iy;
    const float kz = __uint_as_float(w3.w) * r.iz;
    const float ox = __uint_as_float(w0.x), oy = __uint_as_float(w0.y), oz = __uint_as_float(w0.z);
    // entry / exit parameter of the node origin on each axis; the ray's direction signs pick, per axis, which
    // quantised plane set (lo or hi bytes) is the entry side - no per-plane min/max, and an empty child slot
    // (lo = 255 > hi = 0 on every axis) can never satisfy entry <= exit.
    const float cnx = __builtin_fmaf(ox, r.ix, -r.pnx), cfx = __builtin_fmaf(ox, r.ix, -r.pfx);
    const float cny = __builtin_fmaf(oy, r.iy, -r.pny), cfy = __builtin_fmaf(oy, r.iy, -r.pfy);
    const float cnz = __builtin_fmaf(oz, r.iz, -r.pnz), cfz = __builtin_fmaf(oz, r.iz, -r.pfz);
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
        if (key[3] < inf) { if (stk.push(r.sp, link[3])) r.sp++; else r.overflow = true; }
        if (key[2] < inf) { if (stk.push(r.sp, link[2])) r.sp++; else r.overflow = true; }
        if (key[1] < inf) { if (stk.push(r.sp, link[1])) r.sp++; else r.overflow = true; }
        r.node = link[0];
    } else {
        if (COUNT && r.best.tri >= 0) st.culled++;
        r.sp--;
        r.node = stk.pop(r.sp);
    }
}

// The leaf in r.node: test its triangles, then pop.  Returns true when an any-hit ray found its hit.
template <class STK, bool COUNT>
PRT_D bool trav_leaf(const DevScene & sc, TravRay & r, const STK & stk, TraceStats & st) {
    const f3 qp = r.o - (r.o + r.d);                 // raytracer.cpp:88-89, not bitwise -d
    const unsigned int leaf = (unsigned int)~r.node;
    const unsigned int first = leaf >> 2, count = (leaf & 3u) + 1u;
    if (COUNT) { if (first_active_lane()) st.wleaves++; }
    for (unsigned int i = 0; i < count; ++i) {
        const unsigned int ti = first + i;
        const float4 * tp = sc.tris + 3 * (size_t)ti;
        const float4 r0 = tp[0], r1 = tp[1], r2 = tp[2];
        if (COUNT) { st.tris++; if (first_active_lane()) st.wtris++; }
        float t, v, w;
        bool eq;
        bool hit = tri_test(r.o, r.d, qp, mk3(r0.x, r0.y, r0.z), mk3(r0.w, r1.x, r1.y), mk3(r1.z, r1.w, r2.x),
                            mk3(r2.y, r2.z, r2.w), r.best.t, t, v, w, eq);
        if (eq && r.best.tri >= 0 && (int)ti != r.best.tri) {
            // bit-equal t: the reference keeps whichever it visited first
            unsigned int rk = sc.tri_rank[ti];
            if (r.best_rank == 0xFFFFFFFFu) r.best_rank = sc.tri_rank[r.best.tri];
            hit = rk < r.best_rank;
            if (hit) r.best_rank = rk;
        } else if (hit) {
            r.best_rank = 0xFFFFFFFFu;
        }
        if (hit) {
            r.best.t = t;
            r.best.v = v;
            r.best.w = w;
            r.best.tri = (int)ti;
            if (r.kind == TRACE_ANY) return true;
        }
    }
    r.sp--;
    r.node = stk.pop(r.sp);
    return false;
}

// Whole-ray traversal, "while-while" (Aila & Laine): every lane first walks internal nodes until it holds
// a leaf (or runs out of work), and only then does the wave run the triangle code.  With 64 lanes a fused
// node-or-leaf loop would execute the (4x longer) leaf body in almost every iteration.
template <class STK, bool COUNT>
PRT_D HitRec trace_ray_on(const DevScene & sc, f3 o, f3 d, int kind, float pad, const STK & stk, TraceStats & st, bool & overflow) {
    TravRay r;
    trav_init(r, o, d, kind, pad, stk);
    for (;;) {
        while (r.node >= 0) trav_node_step<STK, COUNT>(sc, r, stk, st);
        if (r.node == TRAV_SENTINEL) break;
        if (trav_leaf<STK, COUNT>(sc, r, stk, st)) break;
    }
    overflow = r.overflow;
    return r.best;
}

// Fast stack first; the (never observed) overflow case re-traces on the slow stack.
template <int BLOCK, bool COUNT>
PRT_D HitRec trace_ray(const DevScene & sc, f3 o, f3 d, int kind, float pad, const LdsStack<BLOCK> & fast, const GlobalStack & slow,
                       TraceStats & st) {
    bool overflow;
    HitRec h = trace_ray_on<LdsStack<BLOCK>, COUNT>(sc, o, d, kind, pad, fast, st, overflow);
    if (overflow) h = trace_ray_on<GlobalStack, COUNT>(sc, o, d, kind, pad, slow, st, overflow);
    return h;
}

}  // namespace prt
