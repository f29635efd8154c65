// dev_trace.h - TraceRay for the device: BVH2 traversal + the reference's triangle test.
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
// Traversal stack: per-lane column of a workgroup LDS array (entry e of lane l at stack[e*BLOCK + l],
// so a wave's push/pop of one level is one conflict-free ds_write/ds_read_b32).
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
};

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
    float plx, ply, plz;              // (o + pad) / direction: entry parameter offset of the widened lo planes
    float phx, phy, phz;              // (o - pad) / direction: same for the hi planes
    HitRec best;
    unsigned int best_rank;
    int node, sp, kind;
};

enum { TRAV_SENTINEL = (int)0x80000000 };   // bottom-of-stack marker; never a valid leaf link (first_tri < 2^29)

template <int BLOCK>
PRT_D void trav_init(TravRay & r, f3 o, f3 d, int kind, float pad, int * stack) {
    r.o = o;
    r.d = d;
    // direction components are clamped away from 0 so no inf/NaN enters the box test
    const float tiny = 1e-30f;
    float dx = fabsf(d.x) < tiny ? (d.x < 0.0f ? -tiny : tiny) : d.x;
    float dy = fabsf(d.y) < tiny ? (d.y < 0.0f ? -tiny : tiny) : d.y;
    float dz = fabsf(d.z) < tiny ? (d.z < 0.0f ? -tiny : tiny) : d.z;
    r.ix = 1.0f / dx; r.iy = 1.0f / dy; r.iz = 1.0f / dz;
    r.plx = (o.x + pad) * r.ix; r.ply = (o.y + pad) * r.iy; r.plz = (o.z + pad) * r.iz;
    r.phx = (o.x - pad) * r.ix; r.phy = (o.y - pad) * r.iy; r.phz = (o.z - pad) * r.iz;
    r.best.t = 3.402823466e+38f;
    r.best.v = r.best.w = 0.0f;
    r.best.tri = -1;
    r.best_rank = 0xFFFFFFFFu;
    r.kind = kind;
    stack[0] = TRAV_SENTINEL;
    r.sp = 1;
    r.node = 0;
}

// One internal node: fetch 64 B, test both child boxes, descend into the nearer hit child, push the other.
// The slab test is the one place that uses FMA (plane * 1/d - origin/d in one rounding): it only has to be
// conservative, and the boxes are widened by `pad` >> its rounding error.  6 FMA + 10 min/max per box.
template <int BLOCK, bool COUNT>
PRT_D void trav_node_step(const DevScene & sc, TravRay & r, int * stack, TraceStats & st) {
    const float4 * np = sc.nodes + 4 * (size_t)r.node;
    const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
    if (COUNT) st.nodes++;
    const float ax = __builtin_fmaf(n0.x, r.ix, -r.plx), bx = __builtin_fmaf(n0.y, r.ix, -r.phx);
    const float ay = __builtin_fmaf(n0.z, r.iy, -r.ply), by = __builtin_fmaf(n0.w, r.iy, -r.phy);
    const float az = __builtin_fmaf(n2.x, r.iz, -r.plz), bz = __builtin_fmaf(n2.y, r.iz, -r.phz);
    const float tmin0 = fmaxf(fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz)), 0.0f);
    const float tmax0 = fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)), r.best.t);
    const float cx = __builtin_fmaf(n1.x, r.ix, -r.plx), ex = __builtin_fmaf(n1.y, r.ix, -r.phx);
    const float cy = __builtin_fmaf(n1.z, r.iy, -r.ply), ey = __builtin_fmaf(n1.w, r.iy, -r.phy);
    const float cz = __builtin_fmaf(n2.z, r.iz, -r.plz), ez = __builtin_fmaf(n2.w, r.iz, -r.phz);
    const float tmin1 = fmaxf(fmaxf(fmaxf(fminf(cx, ex), fminf(cy, ey)), fminf(cz, ez)), 0.0f);
    const float tmax1 = fminf(fminf(fminf(fmaxf(cx, ex), fmaxf(cy, ey)), fmaxf(cz, ez)), r.best.t);
    const bool h0 = tmin0 <= tmax0, h1 = tmin1 <= tmax1;
    int l0 = as_i(n3.x), l1 = as_i(n3.y);
    if (h0 && h1) {
        if (tmin1 < tmin0) { int tmp = l0; l0 = l1; l1 = tmp; }
        stack[r.sp * BLOCK] = l1;
        r.sp++;
        r.node = l0;
    } else if (h0) {
        r.node = l0;
    } else if (h1) {
        r.node = l1;
    } else {
        r.sp--;
        r.node = stack[r.sp * BLOCK];
    }
}

// The leaf in r.node: test its triangles, then pop.  Returns true when an any-hit ray found its hit.
template <int BLOCK, bool COUNT>
PRT_D bool trav_leaf(const DevScene & sc, TravRay & r, int * stack, TraceStats & st) {
    const f3 qp = r.o - (r.o + r.d);                 // raytracer.cpp:88-89, not bitwise -d
    const unsigned int leaf = (unsigned int)~r.node;
    const unsigned int first = leaf >> 2, count = (leaf & 3u) + 1u;
    for (unsigned int i = 0; i < count; ++i) {
        const unsigned int ti = first + i;
        const float4 * tp = sc.tris + 3 * (size_t)ti;
        const float4 r0 = tp[0], r1 = tp[1], r2 = tp[2];
        if (COUNT) st.tris++;
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
    r.node = stack[r.sp * BLOCK];
    return false;
}

// Whole-ray traversal, "while-while" (Aila & Laine): every lane first walks internal nodes until it holds
// a leaf (or runs out of work), and only then does the wave run the triangle code.  With 64 lanes a fused
// node-or-leaf loop would execute the (4x longer) leaf body in almost every iteration.
template <int BLOCK, bool COUNT>
PRT_D HitRec trace_ray(const DevScene & sc, f3 o, f3 d, int kind, float pad, int * stack, TraceStats & st) {
    TravRay r;
    trav_init<BLOCK>(r, o, d, kind, pad, stack);
    for (;;) {
        while (r.node >= 0) trav_node_step<BLOCK, COUNT>(sc, r, stack, st);
        if (r.node == TRAV_SENTINEL) break;
        if (trav_leaf<BLOCK, COUNT>(sc, r, stack, st)) break;
    }
    return r.best;
}

}  // namespace prt
