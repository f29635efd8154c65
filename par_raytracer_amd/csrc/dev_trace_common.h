// dev_trace_common.h - TraceRay for the device, the parts every traversal shares: hit record, counters, the reference's
// triangle test.  (The traversal itself: dev_trace4.h - 4-wide BVH, children sorted per step - or dev_trace8.h - 8-wide,
// octant-ordered; dev_trace.h picks one.)
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
//   * hits whose t agree to within a few ulp (coplanar patches, decals, doubled faces, shared edges hit head-on) are
//     where the reference's VISIT ORDER is observable: it runs a sequential filter - early reject `t > best*d`
//     (raytracer.cpp:104), then strict `<` (:149, :220) - over the triangles in sphere-tree order (:208-209), and the two
//     tests can disagree by an ulp.  The traversal here finds the exact minimum of t in any order and FLAGS a ray whose
//     minimum has company within 2^-19 of it; a flagged ray is then decided by resolve_near_ties(): the candidates near
//     the minimum are enumerated in the reference's visit order (tri_rank) and put through the reference's filter, form
//     for form.  Scenes without such geometry never take that path.
//
// Traversal stack: per-lane LDS column that continues in a per-lane global column when it is full (see LdsStack).
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
    unsigned int wrays;                              // k_pool: lanes that held a ray, summed over the wave-level node steps
    unsigned int max_sp, culled;                     // deepest stack use; popped nodes whose entry distance was already beyond the hit
};

PRT_D bool first_active_lane() {
    const unsigned long long m = __ballot(true);
    return (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))) == (__ffsll((long long)m) - 1);
}

PRT_D float as_f(int v) { return __int_as_float(v); }
PRT_D int as_i(float v) { return __float_as_int(v); }

// The geometric part of IntersectRayTriangle (raytracer.cpp:82-125) on pre-differenced data: everything except the two
// comparisons with the caller's best hit.  True when the ray's line meets the front side of the triangle at t >= 0;
// then t / dd is the hit parameter and v / dd, w / dd the barycentrics (dd > 0).
PRT_D bool tri_geom(f3 o, f3 qp, f3 a, f3 ab, f3 ac, f3 n, float & t, float & dd, float & v, float & w) {
    dd = dot3(qp, n);
    if (dd <= 0.0f) return false;
    const f3 ap = o - a;
    t = dot3(ap, n);
    if (t < 0.0f) return false;
    const f3 e = cross3(qp, ap);
    v = dot3(ac, e);
    if (v < 0.0f || v > dd) return false;
    w = -dot3(ab, e);
    if (w < 0.0f || (v + w) > dd) return false;
    return true;
}

// The reference's test as IntersectRayMesh applies it (raytracer.cpp:104, 149): true, with the hit in (out_t, out_v,
// out_w), when IntersectRayTriangle returns true for `best_t` AND the caller keeps it (strict <).  The early reject and
// the final comparison are different float expressions of the same inequality; for a hit within an ulp or two of best_t
// they can disagree, which is what makes the reference's result depend on its visit order.  Used by the known-answer
// tests and by resolve_near_ties(), never by the hot loop.
PRT_D bool tri_test_ref(f3 o, f3 qp, f3 a, f3 ab, f3 ac, f3 n, float best_t, float & out_t, float & out_v, float & out_w) {
    float t, dd, v, w;
    if (!tri_geom(o, qp, a, ab, ac, n, t, dd, v, w)) return false;
    if (t > best_t * dd) return false;
    const float ood = 1.0f / dd;
    const float th = t * ood;
    if (!(th < best_t)) return false;
    out_t = th; out_v = v * ood; out_w = w * ood;
    return true;
}

// Relaxed early reject and the width of "near": a candidate is only rejected early when it is beyond best * (1 + 2^-17),
// so no hit below the current best is ever lost to rounding and the traversal's result is the exact minimum of t; `near`
// is raised for a candidate within 2^-19 (16 ulp) of the current best on either side.  The BVH cannot hide such a
// candidate: every box is widened by pad = 2^-16 x (largest coordinate), and t < 4 x (largest coordinate).
#define PRT_TIE_REJECT 1.00000762939453125f      /* 1 + 2^-17 */
#define PRT_TIE_NEAR   1.0000019073486328125f    /* 1 + 2^-19 */

// Hot-loop triangle test.  Returns true and the hit when the candidate is strictly closer than best_t.
PRT_D bool tri_test(f3 o, f3 d, f3 qp, f3 a, f3 ab, f3 ac, f3 n, float best_t, float & out_t, float & out_v, float & out_w,
                    bool & near) {
    near = false;
    const float dd = dot3(qp, n);
    if (dd <= 0.0f) return false;
    const f3 ap = o - a;
    const float t = dot3(ap, n);
    if (t < 0.0f) return false;
    if (t > (best_t * dd) * PRT_TIE_REJECT) return false;
    const f3 e = cross3(qp, ap);
    const float v = dot3(ac, e);
    if (v < 0.0f || v > dd) return false;
    const float w = -dot3(ab, e);
    if (w < 0.0f || (v + w) > dd) return false;
    const float ood = 1.0f / dd;
    const float th = t * ood;
    near = th * PRT_TIE_NEAR >= best_t;             // (th <= best_t * PRT_TIE_REJECT is already known)
    if (th < best_t) {
        out_t = th;
        out_v = v * ood;
        out_w = w * ood;
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------------------------------
// The reference's bounding-sphere tree, as far as it is OBSERVABLE: TraceRay (raytracer.cpp:159-232) walks it depth first,
// c1 before c0, and never scans a group that lies below a sphere the ray misses or ENTERS LATER THAN ITS BEST HIT SO FAR
// (:176-181).  For a closest hit with room around it that changes nothing (a hit lies inside its group's sphere, which the ray
// therefore enters no later).  When several hits are within an ulp or two of each other - resolve_near_ties() - a candidate that
// sits on the very surface of its group's sphere can be passed over by the reference because the sphere's own rounded entry
// distance exceeds the best hit by an ulp.  RefSphereWalk replays exactly that: candidates come in the reference's visit order
// (tri_rank ascending); every sphere on the way to a candidate's group is tested the first time the walk meets it, against
// the replayed best hit of that moment, with IntersectRaySphere's own float expressions (raytracer.cpp:32-60); a sphere that
// fails takes its whole rank range out.  Spheres are visited in ascending `pre` (their position in the reference's pop order),
// so "met before" is one comparison, and a failed sphere is one range - no storage.
struct RefSphereWalk {
    int mark;                      // the largest `pre` visited so far
    unsigned int skip_until;       // candidates of rank below this lie under a sphere that failed
    PRT_D void reset() { mark = -1; skip_until = 0u; }
    // Would the reference offer the triangle of visit rank `rank` to its filter, with `best` its best hit distance so far?
    PRT_D bool offers(const DevScene & sc, f3 o, f3 d, unsigned int rank, float best) {
        if (!sc.ref_spheres) return true;
        if (rank < skip_until) return false;
        unsigned int si = 0u, hi = sc.tri_count;
        for (;;) {
            const float4 a = sc.ref_spheres[2u * si];
            const uint4 b = reinterpret_cast<const uint4 *>(sc.ref_spheres)[2u * si + 1u];       // c0, c1, split, pre
            if ((int)b.w > mark) {
                mark = (int)b.w;
                // IntersectRaySphere, raytracer.cpp:32-60
                const f3 m = o - mk3(a.x, a.y, a.z);
                const float bb = dot3(m, d);
                const float cc = dot3(m, m) - a.w * a.w;
                bool hit = !(cc > 0.0f && bb > 0.0f);
                float t = 0.0f;
                if (hit) {
                    const float disc = bb * bb - cc;
                    hit = !(disc < 0.0f);
                    if (hit) { t = -bb - sqrtf(disc); if (t < 0.0f) t = 0.0f; }
                }
                if (!hit || t > best) { skip_until = hi; return false; }
            }
            if (!(b.x && b.y)) return true;                       // a leaf: its group is scanned (raytracer.cpp:208-222)
            if (rank < b.z) { si = b.y; hi = b.z; }               // c1's triangles come first in the visit order
            else si = b.x;
        }
    }
};

}  // namespace prt
