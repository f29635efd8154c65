// sphere_tree.cpp - BuildHierarchy: the reference's binary bounding-sphere tree over OBJ groups.
//
// Host-side, one-off preprocessing (bsphere.cpp:8-444); NOT part of the accelerated path - the HIP
// kernels traverse their own per-triangle BVH.  It is kept, bit for bit, for two reasons:
//   1. the CPU oracle must traverse the same tree as the reference so its counters and its timing are a
//      fair stand-in for "the reference CPU path" (SURVEY.md §2, §8d);
//   2. the order in which that tree's leaves are visited (c1 before c0, raytracer.cpp:208-209) is the
//      reference's tie-break between equal-t hits, which the HIP path reproduces as a per-triangle rank.
// Algorithm per group: covariance -> Jacobi eigenvectors -> extreme points along the dominant axis ->
// grow to contain all points (Ericson, "Real-Time Collision Detection" §4.3), then 16 shrink-and-regrow
// passes over a shuffled point list (iterative Ritter).  Tree: greedy agglomeration of the pair whose
// enclosing sphere has the smallest radius, O(n^3) in the group count.  Quirks preserved on purpose:
// the covariance matrix is left asymmetric (element (2,1) stays 0, bsphere.cpp:78) and the growth slack
// 1e-2 is added in double precision (bsphere.cpp:21).
#include <cfloat>
#include <cstdlib>

#include "prt_random.h"
#include "prt_scene.h"

namespace {

void GrowSphereToPoint(Sphere * s, Vector3 p) {                       // bsphere.cpp:14-26
    Vector3 pc = p - s->center;
    float sq_dist = Dot(pc, pc);
    if (sq_dist > (s->radius * s->radius)) {
        float dist = sqrtf(sq_dist);
        float new_radius = (float)((double)((s->radius + dist) * 0.5f) + 1e-2);
        float k = (new_radius - s->radius) / dist;
        s->radius = new_radius;
        s->center += pc * k;
    }
}

Matrix33 Covariance(const std::vector<Vector3> & pts) {              // bsphere.cpp:46-81
    float inv_n = 1.0f / (float)pts.size();
    Vector3 mean;
    for (size_t i = 0; i < pts.size(); ++i) mean += pts[i];
    mean *= inv_n;
    float xx = 0.0f, yy = 0.0f, zz = 0.0f, xy = 0.0f, xz = 0.0f, yz = 0.0f;
    for (size_t i = 0; i < pts.size(); ++i) {
        Vector3 p = pts[i] - mean;
        xx += p.x * p.x;
        yy += p.y * p.y;
        zz += p.z * p.z;
        xy += p.x * p.y;
        xz += p.x * p.z;
        yz += p.y * p.z;
    }
    Matrix33 m;
    m(0, 0) = xx * inv_n;
    m(1, 1) = yy * inv_n;
    m(2, 2) = zz * inv_n;
    m(0, 1) = m(1, 0) = xy * inv_n;
    m(0, 2) = m(2, 0) = xz * inv_n;
    m(1, 2) = yz * inv_n;            // (2,1) intentionally left 0, as in the reference
    return m;
}

void GivensForPair(const Matrix33 & m, u32 p, u32 q, float * c, float * s) {   // bsphere.cpp:83-103
    if (fabsf(m(p, q)) > 0.0001f) {
        float r = (m(q, q) - m(p, p)) / (2.0f * m(p, q));
        float t = (r >= 0.0f) ? 1.0f / (r + sqrtf(1.0f + r * r)) : -1.0f / (-r + sqrtf(1.0f + r * r));
        *c = 1.0f / sqrtf(1.0f + t * t);
        *s = (*c) * t;
    } else {
        *c = 1.0f;
        *s = 0.0f;
    }
}

void JacobiEigen(Matrix33 * a, Matrix33 * v) {                        // bsphere.cpp:105-156
    float prev_off = 0.0f;
    v->SetIdentity();
    for (u32 sweep = 0; sweep < 50; ++sweep) {
        u32 p = 0, q = 1;
        for (u32 i = 0; i < 3; ++i)
            for (u32 j = 0; j < 3; ++j)
                if (i != j && fabsf((*a)(i, j)) > fabsf((*a)(p, q))) { p = i; q = j; }
        float c, s;
        GivensForPair(*a, p, q, &c, &s);
        Matrix33 J;
        J.SetIdentity();
        J(p, p) = c;
        J(p, q) = s;
        J(q, p) = -s;
        J(q, q) = c;
        *v = *v * J;
        *a = (Transpose(J) * (*a)) * J;
        float off = 0.0f;
        for (u32 i = 0; i < 3; ++i)
            for (u32 j = 0; j < 3; ++j)
                if (i != j) off += (*a)(i, j) * (*a)(i, j);
        if (sweep > 2 && off >= prev_off) return;
        prev_off = off;
    }
}

Sphere SphereAlongDominantAxis(const std::vector<Vector3> & pts) {   // bsphere.cpp:158-195
    Matrix33 m = Covariance(pts);
    Matrix33 v;
    JacobiEigen(&m, &v);
    u32 axis = 0;
    float best = fabsf(m(0, 0));
    if (fabsf(m(1, 1)) > best) { axis = 1; best = fabsf(m(1, 1)); }
    if (fabsf(m(2, 2)) > best) { axis = 2; best = fabsf(m(2, 2)); }
    Vector3 dir(v(0, axis), v(1, axis), v(2, axis));

    u32 i_min = 0, i_max = 0;                                         // bsphere.cpp:28-44
    float lo = FLT_MAX, hi = -FLT_MAX;
    for (u32 i = 0; i < pts.size(); ++i) {
        float proj = Dot(pts[i], dir);
        if (proj < lo) { i_min = i; lo = proj; }
        if (proj > hi) { i_max = i; hi = proj; }
    }
    Sphere s;
    s.center = (pts[i_min] + pts[i_max]) * 0.5f;
    s.radius = Length(pts[i_min] - pts[i_max]) * 0.5f;
    for (size_t i = 0; i < pts.size(); ++i) GrowSphereToPoint(&s, pts[i]);
    return s;
}

Sphere RefineRitter(Sphere s, std::vector<Vector3> & pts) {          // bsphere.cpp:197-231
    RandomState rng;
    Random_Seed(&rng, 0x201701260526ull);
    const u32 n = (u32)pts.size();
    Sphere trial = s;
    for (u32 pass = 0; pass < 16; ++pass) {
        trial.radius *= 0.9f;
        for (u32 i = 0; i < n; ++i) {
            u32 remaining = n - i - 1;
            if (remaining) {
                u32 j = (u32)Random_Next(&rng) % remaining + i + 1;
                Vector3 tmp = pts[i];
                pts[i] = pts[j];
                pts[j] = tmp;
            }
            GrowSphereToPoint(&trial, pts[i]);
        }
        if (trial.radius < s.radius) s = trial;
    }
    for (u32 i = 0; i < n; ++i) GrowSphereToPoint(&s, pts[i]);
    return s;
}

Sphere EnclosePair(Sphere s0, Sphere s1) {                            // bsphere.cpp:247-278
    Sphere out;
    Vector3 v = s1.center - s0.center;
    float sq_dist = Dot(v, v);
    float dr = s1.radius - s0.radius;
    if ((dr * dr) >= sq_dist) {
        out = (s1.radius >= s0.radius) ? s1 : s0;
    } else {
        float dist = sqrtf(sq_dist);
        out.radius = (dist + s0.radius + s1.radius) * 0.5f;
        out.center = s0.center;
        if (dist > 0.001f) {
            v /= dist;
            out.center += v * (out.radius - s0.radius);
        }
    }
    out.radius *= 1.0001f;
    return out;
}

struct BuildNode {
    Sphere s;
    BuildNode * child[2];
    MeshGroup * group;
};

u32 FlattenPreorder(BoundingHierarchy * h, BuildNode * n) {           // bsphere.cpp:328-350
    u32 me = (u32)h->spheres.size();
    h->mesh_groups.push_back(n->group);
    h->spheres.push_back(BoundingSphere());
    h->spheres[me].s = n->s;
    u32 c0 = 0, c1 = 0;                                               // 0 (the root) doubles as "no child"
    if (n->child[0]) {
        c0 = FlattenPreorder(h, n->child[0]);
        c1 = FlattenPreorder(h, n->child[1]);
    }
    h->spheres[me].c0 = c0;
    h->spheres[me].c1 = c1;
    delete n;
    return me;
}

}  // namespace

void BuildHierarchy(BoundingHierarchy * h, Mesh * mesh) {
    std::vector<BuildNode *> live;
    for (size_t g = 0; g < mesh->groups.size(); ++g) {                // bsphere.cpp:233-245, 382-394
        MeshGroup * mg = &mesh->groups[g];
        std::vector<Vector3> pts(mg->idx_positions.size());
        for (size_t i = 0; i < pts.size(); ++i) pts[i] = mesh->positions[mg->idx_positions[i]];
        BuildNode * n = new BuildNode;
        n->s = pts.empty() ? Sphere() : RefineRitter(SphereAlongDominantAxis(pts), pts);
        n->child[0] = n->child[1] = NULL;
        n->group = mg;
        live.push_back(n);
    }

    while (live.size() >= 2) {                                        // bsphere.cpp:280-314, 396-425
        float best_r = FLT_MAX;
        size_t bi = live.size(), bj = live.size();
        Sphere merged;
        for (size_t i = 0; i < live.size(); ++i) {
            const Sphere si = live[i]->s;
            for (size_t j = i + 1; j < live.size(); ++j) {
                Sphere parent = EnclosePair(si, live[j]->s);
                if (parent.radius < best_r) {
                    best_r = parent.radius;
                    bi = i;
                    bj = j;
                    merged = parent;
                }
            }
        }
        if (bi >= live.size()) break;          // every candidate radius was NaN / not finite
        BuildNode * parent = new BuildNode;
        parent->s = merged;
        parent->child[0] = live[bi];
        parent->child[1] = live[bj];
        parent->group = NULL;
        live.erase(live.begin() + (long)bj);   // bj > bi: erase the later one first
        live.erase(live.begin() + (long)bi);
        live.push_back(parent);
    }

    h->spheres.clear();
    h->mesh_groups.clear();
    h->mesh = mesh;
    if (!live.empty()) FlattenPreorder(h, live[0]);
}
