// kernels_debug.h - device known-answer kernel: runs single device functions of the hot path on caller-supplied
// vectors so the GPU test-suite can compare them, bit for bit, with what the reference's own functions returned
// (tests/golden/kat.npz).  Test hook only; not part of the render path.
#pragma once

#include "dev_shade.h"

namespace prt {

enum { KAT_RNG_NEXT = 0, KAT_RNG_FLOAT01 = 1, KAT_TRIANGLE = 2, KAT_DIFFUSE_DIR = 3, KAT_CAMERA_RAY = 4, KAT_FRESNEL = 5,
       KAT_TANGENT_TO_WORLD = 6, KAT_RNG_NEXT_COMPACT = 7 };

__global__ void k_debug_kat(int kind, const void * in, void * out, unsigned int n, DevScene sc, DevCamera cam, u64 * ring_ws) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (kind) {
    case KAT_RNG_NEXT: {                       // in: u64 seed[n]; out: u64[n][40]  (RING variant: wraps the 16-word state)
        Rng r;
        rng_seed(r, ((const u64 *)in)[i]);
        for (int k = 0; k < 40; ++k) ((u64 *)out)[(size_t)i * 40 + k] = rng_next<true>(r, ring_ws + i, n);
    } break;
    case KAT_RNG_NEXT_COMPACT: {               // the 2-register variant, valid for the first 15 draws; out: u64[n][15]
        Rng r;
        rng_seed(r, ((const u64 *)in)[i]);
        for (int k = 0; k < 15; ++k) ((u64 *)out)[(size_t)i * 15 + k] = rng_next<false>(r, nullptr, 0);
    } break;
    case KAT_RNG_FLOAT01: {                    // out: float[n][24]
        Rng r;
        rng_seed(r, ((const u64 *)in)[i]);
        for (int k = 0; k < 24; ++k) ((float *)out)[(size_t)i * 24 + k] = rng_float01<true>(r, ring_ws + i, n);
    } break;
    case KAT_TRIANGLE: {                       // in: o d a ab ac n max_t (19 floats, pre-differenced on the host as upload does)
        const float * p = (const float *)in + (size_t)i * 19;
        const f3 o = mk3(p[0], p[1], p[2]), d = mk3(p[3], p[4], p[5]);
        const f3 a = mk3(p[6], p[7], p[8]), ab = mk3(p[9], p[10], p[11]), ac = mk3(p[12], p[13], p[14]), nn = mk3(p[15], p[16], p[17]);
        const f3 qp = o - (o + d);
        // IntersectRayTriangle itself (raytracer.cpp:82-125): the geometric tests and the early reject against out_hit->t;
        // whether the caller keeps the hit (strict <) is IntersectRayMesh's business
        float tn, dd, vn, wn, t = 0.0f, v = 0.0f, w = 0.0f;
        bool hit = tri_geom(o, qp, a, ab, ac, nn, tn, dd, vn, wn) && !(tn > p[18] * dd);
        if (hit) { const float ood = 1.0f / dd; t = tn * ood; v = vn * ood; w = wn * ood; }
        float * q = (float *)out + (size_t)i * 11;
        for (int k = 0; k < 11; ++k) q[k] = 0.0f;
        if (hit) {
            const f3 pos = o + d * t;
            const f3 gn = normalize3(nn);
            q[0] = 1.0f; q[1] = t; q[2] = 1.0f - v - w; q[3] = v; q[4] = w;
            q[5] = pos.x; q[6] = pos.y; q[7] = pos.z; q[8] = gn.x; q[9] = gn.y; q[10] = gn.z;
        }
    } break;
    case KAT_DIFFUSE_DIR: {                    // in: normal xyz, series index; out: direction (table from the uploaded scene)
        const float * p = (const float *)in + (size_t)i * 4;
        const float4 ts = sc.diffuse_dirs[(unsigned int)p[3] & 1023u];
        const f3 dir = tangent_to_world(mk3(p[0], p[1], p[2]), mk3(ts.x, ts.y, ts.z));
        float * q = (float *)out + (size_t)i * 3;
        q[0] = dir.x; q[1] = dir.y; q[2] = dir.z;
    } break;
    case KAT_CAMERA_RAY: {                     // in: pixel position xy; out: direction
        const float * p = (const float *)in + (size_t)i * 2;
        const f3 dir = make_camera_dir(cam, p[0], p[1]);
        float * q = (float *)out + (size_t)i * 3;
        q[0] = dir.x; q[1] = dir.y; q[2] = dir.z;
    } break;
    case KAT_FRESNEL: {                        // in: Ni, normal, incident; out: float
        const float * p = (const float *)in + (size_t)i * 7;
        ((float *)out)[i] = fresnel_amount(1.0f, p[0], mk3(p[1], p[2], p[3]), mk3(p[4], p[5], p[6]));
    } break;
    case KAT_TANGENT_TO_WORLD: {               // in: normal xyz, tangent-space dir xyz; out: direction
        const float * p = (const float *)in + (size_t)i * 6;
        const f3 dir = tangent_to_world(mk3(p[0], p[1], p[2]), mk3(p[3], p[4], p[5]));
        float * q = (float *)out + (size_t)i * 3;
        q[0] = dir.x; q[1] = dir.y; q[2] = dir.z;
    } break;
    default: break;
    }
}

}  // namespace prt
