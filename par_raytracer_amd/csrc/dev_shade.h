// dev_shade.h - TraceRayColor as a resumable per-sample state machine.
//
// The reference shades by recursion (raytracer.cpp:413-577): at every hit one shadow ray per light, then
// (if iters > 0) reflection_samples diffuse children and spec_samples specular children, each a full
// recursive TraceRayColor that may be killed by Russian roulette, then an optional alpha continuation.
// The RNG is consumed in strict depth-first order of that recursion, so per sample there is exactly one
// "next ray" at any time.  sample_advance() runs the recursion as an explicit state machine: it consumes
// the result of the ray it asked for last, advances until the next ray is needed, and returns it (or
// reports that the sample is finished).  All float expressions keep the reference's association so the
// colour matches the CPU path to rounding of powf only; control flow (hit / miss, roulette, directions)
// is bit exact.
//
// Frames: one per recursion level L (iters = bounce_depth - L).  The live frame is held in registers;
// on a child call it is saved to a FrameStore slot and restored when the child returns.
#pragma once

#include "dev_rng.h"
#include "dev_scene.h"
#include "dev_trace.h"

namespace prt {

enum { ST_LIGHT = 0, ST_REFL = 1, ST_SPEC = 2, ST_ALPHA = 3 };
enum { PH_START = 0, PH_CLOSEST = 1, PH_SHADOW = 2 };

struct Frame {
    f3 ray_o, ray_d;      // the ray this TraceRayColor invocation was called with
    f3 hit_pos;           // hit.position                                  raytracer.cpp:121
    f3 hit_p;             // hit.position + hit.normal * ray_bias          raytracer.cpp:425
    f3 hit_n;             // shading normal                                raytracer.cpp:467
    f3 direct, dspec;     // direct_light, direct_specular_light           raytracer.cpp:505-511
    f3 indirect, ispec;   // indirect_light, indirect_specular_light       raytracer.cpp:513-536
    f3 color;             // assembled colour while an alpha continuation is in flight
    float child_w;        // weight the pending child's colour is multiplied by
    float alpha;
    int mat;
    int stage;
    int idx;              // light index / sample index inside the stage
};

struct SampleState {
    Rng rng;
    int level;
    int phase;
    f3 ret;               // final colour once sample_advance() returns false
};

struct RayReq {
    f3 o, d;              // as passed to TraceRay (origin not yet biased)
    int kind;
};

PRT_D f3 tangent_to_world(f3 normal, f3 ts) {          // raytracer.cpp:306-312 and 330-336
    f3 up = fabsf(normal.z) < 0.9999f ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f);
    f3 tangent = normalize3(cross3(up, normal));
    f3 bitangent = normalize3(cross3(normal, tangent));
    return normalize3(tangent * ts.x + bitangent * ts.y + normal * ts.z);
}

PRT_D f3 reflect3(f3 v, f3 normal) { return normal * 2.0f * dot3(v, normal) - v; }   // raytracer.cpp:343-346: (n*2)*dot - v

PRT_D float fresnel_amount(float ior_exit, float ior_enter, f3 normal, f3 incident) {  // raytracer.cpp:348-371
    float r0 = (ior_exit - ior_enter) / (ior_exit + ior_enter);
    r0 *= r0;
    float ct = ref_max(0.0f, -dot3(normal, incident));
    if (ior_exit > ior_enter) {
        float n = ior_exit / ior_enter;
        float st_sq = n * n * (1.0f - ct * ct);
        if (st_sq > 1.0f) return 1.0f;
        ct = sqrtf(1.0f - st_sq);
    }
    float x = 1.0f - ct;
    float x2 = x * x;
    float x3 = x * x2;
    return r0 + (1.0f - r0) * x2 * x3;
}

PRT_D f3 make_camera_dir(const DevCamera & cam, float ox, float oy) {                  // main.cpp:164-177
    float normalized_x = 2.0f * (ox + 0.5f) * cam.inv_width - 1.0f;
    float normalized_y = 1.0f - 2.0f * (oy + 0.5f) * cam.inv_height;
    f3 dir = cam.forward + cam.right_scaled * normalized_x + cam.up_scaled * normalized_y;
    return normalize3(dir);
}

// Seeds the sample's RNG and produces its camera ray (main.cpp:237-241 with the per-sample key).
template <bool RING>
PRT_D void sample_begin(const DevCamera & cam, const DevParams & P, unsigned int pixel, unsigned int samp, SampleState & S,
                        Frame & cur, u64 * ring, size_t ring_stride) {
    rng_seed(S.rng, prt_sample_key(P.seed, pixel, samp));
    // Vector2 sample_offset(NextFloat11, NextFloat11): g++ evaluates right to left, first draw -> .y
    float off_y = rng_float11<RING>(S.rng, ring, ring_stride);
    float off_x = rng_float11<RING>(S.rng, ring, ring_stride);
    unsigned int x = pixel % P.width, y = pixel / P.width;
    float px = (float)x + off_x * 0.5f;
    float py = (float)y + off_y * 0.5f;
    cur.ray_o = cam.position;
    cur.ray_d = make_camera_dir(cam, px, py);
    S.level = 0;
    S.phase = PH_START;
}

// Returns true with `req` filled when a ray must be traced, false when the sample is complete (S.ret).
// `hit` is the result of the previously requested ray (ignored when S.phase == PH_START).
template <bool RING, class FrameStore>
PRT_D bool sample_advance(const DevScene & sc, const DevParams & P, SampleState & S, Frame & cur, FrameStore & store,
                          const HitRec & hit, RayReq & req, unsigned int & shaded_hits, u64 * ring, size_t ring_stride) {
    enum { PC_ENTER, PC_GOT_CLOSEST, PC_LIGHT, PC_GOT_SHADOW, PC_REFL, PC_SPEC, PC_FINAL, PC_RETURN };
    int pc = S.phase == PH_START ? PC_ENTER : (S.phase == PH_CLOSEST ? PC_GOT_CLOSEST : PC_GOT_SHADOW);
    const int depth = (int)P.bounce_depth;
    f3 ret = mk3(0.0f, 0.0f, 0.0f);

    for (;;) {
        switch (pc) {
        case PC_ENTER: {                                            // raytracer.cpp:415-423
            int iters = depth - S.level;
            if (iters < 0) { ret = mk3(0.0f, 0.0f, 0.0f); pc = PC_RETURN; break; }
            if (S.level != 0 && rng_float01<RING>(S.rng, ring, ring_stride) < 0.5f) {
                ret = mk3(0.0f, 0.0f, 0.0f);
                pc = PC_RETURN;
                break;
            }
            req.o = cur.ray_o;
            req.d = cur.ray_d;
            req.kind = TRACE_CLOSEST;
            S.phase = PH_CLOSEST;
            return true;
        }
        case PC_GOT_CLOSEST: {
            if (hit.tri < 0) { ret = P.background; pc = PC_RETURN; break; }        // raytracer.cpp:573-575
            shaded_hits++;
            const f3 ob = cur.ray_o + cur.ray_d * P.ray_bias;                       // raytracer.cpp:163
            const f3 pos = ob + cur.ray_d * hit.t;                                  // raytracer.cpp:121
            const float4 * sp = sc.shade + 4 * (size_t)hit.tri;
            const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
            const f3 gn = normalize3(mk3(s2.y, s2.z, s2.w));                        // raytracer.cpp:122 (n = Cross(ab, ac))
            const int m = as_i(s3.w);
            const DevMaterial mat = sc.materials[m];
            cur.hit_pos = pos;
            cur.hit_p = pos + gn * P.ray_bias;                                      // raytracer.cpp:425
            float alpha = mat.alpha;
            if (mat.alpha <= 1.0f) {                                                // raytracer.cpp:443-453
                if (alpha <= 0.05f) {
                    cur.ray_o = pos + cur.ray_d * P.ray_bias * 2.0f;
                    pc = PC_ENTER;                                                  // same iters; roulette may fire again
                    break;
                }
            }
            const float bwy = hit.v, bwz = hit.w;
            const float bwx = 1.0f - bwy - bwz;                                     // raytracer.cpp:120
            f3 interp = mk3(0.0f, 0.0f, 0.0f);                                      // raytracer.cpp:464-467
            interp = interp + mk3(s0.x, s0.y, s0.z) * bwx;
            interp = interp + mk3(s0.w, s1.x, s1.y) * bwy;
            interp = interp + mk3(s1.z, s1.w, s2.x) * bwz;
            cur.hit_n = normalize3(interp);
            cur.direct = cur.dspec = cur.indirect = cur.ispec = mk3(0.0f, 0.0f, 0.0f);
            cur.mat = m;
            cur.alpha = alpha;
            cur.stage = ST_LIGHT;
            cur.idx = 0;
            pc = PC_LIGHT;
            break;
        }
        case PC_LIGHT: {                                            // raytracer.cpp:507-511, 234-250
            if ((unsigned int)cur.idx < sc.light_count) {
                const DevLight L = sc.lights[cur.idx];
                req.o = cur.hit_p;
                if (L.type == 0) {
                    req.d = mk3(L.facing[0], L.facing[1], L.facing[2]) * -1.0f;
                    req.kind = TRACE_ANY;          // only the boolean is used (raytracer.cpp:385)
                } else {
                    req.d = normalize3(mk3(L.position[0], L.position[1], L.position[2]) - cur.hit_p);
                    req.kind = TRACE_CLOSEST;      // hit.t is compared with the light distance (raytracer.cpp:396)
                }
                S.phase = PH_SHADOW;
                return true;
            }
            cur.stage = ST_REFL;
            cur.idx = 0;
            pc = PC_REFL;
            break;
        }
        case PC_GOT_SHADOW: {                                       // raytracer.cpp:378-411
            const DevLight L = sc.lights[cur.idx];
            const DevMaterial mat = sc.materials[cur.mat];
            f3 light_color = mk3(L.color[0], L.color[1], L.color[2]);
            f3 light_vector;
            bool lit;
            if (L.type == 0) {
                light_vector = mk3(L.facing[0], L.facing[1], L.facing[2]) * -1.0f;
                lit = hit.tri < 0;
            } else {
                const f3 lp = mk3(L.position[0], L.position[1], L.position[2]);
                light_vector = normalize3(lp - cur.hit_p);
                const f3 dv = lp - cur.hit_p;
                const float light_dist_sq = dot3(dv, dv);
                lit = hit.tri < 0 || hit.t * hit.t <= light_dist_sq;               // inverted test preserved
                if (lit) {
                    float falloff_denom = (sqrtf(light_dist_sq) / L.falloff) + 1.0f;
                    light_color = light_color * (1.0f / (falloff_denom * falloff_denom));
                }
            }
            if (lit) {
                float spec_cos = dot3(cur.ray_d * -1.0f, reflect3(light_vector, cur.hit_n));
                f3 dd = light_color * 2.0f * ref_max(0.0f, dot3(cur.hit_n, light_vector));
                f3 ds = light_color * powf(ref_max(0.0f, spec_cos), mat.specular_intensity);
                cur.direct = cur.direct + dd;
                cur.dspec = cur.dspec + ds;
            }
            cur.idx++;
            pc = PC_LIGHT;
            break;
        }
        case PC_REFL: {                                             // raytracer.cpp:515-526
            const int iters = depth - S.level;
            if (iters > 0 && (unsigned int)cur.idx < P.reflection_samples) {
                unsigned int series_i = (unsigned int)(rng_next<RING>(S.rng, ring, ring_stride) % 1024ull);
                const float4 ts = sc.diffuse_dirs[series_i];
                f3 dir = tangent_to_world(cur.hit_n, mk3(ts.x, ts.y, ts.z));
                cur.child_w = ref_max(0.0f, dot3(cur.hit_n, dir));
                f3 o = cur.hit_p;
                store.save(S.level, cur);
                S.level++;
                cur.ray_o = o;
                cur.ray_d = dir;
                pc = PC_ENTER;
                break;
            }
            cur.stage = ST_SPEC;
            cur.idx = 0;
            pc = PC_SPEC;
            break;
        }
        case PC_SPEC: {                                             // raytracer.cpp:528-535
            const int iters = depth - S.level;
            if (iters > 0 && (unsigned int)cur.idx < P.spec_samples) {
                const float4 ts = sc.spec_dirs[(size_t)cur.mat * sc.spec_samples + (unsigned int)cur.idx];
                f3 dir = tangent_to_world(cur.hit_n, mk3(ts.x, ts.y, ts.z));
                cur.child_w = ref_max(0.0f, dot3(dir, cur.ray_d * -1.0f));
                f3 o = cur.hit_p;
                store.save(S.level, cur);
                S.level++;
                cur.ray_o = o;
                cur.ray_d = dir;
                pc = PC_ENTER;
                break;
            }
            pc = PC_FINAL;
            break;
        }
        case PC_FINAL: {                                            // raytracer.cpp:538-552
            const DevMaterial mat = sc.materials[cur.mat];
            float object_reflectivity = 0.04f;
            float fresnel = fresnel_amount(1.0f, mat.index_of_refraction, cur.hit_n, cur.ray_d);
            float w_reflect = (object_reflectivity + (1.0f - object_reflectivity) * fresnel);
            float w_diffuse = 1.0f - w_reflect;
            f3 color = mk3(0.0f, 0.0f, 0.0f);
            color = color + mk3(mat.ambient[0], mat.ambient[1], mat.ambient[2]) * 0.1f;
            color = color + (cur.indirect + cur.direct) * mk3(mat.diffuse[0], mat.diffuse[1], mat.diffuse[2]) * w_diffuse;
            color = color + (cur.ispec + cur.dspec) * mk3(mat.specular[0], mat.specular[1], mat.specular[2]);
            if (cur.alpha < 1.0f) {
                cur.color = color;
                cur.stage = ST_ALPHA;
                f3 o = cur.hit_pos + cur.ray_d * P.ray_bias * 2.0f;
                f3 d = cur.ray_d;
                store.save(S.level, cur);
                S.level++;
                cur.ray_o = o;
                cur.ray_d = d;
                pc = PC_ENTER;
                break;
            }
            ret = color;
            pc = PC_RETURN;
            break;
        }
        case PC_RETURN: {
            if (S.level == 0) {
                S.ret = ret;
                return false;
            }
            S.level--;
            store.load(S.level, cur);
            if (cur.stage == ST_REFL) {
                cur.indirect = cur.indirect + ret * cur.child_w;
                cur.idx++;
                pc = PC_REFL;
            } else if (cur.stage == ST_SPEC) {
                cur.ispec = cur.ispec + ret * cur.child_w;
                cur.idx++;
                pc = PC_SPEC;
            } else {
                ret = cur.color * cur.alpha + ret * (1.0f - cur.alpha);
                pc = PC_RETURN;
            }
            break;
        }
        }
    }
}

}  // namespace prt
