// dev_scene.h - how the scene lies in HBM, and the kernel-side view of it.
//
// The reference walks a pointer graph (Scene -> SceneObject -> MeshGroup -> index buffers -> positions,
// scene.h:22-36, mesh.h:40-57).  For the device everything is flattened at upload into dense,
// 16-byte-aligned records that one lane fetches with dwordx4 loads and no dependent indirection:
//
//   nodes      4-wide BVH, 64 B per node = 16 dwords, child boxes quantised to 8 bits per plane on the
//              node's own power-of-two grid (plane = origin + q * 2^e, rounded outward at build time):
//                d0-2  origin xyz (float: lo corner of the union of the children)
//                d3    2^e_x as a float (the grid step along x; a multiply away from the slab test)
//                d4-6  lo planes x / y / z, one byte per child      d7-9  hi planes x / y / z
//                d10-13 child links                                 d14-15 2^e_y, 2^e_z
//              (children fill the slots from 0; an empty slot has lo = 255 > hi = 0 on every axis)
//              link >= 0: node index; link < 0: leaf, ~link = (first_tri << 2) | (count - 1).
//              One node visit = one 64 B fetch that decides about four subtrees: half the bytes per ray of
//              an uncompressed binary node with the same fetch size, and half the dependent fetches.
//              Nodes are numbered breadth-first, so the top of the tree is one contiguous block.
//   tris       48 B per triangle = 3 x float4, in BVH leaf order, un-indexed and pre-differenced:
//                (a.xyz, ab.x) (ab.yz, ac.xy) (ac.z, n.xyz)   with ab = b-a, ac = c-a, n = Cross(ab, ac)
//              computed on the host with the reference's float expressions (raytracer.cpp:85-91), so the
//              device test consumes the same bits the CPU test would compute per call
//   shade      64 B per triangle = 4 x float4, same order: the three vertex normals, the geometric normal
//              n (again) and the material index - ONE gather per shaded hit, not per test
//   tri_rank   u32 per triangle: position in the reference's own visit order (sphere tree DFS with c1
//              first, then ascending index inside a group; raytracer.cpp:136, 208-209).  Only read when
//              two hits have bit-equal t, to keep the reference's "first hit wins" (strict <, :149, :220)
//   tri_uv, tri_tan  only for scenes with texture maps: 32 B of texture coordinates and 48 B of vertex tangents per
//              triangle, same order; textures as RGBA8 texels + a 256-entry sRGB->linear table (dev_texture.h)
//   materials  64 B each; lights 48 B each; diffuse_dirs 1024 x float4 (the Hammersley set of
//              raytracer.cpp:519-521 pushed through cosf/sinf/sqrtf on the HOST, so no device
//              transcendental ever feeds a direction); spec_dirs [material][spec_samples] likewise.
#pragma once

#include "dev_math.h"

namespace prt {

struct DevMaterial {          // 64 B
    float ambient[3];
    float specular_intensity;
    float diffuse[3];
    float index_of_refraction;
    float specular[3];
    float alpha;
    int flags;
    // texture slots, two 16-bit texture numbers per word (0xFFFF = none):
    //   tex[0] = ambient | diffuse << 16    tex[1] = specular | alpha << 16    tex[2] = bump | 0xFFFF << 16
    unsigned int tex[3];
};

enum { DEV_TEX_NONE = 0xFFFFu };

struct DevTexture {           // 16 B
    unsigned int first_texel;  // into DevScene::texels
    unsigned int size_x, size_y;
    unsigned int pad;
};

struct DevLight {             // 48 B
    float color[3];
    int type;
    float position[3];
    float falloff;
    float facing[3];
    float pad;
};

struct DevScene {
    const float4 * nodes;
    const float4 * tris;
    const float4 * shade;
    const unsigned int * tri_rank;
    const DevMaterial * materials;
    const DevLight * lights;
    const float4 * diffuse_dirs;
    const float4 * spec_dirs;
    unsigned int light_count;
    unsigned int material_count;
    unsigned int spec_samples;
    unsigned int tri_count;
    unsigned int node_count;
    // texture path (only scenes with map_* records; NULL otherwise).  dev_texture.h
    const DevTexture * textures;
    const unsigned int * texels;     // RGBA8, every texture expanded to 4 channels by GetTexel's rules (texture.cpp:27-41)
    const float * srgb_lut;          // 256 x Color_SRGBToLinear(i / 255), host powf
    const float4 * tri_uv;           // 32 B per triangle: (uv0, uv1) (uv2, -)
    const float4 * tri_tan;          // 48 B per triangle: the three vertex tangents, packed like the normals in `shade`
    // near-tie resolution only (dev_trace_common.h RefSphereWalk): the reference's sphere tree, 32 B per sphere:
    // (centre xyz, radius) (c0, c1, visit rank of c0's first triangle, position in the reference's pop order); NULL when the
    // scene came without a usable tree
    const float4 * ref_spheres;
    unsigned int tie_widen_max;                 // resolve_near_ties: widenings of the candidate set before it gives up (8)
    unsigned long long * near_tie_unresolved;   // ... and counts the event here (the render call then fails)
};

struct DevCamera {            // Camera (main.cpp:133-143) with the loop invariants of MakeCameraRay hoisted
    f3 position;
    f3 forward;
    f3 right_scaled;          // (camera_right * tan_a2) * aspect    main.cpp:170
    f3 up_scaled;             // camera_up * tan_a2                  main.cpp:171
    float inv_width, inv_height;
};

struct DevParams {
    float ray_bias;
    unsigned int reflection_samples, spec_samples, bounce_depth;
    f3 background;
    unsigned int spp;
    unsigned long long seed;
    unsigned int width, height;
    float box_pad;            // outward padding of every BVH box, world units (dev_trace.h)
    // which pixels this launch renders: local pixel lp (0 .. n_pixels) -> linear image pixel.
    //   shard_nranks == 1: first_pixel + lp                       (RenderTask's [start_idx, end_idx), main.cpp:273)
    //   otherwise: row-blocks of shard_block_rows rows, block b of this rank = image block b*nranks + rank
    unsigned int first_pixel, shard_block_rows, shard_rank, shard_nranks;
    const unsigned int * pixel_list;   // explicit pixel ids (prt_render_pixel_list) or NULL
    // traversal stack: LDS entries per lane, then a global spill column per lane for the rest of the bound (dev_trace.h
    // LdsStack; null when the LDS column covers the bound)
    int * stack_spill;
    unsigned int stack_lds_entries, stack_spill_stride;
    // full-height global stack columns of k_trace_exact's fixed grid (wavefront pipeline)
    int * exact_stack;
    unsigned int exact_stack_stride;
    unsigned int local_base;                    // large calls run in passes of bounded workspace (prt_api.hip render_pixels)
    // Work order: the first tile_pixels pixels of the set (whole bands of 8 rows; full-width row sets only, width % 8 == 0)
    // are handed out tile by tile - 64 consecutive work items are an 8 x 8 pixel tile, not a 64 x 1 strip - so that the rays
    // a wave traces together start from one compact patch of the image (local_of_work below).  0 = off.
    unsigned int tile_pixels;
    // Work order reversed (option WORK_REVERSE, round 4 experiment): work item wi of a call of n pixels renders what item
    // n - 1 - wi would - the bottom of the image first, so that the samples fetched LAST, whose bounce chains are the frame's
    // tail, are the top rows (sky in outdoor scenes: one ray and done).  n, or 0 = off.  Never with a pixel list.
    unsigned int work_reverse_n;
    // Work order scattered (option WORK_SCATTER, round 4 experiment): the 8-pixel chunks of the tiled region (one wave-load of
    // 8 spp samples each) are handed out in the order c -> c * mul mod n instead of one after the other, so that what a wave
    // takes in one top-up - 8 chunks - comes from 8 places of the image and is a mix of cheap and dear pixels, not one 8 x 8
    // tile of sky or of terrain.  n = chunks in the tiled region (0 = off), mul coprime to n, (n - 1) * mul < 2^32.
    unsigned int work_scatter_n, work_scatter_mul;
    // 1: a shadow ray whose radiance-if-unoccluded is exactly zero is counted, not traced (kernels_wave.h shade_entry);
    // 0 (PRT_TRACE_DEAD_SHADOW_RAYS): traced like every other one
    unsigned int elide_dead_shadow_rays;
    // adaptive sampling (main.cpp:245-258): on when max_spp > spp; k_pool<ADAPT> only
    unsigned int max_spp;
    float variance_threshold;
};

// Work item (position in the order pixels are handed out, whole call) -> local pixel (position in the call's output).
PRT_HD unsigned int local_of_work(unsigned int wi, unsigned int width, unsigned int tile_pixels, unsigned int reverse_n = 0u,
                                  unsigned int scatter_n = 0u, unsigned int scatter_mul = 1u) {
    if (reverse_n) wi = reverse_n - 1u - wi;
    if (wi >= tile_pixels) return wi;
    if (scatter_n) wi = (((wi >> 3) * scatter_mul) % scatter_n) * 8u + (wi & 7u);
    const unsigned int band_px = 8u * width;
    const unsigned int b = wi / band_px, q = wi - b * band_px;
    const unsigned int t = q >> 6, i = q & 63u;
    return b * band_px + (i >> 3) * width + t * 8u + (i & 7u);
}

// Work item lp of the current pass -> linear image pixel.
PRT_HD unsigned int pixel_of_local(const DevParams & P, unsigned int lp) {
    lp += P.local_base;                         // this pass's first work item within the call's pixel set
    if (P.pixel_list) return P.pixel_list[lp];
    lp = local_of_work(lp, P.width, P.tile_pixels, P.work_reverse_n, P.work_scatter_n, P.work_scatter_mul);
    if (P.shard_nranks <= 1) return P.first_pixel + lp;
    const unsigned int row = lp / P.width, x = lp - row * P.width;
    const unsigned int blk = row / P.shard_block_rows, r = row - blk * P.shard_block_rows;
    const unsigned int y = (blk * P.shard_nranks + P.shard_rank) * P.shard_block_rows + r;
    return y * P.width + x;
}

struct DevCounters {          // device-side accumulators (atomics, one add per wave)
    unsigned long long ray_count;
    unsigned long long node_visits;
    unsigned long long tri_tests;
    unsigned long long shaded_hits;
    // wave-level step counts (COUNT builds only): lane utilisation = lane-level count / (64 * wave-level count)
    unsigned long long wave_node_steps, wave_leaf_steps, wave_tri_steps, wave_refills, max_sp, culled;
    unsigned long long wave_node_step_rays;          // k_pool: lanes holding a ray, summed over its wave-level node steps
    unsigned long long drain_node_steps, drain_node_step_rays;   // ... and the part of both taken while a round's list had nothing left to hand out
    // k_pool, COUNT builds: wave-cycles (s_memtime) spent in the top-up / trace / shade phase, in the whole main loop, and
    // (adaptive mode) in the finalise step, which is part of the shade phase
    unsigned long long phase_cycles[5];
    unsigned long long wave_cycles_max, wave_cycles_sum, wave_count;   // k_pool (fast kernel), COUNT builds: the waves' main loops
    // pool pipeline: the most entries any pass wanted to put on its park lists ([0] closest-hit rays + finalise steps, [1] shadow
    // rays); above the lists' capacities the frame is incomplete and render_pixels renders it again with longer lists
    unsigned long long park_peak[2];
    // ... and whether any PASS wanted more than ITS lists held (a call's passes clamp the lists to their own worst case: the
    // peak of one pass says nothing about the capacity of another): the largest such demand, 0 when every pass fitted
    unsigned long long park_over[2];
    unsigned long long elided_shadow_rays;     // shadow rays counted (they are in ray_count) but not traced: they could not change the image
    unsigned long long near_tie_unresolved;    // resolve_near_ties gave up widening its candidate set (the render call fails)
    unsigned long long flow_cycles[8];         // k_flow, COUNT builds: tracers waiting / whole loop; shading wave topping up / shading / waiting / whole loop; batches; hits in them
    unsigned int flow_error, flow_pad;         // k_flow (kernels_flow.h): a watchdog fired (bit mask of which wait; the render call fails)
    // adaptive mode: stopping-rule verdicts whose variance lay within 0.1 % of the threshold (k_pool's finalise step): the only
    // verdicts the device's last bits could turn against the reference's
    unsigned long long variance_close_calls;
};

// Per-sample radiance accumulator of the wavefront and pool pipelines: 2^-32 fixed point in 64-bit integers.  A sample's
// contributions arrive in an order that depends on the pipeline, on the number of lights (the shadow rays of one hit finish
// in either order), on the slow path (a parked ray lands later) - integer addition does not care, so the sample's radiance
// is the same bits every time.  A contribution is truncated to a multiple of 2^-32 (exact for floats >= 2^-8; the whole
// chain stays far inside the 1e-4 tolerance); |sum| < 2^31.
struct Accum { long long x, y, z, w; };

PRT_D long long accum_fix(float c) {
    const float m = fabsf(c);
    const unsigned int hi = (unsigned int)m;                                    // v_cvt_u32_f32: truncates, saturates, NaN -> 0
    const unsigned int lo = (unsigned int)((m - (float)hi) * 4294967296.0f);    // (float)hi is exact wherever the difference is not 0
    const long long v = (long long)((unsigned long long)hi << 32 | lo);
    return c < 0.0f ? -v : v;
}
PRT_D float accum_float(long long v) { return (float)((double)v * (1.0 / 4294967296.0)); }
// Additions are atomics without return value, always: the issuing wave does not wait for them (a read-modify-write would
// park it for a memory round trip at every finished shadow ray), and several shadow rays of one sample may finish at once.
// They are performed in the L2, past the compute unit's vector L1 - so reads must not be served from that L1 either.
PRT_D void accum_add(Accum * a, f3 c) {

    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&a->x), (unsigned long long)accum_fix(c.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&a->y), (unsigned long long)accum_fix(c.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&a->z), (unsigned long long)accum_fix(c.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Additions by the sample's OWNER: a lane that is, for the duration of its phase, the only one that touches *a (the shading
// lane of the sample's one closest-hit ray: kernels_wave.h shade_entry_on).  A plain read-modify-write - loads past the L1,
// where a shadow ray's atomic of the previous phase may have changed the record in the L2 - of one 24-byte run that the lanes
// of a wave (consecutive samples) read and write as whole cache lines, in two halves so that the load is issued with the
// caller's other loads and the rest done when the contribution is known.  Integer addition: the result is the same bits
// whichever way a contribution arrives (the shadow rays' still arrive as atomics).
PRT_D void accum_load_owner(const Accum * a, long long & x, long long & y, long long & z) {
    x = __hip_atomic_load(&a->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    y = __hip_atomic_load(&a->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    z = __hip_atomic_load(&a->z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
PRT_D void accum_store_owner(Accum * a, long long x, long long y, long long z, f3 c) {
    a->x = x + accum_fix(c.x); a->y = y + accum_fix(c.y); a->z = z + accum_fix(c.z);
}
PRT_D f3 accum_read(const Accum * a) {
    const long long x = __hip_atomic_load(&a->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long y = __hip_atomic_load(&a->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long z = __hip_atomic_load(&a->z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return mk3(accum_float(x), accum_float(y), accum_float(z));
}

enum { BVH_LEAF_MAX = 4 };

}  // namespace prt
