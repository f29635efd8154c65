/* prt.h - C ABI of the MI355X ray-trace hot path (libprt_hip.so).
 *
 * The reference (ACEfanatic02/par_raytracer) has no plugin / FFI interface: main.cpp #includes the
 * whole renderer as one translation unit of static functions.  The seam this ABI replaces is
 *
 *     RenderTask(RenderJob*, DebugCounters*)                      main.cpp:267-283
 *       "fill buffer[0 .. end_idx-start_idx) with linear RGBA float4 for the linear pixel indices
 *        [start_idx, end_idx) of a w x h image, given an immutable camera, scene, global params
 *        and an RNG seed"
 *
 * one level below Render(Camera*, Scene*, u32 w, u32 h) -> Framebuffer (main.cpp:301-358), which the
 * host mirror (par_raytracer_amd/host/) re-implements on top of these entry points.
 *
 * Everything here is plain C: pointers, sizes and POD structs; no C++ or torch types.
 * Return convention: 0 = ok, negative = error (message via prt_last_error); nothing aborts: a C++ exception inside
 * the library (std::bad_alloc, std::length_error, std::system_error ...) is caught at the entry point and comes back as
 * PRT_ERR_EXCEPTION.
 * Threading: one host thread per context at a time; one HIP device and one stream per context.
 */
#ifndef PRT_H_
#define PRT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRT_ABI_VERSION 5
/* Error codes (every entry point that returns int): -1 bad argument, -2 no scene uploaded, -5 / -6 internal limits of the
 * wavefront pipeline, -7 park lists kept overflowing, -8 near-tied hits unresolved, -9 the BVH builder produced a tree that fails
 * the upload check (a link or triangle range outside the arrays: refused instead of traversed), -10 a HIP runtime call failed,
 * -11 too many frames in flight (prt_multi_submit), -12 a C++ exception was caught at the entry point. */
enum { PRT_ERR_ARGUMENT = -1, PRT_ERR_NO_SCENE = -2, PRT_ERR_HIP = -10, PRT_ERR_IN_FLIGHT = -11, PRT_ERR_EXCEPTION = -12 };

/* ---- scene description: the reference's pointer graph flattened to POD arrays ---------------- */

/* Material (mesh.h:15-32).  Colours are Vector4 {x,y,z,w}.  Texture slots are indices into
 * prt_scene_desc.textures or -1; the texture path is a "next" row. */
typedef struct prt_material {
    float specular_intensity;      /* Ns */
    float index_of_refraction;     /* Ni */
    float alpha;                   /* d  */
    float ambient_color[4];        /* Ka */
    float diffuse_color[4];        /* Kd */
    float specular_color[4];       /* Ks */
    int32_t ambient_texture;
    int32_t diffuse_texture;
    int32_t specular_texture;
    int32_t alpha_texture;
    int32_t bump_texture;
} prt_material;

/* LightSource (scene.h:3-15). */
enum { PRT_LIGHT_DIRECTIONAL = 0, PRT_LIGHT_POINT = 1 };
typedef struct prt_light {
    int32_t type;
    float color[4];
    float position[3];
    float facing[3];
    float falloff;
} prt_light;

/* MeshGroup (mesh.h:40-47): a contiguous run of the concatenated index buffers. */
typedef struct prt_group {
    uint32_t first_index;          /* offset into idx_* (multiple of 3) */
    uint32_t index_count;          /* 3 * triangles */
    int32_t material;              /* index into materials */
} prt_group;

/* BoundingSphere (bsphere.cpp:316-320), 24 bytes, same field order. */
typedef struct prt_bsphere {
    float center[3];
    float radius;
    uint32_t c0;
    uint32_t c1;
} prt_bsphere;

/* Texture (mesh.h:8-13). */
typedef struct prt_texture {
    uint32_t size_x, size_y, channels;
    const uint8_t * texels;
} prt_texture;

typedef struct prt_scene_desc {
    /* Mesh (mesh.h:49-57) */
    const float * positions;   uint32_t position_count;    /* xyz */
    const float * normals;     uint32_t normal_count;      /* xyz */
    const float * texcoords;   uint32_t texcoord_count;    /* uv  */
    const float * tangents;                                /* xyz per normal, may be NULL */
    /* MeshGroup index buffers of all groups, concatenated in group order */
    const uint32_t * idx_positions;
    const uint32_t * idx_texcoords;
    const uint32_t * idx_normals;
    uint32_t index_count;
    const prt_group * groups;        uint32_t group_count;
    const prt_material * materials;  uint32_t material_count;
    const prt_texture * textures;    uint32_t texture_count;
    /* Scene (scene.h:29-36) */
    const prt_light * lights;        uint32_t light_count;
    /* BoundingHierarchy (bsphere.cpp:322-326), flattened: node i <-> spheres[i]; sphere_group[i] is
     * the group index of a leaf or -1.  The HIP path builds its own per-triangle BVH and uses this
     * tree only to derive the reference's leaf visit order for exact-tie breaking (may be NULL);
     * the CPU oracle traverses it exactly as the reference does. */
    const prt_bsphere * spheres;
    const int32_t * sphere_group;
    uint32_t sphere_count;
} prt_scene_desc;

/* Camera (main.cpp:133-143), same field order. */
typedef struct prt_camera {
    float tan_a2, aspect, inv_width, inv_height;
    float position[3];
    float forward[3];
    float right[3];
    float up[3];
} prt_camera;

/* The subset of gParams (globals.h:9-22) the hot path reads, plus what the reference hard-codes:
 * spp (main.cpp:308-309 -> min_samples = max_samples = spp) and the RNG seed (main.cpp:60-67 ->
 * per-(pixel,sample) key, include/prt_key.h). */
typedef struct prt_params {
    float ray_bias;
    uint32_t reflection_samples;
    uint32_t spec_samples;
    uint32_t bounce_depth;
    float background_color[4];
    uint32_t spp;                  /* samples per pixel; in adaptive mode the fixed first part (min_samples, main.cpp:308) */
    uint32_t pipeline;             /* PRT_PIPELINE_* ; 0 = library default */
    uint64_t seed;
    /* Adaptive sampling (RenderPixel's second loop, main.cpp:245-258), on when max_spp > spp: after `spp` samples with
     * +-0.5 pixel jitter, samples with +-1 pixel jitter are added until the variance of the samples BEFORE the newest is
     * <= variance_threshold or max_spp is reached (the reference hard-codes 10, 50 and 0.01).  The reference's quirks are
     * kept: the newest sample is added to the sum but the sum is divided by the count without it when the rule stops the
     * loop.  One RNG stream per pixel, seeded with prt_sample_key(seed, pixel, 0), as the loop needs; the fixed mode
     * (max_spp <= spp) reseeds per sample.  0 / 0.0f = off / the reference's threshold.
     * The stopping rule compares a float variance of the pixel's sample colours with the threshold.  The device's sample
     * colours are reproducible bit for bit (fixed-point accumulation, see "Determinism" in DESIGN.md) but differ from the CPU
     * reference's in the last bits (device powf, throughput form of the colour polynomial), so a variance that lands within
     * ~1e-6 of the threshold can stop a pixel one sample earlier or later than the reference would.  That is the ONLY way the
     * two can part, and it is counted: prt_render_stats::variance_close_calls is the number of verdicts whose variance lay within
     * 0.1 % of the threshold (a band some fifty times wider than the colours' last bits can move a variance).  0 - as on three of
     * the four adaptive fixtures; the fourth has 2 such verdicts among ~10^5 and equal counts all the same - means every pixel
     * stopped at the reference's sample and ray_count equals the reference's, as it does by construction for fixed spp. */
    uint32_t max_spp;
    float variance_threshold;
} prt_params;

/* DEFAULT = POOL: one launch, wave-private ray pools; the faster pipeline on everything measured except a full 1080p frame
 * of the 1M-triangle scene, where WAVEFRONT (one launch per bounce round, global ray queues) is level or 2 % ahead.
 * With PRT_FLAG_TRYOUT the first DEFAULT call of a (scene, pixel set, sampling) configuration above 1 M samples renders
 * the frame (its first 32 M samples if it is larger than 64 M) with both, twice each, and the context keeps WAVEFRONT if
 * POOL is not at least 3 % faster - worth it for many frames of one configuration, not for a single one.  Adaptive
 * sampling always runs on POOL.  All pipelines produce the same image (tests/test_gpu_parity.py); prt_counters.pipeline
 * reports which ran. */
/* The shipped library has the two production pipelines, WAVEFRONT and POOL.  MEGAKERNEL (round 1's first version, which keeps
 * the reference's float association exactly) and PERSISTENT (a measured negative result) exist only in a library built with
 * -DPRT_EXPERIMENTAL (make hip-experimental); elsewhere asking for them is an error. */
enum { PRT_PIPELINE_DEFAULT = 0, PRT_PIPELINE_MEGAKERNEL = 1, PRT_PIPELINE_WAVEFRONT = 2, PRT_PIPELINE_PERSISTENT = 3,
       PRT_PIPELINE_POOL = 4, PRT_PIPELINE_MASK = 0xFF };
/* OR-ed into prt_params.pipeline: also count BVH node visits and triangle tests (costs a few percent;
 * ray_count and shaded_hits are always counted). */
enum { PRT_FLAG_COUNT_VISITS = 0x100, PRT_FLAG_TRYOUT = 0x200 };

/* DebugCounters (globals.h:3-7) re-cast for a per-triangle BVH.  ray_count has the reference's
 * meaning (one per TraceRay call, raytracer.cpp:161) and must equal the CPU value exactly.  It includes the shadow rays
 * the device does not trace because the radiance that would ride with them is exactly zero (no outcome could change the
 * image; prt_render_stats.elided_shadow_rays says how many): the reference casts and counts those too. */
typedef struct prt_counters {
    uint64_t ray_count;
    uint64_t node_visits;          /* BVH nodes fetched (replaces sphere_check_count) */
    uint64_t tri_tests;            /* triangle tests (the reference tests every triangle of a leaf group) */
    uint64_t shaded_hits;          /* closest-hit shading events */
    double render_ms;              /* ray generation -> resolved framebuffer, device time */
    double trace_kernel_ms;        /* time inside the dominant (traversal) kernel(s) */
    uint32_t trace_kernel_launches;
    uint32_t pipeline;             /* the PRT_PIPELINE_* that ran (what PRT_PIPELINE_DEFAULT resolved to) */
} prt_counters;

typedef struct prt_ctx prt_ctx;

/* Lifecycle.  device_id is a HIP device ordinal.  (A context is one device; prt_multi_* below is the n-device form of
 * SURVEY.md 8(b)'s prt_create(const int * device_ids, int n_dev).) */
prt_ctx * prt_create(int device_id);
void prt_destroy(prt_ctx * ctx);
const char * prt_last_error(const prt_ctx * ctx);   /* ctx may be NULL: last creation error */
int prt_abi_version(void);

/* Options.  Every tuning / test knob of the library is an entry of the context's option table (csrc/prt_options.h).
 * prt_create reads the environment ONCE - PRT_<NAME>=value - and nothing on the upload or render path reads it again;
 * prt_set_option changes an entry afterwards (name with or without the PRT_ prefix, any case; value NULL = default).
 * Three entries change what a render does rather than how fast it is:
 *   TRACE_DEAD_SHADOW_RAYS  0 (default): a shadow ray whose radiance-if-unoccluded is exactly zero is counted in ray_count but
 *                           not traced (no outcome could change the image); 1: traced all the same
 *   BVH_BUILDER             "sah" (default): binned-SAH build on the host; "lbvh": radix tree built on the GPU (applies to the
 *                           next prt_upload_scene)
 *   POOL_EXACT              1: the pool pipeline's slow-path kernel renders everything (test hook; same image)
 * Returns 0, or -1 (unknown name, value that does not parse, RESERVE_CUS after creation). */
int prt_set_option(prt_ctx * ctx, const char * name, const char * value);

/* How the library was built: PRT_BUILD_EXPERIMENTAL - the MEGAKERNEL / PERSISTENT pipelines are present;
 * PRT_BUILD_BVH4 - the 4-wide sorted BVH traversal (the default build; clear in a -DPRT_BVH8 library, which traverses the
 * 8-wide compressed tree instead: same results). */
enum { PRT_BUILD_EXPERIMENTAL = 1, PRT_BUILD_BVH4 = 2 };
int prt_build_flags(void);

/* Copies the scene to the device, builds the per-triangle BVH and the sampler tables.  At most 2^26 - 1 triangles (the
 * traversal addresses nodes and triangle records by 32-bit byte offsets); more is an error, as is any index out of range. */
int prt_upload_scene(prt_ctx * ctx, const prt_scene_desc * scene);

/* Replaces RenderTask (main.cpp:267-283): pixels [start_idx, end_idx) of a w x h image, 16 B/pixel,
 * row-major, w component = 1.  rgba_out is a HOST pointer of (end_idx-start_idx)*4 floats.
 * An empty request (start_idx == end_idx here, a shard that owns no rows, a pixel list of length 0) is not an error:
 * nothing is rendered, the counters come back zero and the output pointer may be NULL. */
int prt_render(prt_ctx * ctx, const prt_camera * cam, const prt_params * params,
               uint32_t width, uint32_t height, uint32_t start_idx, uint32_t end_idx,
               float * rgba_out, prt_counters * counters);

/* Same, but d_rgba_out is a DEVICE pointer on the context's device (hipMalloc'ed or a torch tensor's
 * data_ptr): used by the multi-GPU path, which gathers the shards with RCCL afterwards.  The call
 * returns after the context's stream has drained, so the buffer may be consumed on any stream.  The library's
 * streams are NOT ordered against the caller's: work of the caller that still touches the buffer (a fill queued on
 * another stream, a collective reading the previous frame) must have finished - or be waited for - before the call. */
int prt_render_device(prt_ctx * ctx, const prt_camera * cam, const prt_params * params,
                      uint32_t width, uint32_t height, uint32_t start_idx, uint32_t end_idx,
                      void * d_rgba_out, prt_counters * counters);

/* Interleaved scan-line-block sharding: rank r of n renders the blocks of
 * `block_rows` rows whose block index % n == r, packed densely into d_rgba_out in ascending row
 * order, in ONE launch.  prt_shard_rows reports how many rows that is. */
uint32_t prt_shard_rows(uint32_t height, uint32_t block_rows, uint32_t rank, uint32_t nranks);
int prt_render_shard_device(prt_ctx * ctx, const prt_camera * cam, const prt_params * params,
                            uint32_t width, uint32_t height, uint32_t block_rows, uint32_t rank,
                            uint32_t nranks, void * d_rgba_out, prt_counters * counters);
/* Host-output variant of the same (used by the C++ driver's multi-GPU Render()). */
int prt_render_shard(prt_ctx * ctx, const prt_camera * cam, const prt_params * params,
                     uint32_t width, uint32_t height, uint32_t block_rows, uint32_t rank,
                     uint32_t nranks, float * rgba_out, prt_counters * counters);

/* Arbitrary pixel subset: pixel_ids are linear indices (y*width + x), rgba_out (HOST) gets n_pixels
 * float4 in list order.  Pixels are independently seeded, so any subset reproduces the same values as
 * the full frame; the parity tests use it to render exactly the sparse lattice a CPU fixture holds. */
int prt_render_pixel_list(prt_ctx * ctx, const prt_camera * cam, const prt_params * params,
                          uint32_t width, uint32_t height, const uint32_t * pixel_ids, uint32_t n_pixels,
                          float * rgba_out, prt_counters * counters);

/* ---- several devices behind one handle --------------------------------------------------------------------------------
 * SURVEY.md 8(b)'s prt_create(const int * device_ids, int n_dev): what the host mirror's Render() uses for n GPUs, in place
 * of the reference's one-rank-per-core partition + MPI_Gather (main.cpp:311-347).  The scene is replicated (as every MPI rank
 * holds it); device g renders the 8-row blocks b with b % n == g into a packed buffer in its own HBM; the shards go to
 * device_ids[0] as peer-to-peer copies (DMA engines over xGMI: no compute unit, so no contest with the persistent render
 * kernels), a small kernel there puts the rows in place, and one copy takes the frame to the host buffer rgba_out
 * (width * height * 4 floats).  The frame is bit-identical to what prt_render gives on one device.  The same ordinal may be
 * listed more than once (rehearsal of the n-device path on one GPU: the peer copy is then a copy within the device).
 * prt_multi_context(m, i) is device i's ordinary context (options, scene info, stats).
 *
 * Frames in flight (ABI 5).  A persistent render kernel leaves its GPU partly idle while its last rays drain, and the copies,
 * assembly and download of frame k need no compute unit that frame k + 1 could not use.  The handle therefore has
 * prt_multi_depth() lanes (2 unless the environment says PRT_MULTI_DEPTH=1..4 at creation): a lane is one context per device
 * - lane 0 the contexts above, the others clones that share the uploaded scene's device arrays and own only their streams
 * and workspaces - with worker threads that live as long as the handle (one per lane and device; nothing is created per
 * frame).  prt_multi_submit hands a frame to a free lane and returns at once with a ticket (PRT_ERR_IN_FLIGHT if every lane
 * is busy); the lane renders, and its shards travel to device 0, while the caller does something else - submits the next
 * frame, say; prt_multi_wait(ticket) blocks until that frame is assembled and in rgba_out (which must stay valid until
 * then) and frees the lane.  Tickets may be waited for in any order.  prt_multi_render is submit + wait.  All calls on one
 * handle come from one host thread at a time. */
typedef struct prt_multi prt_multi;
prt_multi * prt_multi_create(const int * device_ids, int n_dev);
void prt_multi_destroy(prt_multi * m);                            /* waits for frames still in flight */
const char * prt_multi_last_error(const prt_multi * m);           /* m may be NULL: last creation error */
int prt_multi_device_count(const prt_multi * m);
int prt_multi_depth(const prt_multi * m);                         /* frames that can be in flight at once */
prt_ctx * prt_multi_context(prt_multi * m, int i);
int prt_multi_upload_scene(prt_multi * m, const prt_scene_desc * scene);   /* not while a frame is in flight */
int prt_multi_render(prt_multi * m, const prt_camera * cam, const prt_params * params, uint32_t width, uint32_t height,
                     float * rgba_out, prt_counters * counters);
int prt_multi_submit(prt_multi * m, const prt_camera * cam, const prt_params * params, uint32_t width, uint32_t height,
                     float * rgba_out, uint64_t * ticket);
int prt_multi_wait(prt_multi * m, uint64_t ticket, prt_counters * counters);

/* Introspection for DESIGN.md / bench.py: sizes of what upload built. */
typedef struct prt_scene_info {
    uint32_t triangle_count;
    uint32_t bvh_node_count;
    uint32_t bvh_max_depth;
    uint32_t bvh_node_bytes;       /* bytes fetched per node visit */
    uint32_t tri_record_bytes;     /* bytes fetched per triangle test */
    uint32_t shade_record_bytes;   /* bytes fetched per shaded hit */
    uint64_t device_bytes;         /* total resident scene bytes */
    double bvh_build_ms;
} prt_scene_info;
int prt_get_scene_info(const prt_ctx * ctx, prt_scene_info * info);

/* Diagnostics of the context's last render call made with PRT_FLAG_COUNT_VISITS (bench.py's roofline block):
 * wave-level step counts - lane utilisation of a loop = lane-level count / (64 x wave-level count) -, where the waves of the
 * pool pipeline spent their time, and how many rays took the slow path (near-tied hits, include/prt.h "Determinism"). */
typedef struct prt_render_stats {
    uint64_t node_visits, tri_tests;           /* lane level, as in prt_counters */
    uint64_t wave_node_steps, wave_tri_steps, wave_leaf_visits, wave_refills;
    uint64_t deepest_stack;                    /* entries of a traversal stack column ever in use */
    uint64_t phase_cycles[5];                  /* pool pipeline: top-up, trace, shade, whole main loop, adaptive finalise step */
    uint64_t parked_rays, parked_shadow_rays;  /* pool pipeline: most rays any pass handed to its slow launches */
    uint64_t elided_shadow_rays;               /* shadow rays that are in ray_count but were not traced: their radiance-if-unoccluded
                                                * was exactly zero, so no outcome could change the image (option
                                                * TRACE_DEAD_SHADOW_RAYS=1 traces them all the same) */
    uint64_t variance_close_calls;             /* adaptive mode: verdicts of the stopping rule whose variance lay within 0.1 % of the
                                                * threshold.  0 = every pixel stopped at the reference's sample (see prt_params) */
    uint32_t stack_lds_entries, stack_bound;   /* LDS stack column height used, worst-case bound of the tree */
} prt_render_stats;
int prt_get_render_stats(const prt_ctx * ctx, prt_render_stats * stats);

/* Host-only self check of the acceleration structure prt_upload_scene builds (runs without a GPU; the
 * CPU test-suite calls it): every triangle inside every ancestor's de-quantised box, every triangle in
 * exactly one leaf, links in range.  out[6] = { violations, nodes, depth, stack bound, leaves, triangle refs }. */
int prt_debug_check_bvh(const prt_scene_desc * scene, uint64_t * out);
/* The same check on the tree of the GPU LBVH builder (option BVH_BUILDER=lbvh at upload: radix tree built on the
 * device, bvh_lbvh.h; an alternative for scenes that change every frame - ~10x faster to build, slower to traverse). */
int prt_debug_check_bvh_lbvh(prt_ctx * ctx, const prt_scene_desc * scene, uint64_t * out);

/* Device known-answer hook (GPU test-suite): runs one device function of the hot path on `n` caller-supplied
 * records (host pointers) and returns its outputs, so tests can compare them bit for bit with the reference's.
 * kinds and record layouts: csrc/kernels_debug.h.  cam may be NULL except for the camera-ray kind; a scene
 * must be uploaded (the diffuse-direction kind reads its table). */
int prt_debug_device_kat(prt_ctx * ctx, int kind, const void * in, size_t in_bytes, void * out, size_t out_bytes,
                         uint32_t n, const prt_camera * cam);

/* Test hook of the exception guard every entry point runs under ("nothing aborts"): throws, inside a guarded entry point,
 * 1 std::bad_alloc, 2 a real std::length_error (a vector asked for more than max_size), 3 std::runtime_error, 4 an int -
 * and returns what the caller of any entry point would get: PRT_ERR_EXCEPTION, with the message in prt_last_error(ctx).
 * kind 0 returns 0.  ctx may be NULL (no GPU needed). */
int prt_debug_throw(prt_ctx * ctx, int kind);

#ifdef __cplusplus
}
#endif
#endif /* PRT_H_ */
