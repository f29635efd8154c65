// prt_api.hip - implementation of the C ABI in include/prt.h: context, scene upload, render dispatch.
//
// One prt_ctx = one HIP device + one stream + one resident scene.  Upload flattens the scene into the
// records of dev_scene.h and builds the per-triangle BVH on the host (bvh_build.cpp).  Render launches
// the selected pipeline on the context's stream; nothing in the render path allocates when the
// workspace from a previous call is large enough.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/prt.h"
#include "bvh_build.h"
#include "prt_options.h"
#include "dev_scene.h"
#include "kernels_wave.h"
#include "kernels_pool.h"
#if defined(PRT_FLOW_EXPERIMENT)
// round 4's round-free experiment (three tracer waves + one shading wave per workgroup): correct, and slower than the round-based
// kernel because one shading wave per workgroup cannot keep up (profiles/r04_flow_kernel.txt); not in the shipped library
#include "kernels_flow.h"
#endif
#include "kernels_resolve.h"
#if defined(PRT_EXPERIMENTAL)
// the round-1 megakernel (the exact-association cross-check) and the persistent single-launch experiment: not in the
// shipped library; `make hip-experimental` builds a library that has them (tests/test_gpu_parity.py uses it when it is there)
#include "kernels_mega.h"
#include "kernels_persist.h"
#endif
#include "bvh_lbvh.h"
#include "kernels_debug.h"

using namespace prt;

#ifndef PRT_ADAPT_WAVES
#define PRT_ADAPT_WAVES 4         // waves per SIMD of the untextured adaptive kernel (128 VGPRs)
#endif
#ifndef PRT_POOL_SHARED_DEFAULT
#define PRT_POOL_SHARED_DEFAULT 1 // block-shared pools (kernels_pool.h) for fixed-spp renders when the option POOL_SHARED is not set
#endif
#ifndef PRT_POOL_EXCHANGE_DEFAULT
#define PRT_POOL_EXCHANGE_DEFAULT 0   // block-shared pools: rays a wave hands over at the end of a round when the option POOL_EXCHANGE is not set (0: none)
#endif
#ifndef PRT_POOL_FLOW_DEFAULT
#define PRT_POOL_FLOW_DEFAULT 0   // the pool pipeline without rounds (kernels_flow.h) for fixed-spp renders when the option POOL_FLOW is not set
#endif
#ifndef PRT_POOL_BLOCK
#define PRT_POOL_BLOCK 256        // threads per workgroup of the fixed-spp pool kernel (experiments: 320, 640 with block-shared pools)
#endif
#ifndef PRT_DEEP_WAVES
#define PRT_DEEP_WAVES PRT_POOL_WAVES   // waves per SIMD of the fixed-spp kernel for bounce trees of more than 15 draws (C5)
#endif
#ifndef PRT_POOL_WAVES
#define PRT_POOL_WAVES 5          // waves per SIMD of the fixed-spp pool kernel: 96 VGPRs (6 = 80 VGPRs spills, profiles/r03_ab_bvh8.txt)
#endif

// The acceleration structure the library is built with (dev_trace.h): the 4-wide sorted tree, or -DPRT_BVH8 the 8-wide one
// of round 3.
#if !defined(PRT_BVH8)
typedef Bvh4Result BvhWide;
#define PRT_BUILD_WIDE build_bvh4q
#define PRT_BUILD_WIDE_FROM_RADIX build_bvh4q_from_radix_tree
#else
typedef Bvh8Result BvhWide;
#define PRT_BUILD_WIDE build_bvh8q
#define PRT_BUILD_WIDE_FROM_RADIX build_bvh8q_from_radix_tree
#endif

namespace {

std::string g_create_error;

#define HIP_TRY(ctx, call)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            (ctx)->error = std::string(#call) + ": " + hipGetErrorString(e_);                                \
            return -10;                                                                                      \
        }                                                                                                    \
    } while (0)

template <typename T>
struct DevBuf {
    T * p = nullptr;
    size_t n = 0;
    bool borrowed = false;               // p belongs to another DevBuf (clone_context: the scene arrays of the context cloned)
    void borrow(const DevBuf & o) { release(); p = o.p; n = o.n; borrowed = o.p != nullptr; }
    hipError_t ensure(size_t count) {
        if (count <= n && p) return hipSuccess;
        if (p && !borrowed) (void)hipFree(p);
        borrowed = false;
        p = nullptr;
        n = 0;
        hipError_t e = hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    hipError_t upload(const std::vector<T> & v) {
        hipError_t e = ensure(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    void release() {
        if (p && !borrowed) (void)hipFree(p);
        borrowed = false;
        p = nullptr;
        n = 0;
    }
    size_t bytes() const { return n * sizeof(T); }
};

}  // namespace

enum { PRT_MAX_CHAINS = 8 };

struct prt_ctx {
    int device = 0;
    PrtOptions opt;                       // every knob, read from the environment once at creation (prt_options.h, prt_set_option)
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
    std::string error;
    bool has_scene = false;

    DevBuf<float4> nodes, tris, shade, diffuse_dirs, spec_dirs;
    DevBuf<unsigned int> tri_rank;
    DevBuf<DevMaterial> materials;
    DevBuf<DevLight> lights;
    DevScene scene;
    prt_scene_info info;
    float scene_abs_max = 0.0f;
    bool any_translucent = false;
    bool point_lights = false;            // any point light: its shadow rays use the hit distance (raytracer.cpp:396), so they can be near-tied too
    bool textured = false;                // any material has a texture map: the TEX kernel variants run
    // PRT_PIPELINE_DEFAULT: which pipeline won the try-out for a (scene, pixel set, sampling) configuration
    struct TuneEntry { uint64_t key[6]; unsigned int pipeline; };
    std::vector<TuneEntry> tuned;
    uint64_t scene_epoch = 0;
    DevBuf<DevTexture> textures;
    DevBuf<unsigned int> texels;
    DevBuf<float> srgb_lut;
    DevBuf<float4> tri_uv, tri_tan;
    DevBuf<float4> ref_spheres;           // the reference's sphere tree for near-tie resolution (dev_trace_common.h RefSphereWalk)
    std::vector<float> material_ns;      // for rebuilding spec_dirs when spec_samples changes
    unsigned int spec_table_samples = 0;

    // render workspace
    DevBuf<float4> sample_rgb;
    DevBuf<float4> frame_out;
    DevBuf<DevCounters> counters;
    DevBuf<unsigned long long> ring_ws;
    DevBuf<unsigned int> pixel_list;
    // wavefront pipeline workspace
    // one set per chain: the frame is split into two halves that run their rounds on two streams, so that one
    // half's (latency-bound) k_shade overlaps the other half's (issue-bound) k_trace
    struct ChainWs {
        DevBuf<float4> f4;                // one slab carved into the float4 arrays of WaveBuffers
        DevBuf<ulonglong2> rng;
        DevBuf<unsigned int> counts;
        DevBuf<unsigned int> overflow;
        DevBuf<int> slow_stack;
        hipStream_t stream = nullptr;
        hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr, ev_done = nullptr, ev_first = nullptr;
        unsigned int * host_counts = nullptr;   // pinned
    } chain[PRT_MAX_CHAINS];
    DevBuf<unsigned int> wf_counts;       // persistent / pool pipelines' sample counter
    DevBuf<float4> pool_f4;               // pool pipeline: the waves' private ray lists
    prt_render_stats last_stats;          // of the last render call (prt_get_render_stats)
    DevBuf<float4> pool_park;             // pool pipeline: rays parked for the slow launches (kernels_pool.h PoolBuffers::park)
    size_t pool_park_cap = 1u << 18, pool_spark_cap = 1u << 18;     // entries; enlarged when a frame needed more (render_pixels)
    DevCounters * host_counters = nullptr;                            // pinned: the render's counters arrive here on the context's stream
    DevBuf<unsigned int> pool_fin;        // adaptive mode: per-wave lists of pixels to finalise
    DevBuf<PoolArgs> pool_args;           // k_pool's arguments (read per phase from memory, kernels_pool.h)
#if defined(PRT_FLOW_EXPERIMENT)
    DevBuf<FlowArgs> flow_args;           // k_flow's (kernels_flow.h)
    DevBuf<unsigned int> flow_u32;        // ... its slot id rings and free-slot stacks
#endif
    DevBuf<float> pool_xchg;                // block-shared pools: the rays waves hand to each other at the end of a round (PoolBuffers::xchg)
    DevBuf<unsigned long long> wave_times;  // DEBUG_UTIL + counting render: (start, counter dry, exit) wall clock of every wave of the fast kernel
    unsigned int wave_times_n = 0;
    DevBuf<float4> adapt_f4;              // adaptive mode: scratch [max_spp][n] + running sums [n] + final colours [n]
    int cu_count = 0;                     // compute units this context may fill (all of the device's minus PRT_RESERVE_CUS)
    int reserved_cus = 0;
    unsigned int stack_bound = 0;
    DevBuf<int> stack_spill;
};

namespace {

f3 ld3(const float * p) { return mk3(p[0], p[1], p[2]); }

// Dwords of `entries` traversal-stack entries for `lanes` lanes: the ONE place that knows how many dwords an entry of this
// build's traversal has (STACK_ENTRY_INTS: 1 for the 4-wide tree's links, 2 for the 8-wide tree's (base, masks) pairs).
// Every stack area - the LDS columns, their spill columns, the exact kernels' full-height global columns - is sized here.
size_t stack_dwords(size_t entries, size_t lanes) { return entries * lanes * (size_t)STACK_ENTRY_INTS; }


// raytracer.cpp:273-288 on the host.
float radical_inverse_vdc(uint32_t bits) {
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    return (float)(bits * 2.3283064365386963e-10);
}

const float kPi32 = 3.1415927f;

// Tangent-space direction of GetDiffuseReflectionRay for Xi = Hammersley(i, 1024) (raytracer.cpp:322-328).
// Host libm: the same cosf / sinf the reference's CPU path calls.
float4 diffuse_tangent_dir(uint32_t i) {
    float xi_x = (float)i / (float)1024u;
    float xi_y = radical_inverse_vdc(i);
    float phi = xi_y * 2.0f * kPi32;
    float cp = cosf(phi);
    float sp = sinf(phi);
    float ct = sqrtf(1.0f - xi_x);
    float st = sqrtf(1.0f - ct * ct);
    return make_float4(cp * st, sp * st, ct, 0.0f);
}

// ImportanceSamplePhong(Hammersley(samp, count), e) (raytracer.cpp:290-300).
float4 phong_tangent_dir(uint32_t samp, uint32_t count, float e) {
    float xi_x = (float)samp / (float)count;
    float xi_y = radical_inverse_vdc(samp);
    float phi = 2.0f * kPi32 * xi_x;
    float cp = cosf(phi);
    float sp = sinf(phi);
    float ct = powf(1.0f - xi_y, 1.0f / (e + 1.0f));
    float st = sqrtf(1.0f - (ct * ct));
    return make_float4(cp * st, sp * st, ct, 0.0f);
}

// Worst-case RNG draws of one sample: 2 jitter draws + the bounce tree (raytracer.cpp:416-417, 520).
uint64_t max_rng_draws(uint32_t depth, uint32_t refl, uint32_t spec) {
    // node(i): draws inside a surviving invocation with iters = i, child(i) = 1 roulette draw + node(i)
    uint64_t node = 0;                                // iters = 0: no children
    for (uint32_t i = 1; i <= depth; ++i) {
        uint64_t child = 1 + node;
        node = (uint64_t)refl * (1 + child) + (uint64_t)spec * child;
        if (node > (1ull << 40)) break;
    }
    return 2 + node;
}

int build_spec_table(prt_ctx * ctx, unsigned int spec_samples) {
    unsigned int n = std::max(1u, spec_samples);
    std::vector<float4> table(ctx->material_ns.size() * (size_t)n);
    for (size_t m = 0; m < ctx->material_ns.size(); ++m)
        for (unsigned int s = 0; s < n; ++s) table[m * n + s] = phong_tangent_dir(s, n, ctx->material_ns[m]);
    HIP_TRY(ctx, ctx->spec_dirs.upload(table));
    ctx->spec_table_samples = n;
    ctx->scene.spec_dirs = ctx->spec_dirs.p;
    ctx->scene.spec_samples = n;
    return 0;
}

struct PixelSet {
    uint32_t n_pixels;
    uint32_t first_pixel;                      // contiguous range when nranks <= 1
    uint32_t block_rows, rank, nranks;         // interleaved row blocks otherwise
    const unsigned int * d_pixel_list;         // explicit list (device pointer) or NULL
};

#if defined(PRT_EXPERIMENTAL)
template <int MAXLEV, bool RING>
void launch_mega(prt_ctx * ctx, bool count, unsigned int grid, size_t lds, const DevCamera & cam, const DevParams & P,
                 unsigned int n_samples) {
    constexpr int BLOCK = 256;
    if (count)
        hipLaunchKernelGGL((k_render_mega<BLOCK, MAXLEV, RING, true>), dim3(grid), dim3(BLOCK), lds, ctx->stream, ctx->scene, cam, P,
                           n_samples, ctx->sample_rgb.p, ctx->counters.p, ctx->ring_ws.p);
    else
        hipLaunchKernelGGL((k_render_mega<BLOCK, MAXLEV, RING, false>), dim3(grid), dim3(BLOCK), lds, ctx->stream, ctx->scene, cam, P,
                           n_samples, ctx->sample_rgb.p, ctx->counters.p, ctx->ring_ws.p);
}

template <int MAXLEV, bool RING>
int launch_persistent(prt_ctx * ctx, bool count, size_t lds, const DevCamera & cam, const DevParams & P, unsigned int n_samples,
                      int keep_min, int node_min, int blocks_cap) {
    constexpr int BLOCK = 256;
    int per_cu = 0;
    hipError_t oe = count ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_render_persistent<BLOCK, MAXLEV, RING, true>, BLOCK, lds)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_render_persistent<BLOCK, MAXLEV, RING, false>, BLOCK, lds);
    if (oe != hipSuccess || per_cu < 1) per_cu = 2;
    per_cu = std::min(per_cu, blocks_cap);
    const unsigned int max_blocks = (unsigned int)per_cu * (unsigned int)ctx->cu_count;
    const unsigned int grid = std::max(1u, std::min(max_blocks, (n_samples + BLOCK - 1) / BLOCK));
    // samples reserved per head atomic: whole waves, 64..512, ~1/16 of a wave's fair share
    unsigned int chunk = n_samples / (grid * (BLOCK / 64) * 16u);
    chunk = std::max(64u, std::min(512u, (chunk / 64u) * 64u));
    if (RING) {
        hipError_t e = ctx->ring_ws.ensure((size_t)16 * grid * BLOCK);
        if (e != hipSuccess) { ctx->error = std::string("ring workspace: ") + hipGetErrorString(e); return -10; }
    }
    if (count)
        hipLaunchKernelGGL((k_render_persistent<BLOCK, MAXLEV, RING, true>), dim3(grid), dim3(BLOCK), lds, ctx->stream, ctx->scene, cam, P,
                           n_samples, ctx->sample_rgb.p, ctx->counters.p, ctx->ring_ws.p, ctx->wf_counts.p, keep_min, node_min, chunk);
    else
        hipLaunchKernelGGL((k_render_persistent<BLOCK, MAXLEV, RING, false>), dim3(grid), dim3(BLOCK), lds, ctx->stream, ctx->scene, cam, P,
                           n_samples, ctx->sample_rgb.p, ctx->counters.p, ctx->ring_ws.p, ctx->wf_counts.p, keep_min, node_min, chunk);
    return 0;
}

#endif  // PRT_EXPERIMENTAL

// The wavefront pipeline (kernels_wave.h): raygen, then rounds of {persistent trace, shade} until no ray is
// left.  Queue sizes come back to the host once per round (8 bytes, pinned): they size the next launches, end
// the loop and sum to ray_count (every queued ray is one TraceRay call, raytracer.cpp:161).
//
// Large frames are split into two CHAINS (sample halves) that run their rounds on two streams, the second chain
// starting when the first has finished its first trace: from then on one chain's k_shade (latency / HBM bound)
// tends to run beside the other chain's k_trace (vector-issue bound).  The persistent trace grid is capped so
// that a shade workgroup still fits on every CU; k_shade uses 256-thread workgroups in this mode for the same
// reason.  One host thread drives both chains, waiting on whichever round finishes next.
struct Chain {
    WaveBuffers B;
    DevParams P;
    prt_ctx::ChainWs * ws;
    unsigned int n_closest = 0, n_shadow = 0, round = 0, launches = 0;
    int cur = 0;
    bool active = false;
    unsigned long long rays = 0;
    float trace_ms = 0.0f;
};

int chain_setup(prt_ctx * ctx, Chain & c, int index, const DevParams & P, bool ring, unsigned int base, unsigned int n_samples) {
    prt_ctx::ChainWs & w = ctx->chain[index];
    c.ws = &w;
    c.P = P;
    const size_t N = n_samples;
    const unsigned int levels = std::max(1u, P.bounce_depth);
    const unsigned int fr4 = ctx->textured ? 7u : ring ? 5u : 4u;
    const unsigned int n_lights = std::max(1u, ctx->scene.light_count);
    // float4 slab: frames + 2x3 closest queues + hits + 3 shadow arrays (accum aliases the shared sample_rgb)
    const size_t f4_total = (size_t)levels * fr4 * N + 6 * N + N + 3 * N * n_lights;
    HIP_TRY(ctx, w.f4.ensure(f4_total));
    HIP_TRY(ctx, w.rng.ensure(ring ? 2 * N : N));
    HIP_TRY(ctx, w.counts.ensure(16));
    HIP_TRY(ctx, w.overflow.ensure(N * (1 + (size_t)n_lights)));
    {
        // k_trace_exact's fixed grid: full-height stack columns for the rays k_trace hands over (near ties, overflowed columns)
        const size_t exact_lanes = 64 * 256;
        HIP_TRY(ctx, w.slow_stack.ensure(stack_dwords(std::max(ctx->stack_bound, 4u), exact_lanes)));
        c.P.exact_stack = w.slow_stack.p;
        c.P.exact_stack_stride = (unsigned int)exact_lanes;
    }
    WaveBuffers & B = c.B;
    memset(&B, 0, sizeof(B));
    B.n_samples = n_samples;
    B.sample_base = base;
    B.accum = reinterpret_cast<Accum *>(ctx->sample_rgb.p) + base;
    B.rng = w.rng.p;
    B.rng_aux = ring ? w.rng.p + N : nullptr;
    B.ring = ring ? ctx->ring_ws.p + (size_t)16 * base : nullptr;   // this chain's own [16][n_samples] block of the shared ring workspace
    B.ring_step = 1;
    B.ring_stride = n_samples;
    float4 * f = w.f4.p;
    B.frames = f; f += (size_t)levels * fr4 * N;
    for (int q = 0; q < 2; ++q) { B.rq_o[q] = f; f += N; B.rq_d[q] = f; f += N; B.rq_t[q] = f; f += N; }
    B.hits = f; f += N;
    B.sq_o = f; f += N * n_lights;
    B.sq_d = f; f += N * n_lights;
    B.sq_c = f; f += N * n_lights;
    B.counts = w.counts.p;
    B.overflow = w.overflow.p;
    c.n_closest = n_samples;
    c.n_shadow = 0;
    c.active = n_samples > 0;
    return 0;
}

struct WaveTuning {
    int per_cu, keep_min, node_min, multi_light, shade_block;
    unsigned int chunk_min;
    bool ring, count_visits;
    size_t lds;
};

// Enqueue one round of a chain: counters reset, trace, overflow re-trace, shade, counts -> pinned host memory.
int chain_issue_round(prt_ctx * ctx, Chain & c, const WaveTuning & t) {
    constexpr int BLOCK = 256;
    hipStream_t stream = c.ws->stream;
    const WaveBuffers & B = c.B;
    c.rays += (unsigned long long)c.n_closest + c.n_shadow;
    HIP_TRY(ctx, hipMemsetAsync(B.counts, 0, 32, stream));
    const unsigned int total = c.n_closest + c.n_shadow;
    const unsigned int max_blocks = (unsigned int)t.per_cu * (unsigned int)ctx->cu_count;
    const unsigned int grid = std::max(1u, std::min(max_blocks, (total + BLOCK - 1) / BLOCK));
    // rays reserved per head atomic: ~1/8 of a wave's fair share, whole waves, 64..512
    unsigned int chunk = total / (grid * (BLOCK / 64) * 8u);
    chunk = std::max(t.chunk_min, std::min(512u, (chunk / 64u) * 64u));
    HIP_TRY(ctx, hipEventRecord(c.ws->ev_t0, stream));
    if (t.count_visits)
        hipLaunchKernelGGL((k_trace<BLOCK, true>), dim3(grid), dim3(BLOCK), t.lds, stream, ctx->scene, c.P, B, c.cur, c.n_closest, c.n_shadow,
                           t.keep_min, t.node_min, chunk, t.multi_light, ctx->counters.p);
    else
        hipLaunchKernelGGL((k_trace<BLOCK, false>), dim3(grid), dim3(BLOCK), t.lds, stream, ctx->scene, c.P, B, c.cur, c.n_closest, c.n_shadow,
                           t.keep_min, t.node_min, chunk, t.multi_light, ctx->counters.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(c.ws->ev_t1, stream));
    if (c.round == 0) HIP_TRY(ctx, hipEventRecord(c.ws->ev_first, stream));
    c.launches++;
    {
        // rays with a near-tied hit; the kernel reads the list length on the device and normally finds it zero
        constexpr unsigned int OVF_BLOCKS = 64;
        if (t.count_visits)
            hipLaunchKernelGGL(k_trace_exact<true>, dim3(OVF_BLOCKS), dim3(256), 0, stream, ctx->scene, c.P, B, c.cur, c.n_closest, t.multi_light, ctx->counters.p);
        else
            hipLaunchKernelGGL(k_trace_exact<false>, dim3(OVF_BLOCKS), dim3(256), 0, stream, ctx->scene, c.P, B, c.cur, c.n_closest, t.multi_light, ctx->counters.p);
        HIP_TRY(ctx, hipGetLastError());
    }
    if (c.n_closest) {
        if (t.shade_block == 1024) {
            const unsigned int sgrid = (c.n_closest + 1023) / 1024;
            if (ctx->textured) hipLaunchKernelGGL((k_shade<true, 1024, true>), dim3(sgrid), dim3(1024), 0, stream, ctx->scene, c.P, B, c.cur, c.n_closest, ctx->counters.p);
            else if (t.ring) hipLaunchKernelGGL((k_shade<true, 1024, false>), dim3(sgrid), dim3(1024), 0, stream, ctx->scene, c.P, B, c.cur, c.n_closest, ctx->counters.p);
            else hipLaunchKernelGGL((k_shade<false, 1024, false>), dim3(sgrid), dim3(1024), 0, stream, ctx->scene, c.P, B, c.cur, c.n_closest, ctx->counters.p);
        } else {
            const unsigned int sgrid = (c.n_closest + 255) / 256;
            if (ctx->textured) hipLaunchKernelGGL((k_shade<true, 256, true>), dim3(sgrid), dim3(256), 0, stream, ctx->scene, c.P, B, c.cur, c.n_closest, ctx->counters.p);
            else if (t.ring) hipLaunchKernelGGL((k_shade<true, 256, false>), dim3(sgrid), dim3(256), 0, stream, ctx->scene, c.P, B, c.cur, c.n_closest, ctx->counters.p);
            else hipLaunchKernelGGL((k_shade<false, 256, false>), dim3(sgrid), dim3(256), 0, stream, ctx->scene, c.P, B, c.cur, c.n_closest, ctx->counters.p);
        }
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipMemcpyAsync(c.ws->host_counts, B.counts, 32, hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipEventRecord(c.ws->ev_done, stream));
    return 0;
}

// Wait for the chain's round in flight; returns with its next queue sizes loaded.
int chain_finish_round(prt_ctx * ctx, Chain & c) {
    HIP_TRY(ctx, hipEventSynchronize(c.ws->ev_done));
    float ms = 0.0f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, c.ws->ev_t0, c.ws->ev_t1));
    c.trace_ms += ms;
    if (ctx->opt.debug_rounds) fprintf(stderr, "[prt] round %u: %u closest + %u shadow rays, k_trace %.3f ms\n", c.round, c.n_closest, c.n_shadow, ms);
    const unsigned int n_lights = std::max(1u, ctx->scene.light_count);
    const unsigned int next_closest = c.n_closest ? c.ws->host_counts[0] : 0;
    const unsigned int next_shadow = c.n_closest ? c.ws->host_counts[1] : 0;
    if (c.n_closest) c.rays += c.ws->host_counts[4];          // shadow rays of this round's hits that were counted, not traced
    if (next_closest > c.B.n_samples || next_shadow > c.B.n_samples * n_lights) { ctx->error = "prt_render: wavefront queue overflow"; return -6; }
    c.n_closest = next_closest;
    c.n_shadow = next_shadow;
    c.cur ^= 1;
    c.round++;
    if (c.round > 100000) { ctx->error = "prt_render: wavefront loop did not terminate"; return -5; }
    c.active = c.n_closest + c.n_shadow > 0;
    return 0;
}

int render_wavefront(prt_ctx * ctx, const DevCamera & cam, const DevParams & P, bool ring, bool count_visits,
                     unsigned int n_samples, size_t lds, unsigned long long * ray_count, float * trace_ms, unsigned int * launches) {
    constexpr int BLOCK = 256;
    WaveTuning t;
    t.lds = lds;
    t.ring = ring;
    t.count_visits = count_visits;
    // persistent grid: as many blocks as are resident
    int per_cu = 0;
    hipError_t oe = count_visits ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace<BLOCK, true>, BLOCK, lds)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace<BLOCK, false>, BLOCK, lds);
    if (oe != hipSuccess || per_cu < 1) per_cu = 2;
    per_cu = std::min(per_cu, 8);
    t.keep_min = 40;
    t.node_min = 32;
    t.chunk_min = 128;
    const PrtOptions & opt = ctx->opt;
    if (opt.chunk_min >= 0) t.chunk_min = (unsigned int)std::max(64ll, std::min(512ll, opt.chunk_min));
    int n_chains = n_samples >= (1u << 23) ? 2 : 1;          // measured +2 % on 16.6 M samples; small frames: not worth the extra launches
    int split_per_cu = 4;
    // tuning knobs for experiments (not part of the ABI)
    if (opt.trace_blocks_per_cu >= 0) { per_cu = std::max(1, std::min(per_cu, (int)opt.trace_blocks_per_cu)); split_per_cu = per_cu; }
    if (opt.keep_min >= 0) t.keep_min = std::max(1, std::min(64, (int)opt.keep_min));
    if (opt.node_min >= 0) t.node_min = std::max(0, std::min(64, (int)opt.node_min));
    if (opt.chains >= 0) n_chains = std::max(1, std::min((int)PRT_MAX_CHAINS, (int)opt.chains));
    // chains start on a pixel and a wave boundary
    const unsigned int unit = P.spp * 64u;
    while (n_chains > 1 && (unsigned long long)unit * n_chains > n_samples) n_chains--;
    t.per_cu = n_chains >= 2 ? std::min(per_cu, split_per_cu) : per_cu;
    t.shade_block = n_chains >= 2 ? 256 : 1024;
    if (opt.shade_block >= 0) t.shade_block = opt.shade_block == 256 ? 256 : 1024;
    t.multi_light = ctx->scene.light_count > 1 ? 1 : 0;

    Chain chain[PRT_MAX_CHAINS];
    int rc = 0;
    {
        unsigned int base = 0;
        const unsigned int per = (unsigned int)((((unsigned long long)n_samples + n_chains - 1) / n_chains + unit - 1) / unit * unit);
        for (int c = 0; c < n_chains; ++c) {
            const unsigned int n = c + 1 == n_chains ? n_samples - base : std::min(per, n_samples - base);
            if ((rc = chain_setup(ctx, chain[c], c, P, ring, base, n))) return rc;
            base += n;
        }
    }

    for (int c = 0; c < n_chains; ++c) {
        hipStream_t st = chain[c].ws->stream;
        if (c >= 1) {
            // the extra streams start after everything already queued on the main stream (scene upload, memsets)
            HIP_TRY(ctx, hipEventRecord(ctx->chain[c].ev_done, ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->chain[c].ev_done, 0));
        }
        const unsigned int gen_grid = (chain[c].B.n_samples + 255) / 256;
        if (gen_grid) {
            if (ring) hipLaunchKernelGGL(k_raygen<true>, dim3(gen_grid), dim3(256), 0, st, cam, chain[c].P, chain[c].B);
            else hipLaunchKernelGGL(k_raygen<false>, dim3(gen_grid), dim3(256), 0, st, cam, chain[c].P, chain[c].B);
            HIP_TRY(ctx, hipGetLastError());
        }
    }
    // first rounds: chain c starts tracing when chain c-1's first trace is done (staggered phases from then on)
    for (int c = 0; c < n_chains; ++c) {
        if (!chain[c].active) continue;
        if (c >= 1 && chain[c - 1].launches) HIP_TRY(ctx, hipStreamWaitEvent(chain[c].ws->stream, chain[c - 1].ws->ev_first, 0));
        if ((rc = chain_issue_round(ctx, chain[c], t))) return rc;
    }
    for (;;) {
        bool any = false;
        for (int c = 0; c < n_chains; ++c) {
            if (!chain[c].active) continue;
            any = true;
            if ((rc = chain_finish_round(ctx, chain[c]))) return rc;
            if (chain[c].active && (rc = chain_issue_round(ctx, chain[c], t))) return rc;
        }
        if (!any) break;
    }
    for (int c = 1; c < n_chains; ++c) {
        // the resolve runs on the main stream: it has to see every chain's results
        HIP_TRY(ctx, hipEventRecord(ctx->chain[c].ev_done, chain[c].ws->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->chain[c].ev_done, 0));
    }
    *ray_count = 0;
    *trace_ms = 0.0f;
    *launches = 0;
    for (int c = 0; c < n_chains; ++c) { *ray_count += chain[c].rays; *trace_ms += chain[c].trace_ms; *launches += chain[c].launches; }
    return 0;
}

// The wave-pool pipeline (kernels_pool.h): one launch; per-sample arrays as in the wavefront pipeline (chain 0's
// workspace, without the global queues), plus cap (+ cap * lights shadow) ray slots per resident wave.
// RINGMEM = 0: no sample can make more than 15 RNG draws (and the scene is opaque, untextured, fixed spp): the
// general-RNG variant runs without its draw ring in memory (dev_rng.h).
template <int BLOCK, int WAVES, bool LDSTAB, bool RING, bool COUNT, bool TEX, bool ADAPT, int RINGMEM, bool EXACT, bool SHARED = false>
int launch_pool_kernel(prt_ctx * ctx, unsigned int grid, size_t lds, const PoolArgs * d_args) {
    hipLaunchKernelGGL((k_pool<BLOCK, WAVES, LDSTAB, RING, COUNT, TEX, ADAPT, RINGMEM, EXACT, SHARED>), dim3(grid), dim3(BLOCK), lds, ctx->stream, d_args, ctx->counters.p);
    HIP_TRY(ctx, hipGetLastError());       // a template variant that cannot launch (LDS, registers) is reported here, by name of its cause
    return 0;
}

template <int BLOCK, int WAVES, bool LDSTAB, bool RING, bool TEX, bool ADAPT, int RINGMEM = 1>
int launch_pool(prt_ctx * ctx, bool count, const DevCamera & cam, DevParams P, unsigned int n_samples, unsigned int stack_entries) {
    // the stack columns double as the shading phase's frame storage (WFRAME_LDS_DWORDS per lane)
    const size_t lds = std::max(stack_dwords(stack_entries, BLOCK), (size_t)((RING && RINGMEM == 1) ? WFRAME_LDS_DWORDS : WFRAME_LDS_DWORDS_NOPOS) * BLOCK) * sizeof(int);
    const PrtOptions & opt = ctx->opt;
    // Three launches (kernels_pool.h PoolBuffers::park): the fast kernel, which parks the rays it cannot finish - a hit with
    // company within a few ulp, a stack column that overflowed -; k_pool_parked_shadows for the parked shadow rays; the EXACT
    // kernel, adopting the parked closest-hit rays.  Normally nothing is parked and the two follow-ups leave at once.
    // Option POOL_EXACT (tests): the EXACT kernel does the whole render.
    const bool exact_only = opt.pool_exact != 0;
    int per_cu = 0;
    // Block-shared pools (kernels_pool.h, round 4): the workgroup owns a pool, not the wave.  Option POOL_SHARED: 1 / 0 force it
    // on / off; the EXACT kernels (the adopting launch, POOL_EXACT) always keep wave-private pools.
    // Default (profiles/r04_ab_shared_pools.txt, same-process A/B): ON for the kernel of the default bounce tree (fixed spp, at
    // most 15 draws per sample, opaque untextured scene: RINGMEM = 0) - the full C4 frame 1.5 - 2 % faster, its 1/2 .. 1/16
    // shards 1.5 - 3.5 % -; OFF for the adaptive mode, whose short rounds lose 11 % to the four waves waiting for each other at
    // every phase boundary, and for deep bounce trees (C5: 886 -> 949 ms; that variant spills 59 dwords shared, 31 private).
    // (the round-free kernel, kernels_flow.h, has its own workgroup-level structure; its follow-up launches use wave-private pools)
#if defined(PRT_FLOW_EXPERIMENT)
    const bool flow = !ADAPT && BLOCK == 256 && !exact_only && (opt.pool_flow >= 0 ? opt.pool_flow != 0 : PRT_POOL_FLOW_DEFAULT != 0);
#else
    const bool flow = false;
#endif
    const bool shared = !flow && !exact_only && (opt.pool_shared >= 0 ? opt.pool_shared != 0 : (PRT_POOL_SHARED_DEFAULT != 0 && !ADAPT && !TEX && RINGMEM == 0));
    hipError_t oe = exact_only ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pool<BLOCK, WAVES, LDSTAB, RING, false, TEX, ADAPT, RINGMEM, true, false>, BLOCK, lds)
                  : shared     ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pool<BLOCK, WAVES, LDSTAB, RING, false, TEX, ADAPT, RINGMEM, false, true>, BLOCK, lds)
                  : count      ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pool<BLOCK, WAVES, LDSTAB, RING, true, TEX, ADAPT, RINGMEM, false, false>, BLOCK, lds)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pool<BLOCK, WAVES, LDSTAB, RING, false, TEX, ADAPT, RINGMEM, false, false>, BLOCK, lds);
    if (oe != hipSuccess || per_cu < 1) per_cu = 1;
    per_cu = std::min(per_cu, 8);
    if (opt.pool_blocks_per_cu >= 0) per_cu = std::max(1, std::min(per_cu, (int)opt.pool_blocks_per_cu));
    if (opt.debug_util) fprintf(stderr, "[prt] k_pool<%d,%d,%d>: %d blocks per CU\n", BLOCK, WAVES, (int)LDSTAB, per_cu);
    const unsigned int max_blocks = (unsigned int)per_cu * (unsigned int)ctx->cu_count;
    const unsigned int grid = std::max(1u, std::min(max_blocks, (n_samples + BLOCK - 1) / BLOCK));
    const unsigned int waves = grid * (BLOCK / 64);
    // slots per wave: half of a wave's fair share of the samples (measured on C4 shards: larger pools leave the
    // waves unbalanced when the sample counter runs out, smaller ones trace at low lane utilisation); whole
    // waves, 64..512
    unsigned int cap = (n_samples / waves / 2u + 63u) / 64u * 64u;
    cap = std::max(64u, std::min(512u, cap));
    if (shared && opt.pool_shared_cap >= 0) cap = (unsigned int)std::max(64ll, std::min(1024ll, opt.pool_shared_cap / 64 * 64));
    // adaptive mode: a pixel stays in its pool for all its samples (up to 50 x several rounds), so what a wave takes it keeps;
    // smaller pools leave more of the frame on the counter for the waves whose pixels end early (C4: 512 slots 195 ms,
    // 256 168 ms, 64 - 192 161 - 164 ms, profiles/r02_adaptive_pool_capacity.txt; round 3, once the variance rule no longer
    // walks the stored samples: 512 160, 256 145, 192 133 - 138, 128 129 - 133, 64 138 ms, profiles/r03_adaptive.txt)
    if (ADAPT) cap = std::min(cap, 128u);
    if (opt.pool_cap >= 0) cap = (unsigned int)std::max(64ll, std::min(4096ll, opt.pool_cap / 64 * 64));
    // experiment POOL_FAIR = k: every wave's pool holds k / 8 of a fair share and takes at most that at once (8 = the whole
    // frame resident from the start, no second generation of samples; profiles/r04_ab_shared_pools.txt section 6)
    unsigned int topup_max = 0xFFFFFFFFu;
    if (opt.pool_fair > 0) {
        const unsigned int fair = (unsigned int)(((unsigned long long)n_samples * (unsigned long long)opt.pool_fair / 8ull + waves - 1u) / waves);
        cap = std::max(64u, std::min(4096u, (fair + 63u) / 64u * 64u));
        topup_max = std::max(1u, fair);
    }
    // block-shared: the same slots, owned by the block's waves together (`cap` is per UNIT from here on)
    const unsigned int units = shared ? grid : waves;
    if (shared) cap *= (unsigned int)(BLOCK / 64);
    const unsigned int n_lights = std::max(1u, ctx->scene.light_count);
    const unsigned int scap = cap * n_lights;

    prt_ctx::ChainWs & w = ctx->chain[0];
    const size_t N = n_samples;
    const unsigned int levels = std::max(1u, P.bounce_depth);
    const unsigned int fr4 = TEX ? 7u : (RING && RINGMEM == 1) ? 5u : 4u;       // kernels_wave.h wframe_save
    HIP_TRY(ctx, w.f4.ensure((size_t)levels * fr4 * N));
    HIP_TRY(ctx, w.rng.ensure(RING ? 2 * N : N));
    HIP_TRY(ctx, ctx->pool_f4.ensure((size_t)units * (7u * (size_t)cap + 3u * (size_t)scap)));
    // [0] sample counter, [1] parked closest-hit / finalise entries, [2] parked shadow rays, [3] the adopting launch's counter
    HIP_TRY(ctx, ctx->wf_counts.ensure(16));          // zeroed by k_pool_store_args below
    // Park lists: sized for what scenes with coincident geometry need in practice, never for the worst case (every sample's
    // ray parked: 48 bytes x samples x (1 + lights)).  The kernels count what they could not store; render_pixels looks at
    // the counts after the frame and, if a list was too short, enlarges it and renders the frame again.
    // The clamps are the true worst cases: a sample is parked at most once (it leaves the pool) plus, in adaptive mode, one
    // deferred finalise step per pixel; parked SHADOW rays pile up until the fast kernel has ended, and every shaded hit of a
    // sample's bounce tree - up to sum over levels of (reflection + specular samples)^level of them - emits one per light.
    size_t hits_per_sample = 1, level_nodes = 1;
    for (unsigned int l = 0; l < P.bounce_depth && hits_per_sample < (1u << 20); ++l) {
        level_nodes *= std::max<size_t>(1, (size_t)P.reflection_samples + P.spec_samples);
        hits_per_sample += level_nodes;
    }
    const size_t worst_spark = N * n_lights * std::min<size_t>(hits_per_sample, 1u << 20);
    const size_t park_cap = std::min<size_t>(ctx->pool_park_cap, N + (ADAPT ? N : 0)), spark_cap = std::min<size_t>(ctx->pool_spark_cap, worst_spark);
    HIP_TRY(ctx, ctx->pool_park.ensure(3 * (park_cap + spark_cap)));
    // the EXACT launch's lanes continue their LDS stack columns in memory; k_pool_parked_shadows has full-height columns
    const unsigned int grid2 = exact_only ? grid : std::max(1u, std::min(grid, (unsigned int)ctx->cu_count));
    const size_t exact_lanes = (size_t)POOL_PARKED_SHADOW_BLOCKS * 256, spill_lanes = (size_t)grid2 * BLOCK;
    const size_t spill_entries = ctx->stack_bound > stack_entries ? ctx->stack_bound - stack_entries : 0;
    const size_t exact_ints = stack_dwords(std::max(ctx->stack_bound, 4u), exact_lanes);
    HIP_TRY(ctx, ctx->stack_spill.ensure(exact_ints + stack_dwords(spill_entries, spill_lanes)));
    P.exact_stack = ctx->stack_spill.p;
    P.exact_stack_stride = (unsigned int)exact_lanes;
    P.stack_spill = spill_entries ? ctx->stack_spill.p + exact_ints : nullptr;
    P.stack_spill_stride = (unsigned int)spill_lanes;

    WaveBuffers B;
    memset(&B, 0, sizeof(B));
    B.n_samples = n_samples;
    B.sample_base = 0;
    B.accum = reinterpret_cast<Accum *>(ctx->sample_rgb.p);
    B.rng = w.rng.p;
    B.rng_aux = RING ? w.rng.p + N : nullptr;
    B.ring = RING && RINGMEM != 0 ? ctx->ring_ws.p : nullptr;
    B.ring_step = ADAPT ? 16u : 1u;                              // WaveBuffers::ring: pixel-major in adaptive mode
    B.ring_stride = ADAPT ? 1u : (unsigned int)n_samples;
    B.frames = w.f4.p;
    PoolBuffers Q;
    memset(&Q, 0, sizeof(Q));
    Q.cq = ctx->pool_f4.p;
    Q.hits = Q.cq + (size_t)units * 6u * cap;
    Q.sq = Q.hits + (size_t)units * cap;
    Q.head = ctx->wf_counts.p;
    Q.cap = cap;
    Q.scap = scap;
    Q.park = ctx->pool_park.p;
    Q.spark = ctx->pool_park.p + 3 * park_cap;
    Q.park_count = ctx->wf_counts.p + 1;
    Q.park_cap = (unsigned int)park_cap;
    Q.spark_cap = (unsigned int)spark_cap;
    Q.adopt = 0;
    Q.fin = nullptr; Q.scratch = nullptr; Q.jobsum = nullptr; Q.final_rgb = nullptr;
    Q.wave_times = nullptr;
    ctx->wave_times_n = 0;
    if (count && opt.debug_util && !exact_only) {
        HIP_TRY(ctx, ctx->wave_times.ensure(3u * (size_t)waves));
        Q.wave_times = ctx->wave_times.p;
        ctx->wave_times_n = waves;
    }
    if (ADAPT) {
        // n_samples counts PIXELS here: the unit in the pool is a pixel that runs its samples one after the other
        HIP_TRY(ctx, ctx->pool_fin.ensure((size_t)units * cap));
        HIP_TRY(ctx, ctx->adapt_f4.ensure(((size_t)P.max_spp + 2u) * N));
        Q.fin = ctx->pool_fin.p;
        Q.scratch = ctx->adapt_f4.p;
        Q.jobsum = Q.scratch + (size_t)P.max_spp * N;
        Q.final_rgb = Q.jobsum + N;
    }
    // guided top-ups: adaptive mode by default (a pixel is a chain of up to max_spp samples: the pixels started last are the frame's tail)
    Q.guided = (unsigned int)std::max(0ll, std::min(64ll, opt.pool_guided >= 0 ? opt.pool_guided : (ADAPT ? 16ll : 0ll)));
    Q.guided_min = (unsigned int)std::max(1ll, std::min(512ll, opt.pool_guided_min >= 0 ? opt.pool_guided_min : 8ll));
    Q.xchg = nullptr; Q.xchg_max = 0;
    {
        const long long xm = opt.pool_exchange >= 0 ? opt.pool_exchange : PRT_POOL_EXCHANGE_DEFAULT;
        if (shared && xm > 0 && PRT_POOL_EXCHANGE_BUILD) {
            Q.xchg_max = (unsigned int)std::min<long long>(xm, POOL_XCHG_MAX);
            HIP_TRY(ctx, ctx->pool_xchg.ensure((size_t)units * (BLOCK / 64) * POOL_XCHG_FIELDS * Q.xchg_max));
            Q.xchg = ctx->pool_xchg.p;
        }
    }
    Q.topup_max = topup_max == 0xFFFFFFFFu ? topup_max : topup_max * (shared ? (unsigned int)(BLOCK / 64) : 1u);
    Q.topup_min = ADAPT ? std::max(64u, cap / 2u) : std::max(64u, cap / 4u);
    if (opt.pool_topup >= 0) Q.topup_min = (unsigned int)std::max(1ll, std::min((long long)cap, opt.pool_topup * (shared ? BLOCK / 64 : 1)));   // the option counts per wave
    int keep_min = 40, node_min = 32;
    if (opt.keep_min >= 0) keep_min = std::max(1, std::min(64, (int)opt.keep_min));
    if (opt.node_min >= 0) node_min = std::max(0, std::min(64, (int)opt.node_min));
    const int multi_light = ctx->scene.light_count > 1 ? 1 : 0;
    PoolArgs A;
    memset(&A, 0, sizeof(A));
    A.sc = ctx->scene;
    A.cam = cam;
    A.P = P;
    A.B = B;
    A.Q = Q;
    A.keep_min = keep_min;
    A.node_min = node_min;
    A.node_frac = 4;
    if (opt.node_frac >= 0) A.node_frac = std::max(0, std::min(8, (int)opt.node_frac));
    A.multi_light = multi_light;
    HIP_TRY(ctx, ctx->pool_args.ensure(2));
#if defined(PRT_FLOW_EXPERIMENT)
    if constexpr (!ADAPT && BLOCK == 256) {
        if (flow) {
            // ---- no rounds (kernels_flow.h): three tracer waves and a shading wave per workgroup, ray slots and rings instead of lists.
            // The follow-up launches are the pool kernel's (their PoolArgs `A` as built above, wave-private layout over the same buffers).
            int fper_cu = 0;
            hipError_t foe = count ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&fper_cu, k_flow<BLOCK, WAVES, RING, true, TEX, RINGMEM>, BLOCK, lds)
                                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&fper_cu, k_flow<BLOCK, WAVES, RING, false, TEX, RINGMEM>, BLOCK, lds);
            if (foe != hipSuccess || fper_cu < 1) fper_cu = 1;
            fper_cu = std::min(fper_cu, 8);
            if (opt.pool_blocks_per_cu >= 0) fper_cu = std::max(1, std::min(fper_cu, (int)opt.pool_blocks_per_cu));
            const unsigned int fgrid = std::max(1u, std::min((unsigned int)fper_cu * (unsigned int)ctx->cu_count, (n_samples + BLOCK - 1) / BLOCK));
            auto pow2 = [](unsigned int v) { unsigned int p = 64; while (p < v) p <<= 1; return p; };
            // samples in flight per workgroup: what the block's waves would hold in their pools
            unsigned int fcap = (n_samples / (fgrid * (BLOCK / 64)) / 2u + 63u) / 64u * 64u;
            fcap = std::max(64u, std::min(512u, fcap));
            if (opt.pool_cap >= 0) fcap = (unsigned int)std::max(64ll, std::min(4096ll, opt.pool_cap / 64 * 64));
            const unsigned int slots = fcap * (BLOCK / 64);
            const unsigned int rc = pow2(slots), rd = pow2(slots), rs = pow2(std::max(1024u, 2u * slots * n_lights));
            HIP_TRY(ctx, ctx->pool_f4.ensure(std::max((size_t)units * (7u * (size_t)cap + 3u * (size_t)scap), (size_t)fgrid * (4u * (size_t)slots + 3u * (size_t)rs))));
            HIP_TRY(ctx, ctx->flow_u32.ensure((size_t)fgrid * ((size_t)rc + 3u * rd + slots)));
            HIP_TRY(ctx, ctx->flow_args.ensure(1));
            // (pool_f4 may have moved: the wave-private layout of the follow-up launches again)
            A.Q.cq = ctx->pool_f4.p;
            A.Q.hits = A.Q.cq + (size_t)units * 6u * cap;
            A.Q.sq = A.Q.hits + (size_t)units * cap;
            FlowArgs FA;
            memset(&FA, 0, sizeof(FA));
            FA.pool = A;
            FA.pool.Q.cq = ctx->pool_f4.p;                                   // [blocks][3][slots]
            FA.pool.Q.hits = FA.pool.Q.cq + (size_t)fgrid * 3u * slots;      // [blocks][slots]
            FA.pool.Q.sq = FA.pool.Q.hits + (size_t)fgrid * slots;           // [blocks][3][rs]
            FA.F.crq = ctx->flow_u32.p;
            FA.F.done = FA.F.crq + (size_t)fgrid * rc;
            FA.F.freelist = FA.F.done + (size_t)fgrid * 3u * rd;
            FA.F.slots = slots;
            FA.F.rc_mask = rc - 1u; FA.F.rd_mask = rd - 1u; FA.F.rs_mask = rs - 1u;
            FA.F.topup_min = 64u;
            FA.F.topup_max = 256u;
            if (opt.pool_topup >= 0) FA.F.topup_min = (unsigned int)std::max(1ll, std::min((long long)slots, opt.pool_topup));
            FA.F.low_water = 3u * 64u;
            FA.F.error = &ctx->counters.p->flow_error;
            if (opt.debug_util) fprintf(stderr, "[prt] k_flow<%d,%d>: %d blocks per CU, %u workgroups, %u slots each, rings %u / %u / %u\n", BLOCK, WAVES, fper_cu, fgrid, slots, rc, rd, rs);
            hipLaunchKernelGGL(k_flow_store_args, dim3(1), dim3(64), 0, ctx->stream, FA, A, ctx->flow_args.p, ctx->pool_args.p, ctx->wf_counts.p, 16u, ctx->wf_counts.p + 3);
            HIP_TRY(ctx, hipGetLastError());
            if (count) hipLaunchKernelGGL((k_flow<BLOCK, WAVES, RING, true, TEX, RINGMEM>), dim3(fgrid), dim3(BLOCK), lds, ctx->stream, ctx->flow_args.p, ctx->counters.p);
            else hipLaunchKernelGGL((k_flow<BLOCK, WAVES, RING, false, TEX, RINGMEM>), dim3(fgrid), dim3(BLOCK), lds, ctx->stream, ctx->flow_args.p, ctx->counters.p);
            HIP_TRY(ctx, hipGetLastError());
            if (count) hipLaunchKernelGGL(k_pool_parked_shadows<true>, dim3(POOL_PARKED_SHADOW_BLOCKS), dim3(256), 0, ctx->stream, ctx->pool_args.p, ctx->counters.p);
            else hipLaunchKernelGGL(k_pool_parked_shadows<false>, dim3(POOL_PARKED_SHADOW_BLOCKS), dim3(256), 0, ctx->stream, ctx->pool_args.p, ctx->counters.p);
            HIP_TRY(ctx, hipGetLastError());
            return count ? launch_pool_kernel<BLOCK, WAVES, LDSTAB, RING, true, TEX, ADAPT, RINGMEM, true>(ctx, grid2, lds, ctx->pool_args.p + 1)
                         : launch_pool_kernel<BLOCK, WAVES, LDSTAB, RING, false, TEX, ADAPT, RINGMEM, true>(ctx, grid2, lds, ctx->pool_args.p + 1);
        }
    }
#endif
    // the adopting launch's units are waves whatever the fast kernel's were: a quarter of a shared pool's slots each, in the
    // same buffers (grid2 <= grid, so its waves' lists fit where the blocks' lists lie)
    hipLaunchKernelGGL(k_pool_store_args, dim3(1), dim3(64), 0, ctx->stream, A, ctx->pool_args.p, ctx->wf_counts.p, 16u,
                       exact_only ? (unsigned int *)nullptr : ctx->wf_counts.p + 3, shared ? (unsigned int)(BLOCK / 64) : 1u);
    HIP_TRY(ctx, hipGetLastError());
    int rc;
    if (exact_only) {
        return count ? launch_pool_kernel<BLOCK, WAVES, LDSTAB, RING, true, TEX, ADAPT, RINGMEM, true>(ctx, grid, lds, ctx->pool_args.p)
                     : launch_pool_kernel<BLOCK, WAVES, LDSTAB, RING, false, TEX, ADAPT, RINGMEM, true>(ctx, grid, lds, ctx->pool_args.p);
    }
    if (shared)
        rc = count ? launch_pool_kernel<BLOCK, WAVES, LDSTAB, RING, true, TEX, ADAPT, RINGMEM, false, true>(ctx, grid, lds, ctx->pool_args.p)
                   : launch_pool_kernel<BLOCK, WAVES, LDSTAB, RING, false, TEX, ADAPT, RINGMEM, false, true>(ctx, grid, lds, ctx->pool_args.p);
    else
        rc = count ? launch_pool_kernel<BLOCK, WAVES, LDSTAB, RING, true, TEX, ADAPT, RINGMEM, false>(ctx, grid, lds, ctx->pool_args.p)
                   : launch_pool_kernel<BLOCK, WAVES, LDSTAB, RING, false, TEX, ADAPT, RINGMEM, false>(ctx, grid, lds, ctx->pool_args.p);
    if (rc) return rc;
    if (count) hipLaunchKernelGGL(k_pool_parked_shadows<true>, dim3(POOL_PARKED_SHADOW_BLOCKS), dim3(256), 0, ctx->stream, ctx->pool_args.p, ctx->counters.p);
    else hipLaunchKernelGGL(k_pool_parked_shadows<false>, dim3(POOL_PARKED_SHADOW_BLOCKS), dim3(256), 0, ctx->stream, ctx->pool_args.p, ctx->counters.p);
    HIP_TRY(ctx, hipGetLastError());
    // the adopting launch: one block per compute unit; they find the park list empty and leave at once - except in scenes
    // with coincident geometry, where they finish what the first launch set aside
    return count ? launch_pool_kernel<BLOCK, WAVES, LDSTAB, RING, true, TEX, ADAPT, RINGMEM, true>(ctx, grid2, lds, ctx->pool_args.p + 1)
                 : launch_pool_kernel<BLOCK, WAVES, LDSTAB, RING, false, TEX, ADAPT, RINGMEM, true>(ctx, grid2, lds, ctx->pool_args.p + 1);
}

template <bool FIXED>
void launch_resolve_t(hipStream_t stream, const void * samples, float4 * out, unsigned int n_px, unsigned int spp, ResolveMap map) {
    const unsigned int grid2 = (unsigned int)(((unsigned long long)n_px * spp + 255ull) / 256ull);
    switch (spp) {
        case 2: hipLaunchKernelGGL((k_resolve_pow2<2, FIXED>), dim3(grid2), dim3(256), 0, stream, samples, out, n_px, map); break;
        case 4: hipLaunchKernelGGL((k_resolve_pow2<4, FIXED>), dim3(grid2), dim3(256), 0, stream, samples, out, n_px, map); break;
        case 8: hipLaunchKernelGGL((k_resolve_pow2<8, FIXED>), dim3(grid2), dim3(256), 0, stream, samples, out, n_px, map); break;
        case 16: hipLaunchKernelGGL((k_resolve_pow2<16, FIXED>), dim3(grid2), dim3(256), 0, stream, samples, out, n_px, map); break;
        case 32: hipLaunchKernelGGL((k_resolve_pow2<32, FIXED>), dim3(grid2), dim3(256), 0, stream, samples, out, n_px, map); break;
        case 64: hipLaunchKernelGGL((k_resolve_pow2<64, FIXED>), dim3(grid2), dim3(256), 0, stream, samples, out, n_px, map); break;
        default: hipLaunchKernelGGL(k_resolve<FIXED>, dim3((n_px + 255) / 256), dim3(256), 0, stream, samples, out, n_px, spp, map);
    }
}
// fixed: the samples are the wavefront / pool pipelines' fixed-point accumulators (dev_scene.h Accum), else float4 colours
void launch_resolve(hipStream_t stream, const void * samples, bool fixed, float4 * out, unsigned int n_px, unsigned int spp, ResolveMap map) {
    if (fixed) launch_resolve_t<true>(stream, samples, out, n_px, spp, map);
    else launch_resolve_t<false>(stream, samples, out, n_px, spp, map);
}

// GPU radix-tree build + host collapse / quantise.  verts: 9 floats per triangle.
int build_bvh_lbvh(prt_ctx * ctx, const float * verts, uint32_t n_tris, uint32_t leaf_max, BvhWide * bvh) {
    float lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };
    for (uint32_t i = 0; i < n_tris * 3u; ++i)
        for (int a = 0; a < 3; ++a) {
            const float v = verts[(size_t)i * 3 + a];
            if (i == 0 || v < lo[a]) lo[a] = v;
            if (i == 0 || v > hi[a]) hi[a] = v;
        }
    LbvhTree tree;
    double device_ms = 0.0;
    HIP_TRY(ctx, build_lbvh_tree(verts, n_tris, lo, hi, ctx->stream, &tree, &device_ms));
    PRT_BUILD_WIDE_FROM_RADIX(n_tris, leaf_max, tree.left.data(), tree.right.data(), tree.first.data(), tree.last.data(),
                              tree.node_box.data(), tree.leaf_box.data(), tree.sorted_ids.data(), bvh, &ctx->opt.bvh);
    if (ctx->opt.debug_util) fprintf(stderr, "[prt] LBVH: %u triangles, radix tree on the device in %.2f ms\n", n_tris, device_ms);
    return 0;
}

// Every address the traversal kernels will form from the tree is checked HERE, on the host, before the tree is uploaded: a
// node link or a triangle range outside the arrays would be a wild read on the device - which the runtime reports by
// aborting the process (the round-3 abort) - and is an upload error instead.  O(nodes), links only;
// prt_debug_check_bvh is the full geometric check.
const char * validate_bvh_links(const Bvh4Result & bvh, uint32_t n_tris) {
    if (bvh.node_count == 0 || bvh.nodes.size() != (size_t)bvh.node_count * 16) return "4-wide BVH: node array size does not match the node count";
    if (bvh.tri_order.size() != n_tris) return "4-wide BVH: triangle order does not cover the triangles";
    for (uint32_t ni = 0; ni < bvh.node_count; ++ni)
        for (int k = 0; k < 4; ++k) {
            const int32_t link = (int32_t)bvh.nodes[(size_t)ni * 16 + 10 + k];
            if (link >= 0) {
                if ((uint32_t)link >= bvh.node_count || (uint32_t)link <= ni) return "4-wide BVH: child link out of range (children follow their parent in breadth-first order)";
            } else {
                const uint32_t leaf = (uint32_t)~link, first = leaf >> 2, cnt = (leaf & 3u) + 1u;
                // an empty slot points at the all-zero dummy record behind the last triangle
                if (first > n_tris || (first < n_tris && first + cnt > n_tris)) return "4-wide BVH: leaf triangle range out of range";
            }
        }
    return nullptr;
}
const char * validate_bvh_links(const Bvh8Result & bvh, uint32_t n_tris) {
    if (bvh.node_count == 0 || bvh.nodes.size() != (size_t)bvh.node_count * BVH8_NODE_DWORDS) return "8-wide BVH: node array size does not match the node count";
    if (bvh.tri_order.size() != n_tris) return "8-wide BVH: triangle order does not cover the triangles";
    for (uint32_t ni = 0; ni < bvh.node_count; ++ni) {
        const uint32_t * d = &bvh.nodes[(size_t)ni * BVH8_NODE_DWORDS];
        const uint32_t imask = d[3] & 0xFFu, lmask = d[3] >> 8 & 0xFFu, c0 = d[6] & 0xFFu, c1 = d[6] >> 8 & 0xFFu;
        if (imask & lmask) return "8-wide BVH: a slot is both an internal node and a leaf";
        uint32_t n_child = 0, n_leaf_tris = 0;
        for (uint32_t sl = 0; sl < 8; ++sl) {
            n_child += imask >> sl & 1u;
            if (lmask >> sl & 1u) n_leaf_tris += 1u + (c0 >> sl & 1u) + 2u * (c1 >> sl & 1u);
        }
        if (n_child && ((uint64_t)d[4] + n_child > bvh.node_count || d[4] <= ni)) return "8-wide BVH: child range out of range";
        if (n_leaf_tris && (uint64_t)d[5] + n_leaf_tris > (uint64_t)n_tris + (n_tris == 0 ? 1u : 0u)) return "8-wide BVH: leaf triangle range out of range";
    }
    return nullptr;
}

int render_pixels(prt_ctx * ctx, const prt_camera * cam_in, const prt_params * params, uint32_t width, uint32_t height,
                  const PixelSet & px, float4 * d_out, prt_counters * counters);

// Renders the pixel set into d_out (device, float4 per pixel, packed in local pixel order).  Synchronous.
int render_pixels_once(prt_ctx * ctx, const prt_camera * cam_in, const prt_params * params, uint32_t width, uint32_t height,
                       const PixelSet & px, float4 * d_out, prt_counters * counters, bool * park_overflow) {
    *park_overflow = false;
    if (!ctx->has_scene) { ctx->error = "prt_render: no scene uploaded"; return -2; }
    if (!cam_in || !params || !width || !height) { ctx->error = "prt_render: null camera / params or empty image"; return -1; }
    if (params->spp == 0 || params->spp > 65535) { ctx->error = "prt_render: spp must be in 1..65535 (prt_key.h packs the sample in 16 bits)"; return -1; }
    if (params->bounce_depth > 16) { ctx->error = "prt_render: bounce_depth > 16 not supported"; return -1; }
    if ((uint64_t)width * height >= (1ull << 32)) { ctx->error = "prt_render: image has 2^32 pixels or more"; return -1; }
    if (px.n_pixels == 0) {                 // RenderTask with start_idx == end_idx (main.cpp:273): nothing to do, not an error
        if (counters) { memset(counters, 0, sizeof(*counters)); counters->pipeline = params->pipeline & PRT_PIPELINE_MASK; }
        return 0;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = ctx->stream;

    if (std::max(1u, params->spec_samples) != ctx->spec_table_samples) {
        HIP_TRY(ctx, hipStreamSynchronize(stream));
        int rc = build_spec_table(ctx, params->spec_samples);
        if (rc) return rc;
        HIP_TRY(ctx, hipDeviceSynchronize());      // the table went through the null stream
    }

    DevCamera cam;
    cam.position = ld3(cam_in->position);
    cam.forward = ld3(cam_in->forward);
    cam.right_scaled = ld3(cam_in->right) * cam_in->tan_a2 * cam_in->aspect;      // main.cpp:170
    cam.up_scaled = ld3(cam_in->up) * cam_in->tan_a2;                              // main.cpp:171
    cam.inv_width = cam_in->inv_width;
    cam.inv_height = cam_in->inv_height;

    DevParams P;
    P.ray_bias = params->ray_bias;
    P.reflection_samples = params->reflection_samples;
    P.spec_samples = params->spec_samples;
    P.bounce_depth = params->bounce_depth;
    P.background = mk3(params->background_color[0], params->background_color[1], params->background_color[2]);
    P.spp = params->spp;
    P.seed = params->seed;
    const bool adaptive = params->max_spp > params->spp;            // RenderPixel's second loop (main.cpp:245-258)
    P.max_spp = adaptive ? params->max_spp : params->spp;
    P.variance_threshold = params->variance_threshold > 0.0f ? params->variance_threshold : 0.01f;    // main.cpp:254
    P.width = width;
    P.height = height;
    float extent = ctx->scene_abs_max;
    for (int a = 0; a < 3; ++a) extent = std::max(extent, fabsf(cam_in->position[a]));
    P.box_pad = extent * (1.0f / 65536.0f);
    P.first_pixel = px.first_pixel;
    P.shard_block_rows = std::max(1u, px.block_rows);
    P.shard_rank = px.rank;
    P.shard_nranks = std::max(1u, px.nranks);
    P.pixel_list = px.d_pixel_list;
    const PrtOptions & opt = ctx->opt;
    ctx->scene.tie_widen_max = (unsigned int)std::max(0ll, std::min(64ll, opt.tie_widen_max));
    P.elide_dead_shadow_rays = opt.trace_dead_shadow_rays ? 0u : 1u;
    // tiled work order: sets made of full-width rows only (a whole frame or a range that starts at a row, row-block shards)
    P.tile_pixels = 0;
    if (!px.d_pixel_list && width % 8u == 0u && !opt.no_tiles) {
        const bool rows = px.nranks > 1 ? std::max(1u, px.block_rows) % 8u == 0u : px.first_pixel % width == 0u;
        if (rows) P.tile_pixels = px.n_pixels / (8u * width) * (8u * width);
    }

    P.work_reverse_n = (opt.work_reverse > 0 && !px.d_pixel_list) ? px.n_pixels : 0u;
    P.work_scatter_n = 0; P.work_scatter_mul = 1;
    if (opt.work_scatter > 0 && !px.d_pixel_list && P.tile_pixels >= 64u) {
        // a multiplier near n / golden ratio spreads consecutive chunks far apart; coprime to n, and (n - 1) * mul must fit 32 bits
        const unsigned int n = P.tile_pixels / 8u;
        unsigned long long mul = std::max<unsigned long long>(3ull, (unsigned long long)((double)n * 0.6180339887));
        mul = std::min<unsigned long long>(mul, 0xFFFFFFFFull / n);
        if (!(mul & 1ull)) mul -= 1ull;                       // odd, and never above the 32-bit limit
        auto gcd = [](unsigned long long a, unsigned long long b) { while (b) { const unsigned long long t = a % b; a = b; b = t; } return a; };
        while (mul > 1ull && gcd(mul, n) != 1ull) mul -= 2ull;
        if (mul > 1ull) { P.work_scatter_n = n; P.work_scatter_mul = (unsigned int)mul; }
    }
    const bool count_visits = (params->pipeline & PRT_FLAG_COUNT_VISITS) != 0;
    // the compact 2-register RNG only covers opaque scenes with <= 15 draws per sample; textured scenes always take the
    // general variant (an alpha map can make any hit translucent)
    const bool ring = adaptive || ctx->any_translucent || ctx->textured || max_rng_draws(P.bounce_depth, P.reflection_samples, P.spec_samples) > 15;
#if defined(PRT_EXPERIMENTAL)
    const int levels = (int)P.bounce_depth + 1;
#endif

    unsigned int pipeline = params->pipeline & PRT_PIPELINE_MASK;
    if (adaptive) {
        // a pixel's samples form a chain on one RNG stream with a data-dependent length: only the pool pipeline, whose
        // work items carry their own state machine, runs it
        if (pipeline != PRT_PIPELINE_DEFAULT && pipeline != PRT_PIPELINE_POOL) { ctx->error = "prt_render: adaptive sampling (max_spp > spp) runs on PRT_PIPELINE_POOL (or DEFAULT) only"; return -1; }
        if (params->max_spp > 4096) { ctx->error = "prt_render: max_spp above 4096 is not supported"; return -1; }
        pipeline = PRT_PIPELINE_POOL;
    }
    if (pipeline == PRT_PIPELINE_DEFAULT) {
        // Measured on MI355X (tools/pool_cross.py, tools/pool_scene_cross.py): the wavefront pipeline's steady state is
        // ~12 % faster (its k_trace runs 6 waves per SIMD, k_pool 4), but each of its ~8 rounds costs a launch ramp, a
        // drain tail and a host round trip, ~1 ms per frame in total.  So the single-launch pool pipeline wins when a
        // frame takes less than ~6 ms on the wavefront pipeline - every C4 shard of a 4..8 GPU run, but also a full 1080p
        // frame of a scene whose rays are short (C3: 4.75 vs 5.85 ms).  Sample count alone cannot tell those apart, so in
        // the range where either can win the first call of a configuration renders the frame with both (twice each: the
        // first run of a pipeline allocates its workspace) and keeps the faster; the images are the same.  Outside
        // that range - and always when PRT_POOL_MAX_SAMPLES is set - the size decides.  (Below 1 M samples: pool.)
        // The try-out is opt-in (PRT_FLAG_TRYOUT): without it DEFAULT is the pool pipeline.
        const unsigned long long ns = (unsigned long long)px.n_pixels * params->spp;
        if (opt.pool_max_samples >= 0) {
            pipeline = ns <= (unsigned long long)opt.pool_max_samples ? PRT_PIPELINE_POOL : PRT_PIPELINE_WAVEFRONT;
        } else if (!(params->pipeline & PRT_FLAG_TRYOUT) || ns <= (1ull << 20)) {
            pipeline = PRT_PIPELINE_POOL;       // no try-out asked for (a one-off render would pay for it five-fold), or too small to matter
        } else {
            prt_ctx::TuneEntry e;
            e.key[0] = ctx->scene_epoch;
            e.key[1] = (uint64_t)px.n_pixels | (uint64_t)params->spp << 32;
            e.key[2] = (uint64_t)width | (uint64_t)height << 32;
            e.key[3] = (uint64_t)params->bounce_depth | (uint64_t)params->reflection_samples << 32;      // full-width fields: no aliasing
            e.key[4] = (uint64_t)params->spec_samples | (uint64_t)px.nranks << 32;
            e.key[5] = (uint64_t)px.block_rows | (uint64_t)(px.d_pixel_list ? 1 : 0) << 32;
            for (const prt_ctx::TuneEntry & t : ctx->tuned)
                if (!memcmp(t.key, e.key, sizeof(e.key))) pipeline = t.pipeline;
            if (pipeline == PRT_PIPELINE_DEFAULT) {
                // calls that will run in several passes (more than 64 M samples) try the two pipelines on their first
                // 32 M samples only (C5, 4K x 64 spp at depth 8: pool 1005 ms, wavefront 1181 ms per frame)
                PixelSet trial = px;
                const bool whole = ns <= (64ull << 20);
                if (!whole) trial.n_pixels = (uint32_t)std::max<unsigned long long>(64, ((32ull << 20) / params->spp) / 64 * 64);
                const unsigned int cand[2] = { PRT_PIPELINE_POOL, PRT_PIPELINE_WAVEFRONT };
                double ms[2] = { 0.0, 0.0 };
                prt_counters c;
                for (int run = 0; run < 4; ++run) {                        // pool, wavefront (cold), pool, wavefront (timed)
                    prt_params pp = *params;
                    pp.pipeline = (params->pipeline & ~(uint32_t)PRT_PIPELINE_MASK) | cand[run & 1];
                    int rc = render_pixels(ctx, cam_in, &pp, width, height, trial, d_out, &c);
                    if (rc) return rc;
                    ms[run & 1] = c.render_ms;
                }
                // run-to-run noise is 2-3 %: the pool pipeline has to win by more than that to displace the other
                e.pipeline = ms[0] <= 0.97 * ms[1] ? PRT_PIPELINE_POOL : PRT_PIPELINE_WAVEFRONT;
                if (ctx->tuned.size() >= 64) ctx->tuned.erase(ctx->tuned.begin());
                ctx->tuned.push_back(e);
                if (opt.debug_util) fprintf(stderr, "[prt] default pipeline try-out: pool %.3f ms, wavefront %.3f ms\n", ms[0], ms[1]);
                // the call itself is then served by the winner like every later one, so that the counters (pipeline, kernel
                // times) describe the pipeline this configuration will keep
                (void)whole;
                pipeline = e.pipeline;
            }
        }
    }
    if (pipeline != PRT_PIPELINE_MEGAKERNEL && pipeline != PRT_PIPELINE_WAVEFRONT && pipeline != PRT_PIPELINE_PERSISTENT && pipeline != PRT_PIPELINE_POOL) { ctx->error = "prt_render: unknown pipeline"; return -1; }
#if !defined(PRT_EXPERIMENTAL)
    if (pipeline == PRT_PIPELINE_MEGAKERNEL || pipeline == PRT_PIPELINE_PERSISTENT) {
        ctx->error = "prt_render: the experimental pipelines (MEGAKERNEL, PERSISTENT) are not built into this library (make hip-experimental)";
        return -1;
    }
#endif
    if (ctx->textured && (pipeline == PRT_PIPELINE_MEGAKERNEL || pipeline == PRT_PIPELINE_PERSISTENT)) {
        ctx->error = "prt_render: textured scenes run on PRT_PIPELINE_WAVEFRONT / PRT_PIPELINE_POOL (or DEFAULT) only";
        return -1;
    }

    // The pool kernel always runs its general-RNG variant: with the draw ring in memory it fits 96 VGPRs with 29 dwords
    // spilled and runs 5 waves per SIMD, where the 2-register RNG variant spills 105 (measured: 14.7 vs 15.3 ms per C4 frame
    // at 4 waves per SIMD; same random numbers either way).
    const bool ring_eff = ring || pipeline == PRT_PIPELINE_POOL;

    // Passes.  The per-sample workspace (radiance, RNG, pending frames, ray queues) is 100 B .. 1 KB per sample, so a
    // 4K x 64 spp frame (530 M samples) does not fit any GPU in one piece: the call's pixel set is rendered in passes of
    // whole pixels, each at most 64 M samples and at most 64 GB of workspace (PRT_PASS_SAMPLES / PRT_PASS_MB override).
    // One pass for everything up to 4K x 8 spp.
    const unsigned int unit_spp = adaptive ? 1u : P.spp;            // work items per pixel: samples, or the pixel itself
    const unsigned long long total_samples = (unsigned long long)px.n_pixels * unit_spp;
    unsigned int pass_pixels = px.n_pixels;
    {
        const unsigned long long lv = std::max(1u, P.bounce_depth), fr4 = ctx->textured ? 7 : ring_eff ? 5 : 4, nl = std::max(1u, ctx->scene.light_count);
        unsigned long long per_sample = 32 + (ring && pipeline != PRT_PIPELINE_PERSISTENT ? 128 : 0);
        if (pipeline == PRT_PIPELINE_WAVEFRONT) per_sample += (lv * fr4 + 7 + 3 * nl) * 16 + (ring ? 32 : 16) + 4 * (1 + nl);
        if (pipeline == PRT_PIPELINE_POOL) per_sample += lv * fr4 * 16 + 32;
        if (adaptive) per_sample += ((unsigned long long)P.max_spp + 2) * 16;
        if (pipeline == PRT_PIPELINE_MEGAKERNEL && ctx->stack_bound > 24) per_sample += 4ull * ctx->stack_bound;
        // A pass ends with a drain (the waves run dry one by one), so fewer, larger passes are faster - C5: 8 passes 1,046 - 1,076
        // ms, 4 passes 1,002, 3 passes 977 - 992 (profiles/r02_c5_pass_size.txt) - and 288 GB are there to be used: up to 192 M
        // samples and 160 GB of workspace per pass, but no more than 80 % of what the device has free plus what this context
        // already holds (a second context on the same GPU - two frames in flight - gets what is left, not an allocation failure).
        unsigned long long max_samples = 192ull << 20, max_mb = 160ull << 10;
        {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                unsigned long long held = ctx->sample_rgb.bytes() + ctx->ring_ws.bytes() + ctx->pool_f4.bytes() + ctx->adapt_f4.bytes() +
                                          ctx->stack_spill.bytes() + ctx->pool_fin.bytes();
                for (int c = 0; c < PRT_MAX_CHAINS; ++c) held += ctx->chain[c].f4.bytes() + ctx->chain[c].rng.bytes() + ctx->chain[c].overflow.bytes() + ctx->chain[c].slow_stack.bytes();
                const unsigned long long budget_mb = (unsigned long long)(0.8 * (double)(free_b + held)) >> 20;
                max_mb = std::max(1024ull, std::min(max_mb, budget_mb));
            }
        }
        if (opt.pass_samples >= 0) max_samples = std::max(1ull, (unsigned long long)opt.pass_samples);
        if (opt.pass_mb >= 0) max_mb = std::max(1ull, (unsigned long long)opt.pass_mb);
        max_samples = std::min(max_samples, std::max(1ull, (max_mb << 20) / per_sample));
        max_samples = std::min(max_samples, 0x7FFFFFFFull);
        if (total_samples > max_samples) {
            unsigned long long pp = std::max(1ull, max_samples / unit_spp);
            if (pp > 64) pp = pp / 64 * 64;                                 // passes start on a wave boundary
            pass_pixels = (unsigned int)std::min<unsigned long long>(pp, px.n_pixels);
        }
    }
    const size_t n_samples64 = (size_t)pass_pixels * unit_spp;              // work items of the largest pass
    if (n_samples64 > 0x7FFFFFFFull) { ctx->error = "prt_render: spp too large for one pixel per pass"; return -1; }
    HIP_TRY(ctx, ctx->sample_rgb.ensure(2 * n_samples64));      // float4 colours (megakernel, persistent) or 32-byte fixed-point accumulators
    HIP_TRY(ctx, ctx->counters.ensure(1));
    if (ring && pipeline != PRT_PIPELINE_PERSISTENT) HIP_TRY(ctx, ctx->ring_ws.ensure(n_samples64 * 16));

    // Traversal stack: LDS column of up to STACK_LDS_CAP entries per lane (occupancy); the rest of the worst-case bound
    // - 3 pushes per 4-wide level are possible, nothing real comes close - lives in a per-lane global column behind it
    // (dev_trace.h LdsStack).
    constexpr int BLOCK = 256;
    unsigned int stack_cap = STACK_LDS_CAP_DEFAULT;
    if (opt.stack_cap >= 0) stack_cap = (unsigned int)std::max(2ll, std::min(40ll, opt.stack_cap));   // test hook: force the spill area into use
    const unsigned int stack_entries = std::min(ctx->stack_bound, stack_cap);
    const size_t lds = stack_dwords(stack_entries, BLOCK) * sizeof(int);
    P.stack_lds_entries = stack_entries;
    P.stack_spill = nullptr;
    P.stack_spill_stride = 0;


    {
        const unsigned int spill_entries = ctx->stack_bound > stack_entries ? ctx->stack_bound - stack_entries : 0;
        // one spill column per lane that may need it: every sample lane (megakernel), every persistent lane (persistent);
        // the pool and wavefront pipelines size theirs where they know their grids (launch_pool, chain_setup)
        const size_t spill_lanes = pipeline == PRT_PIPELINE_MEGAKERNEL ? ((n_samples64 + BLOCK - 1) / BLOCK) * BLOCK
                                 : pipeline == PRT_PIPELINE_PERSISTENT ? (size_t)8 * (size_t)ctx->cu_count * BLOCK
                                                                       : 0;                      // wavefront: per chain, see chain_setup
        if (spill_entries && spill_lanes) {
            if (spill_lanes >= (1ull << 32)) { ctx->error = "prt_render: too many lanes for the stack spill area"; return -1; }
            HIP_TRY(ctx, ctx->stack_spill.ensure(stack_dwords(spill_entries, spill_lanes)));
            P.stack_spill = ctx->stack_spill.p;
            P.stack_spill_stride = (unsigned int)spill_lanes;
        }
    }

    HIP_TRY(ctx, hipMemsetAsync(ctx->counters.p, 0, sizeof(DevCounters), stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev[0], stream));
    unsigned int launches = 0;
    unsigned long long host_ray_count = 0;
    float trace_ms_accum = 0.0f;
    const bool single_launch = pipeline != PRT_PIPELINE_WAVEFRONT;
    for (unsigned int p0 = 0; p0 < px.n_pixels; p0 += pass_pixels) {
        const unsigned int n_px = std::min(pass_pixels, px.n_pixels - p0);
        const unsigned int n_samples = n_px * unit_spp;
        const bool last_pass = p0 + n_px == px.n_pixels;
        P.local_base = p0;
        if (single_launch) HIP_TRY(ctx, hipEventRecord(ctx->ev[2], stream));
        int rc = 0;
#if defined(PRT_EXPERIMENTAL)
        if (pipeline == PRT_PIPELINE_MEGAKERNEL) {
            const unsigned int grid = (n_samples + BLOCK - 1) / BLOCK;
            if (!ring && levels <= 3) launch_mega<3, false>(ctx, count_visits, grid, lds, cam, P, n_samples);
            else if (levels <= 9) launch_mega<9, true>(ctx, count_visits, grid, lds, cam, P, n_samples);
            else launch_mega<17, true>(ctx, count_visits, grid, lds, cam, P, n_samples);
            launches += 1;
        } else if (pipeline == PRT_PIPELINE_PERSISTENT) {
            HIP_TRY(ctx, ctx->wf_counts.ensure(16));
            HIP_TRY(ctx, hipMemsetAsync(ctx->wf_counts.p, 0, 16, stream));
            int keep_min = 40, node_min = 32;
            if (opt.keep_min >= 0) keep_min = std::max(1, std::min(64, (int)opt.keep_min));
            if (opt.node_min >= 0) node_min = std::max(0, std::min(64, (int)opt.node_min));
            int blocks_cap = 8;
            if (opt.trace_blocks_per_cu >= 0) blocks_cap = std::max(1, std::min(8, (int)opt.trace_blocks_per_cu));
            if (!ring && levels <= 3) rc = launch_persistent<3, false>(ctx, count_visits, lds, cam, P, n_samples, keep_min, node_min, blocks_cap);
            else if (levels <= 9) rc = launch_persistent<9, true>(ctx, count_visits, lds, cam, P, n_samples, keep_min, node_min, blocks_cap);
            else rc = launch_persistent<17, true>(ctx, count_visits, lds, cam, P, n_samples, keep_min, node_min, blocks_cap);
            launches += 1;
        } else
#endif
        if (pipeline == PRT_PIPELINE_POOL) {
            // 256-thread blocks, the Hammersley direction table in global memory (staging it in LDS measured 17.04 vs 17.15 ms:
            // nothing).  Small blocks retire - and let the blocks of the next frame's kernel in - at a finer grain: with two
            // frames in flight a 1/8-frame shard takes 2.24 ms per frame instead of 2.59 with 512-thread blocks.
            // Untextured fixed-spp renders: 5 waves per SIMD (96 VGPRs, 23 dwords spilled).  Textured (64 spilled at 96) and
            // adaptive (68 spilled at 96, no gain measured: profiles/r02_experiments.txt item 11) renders: 4 waves per SIMD.
            // 6 waves (80 VGPRs) spill 90-230 dwords: 29 ms.
            // last argument (k_pool's RINGMEM): 0 = no sample passes 15 draws (and the scene is opaque and untextured): no draw ring
            // in memory; 1 = draw ring, materials may be translucent; 2 = draw ring, opaque scene: the translucency paths and the
            // frames' hit position fold away as they do for 0.  2 is used for the adaptive mode (-3.6 % on C4 10..50 spp); the
            // fixed-spp kernel for deep bounce trees allocates worse with it at 96 VGPRs (47 dwords spilled instead of 31: +1 %)
            // and stays on 1 (profiles/r03_adaptive.txt item 7)
            const bool opaque = !ctx->any_translucent && !ctx->textured;
            if (adaptive)
                rc = ctx->textured ? launch_pool<256, 4, false, true, true, true>(ctx, count_visits, cam, P, n_samples, stack_entries)
                   : opaque ? launch_pool<256, PRT_ADAPT_WAVES, false, true, false, true, 2>(ctx, count_visits, cam, P, n_samples, stack_entries)
                            : launch_pool<256, PRT_ADAPT_WAVES, false, true, false, true, 1>(ctx, count_visits, cam, P, n_samples, stack_entries);
            else
                rc = ctx->textured ? launch_pool<256, 4, false, true, true, false>(ctx, count_visits, cam, P, n_samples, stack_entries)
                   : !ring ? launch_pool<PRT_POOL_BLOCK, PRT_POOL_WAVES, false, true, false, false, 0>(ctx, count_visits, cam, P, n_samples, stack_entries)
#if defined(PRT_DEEP_OPAQUE_VARIANT)
                   : opaque ? launch_pool<PRT_POOL_BLOCK, PRT_DEEP_WAVES, false, true, false, false, 2>(ctx, count_visits, cam, P, n_samples, stack_entries)
#endif
                           : launch_pool<PRT_POOL_BLOCK, PRT_DEEP_WAVES, false, true, false, false, 1>(ctx, count_visits, cam, P, n_samples, stack_entries);
            launches += 1;
        } else {
            unsigned long long rays = 0;
            float tms = 0.0f;
            unsigned int nl = 0;
            rc = render_wavefront(ctx, cam, P, ring, count_visits, n_samples, lds, &rays, &tms, &nl);
            host_ray_count += rays;
            trace_ms_accum += tms;
            launches += nl;
        }
        if (rc) return rc;
        HIP_TRY(ctx, hipGetLastError());
        if (single_launch) HIP_TRY(ctx, hipEventRecord(ctx->ev[3], stream));
        // work items of this pass -> their places in the call's output (tiled work order, dev_scene.h local_of_work)
        ResolveMap rmap;
        rmap.base = p0; rmap.width = P.width; rmap.tile_pixels = P.tile_pixels; rmap.reverse_n = P.work_reverse_n; rmap.scatter_n = P.work_scatter_n; rmap.scatter_mul = P.work_scatter_mul;
        if (adaptive)       // k_pool<ADAPT> has already divided by each pixel's own sample count
            hipLaunchKernelGGL(k_resolve<false>, dim3((n_px + 255) / 256), dim3(256), 0, stream,
                               ctx->adapt_f4.p + ((size_t)P.max_spp + 1u) * n_samples, d_out, n_px, 1u, rmap);
        else
            launch_resolve(stream, ctx->sample_rgb.p, pipeline == PRT_PIPELINE_WAVEFRONT || pipeline == PRT_PIPELINE_POOL, d_out, n_px, P.spp, rmap);
        HIP_TRY(ctx, hipGetLastError());
        if (single_launch && (counters || !last_pass)) {
            // the next pass reuses ev[2] / ev[3] (and the workspace is stream ordered anyway): take this pass's time now
            float tms = 0.0f;
            HIP_TRY(ctx, hipEventSynchronize(ctx->ev[3]));
            HIP_TRY(ctx, hipEventElapsedTime(&tms, ctx->ev[2], ctx->ev[3]));
            trace_ms_accum += tms;
        }
    }
    // the counters travel on the context's own stream into pinned memory, in front of the event the host waits for: no
    // synchronous copy through the null stream, which CU-masked (blocking) streams of other contexts would be ordered against
    HIP_TRY(ctx, hipMemcpyAsync(ctx->host_counters, ctx->counters.p, sizeof(DevCounters), hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev[1], stream));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev[1]));

    *park_overflow = false;
    DevCounters h = *ctx->host_counters;
    if (pipeline == PRT_PIPELINE_POOL && (h.park_over[0] || h.park_over[1])) {
        // a park list was too short for one of this call's launches (kernels_pool.h PoolBuffers::park): rays were dropped.  The
        // device decided that per launch, against the capacity that launch's lists had (every pass clamps them to its own
        // worst case).  Longer lists, then the caller renders the frame again: a capacity above the largest demand seen is
        // above every pass's demand, and a pass whose clamp is smaller than that cannot want more than its clamp.
        if (opt.debug_util) fprintf(stderr, "[prt] park lists too short (a pass wanted %llu ray / %llu shadow-ray entries): enlarging\n",
                                    (unsigned long long)h.park_over[0], (unsigned long long)h.park_over[1]);
        ctx->pool_park_cap = std::max<size_t>(ctx->pool_park_cap, (size_t)h.park_over[0] + (size_t)h.park_over[0] / 2);
        ctx->pool_spark_cap = std::max<size_t>(ctx->pool_spark_cap, (size_t)h.park_over[1] + (size_t)h.park_over[1] / 2);
        *park_overflow = true;
        return 0;
    }
    if (h.flow_error) {
        char msg[160];
        snprintf(msg, sizeof(msg), "prt_render: a wait inside a pool kernel exceeded its watchdog (code 0x%x): the frame is incomplete", h.flow_error);
        ctx->error = msg;
        return -5;
    }
    if (h.near_tie_unresolved) {
        // resolve_near_ties ran out of widenings (dev_trace8.h): some hit among near-coincident candidates was decided over an
        // incomplete candidate set.  Never seen on real geometry; reported, not hidden.
        char msg[160];
        snprintf(msg, sizeof(msg), "prt_render: %llu near-tied hits could not be resolved within %lld widenings of the candidate set",
                 (unsigned long long)h.near_tie_unresolved, opt.tie_widen_max);
        ctx->error = msg;
        return -8;
    }
    if (counters) {
        {
            prt_render_stats & rs = ctx->last_stats;
            memset(&rs, 0, sizeof(rs));
            rs.node_visits = h.node_visits; rs.tri_tests = h.tri_tests;
            rs.wave_node_steps = h.wave_node_steps; rs.wave_tri_steps = h.wave_tri_steps; rs.wave_leaf_visits = h.wave_leaf_steps;
            rs.wave_refills = h.wave_refills; rs.deepest_stack = h.max_sp;
            for (int k = 0; k < 5; ++k) rs.phase_cycles[k] = h.phase_cycles[k];
            rs.parked_rays = h.park_peak[0]; rs.parked_shadow_rays = h.park_peak[1];
            rs.elided_shadow_rays = h.elided_shadow_rays;
            rs.variance_close_calls = h.variance_close_calls;
            rs.stack_lds_entries = stack_entries; rs.stack_bound = ctx->stack_bound;
        }
        float ms = 0.0f;
        const float trace_ms = trace_ms_accum;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        if (pipeline == PRT_PIPELINE_WAVEFRONT) h.ray_count = host_ray_count;      // every queued ray is one TraceRay call
        else host_ray_count = h.ray_count;
        if (opt.debug_util && h.wave_node_steps)
            fprintf(stderr, "[prt] lane utilisation: node loop %.1f%% (%llu wave steps), triangle tests %.1f%% (%llu wave steps, %llu leaf visits), %llu refills (%.1f rays each)\n",
                    100.0 * (double)h.node_visits / (64.0 * (double)h.wave_node_steps), (unsigned long long)h.wave_node_steps,
                    100.0 * (double)h.tri_tests / (64.0 * (double)h.wave_tri_steps), (unsigned long long)h.wave_tri_steps,
                    (unsigned long long)h.wave_leaf_steps, (unsigned long long)h.wave_refills,
                    h.wave_refills ? (double)host_ray_count / (double)h.wave_refills : 0.0);
        if (opt.debug_util && h.wave_node_step_rays)
            fprintf(stderr, "[prt] k_pool node loop, lane slots per wave step: %.1f%% walking, %.1f%% holding a ray but not walking (at a leaf, or done), %.1f%% without a ray\n",
                    100.0 * (double)h.node_visits / (64.0 * (double)h.wave_node_steps),
                    100.0 * ((double)h.wave_node_step_rays - (double)h.node_visits) / (64.0 * (double)h.wave_node_steps),
                    100.0 * (64.0 * (double)h.wave_node_steps - (double)h.wave_node_step_rays) / (64.0 * (double)h.wave_node_steps));
        if (opt.debug_util && h.drain_node_steps && h.wave_node_steps)
            fprintf(stderr, "[prt] k_pool round drains (node steps taken after a round's list ran dry): %.1f%% of the wave-level node steps, %.1f%% of their lane slots hold a ray; "
                            "the steps before the list ran dry: %.1f%% of lane slots hold a ray\n",
                    100.0 * (double)h.drain_node_steps / (double)h.wave_node_steps, 100.0 * (double)h.drain_node_step_rays / (64.0 * (double)h.drain_node_steps),
                    100.0 * ((double)h.wave_node_step_rays - (double)h.drain_node_step_rays) / (64.0 * ((double)h.wave_node_steps - (double)h.drain_node_steps)));
        if (opt.debug_util && h.flow_cycles[1] && h.flow_cycles[5])
            fprintf(stderr, "[prt] k_flow: tracer waves wait for rays %.1f%% of their time; the shading wave tops up %.1f%%, shades %.1f%%, waits %.1f%% of its time; %llu shade batches of %.1f hits\n",
                    100.0 * (double)h.flow_cycles[0] / (double)h.flow_cycles[1], 100.0 * (double)h.flow_cycles[2] / (double)h.flow_cycles[5],
                    100.0 * (double)h.flow_cycles[3] / (double)h.flow_cycles[5], 100.0 * (double)h.flow_cycles[4] / (double)h.flow_cycles[5],
                    (unsigned long long)h.flow_cycles[6], h.flow_cycles[6] ? (double)h.flow_cycles[7] / (double)h.flow_cycles[6] : 0.0);
        if (opt.debug_util && h.phase_cycles[3])
            fprintf(stderr, "[prt] k_pool wave time by phase: top-up %.1f%%, trace %.1f%%, shade %.1f%% of the main loop\n",
                    100.0 * (double)h.phase_cycles[0] / (double)h.phase_cycles[3], 100.0 * (double)h.phase_cycles[1] / (double)h.phase_cycles[3],
                    100.0 * (double)h.phase_cycles[2] / (double)h.phase_cycles[3]);
        if (opt.debug_util && h.wave_count)
            fprintf(stderr, "[prt] k_pool waves: %llu, main loop mean %.3f of the longest wave's (what the others idle at the end of the frame: %.1f%%)\n",
                    (unsigned long long)h.wave_count, (double)h.wave_cycles_sum / (double)h.wave_count / (double)h.wave_cycles_max,
                    100.0 * (1.0 - (double)h.wave_cycles_sum / (double)h.wave_count / (double)h.wave_cycles_max));
        if (opt.debug_util && ctx->wave_times_n && pipeline == PRT_PIPELINE_POOL) {
            // where the end of the frame goes: every wave's (start, sample counter dry, exit) on the 100 MHz wall clock
            std::vector<unsigned long long> wt(3u * (size_t)ctx->wave_times_n);
            if (hipMemcpy(wt.data(), ctx->wave_times.p, wt.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                unsigned long long t0 = ~0ull;
                for (size_t i = 0; i < wt.size(); i += 3) t0 = std::min(t0, wt[i]);
                std::vector<double> start, dry, end, chain;
                for (size_t i = 0; i < wt.size(); i += 3) {
                    start.push_back((double)(wt[i] - t0) * 1e-5);
                    end.push_back((double)(wt[i + 2] - t0) * 1e-5);
                    if (wt[i + 1]) { dry.push_back((double)(wt[i + 1] - t0) * 1e-5); chain.push_back((double)(wt[i + 2] - wt[i + 1]) * 1e-5); }
                }
                auto pct = [](std::vector<double> & v, double q) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[std::min(v.size() - 1, (size_t)(q * (double)v.size()))]; };
                fprintf(stderr, "[prt] k_pool waves on the wall clock (ms since the first wave started; last pass): start p50 %.3f max %.3f | counter dry p10 %.3f p50 %.3f p90 %.3f max %.3f | exit p10 %.3f p50 %.3f p90 %.3f p99 %.3f max %.3f | exit - dry p50 %.3f p90 %.3f max %.3f\n",
                        pct(start, 0.5), pct(start, 1.0), pct(dry, 0.1), pct(dry, 0.5), pct(dry, 0.9), pct(dry, 1.0),
                        pct(end, 0.1), pct(end, 0.5), pct(end, 0.9), pct(end, 0.99), pct(end, 1.0), pct(chain, 0.5), pct(chain, 0.9), pct(chain, 1.0));
            }
        }
        if (opt.debug_util && h.phase_cycles[3] && h.phase_cycles[4])
            fprintf(stderr, "[prt] k_pool adaptive finalise step (store the sample, variance rule, next camera ray): %.1f%% of the main loop\n",
                    100.0 * (double)h.phase_cycles[4] / (double)h.phase_cycles[3]);
        if (opt.debug_util && h.wave_node_steps)
            fprintf(stderr, "[prt] deepest stack %llu entries (LDS column %u, bound %u); %llu of %llu node visits (%.1f%%) hit no child after a hit was known\n",
                    (unsigned long long)h.max_sp, stack_entries, ctx->stack_bound, (unsigned long long)h.culled,
                    (unsigned long long)h.node_visits, 100.0 * (double)h.culled / (double)h.node_visits);
        memset(counters, 0, sizeof(*counters));
        counters->ray_count = h.ray_count;
        counters->node_visits = h.node_visits;
        counters->tri_tests = h.tri_tests;
        counters->shaded_hits = h.shaded_hits;
        counters->render_ms = ms;
        counters->trace_kernel_ms = trace_ms;
        counters->trace_kernel_launches = launches;
        counters->pipeline = pipeline;
    }
    return 0;
}

int render_pixels(prt_ctx * ctx, const prt_camera * cam_in, const prt_params * params, uint32_t width, uint32_t height,
                  const PixelSet & px, float4 * d_out, prt_counters * counters) {
    for (int attempt = 0; attempt < 4; ++attempt) {
        bool park_overflow = false;
        const int rc = render_pixels_once(ctx, cam_in, params, width, height, px, d_out, counters, &park_overflow);
        if (rc || !park_overflow) return rc;
    }
    ctx->error = "prt_render: the pool pipeline's park lists kept overflowing";
    return -7;
}

}  // namespace

// =============================================================================================================
// "Nothing aborts" (include/prt.h): the entry points below use std::vector / std::string / std::thread / new, and an
// exception that leaves an extern "C" function is std::terminate - an abort of the CALLER's process.  Every entry point
// that can allocate therefore runs its body inside PRT_API_TRY ... PRT_API_CATCH_*: std::bad_alloc, std::length_error,
// std::system_error and anything else become PRT_ERR_EXCEPTION (-12) with the message in prt_last_error.
namespace {
int api_exception(std::string * err, const char * where) noexcept {
    char what[256] = "unknown C++ exception";
    try { throw; }
    catch (const std::bad_alloc &) { snprintf(what, sizeof(what), "out of host memory (std::bad_alloc)"); }
    catch (const std::exception & e) { snprintf(what, sizeof(what), "C++ exception: %s", e.what()); }
    catch (...) {}
    try { if (err) *err = std::string(where) + ": " + what; } catch (...) {}      // the message itself may not fit: keep the code
    return PRT_ERR_EXCEPTION;
}
}  // namespace
#define PRT_API_TRY try {
#define PRT_API_CATCH_RC(ctx, where) } catch (...) { prt_ctx * c_ = (ctx); return api_exception(c_ ? &c_->error : &g_create_error, where); }
#define PRT_API_CATCH_RC_MULTI(m, where) } catch (...) { prt_multi * m_ = (m); return api_exception(m_ ? &m_->error : &g_create_error, where); }
#define PRT_API_CATCH_PTR(where) } catch (...) { (void)api_exception(&g_create_error, where); return nullptr; }
#define PRT_API_CATCH_VOID } catch (...) {}

extern "C" {

int prt_abi_version(void) { return PRT_ABI_VERSION; }

int prt_build_flags(void) {
    int f = 0;
#if defined(PRT_EXPERIMENTAL)
    f |= PRT_BUILD_EXPERIMENTAL;
#endif
#if !defined(PRT_BVH8)
    f |= PRT_BUILD_BVH4;
#endif
    return f;
}

const char * prt_last_error(const prt_ctx * ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

prt_ctx * prt_create(int device_id) {
    PRT_API_TRY
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = std::string("prt_create: no HIP device available (") + hipGetErrorString(e) + ")";
        return nullptr;
    }
    if (device_id < 0 || device_id >= n) {
        g_create_error = "prt_create: device ordinal out of range";
        return nullptr;
    }
    if ((e = hipSetDevice(device_id)) != hipSuccess) {
        g_create_error = std::string("prt_create: hipSetDevice: ") + hipGetErrorString(e);
        return nullptr;
    }
    prt_ctx * ctx = new prt_ctx;
    ctx->device = device_id;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) ctx->cu_count = prop.multiProcessorCount;
        if (ctx->cu_count <= 0) ctx->cu_count = 256;
    }
    memset(&ctx->scene, 0, sizeof(ctx->scene));
    memset(&ctx->info, 0, sizeof(ctx->info));
    memset(&ctx->last_stats, 0, sizeof(ctx->last_stats));
    prt_options_from_env(ctx->opt);       // the environment is read here and nowhere else
#if defined(PRT_BVH8_OCTANT)
    ctx->opt.bvh.slot_order = 0;          // the traversal of this build expects one slot per octant (dev_trace8.h)
#else
    ctx->opt.bvh.slot_order = 1;          // ... slots sorted along the node's ordering axis
#endif
    // PRT_RESERVE_CUS=k (multi-GPU callers): the context's streams are created with a CU mask that leaves k compute units to
    // others - the RCCL gather of the previous frame must not wait for a wave slot while this context's persistent kernels
    // hold every one of theirs (bench.py sets it for N > 1 with frames in flight).  The persistent grids are sized for the
    // CUs that are left.  WHICH k matters: the mask's bits are XCD-major (32 per XCD), workgroups go round the XCDs, so
    // clearing the last 8 bits takes a quarter of ONE XCD and leaves 35 of the grid's blocks without a slot there (C4 frame
    // +7.5 %); every 32nd bit takes one CU of each XCD (+4.2 % for 3.1 % of the CUs).  Measured: profiles/r03_cu_mask.txt.
    // A CU-masked stream is a BLOCKING stream (ordered against the null stream): callers keep their own work off the null
    // stream (bench.py does; the same file shows what happens otherwise).
    std::vector<uint32_t> cu_mask;
    {
        const int reserve = (int)ctx->opt.reserve_cus;
        if (reserve > 0 && reserve < ctx->cu_count) {
            cu_mask.assign((size_t)(ctx->cu_count + 31) / 32, 0u);
            const int pattern = (int)ctx->opt.reserve_pattern, stride = ctx->cu_count / reserve;
            int kept = 0;
            for (int cu = 0; cu < ctx->cu_count; ++cu) {
                const bool off = pattern == 1 ? cu < reserve
                               : pattern == 2 ? (cu % stride == stride - 1 && cu / stride < reserve)
                               : pattern == 3 ? false
                               : cu >= ctx->cu_count - reserve;
                if (!off) { cu_mask[(size_t)cu >> 5] |= 1u << (cu & 31); ++kept; }
            }
            ctx->reserved_cus = ctx->cu_count - kept;
            ctx->cu_count = kept;
        }
    }
    auto make_stream = [&](hipStream_t * st) -> hipError_t {
        if (!cu_mask.empty()) return hipExtStreamCreateWithCUMask(st, (uint32_t)cu_mask.size(), cu_mask.data());
        return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    };
    e = make_stream(&ctx->stream);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&ctx->ev[i]);
    ctx->chain[0].stream = ctx->stream;
    if (ctx->opt.pool_park_cap >= 0) ctx->pool_park_cap = ctx->pool_spark_cap = (size_t)std::max(1ll, ctx->opt.pool_park_cap);   // tests: start tiny, grow
    if (e == hipSuccess) e = ctx->counters.ensure(1);          // DevScene::near_tie_unresolved points into it
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->host_counters, sizeof(DevCounters), hipHostMallocDefault);
    for (int c = 1; c < PRT_MAX_CHAINS && e == hipSuccess; ++c) e = make_stream(&ctx->chain[c].stream);
    for (int c = 0; c < PRT_MAX_CHAINS && e == hipSuccess; ++c) {
        prt_ctx::ChainWs & w = ctx->chain[c];
        e = hipEventCreate(&w.ev_t0);
        if (e == hipSuccess) e = hipEventCreate(&w.ev_t1);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&w.ev_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&w.ev_first, hipEventDisableTiming);
        if (e == hipSuccess) e = hipHostMalloc((void **)&w.host_counts, 64, hipHostMallocDefault);
    }
    if (e != hipSuccess) {
        g_create_error = std::string("prt_create: stream/event creation: ") + hipGetErrorString(e);
        prt_destroy(ctx);                  // releases whatever was created before the failure (every handle starts out null)
        return nullptr;
    }
    return ctx;
    PRT_API_CATCH_PTR("prt_create")
}

void prt_destroy(prt_ctx * ctx) {
    PRT_API_TRY
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->nodes.release(); ctx->tris.release(); ctx->shade.release(); ctx->diffuse_dirs.release(); ctx->spec_dirs.release();
    ctx->tri_rank.release(); ctx->materials.release(); ctx->lights.release();
    ctx->sample_rgb.release(); ctx->frame_out.release(); ctx->counters.release(); ctx->ring_ws.release(); ctx->pixel_list.release(); ctx->wf_counts.release(); ctx->stack_spill.release(); ctx->pool_f4.release(); ctx->pool_park.release(); ctx->pool_fin.release(); ctx->pool_args.release(); ctx->adapt_f4.release(); ctx->wave_times.release(); ctx->pool_xchg.release();
#if defined(PRT_FLOW_EXPERIMENT)
    ctx->flow_args.release(); ctx->flow_u32.release();
#endif
   
    ctx->textures.release(); ctx->texels.release(); ctx->srgb_lut.release(); ctx->tri_uv.release(); ctx->tri_tan.release();
    ctx->ref_spheres.release();
    for (int c = 0; c < PRT_MAX_CHAINS; ++c) {
        prt_ctx::ChainWs & w = ctx->chain[c];
        w.f4.release(); w.rng.release(); w.counts.release(); w.overflow.release(); w.slow_stack.release();
        if (w.ev_t0) (void)hipEventDestroy(w.ev_t0);
        if (w.ev_t1) (void)hipEventDestroy(w.ev_t1);
        if (w.ev_done) (void)hipEventDestroy(w.ev_done);
        if (w.ev_first) (void)hipEventDestroy(w.ev_first);
        if (w.host_counts) (void)hipHostFree(w.host_counts);
        if (c >= 1 && w.stream) (void)hipStreamDestroy(w.stream);
    }
    for (int i = 0; i < 4; ++i) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->host_counters) (void)hipHostFree(ctx->host_counters);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    PRT_API_CATCH_VOID
}

int prt_set_option(prt_ctx * ctx, const char * name, const char * value) {
    PRT_API_TRY
    if (!ctx || !name) return -1;
    const char * bare = !strncasecmp(name, "PRT_", 4) ? name + 4 : name;
    if (!strcasecmp(bare, "RESERVE_CUS")) { ctx->error = "prt_set_option: RESERVE_CUS shapes the context's streams and is read at creation only (environment PRT_RESERVE_CUS)"; return -1; }
    if (prt_option_set(ctx->opt, bare, value)) { ctx->error = std::string("prt_set_option: unknown option or bad value: ") + name; return -1; }
    ctx->opt.bvh.debug = ctx->opt.debug_util;
    if (!strcasecmp(bare, "POOL_PARK_CAP")) {
        ctx->pool_park_cap = ctx->pool_spark_cap = ctx->opt.pool_park_cap >= 0 ? (size_t)std::max(1ll, ctx->opt.pool_park_cap) : (size_t)1 << 18;
    }
    ctx->tuned.clear();                 // a knob may change which pipeline wins the try-out
    return 0;
    PRT_API_CATCH_RC(ctx, "prt_set_option")
}

int prt_get_render_stats(const prt_ctx * ctx, prt_render_stats * stats) {
    PRT_API_TRY
    if (!ctx || !stats) return -1;
    *stats = ctx->last_stats;
    return 0;
    PRT_API_CATCH_RC(nullptr, "prt_get_render_stats")
}

int prt_upload_scene(prt_ctx * ctx, const prt_scene_desc * s) {
    PRT_API_TRY
    if (!ctx) return -1;
    if (!s || (s->index_count % 3) != 0) { ctx->error = "prt_upload_scene: null scene or index_count not a multiple of 3"; return -1; }
    if (s->material_count == 0) { ctx->error = "prt_upload_scene: at least one material is required"; return -1; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->has_scene = false;
    const uint32_t n_tris = s->index_count / 3;
    if (n_tris >= (1u << 26)) { ctx->error = "prt_upload_scene: 2^26 triangles or more (the traversal addresses nodes and triangles by 32-bit byte offsets)"; return -1; }

    // per-triangle group / material lookup + validation of every index the kernels will follow
    std::vector<int32_t> tri_material(n_tris, 0);
    std::vector<uint8_t> covered(n_tris, 0);
    for (uint32_t g = 0; g < s->group_count; ++g) {
        const prt_group & pg = s->groups[g];
        if (pg.first_index % 3 || pg.index_count % 3 || (uint64_t)pg.first_index + pg.index_count > s->index_count) {
            ctx->error = "prt_upload_scene: group index range is not a whole run of triangles inside the index buffers";
            return -1;
        }
        if (pg.material < 0 || (uint32_t)pg.material >= s->material_count) { ctx->error = "prt_upload_scene: group material out of range"; return -1; }
        for (uint32_t t = pg.first_index / 3; t < (pg.first_index + pg.index_count) / 3; ++t) { tri_material[t] = pg.material; covered[t] = 1; }
    }
    for (uint32_t t = 0; t < n_tris; ++t)
        if (!covered[t]) { ctx->error = "prt_upload_scene: triangle not covered by any group"; return -1; }
    for (uint32_t i = 0; i < s->index_count; ++i) {
        if (s->idx_positions[i] >= s->position_count || s->idx_normals[i] >= s->normal_count ||
            s->idx_texcoords[i] >= s->texcoord_count) {
            ctx->error = "prt_upload_scene: vertex index out of range (the reference would read out of bounds here)";
            return -1;
        }
    }
    bool textured = false, bumped = false;
    for (uint32_t m = 0; m < s->material_count; ++m) {
        const prt_material & pm = s->materials[m];
        const int32_t slots[5] = { pm.ambient_texture, pm.diffuse_texture, pm.specular_texture, pm.alpha_texture, pm.bump_texture };
        for (int k = 0; k < 5; ++k) {
            if (slots[k] < 0) continue;
            textured = true;
            if (k == 4) bumped = true;
            if (!s->textures || (uint32_t)slots[k] >= s->texture_count) { ctx->error = "prt_upload_scene: material texture slot out of range"; return -1; }
            const prt_texture & t = s->textures[slots[k]];
            // size - 2 is the sampling scale (texture.cpp:63-64): below 2 it wraps around as u32 in the reference
            if (!t.texels || t.size_x < 2 || t.size_y < 2 || t.channels < 1 || t.channels > 4) {
                ctx->error = "prt_upload_scene: texture needs texels, 1..4 channels and at least 2 x 2 texels";
                return -1;
            }
        }
    }
    if (textured && s->texture_count >= DEV_TEX_NONE) { ctx->error = "prt_upload_scene: more than 65534 textures"; return -1; }
    if (bumped && !s->tangents) { ctx->error = "prt_upload_scene: a material has a bump map but the scene has no tangents"; return -1; }

    // ---- BVH over un-indexed triangles
    auto t0 = std::chrono::steady_clock::now();
    std::vector<float> verts((size_t)n_tris * 9);
    for (uint32_t t = 0; t < n_tris; ++t)
        for (int c = 0; c < 3; ++c) memcpy(&verts[(size_t)t * 9 + 3 * c], s->positions + 3 * (size_t)s->idx_positions[3 * t + c], 12);
    for (size_t i = 0; i < verts.size(); ++i)
        if (!(fabsf(verts[i]) < 1e18f)) { ctx->error = "prt_upload_scene: vertex coordinate is not finite or exceeds 1e18"; return -1; }
    BvhWide bvh;
    unsigned int hw = std::max(1u, std::thread::hardware_concurrency());
    unsigned int leaf_max = BVH_LEAF_MAX;
    if (ctx->opt.leaf_max >= 0) leaf_max = (unsigned int)std::max(1ll, std::min(4ll, ctx->opt.leaf_max));   // experiment knob
    const float trav_cost = (float)ctx->opt.sah_trav_cost;                                                  // experiment knob
    if (ctx->opt.bvh_builder_lbvh) {
        // fast build for scenes that change every frame: radix tree on the GPU (bvh_lbvh.h), same quantised wide back end
        int rc = build_bvh_lbvh(ctx, verts.data(), n_tris, leaf_max, &bvh);
        if (rc) return rc;
    } else {
        PRT_BUILD_WIDE(verts.data(), n_tris, leaf_max, std::min(hw, 16u), &bvh, trav_cost, &ctx->opt.bvh);
    }
    double build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (const char * bad = validate_bvh_links(bvh, n_tris)) { ctx->error = std::string("prt_upload_scene: the builder produced a broken tree - ") + bad; return -9; }

    // ---- reference visit rank: leaves of the sphere tree in the order TraceRay pops them (c1 first)
    std::vector<uint32_t> rank_of_input(n_tris);
    std::vector<float4> ref_spheres;                          // (centre, radius) (c0, c1, first rank of c0's triangles, pop order)
    {
        std::vector<uint32_t> group_base(s->group_count, 0);
        bool ranked = false;
        if (s->spheres && s->sphere_group && s->sphere_count) {
            std::vector<uint32_t> stack(1, 0u);
            std::vector<uint8_t> seen(s->group_count, 0);
            std::vector<uint32_t> first_rank(s->sphere_count, 0), pop_order(s->sphere_count, 0);
            std::vector<uint8_t> popped(s->sphere_count, 0);
            uint32_t next = 0, visited = 0;
            bool ok = true;
            while (!stack.empty() && ok) {
                uint32_t i = stack.back();
                stack.pop_back();
                if (i >= s->sphere_count || ++visited > 2 * s->sphere_count || popped[i]) { ok = false; break; }
                popped[i] = 1;
                pop_order[i] = visited - 1;
                first_rank[i] = next;                         // the first triangle the reference meets below this sphere
                const prt_bsphere & bs = s->spheres[i];
                if (bs.c0 && bs.c1) {
                    stack.push_back(bs.c0);
                    stack.push_back(bs.c1);
                } else {
                    int32_t g = s->sphere_group[i];
                    if (g < 0 || (uint32_t)g >= s->group_count || seen[g]) { ok = false; break; }
                    seen[g] = 1;
                    group_base[g] = next;
                    next += s->groups[g].index_count / 3;
                }
            }
            ranked = ok && next == n_tris;
            if (ranked) {
                ref_spheres.resize(2 * (size_t)s->sphere_count, make_float4(0, 0, 0, 0));
                for (uint32_t i = 0; i < s->sphere_count; ++i) {
                    const prt_bsphere & bs = s->spheres[i];
                    ref_spheres[2 * (size_t)i] = make_float4(bs.center[0], bs.center[1], bs.center[2], bs.radius);
                    const bool inner = bs.c0 && bs.c1 && popped[i];
                    const uint32_t w[4] = { inner ? bs.c0 : 0u, inner ? bs.c1 : 0u, inner ? first_rank[bs.c0] : 0u, pop_order[i] };
                    float4 f;
                    memcpy(&f, w, 16);
                    ref_spheres[2 * (size_t)i + 1] = f;
                }
            }
        }
        if (ranked) {
            for (uint32_t g = 0; g < s->group_count; ++g) {
                uint32_t first = s->groups[g].first_index / 3, cnt = s->groups[g].index_count / 3;
                for (uint32_t k = 0; k < cnt; ++k) rank_of_input[first + k] = group_base[g] + k;
            }
        } else {
            for (uint32_t t = 0; t < n_tris; ++t) rank_of_input[t] = t;
        }
    }

    // ---- device records in BVH leaf order
    const uint32_t n_rec = n_tris + 1;                              // + the all-zero dummy triangle empty BVH slots point at
    std::vector<float4> tris((size_t)n_rec * 3, make_float4(0, 0, 0, 0)), shade((size_t)n_rec * 4, make_float4(0, 0, 0, 0));
    std::vector<unsigned int> rank(n_rec, 0);
    std::vector<float4> tri_uv(textured ? (size_t)n_rec * 2 : 0, make_float4(0, 0, 0, 0));
    std::vector<float4> tri_tan(bumped ? (size_t)n_rec * 3 : 0, make_float4(0, 0, 0, 0));
    float abs_max = 0.0f;
    for (uint32_t slot = 0; slot < n_tris; ++slot) {
        const uint32_t t = bvh.tri_order[slot];
        const f3 a = ld3(&verts[(size_t)t * 9]), b = ld3(&verts[(size_t)t * 9 + 3]), c = ld3(&verts[(size_t)t * 9 + 6]);
        const f3 ab = b - a, ac = c - a;                          // raytracer.cpp:85-86
        const f3 n = cross3(ab, ac);                              // raytracer.cpp:91
        tris[(size_t)slot * 3 + 0] = make_float4(a.x, a.y, a.z, ab.x);
        tris[(size_t)slot * 3 + 1] = make_float4(ab.y, ab.z, ac.x, ac.y);
        tris[(size_t)slot * 3 + 2] = make_float4(ac.z, n.x, n.y, n.z);
        const float * n0 = s->normals + 3 * (size_t)s->idx_normals[3 * t + 0];
        const float * n1 = s->normals + 3 * (size_t)s->idx_normals[3 * t + 1];
        const float * n2 = s->normals + 3 * (size_t)s->idx_normals[3 * t + 2];
        const float * u0 = s->texcoords + 2 * (size_t)s->idx_texcoords[3 * t + 0];
        const float * u1 = s->texcoords + 2 * (size_t)s->idx_texcoords[3 * t + 1];
        const float * u2 = s->texcoords + 2 * (size_t)s->idx_texcoords[3 * t + 2];
        float mbits;
        int32_t mi = tri_material[t];
        memcpy(&mbits, &mi, 4);
        shade[(size_t)slot * 4 + 0] = make_float4(n0[0], n0[1], n0[2], n1[0]);
        shade[(size_t)slot * 4 + 1] = make_float4(n1[1], n1[2], n2[0], n2[1]);
        // the geometric normal n = Cross(ab, ac) rides in the shading record too, so a shaded hit costs ONE 64 B
        // gather instead of two; texture coordinates and tangents have their own arrays, only for textured scenes
        shade[(size_t)slot * 4 + 2] = make_float4(n2[2], n.x, n.y, n.z);
        shade[(size_t)slot * 4 + 3] = make_float4(0.0f, 0.0f, 0.0f, mbits);
        if (textured) {
            tri_uv[(size_t)slot * 2 + 0] = make_float4(u0[0], u0[1], u1[0], u1[1]);
            tri_uv[(size_t)slot * 2 + 1] = make_float4(u2[0], u2[1], 0.0f, 0.0f);
        }
        if (bumped) {                                             // mesh->tangents[idx_normals[...]], raytracer.cpp:470-473
            const float * g0 = s->tangents + 3 * (size_t)s->idx_normals[3 * t + 0];
            const float * g1 = s->tangents + 3 * (size_t)s->idx_normals[3 * t + 1];
            const float * g2 = s->tangents + 3 * (size_t)s->idx_normals[3 * t + 2];
            tri_tan[(size_t)slot * 3 + 0] = make_float4(g0[0], g0[1], g0[2], g1[0]);
            tri_tan[(size_t)slot * 3 + 1] = make_float4(g1[1], g1[2], g2[0], g2[1]);
            tri_tan[(size_t)slot * 3 + 2] = make_float4(g2[2], 0.0f, 0.0f, 0.0f);
        }
        rank[slot] = rank_of_input[t];
        for (int k = 0; k < 9; ++k) abs_max = std::max(abs_max, fabsf(verts[(size_t)t * 9 + k]));
    }

    std::vector<DevMaterial> mats(s->material_count);
    ctx->material_ns.resize(s->material_count);
    ctx->any_translucent = false;
    for (uint32_t m = 0; m < s->material_count; ++m) {
        const prt_material & pm = s->materials[m];
        DevMaterial & d = mats[m];
        memset(&d, 0, sizeof(d));
        for (int k = 0; k < 3; ++k) { d.ambient[k] = pm.ambient_color[k]; d.diffuse[k] = pm.diffuse_color[k]; d.specular[k] = pm.specular_color[k]; }
        d.specular_intensity = pm.specular_intensity;
        d.index_of_refraction = pm.index_of_refraction;
        d.alpha = pm.alpha;
        auto slot16 = [](int32_t v) { return v < 0 ? (unsigned int)DEV_TEX_NONE : (unsigned int)v; };
        d.tex[0] = slot16(pm.ambient_texture) | slot16(pm.diffuse_texture) << 16;
        d.tex[1] = slot16(pm.specular_texture) | slot16(pm.alpha_texture) << 16;
        d.tex[2] = slot16(pm.bump_texture) | (unsigned int)DEV_TEX_NONE << 16;
        ctx->material_ns[m] = pm.specular_intensity;
        if (!(pm.alpha >= 1.0f)) ctx->any_translucent = true;      // alpha < 1 (or NaN): continuation rays, unbounded draws
    }
    // textures: RGBA8 by GetTexel's channel rules (texture.cpp:27-41) + the 256 values Color_SRGBToLinear can take
    std::vector<DevTexture> dev_tex;
    std::vector<unsigned int> texel_pool;
    std::vector<float> lut;
    if (textured) {
        dev_tex.resize(s->texture_count);
        for (uint32_t ti = 0; ti < s->texture_count; ++ti) {
            const prt_texture & t = s->textures[ti];
            DevTexture & d = dev_tex[ti];
            memset(&d, 0, sizeof(d));
            if (!t.texels || t.size_x < 2 || t.size_y < 2 || t.channels < 1 || t.channels > 4) continue;     // unreferenced and unusable
            const size_t n = (size_t)t.size_x * t.size_y;
            if (texel_pool.size() + n >= (1ull << 32)) { ctx->error = "prt_upload_scene: more than 2^32 texels"; return -1; }
            d.first_texel = (unsigned int)texel_pool.size();
            d.size_x = t.size_x;
            d.size_y = t.size_y;
            texel_pool.resize(texel_pool.size() + n);
            unsigned int * out = texel_pool.data() + d.first_texel;
            for (size_t i = 0; i < n; ++i) {
                const uint8_t * px = t.texels + i * t.channels;
                unsigned int r = px[0], g = t.channels >= 2 ? px[1] : 0u, b = t.channels >= 3 ? px[2] : 0u, a = t.channels >= 4 ? px[3] : 255u;
                if (t.channels == 1) b = g = r;
                out[i] = r | g << 8 | b << 16 | a << 24;
            }
        }
        lut.resize(256);
        const float one_over_255 = 1.0f / 255.0f;                  // texture.cpp:15
        for (int i = 0; i < 256; ++i) {
            const float srgb = (float)i * one_over_255;
            lut[i] = srgb <= 0.04045f ? srgb / 12.92f : powf((srgb + 0.055f) / 1.055f, 2.4f);     // color.h:13-21
        }
    }
    std::vector<DevLight> lights(std::max(1u, s->light_count));
    memset(lights.data(), 0, lights.size() * sizeof(DevLight));
    for (uint32_t l = 0; l < s->light_count; ++l) {
        const prt_light & pl = s->lights[l];
        DevLight & d = lights[l];
        d.type = pl.type;
        for (int k = 0; k < 3; ++k) { d.color[k] = pl.color[k]; d.position[k] = pl.position[k]; d.facing[k] = pl.facing[k]; }
        d.falloff = pl.falloff;
    }
    ctx->point_lights = false;
    for (uint32_t l = 0; l < s->light_count; ++l) ctx->point_lights = ctx->point_lights || s->lights[l].type == PRT_LIGHT_POINT;
    std::vector<float4> ddirs(1024);
    for (uint32_t i = 0; i < 1024; ++i) ddirs[i] = diffuse_tangent_dir(i);

#if defined(PRT_BVH8) && PRT_BVH8_STRIDE != 80
    std::vector<float4> nodes4((size_t)bvh.node_count * (BVH_NODE_STRIDE / 16), make_float4(0, 0, 0, 0));
    for (uint32_t ni = 0; ni < bvh.node_count; ++ni)
        memcpy(reinterpret_cast<char *>(nodes4.data()) + (size_t)ni * BVH_NODE_STRIDE, &bvh.nodes[(size_t)ni * BVH8_NODE_DWORDS], BVH8_NODE_DWORDS * 4);
#else
    std::vector<float4> nodes4(bvh.nodes.size() / 4);
    memcpy(nodes4.data(), bvh.nodes.data(), bvh.nodes.size() * sizeof(uint32_t));
#endif
    ctx->stack_bound = bvh.stack_bound;

    HIP_TRY(ctx, ctx->nodes.upload(nodes4));
    HIP_TRY(ctx, ctx->tris.upload(tris));
    HIP_TRY(ctx, ctx->shade.upload(shade));
    HIP_TRY(ctx, ctx->tri_rank.upload(rank));
    HIP_TRY(ctx, ctx->materials.upload(mats));
    HIP_TRY(ctx, ctx->lights.upload(lights));
    HIP_TRY(ctx, ctx->diffuse_dirs.upload(ddirs));
    if (!ref_spheres.empty()) HIP_TRY(ctx, ctx->ref_spheres.upload(ref_spheres));
    ctx->textured = textured;
    if (textured) {
        HIP_TRY(ctx, ctx->textures.upload(dev_tex));
        HIP_TRY(ctx, ctx->texels.upload(texel_pool));
        HIP_TRY(ctx, ctx->srgb_lut.upload(lut));
        HIP_TRY(ctx, ctx->tri_uv.upload(tri_uv));
        if (bumped) HIP_TRY(ctx, ctx->tri_tan.upload(tri_tan));
    }

    DevScene & sc = ctx->scene;
    sc.nodes = ctx->nodes.p;
    sc.tris = ctx->tris.p;
    sc.shade = ctx->shade.p;
    sc.tri_rank = ctx->tri_rank.p;
    sc.materials = ctx->materials.p;
    sc.lights = ctx->lights.p;
    sc.diffuse_dirs = ctx->diffuse_dirs.p;
    sc.light_count = s->light_count;
    sc.material_count = s->material_count;
    sc.tri_count = n_tris;
    sc.node_count = bvh.node_count;
    sc.textures = textured ? ctx->textures.p : nullptr;
    sc.texels = textured ? ctx->texels.p : nullptr;
    sc.srgb_lut = textured ? ctx->srgb_lut.p : nullptr;
    sc.tri_uv = textured ? ctx->tri_uv.p : nullptr;
    sc.tri_tan = bumped ? ctx->tri_tan.p : nullptr;
    sc.ref_spheres = ref_spheres.empty() ? nullptr : ctx->ref_spheres.p;
    sc.tie_widen_max = 8;                                     // render_pixels_once sets the option's value per call
    sc.near_tie_unresolved = &ctx->counters.p->near_tie_unresolved;
    ctx->spec_table_samples = 0;
    int rc = build_spec_table(ctx, 1);
    if (rc) return rc;

    ctx->scene_abs_max = abs_max;
    prt_scene_info & info = ctx->info;
    info.triangle_count = n_tris;
    info.bvh_node_count = bvh.node_count;
    info.bvh_max_depth = bvh.max_depth;
    info.bvh_node_bytes = BVH_NODE_BYTES;
    info.tri_record_bytes = 48;
    info.shade_record_bytes = 64;
    info.device_bytes = ctx->nodes.bytes() + ctx->tris.bytes() + ctx->shade.bytes() + ctx->tri_rank.bytes() +
                        ctx->materials.bytes() + ctx->lights.bytes() + ctx->diffuse_dirs.bytes() + ctx->spec_dirs.bytes() +
                        (textured ? ctx->textures.bytes() + ctx->texels.bytes() + ctx->srgb_lut.bytes() + ctx->tri_uv.bytes() : 0) +
                        (bumped ? ctx->tri_tan.bytes() : 0) + (ref_spheres.empty() ? 0 : ctx->ref_spheres.bytes());
    info.bvh_build_ms = build_ms;
    HIP_TRY(ctx, hipDeviceSynchronize());      // uploads went through the null stream; renders use the context's non-blocking streams
    ctx->has_scene = true;
    ctx->scene_epoch++;
    ctx->tuned.clear();
    return 0;
    PRT_API_CATCH_RC(ctx, "prt_upload_scene")
}

int prt_get_scene_info(const prt_ctx * ctx, prt_scene_info * info) {
    PRT_API_TRY
    if (!ctx || !info) return -1;
    if (!ctx->has_scene) return -2;
    *info = ctx->info;
    return 0;
    PRT_API_CATCH_RC(nullptr, "prt_get_scene_info")
}

int prt_render_device(prt_ctx * ctx, const prt_camera * cam, const prt_params * params, uint32_t width, uint32_t height,
                      uint32_t start_idx, uint32_t end_idx, void * d_rgba_out, prt_counters * counters) {
    PRT_API_TRY
    if (!ctx) return -1;
    if ((!d_rgba_out && end_idx != start_idx) || end_idx < start_idx || (uint64_t)end_idx > (uint64_t)width * height) {
        ctx->error = "prt_render: bad output pointer or pixel range";
        return -1;
    }
    PixelSet px = { end_idx - start_idx, start_idx, 1, 0, 1, nullptr };
    return render_pixels(ctx, cam, params, width, height, px, (float4 *)d_rgba_out, counters);
    PRT_API_CATCH_RC(ctx, "prt_render_device")
}

int prt_render(prt_ctx * ctx, const prt_camera * cam, const prt_params * params, uint32_t width, uint32_t height,
               uint32_t start_idx, uint32_t end_idx, float * rgba_out, prt_counters * counters) {
    PRT_API_TRY
    if (!ctx) return -1;
    if ((!rgba_out && end_idx != start_idx) || end_idx < start_idx) { ctx->error = "prt_render: bad output pointer or pixel range"; return -1; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n = end_idx - start_idx;
    HIP_TRY(ctx, ctx->frame_out.ensure(n));
    int rc = prt_render_device(ctx, cam, params, width, height, start_idx, end_idx, ctx->frame_out.p, counters);
    if (rc) return rc;
    if (n) HIP_TRY(ctx, hipMemcpy(rgba_out, ctx->frame_out.p, n * sizeof(float4), hipMemcpyDeviceToHost));
    return 0;
    PRT_API_CATCH_RC(ctx, "prt_render")
}

uint32_t prt_shard_rows(uint32_t height, uint32_t block_rows, uint32_t rank, uint32_t nranks) {
    if (!block_rows || !nranks || rank >= nranks) return 0;
    uint32_t rows = 0;
    for (uint64_t b = rank; b * block_rows < height; b += nranks) rows += std::min<uint64_t>(block_rows, height - b * block_rows);
    return rows;
}

int prt_render_shard_device(prt_ctx * ctx, const prt_camera * cam, const prt_params * params, uint32_t width, uint32_t height,
                            uint32_t block_rows, uint32_t rank, uint32_t nranks, void * d_rgba_out, prt_counters * counters) {
    PRT_API_TRY
    if (!ctx) return -1;
    if (!block_rows || !nranks || rank >= nranks) { ctx->error = "prt_render_shard: bad arguments"; return -1; }
    if (!d_rgba_out && prt_shard_rows(height, block_rows, rank, nranks)) { ctx->error = "prt_render_shard: null output"; return -1; }
    PixelSet px = { prt_shard_rows(height, block_rows, rank, nranks) * width, 0, block_rows, rank, nranks, nullptr };
    if (nranks == 1) { px.block_rows = 1; }
    return render_pixels(ctx, cam, params, width, height, px, (float4 *)d_rgba_out, counters);
    PRT_API_CATCH_RC(ctx, "prt_render_shard_device")
}

int prt_render_shard(prt_ctx * ctx, const prt_camera * cam, const prt_params * params, uint32_t width, uint32_t height,
                     uint32_t block_rows, uint32_t rank, uint32_t nranks, float * rgba_out, prt_counters * counters) {
    PRT_API_TRY
    if (!ctx) return -1;
    const size_t n = (size_t)prt_shard_rows(height, block_rows, rank, nranks) * width;
    if (!rgba_out && n) { ctx->error = "prt_render_shard: null output"; return -1; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, ctx->frame_out.ensure(n));
    int rc = prt_render_shard_device(ctx, cam, params, width, height, block_rows, rank, nranks, ctx->frame_out.p, counters);
    if (rc) return rc;
    if (n) HIP_TRY(ctx, hipMemcpy(rgba_out, ctx->frame_out.p, n * sizeof(float4), hipMemcpyDeviceToHost));
    return 0;
    PRT_API_CATCH_RC(ctx, "prt_render_shard")
}

int prt_render_pixel_list(prt_ctx * ctx, const prt_camera * cam, const prt_params * params, uint32_t width, uint32_t height,
                          const uint32_t * pixel_ids, uint32_t n_pixels, float * rgba_out, prt_counters * counters) {
    PRT_API_TRY
    if (!ctx) return -1;
    if ((!pixel_ids || !rgba_out) && n_pixels) { ctx->error = "prt_render_pixel_list: null pixel list or output"; return -1; }
    for (uint32_t i = 0; i < n_pixels; ++i)
        if ((uint64_t)pixel_ids[i] >= (uint64_t)width * height) { ctx->error = "prt_render_pixel_list: pixel id outside the image"; return -1; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, ctx->pixel_list.ensure(n_pixels));
    HIP_TRY(ctx, ctx->frame_out.ensure(n_pixels));
    if (n_pixels) HIP_TRY(ctx, hipMemcpyAsync(ctx->pixel_list.p, pixel_ids, (size_t)n_pixels * 4, hipMemcpyHostToDevice, ctx->stream));
    if (n_pixels) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // pixel_ids is the caller's memory
    PixelSet px = { n_pixels, 0, 1, 0, 1, ctx->pixel_list.p };
    int rc = render_pixels(ctx, cam, params, width, height, px, ctx->frame_out.p, counters);
    if (rc) return rc;
    if (n_pixels) HIP_TRY(ctx, hipMemcpy(rgba_out, ctx->frame_out.p, (size_t)n_pixels * sizeof(float4), hipMemcpyDeviceToHost));
    return 0;
    PRT_API_CATCH_RC(ctx, "prt_render_pixel_list")
}

int prt_debug_device_kat(prt_ctx * ctx, int kind, const void * in, size_t in_bytes, void * out, size_t out_bytes, uint32_t n,
                         const prt_camera * cam_in) {
    PRT_API_TRY
    if (!ctx) return -1;
    if (!ctx->has_scene) { ctx->error = "prt_debug_device_kat: upload a scene first"; return -2; }
    if (!in || !out || !n) { ctx->error = "prt_debug_device_kat: null buffers"; return -1; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // scratch buffers owned by DevBuf objects: released on every return path
    struct Scratch {
        DevBuf<unsigned char> in, out;
        ~Scratch() { in.release(); out.release(); }
    } scratch;
    HIP_TRY(ctx, scratch.in.ensure(in_bytes));
    HIP_TRY(ctx, scratch.out.ensure(out_bytes));
    void * d_in = scratch.in.p;
    void * d_out = scratch.out.p;
    HIP_TRY(ctx, ctx->ring_ws.ensure((size_t)16 * n));
    // everything on the context's stream: it is a non-blocking stream, so work on the null stream (a plain hipMemset)
    // is NOT ordered against the kernel below
    HIP_TRY(ctx, hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(d_out, 0, out_bytes, ctx->stream));
    DevCamera cam;
    memset(&cam, 0, sizeof(cam));
    if (cam_in) {
        cam.position = ld3(cam_in->position);
        cam.forward = ld3(cam_in->forward);
        cam.right_scaled = ld3(cam_in->right) * cam_in->tan_a2 * cam_in->aspect;
        cam.up_scaled = ld3(cam_in->up) * cam_in->tan_a2;
        cam.inv_width = cam_in->inv_width;
        cam.inv_height = cam_in->inv_height;
    }
    hipLaunchKernelGGL(k_debug_kat, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, kind, d_in, d_out, n, ctx->scene, cam, ctx->ring_ws.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost));
    return 0;
    PRT_API_CATCH_RC(ctx, "prt_debug_device_kat")
}

// Host-only self check of the acceleration structure (no GPU needed; used by the CPU test-suite): builds the
// quantised 4-wide BVH for `scene` exactly as prt_upload_scene does and verifies that every triangle lies
// inside the de-quantised box of every ancestor, that every triangle is referenced by exactly one leaf and
// that links are in range.  out[0] = violations, out[1] = nodes, out[2] = depth, out[3] = stack bound,
// out[4] = leaves, out[5] = triangles referenced.
static int check_bvh4q(const std::vector<float> & verts, uint32_t n_tris, const Bvh4Result & bvh, uint64_t * out) {
    uint64_t violations = 0, leaves = 0, refs = 0;
    std::vector<uint8_t> seen(std::max(1u, n_tris), 0);
    struct Item { uint32_t node; float lo[3], hi[3]; };
    std::vector<Item> stack;
    Item root;
    root.node = 0;
    for (int a = 0; a < 3; ++a) { root.lo[a] = -3.0e38f; root.hi[a] = 3.0e38f; }
    stack.push_back(root);
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        if (it.node >= bvh.node_count) { violations++; continue; }
        const uint32_t * d = &bvh.nodes[(size_t)it.node * 16];
        float org[3], scale[3];
        const int scale_dword[3] = { 3, 14, 15 };
        for (int a = 0; a < 3; ++a) {
            memcpy(&org[a], &d[a], 4);
            memcpy(&scale[a], &d[scale_dword[a]], 4);
            if ((d[scale_dword[a]] & 0x807FFFFFu) != 0u || d[scale_dword[a]] == 0u) violations++;      // a positive power of two
        }
        uint32_t count = 0;                                  // children come first; an empty slot has inverted planes on every axis
        while (count < 4 && !(((d[4] >> (8 * count)) & 0xFFu) == 255u && ((d[7] >> (8 * count)) & 0xFFu) == 0u)) ++count;
        if (count < 1) violations++;
        for (uint32_t k = 0; k < 4; ++k) {
            Item ch;
            if (k >= count) {
                // empty slot: inverted box on every axis and a link to the dummy leaf
                const int32_t el = (int32_t)d[10 + k];
                if (el >= 0 || ((uint32_t)~el >> 2) != n_tris) violations++;
                for (int a = 0; a < 3; ++a) if (((d[4 + a] >> (8 * k)) & 0xFFu) != 255u || ((d[7 + a] >> (8 * k)) & 0xFFu) != 0u) violations++;
                continue;
            }
            for (int a = 0; a < 3; ++a) {
                ch.lo[a] = std::max(it.lo[a], org[a] + (float)((d[4 + a] >> (8 * k)) & 0xFFu) * scale[a]);
                ch.hi[a] = std::min(it.hi[a], org[a] + (float)((d[7 + a] >> (8 * k)) & 0xFFu) * scale[a]);
            }
            const int32_t link = (int32_t)d[10 + k];
            if (link >= 0) {
                ch.node = (uint32_t)link;
                stack.push_back(ch);
            } else {
                const uint32_t leaf = (uint32_t)~link, first = leaf >> 2, cnt = (leaf & 3u) + 1u;
                leaves++;
                for (uint32_t i = 0; i < cnt; ++i) {
                    const uint32_t slot = first + i;
                    if (slot == n_tris) continue;                       // the dummy triangle of an empty slot
                    if (slot > n_tris) { violations++; continue; }
                    refs++;
                    if (seen[slot]++) violations++;
                    const uint32_t t = bvh.tri_order[slot];
                    for (int c = 0; c < 3; ++c)
                        for (int a = 0; a < 3; ++a) {
                            const float v = verts[(size_t)t * 9 + 3 * c + a];
                            // de-quantised planes may round by an ulp of the coordinate; the kernels widen every box
                            // by 2^-16 of the scene extent, far more than that
                            const float tol = 4.0f * 1.1920929e-7f * std::max(1.0f, fabsf(v));
                            if (v < ch.lo[a] - tol || v > ch.hi[a] + tol) violations++;
                        }
                }
            }
        }
    }
    for (uint32_t t = 0; t < n_tris; ++t) if (!seen[t]) violations++;
    out[0] = violations; out[1] = bvh.node_count; out[2] = bvh.max_depth; out[3] = bvh.stack_bound; out[4] = leaves; out[5] = refs;
    return 0;
}

// The same for the 8-wide tree (bvh_build.h Bvh8Result): masks disjoint, empty slots inverted, scales positive powers of two,
// implicit child / triangle addresses in range, every triangle inside every ancestor's box and in exactly one leaf.
static int check_bvh8q(const std::vector<float> & verts, uint32_t n_tris, const Bvh8Result & bvh, uint64_t * out) {
    uint64_t violations = 0, leaves = 0, refs = 0;
    std::vector<uint8_t> seen(std::max(1u, n_tris), 0);
    std::vector<uint8_t> node_seen(std::max(1u, bvh.node_count), 0);
    struct Item { uint32_t node; float lo[3], hi[3]; };
    std::vector<Item> stack;
    Item root;
    root.node = 0;
    for (int a = 0; a < 3; ++a) { root.lo[a] = -3.0e38f; root.hi[a] = 3.0e38f; }
    stack.push_back(root);
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        if (it.node >= bvh.node_count) { violations++; continue; }
        if (node_seen[it.node]++) { violations++; continue; }
        const uint32_t * d = &bvh.nodes[(size_t)it.node * BVH8_NODE_DWORDS];
        float org[3], scale[3];
        const int scale_dword[3] = { 3, 6, 7 };
        for (int a = 0; a < 3; ++a) {
            memcpy(&org[a], &d[a], 4);
            const uint32_t sb = d[scale_dword[a]] & 0x7F800000u;
            memcpy(&scale[a], &sb, 4);
            if (sb == 0u || (d[scale_dword[a]] & 0x80000000u)) violations++;      // a positive power of two
        }
        const uint32_t imask = d[3] & 0xFFu, lmask = d[3] >> 8 & 0xFFu, c0 = d[6] & 0xFFu, c1 = d[6] >> 8 & 0xFFu;
        if (imask & lmask) violations++;
        if ((c0 | c1) & ~lmask) violations++;
        if (!(imask | lmask)) violations++;
        uint32_t next_child = d[4], next_tri = d[5];
        for (uint32_t sl = 0; sl < 8; ++sl) {
            uint32_t qlo[3], qhi[3];
            for (int a = 0; a < 3; ++a) {
                qlo[a] = d[8 + 2 * a + (sl >> 2)] >> (8 * (sl & 3u)) & 0xFFu;
                qhi[a] = d[14 + 2 * a + (sl >> 2)] >> (8 * (sl & 3u)) & 0xFFu;
            }
            if (!((imask | lmask) >> sl & 1u)) {
                for (int a = 0; a < 3; ++a) if (qlo[a] != 255u || qhi[a] != 0u) violations++;
                continue;
            }
            Item ch;
            for (int a = 0; a < 3; ++a) {
                ch.lo[a] = std::max(it.lo[a], org[a] + (float)qlo[a] * scale[a]);
                ch.hi[a] = std::min(it.hi[a], org[a] + (float)qhi[a] * scale[a]);
            }
            if (imask >> sl & 1u) {
                ch.node = next_child++;
                stack.push_back(ch);
            } else {
                const uint32_t cnt = 1u + (c0 >> sl & 1u) + 2u * (c1 >> sl & 1u);
                leaves++;
                for (uint32_t i = 0; i < cnt; ++i) {
                    const uint32_t slot = next_tri++;
                    if (slot == n_tris && n_tris == 0) continue;        // the dummy triangle of an empty scene
                    if (slot >= n_tris) { violations++; continue; }
                    refs++;
                    if (seen[slot]++) violations++;
                    const uint32_t t = bvh.tri_order[slot];
                    for (int c = 0; c < 3; ++c)
                        for (int a = 0; a < 3; ++a) {
                            const float v = verts[(size_t)t * 9 + 3 * c + a];
                            const float tol = 4.0f * 1.1920929e-7f * std::max(1.0f, fabsf(v));
                            if (v < ch.lo[a] - tol || v > ch.hi[a] + tol) violations++;
                        }
                }
            }
        }
    }
    for (uint32_t t = 0; t < n_tris; ++t) if (!seen[t]) violations++;
    for (uint32_t n = 0; n < bvh.node_count; ++n) if (!node_seen[n]) violations++;
    out[0] = violations; out[1] = bvh.node_count; out[2] = bvh.max_depth; out[3] = bvh.stack_bound; out[4] = leaves; out[5] = refs;
    return 0;
}

#if !defined(PRT_BVH8)
static int check_bvh_wide(const std::vector<float> & verts, uint32_t n_tris, const BvhWide & bvh, uint64_t * out) { return check_bvh4q(verts, n_tris, bvh, out); }
#else
static int check_bvh_wide(const std::vector<float> & verts, uint32_t n_tris, const BvhWide & bvh, uint64_t * out) { return check_bvh8q(verts, n_tris, bvh, out); }
#endif

int prt_debug_check_bvh(const prt_scene_desc * s, uint64_t * out) {
    PRT_API_TRY
    if (!s || !out || s->index_count % 3) return -1;
    const uint32_t n_tris = s->index_count / 3;
    std::vector<float> verts((size_t)n_tris * 9);
    for (uint32_t t = 0; t < n_tris; ++t)
        for (int c = 0; c < 3; ++c) memcpy(&verts[(size_t)t * 9 + 3 * c], s->positions + 3 * (size_t)s->idx_positions[3 * t + c], 12);
    PrtOptions opt;
    prt_options_from_env(opt);                  // the test-suite selects the collapse rule through the environment
#if defined(PRT_BVH8_OCTANT)
    opt.bvh.slot_order = 0;
#else
    opt.bvh.slot_order = 1;
#endif
    BvhWide bvh;
    PRT_BUILD_WIDE(verts.data(), n_tris, BVH_LEAF_MAX, 4, &bvh, 1.0f, &opt.bvh);
    if (validate_bvh_links(bvh, n_tris)) return -9;          // what prt_upload_scene checks before it uploads a tree
    return check_bvh_wide(verts, n_tris, bvh, out);
    PRT_API_CATCH_RC(nullptr, "prt_debug_check_bvh")
}

// The same check on the tree of the GPU LBVH builder (needs a context: the radix tree is built on its device).
int prt_debug_check_bvh_lbvh(prt_ctx * ctx, const prt_scene_desc * s, uint64_t * out) {
    PRT_API_TRY
    if (!ctx || !s || !out || s->index_count % 3) return -1;
    const uint32_t n_tris = s->index_count / 3;
    std::vector<float> verts((size_t)n_tris * 9);
    for (uint32_t t = 0; t < n_tris; ++t)
        for (int c = 0; c < 3; ++c) memcpy(&verts[(size_t)t * 9 + 3 * c], s->positions + 3 * (size_t)s->idx_positions[3 * t + c], 12);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    BvhWide bvh;
    int rc = build_bvh_lbvh(ctx, verts.data(), n_tris, BVH_LEAF_MAX, &bvh);
    if (rc) return rc;
    if (const char * bad = validate_bvh_links(bvh, n_tris)) { ctx->error = bad; return -9; }
    return check_bvh_wide(verts, n_tris, bvh, out);
    PRT_API_CATCH_RC(ctx, "prt_debug_check_bvh_lbvh")
}

}  // extern "C"

// =============================================================================================================
// Several devices behind one handle (prt_create(const int * device_ids, int n_dev)) - the product's
// multi-GPU path.  Replaces the reference's partition + MPI_Gather (main.cpp:311-347): the scene is replicated (as every
// MPI rank holds it), device g renders the row blocks b with b % n == g into a packed buffer in its own HBM, the shards
// travel to device 0 over the fabric as peer-to-peer copies - DMA engines over xGMI, no compute unit involved, so they
// do not compete with the persistent render kernels for wave slots the way a collective's kernels do -, one small kernel
// on device 0 puts the rows where they belong, and ONE copy takes the frame to the host.
//
// Round 4: FRAMES IN FLIGHT.  A persistent kernel leaves its GPU partly idle while its last rays drain (a 1/8 shard of the
// headline frame: 2.1 ms at one frame at a time, 1.75 ms per frame with two in flight, profiles/r03_shards_frames_in_flight.txt),
// and the copies / assembly / download of frame k need no compute unit that frame k + 1 could not use.  The handle
// therefore holds `depth` LANES (default 2, PRT_MULTI_DEPTH): a lane is one context per device - lane 0 the contexts
// prt_multi_context() returns, the others CLONES that share the uploaded scene's device arrays and own only their streams
// and workspaces - plus the lane's shard / staging / frame buffers.  prt_multi_submit() hands a frame to a free lane and
// returns; the lane's worker threads - one per device, created ONCE with the handle, not per frame - render and send
// their shards; prt_multi_wait() assembles and downloads.  prt_multi_render() is submit + wait.
__global__ void k_assemble_shards(const float4 * staging, float4 * frame, unsigned int width, unsigned int height,
                                  unsigned int block_rows, unsigned int n_dev, unsigned int max_shard_rows) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width * height) return;
    const unsigned int y = i / width, x = i - y * width;
    const unsigned int b = y / block_rows, g = b % n_dev;
    const unsigned int local_row = (b / n_dev) * block_rows + (y - b * block_rows);
    frame[i] = staging[((size_t)g * max_shard_rows + local_row) * width + x];
}

enum { PRT_MULTI_BLOCK_ROWS = 8, PRT_MULTI_MAX_DEPTH = 4 };

struct MultiLane {
    std::vector<prt_ctx *> ctx;              // per device; lane 0: the handle's own contexts, else clones of them
    std::vector<DevBuf<float4> > shard;      // per device: its packed shard (device g's memory)
    DevBuf<float4> staging, frame;           // device 0: every shard side by side; the assembled frame
    std::vector<hipEvent_t> done;            // per device: its peer copy has landed
    // the frame in flight on this lane (written by submit under the handle's mutex, read by the lane's workers)
    bool busy = false;
    uint64_t ticket = 0;
    prt_camera cam;
    prt_params params;
    uint32_t width = 0, height = 0;
    float * rgba_out = nullptr;
    unsigned int pending = 0;                // workers that have not finished this frame yet
    std::vector<uint64_t> job;               // per device: generation of the job handed to its worker
    std::vector<int> rc;
    std::vector<std::string> err;
    std::vector<prt_counters> ctr;
};

struct prt_multi {
    std::vector<int> devices;
    std::vector<MultiLane> lane;
    std::vector<std::thread> workers;        // lanes x devices, alive as long as the handle
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    bool quit = false;
    uint64_t next_ticket = 1;
    std::string error;
};

namespace {

// A second context on src's device that SHARES the uploaded scene's device arrays (it does not own them: src must
// outlive it and must not be re-uploaded while it exists) and has streams, events and workspaces of its own.
prt_ctx * clone_context(prt_ctx * src) {
    prt_ctx * c = prt_create(src->device);
    if (!c) return nullptr;
    c->opt = src->opt;
    c->nodes.borrow(src->nodes); c->tris.borrow(src->tris); c->shade.borrow(src->shade); c->diffuse_dirs.borrow(src->diffuse_dirs);
    c->tri_rank.borrow(src->tri_rank); c->materials.borrow(src->materials); c->lights.borrow(src->lights);
    c->textures.borrow(src->textures); c->texels.borrow(src->texels); c->srgb_lut.borrow(src->srgb_lut);
    c->tri_uv.borrow(src->tri_uv); c->tri_tan.borrow(src->tri_tan); c->ref_spheres.borrow(src->ref_spheres);
    c->scene = src->scene;
    c->scene.near_tie_unresolved = &c->counters.p->near_tie_unresolved;      // its own counters
    c->scene.spec_dirs = nullptr;
    c->spec_table_samples = 0;              // its own specular direction table, built at its first render
    c->material_ns = src->material_ns;
    c->info = src->info;
    c->scene_abs_max = src->scene_abs_max;
    c->any_translucent = src->any_translucent; c->point_lights = src->point_lights; c->textured = src->textured;
    c->stack_bound = src->stack_bound;
    c->pool_park_cap = src->pool_park_cap; c->pool_spark_cap = src->pool_spark_cap;
    c->has_scene = src->has_scene;
    c->scene_epoch = 1;
    return c;
}

void multi_worker(prt_multi * m, unsigned int f, unsigned int g) {
    MultiLane & L = m->lane[f];
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(m->mu);
            m->cv_work.wait(lk, [&] { return m->quit || L.job[g] != seen; });
            if (m->quit) return;
            seen = L.job[g];
        }
        int rc = 0;
        std::string err;
        prt_counters ctr;
        memset(&ctr, 0, sizeof(ctr));
        try {
            const unsigned int n = (unsigned int)L.ctx.size();
            prt_ctx * c = L.ctx[g], * c0 = L.ctx[0];
            const size_t rows = prt_shard_rows(L.height, PRT_MULTI_BLOCK_ROWS, g, n);
            const uint32_t max_rows = prt_shard_rows(L.height, PRT_MULTI_BLOCK_ROWS, 0, n);      // rank 0 owns the most rows
            hipError_t e = hipSetDevice(c->device);
            if (e == hipSuccess) e = L.shard[g].ensure(std::max<size_t>(1, rows * L.width));
            if (e != hipSuccess) { rc = PRT_ERR_HIP; err = std::string("shard buffer: ") + hipGetErrorString(e); }
            if (!rc) {
                rc = prt_render_shard_device(c, &L.cam, &L.params, L.width, L.height, PRT_MULTI_BLOCK_ROWS, g, n, L.shard[g].p, &ctr);
                if (rc) err = prt_last_error(c);
            }
            if (!rc && n > 1) {
                // the shard's trip to device 0, queued on the rendering context's own stream: a shard that is done travels
                // while the others still render
                if (rows) e = hipMemcpyPeerAsync(L.staging.p + (size_t)g * max_rows * L.width, c0->device, L.shard[g].p, c->device,
                                                 rows * L.width * sizeof(float4), c->stream);
                if (e == hipSuccess) e = hipEventRecord(L.done[g], c->stream);
                if (e != hipSuccess) { rc = PRT_ERR_HIP; err = std::string("peer copy: ") + hipGetErrorString(e); }
            }
        } catch (...) {
            rc = api_exception(&err, "prt_multi worker");
        }
        {
            std::lock_guard<std::mutex> lk(m->mu);
            L.rc[g] = rc;
            try { L.err[g] = err; } catch (...) {}
            L.ctr[g] = ctr;
            if (--L.pending == 0) m->cv_done.notify_all();
        }
    }
}

void multi_release_lane(MultiLane & L, bool destroy_ctx) {
    for (size_t g = 0; g < L.ctx.size(); ++g) {
        if (!L.ctx[g]) continue;
        (void)hipSetDevice(L.ctx[g]->device);
        if (L.ctx[g]->stream) (void)hipStreamSynchronize(L.ctx[g]->stream);
        if (g < L.shard.size()) L.shard[g].release();
        if (g < L.done.size() && L.done[g]) { (void)hipEventDestroy(L.done[g]); L.done[g] = nullptr; }
    }
    if (!L.ctx.empty() && L.ctx[0]) {
        (void)hipSetDevice(L.ctx[0]->device);
        L.staging.release();
        L.frame.release();
    }
    if (destroy_ctx)
        for (size_t g = 0; g < L.ctx.size(); ++g) { prt_destroy(L.ctx[g]); L.ctx[g] = nullptr; }
}

// Every stream of every lane drained: after a failure nothing of the failed frame is still writing into staging / frame.
void multi_quiesce(prt_multi * m) {
    for (MultiLane & L : m->lane)
        for (prt_ctx * c : L.ctx)
            if (c && c->stream) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
}

}  // namespace

extern "C" {

prt_multi * prt_multi_create(const int * device_ids, int n_dev) {
    PRT_API_TRY
    if (!device_ids || n_dev < 1) { g_create_error = "prt_multi_create: no devices"; return nullptr; }
    prt_multi * m = new prt_multi;
    m->devices.assign(device_ids, device_ids + n_dev);
    long long depth = 2;
    if (const char * v = getenv("PRT_MULTI_DEPTH")) depth = atoll(v);        // creation only, like every PRT_* variable
    depth = std::max(1ll, std::min((long long)PRT_MULTI_MAX_DEPTH, depth));
    m->lane.resize((size_t)depth);
    for (MultiLane & L : m->lane) {
        L.ctx.assign((size_t)n_dev, nullptr);
        L.shard.resize((size_t)n_dev);
        L.done.assign((size_t)n_dev, nullptr);
        L.job.assign((size_t)n_dev, 0);
        L.rc.assign((size_t)n_dev, 0);
        L.err.resize((size_t)n_dev);
        L.ctr.resize((size_t)n_dev);
    }
    for (int g = 0; g < n_dev; ++g) {
        prt_ctx * c = prt_create(device_ids[g]);
        if (!c) { prt_multi_destroy(m); return nullptr; }
        m->lane[0].ctx[(size_t)g] = c;
    }
    for (MultiLane & L : m->lane)
        for (int g = 0; g < n_dev; ++g) {
            prt_ctx * c0 = m->lane[0].ctx[0], * c = m->lane[0].ctx[(size_t)g];
            if (hipSetDevice(c->device) != hipSuccess || hipEventCreateWithFlags(&L.done[(size_t)g], hipEventDisableTiming) != hipSuccess) {
                g_create_error = "prt_multi_create: event creation failed";
                prt_multi_destroy(m);
                return nullptr;
            }
            // direct loads / stores between the devices where the fabric offers them (the copies work either way)
            if (&L == &m->lane[0] && g > 0 && c->device != c0->device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, c0->device, c->device) == hipSuccess && can) {
                    (void)hipSetDevice(c0->device);
                    (void)hipDeviceEnablePeerAccess(c->device, 0);
                    (void)hipGetLastError();             // "already enabled" is fine
                }
            }
        }
    // the workers: one per (lane, device), for the life of the handle (round 3 created and joined n threads per frame)
    for (unsigned int f = 0; f < (unsigned int)depth; ++f)
        for (unsigned int g = 0; g < (unsigned int)n_dev; ++g) m->workers.emplace_back(multi_worker, m, f, g);
    return m;
    PRT_API_CATCH_PTR("prt_multi_create")
}

void prt_multi_destroy(prt_multi * m) {
    PRT_API_TRY
    if (!m) return;
    {
        std::unique_lock<std::mutex> lk(m->mu);
        // frames still in flight finish first (their buffers and contexts go away below)
        m->cv_done.wait(lk, [&] { for (MultiLane & L : m->lane) if (L.pending) return false; return true; });
        m->quit = true;
    }
    m->cv_work.notify_all();
    for (std::thread & t : m->workers) if (t.joinable()) t.join();
    for (size_t f = m->lane.size(); f-- > 0;) multi_release_lane(m->lane[f], true);      // the clones before the contexts they borrow from
    delete m;
    PRT_API_CATCH_VOID
}

const char * prt_multi_last_error(const prt_multi * m) { return m ? m->error.c_str() : g_create_error.c_str(); }
int prt_multi_device_count(const prt_multi * m) { return m ? (int)m->devices.size() : 0; }
int prt_multi_depth(const prt_multi * m) { return m ? (int)m->lane.size() : 0; }
prt_ctx * prt_multi_context(prt_multi * m, int i) { return m && i >= 0 && (size_t)i < m->devices.size() ? m->lane[0].ctx[(size_t)i] : nullptr; }

int prt_multi_upload_scene(prt_multi * m, const prt_scene_desc * scene) {
    PRT_API_TRY
    if (!m) return -1;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        for (MultiLane & L : m->lane)
            if (L.busy) { m->error = "prt_multi_upload_scene: a frame is still in flight (prt_multi_wait it first)"; return PRT_ERR_IN_FLIGHT; }
    }
    // the other lanes' contexts borrow lane 0's scene arrays: they go before those are replaced, and come back after
    for (size_t f = 1; f < m->lane.size(); ++f)
        for (size_t g = 0; g < m->lane[f].ctx.size(); ++g) { prt_destroy(m->lane[f].ctx[g]); m->lane[f].ctx[g] = nullptr; }
    for (size_t g = 0; g < m->devices.size(); ++g) {
        const int rc = prt_upload_scene(m->lane[0].ctx[g], scene);
        if (rc) { m->error = prt_last_error(m->lane[0].ctx[g]); return rc; }
    }
    for (size_t f = 1; f < m->lane.size(); ++f)
        for (size_t g = 0; g < m->devices.size(); ++g) {
            m->lane[f].ctx[g] = clone_context(m->lane[0].ctx[g]);
            if (!m->lane[f].ctx[g]) { m->error = std::string("prt_multi_upload_scene: ") + g_create_error; return PRT_ERR_HIP; }
        }
    return 0;
    PRT_API_CATCH_RC_MULTI(m, "prt_multi_upload_scene")
}

#define MULTI_TRY(m, call)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) { (m)->error = std::string(#call) + ": " + hipGetErrorString(e_); multi_quiesce(m); return PRT_ERR_HIP; } \
    } while (0)

int prt_multi_submit(prt_multi * m, const prt_camera * cam, const prt_params * params, uint32_t width, uint32_t height,
                     float * rgba_out, uint64_t * ticket) {
    PRT_API_TRY
    if (!m || m->devices.empty()) return -1;
    if (!cam || !params || !ticket) { m->error = "prt_multi_submit: null camera, params or ticket"; return -1; }
    if (!rgba_out && width && height) { m->error = "prt_multi_submit: null output"; return -1; }
    const unsigned int n = (unsigned int)m->devices.size();
    MultiLane * L = nullptr;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        for (MultiLane & c : m->lane)
            if (!c.busy && c.ctx[0]) { L = &c; break; }
        if (!L) { m->error = "prt_multi_submit: every lane has a frame in flight (prt_multi_wait the oldest ticket first)"; return PRT_ERR_IN_FLIGHT; }
        L->busy = true;                      // reserved; the workers start below, when the buffers are there
    }
    const uint32_t max_rows = prt_shard_rows(height, PRT_MULTI_BLOCK_ROWS, 0, n);
    prt_ctx * c0 = L->ctx[0];
    hipError_t e = hipSetDevice(c0->device);
    if (e == hipSuccess && n > 1) e = L->staging.ensure((size_t)n * max_rows * width);
    if (e == hipSuccess && n > 1) e = L->frame.ensure((size_t)width * height);
    if (e != hipSuccess) {
        std::lock_guard<std::mutex> lk(m->mu);
        L->busy = false;
        m->error = std::string("prt_multi_submit: buffers on device 0: ") + hipGetErrorString(e);
        return PRT_ERR_HIP;
    }
    {
        std::lock_guard<std::mutex> lk(m->mu);
        L->cam = *cam; L->params = *params; L->width = width; L->height = height; L->rgba_out = rgba_out;
        L->ticket = m->next_ticket++;
        L->pending = n;
        for (unsigned int g = 0; g < n; ++g) { L->rc[g] = 0; L->err[g].clear(); L->job[g]++; }
        *ticket = L->ticket;
    }
    m->cv_work.notify_all();
    return 0;
    PRT_API_CATCH_RC_MULTI(m, "prt_multi_submit")
}

int prt_multi_wait(prt_multi * m, uint64_t ticket, prt_counters * counters) {
    PRT_API_TRY
    if (!m || m->devices.empty()) return -1;
    const unsigned int n = (unsigned int)m->devices.size();
    MultiLane * L = nullptr;
    {
        std::unique_lock<std::mutex> lk(m->mu);
        for (MultiLane & c : m->lane)
            if (c.busy && c.ticket == ticket) { L = &c; break; }
        if (!L) { m->error = "prt_multi_wait: no frame with this ticket is in flight"; return -1; }
        m->cv_done.wait(lk, [&] { return L->pending == 0; });
    }
    // from here on the lane's workers are idle: its fields are this thread's
    struct Release { prt_multi * m; MultiLane * L; ~Release() { std::lock_guard<std::mutex> lk(m->mu); L->busy = false; } } release = { m, L };
    for (unsigned int g = 0; g < n; ++g)
        if (L->rc[g]) { m->error = L->err[g]; multi_quiesce(m); return L->rc[g]; }
    prt_ctx * c0 = L->ctx[0];
    const size_t n_px = (size_t)L->width * L->height;
    MULTI_TRY(m, hipSetDevice(c0->device));
    if (n == 1) {
        // one device: its "shard" is the frame (the render call returned with its stream drained)
        if (n_px) MULTI_TRY(m, hipMemcpy(L->rgba_out, L->shard[0].p, n_px * sizeof(float4), hipMemcpyDeviceToHost));
    } else {
        const uint32_t max_rows = prt_shard_rows(L->height, PRT_MULTI_BLOCK_ROWS, 0, n);
        for (unsigned int g = 0; g < n; ++g) MULTI_TRY(m, hipStreamWaitEvent(c0->stream, L->done[g], 0));
        if (n_px) {
            hipLaunchKernelGGL(k_assemble_shards, dim3((unsigned int)((n_px + 255) / 256)), dim3(256), 0, c0->stream, L->staging.p, L->frame.p,
                               L->width, L->height, (unsigned int)PRT_MULTI_BLOCK_ROWS, n, max_rows);
            MULTI_TRY(m, hipGetLastError());
            MULTI_TRY(m, hipMemcpyAsync(L->rgba_out, L->frame.p, n_px * sizeof(float4), hipMemcpyDeviceToHost, c0->stream));
        }
        MULTI_TRY(m, hipStreamSynchronize(c0->stream));
    }
    if (counters) {
        prt_counters sum;
        memset(&sum, 0, sizeof(sum));
        for (unsigned int g = 0; g < n; ++g) {
            const prt_counters & c = L->ctr[g];
            sum.ray_count += c.ray_count; sum.node_visits += c.node_visits; sum.tri_tests += c.tri_tests;
            sum.shaded_hits += c.shaded_hits; sum.trace_kernel_launches += c.trace_kernel_launches;
            sum.render_ms = std::max(sum.render_ms, c.render_ms);                   // the devices run concurrently
            sum.trace_kernel_ms = std::max(sum.trace_kernel_ms, c.trace_kernel_ms);
            sum.pipeline = c.pipeline;
        }
        *counters = sum;
    }
    return 0;
    PRT_API_CATCH_RC_MULTI(m, "prt_multi_wait")
}

int prt_multi_render(prt_multi * m, const prt_camera * cam, const prt_params * params, uint32_t width, uint32_t height,
                     float * rgba_out, prt_counters * counters) {
    PRT_API_TRY
    uint64_t ticket = 0;
    const int rc = prt_multi_submit(m, cam, params, width, height, rgba_out, &ticket);
    if (rc) return rc;
    return prt_multi_wait(m, ticket, counters);
    PRT_API_CATCH_RC_MULTI(m, "prt_multi_render")
}

// Test hook of the exception guard (tests/test_host_side.py): throws the given kind of C++ exception inside a guarded
// entry point and returns what the caller of any entry point would get.  ctx may be NULL (no GPU needed).
int prt_debug_throw(prt_ctx * ctx, int kind) {
    PRT_API_TRY
    if (kind == 1) throw std::bad_alloc();
    if (kind == 2) { std::vector<float> v; v.resize(v.max_size() + 1); return (int)v.size(); }     // the real thing: std::length_error
    if (kind == 3) throw std::runtime_error("thrown on request");
    if (kind == 4) throw 42;
    return 0;
    PRT_API_CATCH_RC(ctx, "prt_debug_throw")
}

}  // extern "C"
