// render_host.cpp - Render(): the reference's distribution layer (main.cpp:301-358) over the HIP path.
//
// Reference: every MPI rank loads the full scene, renders the contiguous pixel range
// [rank*cpp, (rank+1)*cpp) on one CPU core and MPI_Gathers float RGBA to rank 0 (main.cpp:311-347).  Here one
// process drives n GPUs through ONE handle (include/prt.h prt_multi_*): the flattened scene is uploaded to each
// (replicated, as every rank does today), GPU r renders the row blocks b with b % n == r (interleaved, because
// contiguous ranges balance badly - NOTES.txt:25) into its own HBM, the shards travel to GPU 0 as peer-to-peer
// copies over the fabric, a kernel there puts the rows in place and one copy brings the frame to the host.
// (bench.py's multi-process path does the same sharding with one rank per GPU and an RCCL gather.)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/prt_host.h"
#include "prt_scene.h"
#include "host_guard.h"
#include "scene_flatten.h"

RenderReport gLastRenderReport;
u32 gRenderGpuCount = 1;

namespace {

// The uploaded contexts are kept between calls, keyed by a scene id that is never reused (NOT by the scene's address: a
// freed scene's address is likely to be handed out again for the next one).  0 = no scene.
struct ContextCache {
    u64 key = 0;
    prt_multi * multi = NULL;
    void Clear() {
        if (multi) prt_multi_destroy(multi);
        multi = NULL;
        key = 0;
    }
    ~ContextCache() { Clear(); }
};
ContextCache gCache;
std::string gRenderError;
std::mutex gRenderMutex;                 // guards gCache and gRenderError: concurrent callers render one after the other
std::atomic<u64> gNextSceneId(1);

int RenderFlat(u64 cache_key, const prt_scene_desc * desc, const prt_camera * cam, const prt_params * params,
               u32 width, u32 height, int n_gpus, float * rgba_out, prt_counters * total) {
    std::lock_guard<std::mutex> lock(gRenderMutex);
    if (n_gpus < 1) n_gpus = 1;
    if (cache_key == 0 || gCache.key != cache_key || prt_multi_device_count(gCache.multi) != n_gpus) {
        gCache.Clear();
        // PRT_HOST_SHARE_DEVICE=1: every "GPU" of the call is device 0 (rehearsal of the n_gpus > 1 path on a one-GPU box)
        const char * share = getenv("PRT_HOST_SHARE_DEVICE");
        std::vector<int> ids((size_t)n_gpus);
        for (int g = 0; g < n_gpus; ++g) ids[(size_t)g] = share && atoi(share) ? 0 : g;
        gCache.multi = prt_multi_create(ids.data(), n_gpus);
        if (!gCache.multi) { gRenderError = prt_multi_last_error(NULL); return -1; }
        if (prt_multi_upload_scene(gCache.multi, desc) != 0) { gRenderError = prt_multi_last_error(gCache.multi); gCache.Clear(); return -2; }
        gCache.key = cache_key;
    }
    prt_counters sum;
    memset(&sum, 0, sizeof(sum));
    const int rc = prt_multi_render(gCache.multi, cam, params, width, height, rgba_out, &sum);
    if (rc) { gRenderError = prt_multi_last_error(gCache.multi); return rc; }
    if (total) *total = sum;
    return 0;
}

}  // namespace

u64 NewRenderSceneId() { return gNextSceneId.fetch_add(1); }

void ForgetRenderScene(u64 scene_id) {
    std::lock_guard<std::mutex> lock(gRenderMutex);
    if (scene_id != 0 && gCache.key == scene_id) gCache.Clear();
}

Framebuffer Render(Camera * cam, Scene * scene, u32 width, u32 height) {
    Framebuffer result;
    result.width = width;
    result.height = height;
    result.pixels = (Vector4 *)calloc(sizeof(Vector4), (size_t)width * height);

    // Like the reference's Render, which is called once per process, every call takes the scene as it is NOW: flatten and
    // upload again under a fresh id (callers that render many frames of one scene use prt_host_render, which caches).
    FlatScene flat;
    FlattenScene(scene, &flat);
    prt_camera pc = ToPrtCamera(cam);
    prt_params pp = ToPrtParams(&gParams);
    prt_counters ctr;
    memset(&ctr, 0, sizeof(ctr));
    int rc = RenderFlat(NewRenderSceneId(), &flat.desc, &pc, &pp, width, height, (int)gRenderGpuCount, (float *)result.pixels, &ctr);
    if (rc != 0) {
        // The reference has no error path here (asserts print and continue, brt.h:41); we report, return the zero-filled
        // frame rather than abort, and leave the code in gLastRenderReport.status for the caller (prt_main exits non-zero).
        std::lock_guard<std::mutex> lock(gRenderMutex);
        fprintf(stderr, "Render: HIP path failed (%d): %s\n", rc, gRenderError.c_str());
    }
    gLastRenderReport.status = rc;
    gLastRenderReport.counters.ray_count = ctr.ray_count;
    gLastRenderReport.counters.sphere_check_count = ctr.node_visits;
    gLastRenderReport.counters.mesh_check_count = ctr.tri_tests;
    gLastRenderReport.shaded_hits = ctr.shaded_hits;
    gLastRenderReport.render_ms = ctr.render_ms;
    gLastRenderReport.trace_kernel_ms = ctr.trace_kernel_ms;
    gLastRenderReport.gpu_count = gRenderGpuCount;
    return result;
}

// from host_capi.cpp's opaque scene
const prt_scene_desc * prt_host_scene_desc(const prt_host_scene * hs);
extern "C" u64 prt_host_scene_id(const prt_host_scene * hs);

extern "C" int prt_host_render(const prt_host_scene * scene, const prt_camera * cam, const prt_params * params, uint32_t width,
                               uint32_t height, int n_gpus, float * rgba_out, prt_counters * counters) {
    if (!scene || !cam || !params || !rgba_out) return -1;
    std::string err;
    try {
        return RenderFlat(prt_host_scene_id(scene), prt_host_scene_desc(scene), cam, params, width, height, n_gpus, rgba_out, counters);
    } catch (...) {
        // nothing leaves a C entry point as an exception (host_guard.h); RenderFlat's lock is released by now
        HostApiException(&err, "prt_host_render");
    }
    try { std::lock_guard<std::mutex> lock(gRenderMutex); gRenderError = err; } catch (...) {}
    return -12;
}

extern "C" const char * prt_host_render_error(void) {
    // a copy per calling thread: the shared string may be rewritten by another thread's render
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lock(gRenderMutex);
    copy = gRenderError;
    return copy.c_str();
}
