// main.cpp - driver with the reference's call sequence (main.cpp:537-612) on top of the HIP path.
//
//   InitParams -> MakeCamera -> ParseOBJ -> CalculateTangents -> BuildHierarchy -> InitScene ->
//   object list -> Render -> WriteFramebufferImage
//
// Same flags as the reference (prefix matching included) plus --spp N, --max_spp M (adaptive mode), --seed S, --obj FILE, --gpus N,
// --pipeline P, which replace what the reference hard-codes (adaptive 10..50 spp, rank-indexed seed
// table, "sponza.obj", MPI rank count).  MPI is gone: one process drives all GPUs of the node.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <time.h>

#include "prt_scene.h"

extern u32 gRenderGpuCount;

static double Now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char ** argv) {
    InitParams(argc, argv);
    for (int i = 1; i + 1 < argc; ++i)
        if (!strncmp("--gpus", argv[i], 6)) gRenderGpuCount = (u32)atoi(argv[i + 1]);
    if (gRenderGpuCount < 1) gRenderGpuCount = 1;

    Camera cam = MakeCamera(gParams.camera_fov, gParams.image_width, gParams.image_height);
    Matrix33 transform;
    transform.SetIdentity();
    double t0 = Now();
    Mesh * mesh = ParseOBJ(gParams.data_dirname, gParams.obj_filename, transform);
    if (!mesh || mesh->groups.empty()) {
        fprintf(stderr, "Cannot load %s from %s (the reference dereferences NULL here, main.cpp:557)\n",
                gParams.obj_filename, gParams.data_dirname);
        return 1;
    }
    CalculateTangents(mesh);
    double t1 = Now();
    BoundingHierarchy hierarchy;
    BuildHierarchy(&hierarchy, mesh);
    double t2 = Now();
    fprintf(stderr, "[Load Mesh] :: %.2f s\n[Build Hierarchy] :: %.2f s\n", t1 - t0, t2 - t1);

    u32 total_tris = 0;
    Scene scene = InitScene();
    PopulateSceneObjects(&scene, &hierarchy, mesh, &total_tris);
    printf("Triangles: %u\n", total_tris);

    double t3 = Now();
    Framebuffer fb = Render(&cam, &scene, gParams.image_width, gParams.image_height);
    double t4 = Now();
    fprintf(stderr, "[Render, sync] :: %.2f s (first call: includes upload + BVH build)\n", t4 - t3);

    const RenderReport & r = gLastRenderReport;
    if (r.status != 0) {
        // no image of a failed render: tone-mapping a zero frame would print one warning per pixel and write a black PNG
        fprintf(stderr, "render failed (%d): no image written\n", r.status);
        return 2;
    }
    u32 total_pixel_count = gParams.image_width * gParams.image_height;
    printf("GPUs %u\n", r.gpu_count);
    printf("Rays cast:          %llu\n", (unsigned long long)r.counters.ray_count);
    printf("Render (device):    %.3f ms  (%.2f Mrays/s)\n", r.render_ms,
           r.render_ms > 0.0 ? (double)r.counters.ray_count / r.render_ms / 1e3 : 0.0);
    if (r.counters.sphere_check_count) {
        printf("BVH nodes visited:  %llu\n", (unsigned long long)r.counters.sphere_check_count);
        printf("   average / pixel: %f\n", (double)r.counters.sphere_check_count / total_pixel_count);
        printf("Triangles tested:   %llu\n", (unsigned long long)r.counters.mesh_check_count);
        printf("   average / pixel: %f\n", (double)r.counters.mesh_check_count / total_pixel_count);
    }

    WriteFramebufferImage(&fb, gParams.image_output_filename);
    fprintf(stderr, "done, exiting...\n");
    return 0;
}
