// host_capi.cpp - C entry points of libprt_host.so (include/prt_host.h) over the C++ host mirror.
#include <cstdio>
#include <cstring>
#include <string>
#include <time.h>

#include "../../include/prt_host.h"
#include "prt_scene.h"
#include "scene_flatten.h"
#include "host_guard.h"
#include <stdexcept>
#include <vector>

float TonemapFramebuffer(const Framebuffer * fb, u8 * rgba8);   // image_out.cpp
u64 NewRenderSceneId();                                          // render_host.cpp
void ForgetRenderScene(u64 scene_id);

struct prt_host_scene {
    Mesh * mesh;
    BoundingHierarchy hierarchy;
    Scene scene;
    FlatScene flat;
    u64 id;                         // key of Render's context cache; never reused, unlike this object's address
    double parse_seconds;
    double hierarchy_seconds;
};

static std::string gHostError;

static double Now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

extern "C" {

const char * prt_host_last_error(void) { return gHostError.c_str(); }

prt_host_scene * prt_host_load_obj(const char * dir, const char * obj_name, int light_mode, const float camera_position[3]) {
    HOST_API_TRY
    Matrix33 identity;
    identity.SetIdentity();
    double t0 = Now();
    Mesh * mesh = ParseOBJ(dir, obj_name, identity);
    if (!mesh) {
        gHostError = std::string("cannot read OBJ file ") + (dir ? dir : "") + "/" + obj_name;
        return NULL;
    }
    if (mesh->groups.empty()) {
        gHostError = "OBJ file has no groups (a `g` line must precede the faces)";
        delete mesh;
        return NULL;
    }
    double t1 = Now();
    CalculateTangents(mesh);
    prt_host_scene * hs = new prt_host_scene;
    hs->id = NewRenderSceneId();
    hs->mesh = mesh;
    hs->parse_seconds = t1 - t0;
    double t2 = Now();
    BuildHierarchy(&hs->hierarchy, mesh);
    hs->hierarchy_seconds = Now() - t2;
    hs->scene = InitScene();
    if (light_mode == 1) {
        hs->scene.light_count = 2;
    } else if (light_mode == 2) {
        hs->scene.light_count = 2;
        LightSource * l = &hs->scene.lights[1];
        l->type = Light_Point;
        l->color = Vector4(1.0f, 0.85f, 0.6f, 1.0f) * 6.0f;
        Vector3 cp = camera_position ? Vector3(camera_position[0], camera_position[1], camera_position[2]) : Vector3();
        l->position = cp + Vector3(0.5f, 1.0f, -2.0f);
        l->falloff = 3.0f;
    }
    PopulateSceneObjects(&hs->scene, &hs->hierarchy, mesh, NULL);
    FlattenScene(&hs->scene, &hs->flat);
    return hs;
    HOST_API_CATCH(&gHostError, "prt_host_load_obj", NULL)
}

void prt_host_free_scene(prt_host_scene * hs) {
    HOST_API_TRY
    if (!hs) return;
    ForgetRenderScene(hs->id);      // drop the device copies: the next scene may well be allocated at this address
    delete hs->mesh;
    delete hs;
    HOST_API_CATCH_VOID(&gHostError, "prt_host_free_scene")
}

const prt_scene_desc * prt_host_scene_desc(const prt_host_scene * hs) { return hs ? &hs->flat.desc : NULL; }
u64 prt_host_scene_id(const prt_host_scene * hs) { return hs ? hs->id : 0; }
double prt_host_scene_hierarchy_seconds(const prt_host_scene * hs) { return hs ? hs->hierarchy_seconds : 0.0; }
double prt_host_scene_parse_seconds(const prt_host_scene * hs) { return hs ? hs->parse_seconds : 0.0; }

void prt_host_make_camera(float fov, uint32_t width, uint32_t height, const float position[3], const float facing[3], prt_camera * out) {
    HOST_API_TRY
    Vector3 saved_p = gParams.camera_position, saved_f = gParams.camera_facing;
    gParams.camera_position = Vector3(position[0], position[1], position[2]);
    gParams.camera_facing = Vector3(facing[0], facing[1], facing[2]);
    Camera cam = MakeCamera(fov, width, height);
    gParams.camera_position = saved_p;
    gParams.camera_facing = saved_f;
    *out = ToPrtCamera(&cam);
    HOST_API_CATCH_VOID(&gHostError, "prt_host_make_camera")
}

void prt_host_default_params(uint32_t spp, uint64_t seed, prt_params * out) {
    HOST_API_TRY
    GlobalParams saved = gParams;
    char * argv0[1] = { (char *)"prt" };
    InitParams(1, argv0);
    gParams.spp = spp;
    gParams.seed = seed;
    *out = ToPrtParams(&gParams);
    gParams = saved;
    HOST_API_CATCH_VOID(&gHostError, "prt_host_default_params")
}

uint8_t * prt_host_load_texture(const char * filename, uint32_t * size_x, uint32_t * size_y, uint32_t * channels) {
    HOST_API_TRY
    Texture * t = filename ? LoadTexture(filename) : NULL;
    if (!t) {
        gHostError = std::string("prt_host_load_texture: ") + (filename ? TextureLoadError() : "null file name");
        return NULL;
    }
    if (size_x) *size_x = t->size_x;
    if (size_y) *size_y = t->size_y;
    if (channels) *channels = t->channels;
    uint8_t * texels = t->texels;
    free(t);
    return texels;
    HOST_API_CATCH(&gHostError, "prt_host_load_texture", NULL)
}

void prt_host_free_texture(uint8_t * texels) { free(texels); }

float prt_host_tonemap(const float * rgba, uint32_t width, uint32_t height, uint8_t * rgba8_out) {
    HOST_API_TRY
    Framebuffer fb;
    fb.pixels = (Vector4 *)rgba;
    fb.width = width;
    fb.height = height;
    return TonemapFramebuffer(&fb, rgba8_out);
    HOST_API_CATCH(&gHostError, "prt_host_tonemap", 0.0f)
}

int prt_host_write_image(const float * rgba, uint32_t width, uint32_t height, const char * filename) {
    HOST_API_TRY
    Framebuffer fb;
    fb.pixels = (Vector4 *)rgba;
    fb.width = width;
    fb.height = height;
    WriteFramebufferImage(&fb, filename);
    return 0;
    HOST_API_CATCH(&gHostError, "prt_host_write_image", -12)
}

// Test hook of the guard above (tests/test_host_side.py): throws inside a guarded entry point; returns -12 with the message in
// prt_host_last_error, as any entry point would.  kind: 1 std::bad_alloc, 2 std::length_error (a vector asked for more than
// max_size), 3 std::runtime_error, 4 an int; 0 returns 0.
int prt_host_debug_throw(int kind) {
    HOST_API_TRY
    if (kind == 1) throw std::bad_alloc();
    if (kind == 2) { std::vector<float> v; v.resize(v.max_size() + 1); return (int)v.size(); }
    if (kind == 3) throw std::runtime_error("thrown on request");
    if (kind == 4) throw 42;
    return 0;
    HOST_API_CATCH(&gHostError, "prt_host_debug_throw", -12)
}

}  // extern "C"
