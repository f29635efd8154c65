// host_scene.cpp - host-side setup mirroring the reference driver: flag parsing, camera, default
// material, lights, scene object list, and the flattening of that pointer graph for the device.
//
// Reference: main.cpp:133-177 (Camera), 360-504 (InitParams), 506-535 (MakeMaterial, InitScene),
// 576-599 (object list).  All arithmetic that feeds the hot path (camera basis, tanf, light facing)
// is done here on the host with the same expressions, so the device receives the same bits the
// reference's CPU path would use.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

#include "prt_scene.h"
#include "scene_flatten.h"

GlobalParams gParams;

namespace {

// The reference matches flags by PREFIX (main.cpp:441): "-bg" is shadowed by "-b", "-width" matches "-w".
bool FlagIs(const char * flag, const char * arg) { return strncmp(flag, arg, strlen(flag)) == 0; }

void Die() {
    fprintf(stderr, "Incorrect arguments.\n");          // main.cpp:360-364
    exit(1);
}

float ArgFloat(int argc, char ** argv, int i) { if (i + 1 >= argc) Die(); return (float)atof(argv[i + 1]); }
u32 ArgU32(int argc, char ** argv, int i) { if (i + 1 >= argc) Die(); return (u32)atoi(argv[i + 1]); }

void ReplaceString(char ** slot, const char * value) {
    if (*slot) free(*slot);
    *slot = strdup(value);
}

}  // namespace

void InitParams(int argc, char ** argv) {
    // Defaults: main.cpp:419-434.  ray_bias is the double literal 1e-3 rounded to float.
    gParams.ray_bias = (float)1e-3;
    gParams.reflection_samples = 1;
    gParams.spec_samples = 1;
    gParams.bounce_depth = 2;
    gParams.background_color = Vector4(0.8275f, 0.8913f, 1.0f, 1.0f) * 1.5f;
    gParams.camera_fov = 60.0f;
    gParams.camera_position = Vector3(475.0f, 250.0f, 0.0f);
    gParams.camera_facing = Normalize(Vector3(1.25f, -0.5f, 1.25f));
    gParams.image_output_filename = strdup("rt_out.png");
    gParams.data_dirname = strdup("/scratch/taylorbr/data/crytek-sponza/");
    gParams.image_width = 720;
    gParams.image_height = 480;
    // The reference hard-codes these (adaptive 10..50 spp main.cpp:308-309, seed table main.cpp:9-67,
    // "sponza.obj" main.cpp:553).  BASELINE's "N spp" is min_samples = max_samples = N.
    gParams.spp = 8;
    gParams.max_spp = 0;
    gParams.seed = 1234;
    gParams.obj_filename = strdup("sponza.obj");
    gParams.pipeline = 0;

    for (int i = 1; i < argc; ++i) {
        const char * arg = argv[i];
        if (FlagIs("--spp", arg)) { gParams.spp = ArgU32(argc, argv, i); ++i; }
        else if (FlagIs("--max_spp", arg)) { gParams.max_spp = ArgU32(argc, argv, i); ++i; }
        else if (FlagIs("--seed", arg)) { if (i + 1 >= argc) Die(); gParams.seed = strtoull(argv[i + 1], NULL, 0); ++i; }
        else if (FlagIs("--obj", arg)) { if (i + 1 >= argc) Die(); ReplaceString(&gParams.obj_filename, argv[i + 1]); ++i; }
        else if (FlagIs("--pipeline", arg)) { gParams.pipeline = ArgU32(argc, argv, i); ++i; }
        else if (FlagIs("--gpus", arg)) { ++i; }                       // consumed by the driver
        else if (FlagIs("--ray-bias", arg)) { gParams.ray_bias = ArgFloat(argc, argv, i); ++i; }
        else if (FlagIs("--width", arg) || FlagIs("-w", arg)) { gParams.image_width = ArgU32(argc, argv, i); ++i; }
        else if (FlagIs("--height", arg) || FlagIs("-h", arg)) { gParams.image_height = ArgU32(argc, argv, i); ++i; }
        else if (FlagIs("--reflection_samples", arg) || FlagIs("-rs", arg)) { gParams.reflection_samples = ArgU32(argc, argv, i); ++i; }
        else if (FlagIs("--specular_samples", arg) || FlagIs("-ss", arg)) { gParams.spec_samples = ArgU32(argc, argv, i); ++i; }
        else if (FlagIs("--bounce_depth", arg) || FlagIs("-b", arg)) { gParams.bounce_depth = ArgU32(argc, argv, i); ++i; }
        else if (FlagIs("--background_color", arg) || FlagIs("-bg", arg)) {
            gParams.background_color = Vector4(ArgFloat(argc, argv, i), ArgFloat(argc, argv, i + 1),
                                               ArgFloat(argc, argv, i + 2), ArgFloat(argc, argv, i + 3));
            i += 4;
        }
        else if (FlagIs("--fov", arg)) { gParams.camera_fov = ArgFloat(argc, argv, i); ++i; }
        else if (FlagIs("--camera_position", arg)) {
            gParams.camera_position = Vector3(ArgFloat(argc, argv, i), ArgFloat(argc, argv, i + 1), ArgFloat(argc, argv, i + 2));
            i += 3;
        }
        else if (FlagIs("--camera_facing", arg)) {
            gParams.camera_facing = Vector3(ArgFloat(argc, argv, i), ArgFloat(argc, argv, i + 1), ArgFloat(argc, argv, i + 2));
            i += 3;
        }
        else if (FlagIs("--output", arg) || FlagIs("-o", arg)) { if (i + 1 >= argc) Die(); ReplaceString(&gParams.image_output_filename, argv[i + 1]); ++i; }
        else if (FlagIs("--data", arg) || FlagIs("-d", arg)) { if (i + 1 >= argc) Die(); ReplaceString(&gParams.data_dirname, argv[i + 1]); ++i; }
    }
}

Camera MakeCamera(float fov, u32 width, u32 height) {                 // main.cpp:145-162
    Camera cam;
    cam.tan_a2 = tanf(DEG2RAD(fov / 2.0f));
    cam.aspect = (float)width / (float)height;
    cam.inv_width = 1.0f / (float)width;
    cam.inv_height = 1.0f / (float)height;
    cam.camera_position = gParams.camera_position;
    Vector3 world_up(0.0f, 1.0f, 0.0f);
    cam.camera_forward = Normalize(gParams.camera_facing);
    cam.camera_right = Normalize(Cross(cam.camera_forward, world_up));
    cam.camera_up = Normalize(Cross(cam.camera_right, cam.camera_forward));
    return cam;
}

Material * MakeMaterial(Vector4 color) {                              // main.cpp:506-517
    Material * m = (Material *)calloc(1, sizeof(Material));
    m->specular_intensity = 10.0f;
    m->index_of_refraction = 1.5f;
    m->alpha = 1.0f;
    m->ambient_color = color;
    m->diffuse_color = color;
    m->specular_color = Vector4(1, 1, 1, 1);
    return m;
}

Scene InitScene() {                                                   // main.cpp:519-535
    LightSource * lights = (LightSource *)calloc(3, sizeof(LightSource));
    lights[0].type = Light_Directional;
    lights[0].color = Vector4(0.9f, 1.0f, 0.95f, 1.0f) * 4.0f;
    lights[0].facing = Normalize(Vector3(1.0f, -1.5f, 0.25f));
    lights[1].type = Light_Directional;
    lights[1].color = Vector4(1.0f, 1.0f, 1.0f, 1.0f) * 1.0f;
    lights[1].facing = Normalize(Vector3(0.0f, -1.0f, 0.0f));
    Scene scene;
    scene.lights = lights;
    scene.light_count = 1;
    return scene;
}

void PopulateSceneObjects(Scene * scene, BoundingHierarchy * h, Mesh * mesh, u32 * out_total_tris) {   // main.cpp:576-599
    u32 total = 0;
    scene->hierarchy = h;
    if (!scene->default_mat) scene->default_mat = MakeMaterial(Vector4(0.75f, 0.5f, 0.75f, 1.0f));
    for (size_t i = 0; i < h->mesh_groups.size(); ++i) {
        MeshGroup * mg = h->mesh_groups[i];
        SceneObject * obj = (SceneObject *)calloc(1, sizeof(SceneObject));
        obj->mesh_group = mg;
        obj->mesh = mesh;
        obj->type = ObjectType_MeshGroup;
        obj->material = scene->default_mat;
        if (mg) {
            total += (u32)(mg->idx_positions.size() / 3);
            if (mg->material) obj->material = mg->material;
        }
        scene->objects.push_back(obj);
    }
    if (out_total_tris) *out_total_tris = total;
}

// -------------------------------------------------------------------------------------------------

prt_camera ToPrtCamera(const Camera * cam) {
    prt_camera c;
    c.tan_a2 = cam->tan_a2; c.aspect = cam->aspect; c.inv_width = cam->inv_width; c.inv_height = cam->inv_height;
    c.position[0] = cam->camera_position.x; c.position[1] = cam->camera_position.y; c.position[2] = cam->camera_position.z;
    c.forward[0] = cam->camera_forward.x; c.forward[1] = cam->camera_forward.y; c.forward[2] = cam->camera_forward.z;
    c.right[0] = cam->camera_right.x; c.right[1] = cam->camera_right.y; c.right[2] = cam->camera_right.z;
    c.up[0] = cam->camera_up.x; c.up[1] = cam->camera_up.y; c.up[2] = cam->camera_up.z;
    return c;
}

prt_params ToPrtParams(const GlobalParams * p) {
    prt_params o;
    memset(&o, 0, sizeof(o));
    o.ray_bias = p->ray_bias;
    o.reflection_samples = p->reflection_samples;
    o.spec_samples = p->spec_samples;
    o.bounce_depth = p->bounce_depth;
    o.background_color[0] = p->background_color.x; o.background_color[1] = p->background_color.y;
    o.background_color[2] = p->background_color.z; o.background_color[3] = p->background_color.w;
    o.spp = p->spp;
    o.pipeline = p->pipeline;
    o.seed = p->seed;
    o.max_spp = p->max_spp;
    o.variance_threshold = 0.0f;                          // the reference's 0.01 (main.cpp:254)
    return o;
}
