// prt_scene.h - the reference's scene data model, kept as the host-side API surface.
//
// BASELINE.json's north_star keeps "the existing Scene/Mesh/Material structs, the OBJ loader and the
// stb_image_write output path as the API surface so main.cpp can call the new path as a drop-in".
// These are the same struct and field names as mesh.h:8-57, scene.h:3-36, bsphere.cpp:316-326,
// globals.h:3-22 and main.cpp:69-73,133-143; the functions declared at the bottom have the reference's
// names and argument meaning.  Implementations are new (par_raytracer_amd/host/*.cpp).
#pragma once

#include <string>
#include <unordered_map>
#include <vector>

#include "prt_math.h"

struct Texture {                    // mesh.h:8-13
    u32 size_x;
    u32 size_y;
    u32 channels;
    u8 * texels;
};

struct Material {                   // mesh.h:15-32
    float specular_intensity;
    float index_of_refraction;
    float alpha;
    char * name;

    Vector4 ambient_color;
    Vector4 diffuse_color;
    Vector4 specular_color;
    Vector4 emissive_color;

    Texture * ambient_texture;
    Texture * diffuse_texture;
    Texture * specular_texture;
    Texture * alpha_texture;
    Texture * bump_texture;
};

struct MaterialLibrary {            // mesh.h:34-36
    std::unordered_map<std::string, Material *> materials;
    std::vector<Material *> in_file_order;   // addition: deterministic flattening order
};

typedef std::vector<u32> IndexBuffer;

struct MeshGroup {                  // mesh.h:40-47
    IndexBuffer idx_positions;
    IndexBuffer idx_texcoords;
    IndexBuffer idx_normals;
    char * name;
    Material * material;
    MeshGroup() : name(NULL), material(NULL) {}
};

struct Mesh {                       // mesh.h:49-57
    std::vector<MeshGroup> groups;
    std::vector<Vector3> positions;
    std::vector<Vector2> texcoords;
    std::vector<Vector3> normals;
    std::vector<Vector3> tangents;
    MaterialLibrary * material_library;
    Mesh() : material_library(NULL) {}
};

enum LightSourceType { Light_Directional, Light_Point };   // scene.h:3-7

struct LightSource {                // scene.h:9-15
    LightSourceType type;
    Vector4 color;
    Vector3 position;
    Vector3 facing;
    float falloff;
};

enum ObjectType { ObjectType_Sphere, ObjectType_MeshGroup };   // scene.h:17-20

struct SceneObject {                // scene.h:22-27
    MeshGroup * mesh_group;
    Mesh * mesh;
    ObjectType type;
    Material * material;
};

struct BoundingSphere {             // bsphere.cpp:316-320
    Sphere s;
    u32 c0;
    u32 c1;
};

struct BoundingHierarchy {          // bsphere.cpp:322-326
    std::vector<BoundingSphere> spheres;
    std::vector<MeshGroup *> mesh_groups;
    Mesh * mesh;
    BoundingHierarchy() : mesh(NULL) {}
};

struct Scene {                      // scene.h:29-36
    std::vector<SceneObject *> objects;
    BoundingHierarchy * hierarchy;
    LightSource * lights;
    u32 light_count;
    Material * default_mat;
    Scene() : hierarchy(NULL), lights(NULL), light_count(0), default_mat(NULL) {}
};

struct DebugCounters {              // globals.h:3-7 (+ what a per-triangle BVH can report)
    u64 ray_count;
    u64 sphere_check_count;         // BVH node visits on the HIP path
    u64 mesh_check_count;           // triangle tests on the HIP path
};

struct GlobalParams {               // globals.h:9-22 (the reference's anonymous global gParams)
    float ray_bias;
    u32 reflection_samples;
    u32 spec_samples;
    u32 bounce_depth;
    Vector4 background_color;
    float camera_fov;
    Vector3 camera_position;
    Vector3 camera_facing;
    char * image_output_filename;
    char * data_dirname;
    u32 image_width;
    u32 image_height;
    // additions (no reference flag exists for these; SURVEY.md §5 "Config / flags")
    u32 spp;
    u32 max_spp;                    // > spp: the reference's adaptive mode (main.cpp:245-258, it hard-codes 10 / 50)
    u64 seed;
    char * obj_filename;
    u32 pipeline;
};
extern GlobalParams gParams;

struct Camera {                     // main.cpp:133-143
    float tan_a2;
    float aspect;
    float inv_width;
    float inv_height;
    Vector3 camera_position;
    Vector3 camera_forward;
    Vector3 camera_right;
    Vector3 camera_up;
};

struct Framebuffer {                // main.cpp:69-73
    Vector4 * pixels;
    u32 width;
    u32 height;
};

// ---- functions with the reference's names -------------------------------------------------------
void InitParams(int argc, char ** argv);                                  // main.cpp:416-504
Camera MakeCamera(float fov, u32 width, u32 height);                      // main.cpp:145-162
Material * MakeMaterial(Vector4 color);                                   // main.cpp:506-517
Scene InitScene();                                                        // main.cpp:519-535
Mesh * ParseOBJ(const char * working_dir, const char * filename, Matrix33 transform);   // obj_parser.cpp:348-426
Texture * LoadTexture(const char * filename);                             // obj_parser.cpp:197-213 (image_in.cpp)
void FreeTexture(Texture * t);
Texture * ConvertHeightMapToNormalMap(const Texture * height_map);        // texture.cpp:102-143
const char * TextureLoadError();
void CalculateTangents(Mesh * mesh);                                      // mesh.h:59-130
void BuildHierarchy(BoundingHierarchy * h, Mesh * mesh);                  // bsphere.cpp:379-444
void PopulateSceneObjects(Scene * scene, BoundingHierarchy * h, Mesh * mesh, u32 * out_total_tris);  // main.cpp:576-599
Framebuffer Render(Camera * cam, Scene * scene, u32 width, u32 height);   // main.cpp:301-358 (HIP path behind it)
void WriteFramebufferImage(Framebuffer * fb, const char * filename);      // main.cpp:101-131
extern "C" int stbi_write_png(char const * filename, int w, int h, int comp, const void * data, int stride_in_bytes);

// Last counters / timings of Render(), for the driver's report (main.cpp:351-356).
struct RenderReport {
    DebugCounters counters;
    u64 shaded_hits;
    double render_ms;
    double trace_kernel_ms;
    u32 gpu_count;
    int status;                     // 0 = the frame was rendered; negative = the HIP path failed (the frame is zero-filled)
};
extern RenderReport gLastRenderReport;
