// scene_flatten.h - Scene pointer graph -> the POD arrays of include/prt.h.
#pragma once

#include <vector>

#include "../../include/prt.h"
#include "prt_scene.h"

// Owns the storage a prt_scene_desc points into.  Valid while the FlatScene lives and is not modified.
struct FlatScene {
    std::vector<float> positions, normals, texcoords, tangents;
    std::vector<u32> idx_positions, idx_texcoords, idx_normals;
    std::vector<prt_group> groups;
    std::vector<prt_material> materials;
    std::vector<prt_texture> textures;          // texel pointers alias the Scene's Texture objects
    std::vector<prt_light> lights;
    std::vector<prt_bsphere> spheres;
    std::vector<s32> sphere_group;
    prt_scene_desc desc;
};

// Walks scene->objects / scene->hierarchy exactly as the reference's shading code would
// (material = group material or scene->default_mat, main.cpp:586-589).  Material 0 is default_mat.
void FlattenScene(const Scene * scene, FlatScene * out);

prt_camera ToPrtCamera(const Camera * cam);
prt_params ToPrtParams(const GlobalParams * p);
