/* prt_flatten_ref.h - the reference's scene graph -> the POD arrays of prt.h (level-1 drop-in).
 *
 * C++11, header only.  Include it AFTER the reference's own headers (mesh.h, scene.h, bsphere.cpp - in the reference,
 * main.cpp includes all of them), i.e. where these names are already declared with the reference's fields:
 *
 *     Scene { objects, hierarchy, lights, light_count, default_mat }            scene.h:29-36
 *     SceneObject { mesh_group, mesh, type, material }                          scene.h:22-27
 *     LightSource { type, color, position, facing, falloff }, Light_Point       scene.h:3-15
 *     Mesh { groups, positions, texcoords, normals, tangents }                  mesh.h:49-57
 *     MeshGroup { idx_positions, idx_texcoords, idx_normals, material }         mesh.h:40-47
 *     Material { Ns, Ni, d, Ka, Kd, Ks, five Texture* }, Texture                mesh.h:8-32
 *     BoundingHierarchy { spheres, mesh_groups, mesh }, BoundingSphere          bsphere.cpp:316-326
 *
 * It uses nothing else of the reference.  This repository's host mirror (par_raytracer_amd/host/prt_scene.h) declares
 * the same names with the same fields, and its FlattenScene is this very function: the flattening that feeds the HIP
 * path from `prt_main` and the one a maintainer of the reference adds next to RenderTask (main.cpp:267-283) are one
 * piece of code.  oracle/ref_harness.cpp compiles it against the reference's real headers (--dump-desc), and
 * tests/test_host_side.py checks that what comes out is byte for byte what the host mirror produces.
 *
 * What it walks, and why in this way:
 *   - TraceRay visits hierarchy->spheres[i] and, at a leaf, scene->objects[i] (raytracer.cpp:172, 220); shading reads
 *     hit.object->material (raytracer.cpp:427).  So the material of a group is the material of the OBJECT that carries the
 *     group - which main() set to the group's own material or scene->default_mat (main.cpp:586-589) - and not necessarily
 *     MeshGroup::material.
 *   - IntersectRayMesh runs over mesh_group->idx_positions in order (raytracer.cpp:136): the three index buffers of all
 *     groups are concatenated in mesh->groups order and a prt_group is a run of them.
 *   - hierarchy->spheres / mesh_groups go out as they are (24-byte nodes; leaf -> group index): the HIP path only derives
 *     the reference's leaf visit order from them, for ties in t.
 * Material 0 is scene->default_mat; the others follow in order of first use by the groups.  Texture slots index
 * RefFlatScene::textures, one entry per distinct Texture object, texel pointers aliasing the scene's own decoded bytes.
 */
#ifndef PRT_FLATTEN_REF_H_
#define PRT_FLATTEN_REF_H_

#include <stdint.h>
#include <string.h>

#include <map>
#include <vector>

#include "prt.h"

/* Owns the storage a prt_scene_desc points into.  Valid while it lives and is not modified (and, for the texel pointers,
 * while the scene's Texture objects live). */
struct RefFlatScene {
    std::vector<float> positions, normals, texcoords, tangents;
    std::vector<uint32_t> idx_positions, idx_texcoords, idx_normals;
    std::vector<prt_group> groups;
    std::vector<prt_material> materials;
    std::vector<prt_texture> textures;
    std::vector<prt_light> lights;
    std::vector<prt_bsphere> spheres;
    std::vector<int32_t> sphere_group;
    prt_scene_desc desc;
};

namespace prt_flatten_detail {

inline int32_t TextureSlot(const Texture * t, std::map<const Texture *, int32_t> * index, std::vector<prt_texture> * out) {
    if (!t) return -1;
    std::map<const Texture *, int32_t>::iterator it = index->find(t);
    if (it != index->end()) return it->second;
    prt_texture pt;
    pt.size_x = t->size_x; pt.size_y = t->size_y; pt.channels = t->channels; pt.texels = t->texels;
    int32_t slot = (int32_t)out->size();
    out->push_back(pt);
    (*index)[t] = slot;
    return slot;
}

template <class V4>
inline void Put4(float * dst, const V4 & v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
template <class V3>
inline void Put3(float * dst, const V3 & v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

inline prt_material ToPrtMaterial(const Material * m, std::map<const Texture *, int32_t> * tex_index, std::vector<prt_texture> * textures) {
    prt_material o;
    memset(&o, 0, sizeof(o));
    o.specular_intensity = m->specular_intensity;
    o.index_of_refraction = m->index_of_refraction;
    o.alpha = m->alpha;
    Put4(o.ambient_color, m->ambient_color);
    Put4(o.diffuse_color, m->diffuse_color);
    Put4(o.specular_color, m->specular_color);
    o.ambient_texture = TextureSlot(m->ambient_texture, tex_index, textures);
    o.diffuse_texture = TextureSlot(m->diffuse_texture, tex_index, textures);
    o.specular_texture = TextureSlot(m->specular_texture, tex_index, textures);
    o.alpha_texture = TextureSlot(m->alpha_texture, tex_index, textures);
    o.bump_texture = TextureSlot(m->bump_texture, tex_index, textures);
    return o;
}

}  /* namespace prt_flatten_detail */

inline void FlattenReferenceScene(const Scene * scene, RefFlatScene * out) {
    using namespace prt_flatten_detail;
    const BoundingHierarchy * h = scene->hierarchy;
    const Mesh * mesh = h->mesh;
    *out = RefFlatScene();

    out->positions.resize(mesh->positions.size() * 3);
    for (size_t i = 0; i < mesh->positions.size(); ++i) Put3(&out->positions[3 * i], mesh->positions[i]);
    out->normals.resize(mesh->normals.size() * 3);
    for (size_t i = 0; i < mesh->normals.size(); ++i) Put3(&out->normals[3 * i], mesh->normals[i]);
    out->texcoords.resize(mesh->texcoords.size() * 2);
    for (size_t i = 0; i < mesh->texcoords.size(); ++i) {
        out->texcoords[2 * i] = mesh->texcoords[i].x;
        out->texcoords[2 * i + 1] = mesh->texcoords[i].y;
    }
    if (mesh->tangents.size() == mesh->normals.size()) {          /* CalculateTangents ran (mesh.h:59-130) */
        out->tangents.resize(mesh->tangents.size() * 3);
        for (size_t i = 0; i < mesh->tangents.size(); ++i) Put3(&out->tangents[3 * i], mesh->tangents[i]);
    }

    /* the material a hit on group g is shaded with: that of the object TraceRay finds the group through */
    const MeshGroup * g0 = mesh->groups.empty() ? NULL : &mesh->groups[0];
    std::vector<const Material *> group_material(mesh->groups.size(), (const Material *)NULL);
    for (size_t i = 0; i < h->mesh_groups.size() && i < scene->objects.size(); ++i) {
        const SceneObject * obj = scene->objects[i];
        if (!obj || !obj->mesh_group) continue;
        const size_t g = (size_t)(obj->mesh_group - g0);
        if (g < group_material.size()) group_material[g] = obj->material;
    }

    std::map<const Material *, int32_t> mat_index;
    std::map<const Texture *, int32_t> tex_index;
    out->materials.push_back(ToPrtMaterial(scene->default_mat, &tex_index, &out->textures));
    mat_index[scene->default_mat] = 0;

    for (size_t g = 0; g < mesh->groups.size(); ++g) {
        const MeshGroup * mg = &mesh->groups[g];
        const Material * m = group_material[g] ? group_material[g] : (mg->material ? mg->material : scene->default_mat);
        if (!mat_index.count(m)) {
            mat_index[m] = (int32_t)out->materials.size();
            out->materials.push_back(ToPrtMaterial(m, &tex_index, &out->textures));
        }
        prt_group pg;
        pg.first_index = (uint32_t)out->idx_positions.size();
        pg.index_count = (uint32_t)mg->idx_positions.size();
        pg.material = mat_index[m];
        out->groups.push_back(pg);
        out->idx_positions.insert(out->idx_positions.end(), mg->idx_positions.begin(), mg->idx_positions.end());
        out->idx_texcoords.insert(out->idx_texcoords.end(), mg->idx_texcoords.begin(), mg->idx_texcoords.end());
        out->idx_normals.insert(out->idx_normals.end(), mg->idx_normals.begin(), mg->idx_normals.end());
    }

    for (uint32_t i = 0; i < scene->light_count; ++i) {
        const LightSource * l = &scene->lights[i];
        prt_light pl;
        memset(&pl, 0, sizeof(pl));
        pl.type = (l->type == Light_Point) ? PRT_LIGHT_POINT : PRT_LIGHT_DIRECTIONAL;
        Put4(pl.color, l->color);
        Put3(pl.position, l->position);
        Put3(pl.facing, l->facing);
        pl.falloff = l->falloff;
        out->lights.push_back(pl);
    }

    for (size_t i = 0; i < h->spheres.size(); ++i) {
        prt_bsphere s;
        Put3(s.center, h->spheres[i].s.center);
        s.radius = h->spheres[i].s.radius;
        s.c0 = h->spheres[i].c0;
        s.c1 = h->spheres[i].c1;
        out->spheres.push_back(s);
        const MeshGroup * mg = h->mesh_groups[i];
        out->sphere_group.push_back(mg ? (int32_t)(mg - g0) : -1);
    }

    prt_scene_desc & d = out->desc;
    memset(&d, 0, sizeof(d));
    d.positions = out->positions.data();   d.position_count = (uint32_t)mesh->positions.size();
    d.normals = out->normals.data();       d.normal_count = (uint32_t)mesh->normals.size();
    d.texcoords = out->texcoords.data();   d.texcoord_count = (uint32_t)mesh->texcoords.size();
    d.tangents = out->tangents.empty() ? NULL : out->tangents.data();
    d.idx_positions = out->idx_positions.data();
    d.idx_texcoords = out->idx_texcoords.data();
    d.idx_normals = out->idx_normals.data();
    d.index_count = (uint32_t)out->idx_positions.size();
    d.groups = out->groups.data();         d.group_count = (uint32_t)out->groups.size();
    d.materials = out->materials.data();   d.material_count = (uint32_t)out->materials.size();
    d.textures = out->textures.empty() ? NULL : out->textures.data();
    d.texture_count = (uint32_t)out->textures.size();
    d.lights = out->lights.data();         d.light_count = (uint32_t)out->lights.size();
    d.spheres = out->spheres.data();
    d.sphere_group = out->sphere_group.data();
    d.sphere_count = (uint32_t)out->spheres.size();
}

#endif /* PRT_FLATTEN_REF_H_ */
