// bvh_build.h - host-side per-triangle BVH2 builder for the HIP traversal kernels.
//
// The reference's acceleration structure is a sphere tree over whole OBJ groups with brute force inside
// each leaf (bsphere.cpp:379-444, raytracer.cpp:136-154).  Its RESULT is the exact closest front-facing
// hit over all triangles, so any conservative structure gives the same answer; the
// device uses a binned-SAH binary BVH with <= 4 triangles per leaf.
#pragma once

#include <cstdint>
#include <vector>

namespace prt {

struct BvhResult {
    std::vector<float> nodes;          // 16 floats (64 B) per internal node, layout in dev_scene.h
    std::vector<uint32_t> tri_order;   // tri_order[i] = input triangle stored at leaf-order slot i
    uint32_t node_count = 0;
    uint32_t max_depth = 0;            // deepest leaf; bounds the traversal stack
    float scene_lo[3] = { 0, 0, 0 }, scene_hi[3] = { 0, 0, 0 };
};

// verts: 9 floats per triangle (a, b, c).  leaf_max <= 4 (2 bits in the leaf link).
void build_bvh2(const float * verts, uint32_t n_tris, uint32_t leaf_max, uint32_t threads, BvhResult * out);

// 4-wide BVH with child boxes quantised to 8 bits per plane relative to the node's own box: 64 B per node
// (layout in dev_scene.h).  Built by collapsing the binned-SAH binary tree (largest-area child first) and
// rounding every child box OUTWARD onto the node's 2^e grid, so it stays conservative.
struct Bvh4Result {
    std::vector<uint32_t> nodes;       // 16 dwords (64 B) per node
    std::vector<uint32_t> tri_order;
    uint32_t node_count = 0;
    uint32_t max_depth = 0;            // in 4-wide nodes
    uint32_t stack_bound = 0;          // entries a traversal can ever hold: 3 per level + sentinel
    float scene_lo[3] = { 0, 0, 0 }, scene_hi[3] = { 0, 0, 0 };
};
void build_bvh4q(const float * verts, uint32_t n_tris, uint32_t leaf_max, uint32_t threads, Bvh4Result * out, float trav_cost = 1.0f);

// The same 4-wide quantised result from a binary radix tree built elsewhere (GPU LBVH, bvh_lbvh.h); see bvh_build.cpp.
void build_bvh4q_from_radix_tree(uint32_t n_tris, uint32_t leaf_max, const int32_t * left, const int32_t * right,
                                 const uint32_t * first, const uint32_t * last, const float * node_box, const float * leaf_box,
                                 const uint32_t * sorted_ids, Bvh4Result * out);

}  // namespace prt
