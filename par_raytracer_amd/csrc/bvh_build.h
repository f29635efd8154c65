// bvh_build.h - host-side per-triangle BVH2 builder for the HIP traversal kernels.
//
// The reference's acceleration structure is a sphere tree over whole OBJ groups with brute force inside
// each leaf (bsphere.cpp:379-444, raytracer.cpp:136-154).  Its RESULT is the exact closest front-facing
// hit over all triangles, so any conservative structure gives the same answer; the
// device uses a binned-SAH binary BVH with <= 4 triangles per leaf.
#pragma once

#include <cstdint>
#include <vector>

namespace prt { struct BvhBuildOptions; }     // prt_options.h: bins, sweep, collapse rule, GPU-builder back end; NULL = defaults

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
void build_bvh4q(const float * verts, uint32_t n_tris, uint32_t leaf_max, uint32_t threads, Bvh4Result * out, float trav_cost = 1.0f,
                 const BvhBuildOptions * opt = nullptr);

// The same 4-wide quantised result from a binary radix tree built elsewhere (GPU LBVH, bvh_lbvh.h); see bvh_build.cpp.
void build_bvh4q_from_radix_tree(uint32_t n_tris, uint32_t leaf_max, const int32_t * left, const int32_t * right,
                                 const uint32_t * first, const uint32_t * last, const float * node_box, const float * leaf_box,
                                 const uint32_t * sorted_ids, Bvh4Result * out, const BvhBuildOptions * opt = nullptr);

// 8-wide compressed BVH: 80 B per node (5 x dwordx4), child boxes quantised to 8 bits per plane on the node's own
// power-of-two grid, children assigned to slots so that (slot XOR ray octant) ascending is a front-to-back visiting order
// (Ylitie, Karras, Laine 2017: no per-step sort), children addressed implicitly (internal children consecutive from
// child_base, leaf triangles consecutive from tri_base, both in slot order).  Layout (dwords; dev_trace8.h reads it):
//    0-2   origin xyz (float: lo corner of the union of the children)
//    3     2^e_x as float bits (bits 23-30) | imask (bits 0-7: slot holds an internal node) | lmask << 8 (slot holds a leaf)
//    4     child_base: node index of the first internal child
//    5     tri_base: leaf-order index of the first triangle of the first leaf slot
//    6     2^e_y as float bits | c0 (bits 0-7) | c1 << 8: a leaf slot holds 1 + c0 + 2 c1 triangles (1..4)
//    7     2^e_z as float bits
//    8-9   lo x of slots 0-3, 4-7 (one byte per slot)     10-11 lo y     12-13 lo z
//    14-15 hi x                                            16-17 hi y     18-19 hi z
// An empty slot has lo = 255 > hi = 0 on every axis and is in neither mask.
struct Bvh8Result {
    std::vector<uint32_t> nodes;       // 20 dwords (80 B) per node
    std::vector<uint32_t> tri_order;   // tri_order[i] = input triangle stored at leaf-order slot i
    uint32_t node_count = 0;
    uint32_t max_depth = 0;            // in 8-wide nodes
    uint32_t stack_bound = 0;          // entries a traversal can ever hold: one group per level + the marker
    float scene_lo[3] = { 0, 0, 0 }, scene_hi[3] = { 0, 0, 0 };
};
enum { BVH8_NODE_DWORDS = 20 };
void build_bvh8q(const float * verts, uint32_t n_tris, uint32_t leaf_max, uint32_t threads, Bvh8Result * out, float trav_cost = 1.0f,
                 const BvhBuildOptions * opt = nullptr);
void build_bvh8q_from_radix_tree(uint32_t n_tris, uint32_t leaf_max, const int32_t * left, const int32_t * right,
                                 const uint32_t * first, const uint32_t * last, const float * node_box, const float * leaf_box,
                                 const uint32_t * sorted_ids, Bvh8Result * out, const BvhBuildOptions * opt = nullptr);

}  // namespace prt
