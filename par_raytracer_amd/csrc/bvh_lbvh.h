// bvh_lbvh.h - binary radix tree over the triangles, built on the GPU ("GPU LBVH later").
//
// The default acceleration structure is a binned-SAH tree built on the host (bvh_build.cpp, 0.5 s for 1 M triangles);
// this is the fast alternative for scenes that change between frames: Morton codes of the triangle centroids (21 bits per
// axis), a device radix sort, Karras' parallel radix-tree construction (one lane per internal node, longest common
// prefix of neighbouring keys, ties broken by position) and a bottom-up box fit (one lane per triangle climbs while it
// is the second child to arrive).  The tree comes back to the host, where the same collapse-to-4-wide / quantise back
// end as the SAH builder turns it into the 64 B nodes the traversal kernels read (bvh_build.cpp
// build_bvh4q_from_radix_tree).  Any conservative tree gives the same image; this one is built ~10x
// faster and traverses slower (§6).
#pragma once

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <vector>

#include "bvh_build.h"

namespace prt {

struct LbvhTree {                        // host copies of the device result, see build_bvh4q_from_radix_tree
    std::vector<int32_t> left, right;
    std::vector<uint32_t> first, last;
    std::vector<float> node_box, leaf_box;
    std::vector<uint32_t> sorted_ids;
};

namespace lbvh {

__device__ inline unsigned long long spread21(unsigned int v) {        // 21 bits -> every third bit of 63
    unsigned long long x = v & 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

// One lane per triangle: its box, and the Morton code of the box centre in the scene's bounds.
__global__ void k_prims(const float * verts, unsigned int n, float lox, float loy, float loz, float sx, float sy, float sz,
                        float * boxes, unsigned long long * keys, unsigned int * ids) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float * v = verts + 9 * (size_t)i;
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = fminf(fminf(v[a], v[3 + a]), v[6 + a]);
        hi[a] = fmaxf(fmaxf(v[a], v[3 + a]), v[6 + a]);
    }
    for (int a = 0; a < 3; ++a) { boxes[6 * (size_t)i + a] = lo[a]; boxes[6 * (size_t)i + 3 + a] = hi[a]; }
    const float cx = (0.5f * (lo[0] + hi[0]) - lox) * sx, cy = (0.5f * (lo[1] + hi[1]) - loy) * sy, cz = (0.5f * (lo[2] + hi[2]) - loz) * sz;
    const unsigned int qx = (unsigned int)fminf(fmaxf(cx, 0.0f), 2097151.0f);
    const unsigned int qy = (unsigned int)fminf(fmaxf(cy, 0.0f), 2097151.0f);
    const unsigned int qz = (unsigned int)fminf(fmaxf(cz, 0.0f), 2097151.0f);
    keys[i] = spread21(qx) | spread21(qy) << 1 | spread21(qz) << 2;
    ids[i] = i;
}

// Length of the common prefix of keys i and j (-1 outside the array); equal keys are told apart by their positions.
__device__ inline int delta(const unsigned long long * keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}

// Karras 2012, one lane per internal node.  Children: >= 0 internal node, < 0 = ~(sorted position of a triangle).
__global__ void k_hierarchy(const unsigned long long * keys, int n, int * left, int * right, unsigned int * first, unsigned int * last,
                            int * parent_internal, int * parent_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2; ; t = (t + 1) / 2) {               // ceil(l / 2), ceil(l / 4), ... 1
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int lc = lo == gamma ? ~gamma : gamma;
    const int rc = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    left[i] = lc;
    right[i] = rc;
    first[i] = (unsigned int)lo;
    last[i] = (unsigned int)hi;
    if (lc < 0) parent_leaf[~lc] = i; else parent_internal[lc] = i;
    if (rc < 0) parent_leaf[~rc] = i; else parent_internal[rc] = i;
    if (i == 0) parent_internal[0] = -1;
}

// One lane per triangle (sorted position): writes its box in sorted order, then climbs; at every internal node the
// first child to arrive stops, the second merges the two child boxes (both complete by then) and goes on.
__global__ void k_fit(const float * boxes, const unsigned int * ids, int n, const int * left, const int * right, const int * parent_internal,
                      const int * parent_leaf, float * leaf_box, float * node_box, unsigned int * arrived) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float * src = boxes + 6 * (size_t)ids[i];
    for (int a = 0; a < 6; ++a) leaf_box[6 * (size_t)i + a] = src[a];
    __threadfence();
    int p = n > 1 ? parent_leaf[i] : -1;
    while (p >= 0) {
        if (atomicAdd(&arrived[p], 1u) == 0u) return;             // the sibling subtree is not finished yet
        __threadfence();
        const int lc = left[p], rc = right[p];
        const float * lb = lc < 0 ? leaf_box + 6 * (size_t)~lc : node_box + 6 * (size_t)lc;
        const float * rb = rc < 0 ? leaf_box + 6 * (size_t)~rc : node_box + 6 * (size_t)rc;
        // children's boxes were written by other lanes: read them past the caches of this CU
        float out[6];
        for (int a = 0; a < 3; ++a) {
            out[a] = fminf(__builtin_nontemporal_load(lb + a), __builtin_nontemporal_load(rb + a));
            out[3 + a] = fmaxf(__builtin_nontemporal_load(lb + 3 + a), __builtin_nontemporal_load(rb + 3 + a));
        }
        for (int a = 0; a < 6; ++a) node_box[6 * (size_t)p + a] = out[a];
        __threadfence();
        p = parent_internal[p];
    }
}

}  // namespace lbvh

// verts: 9 floats per triangle on the HOST (copied in); scene bounds from the caller.  Returns hipSuccess and fills `t`.
inline hipError_t build_lbvh_tree(const float * verts, uint32_t n, const float * scene_lo, const float * scene_hi, hipStream_t stream,
                                  LbvhTree * t, double * device_ms) {
    *t = LbvhTree();
    if (n == 0) return hipSuccess;
    hipError_t e = hipSuccess;
    float * d_verts = nullptr, * d_boxes = nullptr, * d_leaf_box = nullptr, * d_node_box = nullptr;
    unsigned long long * d_keys = nullptr, * d_keys2 = nullptr;
    unsigned int * d_ids = nullptr, * d_ids2 = nullptr, * d_first = nullptr, * d_last = nullptr, * d_arrived = nullptr;
    int * d_left = nullptr, * d_right = nullptr, * d_pi = nullptr, * d_pl = nullptr;
    void * d_tmp = nullptr;
    size_t tmp_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const size_t ni = n > 1 ? n - 1 : 1;
#define LB_TRY(call) do { e = (call); if (e != hipSuccess) goto done; } while (0)
    LB_TRY(hipMalloc((void **)&d_verts, (size_t)n * 36));
    LB_TRY(hipMalloc((void **)&d_boxes, (size_t)n * 24));
    LB_TRY(hipMalloc((void **)&d_leaf_box, (size_t)n * 24));
    LB_TRY(hipMalloc((void **)&d_node_box, ni * 24));
    LB_TRY(hipMalloc((void **)&d_keys, (size_t)n * 8));
    LB_TRY(hipMalloc((void **)&d_keys2, (size_t)n * 8));
    LB_TRY(hipMalloc((void **)&d_ids, (size_t)n * 4));
    LB_TRY(hipMalloc((void **)&d_ids2, (size_t)n * 4));
    LB_TRY(hipMalloc((void **)&d_first, ni * 4));
    LB_TRY(hipMalloc((void **)&d_last, ni * 4));
    LB_TRY(hipMalloc((void **)&d_arrived, ni * 4));
    LB_TRY(hipMalloc((void **)&d_left, ni * 4));
    LB_TRY(hipMalloc((void **)&d_right, ni * 4));
    LB_TRY(hipMalloc((void **)&d_pi, ni * 4));
    LB_TRY(hipMalloc((void **)&d_pl, (size_t)n * 4));
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys, d_keys2, d_ids, d_ids2, (int)n, 0, 63, stream));
    LB_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1));
    LB_TRY(hipEventCreate(&ev0));
    LB_TRY(hipEventCreate(&ev1));
    LB_TRY(hipMemcpyAsync(d_verts, verts, (size_t)n * 36, hipMemcpyHostToDevice, stream));
    LB_TRY(hipMemsetAsync(d_arrived, 0, ni * 4, stream));
    LB_TRY(hipEventRecord(ev0, stream));
    {
        float s[3];
        for (int a = 0; a < 3; ++a) {
            const float ext = scene_hi[a] - scene_lo[a];
            s[a] = ext > 0.0f ? 2097151.0f / ext : 0.0f;
        }
        const unsigned int grid = (n + 255) / 256;
        hipLaunchKernelGGL(lbvh::k_prims, dim3(grid), dim3(256), 0, stream, d_verts, n, scene_lo[0], scene_lo[1], scene_lo[2], s[0], s[1], s[2],
                           d_boxes, d_keys, d_ids);
        LB_TRY(hipGetLastError());
        LB_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_keys, d_keys2, d_ids, d_ids2, (int)n, 0, 63, stream));
        if (n > 1) {
            hipLaunchKernelGGL(lbvh::k_hierarchy, dim3((n - 1 + 255) / 256), dim3(256), 0, stream, d_keys2, (int)n, d_left, d_right, d_first, d_last,
                               d_pi, d_pl);
            LB_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(lbvh::k_fit, dim3(grid), dim3(256), 0, stream, d_boxes, d_ids2, (int)n, d_left, d_right, d_pi, d_pl, d_leaf_box, d_node_box,
                           d_arrived);
        LB_TRY(hipGetLastError());
    }
    LB_TRY(hipEventRecord(ev1, stream));
    LB_TRY(hipStreamSynchronize(stream));
    if (device_ms) { float ms = 0.0f; LB_TRY(hipEventElapsedTime(&ms, ev0, ev1)); *device_ms = ms; }
    t->left.resize(ni); t->right.resize(ni); t->first.resize(ni); t->last.resize(ni);
    t->node_box.resize(ni * 6); t->leaf_box.resize((size_t)n * 6); t->sorted_ids.resize(n);
    if (n > 1) {
        LB_TRY(hipMemcpy(t->left.data(), d_left, ni * 4, hipMemcpyDeviceToHost));
        LB_TRY(hipMemcpy(t->right.data(), d_right, ni * 4, hipMemcpyDeviceToHost));
        LB_TRY(hipMemcpy(t->first.data(), d_first, ni * 4, hipMemcpyDeviceToHost));
        LB_TRY(hipMemcpy(t->last.data(), d_last, ni * 4, hipMemcpyDeviceToHost));
        LB_TRY(hipMemcpy(t->node_box.data(), d_node_box, ni * 24, hipMemcpyDeviceToHost));
    }
    LB_TRY(hipMemcpy(t->leaf_box.data(), d_leaf_box, (size_t)n * 24, hipMemcpyDeviceToHost));
    LB_TRY(hipMemcpy(t->sorted_ids.data(), d_ids2, (size_t)n * 4, hipMemcpyDeviceToHost));
done:
#undef LB_TRY
    (void)hipFree(d_verts); (void)hipFree(d_boxes); (void)hipFree(d_leaf_box); (void)hipFree(d_node_box); (void)hipFree(d_keys); (void)hipFree(d_keys2);
    (void)hipFree(d_ids); (void)hipFree(d_ids2); (void)hipFree(d_first); (void)hipFree(d_last); (void)hipFree(d_arrived); (void)hipFree(d_left);
    (void)hipFree(d_right); (void)hipFree(d_pi); (void)hipFree(d_pl); (void)hipFree(d_tmp);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    return e;
}

}  // namespace prt
