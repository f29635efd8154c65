// bvh_build.cpp - binned surface-area-heuristic BVH2 builder (host, multi-threaded over subtrees).
#include "bvh_build.h"
#include "prt_options.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <thread>

namespace prt {
namespace {

struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; ++a) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; } }
    void grow(const Box & b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    void grow(const float * p) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0.0f) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Prim {
    Box box;
    float c[3];
    uint32_t id;
};

struct TmpNode {
    Box box;
    int32_t left, right;        // TmpNode indices, -1 for a leaf
    uint32_t first, count;      // primitive range (leaf)
    uint32_t depth;
};

enum { MAX_BINS = 64, MAX_FORCED_DEPTH = 56 };

struct Builder {
    std::vector<Prim> prims;
    std::vector<TmpNode> pool;
    std::atomic<uint32_t> next_node;
    std::atomic<uint32_t> max_depth;
    std::atomic<int> threads_free;
    uint32_t leaf_max;
    float trav_cost = 1.0f;     // SAH: cost of one traversal step relative to one triangle test
    int BINS = 16;              // SAH bins per axis (PRT_SAH_BINS, <= MAX_BINS)
    uint32_t sweep_max = 0;     // nodes of at most this many triangles try every split position of every axis (PRT_SAH_SWEEP)
    BvhBuildOptions opt;        // how the caller wants the tree built (prt_options.h); defaults when it said nothing
    void configure(const BvhBuildOptions * o) {
        if (o) opt = *o;
        if (opt.sah_bins >= 0) BINS = std::max(4, std::min((int)MAX_BINS, (int)opt.sah_bins));
        if (opt.sah_sweep >= 0) sweep_max = (uint32_t)opt.sah_sweep;
    }

    uint32_t alloc() { return next_node.fetch_add(1); }

    // Builds the subtree over prims[first, first+count) into node `me`.
    void build(uint32_t me, uint32_t first, uint32_t count, uint32_t depth) {
        TmpNode & n = pool[me];
        n.depth = depth;
        Box bounds, cbounds;
        bounds.reset();
        cbounds.reset();
        for (uint32_t i = first; i < first + count; ++i) {
            bounds.grow(prims[i].box);
            cbounds.grow(prims[i].c);
        }
        n.box = bounds;
        n.left = n.right = -1;
        n.first = first;
        n.count = count;

        auto make_leaf = [&]() {
            uint32_t d = max_depth.load();
            while (depth > d && !max_depth.compare_exchange_weak(d, depth)) {}
        };
        if (count == 1) { make_leaf(); return; }

        // --- binned SAH over the three axes
        int best_axis = -1, best_split = -1;
        float best_cost = FLT_MAX;
        float parent_area = bounds.half_area();
        if (depth < MAX_FORCED_DEPTH) {
            // one pass over the triangles fills the bins of all three axes (the array is 40 MB for a million triangles: the
            // passes, not the arithmetic, are what a level costs)
            Box bin_box[3][MAX_BINS];
            uint32_t bin_n[3][MAX_BINS];
            float cmin[3], scale[3];
            bool use[3];
            for (int axis = 0; axis < 3; ++axis) {
                cmin[axis] = cbounds.lo[axis];
                use[axis] = cbounds.hi[axis] > cmin[axis];
                scale[axis] = use[axis] ? (float)BINS / (cbounds.hi[axis] - cmin[axis]) : 0.0f;
                for (int k = 0; k < BINS; ++k) { bin_box[axis][k].reset(); bin_n[axis][k] = 0; }
            }
            for (uint32_t i = first; i < first + count; ++i) {
                const Prim & p = prims[i];
                for (int axis = 0; axis < 3; ++axis) {
                    if (!use[axis]) continue;
                    int k = (int)((p.c[axis] - cmin[axis]) * scale[axis]);
                    k = k < 0 ? 0 : (k >= BINS ? BINS - 1 : k);
                    bin_box[axis][k].grow(p.box);
                    bin_n[axis][k]++;
                }
            }
            for (int axis = 0; axis < 3; ++axis) {
                if (!use[axis]) continue;
                float right_area[MAX_BINS];
                uint32_t right_n[MAX_BINS];
                Box acc;
                acc.reset();
                uint32_t cnt = 0;
                for (int k = BINS - 1; k > 0; --k) {
                    acc.grow(bin_box[axis][k]);
                    cnt += bin_n[axis][k];
                    right_area[k] = acc.half_area();
                    right_n[k] = cnt;
                }
                acc.reset();
                cnt = 0;
                for (int k = 0; k < BINS - 1; ++k) {
                    acc.grow(bin_box[axis][k]);
                    cnt += bin_n[axis][k];
                    if (cnt == 0 || right_n[k + 1] == 0) continue;
                    float cost = acc.half_area() * (float)cnt + right_area[k + 1] * (float)right_n[k + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = k; }
                }
            }
        }

        // --- small nodes: every split position of every axis (sorted sweep) instead of the bins
        int sweep_axis = -1;
        uint32_t sweep_left = 0;
        if (count <= sweep_max && depth < MAX_FORCED_DEPTH && count >= 2) {
            std::vector<float> right_area(count);
            for (int axis = 0; axis < 3; ++axis) {
                std::sort(prims.begin() + first, prims.begin() + first + count,
                          [axis](const Prim & a, const Prim & b) { return a.c[axis] < b.c[axis] || (a.c[axis] == b.c[axis] && a.id < b.id); });
                Box acc;
                acc.reset();
                for (uint32_t i = count - 1; i > 0; --i) { acc.grow(prims[first + i].box); right_area[i] = acc.half_area(); }
                acc.reset();
                for (uint32_t i = 0; i + 1 < count; ++i) {
                    acc.grow(prims[first + i].box);
                    float cost = acc.half_area() * (float)(i + 1) + right_area[i + 1] * (float)(count - i - 1);
                    if (cost < best_cost) { best_cost = cost; sweep_axis = axis; sweep_left = i + 1; best_axis = axis; }
                }
            }
            if (sweep_axis >= 0 && sweep_axis != 2)
                std::sort(prims.begin() + first, prims.begin() + first + count,
                          [sweep_axis](const Prim & a, const Prim & b) { return a.c[sweep_axis] < b.c[sweep_axis] || (a.c[sweep_axis] == b.c[sweep_axis] && a.id < b.id); });
        }

        // leaf if allowed and cheaper than splitting (traversal step cost 1, triangle test cost 1)
        if (count <= leaf_max) {
            float split_cost = (best_axis >= 0 && parent_area > 0.0f) ? trav_cost + best_cost / parent_area : FLT_MAX;
            if ((float)count <= split_cost) { make_leaf(); return; }
        }

        uint32_t mid;
        if (sweep_axis >= 0) {
            mid = first + sweep_left;                  // the range is sorted along the winning axis
        } else if (best_axis >= 0) {
            float cmin = cbounds.lo[best_axis];
            float scale = (float)BINS / (cbounds.hi[best_axis] - cmin);
            Prim * b = &prims[first];
            Prim * e = b + count;
            Prim * m = std::partition(b, e, [&](const Prim & p) {
                int bin = (int)((p.c[best_axis] - cmin) * scale);
                bin = bin < 0 ? 0 : (bin >= BINS ? BINS - 1 : bin);
                return bin <= best_split;
            });
            mid = first + (uint32_t)(m - b);
        } else {
            mid = first;
        }
        if (mid == first || mid == first + count) {
            // coincident centroids (or forced depth): median split along the widest centroid axis
            int axis = 0;
            float ext = -1.0f;
            for (int a = 0; a < 3; ++a) {
                float e = cbounds.hi[a] - cbounds.lo[a];
                if (e > ext) { ext = e; axis = a; }
            }
            mid = first + count / 2;
            std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
                             [axis](const Prim & a, const Prim & b) { return a.c[axis] < b.c[axis]; });
        }

        uint32_t l = alloc(), r = alloc();
        pool[me].left = (int32_t)l;
        pool[me].right = (int32_t)r;
        uint32_t lcount = mid - first, rcount = count - lcount;
        bool spawn = lcount > 32768 && rcount > 32768 && threads_free.fetch_sub(1) > 0;
        if (spawn) {
            std::thread t([this, l, first, lcount, depth]() { build(l, first, lcount, depth + 1); });
            build(r, mid, rcount, depth + 1);
            t.join();
            threads_free.fetch_add(1);
        } else {
            if (lcount > 32768 && rcount > 32768) threads_free.fetch_add(1);   // undo the failed reservation
            build(l, first, lcount, depth + 1);
            build(r, mid, rcount, depth + 1);
        }
    }
};

inline float int_bits(int32_t v) {
    float f;
    memcpy(&f, &v, 4);
    return f;
}

}  // namespace

void build_bvh2(const float * verts, uint32_t n_tris, uint32_t leaf_max, uint32_t threads, BvhResult * out) {
    *out = BvhResult();
    if (leaf_max < 1) leaf_max = 1;
    if (leaf_max > 4) leaf_max = 4;
    Builder b;
    b.leaf_max = leaf_max;
    b.prims.resize(n_tris);
    Box scene;
    scene.reset();
    for (uint32_t i = 0; i < n_tris; ++i) {
        Prim & p = b.prims[i];
        p.box.reset();
        p.box.grow(verts + 9 * (size_t)i);
        p.box.grow(verts + 9 * (size_t)i + 3);
        p.box.grow(verts + 9 * (size_t)i + 6);
        for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * p.box.lo[a] + 0.5f * p.box.hi[a];
        p.id = i;
        scene.grow(p.box);
    }
    for (int a = 0; a < 3; ++a) { out->scene_lo[a] = n_tris ? scene.lo[a] : 0.0f; out->scene_hi[a] = n_tris ? scene.hi[a] : 0.0f; }

    b.pool.resize(n_tris ? 2 * (size_t)n_tris : 1);
    b.next_node = 1;
    b.max_depth = 0;
    b.threads_free = (int)(threads > 1 ? threads - 1 : 0);
    if (n_tris) b.build(0, 0, n_tris, 0);

    out->tri_order.resize(n_tris);
    for (uint32_t i = 0; i < n_tris; ++i) out->tri_order[i] = b.prims[i].id;
    out->max_depth = b.max_depth.load() + 1;

    // --- emit the device layout.  Internal nodes are numbered in DFS preorder (node 0 = root, a left child
    // directly follows its parent).  Leaves live only as links.  The root is always an internal node: a scene
    // of <= leaf_max triangles becomes a root whose two links name the same leaf (testing a triangle twice
    // cannot change a closest hit: equal t and equal rank never replace), and an empty scene a root over one
    // all-zero dummy triangle, which the test always rejects (d = 0).
    auto leaf_link = [&](const TmpNode & n) -> int32_t { return ~(int32_t)((n.first << 2) | (n.count - 1)); };
    auto emit = [&](uint32_t slot, const Box & b0, int32_t l0, const Box & b1, int32_t l1) {
        float * n = &out->nodes[16 * (size_t)slot];
        n[0] = b0.lo[0]; n[1] = b0.hi[0]; n[2] = b0.lo[1]; n[3] = b0.hi[1];
        n[4] = b1.lo[0]; n[5] = b1.hi[0]; n[6] = b1.lo[1]; n[7] = b1.hi[1];
        n[8] = b0.lo[2]; n[9] = b0.hi[2]; n[10] = b1.lo[2]; n[11] = b1.hi[2];
        n[12] = int_bits(l0); n[13] = int_bits(l1); n[14] = 0.0f; n[15] = 0.0f;
    };

    if (n_tris == 0 || b.pool[0].left < 0) {
        out->nodes.assign(16, 0.0f);
        out->node_count = 1;
        Box zero = { { 0, 0, 0 }, { 0, 0, 0 } };
        if (n_tris == 0) emit(0, zero, ~0, zero, ~0);
        else emit(0, b.pool[0].box, leaf_link(b.pool[0]), b.pool[0].box, leaf_link(b.pool[0]));
        out->max_depth = 1;
        return;
    }

    // count internal nodes, assign preorder slots
    std::vector<int32_t> slot_of(b.next_node.load(), -1);
    std::vector<uint32_t> stack;
    uint32_t n_internal = 0;
    stack.push_back(0);
    while (!stack.empty()) {
        uint32_t t = stack.back();
        stack.pop_back();
        const TmpNode & n = b.pool[t];
        if (n.left < 0) continue;
        slot_of[t] = (int32_t)n_internal++;
        stack.push_back((uint32_t)n.right);
        stack.push_back((uint32_t)n.left);
    }
    out->node_count = n_internal;
    out->nodes.assign(16 * (size_t)n_internal, 0.0f);
    for (uint32_t t = 0; t < slot_of.size(); ++t) {
        if (slot_of[t] < 0) continue;
        const TmpNode & n = b.pool[t];
        const TmpNode & l = b.pool[n.left];
        const TmpNode & r = b.pool[n.right];
        int32_t ll = l.left < 0 ? leaf_link(l) : slot_of[n.left];
        int32_t rl = r.left < 0 ? leaf_link(r) : slot_of[n.right];
        emit((uint32_t)slot_of[t], l.box, ll, r.box, rl);
    }
}


// ---------------------------------------------------------------------------------------------------------
// 4-wide, quantised
// ---------------------------------------------------------------------------------------------------------
namespace {

struct Wide {
    uint32_t tmp[4];        // TmpNode indices of the children
    uint32_t n;
};

inline uint32_t float_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

}  // namespace

namespace {

// Shared back end: TmpNode tree (b.pool, root 0; leaves carry [first, first + count) of b.prims) -> 4-wide quantised nodes.
void finish_bvh4q_impl(Builder & b, uint32_t n_tris, Bvh4Result * out);
void finish_bvh4q(Builder & b, uint32_t n_tris, Bvh4Result * out) {
    const auto t0 = std::chrono::steady_clock::now();
    finish_bvh4q_impl(b, n_tris, out);
    if (b.opt.debug) fprintf(stderr, "[prt] BVH back end (collapse to 4-wide, quantise, reorder): %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}
void finish_bvh4q_impl(Builder & b, uint32_t n_tris, Bvh4Result * out) {
    out->tri_order.resize(n_tris);
    for (uint32_t i = 0; i < n_tris; ++i) out->tri_order[i] = b.prims[i].id;

    auto is_leaf = [&](uint32_t t) { return b.pool[t].left < 0; };
    auto leaf_link = [&](uint32_t t) -> int32_t { return ~(int32_t)((b.pool[t].first << 2) | (b.pool[t].count - 1)); };

    // ---- collapse: each wide node adopts up to 4 binary-tree descendants.
    // Every visit of a wide node costs the same (all four boxes are tested), and under the surface-area heuristic a
    // node is visited in proportion to the area of its box, so the best collapse of a given binary tree is the one
    // with the smallest total area of wide nodes (the leaves stay what they are).  That optimum is a small dynamic
    // programme over the binary tree (Ylitie, Karras, Laine 2017, section 3.1, here for width 4 and fixed leaves):
    //   F(m, k) = least area needed below binary node m if it may occupy up to k child slots of its parent
    //   F(m, 1) = area(m) + min over i of F(left, i) + F(right, 4 - i)          (m becomes a wide node; 0 for a leaf)
    //   F(m, k) = min(F(m, k - 1), min over i of F(left, i) + F(right, k - i))  (m is dissolved into its parent)
    // Measured on MI355X: 18-20 % fewer wide nodes, but only 1-2 % fewer node visits per frame and 2.5 % MORE triangle
    // tests on C4 (the rays of a terrain seen from above are far from the uniform distribution the heuristic assumes):
    // frame time unchanged within noise.  So the default stays the first version - open the child with the largest box
    // until four are there - and the BVH_COLLAPSE=dp option selects this one.
    struct Item { uint32_t tmp; uint32_t slot; uint32_t depth; };
    std::vector<Wide> wide;
    std::vector<Item> work;
    const bool greedy = b.opt.collapse != 1;                 // 4-wide default: greedy
    const uint32_t n_tmp = b.next_node.load();
    struct Dp { float f[3]; uint8_t root_split, split[2]; };   // f[k-1] = F(m, k); split: i of the best distribution, 0 = "use k - 1"
    std::vector<Dp> dp;
    if (!greedy) {
        dp.resize(n_tmp);
        for (uint32_t t = n_tmp; t-- > 0;) {                  // children are allocated after their parent: bottom-up
            Dp & d = dp[t];
            if (is_leaf(t)) { d.f[0] = d.f[1] = d.f[2] = 0.0f; d.root_split = 0; d.split[0] = d.split[1] = 0; continue; }
            const Dp & l = dp[(uint32_t)b.pool[t].left], & r = dp[(uint32_t)b.pool[t].right];
            float best = FLT_MAX;
            for (int i = 1; i <= 3; ++i) {
                const float c = l.f[i - 1] + r.f[3 - i];
                if (c < best) { best = c; d.root_split = (uint8_t)i; }
            }
            d.f[0] = b.pool[t].box.half_area() + best;
            for (int k = 2; k <= 3; ++k) {
                float bk = d.f[k - 2];
                uint8_t sk = 0;
                for (int i = 1; i < k; ++i) {
                    const float c = l.f[i - 1] + r.f[k - i - 1];
                    if (c < bk) { bk = c; sk = (uint8_t)i; }
                }
                d.f[k - 1] = bk;
                d.split[k - 2] = sk;
            }
        }
    }
    auto make_wide = [&](uint32_t root_tmp) -> Wide {
        Wide w;
        w.n = 0;
        if (is_leaf(root_tmp)) { w.tmp[w.n++] = root_tmp; return w; }
        if (!greedy) {
            // unfold the stored decisions: (node, slots it may occupy)
            struct Todo { uint32_t t; int k; };
            Todo stack[8];
            int sp = 0;
            const int i0 = dp[root_tmp].root_split;
            stack[sp++] = Todo{ (uint32_t)b.pool[root_tmp].right, 4 - i0 };
            stack[sp++] = Todo{ (uint32_t)b.pool[root_tmp].left, i0 };
            while (sp > 0) {
                Todo cur = stack[--sp];
                while (cur.k > 1 && !is_leaf(cur.t) && dp[cur.t].split[cur.k - 2] == 0) cur.k--;      // "use k - 1 slots"
                if (cur.k == 1 || is_leaf(cur.t)) { w.tmp[w.n++] = cur.t; continue; }
                const int i = dp[cur.t].split[cur.k - 2];
                stack[sp++] = Todo{ (uint32_t)b.pool[cur.t].right, cur.k - i };
                stack[sp++] = Todo{ (uint32_t)b.pool[cur.t].left, i };
            }
            // largest box first: the slot order is the visiting order of equal keys
            for (uint32_t i = 1; i < w.n; ++i)
                for (uint32_t j = i; j > 0 && b.pool[w.tmp[j]].box.half_area() > b.pool[w.tmp[j - 1]].box.half_area(); --j)
                    std::swap(w.tmp[j], w.tmp[j - 1]);
            return w;
        }
        w.tmp[w.n++] = (uint32_t)b.pool[root_tmp].left;
        w.tmp[w.n++] = (uint32_t)b.pool[root_tmp].right;
        while (w.n < 4) {
            int best = -1;
            float best_area = -1.0f;
            for (uint32_t k = 0; k < w.n; ++k) {
                if (is_leaf(w.tmp[k])) continue;
                float a = b.pool[w.tmp[k]].box.half_area();
                if (a > best_area) { best_area = a; best = (int)k; }
            }
            if (best < 0) break;
            uint32_t t = w.tmp[best];
            w.tmp[best] = (uint32_t)b.pool[t].left;
            w.tmp[w.n++] = (uint32_t)b.pool[t].right;
        }
        return w;
    };

    // BFS numbering: the top of the tree is contiguous (cache / LDS friendly), children of a node are adjacent
    std::vector<uint32_t> depth_of;
    work.push_back(Item{ 0u, 0u, 1u });
    wide.push_back(make_wide(0));
    depth_of.push_back(1);
    std::vector<int32_t> links;           // 4 per wide node
    links.assign(4, 0);
    uint32_t max_depth = 1;
    for (size_t head = 0; head < work.size(); ++head) {
        const Item it = work[head];
        const Wide w = wide[it.slot];
        for (uint32_t k = 0; k < 4; ++k) {
            int32_t link;
            if (k >= w.n) {
                link = ~(int32_t)(n_tris << 2);               // empty slot: a 1-triangle leaf naming the all-zero dummy record
                                                             // the uploader appends at slot n_tris (always rejected: d = 0)
            } else if (is_leaf(w.tmp[k])) {
                link = leaf_link(w.tmp[k]);
            } else {
                uint32_t slot = (uint32_t)wide.size();
                wide.push_back(make_wide(w.tmp[k]));
                links.resize(links.size() + 4, 0);
                work.push_back(Item{ w.tmp[k], slot, it.depth + 1 });
                if (it.depth + 1 > max_depth) max_depth = it.depth + 1;
                link = (int32_t)slot;
            }
            links[(size_t)it.slot * 4 + k] = link;
        }
    }

    // ---- quantise
    const uint32_t n_nodes = (uint32_t)wide.size();
    out->nodes.assign((size_t)n_nodes * 16, 0u);
    for (uint32_t s = 0; s < n_nodes; ++s) {
        const Wide & w = wide[s];
        Box u;
        u.reset();
        for (uint32_t k = 0; k < w.n; ++k) u.grow(b.pool[w.tmp[k]].box);
        uint32_t * d = &out->nodes[(size_t)s * 16];
        uint32_t ebyte[3];
        double scale[3];
        for (int a = 0; a < 3; ++a) {
            double ext = (double)u.hi[a] - (double)u.lo[a];
            int e = -100;
            if (ext > 0.0) {
                e = (int)std::ceil(std::log2(ext / 255.0));
                while (std::ldexp(255.0, e) < ext) ++e;
                if (e < -100) e = -100;
            }
            if (e > 100) e = 100;
            ebyte[a] = (uint32_t)(e + 127);
            scale[a] = std::ldexp(1.0, e);
            d[a] = float_bits(u.lo[a]);
        }
        // the grid steps 2^e as ready-made floats (the exponent byte in place): the traversal multiplies, it does not decode
        d[3] = ebyte[0] << 23;
        d[14] = ebyte[1] << 23;
        d[15] = ebyte[2] << 23;
        for (uint32_t k = 0; k < 4; ++k) {
            uint32_t qlo[3] = { 255, 255, 255 }, qhi[3] = { 0, 0, 0 };      // empty slot: inverted (and masked by its link)
            if (k < w.n) {
                const Box & cb = b.pool[w.tmp[k]].box;
                for (int a = 0; a < 3; ++a) {
                    double lo = std::floor(((double)cb.lo[a] - (double)u.lo[a]) / scale[a]);
                    double hi = std::ceil(((double)cb.hi[a] - (double)u.lo[a]) / scale[a]);
                    if (lo < 0.0) lo = 0.0;
                    if (lo > 255.0) lo = 255.0;
                    if (hi < 0.0) hi = 0.0;
                    if (hi > 255.0) hi = 255.0;
                    qlo[a] = (uint32_t)lo;
                    qhi[a] = (uint32_t)hi;
                }
            }
            for (int a = 0; a < 3; ++a) {
                d[4 + a] |= qlo[a] << (8 * k);
                d[7 + a] |= qhi[a] << (8 * k);
            }
            d[10 + k] = (uint32_t)links[(size_t)s * 4 + k];
        }
    }
    out->node_count = n_nodes;
    out->max_depth = max_depth;
    out->stack_bound = 3 * max_depth + 2;
}

// ---------------------------------------------------------------------------------------------------------
// 8-wide, quantised, octant-ordered (layout: bvh_build.h Bvh8Result)
// ---------------------------------------------------------------------------------------------------------
void finish_bvh8q_impl(Builder & b, uint32_t n_tris, Bvh8Result * out);
void finish_bvh8q(Builder & b, uint32_t n_tris, Bvh8Result * out) {
    const auto t0 = std::chrono::steady_clock::now();
    finish_bvh8q_impl(b, n_tris, out);
    if (b.opt.debug) fprintf(stderr, "[prt] BVH back end (collapse to 8-wide, slot assignment, quantise, reorder): %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}
void finish_bvh8q_impl(Builder & b, uint32_t n_tris, Bvh8Result * out) {
    auto is_leaf = [&](uint32_t t) { return b.pool[t].left < 0; };
    enum : uint32_t { EMPTY = 0xFFFFFFFFu };
    struct Wide8 { uint32_t slot[8]; uint32_t axis = 0; };  // TmpNode index per slot, EMPTY for none; (experiment) the ordering axis

    // ---- collapse (below), then give every child a slot: slot s stands for the
    // octant direction ((s & 1) ? +x : -x, (s & 2) ? +y : -y, (s & 4) ? +z : -z) as seen from the node's centre, and a ray
    // whose direction signs are o = (dx < 0) | (dy < 0) << 1 | (dz < 0) << 2 visits the hit slots in ascending order of
    // (s XOR o): the child on the side the ray comes from first, on every axis.  Children are matched to slots greedily by
    // how far their centres lie in the slot's direction (Ylitie et al. 2017, section 3.2, solve it with an auction; the
    // greedy matching is within a few per cent of it and this runs 150,000 times per upload).
    // The collapse itself is the small dynamic programme of Ylitie et al. (section 3.1) over the binary tree, leaves fixed:
    //   F(m, k) = least total area of wide nodes below binary node m if m may occupy up to k slots of its parent
    //   F(m, 1) = area(m) + min over i of F(left, i) + F(right, 8 - i)          (m becomes a wide node; 0 for a leaf)
    //   F(m, k) = min(F(m, k - 1), min over i of F(left, i) + F(right, k - i))  (m is dissolved into its parent)
    // Every visit of a wide node costs the same (all eight boxes are tested) and a node is visited in proportion to its
    // area.  The greedy rule of the 4-wide back end (open the largest child until the node is full) leaves the 8-wide
    // tree of the 1M-triangle terrain at 3.6 children per node; the programme fills it (tools/bvh_price.cpp).
    // The option BVH_COLLAPSE=greedy selects the greedy rule.
    const bool greedy = b.opt.collapse == 0;                 // 8-wide default: the dynamic programme
    const int W = b.opt.width >= 2 && b.opt.width <= 8 ? (int)b.opt.width : 8;      // children per node (experiments: 6)
    const uint32_t n_tmp = b.next_node.load();
    struct Dp8 { float f[7]; uint8_t root_split, split[6]; };   // f[k-1] = F(m, k); split[k-2]: i of the best distribution, 0 = "use k - 1"
    std::vector<Dp8> dp;
    if (!greedy) {
        dp.resize(n_tmp);
        for (uint32_t t = n_tmp; t-- > 0;) {                  // children are allocated after their parent: bottom-up
            Dp8 & d = dp[t];
            if (is_leaf(t)) { memset(&d, 0, sizeof(d)); continue; }
            const Dp8 & l = dp[(uint32_t)b.pool[t].left], & r = dp[(uint32_t)b.pool[t].right];
            float best = FLT_MAX;
            d.root_split = 1;
            for (int i = 1; i <= W - 1; ++i) {
                const float c = l.f[i - 1] + r.f[W - 1 - i];
                if (c < best) { best = c; d.root_split = (uint8_t)i; }
            }
            d.f[0] = b.pool[t].box.half_area() + best;
            for (int k = 2; k <= W - 1; ++k) {
                float bk = d.f[k - 2];
                uint8_t sk = 0;
                for (int i = 1; i < k; ++i) {
                    const float c = l.f[i - 1] + r.f[k - i - 1];
                    if (c < bk) { bk = c; sk = (uint8_t)i; }
                }
                d.f[k - 1] = bk;
                d.split[k - 2] = sk;
            }
        }
    }
    auto make_wide = [&](uint32_t root_tmp) -> Wide8 {
        uint32_t kids[8];
        uint32_t n = 0;
        if (is_leaf(root_tmp)) {
            kids[n++] = root_tmp;
        } else if (!greedy) {
            // unfold the stored decisions: (node, slots it may occupy)
            struct Todo { uint32_t t; int k; };
            Todo stack[16];
            int sp = 0;
            const int i0 = dp[root_tmp].root_split;
            stack[sp++] = Todo{ (uint32_t)b.pool[root_tmp].right, W - i0 };
            stack[sp++] = Todo{ (uint32_t)b.pool[root_tmp].left, i0 };
            while (sp > 0) {
                Todo cur = stack[--sp];
                while (cur.k > 1 && !is_leaf(cur.t) && dp[cur.t].split[cur.k - 2] == 0) cur.k--;      // "use k - 1 slots"
                if (cur.k == 1 || is_leaf(cur.t)) { kids[n++] = cur.t; continue; }
                const int i = dp[cur.t].split[cur.k - 2];
                stack[sp++] = Todo{ (uint32_t)b.pool[cur.t].right, cur.k - i };
                stack[sp++] = Todo{ (uint32_t)b.pool[cur.t].left, i };
            }
        } else {
            kids[n++] = (uint32_t)b.pool[root_tmp].left;
            kids[n++] = (uint32_t)b.pool[root_tmp].right;
            while (n < (uint32_t)W) {
                int best = -1;
                float best_area = -1.0f;
                for (uint32_t k = 0; k < n; ++k) {
                    if (is_leaf(kids[k])) continue;
                    const float a = b.pool[kids[k]].box.half_area();
                    if (a > best_area) { best_area = a; best = (int)k; }
                }
                if (best < 0) break;
                const uint32_t t = kids[best];
                kids[best] = (uint32_t)b.pool[t].left;
                kids[n++] = (uint32_t)b.pool[t].right;
            }
        }
        Box u;
        u.reset();
        for (uint32_t k = 0; k < n; ++k) u.grow(b.pool[kids[k]].box);
        float cost[8][8];
        for (uint32_t k = 0; k < n; ++k) {
            const Box & cb = b.pool[kids[k]].box;
            float rel[3];
            for (int a = 0; a < 3; ++a) rel[a] = (0.5f * cb.lo[a] + 0.5f * cb.hi[a]) - (0.5f * u.lo[a] + 0.5f * u.hi[a]);
            for (uint32_t s = 0; s < 8; ++s)
                cost[k][s] = ((s & 1u) ? rel[0] : -rel[0]) + ((s & 2u) ? rel[1] : -rel[1]) + ((s & 4u) ? rel[2] : -rel[2]);
        }
        Wide8 w;
        for (uint32_t s = 0; s < 8; ++s) w.slot[s] = EMPTY;
        if (b.opt.slot_order == 1) {
            // experiment: children in ascending order of their centres along the axis on which the centres spread most; a ray
            // takes the hit slots in ascending or descending order by the sign of its direction on that axis (axis in w.axis)
            float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
            float c[8][3];
            for (uint32_t k = 0; k < n; ++k)
                for (int a = 0; a < 3; ++a) {
                    c[k][a] = 0.5f * b.pool[kids[k]].box.lo[a] + 0.5f * b.pool[kids[k]].box.hi[a];
                    lo[a] = std::min(lo[a], c[k][a]); hi[a] = std::max(hi[a], c[k][a]);
                }
            int axis = 0;
            for (int a = 1; a < 3; ++a) if (hi[a] - lo[a] > hi[axis] - lo[axis]) axis = a;
            if (b.opt.axis_rule == 1) {
                // experiment: the longest axis of the node's box
                Box u2; u2.reset();
                for (uint32_t k = 0; k < n; ++k) u2.grow(b.pool[kids[k]].box);
                axis = 0;
                for (int a = 1; a < 3; ++a) if (u2.hi[a] - u2.lo[a] > u2.hi[axis] - u2.lo[axis]) axis = a;
            } else if (b.opt.axis_rule == 2) {
                // experiment: the axis along which the children's boxes, sorted by centre, overlap least (sum over all pairs of
                // the overlap of their intervals, relative to the node's extent on that axis)
                float best = FLT_MAX;
                for (int a = 0; a < 3; ++a) {
                    float ext = 0.0f, lo_a = FLT_MAX, hi_a = -FLT_MAX;
                    for (uint32_t k = 0; k < n; ++k) { lo_a = std::min(lo_a, b.pool[kids[k]].box.lo[a]); hi_a = std::max(hi_a, b.pool[kids[k]].box.hi[a]); }
                    ext = hi_a - lo_a;
                    if (!(ext > 0.0f)) continue;
                    float ov = 0.0f;
                    for (uint32_t i = 0; i < n; ++i)
                        for (uint32_t j = i + 1; j < n; ++j) {
                            const Box & bi = b.pool[kids[i]].box, & bj = b.pool[kids[j]].box;
                            const float o = std::min(bi.hi[a], bj.hi[a]) - std::max(bi.lo[a], bj.lo[a]);
                            if (o > 0.0f) ov += o / ext;
                        }
                    if (ov < best) { best = ov; axis = a; }
                }
            }
            uint32_t order[8];
            for (uint32_t k = 0; k < n; ++k) order[k] = k;
            std::sort(order, order + n, [&](uint32_t x, uint32_t y) { return c[x][axis] < c[y][axis]; });
            for (uint32_t k = 0; k < n; ++k) w.slot[k] = kids[order[k]];
            w.axis = (uint32_t)axis;
            return w;
        }
        bool placed[8] = { false, false, false, false, false, false, false, false };
        for (uint32_t round = 0; round < n; ++round) {
            int bk = -1, bs = -1;
            float bc = -FLT_MAX;
            for (uint32_t k = 0; k < n; ++k) {
                if (placed[k]) continue;
                for (uint32_t s = 0; s < 8; ++s) {
                    if (w.slot[s] != EMPTY) continue;
                    if (cost[k][s] > bc) { bc = cost[k][s]; bk = (int)k; bs = (int)s; }
                }
            }
            if (bk < 0) {                                   // NaN / inf boxes: any free slot will do
                for (uint32_t k = 0; k < n && bk < 0; ++k) if (!placed[k]) bk = (int)k;
                for (uint32_t s = 0; s < 8 && bs < 0; ++s) if (w.slot[s] == EMPTY) bs = (int)s;
            }
            placed[bk] = true;
            w.slot[bs] = kids[bk];
        }
        return w;
    };

    // ---- number the nodes breadth-first - the internal children of a node get consecutive indices in slot order - and lay
    // the triangles out so that the leaves of a node are consecutive in slot order too: a child is then addressed by the
    // node's base index plus the number of like children in lower slots, and the node needs no links.
    struct Item { uint32_t tmp, depth; };
    std::vector<Wide8> wide;
    std::vector<Item> work;
    std::vector<uint32_t> child_base, tri_base;
    out->tri_order.clear();
    out->tri_order.reserve(n_tris);
    work.push_back(Item{ 0u, 1u });
    uint32_t max_depth = 1;
    for (size_t head = 0; head < work.size(); ++head) {
        const Item it = work[head];
        const Wide8 w = make_wide(it.tmp);
        wide.push_back(w);
        child_base.push_back((uint32_t)work.size());
        tri_base.push_back((uint32_t)out->tri_order.size());
        for (uint32_t s = 0; s < 8; ++s) {
            const uint32_t t = w.slot[s];
            if (t == EMPTY) continue;
            if (is_leaf(t)) {
                // (an empty scene's root leaf names the all-zero dummy record the uploader appends at slot n_tris = 0)
                for (uint32_t i = 0; i < b.pool[t].count && b.pool[t].first + i < n_tris; ++i)
                    out->tri_order.push_back(b.prims[b.pool[t].first + i].id);
            } else {
                work.push_back(Item{ t, it.depth + 1 });
                if (it.depth + 1 > max_depth) max_depth = it.depth + 1;
            }
        }
    }

    // ---- quantise
    const uint32_t n_nodes = (uint32_t)wide.size();
    out->nodes.assign((size_t)n_nodes * BVH8_NODE_DWORDS, 0u);
    for (uint32_t ni = 0; ni < n_nodes; ++ni) {
        const Wide8 & w = wide[ni];
        Box u;
        u.reset();
        for (uint32_t s = 0; s < 8; ++s) if (w.slot[s] != EMPTY) u.grow(b.pool[w.slot[s]].box);
        uint32_t * d = &out->nodes[(size_t)ni * BVH8_NODE_DWORDS];
        uint32_t ebyte[3];
        double scale[3];
        for (int a = 0; a < 3; ++a) {
            double ext = (double)u.hi[a] - (double)u.lo[a];
            int e = -100;
            if (ext > 0.0) {
                e = (int)std::ceil(std::log2(ext / 255.0));
                while (std::ldexp(255.0, e) < ext) ++e;
                if (e < -100) e = -100;
            }
            if (e > 100) e = 100;
            ebyte[a] = (uint32_t)(e + 127);
            scale[a] = std::ldexp(1.0, e);
            d[a] = float_bits(u.lo[a]);
        }
        uint32_t imask = 0, lmask = 0, c0 = 0, c1 = 0;
        for (uint32_t s = 0; s < 8; ++s) {
            uint32_t qlo[3] = { 255, 255, 255 }, qhi[3] = { 0, 0, 0 };      // empty slot: inverted, can never be hit
            const uint32_t t = w.slot[s];
            if (t != EMPTY) {
                const Box & cb = b.pool[t].box;
                for (int a = 0; a < 3; ++a) {
                    double lo = std::floor(((double)cb.lo[a] - (double)u.lo[a]) / scale[a]);
                    double hi = std::ceil(((double)cb.hi[a] - (double)u.lo[a]) / scale[a]);
                    if (lo < 0.0) lo = 0.0;
                    if (lo > 255.0) lo = 255.0;
                    if (hi < 0.0) hi = 0.0;
                    if (hi > 255.0) hi = 255.0;
                    qlo[a] = (uint32_t)lo;
                    qhi[a] = (uint32_t)hi;
                }
                if (is_leaf(t)) {
                    const uint32_t extra = b.pool[t].count - 1u;             // 0..3
                    lmask |= 1u << s;
                    c0 |= (extra & 1u) << s;
                    c1 |= (extra >> 1) << s;
                } else {
                    imask |= 1u << s;
                }
            }
            for (int a = 0; a < 3; ++a) {
                d[8 + 2 * a + (s >> 2)] |= qlo[a] << (8 * (s & 3u));
                d[14 + 2 * a + (s >> 2)] |= qhi[a] << (8 * (s & 3u));
            }
        }
        d[3] = ebyte[0] << 23 | imask | lmask << 8;
        d[4] = child_base[ni];
        d[5] = tri_base[ni];
        d[6] = ebyte[1] << 23 | c0 | c1 << 8;
        d[7] = ebyte[2] << 23 | w.axis;
    }
    out->node_count = n_nodes;
    out->max_depth = max_depth;
    out->stack_bound = max_depth + 2;       // one group of unvisited siblings per level, the bottom marker, one to spare
}

}  // namespace

namespace {

// Front end shared by the 4- and 8-wide builds: binned-SAH binary tree over the triangles into b.pool (root 0).
void build_sah_binary(Builder & b, const float * verts, uint32_t n_tris, uint32_t leaf_max, uint32_t threads, float trav_cost,
                      float * scene_lo, float * scene_hi) {
    if (leaf_max < 1) leaf_max = 1;
    if (leaf_max > 4) leaf_max = 4;
    b.leaf_max = leaf_max;
    b.trav_cost = trav_cost;
    b.prims.resize(n_tris);
    Box scene;
    scene.reset();
    for (uint32_t i = 0; i < n_tris; ++i) {
        Prim & p = b.prims[i];
        p.box.reset();
        p.box.grow(verts + 9 * (size_t)i);
        p.box.grow(verts + 9 * (size_t)i + 3);
        p.box.grow(verts + 9 * (size_t)i + 6);
        for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * p.box.lo[a] + 0.5f * p.box.hi[a];
        p.id = i;
        scene.grow(p.box);
    }
    for (int a = 0; a < 3; ++a) { scene_lo[a] = n_tris ? scene.lo[a] : 0.0f; scene_hi[a] = n_tris ? scene.hi[a] : 0.0f; }
    b.pool.resize(n_tris ? 2 * (size_t)n_tris : 1);
    b.next_node = 1;
    b.max_depth = 0;
    b.threads_free = (int)(threads > 1 ? threads - 1 : 0);
    const auto t_build = std::chrono::steady_clock::now();
    if (n_tris) {
        b.build(0, 0, n_tris, 0);
    } else {
        TmpNode & n = b.pool[0];
        n.box.reset();
        for (int a = 0; a < 3; ++a) n.box.lo[a] = n.box.hi[a] = 0.0f;
        n.left = n.right = -1;
        n.first = 0;
        n.count = 1;          // the all-zero dummy triangle the uploader always allocates
        n.depth = 0;
    }
    if (b.opt.debug) fprintf(stderr, "[prt] binned-SAH binary build, %u triangles, %u threads: %.1f ms\n", n_tris, threads, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count());
}

}  // namespace

void build_bvh4q(const float * verts, uint32_t n_tris, uint32_t leaf_max, uint32_t threads, Bvh4Result * out, float trav_cost, const BvhBuildOptions * opt) {
    *out = Bvh4Result();
    Builder b;
    b.configure(opt);
    build_sah_binary(b, verts, n_tris, leaf_max, threads, trav_cost, out->scene_lo, out->scene_hi);
    finish_bvh4q(b, n_tris, out);
}

void build_bvh8q(const float * verts, uint32_t n_tris, uint32_t leaf_max, uint32_t threads, Bvh8Result * out, float trav_cost, const BvhBuildOptions * opt) {
    *out = Bvh8Result();
    Builder b;
    b.configure(opt);
    build_sah_binary(b, verts, n_tris, leaf_max, threads, trav_cost, out->scene_lo, out->scene_hi);
    finish_bvh8q(b, n_tris, out);
}

// Back end for a tree built elsewhere (the GPU LBVH builder, bvh_lbvh.hip): a binary radix tree over the triangles in
// sorted order.  Internal node i has children left[i] / right[i] (>= 0: internal node, < 0: ~sorted position of a
// single triangle), covers sorted positions [first[i], last[i]] and has box node_box[6 i .. 6 i + 5] (lo xyz, hi xyz);
// leaf_box holds the triangles' own boxes in sorted order.  Subtrees of at most leaf_max triangles become leaves.
namespace {
void radix_tree_to_binary(Builder & b, uint32_t n_tris, uint32_t leaf_max, const int32_t * left, const int32_t * right,
                          const uint32_t * first, const uint32_t * last, const float * node_box, const float * leaf_box,
                          const uint32_t * sorted_ids, float * scene_lo, float * scene_hi) {
    if (leaf_max < 1) leaf_max = 1;
    if (leaf_max > 4) leaf_max = 4;
    b.leaf_max = leaf_max;
    b.prims.resize(n_tris);
    Box scene;
    scene.reset();
    for (uint32_t i = 0; i < n_tris; ++i) {
        Prim & p = b.prims[i];
        for (int a = 0; a < 3; ++a) { p.box.lo[a] = leaf_box[6 * (size_t)i + a]; p.box.hi[a] = leaf_box[6 * (size_t)i + 3 + a]; p.c[a] = 0.0f; }
        p.id = sorted_ids[i];
        scene.grow(p.box);
    }
    for (int a = 0; a < 3; ++a) { scene_lo[a] = n_tris ? scene.lo[a] : 0.0f; scene_hi[a] = n_tris ? scene.hi[a] : 0.0f; }
    b.pool.resize(n_tris ? 2 * (size_t)n_tris : 1);
    b.next_node = 1;
    b.max_depth = 0;
    b.threads_free = 0;
    auto make_leaf = [&](TmpNode & n, const Box & box, uint32_t f, uint32_t c, uint32_t depth) {
        n.box = box; n.left = n.right = -1; n.first = f; n.count = c; n.depth = depth;
    };
    if (n_tris == 0) {
        Box zero = { { 0, 0, 0 }, { 0, 0, 0 } };
        make_leaf(b.pool[0], zero, 0, 1, 0);            // the all-zero dummy triangle the uploader always allocates
    } else if (n_tris <= leaf_max || n_tris == 1) {
        make_leaf(b.pool[0], scene, 0, n_tris, 0);
    } else if (b.opt.lbvh_plain) {
        // the radix tree as it is (round 1): fastest back end, 1.6 x slower to traverse than the SAH tree
        struct Todo { uint32_t tmp; int32_t src; uint32_t depth; };
        std::vector<Todo> todo;
        todo.push_back(Todo{ 0u, 0, 0u });
        while (!todo.empty()) {
            const Todo t = todo.back();
            todo.pop_back();
            TmpNode & n = b.pool[t.tmp];
            if (t.src < 0) {                                      // a single triangle
                const uint32_t pos = (uint32_t)~t.src;
                make_leaf(n, b.prims[pos].box, pos, 1, t.depth);
                continue;
            }
            Box box;
            for (int a = 0; a < 3; ++a) { box.lo[a] = node_box[6 * (size_t)t.src + a]; box.hi[a] = node_box[6 * (size_t)t.src + 3 + a]; }
            const uint32_t cnt = last[t.src] - first[t.src] + 1u;
            if (cnt <= leaf_max) { make_leaf(n, box, first[t.src], cnt, t.depth); continue; }
            n.box = box;
            n.depth = t.depth;
            n.first = n.count = 0;
            const uint32_t l = b.alloc(), r = b.alloc();
            n.left = (int32_t)l;
            n.right = (int32_t)r;
            todo.push_back(Todo{ l, left[t.src], t.depth + 1 });
            todo.push_back(Todo{ r, right[t.src], t.depth + 1 });
        }
    } else {
        // Hybrid (default): the device's radix tree only says which triangles belong together - its subtrees of at most
        // `cluster_max` triangles are CLUSTERS, contiguous runs of the Morton order -; the tree itself is surface-area
        // heuristic throughout: binned SAH ACROSS the clusters (a few thousand boxes), and the ordinary SAH builder INSIDE
        // each cluster (independent ranges, built by a pool of threads).  What the Morton order costs is then only where the
        // cluster boundaries lie.
        uint32_t cluster_max = 64;
        if (b.opt.lbvh_cluster >= 0) cluster_max = (uint32_t)std::max(4ll, std::min(1ll << 20, b.opt.lbvh_cluster));
        for (uint32_t i = 0; i < n_tris; ++i)
            for (int a = 0; a < 3; ++a) b.prims[i].c[a] = 0.5f * b.prims[i].box.lo[a] + 0.5f * b.prims[i].box.hi[a];
        struct Cluster { Box box; float c[3]; uint32_t first, count; };
        std::vector<Cluster> clusters;
        {
            std::vector<int32_t> stack;
            stack.push_back(0);
            while (!stack.empty()) {
                const int32_t src = stack.back();
                stack.pop_back();
                Cluster c;
                if (src < 0) {
                    const uint32_t pos = (uint32_t)~src;
                    c.box = b.prims[pos].box; c.first = pos; c.count = 1;
                } else {
                    const uint32_t cnt = last[src] - first[src] + 1u;
                    if (cnt > cluster_max) { stack.push_back(right[src]); stack.push_back(left[src]); continue; }
                    for (int a = 0; a < 3; ++a) { c.box.lo[a] = node_box[6 * (size_t)src + a]; c.box.hi[a] = node_box[6 * (size_t)src + 3 + a]; }
                    c.first = first[src]; c.count = cnt;
                }
                for (int a = 0; a < 3; ++a) c.c[a] = 0.5f * c.box.lo[a] + 0.5f * c.box.hi[a];
                clusters.push_back(c);
            }
        }
        // top tree: binned SAH over the clusters, weighted by their triangle counts; a leaf is one cluster
        struct Job { uint32_t tmp, first, count, depth; };
        std::vector<Job> cluster_jobs;
        std::vector<uint32_t> order(clusters.size());
        for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
        struct Range { uint32_t tmp, lo, hi, depth; };
        std::vector<Range> ranges;
        ranges.push_back(Range{ 0u, 0u, (uint32_t)order.size(), 0u });
        const int TB = 32;
        while (!ranges.empty()) {
            const Range rg = ranges.back();
            ranges.pop_back();
            if (rg.hi - rg.lo == 1) {
                const Cluster & c = clusters[order[rg.lo]];
                cluster_jobs.push_back(Job{ rg.tmp, c.first, c.count, rg.depth });
                continue;
            }
            Box bounds, cb;
            bounds.reset(); cb.reset();
            for (uint32_t i = rg.lo; i < rg.hi; ++i) { bounds.grow(clusters[order[i]].box); cb.grow(clusters[order[i]].c); }
            int best_axis = -1, best_split = -1;
            float best_cost = FLT_MAX;
            for (int axis = 0; axis < 3; ++axis) {
                const float cmin = cb.lo[axis], cmax = cb.hi[axis];
                if (!(cmax > cmin)) continue;
                const float scale = (float)TB / (cmax - cmin);
                Box bin_box[TB]; float bin_w[TB];
                for (int k = 0; k < TB; ++k) { bin_box[k].reset(); bin_w[k] = 0.0f; }
                for (uint32_t i = rg.lo; i < rg.hi; ++i) {
                    const Cluster & c = clusters[order[i]];
                    int k = (int)((c.c[axis] - cmin) * scale);
                    k = k < 0 ? 0 : (k >= TB ? TB - 1 : k);
                    bin_box[k].grow(c.box); bin_w[k] += (float)c.count;
                }
                float right_area[TB], right_w[TB];
                Box acc; acc.reset(); float w = 0.0f;
                for (int k = TB - 1; k > 0; --k) { acc.grow(bin_box[k]); w += bin_w[k]; right_area[k] = acc.half_area(); right_w[k] = w; }
                acc.reset(); w = 0.0f;
                for (int k = 0; k < TB - 1; ++k) {
                    acc.grow(bin_box[k]); w += bin_w[k];
                    if (w == 0.0f || right_w[k + 1] == 0.0f) continue;
                    const float cost = acc.half_area() * w + right_area[k + 1] * right_w[k + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = k; }
                }
            }
            uint32_t mid = rg.lo;
            if (best_axis >= 0) {
                const float cmin = cb.lo[best_axis], scale = (float)TB / (cb.hi[best_axis] - cmin);
                uint32_t * bp = order.data() + rg.lo, * ep = order.data() + rg.hi;
                uint32_t * mp = std::partition(bp, ep, [&](uint32_t ci) {
                    int k = (int)((clusters[ci].c[best_axis] - cmin) * scale);
                    k = k < 0 ? 0 : (k >= TB ? TB - 1 : k);
                    return k <= best_split;
                });
                mid = rg.lo + (uint32_t)(mp - bp);
            }
            // coincident centres - or a run of lopsided splits that has used up the depth budget (the stack bound and the spill
            // areas are sized from the depth): halve the run
            if (mid == rg.lo || mid == rg.hi || rg.depth >= MAX_FORCED_DEPTH) mid = rg.lo + (rg.hi - rg.lo) / 2;
            TmpNode & n = b.pool[rg.tmp];
            n.box = bounds; n.depth = rg.depth; n.first = n.count = 0;
            const uint32_t l = b.alloc(), r = b.alloc();
            b.pool[rg.tmp].left = (int32_t)l;
            b.pool[rg.tmp].right = (int32_t)r;
            ranges.push_back(Range{ l, rg.lo, mid, rg.depth + 1 });
            ranges.push_back(Range{ r, mid, rg.hi, rg.depth + 1 });
        }
        const auto t_bottom = std::chrono::steady_clock::now();
        // bottom trees: the SAH builder on each cluster's range of the (Morton-ordered) triangle array, in parallel
        unsigned int hw = std::thread::hardware_concurrency();
        const unsigned int n_threads = std::max(1u, std::min(16u, hw ? hw : 1u));
        std::atomic<size_t> next_job(0);
        auto work = [&]() {
            for (;;) {
                const size_t k = next_job.fetch_add(1);
                if (k >= cluster_jobs.size()) break;
                const Job & j = cluster_jobs[k];
                b.build(j.tmp, j.first, j.count, j.depth);
            }
        };
        if (n_threads == 1 || cluster_jobs.size() < 64) {
            work();
        } else {
            std::vector<std::thread> pool;
            for (unsigned int t = 0; t < n_threads; ++t) pool.emplace_back(work);
            for (size_t t = 0; t < pool.size(); ++t) pool[t].join();
        }
        if (b.opt.debug) fprintf(stderr, "[prt] LBVH hybrid: %zu clusters (<= %u triangles); SAH inside them on %u threads: %.1f ms\n", clusters.size(), cluster_max, n_threads, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_bottom).count());
    }
}
}  // namespace

void build_bvh4q_from_radix_tree(uint32_t n_tris, uint32_t leaf_max, const int32_t * left, const int32_t * right,
                                 const uint32_t * first, const uint32_t * last, const float * node_box, const float * leaf_box,
                                 const uint32_t * sorted_ids, Bvh4Result * out, const BvhBuildOptions * opt) {
    *out = Bvh4Result();
    Builder b;
    b.configure(opt);
    radix_tree_to_binary(b, n_tris, leaf_max, left, right, first, last, node_box, leaf_box, sorted_ids, out->scene_lo, out->scene_hi);
    finish_bvh4q(b, n_tris, out);
}

void build_bvh8q_from_radix_tree(uint32_t n_tris, uint32_t leaf_max, const int32_t * left, const int32_t * right,
                                 const uint32_t * first, const uint32_t * last, const float * node_box, const float * leaf_box,
                                 const uint32_t * sorted_ids, Bvh8Result * out, const BvhBuildOptions * opt) {
    *out = Bvh8Result();
    Builder b;
    b.configure(opt);
    radix_tree_to_binary(b, n_tris, leaf_max, left, right, first, last, node_box, leaf_box, sorted_ids, out->scene_lo, out->scene_hi);
    finish_bvh8q(b, n_tris, out);
}

}  // namespace prt
