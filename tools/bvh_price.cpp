// bvh_price.cpp - prices a BVH layout on the CPU before it is built into the kernels: node visits, triangle tests and stack
// depth per ray for the 4-wide quantised BVH (sorted children, csrc/dev_trace.h) and the 8-wide octant-ordered one
// (csrc/dev_trace8.h), on the same rays.  Host only; links csrc/bvh_build.cpp.
//
//   g++ -O2 -std=c++17 -pthread -Ipar_raytracer_amd/csrc tools/bvh_price.cpp par_raytracer_amd/csrc/bvh_build.cpp -o /tmp/bvh_price
//   python3 tools/bvh_price_scene.py terrain_1m /tmp/terrain.bin && /tmp/bvh_price /tmp/terrain.bin
//
// Input file: u32 n_tris, 9 floats per triangle, then camera position (3 floats), facing (3), fov (1).
// Rays: primary rays of a 480 x 270 lattice of the 1920 x 1080 frame; from every hit a shadow ray towards the reference's
// default light (main.cpp:522-524) and one cosine-distributed bounce ray with its own shadow ray - the ray mix of depth 2.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "bvh_build.h"
#include "prt_options.h"

using namespace prt;

struct V3 { double x, y, z; };
static V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static V3 operator*(V3 a, double s) { return { a.x * s, a.y * s, a.z * s }; }
static double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static V3 norm(V3 a) { double l = std::sqrt(dot(a, a)); return l > 0 ? a * (1.0 / l) : a; }

struct Ray { V3 o, d; bool any; };
struct Stats { uint64_t rays = 0, nodes = 0, tris = 0, max_sp = 0, sum_sp = 0; };

static std::vector<float> g_verts;     // 9 per triangle, input order

// single-sided like the reference (raytracer.cpp:93); returns t or -1
static double tri_hit(const Ray & r, uint32_t input_tri) {
    const float * v = &g_verts[(size_t)input_tri * 9];
    V3 a{ v[0], v[1], v[2] }, b{ v[3], v[4], v[5] }, c{ v[6], v[7], v[8] };
    V3 ab = b - a, ac = c - a, n = cross(ab, ac);
    V3 qp = r.d * -1.0;
    double d = dot(qp, n);
    if (d <= 0) return -1;
    V3 ap = r.o - a;
    double t = dot(ap, n);
    if (t < 0) return -1;
    V3 e = cross(qp, ap);
    double vv = dot(ac, e);
    if (vv < 0 || vv > d) return -1;
    double ww = -dot(ab, e);
    if (ww < 0 || vv + ww > d) return -1;
    return t / d;
}

static float bits_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

struct Hit { double t = 1e300; int tri = -1; };

// ---- 4-wide: csrc/dev_trace.h trav_node_step / trav_leaf
static Hit trace4(const Bvh4Result & bvh, const Ray & r, Stats & st) {
    Hit best;
    std::vector<int32_t> stack;
    int32_t node = 0;
    const double inv[3] = { 1.0 / (std::fabs(r.d.x) < 1e-30 ? 1e-30 : r.d.x), 1.0 / (std::fabs(r.d.y) < 1e-30 ? 1e-30 : r.d.y), 1.0 / (std::fabs(r.d.z) < 1e-30 ? 1e-30 : r.d.z) };
    const double o[3] = { r.o.x, r.o.y, r.o.z };
    st.rays++;
    for (;;) {
        if (node >= 0) {
            st.nodes++;
            const uint32_t * d = &bvh.nodes[(size_t)node * 16];
            const double org[3] = { bits_f(d[0]), bits_f(d[1]), bits_f(d[2]) };
            const double sc[3] = { bits_f(d[3]), bits_f(d[14]), bits_f(d[15]) };
            double key[4];
            int32_t link[4];
            int n = 0;
            for (int k = 0; k < 4; ++k) {
                double tmin = 0.0, tmax = best.t;
                for (int a = 0; a < 3; ++a) {
                    const double lo = org[a] + ((d[4 + a] >> (8 * k)) & 0xFF) * sc[a], hi = org[a] + ((d[7 + a] >> (8 * k)) & 0xFF) * sc[a];
                    if (((d[4 + a] >> (8 * k)) & 0xFF) > ((d[7 + a] >> (8 * k)) & 0xFF)) { tmin = 1; tmax = 0; break; }
                    double t0 = (lo - o[a]) * inv[a], t1 = (hi - o[a]) * inv[a];
                    if (t0 > t1) std::swap(t0, t1);
                    tmin = std::max(tmin, t0);
                    tmax = std::min(tmax, t1);
                }
                if (tmin <= tmax) { key[n] = tmin; link[n] = (int32_t)d[10 + k]; n++; }
            }
            for (int i = 1; i < n; ++i)
                for (int j = i; j > 0 && key[j] < key[j - 1]; --j) { std::swap(key[j], key[j - 1]); std::swap(link[j], link[j - 1]); }
            if (n) {
                for (int i = n - 1; i >= 1; --i) stack.push_back(link[i]);
                node = link[0];
                st.max_sp = std::max<uint64_t>(st.max_sp, stack.size() + 1);
            } else {
                if (stack.empty()) break;
                node = stack.back();
                stack.pop_back();
            }
        } else {
            const uint32_t leaf = (uint32_t)~node, first = leaf >> 2, cnt = (leaf & 3u) + 1u;
            for (uint32_t i = 0; i < cnt; ++i) {
                if (first + i >= bvh.tri_order.size()) continue;
                st.tris++;
                const double t = tri_hit(r, bvh.tri_order[first + i]);
                if (t >= 0 && t < best.t) { best.t = t; best.tri = (int)(first + i); if (r.any) return best; }
            }
            if (stack.empty()) break;
            node = stack.back();
            stack.pop_back();
        }
    }
    return best;
}

// ---- 8-wide: csrc/dev_trace8.h
static Hit trace8(const Bvh8Result & bvh, const Ray & r, Stats & st, bool ordered) {
    Hit best;
    struct Group { uint32_t base, imask, rest; };
    std::vector<Group> stack;
    int64_t cur = 0;
    const double inv[3] = { 1.0 / (std::fabs(r.d.x) < 1e-30 ? 1e-30 : r.d.x), 1.0 / (std::fabs(r.d.y) < 1e-30 ? 1e-30 : r.d.y), 1.0 / (std::fabs(r.d.z) < 1e-30 ? 1e-30 : r.d.z) };
    const double o[3] = { r.o.x, r.o.y, r.o.z };
    const uint32_t oct = ordered ? ((r.d.x < 0) | (r.d.y < 0) << 1 | (r.d.z < 0) << 2) : 0u;
    auto pick = [&](uint32_t rest) {                       // the slot of `rest` with the smallest (slot XOR octant)
        uint32_t bs = 0, bk = 99;
        for (uint32_t s = 0; s < 8; ++s) if ((rest >> s & 1u) && (s ^ oct) < bk) { bk = s ^ oct; bs = s; }
        return bs;
    };
    st.rays++;
    for (;;) {
        st.nodes++;
        const uint32_t * d = &bvh.nodes[(size_t)cur * BVH8_NODE_DWORDS];
        const double org[3] = { bits_f(d[0]), bits_f(d[1]), bits_f(d[2]) };
        const double sc[3] = { bits_f(d[3] & 0x7F800000u), bits_f(d[6] & 0x7F800000u), bits_f(d[7] & 0x7F800000u) };
        const uint32_t imask = d[3] & 0xFF, lmask = d[3] >> 8 & 0xFF, c0 = d[6] & 0xFF, c1 = d[6] >> 8 & 0xFF;
        uint32_t m = 0;
        for (uint32_t s = 0; s < 8; ++s) {
            double tmin = 0.0, tmax = best.t;
            bool empty = false;
            for (int a = 0; a < 3; ++a) {
                const uint32_t qlo = d[8 + 2 * a + (s >> 2)] >> (8 * (s & 3)) & 0xFF, qhi = d[14 + 2 * a + (s >> 2)] >> (8 * (s & 3)) & 0xFF;
                if (qlo > qhi) { empty = true; break; }
                double t0 = (org[a] + qlo * sc[a] - o[a]) * inv[a], t1 = (org[a] + qhi * sc[a] - o[a]) * inv[a];
                if (t0 > t1) std::swap(t0, t1);
                tmin = std::max(tmin, t0);
                tmax = std::min(tmax, t1);
            }
            if (!empty && tmin <= tmax) m |= 1u << s;
        }
        // leaves of this node first, in slot order
        uint32_t L = m & lmask;
        while (L) {
            const uint32_t s = (uint32_t)__builtin_ctz(L), below = (1u << s) - 1u;
            L &= L - 1;
            const uint32_t off = __builtin_popcount(lmask & below) + __builtin_popcount(c0 & below) + 2 * __builtin_popcount(c1 & below);
            const uint32_t cnt = 1 + (c0 >> s & 1) + 2 * (c1 >> s & 1);
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint32_t slot = d[5] + off + i;
                if (slot >= bvh.tri_order.size()) continue;
                st.tris++;
                const double t = tri_hit(r, bvh.tri_order[slot]);
                if (t >= 0 && t < best.t) { best.t = t; best.tri = (int)slot; if (r.any) return best; }
            }
        }
        Group g{ d[4], imask, m & imask };
        bool from_stack = false;
        if (!g.rest) {
            if (stack.empty()) break;
            g = stack.back();
            from_stack = true;
        }
        const uint32_t s = pick(g.rest);
        cur = g.base + __builtin_popcount(g.imask & ((1u << s) - 1u));
        g.rest &= ~(1u << s);
        if (from_stack) { if (g.rest) stack.back() = g; else stack.pop_back(); }
        else if (g.rest) stack.push_back(g);
        st.max_sp = std::max<uint64_t>(st.max_sp, stack.size() + 1);
    }
    return best;
}


// ---- 8-wide, hit slots (leaves and internal children alike) visited in one order: mode 0 = (slot XOR octant) ascending,
// mode 1 = entry distance ascending (what a per-step sort would give: the bound on what any ordering can do)
static Hit trace8u(const Bvh8Result & bvh, const Ray & r, Stats & st, int mode) {
    Hit best;
    struct Group { const uint32_t * d; uint32_t rest; double key[8]; };
    std::vector<Group> stack;
    const double inv[3] = { 1.0 / (std::fabs(r.d.x) < 1e-30 ? 1e-30 : r.d.x), 1.0 / (std::fabs(r.d.y) < 1e-30 ? 1e-30 : r.d.y), 1.0 / (std::fabs(r.d.z) < 1e-30 ? 1e-30 : r.d.z) };
    const double o[3] = { r.o.x, r.o.y, r.o.z };
    const uint32_t oct = (r.d.x < 0) | (r.d.y < 0) << 1 | (r.d.z < 0) << 2;
    st.rays++;
    auto visit = [&](int64_t cur) {
        st.nodes++;
        Group g;
        g.d = &bvh.nodes[(size_t)cur * BVH8_NODE_DWORDS];
        const uint32_t * d = g.d;
        const double org[3] = { bits_f(d[0]), bits_f(d[1]), bits_f(d[2]) };
        const double sc[3] = { bits_f(d[3] & 0x7F800000u), bits_f(d[6] & 0x7F800000u), bits_f(d[7] & 0x7F800000u) };
        g.rest = 0;
        for (uint32_t s = 0; s < 8; ++s) {
            double tmin = 0.0, tmax = best.t;
            bool empty = false;
            for (int a = 0; a < 3; ++a) {
                const uint32_t qlo = d[8 + 2 * a + (s >> 2)] >> (8 * (s & 3)) & 0xFF, qhi = d[14 + 2 * a + (s >> 2)] >> (8 * (s & 3)) & 0xFF;
                if (qlo > qhi) { empty = true; break; }
                double t0 = (org[a] + qlo * sc[a] - o[a]) * inv[a], t1 = (org[a] + qhi * sc[a] - o[a]) * inv[a];
                if (t0 > t1) std::swap(t0, t1);
                tmin = std::max(tmin, t0);
                tmax = std::min(tmax, t1);
            }
            g.key[s] = mode == 1 ? tmin : mode == 2 ? ((&r.d.x)[d[7] & 3u] >= 0 ? (double)s : (double)(7 - s)) : (double)(s ^ oct);
            if (!empty && tmin <= tmax) g.rest |= 1u << s;
        }
        return g;
    };
    Group g = visit(0);
    for (;;) {
        if (!g.rest) {
            if (stack.empty()) break;
            g = stack.back();
            stack.pop_back();
        }
        uint32_t s = 0;
        double bk = 1e300;
        for (uint32_t k = 0; k < 8; ++k) if ((g.rest >> k & 1u) && g.key[k] < bk) { bk = g.key[k]; s = k; }
        g.rest &= ~(1u << s);
        const uint32_t * d = g.d;
        const uint32_t imask = d[3] & 0xFF, lmask = d[3] >> 8 & 0xFF, c0 = d[6] & 0xFF, c1 = d[6] >> 8 & 0xFF, below = (1u << s) - 1u;
        if (imask >> s & 1u) {
            if (g.rest) stack.push_back(g);
            st.max_sp = std::max<uint64_t>(st.max_sp, stack.size() + 1);
            g = visit(d[4] + __builtin_popcount(imask & below));
        } else {
            const uint32_t off = __builtin_popcount(lmask & below) + __builtin_popcount(c0 & below) + 2 * __builtin_popcount(c1 & below);
            const uint32_t cnt = 1 + (c0 >> s & 1) + 2 * (c1 >> s & 1);
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint32_t slot = d[5] + off + i;
                if (slot >= bvh.tri_order.size()) continue;
                st.tris++;
                const double t = tri_hit(r, bvh.tri_order[slot]);
                if (t >= 0 && t < best.t) { best.t = t; best.tri = (int)slot; if (r.any) return best; }
            }
        }
    }
    return best;
}

static void report(const char * name, const Stats & st, uint32_t nodes, uint32_t node_bytes, uint32_t depth) {
    printf("%-34s %8u nodes (%5.1f MB) depth %2u | %6.2f node visits/ray  %5.2f tri tests/ray  deepest stack %llu | %.0f B fetched/ray in %.1f loads\n",
           name, nodes, nodes * (double)node_bytes / 1e6, depth, (double)st.nodes / st.rays, (double)st.tris / st.rays, (unsigned long long)st.max_sp,
           ((double)st.nodes * node_bytes + (double)st.tris * 48) / st.rays, ((double)st.nodes * (node_bytes / 16) + (double)st.tris * 3) / st.rays);
}

int main(int argc, char ** argv) {
    if (argc < 2) { fprintf(stderr, "usage: bvh_price scene.bin [lattice]\n"); return 2; }
    FILE * f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 1; }
    uint32_t n_tris = 0;
    if (fread(&n_tris, 4, 1, f) != 1) return 1;
    g_verts.resize((size_t)n_tris * 9);
    if (fread(g_verts.data(), 4, g_verts.size(), f) != g_verts.size()) return 1;
    float camf[7];
    if (fread(camf, 4, 7, f) != 7) return 1;
    fclose(f);
    const int lattice = argc > 2 ? atoi(argv[2]) : 4;

    Bvh4Result b4;
    Bvh8Result b8, b8a;
    const uint32_t leaf_max = getenv("PRICE_LEAF_MAX") ? (uint32_t)atoi(getenv("PRICE_LEAF_MAX")) : 4u;
    build_bvh4q(g_verts.data(), n_tris, 4, 8, &b4);
    BvhBuildOptions oct_opt;
    oct_opt.slot_order = 0;
    build_bvh8q(g_verts.data(), n_tris, leaf_max, 8, &b8, 1.0f, &oct_opt);
    BvhBuildOptions axis_opt;
    axis_opt.slot_order = 1;
    if (getenv("PRICE_WIDTH")) axis_opt.width = atoi(getenv("PRICE_WIDTH"));
    if (getenv("PRICE_AXIS_RULE")) axis_opt.axis_rule = atoi(getenv("PRICE_AXIS_RULE"));
    build_bvh8q(g_verts.data(), n_tris, leaf_max, 8, &b8a, 1.0f, &axis_opt);

    // camera (main.cpp:145-177)
    const int W = 1920, H = 1080;
    const V3 pos{ camf[0], camf[1], camf[2] }, fwd = norm(V3{ camf[3], camf[4], camf[5] });
    const double tan_a2 = std::tan(camf[6] * 0.5 * M_PI / 180.0), aspect = (double)W / H;
    const V3 right = norm(cross(fwd, V3{ 0, 1, 0 })), up = norm(cross(right, fwd));
    const V3 light = norm(V3{ 1, -1.5, 0.25 }) * -1.0;

    Stats s4[3], s8[3], s8u[3], s8o[3], s8d[3], s8a[3];            // primary, shadow, bounce
    uint64_t mismatches = 0;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) / 9007199254740992.0; };
    auto trace_all = [&](const Ray & r, int cls) {
        const Hit h4 = trace4(b4, r, s4[cls]);
        const Hit h8 = trace8(b8, r, s8[cls], true);
        trace8(b8, r, s8u[cls], false);
        trace8u(b8, r, s8o[cls], 0);
        trace8u(b8, r, s8d[cls], 1);
        trace8u(b8a, r, s8a[cls], 2);
        if (r.any ? (h4.tri < 0) != (h8.tri < 0) : (h4.tri < 0) != (h8.tri < 0) || (h4.tri >= 0 && std::fabs(h4.t - h8.t) > 1e-9 * std::max(1.0, h4.t))) mismatches++;
        return h8;
    };
    const double bias = 1e-3;
    for (int y = lattice / 2; y < H; y += lattice)
        for (int x = lattice / 2; x < W; x += lattice) {
            const double nx = 2.0 * (x + 0.5) / W - 1.0, ny = 1.0 - 2.0 * (y + 0.5) / H;
            Ray r{ pos, norm(fwd + right * (tan_a2 * aspect * nx) + up * (tan_a2 * ny)), false };
            const Hit h = trace_all(r, 0);
            if (h.tri < 0) continue;
            const float * v = &g_verts[(size_t)b8.tri_order[h.tri] * 9];
            const V3 a{ v[0], v[1], v[2] }, b{ v[3], v[4], v[5] }, c{ v[6], v[7], v[8] };
            const V3 n = norm(cross(b - a, c - a));
            const V3 p = r.o + r.d * h.t + n * bias;
            trace_all(Ray{ p, light, true }, 1);
            // cosine-distributed bounce
            const double u1 = rnd(), u2 = rnd(), ct = std::sqrt(1 - u1), stn = std::sqrt(u1), ph = 2 * M_PI * u2;
            const V3 upv = std::fabs(n.z) < 0.9999 ? V3{ 0, 0, 1 } : V3{ 1, 0, 0 };
            const V3 T = norm(cross(upv, n)), B = norm(cross(n, T));
            Ray br{ p, norm(T * (std::cos(ph) * stn) + B * (std::sin(ph) * stn) + n * ct), false };
            const Hit bh = trace_all(br, 2);
            if (bh.tri < 0) continue;
            const float * v2 = &g_verts[(size_t)b8.tri_order[bh.tri] * 9];
            const V3 a2{ v2[0], v2[1], v2[2] }, b2{ v2[3], v2[4], v2[5] }, c2{ v2[6], v2[7], v2[8] };
            const V3 n2 = norm(cross(b2 - a2, c2 - a2));
            trace_all(Ray{ br.o + br.d * bh.t + n2 * bias, light, true }, 1);
        }
    const char * cls[3] = { "primary", "shadow (any hit)", "bounce" };
    for (int k = 0; k < 3; ++k) {
        printf("-- %s rays: %llu\n", cls[k], (unsigned long long)s4[k].rays);
        report("4-wide, sorted children (64 B)", s4[k], b4.node_count, 64, b4.max_depth);
        report("8-wide, octant order (80 B)", s8[k], b8.node_count, 80, b8.max_depth);
        report("8-wide, slot order (no octant)", s8u[k], b8.node_count, 80, b8.max_depth);
        report("8-wide, one octant order, leaves too", s8o[k], b8.node_count, 80, b8.max_depth);
        report("8-wide, sorted by entry distance", s8d[k], b8.node_count, 80, b8.max_depth);
        report("8-wide, one axis per node", s8a[k], b8a.node_count, 80, b8a.max_depth);
    }
    Stats t4, t8, t8u, t8o, t8d, t8a;
    for (int k = 0; k < 3; ++k) {
        t4.rays += s4[k].rays; t4.nodes += s4[k].nodes; t4.tris += s4[k].tris; t4.max_sp = std::max(t4.max_sp, s4[k].max_sp);
        t8.rays += s8[k].rays; t8.nodes += s8[k].nodes; t8.tris += s8[k].tris; t8.max_sp = std::max(t8.max_sp, s8[k].max_sp);
        t8u.rays += s8u[k].rays; t8u.nodes += s8u[k].nodes; t8u.tris += s8u[k].tris; t8u.max_sp = std::max(t8u.max_sp, s8u[k].max_sp);
        t8o.rays += s8o[k].rays; t8o.nodes += s8o[k].nodes; t8o.tris += s8o[k].tris; t8o.max_sp = std::max(t8o.max_sp, s8o[k].max_sp);
        t8a.rays += s8a[k].rays; t8a.nodes += s8a[k].nodes; t8a.tris += s8a[k].tris; t8a.max_sp = std::max(t8a.max_sp, s8a[k].max_sp);
        t8d.rays += s8d[k].rays; t8d.nodes += s8d[k].nodes; t8d.tris += s8d[k].tris; t8d.max_sp = std::max(t8d.max_sp, s8d[k].max_sp);
    }
    printf("-- all rays: %llu\n", (unsigned long long)t4.rays);
    report("4-wide, sorted children (64 B)", t4, b4.node_count, 64, b4.max_depth);
    report("8-wide, octant order (80 B)", t8, b8.node_count, 80, b8.max_depth);
    report("8-wide, slot order (no octant)", t8u, b8.node_count, 80, b8.max_depth);
    report("8-wide, one octant order, leaves too", t8o, b8.node_count, 80, b8.max_depth);
    report("8-wide, sorted by entry distance", t8d, b8.node_count, 80, b8.max_depth);
    report("8-wide, one axis per node", t8a, b8a.node_count, 80, b8a.max_depth);
    printf("hit / miss or distance mismatches between the two trees: %llu\n", (unsigned long long)mismatches);
    return mismatches ? 1 : 0;
}
