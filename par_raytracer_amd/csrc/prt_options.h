// prt_options.h - every tuning / test knob of the library in one struct, filled ONCE per context.
//
// prt_create() reads the environment (PRT_<NAME>) into the context's PrtOptions; prt_set_option() (include/prt.h) changes one
// entry later.  Nothing on the upload or render path calls getenv: a render reads ctx->opt, which only the context's own host
// thread writes.  The three switches that change what a render DOES (not how fast) - TRACE_DEAD_SHADOW_RAYS, BVH_BUILDER,
// POOL_EXACT - are here too, so that they are visible to a caller of the ABI and not only to whoever started the process.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <strings.h>

namespace prt {

struct BvhBuildOptions {
    long long sah_bins = -1;          // SAH bins per axis (default 16)
    long long sah_sweep = -1;         // nodes of at most this many triangles try every split position (default 0)
    long long collapse = -1;          // 0 = greedy ("open the largest child"), 1 = dynamic programme; -1 = the back end's default
    long long lbvh_plain = 0;         // GPU builder: the radix tree as it is (no SAH inside / across its clusters)
    long long lbvh_cluster = -1;      // GPU builder, hybrid: triangles per Morton cluster (default 64)
    long long slot_order = 1;         // 8-wide: 1 = children sorted along the node's ordering axis (what dev_trace8.h expects), 0 = one slot
                                      // per octant (Ylitie et al.; the traversal built with -DPRT_BVH8_OCTANT).  Set by the library, not a knob
    long long width = 8;              // 8-wide back end: children per node (experiment: 6)
    long long axis_rule = 0;          // ordering axis of a node: 0 = largest spread of the children's centres; experiments: 1 = longest box
                                      // axis, 2 = least overlap of the children's intervals
    long long debug = 0;              // print build timings
};

struct PrtOptions {
    // ---- what a render does
    long long trace_dead_shadow_rays = 0;   // 1: shadow rays whose radiance-if-unoccluded is exactly zero are traced, not only counted
    long long bvh_builder_lbvh = 0;         // BVH_BUILDER=lbvh: radix tree built on the GPU (bvh_lbvh.h) instead of host SAH
    long long pool_exact = 0;               // the pool pipeline's EXACT kernel renders everything (tests)
    long long tie_widen_max = 8;            // resolve_near_ties: widenings before it gives up (tests force 0)
    // ---- diagnostics
    long long debug_util = 0, debug_rounds = 0;
    // ---- tuning; -1 = automatic
    long long chunk_min = -1, trace_blocks_per_cu = -1, keep_min = -1, node_min = -1, node_frac = -1, chains = -1, shade_block = -1;
    long long pool_blocks_per_cu = -1, pool_cap = -1, pool_topup = -1, pool_max_samples = -1, pool_park_cap = -1;
    long long pool_shared = -1;             // pool pipeline: 1 = the four waves of a block share one pool, 0 = wave-private pools; -1 = the build's default
    long long pool_exchange = -1;           // block-shared pools: a wave down to this many rays at the end of a round hands them to the block's other waves (0: off)
    long long pool_guided = -1;             // towards the end a wave takes at most this many eighths of its share of what is left (0: off; default: 16 in adaptive mode)
    long long pool_guided_min = -1;         // ... but at least this many units
    long long pool_fair = -1;               // experiment: pool capacity and largest top-up = this many eighths of a wave's fair share of the samples
    long long work_scatter = 0;             // experiment: hand the 8-pixel chunks of the tiled region out scattered over the image (DevParams::work_scatter_n)
    long long work_reverse = 0;             // experiment: hand the call's pixels out last-to-first (dev_scene.h DevParams::work_reverse_n)
    long long pool_flow = -1;               // pool pipeline without rounds (kernels_flow.h: tracer waves + a shading wave per workgroup); 1 / 0, -1 = the build's default
    long long pool_shared_cap = -1;         // ... slots per WAVE of a shared pool before the x4 (experiments)
    long long pass_samples = -1, pass_mb = -1, stack_cap = -1, no_tiles = 0;
    long long reserve_cus = 0;              // creation only: compute units the context's streams leave free
    long long reserve_pattern = 2;          // creation only: which bits of the CU mask are cleared - 2 = every (n / k)-th (k = 8: one compute unit per XCD),
                                            // experiments: 0 = the last k, 1 = the first k, 3 = none (profiles/r03_cu_mask.txt)
    // ---- BVH build
    long long leaf_max = -1;
    double sah_trav_cost = 1.0;
    BvhBuildOptions bvh;
};

namespace detail {
struct OptEntry { const char * name; long long PrtOptions::* field; long long BvhBuildOptions::* bvh_field; };
inline const OptEntry * option_table(size_t * n) {
    static const OptEntry table[] = {
        { "TRACE_DEAD_SHADOW_RAYS", &PrtOptions::trace_dead_shadow_rays, nullptr }, { "POOL_EXACT", &PrtOptions::pool_exact, nullptr },
        { "TIE_WIDEN_MAX", &PrtOptions::tie_widen_max, nullptr },
        { "DEBUG_UTIL", &PrtOptions::debug_util, nullptr }, { "DEBUG_ROUNDS", &PrtOptions::debug_rounds, nullptr },
        { "CHUNK_MIN", &PrtOptions::chunk_min, nullptr }, { "TRACE_BLOCKS_PER_CU", &PrtOptions::trace_blocks_per_cu, nullptr },
        { "KEEP_MIN", &PrtOptions::keep_min, nullptr }, { "NODE_MIN", &PrtOptions::node_min, nullptr }, { "NODE_FRAC", &PrtOptions::node_frac, nullptr },
        { "CHAINS", &PrtOptions::chains, nullptr }, { "SHADE_BLOCK", &PrtOptions::shade_block, nullptr },
        { "POOL_BLOCKS_PER_CU", &PrtOptions::pool_blocks_per_cu, nullptr }, { "POOL_CAP", &PrtOptions::pool_cap, nullptr },
        { "POOL_TOPUP", &PrtOptions::pool_topup, nullptr }, { "POOL_MAX_SAMPLES", &PrtOptions::pool_max_samples, nullptr },
        { "POOL_PARK_CAP", &PrtOptions::pool_park_cap, nullptr }, { "POOL_SHARED", &PrtOptions::pool_shared, nullptr }, { "POOL_FAIR", &PrtOptions::pool_fair, nullptr }, { "POOL_EXCHANGE", &PrtOptions::pool_exchange, nullptr }, { "POOL_GUIDED", &PrtOptions::pool_guided, nullptr }, { "POOL_GUIDED_MIN", &PrtOptions::pool_guided_min, nullptr }, { "POOL_FLOW", &PrtOptions::pool_flow, nullptr }, { "WORK_REVERSE", &PrtOptions::work_reverse, nullptr }, { "WORK_SCATTER", &PrtOptions::work_scatter, nullptr },
        { "POOL_SHARED_CAP", &PrtOptions::pool_shared_cap, nullptr }, { "PASS_SAMPLES", &PrtOptions::pass_samples, nullptr },
        { "PASS_MB", &PrtOptions::pass_mb, nullptr }, { "STACK_CAP", &PrtOptions::stack_cap, nullptr }, { "NO_TILES", &PrtOptions::no_tiles, nullptr },
        { "RESERVE_CUS", &PrtOptions::reserve_cus, nullptr }, { "RESERVE_PATTERN", &PrtOptions::reserve_pattern, nullptr }, { "LEAF_MAX", &PrtOptions::leaf_max, nullptr },
        { "SAH_BINS", nullptr, &BvhBuildOptions::sah_bins }, { "SAH_SWEEP", nullptr, &BvhBuildOptions::sah_sweep },
        { "LBVH_PLAIN", nullptr, &BvhBuildOptions::lbvh_plain }, { "LBVH_CLUSTER", nullptr, &BvhBuildOptions::lbvh_cluster },
        { "BVH8_WIDTH", nullptr, &BvhBuildOptions::width }, { "BVH8_AXIS_RULE", nullptr, &BvhBuildOptions::axis_rule },
    };
    *n = sizeof(table) / sizeof(table[0]);
    return table;
}
}  // namespace detail

// Sets option `name` (with or without the PRT_ prefix, any case) from its textual value; value NULL restores the default.
// Flags that the environment sets by mere presence (PRT_POOL_EXACT=, PRT_NO_TILES=1 ...) count as 1 for any value but "0".
// Returns 0, or -1 for an unknown name / a value that does not parse.
inline int prt_option_set(PrtOptions & o, const char * name, const char * value) {
    if (!name) return -1;
    if (!strncasecmp(name, "PRT_", 4)) name += 4;
    const PrtOptions defaults;
    if (!strcasecmp(name, "BVH_BUILDER")) {
        if (!value || !strcasecmp(value, "sah") || !*value) { o.bvh_builder_lbvh = 0; return 0; }
        if (!strcasecmp(value, "lbvh")) { o.bvh_builder_lbvh = 1; return 0; }
        return -1;
    }
    if (!strcasecmp(name, "BVH_COLLAPSE")) {
        if (!value || !*value) { o.bvh.collapse = -1; return 0; }
        if (!strcasecmp(value, "greedy")) { o.bvh.collapse = 0; return 0; }
        if (!strcasecmp(value, "dp")) { o.bvh.collapse = 1; return 0; }
        return -1;
    }
    if (!strcasecmp(name, "SAH_TRAV_COST")) {
        if (!value) { o.sah_trav_cost = defaults.sah_trav_cost; return 0; }
        char * end = nullptr;
        const double v = strtod(value, &end);
        if (end == value) return -1;
        o.sah_trav_cost = v;
        return 0;
    }
    size_t n = 0;
    const detail::OptEntry * table = detail::option_table(&n);
    for (size_t i = 0; i < n; ++i) {
        if (strcasecmp(name, table[i].name)) continue;
        long long & dst = table[i].field ? o.*(table[i].field) : o.bvh.*(table[i].bvh_field);
        if (!value) { dst = table[i].field ? defaults.*(table[i].field) : defaults.bvh.*(table[i].bvh_field); return 0; }
        char * end = nullptr;
        const long long v = strtoll(value, &end, 10);
        if (end == value) {
            // presence flags: "PRT_POOL_EXACT=" or "=yes"
            const bool is_flag = table[i].field == &PrtOptions::trace_dead_shadow_rays || table[i].field == &PrtOptions::pool_exact ||
                                 table[i].field == &PrtOptions::debug_util || table[i].field == &PrtOptions::debug_rounds ||
                                 table[i].field == &PrtOptions::no_tiles || table[i].bvh_field == &BvhBuildOptions::lbvh_plain;
            if (!is_flag) return -1;
            dst = 1;
            return 0;
        }
        dst = v;
        return 0;
    }
    return -1;
}

// The environment, once: PRT_<NAME> for every entry of the table (called by prt_create).
inline void prt_options_from_env(PrtOptions & o) {
    size_t n = 0;
    const detail::OptEntry * table = detail::option_table(&n);
    char key[64];
    for (size_t i = 0; i < n; ++i) {
        snprintf(key, sizeof(key), "PRT_%s", table[i].name);
        if (const char * v = getenv(key)) prt_option_set(o, table[i].name, v);
    }
    if (const char * v = getenv("PRT_BVH_BUILDER")) prt_option_set(o, "BVH_BUILDER", v);
    if (const char * v = getenv("PRT_BVH_COLLAPSE")) prt_option_set(o, "BVH_COLLAPSE", v);
    if (const char * v = getenv("PRT_SAH_TRAV_COST")) prt_option_set(o, "SAH_TRAV_COST", v);
    o.bvh.debug = o.debug_util;
}

}  // namespace prt
