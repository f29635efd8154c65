// dev_rng.h - the reference PRNG (random.h:4-61) for the device, bit exact, in 4 VGPRs.
//
// RandomState is 16 x u64 + an index: 136 bytes per lane with a data-dependent index, which on a GPU
// means LDS or scratch.  But the generator is a lagged recurrence.  With p starting at 0 (Random_Seed):
//     draw k reads  state[k & 15]       (as s0)  = the value draw k-1 wrote (seed word 0 for k = 0)
//               and state[(k+1) & 15]   (as s1)  = seed word k+1 for k <= 14, seed word 0 for k = 15,
//                                                   the value draw k-16 wrote for k >= 16,
//     writes state[(k+1) & 15] and returns it times 1181783497276652981.
// and seed word i is the i-th step of the xorshift chain Random_Seed runs.  So while a sample makes at
// most 15 draws - true for the reference defaults: 2 jitter draws + 9 in the bounce tree at depth 2 -
// the whole state is (chain value, previous output): two u64.  The host computes the worst-case draw
// count of the configured bounce tree (prt_api.hip: max_rng_draws) and selects the RING variant when
// it can exceed 15 (deeper trees, translucent materials); that variant spills outputs to a per-sample
// 16-entry ring in global memory and reads it back 16 draws later.
#pragma once

#include "dev_math.h"

namespace prt {

typedef unsigned long long u64;
typedef unsigned int u32;

struct Rng {
    u64 chain;    // xorshift chain value after the last seed word generated so far
    u64 prev;     // last value written to the state array (seed word 0 before the first draw)
    u64 seed0;    // seed word 0, needed again by draw 15
    u32 k;        // draws made so far
};

PRT_HD u64 rng_chain_step(u64 x) {            // random.h:21-24: three RIGHT shifts (sic)
    x ^= x >> 12;
    x ^= x >> 25;
    x ^= x >> 27;
    return x;
}

PRT_HD void rng_seed(Rng & r, u64 seed) {     // random.h:9-27
    if (seed == 0) seed = 0x5555555555555555ULL;
    r.chain = rng_chain_step(seed);
    r.seed0 = r.chain * 2685821657736338717ULL;
    r.prev = r.seed0;
    r.k = 0;
}

// ring: this sample's 16 u64 slots, element i at ring[i * ring_stride].  NULL when !RING - and also (as a compile-time
// constant, k_pool<RINGMEM = false>) when the host knows that no sample of the render can make more than 15 draws: the
// ring is then never read, and leaving out its writes saves 8 scattered bytes per draw.
template <bool RING>
PRT_HD u64 rng_next(Rng & r, u64 * ring, size_t ring_stride) {   // random.h:29-42
    u64 s0 = r.prev;
    u64 s1;
    if (!RING || r.k < 15) {
        r.chain = rng_chain_step(r.chain);
        s1 = r.chain * 2685821657736338717ULL;                   // seed word k+1
    } else if (r.k == 15) {
        s1 = r.seed0;
    } else {
        s1 = ring[(size_t)((r.k + 1) & 15) * ring_stride];       // written by draw k-16
    }
    s1 ^= s1 << 31;
    s1 ^= s1 >> 11;
    s0 &= s0 >> 30;                                              // AND (sic)
    u64 out = s0 ^ s1;
    if (RING && ring) ring[(size_t)((r.k + 1) & 15) * ring_stride] = out;
    r.prev = out;
    r.k++;
    return out * 1181783497276652981ULL;
}

// random.h:49-56: (float)u64 / (float)0xFFFFFFFFFFFFFFFF, i.e. / 2^64 (exact scaling), then Clamp.
PRT_HD float rng_to_float01(u64 v) {
    float f = (float)v * 0x1p-64f;                              // exact
    return ref_min(ref_max(f, 0.0f), 1.0f);
}

template <bool RING>
PRT_HD float rng_float01(Rng & r, u64 * ring, size_t stride) { return rng_to_float01(rng_next<RING>(r, ring, stride)); }

template <bool RING>
PRT_HD float rng_float11(Rng & r, u64 * ring, size_t stride) {   // random.h:58-61
    return (rng_float01<RING>(r, ring, stride) * 2.0f) - 1.0f;
}

}  // namespace prt
