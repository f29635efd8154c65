// prt_random.h - host copy of the reference PRNG semantics (random.h:4-61), quirks included:
// the seeding chain uses three RIGHT shifts, Random_Next ANDs state_0 with itself shifted
// (`s0 &= s0 >> 30`), and float01 divides by (float)0xFFFFFFFFFFFFFFFF == 2^64 then clamps.
// Known-answer vectors: SURVEY.md §8c, tests/golden/kat.npz.
#pragma once

#include "prt_math.h"

struct RandomState {
    u64 _state[16];
    s32 _p;
};

inline void Random_Seed(RandomState * state, u64 seed) {
    if (seed == 0) seed = 0x5555555555555555ULL;
    state->_p = 0;
    u64 x = seed;
    for (u32 i = 0; i < 16; ++i) {
        x ^= x >> 12;
        x ^= x >> 25;
        x ^= x >> 27;
        state->_state[i] = x * 2685821657736338717ULL;
    }
}

inline u64 Random_Next(RandomState * state) {
    u64 s0 = state->_state[state->_p];
    state->_p = (state->_p + 1) & 15;
    u64 s1 = state->_state[state->_p];
    s1 ^= s1 << 31;
    s1 ^= s1 >> 11;
    s0 &= s0 >> 30;
    state->_state[state->_p] = s0 ^ s1;
    return state->_state[state->_p] * 1181783497276652981ULL;
}

inline float Random_NextFloat01(RandomState * state) {
    float f = (float)Random_Next(state) / 18446744073709551616.0f;
    return PrtClamp(f, 0.0f, 1.0f);
}

inline float Random_NextFloat11(RandomState * state) { return Random_NextFloat01(state) * 2.0f - 1.0f; }
