/* prt_key.h - the per-(pixel,sample) RNG key that makes pixel-exact parity definable.
 *
 * The reference carries ONE sequential RandomState per MPI rank across all pixels
 * (main.cpp:203, 230, 319), so its image depends on the rank count.  Every renderer in this
 * repository (the compiled reference harness under oracle/, the CPU restatement and the HIP
 * kernels) instead reseeds with Random_Seed(key(seed, pixel, sample)) (random.h:9-27) before each
 * sample.  This header is the single definition of key(); it is part of the parity contract.
 *
 * key = splitmix64-finaliser(seed + golden * (((pixel << 16) | sample) + 1)),  sample < 65536.
 */
#ifndef PRT_KEY_H_
#define PRT_KEY_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define PRT_KEY_FN static inline __host__ __device__
#else
#define PRT_KEY_FN static inline
#endif

PRT_KEY_FN uint64_t prt_sample_key(uint64_t seed, uint32_t pixel, uint32_t sample) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ULL * (((((uint64_t)pixel) << 16) | (uint64_t)(sample & 0xFFFFu)) + 1ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

#endif /* PRT_KEY_H_ */
