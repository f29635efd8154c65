// ta_coop.hip - does the texture addresser charge a divergent node fetch per LANE or per distinct cache line?
//
// Companion of ta_rate.hip (same random 64-byte nodes, same clocks per wave-step per CU).  k_pool fetches a lane's node with
// 4 x global_load_dwordx4: every instruction touches 64 different lines.  The alternative priced here: the four lanes of a
// quad fetch the four quarters of ONE node per instruction (16 lines per instruction, each read whole), four instructions
// cover the quad's four nodes, and the quarters reach their owner through LDS.
//   mode 0: 4 x dwordx4, lane-private node                         (today)
//   mode 1: 4 x dwordx4, quad-cooperative, results left in registers (the addresser's side of the alternative alone)
//   mode 2: 4 x global_load_lds_dwordx4 quad-cooperative + 4 x ds_read_b128 of the own node (the whole alternative)
//   mode 3: mode 0 with 38 of 64 lanes active                      (does the cost follow the number of active lanes?)
//   mode 4: mode 0 with 16 of 64 lanes active
//   mode 5: 4 x global_load_lds_dwordx4 of the lane's OWN node + 4 x ds_read_b128 (LDS-direct without the quad scheme)
//   mode 6: mode 5 with 38 of 64 lanes active
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int K>
__device__ __forceinline__ unsigned int quad_bcast(unsigned int v) {
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, K | K << 2 | K << 4 | K << 6, 0xF, 0xF, true);
}

template <int MODE>
__global__ __launch_bounds__(256, 5) void k(const uint4 * nodes, unsigned int n_nodes, unsigned int iters, float * out) {
    __shared__ uint4 s_stage[4][4 * 65];                      // per wave: 4 instructions x (64 lanes x 16 B), staggered by 16 B
    const unsigned int lane = threadIdx.x & 63u, q = lane & 3u;
    uint4 * const stage = &s_stage[threadIdx.x >> 6][0];
    unsigned int x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    const bool on = (MODE == 3 || MODE == 6) ? lane < 38u : MODE == 4 ? lane < 16u : true;
    for (unsigned int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        const unsigned int node = (x >> 8) % n_nodes;
        if (MODE == 0 || MODE == 3 || MODE == 4) {
            if (on) {
                const uint4 * np = nodes + 4 * (size_t)node;
                uint4 a = np[0], b = np[1], c = np[2], d = np[3];
                acc += __uint_as_float((a.x ^ b.y ^ c.z ^ d.w) & 0x3FFFFFFFu);
            }
        } else if (MODE == 1) {
            const uint4 a = nodes[4 * (size_t)quad_bcast<0>(node) + q];
            const uint4 b = nodes[4 * (size_t)quad_bcast<1>(node) + q];
            const uint4 c = nodes[4 * (size_t)quad_bcast<2>(node) + q];
            const uint4 d = nodes[4 * (size_t)quad_bcast<3>(node) + q];
            acc += __uint_as_float((a.x ^ b.y ^ c.z ^ d.w) & 0x3FFFFFFFu);
        } else if (MODE == 5 || MODE == 6) {
            if (on) {
                const uint4 * np = nodes + 4 * (size_t)node;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(np + 0), (__attribute__((address_space(3))) void *)(stage + 0 * 65), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(np + 1), (__attribute__((address_space(3))) void *)(stage + 1 * 65), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(np + 2), (__attribute__((address_space(3))) void *)(stage + 2 * 65), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(np + 3), (__attribute__((address_space(3))) void *)(stage + 3 * 65), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const uint4 a = stage[0 * 65 + lane], b = stage[1 * 65 + lane], c = stage[2 * 65 + lane], d = stage[3 * 65 + lane];
                acc += __uint_as_float((a.x ^ b.y ^ c.z ^ d.w) & 0x3FFFFFFFu);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else {
            const uint4 * p0 = nodes + 4 * (size_t)quad_bcast<0>(node) + q;
            const uint4 * p1 = nodes + 4 * (size_t)quad_bcast<1>(node) + q;
            const uint4 * p2 = nodes + 4 * (size_t)quad_bcast<2>(node) + q;
            const uint4 * p3 = nodes + 4 * (size_t)quad_bcast<3>(node) + q;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p0, (__attribute__((address_space(3))) void *)(stage + 0 * 65), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p1, (__attribute__((address_space(3))) void *)(stage + 1 * 65), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p2, (__attribute__((address_space(3))) void *)(stage + 2 * 65), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p3, (__attribute__((address_space(3))) void *)(stage + 3 * 65), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // instruction k put node(4j + k) at stage[k * 65 + 4j .. 4j + 3]; the own node is the one of instruction q
            const uint4 * own = stage + q * 65 + (lane & ~3u);
            const uint4 a = own[0], b = own[1], c = own[2], d = own[3];
            acc += __uint_as_float((a.x ^ b.y ^ c.z ^ d.w) & 0x3FFFFFFFu);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int MODE>
double run(const uint4 * d, unsigned int n_nodes, unsigned int iters, float * o, unsigned int grid) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(d, n_nodes, 8, o);
    (void)hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(d, n_nodes, iters, o);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    const unsigned int grid = (unsigned int)cus * 5u, iters = 4000;
    float * o;
    (void)hipMalloc(&o, (size_t)grid * 256 * 4);
    const unsigned int sizes[4] = { 128u, 448u, 4096u, 1u << 20 };     // 8 KB, 28 KB (inside L1), 256 KB (L2), 64 MB (beyond L2)
    for (int s = 0; s < 4; ++s) {
        const unsigned int n = sizes[s];
        std::vector<unsigned int> h((size_t)n * 16);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned int)(i * 2654435761u) >> 3;
        uint4 * d;
        (void)hipMalloc(&d, (size_t)n * 64);
        (void)hipMemcpy(d, h.data(), (size_t)n * 64, hipMemcpyHostToDevice);
        double ms[7] = { run<0>(d, n, iters, o, grid), run<1>(d, n, iters, o, grid), run<2>(d, n, iters, o, grid), run<3>(d, n, iters, o, grid), run<4>(d, n, iters, o, grid), run<5>(d, n, iters, o, grid), run<6>(d, n, iters, o, grid) };
        const double wave_steps_per_cu = (double)grid * 4 * iters / cus;
        printf("%u nodes (%.1f MB), %d CUs at %.2f GHz:", n, n * 64.0 / 1e6, cus, clk / 1e9);
        for (int m = 0; m < 7; ++m) printf("  mode %d: %.1f clk/step/CU (%.2f ms)", m, ms[m] * 1e-3 * clk / wave_steps_per_cu, ms[m]);
        printf("\n");
        (void)hipFree(d);
    }
    return 0;
}
