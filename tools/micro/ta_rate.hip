// ta_rate.hip - how many divergent vector-memory instructions per node step can a CU's texture addresser / L1 sustain?
//
// Every lane fetches its own random 64-byte node per step (as BVH traversal does) with one of several instruction mixes and
// does next to no arithmetic.  Reports clocks per wave-step per CU; the traversal kernel's vector-issue budget is one step
// per ~130 clocks per CU (525 clocks of VALU per step on each of the 4 SIMDs), so a mix well below that is not a bottleneck.
//   mode 0: 4 x global_load_dwordx4                       (the 64-byte node as k_pool fetches it today)
//   mode 1: 2 x dwordx4 + 6 x buffer_load_format_xyzw     (planes converted byte -> float by the TA: 8_8_8_8 USCALED)
//   mode 2: 1 x dwordx4
//   mode 3: 6 x buffer_load_format_xyzw
//   mode 4: 2 x dwordx4 + 3 x buffer_load_format_xyzw + 1 x dwordx3
//   mode 5: 1 x dwordx4 + 1 x dwordx2 + 6 x buffer_load_format_xyzw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float float4v __attribute__((ext_vector_type(4)));
typedef int int4v __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 5) void k(const uint4 * nodes, unsigned int n_nodes, unsigned int iters, unsigned int word3, float * out) {
    unsigned long long base = (unsigned long long)nodes;
    int4v rsrc;
    rsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned int)base);
    rsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned int)(base >> 32));
    rsrc.z = __builtin_amdgcn_readfirstlane((int)(n_nodes * 64u));
    rsrc.w = __builtin_amdgcn_readfirstlane((int)word3);
    unsigned int x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (unsigned int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        const unsigned int node = (x >> 8) % n_nodes;
        const uint4 * np = nodes + 4 * (size_t)node;
        const unsigned int off = node * 64u;
        if (MODE == 0) {
            uint4 a = np[0], b = np[1], c = np[2], d = np[3];
            acc += __uint_as_float((a.x ^ b.y ^ c.z ^ d.w) & 0x3FFFFFFFu);
        } else if (MODE == 2) {
            uint4 a = np[0];
            acc += __uint_as_float(a.x & 0x3FFFFFFFu);
        } else {
            float4v p0, p1, p2, p3, p4, p5;
            uint4 a = make_uint4(0, 0, 0, 0), d = make_uint4(0, 0, 0, 0);
            if (MODE == 1 || MODE == 4) { a = np[0]; d = np[3]; }
            if (MODE == 5) { a = np[0]; uint2 e = *reinterpret_cast<const uint2 *>(np + 3); d.x = e.x; d.y = e.y; }
            if (MODE == 4) {
                asm volatile("buffer_load_format_xyzw %0, %3, %4, 0 offen offset:16\n\t"
                             "buffer_load_format_xyzw %1, %3, %4, 0 offen offset:20\n\t"
                             "buffer_load_format_xyzw %2, %3, %4, 0 offen offset:24\n\t"
                             "s_waitcnt vmcnt(0)" : "=&v"(p0), "=&v"(p1), "=&v"(p2) : "v"(off), "s"(rsrc) : "memory");
                const uint3 far = *reinterpret_cast<const uint3 *>(reinterpret_cast<const unsigned int *>(np) + 7);
                p3 = p0; p4 = p1; p5 = p2;
                acc += __uint_as_float((far.x ^ far.y ^ far.z) & 0x3FFFFFFFu);
            } else {
                asm volatile("buffer_load_format_xyzw %0, %6, %7, 0 offen offset:16\n\t"
                             "buffer_load_format_xyzw %1, %6, %7, 0 offen offset:20\n\t"
                             "buffer_load_format_xyzw %2, %6, %7, 0 offen offset:24\n\t"
                             "buffer_load_format_xyzw %3, %6, %7, 0 offen offset:28\n\t"
                             "buffer_load_format_xyzw %4, %6, %7, 0 offen offset:32\n\t"
                             "buffer_load_format_xyzw %5, %6, %7, 0 offen offset:36\n\t"
                             "s_waitcnt vmcnt(0)" : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5) : "v"(off), "s"(rsrc) : "memory");
            }
            acc += (p0.x + p1.y) + (p2.z + p3.w) + (p4.x + p5.y) + __uint_as_float((a.x ^ d.y) & 0x3FFFFFFFu);
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int MODE>
double run(const uint4 * d, unsigned int n_nodes, unsigned int iters, unsigned int w3, float * o, unsigned int grid) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(d, n_nodes, 8, w3, o);
    (void)hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(d, n_nodes, iters, w3, o);
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
    const unsigned int w3 = 4u | 5u << 3 | 6u << 6 | 7u << 9 | 2u << 12 | 10u << 15;
    float * o;
    (void)hipMalloc(&o, (size_t)grid * 256 * 4);
    const unsigned int sizes[3] = { 128u, 448u, 4096u };     // 8 KB, 28 KB (inside the 32 KB L1), 256 KB (L2) of nodes
    for (int s = 0; s < 3; ++s) {
        const unsigned int n = sizes[s];
        std::vector<unsigned int> h((size_t)n * 16);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned int)(i * 2654435761u) >> 3;
        uint4 * d;
        (void)hipMalloc(&d, (size_t)n * 64);
        (void)hipMemcpy(d, h.data(), (size_t)n * 64, hipMemcpyHostToDevice);
        double ms[6] = { run<0>(d, n, iters, w3, o, grid), run<1>(d, n, iters, w3, o, grid), run<2>(d, n, iters, w3, o, grid),
                         run<3>(d, n, iters, w3, o, grid), run<4>(d, n, iters, w3, o, grid), run<5>(d, n, iters, w3, o, grid) };
        const double wave_steps_per_cu = (double)grid * 4 * iters / cus;
        printf("%u nodes (%.1f MB), %d CUs at %.2f GHz:", n, n * 64.0 / 1e6, cus, clk / 1e9);
        for (int m = 0; m < 6; ++m) printf("  mode %d: %.1f clk/step/CU (%.2f ms)", m, ms[m] * 1e-3 * clk / wave_steps_per_cu, ms[m]);
        printf("\n");
        (void)hipFree(d);
    }
    return 0;
}
