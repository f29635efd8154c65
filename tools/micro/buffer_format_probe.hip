// buffer_format_probe.hip - can the texture addresser convert the BVH's quantised plane bytes to floats by itself?
//
// buffer_load_format_xyzw with a buffer resource whose data format is 8_8_8_8 and number format USCALED returns the four
// bytes of a dword as four floats (0.0 .. 255.0) - the 24 / 48 v_cvt_f32_ubyteN of a node step would disappear.  This probe
// shows that the format works on gfx950 (it does: every lane prints its bytes as floats); tools/micro/ta_rate.hip prices what
// it would cost (one vector-memory instruction per four planes, where one dwordx4 load now brings sixteen): the texture
// addresser, not the vector ALU, would become the bottleneck.  profiles/r02_experiments.txt item 4.
// (Re-written in round 3: round 2 committed only the binary.)
//
//   hipcc --offload-arch=gfx950 -O2 -o buffer_format_probe buffer_format_probe.hip && ./buffer_format_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k_probe(const uint32_t * src, float * dst, int n) {
    const int lane = threadIdx.x;
    // buffer resource (V#): base address, stride 0, num_records = bytes, word 3 = dst_sel xyzw + the format.
    // gfx9-family word 3: DST_SEL_X..W at bits 0-11 (4 = X, 5 = Y, 6 = Z, 7 = W), NUM_FORMAT bits 12-14 (2 = USCALED),
    // DATA_FORMAT bits 15-18 (10 = 8_8_8_8).
    const uint64_t base = (uint64_t)src;
    v4i rsrc;
    rsrc.x = (int)(uint32_t)base;
    rsrc.y = (int)(uint32_t)(base >> 32);
    rsrc.z = n * 4;
    rsrc.w = (4 << 0) | (5 << 3) | (6 << 6) | (7 << 9) | (2 << 12) | (10 << 15);
    v4f v;
    const int voff = lane * 4;
    asm volatile("buffer_load_format_xyzw %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(voff), "s"(rsrc) : "memory");
    dst[lane * 4 + 0] = v.x; dst[lane * 4 + 1] = v.y; dst[lane * 4 + 2] = v.z; dst[lane * 4 + 3] = v.w;
}

int main() {
    const int n = 64;
    std::vector<uint32_t> h(n);
    for (int i = 0; i < n; ++i) h[i] = (uint32_t)i | (uint32_t)(255 - i) << 8 | (uint32_t)(3 * i & 255) << 16 | 128u << 24;
    uint32_t * d_src; float * d_dst;
    hipMalloc(&d_src, n * 4); hipMalloc(&d_dst, n * 16);
    hipMemcpy(d_src, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d_src, d_dst, n);
    std::vector<float> out(n * 4);
    if (hipMemcpy(out.data(), d_dst, n * 16, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 1; }
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const float want[4] = { (float)i, (float)(255 - i), (float)(3 * i & 255), 128.0f };
        for (int c = 0; c < 4; ++c) if (out[i * 4 + c] != want[c]) bad++;
        if (i == 5 || i == 63) printf("lane %2d: bytes (%u, %u, %u, %u) -> floats %g %g %g %g\n", i, h[i] & 255, h[i] >> 8 & 255, h[i] >> 16 & 255, h[i] >> 24,
                                      out[i * 4], out[i * 4 + 1], out[i * 4 + 2], out[i * 4 + 3]);
    }
    printf("%s: %d of %d components differ\n", bad ? "FORMAT CONVERSION NOT AS EXPECTED" : "8_8_8_8 USCALED converts bytes to floats in the load", bad, n * 4);
    return bad ? 1 : 0;
}
