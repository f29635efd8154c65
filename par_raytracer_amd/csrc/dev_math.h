// dev_math.h - float3 arithmetic with the reference's expression association, host + device.
//
// Parity is about control flow, not ulps: every hit/miss, Russian-roulette and
// shadow decision of the reference depends only on +, -, *, /, sqrtf and integer ops, all of which
// gfx950 reproduces bit for bit PROVIDED mul and add are never fused.  This translation unit is
// compiled with -ffp-contract=off (hipcc contracts by default); nothing here may use fmaf except the
// BVH slab test (dev_trace.h), which only has to be conservative.
//   Dot   = (a.x*b.x + a.y*b.y) + a.z*b.z                      mathlib.h:236
//   Cross = (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)   mathlib.h:241-245
//   Normalize: each component divided by sqrtf(len^2); unchanged if len^2 == 0   mathlib.h:253-262
//   Max(a,b) = a > b ? a : b ; Min(a,b) = a < b ? a : b (macros, NaN picks b)   mathlib.h:7-8
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PRT_HD __host__ __device__ __forceinline__
#define PRT_D __device__ __forceinline__

namespace prt {

struct f3 {
    float x, y, z;
};

PRT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PRT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PRT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PRT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PRT_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PRT_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
PRT_HD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PRT_HD f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
PRT_HD f3 normalize3(f3 a) {
    float length_sq = dot3(a, a);
    if (length_sq == 0.0f) return a;
    return a / sqrtf(length_sq);
}
PRT_HD float ref_max(float a, float b) { return a > b ? a : b; }
PRT_HD float ref_min(float a, float b) { return a < b ? a : b; }

}  // namespace prt
