// Host stand-in for <hip/hip_runtime.h>: just enough of the HIP device vocabulary to compile the traversal headers
// (par_raytracer_amd/csrc/dev_trace*.h) with g++ and run them one "lane" at a time.  Test infrastructure only
// (tests/trace_host_harness.cpp); the product never sees this file.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

struct float4 { float x, y, z, w; };
struct uint4 { unsigned int x, y, z, w; };
struct int2 { int x, y; };
struct ulonglong2 { unsigned long long x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = { x, y, z, w }; return r; }
static inline int2 make_int2(int x, int y) { int2 r = { x, y }; return r; }
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { ulonglong2 r = { x, y }; return r; }
struct dim3_shim { unsigned int x, y, z; };
static dim3_shim threadIdx = { 0, 0, 0 }, blockIdx = { 0, 0, 0 }, blockDim = { 1, 1, 1 }, gridDim = { 1, 1, 1 };

static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline float __uint_as_float(unsigned int v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __popc(unsigned int v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned int)v) : 32; }
static inline unsigned long long __ballot(bool p) { return p ? 1ull : 0ull; }     // one lane
static inline unsigned int __builtin_amdgcn_mbcnt_lo(unsigned int, unsigned int v) { return v; }
static inline unsigned int __builtin_amdgcn_mbcnt_hi(unsigned int, unsigned int v) { return v; }
template <class T> static inline T atomicAdd(T * p, T v) { T o = *p; *p = o + v; return o; }
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_fetch_add(p, v, order, scope) (*(p) += (v))
#define __hip_atomic_load(p, order, scope) (*(p))
