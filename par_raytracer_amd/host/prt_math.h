// prt_math.h - by-value float vectors with the reference's names and evaluation order.
//
// API surface kept from the reference (brt.h:9-18 integer names; mathlib.h:24-386 Vector2/3/4;
// mathlib.h:540-737 Matrix33; geometry.h:4-12 Sphere/Ray).  Only what the live paths use is provided;
// the reference's Matrix22/44, Quaternion and Euler helpers are dead code.
//
// Bit-exactness rules this header encodes ("Expression association"):
//   Dot(a,b)   = (a.x*b.x + a.y*b.y) + a.z*b.z                       mathlib.h:236
//   Cross(a,b) = (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)   mathlib.h:241-245
//   Normalize  = component / sqrtf(len^2), input returned unchanged when len^2 == 0  mathlib.h:253-262
//   Min/Max/Clamp are the comparison forms a<b?a:b / a>b?a:b, not fminf/fmaxf  mathlib.h:7-9
// Translation units including it must be compiled with -ffp-contract=off.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int16_t s16;
typedef int32_t s32;
typedef int64_t s64;

#define PI32 (3.1415927f)
#define DEG2RAD(x) ((x) / 180.0f * PI32)

template <typename T> inline T PrtMin(T a, T b) { return a < b ? a : b; }
template <typename T> inline T PrtMax(T a, T b) { return a > b ? a : b; }
template <typename T> inline T PrtClamp(T n, T lo, T hi) { return PrtMin(PrtMax(n, lo), hi); }

struct Vector2 {
    float x, y;
    Vector2() : x(0.0f), y(0.0f) {}
    Vector2(float x_, float y_) : x(x_), y(y_) {}
    Vector2 & operator+=(Vector2 v) { x += v.x; y += v.y; return *this; }
};
inline Vector2 operator+(Vector2 a, Vector2 b) { return Vector2(a.x + b.x, a.y + b.y); }
inline Vector2 operator-(Vector2 a, Vector2 b) { return Vector2(a.x - b.x, a.y - b.y); }
inline Vector2 operator*(Vector2 a, float s) { return Vector2(a.x * s, a.y * s); }

struct Vector3 {
    float x, y, z;
    Vector3() : x(0.0f), y(0.0f), z(0.0f) {}
    Vector3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    Vector3 & operator+=(Vector3 v) { x += v.x; y += v.y; z += v.z; return *this; }
    Vector3 & operator-=(Vector3 v) { x -= v.x; y -= v.y; z -= v.z; return *this; }
    Vector3 & operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
    Vector3 & operator/=(float s) { x /= s; y /= s; z /= s; return *this; }
};
inline Vector3 operator+(Vector3 a, Vector3 b) { return Vector3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vector3 operator-(Vector3 a, Vector3 b) { return Vector3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vector3 operator*(Vector3 a, Vector3 b) { return Vector3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline Vector3 operator*(Vector3 a, float s) { return Vector3(a.x * s, a.y * s, a.z * s); }
inline Vector3 operator/(Vector3 a, float s) { return Vector3(a.x / s, a.y / s, a.z / s); }
inline Vector3 operator-(Vector3 a) { return a * -1.0f; }
inline float Dot(Vector3 a, Vector3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vector3 Cross(Vector3 a, Vector3 b) {
    return Vector3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
inline float Length(Vector3 v) { return sqrtf(Dot(v, v)); }
inline Vector3 Normalize(Vector3 a) {
    float length_sq = Dot(a, a);
    if (length_sq == 0.0f) return a;
    return a / sqrtf(length_sq);
}

struct Vector4 {
    float x, y, z, w;
    Vector4() : x(0.0f), y(0.0f), z(0.0f), w(0.0f) {}
    Vector4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    Vector4 & operator+=(Vector4 v) { x += v.x; y += v.y; z += v.z; w += v.w; return *this; }
    Vector4 & operator*=(Vector4 v) { x *= v.x; y *= v.y; z *= v.z; w *= v.w; return *this; }
    Vector4 & operator*=(float s) { x *= s; y *= s; z *= s; w *= s; return *this; }
    Vector4 & operator/=(float s) { x /= s; y /= s; z /= s; w /= s; return *this; }
};
inline Vector4 operator+(Vector4 a, Vector4 b) { return Vector4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline Vector4 operator-(Vector4 a, Vector4 b) { return Vector4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline Vector4 operator*(Vector4 a, Vector4 b) { return Vector4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline Vector4 operator*(Vector4 a, float s) { return Vector4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline Vector4 operator/(Vector4 a, float s) { return Vector4(a.x / s, a.y / s, a.z / s, a.w / s); }

// Row-major 3x3, element (i,j) at e[3*i+j] (mathlib.h:540-586).
struct Matrix33 {
    float e[9];
    Matrix33() { for (int i = 0; i < 9; ++i) e[i] = 0.0f; }
    float & operator()(size_t i, size_t j) { return e[i * 3 + j]; }
    float operator()(size_t i, size_t j) const { return e[i * 3 + j]; }
    void SetIdentity() {
        for (int i = 0; i < 9; ++i) e[i] = 0.0f;
        e[0] = e[4] = e[8] = 1.0f;
    }
};
inline Matrix33 operator*(const Matrix33 & a, const Matrix33 & b) {   // mathlib.h:657-698: sums left to right
    Matrix33 r;
    for (size_t i = 0; i < 3; ++i)
        for (size_t j = 0; j < 3; ++j)
            r(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j);
    return r;
}
inline Vector3 operator*(const Matrix33 & a, Vector3 b) {             // mathlib.h:700-715
    return Vector3(a(0, 0) * b.x + a(0, 1) * b.y + a(0, 2) * b.z,
                   a(1, 0) * b.x + a(1, 1) * b.y + a(1, 2) * b.z,
                   a(2, 0) * b.x + a(2, 1) * b.y + a(2, 2) * b.z);
}
inline Matrix33 Transpose(const Matrix33 & m) {
    Matrix33 r;
    for (size_t i = 0; i < 3; ++i)
        for (size_t j = 0; j < 3; ++j) r(i, j) = m(j, i);
    return r;
}

struct Sphere {          // geometry.h:4-7, 16 bytes
    Vector3 center;
    float radius;
};

struct Ray {             // geometry.h:9-12, 24 bytes
    Vector3 origin;
    Vector3 direction;
};
