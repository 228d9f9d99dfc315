// VecMath.h — the handful of float3 helpers the host layer needs (the reference pulls them from NVIDIA's
// sutil vector_math.h, src/vector_math.h:146,396,436,560-606; written fresh here, same expression order so
// Camera::UVWFrame rounds identically).
#pragma once
#include <cmath>

#if !defined(HIP_INCLUDE_HIP_AMD_DETAIL_HIP_VECTOR_TYPES_H) && !defined(HIP_INCLUDE_HIP_HIP_VECTOR_TYPES_H)
struct float3 { float x, y, z; };
struct uint3 { unsigned int x, y, z; };
struct uchar3 { unsigned char x, y, z; };
inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
#endif
inline float3 make_float3(float s) { return make_float3(s, s, s); }

#ifndef M_PIf
#define M_PIf 3.14159265358979323846f
#endif

inline float3 operator+(const float3& a, const float3& b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline float3 operator-(const float3& a, const float3& b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline float3 operator-(const float3& a) { return make_float3(-a.x, -a.y, -a.z); }
inline float3 operator*(const float3& a, float s) { return make_float3(a.x * s, a.y * s, a.z * s); }
inline float3 operator*(float s, const float3& a) { return make_float3(a.x * s, a.y * s, a.z * s); }
inline float3& operator*=(float3& a, float s) { a.x *= s; a.y *= s; a.z *= s; return a; }
inline float3& operator+=(float3& a, const float3& b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
inline float dot(const float3& a, const float3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross(const float3& a, const float3& b)
{
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float length(const float3& v) { return sqrtf(dot(v, v)); }
inline float3 normalize(const float3& v) { float invLen = 1.0f / sqrtf(dot(v, v)); return v * invLen; }
