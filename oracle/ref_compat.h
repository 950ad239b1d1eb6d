/*
 * ref_compat.h — just enough of the HLSL language for a C++ compiler to accept the REFERENCE'S OWN shader text
 * (Assets/Scripts/Tracer/RayCommon.hlsl + RayCompute.compute) after the short list of mechanical rewrites in
 * oracle/make_ref.py.  TEST INFRASTRUCTURE ONLY (see rt_oracle.cpp's header): the result, oracle/_ref/libref.so,
 * exists to pin oracle/rt_oracle.cpp — the hand restatement — to the reference's text.  Control flow, operation
 * order, operand order, quirks Q1-Q13: all of that is whatever the reference's file says, because it IS the
 * reference's file.  What HLSL leaves to its compiler — what '/', normalize, pow, smoothstep, sqrt, log, exp, sin,
 * cos evaluate to in fp32 — is taken from include/rt_math.h, the arithmetic contract shared with the oracle and the
 * HIP kernels (-DRT_MATH_IEEE selects the other reading, as for liboracle_ieee.so).
 *
 * `float` of the shader text becomes `hfloat` (a class around one IEEE binary32) so that '/' can follow the
 * contract and nothing is ever promoted to double; vectors are componentwise with scalar broadcast, like HLSL's.
 * No reference source is copied here: this header declares types and intrinsics of the *language*.
 */
#ifndef RT_REF_COMPAT_H
#define RT_REF_COMPAT_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "../include/rt_math.h"

namespace hlsl_ref {

typedef uint32_t uint;

#define HLSL_ARITH(T) class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type

/* ---------------------------------------------------------------- float */
struct hfloat {
    float v;
    hfloat() = default;
    template <HLSL_ARITH(T)> hfloat(T t) : v((float)t) {} /* int / uint / bool / literal -> float, round to nearest even */
    hfloat& operator+=(hfloat o);
    hfloat& operator-=(hfloat o);
    hfloat& operator*=(hfloat o);
    hfloat& operator/=(hfloat o);
};
/* float literals of the shader text are rewritten to 0.35_h: decimal -> binary32 directly (HLSL literals are float) */
inline hfloat operator""_h(const char* s) { return hfloat(strtof(s, nullptr)); }

inline hfloat operator+(hfloat a, hfloat b) { return hfloat(a.v + b.v); }
inline hfloat operator-(hfloat a, hfloat b) { return hfloat(a.v - b.v); }
inline hfloat operator*(hfloat a, hfloat b) { return hfloat(a.v * b.v); }
inline hfloat operator/(hfloat a, hfloat b) { return hfloat(rt_div(a.v, b.v)); } /* the contract's '/' */
inline hfloat operator-(hfloat a) { return hfloat(-a.v); }
inline bool operator<(hfloat a, hfloat b) { return a.v < b.v; }
inline bool operator>(hfloat a, hfloat b) { return a.v > b.v; }
inline bool operator<=(hfloat a, hfloat b) { return a.v <= b.v; }
inline bool operator>=(hfloat a, hfloat b) { return a.v >= b.v; }
inline bool operator==(hfloat a, hfloat b) { return a.v == b.v; }
inline bool operator!=(hfloat a, hfloat b) { return a.v != b.v; }
inline hfloat& hfloat::operator+=(hfloat o) { return *this = *this + o; }
inline hfloat& hfloat::operator-=(hfloat o) { return *this = *this - o; }
inline hfloat& hfloat::operator*=(hfloat o) { return *this = *this * o; }
inline hfloat& hfloat::operator/=(hfloat o) { return *this = *this / o; }

static const hfloat HLSL_INF = hfloat(RT_INF); /* the text's 1.#INF */

/* -------------------------------------------------------------- vectors */
struct uint2;
struct float2 {
    hfloat x, y;
    float2() = default;
    float2(hfloat x_, hfloat y_) : x(x_), y(y_) {}
    float2(hfloat s) : x(s), y(s) {}
    template <HLSL_ARITH(T)> float2(T s) : x(s), y(s) {}
    float2(const uint2& u); /* uint2 -> float2 promotion */
    float2& operator*=(float2 o);
};
struct uint2 {
    uint x, y;
    uint2() = default;
    uint2(uint x_, uint y_) : x(x_), y(y_) {}
    uint2(const float2& f) : x((uint)f.x.v), y((uint)f.y.v) {} /* ftou: truncation */
};
inline float2::float2(const uint2& u) : x(u.x), y(u.y) {}
struct uint3 {
    uint x, y, z;
    uint3() = default;
    uint3(uint x_, uint y_, uint z_) : x(x_), y(y_), z(z_) {}
    uint2 xy() const { return uint2(x, y); }
};
struct int2 {
    int v[2];
    int& operator[](int i) { return v[i]; }
};

struct float3 {
    hfloat x, y, z;
    float3() = default;
    float3(hfloat x_, hfloat y_, hfloat z_) { x = x_; y = y_; z = z_; }
    float3(float2 xy_, hfloat z_) { x = xy_.x; y = xy_.y; z = z_; }
    float3(hfloat s) { x = s; y = s; z = s; }
    template <HLSL_ARITH(T)> float3(T s) { x = hfloat(s); y = x; z = x; }
    float3 xyz() const { return *this; }
    float3 rgb() const { return *this; }
    hfloat r() const { return x; } /* colour-named components (rvalue uses only) */
    hfloat g() const { return y; }
    hfloat b() const { return z; }
    float2 xz() const { return float2(x, z); }
    float2 zy() const { return float2(z, y); }
    float2 xy() const { return float2(x, y); }
    float3& operator+=(float3 o);
    float3& operator*=(float3 o);
};
struct float4 {
    hfloat x, y, z, w;
    float4() = default;
    float4(hfloat x_, hfloat y_, hfloat z_, hfloat w_) { x = x_; y = y_; z = z_; w = w_; }
    float4(float3 v, hfloat w_) { x = v.x; y = v.y; z = v.z; w = w_; }
    float4(hfloat s) { x = s; y = s; z = s; w = s; }
    template <HLSL_ARITH(T)> float4(T s) { x = hfloat(s); y = x; z = x; w = x; }
    float3 xyz() const { return float3(x, y, z); }
    float3 rgb() const { return float3(x, y, z); }
    float4& operator+=(float4 o);
};

#define HLSL_VEC_OPS(OP)                                                                                       \
    inline float2 operator OP(float2 a, float2 b) { return float2(a.x OP b.x, a.y OP b.y); }                   \
    inline float3 operator OP(float3 a, float3 b) { return float3(a.x OP b.x, a.y OP b.y, a.z OP b.z); }       \
    inline float4 operator OP(float4 a, float4 b) { return float4(a.x OP b.x, a.y OP b.y, a.z OP b.z, a.w OP b.w); }
HLSL_VEC_OPS(+)
HLSL_VEC_OPS(-)
HLSL_VEC_OPS(*)
HLSL_VEC_OPS(/) /* componentwise contract '/': v / s = (v.x * rcp(s), ...) */
inline float2 operator-(float2 a) { return float2(-a.x, -a.y); }
inline float3 operator-(float3 a) { return float3(-a.x, -a.y, -a.z); }
inline float2& float2::operator*=(float2 o) { return *this = *this * o; }
inline float3& float3::operator+=(float3 o) { return *this = *this + o; }
inline float3& float3::operator*=(float3 o) { return *this = *this * o; }
inline float4& float4::operator+=(float4 o) { return *this = *this + o; }

/* Unity Matrix4x4 / HLSL column_major float4x4: memory order m00 m10 m20 m30 m01 ...; _mRC = row R,
 * column C */
struct float4x4 {
    hfloat m[16];
    float3 _m00_m10_m20() const { return float3(m[0], m[1], m[2]); }
    float3 _m01_m11_m21() const { return float3(m[4], m[5], m[6]); }
};

/* ---------------------------------------------------------- intrinsics */
inline rt_f3 to_rt(float3 a) { return rt_v3(a.x.v, a.y.v, a.z.v); }
inline float3 from_rt(rt_f3 a) { return float3(hfloat(a.x), hfloat(a.y), hfloat(a.z)); }

inline hfloat sqrt(hfloat x) { return hfloat(rt_sqrt(x.v)); }
inline hfloat log(hfloat x) { return hfloat(rt_log(x.v)); }
inline hfloat exp(hfloat x) { return hfloat(rt_exp(x.v)); }
inline float3 exp(float3 a) { return float3(exp(a.x), exp(a.y), exp(a.z)); }
inline hfloat sin(hfloat x) { return hfloat(rt_sin(x.v)); }
inline hfloat cos(hfloat x) { return hfloat(rt_cos(x.v)); }
inline hfloat pow(hfloat x, hfloat y) { return hfloat(rt_pow(x.v, y.v)); }
inline hfloat abs(hfloat x) { return hfloat(rt_abs(x.v)); }
inline hfloat sign(hfloat x) { return hfloat(rt_sign(x.v)); } /* HLSL returns int -1/0/1; every use multiplies a float by it */
inline hfloat floor(hfloat x) { return hfloat(rt_floor(x.v)); }
inline float2 floor(float2 a) { return float2(floor(a.x), floor(a.y)); }
inline hfloat min(hfloat a, hfloat b) { return hfloat(rt_min(a.v, b.v)); }
inline hfloat max(hfloat a, hfloat b) { return hfloat(rt_max(a.v, b.v)); }
inline float3 min(float3 a, float3 b) { return float3(min(a.x, b.x), min(a.y, b.y), min(a.z, b.z)); }
inline float3 max(float3 a, float3 b) { return float3(max(a.x, b.x), max(a.y, b.y), max(a.z, b.z)); }
inline hfloat smoothstep(hfloat a, hfloat b, hfloat x) { return hfloat(rt_smoothstep_edges(a.v, b.v, x.v)); }
inline hfloat lerp(hfloat a, hfloat b, hfloat t) { return hfloat(rt_lerp(a.v, b.v, t.v)); }
inline float3 lerp(float3 a, float3 b, float3 t) { return float3(lerp(a.x, b.x, t.x), lerp(a.y, b.y, t.y), lerp(a.z, b.z, t.z)); }
inline hfloat dot(float3 a, float3 b) { return hfloat(rt_dot(to_rt(a), to_rt(b))); }
inline hfloat dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline float3 cross(float3 a, float3 b) { return from_rt(rt_cross(to_rt(a), to_rt(b))); }
inline float3 normalize(float3 a) { return from_rt(rt_normalize(to_rt(a))); }
inline float4 normalize(float4 a)
{
#ifndef RT_MATH_IEEE
    return a * hfloat(rt_rsqrt(dot(a, a).v));
#else
    return a / sqrt(dot(a, a));
#endif
}
inline float3 reflect(float3 i, float3 n) { return from_rt(rt_reflect(to_rt(i), to_rt(n))); }
/* mul(M, v), column vector: row r = sum over columns, left to right (rt_mul_point's order, plus the fourth row) */
inline float4 mul(const float4x4& M, float4 v)
{
    const hfloat* m = M.m;
    return float4(m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w, m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w,
                  m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w, m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w);
}

/* ------------------------------------------------------------- resources */
template <class T> struct StructuredBuffer {
    const T* data = nullptr;
    int64_t count = 0;
    const T& operator[](int64_t i) const
    {
        if (i < 0 || i >= count) { fprintf(stderr, "ref: StructuredBuffer index %lld out of %lld\n", (long long)i, (long long)count); abort(); }
        return data[i];
    }
};
template <class T> struct AppendStructuredBuffer {};
template <class T> struct RWTexture2D {
    T* data = nullptr;
    uint width = 0, height = 0;
    T& operator[](uint2 p) { return data[(size_t)p.y * width + p.x]; }
};

/* the shader's never-output `int2 stats` (RC:339: [0] triangle tests RC:254, [1] box tests RC:271) made visible:
 * make_ref.py zero-initialises the local and attaches one of these, which adds it to a per-thread total on scope exit */
extern thread_local int64_t g_ref_stats[2];
extern thread_local int64_t g_ref_collisions; /* calls of CalculateRayCollision = path segments */
struct RefStatsExport {
    int2& s;
    explicit RefStatsExport(int2& s_) : s(s_) {}
    ~RefStatsExport() { g_ref_stats[0] += s[0]; g_ref_stats[1] += s[1]; g_ref_collisions++; }
};

} /* namespace hlsl_ref */
#endif
