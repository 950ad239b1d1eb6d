/*
 * rt_math.h — the fp32 arithmetic contract of the path tracer.
 *
 * The reference shader (Assets/Scripts/Tracer/RayCommon.hlsl, "RC") runs under
 * an HLSL compiler whose sqrt/div/log/cos/exp/pow are implementation-defined
 * approximations, so "the reference's results" are only defined up to that.
 * This header fixes one strict-IEEE binary32 definition of every non-trivial
 * operation RC uses — written with +,-,*,/ , sqrt and integer ops only, no FMA
 * contraction, no fast-math — so that the CPU oracle (g++ on x86-64) and the
 * HIP kernels (hipcc on gfx950) evaluate the SAME rounding sequence and agree
 * bit for bit.  A Monte-Carlo estimator with Russian roulette is chaotic in
 * its random decisions; bit-exact primitives are what make a 1e-4 relative
 * parity bar meaningful.
 *
 * Compile requirements (enforced by the build scripts):
 *    host:   g++ -O2 -fno-fast-math -ffp-contract=off
 *    device: hipcc -ffp-contract=off (f32 div/sqrt correctly rounded — the
 *            hipcc default — and f32 denormals preserved — the gfx9 default).
 *
 * log/exp follow the classic Sun fdlibm single-precision algorithms, sin/cos
 * the Cephes single-precision ones (Cody–Waite 3-term reduction + minimax
 * polynomials); tests/test_math.py bounds their error against float64 libm.
 */
#ifndef RT_MATH_H
#define RT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

/* RT_WAVE_ALL(p) — device only: true when p holds in EVERY active lane of the wavefront.  The functions below keep their special cases
 * (zeros, infinities, NaNs, subnormals, huge arguments) in per-lane branches; a per-lane branch costs the whole wave its exec-mask
 * bookkeeping (four to six scalar instructions) even when no lane takes it, and the scalar unit is shared by all the waves of a compute
 * unit.  With RT_WAVE_ALL(normal) in front, a wave whose active lanes are all ordinary runs the straight-line path behind ONE uniform
 * branch; a wave with a special lane runs the unchanged per-lane code.  Same operations on the same values per lane either way. */
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_MATH_NO_WAVE_PATHS)
#define RT_WAVE_ALL(p) (__builtin_amdgcn_ballot_w64(!(p)) == 0ull)
#else
#define RT_WAVE_ALL(p) false
#endif

/* ---------------------------------------------------------------- bit casts */
RT_HD uint32_t rt_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
RT_HD float rt_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

#define RT_INF (rt_u2f(0x7f800000u))

/* ------------------------------------------------- exact single operations */
RT_HD float rt_sqrt(float x) /* correctly rounded */
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_RCP_IEEE_SEQUENCE)
    /* Same bits in 5 instructions instead of the compiler's 16: for 2^-95 <= x < 2^96, v_rsq_f32 (1 ulp), s = x r and
     * one exact-residual correction s + (x - s s) r/2 give the correctly rounded square root — checked over all 2^32 bit
     * patterns on gfx950 (tools/ubench/exact_math.hip: 0 mismatches in that range); zeros, negatives, tiny, huge, inf
     * and NaN arguments take the IEEE sequence. */
    if ((rt_f2u(x) >> 23) - 32u <= 190u) { /* (no wave-uniform copy here: it costs the BVH kernels a spill) */
        const float r = __builtin_amdgcn_rsqf(x);
        const float s = x * r;
        const float e = __builtin_fmaf(-s, s, x);
        return __builtin_fmaf(e, 0.5f * r, s);
    }
#endif
    return __builtin_sqrtf(x);
}
RT_HD float rt_abs(float x) { return rt_u2f(rt_f2u(x) & 0x7fffffffu); }
RT_HD float rt_floor(float x) { return __builtin_floorf(x); } /* exact */

/* HLSL min/max: if one operand is NaN the other is returned (RC:223-226 relies
 * on this when invDir = inf meets a zero slab offset). */
RT_HD float rt_min(float a, float b) { return (a < b || b != b) ? a : b; }
RT_HD float rt_max(float a, float b) { return (a > b || b != b) ? a : b; }
/* HLSL sign(): -1, 0 or +1 (0 for NaN) */
RT_HD float rt_sign(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }
/* HLSL saturate(): clamp to [0,1], NaN -> 0 */
RT_HD float rt_saturate(float x) { return rt_min(rt_max(x, 0.0f), 1.0f); }
/* HLSL lerp(a,b,t) as compilers lower it: a + (b-a)*t */
RT_HD float rt_lerp(float a, float b, float t) { return a + (b - a) * t; }
/* HLSL '/': GPUs have no IEEE divide in shaders — a/b executes as a * rcp(b).  The strict
 * form used here keeps that shape with a correctly rounded reciprocal, so a reciprocal of
 * a wave-uniform or repeated denominator is computed once. */
RT_HD float rt_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_RCP_IEEE_SEQUENCE)
    /* Same bits, a third of the issue slots: on gfx950 v_rcp_f32 (1 ulp) followed by ONE fused Newton step is the
     * correctly rounded reciprocal for every input whose biased exponent lies in [3, 251] (x and 1/x both normal) —
     * checked over all 2^32 bit patterns against the compiler's 11-instruction IEEE sequence
     * (tools/ubench/exact_math.hip: 0 mismatches in that range); everything else — zeros, subnormals, the last two
     * binades, inf, NaN — takes the IEEE sequence. */
    const bool ordinary = ((rt_f2u(x) >> 23) & 0xffu) - 3u <= 248u;
    if (RT_WAVE_ALL(ordinary)) { /* wave-uniform branch */
        const float y = __builtin_amdgcn_rcpf(x);
        const float e = __builtin_fmaf(-x, y, 1.0f);
        return __builtin_fmaf(y, e, y);
    }
    if (ordinary) {
        const float y = __builtin_amdgcn_rcpf(x);
        const float e = __builtin_fmaf(-x, y, 1.0f);
        return __builtin_fmaf(y, e, y);
    }
#endif
    return 1.0f / x;
}
/* a / b, correctly rounded, for the two divisions inside rt_log and rt_exp.  On the device their operands are confined
 * to ranges (b in [1.6, 2.5]) over which reciprocal + quotient + one exact-residual correction was checked to equal the
 * IEEE quotient for EVERY argument the functions can produce (tools/ubench/exact_math.hip sweeps all 2^32 reduced
 * arguments: 0 mismatches; the only differing input, f = -0 in rt_log, cannot arise from x - 1). */
RT_HD float rt_div_narrow(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_RCP_IEEE_SEQUENCE)
    float y = __builtin_amdgcn_rcpf(b);
    y = __builtin_fmaf(y, __builtin_fmaf(-b, y, 1.0f), y);
    const float q = a * y;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);
#else
    return a / b;
#endif
}
#ifndef RT_MATH_IEEE
RT_HD float rt_div(float a, float b) { return a * rt_rcp(b); }
#else
/* RT_MATH_IEEE: the alternative reading of what HLSL leaves open — '/' as a correctly rounded
 * IEEE divide, normalize as v / sqrt(dot), smoothstep with its own divide.  Built for the ORACLE
 * only (oracle/liboracle_ieee.so); tests/test_contract_bracket.py measures how far the two readings
 * are apart (far below the Monte-Carlo error), bracketing the contract the reference cannot pin. */
RT_HD float rt_div(float a, float b) { return a / b; }
#endif
/* HLSL smoothstep(a,b,x) = saturate((x-a)/(b-a)) then Hermite; every call site of the shader
 * passes literal edges (RC:175-176), for which the compiler folds 1/(b-a) into a constant:
 * inv_range is that constant (correctly rounded). */
RT_HD float rt_smoothstep(float a, float inv_range, float x)
{
    float t = rt_saturate((x - a) * inv_range);
    return t * t * (3.0f - 2.0f * t);
}
/* the shader's form, smoothstep(a, b, x) with literal edges: 1/(b-a) is the folded constant above */
RT_HD float rt_smoothstep_edges(float a, float b, float x)
{
#ifndef RT_MATH_IEEE
    return rt_smoothstep(a, 1.0f / (b - a), x);
#else
    float t = rt_saturate((x - a) / (b - a));
    return t * t * (3.0f - 2.0f * t);
#endif
}

/* ------------------------------------------------------------------- logf */
RT_HD float rt_log(float x)
{
    const float ln2_hi = 6.9313812256e-01f; /* 0x3f317180 */
    const float ln2_lo = 9.0580006145e-06f; /* 0x3717f7d1 */
    const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f;
    const float Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
    uint32_t ix = rt_f2u(x);
    /* (no wave-uniform path here: a second copy of the polynomial costs every trace kernel spilled registers — measured, profiles/r06_chain_pool.txt) */
    int k = 0;
    if (ix < 0x00800000u || (ix >> 31)) {
        if ((ix << 1) == 0) return -RT_INF;          /* log(+-0) = -inf  */
        if (ix >> 31) return rt_u2f(0x7fc00000u);     /* log(x<0) = NaN   */
        k -= 25;                                       /* subnormal: scale */
        x *= 33554432.0f;
        ix = rt_f2u(x);
    } else if (ix >= 0x7f800000u) {
        return x;                                      /* inf or NaN       */
    } else if (ix == 0x3f800000u) {
        return 0.0f;
    }
    /* x = 2^k * (1+f), sqrt(2)/2 < 1+f < sqrt(2) */
    ix += 0x3f800000u - 0x3f3504f3u;
    k += (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    x = rt_u2f(ix);
    float f = x - 1.0f;
    float s = rt_div_narrow(f, 2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * (Lg2 + w * Lg4);
    float t2 = z * (Lg1 + w * Lg3);
    float R = t2 + t1;
    float hfsq = 0.5f * f * f;
    float dk = (float)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

/* ------------------------------------------------------------------- expf */
/* the straight-line form of rt_exp below for |x| < 87.33655 (the result is normal: k in [-126, 126]) — the same operations with the magnitude
 * classes as selects: k = 0 makes hi = x - 0 = x and lo = 0 exactly, y * 2^0 = y exactly, and |x| <= 2^-14 keeps its 1 + x.  Arguments that
 * underflow to 0 (x <= -103.97..., -inf included: every pow(0, y)) are folded in with a stand-in.  Device only, behind RT_WAVE_ALL. */
RT_HD float rt_exp_ordinary(float x)
{
    const float ln2hi = 6.9314575195e-1f, ln2lo = 1.4286067653e-6f, invln2 = 1.4426950216e+0f;
    const float P1 = 1.6666625440e-1f, P2 = -2.7667332906e-3f;
    uint32_t hx = rt_f2u(x);
    const int sign = (int)(hx >> 31);
    hx &= 0x7fffffffu;
    const bool under = sign && hx >= 0x42cff1b5u;
    if (under) { x = 0.0f; hx = 0u; }
    const bool big = hx > 0x3f851592u, mid = hx > 0x3eb17218u, tiny = !(hx > 0x39000000u);
    const int kb = (int)(invln2 * x + (sign ? -0.5f : 0.5f));
    const int k = big ? kb : (mid ? 1 - sign - sign : 0);
    const float fk = (float)k;
    const float hi = x - fk * ln2hi;
    const float lo = fk * ln2lo;
    const float xr = hi - lo;
    const float xx = xr * xr;
    const float c = xr - xx * (P1 + xx * P2);
    const float y = 1.0f + (rt_div_narrow(xr * c, 2.0f - c) - lo + hi);
    const float r = y * rt_u2f((uint32_t)(k + 127) << 23);
    return under ? 0.0f : (tiny ? 1.0f + x : r);
}
RT_HD float rt_exp(float x)
{
    const float ln2hi = 6.9314575195e-1f;  /* 0x3f317200 */
    const float ln2lo = 1.4286067653e-6f;  /* 0x35bfbe8e */
    const float invln2 = 1.4426950216e+0f;
    const float P1 = 1.6666625440e-1f, P2 = -2.7667332906e-3f;
    uint32_t hx = rt_f2u(x);
    int sign = (int)(hx >> 31);
    hx &= 0x7fffffffu;
    /* wave-uniform branch: every lane's result is normal, or an underflow to 0 */
    if (RT_WAVE_ALL(hx < 0x42aeac50u || (sign && hx >= 0x42cff1b5u && hx <= 0x7f800000u))) return rt_exp_ordinary(x);
    if (hx >= 0x42aeac50u) {               /* |x| >= 87.33655 or NaN */
        if (hx > 0x7f800000u) return x;    /* NaN */
        if (hx >= 0x42b17218u && !sign) return RT_INF;  /* overflow  */
        if (sign && hx >= 0x42cff1b5u) return 0.0f;     /* underflow */
    }
    float hi, lo;
    int k;
    if (hx > 0x3eb17218u) {                /* |x| > 0.5 ln2 */
        if (hx > 0x3f851592u)              /* |x| > 1.5 ln2 */
            k = (int)(invln2 * x + (sign ? -0.5f : 0.5f));
        else
            k = 1 - sign - sign;
        float fk = (float)k;
        hi = x - fk * ln2hi;
        lo = fk * ln2lo;
        x = hi - lo;
    } else if (hx > 0x39000000u) {         /* |x| > 2^-14 */
        k = 0;
        hi = x;
        lo = 0.0f;
    } else {
        return 1.0f + x;
    }
    float xx = x * x;
    float c = x - xx * (P1 + xx * P2);
    float y = 1.0f + (rt_div_narrow(x * c, 2.0f - c) - lo + hi);
    if (k == 0) return y;
    /* y * 2^k with a single rounding even when the result is subnormal */
    if (k > 127) { y *= 1.7014118346e38f; k -= 127; }          /* 2^127 */
    if (k < -126) {
        y *= rt_u2f((uint32_t)(k + 24 + 127) << 23);
        return y * 5.9604644775e-8f;                              /* 2^-24 */
    }
    return y * rt_u2f((uint32_t)(k + 127) << 23);
}

/* HLSL pow(x,y) for x >= 0, lowered the way GPUs do it (exp(y*log x)):
 * pow(0,y>0) = 0, pow(1,y) = 1, pow(x<0,y) = NaN. */
RT_HD float rt_pow(float x, float y) { return rt_exp(y * rt_log(x)); }

/* -------------------------------------------------------------- sinf / cosf */
/* r = x - n*(pi/2), n = round(x*2/pi); three-term Cody–Waite split of pi/2
 * (8 + 11 + 24 significant bits: n*P1 and n*P2 are exact for |n| < 2^13).
 * Accurate for |x| up to a few thousand; the tracer only passes [0, 2*pi]. */
RT_HD int rt_reduce_pio2(float x, float* r)
{
    const float TWO_OVER_PI = 0.636619772f;
    const float P1 = 1.5703125f;
    const float P2 = 4.837512969970703125e-4f;
    const float P3 = 7.54978995489188216e-8f;
    float q = x * TWO_OVER_PI;
    int n = (int)(q + (q < 0.0f ? -0.5f : 0.5f));
    float fn = (float)n;
    float t = x - fn * P1;
    t = t - fn * P2;
    t = t - fn * P3;
    *r = t;
    return n;
}
RT_HD float rt_sin_kernel(float r)
{
    float z = r * r;
    float p = (-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f;
    return r + r * z * p;
}
RT_HD float rt_cos_kernel(float r)
{
    float z = r * r;
    float p = (2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f;
    return 1.0f - 0.5f * z + z * z * p;
}
RT_HD float rt_sin(float x)
{
    if (RT_WAVE_ALL(rt_abs(x) < 3.0e4f)) { /* wave-uniform branch */
        float r;
        int n = rt_reduce_pio2(x, &r);
        float s = (n & 1) ? rt_cos_kernel(r) : rt_sin_kernel(r);
        return (n & 2) ? -s : s;
    }
    if (!(rt_abs(x) < 3.0e4f)) return (x != x || rt_abs(x) == RT_INF) ? rt_u2f(0x7fc00000u) : 0.0f;
    float r;
    int n = rt_reduce_pio2(x, &r);
    float s = (n & 1) ? rt_cos_kernel(r) : rt_sin_kernel(r);
    return (n & 2) ? -s : s;
}
RT_HD float rt_cos(float x)
{
    if (RT_WAVE_ALL(rt_abs(x) < 3.0e4f)) { /* wave-uniform branch */
        float r;
        int n = rt_reduce_pio2(x, &r);
        float c = (n & 1) ? rt_sin_kernel(r) : rt_cos_kernel(r);
        return ((n + 1) & 2) ? -c : c;
    }
    if (!(rt_abs(x) < 3.0e4f)) return (x != x || rt_abs(x) == RT_INF) ? rt_u2f(0x7fc00000u) : 1.0f;
    float r;
    int n = rt_reduce_pio2(x, &r);
    float c = (n & 1) ? rt_sin_kernel(r) : rt_cos_kernel(r);
    return ((n + 1) & 2) ? -c : c;
}

/* cos(x) and sin(x) of the same argument sharing one range reduction; bit-identical
 * to calling rt_cos(x) and rt_sin(x) separately (same operations on the same r). */
RT_HD void rt_sincos(float x, float* s_out, float* c_out)
{
    if (RT_WAVE_ALL(rt_abs(x) < 3.0e4f)) { /* wave-uniform branch */
        float r;
        int n = rt_reduce_pio2(x, &r);
        float sk = rt_sin_kernel(r), ck = rt_cos_kernel(r);
        float s = (n & 1) ? ck : sk;
        float c = (n & 1) ? sk : ck;
        *s_out = (n & 2) ? -s : s;
        *c_out = ((n + 1) & 2) ? -c : c;
        return;
    }
    if (!(rt_abs(x) < 3.0e4f)) {
        *s_out = rt_sin(x);
        *c_out = rt_cos(x);
        return;
    }
    float r;
    int n = rt_reduce_pio2(x, &r);
    float sk = rt_sin_kernel(r), ck = rt_cos_kernel(r);
    float s = (n & 1) ? ck : sk;
    float c = (n & 1) ? sk : ck;
    *s_out = (n & 2) ? -s : s;
    *c_out = ((n + 1) & 2) ? -c : c;
}

/* ---------------------------------------------------------------- vectors */
struct rt_f3 { float x, y, z; };
struct rt_f2 { float x, y; };

RT_HD rt_f3 rt_v3(float x, float y, float z) { rt_f3 r = {x, y, z}; return r; }
RT_HD rt_f3 rt_v3s(float s) { rt_f3 r = {s, s, s}; return r; }
RT_HD rt_f3 operator+(rt_f3 a, rt_f3 b) { return rt_v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD rt_f3 operator-(rt_f3 a, rt_f3 b) { return rt_v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD rt_f3 operator*(rt_f3 a, rt_f3 b) { return rt_v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD rt_f3 operator*(rt_f3 a, float s) { return rt_v3(a.x * s, a.y * s, a.z * s); }
RT_HD rt_f3 operator*(float s, rt_f3 a) { return rt_v3(s * a.x, s * a.y, s * a.z); }
/* float3 / scalar: one reciprocal, three multiplies (see rt_div) */
#ifndef RT_MATH_IEEE
RT_HD rt_f3 operator/(rt_f3 a, float s) { float r = rt_rcp(s); return rt_v3(a.x * r, a.y * r, a.z * r); }
#else
RT_HD rt_f3 operator/(rt_f3 a, float s) { return rt_v3(a.x / s, a.y / s, a.z / s); }
#endif
RT_HD rt_f3 operator-(rt_f3 a) { return rt_v3(-a.x, -a.y, -a.z); }
/* HLSL dot(): left-to-right sum of products */
RT_HD float rt_dot(rt_f3 a, rt_f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_HD rt_f3 rt_cross(rt_f3 a, rt_f3 b)
{
    return rt_v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* HLSL rsqrt(): an approximation instruction on GPUs (1-2 ulp).  The strict definition
 * here is a fixed algorithm in IEEE mul/sub only: the classic exponent-halving seed and
 * three Newton steps y <- y*(1.5 - 0.5*x*y*y) (relative error 3.4e-2 -> 1.7e-3 -> 4.6e-6 ->
 * fp32 rounding, < 1.5 ulp; tests/test_math.py).  Half the cost of sqrt followed by a divide.
 * rsqrt(+-0) = +-inf, rsqrt(x<0) = NaN, rsqrt(inf) = 0, subnormals are pre-scaled by 2^48. */
RT_HD float rt_rsqrt(float x)
{
    if (RT_WAVE_ALL(x >= 1.17549435e-38f && x < RT_INF)) { /* wave-uniform branch: positive, normal, finite in every lane — the steps below with scale = 1 */
        float y = rt_u2f(0x5f375a86u - (rt_f2u(x) >> 1));
        const float h = 0.5f * x;
        y = y * (1.5f - h * y * y);
        y = y * (1.5f - h * y * y);
        y = y + y * (0.5f - h * y * y);
        return y * 1.0f;
    }
    if (!(x > 0.0f)) return (x == 0.0f) ? rt_u2f((rt_f2u(x) & 0x80000000u) | 0x7f800000u) : rt_u2f(0x7fc00000u);
    if (x == RT_INF) return 0.0f;
    float scale = 1.0f;
    if (x < 1.17549435e-38f) { x *= 281474976710656.0f; scale = 16777216.0f; } /* 2^48, 2^24 */
    float y = rt_u2f(0x5f375a86u - (rt_f2u(x) >> 1));
    const float h = 0.5f * x;
    y = y * (1.5f - h * y * y);
    y = y * (1.5f - h * y * y);
    y = y + y * (0.5f - h * y * y); /* last step in residual form: its rounding error stays below 1 ulp */
    return y * scale;
}
/* HLSL normalize(): DXC lowers it to v * rsqrt(dot(v,v)) — followed here with
 * the strict rsqrt above (one sqrt, one divide, three multiplies).  A zero
 * vector gives NaNs (0 * inf), like the shader. */
#ifndef RT_MATH_IEEE
RT_HD rt_f3 rt_normalize(rt_f3 v) { return v * rt_rsqrt(rt_dot(v, v)); }
#else
RT_HD rt_f3 rt_normalize(rt_f3 v) { return v / rt_sqrt(rt_dot(v, v)); }
#endif
RT_HD rt_f3 rt_lerp3(rt_f3 a, rt_f3 b, float t)
{
    return rt_v3(rt_lerp(a.x, b.x, t), rt_lerp(a.y, b.y, t), rt_lerp(a.z, b.z, t));
}
/* HLSL intrinsic reflect(i,n) = i - 2*n*dot(i,n)  (RC:526); also RC:419-422 */
RT_HD rt_f3 rt_reflect(rt_f3 i, rt_f3 n) { return i - (2.0f * rt_dot(i, n)) * n; }

/* mul(M, float4(v, w)).xyz with M column-major (Unity Matrix4x4 memory order):
 * row r = m[r]*v.x + m[4+r]*v.y + m[8+r]*v.z + m[12+r]*w, summed left to right */
RT_HD rt_f3 rt_mul_point(const float* m, rt_f3 v, float w)
{
    return rt_v3(m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * w,
                 m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * w,
                 m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * w);
}

/* linear -> sRGB transfer (what Unity's linear colour space applies when the Display pass
 * writes to the sRGB back buffer), then 8-bit quantisation with round-half-up */
RT_HD float rt_linear_to_srgb(float c)
{
    c = rt_saturate(c);
    return c <= 0.0031308f ? 12.92f * c : 1.055f * rt_pow(c, 1.0f / 2.4f) - 0.055f;
}
RT_HD uint32_t rt_srgb8(float c) { return (uint32_t)(rt_linear_to_srgb(c) * 255.0f + 0.5f); }

/* ------------------------------------------------------------------- RNG */
/* PCG hash step — RC:127-133 */
RT_HD uint32_t rt_next_random(uint32_t* state)
{
    *state = *state * 747796405u + 2891336453u;
    uint32_t result = ((*state >> ((*state >> 28) + 4u)) ^ *state) * 277803737u;
    result = (result >> 22) ^ result;
    return result;
}
/* RC:135-138: NextRandom / 4294967295.0 — the literal is a float, i.e. 2^32;
 * uint->float conversion rounds to nearest even, the divide is exact.
 * Range [0,1] INCLUSIVE (quirk Q4). */
RT_HD float rt_random_value(uint32_t* state)
{
    return (float)rt_next_random(state) / 4294967296.0f;
}

#endif /* RT_MATH_H */
