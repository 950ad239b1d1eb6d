// Exhaustive search for cheaper instruction sequences that reproduce the correctly rounded fp32 reciprocal /
// square root (the arithmetic contract of include/rt_math.h: rt_rcp(x) = 1.0f / x, rt_sqrt = IEEE sqrt) BIT FOR BIT.
// Every one of the 2^32 bit patterns is evaluated; mismatches are counted per candidate, inside and outside the
// guarded range the fast path would accept.   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off exact_math.hip -o exact_math
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float f_of(uint32_t u) { return __uint_as_float(u); }

__device__ __forceinline__ float rcp_a(float x)
{
    float y = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, y, 1.0f);
    return __builtin_fmaf(y, e, y);
}
__device__ __forceinline__ float rcp_b(float x)
{
    float y = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, y, 1.0f);
    y = __builtin_fmaf(y, e, y);
    e = __builtin_fmaf(-x, y, 1.0f);
    return __builtin_fmaf(y, e, y);
}
__device__ __forceinline__ float sqrt_a(float x)
{
    float r = __builtin_amdgcn_rsqf(x);
    float s = x * r;
    float h = 0.5f * r;
    float e = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(e, h, s);
}
__device__ __forceinline__ float sqrt_b(float x)
{
    float r = __builtin_amdgcn_rsqf(x);
    float s = x * r;
    float h = 0.5f * r;
    float e = __builtin_fmaf(-s, s, x);
    s = __builtin_fmaf(e, h, s);
    e = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(e, h, s);
}

// Markstein-style square root: refine both s ~ sqrt(x) and h ~ 1/(2 sqrt(x)), then one exact-residual correction
__device__ __forceinline__ float sqrt_c(float x)
{
    float r = __builtin_amdgcn_rsqf(x);
    float s = x * r;
    float h = 0.5f * r;
    float d = __builtin_fmaf(-s, h, 0.5f);
    s = __builtin_fmaf(s, d, s);
    h = __builtin_fmaf(h, d, h);
    float e = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(e, h, s);
}
// hardware sqrt (1 ulp) + one exact-residual correction with h = 0.5 * rsq(x)
__device__ __forceinline__ float sqrt_d(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    float h = 0.5f * __builtin_amdgcn_rsqf(x);
    float e = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(e, h, s);
}
// a / b through the exact reciprocal of b and one residual correction of the quotient
__device__ __forceinline__ float div_fast(float a, float b)
{
    float y = __builtin_amdgcn_rcpf(b);
    y = __builtin_fmaf(y, __builtin_fmaf(-b, y, 1.0f), y);
    float q = a * y;
    float r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, y, q);
}

// counters: [0] rcp_a in range, [1] rcp_a out of range, [2] rcp_b in, [3] rcp_b out, [4] sqrt_a in, [5] sqrt_a out, [6] sqrt_b in, [7] sqrt_b out
__global__ void sweep(unsigned long long* cnt, uint32_t* firstBad)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned long long c[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t u64 = blockIdx.x * blockDim.x + threadIdx.x; u64 < (1ull << 32); u64 += stride) {
        const uint32_t u = (uint32_t)u64;
        const float x = f_of(u);
        const uint32_t ex = (u >> 23) & 0xff;
        // reciprocal: guarded range = x normal with exponent such that 1/x is normal too: biased exponent in [3, 251]
        const bool rin = ex >= 3 && ex <= 251;
        const uint32_t want = __float_as_uint(1.0f / x);
        uint32_t a = __float_as_uint(rcp_a(x)), b = __float_as_uint(rcp_b(x));
        if (a != want) { c[rin ? 0 : 1]++; if (rin) atomicMin(&firstBad[0], u); }
        if (b != want) { c[rin ? 2 : 3]++; if (rin) atomicMin(&firstBad[1], u); }
        // sqrt: guarded range = positive normal x
        const bool sin_ = !(u >> 31) && ex >= 32 && ex <= 222; /* positive, 2^-95 <= x < 2^96 */
        const uint32_t wants = __float_as_uint(__builtin_sqrtf(x));
        uint32_t sa = __float_as_uint(sqrt_a(x)), sb = __float_as_uint(sqrt_b(x));
        const bool nanBoth = (wants & 0x7fffffffu) > 0x7f800000u;
        if (sa != wants && !(nanBoth && (sa & 0x7fffffffu) > 0x7f800000u)) { c[sin_ ? 4 : 5]++; }
        if (sb != wants && !(nanBoth && (sb & 0x7fffffffu) > 0x7f800000u)) { c[sin_ ? 6 : 7]++; if (sin_) atomicMin(&firstBad[3], u); }
        uint32_t sc = __float_as_uint(sqrt_c(x)), sd = __float_as_uint(sqrt_d(x));
        if (sc != wants && !(nanBoth && (sc & 0x7fffffffu) > 0x7f800000u)) c[sin_ ? 8 : 9]++;
        if (sd != wants && !(nanBoth && (sd & 0x7fffffffu) > 0x7f800000u)) c[sin_ ? 10 : 11]++;
        // the one division of rt_log: s = f / (2 + f) for the reduced argument f in [sqrt(2)/2 - 1, sqrt(2) - 1) (swept wider)
        if (x >= -0.30f && x <= 0.42f) {
            const float b = 2.0f + x;
            if (__float_as_uint(div_fast(x, b)) != __float_as_uint(x / b)) { c[12]++; atomicMin(&firstBad[2], u); }
            c[13]++;
        }
        // the one division of rt_exp: x * c / (2 - c), c = x - x^2 (P1 + x^2 P2), reduced |x| <= 0.5 ln 2 (swept to 0.36)
        if (x >= -0.36f && x <= 0.36f) {
            const float P1 = 1.6666625440e-1f, P2 = -2.7667332906e-3f;
            const float xx = x * x;
            const float cc = x - xx * (P1 + xx * P2);
            const float a = x * cc, b = 2.0f - cc;
            if (__float_as_uint(div_fast(a, b)) != __float_as_uint(a / b)) c[14]++;
            c[15]++;
        }
    }
    for (int k = 0; k < 16; k++)
        if (c[k]) atomicAdd(&cnt[k], c[k]);
}

int main()
{
    unsigned long long* d;
    uint32_t* fb;
    hipMalloc(&d, 128);
    hipMemset(d, 0, 128);
    hipMalloc(&fb, 16);
    hipMemset(fb, 0xff, 16);
    hipLaunchKernelGGL(sweep, dim3(256 * 16), dim3(256), 0, 0, d, fb);
    unsigned long long h[16];
    uint32_t hb[4];
    hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    hipMemcpy(hb, fb, 16, hipMemcpyDeviceToHost);
    const char* names[4] = {"rcp: v_rcp_f32 + 1 fma-Newton step", "rcp: v_rcp_f32 + 2 fma-Newton steps", "sqrt: v_rsq_f32 + 1 fma step", "sqrt: v_rsq_f32 + 2 fma steps"};
    for (int k = 0; k < 4; k++)
        printf("%-40s mismatches vs correctly rounded over all 2^32 inputs: %llu in the guarded range (first 0x%08x), %llu outside\n", names[k], h[2 * k], hb[k], h[2 * k + 1]);
    printf("%-40s mismatches: %llu in the guarded range, %llu outside\n", "sqrt: rsq + Markstein refinement (5 fma)", h[8], h[9]);
    printf("%-40s mismatches: %llu in the guarded range, %llu outside\n", "sqrt: v_sqrt_f32 + residual * 0.5 rsq", h[10], h[11]);
    printf("%-40s mismatches: %llu of %llu arguments (first mismatching f bits 0x%08x)\n", "log's division f/(2+f), fast form", h[12], h[13], hb[2]);
    printf("%-40s mismatches: %llu of %llu arguments\n", "exp's division x*c/(2-c), fast form", h[14], h[15]);
    return 0;
}
