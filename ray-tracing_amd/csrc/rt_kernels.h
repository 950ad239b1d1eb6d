/*
 * rt_kernels.h — the gfx950 path-tracing kernel of libraytrace_hip.so.
 *
 * Replaces the reference's HLSL kernels RayTrace / ResetAccumulated
 * (Assets/Scripts/Tracer/RayCompute.compute:10-32) and everything they call in
 * RayCommon.hlsl ("RC"), with the same per-pixel results:
 *   - persistent single-wave workgroups; a lane owns one pixel at a time and runs
 *     that pixel's serial RNG chain (quirk Q13), taking the next unassigned pixel of
 *     the wave's current 8x8 tile (tiles come from a global atomic queue, longest
 *     pixel chains first) as soon as its pixel is finished;
 *   - instead of the reference's nested sample / bounce / model / node loops a lane
 *     is a state machine and the wave executes one kind of work at a time for the
 *     lanes that need it: camera ray -> spheres (two-phase) + conservative
 *     world-space root filter over the models -> resumable traversal with
 *     majority-phase scheduling and suspension -> shading;
 *   - the per-ray BVH order (near child first, strict '<', RC:256,274-281) is
 *     the reference's, so closest-hit ties resolve identically; the near child
 *     stays in a register instead of being pushed and popped, the far child goes
 *     to a per-lane stack in LDS laid out [level][lane] (bank = lane, conflict
 *     free for ds_read/write_b32); pixel bookkeeping is parked in LDS next to it;
 *   - spheres / model matrices / filter boxes / uniforms are wave-uniform and come
 *     through scalar loads (constant address space);
 *   - no MFMA: branchy scalar fp32, there is no contraction to map.
 *
 * Arithmetic follows include/rt_math.h (strict fp32, no contraction) so the
 * output is bit-identical to oracle/rt_oracle.cpp.
 */
#ifndef RT_KERNELS_H
#define RT_KERNELS_H

#include <hip/hip_runtime.h>

#include "../../include/rt_abi.h"
#include "../../include/rt_math.h"
#include "rt_device.h"

/* Tuning knobs (values measured on MI355X). */
/* traverse() is left once active <= entered * NUM/DEN lanes are still traversing */
#ifndef RT_SUSPEND_NUM
#define RT_SUSPEND_NUM 3
#define RT_SUSPEND_DEN 8
#endif
/* waves per SIMD the register allocator must leave room for (launch_bounds 2nd argument).  Built without the SLP
 * vectoriser (Makefile) the BVH variants need 83 VGPRs: 6 waves = 80 VGPRs costs no spill and is worth 2–7 % per frame
 * (7 waves: 30 spilled dwords, slower); the FLAT variant fits 8 waves = 64 VGPRs (−9 % on config 2). */
#ifndef RT_MIN_WAVES_PER_SIMD
#define RT_MIN_WAVES_PER_SIMD 6
#endif
#ifndef RT_MIN_WAVES_PER_SIMD_MANY
#define RT_MIN_WAVES_PER_SIMD_MANY RT_MIN_WAVES_PER_SIMD /* the > 64-model instantiation (measured at 5 as well: see DESIGN.md §4.12) */
#endif
#ifndef RT_MIN_WAVES_PER_SIMD_FLAT
#define RT_MIN_WAVES_PER_SIMD_FLAT 8
#endif
/* inner steps per vote won by phase B (lanes that reach a leaf or run out wait for the next vote) */
#ifndef RT_INNER_BURST
#define RT_INNER_BURST 3
#endif

namespace rtk {

/* Scene constants (spheres, models) are indexed wave-uniformly and never written
 * during a launch: reading them through the constant address space lets the
 * compiler use scalar loads (s_load_dwordx4/x8 into SGPRs) instead of 64
 * identical vector loads. */
#define RT_CAS __attribute__((address_space(4)))
#define RT_LDS __attribute__((address_space(3)))

/* v_min_f32 / v_max_f32 (IEEE minNum/maxNum: a NaN operand yields the other one,
 * like HLSL).  They may differ from rt_min/rt_max only in the sign of a zero
 * result, which the slab test below never observes (it only compares against 0
 * and returns a literal 0). */
__device__ __forceinline__ float hw_min(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float hw_max(float a, float b) { return __builtin_fmaxf(a, b); }

/* RayBoundingBoxDst — RC:219-231 */
__device__ __forceinline__ float box_dst(rt_f3 pos, rt_f3 invDir, const float* bmin, const float* bmax)
{
    float tminx = (bmin[0] - pos.x) * invDir.x, tmaxx = (bmax[0] - pos.x) * invDir.x;
    float tminy = (bmin[1] - pos.y) * invDir.y, tmaxy = (bmax[1] - pos.y) * invDir.y;
    float tminz = (bmin[2] - pos.z) * invDir.z, tmaxz = (bmax[2] - pos.z) * invDir.z;
    float tNear = hw_max(hw_max(hw_min(tminx, tmaxx), hw_min(tminy, tmaxy)), hw_min(tminz, tmaxz));
    float tFar = hw_min(hw_min(hw_max(tminx, tmaxx), hw_max(tminy, tmaxy)), hw_max(tminz, tmaxz));
    bool hit = tFar >= tNear && tFar > 0.0f;
    return hit ? (tNear > 0.0f ? tNear : 0.0f) : RT_INF;
}

struct SceneHit {
    float dst;      /* closest so far (result.dst) */
    int obj;        /* -1 none; [0,nSpheres) sphere; nSpheres + model index */
    int tri;        /* absolute triangle index (models) */
    float u, v, det;
    bool backface;
};

struct Stats {
    uint32_t inner, leaf, tri, sphere, model, filterViolations;
    uint32_t hotSteps; /* inner steps served by the LDS top-of-tree cache */
    uint32_t uni48, uniMaj; /* lane-steps of inner steps in which >= 48 lanes / >= 3/4 of >= 16 active lanes stand on one node */
    /* wave-level phase profile (stats launches only): how often the wave executed a phase
     * and how many lanes were active in it — lane utilisation per phase = lanes / (64 * execs) */
    uint32_t phExec[RT_N_PHASES], phLanes[RT_N_PHASES];
};
enum { PH_LOOP = 0, PH_RAYGEN, PH_SPHERES, PH_TRAVERSE_CALL, PH_MODEL, PH_INNER, PH_TRI, PH_SHADE_HIT, PH_SKY, PH_SPHERE_ROOTS, PH_GLASS, PH_REFILL };
template <bool STATS>
__device__ __forceinline__ void phase_mark(Stats& st, int ph)
{
    if (STATS) {
        const unsigned long long m = __ballot(1);
        st.phLanes[ph]++;
        if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) st.phExec[ph]++;
    }
}

/* RandomValueNormalDistribution / RandomDirection — RC:141-157 */
__device__ __forceinline__ float rand_normal(uint32_t* state)
{
    float theta = 2 * 3.1415926f * rt_random_value(state);
    float rho = rt_sqrt(-2 * rt_log(rt_random_value(state)));
    return rho * rt_cos(theta);
}
__device__ __forceinline__ rt_f3 rand_direction(uint32_t* state)
{
    /* x, y, z in draw order, one after the other (a rolled loop: the three independent
     * log/cos/sqrt chains would otherwise be interleaved and triple the live registers) */
    float x = 0.0f, y = 0.0f, z = 0.0f;
#pragma clang loop unroll(disable)
    for (int i = 0; i < 3; i++) {
        x = y;
        y = z;
        z = rand_normal(state);
    }
    return rt_normalize(rt_v3(x, y, z));
}
/* RandomPointInCircle — RC:159-164 (PI = 3.1415, RC:2) */
__device__ __forceinline__ rt_f2 rand_circle(uint32_t* state)
{
    float angle = rt_random_value(state) * 2 * 3.1415f;
    float c, s;
    rt_sincos(angle, &s, &c); /* == rt_cos(angle), rt_sin(angle): one shared range reduction */
    float r = rt_sqrt(rt_random_value(state));
    rt_f2 o = {c * r, s * r};
    return o;
}

/* GetEnvironmentLight — RC:167-183 (UseSky checked by the caller) */
template <class Args>
__device__ __forceinline__ rt_f3 environment_light(const Args& a, rt_f3 dir)
{
    float skyGradientT = rt_pow(rt_smoothstep_edges(0.0f, 0.4f, dir.y), 0.35f);
    float groundToSkyT = rt_smoothstep_edges(-0.01f, 0.0f, dir.y);
    rt_f3 skyGradient = rt_lerp3(rt_v3(1, 1, 1), rt_v3(0.08f, 0.37f, 0.73f), skyGradientT);
    float s = rt_div(1000 * 1, a.sunFocus);
    rt_f3 toSun = rt_v3(a.dirToSun[0], a.dirToSun[1], a.dirToSun[2]);
    float sun = rt_pow(rt_max(0.0f, rt_dot(dir, toSun)), s) * a.sunIntensity;
    float gate = (groundToSkyT >= 1.0f) ? 1.0f : 0.0f;
    return rt_lerp3(rt_v3(0.35f, 0.3f, 0.35f), skyGradient, groundToSkyT)
           + sun * rt_v3(a.sunColour[0], a.sunColour[1], a.sunColour[2]) * gate;
}

/* CalculateReflectance — RC:383-405 */
__device__ __forceinline__ float reflectance(rt_f3 inDir, rt_f3 normal, float iorA, float iorB)
{
    float refractRatio = rt_div(iorA, iorB);
    float cosAngleIn = -rt_dot(inDir, normal);
    float sinSqr = refractRatio * refractRatio * (1 - cosAngleIn * cosAngleIn);
    if (sinSqr >= 1) return 1.0f;
    float cosRefr = rt_sqrt(1 - sinSqr);
    float denomPerp = iorA * cosAngleIn + iorB * cosRefr;
    float denomPar = iorA * cosAngleIn + iorB * cosRefr; /* RC:392 repeats RC:391 */
    if (rt_min(denomPerp, denomPar) < 1E-8f) return 1.0f;
    float rPerp = rt_div(iorA * cosAngleIn - iorB * cosRefr, denomPerp);
    rPerp *= rPerp;
    float rPar = rt_div(iorB * cosAngleIn - iorA * cosRefr, denomPar);
    rPar *= rPar;
    return rt_div(rPerp + rPar, 2);
}
/* Refract — RC:408-417 */
__device__ __forceinline__ rt_f3 refract_dir(rt_f3 inDir, rt_f3 normal, float iorA, float iorB)
{
    float refractRatio = rt_div(iorA, iorB);
    float cosAngleIn = -rt_dot(inDir, normal);
    float sinSqr = refractRatio * refractRatio * (1 - cosAngleIn * cosAngleIn);
    if (sinSqr > 1) return rt_v3s(0.0f);
    return refractRatio * inDir + (refractRatio * cosAngleIn - rt_sqrt(1 - sinSqr)) * normal;
}

/* GetMaterialColour — RC:450-466 with mod2 RC:376-379 */
__device__ __forceinline__ float mod2f(float x, float y) { return x - y * rt_floor(rt_div(x, y)); }
__device__ __forceinline__ rt_f3 material_colour(const DMaterial& mat, rt_f3 pos, rt_f3 normal, bool isSpecular)
{
    rt_f3 col = rt_v3(mat.diffuseCol[0], mat.diffuseCol[1], mat.diffuseCol[2]);
    if (mat.flag == RT_MATERIAL_CHECKERED) {
        float px = pos.x, py = pos.z;
        if (rt_abs(normal.x) > rt_abs(normal.y)) { px = pos.z; py = pos.y; }
        if (rt_abs(normal.z) > rt_max(rt_abs(normal.x), rt_abs(normal.y))) { px = pos.x; py = pos.y; }
        px *= 1.5f;
        py *= 1.5f;
        float cx = mod2f(rt_floor(px), 2.0f);
        float cy = mod2f(rt_floor(py), 2.0f);
        if (!(cx == cy)) col = rt_v3(mat.emissionCol[0], mat.emissionCol[1], mat.emissionCol[2]);
    }
    return rt_lerp3(col, rt_v3(mat.specularCol[0], mat.specularCol[1], mat.specularCol[2]), isSpecular ? 1.0f : 0.0f);
}

/* RayTriangle — RC:188-215 on a pre-differenced triangle. Updates the closest
 * hit with the reference's strict '<' (RC:256). */
__device__ __forceinline__ void tri_test(const DTri* __restrict__ tris, int triUnit, rt_f3 pos, rt_f3 dir, bool cull,
                                         float& bestDst, int& bestTri, float& bu, float& bv, float& bdet)
{
    /* A triangle is named by the 16-byte UNIT its record starts at (rt_device.h: three units per record, the host's layout
     * decides where the runs of a leaf lie): a 32-bit byte offset from the uniform array base (SGPR base + VGPR offset
     * addressing, see the inner step) is one shift; rt_upload_scene refuses scenes whose triangle space reaches 4 GiB */
    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tris) + ((uint32_t)triUnit << 4));
    const float4 q0 = p[0], q1 = p[1], q2 = p[2];
    rt_f3 A = rt_v3(q0.x, q0.y, q0.z);
    rt_f3 edgeAB = rt_v3(q0.w, q1.x, q1.y);
    rt_f3 edgeAC = rt_v3(q1.z, q1.w, q2.x);
    rt_f3 face = rt_v3(q2.y, q2.z, q2.w);
    rt_f3 vertRayOffset = pos - A;
    rt_f3 rayOffsetPerp = rt_cross(vertRayOffset, dir);
    float determinant = -rt_dot(dir, face);
    float invDet = rt_rcp(determinant);
    float dst = rt_dot(vertRayOffset, face) * invDet;
    float u = rt_dot(edgeAC, rayOffsetPerp) * invDet;
    float v = -rt_dot(edgeAB, rayOffsetPerp) * invDet;
    float w = 1 - u - v;
    bool keep = (cull ? determinant : rt_abs(determinant)) >= 1E-8f; /* cull ? det >= eps : |det| >= eps */
    bool didHit = keep && dst > 0 && u >= 0 && v >= 0 && w >= 0;
    if (didHit && dst < bestDst) {
        bestDst = dst;
        bestTri = triUnit;
        bu = u;
        bv = v;
        bdet = determinant;
    }
}

/* ---------------------------------------------------------------------------
 * CalculateRayCollision — RC:335-374 plus the sphere buffer hooked at RC:341,
 * split so that a lane can SUSPEND in the middle of it (see rt_trace_kernel):
 *   begin_intersect : result.dst = inf, all spheres, traversal state = "before model 0"
 *   traverse        : the model loop + BVH traversal state machine, resumable
 * ------------------------------------------------------------------------- */
struct Trav {
    unsigned long long cand; /* models [0,64) still to visit (bit m), from the lockstep root filter */
    int m;          /* current model index */
    uint32_t cur;   /* current node code, or RT_CODE_NEXT_MODEL */
    int sp;         /* entries on this lane's stack */
    bool rootStep;  /* the next inner step is the model's root (already counted by the filter) */
    rt_f3 lpos, ldir, linv; /* ray in the model's local space (RC:351-353) */
    int triBase;
    bool cull;
};

template <bool STATS, bool FLAT, bool MANY>
__device__ __forceinline__ void begin_intersect(const KArgs& a, rt_f3 rpos, rt_f3 rdir, uint32_t* extBase, SceneHit& h, Trav& t, Stats& st)
{
    h.dst = RT_INF;
    h.obj = -1;
    h.tri = -1;
    h.u = h.v = h.det = 0.0f;
    h.backface = false;

    /* RaySphere — RC:289-332, in two phases so that the reference's arithmetic (and its sqrt +
     * divides) runs only for (ray, sphere) pairs that can be hits:
     *   phase 1, wave-uniform loop, sphere data in SGPRs: a CONSERVATIVE sign test of the
     *   discriminant -> per-lane bitmask.  It evaluates D = ((o-c).d)^2 - (d.d)(|o-c|^2 - r^2)
     *   in expanded form, (o.d - c.d)^2 - (d.d)(|o|^2 - 2 o.c + (|c|^2 - r^2)), with the per-ray
     *   terms hoisted and fused multiply-adds (half the operations of the reference's form), and
     *   rejects only when D' < -margin.  Rounding-error bounds (u = 2^-24, S = |o| + |c|):
     *       reference form  |disc/4 - D| <= u (d.d) (21 |o-c|^2 + 11 r^2)
     *       expanded form   |D'    - D| <= u (d.d) (22 S^2 + 8 r^2)
     *   so margin = 2^-17 (d.d) (|o|^2 + max_k(|c_k|^2 + r_k^2)) >= 128 u (d.d)(|o|^2 + |c|^2 + r^2)
     *   covers their sum (S^2 <= 2|o|^2 + 2|c|^2) with 1.5x to spare: a rejected pair has a
     *   negative discriminant in the reference's arithmetic too.  The comparison is written so
     *   that NaN / overflow keeps the candidate.  The stats build audits this (must count 0).
     *   phase 2, per lane: walk the set bits in increasing sphere order (strict '<' keeps
     *   the first of equal hits, like the reference's in-order loop) and do the reference's
     *   fp32 operations for that sphere: discriminant, then the roots.
     * 4th float of a sphere record is radius*radius, computed on upload with the same fp32 multiply. */
    const RT_CAS float* sph = (const RT_CAS float*)a.spheres;
    const RT_CAS float* sphq = (const RT_CAS float*)a.sphereQuick;
    const float qa = rt_dot(rdir, rdir);
    const float od = __builtin_fmaf(rpos.x, rdir.x, __builtin_fmaf(rpos.y, rdir.y, rpos.z * rdir.z));
    const float oo = __builtin_fmaf(rpos.x, rpos.x, __builtin_fmaf(rpos.y, rpos.y, rpos.z * rpos.z));
    const float negMargin = -(7.62939453125e-06f * qa * (oo + a.sphereBound)); /* 2^-17 */
    /* Two spheres per step, side by side in packed fp32 instructions: the sphere data sits in SGPRs, and on gfx950 a VALU
     * instruction with an SGPR source issues at the slow rate (4.5 cycles against 2.9, profiles/r03_valu_op_rates.txt) while
     * v_pk_fma/mul/add_f32 take an SGPR pair at no extra cost (5.0 cycles for both halves) — the same operations on the same
     * values per sphere, so the candidate masks do not change. */
    typedef float rt_f2v __attribute__((ext_vector_type(2)));
    const rt_f2v dx2 = {rdir.x, rdir.x}, dy2 = {rdir.y, rdir.y}, dz2 = {rdir.z, rdir.z};
    const rt_f2v ox2 = {rpos.x, rpos.x}, oy2 = {rpos.y, rpos.y}, oz2 = {rpos.z, rpos.z};
    const rt_f2v od2 = {od, od}, oo2 = {oo, oo}, qa2 = {qa, qa}, m2 = {-2.0f, -2.0f};
    for (int base = 0; base < a.nSpheres; base += 32) {
        const int n = (a.nSpheres - base) < 32 ? (a.nSpheres - base) : 32;
        uint32_t cand = 0;
        for (int k = 0; k < n; k += 2) {
            const RT_CAS float* q = sphq + 4 * (base + k); /* pair record (base + k) / 2, eight floats each */
            const rt_f2v cx = {q[0], q[1]}, cy = {q[2], q[3]}, cz = {q[4], q[5]}, kk = {q[6], q[7]};
            const rt_f2v cd = __builtin_elementwise_fma(cx, dx2, __builtin_elementwise_fma(cy, dy2, cz * dz2));
            const rt_f2v co = __builtin_elementwise_fma(cx, ox2, __builtin_elementwise_fma(cy, oy2, cz * oz2));
            const rt_f2v b = od2 - cd;
            const rt_f2v ct = __builtin_elementwise_fma(m2, co, oo2) + kk;
            const rt_f2v dq = __builtin_elementwise_fma(b, b, -(qa2 * ct));
            const bool keep0 = !(dq.x < negMargin);
            const bool keep1 = (k + 1 < n) && !(dq.y < negMargin);
            if (STATS) { /* audit against the reference's discriminant */
                for (int j = 0; j < 2 && k + j < n; j++) {
                    const int s = base + k + j;
                    rt_f3 off = rpos - rt_v3(sph[4 * s + 0], sph[4 * s + 1], sph[4 * s + 2]);
                    float qb = 2 * rt_dot(off, rdir);
                    float qc = rt_dot(off, off) - sph[4 * s + 3];
                    float disc = qb * qb - 4 * qa * qc;
                    if (disc >= 0 && !(j ? keep1 : keep0)) st.filterViolations++;
                }
            }
            cand |= ((keep0 ? 1u : 0u) << k) | ((keep1 ? 1u : 0u) << (k + 1));
        }
        while (cand) {
            phase_mark<STATS>(st, PH_SPHERE_ROOTS);
            const int k = __builtin_ctz(cand);
            cand &= cand - 1;
            const int s = base + k;
            const float4 sp = *reinterpret_cast<const float4*>(a.spheres + 4 * s);
            rt_f3 off = rpos - rt_v3(sp.x, sp.y, sp.z);
            float qb = 2 * rt_dot(off, rdir);
            float qc = rt_dot(off, off) - sp.w;
            float disc = qb * qb - 4 * qa * qc;
            if (!(disc >= 0)) continue; /* RC:304: a false positive of phase 1 ends here */
            float sq = rt_sqrt(disc);
            const float inv2a = rt_rcp(2 * qa); /* both roots share the reciprocal (rt_div) */
            float dstNear = rt_max(0.0f, (-qb - sq) * inv2a);
            float dstFar = (-qb + sq) * inv2a;
            if (dstFar >= 0) {
                bool inside = dstNear == 0;
                float d = inside ? dstFar : dstNear;
                if (d < h.dst) {
                    h.dst = d;
                    h.obj = s;
                    h.backface = inside;
                }
            }
        }
    }
    if (STATS) st.sphere += (uint32_t)a.nSpheres;

    /* Root filter, in lockstep over the models (wave-uniform loop, boxes in SGPRs).
     * In the reference every model costs a ray two matrix-vector products, three divides
     * and the root's two box tests (RC:351-353, 269-270) before most rays find they miss it.
     * A model whose root children are both missed contributes nothing (nothing is pushed,
     * RC:280-281).  Here that decision is taken CONSERVATIVELY in world space: the union of the
     * root's two child boxes was transformed to world space and inflated on upload
     * (rt_context.hip, make_filters), so one slab test with the world ray — no transform, one
     * reciprocal per segment — never rejects a model the reference would descend into; the
     * models that pass get the reference's exact arithmetic in the per-lane traversal, in
     * model order.  The filter only saves work: results and counters do not depend on it
     * (the stats build counts any exact-keep / filter-reject disagreement: must be 0). */
    unsigned long long cand = 0;
    const RT_CAS DFilter* cf = (const RT_CAS DFilter*)a.filters;
    const int nf = FLAT ? 0 : (MANY ? a.nFiltered : (a.nModels < 64 ? a.nModels : 64));
    if (nf > 0) {
        const rt_f3 winv = rt_v3(__builtin_amdgcn_rcpf(rdir.x), __builtin_amdgcn_rcpf(rdir.y), __builtin_amdgcn_rcpf(rdir.z));
        const bool farOrigin = !(rt_abs(rpos.x) <= a.filterMaxOrigin && rt_abs(rpos.y) <= a.filterMaxOrigin && rt_abs(rpos.z) <= a.filterMaxOrigin);
        /* one model: conservative world-box test, the stats build audits every rejection against the exact root step */
        auto test_model = [&](int m) -> bool {
            const RT_CAS DFilter& F = cf[m];
            bool keep = true;
            if (!F.always) {
                float bMin[3] = {F.bMin[0], F.bMin[1], F.bMin[2]}, bMax[3] = {F.bMax[0], F.bMax[1], F.bMax[2]};
                const float dB = box_dst(rpos, winv, bMin, bMax);
                /* box_dst returns +inf for a miss; '<=' (not '<') on the distance keeps it conservative */
                keep = farOrigin || (dB < RT_INF && dB <= h.dst);
            }
            if (STATS) {
                st.inner += F.innerRoot & 1u;
                if (!(F.innerRoot & 1u) && !keep) { /* a rejected leaf-root model: the reference would have run its tests */
                    st.leaf++;
                    st.tri += F.innerRoot >> 8;
                    const RT_CAS DModel& M = ((const RT_CAS DModel*)a.models)[m];
                    rt_f3 lpos = rt_v3(M.w2l[0] * rpos.x + M.w2l[1] * rpos.y + M.w2l[2] * rpos.z + M.w2l[3] * 1.0f,
                                       M.w2l[4] * rpos.x + M.w2l[5] * rpos.y + M.w2l[6] * rpos.z + M.w2l[7] * 1.0f,
                                       M.w2l[8] * rpos.x + M.w2l[9] * rpos.y + M.w2l[10] * rpos.z + M.w2l[11] * 1.0f);
                    rt_f3 ldir = rt_v3(M.w2l[0] * rdir.x + M.w2l[1] * rdir.y + M.w2l[2] * rdir.z + M.w2l[3] * 0.0f,
                                       M.w2l[4] * rdir.x + M.w2l[5] * rdir.y + M.w2l[6] * rdir.z + M.w2l[7] * 0.0f,
                                       M.w2l[8] * rdir.x + M.w2l[9] * rdir.y + M.w2l[10] * rdir.z + M.w2l[11] * 0.0f);
                    uint32_t count = (M.rootCode >> 24) & 0x7fu, start = M.rootCode & RT_CODE_MAX_INLINE_START;
                    if (count == 0) { count = a.bigLeaves[2 * start + 1]; start = a.bigLeaves[2 * start]; }
                    float best = h.dst, bu, bv, bdet;
                    int btri = -1;
                    for (uint32_t i = 0; i < count; i++) tri_test(a.tris, M.triBase + (int)start + 3 * (int)i, lpos, ldir, M.cullBackface != 0, best, btri, bu, bv, bdet);
                    if (btri >= 0) st.filterViolations++; /* audit: a rejected model must not hold a closer hit */
                }
                if ((F.innerRoot & 1u) && !keep) { /* audit the conservative filter against the exact root step */
                    const RT_CAS DModel& M = ((const RT_CAS DModel*)a.models)[m];
                    rt_f3 lpos = rt_v3(M.w2l[0] * rpos.x + M.w2l[1] * rpos.y + M.w2l[2] * rpos.z + M.w2l[3] * 1.0f,
                                       M.w2l[4] * rpos.x + M.w2l[5] * rpos.y + M.w2l[6] * rpos.z + M.w2l[7] * 1.0f,
                                       M.w2l[8] * rpos.x + M.w2l[9] * rpos.y + M.w2l[10] * rpos.z + M.w2l[11] * 1.0f);
                    rt_f3 ldir = rt_v3(M.w2l[0] * rdir.x + M.w2l[1] * rdir.y + M.w2l[2] * rdir.z + M.w2l[3] * 0.0f,
                                       M.w2l[4] * rdir.x + M.w2l[5] * rdir.y + M.w2l[6] * rdir.z + M.w2l[7] * 0.0f,
                                       M.w2l[8] * rdir.x + M.w2l[9] * rdir.y + M.w2l[10] * rdir.z + M.w2l[11] * 0.0f);
                    rt_f3 linv = rt_v3(rt_rcp(ldir.x), rt_rcp(ldir.y), rt_rcp(ldir.z));
                    const RT_CAS DPair& P = *(const RT_CAS DPair*)((const RT_CAS char*)a.pairs + ((size_t)M.rootCode << 4));
                    float pa0[3] = {P.aMin[0], P.aMin[1], P.aMin[2]}, pa1[3] = {P.aMax[0], P.aMax[1], P.aMax[2]};
                    float pb0[3] = {P.bMin[0], P.bMin[1], P.bMin[2]}, pb1[3] = {P.bMax[0], P.bMax[1], P.bMax[2]};
                    if (box_dst(lpos, linv, pa0, pa1) < h.dst || box_dst(lpos, linv, pb0, pb1) < h.dst) st.filterViolations++;
                }
            }
            return keep;
        };
        unsigned long long scalarCand = 0;
        if (!MANY && STATS) { /* up to 64 models: the one-model-per-step loop for the exact counters and the audits ... */
            for (int m = 0; m < nf; m++) scalarCand |= (test_model(m) ? 1ull : 0ull) << m;
        }
        if (!MANY) { /* ... and the packed loop that SHIPS decides the candidates in both builds (the stats build audits it too) */
            /* The shipped form of the same loop, TWO models per step (round 4): the boxes sit in SGPRs, and a VALU instruction
             * with an SGPR source issues at the slow rate on gfx950 while v_pk_add/mul_f32 take an SGPR PAIR for the price of one
             * (the two-spheres-per-step pre-test above, profiles/r03_valu_op_rates.txt).  The pair records (rt_context.hip,
             * append_filter_pairs) hold the two models' boxes side by side — minx0 minx1 miny0 miny1 ... — so that one scalar load
             * fills the pairs and the twelve (b - o) and twelve (.) * (1 / d) of two slab tests are twelve packed instructions.
             * Same operations on the same values per model; the masks are a SUPERSET of the one-model-per-step loop's (a NaN or an
             * infinite tNear keeps the model here where box_dst's comparisons reject it: conservative) — the stats build counts any
             * model that loop keeps and this one drops as a filter violation (must be 0). */
            typedef float rt_f2v __attribute__((ext_vector_type(2)));
            const rt_f2v px = {rpos.x, rpos.x}, py = {rpos.y, rpos.y}, pz = {rpos.z, rpos.z};
            const rt_f2v ix = {winv.x, winv.x}, iy = {winv.y, winv.y}, iz = {winv.z, winv.z};
            const RT_CAS float* fp = (const RT_CAS float*)a.filterPairs;
            /* (the halves of a packed product reach fminf / fmaxf as "maybe a signalling NaN" and would each be canonicalised
             * first — six more instructions per model than the packing saves — so the slab test's min / max are written as the
             * instructions themselves; products of finite or infinite operands are never signalling NaNs) */
            auto keep_of = [&](float t0x, float t1x, float t0y, float t1y, float t0z, float t1z, uint32_t always) -> bool {
                float nx, ny, nz, fx, fy, fz, tNear, tFar;
                asm("v_min_f32 %0, %1, %2" : "=v"(nx) : "v"(t0x), "v"(t1x));
                asm("v_min_f32 %0, %1, %2" : "=v"(ny) : "v"(t0y), "v"(t1y));
                asm("v_min_f32 %0, %1, %2" : "=v"(nz) : "v"(t0z), "v"(t1z));
                asm("v_max_f32 %0, %1, %2" : "=v"(fx) : "v"(t0x), "v"(t1x));
                asm("v_max_f32 %0, %1, %2" : "=v"(fy) : "v"(t0y), "v"(t1y));
                asm("v_max_f32 %0, %1, %2" : "=v"(fz) : "v"(t0z), "v"(t1z));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tNear) : "v"(nx), "v"(ny), "v"(nz));
                asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tFar) : "v"(fx), "v"(fy), "v"(fz));
                const bool hit = tFar >= tNear && tFar > 0.0f;
                const bool near = hit && (tNear > 0.0f ? tNear : 0.0f) <= h.dst; /* box_dst(...) < inf && box_dst(...) <= h.dst */
                return near | farOrigin | (always != 0u);
            };
            for (int m = 0; m < nf; m += 2) {
                const RT_CAS float* q = fp + 8 * m; /* pair record m / 2, sixteen dwords */
                const rt_f2v t0x = (rt_f2v{q[0], q[1]} - px) * ix, t0y = (rt_f2v{q[2], q[3]} - py) * iy, t0z = (rt_f2v{q[4], q[5]} - pz) * iz;
                const rt_f2v t1x = (rt_f2v{q[6], q[7]} - px) * ix, t1y = (rt_f2v{q[8], q[9]} - py) * iy, t1z = (rt_f2v{q[10], q[11]} - pz) * iz;
                const bool keep0 = keep_of(t0x.x, t1x.x, t0y.x, t1y.x, t0z.x, t1z.x, __float_as_uint(q[12]));
                const bool keep1 = (m + 1 < nf) && keep_of(t0x.y, t1x.y, t0y.y, t1y.y, t0z.y, t1z.y, __float_as_uint(q[13]));
                cand |= ((keep0 ? 1ull : 0ull) << m) | ((keep1 ? 1ull : 0ull) << (m + 1));
            }
            if (STATS && (scalarCand & ~cand)) st.filterViolations += (uint32_t)__popcll(scalarCand & ~cand);
        } else {
            /* more than 64 models: chunk boxes first (a chunk no lane of the wave hits costs one box test instead of
             * 16), candidates beyond model 62 go to the lane's LDS extension words */
            for (int w = 0; w <= a.extWords; w++) extBase[w * RT_WAVE] = 0u;
            const RT_CAS DChunk* cc = (const RT_CAS DChunk*)a.chunks;
            for (int c = 0; c < a.nChunks; c++) {
                const RT_CAS DChunk& C = cc[c];
                bool hitC = true;
                if (!C.always) {
                    float bMin[3] = {C.bMin[0], C.bMin[1], C.bMin[2]}, bMax[3] = {C.bMax[0], C.bMax[1], C.bMax[2]};
                    const float dB = box_dst(rpos, winv, bMin, bMax);
                    hitC = farOrigin || (dB < RT_INF && dB <= h.dst);
                }
                /* (the stats build walks every chunk so that the per-model audit sees every rejection) */
                if (!STATS && __ballot(hitC) == 0ull) continue;
                const int n = (int)C.count;
                for (int k = 0; k < n; k++) {
                    const int m = (int)C.members[k];
                    const bool keep = test_model(m);
                    if (STATS && keep && !hitC) st.filterViolations++; /* the chunk box must contain its members' boxes */
                    if (m < 63) {
                        cand |= (keep ? 1ull : 0ull) << m;
                    } else if (keep) {
                        const int w = (m - 63) >> 5;
                        extBase[(1 + w) * RT_WAVE] |= 1u << ((m - 63) & 31);
                        extBase[0] |= 1u << w;
                        cand |= 1ull << 63;
                    }
                }
            }
        }
    }
    if (STATS) st.model += (uint32_t)a.nModels;

    t.cand = cand;
    t.m = -1;
    t.cur = RT_CODE_NEXT_MODEL;
    t.sp = 0;
    t.rootStep = false;
    t.lpos = t.ldir = t.linv = rt_v3s(0.0f);
    t.triBase = 0;
    t.cull = true;
}

/* Model loop (RC:347-371) and per-model BVH traversal (RayTriangleBVH, RC:234-287) as
 * ONE per-lane state machine.  The reference nests them, which on a SIMD machine makes
 * every lane wait, model after model, for the slowest traversal in the wave.  Here each
 * lane walks its own (model, node) sequence — the order of box tests, triangle tests
 * and strict-'<' updates of a given ray is exactly the reference's — and the wave only
 * re-converges by kind of work ("while-while"):
 *   A  lanes that finished a model load the next one and transform the ray (RC:351-353);
 *   B  lanes at an inner node step down until they reach a leaf or run out of nodes;
 *   C  lanes at a leaf test its triangles.
 * h.dst is result.dst carried from model to model (rayLength = result.dst, RC:359).
 *
 * Returns true when this lane has visited every model.  With SUSPEND the wave leaves
 * the loop as soon as at most RT_SUSPEND_NUM/RT_SUSPEND_DEN of the lanes that entered are still traversing;
 * the stragglers keep their state in `t`/`h`/LDS and resume at the next call.  The loop has a
 * single, wave-uniform exit (finished lanes park in RT_CODE_DONE instead of leaving one by
 * one), which keeps the loop-carried state in one set of registers. */
template <bool STATS, bool SUSPEND, bool MANY, bool HOT = false>
__device__ __forceinline__ bool traverse(const KArgs& a, rt_f3 rpos, rt_f3 rdir, uint32_t* stackBase, uint32_t* extBase, SceneHit& h, Trav& t, Stats& st,
                                         const RT_LDS char* hotLds, const uint32_t hotUnits)
{
    const DModel* __restrict__ models = a.models;
    const DPair* __restrict__ pairs = a.pairs;
    const DTri* __restrict__ tris = a.tris;
    /* the loop runs while more than keepAbove lanes are still traversing: entered * a.suspendNum / RT_SUSPEND_DEN (3/8 unless the
     * launch tuner found 4/8 faster for this scene) */
    const int enteredNum = SUSPEND ? __popcll(__ballot(1)) * a.suspendNum : 0;
    phase_mark<STATS>(st, PH_TRAVERSE_CALL);
    /* One kind of work per iteration, chosen for the whole wave: the kind most lanes are
     * waiting for (wave-uniform branch, so only that code is issued).  A lane deep inside
     * a big mesh no longer drags 60 idle lanes through its private box tests: it advances
     * whenever "inner step" is the majority need (every model's root step is one), and
     * otherwise waits, keeping its state.
     * The vote for the next iteration is taken at the bottom of the loop (do-while form, the
     * only exit is wave-uniform): nobody left, or — SUSPEND — at most RT_SUSPEND_NUM/RT_SUSPEND_DEN
     * of the lanes that entered are still traversing. */
    bool atNext, atLeaf, atInner;
    int nA, nB, nC;
    /* WATCHDOG (round 6).  The reference walks nodeOffset + startIndex with no bounds or cycle check (RC:245-252, 264-267); here
     * rt_upload_scene refuses every buffer that could make a walk endless (cycles, depth > 32, indices out of range), so this never fires
     * — but a traversal that does not end would occupy a shared GPU until the driver resets it, and validation is code like any other.
     * Every iteration of the loop below advances at least one lane by one step, and a lane has at most travSteps steps in a scene
     * (every model's pairs and leaves once): more than 64 x travSteps iterations in ONE call cannot happen on validated buffers.  A wave
     * that gets there ends its lanes' walks where they are and raises counter slot 7: rt_get_counters / rt_read_* then FAIL (the image
     * is wrong by then, the device is not lost).  Wave-uniform: one scalar add and compare per vote. */
    uint32_t watchdog = 0;
#define RT_TRAV_VOTE()                                                                                         \
    do {                                                                                                       \
        /* a lane between models that has no model left is done: no more demand for phase A */              \
        if (t.cur == RT_CODE_NEXT_MODEL && !t.cand && (MANY ? (t.m < a.nFiltered - 1 ? a.nFiltered : t.m + 1) : (t.m < 63 ? 64 : t.m + 1)) >= a.nModels) t.cur = RT_CODE_DONE; \
        atNext = t.cur == RT_CODE_NEXT_MODEL;                                                                  \
        atLeaf = (t.cur & RT_CODE_LEAF) != 0;                                                                  \
        atInner = t.cur < RT_CODE_DONE; /* leaf codes have bit 31 set */                                       \
        nA = __popcll(__ballot(atNext)), nB = __popcll(__ballot(atInner)), nC = __popcll(__ballot(atLeaf));    \
    } while (0)
    RT_TRAV_VOTE();
    if ((nA + nB + nC) * RT_SUSPEND_DEN > enteredNum) do {
        if (nA >= nB && nA >= nC) {
            if (atNext) { /* ---- A: next model, RC:349-355 */
                /* the filtered candidates in model order (register mask, then the LDS extension words of scenes
                 * with more than 64 models), then any model beyond the filtered range in order */
                const unsigned long long regBits = MANY ? (t.cand & 0x7fffffffffffffffull) : t.cand;
                if (regBits) {
                    t.m = __ffsll((long long)regBits) - 1;
                    t.cand &= t.cand - 1;
                } else if (MANY && t.cand) { /* bit 63 alone: the next candidate is in the extension */
                    uint32_t summary = extBase[0];
                    const int w = __ffs((int)summary) - 1;
                    uint32_t word = extBase[(1 + w) * RT_WAVE];
                    t.m = 63 + 32 * w + (__ffs((int)word) - 1);
                    word &= word - 1;
                    extBase[(1 + w) * RT_WAVE] = word;
                    if (!word) {
                        summary &= summary - 1;
                        extBase[0] = summary;
                        if (!summary) t.cand = 0;
                    }
                } else {
                    t.m = MANY ? (t.m < a.nFiltered - 1 ? a.nFiltered : t.m + 1) : (t.m < 63 ? 64 : t.m + 1);
                    if (STATS && !(models[t.m].rootCode & RT_CODE_LEAF)) st.inner++;
                }
                t.rootStep = true;
                phase_mark<STATS>(st, PH_MODEL);
                const float4* q = reinterpret_cast<const float4*>(models + t.m);
                const float4 r0 = q[0], r1 = q[1], r2 = q[2];
                const float4 tail = q[6];
                t.lpos = rt_v3(r0.x * rpos.x + r0.y * rpos.y + r0.z * rpos.z + r0.w * 1.0f,
                               r1.x * rpos.x + r1.y * rpos.y + r1.z * rpos.z + r1.w * 1.0f,
                               r2.x * rpos.x + r2.y * rpos.y + r2.z * rpos.z + r2.w * 1.0f);
                t.ldir = rt_v3(r0.x * rdir.x + r0.y * rdir.y + r0.z * rdir.z + r0.w * 0.0f,
                               r1.x * rdir.x + r1.y * rdir.y + r1.z * rdir.z + r1.w * 0.0f,
                               r2.x * rdir.x + r2.y * rdir.y + r2.z * rdir.z + r2.w * 0.0f);
                t.cur = __float_as_uint(tail.x);
                t.triBase = (int)__float_as_uint(tail.y);
                t.cull = __float_as_uint(tail.z) != 0;
                t.sp = 0;
                /* invDir (RC:353) is only read by box tests: a mesh whose root is a leaf has none */
                if (!(t.cur & RT_CODE_LEAF)) t.linv = rt_v3(rt_rcp(t.ldir.x), rt_rcp(t.ldir.y), rt_rcp(t.ldir.z));
            }
        } else if (nB >= nC) {
            /* a short burst of inner steps per vote won (lanes that reach a leaf or run out of nodes
             * wait for the next vote): 3 measured best — 1 pays a vote per step, "until no lane
             * is at an inner node" (the classic while-while) idles most lanes most of the time */
#pragma clang loop unroll(disable)
            for (int burst = 0; burst < RT_INNER_BURST; burst++)
            if (t.cur < RT_CODE_DONE) { /* ---- B: one inner node, RC:262-282 */
                if (STATS && !t.rootStep) st.inner++;
                t.rootStep = false;
                phase_mark<STATS>(st, PH_INNER);
                /* an inner code is the 16-byte unit the pair record starts at (rt_device.h; the host's layout decides where
                 * that is): a 32-bit byte offset from the (wave-uniform) array base — the load takes "SGPR base + VGPR offset",
                 * and the 64-bit shift and add of a full pointer (two slow-class VALU instructions per step on gfx950) become
                 * one fast 32-bit shift; rt_upload_scene refuses scenes whose pair space reaches 4 GiB.
                 * TOP-OF-TREE CACHE (round 6): the layout puts the records a ray is most likely to need — the top of every tree,
                 * ranked by world-space surface area (rt_layout.h, hot set) — at units [0, hotUnits) of the pair space, and the
                 * workgroup copied exactly those into LDS when it started (trace_body).  The BVH kernels are bound by the
                 * vector-memory path (0.83-0.92 L1 accesses per clock per CU, four per ray and step; profiles/r05_memory_path.txt)
                 * while the LDS pipe idles: a step whose node is in the cache costs the TA / L1 nothing.  LDS layout
                 * [quarter][record] (quarter q of record r at byte (q * nHot + r) * 16): the 16 lanes of a ds_read_b128 pass
                 * spread over 16 bank groups by r instead of 4.  Same bytes either way: RC:262-282 to the bit. */
                const uint32_t cur = t.cur;
                typedef float rt_v4f __attribute__((ext_vector_type(4))); /* (a plain vector type: HIP's float4 class cannot be read through an LDS-qualified pointer) */
                rt_v4f q0, q1, q2, q3;
                if (HOT && cur < hotUnits) { /* HOT: the instantiation launched as multi-wave workgroups with the cache; without it the step is round 5's */
                    /* (LDS-typed pointers: with generic ones the optimiser merges the two branches into one select of addresses and
                     * nine flat loads — measured 35-88 % slower than no cache at all) */
                    const RT_LDS char* l = hotLds + (cur << 2);
                    const uint32_t qs = hotUnits << 2; /* bytes between the quarters: nHot * 16 */
                    q0 = *reinterpret_cast<const RT_LDS rt_v4f*>(l);
                    q1 = *reinterpret_cast<const RT_LDS rt_v4f*>(l + qs);
                    q2 = *reinterpret_cast<const RT_LDS rt_v4f*>(l + 2 * qs);
                    q3 = *reinterpret_cast<const RT_LDS rt_v4f*>(l + 3 * qs);
                    if (STATS) st.hotSteps++;
                } else {
                    const rt_v4f* q = reinterpret_cast<const rt_v4f*>(reinterpret_cast<const char*>(pairs) + (uint32_t)(cur << 4));
                    q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3]; /* (loading only the 8 bytes of q3 that are used changes nothing: 9.9) */
                }
                if (STATS) { /* how often most of the wave stands on ONE node (the case for a scalar top-of-tree path) */
                    const unsigned long long act = __ballot(true);
                    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
                    unsigned long long same = __ballot(cur == c0);
                    if (__popcll(same) < 48 && (act & ~same)) { /* second try: the node of the first lane that is elsewhere */
                        const int l2 = __ffsll((long long)(act & ~same)) - 1;
                        const uint32_t c1 = (uint32_t)__shfl((int)cur, l2, 64);
                        same = __ballot(cur == c1);
                    }
                    if (__popcll(same) >= 48) st.uni48++;
                    if (__popcll(act) >= 16 && 4 * __popcll(same) >= 3 * __popcll(act)) st.uniMaj++;
                }
                float aMin[3] = {q0.x, q0.y, q0.z}, aMax[3] = {q0.w, q1.x, q1.y};
                float bMin[3] = {q1.z, q1.w, q2.x}, bMax[3] = {q2.y, q2.z, q2.w};
                uint32_t codeA = __float_as_uint(q3.x), codeB = __float_as_uint(q3.y);
                float dstA = box_dst(t.lpos, t.linv, aMin, aMax);
                float dstB = box_dst(t.lpos, t.linv, bMin, bMax);
                bool isNearestA = dstA <= dstB;
                float dstNear = isNearestA ? dstA : dstB;
                float dstFar = isNearestA ? dstB : dstA;
                uint32_t codeNear = isNearestA ? codeA : codeB;
                uint32_t codeFar = isNearestA ? codeB : codeA;
                /* RC:280-281: push far, then near; the next pop is the near child, so it stays
                 * in `cur`.  dstNear <= dstFar, so far-pushed implies near-pushed. */
                if (dstNear < h.dst) {
                    if (dstFar < h.dst) { stackBase[t.sp * RT_WAVE] = codeFar; t.sp++; }
                    t.cur = codeNear;
                } else if (t.sp == 0) {
                    t.cur = RT_CODE_NEXT_MODEL;
                } else {
                    t.cur = stackBase[(--t.sp) * RT_WAVE];
                }
            }
        } else if (atLeaf) { /* ---- C: one leaf, RC:248-261 */
            uint32_t count = (t.cur >> 24) & 0x7fu;
            uint32_t start = t.cur & RT_CODE_MAX_INLINE_START;
            if (count == 0) { /* indirect (oversized leaf) */
                count = a.bigLeaves[2 * start + 1];
                start = a.bigLeaves[2 * start];
            }
            if (STATS) { st.leaf++; st.tri += count; }
            const int first = t.triBase + (int)start;
            for (uint32_t i = 0; i < count; i++) {
                phase_mark<STATS>(st, PH_TRI);
                const float before = h.dst;
                tri_test(tris, first + 3 * (int)i, t.lpos, t.ldir, t.cull, h.dst, h.tri, h.u, h.v, h.det);
                if (h.dst < before) { /* RC:362-369 (an update strictly lowers dst) */
                    h.obj = a.nSpheres + t.m;
                    h.backface = h.det < 0;
                }
            }
            if (t.sp == 0) t.cur = RT_CODE_NEXT_MODEL;
            else t.cur = stackBase[(--t.sp) * RT_WAVE];
        }
        if (++watchdog > a.travLimit) { /* wave-uniform */
            t.cur = RT_CODE_DONE; t.sp = 0; t.cand = 0;
            if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(a.counters + 7, 1ull);
        }
        RT_TRAV_VOTE();
    } while ((nA + nB + nC) * RT_SUSPEND_DEN > enteredNum);
#undef RT_TRAV_VOTE
    return t.cur == RT_CODE_DONE;
}

/* Scenes in which every model's root is a leaf (quads, small meshes built with
 * Quality.Disabled — BASELINE configs 1 and 2): there is no tree to walk, so the
 * reference's nested form is already convergent.  Wave-uniform model loop, matrices in
 * SGPRs, the root leaf's triangles tested in order; no stack, no suspension. */
template <bool STATS>
__device__ __forceinline__ void traverse_flat(const KArgs& a, rt_f3 rpos, rt_f3 rdir, SceneHit& h, Stats& st)
{
    const RT_CAS DModel* cm = (const RT_CAS DModel*)a.models;
    const DTri* __restrict__ tris = a.tris;
    for (int m = 0; m < a.nModels; m++) {
        const RT_CAS DModel& M = cm[m];
        rt_f3 lpos = rt_v3(M.w2l[0] * rpos.x + M.w2l[1] * rpos.y + M.w2l[2] * rpos.z + M.w2l[3] * 1.0f,
                           M.w2l[4] * rpos.x + M.w2l[5] * rpos.y + M.w2l[6] * rpos.z + M.w2l[7] * 1.0f,
                           M.w2l[8] * rpos.x + M.w2l[9] * rpos.y + M.w2l[10] * rpos.z + M.w2l[11] * 1.0f);
        rt_f3 ldir = rt_v3(M.w2l[0] * rdir.x + M.w2l[1] * rdir.y + M.w2l[2] * rdir.z + M.w2l[3] * 0.0f,
                           M.w2l[4] * rdir.x + M.w2l[5] * rdir.y + M.w2l[6] * rdir.z + M.w2l[7] * 0.0f,
                           M.w2l[8] * rdir.x + M.w2l[9] * rdir.y + M.w2l[10] * rdir.z + M.w2l[11] * 0.0f);
        const uint32_t code = M.rootCode;
        uint32_t count = (code >> 24) & 0x7fu;
        uint32_t start = code & RT_CODE_MAX_INLINE_START;
        if (count == 0) {
            count = a.bigLeaves[2 * start + 1];
            start = a.bigLeaves[2 * start];
        }
        if (STATS) { st.leaf++; st.tri += count; }
        const int first = M.triBase + (int)start;
        const bool cull = M.cullBackface != 0;
        for (uint32_t i = 0; i < count; i++) {
            phase_mark<STATS>(st, PH_TRI);
            const float before = h.dst;
            tri_test(tris, first + 3 * (int)i, lpos, ldir, cull, h.dst, h.tri, h.u, h.v, h.det);
            if (h.dst < before) {
                h.obj = a.nSpheres + m;
                h.backface = h.det < 0;
            }
        }
    }
}

/* Run-to-completion form (debug hook). */
template <bool STATS>
__device__ __forceinline__ void intersect_scene(const KArgs& a, rt_f3 rpos, rt_f3 rdir, uint32_t* stackBase, uint32_t* extBase, SceneHit& h, Stats& st)
{
    Trav t;
    if (a.nChunks) { /* wave-uniform: more than 64 models */
        begin_intersect<STATS, false, true>(a, rpos, rdir, extBase, h, t, st);
        traverse<STATS, false, true>(a, rpos, rdir, stackBase, extBase, h, t, st, (const RT_LDS char*)nullptr, 0u);
    } else {
        begin_intersect<STATS, false, false>(a, rpos, rdir, extBase, h, t, st);
        traverse<STATS, false, false>(a, rpos, rdir, stackBase, extBase, h, t, st, (const RT_LDS char*)nullptr, 0u);
    }
}

/* Position and world normal of the winning hit: RC:319-320 (sphere) or RC:208-209 +
 * RC:367-368 (triangle; quirk Q10: localToWorld, not its inverse transpose). */
__device__ __forceinline__ void resolve_hit(const KArgs& a, rt_f3 rpos, rt_f3 rdir, const SceneHit& h, rt_f3& hpos, rt_f3& normal)
{
    hpos = rpos + rdir * h.dst;
    if (h.obj < a.nSpheres) {
        const float* sp4 = a.spheres + 4 * h.obj;
        rt_f3 centre = rt_v3(sp4[0], sp4[1], sp4[2]);
        normal = rt_normalize(hpos - centre) * (h.backface ? -1.0f : 1.0f);
    } else {
        const DModel& M = a.models[h.obj - a.nSpheres];
        /* the winner's vertex normals: 12 bytes per unit of the triangle space, i.e. the 36-byte record of the triangle at unit h.tri */
        const DTriN& N = *reinterpret_cast<const DTriN*>(reinterpret_cast<const char*>(a.norms) + (uint32_t)h.tri * 12u);
        float w = 1 - h.u - h.v;
        rt_f3 sn = rt_normalize(rt_v3(N.n[0], N.n[1], N.n[2]) * w + rt_v3(N.n[3], N.n[4], N.n[5]) * h.u
                                + rt_v3(N.n[6], N.n[7], N.n[8]) * h.v);
        rt_f3 ln = sn * rt_sign(h.det);
        normal = rt_normalize(rt_v3(M.l2w[0] * ln.x + M.l2w[1] * ln.y + M.l2w[2] * ln.z + M.l2w[3] * 0.0f,
                                    M.l2w[4] * ln.x + M.l2w[5] * ln.y + M.l2w[6] * ln.z + M.l2w[7] * 0.0f,
                                    M.l2w[8] * ln.x + M.l2w[9] * ln.y + M.l2w[10] * ln.z + M.l2w[11] * 0.0f));
    }
}

/* Kernel arguments that only the per-pixel bookkeeping reads (tile queue, camera, targets,
 * sky) are fetched where they are used, through a pointer the optimiser cannot see through:
 * read as plain `a.field` they would all be loaded once up front and then held in — and
 * spilled from — scalar registers across the traversal and shading code. */
__device__ __forceinline__ const RT_CAS KArgs& cold_args()
{
    const RT_CAS void* p = (const RT_CAS void*)__builtin_amdgcn_kernarg_segment_ptr(); /* KArgs is the only kernel argument */
    asm volatile("" : "+s"(p));
    return *(const RT_CAS KArgs*)p;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

/* ---------------------------------------------------------------------------
 * The FLAT variant's CHAIN POOL (round 6): active-ray compaction across the waves of a workgroup.
 *
 * After the intersection every lane's chain wants ONE of two exclusive phases — `sky` (a miss, RC:488-492: two pow, ends the path)
 * or `shade` (a hit, RC:494-538) — and a wave that runs both for half of its lanes each is where the headline kernel loses its lanes
 * (lane utilisation 0.50 / 0.63 in the two phases, 57 % of the instructions of an iteration).  A pixel's samples and bounces are one
 * serial RNG chain (quirk Q13), but WHICH lane of WHICH wave runs the next link is free: the chain is 32 dwords of state.  The waves of
 * a workgroup therefore share two bounded queues in LDS — chains waiting for `sky`, chains waiting for `shade` — and at this point
 * of every iteration a wave
 *   - picks as its target the phase with the larger demand (its own lanes + the queue's chains),
 *   - DEPOSITS the chains of its other lanes (as far as that queue has room), and
 *   - WITHDRAWS chains of the target phase into every lane that is empty now (just deposited, or out of pixels),
 * then runs the phases as before — for (nearly) all 64 lanes the target phase, the other one only for lanes the full queue kept.
 * An emptied lane takes the next pixel at the top of the loop like any idle lane, so a workgroup carries up to 2 x poolCells more chains
 * than it has lanes.  Same arithmetic per chain, same order of a pixel's samples, frames staged and added in frame order as before:
 * same bits (tools/sched_sim_pool.py is the model that said -12 ... -16 % instructions before this was written).
 *
 * Queues: bounded multi-producer / multi-consumer rings with a sequence word per cell.  A wave reserves n cells with ONE compare-and-swap
 * on the queue's tail (deposit) or head (withdraw) by its first lane; lane r then owns position p = base + r, cell p mod C:
 *   deposit : wait until seq[cell] == p (the reader of the previous lap is done), write the payload, seq[cell] = p + 1 (release)
 *   withdraw: wait until seq[cell] == p + 1 (the writer is done), read the payload, seq[cell] = p + C (release: free for the next lap)
 * Every wait is for a wave that is in the middle of straight-line code of the same kind on an EARLIER position, so waits cannot form a
 * cycle; a wave never exits while it holds a chain or a queue is non-empty (trace_body), so every deposited chain is withdrawn by a
 * live wave.  A wait that exceeds KArgs::poolSpinLimit (65,536) polls raises the watchdog counter (slot 7: rt_get_counters / rt_read_* then FAIL)
 * and goes on — a broken pool must cost a wrong image that says so, never a hung device.
 * ------------------------------------------------------------------------- */
#ifndef RT_POOL_ATTEMPTS
#define RT_POOL_ATTEMPTS 3 /* compare-and-swap attempts per exchange (config 2: 1 = -8.8 %, 2 = -10.0 %, 3 = -10.2 % frame time against no pool) */
#endif
#define RT_POOL_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define RT_POOL_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define RT_RFL(v) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(v)))
template <bool STATS>
__device__ __forceinline__ void pool_exchange(const KArgs& a, uint32_t* const pool, const int lane, uint32_t* const pxu, float4* const cold,
                                              bool& laneDone, bool& pathActive, bool& inTrav, rt_f3& rpos, rt_f3& rdir, rt_f3& transmittance, rt_f3& pathLight,
                                              uint32_t& rng, int& bounce, SceneHit& h, const uint32_t segments, Stats& st)
{
    /* (the geometry is a compile-time constant: every mask and offset below is an immediate.  The exchange runs once per wave iteration and is
     * mostly scalar bookkeeping — the scalar unit is shared by the CU's waves, so its instruction count is what the exchange costs) */
    constexpr uint32_t C = RT_POOL_CELLS;
    const bool wantSky = !laneDone && inTrav && h.obj < 0;
    const bool wantShade = !laneDone && inTrav && h.obj >= 0;
    const unsigned long long mS = __ballot(wantSky), mH = __ballot(wantShade);
    const unsigned long long mE = __ballot(laneDone);
    const int nS = __popcll(mS), nH = __popcll(mH);
    if (mE == 0ull && (nS == 0 || nH == 0)) return; /* every lane holds a chain and they all want the same phase: nothing to hand over or take */
    /* Header, queue 0 = sky, queue 1 = shade: dwords (head0, tail1, head1, tail0) — the two counters ONE exchange moves lie in one 64-bit word:
     * target shade = deposit into sky (tail0) + withdraw from shade (head1) = dwords 2..3; target sky = (head0, tail1) = dwords 0..1.
     * ONE compare-and-swap of that word by the wave's first lane reserves both ranges; if another wave moved either counter since the header
     * was read the swap fails and this iteration goes without an exchange (the phases below handle any mix of lanes). */
    bool tgtShade = true;
    int nDep = 0, nW = 0;
    uint32_t wBase = 0u, dBase = 0u;
    bool reserved = false;
#pragma clang loop unroll(disable)
    for (int attempt = 0; attempt < RT_POOL_ATTEMPTS && !reserved; attempt++) {
        asm volatile("" ::: "memory");
        const uint4 hdr = *reinterpret_cast<const uint4*>(pool); /* one 16-byte read, the same address in every lane */
        const uint32_t head0 = RT_RFL(hdr.x), tail1 = RT_RFL(hdr.y), head1 = RT_RFL(hdr.z), tail0 = RT_RFL(hdr.w);
        const int qS = (int)(tail0 - head0), qH = (int)(tail1 - head1);
        tgtShade = nH + qH >= nS + qS;
        /* a stale head only underestimates the room, a stale tail only the chains to take */
        const int nOther = tgtShade ? nS : nH;
        const int room = (int)C - (tgtShade ? qS : qH);
        nDep = nOther < room ? nOther : room;
        nDep = nDep > 0 ? nDep : 0;
        const int empties = __popcll(mE) + nDep;
        const int avail = tgtShade ? qH : qS;
        nW = empties < avail ? empties : avail;
        nW = nW > 0 ? nW : 0;
        if ((nDep | nW) == 0) return;
        wBase = tgtShade ? head1 : head0;
        dBase = tgtShade ? tail0 : tail1;
        const unsigned long long expect = (unsigned long long)wBase | ((unsigned long long)dBase << 32);
        const unsigned long long want = (unsigned long long)(wBase + (uint32_t)nW) | ((unsigned long long)(dBase + (uint32_t)nDep) << 32);
        unsigned long long old = expect;
        if (lane == 0) old = atomicCAS(reinterpret_cast<unsigned long long*>(pool + (tgtShade ? 2 : 0)), expect, want);
        reserved = RT_RFL((uint32_t)old) == wBase && RT_RFL((uint32_t)(old >> 32)) == dBase;
    }
    if (!reserved) return;
    uint32_t* const seq = pool + RT_POOL_HEADER_DWORDS;
    float4* const payload = reinterpret_cast<float4*>(pool + RT_POOL_HEADER_DWORDS + 2u * C);
    const uint32_t qo = tgtShade ? 0u : 1u, qt = 1u - qo;
    /* who does what: the first nDep lanes of the other phase deposit; the first nW lanes that are empty afterwards withdraw (a lane may do both);
     * ranks through v_mbcnt (bits of the mask below this lane) */
    const unsigned long long mOther = tgtShade ? mS : mH;
    const uint32_t rankD = __builtin_amdgcn_mbcnt_hi((uint32_t)(mOther >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mOther, 0u));
    const bool deposits = (tgtShade ? wantSky : wantShade) && (int)rankD < nDep;
    const unsigned long long mEmptyAfter = mE | __ballot(deposits);
    const uint32_t rankW = __builtin_amdgcn_mbcnt_hi((uint32_t)(mEmptyAfter >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mEmptyAfter, 0u));
    const bool withdraws = (laneDone || deposits) && (int)rankW < nW;
    const uint32_t posD = dBase + rankD, cellD = posD & (C - 1u);
    const uint32_t posW = wBase + rankW, cellW = posW & (C - 1u);
    uint32_t* const spD = seq + qo * C + cellD;
    uint32_t* const spW = seq + qt * C + cellW;
    /* both cells' sequence words in one go (every lane reads both; lanes without the role compare against what they read): the deposit's must say
     * "free for posD" (the reader of the previous lap is done), the withdrawal's "written for posW".  Almost always true at once. */
    bool gaveUp = false;
    {
        uint32_t sD = RT_POOL_LOAD(spD), sW = RT_POOL_LOAD(spW);
        /* (a.poolSpinLimit's top bit = the test hook RT_POOL_FAULT: withdrawers poll to the limit as if their cell were not written) */
        const bool fault = (int)a.poolSpinLimit < 0;
        if (__ballot((deposits && sD != posD) || (withdraws && sW != posW + 1u)) || fault) { /* rare: a writer / reader of the cell is in the middle of its copy */
            uint32_t spins = 0u;
            bool waiting = true;
            while (waiting) {
                __builtin_amdgcn_s_sleep(1);
                sD = RT_POOL_LOAD(spD);
                sW = RT_POOL_LOAD(spW);
                const bool notYet = (deposits && sD != posD) || (withdraws && sW != posW + 1u);
                waiting = notYet || (fault && withdraws);
                if (++spins > (a.poolSpinLimit & 0x7fffffffu)) {
                    /* A broken pool costs a wrong image that says so (watchdog counter), never a hung device — and never a wild store: a lane whose cell
                     * did not come DROPS its chain (nothing is written over a cell another wave may still read, nothing is read from a cell nobody
                     * wrote) and goes on as an empty lane; whoever waits for that position gives up the same way. */
                    if (waiting) atomicAdd(a.counters + 7, 1ull);
                    gaveUp = notYet;
                    waiting = false;
                }
            }
        }
    }
    asm volatile("" ::: "memory"); /* the payload accesses below stay below the sequence reads (LDS executes a wave's instructions in order) */
    if (deposits && gaveUp) { laneDone = true; inTrav = false; pathActive = false; }
    if (deposits && !gaveUp) {
        float4* const q = payload + qo * (RT_POOL_QUADS * C) + cellD; /* quad j of the cell: q[j * C] */
        const float4 rec1 = cold[RT_WAVE];
        q[0 * C] = make_float4(rpos.x, rpos.y, rpos.z, __uint_as_float(rng));
        q[1 * C] = make_float4(rdir.x, rdir.y, rdir.z, __uint_as_float((uint32_t)bounce));
        q[2 * C] = make_float4(transmittance.x, transmittance.y, transmittance.z, h.dst);
        q[3 * C] = make_float4(pathLight.x, pathLight.y, pathLight.z, __uint_as_float((uint32_t)h.obj));
        q[4 * C] = make_float4(__uint_as_float((uint32_t)h.tri), h.u, h.v, h.det);
        q[5 * C] = make_float4(__uint_as_float(pxu[0 * RT_WAVE]), __uint_as_float(pxu[1 * RT_WAVE]), __uint_as_float(pxu[2 * RT_WAVE]), __uint_as_float(pxu[3 * RT_WAVE]));
        q[6 * C] = cold[0];
        /* the record's second word = `segments` of the LANE when the pixel was set up (tile cost = the chain's segments so far): it travels as the
         * chain's own count and is re-based on the taker's counter; bit 31 of the frame word carries the hit's backface flag */
        q[7 * C] = make_float4(rec1.x, __uint_as_float(segments - __float_as_uint(rec1.y)), __uint_as_float(__float_as_uint(rec1.z) | (h.backface ? 0x80000000u : 0u)), rec1.w);
        /* published after the payload: a wave's LDS instructions execute in order, so no wait is needed between them — only the compiler must keep the order */
        asm volatile("" ::: "memory");
        RT_POOL_STORE(spD, posD + 1u);
        laneDone = true;
        inTrav = false;
        pathActive = false;
        if (STATS) st.hotSteps++;
    }
    if (withdraws && !gaveUp) {
        const float4* const q = payload + qt * (RT_POOL_QUADS * C) + cellW;
        const float4 q0 = q[0 * C], q1 = q[1 * C], q2 = q[2 * C], q3 = q[3 * C], q4 = q[4 * C], q5 = q[5 * C], q6 = q[6 * C], q7 = q[7 * C];
        asm volatile("" ::: "memory");
        RT_POOL_STORE(spW, posW + C); /* behind the reads in the wave's LDS order: the cell is free for the next lap */
        rpos = rt_v3(q0.x, q0.y, q0.z); rng = __float_as_uint(q0.w);
        rdir = rt_v3(q1.x, q1.y, q1.z); bounce = (int)__float_as_uint(q1.w);
        transmittance = rt_v3(q2.x, q2.y, q2.z); h.dst = q2.w;
        pathLight = rt_v3(q3.x, q3.y, q3.z); h.obj = (int)__float_as_uint(q3.w);
        h.tri = (int)__float_as_uint(q4.x); h.u = q4.y; h.v = q4.z; h.det = q4.w;
        pxu[0 * RT_WAVE] = __float_as_uint(q5.x); pxu[1 * RT_WAVE] = __float_as_uint(q5.y); pxu[2 * RT_WAVE] = __float_as_uint(q5.z); pxu[3 * RT_WAVE] = __float_as_uint(q5.w);
        cold[0] = q6;
        cold[RT_WAVE] = make_float4(q7.x, __uint_as_float(segments - __float_as_uint(q7.y)), __uint_as_float(__float_as_uint(q7.z) & 0x7fffffffu), q7.w);
        h.backface = (__float_as_uint(q7.z) >> 31) != 0u;
        laneDone = false;
        inTrav = true;
        pathActive = true;
    }
}

/* ---------------------------------------------------------------------------
 * The trace kernel: RayTrace (RCC:10-24) -> RayTrace(uv) (RC:545-582) -> Trace (RC:479-542).
 *
 * One wave64 per 8x8 tile, one lane per pixel.  A lane is a small state machine over
 * its pixel's serial work (quirk Q13: one RNG chain per pixel per frame):
 *     [frame finished?] -> camera ray -> { spheres -> models/BVH -> shade } per bounce ...
 * and the wave regroups lanes by what they need next instead of following the
 * reference's nested loops:
 *   - a lane whose path ended starts its pixel's next sample at once;
 *   - lanes enter the traversal loop together but leave it as soon as half of them are
 *     done; the rest stay suspended inside their traversal (state in registers + LDS
 *     stack) while the finished ones shade, bounce and re-enter with their next ray.
 * LDS: the per-lane traversal stack, [level][lane], sized by the host to the deepest
 * BVH of the scene (dynamic shared memory).
 * ------------------------------------------------------------------------- */
template <bool STATS, bool FLAT, bool MANY, bool HOT>
__device__ __forceinline__ void trace_body(const KArgs& a)
{
    /* A workgroup is wavesPerGroup waves (1 for the FLAT variant) that share ONE thing: the LDS copy of the top of the scene's trees
     * (traverse(), phase B).  LDS: [hot cache: hotUnits x 16 B][wave 0: stack, pixel fields, ...][wave 1: ...] ...  After the fill and its
     * one barrier the waves never meet again: each is the persistent wave of rounds 1-5 with the global wave index gw where blockIdx.x was. */
    extern __shared__ uint32_t s_lds[];
    /* FLAT && HOT: the FLAT variant launched as multi-wave workgroups that share the CHAIN POOL (pool_exchange) in place of a tree cache */
    constexpr bool POOL = FLAT && HOT;
    const int lane = HOT ? (int)(threadIdx.x & (RT_WAVE - 1)) : (int)threadIdx.x;
    const int wave = HOT ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const uint32_t hotUnits = HOT ? (uint32_t)a.hotUnits : 0u;
    const RT_LDS char* const hotLds = (const RT_LDS char*)s_lds;
    if (POOL) {
        /* both queues empty: head = tail = 0, cell c free for position c (pool_exchange) */
        for (uint32_t i = threadIdx.x; i < RT_POOL_HEADER_DWORDS + 2u * RT_POOL_CELLS; i += blockDim.x) s_lds[i] = i < RT_POOL_HEADER_DWORDS ? 0u : ((i - RT_POOL_HEADER_DWORDS) & (RT_POOL_CELLS - 1u));
        __syncthreads();
    }
    if (HOT && !FLAT && hotUnits) {
        /* unit i = quarter (i & 3) of record (i >> 2): coalesced 16-byte loads of the pair space's first hotUnits units */
        const float4* src = reinterpret_cast<const float4*>(a.pairs);
        float4* dst = reinterpret_cast<float4*>(s_lds);
        const uint32_t nHot = hotUnits >> 2;
        for (uint32_t i = threadIdx.x; i < hotUnits; i += blockDim.x) dst[(i & 3u) * nHot + (i >> 2)] = src[i];
        __syncthreads();
    }
    uint32_t* const s_stack = HOT ? s_lds + (hotUnits * 4u + (uint32_t)wave * (uint32_t)a.waveLdsDwords) : s_lds;
    const int gw = HOT ? (int)blockIdx.x * a.wavesPerGroup + wave : (int)blockIdx.x; /* this wave among the launch's waves */
    uint32_t* stackBase = &s_stack[lane];

    /* Persistent wave: the wave starts on tile blockIdx.x and, whenever lanes run out of
     * work (their pixel is finished), hands them the next unassigned pixels of its current
     * "pool" tile, pulling a fresh 8x8 tile from a global atomic queue when the pool is
     * used up.  Consecutive pool slots are neighbouring pixels, so the rays a wave holds stay
     * spatially close, but no lane idles while its tile mates finish their longer paths. */
    int poolX0 = 0, poolRow0 = 0, poolY0 = 0; /* pool tile: first column, first local row, first GLOBAL row */
    int poolPos = 0; /* next unassigned slot of the pool tile, 64 = exhausted */
    int poolTile = 0; /* the pool tile's index among this context's tiles (the slot of its cost record) */
    int poolFrame = 0; /* the frame this pool tile is rendered for */
    /* A launch of nFrames > 1 frames hands out (tile, frame) items — queue position q = tile position q / nFrames, frame
     * q % nFrames — and a pixel's per-frame colours go to a staging slab that rt_accumulate_kernel adds up in frame order
     * afterwards: the frames of one pixel are independent chains (each reseeds from Frame, RC:552), only their SUM has an
     * order, so a launch is no longer as long as nFrames chains of its slowest pixel.  An item may cover a GROUP of
     * frameGroup consecutive frames, which its lane runs back to back (one pixel set-up per group instead of per frame):
     * the host picks the group size (1 when the items have to be many, more when the per-frame chains are short). */
#define RT_ITEM(c, q, tilePos)                                  \
    do {                                                        \
        if ((c).nFrames > 1) {                                  \
            tilePos = (q) / (c).frameGroups;                    \
            poolFrame = (c).frame0 + ((q) - tilePos * (c).frameGroups) * (c).frameGroup; \
        } else {                                                \
            tilePos = (q);                                      \
            poolFrame = (c).frame0;                             \
        }                                                       \
    } while (0)
    bool queueEmpty;
    /* wave-uniform, once per tile: every row of an 8-row tile lies in one strip (stripRows % 8 == 0);
     * cyclic strips: local strip ls is global strip ls*partCount + partIndex */
#define RT_SET_POOL(c, tile)                                                                              \
    do {                                                                                                  \
        const int ty_ = (tile) / (c).tilesX;                                                              \
        poolX0 = ((tile) - ty_ * (c).tilesX) * 8;                                                         \
        poolRow0 = ty_ * 8;                                                                               \
        const int ls_ = poolRow0 / (c).stripRows;                                                         \
        poolY0 = (ls_ * (c).partCount + (c).partIndex) * (c).stripRows + (poolRow0 - ls_ * (c).stripRows); \
        poolPos = 0;                                                                                      \
        if (FLAT) poolTile = (tile);                                                                      \
    } while (0)
    {
        const RT_CAS KArgs& c = cold_args();
        int tile = gw;
        if (tile < c.launchItems && c.nFrames > 0 && !c.queueStart) {
            const int q0_ = tile;
            RT_ITEM(c, q0_, tile);
            tile = tile * c.orderStride + c.orderOffset;
            if (c.tileOrder) tile = (int)c.tileOrder[tile];
            RT_SET_POOL(c, tile);
        } else {
            poolPos = 64;
        }
        queueEmpty = (c.nFrames <= 0);
    }

    /* per-lane pixel state */
    bool laneDone = true; /* no pixel assigned */
    /* Pixel bookkeeping that is only touched when a path starts or ends lives in LDS next to the
     * traversal stack ([field][lane], conflict free), not in VGPRs: it would otherwise be carried
     * through — and spilled around — the traversal and shading code. */
    uint32_t* const pxu = &s_stack[(size_t)cold_args().stackEntries * RT_WAVE + lane];
    uint32_t* const extBase = pxu + RT_PIXEL_FIELDS * RT_WAVE; /* candidate-mask extension, scenes with more than 64 models */
    float* const pxf = reinterpret_cast<float*>(pxu);
    enum { PX_SAMPLE = 0, PX_TIX, PX_TIY, PX_TIZ };
    /* What is written once per pixel and read once per sample or per frame lives in a per-wave record in device memory
     * (two float4 per lane, [2][64] per wave, private to the lane that wrote them): LDS decides how many waves a CU holds
     * (stack + pixel fields: 160 KB / 6 waves per SIMD = 26 rows of 256 B), and every wave counts (§4.14).
     *   q[0] = (focus point xyz, pixelIndex)   q[1] = (linear pixel index, segment count at set-up, frame, -) */
    /* The FLAT variant has no traversal stack: its LDS is nearly empty (5 rows of 256 B per wave at 8 waves per SIMD), so
     * its records live in LDS too ([2][64] float4 after the pixel fields) — no record traffic to memory at all, and the
     * camera-ray phase reads its focus point back at LDS latency. */
#define PX_COLD(c) (FLAT ? reinterpret_cast<float4*>(__builtin_assume_aligned(pxu - lane + RT_PIXEL_FIELDS * RT_WAVE, 16)) + lane \
                         : ((c).pxCold + (size_t)gw * (RT_COLD_STRIDE_BYTES / 16) + (size_t)lane)) /* [wave][2][lane]: a wave's store covers 1 KB without gaps */
#define PXU(k) pxu[(k) * RT_WAVE]
#define PXF(k) pxf[(k) * RT_WAVE]
    uint32_t rng = 0;

    bool pathActive = false; /* a ray is waiting to be intersected / is being intersected */
    bool inTrav = false;     /* suspended inside traverse() */
    int bounce = 0;
    rt_f3 rpos = rt_v3s(0.0f), rdir = rt_v3s(0.0f), transmittance = rt_v3s(0.0f), pathLight = rt_v3s(0.0f);
    SceneHit h;
    Trav t;
    h.dst = RT_INF; h.obj = -1; h.tri = -1; h.u = h.v = h.det = 0.0f; h.backface = false;
    t.cand = 0; t.rootStep = false; t.m = 0; t.cur = RT_CODE_NEXT_MODEL; t.sp = 0; t.lpos = t.ldir = t.linv = rt_v3s(0.0f); t.triBase = 0; t.cull = true;
    uint32_t segments = 0;
    Stats st = {};

    for (;;) {
        /* ---- hand pixels to idle lanes (every lane of the wave is active here) */
        unsigned long long idle = __ballot(laneDone);
        /* (pooled workgroups too: holding idle lanes back until 8 / 16 / 32 of them can start pixels together — the set-up code runs for 6 lanes in 64 —
         * while the pool has chains for them to take was measured 6 % SLOWER: fewer chains in flight, emptier intersections) */
        while (idle) {
            const RT_CAS KArgs& c = cold_args();
            if (poolPos >= 64) {
                if (queueEmpty) break;
                int next = 0;
                {
                    if (lane == 0) next = (int)(atomicAdd(c.tileQueue, 1ull) - c.tileQueueBase);
                    next = __builtin_amdgcn_readfirstlane(next);
                }
                if (next >= c.launchItems) { queueEmpty = true; break; }
                {
                    const int q_ = next;
                    RT_ITEM(c, q_, next);
                }
                next = next * c.orderStride + c.orderOffset;
                if (c.tileOrder) next = (int)c.tileOrder[next];
                RT_SET_POOL(c, next);
            }
            const int rank = __popcll(idle & ((1ull << lane) - 1ull));
            const int avail = 64 - poolPos;
            if (laneDone && rank < avail) {
                phase_mark<STATS>(st, PH_REFILL);
                const int slot = poolPos + rank;
                const int x = poolX0 + (slot & 7);
                const int lrow = poolRow0 + (slot >> 3);
                if (x < (int)c.W && lrow < c.localRows) {
                    const int y = poolY0 + (slot >> 3);
                    /* RCC:15: id.xy / (Resolution - 1.0) */
                    const float uvx = (float)(uint32_t)x * c.rcpWm1;
                    const float uvy = (float)(uint32_t)y * c.rcpHm1;
                    /* RC:550-556 */
                    const uint32_t pixelCoordX = (uint32_t)(uvx * (float)c.W);
                    const uint32_t pixelCoordY = (uint32_t)(uvy * (float)c.H);
                    const uint32_t pixelIndex = pixelCoordY * c.W + pixelCoordX;
                    const rt_f3 fpl = rt_v3(uvx - 0.5f, uvy - 0.5f, 1.0f) * rt_v3(c.viewParams[0], c.viewParams[1], c.viewParams[2]);
                    float cam[16];
                    for (int k = 0; k < 16; k++) cam[k] = c.cam[k];
                    const rt_f3 focusPoint = rt_mul_point(cam, fpl, 1.0f);
                    {
                        float4* const cold = PX_COLD(c);
                        cold[0] = make_float4(focusPoint.x, focusPoint.y, focusPoint.z, __uint_as_float(pixelIndex));
                        /* (the fourth word, FLAT variant: the pixel's tile, for the tile cost written when the pixel is finished — no division there.  The BVH
                         * variants keep the division: a pixel ends once per thousands of their instructions, and one more live scalar costs them a spill) */
                        cold[RT_WAVE] = make_float4(__uint_as_float((uint32_t)lrow * c.W + (uint32_t)x), __uint_as_float(segments), __uint_as_float((uint32_t)poolFrame),
                                                    FLAT ? __uint_as_float((uint32_t)poolTile) : 0.0f);
                    }
                    PXU(PX_SAMPLE) = 0;
                    PXF(PX_TIX) = 0.0f; PXF(PX_TIY) = 0.0f; PXF(PX_TIZ) = 0.0f;
                    rng = pixelIndex + (uint32_t)poolFrame * 719393u + (uint32_t)c.seed; /* RC:552 */
                    pathActive = false;
                    inTrav = false;
                    laneDone = false;
                }
            }
            const int wanted = __popcll(idle);
            poolPos += wanted < avail ? wanted : avail;
            idle = __ballot(laneDone);
        }
        if (idle == ~0ull) { /* no lane holds a chain and the queue has no pixel left */
            if (!POOL) break;
            /* ... a wave of a pooled workgroup stays while a queue of the pool holds (or is being handed) a chain: whoever is alive takes it */
            if (RT_RFL(RT_POOL_LOAD(s_lds + 0)) == RT_RFL(RT_POOL_LOAD(s_lds + 3)) && RT_RFL(RT_POOL_LOAD(s_lds + 2)) == RT_RFL(RT_POOL_LOAD(s_lds + 1))) break; /* (head0, tail1, head1, tail0) */
        }
        if (!laneDone) {
        phase_mark<STATS>(st, PH_LOOP);
        if (!inTrav) {
            if (!pathActive) {
                const RT_CAS KArgs& c = cold_args();
                /* FLAT items that cover a group of frames keep the frame's offset within the record's first frame in the
                 * upper half of the sample word (the host only forms groups when spp fits the lower half) */
                const uint32_t sampleWord = PXU(PX_SAMPLE);
                const bool grouped = FLAT && c.frameGroup > 1;
                int sample = grouped ? (int)(sampleWord & 0xffffu) : (int)sampleWord;
                if (sample == c.spp) {
                    /* RC:581 + RCC:18-23: finish this frame of this pixel */
                    float4* const cold = PX_COLD(c);
                    const float4 rec = cold[RT_WAVE];
                    const uint32_t pixLinear = __float_as_uint(rec.x), segStart = __float_as_uint(rec.y);
                    const int frameFirst = (int)__float_as_uint(rec.z);
                    const int frameNow = frameFirst + (grouped ? (int)(sampleWord >> 16) : 0);
                    const size_t pixOff = (size_t)pixLinear * 4;
                    rt_f3 col = rt_v3(PXF(PX_TIX), PXF(PX_TIY), PXF(PX_TIZ)) * c.rcpSpp; /* / NumRaysPerPixel */
                    if (c.nFrames > 1) {
                        /* one of several frames of this launch: the colour waits in its frame's slab for
                         * rt_accumulate_kernel, which performs RCC:18-23 for the frames in order */
                        const size_t slab = (size_t)(frameNow - c.frame0) * c.stagingStride;
                        *reinterpret_cast<float4*>(c.staging + (slab + pixLinear) * 4) = make_float4(col.x, col.y, col.z, 1.0f);
                    } else {
                        *reinterpret_cast<float4*>(c.frameRender + pixOff) = make_float4(col.x, col.y, col.z, 1.0f);
                        if (c.accumulate) {
                            float4 acc = *reinterpret_cast<float4*>(c.accumulated + pixOff);
                            acc.x += col.x;
                            acc.y += col.y;
                            acc.z += col.z;
                            acc.w += 1.0f;
                            *reinterpret_cast<float4*>(c.accumulated + pixOff) = acc;
                        }
                    }
                    const int nextFrame = frameNow + 1;
                    /* the item ends frameGroup frames past its first one (items start at multiples of frameGroup past frame0), or with the launch.
                     * No integer division on this path: SOME lane of a wave finishes a pixel in three iterations out of four on the headline scene, so
                     * whatever this block costs is paid by the whole wave nearly every iteration (rounds 1-5 had three divisions here: ~ 90 instructions) */
                    /* groups exist in the FLAT variant only (the host keeps frameGroup at 1 otherwise): the BVH variants are
                     * register-bound and paid 1.3 % for carrying the branch without ever gaining from it */
                    if (!FLAT || nextFrame >= c.frame0 + c.nFrames || nextFrame - frameFirst >= c.frameGroup) {
                        laneDone = true;
                        if (c.tileCost) { /* longest serial chain (per frame) of this tile's pixels: the next launches' queue order */
                            uint32_t tileIdx;
                            if (FLAT) tileIdx = __float_as_uint(rec.w);
                            else { const uint32_t prow = pixLinear / c.W, pcol = pixLinear - prow * c.W; tileIdx = (prow >> 3) * (uint32_t)c.tilesX + (pcol >> 3); }
                            uint32_t* const slot = c.tileCost + tileIdx;
                            /* (scheduling only: a group size that is not a power of two rounds the per-frame figure up to the next one below) */
                            const uint32_t chain = FLAT ? (segments - segStart) >> c.frameGroupShift : segments - segStart;
                            if (chain > *slot) atomicMax(slot, chain); /* the plain read may be stale (lower): then the atomic decides */
                        }
                    } else { /* the next frame of this item's group: same pixel, fresh seed (RC:552) */
                        rng = __float_as_uint(cold[0].w) + (uint32_t)nextFrame * 719393u + (uint32_t)c.seed;
                        sample = 0;
                        PXU(PX_SAMPLE) = (sampleWord & 0xffff0000u) + 0x10000u; /* (frame offset + 1) << 16: this branch exists for grouped items only */
                        PXF(PX_TIX) = 0.0f; PXF(PX_TIY) = 0.0f; PXF(PX_TIZ) = 0.0f;
                    }
                }
                if (!laneDone && sample < c.spp) {
                    /* RC:565-576: next camera ray of this pixel */
                    phase_mark<STATS>(st, PH_RAYGEN);
                    /* camera constants — RC:547,557-558 */
                    float cam[16];
                    for (int k = 0; k < 16; k++) cam[k] = c.cam[k];
                    const rt_f3 camOrigin = rt_mul_point(cam, rt_v3(0.0f, 0.0f, 0.0f), 1.0f);
                    const rt_f3 camRight = rt_v3(cam[0], cam[1], cam[2]);
                    const rt_f3 camUp = rt_v3(cam[4], cam[5], cam[6]);
                    const float invNumPixelsX = c.rcpW; /* x / numPixels.x */
                    rt_f3 rayOrigin;
                    if (c.raygenNoDefocus) {
                        /* defocusStrength == 0 (every BASELINE scene but config 4): the jitter terms are
                         * camRight * (+-0) + camUp * (+-0), and x + (+-0) == x bit for bit unless x is -0, which the
                         * host excluded for the camera origin — only the two random draws of RandomPointInCircle
                         * (RC:159-164) remain, its sin/cos/sqrt are never observed */
                        rt_next_random(&rng);
                        rt_next_random(&rng);
                        rayOrigin = camOrigin;
                    } else {
                        rt_f2 dj = rand_circle(&rng);
                        rayOrigin = camOrigin + camRight * (dj.x * c.defocus * invNumPixelsX) + camUp * (dj.y * c.defocus * invNumPixelsX);
                    }
                    rt_f2 jj = rand_circle(&rng);
                    const float4 fp4 = PX_COLD(c)[0];
                    const rt_f3 focusPoint = rt_v3(fp4.x, fp4.y, fp4.z);
                    rt_f3 jfp = focusPoint + camRight * (jj.x * c.diverge * invNumPixelsX) + camUp * (jj.y * c.diverge * invNumPixelsX);
                    rpos = rayOrigin;
                    rdir = rt_normalize(jfp - rayOrigin);
                    transmittance = rt_v3s(1.0f);
                    pathLight = rt_v3s(0.0f);
                    if (MANY) extBase[(1 + a.extWords) * RT_WAVE] = 0u; /* this variant keeps the bounce count in LDS (a row behind the mask extension): at its
                                                                          * register budget the compiler spilled it to scratch instead */
                    else bounce = 0;
                    PXU(PX_SAMPLE) = grouped ? (PXU(PX_SAMPLE) & 0xffff0000u) | (uint32_t)(sample + 1) : (uint32_t)(sample + 1);
                    if (c.maxBounce >= 0) pathActive = true;                /* RC:485: the loop runs for i = 0 */
                    else { PXF(PX_TIX) = PXF(PX_TIX) + 0.0f; PXF(PX_TIY) = PXF(PX_TIY) + 0.0f; PXF(PX_TIZ) = PXF(PX_TIZ) + 0.0f; } /* Trace returned 0 (RC:578) */
                }
            }
            if (pathActive) {
                phase_mark<STATS>(st, PH_SPHERES);
                begin_intersect<STATS, FLAT, MANY>(a, rpos, rdir, extBase, h, t, st);
                segments++;
                inTrav = true;
                if (FLAT) traverse_flat<STATS>(a, rpos, rdir, h, st);
            }
        }
        if constexpr (!POOL) {
#include "rt_shade_phase.inl"
        }
        } /* !laneDone */
        if constexpr (POOL) {
            pool_exchange<STATS>(a, s_lds, lane, pxu, PX_COLD(cold_args()), laneDone, pathActive, inTrav, rpos, rdir, transmittance, pathLight, rng, bounce, h, segments, st);
            if (!laneDone) {
#include "rt_shade_phase.inl"
            }
        }
    }

#undef RT_SET_POOL
#undef PXU
#undef PXF
#undef PX_COLD
    /* exact work counters: one set of atomics per wave, spread over slots */
    uint32_t segSum = wave_sum(segments);
    unsigned long long* slot = a.counters + (size_t)((uint32_t)gw % RT_COUNTER_SLOTS) * RT_COUNTER_FIELDS;
    if (STATS) {
        uint32_t in = wave_sum(st.inner), lf = wave_sum(st.leaf), tr = wave_sum(st.tri), sp = wave_sum(st.sphere), md = wave_sum(st.model);
        if (lane == 0) {
            atomicAdd(slot + 0, (unsigned long long)segSum);
            atomicAdd(slot + 1, (unsigned long long)in);
            atomicAdd(slot + 2, (unsigned long long)lf);
            atomicAdd(slot + 3, (unsigned long long)tr);
            atomicAdd(slot + 4, (unsigned long long)sp);
            atomicAdd(slot + 5, (unsigned long long)md);
        }
        {
            uint32_t fv = wave_sum(st.filterViolations);
            if (lane == 0 && fv) atomicAdd(slot + 6, (unsigned long long)fv);
            uint32_t hs = wave_sum(st.hotSteps), u48 = wave_sum(st.uni48), um = wave_sum(st.uniMaj);
            if (lane == 0) {
                atomicAdd(slot + 8 + 2 * RT_N_PHASES + 0, (unsigned long long)hs);
                atomicAdd(slot + 8 + 2 * RT_N_PHASES + 1, (unsigned long long)u48);
                atomicAdd(slot + 8 + 2 * RT_N_PHASES + 2, (unsigned long long)um);
            }
        }
        for (int p = 0; p < RT_N_PHASES; p++) {
            uint32_t e = wave_sum(st.phExec[p]), l = wave_sum(st.phLanes[p]);
            if (lane == 0) {
                atomicAdd(slot + 8 + 2 * p, (unsigned long long)e);
                atomicAdd(slot + 9 + 2 * p, (unsigned long long)l);
            }
        }
    } else if (lane == 0) {
        atomicAdd(slot + 0, (unsigned long long)segSum);
    }
}

/* The kernel, under two names: rt_trace_kernel is a launch that renders a whole frame (or batch of
 * frames) of this context's rows; rt_trace_half_kernel is one of the two launches a frame is split
 * into while the context runs on its own streams (rt_context.hip, launch_frames) — the same code, so
 * that profilers list the two kinds of dispatch, whose durations mean different things (the halves
 * overlap in time), separately. */
/* MANY: scenes with more than 64 models (two-level filter, candidate masks extended into LDS) — a separate
 * instantiation so that the common case keeps its registers */
template <bool STATS, bool FLAT, bool MANY = false, bool HOT = false>
__global__ void __launch_bounds__(HOT ? RT_WAVE * (FLAT ? RT_MAX_WAVES_PER_GROUP_FLAT : RT_MAX_WAVES_PER_GROUP) : RT_WAVE, FLAT ? RT_MIN_WAVES_PER_SIMD_FLAT : MANY ? RT_MIN_WAVES_PER_SIMD_MANY : RT_MIN_WAVES_PER_SIMD) rt_trace_kernel(const KArgs a)
{
    trace_body<STATS, FLAT, MANY, HOT>(a);
}
template <bool STATS, bool FLAT, bool MANY = false, bool HOT = false>
__global__ void __launch_bounds__(HOT ? RT_WAVE * (FLAT ? RT_MAX_WAVES_PER_GROUP_FLAT : RT_MAX_WAVES_PER_GROUP) : RT_WAVE, FLAT ? RT_MIN_WAVES_PER_SIMD_FLAT : MANY ? RT_MIN_WAVES_PER_SIMD_MANY : RT_MIN_WAVES_PER_SIMD) rt_trace_half_kernel(const KArgs a)
{
    trace_body<STATS, FLAT, MANY, HOT>(a);
}

/* ---- test hooks (rt_debug_*): the same device functions, one ray / value per lane */
__global__ void __launch_bounds__(RT_WAVE) rt_debug_intersect_kernel(const KArgs a, const float* origins, const float* dirs, int n, float* out)
{
    __shared__ uint32_t s_stack[(RT_STACK_DEPTH + 33 + 16) * RT_WAVE]; /* stack + candidate-mask extension (summary + 32 words) [+ the RT_LDS_NODE_FETCH slab] */
    int i = blockIdx.x * RT_WAVE + threadIdx.x;
    if (i >= n) return;
    rt_f3 o = rt_v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
    rt_f3 d = rt_v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
    SceneHit h;
    Stats st = {};
    intersect_scene<false>(a, o, d, &s_stack[threadIdx.x], &s_stack[RT_STACK_DEPTH * RT_WAVE + threadIdx.x], h, st);
    float* r = out + 10 * i;
    for (int k = 0; k < 10; k++) r[k] = 0.0f;
    r[2] = h.dst;
    if (h.obj >= 0) {
        rt_f3 hpos, normal;
        resolve_hit(a, o, d, h, hpos, normal);
        r[0] = 1.0f;
        r[1] = h.backface ? 1.0f : 0.0f;
        r[3] = normal.x; r[4] = normal.y; r[5] = normal.z;
        r[6] = hpos.x; r[7] = hpos.y; r[8] = hpos.z;
        r[9] = (float)a.materials[h.obj].flag;
    }
}
/* op codes as oracle_math_eval: 0 log 1 exp 2 sin 3 cos 4 sqrt 5 pow 6 div 7 smoothstep(0,y,x) */
__global__ void rt_debug_math_kernel(int op, const float* x, const float* y, float* out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0.0f;
    switch (op) {
    case 0: r = rt_log(x[i]); break;
    case 1: r = rt_exp(x[i]); break;
    case 2: r = rt_sin(x[i]); break;
    case 3: r = rt_cos(x[i]); break;
    case 4: r = rt_sqrt(x[i]); break;
    case 5: r = rt_pow(x[i], y[i]); break;
    case 6: r = rt_div(x[i], y[i]); break;
    case 8: r = rt_rsqrt(x[i]); break;
    case 9: r = rt_rcp(x[i]); break;
    case 7: r = rt_smoothstep(0.0f, y[i], x[i]); break;
    }
    out[i] = r;
}

/* Queue order for the next frame: tiles sorted by the longest pixel chain seen in them,
 * longest first (counting sort, one workgroup).  A pixel's samples and bounces are one
 * serial chain (quirk Q13), so a frame cannot end before its longest chain does; starting
 * the long ones first keeps the end of the frame full of short work.  Pure scheduling:
 * the image does not depend on it. */
__global__ void __launch_bounds__(1024) rt_order_kernel(const uint32_t* cost, uint32_t* key, uint32_t* order, int nTiles)
{
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t base[1024];
    const int t = threadIdx.x;
    hist[t] = 0;
    /* the costs may still be raised by kernels in flight (atomicMax): both passes below must see the SAME keys, or the result
     * is not a permutation — so the keys are snapshot first (one workgroup: its own global writes are visible after the barrier) */
    for (int i = t; i < nTiles; i += 1024) key[i] = 1023u - (cost[i] < 1023u ? cost[i] : 1023u);
    __syncthreads();
    for (int i = t; i < nTiles; i += 1024) atomicAdd(&hist[key[i]], 1u);
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (int k = 0; k < 1024; k++) { base[k] = run; run += hist[k]; }
    }
    __syncthreads();
    /* stable within a bucket is not needed; keep tile order roughly spatial by walking in index order per thread */
    for (int i = t; i < nTiles; i += 1024) order[atomicAdd(&base[key[i]], 1u)] = (uint32_t)i;
}

/* Display pass — Display.shader:42-47: col = tex / Frame (the blit of RayTraceDisplay.cs:9-23).
 * SRGB8: additionally the linear->sRGB conversion + 8-bit quantisation the back buffer applies. */
__global__ void rt_display_kernel(const float4* src, float4* dst, size_t n, int frame)
{
    const float inv = rt_rcp((float)frame);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float4 v = src[i];
        dst[i] = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
    }
}
__global__ void rt_display_srgb8_kernel(const float4* src, uint32_t* dst, int width, int rows, int frame, int flipY)
{
    const float inv = rt_rcp((float)frame);
    const size_t n = (size_t)width * rows;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float4 v = src[i];
        uint32_t r = rt_srgb8(v.x * inv), g = rt_srgb8(v.y * inv), b = rt_srgb8(v.z * inv);
        size_t row = i / width, col = i - row * width;
        size_t o = flipY ? ((size_t)(rows - 1) - row) * width + col : i;
        dst[o] = r | (g << 8) | (b << 16) | 0xff000000u;
    }
}

/* RCC:18-23 for the n frames of a fused launch, in frame order: AccumulatedRender += colour (alpha += 1) frame after
 * frame — the same additions in the same order as n single-frame launches — and FrameRender = the last frame. */
__global__ void rt_accumulate_kernel(const float4* staging, int nFrames, size_t stride, float4* accumulated, float4* frameRender, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        float4 acc = accumulated[i];
        float4 c = make_float4(0.f, 0.f, 0.f, 1.f);
        for (int f = 0; f < nFrames; f++) {
            c = staging[(size_t)f * stride + i];
            acc.x += c.x;
            acc.y += c.y;
            acc.z += c.z;
            acc.w += 1.0f;
        }
        accumulated[i] = acc;
        frameRender[i] = c;
    }
}

/* ResetAccumulated — RCC:26-32 */
/* rt_gather_rccl: rank `part`'s packed tile (its cyclic strips, local rows 0..rows-1) -> those rows of the whole image */
__global__ void rt_unpack_strips_kernel(const float4* __restrict__ packed, float4* __restrict__ image, int W, int rows, int stripRows, int part, int parts)
{
    const size_t n = (size_t)rows * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(i / W), x = (int)(i - (size_t)l * W);
        const int ls = l / stripRows;
        const int g = (ls * parts + part) * stripRows + (l - ls * stripRows); /* rt_local_to_global_row */
        image[(size_t)g * W + x] = packed[i];
    }
}

__global__ void rt_reset_kernel(float4* accum, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) accum[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

} // namespace rtk

#endif
