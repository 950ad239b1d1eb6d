/*
 * rt_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A literal, scalar, one-pixel-at-a-time restatement of the reference's path
 * tracer in the reference's own operation order:
 *     Assets/Scripts/Tracer/RayCompute.compute   ("RCC")  kernels RayTrace / ResetAccumulated
 *     Assets/Scripts/Tracer/RayCommon.hlsl       ("RC")   everything the kernels call
 *     Assets/Scripts/Types/BVH.cs                ("BVH")  the host BVH builder
 *     Assets/Scripts/Tracer/RayComputeManager.cs ("RCM")  camera params
 * Each function cites the reference lines it follows.  Arithmetic primitives
 * (sqrt/log/cos/..., vector ops, PCG) come from include/rt_math.h, the shared
 * fp32 contract, so the HIP kernels can be compared with this file bit for bit.
 *
 * PINNED TO THE REFERENCE'S OWN TEXT (round 4): the reference has no tests, golden images or recorded outputs, and
 * its HLSL cannot run here as a shader (no Unity / dxc) — but it is plain C-like code.  oracle/make_ref.py compiles
 * RayCommon.hlsl + RayCompute.compute AS THEY STAND (a listed set of syntactic rewrites; oracle/ref_compat.h supplies
 * the HLSL types and maps the intrinsics to include/rt_math.h) into oracle/_ref/libref.so, and tests/test_ref_pin.py
 * demands that this restatement gives the same bits as that library: whole FrameRender / AccumulatedRender images of
 * the BVH configs and of all five reference scenes, the shader's own `stats` counters (RC:254,271), and every function
 * below on random inputs — under both readings of the arithmetic contract.  What remains a choice of this repository
 * is include/rt_math.h itself (what '/', normalize, pow, ... evaluate to in fp32 — left to the compiler by HLSL);
 * tests/test_contract_bracket.py brackets it.  Also pinned by hand-derived known-answer tests per function
 * (tests/test_oracle_kat.py), BVH == brute-force property tests, furnace tests and the golden fixtures (tests/golden/).
 * The sphere buffer (extension S1) is outside the shader text: RaySphere is compared at function level.  BVH.cs is C#, and is
 * pinned the same way: `oracle/make_ref.py --bvh` compiles BVH.cs:26-318 and its nested types AS THEY STAND (eighteen listed
 * syntactic rewrites; oracle/ref_bvh_compat.h supplies Vector3, C# arrays, Mathf and Math) into oracle/_ref/libref_bvh.so, and
 * tests/test_ref_pin.py demands that the builder below — and the product's host builder on one and five threads — emits the same
 * nodes, triangle order and BuildStats byte for byte on every mesh class; tests/test_gpu_ref_pin.py does it for the GPU builder.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (libraytrace_hip.so) never links or calls it.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "../include/rt_abi.h"
#include "../include/rt_math.h"

typedef rt_f3 float3;
typedef rt_f2 float2;

namespace {

static const float PI = 3.1415f; /* RC:2 (quirk Q3: two different pi) */
static const float epsilon = 0.001f; /* RC:468 */

struct Counters {
    uint64_t segments = 0, innerSteps = 0, leafSteps = 0, triTests = 0, sphereTests = 0, modelVisits = 0;
    void add(const Counters& o)
    {
        segments += o.segments; innerSteps += o.innerSteps; leafSteps += o.leafSteps;
        triTests += o.triTests; sphereTests += o.sphereTests; modelVisits += o.modelVisits;
    }
};

/* Optional per-thread log of the work one pixel does, as a byte stream (tools/sched_sim2.py models
 * the kernel's scheduling on it; not a reference feature):
 *   'R' camera ray | 'S' segment start | 'A' model entered | 'B' d = inner step | 'C' n d = leaf with n tests
 *   (d = entries left on the traversal stack when the node is popped) |
 *   'K' miss (sky) | 'O' opaque hit | 'G' glass hit | 'E' path ended */
static thread_local std::vector<uint8_t>* g_schedTrace = nullptr;
static inline void sched_tok(uint8_t t) { if (g_schedTrace) g_schedTrace->push_back(t); }
/* ... and of the NODES it visits, for tools/layout_sim.py (which device-memory layout touches how many cache lines):
 *   0xFFFFFFFF segment start | i = inner step, i = absolute index of the popped node's FIRST child | 0x80000000 | i = leaf, i = its node index */
static thread_local std::vector<uint32_t>* g_nodeTrace = nullptr;
static inline void node_tok(uint32_t t) { if (g_nodeTrace) g_nodeTrace->push_back(t); }

struct Ray { /* RC:35-47 */
    float3 pos, dir, invDir, transmittance;
    int bounceCount;
};
struct TriangleHitInfo { /* RC:55-62 */
    bool didHit, isBackface;
    float dst;
    float3 hitPoint, normal;
};
struct ModelHitInfo { /* RC:97-105 */
    bool didHit, isBackface;
    float3 normal, pos;
    float dst;
    RtMaterial material;
};
struct LightResponse { /* RC:107-113 */
    float3 reflectDir, refractDir;
    float reflectWeight, refractWeight;
};

struct Scene {
    std::vector<RtModel> models;
    std::vector<RtTriangle> triangles;
    std::vector<RtBVHNode> nodes;
    std::vector<RtSphere> spheres;
};

static float3 f3(const float* p) { return rt_v3(p[0], p[1], p[2]); }

/* ------------------------------------------------------------ RC:141-164 */
static float RandomValueNormalDistribution(uint32_t* state)
{
    float theta = 2 * 3.1415926f * rt_random_value(state); /* RC:144 */
    float rho = rt_sqrt(-2 * rt_log(rt_random_value(state))); /* RC:145 */
    return rho * rt_cos(theta); /* RC:146 */
}
static float3 RandomDirection(uint32_t* state)
{
    float x = RandomValueNormalDistribution(state);
    float y = RandomValueNormalDistribution(state);
    float z = RandomValueNormalDistribution(state);
    return rt_normalize(rt_v3(x, y, z)); /* RC:156 */
}
static float2 RandomPointInCircle(uint32_t* rngState)
{
    float angle = rt_random_value(rngState) * 2 * PI; /* RC:161 */
    float2 pointOnCircle = {rt_cos(angle), rt_sin(angle)};
    float r = rt_sqrt(rt_random_value(rngState)); /* RC:163 */
    float2 out = {pointOnCircle.x * r, pointOnCircle.y * r};
    return out;
}

/* ------------------------------------------------------------ RC:167-183 */
static float3 GetEnvironmentLight(const RtParams& P, float3 dir)
{
    if (P.useSky == 0) return rt_v3s(0);
    const float3 GroundColour = rt_v3(0.35f, 0.3f, 0.35f);
    const float3 SkyColourHorizon = rt_v3(1, 1, 1);
    const float3 SkyColourZenith = rt_v3(0.08f, 0.37f, 0.73f);

    float skyGradientT = rt_pow(rt_smoothstep_edges(0, 0.4f, dir.y), 0.35f);
    float groundToSkyT = rt_smoothstep_edges(-0.01f, 0, dir.y);
    float3 skyGradient = rt_lerp3(SkyColourHorizon, SkyColourZenith, skyGradientT);
    float s = rt_div(1000 * 1, P.sunFocus); /* RC:178: (1000*1)/SunFocus */
    float sun = rt_pow(rt_max(0, rt_dot(dir, f3(P.dirToSun))), s) * P.sunIntensity;
    float gate = (groundToSkyT >= 1) ? 1.0f : 0.0f;
    float3 composite = rt_lerp3(GroundColour, skyGradient, groundToSkyT) + sun * f3(P.sunColour) * gate;
    return composite;
}

/* ------------------------------------------------------------ RC:188-215 */
static TriangleHitInfo RayTriangle(const Ray& ray, const RtTriangle& tri, bool cullBackface)
{
    float3 posA = f3(tri.posA), posB = f3(tri.posB), posC = f3(tri.posC);
    float3 edgeAB = posB - posA;
    float3 edgeAC = posC - posA;
    float3 triFaceVector = rt_cross(edgeAB, edgeAC);
    float3 vertRayOffset = ray.pos - posA;
    float3 rayOffsetPerp = rt_cross(vertRayOffset, ray.dir);
    float determinant = -rt_dot(ray.dir, triFaceVector);
    float invDet = rt_rcp(determinant);

    float dst = rt_dot(vertRayOffset, triFaceVector) * invDet;
    float u = rt_dot(edgeAC, rayOffsetPerp) * invDet;
    float v = -rt_dot(edgeAB, rayOffsetPerp) * invDet;
    float w = 1 - u - v;

    TriangleHitInfo hitInfo;
    bool keep = cullBackface ? determinant >= 1E-8f : rt_abs(determinant) >= 1E-8f;
    hitInfo.didHit = keep && dst > 0 && u >= 0 && v >= 0 && w >= 0;
    float3 smoothNormal = rt_normalize(f3(tri.normA) * w + f3(tri.normB) * u + f3(tri.normC) * v);
    hitInfo.normal = smoothNormal * rt_sign(determinant);
    hitInfo.isBackface = determinant < 0;
    hitInfo.hitPoint = ray.pos + ray.dir * dst;
    hitInfo.dst = dst;
    return hitInfo;
}

/* ------------------------------------------------------------ RC:219-231 */
static float RayBoundingBoxDst(const Ray& ray, float3 boxMin, float3 boxMax)
{
    float3 tMin = (boxMin - ray.pos) * ray.invDir;
    float3 tMax = (boxMax - ray.pos) * ray.invDir;
    float3 t1 = rt_v3(rt_min(tMin.x, tMax.x), rt_min(tMin.y, tMax.y), rt_min(tMin.z, tMax.z));
    float3 t2 = rt_v3(rt_max(tMin.x, tMax.x), rt_max(tMin.y, tMax.y), rt_max(tMin.z, tMax.z));
    float tNear = rt_max(rt_max(t1.x, t1.y), t1.z);
    float tFar = rt_min(rt_min(t2.x, t2.y), t2.z);

    bool hit = tFar >= tNear && tFar > 0;
    float dst = hit ? (tNear > 0 ? tNear : 0) : RT_INF;
    return dst;
}

/* ------------------------------------------------------------ RC:234-287 */
static const int ORACLE_STACK = 64; /* reference: 32 (RC:239) — see quirk Q7 */

static TriangleHitInfo RayTriangleBVH(const Scene& sc, const Ray& ray, float rayLength, int nodeOffset,
                                      int triOffset, Counters& stats, bool cullBackface)
{
    TriangleHitInfo result;
    result.didHit = false; /* Q6: uninitialised in the reference, never read unless set */
    result.isBackface = false;
    result.hitPoint = rt_v3s(0);
    result.normal = rt_v3s(0);
    result.dst = rayLength;

    int stack[ORACLE_STACK];
    int stackCount = 0;
    stack[stackCount++] = nodeOffset + 0;

    while (stackCount > 0) {
        const int nodeIndex = stack[--stackCount];
        const RtBVHNode& node = sc.nodes[nodeIndex];
        bool isLeaf = node.triangleCount > 0;

        if (isLeaf) {
            stats.leafSteps++;
            sched_tok('C'); sched_tok((uint8_t)(node.triangleCount > 255 ? 255 : node.triangleCount)); sched_tok((uint8_t)stackCount);
            node_tok(0x80000000u | (uint32_t)nodeIndex);
            for (int i = 0; i < node.triangleCount; i++) {
                const RtTriangle& tri = sc.triangles[triOffset + node.startIndex + i];
                TriangleHitInfo triHitInfo = RayTriangle(ray, tri, cullBackface);
                stats.triTests++; /* RC:254 */

                if (triHitInfo.didHit && triHitInfo.dst < result.dst) {
                    result = triHitInfo;
                }
            }
        } else {
            stats.innerSteps++;
            sched_tok('B'); sched_tok((uint8_t)stackCount);
            node_tok((uint32_t)(nodeOffset + node.startIndex));
            int childIndexA = nodeOffset + node.startIndex + 0;
            int childIndexB = nodeOffset + node.startIndex + 1;
            const RtBVHNode& childA = sc.nodes[childIndexA];
            const RtBVHNode& childB = sc.nodes[childIndexB];

            float dstA = RayBoundingBoxDst(ray, f3(childA.boundsMin), f3(childA.boundsMax));
            float dstB = RayBoundingBoxDst(ray, f3(childB.boundsMin), f3(childB.boundsMax));

            bool isNearestA = dstA <= dstB;
            float dstNear = isNearestA ? dstA : dstB;
            float dstFar = isNearestA ? dstB : dstA;
            int childIndexNear = isNearestA ? childIndexA : childIndexB;
            int childIndexFar = isNearestA ? childIndexB : childIndexA;

            if (dstFar < result.dst) stack[stackCount++] = childIndexFar;
            if (dstNear < result.dst) stack[stackCount++] = childIndexNear;
            if (stackCount > ORACLE_STACK - 2) { fprintf(stderr, "oracle: BVH stack overflow\n"); abort(); }
        }
    }
    return result;
}

/* ------------------------------------------------------------ RC:289-332 */
/* Restated with the hard-coded debug material (RC:321-327) left to the caller,
 * which substitutes the sphere's own material (extension S1). */
static ModelHitInfo RaySphere(float3 rayPos, float3 rayDir, float3 sphereCentre, float sphereRadius)
{
    ModelHitInfo hitInfo;
    memset(&hitInfo, 0, sizeof(hitInfo));
    hitInfo.dst = RT_INF;

    float3 offsetRayOrigin = rayPos - sphereCentre;
    float a = rt_dot(rayDir, rayDir);
    float b = 2 * rt_dot(offsetRayOrigin, rayDir);
    float c = rt_dot(offsetRayOrigin, offsetRayOrigin) - sphereRadius * sphereRadius;
    float discriminant = b * b - 4 * a * c;

    if (discriminant >= 0) {
        float s = rt_sqrt(discriminant);
        float dstNear = rt_max(0, rt_div(-b - s, 2 * a));
        float dstFar = rt_div(-b + s, 2 * a);

        if (dstFar >= 0) {
            hitInfo.didHit = true;
            bool isInside = dstNear == 0;
            hitInfo.isBackface = isInside;
            hitInfo.dst = isInside ? dstFar : dstNear;

            hitInfo.pos = rayPos + rayDir * hitInfo.dst;
            hitInfo.normal = rt_normalize(hitInfo.pos - sphereCentre) * (isInside ? -1.0f : 1.0f);
        }
    }
    return hitInfo;
}

/* ------------------------------------------------------------ RC:335-374 */
static ModelHitInfo CalculateRayCollision(const Scene& sc, const Ray& worldRay, bool forceDontCullBack,
                                          Counters& stats)
{
    ModelHitInfo result;
    memset(&result, 0, sizeof(result)); /* Q6: didHit treated as false */
    result.dst = RT_INF;
    stats.segments++;
    sched_tok('S');
    node_tok(0xFFFFFFFFu);

    /* Extension S1, hooked at the commented-out call RC:341: analytic spheres are
     * tested first, in buffer order, strict '<' keeps the first of equal hits. */
    for (size_t i = 0; i < sc.spheres.size(); i++) {
        const RtSphere& sp = sc.spheres[i];
        ModelHitInfo h = RaySphere(worldRay.pos, worldRay.dir, f3(sp.centre), sp.radius);
        stats.sphereTests++;
        if (h.didHit && h.dst < result.dst) {
            result = h;
            result.material = sp.material;
        }
    }

    Ray localRay;
    localRay.transmittance = rt_v3s(0);
    localRay.bounceCount = 0;

    for (size_t i = 0; i < sc.models.size(); i++) {
        const RtModel& model = sc.models[i];
        stats.modelVisits++;
        sched_tok('A');
        /* RC:351-353 */
        localRay.pos = rt_mul_point(model.worldToLocal, worldRay.pos, 1);
        localRay.dir = rt_mul_point(model.worldToLocal, worldRay.dir, 0);
        localRay.invDir = rt_v3(rt_rcp(localRay.dir.x), rt_rcp(localRay.dir.y), rt_rcp(localRay.dir.z));

        bool cullBackface = model.material.flag != RT_MATERIAL_GLASS;
        if (forceDontCullBack) cullBackface = false;
        TriangleHitInfo hit = RayTriangleBVH(sc, localRay, result.dst, model.nodeOffset, model.triOffset, stats, cullBackface);

        if (hit.dst < result.dst) {
            result.didHit = true;
            result.isBackface = hit.isBackface;
            result.dst = hit.dst;
            result.normal = rt_normalize(rt_mul_point(model.localToWorld, hit.normal, 0)); /* RC:367, quirk Q10 */
            result.pos = worldRay.pos + worldRay.dir * hit.dst;
            result.material = model.material;
        }
    }
    return result;
}

/* ------------------------------------------------------------ RC:383-437 */
static float CalculateReflectance(float3 inDir, float3 normal, float iorA, float iorB)
{
    float refractRatio = rt_div(iorA, iorB);
    float cosAngleIn = -rt_dot(inDir, normal);
    float sinSqrAngleOfRefraction = refractRatio * refractRatio * (1 - cosAngleIn * cosAngleIn);
    if (sinSqrAngleOfRefraction >= 1) return 1;

    float cosAngleOfRefraction = rt_sqrt(1 - sinSqrAngleOfRefraction);
    float denominatorPerpendicular = iorA * cosAngleIn + iorB * cosAngleOfRefraction;
    float denominatorParallel = iorA * cosAngleIn + iorB * cosAngleOfRefraction; /* RC:392: same expression, kept */

    if (rt_min(denominatorPerpendicular, denominatorParallel) < 1E-8f) return 1;

    float rPerpendicular = rt_div(iorA * cosAngleIn - iorB * cosAngleOfRefraction, denominatorPerpendicular);
    rPerpendicular *= rPerpendicular;
    float rParallel = rt_div(iorB * cosAngleIn - iorA * cosAngleOfRefraction, denominatorParallel);
    rParallel *= rParallel;

    return rt_div(rPerpendicular + rParallel, 2);
}
static float3 Refract(float3 inDir, float3 normal, float iorA, float iorB)
{
    float refractRatio = rt_div(iorA, iorB);
    float cosAngleIn = -rt_dot(inDir, normal);
    float sinSqrAngleOfRefraction = refractRatio * refractRatio * (1 - cosAngleIn * cosAngleIn);
    if (sinSqrAngleOfRefraction > 1) return rt_v3s(0);

    float3 refractDir = refractRatio * inDir + (refractRatio * cosAngleIn - rt_sqrt(1 - sinSqrAngleOfRefraction)) * normal;
    return refractDir;
}
static float3 Reflect(float3 inDir, float3 normal)
{
    return inDir - (2 * rt_dot(inDir, normal)) * normal; /* RC:421 */
}
static LightResponse CalculateReflectionAndRefraction(float3 inDir, float3 normal, float iorA, float iorB)
{
    LightResponse result;
    result.reflectDir = Reflect(inDir, normal);
    result.refractDir = Refract(inDir, normal, iorA, iorB);
    result.reflectWeight = CalculateReflectance(inDir, normal, iorA, iorB);
    result.refractWeight = 1 - result.reflectWeight;
    return result;
}

/* ------------------------------------------------------------ RC:376-379, 450-466 */
static float mod2(float x, float y) { return x - y * rt_floor(rt_div(x, y)); }

static float3 GetMaterialColour(const RtMaterial& mat, float3 pos, float3 normal, bool isSpecularBounce)
{
    float3 col = rt_v3(mat.diffuseCol[0], mat.diffuseCol[1], mat.diffuseCol[2]);

    if (mat.flag == RT_MATERIAL_CHECKERED) {
        float2 checkerPoint = {pos.x, pos.z};
        if (rt_abs(normal.x) > rt_abs(normal.y)) { checkerPoint.x = pos.z; checkerPoint.y = pos.y; }
        if (rt_abs(normal.z) > rt_max(rt_abs(normal.x), rt_abs(normal.y))) { checkerPoint.x = pos.x; checkerPoint.y = pos.y; }

        checkerPoint.x *= 1.5f;
        checkerPoint.y *= 1.5f;
        float cx = mod2(rt_floor(checkerPoint.x), 2.0f);
        float cy = mod2(rt_floor(checkerPoint.y), 2.0f);
        col = cx == cy ? col : rt_v3(mat.emissionCol[0], mat.emissionCol[1], mat.emissionCol[2]);
    }
    return rt_lerp3(col, rt_v3(mat.specularCol[0], mat.specularCol[1], mat.specularCol[2]), isSpecularBounce ? 1.0f : 0.0f);
}

/* ------------------------------------------------------------ RC:439-448 */
static Ray CreateRay(float3 origin, float3 dir, float3 transmittance, int bounceIndex)
{
    Ray ray;
    ray.pos = origin;
    ray.dir = dir;
    ray.invDir = rt_v3(rt_rcp(dir.x), rt_rcp(dir.y), rt_rcp(dir.z));
    ray.transmittance = transmittance;
    ray.bounceCount = bounceIndex;
    return ray;
}

/* ------------------------------------------------------------ RC:479-542 */
static float3 Trace(const Scene& sc, const RtParams& P, Ray initialRay, uint32_t* rngState, Counters& stats)
{
    float3 totalLight = rt_v3s(0);
    Ray ray = initialRay;

    for (int i = ray.bounceCount; i <= P.maxBounceCount; i++) { /* Q5: inclusive */
        ModelHitInfo hit = CalculateRayCollision(sc, ray, false, stats);
        if (!hit.didHit) {
            if (P.useSky) {
                totalLight = totalLight + ray.transmittance * GetEnvironmentLight(P, ray.dir);
            }
            sched_tok('K');
            break;
        }

        const RtMaterial& material = hit.material;
        sched_tok(material.flag == RT_MATERIAL_GLASS ? 'G' : 'O');

        if (material.flag == RT_MATERIAL_GLASS) {
            /* RC:502: exp(-hit.dst * absorption.rgb * absorptionStrength) */
            if (hit.isBackface) {
                float3 ab = rt_v3(material.absorption[0], material.absorption[1], material.absorption[2]);
                float3 e = (-hit.dst * ab) * material.absorptionStrength;
                ray.transmittance = ray.transmittance * rt_v3(rt_exp(e.x), rt_exp(e.y), rt_exp(e.z));
            }

            float iorCurrent = hit.isBackface ? material.ior : 1;
            float iorNext = hit.isBackface ? 1 : material.ior;
            LightResponse lr = CalculateReflectionAndRefraction(ray.dir, hit.normal, iorCurrent, iorNext);

            float3 diffuseDir = rt_normalize(hit.normal + RandomDirection(rngState));
            lr.reflectDir = rt_normalize(rt_lerp3(diffuseDir, lr.reflectDir, material.specularProbability));
            lr.refractDir = rt_normalize(rt_lerp3(-diffuseDir, lr.refractDir, material.smoothness));

            bool followReflection = rt_random_value(rngState) <= lr.reflectWeight;
            ray.dir = followReflection ? lr.reflectDir : lr.refractDir;
            ray.pos = hit.pos + (epsilon * hit.normal) * rt_sign(rt_dot(hit.normal, ray.dir));
        } else {
            bool isSpecularBounce = material.specularProbability >= rt_random_value(rngState);

            ray.pos = hit.pos + (hit.normal * epsilon);
            float3 diffuseDir = rt_normalize(hit.normal + RandomDirection(rngState));
            float3 specularDir = rt_reflect(ray.dir, hit.normal);
            ray.dir = rt_normalize(rt_lerp3(diffuseDir, specularDir, material.smoothness * (isSpecularBounce ? 1.0f : 0.0f)));

            float3 emittedLight = rt_v3(material.emissionCol[0], material.emissionCol[1], material.emissionCol[2]) * material.emissionStrength;
            totalLight = totalLight + emittedLight * ray.transmittance;
            ray.transmittance = ray.transmittance * GetMaterialColour(material, hit.pos, hit.normal, isSpecularBounce);
        }

        float p = rt_max(ray.transmittance.x, rt_max(ray.transmittance.y, ray.transmittance.z));
        if (rt_random_value(rngState) >= p) break;
        ray.transmittance = ray.transmittance * rt_rcp(p);
    }
    return totalLight;
}

/* ------------------------------------------------------------ RC:545-582 */
static float3 RayTracePixel(const Scene& sc, const RtParams& P, float2 uv, uint32_t numPixelsX, uint32_t numPixelsY,
                            Counters& stats)
{
    const float* M = P.camLocalToWorld;
    float3 camOrigin = rt_mul_point(M, rt_v3(0, 0, 0), 1);

    /* RC:550-552 (quirks Q1, Q2) */
    uint32_t pixelCoordX = (uint32_t)(uv.x * (float)numPixelsX);
    uint32_t pixelCoordY = (uint32_t)(uv.y * (float)numPixelsY);
    uint32_t pixelIndex = pixelCoordY * numPixelsX + pixelCoordX;
    uint32_t rngState = pixelIndex + (uint32_t)P.frame * 719393u + (uint32_t)P.renderSeed;

    float3 focusPointLocal = rt_v3(uv.x - 0.5f, uv.y - 0.5f, 1) * f3(P.viewParams);
    float3 focusPoint = rt_mul_point(M, focusPointLocal, 1);
    float3 camRight = rt_v3(M[0], M[1], M[2]); /* _m00_m10_m20 */
    float3 camUp = rt_v3(M[4], M[5], M[6]);    /* _m01_m11_m21 */

    float3 totalIncomingLight = rt_v3s(0);

    for (int rayIndex = 0; rayIndex < P.numRaysPerPixel; rayIndex++) {
        float2 dj = RandomPointInCircle(&rngState);
        float2 defocusJitter = {rt_div(dj.x * P.defocusStrength, (float)numPixelsX), rt_div(dj.y * P.defocusStrength, (float)numPixelsX)};
        float3 rayOrigin = camOrigin + camRight * defocusJitter.x + camUp * defocusJitter.y;

        float2 jj = RandomPointInCircle(&rngState);
        float2 jitter = {rt_div(jj.x * P.divergeStrength, (float)numPixelsX), rt_div(jj.y * P.divergeStrength, (float)numPixelsX)};
        float3 jitteredFocusPoint = focusPoint + camRight * jitter.x + camUp * jitter.y;
        float3 rayDir = rt_normalize(jitteredFocusPoint - rayOrigin);

        Ray ray = CreateRay(rayOrigin, rayDir, rt_v3s(1), 0);
        sched_tok('R');
        totalIncomingLight = totalIncomingLight + Trace(sc, P, ray, &rngState, stats);
    }
    return totalIncomingLight / (float)P.numRaysPerPixel;
}

/* ====================================================== BVH.cs:26-318 */
struct BVHTriangle { /* BVH:459-496 */
    float CentreX, CentreY, CentreZ, MinX, MinY, MinZ, MaxX, MaxY, MaxZ;
    int Index;
};
static float min3(float a, float b, float c) { return a < b ? (a < c ? a : c) : (b < c ? b : c); }
static float max3(float a, float b, float c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

struct BvhBuilder {
    std::vector<RtBVHNode> nodes;
    std::vector<BVHTriangle> tris;
    int quality;
    RtBvhStats stats;

    static constexpr float FMAX = 3.40282347e+38f; /* float.MaxValue; float.MinValue = -FMAX */

    void RecordNode(int depth, bool isLeaf, int triCount = 0) /* BVH:539-554 */
    {
        stats.totalNodeCount++;
        if (isLeaf) {
            stats.leafNodeCount++;
            stats.leafDepthSum += depth;
            if (depth < stats.leafDepthMin) stats.leafDepthMin = depth;
            if (depth > stats.leafDepthMax) stats.leafDepthMax = depth;
            stats.triangleCount += triCount;
            if (triCount > stats.leafMaxTriCount) stats.leafMaxTriCount = triCount;
            if (triCount < stats.leafMinTriCount) stats.leafMinTriCount = triCount;
        }
    }
    static float NodeCost(float x, float y, float z, int numTriangles) /* BVH:313-318 */
    {
        if (numTriangles == 0) return 0;
        float area = x * y + x * z + y * z;
        return area * numTriangles;
    }
    float EvaluateSplit(int splitAxis, float splitPos, int start, int count) /* BVH:253-311 */
    {
        int numOnLeft = 0, numOnRight = 0;
        float xMinL = FMAX, xMaxL = -FMAX, yMinL = FMAX, yMaxL = -FMAX, zMinL = FMAX, zMaxL = -FMAX;
        float xMinR = FMAX, xMaxR = -FMAX, yMinR = FMAX, yMaxR = -FMAX, zMinR = FMAX, zMaxR = -FMAX;
        int end = start + count;
        for (int i = start; i < end; i++) {
            const BVHTriangle& tri = tris[i];
            float c = splitAxis == 0 ? tri.CentreX : splitAxis == 1 ? tri.CentreY : tri.CentreZ;
            if (c < splitPos) {
                if (tri.MinX < xMinL) xMinL = tri.MinX;
                if (tri.MinY < yMinL) yMinL = tri.MinY;
                if (tri.MinZ < zMinL) zMinL = tri.MinZ;
                if (tri.MaxX > xMaxL) xMaxL = tri.MaxX;
                if (tri.MaxY > yMaxL) yMaxL = tri.MaxY;
                if (tri.MaxZ > zMaxL) zMaxL = tri.MaxZ;
                numOnLeft++;
            } else {
                if (tri.MinX < xMinR) xMinR = tri.MinX;
                if (tri.MinY < yMinR) yMinR = tri.MinY;
                if (tri.MinZ < zMinR) zMinR = tri.MinZ;
                if (tri.MaxX > xMaxR) xMaxR = tri.MaxX;
                if (tri.MaxY > yMaxR) yMaxR = tri.MaxY;
                if (tri.MaxZ > zMaxR) zMaxR = tri.MaxZ;
                numOnRight++;
            }
        }
        float costA = NodeCost(xMaxL - xMinL, yMaxL - yMinL, zMaxL - zMinL, numOnLeft);
        float costB = NodeCost(xMaxR - xMinR, yMaxR - yMinR, zMaxR - zMinR, numOnRight);
        return costA + costB;
    }
    void ChooseSplit(const RtBVHNode& node, int start, int count, int* axisOut, float* posOut, float* costOut) /* BVH:183-250 */
    {
        if (count <= 1) { *axisOut = 0; *posOut = 0; *costOut = RT_INF; return; }
        float sizeX = node.boundsMax[0] - node.boundsMin[0];
        float sizeY = node.boundsMax[1] - node.boundsMin[1];
        float sizeZ = node.boundsMax[2] - node.boundsMin[2];

        if (quality == RT_BVH_QUALITY_LOW) {
            int largestAxisIndex = (sizeX > sizeY && sizeX > sizeZ) ? 0 : (sizeY > sizeZ ? 1 : 2);
            float pos = largestAxisIndex == 0 ? node.boundsMin[0] + sizeX * 0.5f
                      : largestAxisIndex == 1 ? node.boundsMin[1] + sizeY * 0.5f
                                              : node.boundsMin[2] + sizeZ * 0.5f;
            *axisOut = largestAxisIndex; *posOut = pos; *costOut = EvaluateSplit(largestAxisIndex, pos, start, count);
            return;
        }

        float bestSplitPos = 0;
        int bestSplitAxis = 0;
        int maxSplitTests = count < 10 ? 3 : 5;
        float maxAxis = max3(sizeX, sizeY, sizeZ) ; /* Mathf.Max(sizeX,sizeY,sizeZ) */
        float bestCost = FMAX;

        for (int axis = 0; axis < 3; axis++) {
            float axisSize = axis == 0 ? sizeX : axis == 1 ? sizeY : sizeZ;
            float axisMin = node.boundsMin[axis];

            /* Mathf.CeilToInt(axisSize / maxAxis * maxSplitTests): NaN (0/0) casts to int.MinValue in C# */
            float v = axisSize / maxAxis * maxSplitTests;
            int numSplitTests = (v != v) ? INT32_MIN : (int)ceilf(v);
            numSplitTests = numSplitTests < 1 ? 1 : (numSplitTests > maxSplitTests ? maxSplitTests : numSplitTests);

            for (int i = 0; i < numSplitTests; i++) {
                float splitT = (i + 1) / (numSplitTests + 1.0f);
                float splitPos = axisMin + axisSize * splitT;
                float cost = EvaluateSplit(axis, splitPos, start, count);
                if (cost < bestCost) {
                    bestCost = cost;
                    bestSplitPos = splitPos;
                    bestSplitAxis = axis;
                }
            }
        }
        *axisOut = bestSplitAxis; *posOut = bestSplitPos; *costOut = bestCost;
    }
    void Split(int parentIndex, int triGlobalStart, int triNum, int depth) /* BVH:89-181 */
    {
        const int MaxDepth = 32;
        RtBVHNode parent = nodes[parentIndex];
        float sizeX = parent.boundsMax[0] - parent.boundsMin[0];
        float sizeY = parent.boundsMax[1] - parent.boundsMin[1];
        float sizeZ = parent.boundsMax[2] - parent.boundsMin[2];
        float parentCost = NodeCost(sizeX, sizeY, sizeZ, triNum);

        int splitAxis; float splitPos, cost;
        ChooseSplit(parent, triGlobalStart, triNum, &splitAxis, &splitPos, &cost);

        if (cost < parentCost && depth < MaxDepth) {
            float xMinL = FMAX, xMaxL = -FMAX, yMinL = FMAX, yMaxL = -FMAX, zMinL = FMAX, zMaxL = -FMAX;
            float xMinR = FMAX, xMaxR = -FMAX, yMinR = FMAX, yMaxR = -FMAX, zMinR = FMAX, zMaxR = -FMAX;
            int numOnLeft = 0;

            for (int i = triGlobalStart; i < triGlobalStart + triNum; i++) {
                BVHTriangle tri = tris[i];
                float c = splitAxis == 0 ? tri.CentreX : splitAxis == 1 ? tri.CentreY : tri.CentreZ;
                if (c < splitPos) {
                    if (tri.MinX < xMinL) xMinL = tri.MinX;
                    if (tri.MinY < yMinL) yMinL = tri.MinY;
                    if (tri.MinZ < zMinL) zMinL = tri.MinZ;
                    if (tri.MaxX > xMaxL) xMaxL = tri.MaxX;
                    if (tri.MaxY > yMaxL) yMaxL = tri.MaxY;
                    if (tri.MaxZ > zMaxL) zMaxL = tri.MaxZ;

                    BVHTriangle swap = tris[triGlobalStart + numOnLeft];
                    tris[triGlobalStart + numOnLeft] = tri;
                    tris[i] = swap;
                    numOnLeft++;
                } else {
                    if (tri.MinX < xMinR) xMinR = tri.MinX;
                    if (tri.MinY < yMinR) yMinR = tri.MinY;
                    if (tri.MinZ < zMinR) zMinR = tri.MinZ;
                    if (tri.MaxX > xMaxR) xMaxR = tri.MaxX;
                    if (tri.MaxY > yMaxR) yMaxR = tri.MaxY;
                    if (tri.MaxZ > zMaxR) zMaxR = tri.MaxZ;
                }
            }

            int numOnRight = triNum - numOnLeft;
            int triStartLeft = triGlobalStart + 0;
            int triStartRight = triGlobalStart + numOnLeft;

            RtBVHNode childLeft = {{xMinL, yMinL, zMinL}, {xMaxL, yMaxL, zMaxL}, triStartLeft, 0};
            RtBVHNode childRight = {{xMinR, yMinR, zMinR}, {xMaxR, yMaxR, zMaxR}, triStartRight, 0};
            int childIndexLeft = (int)nodes.size();
            nodes.push_back(childLeft);
            int childIndexRight = (int)nodes.size();
            nodes.push_back(childRight);

            parent.startIndex = childIndexLeft;
            nodes[parentIndex] = parent;
            RecordNode(depth, false);

            Split(childIndexLeft, triGlobalStart, numOnLeft, depth + 1);
            Split(childIndexRight, triGlobalStart + numOnLeft, numOnRight, depth + 1);
        } else {
            parent.startIndex = triGlobalStart;
            parent.triangleCount = triNum;
            nodes[parentIndex] = parent;
            RecordNode(depth, true, triNum);
        }
    }
};

struct OracleContext {
    Scene scene;
    RtParams params;
    bool haveParams = false;
    int W = 0, H = 0;
    int frame = 1;
    int threads = 1;
    int rowBegin = 0, rowEnd = -1; /* optional row window for bounded timing samples */
    std::vector<float> frameRender, accumulated;
    Counters counters;
    uint64_t pixelFrames = 0;
    double cpuMs = 0;
    char err[256] = {0};
};

} // namespace

extern "C" {

typedef struct OracleContext OracleContext;

int oracle_create(OracleContext** out)
{
    if (!out) return RT_ERR_INVALID_ARG;
    *out = new OracleContext();
    return RT_OK;
}
void oracle_destroy(OracleContext* ctx) { delete ctx; }
const char* oracle_last_error(const OracleContext* ctx) { return ctx ? ctx->err : ""; }

int oracle_set_threads(OracleContext* ctx, int n)
{
    if (!ctx || n < 1) return RT_ERR_INVALID_ARG;
    ctx->threads = n;
    return RT_OK;
}
/* Restrict rendering to image rows [row_begin,row_end) (others untouched); row_end<0 = all */
int oracle_set_row_window(OracleContext* ctx, int row_begin, int row_end)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    ctx->rowBegin = row_begin;
    ctx->rowEnd = row_end;
    return RT_OK;
}
int oracle_resize(OracleContext* ctx, int w, int h) /* RCM:126-141 */
{
    if (!ctx || w <= 0 || h <= 0) return RT_ERR_INVALID_ARG;
    ctx->W = w; ctx->H = h;
    ctx->frameRender.assign((size_t)w * h * 4, 0.0f);
    ctx->accumulated.assign((size_t)w * h * 4, 0.0f);
    return RT_OK;
}
int oracle_upload_scene(OracleContext* ctx, const RtModel* models, int n_models, const RtTriangle* tris, int n_tris,
                        const RtBVHNode* nodes, int n_nodes, const RtSphere* spheres, int n_spheres) /* RCM:143-161 */
{
    if (!ctx || n_models < 0 || n_tris < 0 || n_nodes < 0 || n_spheres < 0) return RT_ERR_INVALID_ARG;
    ctx->scene.models.assign(models, models + n_models);
    ctx->scene.triangles.assign(tris, tris + n_tris);
    ctx->scene.nodes.assign(nodes, nodes + n_nodes);
    ctx->scene.spheres.assign(spheres, spheres + n_spheres);
    return RT_OK;
}
int oracle_update_models(OracleContext* ctx, const RtModel* models, int n) /* RCM:192-204 */
{
    if (!ctx || n != (int)ctx->scene.models.size()) return RT_ERR_INVALID_ARG;
    ctx->scene.models.assign(models, models + n);
    return RT_OK;
}
int oracle_update_spheres(OracleContext* ctx, const RtSphere* spheres, int n)
{
    if (!ctx || n != (int)ctx->scene.spheres.size()) return RT_ERR_INVALID_ARG;
    ctx->scene.spheres.assign(spheres, spheres + n);
    return RT_OK;
}
int oracle_set_params(OracleContext* ctx, const RtParams* p) /* RCM:163-190 */
{
    if (!ctx || !p) return RT_ERR_INVALID_ARG;
    if (p->abi_version != RT_ABI_VERSION || p->struct_size != sizeof(RtParams)) return RT_ERR_ABI_MISMATCH;
    ctx->params = *p;
    ctx->frame = p->frame;
    ctx->haveParams = true;
    return RT_OK;
}
int oracle_reset_accumulation(OracleContext* ctx) /* RCM:69-76, RCC:26-32 */
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    std::fill(ctx->accumulated.begin(), ctx->accumulated.end(), 0.0f);
    ctx->frame = 1;
    return RT_OK;
}

/* RCC:10-24 for every pixel of the row window */
int oracle_render_frame(OracleContext* ctx)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!ctx->haveParams || ctx->W == 0) return RT_ERR_STATE;
    RtParams P = ctx->params;
    P.frame = ctx->frame;
    const int W = ctx->W, H = ctx->H;
    int r0 = ctx->rowBegin < 0 ? 0 : ctx->rowBegin;
    int r1 = (ctx->rowEnd < 0 || ctx->rowEnd > H) ? H : ctx->rowEnd;
    if (r0 > r1) r0 = r1;
    auto t0 = std::chrono::steady_clock::now();

    std::atomic<int> nextRow(r0);
    int nthreads = ctx->threads;
    std::vector<Counters> perThread(nthreads);
    auto worker = [&](int tid) {
        Counters& st = perThread[tid];
        for (;;) {
            int y = nextRow.fetch_add(1);
            if (y >= r1) break;
            for (int x = 0; x < W; x++) {
                /* RCC:13-15 */
                float2 uv = {rt_div((float)(uint32_t)x, (float)(uint32_t)W - 1.0f), rt_div((float)(uint32_t)y, (float)(uint32_t)H - 1.0f)};
                float3 pixelCol = RayTracePixel(ctx->scene, P, uv, (uint32_t)W, (uint32_t)H, st);
                size_t o = ((size_t)y * W + x) * 4;
                /* RCC:18 */
                ctx->frameRender[o + 0] = pixelCol.x;
                ctx->frameRender[o + 1] = pixelCol.y;
                ctx->frameRender[o + 2] = pixelCol.z;
                ctx->frameRender[o + 3] = 1.0f;
                if (P.accumulate) { /* RCC:20-23 */
                    ctx->accumulated[o + 0] += pixelCol.x;
                    ctx->accumulated[o + 1] += pixelCol.y;
                    ctx->accumulated[o + 2] += pixelCol.z;
                    ctx->accumulated[o + 3] += 1.0f;
                }
            }
        }
    };
    if (nthreads == 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; t++) pool.emplace_back(worker, t);
        for (auto& t : pool) t.join();
    }
    for (auto& c : perThread) ctx->counters.add(c);
    ctx->pixelFrames += (uint64_t)(r1 - r0) * W;
    ctx->cpuMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (P.accumulate) ctx->frame++; /* RCM:94 */
    return RT_OK;
}
int oracle_render_frames(OracleContext* ctx, int n)
{
    for (int i = 0; i < n; i++) {
        int rc = oracle_render_frame(ctx);
        if (rc) return rc;
    }
    return RT_OK;
}
int oracle_get_frame(const OracleContext* ctx) { return ctx ? ctx->frame : RT_ERR_INVALID_ARG; }

int oracle_read_frame(OracleContext* ctx, float* rgba, size_t bytes)
{
    if (!ctx || !rgba || bytes != ctx->frameRender.size() * 4) return RT_ERR_INVALID_ARG;
    memcpy(rgba, ctx->frameRender.data(), bytes);
    return RT_OK;
}
int oracle_read_accumulated(OracleContext* ctx, float* rgba, size_t bytes)
{
    if (!ctx || !rgba || bytes != ctx->accumulated.size() * 4) return RT_ERR_INVALID_ARG;
    memcpy(rgba, ctx->accumulated.data(), bytes);
    return RT_OK;
}
/* Display.shader:42-47 with RayTraceDisplay.cs:14-16 choosing texture and Frame */
int oracle_display(OracleContext* ctx, int frame, int use_accumulated, float* rgba, size_t bytes)
{
    if (!ctx || !rgba || frame == 0) return RT_ERR_INVALID_ARG;
    const std::vector<float>& src = use_accumulated ? ctx->accumulated : ctx->frameRender;
    if (bytes != src.size() * 4) return RT_ERR_INVALID_ARG;
    for (size_t i = 0; i < src.size(); i++) rgba[i] = rt_div(src[i], (float)frame); /* float4 / int */
    return RT_OK;
}
int oracle_display_srgb8(OracleContext* ctx, int frame, int use_accumulated, int flip_y, uint8_t* rgba8, size_t bytes)
{
    if (!ctx || !rgba8 || frame == 0) return RT_ERR_INVALID_ARG;
    const std::vector<float>& src = use_accumulated ? ctx->accumulated : ctx->frameRender;
    if (bytes != src.size()) return RT_ERR_INVALID_ARG;
    const float inv = rt_rcp((float)frame);
    for (int y = 0; y < ctx->H; y++)
        for (int x = 0; x < ctx->W; x++) {
            size_t i = ((size_t)y * ctx->W + x) * 4;
            size_t o = ((size_t)(flip_y ? ctx->H - 1 - y : y) * ctx->W + x) * 4;
            rgba8[o + 0] = (uint8_t)rt_srgb8(src[i + 0] * inv);
            rgba8[o + 1] = (uint8_t)rt_srgb8(src[i + 1] * inv);
            rgba8[o + 2] = (uint8_t)rt_srgb8(src[i + 2] * inv);
            rgba8[o + 3] = 255;
        }
    return RT_OK;
}
int oracle_write_accumulated(OracleContext* ctx, const float* rgba, size_t bytes)
{
    if (!ctx || !rgba || bytes != ctx->accumulated.size() * 4) return RT_ERR_INVALID_ARG;
    memcpy(ctx->accumulated.data(), rgba, bytes);
    return RT_OK;
}
int oracle_reset_counters(OracleContext* ctx)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    ctx->counters = Counters();
    ctx->pixelFrames = 0;
    ctx->cpuMs = 0;
    return RT_OK;
}
int oracle_get_counters(OracleContext* ctx, RtCounters* out)
{
    if (!ctx || !out) return RT_ERR_INVALID_ARG;
    out->segments = ctx->counters.segments;
    out->innerSteps = ctx->counters.innerSteps;
    out->leafSteps = ctx->counters.leafSteps;
    out->triTests = ctx->counters.triTests;
    out->sphereTests = ctx->counters.sphereTests;
    out->modelVisits = ctx->counters.modelVisits;
    out->pixelFrames = ctx->pixelFrames;
    out->gpuMs = ctx->cpuMs;
    return RT_OK;
}

/* BVH.cs:26-87 (constructor) */
int oracle_build_bvh(const float* verts, const float* normals, int n_verts, const int32_t* indices, int n_indices,
                     int quality, RtBVHNode* out_nodes, int* out_n_nodes, RtTriangle* out_tris, RtBvhStats* out_stats)
{
    if (!verts || !normals || !indices || !out_nodes || !out_n_nodes || !out_tris || n_indices < 0 || n_indices % 3) return RT_ERR_INVALID_ARG;
    for (int i = 0; i < n_indices; i++)
        if (indices[i] < 0 || indices[i] >= n_verts) return RT_ERR_INVALID_ARG;
    auto t0 = std::chrono::steady_clock::now();
    BvhBuilder b;
    b.quality = quality;
    memset(&b.stats, 0, sizeof(b.stats));
    b.stats.leafDepthMin = INT32_MAX;
    b.stats.leafMinTriCount = INT32_MAX;
    b.stats.quality = quality;
    const float FMAX = BvhBuilder::FMAX;
    int ntri = n_indices / 3;
    b.tris.resize(ntri);

    float xMin = FMAX, xMax = -FMAX, yMin = FMAX, yMax = -FMAX, zMin = FMAX, zMax = -FMAX;
    for (int i = 0; i < n_indices; i += 3) { /* BVH:44-59 */
        const float* a = verts + 3 * indices[i + 0];
        const float* bb = verts + 3 * indices[i + 1];
        const float* c = verts + 3 * indices[i + 2];
        BVHTriangle tri;
        tri.CentreX = (a[0] + bb[0] + c[0]) / 3;
        tri.CentreY = (a[1] + bb[1] + c[1]) / 3;
        tri.CentreZ = (a[2] + bb[2] + c[2]) / 3;
        tri.MinX = min3(a[0], bb[0], c[0]); tri.MinY = min3(a[1], bb[1], c[1]); tri.MinZ = min3(a[2], bb[2], c[2]);
        tri.MaxX = max3(a[0], bb[0], c[0]); tri.MaxY = max3(a[1], bb[1], c[1]); tri.MaxZ = max3(a[2], bb[2], c[2]);
        tri.Index = i;
        b.tris[i / 3] = tri;
        if (tri.MinX < xMin) xMin = tri.MinX;
        if (tri.MinY < yMin) yMin = tri.MinY;
        if (tri.MinZ < zMin) zMin = tri.MinZ;
        if (tri.MaxX > xMax) xMax = tri.MaxX;
        if (tri.MaxY > yMax) yMax = tri.MaxY;
        if (tri.MaxZ > zMax) zMax = tri.MaxZ;
    }
    RtBVHNode root = {{xMin, yMin, zMin}, {xMax, yMax, zMax}, -1, -1}; /* BVH:61 */
    b.nodes.push_back(root);
    if (quality == RT_BVH_QUALITY_DISABLED) {
        b.nodes[0].startIndex = 0;
        b.nodes[0].triangleCount = ntri;
    } else {
        b.Split(0, 0, ntri, 0);
    }
    for (int i = 0; i < ntri; i++) { /* BVH:69-80 */
        int base = b.tris[i].Index;
        RtTriangle& t = out_tris[i];
        for (int k = 0; k < 3; k++) {
            t.posA[k] = verts[3 * indices[base + 0] + k];
            t.posB[k] = verts[3 * indices[base + 1] + k];
            t.posC[k] = verts[3 * indices[base + 2] + k];
            t.normA[k] = normals[3 * indices[base + 0] + k];
            t.normB[k] = normals[3 * indices[base + 1] + k];
            t.normC[k] = normals[3 * indices[base + 2] + k];
        }
    }
    /* out_nodes holds 2*max(1,ntri) nodes (rt_abi.h).  BVH.cs itself would go on (List<Node>) and emit a
     * tree with empty leaves when every split cost overflows — malformed for the shader (RC:246); both
     * builders refuse it the same way. */
    if (b.nodes.size() > 2 * (size_t)(ntri > 0 ? ntri : 1) || (ntri > 0 && b.stats.leafMinTriCount == 0)) {
        *out_n_nodes = 0;
        return RT_ERR_SCENE;
    }
    memcpy(out_nodes, b.nodes.data(), b.nodes.size() * sizeof(RtBVHNode));
    *out_n_nodes = (int)b.nodes.size();
    if (out_stats) {
        *out_stats = b.stats;
        out_stats->timeMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RT_OK;
}

/* RCM:185-188.  Mathf.Tan / Mathf.Deg2Rad are float: tan evaluated in double and
 * rounded, as UnityEngine.Mathf does ((float)Math.Tan(f)). */
int oracle_camera_view_params(float fov_deg, float aspect, float focus_distance, float out[3])
{
    const float Deg2Rad = 0.0174532924f;
    float planeHeight = focus_distance * (float)tan((double)(fov_deg * 0.5f * Deg2Rad)) * 2;
    float planeWidth = planeHeight * aspect;
    out[0] = planeWidth; out[1] = planeHeight; out[2] = focus_distance;
    return RT_OK;
}

/* ---------------------------------------------- function-level entry points
 * for known-answer tests and for CPU<->GPU comparison of the primitives */
uint32_t oracle_next_random(uint32_t* state) { return rt_next_random(state); }
float oracle_random_value(uint32_t* state) { return rt_random_value(state); }
void oracle_random_direction(uint32_t* state, float out[3])
{
    float3 d = RandomDirection(state);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
void oracle_random_point_in_circle(uint32_t* state, float out[2])
{
    float2 p = RandomPointInCircle(state);
    out[0] = p.x; out[1] = p.y;
}
float oracle_ray_box(const float pos[3], const float dir[3], const float bmin[3], const float bmax[3])
{
    Ray r = CreateRay(f3(pos), f3(dir), rt_v3s(1), 0);
    return RayBoundingBoxDst(r, f3(bmin), f3(bmax));
}
/* out: didHit, isBackface, dst, normal.xyz */
void oracle_ray_triangle(const float pos[3], const float dir[3], const RtTriangle* tri, int cull, float out[6])
{
    Ray r = CreateRay(f3(pos), f3(dir), rt_v3s(1), 0);
    TriangleHitInfo h = RayTriangle(r, *tri, cull != 0);
    out[0] = h.didHit; out[1] = h.isBackface; out[2] = h.dst; out[3] = h.normal.x; out[4] = h.normal.y; out[5] = h.normal.z;
}
/* out: didHit, isBackface, dst, normal.xyz */
void oracle_ray_sphere(const float pos[3], const float dir[3], const float centre[3], float radius, float out[6])
{
    ModelHitInfo h = RaySphere(f3(pos), f3(dir), f3(centre), radius);
    out[0] = h.didHit; out[1] = h.isBackface; out[2] = h.dst; out[3] = h.normal.x; out[4] = h.normal.y; out[5] = h.normal.z;
}
float oracle_reflectance(const float inDir[3], const float normal[3], float iorA, float iorB)
{
    return CalculateReflectance(f3(inDir), f3(normal), iorA, iorB);
}
void oracle_refract(const float inDir[3], const float normal[3], float iorA, float iorB, float out[3])
{
    float3 r = Refract(f3(inDir), f3(normal), iorA, iorB);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void oracle_environment_light(const RtParams* p, const float dir[3], float out[3])
{
    float3 c = GetEnvironmentLight(*p, f3(dir));
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
void oracle_material_colour(const RtMaterial* m, const float pos[3], const float normal[3], int isSpecular, float out[3])
{
    float3 c = GetMaterialColour(*m, f3(pos), f3(normal), isSpecular != 0);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
/* Closest hit of one world ray against the uploaded scene (CalculateRayCollision).
 * out: didHit, isBackface, dst, normal.xyz, pos.xyz, material.flag */
void oracle_ray_collision(OracleContext* ctx, const float pos[3], const float dir[3], float out[10])
{
    Counters c;
    Ray r = CreateRay(f3(pos), f3(dir), rt_v3s(1), 0);
    ModelHitInfo h = CalculateRayCollision(ctx->scene, r, false, c);
    out[0] = h.didHit; out[1] = h.isBackface; out[2] = h.dst;
    out[3] = h.normal.x; out[4] = h.normal.y; out[5] = h.normal.z;
    out[6] = h.pos.x; out[7] = h.pos.y; out[8] = h.pos.z; out[9] = (float)h.material.flag;
}
/* Brute-force closest hit (every triangle of every model, same RayTriangle, same
 * strict '<' in buffer order) — the property-test partner of the BVH traversal.
 * out: didHit, dst */
void oracle_ray_collision_bruteforce(OracleContext* ctx, const float pos[3], const float dir[3], float out[2])
{
    const Scene& sc = ctx->scene;
    float best = RT_INF;
    bool any = false;
    for (size_t s = 0; s < sc.spheres.size(); s++) {
        ModelHitInfo h = RaySphere(f3(pos), f3(dir), f3(sc.spheres[s].centre), sc.spheres[s].radius);
        if (h.didHit && h.dst < best) { best = h.dst; any = true; }
    }
    for (size_t m = 0; m < sc.models.size(); m++) {
        const RtModel& model = sc.models[m];
        Ray lr;
        lr.pos = rt_mul_point(model.worldToLocal, f3(pos), 1);
        lr.dir = rt_mul_point(model.worldToLocal, f3(dir), 0);
        lr.invDir = rt_v3(rt_rcp(lr.dir.x), rt_rcp(lr.dir.y), rt_rcp(lr.dir.z));
        bool cull = model.material.flag != RT_MATERIAL_GLASS;
        /* all triangles of the model's mesh = the triangle range covered by its BVH */
        int lo = INT32_MAX, hi = -1;
        std::vector<int> st; st.push_back(model.nodeOffset);
        while (!st.empty()) {
            const RtBVHNode& n = sc.nodes[st.back()]; st.pop_back();
            if (n.triangleCount > 0) {
                if (n.startIndex < lo) lo = n.startIndex;
                if (n.startIndex + n.triangleCount > hi) hi = n.startIndex + n.triangleCount;
            } else { st.push_back(model.nodeOffset + n.startIndex); st.push_back(model.nodeOffset + n.startIndex + 1); }
        }
        for (int t = lo; t < hi; t++) {
            TriangleHitInfo h = RayTriangle(lr, sc.triangles[model.triOffset + t], cull);
            if (h.didHit && h.dst < best) { best = h.dst; any = true; }
        }
    }
    out[0] = any; out[1] = best;
}
/* One pixel, no framebuffer: RCC:15 + RC:545-582 */
void oracle_trace_pixel(OracleContext* ctx, int x, int y, int frame, float out[3])
{
    RtParams P = ctx->params;
    P.frame = frame;
    Counters c;
    float2 uv = {rt_div((float)(uint32_t)x, (float)(uint32_t)ctx->W - 1.0f), rt_div((float)(uint32_t)y, (float)(uint32_t)ctx->H - 1.0f)};
    float3 col = RayTracePixel(ctx->scene, P, uv, (uint32_t)ctx->W, (uint32_t)ctx->H, c);
    out[0] = col.x; out[1] = col.y; out[2] = col.z;
}
/* oracle_trace_pixel with the work log switched on; returns the log's length (bytes copied: min(len, cap)). */
int oracle_trace_pixel_schedule(OracleContext* ctx, int x, int y, int frame, uint8_t* buf, int cap)
{
    std::vector<uint8_t> log;
    g_schedTrace = &log;
    float out[3];
    oracle_trace_pixel(ctx, x, y, frame, out);
    g_schedTrace = nullptr;
    int n = (int)log.size();
    if (buf) memcpy(buf, log.data(), (size_t)(n < cap ? n : cap));
    return n;
}
/* oracle_trace_pixel with the node log switched on; returns the log's length in entries (copied: min(len, cap)). */
int oracle_trace_pixel_nodes(OracleContext* ctx, int x, int y, int frame, uint32_t* buf, int cap)
{
    std::vector<uint32_t> log;
    g_nodeTrace = &log;
    float out[3];
    oracle_trace_pixel(ctx, x, y, frame, out);
    g_nodeTrace = nullptr;
    int n = (int)log.size();
    if (buf) memcpy(buf, log.data(), (size_t)(n < cap ? n : cap) * 4);
    return n;
}
/* rt_math.h primitives over arrays. op: 0 log 1 exp 2 sin 3 cos 4 sqrt 5 pow(x,y) 6 div(x/y) 7 smoothstep(0,y,x) */
void oracle_math_eval(int op, const float* x, const float* y, float* out, int n)
{
    for (int i = 0; i < n; i++) {
        switch (op) {
        case 0: out[i] = rt_log(x[i]); break;
        case 1: out[i] = rt_exp(x[i]); break;
        case 2: out[i] = rt_sin(x[i]); break;
        case 3: out[i] = rt_cos(x[i]); break;
        case 4: out[i] = rt_sqrt(x[i]); break;
        case 5: out[i] = rt_pow(x[i], y[i]); break;
        case 6: out[i] = rt_div(x[i], y[i]); break;
        case 7: out[i] = rt_smoothstep(0.0f, y[i], x[i]); break;
        case 8: out[i] = rt_rsqrt(x[i]); break;
        case 9: out[i] = rt_rcp(x[i]); break;
        default: out[i] = 0; break;
        }
    }
}
#ifndef RT_MATH_IEEE
const char* oracle_version(void) { return "rt_oracle (CPU restatement, test infrastructure) abi=1"; }
#else
const char* oracle_version(void) { return "rt_oracle (CPU restatement, test infrastructure) abi=1 RT_MATH_IEEE"; }
#endif

} /* extern "C" */
