/*
 * ref_driver.h — the dispatcher around the compiled reference shader text (oracle/_ref/libref.so).
 * TEST INFRASTRUCTURE ONLY.  Appended by oracle/make_ref.py to the translation unit that holds the reference's
 * RayCommon.hlsl + RayCompute.compute (namespace hlsl_ref); this file is the part of RayComputeManager.cs that
 * binds buffers, sets the uniforms and dispatches (RCM:126-204), nothing of the path itself.
 *
 *   ref_upload_scene      ≙ ComputeBuffer.SetData + SetBuffer(ModelInfo / Triangles / Nodes)   RCM:143-161, 192-204
 *   ref_set_params        ≙ SetInt / SetFloat / SetVector / SetMatrix                          RCM:163-190
 *   ref_render_frame      ≙ Dispatch(RayTrace) over ceil(W/8) x ceil(H/8) groups               RCM:90, RCC:10-24
 *   ref_reset_accumulation≙ Dispatch(ResetAccumulated)                                         RCM:69-76, RCC:26-32
 * plus function-level entry points so that every reference function can be compared with the oracle's restatement
 * on its own.  The reference has no sphere buffer (RaySphere's only call is commented out, RC:341): scenes with
 * analytic spheres are refused, RaySphere itself is exposed at function level.  -DREF_SPHERES (libref_spheres.so,
 * make_ref.py --spheres): the translation unit carries the declared semantic rewrite S1 — a `Spheres` buffer tested in
 * front of the model loop through the reference's own RaySphere — and this file binds that buffer.
 */
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "../include/rt_abi.h"

namespace hlsl_ref {
thread_local int64_t g_ref_stats[2] = {0, 0};
thread_local int64_t g_ref_collisions = 0;
}

namespace {
using namespace hlsl_ref;

static_assert(sizeof(hfloat) == 4, "hfloat is one binary32");
static_assert(sizeof(RayTracingMaterial) == sizeof(RtMaterial) && sizeof(RtMaterial) == 88, "RC:64-76");
static_assert(sizeof(Model) == sizeof(RtModel) && sizeof(RtModel) == 224, "RC:78-85");
static_assert(sizeof(Triangle) == sizeof(RtTriangle) && sizeof(RtTriangle) == 72, "RC:49-53");
static_assert(sizeof(BVHNode) == sizeof(RtBVHNode) && sizeof(RtBVHNode) == 32, "RC:87-95");
static_assert(sizeof(float4) == 16 && sizeof(float4x4) == 64, "texel / matrix");
#ifdef REF_SPHERES
static_assert(sizeof(Sphere) == sizeof(RtSphere) && sizeof(RtSphere) == 104, "S1: float3 centre, float radius, RayTracingMaterial");
#endif

struct RefContext {
    std::vector<RtModel> models;
    std::vector<RtTriangle> triangles;
    std::vector<RtBVHNode> nodes;
    std::vector<RtSphere> spheres;
    std::vector<float4> frameRender, accumulated;
    RtParams params = {};
    bool haveParams = false;
    int W = 0, H = 0, frame = 1, threads = 1;
    int rowBegin = -1, rowEnd = -1; /* ref_set_row_window: dispatch only the thread ids of these rows (checkers of large images) */
    int64_t triTests = 0, boxTests = 0, segments = 0;
    uint64_t pixelFrames = 0;
    char err[256] = {0};
};
/* the shader's uniforms and buffers are globals of the translation unit: one context at a time */
std::mutex g_bind_mutex;

int fail(RefContext* c, int rc, const char* msg)
{
    if (c) snprintf(c->err, sizeof(c->err), "%s", msg);
    return rc;
}

/* height of the tallest BVH reachable from a model root; RC:239's stack holds 32 entries and nothing checks it */
int max_live_stack_entries(const RefContext* c)
{
    int worst = 0;
    for (const RtModel& m : c->models) {
        std::vector<std::pair<int, int>> st; /* node, depth */
        st.push_back({m.nodeOffset, 1});
        while (!st.empty()) {
            auto [n, d] = st.back();
            st.pop_back();
            if (n < 0 || n >= (int)c->nodes.size()) return 1 << 20;
            if (d > worst) worst = d;
            if (d > 64) return d;
            const RtBVHNode& node = c->nodes[n];
            if (node.triangleCount <= 0) {
                st.push_back({m.nodeOffset + node.startIndex, d + 1});
                st.push_back({m.nodeOffset + node.startIndex + 1, d + 1});
            }
        }
    }
    return worst;
}

void bind(RefContext* c)
{
    ModelInfo.data = reinterpret_cast<const Model*>(c->models.data());
    ModelInfo.count = (int64_t)c->models.size();
    Triangles.data = reinterpret_cast<const Triangle*>(c->triangles.data());
    Triangles.count = (int64_t)c->triangles.size();
    Nodes.data = reinterpret_cast<const BVHNode*>(c->nodes.data());
    Nodes.count = (int64_t)c->nodes.size();
#ifdef REF_SPHERES
    Spheres.data = reinterpret_cast<const Sphere*>(c->spheres.data());
    Spheres.count = (int64_t)c->spheres.size();
    sphereCount = (int)c->spheres.size();
#endif
    modelCount = (int)c->models.size();       /* RCM:156 */
    triangleCount = (int)c->triangles.size(); /* RCM:157 (unused by the kernel) */
    FrameRender.data = c->frameRender.data();
    FrameRender.width = (uint)c->W;
    FrameRender.height = (uint)c->H;
    AccumulatedRender.data = c->accumulated.data();
    AccumulatedRender.width = (uint)c->W;
    AccumulatedRender.height = (uint)c->H;
    Resolution = uint2((uint)c->W, (uint)c->H); /* RCM:139 */
    const RtParams& p = c->params;
    MaxBounceCount = p.maxBounceCount; /* RCM:165-180 */
    NumRaysPerPixel = p.numRaysPerPixel;
    Frame = c->frame;
    renderSeed = p.renderSeed;
    UseSky = p.useSky;
    accumulate = p.accumulate != 0;
    DefocusStrength = hfloat(p.defocusStrength);
    DivergeStrength = hfloat(p.divergeStrength);
    SunFocus = hfloat(p.sunFocus);
    SunIntensity = hfloat(p.sunIntensity);
    SunColour = float3(hfloat(p.sunColour[0]), hfloat(p.sunColour[1]), hfloat(p.sunColour[2]));
    dirToSun = float3(hfloat(p.dirToSun[0]), hfloat(p.dirToSun[1]), hfloat(p.dirToSun[2]));
    ViewParams = float3(hfloat(p.viewParams[0]), hfloat(p.viewParams[1]), hfloat(p.viewParams[2])); /* RCM:188 */
    for (int i = 0; i < 16; i++) CamLocalToWorldMatrix.m[i] = hfloat(p.camLocalToWorld[i]);             /* RCM:189 */
}

/* Dispatch(kernel, ceil(W/8), ceil(H/8), 1) with [numthreads(8,8,1)]: every thread id of the padded grid */
template <class K> void dispatch(RefContext* c, K kernel)
{
    const int gw = (c->W + 7) / 8 * 8;
    int gh = (c->H + 7) / 8 * 8, g0 = 0;
    if (c->rowBegin >= 0) g0 = c->rowBegin < gh ? c->rowBegin : gh;
    if (c->rowEnd >= 0 && c->rowEnd < gh) gh = c->rowEnd > g0 ? c->rowEnd : g0;
    std::atomic<int> nextRow(g0);
    std::mutex sum;
    auto worker = [&]() {
        g_ref_stats[0] = g_ref_stats[1] = 0;
        g_ref_collisions = 0;
        for (;;) {
            int y = nextRow.fetch_add(1);
            if (y >= gh) break;
            for (int x = 0; x < gw; x++) kernel(uint3((uint)x, (uint)y, 0u));
        }
        std::lock_guard<std::mutex> g(sum);
        c->triTests += g_ref_stats[0];
        c->boxTests += g_ref_stats[1];
        c->segments += g_ref_collisions;
    };
    if (c->threads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < c->threads; t++) pool.emplace_back(worker);
        for (auto& t : pool) t.join();
    }
}
float3 f3(const float* p) { return float3(hfloat(p[0]), hfloat(p[1]), hfloat(p[2])); }
void put3(float* o, float3 v) { o[0] = v.x.v; o[1] = v.y.v; o[2] = v.z.v; }
} // namespace

extern "C" {
typedef struct RefContext RefContext;

int ref_create(RefContext** out)
{
    if (!out) return RT_ERR_INVALID_ARG;
    *out = new RefContext();
    return RT_OK;
}
void ref_destroy(RefContext* c) { delete c; }
const char* ref_last_error(const RefContext* c) { return c ? c->err : ""; }
int ref_set_threads(RefContext* c, int n)
{
    if (!c || n < 1) return RT_ERR_INVALID_ARG;
    c->threads = n;
    return RT_OK;
}
/* the dispatcher's (not the reference's): thread ids of rows [row_begin, row_end) only; negative = all */
int ref_set_row_window(RefContext* c, int row_begin, int row_end)
{
    if (!c) return RT_ERR_INVALID_ARG;
    c->rowBegin = row_begin;
    c->rowEnd = row_end;
    return RT_OK;
}
int ref_resize(RefContext* c, int w, int h) /* RCM:126-133 */
{
    if (!c || w <= 0 || h <= 0) return RT_ERR_INVALID_ARG;
    c->W = w;
    c->H = h;
    c->frameRender.assign((size_t)w * h, float4(0));
    c->accumulated.assign((size_t)w * h, float4(0));
    return RT_OK;
}
int ref_upload_scene(RefContext* c, const RtModel* models, int n_models, const RtTriangle* tris, int n_tris,
                     const RtBVHNode* nodes, int n_nodes, const RtSphere* spheres, int n_spheres)
{
    if (!c || n_models < 0 || n_tris < 0 || n_nodes < 0) return RT_ERR_INVALID_ARG;
#ifdef REF_SPHERES
    if (n_spheres < 0 || (n_spheres && !spheres)) return RT_ERR_INVALID_ARG;
    c->spheres.assign(spheres, spheres + n_spheres);
#else
    if (n_spheres != 0) return fail(c, RT_ERR_SCENE, "the reference has no sphere buffer (RC:341 is commented out)");
    (void)spheres;
#endif
    c->models.assign(models, models + n_models);
    c->triangles.assign(tris, tris + n_tris);
    c->nodes.assign(nodes, nodes + n_nodes);
    if (max_live_stack_entries(c) > 31) return fail(c, RT_ERR_SCENE, "BVH deeper than the reference's int stack[32] (RC:239) can hold");
    return RT_OK;
}
int ref_update_models(RefContext* c, const RtModel* models, int n) /* RCM:192-204 */
{
    if (!c || n != (int)c->models.size()) return RT_ERR_INVALID_ARG;
    c->models.assign(models, models + n);
    return RT_OK;
}
#ifdef REF_SPHERES
int ref_update_spheres(RefContext* c, const RtSphere* spheres, int n)
{
    if (!c || n != (int)c->spheres.size()) return RT_ERR_INVALID_ARG;
    c->spheres.assign(spheres, spheres + n);
    return RT_OK;
}
#else
int ref_update_spheres(RefContext* c, const RtSphere*, int n) { return (c && n == 0) ? RT_OK : RT_ERR_SCENE; }
#endif
int ref_set_params(RefContext* c, const RtParams* p)
{
    if (!c || !p) return RT_ERR_INVALID_ARG;
    if (p->abi_version != RT_ABI_VERSION || p->struct_size != sizeof(RtParams)) return RT_ERR_ABI_MISMATCH;
    c->params = *p;
    c->frame = p->frame;
    c->haveParams = true;
    return RT_OK;
}
int ref_reset_accumulation(RefContext* c) /* RCM:69-76 */
{
    if (!c || c->W == 0) return RT_ERR_STATE;
    std::lock_guard<std::mutex> g(g_bind_mutex);
    bind(c);
    dispatch(c, [](uint3 id) { ResetAccumulated(id); });
    c->frame = 1;
    return RT_OK;
}
int ref_render_frame(RefContext* c) /* RCM:84-95 */
{
    if (!c) return RT_ERR_INVALID_ARG;
    if (!c->haveParams || c->W == 0) return RT_ERR_STATE;
    std::lock_guard<std::mutex> g(g_bind_mutex);
    bind(c);
    dispatch(c, [](uint3 id) { RayTrace(id); });
    c->pixelFrames += (uint64_t)c->W * c->H;
    if (c->params.accumulate) c->frame++; /* RCM:94 */
    return RT_OK;
}
int ref_render_frames(RefContext* c, int n)
{
    for (int i = 0; i < n; i++) {
        int rc = ref_render_frame(c);
        if (rc) return rc;
    }
    return RT_OK;
}
int ref_get_frame(const RefContext* c) { return c ? c->frame : RT_ERR_INVALID_ARG; }
int ref_read_frame(RefContext* c, float* rgba, size_t bytes)
{
    if (!c || !rgba || bytes != c->frameRender.size() * 16) return RT_ERR_INVALID_ARG;
    memcpy(rgba, c->frameRender.data(), bytes);
    return RT_OK;
}
int ref_read_accumulated(RefContext* c, float* rgba, size_t bytes)
{
    if (!c || !rgba || bytes != c->accumulated.size() * 16) return RT_ERR_INVALID_ARG;
    memcpy(rgba, c->accumulated.data(), bytes);
    return RT_OK;
}
int ref_reset_counters(RefContext* c)
{
    if (!c) return RT_ERR_INVALID_ARG;
    c->triTests = c->boxTests = c->segments = 0;
    c->pixelFrames = 0;
    return RT_OK;
}
/* segments = calls of CalculateRayCollision; triTests = stats[0] (RC:254); innerSteps = stats[1] / 2 (RC:271 adds 2 per
 * inner node); the reference counts nothing else */
int ref_get_counters(RefContext* c, RtCounters* out)
{
    if (!c || !out) return RT_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    out->segments = (uint64_t)c->segments;
    out->triTests = (uint64_t)c->triTests;
    out->innerSteps = (uint64_t)c->boxTests / 2;
    out->pixelFrames = c->pixelFrames;
    return RT_OK;
}
#ifdef REF_SPHERES
const char* ref_version(void) { return "reference HLSL text + the declared sphere hook S1 (RC:341), compiled through oracle/ref_compat.h (test infrastructure)"; }
#elif !defined(RT_MATH_IEEE)
const char* ref_version(void) { return "reference HLSL text compiled through oracle/ref_compat.h (test infrastructure)"; }
#else
const char* ref_version(void) { return "reference HLSL text compiled through oracle/ref_compat.h (test infrastructure) RT_MATH_IEEE"; }
#endif

/* ------------------------------------------------ function-level entry points (same shapes as oracle_*) */
uint32_t ref_next_random(uint32_t* state) { return NextRandom(*state); }
float ref_random_value(uint32_t* state) { return RandomValue(*state).v; }
void ref_random_direction(uint32_t* state, float out[3]) { put3(out, RandomDirection(*state)); }
void ref_random_point_in_circle(uint32_t* state, float out[2])
{
    float2 p = RandomPointInCircle(*state);
    out[0] = p.x.v;
    out[1] = p.y.v;
}
float ref_ray_box(const float pos[3], const float dir[3], const float bmin[3], const float bmax[3])
{
    Ray r = CreateRay(f3(pos), f3(dir), 1, 0);
    return RayBoundingBoxDst(r, f3(bmin), f3(bmax)).v;
}
void ref_ray_triangle(const float pos[3], const float dir[3], const RtTriangle* tri, int cull, float out[6])
{
    Ray r = CreateRay(f3(pos), f3(dir), 1, 0);
    Triangle t;
    memcpy(&t, tri, sizeof(t));
    TriangleHitInfo h = RayTriangle(r, t, cull != 0);
    out[0] = h.didHit; out[1] = h.isBackface; out[2] = h.dst.v;
    put3(out + 3, h.normal);
}
void ref_ray_sphere(const float pos[3], const float dir[3], const float centre[3], float radius, float out[6])
{
    ModelHitInfo h = RaySphere(f3(pos), f3(dir), f3(centre), hfloat(radius));
    out[0] = h.didHit; out[1] = h.isBackface; out[2] = h.dst.v;
    put3(out + 3, h.normal);
}
float ref_reflectance(const float inDir[3], const float normal[3], float iorA, float iorB)
{
    return CalculateReflectance(f3(inDir), f3(normal), hfloat(iorA), hfloat(iorB)).v;
}
void ref_refract(const float inDir[3], const float normal[3], float iorA, float iorB, float out[3])
{
    put3(out, Refract(f3(inDir), f3(normal), hfloat(iorA), hfloat(iorB)));
}
void ref_environment_light(const RtParams* p, const float dir[3], float out[3])
{
    std::lock_guard<std::mutex> g(g_bind_mutex);
    UseSky = p->useSky;
    SunFocus = hfloat(p->sunFocus);
    SunIntensity = hfloat(p->sunIntensity);
    SunColour = f3(p->sunColour);
    dirToSun = f3(p->dirToSun);
    put3(out, GetEnvironmentLight(f3(dir)));
}
void ref_material_colour(const RtMaterial* m, const float pos[3], const float normal[3], int isSpecular, float out[3])
{
    RayTracingMaterial mat;
    memcpy(&mat, m, sizeof(mat));
    put3(out, GetMaterialColour(mat, f3(pos), f3(normal), isSpecular != 0));
}
/* out: didHit, isBackface, dst, normal.xyz, pos.xyz, material.flag */
void ref_ray_collision(RefContext* c, const float pos[3], const float dir[3], float out[10])
{
    std::lock_guard<std::mutex> g(g_bind_mutex);
    bind(c);
    Ray r = CreateRay(f3(pos), f3(dir), 1, 0);
    ModelHitInfo h = CalculateRayCollision(r, false);
    out[0] = h.didHit; out[1] = h.isBackface; out[2] = h.dst.v;
    put3(out + 3, h.normal);
    put3(out + 6, h.pos);
    out[9] = (float)h.material.flag;
}
/* one thread of the RayTrace kernel (RCC:10-24) at `frame`; returns the texel it stored in FrameRender.  The accumulation
 * buffer is left as it was. */
void ref_trace_pixel(RefContext* c, int x, int y, int frame, float out[3])
{
    std::lock_guard<std::mutex> g(g_bind_mutex);
    int keep = c->frame;
    c->frame = frame;
    bind(c);
    c->frame = keep;
    accumulate = false;
    RayTrace(uint3((uint)x, (uint)y, 0u));
    put3(out, c->frameRender[(size_t)y * c->W + x].xyz());
}
} /* extern "C" */
