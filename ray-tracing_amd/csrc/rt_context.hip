/*
 * rt_context.hip — the C ABI of libraytrace_hip.so (include/rt_abi.h): context,
 * scene validation + re-layout for HBM, render-target ownership, launches.
 *
 * Replaces the call surface RayComputeManager.cs ("RCM") drives on Unity's
 * ComputeShader/ComputeBuffer objects; see rt_abi.h for the per-function map.
 * There is no CPU fallback: without a HIP device rt_create fails.
 */
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dlfcn.h>
#include <sys/mman.h>

#include <cmath>
#include <string>
#include <vector>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <unordered_map>

#include "rt_kernels.h"

/* Frames per fused launch: a budget, not a constant.  A launch ends with a tail as long as one pixel chain, so short
 * frames want many per launch (a rank that renders 1/8 of config 2: 0.108 ms per frame, 7.9 work items per resident wave at 16 frames);
 * long frames must not turn into launches that last seconds.  RT_FUSE_MIN frames always; more — up to RT_FUSE_MAX — while the launch
 * stays under RT_FUSE_TARGET_MS at the frame time MEASURED on the previous fused launches (stop event to stop event — or, for a launch
 * without a predecessor, its own start to stop — polled, never waited for).  BVH scenes (>= 1.3 ms per frame) stay at 16.  RT_FUSE_CAP=n
 * pins it. */
#define RT_FUSE_MIN 16
#define RT_FUSE_MAX 64
#define RT_FUSE_TARGET_MS 20.0
#define RT_FUSE_SLAB_BYTES ((size_t)1536 << 20) /* a staging slab: 16 bytes per pixel and frame; 1920x1080: 45 frames, 3840x2160: 16 (2.1 GB), an eighth of 1080p: 64 */
#include "rt_layout.h"
#define RT_VERSION_STRING "raytrace_hip gfx950 abi=1"

static thread_local char g_err[512] = "";

struct RtContext {
    int device = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    /* Second render stream (see launch_frames): while the context runs on its own stream, every
     * frame is launched as two kernels over disjoint halves of the tiles, one per stream, so that
     * the drain of one kernel overlaps the other instead of idling the chip. */
    hipStream_t sideStream = nullptr;
    bool twoStreams = true; /* RT_TWO_STREAMS=0: one kernel per launch on the main stream */
    char err[512] = {0};

    /* image */
    int W = 0, H = 0;
    int stripRows = 8, partIndex = 0, partCount = 1;
    int localRows = 0;
    float* ownFrame = nullptr;
    float* ownAccum = nullptr;
    size_t ownBytes = 0;
    float* boundFrame = nullptr;
    float* boundAccum = nullptr;

    /* scene (device) */
    float* dSpheres = nullptr; /* nSpheres x (c, r*r), then nSpheres x (c, |c|^2 - r*r) */
    float sphereBound = 0;     /* max over spheres of |c|^2 + r*r, rounded up */
    DMaterial* dMaterials = nullptr;
    DModel* dModels = nullptr;
    /* the traversal's records, addressed in 16-byte units (rt_layout.h decides where they lie) */
    unsigned char* dPairs = nullptr; /* pair space */
    unsigned char* dTris = nullptr;  /* triangle space; null when the layout keeps the triangles in the pair space (arena) */
    unsigned char* dNorms = nullptr; /* 12 bytes per unit of the triangle space */
    bool arenaLayout = false;
    RtLayout layout;                 /* RT_LAYOUT at rt_create */
    std::string layoutUsed = "dense";
    uint32_t* dBigLeaves = nullptr;
    DFilter* dFilters = nullptr;
    DChunk* dChunks = nullptr;  /* two-level model hierarchy, scenes with more than 64 models */
    int nChunks = 0, nFiltered = 0, extWords = 0;
    float filterMaxOrigin = 0.0f;
    std::vector<RtBVHNode> hRootChildren;
    std::vector<RtSphere> hSpheres; /* per model: the root's two children (for rt_update_models) */
    int nSpheres = 0, nModels = 0, nTris = 0, nPairs = 0;
    bool flatScene = false; /* every model root is a leaf: the FLAT kernel variant applies */
    int stackEntries = 1; /* deepest BVH of the scene = most entries a lane can push */
    int wavesPerGroup = 1; /* the BVH variants' workgroups: waves that share one LDS top-of-tree cache (plan_groups) */
    uint32_t hotUnits = 0; /* units [0, hotUnits) of the pair space are that cache's records */
    uint32_t poolSpinLimit = 1u << 16; /* the chain pool's watchdog (rt_kernels.h, pool_exchange); RT_POOL_FAULT=1 (test hook): 64 polls + the fault bit */
    int poolMinItems = 4; /* RT_POOL_MIN_ITEMS: (tile, frame) items per resident wave a launch needs to run as pooled workgroups (choose_variant) */
    int poolWaves = 1, poolCells = 0; /* the FLAT variant's workgroups: waves that share one LDS chain pool (rt_kernels.h, pool_exchange); poolCells = RT_POOL_CELLS or 0 = no pool */
    uint32_t travLimit = 1u << 20; /* traversal watchdog (rt_kernels.h, traverse): 64 x the steps one ray can take in this scene */
    bool haveScene = false;
    /* scene (host mirrors needed by rt_update_models) */
    std::vector<RtModel> hModels;
    std::vector<uint32_t> hRootCodes;
    std::vector<int32_t> hTriBase; /* per model: first unit of its triangles in the triangle space */

    /* uniforms */
    RtParams params;
    bool haveParams = false;
    int frame = 1;

    /* counters */
    unsigned long long* dCounters = nullptr;
    unsigned long long* dTileQueue = nullptr;      /* monotonic tile counters of the persistent kernel, one per render stream */
    unsigned long long tileQueueNext[2] = {0, 0};  /* value each counter will have when that stream's next launch starts */
    uint32_t* dTileCost = nullptr;  /* per tile: longest pixel chain (segments per frame) seen so far */
    /* queue position -> tile, longest chain first.  Two buffers: a re-sort writes the one no running kernel reads (the kernels in
     * flight keep the order they were launched with), so it does not have to join the streams */
    uint32_t* dTileOrder[2] = {nullptr, nullptr};
    uint32_t* dTileKey = nullptr;   /* the sort's snapshot of the costs (kernels in flight keep raising them) */
    int orderCur = 0;
    int orderTiles = 0;             /* tiles the two arrays are sized for; 0 = none */
    bool orderValid = false;
    long long framesSinceResize = 0;
    long long nextSortAt = 1;
    bool lptEnabled = true;
    int numCUs = 256;
    int occPerCU[16] = {0};
    size_t occBytes[16] = {0};
    bool verbose = false;
    /* rt_render_frame calls that arrive while earlier frames are still executing are held back (at most
     * fuseCap) and leave as ONE fused launch at the next call that needs them (flush_pending) */
    void* dPxCold = nullptr; /* pixel records of the resident waves: 2 launch slots (main / side stream) x pxColdWaves x 2 KB */
    long long pxColdWaves = 0;
    int frameGroupOverride = 0; /* RT_FRAME_GROUP: frames per (tile, frame group) item of fused launches (tuning hook) */
    bool coalesce = true;  /* RT_COALESCE=0: every rt_render_frame launches at once */
    int pending = 0;       /* frames [frame - pending, frame) requested but not launched yet */
    bool fuseFrames = true; /* rt_render_frames(n): up to fuseCap frames per launch (RT_FUSE_FRAMES=0: one launch per frame) */
    int fuseCap = RT_FUSE_MIN;  /* frames per fused launch right now (see RT_FUSE_MIN) */
    bool fuseCapPinned = false; /* RT_FUSE_CAP */
    /* the stop event of each fused launch's trace kernel; ms per frame = (stop - previous launch's stop) / frames once both have passed */
    struct FuseProbe { hipEvent_t start = nullptr, stop = nullptr; int frames = 0; int prev = -1; bool live = false; } fuseProbe[6];
    int fuseLast = -1;
    int gridOverride = 0; /* test hook: force the persistent grid size */
    bool stats = false;
    uint64_t pixelFrames = 0;
    /* pinned staging ring of the per-frame update calls (rt_update_models / rt_update_spheres): the
     * uploads are enqueued on the render stream, no host synchronisation */
    struct Staging { void* host = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool inFlight = false; };
    Staging staging[24]; /* one animated frame uses up to 8 (rt_update_models 4 + rt_update_spheres 4): three frames in flight */
    int stagingNext = 0;
    uint64_t updateUploads = 0, updateSkips = 0; /* diagnostics: uploads enqueued / calls that changed nothing */
    /* Launch tuner (scheduling only, results do not depend on it): the suspension threshold of the traversal loop has two
     * good values, 3/8 and 4/8 of the entrants, and which one is faster depends on how much of a segment is traversal
     * (config 3: 3/8 by 9 %, config 6: 4/8 by 2 %).  Once 48 frames have been rendered, fused launches of >= 8 frames
     * alternate between them, timed with events that are only polled, never waited for; after 3 samples each the
     * faster stays (RT_SUSPEND=3|4 pins it). */
    struct Tuner {
        int decided = 3;       /* value in use once `done` */
        bool done = false;
        double ms[2] = {0, 0}; /* accumulated ms per frame for suspendNum 3, 4 */
        int n[2] = {0, 0};
        int next = 0;          /* which candidate the next measured launch uses */
        struct Probe { hipEvent_t start = nullptr, stop = nullptr; int cand = 0, frames = 0; bool live = false; } probe[6];
    } tuner;
    /* per-frame colours of a fused launch (rt_device.h, KArgs::staging): two slabs, so that consecutive fused launches
     * can alternate between the context's two streams — launch k+1 starts while launch k drains (launch_frames) */
    float* dStaging[2] = {nullptr, nullptr};
    size_t stagingBytes[2] = {0, 0};
    bool stagingUnavailable = false; /* the slab could not be allocated at this image size: fused launches go out frame by frame (cleared by rt_resize) */
    int stagedNext = 0;              /* stream / slab of the next fused launch */
    bool fusedBehindFirstPart = true; /* RT_FUSED_BEHIND_FIRST_PART=0: plain alternation also after a two-part frame (A/B switch) */
    bool alternate = true;           /* fused launches alternate between the two streams; off while the second slab does not fit (until the next rt_resize) */
    bool alternateWanted = true;     /* RT_ALTERNATE=0: every fused launch on the main stream (round-3 behaviour) */
    /* ---- everything the context's two render streams wait for ACROSS each other, in one place (the rules: LaunchOrder's functions below)
     * s = 0 the main stream, 1 the side stream */
    struct Order {
        hipEvent_t evFork = nullptr, evJoin = nullptr;
        bool sideDirty = false; /* the side stream holds work the main stream has not been ordered after */
        bool needFork = true;   /* the main stream holds non-render work the side stream must follow */
        hipEvent_t evSort = nullptr;    /* after the last sort of the tile order; sortPending[s]: stream s has not been ordered after it yet */
        bool sortPending[2] = {false, false};
        hipEvent_t evOrderRetire[2] = {nullptr, nullptr}; /* per order buffer: recorded on the OTHER stream when the buffer went out of use; its next rewrite waits for it */
        bool retireValid[2] = {false, false};
        /* the accumulation buffer is added to in launch order whichever stream a launch runs on: evAccWriter[s] marks the last kernel on
         * stream s that writes it; accWriterPending[s] = the other stream has not been ordered after it yet; accWriterFull[s] = that kernel
         * touches every pixel (an accumulate kernel, a one-part frame; false = one half of a two-part frame) */
        hipEvent_t evAccWriter[2] = {nullptr, nullptr};
        bool accWriterPending[2] = {false, false};
        bool accWriterFull[2] = {false, false};
        /* the last WHOLE-IMAGE writer of a stream is tracked on its own: a half kernel launched behind it on the same stream re-records
         * evAccWriter[s] with full = false, and the other stream's half of that very frame would then no longer wait for the whole-image
         * writer (rt_render_frames(17): 16 fused frames + 1 two-part frame; found by tools/soak.py, round 4) */
        hipEvent_t evAccFull[2] = {nullptr, nullptr};
        bool accFullPending[2] = {false, false};
    } ord;
    int lastLaunched = 0;            /* frames the last launch_frames call really enqueued (flush_pending rolls back the rest) */
    void* dDisplay = nullptr;  /* scratch of the display pass, kept between calls (grows on demand) */
    size_t displayBytes = 0;
    hipEvent_t evStart = nullptr, evStop = nullptr;
    double gpuMs = 0;
    int timerState = 0; /* 0 idle, 1 begun, 2 ended (elapsed not yet read) */
};

static int fail(RtContext* ctx, int status, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s", buf);
    snprintf(g_err, sizeof(g_err), "%s", buf);
    return status;
}

#define HIP_TRY(ctx, call)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

/* The main stream, ordered after everything the second render stream has been given.  Every
 * use of the stream other than a render launch goes through here. */
static hipStream_t joined(RtContext* ctx)
{
    if (ctx->ord.sideDirty) {
        hipEventRecord(ctx->ord.evJoin, ctx->sideStream);
        hipStreamWaitEvent(ctx->stream, ctx->ord.evJoin, 0);
        ctx->ord.sideDirty = false;
        ctx->ord.accWriterPending[1] = false; /* whatever the side stream adds to the accumulation buffer now precedes the main stream's next kernel */
        ctx->ord.accFullPending[1] = false;
    }
    ctx->ord.needFork = true;
    return ctx->stream;
}

static int local_rows_for(int H, int stripRows, int partIndex, int partCount)
{
    int rows = 0;
    int nStrips = (H + stripRows - 1) / stripRows;
    for (int s = partIndex; s < nStrips; s += partCount) {
        int r0 = s * stripRows;
        int r1 = r0 + stripRows < H ? r0 + stripRows : H;
        rows += r1 - r0;
    }
    return rows;
}

static void flush_timer(RtContext* ctx)
{
    if (ctx->timerState == 2) {
        hipEventSynchronize(ctx->evStop);
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx->evStart, ctx->evStop) == hipSuccess) ctx->gpuMs += ms;
        ctx->timerState = 0;
    }
}

/* Stream-ordered host->device upload: the bytes are copied into a pinned staging slot now and the
 * transfer is enqueued on the (joined) render stream, i.e. after every frame already enqueued and
 * before the next one; the caller's memory is free on return and the host waits for the GPU only when the
 * slot it is about to reuse is still in flight: a host that animates models AND spheres every frame uses 8 of the 24
 * slots per frame, i.e. it blocks once it is three frames ahead of the GPU. */
static int stage_upload(RtContext* ctx, void* dst, const void* src, size_t bytes)
{
    if (!bytes) return RT_OK;
    RtContext::Staging& s = ctx->staging[ctx->stagingNext];
    ctx->stagingNext = (ctx->stagingNext + 1) % (int)(sizeof(ctx->staging) / sizeof(ctx->staging[0]));
    if (s.inFlight) {
        HIP_TRY(ctx, hipEventSynchronize(s.done));
        s.inFlight = false;
    }
    if (s.cap < bytes) {
        if (s.host) hipHostFree(s.host);
        s.host = nullptr;
        s.cap = 0;
        size_t cap = bytes < 4096 ? 4096 : bytes + bytes / 2;
        HIP_TRY(ctx, hipHostMalloc(&s.host, cap, hipHostMallocDefault));
        s.cap = cap;
    }
    if (!s.done) HIP_TRY(ctx, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    memcpy(s.host, src, bytes);
    hipStream_t st = joined(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(dst, s.host, bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipEventRecord(s.done, st));
    s.inFlight = true;
    ctx->updateUploads++;
    return RT_OK;
}

static int launch_frames(RtContext* ctx, int frame0, int nFrames);
static bool gpu_idle(RtContext* ctx);

/* Launch the frames rt_render_frame held back.  Called first thing by every entry point that reads or changes
 * what those frames depend on, or that hands results to the host. */
static int flush_pending(RtContext* ctx)
{
    if (!ctx || ctx->pending == 0) return RT_OK;
    const int n = ctx->pending;
    ctx->pending = 0;
    hipSetDevice(ctx->device);
    ctx->lastLaunched = 0;
    const int rc = launch_frames(ctx, ctx->frame - n, n);
    if (rc != RT_OK) ctx->frame -= n - ctx->lastLaunched; /* the frames that never ran: the frame counter says so (the caller may retry them) */
    return rc;
}
#define RT_FLUSH(ctx)                         \
    do {                                      \
        int frc_ = flush_pending(ctx);        \
        if (frc_ != RT_OK) return frc_;       \
    } while (0)

extern "C" {

const char* rt_version(void) { return RT_VERSION_STRING; }

const char* rt_last_error(const RtContext* ctx) { return ctx ? ctx->err : g_err; }

int rt_create(int device_id, RtContext** out)
{
    if (!out) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, RT_ERR_NO_DEVICE, "rt_create: no HIP device (%s); libraytrace_hip has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: device %d out of range [0,%d)", device_id, n);
    RtContext* ctx = new RtContext();
    ctx->device = device_id;
    auto init = [&]() -> int {
        HIP_TRY(ctx, hipSetDevice(device_id));
        hipDeviceProp_t prop;
        HIP_TRY(ctx, hipGetDeviceProperties(&prop, device_id));
        ctx->numCUs = prop.multiProcessorCount;
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->ownStream, hipStreamNonBlocking));
        ctx->stream = ctx->ownStream;
        HIP_TRY(ctx, hipMalloc(&ctx->dCounters, sizeof(unsigned long long) * RT_COUNTER_SLOTS * RT_COUNTER_FIELDS));
        /* Every fill goes through the context's own stream: hipMemset on the null stream may return before the fill has
         * run, and the render streams are non-blocking, i.e. NOT ordered after the null stream — a late fill would then
         * wipe what the first kernels have already counted (seen with two processes sharing one GPU: 4 % of a launch's
         * segments lost once in 20 runs). */
        HIP_TRY(ctx, hipMemsetAsync(ctx->dCounters, 0, sizeof(unsigned long long) * RT_COUNTER_SLOTS * RT_COUNTER_FIELDS, ctx->ownStream));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->sideStream, hipStreamNonBlocking)); /* (a higher priority for it measured no different) */
        HIP_TRY(ctx, hipMalloc(&ctx->dTileQueue, 2 * sizeof(unsigned long long)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->dTileQueue, 0, 2 * sizeof(unsigned long long), ctx->ownStream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->ownStream));
        HIP_TRY(ctx, hipEventCreate(&ctx->evStart));
        HIP_TRY(ctx, hipEventCreate(&ctx->evStop));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ord.evFork, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ord.evJoin, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ord.evAccWriter[0], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ord.evAccWriter[1], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ord.evAccFull[0], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ord.evAccFull[1], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ord.evSort, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ord.evOrderRetire[0], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ord.evOrderRetire[1], hipEventDisableTiming));
        return RT_OK;
    };
    if (int rc = init()) { /* the message stays readable through rt_last_error(NULL) */
        rt_destroy(ctx);
        return rc;
    }
    memset(&ctx->params, 0, sizeof(ctx->params));
    if (const char* g = getenv("RT_GRID")) ctx->gridOverride = atoi(g); /* tuning hook */
    if (const char* g = getenv("RT_POOL_MIN_ITEMS")) ctx->poolMinItems = atoi(g);
    if (const char* g = getenv("RT_POOL_FAULT")) { if (atoi(g)) ctx->poolSpinLimit = 64u | 0x80000000u; }
    if (getenv("RT_VERBOSE")) ctx->verbose = true;
    if (const char* f = getenv("RT_FUSE_FRAMES")) ctx->fuseFrames = atoi(f) != 0;
    if (const char* l = getenv("RT_LPT")) ctx->lptEnabled = atoi(l) != 0;
    if (const char* l = getenv("RT_FUSED_BEHIND_FIRST_PART")) ctx->fusedBehindFirstPart = atoi(l) != 0;
    if (const char* t = getenv("RT_TWO_STREAMS")) ctx->twoStreams = atoi(t) != 0;
    if (const char* c = getenv("RT_COALESCE")) ctx->coalesce = atoi(c) != 0;
    if (const char* fg = getenv("RT_FRAME_GROUP")) ctx->frameGroupOverride = atoi(fg);
    if (const char* al = getenv("RT_ALTERNATE")) ctx->alternate = ctx->alternateWanted = atoi(al) != 0;
    if (const char* fc = getenv("RT_FUSE_CAP")) {
        const int v = atoi(fc);
        if (v >= 1) { ctx->fuseCap = v > RT_FUSE_MAX ? RT_FUSE_MAX : v; ctx->fuseCapPinned = true; }
    }
    *out = ctx;
    return RT_OK;
}

static void free_scene(RtContext* ctx)
{
    hipFree(ctx->dSpheres); ctx->dSpheres = nullptr;
    hipFree(ctx->dMaterials); ctx->dMaterials = nullptr;
    hipFree(ctx->dModels); ctx->dModels = nullptr;
    hipFree(ctx->dPairs); ctx->dPairs = nullptr;
    hipFree(ctx->dTris); ctx->dTris = nullptr;
    hipFree(ctx->dNorms); ctx->dNorms = nullptr;
    hipFree(ctx->dBigLeaves); ctx->dBigLeaves = nullptr;
    hipFree(ctx->dFilters); ctx->dFilters = nullptr;
    hipFree(ctx->dChunks); ctx->dChunks = nullptr;
    ctx->nChunks = 0;
    ctx->haveScene = false;
}

void rt_destroy(RtContext* ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    flush_pending(ctx);
    if (ctx->sideStream) hipStreamSynchronize(ctx->sideStream);
    hipStreamSynchronize(ctx->stream);
    free_scene(ctx);
    hipFree(ctx->ownFrame);
    hipFree(ctx->ownAccum);
    hipFree(ctx->dCounters);
    hipFree(ctx->dTileQueue);
    hipFree(ctx->dTileCost);
    hipFree(ctx->dTileOrder[0]);
    hipFree(ctx->dTileOrder[1]);
    hipFree(ctx->dTileKey);
    if (ctx->ord.evSort) hipEventDestroy(ctx->ord.evSort);
    for (int i = 0; i < 2; i++) if (ctx->ord.evOrderRetire[i]) hipEventDestroy(ctx->ord.evOrderRetire[i]);
    hipFree(ctx->dDisplay);
    hipFree(ctx->dStaging[0]);
    hipFree(ctx->dStaging[1]);
    for (int i = 0; i < 2; i++) if (ctx->ord.evAccWriter[i]) hipEventDestroy(ctx->ord.evAccWriter[i]);
    for (int i = 0; i < 2; i++) if (ctx->ord.evAccFull[i]) hipEventDestroy(ctx->ord.evAccFull[i]);
    hipFree(ctx->dPxCold);
    for (auto& pr : ctx->tuner.probe) {
        if (pr.start) hipEventDestroy(pr.start);
        if (pr.stop) hipEventDestroy(pr.stop);
    }
    for (RtContext::Staging& st : ctx->staging) {
        if (st.host) hipHostFree(st.host);
        if (st.done) hipEventDestroy(st.done);
    }
    for (auto& fp : ctx->fuseProbe) { if (fp.start) hipEventDestroy(fp.start); if (fp.stop) hipEventDestroy(fp.stop); }
    if (ctx->evStart) hipEventDestroy(ctx->evStart);
    if (ctx->evStop) hipEventDestroy(ctx->evStop);
    if (ctx->ord.evFork) hipEventDestroy(ctx->ord.evFork);
    if (ctx->ord.evJoin) hipEventDestroy(ctx->ord.evJoin);
    if (ctx->sideStream) hipStreamDestroy(ctx->sideStream);
    if (ctx->ownStream) hipStreamDestroy(ctx->ownStream);
    delete ctx;
}

int rt_set_stream(RtContext* ctx, void* hip_stream)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    hipStreamSynchronize(joined(ctx));
    flush_timer(ctx);
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->ownStream;
    return RT_OK;
}

int rt_set_partition(RtContext* ctx, int strip_rows, int part_index, int part_count)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    if (strip_rows <= 0 || strip_rows % 8 || part_count <= 0 || part_index < 0 || part_index >= part_count)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_set_partition: strip_rows must be a positive multiple of 8, 0 <= index < count");
    ctx->stripRows = strip_rows;
    ctx->partIndex = part_index;
    ctx->partCount = part_count;
    if (ctx->W > 0) return rt_resize(ctx, ctx->W, ctx->H);
    return RT_OK;
}

int rt_resize(RtContext* ctx, int width, int height)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    if (width <= 0 || height <= 0) return fail(ctx, RT_ERR_INVALID_ARG, "rt_resize: %dx%d", width, height);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
    ctx->W = width;
    ctx->H = height;
    ctx->localRows = local_rows_for(height, ctx->stripRows, ctx->partIndex, ctx->partCount);
    size_t bytes = (size_t)ctx->localRows * width * 16;
    if (bytes != ctx->ownBytes) {
        hipFree(ctx->ownFrame); ctx->ownFrame = nullptr;
        hipFree(ctx->ownAccum); ctx->ownAccum = nullptr;
        ctx->ownBytes = 0;
        if (bytes) {
            HIP_TRY(ctx, hipMalloc(&ctx->ownFrame, bytes));
            HIP_TRY(ctx, hipMalloc(&ctx->ownAccum, bytes));
            ctx->ownBytes = bytes;
        }
    }
    if (bytes) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->ownFrame, 0, bytes, joined(ctx)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->ownAccum, 0, bytes, joined(ctx)));
    }
    ctx->boundFrame = ctx->boundAccum = nullptr;
    ctx->orderTiles = 0; /* tile costs belong to the old geometry */
    for (int i = 0; i < 2; i++)
        if (ctx->dStaging[i]) {
            hipFree(ctx->dStaging[i]); /* sized for the old image: re-made by the next fused launch */
            ctx->dStaging[i] = nullptr;
            ctx->stagingBytes[i] = 0;
        }
    ctx->stagingUnavailable = false;
    ctx->alternate = ctx->alternateWanted; /* (a slab that did not fit the old image may fit this one: ADVICE r4) */
    if (!ctx->fuseCapPinned) ctx->fuseCap = RT_FUSE_MIN; /* frame times of the old geometry say nothing */
    for (auto& fp : ctx->fuseProbe) fp.live = false;
    ctx->fuseLast = -1;
    return RT_OK;
}

int rt_local_rows(const RtContext* ctx) { return ctx ? ctx->localRows : RT_ERR_INVALID_ARG; }

int rt_local_to_global_row(const RtContext* ctx, int local_row)
{
    if (!ctx || local_row < 0 || local_row >= ctx->localRows) return RT_ERR_INVALID_ARG;
    int ls = local_row / ctx->stripRows;
    return (ls * ctx->partCount + ctx->partIndex) * ctx->stripRows + (local_row - ls * ctx->stripRows);
}

int rt_bind_render_targets(RtContext* ctx, void* d_frame, void* d_accum)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    if (((uintptr_t)d_frame | (uintptr_t)d_accum) & 15) return fail(ctx, RT_ERR_INVALID_ARG, "render targets must be 16-byte aligned");
    HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
    ctx->boundFrame = (float*)d_frame;
    ctx->boundAccum = (float*)d_accum;
    return RT_OK;
}

int rt_get_render_targets(RtContext* ctx, void** d_frame, void** d_accum)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    /* a host that reads the targets after its own device synchronise must find every requested frame at least launched */
    RT_FLUSH(ctx);
    if (d_frame) *d_frame = ctx->boundFrame ? ctx->boundFrame : ctx->ownFrame;
    if (d_accum) *d_accum = ctx->boundAccum ? ctx->boundAccum : ctx->ownAccum;
    return RT_OK;
}

} /* extern "C" */

/* ---------------------------------------------------------------- scene */
/* Device sphere records.  First the exact one the reference's arithmetic reads — centre and
 * radius*radius (RC:299, same fp32 multiply) — then, for all spheres again, the record of the
 * conservative discriminant pre-test in begin_intersect: centre and |c|^2 - r*r (rounded from
 * double).  *bound = max_k(|c_k|^2 + r_k^2), rounded up: it scales the pre-test's error margin. */
static void pack_spheres(const RtSphere* spheres, int n, std::vector<float>& out, float* bound)
{
    /* n exact records (c, r*r), then ceil(n/2) PAIR records of the conservative pre-test: (cx0, cx1, cy0, cy1, cz0, cz1, K0, K1)
     * with K = |c|^2 - r*r — two spheres side by side, so that one scalar load fills the SGPR pairs a packed fp32
     * instruction takes (begin_intersect); an odd last sphere is paired with itself */
    const size_t pairs = ((size_t)n + 1) / 2;
    out.assign((size_t)n * 4 + pairs * 8, 0.0f);
    double maxM = 0;
    for (int i = 0; i < n; i++) {
        const float* c = spheres[i].centre;
        const float r2 = spheres[i].radius * spheres[i].radius;
        memcpy(&out[4 * (size_t)i], c, 12);
        out[4 * (size_t)i + 3] = r2;
        const double cc = (double)c[0] * c[0] + (double)c[1] * c[1] + (double)c[2] * c[2];
        const float K = (float)(cc - (double)r2);
        float* q = &out[4 * (size_t)n + 8 * (size_t)(i / 2)];
        const int h = i & 1;
        q[0 + h] = c[0]; q[2 + h] = c[1]; q[4 + h] = c[2]; q[6 + h] = K;
        if (!h && i == n - 1) { q[1] = c[0]; q[3] = c[1]; q[5] = c[2]; q[7] = K; }
        if (cc + (double)r2 > maxM) maxM = cc + (double)r2;
    }
    *bound = (float)(maxM * 1.000001);
}

static void pack_material(const RtMaterial& m, DMaterial& d)
{
    memset(&d, 0, sizeof(d));
    memcpy(d.diffuseCol, m.diffuseCol, 16);
    memcpy(d.emissionCol, m.emissionCol, 16);
    memcpy(d.specularCol, m.specularCol, 16);
    memcpy(d.absorption, m.absorption, 16);
    d.absorptionStrength = m.absorptionStrength;
    d.emissionStrength = m.emissionStrength;
    d.smoothness = m.smoothness;
    d.specularProbability = m.specularProbability;
    d.ior = m.ior;
    d.flag = m.flag;
}
static void pack_model(const RtModel& m, uint32_t rootCode, int32_t triBaseUnits, DModel& d)
{
    memset(&d, 0, sizeof(d));
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) {
            d.w2l[r * 4 + c] = m.worldToLocal[c * 4 + r];
            d.l2w[r * 4 + c] = m.localToWorld[c * 4 + r];
        }
    d.rootCode = rootCode;
    d.triBase = triBaseUnits;
    d.cullBackface = m.material.flag != RT_MATERIAL_GLASS; /* RC:355 */
}

/* World-space, inflated boxes of a model's two root children — the conservative root filter
 * of begin_intersect.  Corners go through inverse(worldToLocal) in double precision; the
 * inflation (1e-4 of the scene extent plus 1e-5 of the model's own coordinate range, mapped to
 * world units) is two to three orders of magnitude above the fp32 rounding of the reference's
 * local-space slab test.  A matrix that is not affine-invertible in a well-conditioned way
 * disables the filter for that model. */
static bool invert_affine(const float* m /* column-major 4x4 */, double inv[12] /* 3 rows x 4 */)
{
    double a[3][3], t[3];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) a[r][c] = m[c * 4 + r];
        t[r] = m[12 + r];
    }
    if (m[3] != 0.0f || m[7] != 0.0f || m[11] != 0.0f || m[15] != 1.0f) return false;
    double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                 a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    double scale = 0;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) scale = fmax(scale, fabs(a[r][c]));
    if (!(fabs(det) > 1e-9 * scale * scale * scale) || !std::isfinite(det)) return false;
    double id = 1.0 / det;
    double b[3][3];
    b[0][0] = (a[1][1] * a[2][2] - a[1][2] * a[2][1]) * id; b[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * id; b[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * id;
    b[1][0] = (a[1][2] * a[2][0] - a[1][0] * a[2][2]) * id; b[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * id; b[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) * id;
    b[2][0] = (a[1][0] * a[2][1] - a[1][1] * a[2][0]) * id; b[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) * id; b[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) * id;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) inv[r * 4 + c] = b[r][c];
        inv[r * 4 + 3] = -(b[r][0] * t[0] + b[r][1] * t[1] + b[r][2] * t[2]);
        for (int c = 0; c < 4; c++)
            if (!std::isfinite(inv[r * 4 + c])) return false;
    }
    return true;
}

/* returns false if the model cannot be filtered; otherwise world boxes (not yet inflated) in wmin/wmax[2][3] */
static bool world_boxes(const RtModel& m, const RtBVHNode children[2], double wmin[2][3], double wmax[2][3], double* localRange)
{
    double inv[12];
    if (!invert_affine(m.worldToLocal, inv)) return false;
    double range = 0, normS = 0;
    for (int r = 0; r < 3; r++) normS = fmax(normS, fabs(inv[r * 4]) + fabs(inv[r * 4 + 1]) + fabs(inv[r * 4 + 2]));
    for (int k = 0; k < 2; k++) {
        for (int d = 0; d < 3; d++) {
            if (!std::isfinite(children[k].boundsMin[d]) || !std::isfinite(children[k].boundsMax[d])) return false;
            wmin[k][d] = INFINITY;
            wmax[k][d] = -INFINITY;
            range = fmax(range, fmax(fabs((double)children[k].boundsMin[d]), fabs((double)children[k].boundsMax[d])));
        }
        for (int corner = 0; corner < 8; corner++) {
            double p[3];
            for (int d = 0; d < 3; d++) p[d] = (corner >> d & 1) ? children[k].boundsMax[d] : children[k].boundsMin[d];
            for (int r = 0; r < 3; r++) {
                double w = inv[r * 4] * p[0] + inv[r * 4 + 1] * p[1] + inv[r * 4 + 2] * p[2] + inv[r * 4 + 3];
                wmin[k][r] = fmin(wmin[k][r], w);
                wmax[k][r] = fmax(wmax[k][r], w);
            }
        }
    }
    *localRange = range * normS;
    return true;
}

static void make_filters(const RtModel* models, int n_models, const std::vector<uint32_t>& rootCodes, const std::vector<RtBVHNode>& rootChildren,
                         const RtSphere* spheres, int n_spheres, std::vector<DFilter>& out, float* maxOrigin)
{
    out.assign(n_models, DFilter());
    std::vector<double> lr(n_models, 0.0);
    std::vector<char> ok(n_models, 0);
    std::vector<double> bmin((size_t)n_models * 6), bmax((size_t)n_models * 6);
    double extent = 0;
    for (int i = 0; i < n_models; i++) {
        DFilter& f = out[i];
        memset(&f, 0, sizeof(f));
        /* innerRoot: bit 0 = the root is an inner node; a leaf root carries its triangle count in bits 8.. (exact
         * counters of rejected models).  rootChildren holds the root's two child boxes, or — leaf root — the
         * bounds of the leaf's triangles twice (computed from the triangles on upload, never taken from the
         * root node, whose bounds the reference does not read) */
        const bool leafRoot = (rootCodes[i] & RT_CODE_LEAF) != 0;
        f.innerRoot = leafRoot ? ((uint32_t)rootChildren[2 * (size_t)i].triangleCount << 8) : 1u;
        f.always = 1;
        if (leafRoot && rootChildren[2 * (size_t)i].triangleCount <= 0) continue; /* no box available */
        double wmin[2][3], wmax[2][3];
        if (!world_boxes(models[i], &rootChildren[2 * (size_t)i], wmin, wmax, &lr[i])) continue;
        ok[i] = 1;
        for (int k = 0; k < 2; k++)
            for (int d = 0; d < 3; d++) {
                bmin[(size_t)i * 6 + k * 3 + d] = wmin[k][d];
                bmax[(size_t)i * 6 + k * 3 + d] = wmax[k][d];
                extent = fmax(extent, fmax(fabs(wmin[k][d]), fabs(wmax[k][d])));
            }
    }
    for (int i = 0; i < n_spheres; i++)
        for (int d = 0; d < 3; d++) extent = fmax(extent, fabs((double)spheres[i].centre[d]) + fabs((double)spheres[i].radius));
    if (!std::isfinite(extent)) extent = 0;
    for (int i = 0; i < n_models; i++) {
        if (!ok[i]) continue;
        DFilter& f = out[i];
        const double margin = 1e-4 * extent + 1e-5 * lr[i] + 1e-30;
        bool fin = true;
        for (int d = 0; d < 3; d++) { /* one box: the union of the two children (measured cheaper than testing both) */
            const double lo2 = fmin(bmin[(size_t)i * 6 + d], bmin[(size_t)i * 6 + 3 + d]);
            const double hi2 = fmax(bmax[(size_t)i * 6 + d], bmax[(size_t)i * 6 + 3 + d]);
            float lo = nextafterf((float)(lo2 - margin), -INFINITY);
            float hi = nextafterf((float)(hi2 + margin), INFINITY);
            f.bMin[d] = lo;
            f.bMax[d] = hi;
            fin = fin && std::isfinite(lo) && std::isfinite(hi);
        }
        f.always = fin ? 0u : 1u;
    }
    /* rays starting farther than this from the origin have coarser fp32 spacing than the margin allows for */
    *maxOrigin = (float)(8.0 * extent);
}

/* The device array behind KArgs::filters / filterPairs: the n DFilter records, then ceil(n / 2) pair records (two DFilter
 * slots each) with the same boxes side by side for the packed root filter of rt_kernels.h. */
static std::vector<DFilter> append_filter_pairs(const std::vector<DFilter>& f)
{
    const size_t n = f.size(), np = (n + 1) / 2;
    std::vector<DFilter> out(n + 2 * np);
    memset(out.data(), 0, out.size() * sizeof(DFilter));
    for (size_t i = 0; i < n; i++) out[i] = f[i];
    static_assert(sizeof(DFilter) == 32, "a pair record is two DFilter slots = sixteen dwords");
    for (size_t p = 0; p < np; p++) {
        float* q = reinterpret_cast<float*>(&out[n + 2 * p]);
        for (int h = 0; h < 2; h++) {
            const size_t m = 2 * p + h;
            uint32_t always = 1u; /* a missing second model never reaches the mask (the kernel checks m + 1 < n) */
            if (m < n) {
                for (int d = 0; d < 3; d++) {
                    q[2 * d + h] = f[m].bMin[d];
                    q[6 + 2 * d + h] = f[m].bMax[d];
                }
                always = f[m].always;
            }
            memcpy(&q[12 + h], &always, 4);
        }
    }
    return out;
}

/* Chunks of the two-level model hierarchy (rt_device.h, DChunk): only built for more than 64 models.
 * Models the filter cannot reject (`always`) are kept in chunks of their own so that they do not spoil the
 * boxes of the others; the rest is clustered by the Morton code of the filter box centre. */
#define RT_MAX_FILTER_EXT_WORDS 32
static void make_chunks(const std::vector<DFilter>& filters, std::vector<DChunk>& chunks, int* nFiltered, int* extWords)
{
    const int n = (int)filters.size();
    chunks.clear();
    *nFiltered = n < 64 ? n : 64;
    *extWords = 0;
    if (n <= 64) return;
    const int maxModels = 63 + 32 * RT_MAX_FILTER_EXT_WORDS;
    const int nf = n < maxModels ? n : maxModels;
    *nFiltered = nf;
    *extWords = (nf - 63 + 31) / 32;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < nf; i++)
        if (!filters[i].always)
            for (int d = 0; d < 3; d++) {
                lo[d] = fmin(lo[d], (double)filters[i].bMin[d]);
                hi[d] = fmax(hi[d], (double)filters[i].bMax[d]);
            }
    auto spread = [](uint32_t v) { /* 10 bits -> every third bit */
        v &= 1023u;
        v = (v | (v << 16)) & 0x030000ffu;
        v = (v | (v << 8)) & 0x0300f00fu;
        v = (v | (v << 4)) & 0x030c30c3u;
        v = (v | (v << 2)) & 0x09249249u;
        return v;
    };
    std::vector<std::pair<uint64_t, int>> order;
    for (int i = 0; i < nf; i++) {
        uint64_t key;
        if (filters[i].always) {
            key = (uint64_t)i; /* first, in index order */
        } else {
            uint32_t q[3];
            for (int d = 0; d < 3; d++) {
                const double c = 0.5 * ((double)filters[i].bMin[d] + (double)filters[i].bMax[d]);
                const double t = hi[d] > lo[d] ? (c - lo[d]) / (hi[d] - lo[d]) : 0.0;
                q[d] = (uint32_t)(t < 0 ? 0 : t > 1 ? 1023 : t * 1023.0);
            }
            key = (1ull << 40) | ((uint64_t)(spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2)) << 8);
        }
        order.push_back({key, i});
    }
    std::stable_sort(order.begin(), order.end(), [](const std::pair<uint64_t, int>& a, const std::pair<uint64_t, int>& b) { return a.first < b.first; });
    for (size_t p = 0; p < order.size();) {
        DChunk c;
        memset(&c, 0, sizeof(c));
        const bool alw = filters[order[p].second].always != 0;
        c.always = alw ? 1u : 0u;
        for (int d = 0; d < 3; d++) { c.bMin[d] = INFINITY; c.bMax[d] = -INFINITY; }
        while (p < order.size() && c.count < RT_CHUNK_MODELS && (filters[order[p].second].always != 0) == alw) {
            const DFilter& f = filters[order[p].second];
            c.members[c.count++] = (uint32_t)order[p].second;
            c.innerRoots += f.innerRoot & 1u;
            if (!alw)
                for (int d = 0; d < 3; d++) {
                    c.bMin[d] = fminf(c.bMin[d], f.bMin[d]);
                    c.bMax[d] = fmaxf(c.bMax[d], f.bMax[d]);
                }
            p++;
        }
        std::sort(c.members, c.members + c.count);
        chunks.push_back(c);
    }
}

struct SceneBuilder {
    const RtBVHNode* nodes;
    int nNodes, nTris;
    std::vector<DPair> pairs;
    std::vector<uint32_t> bigLeaves;
    std::vector<int32_t> pairOfFirstChild; /* absolute first-child node index -> pair id, -1 unseen, -2 in progress */
    std::vector<int32_t> pairDepth;        /* height of the subtree below pair (levels) */
    std::vector<int32_t> pairNodeOffset;   /* the nodeOffset the pair's inner children were resolved with (RC:265-266: child = nodeOffset + startIndex) */
    std::vector<long long> pairLeafEnd;    /* largest startIndex + triangleCount of the leaves below pair (mesh-relative) */
    std::string error;
    /* parallel conversion (one builder per mesh): pairs go into a segment of the scene's array, with their final ids; the memo
     * covers only the mesh's own node window.  A mesh that leaves its window or overflows its segment sets `outside` and the
     * caller falls back to the sequential walk, which has neither limit. */
    DPair* seg = nullptr;
    size_t segCap = 0, segCount = 0;
    uint32_t idBase = 0;
    int memoLo = 0;
    bool outside = false;

    /* code of a leaf node whose triangles are [start, start+count) relative to triOffset */
    bool leaf_code(const RtBVHNode& n, int triOffset, uint32_t* code)
    {
        long long lo = (long long)triOffset + n.startIndex, hi = lo + n.triangleCount;
        if (n.startIndex < 0 || lo < 0 || hi > nTris) {
            error = "leaf triangle range out of bounds";
            return false;
        }
        if (n.triangleCount <= RT_CODE_MAX_INLINE_COUNT && (uint32_t)n.startIndex <= RT_CODE_MAX_INLINE_START) {
            *code = RT_CODE_LEAF | ((uint32_t)n.triangleCount << 24) | (uint32_t)n.startIndex;
        } else {
            uint32_t idx = (uint32_t)(bigLeaves.size() / 2);
            if (idx > RT_CODE_MAX_INLINE_START) { error = "too many oversized leaves"; return false; }
            bigLeaves.push_back((uint32_t)n.startIndex);
            bigLeaves.push_back((uint32_t)n.triangleCount);
            *code = RT_CODE_LEAF | idx;
        }
        return true;
    }

    /* Converts the subtree under node `abs` (absolute index) of a mesh whose node 0 is at nodeOffset.
     * Returns its code and height (leaf = 0). Iterative post-order walk, memoised per sibling pair. */
    bool convert(int nodeOffset, int triOffset, int absRoot, uint32_t* codeOut, int* heightOut, long long* endOut = nullptr)
    {
        struct Frame { int abs; int stage; int firstChild; uint32_t codeA, codeB; int hA, hB; long long endA; };
        std::vector<Frame> stack;
        stack.push_back({absRoot, 0, -1, 0, 0, 0, 0, 0});
        uint32_t retCode = 0;
        int retHeight = 0;
        long long retEnd = 0; /* largest leaf end (mesh-relative) of the subtree just returned */
        while (!stack.empty()) {
            Frame& f = stack.back();
            const RtBVHNode& n = nodes[f.abs];
            if (f.stage == 0) {
                if (n.triangleCount > 0) { /* leaf — RC:246 */
                    if (!leaf_code(n, triOffset, &retCode)) return false;
                    retHeight = 0;
                    retEnd = (long long)n.startIndex + n.triangleCount;
                    stack.pop_back();
                    continue;
                }
                long long fc = (long long)nodeOffset + n.startIndex;
                if (n.startIndex < 0 || fc < 0 || fc + 1 >= nNodes) { error = "inner node child index out of bounds"; return false; }
                f.firstChild = (int)fc;
                if (f.firstChild < memoLo || (size_t)(f.firstChild - memoLo) >= pairOfFirstChild.size()) { outside = true; error = "node outside the mesh's window"; return false; }
                int known = pairOfFirstChild[f.firstChild - memoLo];
                if (known == -2) { error = "cycle in BVH node graph"; return false; }
                if (known >= 0) {
                    /* a pair with inner children means what it means under ONE nodeOffset (RC:265-266 adds the model's nodeOffset to a child
                     * index): a mesh whose tree wanders into another mesh's nodes would need a second, different conversion of the same nodes —
                     * refused like a cycle (the reference would traverse it; no builder produces it) */
                    if (pairDepth[known - idBase] > 1 && pairNodeOffset[known - idBase] != nodeOffset) { error = "node pair reached under two different nodeOffsets"; return false; }
                    /* converted for an earlier model that shares these nodes: its leaves were range-checked
                     * against THAT model's triOffset, so check this one's against the subtree's largest leaf end */
                    if ((long long)triOffset + pairLeafEnd[known - idBase] > nTris) { error = "leaf triangle range out of bounds"; return false; }
                    retCode = (uint32_t)known;
                    retHeight = pairDepth[known - idBase];
                    retEnd = pairLeafEnd[known - idBase];
                    stack.pop_back();
                    continue;
                }
                if ((int)stack.size() > RT_MAX_BVH_DEPTH + 1) { error = "BVH deeper than RT_MAX_BVH_DEPTH"; return false; }
                pairOfFirstChild[f.firstChild - memoLo] = -2;
                f.stage = 1;
                int child = f.firstChild;
                stack.push_back({child, 0, -1, 0, 0, 0, 0, 0});
                continue;
            }
            if (f.stage == 1) {
                f.codeA = retCode;
                f.hA = retHeight;
                f.endA = retEnd;
                f.stage = 2;
                int child = f.firstChild + 1;
                stack.push_back({child, 0, -1, 0, 0, 0, 0, 0});
                continue;
            }
            /* stage 2: both children done */
            f.codeB = retCode;
            f.hB = retHeight;
            const RtBVHNode& A = nodes[f.firstChild];
            const RtBVHNode& B = nodes[f.firstChild + 1];
            DPair p;
            memset(&p, 0, sizeof(p));
            memcpy(p.aMin, A.boundsMin, 12); memcpy(p.aMax, A.boundsMax, 12);
            memcpy(p.bMin, B.boundsMin, 12); memcpy(p.bMax, B.boundsMax, 12);
            p.codeA = f.codeA;
            p.codeB = f.codeB;
            int id;
            if (seg) {
                if (segCount == segCap) { outside = true; error = "more node pairs than the mesh's window holds"; return false; }
                seg[segCount] = p;
                id = (int)(idBase + segCount++);
            } else {
                id = (int)pairs.size();
                pairs.push_back(p);
            }
            int h = 1 + (f.hA > f.hB ? f.hA : f.hB);
            pairDepth.push_back(h);
            pairNodeOffset.push_back(nodeOffset);
            pairLeafEnd.push_back(f.endA > retEnd ? f.endA : retEnd);
            pairOfFirstChild[f.firstChild - memoLo] = id;
            retCode = (uint32_t)id;
            retHeight = h;
            retEnd = pairLeafEnd.back();
            stack.pop_back();
        }
        *codeOut = retCode;
        *heightOut = retHeight;
        if (endOut) *endOut = retEnd;
        return true;
    }
};

/* host worker threads for the scene preparation: f(k) for k in [0, n), at most RT_HOST_THREADS (default 16) at a time */
template <typename F>
static void parallel_jobs(int n, F f)
{
    int nThreads = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("RT_HOST_THREADS")) nThreads = atoi(e);
    if (nThreads > 16) nThreads = 16;
    if (nThreads > n) nThreads = n;
    if (nThreads <= 1) {
        for (int k = 0; k < n; k++) f(k);
        return;
    }
    std::atomic<int> next(0);
    std::vector<std::thread> pool;
    for (int t = 0; t < nThreads; t++)
        pool.emplace_back([&] { for (int k = next.fetch_add(1); k < n; k = next.fetch_add(1)) f(k); });
    for (auto& th : pool) th.join();
}

template <typename T>
static int upload_vec(RtContext* ctx, T** dptr, const void* src, size_t count)
{
    size_t bytes = count * sizeof(T);
    HIP_TRY(ctx, hipMalloc(dptr, bytes ? bytes : sizeof(T)));
    if (bytes) HIP_TRY(ctx, hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice));
    return RT_OK;
}

/* the filter boxes moved: the chunk boxes (and the spatial clustering) follow, stream-ordered */
static int refresh_chunks(RtContext* ctx, const std::vector<DFilter>& filters)
{
    if (!ctx->nChunks) return RT_OK;
    std::vector<DChunk> chunks;
    int nf = 0, ew = 0;
    make_chunks(filters, chunks, &nf, &ew);
    if ((int)chunks.size() != ctx->nChunks) { /* a model became (un)filterable: the split into chunks changed size — rare, synchronous */
        HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
        hipFree(ctx->dChunks);
        ctx->dChunks = nullptr;
        int rc = upload_vec(ctx, &ctx->dChunks, chunks.data(), chunks.size());
        if (rc) return rc;
        ctx->nChunks = (int)chunks.size();
        return RT_OK;
    }
    return stage_upload(ctx, ctx->dChunks, chunks.data(), sizeof(DChunk) * chunks.size());
}

/* The host side of rt_upload_scene: everything validated and re-laid out ONCE, ready to be uploaded to any number of
 * contexts (rt_multi_upload_scene prepares once for all its devices). */
struct PreparedScene {
    std::vector<float> sph;
    float sphereBound = 0;
    std::vector<DMaterial> mats;
    std::vector<DModel> dmodels;
    PodVec<DPair> pairs; /* canonical form (SceneBuilder::convert); consumed by the layout */
    LaidOutScene lay;    /* what is uploaded: pair / triangle / normal spaces, final codes */
    size_t nPairs = 0;
    std::vector<DFilter> filters;
    std::vector<DChunk> chunks;
    int nFiltered = 0, extWords = 0;
    float maxOrigin = 0;
    std::vector<RtBVHNode> rootChildren;
    std::vector<uint32_t> rootCodes;
    std::vector<RtModel> hModels;
    std::vector<RtSphere> hSpheres;
    int nTris = 0, maxHeight = 1;
    bool flat = true;
    int wavesPerGroup = 1; /* plan_groups: what the layout's cache prefix was sized for */
};

/* ---- the BVH trace kernels' workgroups (round 6).  A CU keeps 4 x RT_MIN_WAVES_PER_SIMD waves of them (VGPR bound) if the LDS allows:
 * a wave needs (stack + pixel fields + mask extension) x 256 B.  What the CU's 160 KB leave over becomes the top-of-tree cache — one copy
 * per WORKGROUP, so the fewer, larger workgroups the more records it holds: 12 waves (2 groups per CU) unless RT_WAVES_PER_GROUP says
 * otherwise.  Pure arithmetic on the scene's tree height and model count (no device query): every context, the multi-device upload and
 * rt_debug_layout agree on it.  choose_variant confirms the occupancy with the runtime's own query. */
#define RT_LDS_BYTES_PER_CU (160 * 1024)
struct GroupPlan { int wavesPerGroup = 1; int cacheRecords = 0; };
static size_t wave_lds_bytes(int stackEntries, int extWords)
{
    return (size_t)(stackEntries + RT_PIXEL_FIELDS + (extWords ? 2 + extWords : 0)) * RT_WAVE * sizeof(uint32_t);
}
static GroupPlan plan_groups(int maxHeight, int nModels)
{
    GroupPlan g;
    const int nf = nModels <= 64 ? 0 : (nModels < 63 + 32 * 32 ? nModels : 63 + 32 * 32); /* make_chunks: extWords */
    const int extWords = nf ? (nf - 63 + 31) / 32 : 0;
    const size_t waveBytes = wave_lds_bytes(maxHeight, extWords);
    int wavesPerCU = 4 * RT_MIN_WAVES_PER_SIMD;
    if ((size_t)wavesPerCU * waveBytes > RT_LDS_BYTES_PER_CU) wavesPerCU = (int)(RT_LDS_BYTES_PER_CU / waveBytes);
    if (wavesPerCU < 1) return g;
    int want = RT_MAX_WAVES_PER_GROUP;
    if (const char* e = getenv("RT_WAVES_PER_GROUP")) want = atoi(e);
    if (want < 1) want = 1;
    if (want > RT_MAX_WAVES_PER_GROUP) want = RT_MAX_WAVES_PER_GROUP;
    while (want > 1 && wavesPerCU % want) want--; /* whole groups fill the CU's wave slots */
    const int groupsPerCU = wavesPerCU / want;
    /* LDS is handed out in blocks: keep 1 KB per group clear of the nominal share */
    const long long share = (long long)RT_LDS_BYTES_PER_CU / groupsPerCU - 1024 - (long long)want * (long long)waveBytes;
    long long records = share > 0 ? share / (long long)sizeof(DPair) : 0;
    if (const char* e = getenv("RT_HOT_KB")) { const long long cap = atoll(e) * 1024 / (long long)sizeof(DPair); if (cap < records) records = cap < 0 ? 0 : cap; }
    if (records > (1 << 16)) records = 1 << 16;
    if (records < 16) return g; /* not worth a workgroup: single-wave groups, no cache */
    g.wavesPerGroup = want;
    g.cacheRecords = (int)records;
    return g;
}

/* errors are reported on `ctx` (may be any context of the caller) */
static int prepare_scene(RtContext* ctx, const RtModel* models, int n_models, const RtTriangle* triangles, int n_triangles,
                         const RtBVHNode* nodes, int n_nodes, const RtSphere* spheres, int n_spheres, PreparedScene& ps, const char* layoutOverride = nullptr)
{
    if (n_models < 0 || n_triangles < 0 || n_nodes < 0 || n_spheres < 0 || (n_models && !models) || (n_triangles && !triangles) ||
        (n_nodes && !nodes) || (n_spheres && !spheres))
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_upload_scene: bad pointer/count");

    /* ---- validate + re-lay out the BVHs reachable from the models */
    SceneBuilder sb;
    sb.nodes = nodes;
    sb.nNodes = n_nodes;
    sb.nTris = n_triangles;
    std::vector<uint32_t>& rootCodes = ps.rootCodes;
    std::vector<RtBVHNode>& rootChildren = ps.rootChildren;
    rootCodes.assign(n_models, 0u);
    rootChildren.assign((size_t)n_models * 2, RtBVHNode());
    std::vector<int> heights(n_models, 0);
    for (int i = 0; i < n_models; i++) {
        const RtModel& m = models[i];
        if (m.nodeOffset < 0 || m.nodeOffset >= n_nodes || m.triOffset < 0 || m.triOffset > n_triangles)
            return fail(ctx, RT_ERR_SCENE, "model %d: nodeOffset/triOffset out of range", i);
        if (nodes[m.nodeOffset].triangleCount == 0)
            return fail(ctx, RT_ERR_SCENE, "model %d: root node has triangleCount 0 (empty mesh) — undefined in the reference (RC:246)", i);
    }
    /* the distinct meshes (by root node), in the order the models name them */
    struct MeshJob { int nodeOffset, triOffset, firstModel; size_t segStart = 0; SceneBuilder sb; uint32_t code = 0; int height = 0; long long leafEnd = 0; bool ok = false; };
    std::vector<MeshJob> jobs;
    std::vector<int> jobOfModel(n_models, 0);
    {
        std::unordered_map<int, int> jobOfRoot; /* (a scene of 10^5 models with a mesh each must not pay 10^10 comparisons here) */
        for (int i = 0; i < n_models && jobs.size() <= 256; i++) { /* more than 256 meshes: the sequential walk below, no jobs needed */
            auto it = jobOfRoot.find(models[i].nodeOffset);
            if (it == jobOfRoot.end()) {
                it = jobOfRoot.emplace(models[i].nodeOffset, (int)jobs.size()).first;
                jobs.emplace_back();
                jobs.back().nodeOffset = models[i].nodeOffset; jobs.back().triOffset = models[i].triOffset; jobs.back().firstModel = i;
            }
            jobOfModel[i] = it->second;
        }
    }
    bool merged = false;
    size_t nPairs = 0;
    if (jobs.size() >= 2 && jobs.size() <= 256 && n_nodes >= (1 << 16) && !getenv("RT_SEQUENTIAL_PREPARE")) {
        /* large scene with several meshes: one worker per mesh.  A mesh's nodes are expected in the window from its root to the next
         * mesh's root (how CreateAllMeshData lays them out, RCM:206-236); a window of w nodes holds at most w / 2 pairs, so every mesh
         * gets its segment of ONE uninitialised pair array up front and writes final ids — nothing is merged or rebased. */
        std::vector<int> order(jobs.size());
        for (size_t j = 0; j < jobs.size(); j++) order[j] = (int)j;
        std::sort(order.begin(), order.end(), [&](int x, int y) { return jobs[x].nodeOffset < jobs[y].nodeOffset; });
        std::vector<size_t> segStart(jobs.size() + 1, 0);
        std::vector<int> winEnd(jobs.size(), n_nodes);
        for (size_t k = 0; k < order.size(); k++) {
            const int j = order[k];
            winEnd[j] = k + 1 < order.size() ? jobs[order[k + 1]].nodeOffset : n_nodes;
            jobs[j].segStart = segStart[k];
            segStart[k + 1] = segStart[k] + (size_t)(winEnd[j] - jobs[j].nodeOffset) / 2 + 1;
        }
        if (segStart[order.size()] < ((size_t)1 << 26) && ps.pairs.resize_uninit(segStart[order.size()])) {
            parallel_jobs((int)jobs.size(), [&](int j) {
                MeshJob& mj = jobs[j];
                mj.sb.nodes = nodes; mj.sb.nNodes = n_nodes; mj.sb.nTris = n_triangles;
                mj.sb.memoLo = mj.nodeOffset;
                mj.sb.pairOfFirstChild.assign((size_t)(winEnd[j] - mj.nodeOffset) + 1, -1);
                mj.sb.seg = ps.pairs.data() + mj.segStart;
                mj.sb.segCap = (size_t)(winEnd[j] - mj.nodeOffset) / 2 + 1;
                mj.sb.idBase = (uint32_t)mj.segStart;
                mj.sb.pairDepth.reserve(mj.sb.segCap);
                mj.sb.pairLeafEnd.reserve(mj.sb.segCap);
                mj.ok = mj.sb.convert(mj.nodeOffset, mj.triOffset, mj.nodeOffset, &mj.code, &mj.height, &mj.leafEnd);
                memset(static_cast<void*>(mj.sb.seg + mj.sb.segCount), 0, (mj.sb.segCap - mj.sb.segCount) * sizeof(DPair)); /* the unused tail of the segment */
            });
            merged = true;
            for (const MeshJob& mj : jobs)
                if (mj.sb.outside || !mj.sb.bigLeaves.empty()) merged = false; /* (oversized leaves index a table the meshes would share) */
        }
    }
    if (merged) {
        for (int i = 0; i < n_models; i++) { /* errors in model order, as the sequential walk reports them */
            const MeshJob& mj = jobs[jobOfModel[i]];
            if (!mj.ok) return fail(ctx, RT_ERR_SCENE, "model %d: %s", i, mj.sb.error.c_str());
            if ((long long)models[i].triOffset + mj.leafEnd > n_triangles) return fail(ctx, RT_ERR_SCENE, "model %d: leaf triangle range out of bounds", i);
            rootCodes[i] = mj.code;
            heights[i] = mj.height;
        }
        for (const MeshJob& mj : jobs)
            if (mj.segStart + mj.sb.segCount > nPairs) nPairs = mj.segStart + mj.sb.segCount;
        ps.pairs.shrink(nPairs);
        jobs.clear();
    } else {
        jobs.clear();
        sb.pairOfFirstChild.assign((size_t)n_nodes + 1, -1);
        for (int i = 0; i < n_models; i++) {
            const RtModel& m = models[i];
            if (!sb.convert(m.nodeOffset, m.triOffset, m.nodeOffset, &rootCodes[i], &heights[i]))
                return fail(ctx, RT_ERR_SCENE, "model %d: %s", i, sb.error.c_str());
        }
        nPairs = sb.pairs.size();
        if (nPairs < ((size_t)1 << 26)) {
            if (!ps.pairs.resize_uninit(nPairs)) return fail(ctx, RT_ERR_OOM, "rt_upload_scene: out of host memory");
            if (nPairs) memcpy(static_cast<void*>(ps.pairs.data()), sb.pairs.data(), nPairs * sizeof(DPair));
        }
        std::vector<DPair>().swap(sb.pairs);
    }
    int maxHeight = 1;
    for (int i = 0; i < n_models; i++) {
        const RtModel& m = models[i];
        const RtBVHNode& root = nodes[m.nodeOffset];
        const int height = heights[i];
        if (height > RT_MAX_BVH_DEPTH) return fail(ctx, RT_ERR_SCENE, "model %d: BVH depth %d > %d", i, height, RT_MAX_BVH_DEPTH);
        if (height > maxHeight) maxHeight = height;
        if (!(rootCodes[i] & RT_CODE_LEAF)) {
            rootChildren[2 * (size_t)i] = nodes[m.nodeOffset + root.startIndex];
            rootChildren[2 * (size_t)i + 1] = nodes[m.nodeOffset + root.startIndex + 1];
        } else { /* leaf root: the bounds of its triangles (validated by leaf_code above), count in triangleCount */
            RtBVHNode b;
            memset(&b, 0, sizeof(b));
            for (int d = 0; d < 3; d++) { b.boundsMin[d] = INFINITY; b.boundsMax[d] = -INFINITY; }
            bool fin = true;
            for (int t = 0; t < root.triangleCount; t++) {
                const RtTriangle& tr = triangles[(size_t)m.triOffset + root.startIndex + t];
                const float* vs[3] = {tr.posA, tr.posB, tr.posC};
                for (int v = 0; v < 3; v++)
                    for (int d = 0; d < 3; d++) {
                        fin = fin && std::isfinite(vs[v][d]);
                        b.boundsMin[d] = fminf(b.boundsMin[d], vs[v][d]);
                        b.boundsMax[d] = fmaxf(b.boundsMax[d], vs[v][d]);
                    }
            }
            b.triangleCount = (fin && root.triangleCount < (1 << 23)) ? root.triangleCount : 0; /* 0 = never filtered */
            rootChildren[2 * (size_t)i] = b;
            rootChildren[2 * (size_t)i + 1] = b;
        }
    }

    const auto tConv = std::chrono::steady_clock::now();
    /* the kernels address pairs and triangles with 32-bit byte offsets from the array bases (rt_kernels.h) */
    if (nPairs >= ((size_t)1 << 26) || (size_t)n_triangles * sizeof(DTri) >= ((size_t)1 << 32))
        return fail(ctx, RT_ERR_SCENE, "scene too large for 32-bit offsets: %zu node pairs (limit 2^26), %d triangles (limit 2^32 / 48)", nPairs, n_triangles);

    /* ---- the layout: canonical pairs + the caller's triangles -> the pair / triangle / normal spaces the kernels address in
     * 16-byte units (rt_layout.h); triangles are pre-differenced on the way (RC:190-192 are ray independent, same fp32 ops) */
    {
        RtLayout L;
        const char* want = layoutOverride ? layoutOverride : getenv("RT_LAYOUT");
        if (!parse_layout(want ? want : RT_LAYOUT_DEFAULT, &L)) return fail(ctx, RT_ERR_INVALID_ARG, "RT_LAYOUT=%s: unknown layout", want ? want : RT_LAYOUT_DEFAULT);
        {   /* the top-of-tree cache: as many records as the workgroups' LDS holds (or what RT_LAYOUT's cache=N says, within that) */
            bool anyInner = false;
            for (int i = 0; i < n_models; i++) anyInner = anyInner || !(rootCodes[i] & RT_CODE_LEAF);
            const GroupPlan gp = anyInner ? plan_groups(maxHeight, n_models) : GroupPlan();
            ps.wavesPerGroup = gp.wavesPerGroup;
            /* Default rule (no `cache` word): the cache is used where it was measured to pay — scenes whose trees are small enough that the
             * records the LDS holds are at least 1/16 of all node pairs (configs 3 and 6: 96 % / 73 % of the inner steps served, frame time
             * 0 / - 1.1 % against the single-wave kernel; config 4 / 5 with 0.2 % / 0.01 % coverage: 63 % / 39 % served, + 3.6 % / - 0.3 %:
             * profiles/r06_groups_and_streams.txt).  Without it the BVH variants are round 5's single-wave workgroups on two streams. */
            if (L.cacheRecords == -1) L.cacheRecords = ((size_t)gp.cacheRecords * 16 >= nPairs) ? gp.cacheRecords : 0;
            if (L.cacheRecords < 0 || L.cacheRecords > gp.cacheRecords) L.cacheRecords = gp.cacheRecords;
            if (L.dense()) L.cacheRecords = 0; /* the dense layout moves nothing */
        }
        LayoutEngine eng;
        eng.canon = ps.pairs.data();
        eng.nCanon = nPairs;
        eng.canonBig = &sb.bigLeaves;
        eng.models = models;
        eng.nModels = n_models;
        eng.rootCodes = rootCodes.data();
        eng.tris = triangles;
        eng.nTris = n_triangles;
        eng.parallel = [](int n, void* c, void (*f)(void*, int)) { parallel_jobs(n, [&](int k) { f(c, k); }); };
        if (!eng.run(L, ps.pairs, ps.lay)) {
            const bool oom = ps.lay.error == "out of host memory";
            return fail(ctx, oom ? RT_ERR_OOM : RT_ERR_SCENE, "rt_upload_scene: %s", ps.lay.error.c_str());
        }
        ps.nPairs = nPairs;
        rootCodes = ps.lay.rootCodes; /* final codes from here on (only their leaf bit is read below) */
    }
    if (getenv("RT_DEBUG_UPLOAD"))
        fprintf(stderr, "[rt] prepare_scene: layout %s (%zu + %zu + %zu bytes) in %.2f ms\n", ps.lay.used.name().c_str(), ps.lay.pairBuf.size(), ps.lay.triBuf.size(),
                ps.lay.normBuf.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tConv).count());
    pack_spheres(spheres, n_spheres, ps.sph, &ps.sphereBound);
    ps.mats.resize((size_t)n_spheres + n_models);
    for (int i = 0; i < n_spheres; i++) pack_material(spheres[i].material, ps.mats[i]);
    ps.dmodels.resize(n_models);
    for (int i = 0; i < n_models; i++) {
        pack_model(models[i], rootCodes[i], ps.lay.triBase[i], ps.dmodels[i]);
        pack_material(models[i].material, ps.mats[n_spheres + i]);
    }
    make_filters(models, n_models, rootCodes, rootChildren, spheres, n_spheres, ps.filters, &ps.maxOrigin);
    make_chunks(ps.filters, ps.chunks, &ps.nFiltered, &ps.extWords);
    ps.filters = append_filter_pairs(ps.filters); /* uploaded as one array */
    ps.hModels.assign(models, models + n_models);
    ps.hSpheres.assign(spheres, spheres + n_spheres);
    ps.nTris = n_triangles;
    ps.maxHeight = maxHeight;
    ps.flat = true;
    for (int i = 0; i < n_models; i++)
        if (!(rootCodes[i] & RT_CODE_LEAF)) ps.flat = false;
    return RT_OK;
}

/* device copy of one scene array: from the host vector, or — `peer` — from the context that already holds it, device to
 * device on this context's stream (xGMI when the devices differ; the caller synchronises the stream) */
template <typename T, typename V>
static int commit_vec(RtContext* ctx, T** dptr, const V& v, T* const* peerPtr, const RtContext* peer)
{
    if (!peer) return upload_vec(ctx, dptr, v.data(), v.size());
    const size_t bytes = v.size() * sizeof(T);
    HIP_TRY(ctx, hipMalloc(dptr, bytes ? bytes : sizeof(T)));
    if (bytes) HIP_TRY(ctx, hipMemcpyPeerAsync(*dptr, ctx->device, *peerPtr, peer->device, bytes, ctx->stream));
    return RT_OK;
}

static int commit_scene(RtContext* ctx, const PreparedScene& ps, const RtContext* peer)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
    free_scene(ctx);
    int rc;
    if ((rc = commit_vec(ctx, &ctx->dSpheres, ps.sph, peer ? &peer->dSpheres : nullptr, peer))) return rc;
    if ((rc = commit_vec(ctx, &ctx->dMaterials, ps.mats, peer ? &peer->dMaterials : nullptr, peer))) return rc;
    if ((rc = commit_vec(ctx, &ctx->dModels, ps.dmodels, peer ? &peer->dModels : nullptr, peer))) return rc;
    if ((rc = commit_vec(ctx, &ctx->dPairs, ps.lay.pairBuf, peer ? &peer->dPairs : nullptr, peer))) return rc;
    if (!ps.lay.arena && (rc = commit_vec(ctx, &ctx->dTris, ps.lay.triBuf, peer ? &peer->dTris : nullptr, peer))) return rc;
    if ((rc = commit_vec(ctx, &ctx->dNorms, ps.lay.normBuf, peer ? &peer->dNorms : nullptr, peer))) return rc;
    if ((rc = commit_vec(ctx, &ctx->dBigLeaves, ps.lay.bigLeaves, peer ? &peer->dBigLeaves : nullptr, peer))) return rc;
    ctx->arenaLayout = ps.lay.arena;
    ctx->layoutUsed = ps.lay.used.name();
    if ((rc = commit_vec(ctx, &ctx->dFilters, ps.filters, peer ? &peer->dFilters : nullptr, peer))) return rc;
    if ((rc = commit_vec(ctx, &ctx->dChunks, ps.chunks, peer ? &peer->dChunks : nullptr, peer))) return rc;
    ctx->nChunks = (int)ps.chunks.size();
    ctx->nFiltered = ps.nFiltered;
    ctx->extWords = ps.extWords;
    ctx->filterMaxOrigin = ps.maxOrigin;
    ctx->sphereBound = ps.sphereBound;
    ctx->hRootChildren = ps.rootChildren;
    ctx->hSpheres = ps.hSpheres;
    ctx->nSpheres = (int)ps.hSpheres.size();
    ctx->nModels = (int)ps.hModels.size();
    ctx->nTris = ps.nTris;
    ctx->nPairs = (int)ps.nPairs;
    ctx->hTriBase = ps.lay.triBase;
    ctx->stackEntries = ps.flat ? 0 : ps.maxHeight; /* (the FLAT variant pushes nothing: its LDS is pixel bookkeeping only) */
    ctx->flatScene = ps.flat;
    ctx->hotUnits = ps.flat ? 0u : ps.lay.hotUnits;
    {   /* one ray, one segment: every model once (a step each), every pair and every leaf of its tree at most once — and the same tree once
         * per model that uses it.  64 lanes, one step of one lane per iteration at least; the factor 2 is slack, not arithmetic. */
        const unsigned long long steps = (unsigned long long)ps.hModels.size() * (2ull * ps.nPairs + 2ull) + 16ull;
        const unsigned long long lim = 2ull * 64ull * steps;
        ctx->travLimit = lim > 0x7fffffffull ? 0x7fffffffu : (uint32_t)lim;
        if (const char* e = getenv("RT_TRAV_LIMIT")) ctx->travLimit = (uint32_t)strtoul(e, nullptr, 10); /* test hook: make the watchdog fire */
    }
    ctx->wavesPerGroup = ctx->hotUnits ? ps.wavesPerGroup : 1;
    {   /* the FLAT variant's chain pool (rt_kernels.h, pool_exchange): 16-wave workgroups (two per CU at the variant's 8 waves per SIMD), two
         * queues of 64 cells of 128 bytes — 17 KB of the workgroup's 70 KB.  RT_POOL=0: single-wave workgroups without a pool (rounds 1-5);
         * RT_POOL_WAVES: A/B runs and tests */
        int on = 1, waves = RT_MAX_WAVES_PER_GROUP_FLAT;
        if (const char* e = getenv("RT_POOL")) on = atoi(e);
        if (const char* e = getenv("RT_POOL_WAVES")) waves = atoi(e);
        if (waves < 1) waves = 1;
        if (waves > RT_MAX_WAVES_PER_GROUP_FLAT) waves = RT_MAX_WAVES_PER_GROUP_FLAT;
        const bool pooled = ps.flat && on;
        ctx->poolWaves = pooled ? waves : 1;
        ctx->poolCells = pooled ? (int)RT_POOL_CELLS : 0;
    }
    ctx->hModels = ps.hModels;
    ctx->hRootCodes = ps.rootCodes;
    ctx->haveScene = true;
    ctx->tuner.done = getenv("RT_SUSPEND") != nullptr; /* RT_SUSPEND=3|4 pins the threshold (tests, A/B runs) */
    ctx->tuner.decided = ctx->tuner.done ? atoi(getenv("RT_SUSPEND")) : 3;
    if (ctx->tuner.decided < 1 || ctx->tuner.decided > 7) ctx->tuner.decided = 3;
    ctx->tuner.ms[0] = ctx->tuner.ms[1] = 0;
    ctx->tuner.n[0] = ctx->tuner.n[1] = 0;
    ctx->tuner.next = 0;
    for (auto& pr : ctx->tuner.probe) pr.live = false;
    return RT_OK;
}

extern "C" {

int rt_upload_scene(RtContext* ctx, const RtModel* models, int n_models, const RtTriangle* triangles, int n_triangles,
                    const RtBVHNode* nodes, int n_nodes, const RtSphere* spheres, int n_spheres)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    PreparedScene ps;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = prepare_scene(ctx, models, n_models, triangles, n_triangles, nodes, n_nodes, spheres, n_spheres, ps);
    if (rc) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    rc = commit_scene(ctx, ps, nullptr);
    if (getenv("RT_DEBUG_UPLOAD"))
        fprintf(stderr, "[rt] rt_upload_scene: prepare %.2f ms, commit %.2f ms (%d triangles, %zu node pairs)\n",
                std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count(), n_triangles, ps.nPairs);
    return rc;
}

int rt_validate_scene(const RtModel* models, int n_models, const RtTriangle* triangles, int n_triangles,
                      const RtBVHNode* nodes, int n_nodes, const RtSphere* spheres, int n_spheres, RtSceneInfo* out_info)
{
    PreparedScene ps;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = prepare_scene(nullptr, models, n_models, triangles, n_triangles, nodes, n_nodes, spheres, n_spheres, ps);
    if (out_info) {
        memset(out_info, 0, sizeof(*out_info));
        out_info->prepare_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rc == RT_OK) {
            out_info->n_pairs = (int32_t)ps.nPairs;
            out_info->max_height = ps.maxHeight;
            out_info->flat = ps.flat ? 1 : 0;
            out_info->n_filtered = ps.nFiltered;
        }
    }
    return rc;
}

int rt_debug_layout(const RtModel* models, int n_models, const RtTriangle* triangles, int n_triangles,
                    const RtBVHNode* nodes, int n_nodes, const char* layout, RtLayoutDump* out)
{
    if (!out) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_debug_layout: null output");
    memset(out, 0, sizeof(*out));
    PreparedScene ps;
    const int rc = prepare_scene(nullptr, models, n_models, triangles, n_triangles, nodes, n_nodes, nullptr, 0, ps, layout);
    if (rc) return rc;
    auto take = [](PodVec<unsigned char>& v, unsigned char** p, size_t* n) { *p = v.p; *n = v.n; v.p = nullptr; v.n = 0; };
    take(ps.lay.pairBuf, &out->pair_space, &out->pair_bytes);
    take(ps.lay.triBuf, &out->tri_space, &out->tri_bytes);
    take(ps.lay.normBuf, &out->norm_space, &out->norm_bytes);
    auto dup = [](const void* src, size_t bytes) { void* q = malloc(bytes ? bytes : 1); if (q && bytes) memcpy(q, src, bytes); return q; };
    out->n_big_leaves = ps.lay.bigLeaves.size() / 2;
    out->big_leaves = (uint32_t*)dup(ps.lay.bigLeaves.data(), ps.lay.bigLeaves.size() * 4);
    out->root_codes = (uint32_t*)dup(ps.lay.rootCodes.data(), ps.lay.rootCodes.size() * 4);
    out->tri_base = (int32_t*)dup(ps.lay.triBase.data(), ps.lay.triBase.size() * 4);
    out->n_models = n_models;
    out->arena = ps.lay.arena ? 1 : 0;
    snprintf(out->used, sizeof(out->used), "%s", ps.lay.used.name().c_str());
    return RT_OK;
}

void rt_debug_layout_free(RtLayoutDump* d)
{
    if (!d) return;
    free(d->pair_space); free(d->tri_space); free(d->norm_space); free(d->big_leaves); free(d->root_codes); free(d->tri_base);
    memset(d, 0, sizeof(*d));
}

int rt_update_models(RtContext* ctx, const RtModel* models, int n_models)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    if (!ctx->haveScene) return fail(ctx, RT_ERR_STATE, "rt_update_models before rt_upload_scene");
    if (n_models != ctx->nModels || (n_models && !models)) return fail(ctx, RT_ERR_INVALID_ARG, "rt_update_models: model count changed (%d != %d)", n_models, ctx->nModels);
    if (n_models == 0) return RT_OK;
    /* RCM:192-204 runs every frame; in a static scene nothing changed: no device work at all */
    if (memcmp(models, ctx->hModels.data(), sizeof(RtModel) * (size_t)n_models) == 0) {
        ctx->updateSkips++;
        return RT_OK;
    }
    RT_FLUSH(ctx); /* frames held back were requested with the old models */
    std::vector<DModel> dmodels(n_models);
    std::vector<DMaterial> mats(n_models);
    bool matricesChanged = false;
    for (int i = 0; i < n_models; i++) {
        if (models[i].nodeOffset != ctx->hModels[i].nodeOffset || models[i].triOffset != ctx->hModels[i].triOffset)
            return fail(ctx, RT_ERR_INVALID_ARG, "rt_update_models: model %d changed its BVH offsets; re-upload the scene", i);
        pack_model(models[i], ctx->hRootCodes[i], ctx->hTriBase[i], dmodels[i]);
        pack_material(models[i].material, mats[i]);
        if (memcmp(models[i].worldToLocal, ctx->hModels[i].worldToLocal, sizeof(models[i].worldToLocal)) != 0 ||
            memcmp(models[i].localToWorld, ctx->hModels[i].localToWorld, sizeof(models[i].localToWorld)) != 0)
            matricesChanged = true;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    /* stream-ordered: lands after the frames already enqueued, before the next launch */
    int rc;
    if ((rc = stage_upload(ctx, ctx->dModels, dmodels.data(), sizeof(DModel) * n_models))) return rc;
    if ((rc = stage_upload(ctx, ctx->dMaterials + ctx->nSpheres, mats.data(), sizeof(DMaterial) * n_models))) return rc;
    if (matricesChanged) { /* the world-space root filter boxes depend on the matrices only */
        std::vector<DFilter> filters;
        make_filters(models, n_models, ctx->hRootCodes, ctx->hRootChildren, ctx->hSpheres.data(), (int)ctx->hSpheres.size(), filters, &ctx->filterMaxOrigin);
        {
            const std::vector<DFilter> up = append_filter_pairs(filters);
            if ((rc = stage_upload(ctx, ctx->dFilters, up.data(), sizeof(DFilter) * up.size()))) return rc;
        }
        if ((rc = refresh_chunks(ctx, filters))) return rc;
    }
    ctx->hModels.assign(models, models + n_models);
    return RT_OK;
}

int rt_update_spheres(RtContext* ctx, const RtSphere* spheres, int n_spheres)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    if (!ctx->haveScene) return fail(ctx, RT_ERR_STATE, "rt_update_spheres before rt_upload_scene");
    if (n_spheres != ctx->nSpheres || (n_spheres && !spheres)) return fail(ctx, RT_ERR_INVALID_ARG, "rt_update_spheres: sphere count changed");
    if (n_spheres == 0) return RT_OK;
    if (memcmp(spheres, ctx->hSpheres.data(), sizeof(RtSphere) * (size_t)n_spheres) == 0) {
        ctx->updateSkips++;
        return RT_OK;
    }
    RT_FLUSH(ctx);
    std::vector<float> sph;
    pack_spheres(spheres, n_spheres, sph, &ctx->sphereBound);
    std::vector<DMaterial> mats(n_spheres);
    for (int i = 0; i < n_spheres; i++) pack_material(spheres[i].material, mats[i]);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = stage_upload(ctx, ctx->dSpheres, sph.data(), sph.size() * 4))) return rc;
    if ((rc = stage_upload(ctx, ctx->dMaterials, mats.data(), sizeof(DMaterial) * n_spheres))) return rc;
    ctx->hSpheres.assign(spheres, spheres + n_spheres);
    if (ctx->nModels) { /* the filter margins scale with the scene extent, which includes the spheres */
        std::vector<DFilter> filters;
        make_filters(ctx->hModels.data(), ctx->nModels, ctx->hRootCodes, ctx->hRootChildren, spheres, n_spheres, filters, &ctx->filterMaxOrigin);
        {
            const std::vector<DFilter> up = append_filter_pairs(filters);
            if ((rc = stage_upload(ctx, ctx->dFilters, up.data(), sizeof(DFilter) * up.size()))) return rc;
        }
        if ((rc = refresh_chunks(ctx, filters))) return rc;
    }
    return RT_OK;
}

int rt_set_params(RtContext* ctx, const RtParams* p)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    if (!p) return fail(ctx, RT_ERR_INVALID_ARG, "rt_set_params: null params");
    if (p->abi_version != RT_ABI_VERSION || p->struct_size != sizeof(RtParams))
        return fail(ctx, RT_ERR_ABI_MISMATCH, "rt_set_params: abi_version %u / struct_size %u, library has %u / %zu", p->abi_version,
                    p->struct_size, (unsigned)RT_ABI_VERSION, sizeof(RtParams));
    if (ctx->haveParams && p->frame == ctx->frame) { /* SetShaderParams runs every frame (RCM:122): usually nothing but Frame moved, and it moved to where the context already is */
        RtParams a = *p, b = ctx->params;
        a.frame = b.frame = 0;
        if (memcmp(&a, &b, sizeof(RtParams)) == 0) return RT_OK;
    }
    RT_FLUSH(ctx);
    ctx->params = *p;
    ctx->frame = p->frame;
    ctx->haveParams = true;
    return RT_OK;
}

int rt_reset_accumulation(RtContext* ctx)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    if (ctx->W == 0) return fail(ctx, RT_ERR_STATE, "rt_reset_accumulation before rt_resize");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float* accum = ctx->boundAccum ? ctx->boundAccum : ctx->ownAccum;
    size_t n = (size_t)ctx->localRows * ctx->W;
    if (n) {
        int blocks = (int)((n + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(rtk::rt_reset_kernel, dim3(blocks), dim3(256), 0, joined(ctx), (float4*)accum, n);
        HIP_TRY(ctx, hipGetLastError());
    }
    ctx->frame = 1; /* RCM:71 */
    return RT_OK;
}

static void fill_args(RtContext* ctx, int frame0, int nFrames, KArgs& a)
{
    memset(&a, 0, sizeof(a));
    a.spheres = ctx->dSpheres;
    a.sphereQuick = ctx->dSpheres + 4 * (size_t)ctx->nSpheres;
    a.sphereBound = ctx->sphereBound;
    a.materials = ctx->dMaterials;
    a.models = ctx->dModels;
    a.pairs = reinterpret_cast<const DPair*>(ctx->dPairs);
    a.tris = reinterpret_cast<const DTri*>(ctx->arenaLayout ? ctx->dPairs : ctx->dTris);
    a.norms = reinterpret_cast<const DTriN*>(ctx->dNorms);
    a.bigLeaves = ctx->dBigLeaves;
    a.filters = ctx->dFilters;
    a.filterPairs = reinterpret_cast<const float*>(ctx->dFilters + ctx->nModels);
    a.chunks = ctx->dChunks;
    a.nChunks = ctx->nChunks;
    a.nFiltered = ctx->nFiltered;
    a.extWords = ctx->extWords;
    a.filterMaxOrigin = ctx->filterMaxOrigin;
    a.nSpheres = ctx->nSpheres;
    a.nModels = ctx->nModels;
    a.frameRender = ctx->boundFrame ? ctx->boundFrame : ctx->ownFrame;
    a.accumulated = ctx->boundAccum ? ctx->boundAccum : ctx->ownAccum;
    a.W = (uint32_t)ctx->W;
    a.H = (uint32_t)ctx->H;
    a.localRows = ctx->localRows;
    a.stripRows = ctx->stripRows;
    a.partIndex = ctx->partIndex;
    a.partCount = ctx->partCount;
    a.tilesX = (ctx->W + 7) / 8;
    a.tilesY = (ctx->localRows + 7) / 8;
    const RtParams& p = ctx->params;
    a.maxBounce = p.maxBounceCount;
    a.spp = p.numRaysPerPixel;
    a.frame0 = frame0;
    a.nFrames = nFrames;
    a.seed = p.renderSeed;
    a.useSky = p.useSky;
    a.accumulate = p.accumulate;
    a.defocus = p.defocusStrength;
    a.diverge = p.divergeStrength;
    a.sunFocus = p.sunFocus;
    a.sunIntensity = p.sunIntensity;
    memcpy(a.sunColour, p.sunColour, 12);
    memcpy(a.dirToSun, p.dirToSun, 12);
    memcpy(a.viewParams, p.viewParams, 12);
    memcpy(a.cam, p.camLocalToWorld, 64);
    a.rcpWm1 = rt_rcp((float)a.W - 1.0f);
    a.rcpHm1 = rt_rcp((float)a.H - 1.0f);
    a.rcpW = rt_rcp((float)a.W);
    a.rcpSpp = rt_rcp((float)a.spp);
    {
        /* x + (+-0) == x bit for bit unless x is -0 (then the sign of the sum follows the random jitter's sign):
         * the kernel may skip the defocus jitter's sin/cos/sqrt when no component of the camera origin — computed
         * here with the kernel's own rt_mul_point — is a negative zero */
        bool fin = true;
        for (int k = 0; k < 16; k++) fin = fin && std::isfinite(a.cam[k]);
        const rt_f3 o = rt_mul_point(a.cam, rt_v3(0.0f, 0.0f, 0.0f), 1.0f);
        const bool noNegZero = rt_f2u(o.x) != 0x80000000u && rt_f2u(o.y) != 0x80000000u && rt_f2u(o.z) != 0x80000000u;
        a.raygenNoDefocus = (a.defocus == 0.0f && fin && noNegZero && std::isfinite(a.rcpW)) ? 1 : 0;
    }
    a.counters = ctx->dCounters;
    a.travLimit = ctx->travLimit;
    a.wavesPerGroup = 1; /* (choose_variant decides the workgroup shape of the trace kernels; the debug hooks run single waves without a cache) */
    a.hotUnits = 0;
    a.waveLdsDwords = 0;
    a.poolCells = 0;
    a.poolSpinLimit = 1u << 16;
}

} /* extern "C" */

/* =====================================================================================================================
 * The launch path.  rt_render_frame / rt_render_frames / flush_pending end in launch_frames(ctx, frame0, nFrames):
 *     choose_variant    which kernel instantiation, how much LDS, how many workgroups the chip keeps resident (LaunchPlan)
 *     prepare_*         the buffers a launch needs — pixel records, tile order, staging slab; may synchronise (rarely)
 *     enqueue_*         kernels and events, in the order LaunchOrder's rules demand
 * Dispatch semantics kept: RayComputeManager.cs:84-95 (one RayTrace dispatch per frame, Frame counts the accumulated frames).
 * ===================================================================================================================== */
struct LaunchPlan {
    void (*kern)(const KArgs) = nullptr;     /* a launch that renders a whole frame or a batch of frames */
    void (*kernHalf)(const KArgs) = nullptr; /* the same code under a second name: the two halves of a two-part frame (rt_kernels.h) */
    size_t ldsBytes = 0;
    int blockThreads = RT_WAVE;
    int variant = 0;             /* slot of the occupancy cache */
    int wavesPerGroup = 1;       /* waves of a workgroup: each is one persistent wave of the launch */
    long long resident = 0;      /* workgroups the chip keeps resident */
};

/* ---------------------------------------------------------------------------------------------------------------------
 * LaunchOrder — every wait between the context's two render streams (s = 0 main, 1 side), as named steps.
 *
 *   what                          who writes it                     rule
 *   non-render work (uploads,     the main stream, through           fork_side: before the side stream's next kernel it waits for evFork,
 *    resets, read-backs)           joined()                          recorded on the main stream (needFork); joined() itself makes the main
 *                                                                    stream wait for what the side stream holds (evJoin, sideDirty)
 *   tile order buffer b           rt_order_kernel on stream ss       begin_sort: the rewrite of b waits for b's last readers on the other stream
 *                                                                    (evOrderRetire[b]) and — the sorts share one key snapshot — for the previous
 *                                                                    sort if that ran on the other stream; end_sort records evSort, marks the
 *                                                                    other stream (sortPending) and retires the buffer going out of use;
 *                                                                    before_order_read: a stream's next trace kernel waits for evSort once
 *   accumulation buffer           single-frame trace kernels          before_acc_write(s, whole): wait for the OTHER stream's last whole-image
 *                                  (whole image, or the half of a     writer (evAccFull) and, if this kernel or that stream's last writer covers
 *                                  two-part frame), accumulate        the whole image, for its last writer of any kind (evAccWriter).  Halves of
 *                                  kernels (whole image)              consecutive two-part frames need no wait: between two sorts a stream's half
 *                                                                    is the same set of pixels — end_sort therefore declares both last writers
 *                                                                    "whole".  after_acc_write records the events.
 *   staging slab i, pixel         launches on stream i only          stream order (nothing here)
 *    records i, tile counter i
 * --------------------------------------------------------------------------------------------------------------------- */
namespace LaunchOrder {
static inline hipStream_t stream_of(RtContext* ctx, int s) { return s ? ctx->sideStream : ctx->stream; }

static int fork_side(RtContext* ctx)
{
    if (!ctx->ord.needFork) return RT_OK;
    HIP_TRY(ctx, hipEventRecord(ctx->ord.evFork, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->sideStream, ctx->ord.evFork, 0));
    ctx->ord.needFork = false;
    return RT_OK;
}
static void side_used(RtContext* ctx) { ctx->ord.sideDirty = true; }

static int begin_sort(RtContext* ctx, int ss, int target)
{
    RtContext::Order& o = ctx->ord;
    if (ss == 1) { int rc = fork_side(ctx); if (rc) return rc; }
    hipStream_t S = stream_of(ctx, ss);
    if (o.retireValid[target]) HIP_TRY(ctx, hipStreamWaitEvent(S, o.evOrderRetire[target], 0));
    if (o.sortPending[ss]) HIP_TRY(ctx, hipStreamWaitEvent(S, o.evSort, 0)); /* (two sorts in consecutive launches on alternating streams) */
    return RT_OK;
}
static int end_sort(RtContext* ctx, int ss, bool twoStreams, int retiring /* buffer going out of use, or -1 */)
{
    RtContext::Order& o = ctx->ord;
    HIP_TRY(ctx, hipEventRecord(o.evSort, stream_of(ctx, ss)));
    o.sortPending[ss] = false;
    o.sortPending[1 - ss] = twoStreams;
    if (retiring >= 0 && twoStreams) { /* whatever the other stream holds so far may still read it */
        HIP_TRY(ctx, hipEventRecord(o.evOrderRetire[retiring], stream_of(ctx, 1 - ss)));
        o.retireValid[retiring] = true;
    }
    if (ss == 1) side_used(ctx);
    /* a new order moves pixels between the two halves of a two-part frame: the next such frame's halves must come after BOTH
     * streams' last kernels that add into the accumulation buffer */
    o.accWriterFull[0] = o.accWriterFull[1] = true;
    return RT_OK;
}
static int before_order_read(RtContext* ctx, int s)
{
    if (!ctx->ord.sortPending[s]) return RT_OK;
    HIP_TRY(ctx, hipStreamWaitEvent(stream_of(ctx, s), ctx->ord.evSort, 0));
    ctx->ord.sortPending[s] = false;
    return RT_OK;
}
static void forget_order(RtContext* ctx) /* the order buffers were re-made (another image size) */
{
    ctx->ord.sortPending[0] = ctx->ord.sortPending[1] = false;
    ctx->ord.retireValid[0] = ctx->ord.retireValid[1] = false;
}

static int before_acc_write(RtContext* ctx, int s, bool whole)
{
    RtContext::Order& o = ctx->ord;
    hipStream_t st = stream_of(ctx, s);
    if (o.accFullPending[1 - s]) { /* once waited for, everything later on this stream follows it */
        HIP_TRY(ctx, hipStreamWaitEvent(st, o.evAccFull[1 - s], 0));
        o.accFullPending[1 - s] = false;
    }
    if (o.accWriterPending[1 - s] && (whole || o.accWriterFull[1 - s])) HIP_TRY(ctx, hipStreamWaitEvent(st, o.evAccWriter[1 - s], 0));
    return RT_OK;
}
static int after_acc_write(RtContext* ctx, int s, bool whole)
{
    RtContext::Order& o = ctx->ord;
    HIP_TRY(ctx, hipEventRecord(o.evAccWriter[s], stream_of(ctx, s)));
    o.accWriterPending[s] = true;
    o.accWriterFull[s] = whole;
    if (whole) {
        HIP_TRY(ctx, hipEventRecord(o.evAccFull[s], stream_of(ctx, s)));
        o.accFullPending[s] = true;
    }
    return RT_OK;
}
/* a whole-image writer on stream s was ordered after everything the other stream wrote (before_acc_write(s, true)): later writers wait for it */
static void other_stream_settled(RtContext* ctx, int s) { ctx->ord.accWriterPending[1 - s] = false; }
} // namespace LaunchOrder

/* ---- choose variant: kernel instantiation, LDS, resident workgroups */
static int choose_variant(RtContext* ctx, KArgs& a, LaunchPlan& plan, bool* manyOut)
{
    const size_t coldBytes = ctx->flatScene ? (size_t)2 * RT_WAVE * 16 : 0; /* the FLAT variant keeps its pixel records in LDS (rt_kernels.h, PX_COLD) */
    /* a wave: traversal stack + pixel fields + (more than 64 models) the mask extension: summary + words + the MANY variant's bounce row;
     * a workgroup of the BVH variants: the top-of-tree cache, then its waves' regions (plan_groups) */
    const size_t waveBytes = wave_lds_bytes(ctx->stackEntries, ctx->extWords) + coldBytes;
    /* the shared region in front of the waves' regions: the BVH variants' top-of-tree cache, or the FLAT variant's chain pool */
    /* ... for launches with enough work: a pooled workgroup is 16 persistent waves that leave together, and a launch with a few items per wave is
     * all tail (config 1, 256 x 256: + 11 % with the pool) — at least RT_POOL_MIN_ITEMS (tile, frame) pairs per wave the chip keeps resident */
    const bool pooled = ctx->flatScene && ctx->poolCells > 0 &&
                        (long long)a.tilesX * a.tilesY * (a.nFrames > 0 ? a.nFrames : 1) >= (long long)ctx->poolMinItems * ctx->numCUs * 4 * RT_MIN_WAVES_PER_SIMD_FLAT;
    const uint32_t hotUnits = ctx->flatScene ? (pooled ? (uint32_t)(RT_POOL_DWORDS / 4u) : 0u) : ctx->hotUnits;
    const int wpb = ctx->flatScene ? (pooled ? ctx->poolWaves : 1) : (hotUnits ? ctx->wavesPerGroup : 1);
    plan.wavesPerGroup = wpb;
    plan.blockThreads = RT_WAVE * wpb;
    plan.ldsBytes = (size_t)hotUnits * 16 + (size_t)wpb * waveBytes;
    a.wavesPerGroup = wpb;
    a.hotUnits = (int32_t)hotUnits;
    a.poolCells = pooled ? ctx->poolCells : 0;
    a.poolSpinLimit = ctx->poolSpinLimit;
    a.waveLdsDwords = (int32_t)(waveBytes / sizeof(uint32_t));
    a.stackEntries = ctx->stackEntries;
    const bool many = ctx->nChunks > 0 && !ctx->flatScene;
    *manyOut = many;
    const bool hot = hotUnits > 0; /* the instantiations launched as multi-wave workgroups: with the LDS top-of-tree cache, or (FLAT) the chain pool */
    plan.kern = ctx->flatScene ? (pooled ? (ctx->stats ? rtk::rt_trace_kernel<true, true, false, true> : rtk::rt_trace_kernel<false, true, false, true>)
                                         : (ctx->stats ? rtk::rt_trace_kernel<true, true> : rtk::rt_trace_kernel<false, true>))
                : many         ? (hot ? (ctx->stats ? rtk::rt_trace_kernel<true, false, true, true> : rtk::rt_trace_kernel<false, false, true, true>)
                                      : (ctx->stats ? rtk::rt_trace_kernel<true, false, true> : rtk::rt_trace_kernel<false, false, true>))
                               : (hot ? (ctx->stats ? rtk::rt_trace_kernel<true, false, false, true> : rtk::rt_trace_kernel<false, false, false, true>)
                                      : (ctx->stats ? rtk::rt_trace_kernel<true, false> : rtk::rt_trace_kernel<false, false>));
    plan.kernHalf = ctx->flatScene ? (pooled ? (ctx->stats ? rtk::rt_trace_half_kernel<true, true, false, true> : rtk::rt_trace_half_kernel<false, true, false, true>)
                                             : (ctx->stats ? rtk::rt_trace_half_kernel<true, true> : rtk::rt_trace_half_kernel<false, true>))
                    : many         ? (hot ? (ctx->stats ? rtk::rt_trace_half_kernel<true, false, true, true> : rtk::rt_trace_half_kernel<false, false, true, true>)
                                          : (ctx->stats ? rtk::rt_trace_half_kernel<true, false, true> : rtk::rt_trace_half_kernel<false, false, true>))
                                   : (hot ? (ctx->stats ? rtk::rt_trace_half_kernel<true, false, false, true> : rtk::rt_trace_half_kernel<false, false, false, true>)
                                          : (ctx->stats ? rtk::rt_trace_half_kernel<true, false> : rtk::rt_trace_half_kernel<false, false>));
    plan.variant = (ctx->flatScene ? 2 : many ? 4 : 0) + (ctx->stats ? 1 : 0) + (hot ? 6 : 0);
    if (ctx->occBytes[plan.variant] != plan.ldsBytes + 1) { /* occupancy query cached per (variant, LDS bytes) */
        int perCU = 0;
        if (plan.ldsBytes > 48 * 1024) { /* more dynamic LDS than the default limit of a launch */
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(plan.kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.ldsBytes));
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(plan.kernHalf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.ldsBytes));
        }
        HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, plan.kern, plan.blockThreads, plan.ldsBytes));
        ctx->occPerCU[plan.variant] = perCU > 0 ? perCU : 1;
        ctx->occBytes[plan.variant] = plan.ldsBytes + 1;
        if (getenv("RT_DEBUG_LAUNCH"))
            fprintf(stderr, "[rt] kernel variant %d: %d stack entries, %zu B of LDS per workgroup of %d threads (%u B of it the top-of-tree cache: %u records), %d workgroups per CU\n",
                    plan.variant, ctx->stackEntries, plan.ldsBytes, plan.blockThreads, hotUnits * 16u, hotUnits / 4u, perCU);
    }
    plan.resident = (long long)ctx->occPerCU[plan.variant] * ctx->numCUs;
    return RT_OK;
}

/* ---- prepare buffers: the resident waves' pixel records */
static int prepare_records(RtContext* ctx, const LaunchPlan& plan)
{
    const long long resWaves = plan.resident * plan.wavesPerGroup;
    const long long waves = (ctx->gridOverride > resWaves ? ctx->gridOverride : resWaves) + plan.wavesPerGroup;
    if (ctx->pxColdWaves < waves) { /* kernels in flight use the old block: freed after a synchronise, not now */
        HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
        hipFree(ctx->dPxCold); ctx->dPxCold = nullptr; ctx->pxColdWaves = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->dPxCold, (size_t)2 * waves * RT_COLD_STRIDE_BYTES));
        ctx->pxColdWaves = waves;
    }
    return RT_OK;
}

/* ---- prepare buffers: longest-chain-first queue order, learnt from the frames already rendered at this size.  Re-sorted once 1, 2, 4, 8,
 * ... frames have been recorded; the sort runs on the stream of the launch that first uses it, writes the order buffer that no running
 * kernel reads, works on its own snapshot of the costs (running kernels keep raising them) — it never joins the two streams (a join in the
 * middle of back-to-back launches serialises the next launch behind the drain of the previous one; round 4). */
static int prepare_tile_order(RtContext* ctx, KArgs& a, int tiles, int nFrames, int lane, bool twoOwn)
{
    if (!ctx->lptEnabled) return RT_OK;
    if (ctx->orderTiles != tiles) {
        HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
        hipFree(ctx->dTileCost); ctx->dTileCost = nullptr;
        hipFree(ctx->dTileOrder[0]); hipFree(ctx->dTileOrder[1]); ctx->dTileOrder[0] = ctx->dTileOrder[1] = nullptr;
        hipFree(ctx->dTileKey); ctx->dTileKey = nullptr;
        HIP_TRY(ctx, hipMalloc(&ctx->dTileCost, sizeof(uint32_t) * tiles));
        HIP_TRY(ctx, hipMalloc(&ctx->dTileOrder[0], sizeof(uint32_t) * tiles));
        HIP_TRY(ctx, hipMalloc(&ctx->dTileOrder[1], sizeof(uint32_t) * tiles));
        HIP_TRY(ctx, hipMalloc(&ctx->dTileKey, sizeof(uint32_t) * tiles));
        HIP_TRY(ctx, hipMemsetAsync(ctx->dTileCost, 0, sizeof(uint32_t) * tiles, joined(ctx)));
        ctx->orderTiles = tiles;
        ctx->orderValid = false;
        ctx->orderCur = 0;
        LaunchOrder::forget_order(ctx);
        ctx->framesSinceResize = 0;
        ctx->nextSortAt = 1;
    }
    if (ctx->framesSinceResize >= ctx->nextSortAt) {
        while (ctx->nextSortAt <= ctx->framesSinceResize) ctx->nextSortAt *= 2;
        const int target = ctx->orderValid ? 1 - ctx->orderCur : 0;
        const int ss = lane; /* a single frame's two halves: sorted on the main stream, the side stream waits */
        int rc = LaunchOrder::begin_sort(ctx, ss, target);
        if (rc) return rc;
        hipLaunchKernelGGL(rtk::rt_order_kernel, dim3(1), dim3(1024), 0, LaunchOrder::stream_of(ctx, ss), ctx->dTileCost, ctx->dTileKey, ctx->dTileOrder[target], tiles);
        HIP_TRY(ctx, hipGetLastError());
        if ((rc = LaunchOrder::end_sort(ctx, ss, twoOwn, ctx->orderValid ? ctx->orderCur : -1))) return rc;
        ctx->orderCur = target;
        ctx->orderValid = true;
    }
    a.tileCost = ctx->dTileCost;
    a.tileOrder = ctx->orderValid ? ctx->dTileOrder[ctx->orderCur] : nullptr;
    ctx->framesSinceResize += nFrames;
    return RT_OK;
}

/* ---- prepare buffers: the staging slab of a fused launch ([frame][pixel] colours awaiting rt_accumulate_kernel), one per stream so that
 * consecutive fused launches can alternate between the streams.  Returns 1 when the launch cannot be staged on this lane (the caller
 * falls back), 0 when a.staging is set. */
static int prepare_staging(RtContext* ctx, KArgs& a, int nFrames, size_t nPix, int lane, bool twoOwn, bool* unavailable)
{
    *unavailable = false;
    const size_t need = (size_t)nFrames * nPix * 16;
    if (!ctx->stagingUnavailable && ctx->stagingBytes[lane] < need) {
        HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
        /* both slabs are made by the first fused launch (a warm-up launch then pays for both: an allocation of this size takes
         * milliseconds); sized for the largest batch this context forms; if memory is short, for this batch */
        for (int sl = 0; sl < (twoOwn && ctx->alternate ? 2 : 1); sl++) {
            const int b = sl == 0 ? lane : 1 - lane;
            if (ctx->stagingBytes[b] >= need) continue;
            hipFree(ctx->dStaging[b]);
            ctx->dStaging[b] = nullptr;
            ctx->stagingBytes[b] = 0;
            /* room for the largest batch this context will ever form, so that a growing cap never re-makes a slab in the middle of a
             * render: as many frames as fit RT_FUSE_SLAB_BYTES, between RT_FUSE_MIN and RT_FUSE_MAX (the cap is held to what the slabs hold) */
            /* (sized ONCE, at the first fused launch, on purpose: growing a slab later costs a synchronise + an allocation of this size in the
             * middle of a progressive render.  BVH scenes never batch more than RT_FUSE_MIN frames: a host that runs many contexts of such
             * scenes lowers the budget with RT_FUSE_SLAB_MB — ADVICE r5) */
            size_t slabBudget = RT_FUSE_SLAB_BYTES;
            if (const char* e = getenv("RT_FUSE_SLAB_MB")) { const long long mb = atoll(e); if (mb > 0) slabBudget = (size_t)mb << 20; }
            size_t capFrames = nPix ? slabBudget / (nPix * 16) : RT_FUSE_MAX;
            capFrames = capFrames < RT_FUSE_MIN ? RT_FUSE_MIN : capFrames > RT_FUSE_MAX ? RT_FUSE_MAX : capFrames;
            if (capFrames < (size_t)nFrames) capFrames = (size_t)nFrames;
            const size_t cap = capFrames * nPix * 16;
            size_t got = cap;
            if (hipMalloc(&ctx->dStaging[b], cap) != hipSuccess) {
                (void)hipGetLastError();
                ctx->dStaging[b] = nullptr;
                got = need;
                if (hipMalloc(&ctx->dStaging[b], need) != hipSuccess) {
                    (void)hipGetLastError();
                    ctx->dStaging[b] = nullptr;
                    got = 0;
                }
            }
            ctx->stagingBytes[b] = got;
            if (got < need && b != lane) ctx->alternate = false; /* no room for the second slab: fused launches stay on one stream (until rt_resize) */
        }
    }
    if (ctx->stagingUnavailable || ctx->stagingBytes[lane] < need) {
        *unavailable = true;
        return RT_OK;
    }
    a.staging = ctx->dStaging[lane];
    a.stagingStride = (uint32_t)nPix;
    return RT_OK;
}

/* ---- launch tuner (suspension threshold 3/8 or 4/8) and the frames-per-launch budget: collect what has passed, decide for this launch */
static RtContext::Tuner::Probe* tune_launch(RtContext* ctx, KArgs& a, bool staged, int nFrames)
{
    RtContext::Tuner& tn = ctx->tuner;
    RtContext::Tuner::Probe* probe = nullptr;
    a.suspendNum = tn.decided;
    if (!tn.done && !ctx->flatScene) {
        for (auto& pr : tn.probe)
            if (pr.live && hipEventQuery(pr.stop) == hipSuccess) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, pr.start, pr.stop) == hipSuccess && pr.frames > 0) {
                    tn.ms[pr.cand] += ms / pr.frames;
                    tn.n[pr.cand]++;
                }
                pr.live = false;
            }
        if (tn.n[0] >= 3 && tn.n[1] >= 3) {
            tn.decided = (tn.ms[1] / tn.n[1] < 0.99 * (tn.ms[0] / tn.n[0])) ? 4 : 3; /* 4/8 has to win by more than the noise */
            tn.done = true;
            a.suspendNum = tn.decided;
            if (ctx->verbose) fprintf(stderr, "[raytrace_hip] launch tuner: 3/8 %.3f ms/frame, 4/8 %.3f ms/frame -> %d/8\n", tn.ms[0] / tn.n[0], tn.ms[1] / tn.n[1], tn.decided);
        } else if (staged && nFrames >= 8 && !ctx->stats && ctx->framesSinceResize >= 48) { /* only long progressive renders are tuned: short runs keep 3/8 */
            for (auto& pr : tn.probe)
                if (!pr.live) { probe = &pr; break; }
            if (probe) {
                if (!probe->start) { hipEventCreate(&probe->start); hipEventCreate(&probe->stop); }
                probe->cand = tn.next;
                probe->frames = nFrames;
                tn.next ^= 1;
                a.suspendNum = probe->cand ? 4 : 3;
            }
        }
    }
    /* frames per fused launch (RT_FUSE_MIN ...): the cap the NEXT batches are cut to */
    if (!ctx->fuseCapPinned)
        for (auto& fp : ctx->fuseProbe) {
            if (!fp.live || hipEventQuery(fp.stop) != hipSuccess) continue;
            fp.live = false;
            if (fp.frames <= 0) continue;
            float ms = 0;
            /* the previous fused launch's end to this one's; no predecessor (the first fused launch, a chain broken by full probes): its own
             * start to its end (includes what it waited for the other stream's kernels: over-estimates, i.e. errs towards shorter batches).
             * (a slot re-recorded by a later launch gives a negative or failing difference: skipped) */
            hipEvent_t from = fp.prev >= 0 ? ctx->fuseProbe[fp.prev].stop : fp.start;
            if (!from || hipEventElapsedTime(&ms, from, fp.stop) != hipSuccess || !(ms > 0)) { (void)hipGetLastError(); continue; }
            const double perFrame = (double)ms / fp.frames;
            int cap = (int)ceil(RT_FUSE_TARGET_MS / perFrame);
            cap = cap < RT_FUSE_MIN ? RT_FUSE_MIN : cap > RT_FUSE_MAX ? RT_FUSE_MAX : cap;
            {   /* never more than the staging slabs hold (they are sized once, prepare_staging) */
                const size_t nPixNow = (size_t)ctx->localRows * ctx->W;
                size_t slab = 0;
                for (int b = 0; b < 2; b++)
                    if (ctx->stagingBytes[b] && (slab == 0 || ctx->stagingBytes[b] < slab)) slab = ctx->stagingBytes[b];
                if (nPixNow && slab) {
                    const size_t holds = slab / (nPixNow * 16);
                    if ((size_t)cap > holds) cap = holds < RT_FUSE_MIN ? RT_FUSE_MIN : (int)holds;
                }
            }
            if (cap != ctx->fuseCap && ctx->verbose) fprintf(stderr, "[raytrace_hip] fused launches: %.3f ms per frame -> up to %d frames per launch\n", perFrame, cap);
            ctx->fuseCap = cap;
        }
    return probe;
}
/* around the trace kernel of a fused launch: returns the probe slot (or -1) */
static int mark_fused_launch_begin(RtContext* ctx, hipStream_t st)
{
    if (ctx->fuseCapPinned) return -1;
    int slot = -1;
    for (int i = 0; i < (int)(sizeof(ctx->fuseProbe) / sizeof(ctx->fuseProbe[0])); i++)
        if (!ctx->fuseProbe[i].live && i != ctx->fuseLast) { slot = i; break; }
    if (slot < 0) { ctx->fuseLast = -1; return -1; } /* all in flight: the chain of stop events breaks here */
    RtContext::FuseProbe& fp = ctx->fuseProbe[slot];
    if (!fp.stop) { hipEventCreate(&fp.start); hipEventCreate(&fp.stop); }
    if (!fp.start || !fp.stop) return -1;
    if (ctx->fuseLast < 0 && hipEventRecord(fp.start, st) != hipSuccess) return -1; /* only a launch without a predecessor needs its own start */
    return slot;
}
static void mark_fused_launch_end(RtContext* ctx, int slot, hipStream_t st, int nFrames)
{
    if (slot < 0) return;
    RtContext::FuseProbe& fp = ctx->fuseProbe[slot];
    if (hipEventRecord(fp.stop, st) == hipSuccess) {
        fp.frames = nFrames;
        fp.prev = ctx->fuseLast; /* its event stays recorded (a slot is reused only when it is not the last one) */
        fp.live = true;
        ctx->fuseLast = slot;
    }
}

/* ---- enqueue: the trace kernel(s) of one launch.
 * Persistent launches.  Queue position q of part p is entry q*parts + p of the (longest chain first) tile order, so the parts are
 * disjoint and equally heavy.  Each kernel is given as many single-wave workgroups as the chip keeps resident, never more than it has
 * work items: its first `grid` positions are taken by blockIdx, the rest through the part's atomic queue, which counts monotonically
 * across launches (this launch's positions start at tileQueueBase).
 * Two parts on two streams for a single frame while the context runs on its own stream: a kernel ends with a drain phase as long as one
 * pixel's serial chain, during which waves retire one by one; the other stream's kernel — different pixels, no dependence — picks up
 * every slot that frees.  With a caller-provided stream the caller's stream order is the contract: one kernel on that stream. */
static int enqueue_trace(RtContext* ctx, KArgs& a, const LaunchPlan& plan, int tiles, int nFrames, int lane, int parts, RtContext::Tuner::Probe* probe)
{
    const bool staged = nFrames > 1;
    for (int p = 0; p < parts; p++) {
        const int partTiles = (tiles - p + parts - 1) / parts;
        /* frames per item: 1 = the most items and the shortest tail.  The FLAT scenes' per-frame chains are short and uniform; there one
         * pixel set-up per group of frames is worth 7 % (config 2: 0.765 -> 0.710 ms/frame at groups of 4) as long as every resident wave
         * still gets >= 8 items; fewer items than that, or groups of 8+, lose it to the tail again, and the BVH scenes gain nothing
         * measurable (profiles/r02_frame_group_sweep.txt; their kernel variants are compiled without groups). */
        int group = 1;
        if (staged && ctx->flatScene && ctx->params.numRaysPerPixel < 65536) {
            if (ctx->frameGroupOverride > 0) group = ctx->frameGroupOverride;
            else
                while (group < 4 && 2 * group <= nFrames && (long long)partTiles * ((nFrames + 2 * group - 1) / (2 * group)) >= 8 * plan.resident * plan.wavesPerGroup) group *= 2;
            if (group > nFrames) group = nFrames;
        }
        a.frameGroup = group;
        a.frameGroupShift = 0;
        while ((2 << a.frameGroupShift) <= group) a.frameGroupShift++;
        a.frameGroups = staged ? (nFrames + group - 1) / group : 1;
        const long long items = (long long)partTiles * a.frameGroups;
        /* the grid: workgroups of wavesPerGroup persistent waves each — as many as the chip keeps resident, never more waves than there
         * are items (rounded up to whole workgroups: a wave without an item of its own goes to the queue, finds it empty and ends) */
        const int wpb = plan.wavesPerGroup;
        long long wantWaves = plan.resident * wpb < items ? plan.resident * wpb : items;
        if (ctx->gridOverride > 0) wantWaves = ctx->gridOverride < items ? ctx->gridOverride : items;
        const int grid = (int)((wantWaves + wpb - 1) / wpb);
        const unsigned long long gridWaves = (unsigned long long)grid * wpb;
        const unsigned long long byIndex = gridWaves < (unsigned long long)items ? gridWaves : (unsigned long long)items; /* positions taken by wave index */
        a.launchTiles = partTiles;
        a.launchItems = (int)items;
        a.orderOffset = p;
        a.orderStride = parts;
        /* One kernel alone: all its workgroups become resident at once, the first `grid` positions go by blockIdx and only the rest
         * through the queue.  Two kernels sharing the chip: workgroups are dispatched as slots free up, possibly late, so every
         * position — the longest chains first — comes from the queue. */
        a.queueStart = parts == 2 ? 1 : 0;
        if (ctx->verbose) fprintf(stderr, "[raytrace_hip] launch variant=%d part=%d/%d tiles=%d grid=%d x %d waves perCU=%d lds=%zu\n", plan.variant, p, parts, partTiles, grid, wpb, ctx->occPerCU[plan.variant], plan.ldsBytes);
        /* stream, pixel-record slot and tile-queue counter of this kernel: part p of a two-part frame, or the fused launch's lane */
        const int q = parts == 2 ? p : lane;
        hipStream_t st = LaunchOrder::stream_of(ctx, q);
        a.pxCold = (float4*)((char*)ctx->dPxCold + (size_t)q * ctx->pxColdWaves * RT_COLD_STRIDE_BYTES);
        a.tileQueue = ctx->dTileQueue + q;
        a.tileQueueBase = ctx->tileQueueNext[q] - (a.queueStart ? 0ull : byIndex);
        int rc;
        if ((rc = LaunchOrder::before_order_read(ctx, q))) return rc;
        if (!staged && (rc = LaunchOrder::before_acc_write(ctx, q, parts == 1))) return rc; /* the trace kernel itself adds into the accumulation buffer (RCC:20-23) */
        const int fuseSlot = staged ? mark_fused_launch_begin(ctx, st) : -1;
        if (probe) hipEventRecord(probe->start, st);
        hipLaunchKernelGGL(parts == 2 ? plan.kernHalf : plan.kern, dim3(grid), dim3(plan.blockThreads), plan.ldsBytes, st, a);
        if (probe) { hipEventRecord(probe->stop, st); probe->live = true; }
        HIP_TRY(ctx, hipGetLastError()); /* a refused launch ran no wave: the device counter did not move */
        mark_fused_launch_end(ctx, fuseSlot, st, nFrames);
        /* every position not taken by wave index is one successful fetch, and each of the grid's waves overshoots once */
        ctx->tileQueueNext[q] += (unsigned long long)items - (a.queueStart ? 0ull : byIndex) + gridWaves;
        if (q == 1) LaunchOrder::side_used(ctx);
        if (!staged && (rc = LaunchOrder::after_acc_write(ctx, q, parts == 1))) return rc;
    }
    if (!staged && parts == 1) LaunchOrder::other_stream_settled(ctx, 0); /* ordered before the main stream's kernel just launched */
    return RT_OK;
}

/* ---- enqueue: RCC:18-23 for a fused launch's frames, in frame order, after every earlier frame's */
static int enqueue_accumulate(RtContext* ctx, const KArgs& a, int nFrames, size_t nPix, int lane)
{
    int blocks = (int)((nPix + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    int rc = LaunchOrder::before_acc_write(ctx, lane, true);
    if (rc) return rc;
    hipLaunchKernelGGL(rtk::rt_accumulate_kernel, dim3(blocks), dim3(256), 0, LaunchOrder::stream_of(ctx, lane), (const float4*)ctx->dStaging[lane], nFrames, nPix,
                       (float4*)a.accumulated, (float4*)a.frameRender, nPix);
    HIP_TRY(ctx, hipGetLastError());
    if ((rc = LaunchOrder::after_acc_write(ctx, lane, true))) return rc;
    LaunchOrder::other_stream_settled(ctx, lane); /* this accumulate is ordered after it; later writers wait for this one */
    return RT_OK;
}

static int launch_frames(RtContext* ctx, int frame0, int nFrames)
{
    KArgs a;
    fill_args(ctx, frame0, nFrames, a);
    const int tiles = a.tilesX * a.tilesY;
    if (tiles == 0) return RT_OK;
    LaunchPlan plan;
    bool many = false;
    int rc = choose_variant(ctx, a, plan, &many);
    if (rc) return rc;
    if ((rc = prepare_records(ctx, plan))) return rc;
    /* Fused launches (several frames: (tile, frame) items, per-frame colours staged and summed in frame order afterwards) alternate
     * between the context's two streams (own streams only), each with its own staging slab: the trace kernel of launch k+1 — other
     * frames, nothing shared but the scene — starts while launch k drains, and only the rt_accumulate_kernels, which add into the
     * accumulation buffer in FRAME order, are chained by events. */
    const bool staged = nFrames > 1;
    const size_t nPix = (size_t)ctx->localRows * ctx->W;
    /* Two kernels sharing the chip refill each other's freed wave slots one by one — with single-wave workgroups.  A 12-wave workgroup frees
     * its slots and its LDS only when its LAST wave ends, and the other kernel's workgroups need twelve slots at once: overlapping launches then
     * wait for each other instead of filling gaps (config 4 / 5: 7.8 / 50.2 ms per frame for K back-to-back rt_render_frame against 6.2 / 42.1 on
     * one stream, profiles/r06_groups_and_streams.txt).  The BVH variants' group launches therefore stay on ONE stream, one kernel per launch;
     * RT_GROUP_STREAMS=2 restores the overlap for A/B runs. */
    static const bool groupsOverlap = getenv("RT_GROUP_STREAMS") && atoi(getenv("RT_GROUP_STREAMS")) == 2;
    const bool twoOwn = ctx->twoStreams && ctx->stream == ctx->ownStream && ctx->sideStream && (plan.wavesPerGroup == 1 || groupsOverlap);
    const int lane = (staged && twoOwn && ctx->alternate) ? ctx->stagedNext : 0;
    if ((rc = prepare_tile_order(ctx, a, tiles, nFrames, lane, twoOwn))) return rc;
    if (staged) {
        bool unavailable = false;
        if ((rc = prepare_staging(ctx, a, nFrames, nPix, lane, twoOwn, &unavailable))) return rc;
        if (unavailable) {
            if (ctx->lptEnabled) ctx->framesSinceResize -= nFrames; /* counted again by the launches below */
            if (lane == 1) { /* no room for the second slab: every fused launch on the main stream from now on */
                ctx->alternate = false;
                ctx->stagedNext = 0;
                return launch_frames(ctx, frame0, nFrames);
            }
            /* not even one slab: the frames go out one launch each (a single-frame launch needs no staging) — a render call never
             * fails for want of scratch — and the allocation is not tried again until the image is resized */
            ctx->stagingUnavailable = true;
            for (int f = 0; f < nFrames; f++)
                if ((rc = launch_frames(ctx, frame0 + f, 1))) return rc; /* each counts itself in ctx->lastLaunched */
            return RT_OK;
        }
    }
    RtContext::Tuner::Probe* probe = tune_launch(ctx, a, staged, nFrames);
    const int parts = (!staged && twoOwn && tiles >= 2) ? 2 : 1;
    if ((parts == 2 || lane == 1) && (rc = LaunchOrder::fork_side(ctx))) return rc; /* the side stream follows what the main stream holds so far */
    if ((rc = enqueue_trace(ctx, a, plan, tiles, nFrames, lane, parts, probe))) return rc;
    if (staged) {
        if ((rc = enqueue_accumulate(ctx, a, nFrames, nPix, lane))) return rc;
        if (twoOwn && ctx->alternate) ctx->stagedNext = 1 - lane;
    } else if (parts == 2 && ctx->fusedBehindFirstPart) {
        /* a fused launch that follows a two-part frame goes behind the part that started first and ends first (the main stream's), so
         * its waves take the slots the other part's drain leaves empty instead of waiting for that drain to end */
        ctx->stagedNext = 0;
    }
    ctx->pixelFrames += (uint64_t)ctx->localRows * ctx->W * nFrames;
    ctx->lastLaunched += nFrames;
    return RT_OK;
}

extern "C" {

static int check_renderable(RtContext* ctx)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    if (ctx->W == 0) return fail(ctx, RT_ERR_STATE, "render before rt_resize");
    if (!ctx->haveScene) return fail(ctx, RT_ERR_STATE, "render before rt_upload_scene");
    if (!ctx->haveParams) return fail(ctx, RT_ERR_STATE, "render before rt_set_params");
    if (ctx->params.numRaysPerPixel < 0) return fail(ctx, RT_ERR_INVALID_ARG, "numRaysPerPixel < 0");
    return RT_OK;
}

static bool gpu_idle(RtContext* ctx)
{
    if (hipStreamQuery(ctx->stream) != hipSuccess) return false;
    if (ctx->sideStream && ctx->ord.sideDirty && hipStreamQuery(ctx->sideStream) != hipSuccess) return false;
    return true;
}

int rt_render_frame(RtContext* ctx)
{
    int rc = check_renderable(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    /* Frames requested while earlier ones still execute are held back and leave as one fused launch (each pixel
     * runs its frames back to back: same bits as one launch per frame, without a chip-wide drain per frame) when
     * fuseCap have gathered or at the next call that needs them.  An idle GPU starts at once. */
    if (ctx->coalesce && ctx->fuseFrames && ctx->params.accumulate && ctx->stream == ctx->ownStream) {
        if (gpu_idle(ctx)) {
            RT_FLUSH(ctx);
            rc = launch_frames(ctx, ctx->frame, 1);
            if (rc) return rc;
            ctx->frame++;
            return RT_OK;
        }
        ctx->pending++;
        ctx->frame++; /* RCM:94 */
        if (ctx->pending >= ctx->fuseCap) RT_FLUSH(ctx);
        return RT_OK;
    }
    RT_FLUSH(ctx);
    rc = launch_frames(ctx, ctx->frame, 1);
    if (rc) return rc;
    if (ctx->params.accumulate) ctx->frame++; /* RCM:94 */
    return RT_OK;
}

/* Launch what rt_render_frame holds back, without waiting for it. */
int rt_flush(RtContext* ctx)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    return RT_OK;
}

int rt_render_frames(RtContext* ctx, int n)
{
    int rc = check_renderable(ctx);
    if (rc) return rc;
    if (n < 0) return fail(ctx, RT_ERR_INVALID_ARG, "rt_render_frames: n < 0");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    RT_FLUSH(ctx);
    if (ctx->fuseFrames && ctx->params.accumulate) {
        /* Batched form: each pixel runs its frames back to back inside one launch — same Frame
         * seeds, same order of additions into the sum, FrameRender = the last frame — so the
         * chip-wide drain at the end of a launch is paid once per batch instead of once per
         * frame.  Batches are capped to keep launches short. */
        while (n > 0) {
            const int k = n < ctx->fuseCap ? n : ctx->fuseCap;
            rc = launch_frames(ctx, ctx->frame, k);
            if (rc) return rc;
            ctx->frame += k;
            n -= k;
        }
        return RT_OK;
    }
    for (int i = 0; i < n; i++) {
        rc = launch_frames(ctx, ctx->frame, 1);
        if (rc) return rc;
        if (ctx->params.accumulate) ctx->frame++;
    }
    return RT_OK;
}

int rt_synchronize(RtContext* ctx)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
    flush_timer(ctx);
    return RT_OK;
}

int rt_get_frame(const RtContext* ctx) { return ctx ? ctx->frame : RT_ERR_INVALID_ARG; }

int rt_timer_begin(RtContext* ctx)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    flush_timer(ctx);
    if (ctx->timerState == 1) return fail(ctx, RT_ERR_STATE, "rt_timer_begin: timer already running");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(ctx->evStart, joined(ctx)));
    ctx->timerState = 1;
    return RT_OK;
}
int rt_timer_end(RtContext* ctx)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    if (ctx->timerState != 1) return fail(ctx, RT_ERR_STATE, "rt_timer_end without rt_timer_begin");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(ctx->evStop, joined(ctx)));
    ctx->timerState = 2;
    return RT_OK;
}

static int read_target(RtContext* ctx, const float* src, float* rgba, size_t bytes)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    size_t want = (size_t)ctx->localRows * ctx->W * 16;
    if (!rgba || bytes != want) return fail(ctx, RT_ERR_INVALID_ARG, "read: need exactly %zu bytes, got %zu", want, bytes);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
    flush_timer(ctx);
    if (bytes) HIP_TRY(ctx, hipMemcpy(rgba, src, bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}
int rt_read_frame(RtContext* ctx, float* rgba, size_t bytes)
{
    return read_target(ctx, ctx ? (ctx->boundFrame ? ctx->boundFrame : ctx->ownFrame) : nullptr, rgba, bytes);
}
int rt_read_accumulated(RtContext* ctx, float* rgba, size_t bytes)
{
    return read_target(ctx, ctx ? (ctx->boundAccum ? ctx->boundAccum : ctx->ownAccum) : nullptr, rgba, bytes);
}

/* ---- display pass + checkpoint ------------------------------------------------ */
static int display_common(RtContext* ctx, int frame, int use_accumulated, const float** src)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    if (ctx->W == 0) return fail(ctx, RT_ERR_STATE, "display before rt_resize");
    if (frame == 0) return fail(ctx, RT_ERR_INVALID_ARG, "display: Frame must not be 0");
    *src = use_accumulated ? (ctx->boundAccum ? ctx->boundAccum : ctx->ownAccum) : (ctx->boundFrame ? ctx->boundFrame : ctx->ownFrame);
    return RT_OK;
}

static int display_scratch(RtContext* ctx, size_t bytes, void** out)
{
    if (ctx->displayBytes < bytes) {
        HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
        hipFree(ctx->dDisplay);
        ctx->dDisplay = nullptr;
        ctx->displayBytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->dDisplay, bytes));
        ctx->displayBytes = bytes;
    }
    *out = ctx->dDisplay;
    return RT_OK;
}

int rt_display(RtContext* ctx, int frame, int use_accumulated, float* rgba, size_t bytes)
{
    const float* src = nullptr;
    int rc = display_common(ctx, frame, use_accumulated, &src);
    if (rc) return rc;
    const size_t n = (size_t)ctx->localRows * ctx->W;
    if (!rgba || bytes != n * 16) return fail(ctx, RT_ERR_INVALID_ARG, "rt_display: need exactly %zu bytes", n * 16);
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float4* tmp = nullptr;
    if ((rc = display_scratch(ctx, bytes, (void**)&tmp))) return rc;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rtk::rt_display_kernel, dim3(blocks), dim3(256), 0, joined(ctx), (const float4*)src, tmp, n, frame);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(joined(ctx));
    if (e == hipSuccess) e = hipMemcpy(rgba, tmp, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, "rt_display: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_display_srgb8(RtContext* ctx, int frame, int use_accumulated, int flip_y, uint8_t* rgba8, size_t bytes)
{
    const float* src = nullptr;
    int rc = display_common(ctx, frame, use_accumulated, &src);
    if (rc) return rc;
    const size_t n = (size_t)ctx->localRows * ctx->W;
    if (!rgba8 || bytes != n * 4) return fail(ctx, RT_ERR_INVALID_ARG, "rt_display_srgb8: need exactly %zu bytes", n * 4);
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t* tmp = nullptr;
    if ((rc = display_scratch(ctx, bytes, (void**)&tmp))) return rc;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rtk::rt_display_srgb8_kernel, dim3(blocks), dim3(256), 0, joined(ctx), (const float4*)src, tmp, ctx->W, ctx->localRows, frame, flip_y);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(joined(ctx));
    if (e == hipSuccess) e = hipMemcpy(rgba8, tmp, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, "rt_display_srgb8: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_write_accumulated(RtContext* ctx, const float* rgba, size_t bytes)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    const size_t want = (size_t)ctx->localRows * ctx->W * 16;
    if (!rgba || bytes != want) return fail(ctx, RT_ERR_INVALID_ARG, "rt_write_accumulated: need exactly %zu bytes, got %zu", want, bytes);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
    if (bytes) HIP_TRY(ctx, hipMemcpy(ctx->boundAccum ? ctx->boundAccum : ctx->ownAccum, rgba, bytes, hipMemcpyHostToDevice));
    return RT_OK;
}

int rt_enable_stats(RtContext* ctx, int enabled)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    ctx->stats = enabled != 0;
    return RT_OK;
}

int rt_reset_counters(RtContext* ctx)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
    flush_timer(ctx);
    /* stream-ordered: the next launch on this stream starts after the fill (a null-stream hipMemset is not ordered
     * against the non-blocking render streams and may land after the next kernel's first waves have counted) */
    HIP_TRY(ctx, hipMemsetAsync(ctx->dCounters, 0, sizeof(unsigned long long) * RT_COUNTER_SLOTS * RT_COUNTER_FIELDS, joined(ctx)));
    ctx->pixelFrames = 0;
    ctx->gpuMs = 0;
    return RT_OK;
}

int rt_get_counters(RtContext* ctx, RtCounters* out)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    if (!out) return fail(ctx, RT_ERR_INVALID_ARG, "rt_get_counters: null out");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
    flush_timer(ctx);
    std::vector<unsigned long long> h((size_t)RT_COUNTER_SLOTS * RT_COUNTER_FIELDS);
    HIP_TRY(ctx, hipMemcpy(h.data(), ctx->dCounters, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long sum[RT_COUNTER_FIELDS] = {0};
    for (int s = 0; s < RT_COUNTER_SLOTS; s++)
        for (int f = 0; f < RT_COUNTER_FIELDS; f++) sum[f] += h[(size_t)s * RT_COUNTER_FIELDS + f];
    out->segments = sum[0];
    out->innerSteps = sum[1];
    out->leafSteps = sum[2];
    out->triTests = sum[3];
    out->sphereTests = sum[4];
    out->modelVisits = sum[5];
    out->pixelFrames = ctx->pixelFrames;
    out->gpuMs = ctx->gpuMs;
    /* slot 7 = the watchdogs (rt_kernels.h): traverse — a wave ended its lanes' walks because no validated scene needs that many steps; pool_exchange — a wave
     * gave up waiting for a cell of the FLAT variant's chain pool */
    if (h[7]) return fail(ctx, RT_ERR_HIP, "a kernel watchdog fired %llu times: a traversal did not end (scene validation has a hole, or device memory is corrupt) or, in a scene without trees, a cell of the chain pool was never handed over; the images since the last rt_reset_counters are not valid", h[7]);
    return RT_OK;
}

/* ---- test hooks ------------------------------------------------------------ */
/* Wave-level phase profile of the stats launches since rt_reset_counters:
 * out[2*p] = times a wave executed phase p, out[2*p+1] = lanes active in it
 * (p: 0 loop, 1 raygen, 2 spheres, 3 traverse call, 4 model setup, 5 inner step,
 *  6 triangle test, 7 shade hit, 8 sky, 9 sphere roots, 10 glass branch, 11 pixel refill). */
int rt_debug_phase_profile(RtContext* ctx, uint64_t* out, int n)
{
    if (!ctx || !out || n < 2 * RT_N_PHASES) return fail(ctx, RT_ERR_INVALID_ARG, "rt_debug_phase_profile: need %d entries", 2 * RT_N_PHASES);
    RT_FLUSH(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
    std::vector<unsigned long long> h((size_t)RT_COUNTER_SLOTS * RT_COUNTER_FIELDS);
    HIP_TRY(ctx, hipMemcpy(h.data(), ctx->dCounters, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int f = 0; f < 2 * RT_N_PHASES; f++) {
        out[f] = 0;
        for (int s = 0; s < RT_COUNTER_SLOTS; s++) out[f] += h[(size_t)s * RT_COUNTER_FIELDS + 8 + f];
    }
    if (n > 2 * RT_N_PHASES) { /* audit of the conservative root filter: must be 0 */
        out[2 * RT_N_PHASES] = 0;
        for (int s = 0; s < RT_COUNTER_SLOTS; s++) out[2 * RT_N_PHASES] += h[(size_t)s * RT_COUNTER_FIELDS + 6];
    }
    /* then: inner steps served by the LDS top-of-tree cache; inner steps (lane-steps) taken while >= 48 lanes of the wave stood on ONE node;
     * the same for >= 3/4 of >= 16 active lanes; the rest 0 */
    for (int p = 0; p < RT_N_PHASES && 2 * RT_N_PHASES + 1 + p < n; p++) {
        out[2 * RT_N_PHASES + 1 + p] = 0;
        if (p < 3)
            for (int s = 0; s < RT_COUNTER_SLOTS; s++) out[2 * RT_N_PHASES + 1 + p] += h[(size_t)s * RT_COUNTER_FIELDS + 8 + 2 * RT_N_PHASES + p];
    }
    return RT_OK;
}

/* device scratch that is released on every exit path of the two hooks below */
struct DevScratch {
    void* p = nullptr;
    ~DevScratch() { if (p) hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
    float* f() const { return (float*)p; }
};

int rt_debug_fused_frames_cap(const RtContext* ctx) { return ctx ? ctx->fuseCap : RT_ERR_INVALID_ARG; }

int rt_debug_intersect(RtContext* ctx, const float* origins, const float* dirs, int n, float* out10)
{
    if (ctx) RT_FLUSH(ctx);
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    if (!ctx->haveScene) return fail(ctx, RT_ERR_STATE, "rt_debug_intersect before rt_upload_scene");
    if (n < 0 || (n && (!origins || !dirs || !out10))) return fail(ctx, RT_ERR_INVALID_ARG, "rt_debug_intersect: bad arguments");
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevScratch dO, dD, dR;
    HIP_TRY(ctx, dO.alloc((size_t)n * 12));
    HIP_TRY(ctx, dD.alloc((size_t)n * 12));
    HIP_TRY(ctx, dR.alloc((size_t)n * 40));
    HIP_TRY(ctx, hipMemcpy(dO.p, origins, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dD.p, dirs, (size_t)n * 12, hipMemcpyHostToDevice));
    KArgs a;
    fill_args(ctx, 1, 1, a);
    hipLaunchKernelGGL(rtk::rt_debug_intersect_kernel, dim3((n + RT_WAVE - 1) / RT_WAVE), dim3(RT_WAVE), 0, joined(ctx), a, dO.f(), dD.f(), n, dR.f());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
    HIP_TRY(ctx, hipMemcpy(out10, dR.p, (size_t)n * 40, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_debug_math_eval(RtContext* ctx, int op, const float* x, const float* y, float* out, int n)
{
    if (ctx) RT_FLUSH(ctx);
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    if (n < 0 || (n && (!x || !y || !out))) return fail(ctx, RT_ERR_INVALID_ARG, "rt_debug_math_eval: bad arguments");
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevScratch dX, dY, dR;
    HIP_TRY(ctx, dX.alloc((size_t)n * 4));
    HIP_TRY(ctx, dY.alloc((size_t)n * 4));
    HIP_TRY(ctx, dR.alloc((size_t)n * 4));
    HIP_TRY(ctx, hipMemcpy(dX.p, x, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dY.p, y, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rtk::rt_debug_math_kernel, dim3((n + 255) / 256), dim3(256), 0, joined(ctx), op, dX.f(), dY.f(), dR.f(), n);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
    HIP_TRY(ctx, hipMemcpy(out, dR.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}


/* ------------------------------------------------------------------------------------------
 * Several GPUs from one host process: n contexts with cyclic 8-row strips, gather at readback.
 * ---------------------------------------------------------------------------------------- */
struct RtMulti {
    std::vector<RtContext*> ctx;
    /* pinned staging of the gather: every context's packed rows, copied device -> host concurrently */
    void* pinned = nullptr;
    size_t pinnedBytes = 0;
    double lastGatherMs = 0;
    /* peer access between the distinct devices of this multi-context (hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess at
     * creation): peerPairs = ordered pairs of distinct devices, peerEnabled = of those, the pairs with direct access (xGMI or
     * PCIe P2P).  Without it hipMemcpyPeerAsync still works — the runtime stages through host memory — only slower. */
    int peerPairs = 0, peerEnabled = 0;
};

/* launch what every context holds back BEFORE waiting for any of them: a per-context synchronise in a loop would start
 * device i+1's held frames only after device i has finished */
static int multi_flush_all(RtMulti* m)
{
    for (RtContext* c : m->ctx) {
        int rc = rt_flush(c);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}

/* wait for whatever the contexts' streams hold (error paths of the gathers: nothing may still be writing into the caller's
 * buffer when an error is returned) */
static void multi_drain(RtMulti* m)
{
    for (RtContext* c : m->ctx) {
        if (hipSetDevice(c->device) != hipSuccess) { (void)hipGetLastError(); continue; }
        (void)hipStreamSynchronize(joined(c));
        (void)hipGetLastError();
    }
}

int rt_create_multi(const int* device_ids, int n_devices, RtMulti** out)
{
    if (!out) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create_multi: out is null");
    *out = nullptr;
    if (!device_ids || n_devices <= 0) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create_multi: no devices");
    RtMulti* m = new RtMulti();
    for (int i = 0; i < n_devices; i++) {
        RtContext* c = nullptr;
        int rc = rt_create(device_ids[i], &c);
        if (rc == RT_OK) rc = rt_set_partition(c, 8, i, n_devices);
        if (rc != RT_OK) {
            if (c) rt_destroy(c);
            rt_destroy_multi(m);
            return rc;
        }
        m->ctx.push_back(c);
    }
    /* direct device-to-device copies (scene replication at upload, rt_gather_*_to_device) need peer access enabled in both
     * directions; "already enabled" is fine, "cannot" leaves the pair on the runtime's host-staged path */
    for (int i = 0; i < n_devices; i++)
        for (int j = 0; j < n_devices; j++) {
            const int di = m->ctx[i]->device, dj = m->ctx[j]->device;
            if (di == dj) continue;
            bool seen = false; /* the same pair listed twice (virtual shards on one device list) counts once */
            for (int i2 = 0; i2 < n_devices && !seen; i2++)
                for (int j2 = 0; j2 < n_devices && !seen; j2++)
                    if ((i2 < i || (i2 == i && j2 < j)) && m->ctx[i2]->device == di && m->ctx[j2]->device == dj) seen = true;
            if (seen) continue;
            m->peerPairs++;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, di, dj) != hipSuccess) { (void)hipGetLastError(); can = 0; }
            if (!can) continue;
            if (hipSetDevice(di) != hipSuccess) { (void)hipGetLastError(); continue; }
            const hipError_t e = hipDeviceEnablePeerAccess(dj, 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) m->peerEnabled++;
            (void)hipGetLastError();
        }
    *out = m;
    return RT_OK;
}

/* How device-to-device copies between this multi-context's GPUs travel: 2 = every pair of distinct devices has direct peer access
 * (xGMI / PCIe P2P), 1 = some pairs, 0 = none (the runtime stages through host memory), -1 = one device only (nothing to copy
 * between devices).  *pairs / *enabled (optional) receive the counts. */
int rt_multi_peer_access(const RtMulti* m, int* pairs, int* enabled)
{
    if (!m) return RT_ERR_INVALID_ARG;
    if (pairs) *pairs = m->peerPairs;
    if (enabled) *enabled = m->peerEnabled;
    if (m->peerPairs == 0) return -1;
    return m->peerEnabled == m->peerPairs ? 2 : (m->peerEnabled > 0 ? 1 : 0);
}

void rt_destroy_multi(RtMulti* m)
{
    if (!m) return;
    for (RtContext* c : m->ctx) rt_destroy(c);
    if (m->pinned) hipHostFree(m->pinned);
    delete m;
}

int rt_multi_count(const RtMulti* m) { return m ? (int)m->ctx.size() : 0; }
RtContext* rt_multi_context(RtMulti* m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr; }

#define RT_MULTI_FORWARD(call)                                                 \
    do {                                                                       \
        if (!m) return fail(nullptr, RT_ERR_INVALID_ARG, "null multi context"); \
        for (RtContext* c : m->ctx) {                                          \
            int rc_ = (call);                                                  \
            if (rc_ != RT_OK) return rc_;                                      \
        }                                                                      \
        return RT_OK;                                                          \
    } while (0)

int rt_multi_resize(RtMulti* m, int width, int height) { RT_MULTI_FORWARD(rt_resize(c, width, height)); }
int rt_multi_upload_scene(RtMulti* m, const RtModel* models, int n_models, const RtTriangle* triangles, int n_triangles,
                          const RtBVHNode* nodes, int n_nodes, const RtSphere* spheres, int n_spheres)
{
    /* validated and re-laid out once; context 0 gets it from the host, the others copy context 0's device arrays
     * (hipMemcpyPeerAsync on their own streams: all destinations at once, over xGMI between different devices) */
    if (!m || m->ctx.empty()) return fail(nullptr, RT_ERR_INVALID_ARG, "null multi context");
    int rc = multi_flush_all(m);
    if (rc) return rc;
    PreparedScene ps;
    RtContext* c0 = m->ctx[0];
    if ((rc = prepare_scene(c0, models, n_models, triangles, n_triangles, nodes, n_nodes, spheres, n_spheres, ps))) return rc;
    if ((rc = commit_scene(c0, ps, nullptr))) return rc;
    const bool peerCopies = getenv("RT_MULTI_PEER_UPLOAD") == nullptr || atoi(getenv("RT_MULTI_PEER_UPLOAD")) != 0;
    for (size_t i = 1; i < m->ctx.size(); i++) {
        rc = commit_scene(m->ctx[i], ps, peerCopies ? c0 : nullptr);
        if (rc != RT_OK && peerCopies) rc = commit_scene(m->ctx[i], ps, nullptr); /* no peer path between the two: from the host */
        if (rc != RT_OK) return rc;
    }
    for (size_t i = 1; i < m->ctx.size(); i++) { /* the copies read context 0's arrays: done before anyone may replace them */
        RtContext* c = m->ctx[i];
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return RT_OK;
}
int rt_multi_update_models(RtMulti* m, const RtModel* models, int n_models) { RT_MULTI_FORWARD(rt_update_models(c, models, n_models)); }
int rt_multi_update_spheres(RtMulti* m, const RtSphere* spheres, int n_spheres) { RT_MULTI_FORWARD(rt_update_spheres(c, spheres, n_spheres)); }
int rt_multi_set_params(RtMulti* m, const RtParams* params) { RT_MULTI_FORWARD(rt_set_params(c, params)); }
int rt_multi_reset_accumulation(RtMulti* m) { RT_MULTI_FORWARD(rt_reset_accumulation(c)); }
int rt_multi_render_frame(RtMulti* m) { RT_MULTI_FORWARD(rt_render_frame(c)); }
int rt_multi_render_frames(RtMulti* m, int n) { RT_MULTI_FORWARD(rt_render_frames(c, n)); }
int rt_multi_synchronize(RtMulti* m)
{
    if (!m) return fail(nullptr, RT_ERR_INVALID_ARG, "null multi context");
    int rc = multi_flush_all(m);
    if (rc) return rc;
    RT_MULTI_FORWARD(rt_synchronize(c));
}
double rt_multi_last_gather_ms(const RtMulti* m) { return m ? m->lastGatherMs : 0.0; }

static int multi_gather(RtMulti* m, float* rgba, size_t bytes, bool accumulated)
{
    if (!m || m->ctx.empty()) return fail(nullptr, RT_ERR_INVALID_ARG, "null multi context");
    RtContext* c0 = m->ctx[0];
    const int W = c0->W, H = c0->H;
    if (!rgba || bytes != (size_t)W * H * 16) return fail(c0, RT_ERR_INVALID_ARG, "rt_gather: bytes %zu != H*W*16 = %zu", bytes, (size_t)W * H * 16);
    const size_t rowBytes = (size_t)W * 16;
    size_t total = 0;
    for (RtContext* c : m->ctx) {
        if (c->W != W || c->H != H) return fail(c0, RT_ERR_STATE, "rt_gather: contexts disagree on the resolution");
        total += (size_t)c->localRows * rowBytes;
    }
    if (m->pinnedBytes < total) {
        if (m->pinned) hipHostFree(m->pinned);
        m->pinned = nullptr;
        m->pinnedBytes = 0;
        HIP_TRY(c0, hipHostMalloc(&m->pinned, total ? total : 16, hipHostMallocDefault));
        m->pinnedBytes = total;
    }
    /* every device's held frames are launched, then every device's tile is on its way to pinned host memory (each on
     * its own stream and its own link), and only then does the host wait — device by device, scattering the tile that
     * has arrived while the others are still in flight */
    int rc = multi_flush_all(m);
    if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    size_t off = 0;
    for (RtContext* c : m->ctx) {
        const size_t n = (size_t)c->localRows * rowBytes;
        if (n) {
            const float* src = accumulated ? (c->boundAccum ? c->boundAccum : c->ownAccum) : (c->boundFrame ? c->boundFrame : c->ownFrame);
            hipError_t e = hipSetDevice(c->device);
            if (e == hipSuccess) e = hipMemcpyAsync((char*)m->pinned + off, src, n, hipMemcpyDeviceToHost, joined(c));
            if (e != hipSuccess) { /* copies already enqueued still write into m->pinned: wait for them before handing the error back */
                multi_drain(m);
                return fail(c, RT_ERR_HIP, "rt_gather: %s", hipGetErrorString(e));
            }
        }
        off += n;
    }
    off = 0;
    for (RtContext* c : m->ctx) {
        const int rows = c->localRows;
        const size_t n = (size_t)rows * rowBytes;
        if (!n) continue;
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        flush_timer(c);
        /* packed local rows -> their global rows; a strip's rows are contiguous in both */
        for (int l = 0; l < rows;) {
            const int g = rt_local_to_global_row(c, l);
            int run = c->stripRows - (g % c->stripRows);
            if (run > rows - l) run = rows - l;
            memcpy((char*)rgba + (size_t)g * rowBytes, (const char*)m->pinned + off + (size_t)l * rowBytes, (size_t)run * rowBytes);
            l += run;
        }
        off += n;
    }
    m->lastGatherMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}
/* The gather with a DEVICE destination: every context's strips go straight into their global rows of an image on the
 * device of context `root` — device-to-device copies on each source context's own stream (ordered after its frames; xGMI
 * between different GPUs, all sources concurrently), no host memory in between.  For hosts that display or post-process on
 * one of the GPUs. */
static int multi_gather_device(RtMulti* m, int root, void* d_rgba, size_t bytes, bool accumulated)
{
    if (!m || m->ctx.empty()) return fail(nullptr, RT_ERR_INVALID_ARG, "null multi context");
    RtContext* c0 = m->ctx[0];
    if (root < 0 || root >= (int)m->ctx.size()) return fail(c0, RT_ERR_INVALID_ARG, "rt_gather_*_to_device: root %d out of range", root);
    const int W = c0->W, H = c0->H;
    if (!d_rgba || bytes != (size_t)W * H * 16) return fail(c0, RT_ERR_INVALID_ARG, "rt_gather_*_to_device: bytes %zu != H*W*16 = %zu", bytes, (size_t)W * H * 16);
    const size_t rowBytes = (size_t)W * 16;
    const int rootDev = m->ctx[root]->device;
    for (RtContext* c : m->ctx) /* before anything is enqueued */
        if (c->W != W || c->H != H) return fail(c0, RT_ERR_STATE, "rt_gather: contexts disagree on the resolution");
    int rc = multi_flush_all(m);
    if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    for (RtContext* c : m->ctx) {
        const int rows = c->localRows;
        if (!rows) continue;
        const char* src = (const char*)(accumulated ? (c->boundAccum ? c->boundAccum : c->ownAccum) : (c->boundFrame ? c->boundFrame : c->ownFrame));
        hipError_t e = hipSetDevice(c->device);
        hipStream_t st = joined(c);
        for (int l = 0; l < rows && e == hipSuccess;) { /* a strip's rows are contiguous in the packed tile and in the image */
            const int g = rt_local_to_global_row(c, l);
            int run = c->stripRows - (g % c->stripRows);
            if (run > rows - l) run = rows - l;
            e = hipMemcpyPeerAsync((char*)d_rgba + (size_t)g * rowBytes, rootDev, src + (size_t)l * rowBytes, c->device, (size_t)run * rowBytes, st);
            l += run;
        }
        if (e != hipSuccess) { /* earlier contexts' copies into the caller's buffer are still in flight: wait before returning */
            multi_drain(m);
            return fail(c, RT_ERR_HIP, "rt_gather_*_to_device: %s", hipGetErrorString(e));
        }
    }
    for (RtContext* c : m->ctx) {
        if (!c->localRows) continue;
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        flush_timer(c);
    }
    m->lastGatherMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}
int rt_gather_accumulated_to_device(RtMulti* m, int root, void* d_rgba, size_t bytes) { return multi_gather_device(m, root, d_rgba, bytes, true); }
int rt_gather_frame_to_device(RtMulti* m, int root, void* d_rgba, size_t bytes) { return multi_gather_device(m, root, d_rgba, bytes, false); }
int rt_gather_accumulated(RtMulti* m, float* rgba, size_t bytes) { return multi_gather(m, rgba, bytes, true); }
int rt_gather_frame(RtMulti* m, float* rgba, size_t bytes) { return multi_gather(m, rgba, bytes, false); }

/* ---- RCCL, loaded on first use (no link-time dependency): the handful of entry points rt_gather_rccl needs */
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Rccl* rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("RT_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n) continue;
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        r.GroupStart = (int (*)())dlsym(r.lib, "ncclGroupStart");
        r.GroupEnd = (int (*)())dlsym(r.lib, "ncclGroupEnd");
        r.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(r.lib, "ncclSend");
        r.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(r.lib, "ncclRecv");
        r.CommCount = (int (*)(void*, int*))dlsym(r.lib, "ncclCommCount");
        r.CommUserRank = (int (*)(void*, int*))dlsym(r.lib, "ncclCommUserRank");
        r.GetErrorString = (const char* (*)(int))dlsym(r.lib, "ncclGetErrorString");
        r.ok = r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.CommCount && r.CommUserRank;
    });
    return &r;
}
} // namespace

int rt_gather_rccl(RtContext* ctx, void* nccl_comm, int root, int use_accumulated, void* d_rgba, size_t bytes)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_FLUSH(ctx);
    if (!nccl_comm) return fail(ctx, RT_ERR_INVALID_ARG, "rt_gather_rccl: null communicator");
    if (ctx->W == 0) return fail(ctx, RT_ERR_STATE, "rt_gather_rccl before rt_resize");
    Rccl* R = rccl();
    if (!R->ok) return fail(ctx, RT_ERR_STATE, "rt_gather_rccl: librccl.so could not be loaded (%s)", R->lib ? "missing symbols" : "dlopen failed; RT_RCCL_LIB names another path");
    int world = 0, rank = -1;
    if (R->CommCount(nccl_comm, &world) != 0 || R->CommUserRank(nccl_comm, &rank) != 0) return fail(ctx, RT_ERR_INVALID_ARG, "rt_gather_rccl: not a communicator");
    if (world != ctx->partCount || rank != ctx->partIndex)
        return fail(ctx, RT_ERR_STATE, "rt_gather_rccl: communicator rank %d of %d, context partition %d of %d", rank, world, ctx->partIndex, ctx->partCount);
    if (root < 0 || root >= world) return fail(ctx, RT_ERR_INVALID_ARG, "rt_gather_rccl: root %d out of range", root);
    const int W = ctx->W, H = ctx->H;
    const bool isRoot = rank == root;
    /* a collective: what only ONE rank can get wrong must not keep it out of the exchange (its peers would block in ncclSend for ever):
     * a root with a bad destination still receives every tile, and says so afterwards (include/rt_abi.h) */
    const bool badDst = isRoot && (!d_rgba || bytes != (size_t)W * H * 16);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = joined(ctx);
    const float* src = use_accumulated ? (ctx->boundAccum ? ctx->boundAccum : ctx->ownAccum) : (ctx->boundFrame ? ctx->boundFrame : ctx->ownFrame);
    /* root: one staging area for every rank's packed tile (H rows in all), in the display scratch */
    std::vector<size_t> rowOff(world + 1, 0);
    for (int r = 0; r < world; r++) rowOff[r + 1] = rowOff[r] + (size_t)local_rows_for(H, ctx->stripRows, r, world);
    char* staging = nullptr;
    if (isRoot) {
        void* p = nullptr;
        const int rc = display_scratch(ctx, rowOff[world] * (size_t)W * 16, &p);
        if (rc) return rc;
        staging = (char*)p;
    }
    auto check = [&](int e, const char* what) -> int {
        if (e == 0) return RT_OK;
        return fail(ctx, RT_ERR_HIP, "rt_gather_rccl: %s: %s", what, R->GetErrorString ? R->GetErrorString(e) : "RCCL error");
    };
    /* every rank sends its tile to root, root receives from every rank — its own included, so that a one-rank communicator runs the
     * same code; one group: the transfers progress together */
    int rc = check(R->GroupStart(), "ncclGroupStart");
    if (rc) return rc;
    int e = 0;
    if (ctx->localRows) e = R->Send(src, (size_t)ctx->localRows * W * 4, /*ncclFloat32*/ 7, root, nccl_comm, st);
    if (isRoot)
        for (int r = 0; r < world && e == 0; r++) {
            const size_t rows = rowOff[r + 1] - rowOff[r];
            if (rows) e = R->Recv(staging + rowOff[r] * (size_t)W * 16, rows * W * 4, 7, r, nccl_comm, st);
        }
    const int eEnd = R->GroupEnd();
    if ((rc = check(e, "ncclSend / ncclRecv"))) return rc;
    if ((rc = check(eEnd, "ncclGroupEnd"))) return rc;
    if (isRoot && !badDst)
        for (int r = 0; r < world; r++) {
            const int rows = (int)(rowOff[r + 1] - rowOff[r]);
            if (!rows) continue;
            size_t n = (size_t)rows * W;
            int blocks = (int)((n + 255) / 256);
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(rtk::rt_unpack_strips_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)(staging + rowOff[r] * (size_t)W * 16), (float4*)d_rgba, W, rows,
                               ctx->stripRows, r, world);
            HIP_TRY(ctx, hipGetLastError());
        }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    flush_timer(ctx);
    if (badDst) return fail(ctx, RT_ERR_INVALID_ARG, "rt_gather_rccl: bytes %zu != H*W*16 = %zu (the tiles were received and dropped)", bytes, (size_t)W * H * 16);
    return RT_OK;
}

int rt_multi_get_counters(RtMulti* m, RtCounters* out)
{
    if (!m || !out) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_multi_get_counters: null argument");
    memset(out, 0, sizeof(*out));
    {
        int rc = multi_flush_all(m);
        if (rc) return rc;
    }
    for (RtContext* c : m->ctx) {
        RtCounters k;
        int rc = rt_get_counters(c, &k);
        if (rc != RT_OK) return rc;
        out->segments += k.segments; out->innerSteps += k.innerSteps; out->leafSteps += k.leafSteps; out->triTests += k.triTests;
        out->sphereTests += k.sphereTests; out->modelVisits += k.modelVisits; out->pixelFrames += k.pixelFrames;
        if (k.gpuMs > out->gpuMs) out->gpuMs = k.gpuMs;
    }
    return RT_OK;
}

} /* extern "C" */
