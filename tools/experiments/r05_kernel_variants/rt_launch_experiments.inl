/*
 * rt_launch_experiments.inl — the HOST halves of the kernel experiments, included by rt_context.hip through its one
 * experiment hook (-DRT_EXPERIMENTS; the product build compiles without this file).  Every experiment here is a measured
 * negative or neutral result that stays reproducible ("rejected experiments"):
 *
 *   make queued        -DRT_QUEUED_EXPERIMENT   queued stages, chains parked in device memory (rt_kernels_q.h; RT_QUEUED=1)
 *   make wg            -DRT_WG_EXPERIMENT       traversal waves + shading waves around an LDS pool (rt_kernels_wg.h; RT_WG=1)
 *   make lds-fetch     -DRT_LDS_NODE_FETCH      node quarters staged through LDS (rt_kernels.h)
 *   make xcd           -DRT_XCD_EXPERIMENT      per-XCD ranges of the tile queue (rt_kernels.h; RT_XCD_AFFINITY=1|2)
 *
 * The hooks, in the order launch_frames calls them:
 *   exp_init / exp_destroy          environment, device buffers
 *   exp_choose                      may replace the kernels, the LDS size and the workgroup size of a LaunchPlan
 *   exp_prepare                     per-wave records of the replaced kernels
 *   exp_pre_launch / exp_post_launch  per kernel: record pointers, private tile counters
 */
#ifdef RT_WG_EXPERIMENT
#define RT_QUEUED_EXPERIMENT
#include "rt_kernels_wg.h"
#elif defined(RT_QUEUED_EXPERIMENT)
#include "rt_kernels_q.h"
#endif

struct ExperimentState {
    bool queued = false;        /* RT_QUEUED=1 */
    bool wg = false;            /* RT_WG=1: the workgroup form (make wg) */
    int wgTravWaves = 6;        /* RT_WG_NT */
    int qFlushMin = 16, qRefillMin = 16, qStarveMin = 32; /* RT_Q_FLUSH / RT_Q_REFILL / RT_Q_STARVE (scheduling only) */
    void* dQRecords = nullptr;  /* 2 launch slots x qWaves x RT_Q_WAVE_DWORDS dwords */
    long long qWaves = 0;
    void* dWgRecords = nullptr; /* workgroup form: 2 launch slots x wgUnits x RT_WG_GLOBAL_DWORDS dwords of pixel records */
    long long wgUnits = 0;
    /* RT_XCD_AFFINITY=1|2 (fused launches): per-XCD ranges of the tile queue (KArgs::xcdQueues); 2 = LPT off and a static order in which
     * the eight ranges are eight compact blocks of the image (4 x 2) */
    int xcdAffinity = 0;
    unsigned long long* dXcdQueues = nullptr; /* 2 launch slots x 8 counters */
    uint32_t* dBlockOrder = nullptr;
    int blockOrderTiles = 0;
    /* this launch */
    bool queuedNow = false, wgNow = false, xcdNow = false;
};
#define RT_HAVE_EXPERIMENT_STATE 1

static void exp_init(RtContext* ctx, ExperimentState& x)
{
    if (const char* q = getenv("RT_QUEUED")) x.queued = atoi(q) != 0;
    if (const char* q = getenv("RT_WG")) x.wg = atoi(q) != 0;
    if (const char* q = getenv("RT_WG_NT")) x.wgTravWaves = atoi(q);
    if (const char* q = getenv("RT_Q_FLUSH")) x.qFlushMin = atoi(q);
    if (const char* q = getenv("RT_Q_REFILL")) x.qRefillMin = atoi(q);
    if (const char* q = getenv("RT_Q_STARVE")) x.qStarveMin = atoi(q);
    if (const char* l = getenv("RT_XCD_AFFINITY")) x.xcdAffinity = atoi(l);
    if (x.xcdAffinity == 2) ctx->lptEnabled = false;
}

static void exp_destroy(ExperimentState& x)
{
    hipFree(x.dQRecords);
    hipFree(x.dWgRecords);
    hipFree(x.dXcdQueues);
    hipFree(x.dBlockOrder);
}

static int exp_choose(RtContext* ctx, ExperimentState& x, KArgs& a, LaunchPlan& plan, bool many)
{
    (void)a; (void)many;
    x.queuedNow = x.wgNow = false;
#ifdef RT_LDS_NODE_FETCH
    plan.ldsBytes += 4 * RT_WAVE * 16; /* the node slab (rt_kernels.h) */
#endif
#ifdef RT_QUEUED_EXPERIMENT
    if (x.queued && !ctx->flatScene && !many) { /* BVH scenes with up to 64 models */
        x.queuedNow = true;
        plan.kern = ctx->stats ? rtk::rt_trace_q_kernel<true> : rtk::rt_trace_q_kernel<false>;
        plan.kernHalf = ctx->stats ? rtk::rt_trace_q_half_kernel<true> : rtk::rt_trace_q_half_kernel<false>;
        a.qFlushMin = x.qFlushMin < 1 ? 1 : x.qFlushMin > 64 ? 64 : x.qFlushMin;
        a.qRefillMin = x.qRefillMin < 1 ? 1 : x.qRefillMin > 64 ? 64 : x.qRefillMin;
        a.qStarveMin = x.qStarveMin < 0 ? 0 : x.qStarveMin;
        plan.variant = 6 + (ctx->stats ? 1 : 0);
    }
#endif
#ifdef RT_WG_EXPERIMENT
    if (x.wg && !ctx->flatScene && !many) {
        x.wgNow = true;
        x.queuedNow = false;
        plan.kern = ctx->stats ? rtk::rt_trace_wg_kernel<true> : rtk::rt_trace_wg_kernel<false>;
        plan.kernHalf = plan.kern;
        a.wgTravWaves = x.wgTravWaves < 1 ? 1 : x.wgTravWaves > RT_WG_WAVES - 1 ? RT_WG_WAVES - 1 : x.wgTravWaves;
        a.qFlushMin = x.qFlushMin < 1 ? 1 : x.qFlushMin > 64 ? 64 : x.qFlushMin;
        a.qStarveMin = getenv("RT_Q_STARVE") ? (x.qStarveMin < 0 ? 0 : x.qStarveMin) : 64;
        {   /* the largest pool that keeps three workgroups (24 waves) on a CU's 160 KB of LDS */
            const size_t fixed = (size_t)a.wgTravWaves * ctx->stackEntries * RT_WAVE * sizeof(uint32_t) + sizeof(rtk::WgShared) + 16;
            const size_t budget = 51 * 1024; /* 3 x 51 KB + allocation granules < 160 KB */
            long long pool = budget > fixed ? (long long)((budget - fixed) / (RT_WG_REC * sizeof(uint32_t))) : 0;
            if (const char* e = getenv("RT_WG_POOL")) pool = atoll(e);
            a.wgPool = (int)(pool > RT_WG_POOL_MAX ? RT_WG_POOL_MAX : pool < 128 ? 128 : pool);
        }
        plan.ldsBytes = ((size_t)a.wgTravWaves * ctx->stackEntries * RT_WAVE + (size_t)a.wgPool * RT_WG_REC) * sizeof(uint32_t) + sizeof(rtk::WgShared) + 16;
        plan.blockThreads = RT_WG_THREADS;
        plan.twoPartsAllowed = false;
        plan.variant = 8 + (ctx->stats ? 1 : 0);
    }
#endif
    return RT_OK;
}

/* per-wave records of the replaced kernels (kernels in flight use the old block: it is freed after a synchronise) */
static int exp_prepare(RtContext* ctx, ExperimentState& x, long long waves)
{
    (void)waves;
#ifdef RT_WG_EXPERIMENT
    if (x.wgNow && x.wgUnits < waves) {
        HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
        hipFree(x.dWgRecords); x.dWgRecords = nullptr; x.wgUnits = 0;
        HIP_TRY(ctx, hipMalloc(&x.dWgRecords, (size_t)2 * waves * RT_WG_GLOBAL_DWORDS * sizeof(uint32_t)));
        x.wgUnits = waves;
    }
#endif
#ifdef RT_QUEUED_EXPERIMENT
    if (x.queuedNow && x.qWaves < waves) {
        HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
        hipFree(x.dQRecords); x.dQRecords = nullptr; x.qWaves = 0;
        HIP_TRY(ctx, hipMalloc(&x.dQRecords, (size_t)2 * waves * RT_Q_WAVE_DWORDS * sizeof(uint32_t)));
        x.qWaves = waves;
    }
#endif
    (void)ctx;
    return RT_OK;
}

/* before one kernel of the launch goes out on stream st (slot q); *ownQueue = the kernel counts its tiles from a private zero */
static int exp_pre_launch(RtContext* ctx, ExperimentState& x, KArgs& a, int q, hipStream_t st, bool staged, int tiles, bool* ownQueue)
{
    (void)staged; (void)tiles; (void)st; (void)q;
    *ownQueue = false;
    x.xcdNow = false;
#ifdef RT_QUEUED_EXPERIMENT
    a.qRecords = x.queuedNow ? (uint32_t*)x.dQRecords + (size_t)q * x.qWaves * RT_Q_WAVE_DWORDS : nullptr;
#endif
#ifdef RT_XCD_EXPERIMENT
    if (x.xcdAffinity > 0 && staged && !x.wgNow && !x.queuedNow) {
        x.xcdNow = true;
        if (!x.dXcdQueues) HIP_TRY(ctx, hipMalloc(&x.dXcdQueues, 16 * sizeof(unsigned long long)));
        if (x.xcdAffinity == 2) {
            if (x.blockOrderTiles != tiles) { /* position -> tile: eight blocks (4 across, 2 down), rows within a block */
                std::vector<uint32_t> order;
                order.reserve(tiles);
                for (int by = 0; by < 2; by++)
                    for (int bx = 0; bx < 4; bx++) {
                        const int x0 = a.tilesX * bx / 4, x1 = a.tilesX * (bx + 1) / 4, y0 = a.tilesY * by / 2, y1 = a.tilesY * (by + 1) / 2;
                        for (int y = y0; y < y1; y++)
                            for (int xx = x0; xx < x1; xx++) order.push_back((uint32_t)(y * a.tilesX + xx));
                    }
                HIP_TRY(ctx, hipStreamSynchronize(joined(ctx)));
                hipFree(x.dBlockOrder); x.dBlockOrder = nullptr;
                HIP_TRY(ctx, hipMalloc(&x.dBlockOrder, sizeof(uint32_t) * tiles));
                HIP_TRY(ctx, hipMemcpy(x.dBlockOrder, order.data(), sizeof(uint32_t) * tiles, hipMemcpyHostToDevice));
                x.blockOrderTiles = tiles;
            }
            a.tileOrder = x.dBlockOrder;
        }
        a.xcdQueues = x.dXcdQueues + 8 * q;
        a.queueStart = 1;
        HIP_TRY(ctx, hipMemsetAsync(a.xcdQueues, 0, 8 * sizeof(unsigned long long), st));
        *ownQueue = true;
    }
#endif
#ifdef RT_WG_EXPERIMENT
    if (x.wgNow) { /* its own tile counter from zero (the number of overshooting fetches of a workgroup launch is not fixed) */
        a.qRecords = (uint32_t*)x.dWgRecords + (size_t)q * x.wgUnits * RT_WG_GLOBAL_DWORDS;
        HIP_TRY(ctx, hipMemsetAsync(ctx->dTileQueue + q, 0, sizeof(unsigned long long), st));
        a.tileQueueBase = 0ull;
        *ownQueue = true;
    }
#endif
    (void)ctx;
    return RT_OK;
}

static int exp_post_launch(RtContext* ctx, ExperimentState& x, int q, hipStream_t st)
{
    (void)q; (void)st;
#ifdef RT_WG_EXPERIMENT
    if (x.wgNow) { /* the product kernels' monotonic counter restarts from zero, on the device and in the host's book */
        HIP_TRY(ctx, hipMemsetAsync(ctx->dTileQueue + q, 0, sizeof(unsigned long long), st));
        ctx->tileQueueNext[q] = 0ull;
    }
#endif
    (void)ctx; (void)x;
    return RT_OK;
}
