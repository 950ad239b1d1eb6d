/*
 * rt_kernels_wg.h — EXPERIMENT (make wg): the queued stages of rt_kernels_q.h with the parked chains ON THE CHIP.
 *
 * rt_kernels_q.h showed what batching buys (every shade / glass / camera batch with full lanes: a third fewer instructions) and what
 * it costs when the parked pixel chains live in device memory (3x slower: 1.18 M chains fit no cache).  The chip has no room for
 * three chains per lane at the shipped occupancy — if every wave keeps a traversal stack.  Here a workgroup of eight waves splits
 * the roles: NT TRAVERSAL waves own the LDS stacks and do nothing but walk BVHs (their lanes are refilled from a queue as they
 * finish); the other 8 - NT SHADING waves own no stack and run the shade / glass / camera batches (RC:488-538, 545-582, the spheres
 * and the root filter) for chains they take from queues.  A chain between two segments is 22 dwords in a slot of a workgroup-wide
 * LDS pool; its slot number travels through multi-producer / multi-consumer rings in LDS (ticket by atomic add, entries with an
 * "empty" sentinel).  LDS per workgroup = NT stacks + 192 slots x 23 dwords + five rings: three workgroups = 24 waves per CU as
 * before, because two of eight waves have no stack.  A traversal lane that finishes SWAPS: it takes the next waiting ray's slot,
 * reads that chain, writes its finished one into the same slot and passes the slot on to the shading queues.
 *
 * Per-pixel data that is touched once per path (running sum, focus point, ...) lives in a per-workgroup array in device memory,
 * like the PX_COLD records of rt_kernels.h.  Chains never change their own arithmetic or order: images and counters are the oracle's.
 */
#ifndef RT_KERNELS_WG_H
#define RT_KERNELS_WG_H

#include "rt_kernels_q.h"

#define RT_WG_WAVES 8
#define RT_WG_THREADS (RT_WG_WAVES * RT_WAVE)
#define RT_WG_POOL_MAX 448 /* parked chains per workgroup: the host picks the largest pool that keeps three workgroups on a CU */
#define RT_WG_REC 17       /* dwords per slot: odd, so that consecutive slots start in different banks */
#define RT_WG_RING 512     /* entries per ring (>= RT_WG_POOL_MAX, power of two) */
#define RT_WG_MAX_CHAINS (RT_WG_WAVES * RT_WAVE + RT_WG_POOL_MAX) /* every chain is in a lane or in a slot */
#define RT_WG_PIX_DWORDS 16
#define RT_WG_GLOBAL_DWORDS (RT_WG_MAX_CHAINS * RT_WG_PIX_DWORDS)
#define RT_WG_EMPTY 0xffffu
#define RT_WG_NO_PIXEL 0x7fffu

namespace rtk {

/* slot record (LDS) */
enum {
    WF_RPX = 0, WF_RPY, WF_RPZ, WF_RDX, WF_RDY, WF_RDZ, WF_TRX, WF_TRY, WF_TRZ, WF_RNG,
    WF_BP,   /* bounce (bits 0-15) | pixel record (bits 16-30) | bit 31: a finished path's light waits in the pixel record (RC:578) */
    WF_HDST, WF_HOBJ, /* result so far: the spheres' on the way to the traversal lanes, the final one on the way back */
    WF_X0, WF_X1, WF_X2, WF_X3 /* to the traversal lanes: candidate mask (X0, X1); back: winning triangle, u, v, det */
};
#define WF_CANDLO WF_X0
#define WF_CANDHI WF_X1
#define WF_HTRI WF_X0
#define WF_HU WF_X1
#define WF_HV WF_X2
#define WF_HDET WF_X3
#define RT_WG_PENDING 0x80000000u
/* pixel record (device memory, per workgroup): touched once per path, not per segment.  The light a path gathers (totalLight of Trace,
 * RC:481) lives here too: it only changes at emissive hits and at the sky, and x + (+-0) == x for every x a sum that starts at +0 can
 * hold, so the segments that add nothing never touch it */
enum { WP_TIX = 0, WP_TIY, WP_TIZ, WP_FPX, WP_FPY, WP_FPZ, WP_PIXIDX, WP_PIXLIN, WP_FRAME, WP_SEGS, WP_SAMPLE, WP_PLX, WP_PLY, WP_PLZ };

struct WgQueue {
    uint32_t head, tail;
    uint16_t ring[RT_WG_RING];
};
enum { WQ_FREE = 0, WQ_CAM, WQ_RAY, WQ_HIT, WQ_GLASS, RT_WG_QUEUES };
struct WgShared {
    WgQueue q[RT_WG_QUEUES];
    uint32_t live;           /* chains that have a pixel */
    uint32_t created;        /* pixel records handed out (chains ever created) */
    uint32_t tilesExhausted; /* the launch has no unassigned pixel left */
    uint32_t abort;          /* watchdog: a wave waited too long */
    uint32_t camLock;        /* one wave at a time hands out pixels (the pool tile below belongs to the lock holder) */
    int32_t poolX0, poolRow0, poolY0, poolPos, poolFrame, queueEmpty; /* the workgroup's pool tile: next unassigned pixels of the current (tile, frame) item */
};
#define RT_WG_SHARED_DWORDS ((sizeof(rtk::WgShared) + 3) / 4)

__device__ __forceinline__ uint32_t wq_count(const WgQueue* q)
{
    return *(volatile const uint32_t*)&q->tail - *(volatile const uint32_t*)&q->head;
}
/* lanes with pred append their slot (one ticket per wave) */
__device__ __forceinline__ void wq_push(WgQueue* q, bool pred, int slot)
{
    const unsigned long long m = __ballot(pred);
    if (!m) return;
    const int n = __popcll(m);
    uint32_t base = 0;
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) base = atomicAdd(&q->tail, (uint32_t)n);
    base = __builtin_amdgcn_readlane(base, __ffsll((long long)m) - 1);
    if (pred) *(volatile uint16_t*)&q->ring[(base + (uint32_t)q_rank(m)) & (RT_WG_RING - 1)] = (uint16_t)slot;
}
/* lanes with pred (in rank order) take entries while there are any; returns how many were taken (wave-uniform); the lanes with
 * rank < that get their slot.  `spins` feeds the watchdog. */
__device__ __forceinline__ int wq_pop(WgQueue* q, bool pred, int& slot, uint32_t& spins)
{
    const unsigned long long m = __ballot(pred);
    if (!m) return 0;
    const int want = __popcll(m);
    uint32_t h = 0, n = 0;
    if ((threadIdx.x & 63) == 0) {
        for (;;) {
            h = *(volatile uint32_t*)&q->head;
            const uint32_t avail = *(volatile uint32_t*)&q->tail - h;
            n = avail < (uint32_t)want ? avail : (uint32_t)want;
            if (!n || atomicCAS(&q->head, h, h + n) == h) break;
        }
    }
    h = __builtin_amdgcn_readfirstlane(h);
    n = __builtin_amdgcn_readfirstlane(n);
    const int rank = q_rank(m);
    if (pred && rank < (int)n) {
        volatile uint16_t* e = &q->ring[(h + (uint32_t)rank) & (RT_WG_RING - 1)];
        uint32_t v = *e;
        while (v == RT_WG_EMPTY && spins < (1u << 24)) { /* the producer has its ticket and is about to write */
            spins++;
            v = *e;
        }
        *e = (uint16_t)RT_WG_EMPTY;
        slot = (int)v;
    }
    return (int)n;
}

template <bool STATS>
__device__ __forceinline__ void trace_body_wg(const KArgs& a)
{
    extern __shared__ uint32_t s_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int NT, stackEntries, POOL;
    {
        const RT_CAS KArgs& c = cold_args();
        NT = c.wgTravWaves;
        stackEntries = c.stackEntries;
        POOL = c.wgPool;
    }
    const bool isT = wave < NT;
    WgShared* const sh = reinterpret_cast<WgShared*>(s_lds + (size_t)NT * stackEntries * RT_WAVE);
    uint32_t* const pool = s_lds + (size_t)NT * stackEntries * RT_WAVE + (sizeof(WgShared) + 15) / 16 * 4;
    uint32_t* const stackBase = s_lds + (size_t)(isT ? wave : 0) * stackEntries * RT_WAVE + lane;
    uint32_t* const pix = cold_args().qRecords + (size_t)blockIdx.x * RT_WG_GLOBAL_DWORDS;
#define WS(slot, f) pool[(slot) * RT_WG_REC + (f)]
#define WSF(slot, f) __uint_as_float(WS(slot, f))
#define WP(rec, f) pix[(rec) * RT_WG_PIX_DWORDS + (f)]

    /* ---- initialise the shared state: every slot free */
    for (int i = threadIdx.x; i < RT_WG_QUEUES * RT_WG_RING; i += RT_WG_THREADS) sh->q[i / RT_WG_RING].ring[i % RT_WG_RING] = (uint16_t)RT_WG_EMPTY;
    __syncthreads();
    for (int i = threadIdx.x; i < POOL; i += RT_WG_THREADS) sh->q[WQ_FREE].ring[i] = (uint16_t)i;
    if (threadIdx.x < RT_WG_QUEUES) {
        sh->q[threadIdx.x].head = 0;
        sh->q[threadIdx.x].tail = threadIdx.x == WQ_FREE ? (uint32_t)POOL : 0u;
    }
    if (threadIdx.x == 0) {
        sh->live = 0; sh->created = 0; sh->tilesExhausted = cold_args().nFrames <= 0 ? 1u : 0u; sh->abort = 0; sh->camLock = 0;
        sh->poolX0 = sh->poolRow0 = sh->poolY0 = sh->poolFrame = 0; sh->poolPos = 64; sh->queueEmpty = cold_args().nFrames <= 0 ? 1 : 0;
    }
    __syncthreads();

    uint32_t segments = 0, spins = 0;
    Stats st = {};
#ifdef RT_PHASE_TIMES
    st.phPrev = -1;
#endif
    const int flushMin = a.qFlushMin;
    const uint32_t starveMin = (uint32_t)a.qStarveMin; /* shading waves take partial batches when fewer rays than this wait */

    if (isT) {
        /* =================================================================== TRAVERSAL WAVE */
        bool busy = false;
        rt_f3 rpos = rt_v3s(0.0f), rdir = rt_v3s(0.0f), tr = rt_v3s(0.0f);
        uint32_t rng = 0, bp = 0;
        SceneHit h;
        Trav t;
        h.dst = RT_INF; h.obj = -1; h.tri = -1; h.u = h.v = h.det = 0.0f; h.backface = false;
        t.cand = 0; t.rootStep = false; t.m = 0; t.cur = RT_CODE_DONE; t.sp = 0; t.lpos = t.ldir = t.linv = rt_v3s(0.0f); t.triBase = 0; t.cull = true;
        bool done = false; /* busy && finished traversing: waits to be handed over */
        auto read_chain = [&](int s) {
            rpos = rt_v3(WSF(s, WF_RPX), WSF(s, WF_RPY), WSF(s, WF_RPZ));
            rdir = rt_v3(WSF(s, WF_RDX), WSF(s, WF_RDY), WSF(s, WF_RDZ));
            tr = rt_v3(WSF(s, WF_TRX), WSF(s, WF_TRY), WSF(s, WF_TRZ));
            rng = WS(s, WF_RNG); bp = WS(s, WF_BP);
            h.dst = WSF(s, WF_HDST);
            q_unpack_obj(WS(s, WF_HOBJ), h.obj, h.backface);
            h.tri = -1; h.u = h.v = h.det = 0.0f;
            t.cand = (unsigned long long)WS(s, WF_CANDLO) | ((unsigned long long)WS(s, WF_CANDHI) << 32);
            t.m = -1; t.cur = RT_CODE_NEXT_MODEL; t.sp = 0; t.rootStep = false;
            t.lpos = t.ldir = t.linv = rt_v3s(0.0f); t.triBase = 0; t.cull = true;
        };
        for (;;) {
            phase_mark<STATS>(st, PH_LOOP);
            if (*(volatile uint32_t*)&sh->abort) break;
            /* ---- finished lanes hand their chain over: swap with a waiting ray where there is one, else park in a free slot */
            bool handed = false;
            int s = 0;
            {
                const int nSwap = wq_pop(&sh->q[WQ_RAY], done, s, spins);
                const bool swapped = done && q_rank(__ballot(done)) < nSwap;
                int sFree = 0;
                const bool wantFree = done && !swapped;
                const int nFree = wq_pop(&sh->q[WQ_FREE], wantFree, sFree, spins);
                const bool parked = wantFree && q_rank(__ballot(wantFree)) < nFree;
                const int target = swapped ? s : sFree;
                if (swapped || parked) {
                    /* the finished chain's state (registers) <-> the slot (swap: read the waiting chain first) */
                    const rt_f3 fr = rpos, fd = rdir, ft = tr;
                    const uint32_t frng = rng, fbp = bp;
                    const SceneHit fh = h;
                    if (swapped) read_chain(target);
                    {
                        WS(target, WF_RPX) = __float_as_uint(fr.x); WS(target, WF_RPY) = __float_as_uint(fr.y); WS(target, WF_RPZ) = __float_as_uint(fr.z);
                        WS(target, WF_RDX) = __float_as_uint(fd.x); WS(target, WF_RDY) = __float_as_uint(fd.y); WS(target, WF_RDZ) = __float_as_uint(fd.z);
                        WS(target, WF_TRX) = __float_as_uint(ft.x); WS(target, WF_TRY) = __float_as_uint(ft.y); WS(target, WF_TRZ) = __float_as_uint(ft.z);
                        WS(target, WF_RNG) = frng; WS(target, WF_BP) = fbp;
                        WS(target, WF_HDST) = __float_as_uint(fh.dst); WS(target, WF_HOBJ) = q_pack_obj(fh.obj, fh.backface); WS(target, WF_HTRI) = (uint32_t)fh.tri;
                        WS(target, WF_HU) = __float_as_uint(fh.u); WS(target, WF_HV) = __float_as_uint(fh.v); WS(target, WF_HDET) = __float_as_uint(fh.det);
                    }
                    handed = true;
                    bool glass = false;
                    if (fh.obj >= 0) glass = a.materials[fh.obj].flag == RT_MATERIAL_GLASS;
                    done = false;
                    busy = swapped;
                    /* queue of the finished chain's kind */
                    wq_push(&sh->q[WQ_GLASS], glass, target);
                    wq_push(&sh->q[WQ_HIT], !glass, target);
                }
            }
            (void)handed;
            /* ---- empty lanes take waiting rays; their slots become free */
            {
                int s2 = 0;
                const bool want = !busy;
                const int n = wq_pop(&sh->q[WQ_RAY], want, s2, spins);
                const bool got = want && q_rank(__ballot(want)) < n;
                if (got) {
                    read_chain(s2);
                    busy = true;
                }
                wq_push(&sh->q[WQ_FREE], got, s2);
            }
            /* ---- traverse until flushMin more lanes have finished */
            const unsigned long long trav = __ballot(busy && !done);
            if (trav) {
                const int nNow = __popcll(trav);
                if (busy && !done) done = traverse<STATS, true, false>(a, rpos, rdir, stackBase, stackBase, h, t, st, nNow > flushMin ? nNow - flushMin : 0);
            } else if (__ballot(busy) == 0ull) {
                /* nothing in this wave: finished?  else wait a little for the shading waves */
                if (*(volatile uint32_t*)&sh->tilesExhausted && *(volatile uint32_t*)&sh->live == 0u) break;
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 24)) { sh->abort = 1; break; }
            } else {
                /* lanes wait to hand over (no waiting ray, no free slot): the shading waves will make room */
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 24)) { sh->abort = 1; break; }
            }
        }
    } else {
        /* =================================================================== SHADING WAVE */
        int poolX0 = 0, poolRow0 = 0, poolY0 = 0, poolPos = 64, poolFrame = 0;
        bool queueEmpty = false;
#define RTW_ITEM(c, q, tilePos)                                                                               \
    do {                                                                                                      \
        if ((c).nFrames > 1) { tilePos = (q) / (c).frameGroups; poolFrame = (c).frame0 + ((q) - tilePos * (c).frameGroups) * (c).frameGroup; } \
        else { tilePos = (q); poolFrame = (c).frame0; }                                                       \
    } while (0)
#define RTW_SET_POOL(c, tile)                                                                                 \
    do {                                                                                                      \
        const int ty_ = (tile) / (c).tilesX;                                                                  \
        poolX0 = ((tile) - ty_ * (c).tilesX) * 8;                                                             \
        poolRow0 = ty_ * 8;                                                                                   \
        const int ls_ = poolRow0 / (c).stripRows;                                                             \
        poolY0 = (ls_ * (c).partCount + (c).partIndex) * (c).stripRows + (poolRow0 - ls_ * (c).stripRows);    \
        poolPos = 0;                                                                                          \
    } while (0)
        /* spheres + root filter of a new ray, into its slot, on to the traversal waves (or straight to the hit queues) */
        auto launch_ray = [&](bool pred, int slot, rt_f3 o, rt_f3 d, rt_f3 tr, uint32_t rng, uint32_t bp) {
            SceneHit bh;
            Trav bt;
            bh.dst = RT_INF; bh.obj = -1; bh.backface = false;
            bt.cand = 0;
            if (pred) {
                phase_mark<STATS>(st, PH_SPHERES);
                begin_intersect<STATS, false, false>(a, o, d, stackBase, bh, bt, st);
                segments++;
                WS(slot, WF_RPX) = __float_as_uint(o.x); WS(slot, WF_RPY) = __float_as_uint(o.y); WS(slot, WF_RPZ) = __float_as_uint(o.z);
                WS(slot, WF_RDX) = __float_as_uint(d.x); WS(slot, WF_RDY) = __float_as_uint(d.y); WS(slot, WF_RDZ) = __float_as_uint(d.z);
                WS(slot, WF_TRX) = __float_as_uint(tr.x); WS(slot, WF_TRY) = __float_as_uint(tr.y); WS(slot, WF_TRZ) = __float_as_uint(tr.z);
                WS(slot, WF_RNG) = rng; WS(slot, WF_BP) = bp;
                WS(slot, WF_HDST) = __float_as_uint(bh.dst); WS(slot, WF_HOBJ) = q_pack_obj(bh.obj, bh.backface);
            }
            const bool toTrav = pred && bt.cand != 0ull;
            if (toTrav) { WS(slot, WF_CANDLO) = (uint32_t)bt.cand; WS(slot, WF_CANDHI) = (uint32_t)(bt.cand >> 32); }
            else if (pred) { WS(slot, WF_HTRI) = 0xffffffffu; WS(slot, WF_HU) = 0u; WS(slot, WF_HV) = 0u; WS(slot, WF_HDET) = 0u; } /* the spheres' result is final */
            bool glass = false;
            if (pred && !toTrav && bh.obj >= 0) glass = a.materials[bh.obj].flag == RT_MATERIAL_GLASS;
            wq_push(&sh->q[WQ_RAY], toTrav, slot);
            wq_push(&sh->q[WQ_GLASS], pred && !toTrav && glass, slot);
            wq_push(&sh->q[WQ_HIT], pred && !toTrav && !glass, slot);
        };
        for (;;) {
            phase_mark<STATS>(st, PH_LOOP);
            if (*(volatile uint32_t*)&sh->abort) break;
            const uint32_t nh = wq_count(&sh->q[WQ_HIT]), ng = wq_count(&sh->q[WQ_GLASS]), nc = wq_count(&sh->q[WQ_CAM]), nr = wq_count(&sh->q[WQ_RAY]);
            const uint32_t nfree = wq_count(&sh->q[WQ_FREE]);
            const bool exhausted = *(volatile uint32_t*)&sh->tilesExhausted != 0u;
            /* new chains are made from free slots while the launch has pixels, as long as the traversal waves want rays */
            const bool canCreate = !exhausted && nfree > 64u && nr < 128u && *(volatile uint32_t*)&sh->created + 64u <= (uint32_t)(NT * RT_WAVE + POOL);
            int which = -1; /* WQ_HIT / WQ_GLASS / WQ_CAM, or WQ_FREE = create */
            (void)nfree;
            if (nh >= RT_WAVE) which = WQ_HIT;
            else if (ng >= RT_WAVE) which = WQ_GLASS;
            else if (nc >= RT_WAVE) which = WQ_CAM;
            else if (canCreate) which = WQ_FREE;
            else if (nr < starveMin && (nh | ng | nc)) which = (nh >= ng && nh >= nc) ? WQ_HIT : (ng >= nc ? WQ_GLASS : WQ_CAM); /* the traversal lanes run dry */
            if (which < 0) {
                if (exhausted && *(volatile uint32_t*)&sh->live == 0u) break;
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1u << 24)) { sh->abort = 1; break; }
                continue;
            }
            const bool camStage = which == WQ_CAM || which == WQ_FREE;
            if (camStage) { /* the pixel hand-out is serialised: its state is the workgroup's */
                uint32_t got = 0;
                if (lane == 0) got = atomicCAS(&sh->camLock, 0u, 1u) == 0u ? 1u : 0u;
                got = __builtin_amdgcn_readfirstlane(got);
                if (!got) { /* the other shading wave is at it: anything else to do? */
                    which = nh ? WQ_HIT : (ng ? WQ_GLASS : -1);
                    if (which < 0) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1u << 24)) { sh->abort = 1; break; }
                        continue;
                    }
                } else {
                    poolX0 = sh->poolX0; poolRow0 = sh->poolRow0; poolY0 = sh->poolY0; poolPos = sh->poolPos; poolFrame = sh->poolFrame;
                    queueEmpty = sh->queueEmpty != 0;
                }
            }
            const bool camHeld = camStage && (which == WQ_CAM || which == WQ_FREE);
            int slot = 0;
            const int n = wq_pop(&sh->q[which], true, slot, spins);
            const bool act = lane < n;
            if (n == 0) {
                if (camHeld && lane == 0) { __builtin_amdgcn_s_waitcnt(0xc07f); sh->camLock = 0u; }
                continue;
            }
            if (which == WQ_HIT || which == WQ_GLASS) {
                /* ---- shade batch */
                rt_f3 o = rt_v3s(0.0f), d = rt_v3s(0.0f), tr = rt_v3s(0.0f);
                uint32_t rng = 0, bp = 0;
                int bounce = 0;
                bool endPath = false;
                if (act) {
                    o = rt_v3(WSF(slot, WF_RPX), WSF(slot, WF_RPY), WSF(slot, WF_RPZ));
                    d = rt_v3(WSF(slot, WF_RDX), WSF(slot, WF_RDY), WSF(slot, WF_RDZ));
                    tr = rt_v3(WSF(slot, WF_TRX), WSF(slot, WF_TRY), WSF(slot, WF_TRZ));
                    rng = WS(slot, WF_RNG);
                    bp = WS(slot, WF_BP);
                    bounce = (int)(bp & 0xffffu);
                    const uint32_t segsOfPath = (uint32_t)bounce + 1u; /* this is segment number `bounce` of its path */
                    SceneHit sh_;
                    sh_.dst = WSF(slot, WF_HDST);
                    q_unpack_obj(WS(slot, WF_HOBJ), sh_.obj, sh_.backface);
                    sh_.tri = (int)WS(slot, WF_HTRI);
                    sh_.u = WSF(slot, WF_HU); sh_.v = WSF(slot, WF_HV); sh_.det = WSF(slot, WF_HDET);
                    rt_f3 gained = rt_v3s(0.0f); /* what this segment adds to the path's light: 0 + e == e; only e != +-0 changes the sum */
                    endPath = q_shade<STATS>(a, sh_, rng, o, d, tr, gained, bounce, st);
                    const int rec = (int)((bp >> 16) & 0x7fffu);
                    if (gained.x != 0.0f || gained.y != 0.0f || gained.z != 0.0f) {
                        WP(rec, WP_PLX) = __float_as_uint(__uint_as_float(WP(rec, WP_PLX)) + gained.x);
                        WP(rec, WP_PLY) = __float_as_uint(__uint_as_float(WP(rec, WP_PLY)) + gained.y);
                        WP(rec, WP_PLZ) = __float_as_uint(__uint_as_float(WP(rec, WP_PLZ)) + gained.z);
                    }
                    bp = (bp & 0x7fff0000u) | (uint32_t)(bounce & 0xffff);
                    if (endPath) {
                        WS(slot, WF_RNG) = rng;
                        WS(slot, WF_BP) = bp | RT_WG_PENDING;
                        WP(rec, WP_SEGS) = WP(rec, WP_SEGS) + segsOfPath; /* the pixel's chain length (tile cost), once per path */
                    }
                }
                wq_push(&sh->q[WQ_CAM], act && endPath, slot);
                launch_ray(act && !endPath, slot, o, d, tr, rng, bp);
            } else {
                /* ---- camera batch (which == WQ_CAM: chains whose path ended; WQ_FREE: free slots that become new chains) */
                const RT_CAS KArgs& c = cold_args();
                uint32_t bp = (uint32_t)RT_WG_NO_PIXEL << 16, rng = 0, sample = 0;
                rt_f3 ti = rt_v3s(0.0f);
                bool needPixel = act;
                int rec = -1;
                if (act && which == WQ_CAM) {
                    bp = WS(slot, WF_BP);
                    rec = (int)((bp >> 16) & 0x7fffu);
                    rng = WS(slot, WF_RNG);
                    ti = rt_v3(__uint_as_float(WP(rec, WP_TIX)), __uint_as_float(WP(rec, WP_TIY)), __uint_as_float(WP(rec, WP_TIZ)));
                    sample = WP(rec, WP_SAMPLE);
                    if (bp & RT_WG_PENDING) /* RC:578 */
                        ti = rt_v3(ti.x + __uint_as_float(WP(rec, WP_PLX)), ti.y + __uint_as_float(WP(rec, WP_PLY)), ti.z + __uint_as_float(WP(rec, WP_PLZ)));
                    needPixel = false;
                    if ((int)sample == c.spp) { /* RC:581 + RCC:18-23 */
                        const uint32_t pixLinear = WP(rec, WP_PIXLIN);
                        const int frameNow = (int)WP(rec, WP_FRAME);
                        const size_t pixOff = (size_t)pixLinear * 4;
                        rt_f3 col = ti * c.rcpSpp;
                        if (c.nFrames > 1) {
                            const size_t slab = (size_t)(frameNow - c.frame0) * c.stagingStride;
                            *reinterpret_cast<float4*>(c.staging + (slab + pixLinear) * 4) = make_float4(col.x, col.y, col.z, 1.0f);
                        } else {
                            *reinterpret_cast<float4*>(c.frameRender + pixOff) = make_float4(col.x, col.y, col.z, 1.0f);
                            if (c.accumulate) {
                                float4 acc = *reinterpret_cast<float4*>(c.accumulated + pixOff);
                                acc.x += col.x; acc.y += col.y; acc.z += col.z; acc.w += 1.0f;
                                *reinterpret_cast<float4*>(c.accumulated + pixOff) = acc;
                            }
                        }
                        if (c.tileCost) {
                            const uint32_t prow = pixLinear / c.W, pcol = pixLinear - prow * c.W;
                            uint32_t* const cslot = c.tileCost + (prow >> 3) * (uint32_t)c.tilesX + (pcol >> 3);
                            const uint32_t chain = WP(rec, WP_SEGS);
                            if (chain > *cslot) atomicMax(cslot, chain);
                        }
                        needPixel = true;
                    }
                }
                /* a slot that becomes a chain gets its pixel record (creation) */
                {
                    const bool create = act && which == WQ_FREE;
                    const unsigned long long cm = __ballot(create);
                    if (cm) {
                        uint32_t base = 0;
                        if (lane == __ffsll((long long)cm) - 1) base = atomicAdd(&sh->created, (uint32_t)__popcll(cm));
                        base = __builtin_amdgcn_readlane(base, __ffsll((long long)cm) - 1);
                        if (create) rec = (int)(base + (uint32_t)q_rank(cm));
                    }
                }
                rt_f3 focusPoint = rt_v3s(0.0f);
                bool fresh = false;
                unsigned long long idle = __ballot(needPixel);
                while (idle) {
                    if (poolPos >= 64) {
                        if (queueEmpty) break;
                        int next = 0;
                        if (lane == 0) next = (int)(atomicAdd(c.tileQueue, 1ull) - c.tileQueueBase);
                        next = __builtin_amdgcn_readfirstlane(next);
                        if (next >= c.launchItems) { queueEmpty = true; break; }
                        {
                            const int q_ = next;
                            RTW_ITEM(c, q_, next);
                        }
                        next = next * c.orderStride + c.orderOffset;
                        if (c.tileOrder) next = (int)c.tileOrder[next];
                        RTW_SET_POOL(c, next);
                    }
                    const int rank = q_rank(idle);
                    const int avail = 64 - poolPos;
                    if (needPixel && rank < avail) {
                        phase_mark<STATS>(st, PH_REFILL);
                        const int pslot = poolPos + rank;
                        const int x = poolX0 + (pslot & 7);
                        const int lrow = poolRow0 + (pslot >> 3);
                        if (x < (int)c.W && lrow < c.localRows) {
                            const int y = poolY0 + (pslot >> 3);
                            uint32_t pixelIndex;
                            q_pixel_setup(c, x, y, pixelIndex, focusPoint);
                            WP(rec, WP_FPX) = __float_as_uint(focusPoint.x); WP(rec, WP_FPY) = __float_as_uint(focusPoint.y); WP(rec, WP_FPZ) = __float_as_uint(focusPoint.z);
                            WP(rec, WP_PIXIDX) = pixelIndex;
                            WP(rec, WP_PIXLIN) = (uint32_t)lrow * c.W + (uint32_t)x;
                            WP(rec, WP_SEGS) = 0u;
                            WP(rec, WP_FRAME) = (uint32_t)poolFrame;
                            rng = pixelIndex + (uint32_t)poolFrame * 719393u + (uint32_t)c.seed; /* RC:552 */
                            ti = rt_v3s(0.0f);
                            sample = 0;
                            needPixel = false;
                            fresh = true;
                        }
                    }
                    const int wanted = __popcll(idle);
                    poolPos += wanted < avail ? wanted : avail;
                    idle = __ballot(needPixel);
                }
                /* bookkeeping of live chains: a created chain that got a pixel is born, a chain that finished a pixel and got none dies */
                {
                    const bool born = act && which == WQ_FREE && !needPixel;
                    const bool died = act && which == WQ_CAM && needPixel;
                    const int nb = __popcll(__ballot(born)), nd = __popcll(__ballot(died));
                    if (lane == 0 && nb != nd) atomicAdd(&sh->live, (uint32_t)(nb - nd));
                }
                const bool live = act && !needPixel;
                rt_f3 o = rt_v3s(0.0f), d = rt_v3s(0.0f);
                bool shoot = false, again = false;
                if (live) {
                    bp = (uint32_t)rec << 16;
                    if ((int)sample < c.spp) { /* RC:565-576 */
                        phase_mark<STATS>(st, PH_RAYGEN);
                        if (!fresh) focusPoint = rt_v3(__uint_as_float(WP(rec, WP_FPX)), __uint_as_float(WP(rec, WP_FPY)), __uint_as_float(WP(rec, WP_FPZ)));
                        q_camera_ray(c, focusPoint, rng, o, d);
                        sample++;
                        WP(rec, WP_PLX) = 0u; WP(rec, WP_PLY) = 0u; WP(rec, WP_PLZ) = 0u; /* totalLight = 0, RC:481 */
                        if (c.maxBounce >= 0) shoot = true;
                        else {
                            WS(slot, WF_RNG) = rng; WS(slot, WF_BP) = bp | RT_WG_PENDING;
                            again = true;
                        }
                    } else {
                        WS(slot, WF_RNG) = rng; WS(slot, WF_BP) = bp;
                        again = true;
                    }
                    WP(rec, WP_TIX) = __float_as_uint(ti.x); WP(rec, WP_TIY) = __float_as_uint(ti.y); WP(rec, WP_TIZ) = __float_as_uint(ti.z);
                    WP(rec, WP_SAMPLE) = sample;
                }
                /* a slot without a chain goes back to the free list */
                /* hand the pool tile back and let the other shading wave hand out pixels */
                if (lane == 0) {
                    sh->poolX0 = poolX0; sh->poolRow0 = poolRow0; sh->poolY0 = poolY0; sh->poolPos = poolPos; sh->poolFrame = poolFrame;
                    sh->queueEmpty = queueEmpty ? 1 : 0;
                    if (queueEmpty && poolPos >= 64) sh->tilesExhausted = 1u;
                    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0): the state is written before the lock opens */
                    sh->camLock = 0u;
                }
                wq_push(&sh->q[WQ_FREE], act && !live, slot);
                wq_push(&sh->q[WQ_CAM], again, slot);
                launch_ray(shoot, slot, o, d, rt_v3s(1.0f), rng, bp);
            }
        }
#undef RTW_ITEM
#undef RTW_SET_POOL
    }
#undef WS
#undef WSF
#undef WP

    if (spins >= (1u << 24) || sh->abort) {
        if (lane == 0) atomicAdd(a.counters + (size_t)(blockIdx.x % RT_COUNTER_SLOTS) * RT_COUNTER_FIELDS + 7, 1ull); /* watchdog fired: the host reports it */
    }
    uint32_t segSum = wave_sum(segments);
    unsigned long long* cslot = a.counters + (size_t)((blockIdx.x * RT_WG_WAVES + wave) % RT_COUNTER_SLOTS) * RT_COUNTER_FIELDS;
    if (STATS) {
        uint32_t in = wave_sum(st.inner), lf = wave_sum(st.leaf), tr = wave_sum(st.tri), sp = wave_sum(st.sphere), md = wave_sum(st.model);
        if (lane == 0) {
            atomicAdd(cslot + 0, (unsigned long long)segSum);
            atomicAdd(cslot + 1, (unsigned long long)in);
            atomicAdd(cslot + 2, (unsigned long long)lf);
            atomicAdd(cslot + 3, (unsigned long long)tr);
            atomicAdd(cslot + 4, (unsigned long long)sp);
            atomicAdd(cslot + 5, (unsigned long long)md);
        }
        {
            uint32_t fv = wave_sum(st.filterViolations);
            if (lane == 0 && fv) atomicAdd(cslot + 6, (unsigned long long)fv);
        }
        for (int p = 0; p < RT_N_PHASES; p++) {
            uint32_t e = wave_sum(st.phExec[p]), l = wave_sum(st.phLanes[p]);
            if (lane == 0) {
                atomicAdd(cslot + 8 + 2 * p, (unsigned long long)e);
                atomicAdd(cslot + 9 + 2 * p, (unsigned long long)l);
            }
        }
    } else if (lane == 0) {
        atomicAdd(cslot + 0, (unsigned long long)segSum);
    }
}

template <bool STATS>
__global__ void __launch_bounds__(RT_WG_THREADS, RT_MIN_WAVES_PER_SIMD) rt_trace_wg_kernel(const KArgs a)
{
    trace_body_wg<STATS>(a);
}

} // namespace rtk
#endif
