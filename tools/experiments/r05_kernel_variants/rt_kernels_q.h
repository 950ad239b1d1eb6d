/*
 * rt_kernels_q.h — the QUEUED-STAGES form of the trace kernel (BVH scenes with up to 64 models).
 *
 * Same per-pixel arithmetic as rt_kernels.h (the device functions are shared: begin_intersect, traverse, resolve_hit,
 * the shading helpers), different ownership of rays.  In rt_kernels.h a lane owns ONE pixel chain and the wave runs each
 * kind of work with whatever lanes happen to need it: shading at 0.6 of the lanes, the glass branch at 0.08, camera rays
 * at 0.18, pixel set-up at 0.03 — half of a BVH scene's instructions are issued in those phases
 * (profiles/r02_phase_profile_16_frames_per_launch.txt).  Here a wave owns RT_Q_R x 64 pixel chains ("slots").  A chain
 * between two segments is small — ray, throughput, light, RNG, bounce: the reference's loop-carried state of Trace(),
 * RC:479-542, plus the pixel's running sum — and its traversal stack is empty, so it can wait in a record in device
 * memory (touched once per stage, never per node) while only its one-byte slot number moves through per-wave queues in LDS:
 *
 *     camQ  --[camera batch: finish pixel / next pixel / camera ray RC:545-582, spheres + root filter]-->  rayQ
 *     rayQ  --[the wave's 64 traversal lanes: refilled as they finish, A / B / C majority vote as before]-->  hitQ | glassQ
 *     hitQ, glassQ  --[shade batch of up to 64 hits of one kind, RC:488-538; survivors: spheres + root filter]-->  rayQ,
 *                                                                                          ended paths --> camQ
 *
 * Every batch runs with (up to) all 64 lanes doing the same kind of work; the traversal lanes never wait for shading.
 * Nothing about a chain's own sequence of fp32 operations, RNG draws or visiting order changes (quirk Q13 holds per
 * slot), so images and exact counters are the oracle's bit for bit — the scheduling is the only difference.
 *
 * LDS per wave: the traversal stack [stackEntries][64] as before + four 256-byte rings of slot numbers (they take the place
 * of the four pixel-bookkeeping rows of rt_kernels.h).  Device memory per wave: RT_Q_STRIDE x slots dwords of records,
 * [slot][field], + the parked traversal state of the 64 lanes while a batch runs, [lane][field].
 */
#ifndef RT_KERNELS_Q_H
#define RT_KERNELS_Q_H

#include "../rt_kernels.h"

#ifndef RT_Q_R
#define RT_Q_R 3 /* pixel chains per traversal lane */
#endif
#define RT_Q_SLOTS (RT_WAVE * RT_Q_R)
#if RT_Q_SLOTS > 256
#error "slot numbers are bytes"
#endif

namespace rtk {

/* record of a chain: RT_Q_STRIDE dwords per slot, [slot][field].  Every field is reached with an immediate offset from ONE
 * per-lane byte offset (slot x stride) off the wave-uniform base — a [field][slot] layout needs a base per field, 33 SGPR pairs
 * that spill — and the fields a stage reads or writes together are adjacent, so the accesses merge into dwordx4. */
enum {
    QF_RPX = 0, QF_RPY, QF_RPZ, QF_RDX, QF_RDY, QF_RDZ,          /* ray (RC:35-47 pos, dir) */
    QF_HDST, QF_HOBJ,                                            /* result so far: spheres, then the traversal's */
    QF_CANDLO, QF_CANDHI,                                        /* models that pass the root filter */
    QF_HTRI, QF_HU, QF_HV, QF_HDET,                              /* winning triangle */
    QF_RNG, QF_BOUNCE,                                           /* rngState, i of RC:485 */
    QF_TRX, QF_TRY, QF_TRZ, QF_PLX, QF_PLY, QF_PLZ,              /* ray.transmittance, totalLight of Trace() */
    QF_STATE, QF_SEGS,                                           /* samples started | flags; segments of this pixel-frame */
    QF_TIX, QF_TIY, QF_TIZ,                                      /* totalIncomingLight RC:561 */
    QF_FPX, QF_FPY, QF_FPZ, QF_PIXIDX, QF_PIXLIN, QF_FRAME,      /* per pixel: focus point RC:556, pixelIndex RC:551, ... */
    RT_Q_FIELDS
};
#define RT_Q_STRIDE 36 /* dwords per slot (RT_Q_FIELDS rounded up to a multiple of 4) */
/* parked traversal state of a lane, [lane][field] dwords */
enum {
    QS_CANDLO = 0, QS_CANDHI, QS_M, QS_CUR, QS_SP, QS_LPX, QS_LPY, QS_LPZ, QS_LDX, QS_LDY, QS_LDZ, QS_LIX, QS_LIY, QS_LIZ, QS_TRIBASE,
    QS_HDST, QS_HOBJ, QS_HTRI, QS_HU, QS_HV, QS_HDET, QS_SLOT,
    RT_Q_SAVE_FIELDS
};
#define RT_Q_SAVE_STRIDE 24
#define RT_Q_WAVE_DWORDS (RT_Q_STRIDE * RT_Q_SLOTS + RT_Q_SAVE_STRIDE * RT_WAVE)

#define RT_QSTATE_SAMPLES 0xffffu
#define RT_QSTATE_HAS_PIXEL 0x10000u
#define RT_QSTATE_PENDING 0x20000u /* a finished path's light waits to be added to the pixel's sum (RC:578) */

/* dword at (wave-uniform base) + (32-bit byte offset): "SGPR base + VGPR offset" addressing, no 64-bit VALU arithmetic */
__device__ __forceinline__ uint32_t& q_at(uint32_t* base, uint32_t byteOff)
{
    return *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(base) + byteOff);
}
#define QR(f) q_at(rec, so + (uint32_t)(f) * 4u)
#define QSO(slot) ((uint32_t)(slot) * (RT_Q_STRIDE * 4u))
#define QRF(f) __uint_as_float(QR(f))
#define QRSET(f, v) QR(f) = (v)
#define QRSETF(f, v) QR(f) = __float_as_uint(v)

__device__ __forceinline__ uint32_t q_pack_obj(int obj, bool backface) { return obj < 0 ? 0xffffffffu : ((uint32_t)obj | (backface ? 0x40000000u : 0u)); }
__device__ __forceinline__ void q_unpack_obj(uint32_t w, int& obj, bool& backface)
{
    obj = w == 0xffffffffu ? -1 : (int)(w & 0x3fffffffu);
    backface = w != 0xffffffffu && (w & 0x40000000u) != 0;
}

/* number of set bits of m below this lane (v_mbcnt: no 64-bit lane mask to keep around) */
__device__ __forceinline__ int q_rank(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

/* one per-wave ring of slot numbers in LDS (only its own wave touches it: no atomics, no barriers) */
struct QRing {
    uint8_t* ring; /* 256 bytes, wave-uniform */
    int head, tail; /* free-running, wave-uniform */
    __device__ __forceinline__ int count() const { return tail - head; }
    /* lanes with pred append their slot, in lane order */
    __device__ __forceinline__ void push(bool pred, int slot)
    {
        const unsigned long long m = __ballot(pred);
        if (pred) ring[(tail + q_rank(m)) & 255] = (uint8_t)slot;
        tail += __popcll(m);
    }
    /* entry head + k (k < count()), without consuming */
    __device__ __forceinline__ int peek(int k) const { return (int)ring[(head + k) & 255]; }
};

/* RCC:15 + RC:550-556 for pixel (x, y): seed index and focus point */
__device__ __forceinline__ void q_pixel_setup(const RT_CAS KArgs& c, int x, int y, uint32_t& pixelIndex, rt_f3& focusPoint)
{
    const float uvx = (float)(uint32_t)x * c.rcpWm1;
    const float uvy = (float)(uint32_t)y * c.rcpHm1;
    const uint32_t pixelCoordX = (uint32_t)(uvx * (float)c.W);
    const uint32_t pixelCoordY = (uint32_t)(uvy * (float)c.H);
    pixelIndex = pixelCoordY * c.W + pixelCoordX;
    const rt_f3 fpl = rt_v3(uvx - 0.5f, uvy - 0.5f, 1.0f) * rt_v3(c.viewParams[0], c.viewParams[1], c.viewParams[2]);
    float cam[16];
    for (int k = 0; k < 16; k++) cam[k] = c.cam[k];
    focusPoint = rt_mul_point(cam, fpl, 1.0f);
}
/* RC:565-576: the next camera ray of a pixel (see trace_body for the no-defocus shortcut) */
__device__ __forceinline__ void q_camera_ray(const RT_CAS KArgs& c, rt_f3 focusPoint, uint32_t& rng, rt_f3& rpos, rt_f3& rdir)
{
    float cam[16];
    for (int k = 0; k < 16; k++) cam[k] = c.cam[k];
    const rt_f3 camOrigin = rt_mul_point(cam, rt_v3(0.0f, 0.0f, 0.0f), 1.0f);
    const rt_f3 camRight = rt_v3(cam[0], cam[1], cam[2]);
    const rt_f3 camUp = rt_v3(cam[4], cam[5], cam[6]);
    const float invNumPixelsX = c.rcpW;
    rt_f3 rayOrigin;
    if (c.raygenNoDefocus) {
        rt_next_random(&rng);
        rt_next_random(&rng);
        rayOrigin = camOrigin;
    } else {
        rt_f2 dj = rand_circle(&rng);
        rayOrigin = camOrigin + camRight * (dj.x * c.defocus * invNumPixelsX) + camUp * (dj.y * c.defocus * invNumPixelsX);
    }
    rt_f2 jj = rand_circle(&rng);
    rt_f3 jfp = focusPoint + camRight * (jj.x * c.diverge * invNumPixelsX) + camUp * (jj.y * c.diverge * invNumPixelsX);
    rpos = rayOrigin;
    rdir = rt_normalize(jfp - rayOrigin);
}
/* One iteration of Trace's bounce loop after the intersection (RC:488-538), as in trace_body.  Returns true when the path ends. */
template <bool STATS>
__device__ __forceinline__ bool q_shade(const KArgs& a, const SceneHit& h, uint32_t& rng, rt_f3& rpos, rt_f3& rdir, rt_f3& transmittance,
                                        rt_f3& pathLight, int& bounce, Stats& st)
{
    if (h.obj < 0) {
        phase_mark<STATS>(st, PH_SKY);
        const RT_CAS KArgs& c = cold_args();
        if (c.useSky) pathLight = pathLight + transmittance * environment_light(c, rdir);
        return true;
    }
    phase_mark<STATS>(st, PH_SHADE_HIT);
    rt_f3 hpos, normal;
    resolve_hit(a, rpos, rdir, h, hpos, normal);
    const DMaterial mat = a.materials[h.obj];
    const bool isGlass = mat.flag == RT_MATERIAL_GLASS;
    float uSpec = 0.0f;
    if (!isGlass) uSpec = rt_random_value(&rng); /* RC:521 */
    const rt_f3 diffuseDir = rt_normalize(normal + rand_direction(&rng)); /* RC:509 / RC:525 */
    const rt_f3 specularDir = rt_reflect(rdir, normal);
    rt_f3 lerpA = diffuseDir, lerpB = specularDir;
    float lerpT;
    if (isGlass) {
        phase_mark<STATS>(st, PH_GLASS);
        if (h.backface) { /* RC:502 */
            rt_f3 e = ((-h.dst) * rt_v3(mat.absorption[0], mat.absorption[1], mat.absorption[2])) * mat.absorptionStrength;
            transmittance = transmittance * rt_v3(rt_exp(e.x), rt_exp(e.y), rt_exp(e.z));
        }
        float iorCurrent = h.backface ? mat.ior : 1.0f;
        float iorNext = h.backface ? 1.0f : mat.ior;
        const rt_f3 refractDir = refract_dir(rdir, normal, iorCurrent, iorNext);
        const float reflectWeight = reflectance(rdir, normal, iorCurrent, iorNext);
        const bool followReflection = rt_random_value(&rng) <= reflectWeight; /* RC:515 */
        lerpT = mat.specularProbability;
        if (!followReflection) {
            lerpA = -diffuseDir;
            lerpB = refractDir;
            lerpT = mat.smoothness;
        }
    } else {
        const bool isSpecular = mat.specularProbability >= uSpec;
        lerpT = mat.smoothness * (isSpecular ? 1.0f : 0.0f);
        rt_f3 emitted = rt_v3(mat.emissionCol[0], mat.emissionCol[1], mat.emissionCol[2]) * mat.emissionStrength;
        pathLight = pathLight + emitted * transmittance;
        transmittance = transmittance * material_colour(mat, hpos, normal, isSpecular);
    }
    rdir = rt_normalize(rt_lerp3(lerpA, lerpB, lerpT));
    rpos = isGlass ? hpos + (0.001f * normal) * rt_sign(rt_dot(normal, rdir)) : hpos + (normal * 0.001f);
    float p = rt_max(transmittance.x, rt_max(transmittance.y, transmittance.z)); /* RC:535-538 */
    if (rt_random_value(&rng) >= p) return true;
    transmittance = transmittance * rt_rcp(p);
    bounce++;
    return bounce > a.maxBounce; /* RC:485: i <= MaxBounceCount */
}

template <bool STATS>
__device__ __forceinline__ void trace_body_q(const KArgs& a)
{
    extern __shared__ uint32_t s_stack[];
    const int lane = threadIdx.x;
    uint32_t* const stackBase = &s_stack[lane];
    uint32_t* rec;       /* this wave's records, [field][slot] */
    uint32_t* park;      /* parked traversal state, [field][lane] */
    QRing rayQ, hitQ, glassQ, camQ;
    int flushMin, refillMin, starveMin;
    {
        const RT_CAS KArgs& c = cold_args();
        rec = c.qRecords + (size_t)blockIdx.x * RT_Q_WAVE_DWORDS;
        park = rec + RT_Q_STRIDE * RT_Q_SLOTS;
        uint8_t* rings = reinterpret_cast<uint8_t*>(s_stack + (size_t)c.stackEntries * RT_WAVE);
        rayQ.ring = rings; hitQ.ring = rings + 256; glassQ.ring = rings + 512; camQ.ring = rings + 768;
        flushMin = c.qFlushMin;
        refillMin = c.qRefillMin;
        starveMin = c.qStarveMin;
    }
    rayQ.head = rayQ.tail = hitQ.head = hitQ.tail = glassQ.head = glassQ.tail = 0;
    /* every slot starts without a pixel, waiting for the camera stage */
    for (int k = 0; k < RT_Q_R; k++) {
        camQ.ring[lane + RT_WAVE * k] = (uint8_t)(lane + RT_WAVE * k);
        q_at(rec, QSO(lane + RT_WAVE * k) + (uint32_t)QF_STATE * 4u) = 0u;
    }
    camQ.head = 0;
    camQ.tail = RT_Q_SLOTS;

    /* the wave's pool tile (as in trace_body): next unassigned pixels of the current (tile, frame) item */
    int poolX0 = 0, poolRow0 = 0, poolY0 = 0, poolPos = 64, poolFrame = 0;
    bool queueEmpty;
#define RTQ_ITEM(c, q, tilePos)                                                                               \
    do {                                                                                                      \
        if ((c).nFrames > 1) { tilePos = (q) / (c).frameGroups; poolFrame = (c).frame0 + ((q) - tilePos * (c).frameGroups) * (c).frameGroup; } \
        else { tilePos = (q); poolFrame = (c).frame0; }                                                       \
    } while (0)
#define RTQ_SET_POOL(c, tile)                                                                                 \
    do {                                                                                                      \
        const int ty_ = (tile) / (c).tilesX;                                                                  \
        poolX0 = ((tile) - ty_ * (c).tilesX) * 8;                                                             \
        poolRow0 = ty_ * 8;                                                                                   \
        const int ls_ = poolRow0 / (c).stripRows;                                                             \
        poolY0 = (ls_ * (c).partCount + (c).partIndex) * (c).stripRows + (poolRow0 - ls_ * (c).stripRows);    \
        poolPos = 0;                                                                                          \
    } while (0)
    {
        const RT_CAS KArgs& c = cold_args();
        int tile = (int)blockIdx.x;
        if (tile < c.launchItems && c.nFrames > 0 && !c.queueStart) {
            const int q0_ = tile;
            RTQ_ITEM(c, q0_, tile);
            tile = tile * c.orderStride + c.orderOffset;
            if (c.tileOrder) tile = (int)c.tileOrder[tile];
            RTQ_SET_POOL(c, tile);
        }
        queueEmpty = (c.nFrames <= 0);
    }

    /* the wave's traversal lanes */
    bool busy = false;
    int mySlot = 0;
    rt_f3 rpos = rt_v3s(0.0f), rdir = rt_v3s(0.0f);
    SceneHit h;
    Trav t;
    h.dst = RT_INF; h.obj = -1; h.tri = -1; h.u = h.v = h.det = 0.0f; h.backface = false;
    t.cand = 0; t.rootStep = false; t.m = 0; t.cur = RT_CODE_DONE; t.sp = 0; t.lpos = t.ldir = t.linv = rt_v3s(0.0f); t.triBase = 0; t.cull = true;
    uint32_t segments = 0;
    Stats st = {};
#ifdef RT_PHASE_TIMES
    st.phPrev = -1;
#endif

    /* a hit goes to the queue of its kind: glass (RC:499-518) or everything else (opaque RC:519-533, sky RC:489-495) */
    auto push_hit = [&](bool pred, int slot, int obj) {
        bool glass = false;
        if (pred && obj >= 0) glass = a.materials[obj].flag == RT_MATERIAL_GLASS;
        glassQ.push(pred && glass, slot);
        hitQ.push(pred && !glass, slot);
    };
    /* traversal lanes <-> park: a batch needs the registers, so the 64 lanes' traversal state waits in device memory meanwhile.
     * Every lane stores and loads (a free lane's values are never used) so that nothing stays live across the batch; with no
     * busy lane at all the state is simply reset. */
    auto park_lanes = [&]() -> bool {
        const bool any = __ballot(busy) != 0ull;
        if (!any) return false;
        const uint32_t lo = (uint32_t)lane * (RT_Q_SAVE_STRIDE * 4u);
#define PK(f, v) q_at(park, lo + (uint32_t)(f) * 4u) = (v)
        PK(QS_CANDLO, (uint32_t)t.cand); PK(QS_CANDHI, (uint32_t)(t.cand >> 32));
        PK(QS_M, ((uint32_t)(t.m + 1) & 0xffffu) | (t.rootStep ? 0x80000000u : 0u) | (t.cull ? 0x40000000u : 0u));
        PK(QS_CUR, t.cur); PK(QS_SP, (uint32_t)t.sp);
        PK(QS_LPX, __float_as_uint(t.lpos.x)); PK(QS_LPY, __float_as_uint(t.lpos.y)); PK(QS_LPZ, __float_as_uint(t.lpos.z));
        PK(QS_LDX, __float_as_uint(t.ldir.x)); PK(QS_LDY, __float_as_uint(t.ldir.y)); PK(QS_LDZ, __float_as_uint(t.ldir.z));
        PK(QS_LIX, __float_as_uint(t.linv.x)); PK(QS_LIY, __float_as_uint(t.linv.y)); PK(QS_LIZ, __float_as_uint(t.linv.z));
        PK(QS_TRIBASE, (uint32_t)t.triBase);
        PK(QS_HDST, __float_as_uint(h.dst)); PK(QS_HOBJ, q_pack_obj(h.obj, h.backface)); PK(QS_HTRI, (uint32_t)h.tri);
        PK(QS_HU, __float_as_uint(h.u)); PK(QS_HV, __float_as_uint(h.v)); PK(QS_HDET, __float_as_uint(h.det));
        PK(QS_SLOT, (uint32_t)mySlot);
#undef PK
        return true;
    };
    auto unpark_lanes = [&](bool parked) {
        if (!parked) {
            h.dst = RT_INF; h.obj = -1; h.tri = -1; h.u = h.v = h.det = 0.0f; h.backface = false;
            t.cand = 0; t.rootStep = false; t.m = 0; t.cur = RT_CODE_DONE; t.sp = 0; t.lpos = t.ldir = t.linv = rt_v3s(0.0f); t.triBase = 0; t.cull = true;
            mySlot = 0;
            rpos = rdir = rt_v3s(0.0f);
            return;
        }
        const uint32_t lo = (uint32_t)lane * (RT_Q_SAVE_STRIDE * 4u);
#define UP(f) q_at(park, lo + (uint32_t)(f) * 4u)
        t.cand = (unsigned long long)UP(QS_CANDLO) | ((unsigned long long)UP(QS_CANDHI) << 32);
        const uint32_t mw = UP(QS_M);
        t.m = (int)(mw & 0xffffu) - 1;
        t.rootStep = (mw & 0x80000000u) != 0;
        t.cull = (mw & 0x40000000u) != 0;
        t.cur = UP(QS_CUR); t.sp = (int)UP(QS_SP);
        t.lpos = rt_v3(__uint_as_float(UP(QS_LPX)), __uint_as_float(UP(QS_LPY)), __uint_as_float(UP(QS_LPZ)));
        t.ldir = rt_v3(__uint_as_float(UP(QS_LDX)), __uint_as_float(UP(QS_LDY)), __uint_as_float(UP(QS_LDZ)));
        t.linv = rt_v3(__uint_as_float(UP(QS_LIX)), __uint_as_float(UP(QS_LIY)), __uint_as_float(UP(QS_LIZ)));
        t.triBase = (int)UP(QS_TRIBASE);
        h.dst = __uint_as_float(UP(QS_HDST));
        q_unpack_obj(UP(QS_HOBJ), h.obj, h.backface);
        h.tri = (int)UP(QS_HTRI);
        h.u = __uint_as_float(UP(QS_HU)); h.v = __uint_as_float(UP(QS_HV)); h.det = __uint_as_float(UP(QS_HDET));
        mySlot = (int)(UP(QS_SLOT) & 0xffu);
#undef UP
        const uint32_t so = QSO(mySlot); /* the world ray is still in the record */
        rpos = rt_v3(QRF(QF_RPX), QRF(QF_RPY), QRF(QF_RPZ));
        rdir = rt_v3(QRF(QF_RDX), QRF(QF_RDY), QRF(QF_RDZ));
    };
    /* spheres + root filter for a new ray (the first half of CalculateRayCollision, RC:335-346), then into the record and
     * on to the traversal lanes — or straight to shading when no model can be hit */
    auto launch_ray = [&](bool pred, int slot, uint32_t so, rt_f3 o, rt_f3 d, rt_f3 tr, rt_f3 pl, uint32_t rng, int bounce) {
        SceneHit bh;
        Trav bt;
        bh.dst = RT_INF; bh.obj = -1; bh.backface = false;
        bt.cand = 0;
        if (pred) {
            phase_mark<STATS>(st, PH_SPHERES);
            begin_intersect<STATS, false, false>(a, o, d, stackBase, bh, bt, st);
            segments++;
            QRSETF(QF_RPX, o.x); QRSETF(QF_RPY, o.y); QRSETF(QF_RPZ, o.z);
            QRSETF(QF_RDX, d.x); QRSETF(QF_RDY, d.y); QRSETF(QF_RDZ, d.z);
            QRSETF(QF_TRX, tr.x); QRSETF(QF_TRY, tr.y); QRSETF(QF_TRZ, tr.z);
            QRSETF(QF_PLX, pl.x); QRSETF(QF_PLY, pl.y); QRSETF(QF_PLZ, pl.z);
            QRSET(QF_RNG, rng); QRSET(QF_BOUNCE, (uint32_t)bounce);
            QRSETF(QF_HDST, bh.dst); QRSET(QF_HOBJ, q_pack_obj(bh.obj, bh.backface));
            QRSET(QF_CANDLO, (uint32_t)bt.cand); QRSET(QF_CANDHI, (uint32_t)(bt.cand >> 32));
            QRSET(QF_HTRI, 0xffffffffu); QRSET(QF_HU, 0u); QRSET(QF_HV, 0u); QRSET(QF_HDET, 0u);
            QRSET(QF_SEGS, QR(QF_SEGS) + 1u);
        }
        /* with no candidate among the (<= 64) models the traversal has nothing to do: the sphere result is the result */
        const bool toTrav = pred && (bt.cand != 0ull || a.nModels > 64);
        rayQ.push(toTrav, slot);
        push_hit(pred && !toTrav, slot, bh.obj);
    };

    /* ---- shade batch: up to 64 hits of one kind (the caller took n slot numbers off hitQ or glassQ) */
    auto shade_batch = [&](int n, int slot) {
        const bool act = lane < n;
        const uint32_t so = QSO(slot);
        const bool parked = park_lanes();
        rt_f3 o = rt_v3s(0.0f), d = rt_v3s(0.0f), tr = rt_v3s(0.0f), pl = rt_v3s(0.0f);
        uint32_t rng = 0;
        int bounce = 0;
        bool endPath = false;
        if (act) {
            o = rt_v3(QRF(QF_RPX), QRF(QF_RPY), QRF(QF_RPZ));
            d = rt_v3(QRF(QF_RDX), QRF(QF_RDY), QRF(QF_RDZ));
            tr = rt_v3(QRF(QF_TRX), QRF(QF_TRY), QRF(QF_TRZ));
            pl = rt_v3(QRF(QF_PLX), QRF(QF_PLY), QRF(QF_PLZ));
            rng = QR(QF_RNG);
            bounce = (int)QR(QF_BOUNCE);
            SceneHit sh;
            sh.dst = QRF(QF_HDST);
            q_unpack_obj(QR(QF_HOBJ), sh.obj, sh.backface);
            sh.tri = (int)QR(QF_HTRI);
            sh.u = QRF(QF_HU); sh.v = QRF(QF_HV); sh.det = QRF(QF_HDET);
            endPath = q_shade<STATS>(a, sh, rng, o, d, tr, pl, bounce, st);
            if (endPath) { /* Trace returns: its light waits for the camera stage (RC:578), the RNG state runs on (Q13) */
                QRSETF(QF_PLX, pl.x); QRSETF(QF_PLY, pl.y); QRSETF(QF_PLZ, pl.z);
                QRSET(QF_RNG, rng);
                QRSET(QF_STATE, QR(QF_STATE) | RT_QSTATE_PENDING);
            }
        }
        camQ.push(act && endPath, slot);
        launch_ray(act && !endPath, slot, so, o, d, tr, pl, rng, bounce);
        unpark_lanes(parked);
    };

    /* ---- camera batch: up to 64 slots whose path ended or that have no pixel.  Returns false when none of them could go on. */
    auto camera_batch = [&]() {
        const int n = camQ.count() < RT_WAVE ? camQ.count() : RT_WAVE;
        const bool act = lane < n;
        const int slot = act ? camQ.peek(lane) : 0;
        camQ.head += n;
        const uint32_t so = QSO(slot);
        const bool parked = park_lanes();
        const RT_CAS KArgs& c = cold_args();
        uint32_t state = act ? QR(QF_STATE) : 0u;
        uint32_t rng = 0;
        rt_f3 ti = rt_v3s(0.0f);
        bool needPixel = act && !(state & RT_QSTATE_HAS_PIXEL);
        if (act && (state & RT_QSTATE_HAS_PIXEL)) {
            rng = QR(QF_RNG);
            ti = rt_v3(QRF(QF_TIX), QRF(QF_TIY), QRF(QF_TIZ));
            if (state & RT_QSTATE_PENDING) { /* RC:578: totalIncomingLight += Trace(...) */
                ti = rt_v3(ti.x + QRF(QF_PLX), ti.y + QRF(QF_PLY), ti.z + QRF(QF_PLZ));
                state &= ~RT_QSTATE_PENDING;
            }
            if ((int)(state & RT_QSTATE_SAMPLES) == c.spp) { /* RC:581 + RCC:18-23: this frame of this pixel is finished */
                const uint32_t pixLinear = QR(QF_PIXLIN);
                const int frameNow = (int)QR(QF_FRAME);
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
                    const uint32_t chain = QR(QF_SEGS);
                    if (chain > *cslot) atomicMax(cslot, chain);
                }
                state = 0u;
                needPixel = true;
            }
        }
        /* hand the next unassigned pixels of the pool tile to the slots that need one (ballot + prefix rank, as in trace_body) */
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
                    RTQ_ITEM(c, q_, next);
                }
                next = next * c.orderStride + c.orderOffset;
                if (c.tileOrder) next = (int)c.tileOrder[next];
                RTQ_SET_POOL(c, next);
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
                    QRSETF(QF_FPX, focusPoint.x); QRSETF(QF_FPY, focusPoint.y); QRSETF(QF_FPZ, focusPoint.z);
                    QRSET(QF_PIXIDX, pixelIndex);
                    QRSET(QF_PIXLIN, (uint32_t)lrow * c.W + (uint32_t)x);
                    QRSET(QF_SEGS, 0u);
                    QRSET(QF_FRAME, (uint32_t)poolFrame);
                    rng = pixelIndex + (uint32_t)poolFrame * 719393u + (uint32_t)c.seed; /* RC:552 */
                    ti = rt_v3s(0.0f);
                    state = RT_QSTATE_HAS_PIXEL;
                    needPixel = false;
                    fresh = true;
                }
            }
            const int wanted = __popcll(idle);
            poolPos += wanted < avail ? wanted : avail;
            idle = __ballot(needPixel);
        }
        /* slots that got no pixel: the launch has none left for them — they leave the queues for good */
        const bool live = act && (state & RT_QSTATE_HAS_PIXEL);
        rt_f3 o = rt_v3s(0.0f), d = rt_v3s(0.0f);
        bool shoot = false, again = false;
        if (live) {
            const int sample = (int)(state & RT_QSTATE_SAMPLES);
            if (sample < c.spp) { /* RC:565-576: the pixel's next camera ray */
                phase_mark<STATS>(st, PH_RAYGEN);
                if (!fresh) focusPoint = rt_v3(QRF(QF_FPX), QRF(QF_FPY), QRF(QF_FPZ));
                q_camera_ray(c, focusPoint, rng, o, d);
                state = (state & ~RT_QSTATE_SAMPLES) | (uint32_t)(sample + 1);
                if (c.maxBounce >= 0) shoot = true; /* RC:485: the loop runs for i = 0 */
                else { /* Trace returns 0 at once (RC:578 adds it) */
                    QRSETF(QF_PLX, 0.0f); QRSETF(QF_PLY, 0.0f); QRSETF(QF_PLZ, 0.0f);
                    QRSET(QF_RNG, rng);
                    state |= RT_QSTATE_PENDING;
                    again = true;
                }
            } else {
                again = true; /* spp == 0: finished before it began */
                QRSET(QF_RNG, rng);
            }
            QRSETF(QF_TIX, ti.x); QRSETF(QF_TIY, ti.y); QRSETF(QF_TIZ, ti.z);
        }
        if (act) QRSET(QF_STATE, state);
        camQ.push(again, slot);
        launch_ray(shoot, slot, so, o, d, rt_v3s(1.0f), rt_v3s(0.0f), rng, 0);
        unpark_lanes(parked);
    };

    enum { W_TRAV = 0, W_HIT, W_GLASS, W_CAM };
    for (;;) {
        phase_mark<STATS>(st, PH_LOOP);
        const int nBusy = __popcll(__ballot(busy));
        const int nh = hitQ.count(), ng = glassQ.count(), nc = camQ.count(), nr = rayQ.count();
        int which;
        if (nh >= RT_WAVE) which = W_HIT;
        else if (ng >= RT_WAVE) which = W_GLASS;
        else if (nc >= RT_WAVE) which = W_CAM;
        else if (nBusy + nr >= starveMin || nh + ng + nc == 0) {
            if (nBusy + nr == 0) break; /* no chain left in this wave */
            which = W_TRAV;
        } else { /* the traversal lanes are running dry: the fullest partial batch goes now */
            which = (nh >= ng && nh >= nc) ? W_HIT : (ng >= nc ? W_GLASS : W_CAM);
        }
        if (which == W_TRAV) {
            /* refill free traversal lanes from rayQ */
            const int nFree = RT_WAVE - nBusy;
            if (nr && (nFree >= refillMin || nBusy == 0)) {
                const unsigned long long freeMask = ~__ballot(busy);
                const int n = nFree < nr ? nFree : nr;
                const int rank = q_rank(freeMask);
                if (!busy && rank < n) {
                    mySlot = rayQ.peek(rank);
                    const uint32_t so = QSO(mySlot);
                    rpos = rt_v3(QRF(QF_RPX), QRF(QF_RPY), QRF(QF_RPZ));
                    rdir = rt_v3(QRF(QF_RDX), QRF(QF_RDY), QRF(QF_RDZ));
                    h.dst = QRF(QF_HDST);
                    q_unpack_obj(QR(QF_HOBJ), h.obj, h.backface);
                    h.tri = -1; h.u = h.v = h.det = 0.0f;
                    t.cand = (unsigned long long)QR(QF_CANDLO) | ((unsigned long long)QR(QF_CANDHI) << 32);
                    t.m = -1; t.cur = RT_CODE_NEXT_MODEL; t.sp = 0; t.rootStep = false;
                    t.lpos = t.ldir = t.linv = rt_v3s(0.0f); t.triBase = 0; t.cull = true;
                    busy = true;
                }
                rayQ.head += n;
            }
            /* traverse until flushMin more lanes have finished (or everybody has) */
            const int nNow = __popcll(__ballot(busy));
            bool done = false;
            if (busy) done = traverse<STATS, true, false>(a, rpos, rdir, stackBase, stackBase, h, t, st, nNow > flushMin ? nNow - flushMin : 0);
            /* finished lanes hand their result over and become free */
            const bool fin = busy && done;
            if (fin) {
                const uint32_t so = QSO(mySlot);
                QRSETF(QF_HDST, h.dst); QRSET(QF_HOBJ, q_pack_obj(h.obj, h.backface)); QRSET(QF_HTRI, (uint32_t)h.tri);
                QRSETF(QF_HU, h.u); QRSETF(QF_HV, h.v); QRSETF(QF_HDET, h.det);
            }
            push_hit(fin, mySlot, h.obj);
            if (fin) busy = false;
        } else if (which == W_CAM) {
            camera_batch();
        } else {
            QRing& src = which == W_HIT ? hitQ : glassQ;
            const int n = src.count() < RT_WAVE ? src.count() : RT_WAVE;
            const int slot = lane < n ? src.peek(lane) : 0;
            src.head += n;
            shade_batch(n, slot);
        }
    }
#undef RTQ_ITEM
#undef RTQ_SET_POOL

    uint32_t segSum = wave_sum(segments);
    unsigned long long* cslot = a.counters + (size_t)(blockIdx.x % RT_COUNTER_SLOTS) * RT_COUNTER_FIELDS;
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
#ifdef RT_PHASE_TIMES
                atomicAdd(cslot + 8 + 2 * RT_N_PHASES + p, (unsigned long long)st.phT[p]);
#endif
            }
        }
    } else if (lane == 0) {
        atomicAdd(cslot + 0, (unsigned long long)segSum);
    }
}

#ifndef RT_MIN_WAVES_PER_SIMD_Q
#define RT_MIN_WAVES_PER_SIMD_Q RT_MIN_WAVES_PER_SIMD
#endif
template <bool STATS>
__global__ void __launch_bounds__(RT_WAVE, RT_MIN_WAVES_PER_SIMD_Q) rt_trace_q_kernel(const KArgs a)
{
    trace_body_q<STATS>(a);
}
template <bool STATS>
__global__ void __launch_bounds__(RT_WAVE, RT_MIN_WAVES_PER_SIMD_Q) rt_trace_q_half_kernel(const KArgs a)
{
    trace_body_q<STATS>(a);
}

} // namespace rtk
#endif
