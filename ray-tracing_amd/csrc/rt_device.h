/*
 * rt_device.h — HBM-resident scene layout and kernel argument block of
 * libraytrace_hip.so (internal; the boundary layouts are in include/rt_abi.h).
 *
 * The boundary hands us the reference's structured buffers (72-B triangles,
 * 32-B nodes, 224-B models).  rt_upload_scene re-lays them out once for the
 * traversal kernel; results do not change because every derived value is
 * computed with the same fp32 operation the reference's per-ray code performs
 * (RayCommon.hlsl:190-192 are ray-independent):
 *
 *  DPair (64 B, 16-B aligned): both children of one inner node — the only
 *      thing RayTriangleBVH (RC:262-282) reads per inner step — with each
 *      child's (startIndex, triangleCount) pre-decoded into a 32-bit "code",
 *      so a popped entry never re-reads its own node (reference: 96 B per
 *      inner step; here: one 64-B record).  Where the records lie — every one
 *      is named by the 16-byte unit it starts at — is rt_layout.h's choice.
 *  DTri (48 B = 3 x float4): posA, edgeAB, edgeAC, cross(edgeAB, edgeAC).
 *  DTriN (36 B): the three vertex normals, read once per segment for the
 *      winning triangle only (the reference normalises a normal per test and
 *      discards all but the winner's).
 *  DModel (128 B): worldToLocal / localToWorld as three 4-float rows each,
 *      root code, triangle base, cull flag — read with scalar loads.
 *  DMaterial (96 B): RtMaterial padded to 6 x float4.
 */
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <stdint.h>

#define RT_WAVE 64
#define RT_STACK_DEPTH 34            /* >= RT_MAX_BVH_DEPTH + 2 */
#define RT_COUNTER_SLOTS 1024        /* counters are spread over slots to avoid same-address atomics */
#define RT_PIXEL_FIELDS 4
#define RT_N_PHASES 12
#define RT_MAX_WAVES_PER_GROUP 12      /* the BVH trace kernels' workgroups: up to 12 waves share one LDS top-of-tree cache (rt_kernels.h) */
/* the FLAT trace kernel's workgroups (round 6): up to 16 waves share one LDS CHAIN POOL (rt_kernels.h, pool_exchange): two queues (chains
 * waiting for the sky phase / for the shade phase) of RT_POOL_CELLS cells each (a power of two; 32 measured 10 % slower than 64 on the headline
 * scene, 128 the same as 64: profiles/r06_chain_pool.txt); a cell = RT_POOL_QUADS x 16 bytes of chain state.
 * LDS of a workgroup: [header: RT_POOL_HEADER_DWORDS][seq: 2 x RT_POOL_CELLS dwords][payload: 2 x RT_POOL_QUADS x RT_POOL_CELLS x 16 B][wave regions] */
#define RT_MAX_WAVES_PER_GROUP_FLAT 16
#define RT_POOL_CELLS 64u
#define RT_POOL_QUADS 8u
#define RT_POOL_HEADER_DWORDS 16u
#define RT_POOL_DWORDS (RT_POOL_HEADER_DWORDS + 2u * RT_POOL_CELLS + 2u * RT_POOL_QUADS * RT_POOL_CELLS * 4u)
/* bytes of a wave's record in KArgs::pxCold: two float4 per lane (+ the traversal stack in the RT_GLOBAL_STACK experiment) */
#define RT_COLD_STRIDE_BYTES (2 * RT_WAVE * 16)
#define RT_COUNTER_FIELDS (8 + 2 * RT_N_PHASES + 4) /* ... + hot-cache steps, node-uniform steps (>= 48 lanes, >= 3/4 of the active lanes) */

/* Every record of the traversal is named by the 16-byte UNIT it starts at (rt_layout.h decides where the records lie):
 * node codes: bit31 = leaf.  leaf: [30:24] = triangle count (1..127), [23:0] = first unit of the leaf's run of DTri records
 * (three units each) relative to the model's triBase; count field 0 = indirect, [23:0] indexes bigLeaves {unit, count}.
 * inner: [30:0] = unit of the DPair in the pair space. */
#define RT_CODE_LEAF 0x80000000u
#define RT_CODE_NEXT_MODEL 0x7fffffffu /* traversal state: this model is finished */
#define RT_CODE_DONE 0x7ffffffeu       /* traversal state: every model visited (inner codes are below this) */
#define RT_CODE_MAX_INLINE_COUNT 127
#define RT_CODE_MAX_INLINE_START 0x00ffffffu

struct DPair {
    float aMin[3], aMax[3];
    float bMin[3], bMax[3];
    uint32_t codeA, codeB;
    uint32_t pad[2];
};
#define RT_PAIR_FORMAT 0
struct DTri {
    float ax, ay, az, abx;
    float aby, abz, acx, acy;
    float acz, fx, fy, fz;
};
struct DTriN {
    float n[9];
};
/* the kernels address these records with shifted 32-bit byte offsets (rt_kernels.h: unit << 4) */
static_assert(sizeof(DPair) == 64 && sizeof(DTri) == 48 && sizeof(DTriN) == 36, "rt_kernels.h hard-codes the record sizes");
struct DModel {
    float w2l[12]; /* row r: m[r], m[4+r], m[8+r], m[12+r] of worldToLocal */
    float l2w[12];
    uint32_t rootCode;    /* 16-B aligned tail: (rootCode, triBase, cullBackface, -) */
    int32_t triBase;      /* first unit of the model's triangles in the triangle space */
    int32_t cullBackface; /* material.flag != GLASS (RC:355) */
    int32_t pad[5];
};
/* Conservative world-space stand-in for a model's root step (see begin_intersect): the union of
 * the root's two child boxes, transformed to world space and inflated; `always` = no filtering
 * (leaf root, or a matrix that cannot be inverted robustly). 32 B, scalar-loaded. */
struct DFilter {
    float bMin[3], bMax[3];
    uint32_t always;
    uint32_t innerRoot;
};
/* Two-level model hierarchy for scenes with more than 64 models (the "TLAS" of SURVEY.md §8(f)): models
 * are clustered in space (Morton order of their filter boxes) into chunks of up to 16; a chunk carries the
 * union of its members' filter boxes.  The lockstep filter first tests the chunk box and skips all 16 members
 * when no lane of the wave hits it.  Members are visited later in MODEL-INDEX order (bit masks), so the
 * clustering never changes results.  96 B, scalar-loaded. */
#define RT_CHUNK_MODELS 16
struct DChunk {
    float bMin[3], bMax[3];
    uint32_t always;      /* a member cannot be filtered: the chunk box is meaningless */
    uint32_t count;
    uint32_t innerRoots;  /* members whose root is an inner node (exact counters of skipped chunks) */
    uint32_t pad[3];
    uint32_t members[RT_CHUNK_MODELS];
};
struct DMaterial {
    float diffuseCol[4], emissionCol[4], specularCol[4], absorption[4];
    float absorptionStrength, emissionStrength, smoothness, specularProbability;
    float ior;
    int32_t flag;
    int32_t pad[2];
};

struct KArgs {
    /* scene */
    const float* spheres;        /* nSpheres x (cx, cy, cz, r*r) */
    const float* sphereQuick;    /* ceil(nSpheres / 2) x (cx0, cx1, cy0, cy1, cz0, cz1, K0, K1), K = |c|^2 - r*r: the conservative pre-test, two spheres per record */
    float sphereBound;           /* max_k(|c_k|^2 + r_k^2): scales the pre-test's error margin */
    const DMaterial* materials;  /* [0,nSpheres) spheres, then models */
    const DModel* models;
    const DPair* pairs;          /* the pair space */
    const DTri* tris;            /* the triangle space (the arena layout: == pairs) */
    const DTriN* norms;          /* 12 bytes per unit of the triangle space: the normals of the triangle at unit u start at byte 12 u */
    const uint32_t* bigLeaves;   /* pairs of (first unit, count) */
    const DFilter* filters;      /* one per model */
    const float* filterPairs;    /* the same boxes two models side by side (minx0 minx1 miny0 miny1 minz0 minz1 maxx0 ... always0 always1 - -),
                                  * sixteen dwords per pair, behind the DFilter array: the packed two-models-per-step root filter */
    float filterMaxOrigin;       /* ray origins farther than this from 0 skip the filter */
    int32_t nSpheres, nModels;
    /* models [0, nFiltered) go through the conservative filter.  nModels <= 64: bit m of the lane's 64-bit
     * candidate mask is model m.  More models: bits 0..62 = models 0..62, bit 63 = "more candidates in the LDS
     * extension" ([1 + extWords][64] after the pixel bookkeeping: a summary word, then 32 models per word,
     * word k = models 63+32k ..), filled chunk by chunk (DChunk) */
    const DChunk* chunks;
    int32_t nChunks, nFiltered, extWords;
    /* render targets: rows owned by this context, packed */
    float* frameRender;
    float* accumulated;
    uint32_t W, H;               /* GLOBAL resolution (uniform Resolution) */
    int32_t localRows;
    int32_t stripRows, partIndex, partCount;
    int32_t tilesX, tilesY;
    int32_t stackEntries;        /* a wave's LDS: [stackEntries][64] traversal stack, then [RT_PIXEL_FIELDS][64] pixel bookkeeping */
    /* the BVH variants' workgroups (round 6): wavesPerGroup waves, LDS = [hot cache: hotUnits x 16 B][wave 0's region][wave 1's] ...;
     * units [0, hotUnits) of the pair space are the top-of-tree records the workgroup copies into LDS when it starts (rt_layout.h) */
    int32_t wavesPerGroup, hotUnits, waveLdsDwords;
    uint32_t travLimit;          /* traversal watchdog: iterations of one traverse() call no validated scene can reach (rt_kernels.h) */
    /* uniforms (RtParams) */
    int32_t maxBounce, spp, frame0, nFrames, seed, useSky, accumulate;
    float defocus, diverge, sunFocus, sunIntensity;
    float sunColour[3], dirToSun[3], viewParams[3];
    float cam[16];
    /* reciprocals of launch constants, computed on the host with the same correctly rounded fp32
     * divide the device would use (x / c == x * rcp(c), include/rt_math.h rt_div) */
    float rcpWm1, rcpHm1;        /* 1 / (Resolution - 1)  — RCC:15 */
    float rcpW;                  /* 1 / numPixels.x       — RC:567,572 */
    float rcpSpp;                /* 1 / NumRaysPerPixel   — RC:581 */
    int32_t suspendNum;          /* traverse() is left once active <= entered * suspendNum / 8 lanes are still traversing */
    int32_t raygenNoDefocus;     /* defocusStrength == 0, camera matrix finite, no component of the camera origin is -0 */
    /* counters: RT_COUNTER_SLOTS x RT_COUNTER_FIELDS u64 */
    unsigned long long* counters;
    /* persistent waves: this launch renders launchTiles tiles; its queue position q is entry
     * q*orderStride + orderOffset of the tile order.  Positions beyond the grid come from a global
     * atomic counter (monotonic across launches; this launch's positions start at tileQueueBase) */
    int32_t launchTiles, orderOffset, orderStride;
    int32_t launchItems;         /* queue positions of this launch: launchTiles, or launchTiles * frameGroups (tile, frame group) items */
    float4* pxCold;              /* per-wave pixel records of this launch: [grid][64 lanes][2] float4 (rt_kernels.h, PX_COLD) */
    int32_t frameGroup;          /* consecutive frames per item (>= 1) */
    int32_t frameGroups;         /* ceil(nFrames / frameGroup) */
    float* staging;              /* nFrames > 1: [frame - frame0][stagingStride pixels] RGBA colours awaiting rt_accumulate_kernel */
    uint32_t stagingStride;
    int32_t queueStart;          /* 1: the first position of every wave comes from the queue too (not blockIdx) */
    unsigned long long* tileQueue;
    unsigned long long tileQueueBase;
    /* longest-chain-first scheduling: queue position -> tile (null = identity), and the
     * per-tile record of the longest pixel chain seen so far (segments in one frame) */
    const uint32_t* tileOrder;
    uint32_t* tileCost;
    /* (round 6; at the end: the BVH variants' register allocation is sensitive to where their arguments lie) */
    int32_t poolCells;           /* the FLAT variant's chain pool: RT_POOL_CELLS, or 0 = single-wave workgroups without a pool;
                                  * hotUnits then = the pool region's size in 16-byte units (the wave regions start behind it) */
    int32_t frameGroupShift;     /* floor(log2(frameGroup)): the tile cost's per-frame figure without a division */
    uint32_t poolSpinLimit;      /* polls of a cell's sequence word before the chain pool's watchdog gives up (65,536); top bit: the RT_POOL_FAULT test hook */
};

#endif
