/*
 * rt_bvh_gpu.hip — rt_build_bvh_gpu: the reference's BVH constructor (Assets/Scripts/Types/BVH.cs:26-318)
 * on the GPU, byte-identical to rt_build_bvh (same nodes, same triangle order, same statistics).
 *
 * BVH.cs is a depth-first recursion whose three ingredients look sequential; none of them is:
 *   - EvaluateSplit (BVH:253-311) scans a node's triangles keeping running min/max with "if (t < cur) cur = t":
 *     the first of equal values stays (signed zeros).  Any ORDERED reduction of ordered chunks gives the same
 *     bits: here 128-triangle runs reduced sequentially by one thread per candidate plane, four runs per
 *     512-triangle chunk, chunks combined in order;
 *   - the in-place partition (BVH:117-153) is a Lomuto scan: a triangle that goes left is swapped with the FIRST
 *     triangle of the right-hand block, which thereby moves to the back of that block — a rotating queue.  Number
 *     the scan's events from the first right-hand triangle on: event p writes "tape" position p; a right-hand
 *     triangle writes itself, the r-th left-hand triangle writes a copy of tape position r; the final right-hand
 *     block is the last nRight tape positions, the left-hand block is stable.  That is one prefix sum of the
 *     left flags plus a pointer-jumping pass over tape[p] -> tape[r(p)] (r(p) < p);
 *   - node indices are handed out in depth-first order at split time (BVH:161-162): the children of the k-th
 *     inner node in PRE-order get indices 1+2k, 2+2k.  The tree is built level by level into a breadth-first
 *     array; subtree inner-node counts (bottom-up) and pre-order ranks (top-down) give every node its index.
 * The arithmetic (candidate planes, costs, strict comparisons, the (axis 0, pos 0) fallback) is the host builder's,
 * compiled with the same flags (no contraction, correctly rounded divide).
 *
 * Nothing in the level loop knows that there is ONE root: rt_build_bvh_gpu_batch builds the meshes of a scene as a FOREST —
 * K roots at level 0 over the concatenated triangle array, every level's kernels run once for the nodes of all meshes, and
 * the numbering and the statistics are kept per mesh (a node carries its mesh).  Twelve 82k-triangle meshes cost what one
 * 983k-triangle mesh costs, not twelve times ~21 levels of small launches.
 */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <mutex>
#include <vector>

#include "../../include/rt_abi.h"

namespace gbvh {

#define GB_FMAX 3.40282347e+38f
#define GB_CHUNK 512          /* triangles per sweep chunk */
#define GB_RUN 128            /* triangles per sequential run inside a chunk */
#define GB_NCAND 16           /* 15 candidate planes (BVH:211-246) + the (axis 0, pos 0) fallback split */

struct GTri {
    float c[3], mn[3], mx[3];
    int index;
};
struct GBox {
    float lmn[3], lmx[3], rmn[3], rmx[3];
    int nLeft;
};
struct GCand {
    int axis;
    float pos;
};
struct GNode { /* breadth-first record */
    float bmin[3], bmax[3];
    int start, count, depth;
    int left;        /* breadth-first index of the left child (right = left + 1); -1: leaf or undecided */
    int innerCount;  /* inner nodes in this subtree */
    int preIdx;      /* pre-order rank among inner nodes */
    int id;          /* final node index */
    int nCand;       /* candidates of this level's sweep */
    int chunkBase;   /* first sweep chunk */
    int splitAxis;
    float splitPos;
    int nLeft;
    int mesh;        /* which mesh of the batch (its root is node `mesh` of level 0) */
};

__device__ __forceinline__ float max3f(float a, float b, float c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

__device__ __forceinline__ void box_reset(GBox& b)
{
    for (int k = 0; k < 3; k++) { b.lmn[k] = GB_FMAX; b.lmx[k] = -GB_FMAX; b.rmn[k] = GB_FMAX; b.rmx[k] = -GB_FMAX; }
    b.nLeft = 0;
}
/* append the box of a LATER run: the earlier one keeps equal values */
__device__ __forceinline__ void box_append(GBox& a, const GBox& o)
{
    for (int k = 0; k < 3; k++) {
        if (o.lmn[k] < a.lmn[k]) a.lmn[k] = o.lmn[k];
        if (o.lmx[k] > a.lmx[k]) a.lmx[k] = o.lmx[k];
        if (o.rmn[k] < a.rmn[k]) a.rmn[k] = o.rmn[k];
        if (o.rmx[k] > a.rmx[k]) a.rmx[k] = o.rmx[k];
    }
    a.nLeft += o.nLeft;
}

/* mesh of triangle t: triBase[k] <= t < triBase[k + 1] */
__device__ __forceinline__ int mesh_of(const int* triBase, int nMeshes, int t)
{
    int lo = 0, hi = nMeshes - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (triBase[mid] <= t) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
/* BVH:44-52.  The batch's vertex and index arrays are concatenated; a mesh's indices are relative to its own vertices. */
__global__ void k_prepare(const float* verts, const int* indices, int ntri, const int* triBase, const int* vertBase, int nMeshes, GTri* tris, int* triNode)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntri) return;
    const int mesh = nMeshes > 1 ? mesh_of(triBase, nMeshes, t) : 0;
    const float* vb = verts + 3 * (size_t)vertBase[mesh];
    const float* a = vb + 3 * indices[3 * t + 0];
    const float* b = vb + 3 * indices[3 * t + 1];
    const float* c = vb + 3 * indices[3 * t + 2];
    GTri g;
    for (int k = 0; k < 3; k++) {
        g.c[k] = (a[k] + b[k] + c[k]) / 3;
        g.mn[k] = a[k] < b[k] ? (a[k] < c[k] ? a[k] : c[k]) : (b[k] < c[k] ? b[k] : c[k]);
        g.mx[k] = a[k] > b[k] ? (a[k] > c[k] ? a[k] : c[k]) : (b[k] > c[k] ? b[k] : c[k]);
    }
    g.index = 3 * t; /* position in the concatenated index array */
    tris[t] = g;
    triNode[t] = mesh; /* the roots are nodes 0 .. nMeshes-1 */
}

/* ChooseSplit's candidate planes (BVH:183-250), in evaluation order; slot 15 = the (0, 0) fallback */
__global__ void k_candidates(GNode* nodes, int first, int nActive, int quality, GCand* cands, int* counts)
{
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a == nActive) counts[nActive] = 0; /* the scan's total lands here (the grid covers nActive + 1 threads) */
    if (a >= nActive) return;
    GNode& n = nodes[first + a];
    GCand* cd = cands + (size_t)a * GB_NCAND;
    for (int j = 0; j < GB_NCAND; j++) { cd[j].axis = 0; cd[j].pos = 0.0f; }
    int nc = 0;
    if (n.count > 1) { /* BVH:185 */
        const float size[3] = {n.bmax[0] - n.bmin[0], n.bmax[1] - n.bmin[1], n.bmax[2] - n.bmin[2]};
        if (quality == RT_BVH_QUALITY_LOW) {
            int ax = (size[0] > size[1] && size[0] > size[2]) ? 0 : (size[1] > size[2] ? 1 : 2);
            cd[0].axis = ax;
            cd[0].pos = n.bmin[ax] + size[ax] * 0.5f;
            nc = 1;
        } else {
            const int maxSplitTests = n.count < 10 ? 3 : 5;
            const float maxAxis = max3f(size[0], size[1], size[2]);
            for (int axis = 0; axis < 3; axis++) {
                float v = size[axis] / maxAxis * maxSplitTests;
                int m = (v != v) ? INT32_MIN : (int)ceilf(v); /* CeilToInt(NaN) == int.MinValue */
                m = m < 1 ? 1 : (m > maxSplitTests ? maxSplitTests : m);
                for (int i = 0; i < m; i++) {
                    float splitT = (i + 1) / (m + 1.0f);
                    cd[nc].axis = axis;
                    cd[nc].pos = n.bmin[axis] + size[axis] * splitT;
                    nc++;
                }
            }
        }
    }
    n.nCand = nc;
    counts[a] = nc > 0 ? (n.count + GB_CHUNK - 1) / GB_CHUNK : 0; /* sweep chunks of this node */
}

/* chunk -> node: chunkBase is the exclusive prefix sum of the nodes' chunk counts (non-decreasing); chunk c belongs to
 * the LAST node whose base is <= c (nodes without chunks share their base with the next node).  One thread per chunk
 * (the root of a 327k-triangle mesh has 640 of them; a thread per node walked them one by one). */
__global__ void k_chunk_map(GNode* nodes, int first, int nActive, const int* chunkBase, int maxChunks, int* chunkNode)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nActive) nodes[first + c].chunkBase = chunkBase[c];
    if (c >= maxChunks || c >= chunkBase[nActive]) return;
    int lo = 0, hi = nActive - 1; /* find the largest a with chunkBase[a] <= c */
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (chunkBase[mid] <= c) lo = mid;
        else hi = mid - 1;
    }
    chunkNode[c] = lo;
}

/* EvaluateSplit (BVH:253-311) over one chunk of one node: thread (candidate j, run s) scans its run of up to
 * GB_RUN triangles sequentially with the reference's strict comparisons; the four runs are appended in order. */
__global__ void __launch_bounds__(64) k_sweep(const GNode* nodes, int first, const int* chunkNode, const int* nChunksPtr, const GCand* cands, const GTri* tris,
                                              GBox* partial)
{
    __shared__ float s_c[3][GB_CHUNK];
    __shared__ float s_mn[3][GB_CHUNK];
    __shared__ float s_mx[3][GB_CHUNK];
    __shared__ GBox s_run[4][GB_NCAND];
    const int chunk = blockIdx.x;
    if (chunk >= *nChunksPtr) return; /* the grid is an upper bound (no read-back of the chunk count) */
    const int a = chunkNode[chunk];
    const GNode& n = nodes[first + a];
    const int begin = n.start + (chunk - n.chunkBase) * GB_CHUNK;
    const int end = (begin + GB_CHUNK < n.start + n.count) ? begin + GB_CHUNK : n.start + n.count;
    const int m = end - begin;
    for (int i = threadIdx.x; i < m; i += 64) {
        const GTri t = tris[begin + i];
        for (int k = 0; k < 3; k++) { s_c[k][i] = t.c[k]; s_mn[k][i] = t.mn[k]; s_mx[k][i] = t.mx[k]; }
    }
    __syncthreads();
    const int j = threadIdx.x & 15, s = threadIdx.x >> 4;
    const int nc = n.nCand;
    const bool work = (j < nc) || (j == GB_NCAND - 1);
    GBox b;
    box_reset(b);
    if (work) {
        const GCand cd = cands[(size_t)a * GB_NCAND + j];
        const int r0 = s * GB_RUN, r1 = (r0 + GB_RUN < m) ? r0 + GB_RUN : m;
        for (int i = r0; i < r1; i++) {
            if (s_c[cd.axis][i] < cd.pos) {
                for (int k = 0; k < 3; k++) {
                    if (s_mn[k][i] < b.lmn[k]) b.lmn[k] = s_mn[k][i];
                    if (s_mx[k][i] > b.lmx[k]) b.lmx[k] = s_mx[k][i];
                }
                b.nLeft++;
            } else {
                for (int k = 0; k < 3; k++) {
                    if (s_mn[k][i] < b.rmn[k]) b.rmn[k] = s_mn[k][i];
                    if (s_mx[k][i] > b.rmx[k]) b.rmx[k] = s_mx[k][i];
                }
            }
        }
    }
    s_run[s][j] = b;
    __syncthreads();
    if (s == 0 && work) {
        GBox acc = s_run[0][j];
        for (int r = 1; r < 4; r++) box_append(acc, s_run[r][j]);
        partial[(size_t)chunk * GB_NCAND + j] = acc;
    }
}

__device__ __forceinline__ float node_cost(const float* mn, const float* mx, int n) /* BVH:313-318 */
{
    if (n == 0) return 0;
    float x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2];
    float area = x * y + x * z + y * z;
    return area * n;
}

/* per active node: combine the chunk partials in order, choose the split (BVH:200-208), decide (BVH:101).
 * 16 x SEGS threads per node: thread (candidate j, segment s) appends its quarter (SEGS = 4) or all (SEGS = 1) of the
 * node's chunks in order, the segments are appended in order, lane 0 of the group picks the first cheapest candidate.
 * (One thread per node did all 16 candidates x up to 640 chunks alone: 60 % of the whole build's GPU time.) */
__device__ __forceinline__ GBox box_shfl(const GBox& b, int srcLane)
{
    GBox r;
    for (int k = 0; k < 3; k++) {
        r.lmn[k] = __shfl(b.lmn[k], srcLane, 64); r.lmx[k] = __shfl(b.lmx[k], srcLane, 64);
        r.rmn[k] = __shfl(b.rmn[k], srcLane, 64); r.rmx[k] = __shfl(b.rmx[k], srcLane, 64);
    }
    r.nLeft = __shfl(b.nLeft, srcLane, 64);
    return r;
}
template <int SEGS>
__global__ void __launch_bounds__(256) k_choose(GNode* nodes, int first, int nActive, int quality, const GCand* cands, const GBox* partial, const int* chunkCounts,
                                                GBox* chosen, int* splitFlag)
{
    constexpr int TPN = GB_NCAND * SEGS; /* threads per node: 16 or 64, a divisor of the wave size */
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    if (gt == 0) splitFlag[nActive] = 0;
    const int a0 = gt / TPN;
    const bool valid = a0 < nActive;
    const int a = valid ? a0 : 0; /* lanes past the end go through the motions (the shuffles below are wave-wide) */
    const int lane = threadIdx.x & 63, group0 = lane & ~(TPN - 1);
    const int j = lane & (GB_NCAND - 1), seg = (lane & (TPN - 1)) >> 4;
    GNode& n = nodes[first + a];
    const int nc = n.nCand, count = n.count;
    const bool mine = nc > 0 && (j < nc || j == GB_NCAND - 1);
    GBox acc;
    box_reset(acc);
    if (mine) {
        const int nChunks = chunkCounts[a];
        const int per = (nChunks + SEGS - 1) / SEGS;
        const int c0 = seg * per, c1 = (c0 + per < nChunks) ? c0 + per : nChunks;
        for (int c = c0; c < c1; c++) {
            const GBox p = partial[(size_t)(n.chunkBase + c) * GB_NCAND + j];
            if (c == c0) acc = p;      /* the reference starts from the first triangle's values: box_reset's sentinels would also do, */
            else box_append(acc, p);   /* but a copy keeps the bits of a chunk that is alone exactly as the sweep left them */
        }
    }
    if (SEGS > 1) { /* later segments appended in order onto segment 0 (an empty segment appends sentinels: no change) */
        for (int sgm = 1; sgm < SEGS; sgm++) {
            const GBox o = box_shfl(acc, group0 + sgm * GB_NCAND + j);
            const int oc = __shfl(mine ? 1 : 0, group0 + sgm * GB_NCAND + j, 64);
            if (seg == 0 && mine && oc) {
                const int nChunks = chunkCounts[a];
                const int per = (nChunks + SEGS - 1) / SEGS;
                if (sgm * per < nChunks) box_append(acc, o);
            }
        }
    }
    /* candidate costs in lanes (j, segment 0) */
    float cj = GB_FMAX;
    if (mine && j != GB_NCAND - 1) cj = node_cost(acc.lmn, acc.lmx, acc.nLeft) + node_cost(acc.rmn, acc.rmx, count - acc.nLeft);
    int best = -1;
    float cost = INFINITY; /* count <= 1: BVH:185 */
    if (nc > 0) {
        if (quality == RT_BVH_QUALITY_LOW) {
            best = 0;
            cost = __shfl(cj, group0, 64);
        } else {
            float bestCost = GB_FMAX;
            for (int k = 0; k < GB_NCAND - 1; k++) { /* BVH:200-208: strict '<', the first of equal costs stays */
                const float ck = __shfl(cj, group0 + k, 64);
                if (k < nc && ck < bestCost) { bestCost = ck; best = k; }
            }
            cost = bestCost;
        }
    } else {
        for (int k = 0; k < GB_NCAND - 1; k++) (void)__shfl(cj, group0 + k, 64); /* keep the wave's shuffles aligned */
    }
    const float sx = n.bmax[0] - n.bmin[0], sy = n.bmax[1] - n.bmin[1], sz = n.bmax[2] - n.bmin[2];
    float parentCost = 0;
    if (count != 0) { float area = sx * sy + sx * sz + sy * sz; parentCost = area * count; }
    const bool split = cost < parentCost && n.depth < 32; /* BVH:101 */
    const int pick = best >= 0 ? best : GB_NCAND - 1;     /* bestSplitAxis/Pos stay (0, 0) when no candidate won: BVH:204-205 */
    if (valid && seg == 0 && j == pick && split) chosen[a] = acc;
    if (valid && seg == 0 && j == 0) {
        splitFlag[a] = split ? 1 : 0;
        n.left = -1;
    }
    if (valid && seg == 0 && j == pick && split) {
        if (best >= 0) { n.splitAxis = cands[(size_t)a * GB_NCAND + best].axis; n.splitPos = cands[(size_t)a * GB_NCAND + best].pos; }
        else { n.splitAxis = 0; n.splitPos = 0.0f; }
        n.nLeft = acc.nLeft;
    }
}

/* children of the splitting nodes: breadth-first slots firstChild + 2 * rank */
__global__ void k_children(GNode* nodes, int first, int nActive, const int* splitFlag, const int* splitRank, const GBox* chosen, int firstChild)
{
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nActive || !splitFlag[a]) return;
    GNode& n = nodes[first + a];
    const int li = firstChild + 2 * splitRank[a];
    n.left = li;
    const GBox& b = chosen[a];
    GNode l, r;
    memset(&l, 0, sizeof(l));
    memset(&r, 0, sizeof(r));
    for (int k = 0; k < 3; k++) { l.bmin[k] = b.lmn[k]; l.bmax[k] = b.lmx[k]; r.bmin[k] = b.rmn[k]; r.bmax[k] = b.rmx[k]; }
    l.start = n.start; l.count = b.nLeft; l.depth = n.depth + 1; l.left = -1; l.mesh = n.mesh;
    r.start = n.start + b.nLeft; r.count = n.count - b.nLeft; r.depth = n.depth + 1; r.left = -1; r.mesh = n.mesh;
    nodes[li] = l;
    nodes[li + 1] = r;
}

/* ---- partition (see the header): flags, prefix sum, tape pointers, pointer jumping, scatter */
__global__ void k_flags(const GNode* nodes, const int* triNode, const GTri* tris, int ntri, int* flag)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == ntri) flag[ntri] = 0;
    if (g >= ntri) return;
    const int nd = triNode[g];
    int f = 0;
    if (nd >= 0 && nodes[nd].left >= 0) f = tris[g].c[nodes[nd].splitAxis] < nodes[nd].splitPos ? 1 : 0;
    flag[g] = f;
}
__global__ void k_tape(const GNode* nodes, const int* triNode, const int* flag, const int* S, int ntri, int* src, int* firstR)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ntri) return;
    src[g] = g;
    const int nd = triNode[g];
    if (nd < 0 || nodes[nd].left < 0) return;
    const GNode& n = nodes[nd];
    if (!flag[g] && S[g] - S[n.start] == g - n.start) firstR[nd] = g; /* exactly one right-hand triangle has only lefts before it */
    /* every triangle before the first right-hand one goes left: firstR = start + (length of the leading run of lefts).
     * nLeft lefts in all; the leading run is found from the prefix sums: position p is in it iff S[p] - S[start] == p - start
     * and flag[p] == 1.  The first right-hand triangle is the first position where that fails. */
    const int start = n.start;
    const int cl = S[g] - S[start]; /* lefts before g */
    if (flag[g]) {
        /* r-th rotation -> tape position r; only rotations after the first right-hand triangle exist */
        const int lead = cl == g - start; /* still inside the leading run: stays where it is */
        if (!lead) {
            /* firstR = start + (number of leading lefts) = the position of the first right-hand triangle: the leading run
             * length L0 satisfies S[start + L0] - S[start] == L0 and flag[start + L0] == 0.  Lefts before g = cl, of which L0
             * lead; r = cl - L0.  L0 is not known locally: recover it from the right-hand triangles (see k_tape2). */
            src[g] = -1 - cl; /* provisional: resolved in k_tape2 once firstR of the node is known */
        }
    }
}
__global__ void k_tape2(const GNode* nodes, const int* triNode, const int* firstR, int ntri, int* src)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ntri) return;
    if (src[g] >= 0) return;
    const int nd = triNode[g];
    const int cl = -1 - src[g];
    const int fr = firstR[nd];
    const int lead = fr - nodes[nd].start; /* lefts in the leading run */
    src[g] = fr + (cl - lead);             /* the r-th rotation copies tape position r = global position firstR + r */
}
__global__ void k_jump(int ntri, int* src, int* changed)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ntri) return;
    const int s = src[g];
    const int t = src[s];
    if (t != s) {
        src[g] = t;
        if (changed) *changed = 1;
    }
}
/* after a few synchronous jumps: every thread follows what is left of its chain to the root (a position that copies
 * itself).  Read-only walk over pointers that other threads only ever move FURTHER along the same chain, so a stale
 * read is still on the chain; the walk ends at the same root whatever the interleaving.  Replaces a host loop of
 * "three passes, read a flag back" rounds — two host synchronisations per level less. */
__global__ void k_follow(int ntri, int* src)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ntri) return;
    const volatile int* vs = src;
    int s = vs[g];
    for (;;) {
        const int t = vs[s];
        if (t == s) break;
        s = t;
    }
    src[g] = s;
}
__global__ void k_scatter(const GNode* nodes, const int* triNode, const int* flag, const int* S, const int* firstR, const int* src, const GTri* tris, int ntri,
                          GTri* outTris, int* outTriNode)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ntri) return;
    const int nd = triNode[g];
    if (nd < 0 || nodes[nd].left < 0) { /* finished leaf, or a node that became a leaf at this level */
        outTris[g] = tris[g];
        outTriNode[g] = -1;
        return;
    }
    const GNode& n = nodes[nd];
    const int start = n.start;
    const int nLeft = n.nLeft;
    if (flag[g]) { /* left-hand block is stable */
        const int dest = start + (S[g] - S[start]);
        outTris[dest] = tris[g];
        outTriNode[dest] = n.left;
    }
    const int fr = firstR[nd];
    if (fr >= 0 && g >= fr) { /* tape position p = g - firstR; the last nRight positions are the right-hand block */
        const int p = g - fr;
        const int laterL = nLeft - (fr - start);
        if (p >= laterL) {
            const int dest = start + nLeft + (p - laterL);
            outTris[dest] = tris[src[g]];
            outTriNode[dest] = n.left + 1;
        }
    }
}

/* ---- final numbering: subtree inner counts (bottom-up), pre-order ranks and node indices (top-down) */
__global__ void k_inner_count(GNode* nodes, int first, int count)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    GNode& n = nodes[first + i];
    n.innerCount = n.left >= 0 ? 1 + nodes[n.left].innerCount + nodes[n.left + 1].innerCount : 0;
}
__global__ void k_number(GNode* nodes, int first, int count)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    GNode& n = nodes[first + i];
    if (n.left < 0) return;
    GNode& l = nodes[n.left];
    GNode& r = nodes[n.left + 1];
    l.id = 1 + 2 * n.preIdx;
    r.id = 2 + 2 * n.preIdx;
    l.preIdx = n.preIdx + 1;
    r.preIdx = n.preIdx + 1 + l.innerCount;
}
__device__ __forceinline__ int wave_add(int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ __forceinline__ int wave_max(int v) { for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o, 64); v = t > v ? t : v; } return v; }
__device__ __forceinline__ int wave_min(int v) { for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o, 64); v = t < v ? t : v; } return v; }
/* first node of every mesh in the output: a mesh has 1 + 2 * (inner nodes) nodes */
__global__ void k_node_base(const GNode* nodes, int nMeshes, int* nodeBase)
{
    if (blockIdx.x || threadIdx.x) return;
    int acc = 0;
    for (int k = 0; k < nMeshes; k++) { nodeBase[k] = acc; acc += 1 + 2 * nodes[k].innerCount; }
    nodeBase[nMeshes] = acc;
}
__global__ void k_emit(const GNode* nodes, int total, const int* nodeBase, const int* triBase, RtBVHNode* out,
                       int* stats /* per mesh: leafCount, depthSum, depthMax, depthMin, triMax, triMin, triSum, - */)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int leaf = 0, depth = 0, cnt = 0, mesh = -1;
    if (i < total) {
        const GNode& n = nodes[i];
        mesh = n.mesh;
        RtBVHNode o;
        for (int k = 0; k < 3; k++) { o.boundsMin[k] = n.bmin[k]; o.boundsMax[k] = n.bmax[k]; }
        if (n.left >= 0) {
            o.startIndex = 1 + 2 * n.preIdx; /* BVH:165 */
            o.triangleCount = n.depth == 0 ? -1 : 0; /* the root keeps its constructor value (BVH:61) */
        } else {
            o.startIndex = n.start - triBase[mesh]; /* relative to the mesh's first triangle */
            o.triangleCount = n.count;
            leaf = 1; depth = n.depth; cnt = n.count;
        }
        out[nodeBase[mesh] + n.id] = o;
    }
    /* BuildStats (BVH:557-575): one set of atomics per wave and mesh instead of seven per leaf on the same seven words
     * (a level's nodes are grouped by mesh, so a wave rarely sees more than one) */
    unsigned long long todo = __ballot(leaf != 0);
    while (todo) {
        const int m = __shfl(mesh, __ffsll((long long)todo) - 1, 64);
        const bool mine = leaf && mesh == m;
        const int leaves = wave_add(mine ? 1 : 0);
        const int dSum = wave_add(mine ? depth : 0), tSum = wave_add(mine ? cnt : 0);
        const int dMax = wave_max(mine ? depth : INT32_MIN), dMin = wave_min(mine ? depth : INT32_MAX);
        const int tMax = wave_max(mine ? cnt : INT32_MIN), tMin = wave_min(mine ? cnt : INT32_MAX);
        if ((threadIdx.x & 63) == 0) {
            int* st = stats + 8 * m;
            atomicAdd(&st[0], leaves);
            atomicAdd(&st[1], dSum);
            atomicMax(&st[2], dMax);
            atomicMin(&st[3], dMin);
            atomicMax(&st[4], tMax);
            atomicMin(&st[5], tMin);
            atomicAdd(&st[6], tSum);
        }
        todo &= ~__ballot(mine);
    }
}
__global__ void k_tri_index(const GTri* tris, int ntri, int* out)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < ntri) out[g] = tris[g].index;
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    /* no destructor: a hipFree at process exit may run after the HIP runtime is gone; rt_build_bvh_gpu_release() frees the
     * pool explicitly, the process's teardown frees the rest */
    void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
    template <typename T>
    T* get(size_t n)
    {
        size_t bytes = n * sizeof(T);
        if (bytes > cap) {
            if (p) hipFree(p);
            p = nullptr;
            cap = 0;
            size_t want = bytes + bytes / 4 + 256;
            if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return nullptr; }
            cap = want;
        }
        return (T*)p;
    }
};

#define GB_TRY(call)                                   \
    do {                                               \
        hipError_t e_ = (call);                        \
        if (e_ != hipSuccess) return RT_ERR_HIP;       \
    } while (0)

static inline int blocks(size_t n, int t = 256) { return (int)((n + t - 1) / t); }

struct Pool { int device = -1; DevBuf b[23]; };
/* ONE pool for the process, used under g_poolMutex (round 4, ADVICE r3: the pool was thread_local, so every short-lived worker
 * thread of a host that builds meshes on a thread pool left up to 4 GiB of device memory behind, and rt_build_bvh_gpu_release only
 * reached the calling thread's).  Builds from several threads therefore run one after the other — each one fills the GPU anyway. */
static Pool g_pool;
static std::mutex g_poolMutex;
static const size_t GB_POOL_KEEP = (size_t)4 << 30; /* scratch kept between calls: at most 4 GiB of the 288 (a 1.3M-triangle mesh needs ~1.5 GiB) */

static void pool_release()
{
    if (g_pool.device >= 0) {
        int prev = -1;
        const bool have = hipGetDevice(&prev) == hipSuccess;
        if (hipSetDevice(g_pool.device) == hipSuccess)
            for (DevBuf& d : g_pool.b) d.release();
        if (have) hipSetDevice(prev);
    }
    g_pool.device = -1;
}

struct MeshIn {
    const float* verts;
    const float* normals;
    int n_verts;
    const int32_t* indices;
    int n_indices;
};

/* f(k) for k in [0, n) on at most `cap` host threads */
template <typename F>
static void host_parallel(int n, int cap, F f)
{
    unsigned hc = std::thread::hardware_concurrency();
    int nth = hc ? (int)hc : 1;
    if (nth > cap) nth = cap;
    if (nth > n) nth = n;
    if (nth <= 1) {
        for (int k = 0; k < n; k++) f(k);
        return;
    }
    std::atomic<int> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < nth; t++)
        th.emplace_back([&] { for (int k = next.fetch_add(1); k < n; k = next.fetch_add(1)) f(k); });
    for (auto& t : th) t.join();
}

/* K >= 1 meshes as one forest.  Mesh k's nodes go to out_nodes + nodeOffset[k] (nodeOffset = running sum of the node counts),
 * its triangles to out_tris + triBase[k]; out_n_nodes / out_node_offset / out_tri_offset / out_stats hold K entries (the offset
 * arrays and out_stats may be null).  RT_BVH_QUALITY_DISABLED and empty meshes make their one-node tree on the host (K == 1 only;
 * the batch entry builds such scenes mesh by mesh). */
static int build_forest(int device, int K, const MeshIn* in, int quality, RtBVHNode* out_nodes, int* out_n_nodes, int* out_node_offset,
                        RtTriangle* out_tris, int* out_tri_offset, RtBvhStats* out_stats)
{
    if (K < 1 || !in || !out_nodes || !out_n_nodes || !out_tris) return RT_ERR_INVALID_ARG;
    if (quality != RT_BVH_QUALITY_LOW && quality != RT_BVH_QUALITY_HIGH && quality != RT_BVH_QUALITY_DISABLED) return RT_ERR_INVALID_ARG;
    std::vector<int> triBase(K + 1, 0), vertBase(K + 1, 0);
    for (int k = 0; k < K; k++) {
        const MeshIn& m = in[k];
        if (!m.verts || !m.normals || !m.indices || m.n_verts < 0 || m.n_indices < 0 || m.n_indices % 3) return RT_ERR_INVALID_ARG;
        const long long tb = (long long)triBase[k] + m.n_indices / 3, vb = (long long)vertBase[k] + m.n_verts;
        if (tb > 0x2aaaaaaaLL || vb > 0x2aaaaaaaLL) return RT_ERR_SCENE; /* 3 * position must fit an int */
        triBase[k + 1] = (int)tb;
        vertBase[k + 1] = (int)vb;
    }
    auto t0 = std::chrono::steady_clock::now();
    const bool dbg = getenv("RT_BVH_DEBUG") != nullptr;
    auto lap = [&](const char* what) {
        if (dbg) fprintf(stderr, "[bvh-gpu] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RT_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return RT_ERR_INVALID_ARG;
    GB_TRY(hipSetDevice(device));
    const int ntri = triBase[K];

    /* index validation + root bounds (BVH:53-58, in triangle order): ordered blocks of each mesh's triangles on a few host threads,
     * a mesh's block boxes appended in order with the same strict comparisons (the first of equal values stays) */
    struct Part { int mesh, b0, e0; float mn[3], mx[3]; bool bad; };
    std::vector<Part> parts;
    for (int k = 0; k < K; k++) {
        const int nt = triBase[k + 1] - triBase[k];
        int nb = nt / 16384;
        if (nb > 8) nb = 8;
        if (nb < 1) nb = 1;
        for (int b = 0; b < nb; b++) {
            Part P;
            P.mesh = k; P.b0 = (int)((long long)nt * b / nb); P.e0 = (int)((long long)nt * (b + 1) / nb); P.bad = false;
            parts.push_back(P);
        }
    }
    host_parallel((int)parts.size(), 16, [&](int pi) {
        Part& P = parts[pi];
        const MeshIn& m = in[P.mesh];
        for (int d = 0; d < 3; d++) { P.mn[d] = GB_FMAX; P.mx[d] = -GB_FMAX; }
        for (int t = P.b0; t < P.e0; t++) {
            const int ia = m.indices[3 * t], ib = m.indices[3 * t + 1], ic = m.indices[3 * t + 2];
            if (ia < 0 || ia >= m.n_verts || ib < 0 || ib >= m.n_verts || ic < 0 || ic >= m.n_verts) { P.bad = true; return; }
            const float* a = m.verts + 3 * ia, * b = m.verts + 3 * ib, * c = m.verts + 3 * ic;
            for (int d = 0; d < 3; d++) {
                float mn = a[d] < b[d] ? (a[d] < c[d] ? a[d] : c[d]) : (b[d] < c[d] ? b[d] : c[d]);
                float mx = a[d] > b[d] ? (a[d] > c[d] ? a[d] : c[d]) : (b[d] > c[d] ? b[d] : c[d]);
                if (mn < P.mn[d]) P.mn[d] = mn;
                if (mx > P.mx[d]) P.mx[d] = mx;
            }
        }
    });
    std::vector<GNode> roots(K);
    for (int k = 0; k < K; k++) {
        GNode& r = roots[k];
        memset(&r, 0, sizeof(r));
        for (int d = 0; d < 3; d++) { r.bmin[d] = GB_FMAX; r.bmax[d] = -GB_FMAX; }
        r.start = triBase[k]; r.count = triBase[k + 1] - triBase[k]; r.depth = 0; r.left = -1; r.mesh = k;
    }
    for (const Part& P : parts) {
        if (P.bad) return RT_ERR_INVALID_ARG;
        GNode& r = roots[P.mesh];
        for (int d = 0; d < 3; d++) {
            if (P.mn[d] < r.bmin[d]) r.bmin[d] = P.mn[d];
            if (P.mx[d] > r.bmax[d]) r.bmax[d] = P.mx[d];
        }
    }

    lap("validate + root box");
    std::vector<int> order(ntri);
    std::vector<RtBVHNode> nodesOut; /* host-made tree (no BVH / empty mesh; K == 1) */
    std::vector<int> nodeBase(K + 1, 0);
    bool nodesInPlace = false;       /* the device trees were copied straight into out_nodes (no staging copy of ~21 MB at 327k triangles) */
    /* BVH:69-80: triangles in leaf order with vertex normals — host threads, started as soon as the order is known so that
     * they run while the node array is still on its way back from the device */
    std::vector<std::thread> gatherThreads;
    bool gatherStarted = false;
    auto start_gather = [&]() {
        gatherStarted = true;
        unsigned hc = std::thread::hardware_concurrency();
        int threads = hc ? (int)(hc > 16 ? 16 : hc) : 1;
        if (threads > ntri / 8192) threads = ntri / 8192;
        if (threads < 1) threads = 1;
        const int* ord = order.data();
        const int* tb = triBase.data();
        auto fill = [=](int b0, int e0) {
            int k = 0;
            while (k + 1 < K && tb[k + 1] <= b0) k++;
            for (int i = b0; i < e0; i++) {
                while (tb[k + 1] <= i) k++; /* (empty meshes have no positions) */
                const MeshIn& m = in[k];
                const int b = ord[i] - 3 * tb[k]; /* position in the mesh's own index array */
                RtTriangle& t = out_tris[i];
                const int ia = m.indices[b + 0], ib = m.indices[b + 1], ic = m.indices[b + 2];
                for (int c = 0; c < 3; c++) {
                    t.posA[c] = m.verts[3 * ia + c];
                    t.posB[c] = m.verts[3 * ib + c];
                    t.posC[c] = m.verts[3 * ic + c];
                    t.normA[c] = m.normals[3 * ia + c];
                    t.normB[c] = m.normals[3 * ib + c];
                    t.normC[c] = m.normals[3 * ic + c];
                }
            }
        };
        if (threads == 1) fill(0, ntri);
        else
            for (int t = 0; t < threads; t++) gatherThreads.emplace_back(fill, (int)((long long)ntri * t / threads), (int)((long long)ntri * (t + 1) / threads));
    };
    struct Joiner { std::vector<std::thread>& v; ~Joiner() { for (auto& t : v) if (t.joinable()) t.join(); } } joiner{gatherThreads}; /* every return path */
    std::vector<int> statsH((size_t)8 * K);
    for (int k = 0; k < K; k++) { int* st = &statsH[(size_t)8 * k]; st[0] = 0; st[1] = 0; st[2] = 0; st[3] = INT32_MAX; st[4] = 0; st[5] = INT32_MAX; st[6] = 0; st[7] = 0; }
    if (quality == RT_BVH_QUALITY_DISABLED || ntri == 0) { /* BVH:62-66 (and the empty mesh: Split makes the root a leaf) */
        if (K != 1) return RT_ERR_INVALID_ARG; /* the batch entry builds these mesh by mesh */
        RtBVHNode root;
        memcpy(root.boundsMin, roots[0].bmin, 12);
        memcpy(root.boundsMax, roots[0].bmax, 12);
        root.startIndex = 0;
        root.triangleCount = ntri;
        nodesOut.push_back(root);
        nodeBase[1] = 1;
        for (int t = 0; t < ntri; t++) order[t] = 3 * t;
        statsH[0] = (quality == RT_BVH_QUALITY_DISABLED) ? 0 : 1;
        if (quality != RT_BVH_QUALITY_DISABLED) { statsH[3] = 0; statsH[5] = 0; }
    } else {
        for (int k = 0; k < K; k++)
            if (roots[k].count == 0) return RT_ERR_INVALID_ARG; /* (K > 1: the batch entry keeps empty meshes out of the forest) */
        /* device scratch is kept between calls (a scene build calls this once per mesh): hipMalloc/hipFree of twenty
         * buffers would otherwise cost more than the build of a small mesh */
        Pool& pool = g_pool;
        if (pool.device != device) { pool_release(); pool.device = device; }
        DevBuf &bVerts = pool.b[0], &bIdx = pool.b[1], &bTrisA = pool.b[2], &bTrisB = pool.b[3], &bNodeA = pool.b[4], &bNodeB = pool.b[5], &bFlag = pool.b[6],
               &bScan = pool.b[7], &bSrc = pool.b[8], &bNodes = pool.b[9], &bCands = pool.b[10], &bPartial = pool.b[11], &bChosen = pool.b[12], &bCounts = pool.b[13],
               &bBase = pool.b[14], &bChunkNode = pool.b[15], &bSplit = pool.b[16], &bRank = pool.b[17], &bFirstR = pool.b[18], &bTemp = pool.b[19], &bMisc = pool.b[20],
               &bOut = pool.b[21], &bTables = pool.b[22];
        const int nVertsAll = vertBase[K];
        float* dVerts = bVerts.get<float>((size_t)nVertsAll * 3 + 1);
        int* dIdx = bIdx.get<int>((size_t)ntri * 3 + 1);
        GTri* trisA = bTrisA.get<GTri>(ntri);
        GTri* trisB = bTrisB.get<GTri>(ntri);
        int* nodeOfA = bNodeA.get<int>(ntri);
        int* nodeOfB = bNodeB.get<int>(ntri);
        int* flag = bFlag.get<int>((size_t)ntri + 1);
        int* S = bScan.get<int>((size_t)ntri + 1);
        int* src = bSrc.get<int>(ntri);
        const size_t maxNodes = 2 * (size_t)ntri + 66 * (size_t)K; /* 2*ntri - 1 per well-formed tree; empty-child chains are refused below */
        GNode* nodes = bNodes.get<GNode>(maxNodes);
        int* misc = bMisc.get<int>((size_t)8 * K + 16);
        int* tables = bTables.get<int>((size_t)3 * (K + 1)); /* triBase | vertBase | nodeBase */
        if (!dVerts || !dIdx || !trisA || !trisB || !nodeOfA || !nodeOfB || !flag || !S || !src || !nodes || !misc || !tables) return RT_ERR_OOM;
        int* dTriBase = tables, * dVertBase = tables + (K + 1), * dNodeBase = tables + 2 * (K + 1);
        for (int k = 0; k < K; k++) {
            if (in[k].n_verts) GB_TRY(hipMemcpy(dVerts + 3 * (size_t)vertBase[k], in[k].verts, sizeof(float) * 3 * (size_t)in[k].n_verts, hipMemcpyHostToDevice));
            if (in[k].n_indices) GB_TRY(hipMemcpy(dIdx + 3 * (size_t)triBase[k], in[k].indices, sizeof(int) * (size_t)in[k].n_indices, hipMemcpyHostToDevice));
        }
        GB_TRY(hipMemcpy(dTriBase, triBase.data(), sizeof(int) * (size_t)(K + 1), hipMemcpyHostToDevice));
        GB_TRY(hipMemcpy(dVertBase, vertBase.data(), sizeof(int) * (size_t)(K + 1), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_prepare, dim3(blocks(ntri)), dim3(256), 0, 0, dVerts, dIdx, ntri, dTriBase, dVertBase, K, trisA, nodeOfA);
        GB_TRY(hipMemcpy(nodes, roots.data(), sizeof(GNode) * (size_t)K, hipMemcpyHostToDevice));

        lap("alloc + upload");
        std::vector<int> levelFirst;
        int first = 0, nActive = K, total = K;
        size_t tempBytes = 0;
        hipcub::DeviceScan::ExclusiveSum(nullptr, tempBytes, flag, S, ntri + 1);
        void* temp = bTemp.get<char>(tempBytes + 256);
        if (!temp) return RT_ERR_OOM;
        while (nActive > 0) {
            levelFirst.push_back(first);
            GCand* cands = bCands.get<GCand>((size_t)nActive * GB_NCAND);
            int* counts = bCounts.get<int>((size_t)nActive + 1);
            int* base = bBase.get<int>((size_t)nActive + 1);
            int* splitFlag = bSplit.get<int>((size_t)nActive + 1);
            int* splitRank = bRank.get<int>((size_t)nActive + 1);
            GBox* chosen = bChosen.get<GBox>(nActive);
            if (!cands || !counts || !base || !splitFlag || !splitRank || !chosen) return RT_ERR_OOM;
            hipLaunchKernelGGL(k_candidates, dim3(blocks(nActive + 1)), dim3(256), 0, 0, nodes, first, nActive, quality, cands, counts);
            size_t tb = 0;
            hipcub::DeviceScan::ExclusiveSum(nullptr, tb, counts, base, nActive + 1);
            if (tb > tempBytes) { tempBytes = tb; temp = bTemp.get<char>(tempBytes + 256); if (!temp) return RT_ERR_OOM; }
            hipcub::DeviceScan::ExclusiveSum(temp, tb, counts, base, nActive + 1);
            {
                const int maxChunks = nActive + ntri / GB_CHUNK + 1; /* every node with candidates has ceil(count / 512) chunks */
                int* chunkNode = bChunkNode.get<int>(maxChunks);
                GBox* partial = bPartial.get<GBox>((size_t)maxChunks * GB_NCAND);
                if (!chunkNode || !partial) return RT_ERR_OOM;
                hipLaunchKernelGGL(k_chunk_map, dim3(blocks(maxChunks > nActive ? maxChunks : nActive)), dim3(256), 0, 0, nodes, first, nActive, base, maxChunks, chunkNode);
                hipLaunchKernelGGL(k_sweep, dim3(maxChunks), dim3(64), 0, 0, nodes, first, chunkNode, base + nActive, cands, trisA, partial);
                if (nActive <= 8192) /* few, possibly huge nodes: 64 threads each */
                    hipLaunchKernelGGL(k_choose<4>, dim3(blocks((size_t)nActive * 64)), dim3(256), 0, 0, nodes, first, nActive, quality, cands, partial, counts, chosen, splitFlag);
                else
                    hipLaunchKernelGGL(k_choose<1>, dim3(blocks((size_t)nActive * 16)), dim3(256), 0, 0, nodes, first, nActive, quality, cands, partial, counts, chosen, splitFlag);
            }
            tb = 0;
            hipcub::DeviceScan::ExclusiveSum(nullptr, tb, splitFlag, splitRank, nActive + 1);
            if (tb > tempBytes) { tempBytes = tb; temp = bTemp.get<char>(tempBytes + 256); if (!temp) return RT_ERR_OOM; }
            hipcub::DeviceScan::ExclusiveSum(temp, tb, splitFlag, splitRank, nActive + 1);
            int nSplit = 0;
            GB_TRY(hipMemcpy(&nSplit, splitRank + nActive, sizeof(int), hipMemcpyDeviceToHost));
            if ((size_t)total + 2 * (size_t)nSplit > maxNodes) return RT_ERR_SCENE; /* degenerate input: see rt_build_bvh */
            if (nSplit > 0) {
                int* firstR = bFirstR.get<int>(total);
                if (!firstR) return RT_ERR_OOM;
                hipLaunchKernelGGL(k_children, dim3(blocks(nActive)), dim3(256), 0, 0, nodes, first, nActive, splitFlag, splitRank, chosen, total);
                GB_TRY(hipMemsetAsync(firstR, 0xff, sizeof(int) * (size_t)total, 0));
                hipLaunchKernelGGL(k_flags, dim3(blocks(ntri + 1)), dim3(256), 0, 0, nodes, nodeOfA, trisA, ntri, flag);
                tb = tempBytes;
                hipcub::DeviceScan::ExclusiveSum(temp, tb, flag, S, ntri + 1);
                hipLaunchKernelGGL(k_tape, dim3(blocks(ntri)), dim3(256), 0, 0, nodes, nodeOfA, flag, S, ntri, src, firstR);
                hipLaunchKernelGGL(k_tape2, dim3(blocks(ntri)), dim3(256), 0, 0, nodes, nodeOfA, firstR, ntri, src);
                /* pointer jumping: three synchronous passes shorten every chain eightfold, then each position follows the rest */
                for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k_jump, dim3(blocks(ntri)), dim3(256), 0, 0, ntri, src, (int*)nullptr);
                hipLaunchKernelGGL(k_follow, dim3(blocks(ntri)), dim3(256), 0, 0, ntri, src);
                hipLaunchKernelGGL(k_scatter, dim3(blocks(ntri)), dim3(256), 0, 0, nodes, nodeOfA, flag, S, firstR, src, trisA, ntri, trisB, nodeOfB);
                std::swap(trisA, trisB);
                std::swap(nodeOfA, nodeOfB);
            }
            first = total;
            nActive = 2 * nSplit;
            total += 2 * nSplit;
        }
        GB_TRY(hipGetLastError());
        if (dbg) { hipDeviceSynchronize(); lap("levels"); }
        /* numbering, per mesh: every root is pre-order rank 0, node 0 of ITS tree */
        for (int l = (int)levelFirst.size() - 1; l >= 0; l--) {
            const int lf = levelFirst[l], le = (l + 1 < (int)levelFirst.size()) ? levelFirst[l + 1] : total;
            if (le > lf) hipLaunchKernelGGL(k_inner_count, dim3(blocks(le - lf)), dim3(256), 0, 0, nodes, lf, le - lf);
        }
        for (int l = 0; l < (int)levelFirst.size(); l++) {
            const int lf = levelFirst[l], le = (l + 1 < (int)levelFirst.size()) ? levelFirst[l + 1] : total;
            if (le > lf) hipLaunchKernelGGL(k_number, dim3(blocks(le - lf)), dim3(256), 0, 0, nodes, lf, le - lf);
        }
        hipLaunchKernelGGL(k_node_base, dim3(1), dim3(64), 0, 0, nodes, K, dNodeBase);
        RtBVHNode* dOut = bOut.get<RtBVHNode>(total);
        if (!dOut) return RT_ERR_OOM;
        GB_TRY(hipMemcpy(misc, statsH.data(), sizeof(int) * statsH.size(), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_emit, dim3(blocks(total)), dim3(256), 0, 0, nodes, total, dNodeBase, dTriBase, dOut, misc);
        hipLaunchKernelGGL(k_tri_index, dim3(blocks(ntri)), dim3(256), 0, 0, trisA, ntri, S);
        GB_TRY(hipGetLastError());
        GB_TRY(hipMemcpy(nodeBase.data(), dNodeBase, sizeof(int) * (size_t)(K + 1), hipMemcpyDeviceToHost));
        if (nodeBase[K] != total) return RT_ERR_HIP; /* (cannot happen: every node belongs to one mesh's tree) */
        for (int k = 0; k < K; k++) /* a mesh's part of out_nodes holds 2 * its triangles; as rt_build_bvh */
            if ((size_t)(nodeBase[k + 1] - nodeBase[k]) > 2 * (size_t)(triBase[k + 1] - triBase[k])) return RT_ERR_SCENE;
        GB_TRY(hipMemcpy(order.data(), S, sizeof(int) * (size_t)ntri, hipMemcpyDeviceToHost));
        start_gather();
        GB_TRY(hipMemcpy(out_nodes, dOut, sizeof(RtBVHNode) * (size_t)total, hipMemcpyDeviceToHost));
        nodesInPlace = true;
        GB_TRY(hipMemcpy(statsH.data(), misc, sizeof(int) * statsH.size(), hipMemcpyDeviceToHost));
        lap("numbering + readback");
    }
    for (int k = 0; k < K; k++) {
        const int* st = &statsH[(size_t)8 * k];
        const int nt = triBase[k + 1] - triBase[k];
        if ((size_t)(nodeBase[k + 1] - nodeBase[k]) > 2 * (size_t)(nt > 0 ? nt : 1) || (nt > 0 && st[0] > 0 && st[5] == 0)) { /* as rt_build_bvh */
            for (int q = 0; q < K; q++) out_n_nodes[q] = 0;
            return RT_ERR_SCENE;
        }
    }
    if (!gatherStarted) start_gather();
    if (!nodesInPlace) memcpy(out_nodes, nodesOut.data(), nodesOut.size() * sizeof(RtBVHNode));
    for (auto& t : gatherThreads) t.join();
    gatherThreads.clear();
    lap("triangle gather");
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int k = 0; k < K; k++) {
        out_n_nodes[k] = nodeBase[k + 1] - nodeBase[k];
        if (out_node_offset) out_node_offset[k] = nodeBase[k];
        if (out_tri_offset) out_tri_offset[k] = triBase[k];
        if (out_stats) {
            const int* st = &statsH[(size_t)8 * k];
            RtBvhStats& o = out_stats[k];
            memset(&o, 0, sizeof(o));
            o.triangleCount = st[6];
            o.totalNodeCount = out_n_nodes[k] - (quality == RT_BVH_QUALITY_DISABLED ? 1 : 0);
            o.leafNodeCount = st[0];
            o.leafDepthMax = st[2];
            o.leafDepthMin = st[3];
            o.leafDepthSum = st[1];
            o.leafMaxTriCount = st[4];
            o.leafMinTriCount = st[5];
            o.quality = quality;
            o.timeMs = ntri > 0 ? ms * (double)(triBase[k + 1] - triBase[k]) / (double)ntri : ms; /* the batch's time, by share of the triangles */
        }
    }
    return RT_OK;
}

/* pool lock, caller's device restored, scratch released when it grew beyond GB_POOL_KEEP */
static int build_locked(int device, int K, const MeshIn* in, int quality, RtBVHNode* out_nodes, int* out_n_nodes, int* out_node_offset,
                        RtTriangle* out_tris, int* out_tri_offset, RtBvhStats* out_stats)
{
    std::lock_guard<std::mutex> poolLock(g_poolMutex);
    int prev = -1;
    const bool havePrev = hipGetDevice(&prev) == hipSuccess; /* the caller's current device is the caller's business: put it back */
    if (out_n_nodes) for (int k = 0; k < K; k++) out_n_nodes[k] = 0; /* every error path leaves 0 nodes */
    const int rc = build_forest(device, K, in, quality, out_nodes, out_n_nodes, out_node_offset, out_tris, out_tri_offset, out_stats);
    if (rc != RT_OK && out_n_nodes) for (int k = 0; k < K; k++) out_n_nodes[k] = 0;
    size_t held = 0;
    for (const DevBuf& d : g_pool.b) held += d.cap;
    if (held > GB_POOL_KEEP) pool_release(); /* a huge mesh does not pin its scratch for the life of the process */
    if (havePrev) hipSetDevice(prev);
    return rc;
}

int build(int device, const float* verts, const float* normals, int n_verts, const int32_t* indices, int n_indices, int quality,
          RtBVHNode* out_nodes, int* out_n_nodes, RtTriangle* out_tris, RtBvhStats* out_stats)
{
    if (!out_n_nodes) return RT_ERR_INVALID_ARG;
    const MeshIn m = {verts, normals, n_verts, indices, n_indices};
    return build_locked(device, 1, &m, quality, out_nodes, out_n_nodes, nullptr, out_tris, nullptr, out_stats);
}

} // namespace gbvh

extern "C" void rt_build_bvh_gpu_release(void)
{
    std::lock_guard<std::mutex> poolLock(gbvh::g_poolMutex);
    gbvh::pool_release();
}

/* The meshes of a scene in one call (CreateAllMeshData, RCM:206-236): mesh k's nodes and triangles are written behind mesh
 * k-1's, i.e. the arrays come out concatenated the way the dispatcher uploads them (offsets returned per mesh). */
extern "C" int rt_build_bvh_gpu_batch(int device_id, int n_meshes, const float* const* verts, const float* const* normals, const int* n_verts,
                                      const int32_t* const* indices, const int* n_indices, int quality, RtBVHNode* out_nodes, int* out_n_nodes,
                                      int* out_node_offset, RtTriangle* out_tris, int* out_tri_offset, RtBvhStats* out_stats)
{
    if (n_meshes < 0 || (n_meshes > 0 && (!verts || !normals || !n_verts || !indices || !n_indices || !out_nodes || !out_n_nodes || !out_node_offset ||
                                          !out_tris || !out_tri_offset)))
        return RT_ERR_INVALID_ARG;
    if (n_meshes == 0) return RT_OK;
    bool forest = n_meshes > 1 && quality != RT_BVH_QUALITY_DISABLED && !getenv("RT_BVH_NO_FOREST");
    long long tris = 0;
    for (int k = 0; k < n_meshes; k++) {
        if (n_indices[k] < 0 || n_indices[k] % 3) return RT_ERR_INVALID_ARG;
        if (n_indices[k] == 0) forest = false; /* an empty mesh's one-node tree is made on the host */
        tris += n_indices[k] / 3;
    }
    if (tris > 0x2aaaaaaaLL) forest = false;
    if (forest) {
        /* all meshes as one forest: every level's kernels run once for the whole scene */
        std::vector<gbvh::MeshIn> in(n_meshes);
        for (int k = 0; k < n_meshes; k++) in[k] = {verts[k], normals[k], n_verts[k], indices[k], n_indices[k]};
        const int rc = gbvh::build_locked(device_id, n_meshes, in.data(), quality, out_nodes, out_n_nodes, out_node_offset, out_tris, out_tri_offset, out_stats);
        if (rc == RT_OK) return rc;
        /* refused (an index out of range, a degenerate mesh, no memory for the whole scene at once): mesh by mesh, so that the
         * status and the meshes written before the offending one are what they always were */
    }
    long long nodeOff = 0, triOff = 0;
    for (int k = 0; k < n_meshes; k++) {
        out_node_offset[k] = (int)nodeOff;
        out_tri_offset[k] = (int)triOff;
        out_n_nodes[k] = 0;
        const int rc = gbvh::build(device_id, verts[k], normals[k], n_verts[k], indices[k], n_indices[k], quality, out_nodes + nodeOff, &out_n_nodes[k],
                                   out_tris + triOff, out_stats ? &out_stats[k] : nullptr);
        if (rc != RT_OK) return rc;
        nodeOff += out_n_nodes[k];
        triOff += n_indices[k] / 3;
        if (nodeOff > 0x7fffffffLL || triOff > 0x7fffffffLL) return RT_ERR_SCENE;
    }
    return RT_OK;
}

extern "C" int rt_build_bvh_gpu(int device_id, const float* verts, const float* normals, int n_verts, const int32_t* indices, int n_indices,
                                int quality, RtBVHNode* out_nodes, int* out_n_nodes, RtTriangle* out_tris, RtBvhStats* out_stats)
{
    return gbvh::build(device_id, verts, normals, n_verts, indices, n_indices, quality, out_nodes, out_n_nodes, out_tris, out_stats);
}
