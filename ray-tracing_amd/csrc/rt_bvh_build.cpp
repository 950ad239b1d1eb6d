/*
 * rt_bvh_build.cpp — host helpers of libraytrace_hip.so that produce the buffers
 * the kernel consumes: rt_build_bvh (≙ the BVH constructor, Assets/Scripts/Types/
 * BVH.cs:26-318) and rt_camera_view_params (≙ RayComputeManager.cs:183-190).
 *
 * The tree must be the reference's tree, node for node and triangle for triangle
 * (leaf order decides closest-hit ties, RC:256), so the split rule, the candidate
 * planes, the strict comparisons and the in-place partition are the reference's.
 * The structure of the computation is not: the recursion is an explicit work
 * stack, and all candidate planes of a node (up to 15) are scored in ONE sweep
 * over the node's triangles instead of one sweep per candidate (EvaluateSplit,
 * BVH:253-311, is called up to 15 times per node in the reference).  Each
 * candidate keeps the reference's sequential "if (t < cur) cur = t" updates in
 * triangle order, so even the sign of a zero bound comes out identical.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "../../include/rt_abi.h"

namespace {

const float FMAX = 3.40282347e+38f; /* float.MaxValue; float.MinValue == -FMAX */

struct BuildTri { /* BVH:459-496 */
    float c[3];
    float mn[3];
    float mx[3];
    int index;
};

struct Box {
    float mn[3], mx[3];
    void reset() { mn[0] = mn[1] = mn[2] = FMAX; mx[0] = mx[1] = mx[2] = -FMAX; }
    void grow(const BuildTri& t)
    {
        for (int k = 0; k < 3; k++) {
            if (t.mn[k] < mn[k]) mn[k] = t.mn[k];
            if (t.mx[k] > mx[k]) mx[k] = t.mx[k];
        }
    }
};

inline float node_cost(const Box& b, int n) /* BVH:313-318 applied to max-min sizes */
{
    if (n == 0) return 0;
    float x = b.mx[0] - b.mn[0], y = b.mx[1] - b.mn[1], z = b.mx[2] - b.mn[2];
    float area = x * y + x * z + y * z;
    return area * n;
}
inline float size_cost(float x, float y, float z, int n)
{
    if (n == 0) return 0;
    float area = x * y + x * z + y * z;
    return area * n;
}
inline float min3(float a, float b, float c) { return a < b ? (a < c ? a : c) : (b < c ? b : c); }
inline float max3(float a, float b, float c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

struct Candidate {
    int axis;
    float pos;
    Box left, right;
    int nLeft, nRight;
};

struct Work {
    int node, start, count, depth;
};

} // namespace

extern "C" int rt_build_bvh(const float* verts, const float* normals, int n_verts, const int32_t* indices, int n_indices, int quality,
                            RtBVHNode* out_nodes, int* out_n_nodes, RtTriangle* out_tris, RtBvhStats* out_stats)
{
    if (!verts || !normals || !indices || !out_nodes || !out_n_nodes || !out_tris || n_verts < 0 || n_indices < 0 || n_indices % 3)
        return RT_ERR_INVALID_ARG;
    if (quality != RT_BVH_QUALITY_LOW && quality != RT_BVH_QUALITY_HIGH && quality != RT_BVH_QUALITY_DISABLED) return RT_ERR_INVALID_ARG;
    for (int i = 0; i < n_indices; i++)
        if (indices[i] < 0 || indices[i] >= n_verts) return RT_ERR_INVALID_ARG;
    auto t0 = std::chrono::steady_clock::now();

    const int ntri = n_indices / 3;
    std::vector<BuildTri> tris(ntri);
    Box rootBox;
    rootBox.reset();
    for (int t = 0; t < ntri; t++) { /* BVH:44-59 */
        const float* a = verts + 3 * indices[3 * t + 0];
        const float* b = verts + 3 * indices[3 * t + 1];
        const float* c = verts + 3 * indices[3 * t + 2];
        BuildTri& bt = tris[t];
        for (int k = 0; k < 3; k++) {
            bt.c[k] = (a[k] + b[k] + c[k]) / 3;
            bt.mn[k] = min3(a[k], b[k], c[k]);
            bt.mx[k] = max3(a[k], b[k], c[k]);
        }
        bt.index = 3 * t;
        rootBox.grow(bt);
    }

    RtBvhStats st;
    memset(&st, 0, sizeof(st));
    st.leafDepthMin = INT32_MAX;
    st.leafMinTriCount = INT32_MAX;
    st.quality = quality;

    std::vector<RtBVHNode> nodes;
    nodes.reserve(2 * (size_t)(ntri > 0 ? ntri : 1));
    RtBVHNode root;
    memcpy(root.boundsMin, rootBox.mn, 12);
    memcpy(root.boundsMax, rootBox.mx, 12);
    root.startIndex = -1; /* BVH:61 — an inner root keeps triangleCount == -1 */
    root.triangleCount = -1;
    nodes.push_back(root);

    auto record_leaf = [&](int depth, int n) { /* BVH:539-554 */
        st.totalNodeCount++;
        st.leafNodeCount++;
        st.leafDepthSum += depth;
        if (depth < st.leafDepthMin) st.leafDepthMin = depth;
        if (depth > st.leafDepthMax) st.leafDepthMax = depth;
        st.triangleCount += n;
        if (n > st.leafMaxTriCount) st.leafMaxTriCount = n;
        if (n < st.leafMinTriCount) st.leafMinTriCount = n;
    };

    if (quality == RT_BVH_QUALITY_DISABLED) { /* BVH:62-66 */
        nodes[0].startIndex = 0;
        nodes[0].triangleCount = ntri;
    } else {
        const int MaxDepth = 32; /* BVH:91 */
        std::vector<Work> work;
        work.push_back({0, 0, ntri, 0});
        Candidate cand[15];
        while (!work.empty()) {
            Work w = work.back();
            work.pop_back();
            RtBVHNode parent = nodes[w.node];
            float sizeX = parent.boundsMax[0] - parent.boundsMin[0];
            float sizeY = parent.boundsMax[1] - parent.boundsMin[1];
            float sizeZ = parent.boundsMax[2] - parent.boundsMin[2];
            float parentCost = size_cost(sizeX, sizeY, sizeZ, w.count);

            /* ---- ChooseSplit (BVH:183-250): list the candidate planes in evaluation order */
            int nc = 0;
            if (w.count > 1) {
                float size[3] = {sizeX, sizeY, sizeZ};
                if (quality == RT_BVH_QUALITY_LOW) {
                    int ax = (sizeX > sizeY && sizeX > sizeZ) ? 0 : (sizeY > sizeZ ? 1 : 2);
                    cand[nc].axis = ax;
                    cand[nc].pos = parent.boundsMin[ax] + size[ax] * 0.5f;
                    nc++;
                } else {
                    int maxSplitTests = w.count < 10 ? 3 : 5;
                    float maxAxis = max3(sizeX, sizeY, sizeZ);
                    for (int axis = 0; axis < 3; axis++) {
                        float v = size[axis] / maxAxis * maxSplitTests;
                        int n = (v != v) ? INT32_MIN : (int)ceilf(v); /* CeilToInt(NaN) == int.MinValue */
                        n = n < 1 ? 1 : (n > maxSplitTests ? maxSplitTests : n);
                        for (int i = 0; i < n; i++) {
                            float splitT = (i + 1) / (n + 1.0f);
                            cand[nc].axis = axis;
                            cand[nc].pos = parent.boundsMin[axis] + size[axis] * splitT;
                            nc++;
                        }
                    }
                }
            }
            /* ---- score every candidate in one sweep (EvaluateSplit, BVH:253-311) */
            for (int j = 0; j < nc; j++) {
                cand[j].left.reset();
                cand[j].right.reset();
                cand[j].nLeft = cand[j].nRight = 0;
            }
            const int end = w.start + w.count;
            for (int i = w.start; i < end; i++) {
                const BuildTri& t = tris[i];
                for (int j = 0; j < nc; j++) {
                    Candidate& cd = cand[j];
                    if (t.c[cd.axis] < cd.pos) {
                        cd.left.grow(t);
                        cd.nLeft++;
                    } else {
                        cd.right.grow(t);
                        cd.nRight++;
                    }
                }
            }
            int best = -1;
            float bestCost = (quality == RT_BVH_QUALITY_LOW) ? 0.0f : FMAX;
            float cost = INFINITY; /* count <= 1: BVH:185 */
            if (quality == RT_BVH_QUALITY_LOW) {
                if (nc) { best = 0; cost = node_cost(cand[0].left, cand[0].nLeft) + node_cost(cand[0].right, cand[0].nRight); }
            } else if (nc) {
                /* bestSplitAxis/Pos default to (0, 0) if no candidate beats float.MaxValue (BVH:204-208) */
                for (int j = 0; j < nc; j++) {
                    float cj = node_cost(cand[j].left, cand[j].nLeft) + node_cost(cand[j].right, cand[j].nRight);
                    if (cj < bestCost) {
                        bestCost = cj;
                        best = j;
                    }
                }
                cost = bestCost;
            }

            if (cost < parentCost && w.depth < MaxDepth) { /* BVH:101 */
                int splitAxis = best >= 0 ? cand[best].axis : 0;
                float splitPos = best >= 0 ? cand[best].pos : 0.0f;
                /* in-place partition in the reference's order (BVH:118-152) */
                Box L, R;
                L.reset();
                R.reset();
                int numOnLeft = 0;
                for (int i = w.start; i < end; i++) {
                    BuildTri t = tris[i];
                    if (t.c[splitAxis] < splitPos) {
                        L.grow(t);
                        tris[i] = tris[w.start + numOnLeft];
                        tris[w.start + numOnLeft] = t;
                        numOnLeft++;
                    } else {
                        R.grow(t);
                    }
                }
                int numOnRight = w.count - numOnLeft;
                RtBVHNode cl, cr;
                memcpy(cl.boundsMin, L.mn, 12); memcpy(cl.boundsMax, L.mx, 12);
                cl.startIndex = w.start; cl.triangleCount = 0;
                memcpy(cr.boundsMin, R.mn, 12); memcpy(cr.boundsMax, R.mx, 12);
                cr.startIndex = w.start + numOnLeft; cr.triangleCount = 0;
                int li = (int)nodes.size();
                nodes.push_back(cl);
                nodes.push_back(cr);
                nodes[w.node].startIndex = li; /* BVH:165 */
                st.totalNodeCount++;
                /* depth-first, left subtree first: push right, then left */
                work.push_back({li + 1, w.start + numOnLeft, numOnRight, w.depth + 1});
                work.push_back({li, w.start, numOnLeft, w.depth + 1});
            } else { /* BVH:173-180 */
                nodes[w.node].startIndex = w.start;
                nodes[w.node].triangleCount = w.count;
                record_leaf(w.depth, w.count);
            }
        }
    }

    for (int i = 0; i < ntri; i++) { /* BVH:69-80: triangles in leaf order with vertex normals */
        int base = tris[i].index;
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
    memcpy(out_nodes, nodes.data(), nodes.size() * sizeof(RtBVHNode));
    *out_n_nodes = (int)nodes.size();
    if (out_stats) {
        *out_stats = st;
        out_stats->timeMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RT_OK;
}

/* RCM:185-188.  UnityEngine.Mathf.Tan is (float)Math.Tan(double). */
extern "C" int rt_camera_view_params(float fov_deg, float aspect, float focus_distance, float out[3])
{
    if (!out) return RT_ERR_INVALID_ARG;
    const float Deg2Rad = 0.0174532924f;
    float planeHeight = focus_distance * (float)tan((double)(fov_deg * 0.5f * Deg2Rad)) * 2;
    float planeWidth = planeHeight * aspect;
    out[0] = planeWidth;
    out[1] = planeHeight;
    out[2] = focus_distance;
    return RT_OK;
}
