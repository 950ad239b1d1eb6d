/*
 * ref_bvh_driver.h — C entry point over the reference's BVH class as compiled by oracle/make_ref.py (TEST INFRASTRUCTURE ONLY).
 * Signature and refusals of rt_build_bvh (include/rt_abi.h): the caller's flat float / int arrays are viewed as the constructor's
 * Vector3[] / int[] arguments (BVH.cs:26), Nodes / Triangles / stats are copied out field by field.
 */
#pragma once
#include "../include/rt_abi.h"

static_assert(sizeof(Vector3) == 12, "Vector3 is three floats");

extern "C" int ref_bvh_build(const float* verts, const float* normals, int n_verts, const int32_t* indices, int n_indices, int quality,
                             RtBVHNode* out_nodes, int* out_n_nodes, RtTriangle* out_tris, RtBvhStats* out_stats)
{
    using Seb::AccelerationStructures::BVH;
    if (!verts || !normals || !indices || !out_nodes || !out_n_nodes || !out_tris || n_indices < 0 || n_indices % 3) return RT_ERR_INVALID_ARG;
    if (quality != RT_BVH_QUALITY_LOW && quality != RT_BVH_QUALITY_HIGH && quality != RT_BVH_QUALITY_DISABLED) return RT_ERR_INVALID_ARG;
    for (int i = 0; i < n_indices; i++)
        if (indices[i] < 0 || indices[i] >= n_verts) return RT_ERR_INVALID_ARG; /* C# would throw IndexOutOfRangeException */
    *out_n_nodes = 0;
    const int ntri = n_indices / 3;
    int rc = RT_OK;
    {
        CsArray<Vector3> v(reinterpret_cast<Vector3*>(const_cast<float*>(verts)), n_verts);
        CsArray<Vector3> nrm(reinterpret_cast<Vector3*>(const_cast<float*>(normals)), n_verts);
        CsArray<int> idx(const_cast<int*>(indices), n_indices);
        const BVH::Quality q = quality == RT_BVH_QUALITY_LOW ? BVH::Quality::Low : quality == RT_BVH_QUALITY_HIGH ? BVH::Quality::High : BVH::Quality::Disabled;
        /* BVH.cs's node list is unbounded: a mesh whose every split cost overflows grows a chain of empty leaves 32 levels deep per
         * branch (see rt_build_bvh).  The recursion is bounded by MaxDepth, so the build always ends. */
        BVH bvh(v, idx, nrm, q);
        const int nNodes = bvh.Nodes.Length;
        /* rt_build_bvh's refusals: more nodes than the caller's 2 * max(1, triangles), or an empty leaf (malformed for RC:246) */
        if ((size_t)nNodes > 2 * (size_t)(ntri > 0 ? ntri : 1) || (ntri > 0 && bvh.stats.LeafMinTriCount == 0)) {
            rc = RT_ERR_SCENE;
        } else {
            for (int i = 0; i < nNodes; i++) {
                const BVH::Node& n = bvh.Nodes[i];
                RtBVHNode o = {{n.MinX, n.MinY, n.MinZ}, {n.MaxX, n.MaxY, n.MaxZ}, n.StartIndex, n.TriangleCount};
                out_nodes[i] = o;
            }
            for (int i = 0; i < bvh.Triangles.Length; i++) {
                const BVH::Triangle& t = bvh.Triangles[i];
                RtTriangle o = {{t.A.x, t.A.y, t.A.z}, {t.B.x, t.B.y, t.B.z}, {t.C.x, t.C.y, t.C.z},
                                {t.normalA.x, t.normalA.y, t.normalA.z}, {t.normalB.x, t.normalB.y, t.normalB.z}, {t.normalC.x, t.normalC.y, t.normalC.z}};
                out_tris[i] = o;
            }
            *out_n_nodes = nNodes;
            if (out_stats) {
                memset(out_stats, 0, sizeof(*out_stats));
                out_stats->triangleCount = bvh.stats.TriangleCount;
                out_stats->totalNodeCount = bvh.stats.TotalNodeCount;
                out_stats->leafNodeCount = bvh.stats.LeafNodeCount;
                out_stats->leafDepthMax = bvh.stats.LeafDepthMax;
                out_stats->leafDepthMin = bvh.stats.LeafDepthMin;
                out_stats->leafDepthSum = bvh.stats.LeafDepthSum;
                out_stats->leafMaxTriCount = bvh.stats.LeafMaxTriCount;
                out_stats->leafMinTriCount = bvh.stats.LeafMinTriCount;
                out_stats->quality = quality;
                out_stats->timeMs = (double)bvh.stats.TimeMs;
            }
        }
    }
    cs_collect();
    return rc;
}

extern "C" const char* ref_bvh_version(void) { return "reference BVH.cs:26-318 compiled as C++ (oracle/make_ref.py)"; }
