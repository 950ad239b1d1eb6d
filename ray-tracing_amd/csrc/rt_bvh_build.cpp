/*
 * rt_bvh_build.cpp — host helpers of libraytrace_hip.so that produce the buffers
 * the kernel consumes: rt_build_bvh / rt_build_bvh_mt (≙ the BVH constructor,
 * Assets/Scripts/Types/BVH.cs:26-318) and rt_camera_view_params (≙
 * RayComputeManager.cs:183-190).
 *
 * The tree must be the reference's tree, node for node and triangle for triangle
 * (leaf order decides closest-hit ties, RC:256), so the split rule, the candidate
 * planes, the strict comparisons and the in-place partition are the reference's.
 * The structure of the computation is not:
 *   - the recursion is an explicit work stack;
 *   - the build is multi-threaded: the sweep of a big node is split into chunks that
 *     are reduced IN CHUNK ORDER (the reference's sequential "if (t < cur) cur = t"
 *     keeps the first of equal values — signed zeros included — and so does an ordered
 *     reduction of ordered chunks), and once the top of the tree has been cut into
 *     enough subtrees each subtree is built by one thread into a private node list.
 *     Node indices are assigned afterwards by replaying the reference's allocation
 *     order (children allocated at the split, left subtree before right subtree), so
 *     the output is byte-identical for any thread count (tests/test_bvh.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <chrono>
#include <thread>
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
    /* append the box of a LATER chunk: same "first of equals stays" rule as grow() */
    void append(const Box& o)
    {
        for (int k = 0; k < 3; k++) {
            if (o.mn[k] < mn[k]) mn[k] = o.mn[k];
            if (o.mx[k] > mx[k]) mx[k] = o.mx[k];
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

struct Stats { /* BVH:518-576, mergeable */
    int triangleCount = 0, totalNodeCount = 0, leafNodeCount = 0;
    int leafDepthMax = 0, leafDepthMin = INT32_MAX, leafDepthSum = 0;
    int leafMaxTriCount = 0, leafMinTriCount = INT32_MAX;
    void leaf(int depth, int n)
    {
        totalNodeCount++;
        leafNodeCount++;
        leafDepthSum += depth;
        if (depth < leafDepthMin) leafDepthMin = depth;
        if (depth > leafDepthMax) leafDepthMax = depth;
        triangleCount += n;
        if (n > leafMaxTriCount) leafMaxTriCount = n;
        if (n < leafMinTriCount) leafMinTriCount = n;
    }
    void inner() { totalNodeCount++; }
    void merge(const Stats& o)
    {
        triangleCount += o.triangleCount; totalNodeCount += o.totalNodeCount; leafNodeCount += o.leafNodeCount;
        leafDepthSum += o.leafDepthSum;
        if (o.leafDepthMax > leafDepthMax) leafDepthMax = o.leafDepthMax;
        if (o.leafDepthMin < leafDepthMin) leafDepthMin = o.leafDepthMin;
        if (o.leafMaxTriCount > leafMaxTriCount) leafMaxTriCount = o.leafMaxTriCount;
        if (o.leafMinTriCount < leafMinTriCount) leafMinTriCount = o.leafMinTriCount;
    }
};

/* ChooseSplit (BVH:183-250): the candidate planes of a node, in evaluation order */
int list_candidates(const RtBVHNode& node, int count, int quality, Candidate* cand)
{
    if (count <= 1) return 0; /* BVH:185 */
    const float size[3] = {node.boundsMax[0] - node.boundsMin[0], node.boundsMax[1] - node.boundsMin[1], node.boundsMax[2] - node.boundsMin[2]};
    int nc = 0;
    if (quality == RT_BVH_QUALITY_LOW) {
        int ax = (size[0] > size[1] && size[0] > size[2]) ? 0 : (size[1] > size[2] ? 1 : 2);
        cand[nc].axis = ax;
        cand[nc].pos = node.boundsMin[ax] + size[ax] * 0.5f;
        return 1;
    }
    const int maxSplitTests = count < 10 ? 3 : 5;
    const float maxAxis = max3(size[0], size[1], size[2]);
    for (int axis = 0; axis < 3; axis++) {
        float v = size[axis] / maxAxis * maxSplitTests;
        int n = (v != v) ? INT32_MIN : (int)ceilf(v); /* CeilToInt(NaN) == int.MinValue */
        n = n < 1 ? 1 : (n > maxSplitTests ? maxSplitTests : n);
        for (int i = 0; i < n; i++) {
            float splitT = (i + 1) / (n + 1.0f);
            cand[nc].axis = axis;
            cand[nc].pos = node.boundsMin[axis] + size[axis] * splitT;
            nc++;
        }
    }
    return nc;
}

/* EvaluateSplit (BVH:253-311) for every candidate over tris[begin,end), sequential update order.
 * One tight pass per candidate with the twelve running bounds in registers (a single pass
 * updating all 15 candidates' boxes in memory measured slower). */
void sweep(const BuildTri* tris, int begin, int end, Candidate* cand, int nc)
{
    for (int j = 0; j < nc; j++) {
        const int axis = cand[j].axis;
        const float pos = cand[j].pos;
        float lmn0 = FMAX, lmn1 = FMAX, lmn2 = FMAX, lmx0 = -FMAX, lmx1 = -FMAX, lmx2 = -FMAX;
        float rmn0 = FMAX, rmn1 = FMAX, rmn2 = FMAX, rmx0 = -FMAX, rmx1 = -FMAX, rmx2 = -FMAX;
        int nl = 0;
        for (int i = begin; i < end; i++) {
            const BuildTri& t = tris[i];
            if (t.c[axis] < pos) {
                if (t.mn[0] < lmn0) lmn0 = t.mn[0];
                if (t.mn[1] < lmn1) lmn1 = t.mn[1];
                if (t.mn[2] < lmn2) lmn2 = t.mn[2];
                if (t.mx[0] > lmx0) lmx0 = t.mx[0];
                if (t.mx[1] > lmx1) lmx1 = t.mx[1];
                if (t.mx[2] > lmx2) lmx2 = t.mx[2];
                nl++;
            } else {
                if (t.mn[0] < rmn0) rmn0 = t.mn[0];
                if (t.mn[1] < rmn1) rmn1 = t.mn[1];
                if (t.mn[2] < rmn2) rmn2 = t.mn[2];
                if (t.mx[0] > rmx0) rmx0 = t.mx[0];
                if (t.mx[1] > rmx1) rmx1 = t.mx[1];
                if (t.mx[2] > rmx2) rmx2 = t.mx[2];
            }
        }
        Candidate& cd = cand[j];
        cd.left.mn[0] = lmn0; cd.left.mn[1] = lmn1; cd.left.mn[2] = lmn2; cd.left.mx[0] = lmx0; cd.left.mx[1] = lmx1; cd.left.mx[2] = lmx2;
        cd.right.mn[0] = rmn0; cd.right.mn[1] = rmn1; cd.right.mn[2] = rmn2; cd.right.mx[0] = rmx0; cd.right.mx[1] = rmx1; cd.right.mx[2] = rmx2;
        cd.nLeft = nl;
        cd.nRight = (end - begin) - nl;
    }
}

/* the same sweep with the range cut into ordered chunks, one thread each, reduced in order */
void sweep_parallel(const BuildTri* tris, int begin, int end, Candidate* cand, int nc, int threads)
{
    const int n = end - begin;
    if (threads > n / 16384) threads = n / 16384; /* at least 16k triangles per chunk */
    if (threads <= 1 || nc == 0) {
        sweep(tris, begin, end, cand, nc);
        return;
    }
    std::vector<Candidate> part((size_t)threads * 15);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) {
        Candidate* mine = &part[(size_t)t * 15];
        for (int j = 0; j < nc; j++) { mine[j].axis = cand[j].axis; mine[j].pos = cand[j].pos; }
        const int b = begin + (int)((long long)n * t / threads), e = begin + (int)((long long)n * (t + 1) / threads);
        pool.emplace_back([=] { sweep(tris, b, e, mine, nc); });
    }
    for (auto& th : pool) th.join();
    for (int j = 0; j < nc; j++) {
        cand[j] = part[j];
        for (int t = 1; t < threads; t++) {
            const Candidate& p = part[(size_t)t * 15 + j];
            cand[j].left.append(p.left);
            cand[j].right.append(p.right);
            cand[j].nLeft += p.nLeft;
            cand[j].nRight += p.nRight;
        }
    }
}

/* One node of the reference's Split (BVH:89-181): decide, and partition in place if it splits.
 * Returns true with the two child records filled in. */
bool split_node(BuildTri* tris, const RtBVHNode& parent, int start, int count, int depth, int quality, int sweepThreads,
                RtBVHNode* cl, RtBVHNode* cr, int* numOnLeftOut)
{
    const int MaxDepth = 32; /* BVH:91 */
    const float sizeX = parent.boundsMax[0] - parent.boundsMin[0];
    const float sizeY = parent.boundsMax[1] - parent.boundsMin[1];
    const float sizeZ = parent.boundsMax[2] - parent.boundsMin[2];
    const float parentCost = size_cost(sizeX, sizeY, sizeZ, count);

    Candidate cand[15];
    const int nc = list_candidates(parent, count, quality, cand);
    sweep_parallel(tris, start, start + count, cand, nc, sweepThreads);

    int best = -1;
    float cost = INFINITY; /* count <= 1: BVH:185 */
    if (quality == RT_BVH_QUALITY_LOW) {
        if (nc) { best = 0; cost = node_cost(cand[0].left, cand[0].nLeft) + node_cost(cand[0].right, cand[0].nRight); }
    } else if (nc) {
        /* bestSplitAxis/Pos stay (0, 0) if no candidate beats float.MaxValue (BVH:204-208) */
        float bestCost = FMAX;
        for (int j = 0; j < nc; j++) {
            float cj = node_cost(cand[j].left, cand[j].nLeft) + node_cost(cand[j].right, cand[j].nRight);
            if (cj < bestCost) {
                bestCost = cj;
                best = j;
            }
        }
        cost = bestCost;
    }
    if (!(cost < parentCost && depth < MaxDepth)) return false; /* BVH:101 */

    const int splitAxis = best >= 0 ? cand[best].axis : 0;
    const float splitPos = best >= 0 ? cand[best].pos : 0.0f;
    /* in-place partition in the reference's order (BVH:118-152) */
    Box L, R;
    L.reset();
    R.reset();
    int numOnLeft = 0;
    const int end = start + count;
    for (int i = start; i < end; i++) {
        BuildTri t = tris[i];
        if (t.c[splitAxis] < splitPos) {
            L.grow(t);
            tris[i] = tris[start + numOnLeft];
            tris[start + numOnLeft] = t;
            numOnLeft++;
        } else {
            R.grow(t);
        }
    }
    memcpy(cl->boundsMin, L.mn, 12); memcpy(cl->boundsMax, L.mx, 12);
    cl->startIndex = start; cl->triangleCount = 0;
    memcpy(cr->boundsMin, R.mn, 12); memcpy(cr->boundsMax, R.mx, 12);
    cr->startIndex = start + numOnLeft; cr->triangleCount = 0;
    *numOnLeftOut = numOnLeft;
    return true;
}

/* fn(begin, end) over [0,n) cut into one contiguous chunk per thread */
template <typename F>
void parallel_chunks(int n, int threads, F fn)
{
    if (threads > n / 8192) threads = n / 8192;
    if (threads <= 1) { fn(0, n); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) {
        const int b = (int)((long long)n * t / threads), e = (int)((long long)n * (t + 1) / threads);
        pool.emplace_back([=] { fn(b, e); });
    }
    for (auto& th : pool) th.join();
}

struct Work {
    int node, start, count, depth;
};

/* Serial build of the subtree whose root record is nodes[0] (already holding its bounds);
 * nodes are appended in the reference's allocation order, indices local to `nodes`. */
void build_subtree(BuildTri* tris, std::vector<RtBVHNode>& nodes, int start, int count, int depth, int quality, Stats& st,
                   std::vector<uint8_t>* innerFlags = nullptr)
{
    std::vector<Work> work;
    work.push_back({0, start, count, depth});
    while (!work.empty()) {
        Work w = work.back();
        work.pop_back();
        RtBVHNode cl, cr;
        int numOnLeft = 0;
        if (split_node(tris, nodes[w.node], w.start, w.count, w.depth, quality, 1, &cl, &cr, &numOnLeft)) {
            int li = (int)nodes.size();
            nodes.push_back(cl);
            nodes.push_back(cr);
            nodes[w.node].startIndex = li; /* BVH:165 */
            if (innerFlags) {
                innerFlags->resize(nodes.size(), 0);
                (*innerFlags)[w.node] = 1;
            }
            st.inner();
            /* depth-first, left subtree first: push right, then left */
            work.push_back({li + 1, w.start + numOnLeft, w.count - numOnLeft, w.depth + 1});
            work.push_back({li, w.start, numOnLeft, w.depth + 1});
        } else { /* BVH:173-180 */
            nodes[w.node].startIndex = w.start;
            nodes[w.node].triangleCount = w.count;
            st.leaf(w.depth, w.count);
        }
    }
}

/* top of the tree while it is being cut into per-thread subtrees */
struct TopNode {
    RtBVHNode rec;
    int start, count, depth;
    int child[2] = {-1, -1}; /* indices into the top-node list, -1: none */
    int task = -1;           /* subtree task index if this node was handed to a thread */
    bool leaf = false;
};
struct Task {
    int top;
    std::vector<RtBVHNode> nodes; /* local subtree, nodes[0] = its root */
    std::vector<uint8_t> inner;   /* nodes[k] is an inner node (its startIndex is a local node index) */
    Stats st;
};

int default_threads()
{
    if (const char* e = getenv("RT_BVH_THREADS")) {
        int n = atoi(e);
        if (n >= 1) return n > 64 ? 64 : n;
    }
    unsigned hc = std::thread::hardware_concurrency();
    int n = hc ? (int)hc : 1;
    return n > 16 ? 16 : n;
}

int build(const float* verts, const float* normals, int n_verts, const int32_t* indices, int n_indices, int quality, int threads,
          RtBVHNode* out_nodes, int* out_n_nodes, RtTriangle* out_tris, RtBvhStats* out_stats)
{
    if (!verts || !normals || !indices || !out_nodes || !out_n_nodes || !out_tris || n_verts < 0 || n_indices < 0 || n_indices % 3)
        return RT_ERR_INVALID_ARG;
    if (quality != RT_BVH_QUALITY_LOW && quality != RT_BVH_QUALITY_HIGH && quality != RT_BVH_QUALITY_DISABLED) return RT_ERR_INVALID_ARG;
    for (int i = 0; i < n_indices; i++)
        if (indices[i] < 0 || indices[i] >= n_verts) return RT_ERR_INVALID_ARG;
    auto t0 = std::chrono::steady_clock::now();
    if (threads <= 0) threads = default_threads();

    const int ntri = n_indices / 3;
    std::vector<BuildTri> tris(ntri);
    BuildTri* const TT = tris.data();
    parallel_chunks(ntri, threads, [=](int b0, int e0) { /* BVH:44-52 */
        for (int t = b0; t < e0; t++) {
            const float* a = verts + 3 * indices[3 * t + 0];
            const float* b = verts + 3 * indices[3 * t + 1];
            const float* c = verts + 3 * indices[3 * t + 2];
            BuildTri& bt = TT[t];
            for (int k = 0; k < 3; k++) {
                bt.c[k] = (a[k] + b[k] + c[k]) / 3;
                bt.mn[k] = min3(a[k], b[k], c[k]);
                bt.mx[k] = max3(a[k], b[k], c[k]);
            }
            bt.index = 3 * t;
        }
    });
    Box rootBox;
    rootBox.reset();
    for (int t = 0; t < ntri; t++) rootBox.grow(tris[t]); /* BVH:53-58, in triangle order */

    Stats st;
    RtBVHNode root;
    memcpy(root.boundsMin, rootBox.mn, 12);
    memcpy(root.boundsMax, rootBox.mx, 12);
    root.startIndex = -1; /* BVH:61 — an inner root keeps triangleCount == -1 */
    root.triangleCount = -1;

    std::vector<RtBVHNode> nodes;
    if (quality == RT_BVH_QUALITY_DISABLED) { /* BVH:62-66 */
        root.startIndex = 0;
        root.triangleCount = ntri;
        nodes.push_back(root);
    } else if (threads <= 1 || ntri < 8192) {
        nodes.reserve(2 * (size_t)(ntri > 0 ? ntri : 1));
        nodes.push_back(root);
        build_subtree(tris.data(), nodes, 0, ntri, 0, quality, st);
    } else {
        /* ---- 1+2. a shared work list of tree nodes: a thread takes a node; small ones (<= grain
         *           triangles) it builds to the leaves into a private node list, big ones it splits
         *           (the very biggest with the chunk-parallel sweep) and puts the children back */
        const int grain = 4096;
        const size_t maxTop = (size_t)ntri / 1024 * 4 + 64;
        std::vector<TopNode> top;
        top.reserve(maxTop + 2 * (size_t)threads); /* indices, never references, are held across the lock */
        std::vector<Task> tasks;
        tasks.reserve(maxTop);
        TopNode r;
        r.rec = root; r.start = 0; r.count = ntri; r.depth = 0;
        top.push_back(r);
        std::mutex mu;
        std::condition_variable cv;
        std::vector<int> queue(1, 0);
        int outstanding = 1;
        BuildTri* T = tris.data();
        auto worker = [&]() {
            for (;;) {
                int i;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !queue.empty() || outstanding == 0; });
                    if (queue.empty()) return;
                    i = queue.back();
                    queue.pop_back();
                }
                /* every access to `top` happens under the mutex (push_back may reallocate it) */
                TopNode tn;
                bool listFull;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    tn = top[i];
                    listFull = top.size() + 2 * (size_t)threads > maxTop; /* room for every worker's two children */
                }
                if (tn.count <= grain || listFull) {
                    /* build into thread-local containers (no false sharing between neighbouring tasks) */
                    std::vector<RtBVHNode> local;
                    std::vector<uint8_t> inner;
                    Stats lst;
                    local.reserve(2 * (size_t)tn.count + 1);
                    inner.reserve(2 * (size_t)tn.count + 1);
                    local.push_back(tn.rec);
                    build_subtree(T, local, tn.start, tn.count, tn.depth, quality, lst, &inner);
                    inner.resize(local.size(), 0);
                    std::lock_guard<std::mutex> lk(mu);
                    top[i].task = (int)tasks.size();
                    tasks.emplace_back();
                    tasks.back().top = i;
                    tasks.back().nodes.swap(local);
                    tasks.back().inner.swap(inner);
                    tasks.back().st = lst;
                    outstanding--;
                } else {
                    RtBVHNode cl, cr;
                    int numOnLeft = 0;
                    const int sweepThreads = tn.count >= ntri / 2 ? threads : 1; /* only the first levels: the list is still short */
                    const bool did = split_node(T, tn.rec, tn.start, tn.count, tn.depth, quality, sweepThreads, &cl, &cr, &numOnLeft);
                    std::lock_guard<std::mutex> lk(mu);
                    if (did) {
                        TopNode a, b;
                        a.rec = cl; a.start = tn.start; a.count = numOnLeft; a.depth = tn.depth + 1;
                        b.rec = cr; b.start = tn.start + numOnLeft; b.count = tn.count - numOnLeft; b.depth = tn.depth + 1;
                        top[i].child[0] = (int)top.size();
                        top.push_back(a);
                        top[i].child[1] = (int)top.size();
                        top.push_back(b);
                        queue.push_back(top[i].child[1]);
                        queue.push_back(top[i].child[0]);
                        outstanding += 1;
                    } else {
                        top[i].leaf = true;
                        outstanding--;
                    }
                }
                cv.notify_all();
            }
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(worker);
        for (auto& th : pool) th.join();
        if (getenv("RT_BVH_DEBUG")) fprintf(stderr, "[bvh] threads=%d top=%zu tasks=%zu t_built=%.1f ms\n", threads, top.size(), tasks.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        /* ---- 3. number the nodes in the reference's allocation order (BVH:161-171) */
        size_t total = 1;
        for (const TopNode& tn : top)
            if (tn.child[0] >= 0) total += 2;
        for (const Task& task : tasks) total += task.nodes.size() - 1;
        nodes.resize(total);
        int nextFree = 1;
        struct Emit { int top, gidx; };
        std::vector<Emit> stack;
        stack.push_back({0, 0});
        while (!stack.empty()) {
            Emit e = stack.back();
            stack.pop_back();
            const TopNode& tn = top[e.top];
            if (tn.task >= 0) {
                const Task& task = tasks[tn.task];
                const int base = nextFree; /* local k >= 1 -> base + k - 1; local 0 -> e.gidx */
                for (size_t k = 0; k < task.nodes.size(); k++) {
                    RtBVHNode n = task.nodes[k];
                    if (task.inner[k]) n.startIndex = base + n.startIndex - 1;
                    nodes[k == 0 ? (size_t)e.gidx : (size_t)(base + (int)k - 1)] = n;
                }
                nextFree += (int)task.nodes.size() - 1;
                st.merge(task.st);
            } else if (tn.leaf) {
                RtBVHNode n = tn.rec;
                n.startIndex = tn.start;
                n.triangleCount = tn.count;
                nodes[e.gidx] = n;
                st.leaf(tn.depth, tn.count);
            } else {
                RtBVHNode n = tn.rec;
                n.startIndex = nextFree;
                nodes[e.gidx] = n;
                st.inner();
                const int l = nextFree, rr = nextFree + 1;
                nextFree += 2;
                stack.push_back({tn.child[1], rr}); /* right after the whole left subtree */
                stack.push_back({tn.child[0], l});
            }
        }
    }

    parallel_chunks(ntri, threads, [=](int b0, int e0) { /* BVH:69-80: triangles in leaf order with vertex normals */
        for (int i = b0; i < e0; i++) {
            int base = TT[i].index;
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
    });
    /* A well-formed tree has at most 2*ntri-1 nodes.  The reference's (axis 0, pos 0) fallback split
     * (BVH:213-217) can peel off EMPTY children when every candidate cost is inf/NaN (coordinates
     * around 1e19 and beyond): such a tree has 0-triangle "leaves" the shader would read as inner
     * nodes (RC:246) and more nodes than the documented capacity of out_nodes — refuse it. */
    if (nodes.size() > 2 * (size_t)(ntri > 0 ? ntri : 1)) {
        *out_n_nodes = 0;
        return RT_ERR_SCENE;
    }
    if (ntri > 0 && st.leafMinTriCount == 0) { /* an empty leaf, even if the node count stayed in bounds */
        *out_n_nodes = 0;
        return RT_ERR_SCENE;
    }
    memcpy(out_nodes, nodes.data(), nodes.size() * sizeof(RtBVHNode));
    *out_n_nodes = (int)nodes.size();
    if (out_stats) {
        memset(out_stats, 0, sizeof(*out_stats));
        out_stats->triangleCount = st.triangleCount;
        out_stats->totalNodeCount = st.totalNodeCount;
        out_stats->leafNodeCount = st.leafNodeCount;
        out_stats->leafDepthMax = st.leafDepthMax;
        out_stats->leafDepthMin = st.leafDepthMin;
        out_stats->leafDepthSum = st.leafDepthSum;
        out_stats->leafMaxTriCount = st.leafMaxTriCount;
        out_stats->leafMinTriCount = st.leafMinTriCount;
        out_stats->quality = quality;
        out_stats->timeMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RT_OK;
}

} // namespace

extern "C" int rt_build_bvh(const float* verts, const float* normals, int n_verts, const int32_t* indices, int n_indices, int quality,
                            RtBVHNode* out_nodes, int* out_n_nodes, RtTriangle* out_tris, RtBvhStats* out_stats)
{
    return build(verts, normals, n_verts, indices, n_indices, quality, 0, out_nodes, out_n_nodes, out_tris, out_stats);
}

extern "C" int rt_build_bvh_mt(const float* verts, const float* normals, int n_verts, const int32_t* indices, int n_indices, int quality,
                               int n_threads, RtBVHNode* out_nodes, int* out_n_nodes, RtTriangle* out_tris, RtBvhStats* out_stats)
{
    return build(verts, normals, n_verts, indices, n_indices, quality, n_threads, out_nodes, out_n_nodes, out_tris, out_stats);
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
