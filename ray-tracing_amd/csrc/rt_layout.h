/*
 * rt_layout.h — WHERE the traversal's records lie in device memory (internal, host side; plain C++: rt_debug_layout hands the
 * result to tests/test_layout.py and to the line-touch model tools/layout_sim/ without a device).
 *
 * SceneBuilder::convert (rt_context.hip) validates the caller's BVHs and emits them in a CANONICAL form: node pairs in
 * post-order with pair INDICES in the inner codes and triangle INDICES in the leaf codes.  LayoutEngine::run() turns that into
 * what the kernels address (rt_device.h): every record is named by the 16-byte UNIT it starts at —
 *     inner code = unit of the DPair in the pair space, leaf code = first unit of the leaf's run of DTri records relative
 *     to the model's triBase (three units per triangle), normals = 12 bytes per unit of the triangle space —
 * so the order and the spacing of the records are the host's to choose and no layout can change a bit of the result:
 * RayTriangleBVH (RC:234-287) reads the same boxes, the same triangles, in the same order.
 *
 * RT_LAYOUT (comma separated; default = the layout measured best, profiles/r05_ab_layout.txt):
 *     dense     pairs in post-order, triangles in the caller's order, no padding (rounds 1-4)
 *     pre       pairs in pre-order (a pair is followed by its FIRST child's subtree)
 *     hot=K     the top K levels of every tree breadth-first in one block at the front of the tree's region
 *     align     a leaf's run never crosses a 128-byte line it need not cross (padding)
 *     arena     ONE space for pairs and triangles: a pair is followed by the runs of its leaf children
 *     palign    (arena) a pair never straddles a 128-byte line
 *     cache=N   (round 6) the TOP-OF-TREE CACHE: the N node pairs a ray is most likely to need — chosen greedily from the roots
 *               down by the world-space surface area of the node they belong to, summed over the models that share the tree —
 *               lie at units [0, 4N) of the pair space, in front of every instance; the BVH kernels' workgroups copy exactly that
 *               prefix into LDS (rt_kernels.h, traverse phase B).  cache (no number) / default = as many as the LDS of a
 *               workgroup has room for (rt_context.hip, plan_groups); cache=0 = none; no word at all = the default rule: as many as fit
 *               when they cover at least 1/16 of the scene's node pairs, else none (prepare_scene: where the cache was measured to pay)
 * Anything but `dense` needs a regular scene (no node pair shared between meshes, referenced triangles not much more
 * than the triangles there are); an irregular one silently gets `dense`.
 */
#ifndef RT_LAYOUT_H
#define RT_LAYOUT_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <math.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rt_abi.h"
#include "../../include/rt_math.h"
#include "rt_device.h"

/* Uninitialised storage for plain records that are about to be written in full (std::vector::resize would first zero
 * ~150 MB for a million triangles, on one thread: page faults, a third of rt_upload_scene's host time). */
template <typename T>
struct PodVec {
    T* p = nullptr;
    size_t n = 0;
    PodVec() = default;
    PodVec(const PodVec&) = delete;
    PodVec& operator=(const PodVec&) = delete;
    ~PodVec() { free(p); }
    bool resize_uninit(size_t k)
    {
        free(p);
        p = nullptr;
        const size_t bytes = k * sizeof(T), huge = (size_t)2 << 20;
        if (bytes >= 2 * huge) { /* fresh pages are the cost of a large scene's preparation: ask for 2 MB ones */
            void* q = nullptr;
            if (posix_memalign(&q, huge, (bytes + huge - 1) / huge * huge) == 0) {
                madvise(q, (bytes + huge - 1) / huge * huge, MADV_HUGEPAGE);
                p = static_cast<T*>(q);
            }
        } else if (k) {
            p = static_cast<T*>(malloc(bytes));
        }
        n = p ? k : 0;
        return k == 0 || p != nullptr;
    }
    void shrink(size_t k) { if (k < n) n = k; }
    void swap(PodVec& o) { std::swap(p, o.p); std::swap(n, o.n); }
    T* data() { return p; }
    const T* data() const { return p; }
    size_t size() const { return n; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

#define RT_UNIT_BYTES 16u
#define RT_PAIR_UNITS 4u /* sizeof(DPair) / 16 */
#define RT_TRI_UNITS 3u  /* sizeof(DTri) / 16 */
#define RT_LINE_UNITS 8u /* one 128-byte cache line */
#define RT_NORM_BYTES_PER_UNIT 12u

struct RtLayout {
    int hotLevels = 0;
    bool preorder = false;
    int triMode = 0; /* 0 dense, 1 align, 2 arena */
    bool pairAlign = false;
    int cacheRecords = -1; /* node pairs in the hot prefix of the pair space: -1 = default rule (what the workgroups' LDS holds IF that covers
                            * enough of the scene's pairs to pay, rt_context.hip prepare_scene), -2 = what the LDS holds, 0 = none */
    bool dense() const { return hotLevels == 0 && !preorder && triMode == 0; }
    std::string name() const
    {
        if (dense()) return "dense";
        std::string s;
        auto add = [&](const std::string& t) { s += (s.empty() ? "" : ",") + t; };
        if (preorder) add("pre");
        if (hotLevels) add("hot=" + std::to_string(hotLevels));
        if (triMode == 1) add("align");
        if (triMode == 2) add("arena");
        if (pairAlign) add("palign");
        if (cacheRecords > 0) add("cache=" + std::to_string(cacheRecords));
        return s;
    }
};

#ifndef RT_LAYOUT_DEFAULT
#define RT_LAYOUT_DEFAULT "pre,arena"
#endif

/* a decimal count in [0, max], nothing behind it (atoi read "hot=abc" as 0) */
static inline bool parse_count(const char* t, long max, int* out)
{
    if (!*t) return false;
    char* end = nullptr;
    const long v = strtol(t, &end, 10);
    if (*end || v < 0 || v > max) return false;
    *out = (int)v;
    return true;
}

/* unknown words are an error (returns false): a mistyped A/B run must not silently measure the default */
static inline bool parse_layout(const char* s, RtLayout* out)
{
    RtLayout L;
    std::string str(s ? s : "");
    size_t i = 0;
    while (i <= str.size()) {
        size_t j = str.find(',', i);
        if (j == std::string::npos) j = str.size();
        std::string w = str.substr(i, j - i);
        if (w == "" || w == "dense" || w == "post") { }
        else if (w == "pre") L.preorder = true;
        else if (w.compare(0, 4, "hot=") == 0) { if (!parse_count(w.c_str() + 4, 24, &L.hotLevels)) return false; }
        else if (w == "cache") L.cacheRecords = -2; /* as many as fit, whatever the scene's size (-1, the default, applies the coverage rule) */
        else if (w.compare(0, 6, "cache=") == 0) { if (!parse_count(w.c_str() + 6, 1 << 16, &L.cacheRecords)) return false; }
        else if (w == "align") L.triMode = 1;
        else if (w == "arena") L.triMode = 2;
        else if (w == "palign") L.pairAlign = true;
        else return false;
        i = j + 1;
    }
    if (L.pairAlign && L.triMode != 2) return false;
    *out = L;
    return true;
}

/* RC:190-192 are ray independent: the triangle pre-differenced with the same fp32 operations, and its vertex normals */
static inline void make_dtri(const RtTriangle& t, DTri& d, float* n9)
{
    rt_f3 A = rt_v3(t.posA[0], t.posA[1], t.posA[2]);
    rt_f3 B = rt_v3(t.posB[0], t.posB[1], t.posB[2]);
    rt_f3 Cc = rt_v3(t.posC[0], t.posC[1], t.posC[2]);
    rt_f3 ab = B - A, ac = Cc - A;
    rt_f3 f = rt_cross(ab, ac);
    d.ax = A.x; d.ay = A.y; d.az = A.z;
    d.abx = ab.x; d.aby = ab.y; d.abz = ab.z;
    d.acx = ac.x; d.acy = ac.y; d.acz = ac.z;
    d.fx = f.x; d.fy = f.y; d.fz = f.z;
    memcpy(n9 + 0, t.normA, 12);
    memcpy(n9 + 3, t.normB, 12);
    memcpy(n9 + 6, t.normC, 12);
}

struct LaidOutScene {
    /* pair space (arena: pairs AND triangles), triangle space (arena: empty — the kernels get the pair space twice), normals */
    PodVec<unsigned char> pairBuf, triBuf, normBuf;
    bool arena = false;
    std::vector<uint32_t> bigLeaves;
    std::vector<uint32_t> rootCodes; /* per model, final */
    std::vector<int32_t> triBase;    /* per model, units */
    RtLayout used;
    uint32_t hotUnits = 0; /* units [0, hotUnits) of the pair space = the top-of-tree cache's records (4 units each) */
    std::string error;
};

struct LayoutEngine {
    /* inputs */
    DPair* canon = nullptr; /* canonical pairs (rewritten in place by the dense layout) */
    size_t nCanon = 0;
    const std::vector<uint32_t>* canonBig = nullptr;
    const RtModel* models = nullptr;
    int nModels = 0;
    const uint32_t* rootCodes = nullptr; /* canonical */
    const RtTriangle* tris = nullptr;
    int nTris = 0;
    /* run f(k) for k in [0, n) on the caller's worker threads */
    void (*parallel)(int n, void* ctx, void (*f)(void*, int)) = nullptr;

    template <typename F>
    void par(int n, F f)
    {
        if (!parallel) { for (int k = 0; k < n; k++) f(k); return; }
        parallel(n, &f, [](void* c, int k) { (*static_cast<F*>(c))(k); });
    }

    void decode_leaf(uint32_t code, uint32_t* start, uint32_t* count) const
    {
        *count = (code >> 24) & 0x7fu;
        *start = code & RT_CODE_MAX_INLINE_START;
        if (*count == 0) {
            *count = (*canonBig)[2 * (size_t)*start + 1];
            *start = (*canonBig)[2 * (size_t)*start];
        }
    }
    static bool encode_leaf(uint32_t relUnit, uint32_t count, std::vector<uint32_t>& big, uint32_t* code)
    {
        if (count <= RT_CODE_MAX_INLINE_COUNT && relUnit <= RT_CODE_MAX_INLINE_START) {
            *code = RT_CODE_LEAF | (count << 24) | relUnit;
            return true;
        }
        const size_t idx = big.size() / 2;
        if (idx > RT_CODE_MAX_INLINE_START) return false;
        big.push_back(relUnit);
        big.push_back(count);
        *code = RT_CODE_LEAF | (uint32_t)idx;
        return true;
    }

    /* ---- dense: nothing moves; the codes go from indices to units */
    bool run_dense(PodVec<DPair>& canonStore, LaidOutScene& out)
    {
        out.arena = false;
        out.used = RtLayout();
        out.rootCodes.assign(nModels, 0u);
        out.triBase.assign(nModels, 0);
        /* roots first, in model order (deterministic table of oversized leaves) */
        for (int m = 0; m < nModels; m++) {
            out.triBase[m] = (int32_t)((uint32_t)models[m].triOffset * RT_TRI_UNITS);
            const uint32_t c = rootCodes[m];
            if (c & RT_CODE_LEAF) {
                uint32_t start, count;
                decode_leaf(c, &start, &count);
                if (!encode_leaf(start * RT_TRI_UNITS, count, out.bigLeaves, &out.rootCodes[m])) { out.error = "too many oversized leaves"; return false; }
            } else {
                out.rootCodes[m] = c * RT_PAIR_UNITS;
            }
        }
        const int block = 1 << 14;
        const int nBlocks = (int)((nCanon + block - 1) / block);
        std::vector<std::vector<size_t>> later(nBlocks);
        par(nBlocks, [&](int b) {
            const size_t i1 = (size_t)(b + 1) * block < nCanon ? (size_t)(b + 1) * block : nCanon;
            for (size_t i = (size_t)b * block; i < i1; i++) {
                uint32_t* codes[2] = {&canon[i].codeA, &canon[i].codeB};
                for (int s = 0; s < 2; s++) {
                    const uint32_t c = *codes[s];
                    if (!(c & RT_CODE_LEAF)) { *codes[s] = c * RT_PAIR_UNITS; continue; }
                    const uint32_t count = (c >> 24) & 0x7fu, start = c & RT_CODE_MAX_INLINE_START;
                    if (count && start * RT_TRI_UNITS <= RT_CODE_MAX_INLINE_START) *codes[s] = RT_CODE_LEAF | (count << 24) | (start * RT_TRI_UNITS);
                    else later[b].push_back(2 * i + s);
                }
            }
        });
        for (int b = 0; b < nBlocks; b++)
            for (size_t k : later[b]) {
                uint32_t* code = (k & 1) ? &canon[k >> 1].codeB : &canon[k >> 1].codeA;
                uint32_t start, count;
                decode_leaf(*code, &start, &count);
                if (!encode_leaf(start * RT_TRI_UNITS, count, out.bigLeaves, code)) { out.error = "too many oversized leaves"; return false; }
            }
        /* the canonical array IS the pair space */
        {
            PodVec<unsigned char> tmp;
            tmp.p = reinterpret_cast<unsigned char*>(canonStore.p);
            tmp.n = canonStore.n * sizeof(DPair);
            canonStore.p = nullptr;
            canonStore.n = 0;
            out.pairBuf.swap(tmp);
        }
        if (!out.triBuf.resize_uninit((size_t)nTris * sizeof(DTri)) || !out.normBuf.resize_uninit((size_t)nTris * sizeof(DTriN))) { out.error = "out of host memory"; return false; }
        const int triBlock = 1 << 15;
        DTri* dt = reinterpret_cast<DTri*>(out.triBuf.data());
        DTriN* dn = reinterpret_cast<DTriN*>(out.normBuf.data());
        par((nTris + triBlock - 1) / triBlock, [&](int blk) {
            const int i1 = (blk + 1) * triBlock < nTris ? (blk + 1) * triBlock : nTris;
            for (int i = blk * triBlock; i < i1; i++) make_dtri(tris[i], dt[i], dn[i].n);
        });
        return true;
    }

    /* ---- everything else: a walk per mesh instance decides the places.  Every instance is laid out on its own (relative units, in
     * parallel), a prefix sum over the instances' sizes gives the bases: [instance 0: hot block | the rest][instance 1: ...] ... */
    struct Placed { uint32_t pair; int side; uint32_t unit; uint32_t start, count; }; /* pair = UINT32_MAX: the instance's leaf root; unit relative to the instance's triangle region */
    struct Inst {
        uint32_t root;
        int triOffset;
        std::vector<int> modelsOf;
        std::vector<Placed> leaves;
        std::vector<std::pair<uint32_t, uint32_t>> pairs; /* (canonical pair, unit relative to the instance's pair region) */
        uint64_t pairUnits = 0, triUnits = 0;             /* region sizes (arena: pairUnits only) */
        uint64_t pairBase = 0, triBase = 0;               /* absolute first unit of the regions */
        uint64_t placedTris = 0;
        uint32_t triBaseCode = 0;                         /* what the leaf codes are relative to */
        bool irregular = false;
    };

    /* one instance, relative placement; owner[] = the instance that claimed a pair (a second claim = not a forest of meshes) */
    void lay_instance(const RtLayout& L, Inst& I, int k, std::atomic<int32_t>* owner)
    {
        const bool arena = L.triMode == 2;
        uint64_t pairCur = 0, triCur = 0;
        auto claim = [&](uint32_t p) {
            int32_t expect = -1;
            if (!owner[p].compare_exchange_strong(expect, (int32_t)k)) I.irregular = true;
        };
        auto place_pair = [&](uint32_t p) {
            if (L.pairAlign && (pairCur % RT_LINE_UNITS) > RT_LINE_UNITS - RT_PAIR_UNITS) pairCur = (pairCur / RT_LINE_UNITS + 1) * RT_LINE_UNITS;
            I.pairs.push_back({p, (uint32_t)pairCur});
            pairCur += RT_PAIR_UNITS;
        };
        auto place_run = [&](uint32_t pair, int side, uint32_t start, uint32_t count) {
            uint64_t& cur = arena ? pairCur : triCur;
            const uint64_t n = (uint64_t)count * RT_TRI_UNITS;
            if (L.triMode == 1) { /* a line crossing that padding can remove is removed */
                const uint64_t lines = (cur % RT_LINE_UNITS + n + RT_LINE_UNITS - 1) / RT_LINE_UNITS, least = (n + RT_LINE_UNITS - 1) / RT_LINE_UNITS;
                if (lines > least) cur = (cur / RT_LINE_UNITS + 1) * RT_LINE_UNITS;
            }
            I.leaves.push_back({pair, side, (uint32_t)cur, start, count});
            cur += n;
            I.placedTris += count;
        };
        auto leaf_children = [&](uint32_t p) {
            const uint32_t codes[2] = {canon[p].codeA, canon[p].codeB};
            for (int s = 0; s < 2; s++)
                if (codes[s] & RT_CODE_LEAF) {
                    uint32_t start, count;
                    decode_leaf(codes[s], &start, &count);
                    place_run(p, s, start, count);
                }
        };
        if (I.root & RT_CODE_LEAF) {
            uint32_t start, count;
            decode_leaf(I.root, &start, &count);
            place_run(UINT32_MAX, 0, start, count);
        } else {
            claim(I.root);
            /* 1. the hot block: the top levels breadth-first (hot pairs are marked by unit != UINT32_MAX in `hotUnit`) */
            std::vector<uint32_t> hotList;
            if (L.hotLevels) {
                std::vector<uint32_t> level(1, I.root), next;
                for (int d = 0; d < L.hotLevels && !level.empty() && !I.irregular; d++) {
                    next.clear();
                    for (uint32_t p : level) {
                        hotList.push_back(p);
                        place_pair(p);
                        const uint32_t codes[2] = {canon[p].codeA, canon[p].codeB};
                        for (int s = 0; s < 2; s++)
                            if (!(codes[s] & RT_CODE_LEAF)) { claim(codes[s]); next.push_back(codes[s]); }
                    }
                    level.swap(next);
                }
                if (arena) /* leaves hanging off the hot pairs: behind the block */
                    for (uint32_t p : hotList) leaf_children(p);
            }
            std::sort(hotList.begin(), hotList.end());
            auto is_hot = [&](uint32_t p) { return !hotList.empty() && std::binary_search(hotList.begin(), hotList.end(), p); };
            /* 2. the rest, depth-first; a child the hot pass already claimed (hot itself, or the frontier below it) is ours */
            const size_t nHot = hotList.size();
            std::vector<uint32_t> frontier; /* claimed by the hot pass, not hot: the level below the block */
            if (nHot) {
                for (uint32_t p : hotList) {
                    const uint32_t codes[2] = {canon[p].codeA, canon[p].codeB};
                    for (int s = 0; s < 2; s++)
                        if (!(codes[s] & RT_CODE_LEAF) && !is_hot(codes[s])) frontier.push_back(codes[s]);
                }
                std::sort(frontier.begin(), frontier.end());
            }
            auto preclaimed = [&](uint32_t p) { return nHot && (is_hot(p) || std::binary_search(frontier.begin(), frontier.end(), p)); };
            struct Frame { uint32_t p; int stage; bool hot; };
            std::vector<Frame> stack;
            stack.push_back({I.root, 0, is_hot(I.root)});
            while (!stack.empty() && !I.irregular) {
                Frame& f = stack.back();
                const uint32_t p = f.p;
                if (f.stage == 0) {
                    if (L.preorder) {
                        if (!f.hot) {
                            place_pair(p);
                            if (arena) leaf_children(p);
                        }
                        if (!arena) leaf_children(p);
                    }
                    f.stage = 1;
                    const uint32_t c = canon[p].codeA;
                    if (!(c & RT_CODE_LEAF)) {
                        if (!preclaimed(c)) claim(c);
                        const bool h = is_hot(c);
                        if ((int)stack.size() > RT_MAX_BVH_DEPTH + 2) { I.irregular = true; break; } /* (convert() bounds the height; a shared pair below the hot block could loop) */
                        stack.push_back({c, 0, h});
                    }
                    continue;
                }
                if (f.stage == 1) {
                    f.stage = 2;
                    const uint32_t c = canon[p].codeB;
                    if (!(c & RT_CODE_LEAF)) {
                        if (!preclaimed(c)) claim(c);
                        const bool h = is_hot(c);
                        if ((int)stack.size() > RT_MAX_BVH_DEPTH + 2) { I.irregular = true; break; }
                        stack.push_back({c, 0, h});
                    }
                    continue;
                }
                if (!L.preorder) {
                    if (!f.hot) {
                        place_pair(p);
                        if (arena) leaf_children(p);
                    }
                    if (!arena) leaf_children(p);
                }
                stack.pop_back();
            }
        }
        I.pairUnits = pairCur;
        I.triUnits = triCur;
    }

    bool run(const RtLayout& L, PodVec<DPair>& canonStore, LaidOutScene& out)
    {
        if (L.dense()) return run_dense(canonStore, out);
        /* RT_DEBUG_UPLOAD: where the time of a layout goes */
        const bool timing = getenv("RT_DEBUG_UPLOAD") != nullptr;
        auto tPrev = std::chrono::steady_clock::now();
#define RT_LAYOUT_T(what)                                                                                                      \
        do {                                                                                                                   \
            if (timing) {                                                                                                      \
                const auto now = std::chrono::steady_clock::now();                                                             \
                fprintf(stderr, "[rt] layout: %-16s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tPrev).count()); \
                tPrev = now;                                                                                                   \
            }                                                                                                                  \
        } while (0)
        /* instances = distinct (root, triOffset), in the order the models name them */
        std::vector<Inst> insts;
        {
            std::map<std::pair<uint32_t, int>, int> seen;
            for (int m = 0; m < nModels; m++) {
                auto key = std::make_pair(rootCodes[m], (int)models[m].triOffset);
                auto it = seen.find(key);
                if (it == seen.end()) {
                    it = seen.emplace(key, (int)insts.size()).first;
                    insts.emplace_back();
                    insts.back().root = rootCodes[m];
                    insts.back().triOffset = models[m].triOffset;
                }
                insts[it->second].modelsOf.push_back(m);
            }
        }
        const bool arena = L.triMode == 2;
        std::unique_ptr<std::atomic<int32_t>[]> owner(new std::atomic<int32_t>[nCanon ? nCanon : 1]);
        par((int)((nCanon + (1 << 16) - 1) >> 16), [&](int b) {
            const size_t i1 = ((size_t)b + 1) << 16 < nCanon ? ((size_t)b + 1) << 16 : nCanon;
            for (size_t i = (size_t)b << 16; i < i1; i++) owner[i].store(-1, std::memory_order_relaxed);
        });
        RT_LAYOUT_T("owner init");
        par((int)insts.size(), [&](int k) { lay_instance(L, insts[k], k, owner.get()); });
        RT_LAYOUT_T("walks");
        bool irregular = false;
        for (Inst& I : insts) irregular = irregular || I.irregular;
        /* ---- the top-of-tree cache's records: greedy from the roots down by expected visits.  A node is entered by the rays that hit its box:
         * under the usual surface-area argument that is proportional to the box's area in WORLD space — 2 (dy dz |c2 x c3| + dx dz |c1 x c3| +
         * dx dy |c1 x c2|) for a box of extent d under a linear map with columns c — summed over the models that share the tree.  The node's
         * box is the union of the two child boxes its pair record holds.  A child's area never exceeds its parent's, so the set stays the
         * connected top of every tree. */
        std::vector<uint32_t> hot;
        if (L.cacheRecords > 0 && !irregular) {
            struct Cand { double w; uint32_t pair; int inst; };
            auto worse = [](const Cand& a, const Cand& b) { return a.w < b.w || (a.w == b.w && a.pair > b.pair); };
            std::vector<Cand> heap;
            std::vector<double> coef(3 * insts.size(), 0.0);
            for (size_t k = 0; k < insts.size(); k++)
                for (int m : insts[k].modelsOf) {
                    const float* M = models[m].localToWorld; /* column-major: columns at 0, 4, 8 */
                    const double c1[3] = {M[0], M[1], M[2]}, c2[3] = {M[4], M[5], M[6]}, c3[3] = {M[8], M[9], M[10]};
                    auto crossLen = [](const double* a, const double* b) {
                        const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
                        return sqrt(x * x + y * y + z * z);
                    };
                    const double kx = crossLen(c2, c3), ky = crossLen(c1, c3), kz = crossLen(c1, c2);
                    coef[3 * k + 0] += std::isfinite(kx) ? kx : 0.0;
                    coef[3 * k + 1] += std::isfinite(ky) ? ky : 0.0;
                    coef[3 * k + 2] += std::isfinite(kz) ? kz : 0.0;
                }
            auto weight = [&](uint32_t p, int k) {
                const DPair& d = canon[p];
                double e[3];
                for (int a = 0; a < 3; a++) {
                    const double lo = d.aMin[a] < d.bMin[a] ? d.aMin[a] : d.bMin[a], hi = d.aMax[a] > d.bMax[a] ? d.aMax[a] : d.bMax[a];
                    e[a] = hi - lo;
                    if (!(e[a] >= 0.0) || !std::isfinite(e[a])) e[a] = 0.0;
                }
                const double w = e[1] * e[2] * coef[3 * k] + e[0] * e[2] * coef[3 * k + 1] + e[0] * e[1] * coef[3 * k + 2];
                return std::isfinite(w) ? w : 0.0;
            };
            for (size_t k = 0; k < insts.size(); k++)
                if (!(insts[k].root & RT_CODE_LEAF)) heap.push_back({weight(insts[k].root, (int)k), insts[k].root, (int)k});
            std::make_heap(heap.begin(), heap.end(), worse);
            while (!heap.empty() && (int)hot.size() < L.cacheRecords) {
                std::pop_heap(heap.begin(), heap.end(), worse);
                const Cand c = heap.back();
                heap.pop_back();
                hot.push_back(c.pair);
                const uint32_t codes[2] = {canon[c.pair].codeA, canon[c.pair].codeB};
                for (int s2 = 0; s2 < 2; s2++)
                    if (!(codes[s2] & RT_CODE_LEAF)) {
                        heap.push_back({weight(codes[s2], c.inst), codes[s2], c.inst});
                        std::push_heap(heap.begin(), heap.end(), worse);
                    }
            }
        }
        const uint64_t hotUnits = (uint64_t)hot.size() * RT_PAIR_UNITS;
        uint64_t pairCur = hotUnits, triCur = 0, placedTris = 0;
        for (Inst& I : insts) {
            /* an instance starts on a line (its own padding rules are relative to its start) */
            pairCur = (pairCur + RT_LINE_UNITS - 1) / RT_LINE_UNITS * RT_LINE_UNITS;
            triCur = (triCur + RT_LINE_UNITS - 1) / RT_LINE_UNITS * RT_LINE_UNITS;
            I.pairBase = pairCur;
            I.triBase = arena ? pairCur : triCur;
            pairCur += I.pairUnits;
            triCur += I.triUnits;
            placedTris += I.placedTris;
        }
        if (placedTris > 2ull * (uint64_t)nTris + 1024) irregular = true; /* leaves that overlap each other en masse */
        if (irregular || pairCur * RT_UNIT_BYTES >= ((uint64_t)1 << 32) || triCur * RT_UNIT_BYTES >= ((uint64_t)1 << 32)) {
            const bool ok = run_dense(canonStore, out);
            out.used = RtLayout();
            return ok;
        }

        /* final codes: the inline ones in parallel, the oversized leaves afterwards in instance order (a deterministic table) */
        out.arena = arena;
        out.used = L;
        out.used.cacheRecords = (int)hot.size();
        out.hotUnits = (uint32_t)hotUnits;
        out.rootCodes.assign(nModels, 0u);
        out.triBase.assign(nModels, 0);
        std::vector<uint32_t> unitOf(nCanon, UINT32_MAX), leafCode(2 * nCanon, 0u);
        std::vector<uint32_t> rootCode(insts.size(), 0u);
        std::vector<std::vector<size_t>> later(insts.size());
        par((int)insts.size(), [&](int k) {
            Inst& I = insts[k];
            for (const auto& pu : I.pairs) unitOf[pu.first] = (uint32_t)(I.pairBase + pu.second);
            uint32_t lo = UINT32_MAX;
            for (const Placed& q : I.leaves) lo = q.unit < lo ? q.unit : lo;
            I.triBaseCode = I.leaves.empty() ? 0u : (uint32_t)(I.triBase + lo);
            for (size_t n = 0; n < I.leaves.size(); n++) {
                const Placed& q = I.leaves[n];
                const uint32_t rel = (uint32_t)(I.triBase + q.unit) - I.triBaseCode;
                if (q.count <= RT_CODE_MAX_INLINE_COUNT && rel <= RT_CODE_MAX_INLINE_START) {
                    const uint32_t code = RT_CODE_LEAF | (q.count << 24) | rel;
                    if (q.pair == UINT32_MAX) rootCode[k] = code;
                    else leafCode[2 * (size_t)q.pair + q.side] = code;
                } else {
                    later[k].push_back(n);
                }
            }
        });
        /* the cache's records live in the prefix; the place the instance walk gave them stays an (unread, zeroed) hole */
        for (size_t i = 0; i < hot.size(); i++) unitOf[hot[i]] = (uint32_t)(i * RT_PAIR_UNITS);
        for (size_t k = 0; k < insts.size(); k++) {
            Inst& I = insts[k];
            for (size_t n : later[k]) {
                const Placed& q = I.leaves[n];
                uint32_t code;
                if (!encode_leaf((uint32_t)(I.triBase + q.unit) - I.triBaseCode, q.count, out.bigLeaves, &code)) { out.error = "too many oversized leaves"; return false; }
                if (q.pair == UINT32_MAX) rootCode[k] = code;
                else leafCode[2 * (size_t)q.pair + q.side] = code;
            }
            if (!(I.root & RT_CODE_LEAF)) rootCode[k] = unitOf[I.root];
            for (int m : I.modelsOf) {
                out.rootCodes[m] = rootCode[k];
                out.triBase[m] = (int32_t)I.triBaseCode;
            }
        }

        RT_LAYOUT_T("codes");
        /* the buffers: zeroed (padding, the normals' holes under pair records) and filled in parallel */
        const size_t pairBytes = (size_t)pairCur * RT_UNIT_BYTES, triBytes = (size_t)triCur * RT_UNIT_BYTES;
        const size_t normBytes = (size_t)(arena ? pairCur : triCur) * RT_NORM_BYTES_PER_UNIT;
        if (!out.pairBuf.resize_uninit(pairBytes ? pairBytes : 64) || !out.triBuf.resize_uninit(arena ? 0 : (triBytes ? triBytes : 48)) ||
            !out.normBuf.resize_uninit(normBytes ? normBytes : 36)) { out.error = "out of host memory"; return false; }
        auto par_zero = [&](unsigned char* p, size_t n) {
            const size_t blk = (size_t)4 << 20;
            par((int)((n + blk - 1) / blk), [&](int b) { memset(p + (size_t)b * blk, 0, (size_t)(b + 1) * blk < n ? blk : n - (size_t)b * blk); });
        };
        /* zero first, in parallel 4-MB blocks: padding, the normals' holes under pair records — and the first touch of the fresh pages by ONE
         * thread per block (filling without it, sixteen threads faulting the same 2-MB pages, measured 20 % slower on a million triangles) */
        par_zero(out.pairBuf.data(), out.pairBuf.size());
        if (!arena) par_zero(out.triBuf.data(), out.triBuf.size());
        par_zero(out.normBuf.data(), out.normBuf.size());
        RT_LAYOUT_T("alloc + zero");
        unsigned char* const triSpace = arena ? out.pairBuf.data() : out.triBuf.data();
        /* work items: (instance, block of its pairs) and (instance, block of its leaves) */
        struct Item { int k; bool leaves; size_t a, b; };
        std::vector<Item> items;
        const size_t blk = 1 << 13;
        for (size_t k = 0; k < insts.size(); k++) {
            for (size_t a = 0; a < insts[k].pairs.size(); a += blk) items.push_back({(int)k, false, a, std::min(a + blk, insts[k].pairs.size())});
            for (size_t a = 0; a < insts[k].leaves.size(); a += blk) items.push_back({(int)k, true, a, std::min(a + blk, insts[k].leaves.size())});
        }
        par((int)items.size(), [&](int n) {
            const Item& it = items[n];
            const Inst& I = insts[it.k];
            if (!it.leaves) {
                for (size_t a = it.a; a < it.b; a++) {
                    const uint32_t i = I.pairs[a].first;
                    DPair d = canon[i];
                    d.codeA = (d.codeA & RT_CODE_LEAF) ? leafCode[2 * (size_t)i] : unitOf[d.codeA];
                    d.codeB = (d.codeB & RT_CODE_LEAF) ? leafCode[2 * (size_t)i + 1] : unitOf[d.codeB];
                    memcpy(out.pairBuf.data() + (size_t)unitOf[i] * RT_UNIT_BYTES, &d, sizeof(d));
                }
            } else {
                for (size_t a = it.a; a < it.b; a++) {
                    const Placed& q = I.leaves[a];
                    for (uint32_t t = 0; t < q.count; t++) {
                        const size_t unit = (size_t)(I.triBase + q.unit) + (size_t)t * RT_TRI_UNITS;
                        DTri d;
                        float n9[9];
                        make_dtri(tris[(size_t)I.triOffset + q.start + t], d, n9);
                        memcpy(triSpace + unit * RT_UNIT_BYTES, &d, sizeof(d));
                        memcpy(out.normBuf.data() + unit * RT_NORM_BYTES_PER_UNIT, n9, sizeof(n9));
                    }
                }
            }
        });
        RT_LAYOUT_T("fill");
        canonStore.resize_uninit(0);
        RT_LAYOUT_T("free canonical");
        return true;
    }
};

#endif
