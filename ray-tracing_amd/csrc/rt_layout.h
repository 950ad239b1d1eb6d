/*
 * rt_layout.h — WHERE the traversal's records lie in device memory (internal, host side; plain C++: tools/layout_sim.cpp
 * replays it on the CPU).
 *
 * SceneBuilder::convert (rt_context.hip) validates the caller's BVHs and emits them in a CANONICAL form: node pairs in
 * post-order with pair INDICES in the inner codes and triangle INDICES in the leaf codes.  apply_layout() turns that into
 * what the kernels address (rt_device.h): every record is named by the 16-byte UNIT it starts at —
 *     inner code = unit of the DPair in the pair space, leaf code = first unit of the leaf's run of DTri records relative
 *     to the model's triBase (three units per triangle), normals = 12 bytes per unit of the triangle space —
 * so the order and the spacing of the records are the host's to choose and no layout can change a bit of the result:
 * RayTriangleBVH (RC:234-287) reads the same boxes, the same triangles, in the same order.
 *
 * RT_LAYOUT (comma separated; default = the layout measured best, profiles/r05_ab_layout.txt):
 *     dense     pairs in post-order, triangles in the caller's order, no padding (rounds 1-4)
 *     pre       pairs in pre-order (a pair is followed by its FIRST child's subtree)
 *     hot=K     the top K levels of every tree breadth-first in one block ahead of everything else
 *     align     a leaf's run never crosses a 128-byte line it need not cross (padding)
 *     arena     ONE space for pairs and triangles: a pair is followed by the runs of its leaf children
 *     palign    (arena) a pair never straddles a 128-byte line
 * Anything but `dense` needs a regular scene (no node pair shared between meshes, referenced triangles not much more
 * than the triangles there are); an irregular one silently gets `dense`.
 */
#ifndef RT_LAYOUT_H
#define RT_LAYOUT_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <map>
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
        return s;
    }
};

#ifndef RT_LAYOUT_DEFAULT
#define RT_LAYOUT_DEFAULT "dense"
#endif

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
        else if (w.compare(0, 4, "hot=") == 0) { L.hotLevels = atoi(w.c_str() + 4); if (L.hotLevels < 0 || L.hotLevels > 24) return false; }
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

    /* ---- everything else: a walk per mesh instance decides the places */
    struct Placed { uint32_t pair; int side; uint32_t unit; uint32_t start, count; }; /* pair = UINT32_MAX: the instance's leaf root */
    struct Inst {
        uint32_t root;
        int triOffset;
        std::vector<int> modelsOf;
        std::vector<Placed> leaves;
        uint32_t triBase = 0;
    };

    bool run(const RtLayout& L, PodVec<DPair>& canonStore, LaidOutScene& out)
    {
        if (L.dense()) return run_dense(canonStore, out);
        /* instances = distinct (root, triOffset), in the order the models name them */
        std::vector<Inst> insts;
        std::vector<int> instOfModel(nModels, 0);
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
                instOfModel[m] = it->second;
            }
        }
        const bool arena = L.triMode == 2;
        std::vector<int32_t> owner(nCanon, -1);   /* instance that placed / will place the pair */
        std::vector<uint32_t> unitOf(nCanon, UINT32_MAX);
        uint64_t pairCur = 0, triCur = 0;         /* next free unit of the two spaces (arena: pairCur only) */
        uint64_t placedTris = 0;
        bool irregular = false;

        auto place_pair = [&](uint32_t p) {
            if (L.pairAlign && (pairCur % RT_LINE_UNITS) > RT_LINE_UNITS - RT_PAIR_UNITS) pairCur = (pairCur / RT_LINE_UNITS + 1) * RT_LINE_UNITS;
            unitOf[p] = (uint32_t)pairCur;
            pairCur += RT_PAIR_UNITS;
        };
        auto place_run = [&](Inst& I, uint32_t pair, int side, uint32_t start, uint32_t count) {
            uint64_t& cur = arena ? pairCur : triCur;
            const uint64_t n = (uint64_t)count * RT_TRI_UNITS;
            if (L.triMode == 1) { /* a line crossing that padding can remove is removed */
                const uint64_t lines = (cur % RT_LINE_UNITS + n + RT_LINE_UNITS - 1) / RT_LINE_UNITS, least = (n + RT_LINE_UNITS - 1) / RT_LINE_UNITS;
                if (lines > least) cur = (cur / RT_LINE_UNITS + 1) * RT_LINE_UNITS;
            }
            I.leaves.push_back({pair, side, (uint32_t)cur, start, count});
            cur += n;
            placedTris += count;
        };
        auto leaf_children = [&](Inst& I, uint32_t p) {
            const uint32_t codes[2] = {canon[p].codeA, canon[p].codeB};
            for (int s = 0; s < 2; s++)
                if (codes[s] & RT_CODE_LEAF) {
                    uint32_t start, count;
                    decode_leaf(codes[s], &start, &count);
                    place_run(I, p, s, start, count);
                }
        };

        /* 1. the hot block: the top levels of every tree, breadth-first, tree after tree */
        std::vector<char> hot(nCanon, 0);
        for (size_t k = 0; k < insts.size() && !irregular; k++) {
            Inst& I = insts[k];
            if (I.root & RT_CODE_LEAF) continue;
            if (owner[I.root] >= 0) { irregular = true; break; } /* two instances over the same pairs (same nodes, other triangles) */
            owner[I.root] = (int32_t)k;
            if (!L.hotLevels) continue;
            std::vector<uint32_t> level(1, I.root), next;
            for (int d = 0; d < L.hotLevels && !level.empty(); d++) {
                next.clear();
                for (uint32_t p : level) {
                    hot[p] = 1;
                    place_pair(p);
                    const uint32_t codes[2] = {canon[p].codeA, canon[p].codeB};
                    for (int s = 0; s < 2; s++)
                        if (!(codes[s] & RT_CODE_LEAF)) {
                            if (owner[codes[s]] >= 0) { irregular = true; break; }
                            owner[codes[s]] = (int32_t)k;
                            next.push_back(codes[s]);
                        }
                    if (irregular) break;
                }
                level.swap(next);
                if (irregular) break;
            }
        }
        if (L.hotLevels && arena && !irregular) /* leaves hanging off the hot pairs: behind the block */
            for (size_t k = 0; k < insts.size(); k++) {
                Inst& I = insts[k];
                if (I.root & RT_CODE_LEAF) continue;
                std::vector<uint32_t> level(1, I.root), next;
                while (!level.empty()) {
                    next.clear();
                    for (uint32_t p : level) {
                        if (!hot[p]) continue;
                        leaf_children(I, p);
                        if (!(canon[p].codeA & RT_CODE_LEAF)) next.push_back(canon[p].codeA);
                        if (!(canon[p].codeB & RT_CODE_LEAF)) next.push_back(canon[p].codeB);
                    }
                    level.swap(next);
                }
            }

        /* 2. the rest, depth-first, instance after instance */
        struct Frame { uint32_t p; int stage; };
        std::vector<Frame> stack;
        std::vector<char> walked(nCanon, 0); /* a pair reached twice (a node graph that is not a forest) is irregular */
        for (size_t k = 0; k < insts.size() && !irregular; k++) {
            Inst& I = insts[k];
            if (I.root & RT_CODE_LEAF) {
                uint32_t start, count;
                decode_leaf(I.root, &start, &count);
                place_run(I, UINT32_MAX, 0, start, count);
                continue;
            }
            stack.clear();
            stack.push_back({I.root, 0});
            walked[I.root] = 1;
            while (!stack.empty() && !irregular) {
                Frame& f = stack.back();
                const uint32_t p = f.p;
                if (f.stage == 0) {
                    if (!hot[p] && L.preorder) {
                        place_pair(p);
                        if (arena) leaf_children(I, p);
                    }
                    if (!arena && L.preorder) leaf_children(I, p);
                    f.stage = 1;
                    const uint32_t c = canon[p].codeA;
                    if (!(c & RT_CODE_LEAF)) {
                        if (walked[c] || (owner[c] >= 0 && owner[c] != (int32_t)k)) { irregular = true; break; }
                        owner[c] = (int32_t)k;
                        walked[c] = 1;
                        stack.push_back({c, 0});
                    }
                    continue;
                }
                if (f.stage == 1) {
                    f.stage = 2;
                    const uint32_t c = canon[p].codeB;
                    if (!(c & RT_CODE_LEAF)) {
                        if (walked[c] || (owner[c] >= 0 && owner[c] != (int32_t)k)) { irregular = true; break; }
                        owner[c] = (int32_t)k;
                        walked[c] = 1;
                        stack.push_back({c, 0});
                    }
                    continue;
                }
                if (!L.preorder) {
                    if (!hot[p]) {
                        place_pair(p);
                        if (arena) leaf_children(I, p);
                    }
                    if (!arena) leaf_children(I, p);
                }
                stack.pop_back();
            }
            if (placedTris > 2ull * (uint64_t)nTris + 1024) irregular = true; /* leaves that overlap each other en masse */
        }
        if (placedTris > 2ull * (uint64_t)nTris + 1024) irregular = true;
        if (irregular || pairCur * RT_UNIT_BYTES >= ((uint64_t)1 << 32) || triCur * RT_UNIT_BYTES >= ((uint64_t)1 << 32)) {
            RtLayout d;
            const bool ok = run_dense(canonStore, out);
            out.used = d;
            return ok;
        }

        /* 3. final codes (sequential: the table of oversized leaves is deterministic) */
        out.arena = arena;
        out.used = L;
        out.rootCodes.assign(nModels, 0u);
        out.triBase.assign(nModels, 0);
        std::vector<uint32_t> leafCode(2 * nCanon, 0u);
        for (Inst& I : insts) {
            uint32_t lo = UINT32_MAX;
            for (const Placed& q : I.leaves) lo = q.unit < lo ? q.unit : lo;
            I.triBase = I.leaves.empty() ? 0u : lo;
            uint32_t rootCode = (I.root & RT_CODE_LEAF) ? 0u : unitOf[I.root];
            for (const Placed& q : I.leaves) {
                uint32_t code;
                if (!encode_leaf(q.unit - I.triBase, q.count, out.bigLeaves, &code)) { out.error = "too many oversized leaves"; return false; }
                if (q.pair == UINT32_MAX) rootCode = code;
                else leafCode[2 * (size_t)q.pair + q.side] = code;
            }
            for (int m : I.modelsOf) {
                out.rootCodes[m] = rootCode;
                out.triBase[m] = (int32_t)I.triBase;
            }
        }

        /* 4. the buffers */
        const size_t pairBytes = (size_t)pairCur * RT_UNIT_BYTES, triBytes = (size_t)triCur * RT_UNIT_BYTES;
        const size_t normBytes = (size_t)(arena ? pairCur : triCur) * RT_NORM_BYTES_PER_UNIT;
        if (!out.pairBuf.resize_uninit(pairBytes ? pairBytes : 64) || !out.triBuf.resize_uninit(arena ? 0 : (triBytes ? triBytes : 48)) ||
            !out.normBuf.resize_uninit(normBytes ? normBytes : 36)) { out.error = "out of host memory"; return false; }
        memset(out.pairBuf.data(), 0, out.pairBuf.size());
        if (!arena) memset(out.triBuf.data(), 0, out.triBuf.size());
        memset(out.normBuf.data(), 0, out.normBuf.size());
        const int block = 1 << 14;
        par((int)((nCanon + block - 1) / block), [&](int b) {
            const size_t i1 = (size_t)(b + 1) * block < nCanon ? (size_t)(b + 1) * block : nCanon;
            for (size_t i = (size_t)b * block; i < i1; i++) {
                if (unitOf[i] == UINT32_MAX) continue; /* not reachable from any model */
                DPair d = canon[i];
                d.codeA = (d.codeA & RT_CODE_LEAF) ? leafCode[2 * i] : unitOf[d.codeA];
                d.codeB = (d.codeB & RT_CODE_LEAF) ? leafCode[2 * i + 1] : unitOf[d.codeB];
                memcpy(out.pairBuf.data() + (size_t)unitOf[i] * RT_UNIT_BYTES, &d, sizeof(d));
            }
        });
        unsigned char* const triSpace = arena ? out.pairBuf.data() : out.triBuf.data();
        par((int)insts.size(), [&](int k) {
            const Inst& I = insts[k];
            for (const Placed& q : I.leaves)
                for (uint32_t t = 0; t < q.count; t++) {
                    const size_t unit = (size_t)q.unit + (size_t)t * RT_TRI_UNITS;
                    DTri d;
                    float n9[9];
                    make_dtri(tris[(size_t)I.triOffset + q.start + t], d, n9);
                    memcpy(triSpace + unit * RT_UNIT_BYTES, &d, sizeof(d));
                    memcpy(out.normBuf.data() + unit * RT_NORM_BYTES_PER_UNIT, n9, sizeof(n9));
                }
        });
        canonStore.resize_uninit(0);
        return true;
    }
};

#endif
