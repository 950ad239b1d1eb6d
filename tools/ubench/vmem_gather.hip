// What does the vector-memory path (TA + TCP = the CU's L1) charge a wave64 `global_load_dwordx4` for?  Round 5: the BVH trace kernels
// sit at 0.83-0.84 L1 accesses per clock per CU with the TA 76-85 % busy (profiles/r05_memory_path.txt) — this measures the rule behind
// that number: cycles per load instruction per CU for address patterns that differ only in how the 64 lanes' 16-byte pieces fall into
// 64-byte / 128-byte blocks, everything L1- or L2-resident.  24 waves per CU (6 per SIMD) like the trace kernel, 4 independent loads per
// iteration, clock measured inside the kernel (s_memtime / s_memrealtime).  Run under rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum
// TA_TA_BUSY_sum for the access counts (tools/r05_call2.sh).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { P_SAME, P_SEQ, P_LINE, P_HALF, P_QUAD64, P_QUAD128, P_FARPAIR, P_RAND64, P_RAND64_25, P_NODE_OWN, P_NODE_QUAD, P_NODE_OWN_25, P_NODE_QUAD_25, P_COUNT };
static const char* NAMES[P_COUNT] = {
    "same16            all lanes the same 16 bytes",
    "seq               lane i reads bytes [16 i, 16 i + 16): 1 KB contiguous",
    "line              every lane another 128-byte line",
    "half              lanes 2k, 2k+1 share a line, different 64-byte halves",
    "quad64            lanes 4k..4k+3 = the four 16-byte pieces of one 64-byte block, 16 lines",
    "quad128           lanes 4k..4k+3 in one 128-byte line, two per 64-byte half",
    "farpair           lanes i and i + 32 share a 64-byte block (not neighbours)",
    "rand64            every lane a random 64-byte record (first 16 bytes)",
    "rand64_25of64     the same, 25 of 64 lanes active",
    "node_own          4 loads: every lane its own random 64-byte record, quarters 0..3 (the trace kernel's inner step)",
    "node_quad         4 loads: load j = the record of lane 4k + j, lane 4k + q reads quarter q (quad-cooperative)",
    "node_own_25of64   node_own with 25 of 64 lanes active",
    "node_quad_25of64  node_quad where 25 of 64 lanes own a record (a load runs for the quads whose lane j does)",
};

__global__ void __launch_bounds__(256) gather(const char* __restrict__ base, const uint32_t* __restrict__ offs, const unsigned long long* __restrict__ masks,
                                              int pattern, int iters, uint32_t regionMask, float* out, unsigned long long* clocks)
{
    unsigned long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_readcyclecounter(); r0 = wall_clock64(); }
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    // per-lane offsets of the four loads of one iteration (bytes), fixed per wave; the iteration adds a wave-uniform stride
    uint32_t o0 = offs[((size_t)pattern * 4 + 0) * 64 + lane], o1 = offs[((size_t)pattern * 4 + 1) * 64 + lane];
    uint32_t o2 = offs[((size_t)pattern * 4 + 2) * 64 + lane], o3 = offs[((size_t)pattern * 4 + 3) * 64 + lane];
    const unsigned long long m0 = masks[pattern * 4 + 0], m1 = masks[pattern * 4 + 1], m2 = masks[pattern * 4 + 2], m3 = masks[pattern * 4 + 3];
    const bool a0 = (m0 >> lane) & 1, a1 = (m1 >> lane) & 1, a2 = (m2 >> lane) & 1, a3 = (m3 >> lane) & 1;
    float acc = 0.0f;
    uint32_t shift = (uint32_t)wave * 8192u;
    for (int i = 0; i < iters; i++) {
        float4 v0 = make_float4(0, 0, 0, 0), v1 = v0, v2 = v0, v3 = v0;
        if (a0) v0 = *reinterpret_cast<const float4*>(base + ((o0 + shift) & regionMask));
        if (a1) v1 = *reinterpret_cast<const float4*>(base + ((o1 + shift) & regionMask));
        if (a2) v2 = *reinterpret_cast<const float4*>(base + ((o2 + shift) & regionMask));
        if (a3) v3 = *reinterpret_cast<const float4*>(base + ((o3 + shift) & regionMask));
        acc += (v0.x + v0.y + v0.z + v0.w) + (v1.x + v1.y + v1.z + v1.w) + (v2.x + v2.y + v2.z + v2.w) + (v3.x + v3.y + v3.z + v3.w);
        shift += 8192u * 977u; /* another 8-KB-aligned window of the region next time */
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = __builtin_readcyclecounter() - t0; clocks[1] = wall_clock64() - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

int main(int argc, char** argv)
{
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    const int fixedIters = argc > 2 ? atoi(argv[2]) : 0; /* counter runs (rocprofv3 --pmc): ONE dispatch per pattern with this many iterations */
    const size_t regions[2] = {16u << 10, 2u << 20}; /* 16 KB: L1-resident per CU; 2 MB: L2-resident */
    char* d; hipMalloc(&d, 4u << 20); hipMemset(d, 0, 4u << 20);
    uint32_t h_offs[P_COUNT * 4 * 64];
    unsigned long long h_masks[P_COUNT * 4];
    uint32_t seed = 12345;
    unsigned long long m25 = 0; { int n = 0; while (n < 25) { int b = rnd(seed) % 64; if (!((m25 >> b) & 1)) { m25 |= 1ull << b; n++; } } }
    for (int p = 0; p < P_COUNT; p++)
        for (int j = 0; j < 4; j++) {
            h_masks[p * 4 + j] = ~0ull;
            for (int l = 0; l < 64; l++) {
                uint32_t o = 0;
                const uint32_t win = (uint32_t)j * 2048u; /* the four loads of an iteration go to different 2-KB windows (independent lines) */
                switch (p) {
                case P_SAME: o = win; break;
                case P_SEQ: o = win + 16u * l; break;
                case P_LINE: o = (uint32_t)j * 16u + 128u * l; break; /* 64 lines = the whole 8-KB window: the four loads use other pieces of them */
                case P_HALF: o = win / 2 * 0 + (uint32_t)j * 16u + 64u * l; break;
                case P_QUAD64: o = win + 128u * (l / 4) + 16u * (l % 4); break;
                case P_QUAD128: o = win + 128u * (l / 4) + 32u * (l % 4); break;
                case P_FARPAIR: o = win + 64u * (l % 32) + 16u * (l / 32); break;
                default: break;
                }
                h_offs[(p * 4 + j) * 64 + l] = o;
            }
        }
    /* random 64-byte records inside an 8-KB window (128 records): one per lane */
    uint32_t rec[64];
    for (int l = 0; l < 64; l++) rec[l] = (rnd(seed) % 128u) * 64u;
    for (int j = 0; j < 4; j++)
        for (int l = 0; l < 64; l++) {
            uint32_t r2[64];
            h_offs[(P_RAND64 * 4 + j) * 64 + l] = ((rec[l] + 64u * 31u * j) & 8191u);       /* four other random records */
            h_offs[(P_RAND64_25 * 4 + j) * 64 + l] = h_offs[(P_RAND64 * 4 + j) * 64 + l];
            h_masks[P_RAND64_25 * 4 + j] = m25;
            h_offs[(P_NODE_OWN * 4 + j) * 64 + l] = rec[l] + 16u * j;
            h_offs[(P_NODE_OWN_25 * 4 + j) * 64 + l] = rec[l] + 16u * j;
            h_masks[P_NODE_OWN_25 * 4 + j] = m25;
            h_offs[(P_NODE_QUAD * 4 + j) * 64 + l] = rec[(l & ~3) + j] + 16u * (l & 3);
            h_offs[(P_NODE_QUAD_25 * 4 + j) * 64 + l] = rec[(l & ~3) + j] + 16u * (l & 3);
            (void)r2;
        }
    for (int j = 0; j < 4; j++) { /* load j runs for the quads whose lane 4k + j owns a record */
        unsigned long long m = 0;
        for (int q = 0; q < 16; q++)
            if ((m25 >> (4 * q + j)) & 1) m |= 0xfull << (4 * q);
        h_masks[P_NODE_QUAD_25 * 4 + j] = m;
    }
    uint32_t* d_offs; unsigned long long* d_masks; float* d_out; unsigned long long* d_clk;
    hipMalloc(&d_offs, sizeof(h_offs)); hipMemcpy(d_offs, h_offs, sizeof(h_offs), hipMemcpyHostToDevice);
    hipMalloc(&d_masks, sizeof(h_masks)); hipMemcpy(d_masks, h_masks, sizeof(h_masks), hipMemcpyHostToDevice);
    const int blocks = 256 * 6; /* 6 workgroups of 4 waves per CU = 24 waves per CU */
    hipMalloc(&d_out, (size_t)blocks * 256 * 4); hipMalloc(&d_clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("# cycles per wave64 global_load_dwordx4 per CU (256 CUs, 24 waves per CU, 4 independent loads per iteration), clock measured in the kernel\n");
    for (int r = 0; r < 2; r++)
        for (int p = 0; p < P_COUNT; p++) {
            if (only >= 0 && only != r * P_COUNT + p) continue;
            const uint32_t mask = (uint32_t)regions[r] - 1;
            int iters = 2000;
            if (fixedIters > 0) {
                gather<<<blocks, 256>>>(d, d_offs, d_masks, p, fixedIters, mask, d_out, d_clk); hipDeviceSynchronize();
                int lpi = 0;
                for (int j = 0; j < 4; j++) if (h_masks[p * 4 + j]) lpi++;
                printf("dispatch %d: %s pattern %d, %d iterations x %d loads x %d waves per CU | %s\n", r * P_COUNT + p, r ? "L2" : "L1", p, fixedIters, lpi, blocks * 4 / 256, NAMES[p]);
                continue;
            }
            gather<<<blocks, 256>>>(d, d_offs, d_masks, p, iters, mask, d_out, d_clk); hipDeviceSynchronize();
            hipEventRecord(e0); gather<<<blocks, 256>>>(d, d_offs, d_masks, p, iters, mask, d_out, d_clk); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            iters = (int)(iters * (10.0 / (ms > 0.01f ? ms : 0.01f))) + 2000; /* ~10 ms */
            hipEventRecord(e0); gather<<<blocks, 256>>>(d, d_offs, d_masks, p, iters, mask, d_out, d_clk); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, d_clk, sizeof(h), hipMemcpyDeviceToHost);
            const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
            int loadsPerIter = 0, lanes = 0;
            for (int j = 0; j < 4; j++) if (h_masks[p * 4 + j]) { loadsPerIter++; lanes += __builtin_popcountll(h_masks[p * 4 + j]); }
            const double instrPerCU = (double)blocks * 4 / 256.0 * (double)iters * loadsPerIter;
            printf("%s region %4zu KB: %8.3f ms, %.3f GHz -> %6.2f cycles per load per CU (%.1f lanes per load) | %s\n", r ? "L2" : "L1", regions[r] >> 10, ms, ghz,
                   ms * 1e-3 * ghz * 1e9 / instrPerCU, (double)lanes / loadsPerIter, NAMES[p]);
            fflush(stdout);
        }
    return 0;
}
