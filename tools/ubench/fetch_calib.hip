// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access pattern of the BVH traversal: every lane of a wave reads ONE
// 64-byte record (4 x global_load_dwordx4, the DPair fetch of rt_kernels.h) at an unrelated address.  MI355X_MICROARCH.md calibrates
// FETCH_SIZE x 2 for wide coalesced streaming reads only.  Known bytes here: every record of a buffer far larger than the caches
// (L2 4 MB per XCD, Infinity Cache 256 MB) is read exactly once, in a pseudo-random permutation (each 64-byte record is its own
// request; the two halves of a 128-byte line are touched at unrelated times).  Patterns:
//   0  streaming  : lane i of consecutive waves reads consecutive 16-byte pieces (the guide's calibrated case)
//   1  divergent64: one aligned 64-byte record per lane per step, random permutation of all records
//   2  divergent48: one 48-byte record (3 x dwordx4, the DTri fetch) per lane per step, random permutation
//   3  line128    : one aligned 128-byte line per lane, lower half first, upper half afterwards (whole-line or sector fills?)
// Run under rocprofv3 --pmc FETCH_SIZE / TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum / TCC_MISS_sum (tools/fetch_calib.sh); prints the
// bytes the kernel asked for, so that counter / bytes is the factor.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t perm(uint32_t i, uint32_t nMask)
{
    // a bijection on [0, 2^k): odd multiplier + xorshift + odd multiplier, all mod 2^k
    i = (i * 2654435761u) & nMask;
    i ^= i >> 7;
    i = (i * 40503u + 12345u) & nMask;
    i ^= i >> 11;
    i = (i * 2246822519u) & nMask;
    return i;
}

template <int PATTERN>
__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ buf, uint32_t nRecMask, uint32_t recsPerThread, float* out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nThreads = gridDim.x * blockDim.x;
    float acc = 0.0f;
    for (uint32_t k = 0; k < recsPerThread; k++) {
        const uint32_t i = k * nThreads + tid;
        if (PATTERN == 0) {
            const float4 v = buf[i];
            acc += v.x + v.y + v.z + v.w;
        } else if (PATTERN == 1) {
            const float4* r = buf + (size_t)perm(i, nRecMask) * 4;
            const float4 a = r[0], b = r[1], c = r[2], d = r[3];
            acc += a.x + b.y + c.z + d.w;
        } else if (PATTERN == 2) {
            const float4* r = buf + (size_t)perm(i, nRecMask) * 3;
            const float4 a = r[0], b = r[1], c = r[2];
            acc += a.x + b.y + c.z;
        } else { /* 3: one aligned 128-byte line per lane, first its lower 64 bytes, then — dependent on them — its upper 64 bytes:
                  * does a miss bring the whole line (the second half hits) or one 64-byte sector (it misses again)? */
            const float4* r = buf + (size_t)perm(i, nRecMask) * 8;
            const float4 a = r[0], b = r[1], c = r[2], d = r[3];
            const float s0 = a.x + b.y + c.z + d.w;
            const float4* r2 = r + 4 + (s0 == 77.0f ? 1 : 0); /* address depends on the first half's data */
            const float4 e = r2[0], f = r2[1], g = r2[2], h = r2[3];
            acc += s0 + e.x + f.y + g.z + h.w;
        }
    }
    if (acc == 123.456f) out[0] = acc; // never true: keeps the loads
}

int main(int argc, char** argv)
{
    const int pattern = argc > 1 ? atoi(argv[1]) : 1;
    const int logRecs = argc > 2 ? atoi(argv[2]) : 25; // 2^25 records
    const uint32_t nRec = 1u << logRecs;
    const size_t recBytes = pattern == 0 ? 16 : pattern == 1 ? 64 : pattern == 2 ? 48 : 128;
    const size_t bytes = (size_t)nRec * recBytes;
    float4* buf;
    float* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    const uint32_t threads = 256 * 2048, recsPerThread = nRec / threads;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (pattern == 0) k_read<0><<<2048, 256>>>(buf, nRec - 1, recsPerThread, out);
    else if (pattern == 1) k_read<1><<<2048, 256>>>(buf, nRec - 1, recsPerThread, out);
    else if (pattern == 2) k_read<2><<<2048, 256>>>(buf, nRec - 1, recsPerThread, out);
    else k_read<3><<<2048, 256>>>(buf, nRec - 1, recsPerThread, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("pattern %d: %u records x %zu B = %.3f GB read exactly once, %.3f ms, %.1f GB/s\n", pattern, nRec, recBytes, bytes / 1e9, ms, bytes / 1e6 / ms);
    return 0;
}
