// Microbenchmark (round 6): does the instruction front end bind a kernel whose waves stand at DIFFERENT places of a large code body?
// The trace kernels are 16-32 KB of code and their waves (8 per SIMD, 32 per CU) are each somewhere else in it; the tight loops of valu_rate.hip
// (2.6 cycles per v_mul_f32 per SIMD) say nothing about that.  Here every wave runs one of NB code blocks of 512 independent v_mul_f32 (2 KB each,
// kept apart by an asm marker), chosen by its wave index, in a loop: NB = 1 -> all waves in the same 2 KB; NB = 8 / 16 / 32 -> 16 / 32 / 64 KB of
// code live at once.  Output: cycles per wave64 instruction per SIMD at 8 waves per SIMD (assuming 2.4 GHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define M8(c) a0 *= c; a1 *= c; a2 *= c; a3 *= c; a4 *= c; a5 *= c; a6 *= c; a7 *= c;
#define M64(c) M8(c) M8(c) M8(c) M8(c) M8(c) M8(c) M8(c) M8(c)
#define M512(c) M64(c) M64(c) M64(c) M64(c) M64(c) M64(c) M64(c) M64(c)
#define BLK(k) case k: asm volatile("s_nop %0" :: "n"(k % 8)); M512(c) break;
#define BLK8(k) BLK(k) BLK(k + 1) BLK(k + 2) BLK(k + 3) BLK(k + 4) BLK(k + 5) BLK(k + 6) BLK(k + 7)
template <int NB> __global__ void __launch_bounds__(256) k(float* out, float seed, int iters)
{
    float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
    const float c = seed * 0.999f;
    const int wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    const int b = __builtin_amdgcn_readfirstlane((wave * 7) % NB);
    for (int i = 0; i < iters; i++) {
        switch (b) {
            BLK8(0)
            BLK8(8)
            BLK8(16)
            BLK8(24)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int NB> void run(float* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 256;
    k<NB><<<blocks, 256>>>(d, 1.0001f, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<NB><<<blocks, 256>>>(d, 1.0001f, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waveInstr = (double)blocks * 4 * iters * 512;
    printf("NB = %2d (%2d KB of code in use): %8.3f ms -> %.2f cycles per wave-instr per SIMD\n", NB, NB * 2, ms, ms * 1e-3 * 2.4e9 * 1024 / waveInstr);
}
int main()
{
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<1>(d); run<2>(d); run<4>(d); run<8>(d); run<16>(d); run<32>(d);
    return 0;
}
