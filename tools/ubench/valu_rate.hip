// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the tracer uses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define N_ITER 4096
typedef float float2v __attribute__((ext_vector_type(2)));
template <int OP> __global__ void __launch_bounds__(256) k(float* out, float seed, uint32_t useed)
{
    float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
    uint32_t u0 = useed + threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float c = seed * 0.999f;
    for (int i = 0; i < N_ITER; i++) {
        if (OP == 0) { a0 = a0 * c; a1 = a1 * c; a2 = a2 * c; a3 = a3 * c; a4 = a4 * c; a5 = a5 * c; a6 = a6 * c; a7 = a7 * c; }            // v_mul_f32
        if (OP == 1) { a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; a4 = a4 + c; a5 = a5 + c; a6 = a6 + c; a7 = a7 + c; }            // v_add_f32
        if (OP == 2) { a0 = __builtin_fmaf(a0, c, c); a1 = __builtin_fmaf(a1, c, c); a2 = __builtin_fmaf(a2, c, c); a3 = __builtin_fmaf(a3, c, c); a4 = __builtin_fmaf(a4, c, c); a5 = __builtin_fmaf(a5, c, c); a6 = __builtin_fmaf(a6, c, c); a7 = __builtin_fmaf(a7, c, c); }
        if (OP == 3) { p0 = p0 * c; p1 = p1 * c; p2 = p2 * c; p3 = p3 * c; p0 = p0 * c; p1 = p1 * c; p2 = p2 * c; p3 = p3 * c; }                    // v_pk_mul_f32 (8 instr)
        if (OP == 4) { a0 = __builtin_fminf(a0, c); a1 = __builtin_fmaxf(a1, c); a2 = __builtin_fminf(a2, c); a3 = __builtin_fmaxf(a3, c); a4 = __builtin_fminf(a4, c); a5 = __builtin_fmaxf(a5, c); a6 = __builtin_fminf(a6, c); a7 = __builtin_fmaxf(a7, c); a0 += 1; a1 += 1; a2 += 1; a3 += 1; a4 += 1; a5 += 1; a6 += 1; a7 += 1; } // 8 min/max + 8 add
        if (OP == 5) { a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3); a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7); } // v_rcp_f32
        if (OP == 6) { a0 = c / a0; a1 = c / a1; a2 = c / a2; a3 = c / a3; a4 = c / a4; a5 = c / a5; a6 = c / a6; a7 = c / a7; }              // IEEE divide sequence
        if (OP == 7) { a0 = __builtin_sqrtf(a0); a1 = __builtin_sqrtf(a1); a2 = __builtin_sqrtf(a2); a3 = __builtin_sqrtf(a3); a4 = __builtin_sqrtf(a4); a5 = __builtin_sqrtf(a5); a6 = __builtin_sqrtf(a6); a7 = __builtin_sqrtf(a7); a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c; } // IEEE sqrt sequence + add
        if (OP == 8) { u0 = u0 * 747796405u + 2891336453u; u1 = u1 * 747796405u + 2891336453u; u2 = u2 * 747796405u + 2891336453u; u3 = u3 * 747796405u + 2891336453u; u0 = u0 * 277803737u; u1 = u1 * 277803737u; u2 = u2 * 277803737u; u3 = u3 * 277803737u; } // int mul
        if (OP == 9) { u0 = (u0 >> 5) ^ u0; u1 = (u1 >> 5) ^ u1; u2 = (u2 >> 5) ^ u2; u3 = (u3 >> 5) ^ u3; u0 += 1; u1 += 1; u2 += 1; u3 += 1; }   // shift/xor/add
        if (OP == 10) { a0 = (a0 < c) ? a1 : a0; a1 = (a1 < c) ? a2 : a1; a2 = (a2 < c) ? a3 : a2; a3 = (a3 < c) ? a4 : a3; a4 = (a4 < c) ? a5 : a4; a5 = (a5 < c) ? a6 : a5; a6 = (a6 < c) ? a7 : a6; a7 = (a7 < c) ? a0 : a7; } // cmp + cndmask
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(u0 + u1 + u2 + u3);
}
template <int OP> void run(const char* name, int instrPerIter, float* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8; // 8 waves/SIMD
    k<OP><<<blocks, 256>>>(d, 1.0001f, 12345u); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, 1.0001f, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waveInstr = (double)blocks * 4 * N_ITER * instrPerIter;      // waves * iterations * instr
    double cyclesPerSimd = ms * 1e-3 * 2.4e9;                             // assume 2.4 GHz
    printf("%-28s %8.3f ms  -> %.2f cycles per wave-instr per SIMD (source-level ops; at 2.4 GHz)\n", name, ms, cyclesPerSimd * 1024 / waveInstr);
}
int main()
{
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_mul_f32 x8", 8, d); run<1>("v_add_f32 x8", 8, d); run<2>("v_fma_f32 x8", 8, d); run<3>("v_pk_mul_f32 x8 (16 flops)", 8, d);
    run<4>("min/max x8 + add x8", 16, d); run<5>("v_rcp_f32 x8", 8, d); run<6>("IEEE div x8 (per division)", 8, d); run<7>("IEEE sqrt x8 + add x8 (per pair)", 8, d);
    run<8>("u32 mul-add x4 + mul x4", 8, d); run<9>("shift+xor x4, add x4 (12 ops)", 12, d); run<10>("cmp+cndmask x8 (16 ops)", 16, d);
    return 0;
}
