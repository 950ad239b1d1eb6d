// Issue cost of single VALU instructions on gfx950, pinned with inline asm (8 independent dependent-chains per lane,
// 8 waves per SIMD resident).  Round 5: every kernel runs >= 20 ms (the iteration count is a run-time argument,
// calibrated per instruction) and measures its own clock: wave 0 reads s_memtime (shader clock) and s_memrealtime (constant 100 MHz)
// at both ends, so "cycles per wave64 instruction per SIMD" is at the clock the instruction really ran at — round 3 assumed 2.4 GHz
// on 0.3-ms kernels, launch ramp and clock state included.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_ITER 4096 /* calibration launch; the measured launch is scaled to >= 20 ms */
#define REP8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define K(NAME, ASM)                                                                                                   \
    __global__ void __launch_bounds__(256) k_##NAME(float* out, float seed, int iters, unsigned long long* clocks)     \
    {                                                                                                                  \
        unsigned long long t0 = 0, r0 = 0;                                                                             \
        if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_readcyclecounter(); r0 = wall_clock64(); }          \
        float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f; \
        float c = seed * 0.999f, d = seed * 1.001f;                                                                    \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                          \
        f2 pa0 = {a0, a1}, pa1 = {a2, a3}, pa2 = {a4, a5}, pa3 = {a6, a7}, pa4 = {a1, a0}, pa5 = {a3, a2}, pa6 = {a5, a4}, pa7 = {a7, a6}, pc = {c, d}; \
        unsigned long long qa0 = threadIdx.x, qa1 = qa0 + 1, qa2 = qa0 + 2, qa3 = qa0 + 3, qa4 = qa0 + 4, qa5 = qa0 + 5, qa6 = qa0 + 6, qa7 = qa0 + 7, qc = blockIdx.x; \
        asm volatile("s_mov_b32 s20, 0x3f7fbe77\n s_mov_b32 s21, 0x3f800347" ::: "s20", "s21");                                                                    \
        unsigned long long msk = __ballot(a0 < c);                                                                   \
        asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(a0), "v"(c) : "vcc");                                           \
        for (int i = 0; i < iters; i++) {                                                                              \
            _Pragma("unroll") for (int u = 0; u < 1; u++) { REP8(ASM) }                                                \
        }                                                                                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = __builtin_readcyclecounter() - t0; clocks[1] = wall_clock64() - r0; } \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c + d + pa0.x + pa1.y + pa2.x + pa3.y + pa4.x + pa5.y + pa6.x + pa7.y + (float)(qa0 + qa1 + qa2 + qa3 + qa4 + qa5 + qa6 + qa7);                   \
    }
#define OP_MUL(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_SUB(x) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_MIN(x) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_MAX(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_MIN3(x) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
#define OP_MAX3(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
#define OP_MED3(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
#define OP_CND(x) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(c));
#define OP_CNDS(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(c), "s"(msk));
#define OP_MINU(x) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_MAXI(x) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_MUL64(x) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_SUBS(x) asm volatile("v_sub_f32_e32 %0, s20, %0" : "+v"(x));
#define OP_MULNEG(x) asm volatile("v_mul_f32_e64 %0, -%0, %1" : "+v"(x) : "v"(c));
#define OP_MAXABS(x) asm volatile("v_max_f32_e64 %0, |%0|, %1" : "+v"(x) : "v"(c));
#define OP_CMP(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(x), "v"(c) : "vcc");
#define OP_CMPS(x) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" ::"v"(x), "v"(c) : "s20", "s21");
#define OP_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
#define OP_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_MUL24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_MAD24(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
#define OP_XOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_LSHR(x) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(x));
#define OP_LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(c));
#define OP_ADDU(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_CVT(x) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x));
#define OP_RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
#define OP_RSQ(x) asm volatile("v_rsq_f32 %0, %0" : "+v"(x));
#define OP_PKMUL(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##x) : "v"(pc));
#define OP_PKMULS(x) asm volatile("v_pk_mul_f32 %0, %0, s[20:21]" : "+v"(p##x));
#define OP_PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p##x) : "v"(pc));
#define OP_PKFMAS(x) asm volatile("v_pk_fma_f32 %0, %0, s[20:21], %1" : "+v"(p##x) : "v"(pc));
#define OP_PKADD(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##x) : "v"(pc));
#define OP_FMAS(x) asm volatile("v_fma_f32 %0, %0, s20, %1" : "+v"(x) : "v"(c));
#define OP_MULS(x) asm volatile("v_mul_f32_e32 %0, s20, %0" : "+v"(x));
#define OP_MULLIT(x) asm volatile("v_mul_f32_e32 %0, 0x3f317218, %0" : "+v"(x));
#define OP_ADDLIT(x) asm volatile("v_add_f32_e32 %0, 0x3f317218, %0" : "+v"(x));
#define OP_FMALIT(x) asm volatile("v_fmac_f32_e32 %0, 0x3f317218, %1" : "+v"(x) : "v"(c));
#define OP_FMAK(x) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f317218" : "+v"(x) : "v"(c));
#define OP_MULINL(x) asm volatile("v_mul_f32_e32 %0, 0.5, %0" : "+v"(x));
#define OP_FMAC(x) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
#define OP_MAD64(x) asm volatile("v_mad_i64_i32 %0, vcc, %1, 48, %0" : "+v"(q##x) : "v"(x) : "vcc");
#define OP_LSHLADD64(x) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q##x) : "v"(qc));
#define OP_LSHL64(x) asm volatile("v_lshlrev_b64 %0, 6, %0" : "+v"(q##x));
#define OP_MOV(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(c));
#define OP_BFE(x) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(x));
#define OP_ANDOR(x) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
K(mul, OP_MUL) K(sub, OP_SUB) K(min, OP_MIN) K(max, OP_MAX) K(min3, OP_MIN3) K(max3, OP_MAX3) K(med3, OP_MED3) K(cnd, OP_CND) K(cmp, OP_CMP) K(cmps, OP_CMPS)
K(cnds, OP_CNDS) K(minu, OP_MINU) K(maxi, OP_MAXI) K(mul64, OP_MUL64) K(subs, OP_SUBS) K(mulneg, OP_MULNEG) K(maxabs, OP_MAXABS)
K(pkmul, OP_PKMUL) K(pkmuls, OP_PKMULS) K(pkfma, OP_PKFMA) K(pkfmas, OP_PKFMAS) K(pkadd, OP_PKADD) K(fmas, OP_FMAS) K(muls, OP_MULS)
K(mullit, OP_MULLIT) K(addlit, OP_ADDLIT) K(fmalit, OP_FMALIT) K(fmak, OP_FMAK) K(mulinl, OP_MULINL) K(fmac, OP_FMAC)
K(mad64, OP_MAD64) K(lshladd64, OP_LSHLADD64) K(lshl64, OP_LSHL64)
K(fma, OP_FMA) K(mullo, OP_MULLO) K(mul24, OP_MUL24) K(mad24, OP_MAD24) K(xorb, OP_XOR) K(lshr, OP_LSHR) K(lshladd, OP_LSHLADD) K(addu, OP_ADDU)
K(cvt, OP_CVT) K(rcp, OP_RCP) K(rsq, OP_RSQ) K(mov, OP_MOV) K(bfe, OP_BFE) K(andor, OP_ANDOR)
template <typename F> void run(const char* name, F kern, float* d, unsigned long long* clocks, int blocks, const char* mode)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 256>>>(d, 1.0001f, N_ITER, clocks); hipDeviceSynchronize();
    hipEventRecord(e0); kern<<<blocks, 256>>>(d, 1.0001f, N_ITER, clocks); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int iters = (int)(N_ITER * (22.0 / (ms > 0.01f ? ms : 0.01f))) + N_ITER; /* >= 20 ms */
    hipEventRecord(e0); kern<<<blocks, 256>>>(d, 1.0001f, iters, clocks); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clocks, sizeof(h), hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9; /* shader cycles / seconds of the 100 MHz counter, wave 0's life */
    /* `sparse` (one wave per SIMD): wave 0's own cycles / its iters x 8 instructions = the issue interval of ONE wave.
     * `chip`: the kernel's time x the measured clock x 1,024 SIMDs / all wave-instructions — total work over total time, whatever number of waves
     * the SIMDs really keep resident (round 5's first version divided wave 0's cycles by an assumed eight: void) */
    const double perSimd = blocks >= 2048 ? ms * 1e-3 * ghz * 1e9 * 1024.0 / ((double)blocks * 4 * (double)iters * 8) : (double)h[0] / ((double)iters * 8);
    printf("%-7s %-14s %8.3f ms  clock %.3f GHz -> %5.2f cycles per wave64 instruction per SIMD\n", mode, name, ms, ghz, perSimd);
    fflush(stdout);
}
int main()
{
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    unsigned long long* clocks; hipMalloc(&clocks, 16);
    /* two loads: `chip` = 2,048 workgroups, every SIMD of the chip holds 8 waves (the power-limited sustained rate); `sparse` = 16 workgroups, one wave per
     * SIMD on 16 CUs, 8 independent chains per lane (what a SIMD can issue when the chip is not power-limited) */
#define R(n) run(#n, k_##n, d, clocks, 16, "sparse"); run(#n, k_##n, d, clocks, 2048, "chip");
    R(mad64) R(lshladd64) R(lshl64) R(mul) R(mullit) R(addlit) R(fmalit) R(fmak) R(mulinl) R(fmac) R(muls) R(sub) R(fma) R(fmas) R(pkmul) R(pkmuls) R(pkadd) R(pkfma) R(pkfmas) R(min) R(max) R(min3) R(max3) R(med3) R(cnd) R(cnds) R(cmp) R(cmps) R(minu) R(maxi) R(mul64) R(subs) R(mulneg) R(maxabs) R(mov) R(xorb) R(lshr) R(lshladd) R(addu) R(bfe) R(andor) R(mullo) R(mul24) R(mad24) R(cvt) R(rcp) R(rsq)
    return 0;
}
