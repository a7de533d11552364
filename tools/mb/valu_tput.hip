// Ad-hoc microbenchmark (not part of the product or the test suite): VALU THROUGHPUT per SIMD on gfx950.
// 8 waves per SIMD on every CU, 8 independent chains per wave, wall time by HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 2048
#define BENCH(NAME, ASM)                                                                         \
    __global__ __launch_bounds__(1024) void NAME(uint32_t *out, uint32_t seed)                   \
    {                                                                                            \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;     \
        uint32_t a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;         \
        uint32_t m = seed | 0xD2511F53u;                                                         \
        for (int i = 0; i < N_ITER; ++i) {                                                       \
            asm volatile(ASM("%0") ASM("%1") ASM("%2") ASM("%3") ASM("%4") ASM("%5") ASM("%6") ASM("%7") \
                         ASM("%0") ASM("%1") ASM("%2") ASM("%3") ASM("%4") ASM("%5") ASM("%6") ASM("%7") \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(m) : "vcc", "s2", "s3");                                          \
        }                                                                                        \
        out[threadIdx.x + blockIdx.x * blockDim.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;      \
    }
#define A_FMA(R) "v_fma_f32 " R ", " R ", %8, " R "\n"
#define A_MULHI(R) "v_mul_hi_u32 " R ", " R ", %8\n"
#define A_MULLO(R) "v_mul_lo_u32 " R ", " R ", %8\n"
#define A_XOR(R) "v_xor_b32 " R ", " R ", %8\n"
#define A_BITOP3(R) "v_bitop3_b32 " R ", " R ", %8, " R " bitop3:0x96\n"
#define A_MAX3(R) "v_max3_f32 " R ", " R ", %8, " R "\n"
#define A_MIN(R) "v_min_f32 " R ", " R ", %8\n"
#define A_RCP(R) "v_rcp_f32 " R ", " R "\n"
#define A_SQRT(R) "v_sqrt_f32 " R ", " R "\n"
#define A_CVT(R) "v_cvt_f32_u32 " R ", " R "\n"
#define A_LSHLADD(R) "v_lshl_add_u32 " R ", " R ", 3, %8\n"
#define A_ADDU(R) "v_add_u32 " R ", " R ", %8\n"
#define A_MINI(R) "v_min_i32 " R ", " R ", %8\n"
#define A_MAXU(R) "v_max_u32 " R ", " R ", %8\n"
#define A_MIN3(R) "v_min3_f32 " R ", " R ", %8, " R "\n"
#define A_MED3(R) "v_med3_f32 " R ", " R ", %8, " R "\n"
#define A_MULF(R) "v_mul_f32 " R ", " R ", %8\n"
#define A_ADDF(R) "v_add_f32 " R ", " R ", %8\n"
#define A_SUBF(R) "v_sub_f32 " R ", " R ", %8\n"
#define A_AND(R) "v_and_b32 " R ", " R ", %8\n"
#define A_LSHL(R) "v_lshlrev_b32 " R ", 3, " R "\n"
#define A_MOV(R) "v_mov_b32 " R ", %8\n"
#define A_CND(R) "v_cndmask_b32 " R ", " R ", %8, vcc\n"
#define A_CMP(R) "v_cmp_lt_f32 vcc, " R ", %8\n"
#define A_FMAC(R) "v_fmac_f32 " R ", %8, %8\n"
#define A_MAX3I(R) "v_max3_i32 " R ", " R ", %8, " R "\n"
#define A_MADU24(R) "v_mad_u32_u24 " R ", " R ", %8, " R "\n"
#define A_BFE(R) "v_bfe_u32 " R ", " R ", 3, 8\n"
#define A_ADD3(R) "v_add3_u32 " R ", " R ", %8, " R "\n"
#define A_CNDS(R) "v_cndmask_b32 " R ", " R ", %8, s[2:3]\n"
#define A_DIVFIX(R) "v_div_fixup_f32 " R ", " R ", %8, " R "\n"
#define A_DIVFMAS(R) "v_div_fmas_f32 " R ", " R ", %8, " R "\n"
#define A_DIVSCALE(R) "v_div_scale_f32 " R ", vcc, " R ", %8, " R "\n"
#define A_CMPCLASS(R) "v_cmp_class_f32 vcc, " R ", %8\n"
#define A_NOT(R) "v_not_b32 " R ", " R "\n"
#define A_SUBU(R) "v_sub_u32 " R ", " R ", %8\n"
#define A_LSHR(R) "v_lshrrev_b32 " R ", 3, " R "\n"
#define A_MBCNT(R) "v_mbcnt_lo_u32_b32 " R ", %8, " R "\n"
#define A_CMPU(R) "v_cmp_lt_u32 vcc, " R ", %8\n"
#define A_OR(R) "v_or_b32 " R ", " R ", %8\n"
#define A_LSHLOR(R) "v_lshl_or_b32 " R ", " R ", 3, %8\n"
#define A_MULU24(R) "v_mul_u32_u24 " R ", " R ", %8\n"
#define A_PERM(R) "v_perm_b32 " R ", " R ", %8, " R "\n"
#define A_CVTB0(R) "v_cvt_f32_ubyte0 " R ", " R "\n"
#define A_CVTB3(R) "v_cvt_f32_ubyte3 " R ", " R "\n"
#define A_ANDOR(R) "v_and_or_b32 " R ", " R ", %8, " R "\n"
#define A_BFI(R) "v_bfi_b32 " R ", " R ", %8, " R "\n"
#define A_ALIGNBIT(R) "v_alignbit_b32 " R ", " R ", %8, 8\n"
#define A_MINU(R) "v_min_u32 " R ", " R ", %8\n"
#define A_FMAMIX(R) "v_fma_mix_f32 " R ", " R ", %8, " R " op_sel_hi:[1,0,0]\n"
#define A_FMAMIXHI(R) "v_fma_mix_f32 " R ", " R ", %8, " R " op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define A_CVTF16(R) "v_cvt_f32_f16 " R ", " R "\n"
#define A_LSHL16(R) "v_lshlrev_b32 " R ", 16, " R "\n"
#define A_LSHR16(R) "v_lshrrev_b32 " R ", 16, " R "\n"
#define A_MAXF(R) "v_max_f32 " R ", " R ", %8\n"
BENCH(k_fma, A_FMA) BENCH(k_mulhi, A_MULHI) BENCH(k_mullo, A_MULLO) BENCH(k_xor, A_XOR) BENCH(k_bitop3, A_BITOP3)
BENCH(k_max3, A_MAX3) BENCH(k_min, A_MIN) BENCH(k_rcp, A_RCP) BENCH(k_sqrt, A_SQRT) BENCH(k_cvt, A_CVT) BENCH(k_lshladd, A_LSHLADD) BENCH(k_addu, A_ADDU) BENCH(k_mini, A_MINI) BENCH(k_maxu, A_MAXU) BENCH(k_min3, A_MIN3) BENCH(k_med3, A_MED3) BENCH(k_mulf, A_MULF) BENCH(k_addf, A_ADDF) BENCH(k_subf, A_SUBF) BENCH(k_and, A_AND) BENCH(k_lshl, A_LSHL) BENCH(k_mov, A_MOV) BENCH(k_cnd, A_CND) BENCH(k_cmp, A_CMP) BENCH(k_fmac, A_FMAC) BENCH(k_max3i, A_MAX3I) BENCH(k_madu24, A_MADU24) BENCH(k_bfe, A_BFE) BENCH(k_add3, A_ADD3)
BENCH(k_cnd_sgpr, A_CNDS) BENCH(k_divfixup, A_DIVFIX) BENCH(k_divfmas, A_DIVFMAS) BENCH(k_divscale, A_DIVSCALE) BENCH(k_cmpclass, A_CMPCLASS) BENCH(k_not, A_NOT)
BENCH(k_subu, A_SUBU) BENCH(k_lshr, A_LSHR) BENCH(k_mbcnt, A_MBCNT) BENCH(k_cmpu, A_CMPU) BENCH(k_or, A_OR) BENCH(k_lshlor, A_LSHLOR) BENCH(k_mulu24, A_MULU24)
BENCH(k_perm, A_PERM) BENCH(k_cvtb0, A_CVTB0) BENCH(k_cvtb3, A_CVTB3) BENCH(k_andor, A_ANDOR) BENCH(k_bfi, A_BFI) BENCH(k_alignbit, A_ALIGNBIT) BENCH(k_minu, A_MINU)
BENCH(k_fmamix, A_FMAMIX) BENCH(k_fmamixhi, A_FMAMIXHI) BENCH(k_cvtf16, A_CVTF16) BENCH(k_lshl16, A_LSHL16) BENCH(k_lshr16, A_LSHR16) BENCH(k_maxf, A_MAXF)
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(1024) void k_pkfma(uint32_t *out, uint32_t seed)
{
    f2 a0 = {1.0f + threadIdx.x, 2.0f}, a1 = a0 * 3.0f, a2 = a0 * 5.0f, a3 = a0 * 7.0f, a4 = a0 * 1.5f, a5 = a0 * 2.5f, a6 = a0 * 3.5f, a7 = a0 * 4.5f;
    f2 m = {1.0001f, 0.5f};
    for (int i = 0; i < N_ITER; ++i) {
#define P(R) "v_pk_fma_f32 " R ", " R ", %8, %8 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
        asm volatile(P("%0") P("%1") P("%2") P("%3") P("%4") P("%5") P("%6") P("%7") P("%0") P("%1") P("%2") P("%3") P("%4") P("%5") P("%6") P("%7")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    }
    out[threadIdx.x + blockIdx.x * blockDim.x] = __float_as_uint(a0.x + a1.x + a2.y + a3.y + a4.x + a5.y + a6.x + a7.y);
}
__global__ __launch_bounds__(1024) void k_mad64(uint32_t *out, uint32_t seed)
{
    unsigned long long a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15;
    uint32_t m = seed | 0xD2511F53u;
    for (int i = 0; i < N_ITER; ++i) {
#define Q(R) "v_mad_u64_u32 " R ", vcc, %8, %8, " R "\n"
        asm volatile(Q("%0") Q("%1") Q("%2") Q("%3") Q("%4") Q("%5") Q("%6") Q("%7") Q("%0") Q("%1") Q("%2") Q("%3") Q("%4") Q("%5") Q("%6") Q("%7")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
    }
    out[threadIdx.x + blockIdx.x * blockDim.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
int main()
{
    uint32_t *d;
    hipMalloc(&d, 512 * 1024 * 4 + 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    printf("%d CUs, clockRate %.2f GHz; cycles below assume that clock\n", cus, ghz);
#define RUN(K)                                                                                              \
    {                                                                                                       \
        hipLaunchKernelGGL(K, dim3(2 * cus), dim3(1024), 0, 0, d, 1u);                                      \
        hipEventRecord(e0, 0);                                                                              \
        hipLaunchKernelGGL(K, dim3(2 * cus), dim3(1024), 0, 0, d, 1u);                                      \
        hipEventRecord(e1, 0); hipEventSynchronize(e1);                                                     \
        float ms; hipEventElapsedTime(&ms, e0, e1);                                                         \
        const double inst_per_simd = 8.0 * N_ITER * 16;  /* 8 waves per SIMD */                             \
        printf("%-10s %.3f ms  %.2f cycles per wave-instruction per SIMD\n", #K, ms, ms * 1e-3 * ghz * 1e9 / inst_per_simd); \
    }
    RUN(k_fma) RUN(k_fma) RUN(k_min) RUN(k_fma) RUN(k_mini) RUN(k_maxu) RUN(k_min3) RUN(k_med3) RUN(k_max3i) RUN(k_mulf) RUN(k_addf) RUN(k_subf) RUN(k_fmac) RUN(k_and) RUN(k_lshl) RUN(k_mov) RUN(k_cnd) RUN(k_cmp) RUN(k_madu24) RUN(k_bfe) RUN(k_add3) RUN(k_addu) RUN(k_xor) RUN(k_bitop3) RUN(k_min) RUN(k_max3) RUN(k_lshladd) RUN(k_cvt) RUN(k_mulhi) RUN(k_mullo) RUN(k_mad64) RUN(k_pkfma) RUN(k_rcp) RUN(k_sqrt) RUN(k_cnd_sgpr) RUN(k_divfixup) RUN(k_divfmas) RUN(k_divscale) RUN(k_cmpclass) RUN(k_not) RUN(k_subu) RUN(k_lshr) RUN(k_mbcnt) RUN(k_cmpu) RUN(k_or) RUN(k_lshlor) RUN(k_mulu24) RUN(k_perm) RUN(k_cvtb0) RUN(k_cvtb3) RUN(k_andor) RUN(k_bfi) RUN(k_alignbit) RUN(k_minu)
    RUN(k_fmamix) RUN(k_fmamixhi) RUN(k_cvtf16) RUN(k_lshl16) RUN(k_lshr16) RUN(k_maxf) RUN(k_fma)
    return 0;
}
