// Ad-hoc microbenchmark (not part of the product or the test suite): issue cost of a few VALU
// instructions on gfx950, one wave per SIMD, 8 independent chains each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 4096
#define BENCH(NAME, ASM)                                                                         \
    __global__ void NAME(uint32_t *out, uint32_t seed)                                           \
    {                                                                                            \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;     \
        uint32_t a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;         \
        uint32_t m = seed | 0xD2511F53u;                                                         \
        long long t0 = clock64();                                                                \
        for (int i = 0; i < N_ITER; ++i) {                                                       \
            asm volatile(ASM("%0") ASM("%1") ASM("%2") ASM("%3") ASM("%4") ASM("%5") ASM("%6") ASM("%7") \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(m));                                                              \
        }                                                                                        \
        long long t1 = clock64();                                                                \
        out[threadIdx.x + blockIdx.x * blockDim.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;      \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (uint32_t)(t1 - t0);             \
    }
#define A_MULHI(R) "v_mul_hi_u32 " R ", " R ", %8\n"
#define A_MULLO(R) "v_mul_lo_u32 " R ", " R ", %8\n"
#define A_FMA(R) "v_fma_f32 " R ", " R ", %8, " R "\n"
#define A_XOR(R) "v_xor_b32 " R ", " R ", %8\n"
#define A_RCP(R) "v_rcp_f32 " R ", " R "\n"
#define A_SQRT(R) "v_sqrt_f32 " R ", " R "\n"
#define A_MAX3(R) "v_max3_f32 " R ", " R ", %8, " R "\n"
#define A_MUL24(R) "v_mul_u32_u24 " R ", " R ", %8\n"
#define A_MULHI24(R) "v_mul_hi_u32_u24 " R ", " R ", %8\n"
#define A_BITOP3(R) "v_bitop3_b32 " R ", " R ", %8, " R " bitop3:0x96\n"
#define A_CNDMASK(R) "v_cndmask_b32 " R ", " R ", %8, vcc\n"
#define A_CVT(R) "v_cvt_f32_u32 " R ", " R "\n"
#define A_MIN(R) "v_min_f32 " R ", " R ", %8\n"
#define A_LSHLADD(R) "v_lshl_add_u32 " R ", " R ", 3, %8\n"
BENCH(k_mulhi, A_MULHI)
BENCH(k_mullo, A_MULLO)
BENCH(k_fma, A_FMA)
BENCH(k_xor, A_XOR)
BENCH(k_rcp, A_RCP)
BENCH(k_sqrt, A_SQRT)
BENCH(k_max3, A_MAX3)
BENCH(k_mul24, A_MUL24)
BENCH(k_mulhi24, A_MULHI24)
BENCH(k_bitop3, A_BITOP3)
BENCH(k_cndmask, A_CNDMASK)
BENCH(k_cvt, A_CVT)
BENCH(k_min, A_MIN)
BENCH(k_lshladd, A_LSHLADD)
__global__ void k_pkfma(uint32_t *out, uint32_t seed)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {1.0f + threadIdx.x, 2.0f}, a1 = a0 * 3.0f, a2 = a0 * 5.0f, a3 = a0 * 7.0f;
    f2 m = {1.0001f, 0.5f};
    long long t0 = clock64();
    for (int i = 0; i < N_ITER; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %4, %4 op_sel:[0,0,1] op_sel_hi:[1,0,1]\nv_pk_fma_f32 %1, %1, %4, %4 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                     "v_pk_fma_f32 %2, %2, %4, %4 op_sel:[0,0,1] op_sel_hi:[1,0,1]\nv_pk_fma_f32 %3, %3, %4, %4 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                     "v_pk_fma_f32 %0, %0, %4, %4 op_sel:[0,0,1] op_sel_hi:[1,0,1]\nv_pk_fma_f32 %1, %1, %4, %4 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                     "v_pk_fma_f32 %2, %2, %4, %4 op_sel:[0,0,1] op_sel_hi:[1,0,1]\nv_pk_fma_f32 %3, %3, %4, %4 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
    }
    long long t1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = __float_as_uint(a0.x + a1.x + a2.y + a3.y);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (uint32_t)(t1 - t0);
}
__global__ void k_pkfma_plain(uint32_t *out, uint32_t seed)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {1.0f + threadIdx.x, 2.0f}, a1 = a0 * 3.0f, a2 = a0 * 5.0f, a3 = a0 * 7.0f;
    f2 m = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
    long long t0 = clock64();
    for (int i = 0; i < N_ITER; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5\n"
                     "v_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
    }
    long long t1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = __float_as_uint(a0.x + a1.x + a2.y + a3.y);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (uint32_t)(t1 - t0);
}
__global__ void k_mad64(uint32_t *out, uint32_t seed)
{
    unsigned long long a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    uint32_t m = seed | 0xD2511F53u;
    long long t0 = clock64();
    for (int i = 0; i < N_ITER; ++i) {
        asm volatile("v_mad_u64_u32 %0, vcc, %4, %4, %0\nv_mad_u64_u32 %1, vcc, %4, %4, %1\nv_mad_u64_u32 %2, vcc, %4, %4, %2\nv_mad_u64_u32 %3, vcc, %4, %4, %3\n"
                     "v_mad_u64_u32 %0, vcc, %4, %4, %0\nv_mad_u64_u32 %1, vcc, %4, %4, %1\nv_mad_u64_u32 %2, vcc, %4, %4, %2\nv_mad_u64_u32 %3, vcc, %4, %4, %3\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");
    }
    long long t1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (uint32_t)(t1 - t0);
}
// IEEE divide and sqrt as the compiler emits them under -fhip-fp32-correctly-rounded-divide-sqrt
__global__ void k_div(float *out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    long long t0 = clock64();
    for (int i = 0; i < N_ITER; ++i) { a0 = seed / a0 + 1.5f; a1 = seed / a1 + 1.5f; a2 = seed / a2 + 1.5f; a3 = seed / a3 + 1.5f; }
    long long t1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<uint32_t *>(out)[1 << 20] = (uint32_t)(t1 - t0);
}
__global__ void k_sqrt_ieee(float *out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    long long t0 = clock64();
    for (int i = 0; i < N_ITER; ++i) { a0 = sqrtf(a0) + seed; a1 = sqrtf(a1) + seed; a2 = sqrtf(a2) + seed; a3 = sqrtf(a3) + seed; }
    long long t1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<uint32_t *>(out)[1 << 20] = (uint32_t)(t1 - t0);
}
int main()
{
    uint32_t *d;
    hipMalloc(&d, ((1 << 20) + 16) * 4);
    uint32_t cyc;
#define RUN(K, NOPS, ARG)                                                                  \
    for (int waves = 1; waves <= 4; waves *= 2) {                                             \
        hipLaunchKernelGGL(K, dim3(1), dim3(256 * waves), 0, 0, (decltype(ARG) *)d, ARG);  \
        hipDeviceSynchronize();                                                            \
        hipMemcpy(&cyc, d + (1 << 20), 4, hipMemcpyDeviceToHost);                          \
        printf("%-12s waves/SIMD %d: %.2f clock64 ticks per instruction per wave (ticks are 100 MHz: x%.1f for shader cycles)\n", #K, waves, (double)cyc / (N_ITER * NOPS), 24.0); \
    }
    RUN(k_fma, 8, 1u) RUN(k_xor, 8, 1u) RUN(k_max3, 8, 1u) RUN(k_mulhi, 8, 1u) RUN(k_mullo, 8, 1u) RUN(k_mul24, 8, 1u) RUN(k_mulhi24, 8, 1u) RUN(k_bitop3, 8, 1u) RUN(k_cndmask, 8, 1u) RUN(k_cvt, 8, 1u) RUN(k_min, 8, 1u) RUN(k_lshladd, 8, 1u)
    RUN(k_pkfma, 8, 1u) RUN(k_pkfma_plain, 8, 1u) RUN(k_mad64, 8, 1u) RUN(k_rcp, 8, 1u) RUN(k_sqrt, 8, 1u) RUN(k_div, 4, 1.25f) RUN(k_sqrt_ieee, 4, 1.25f)
    return 0;
}
