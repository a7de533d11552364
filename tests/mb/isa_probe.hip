#include <hip/hip_runtime.h>
#include <cstdint>
__global__ void k(uint32_t *o, const uint32_t *x)
{
    uint32_t a = x[threadIdx.x], b = x[threadIdx.x + 64], c = x[threadIdx.x + 128];
    unsigned long long p;
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(a), "v"(b) : "vcc");
    uint32_t r;
    asm volatile("v_xor3_b32 %0, %1, %2, %3" : "=v"(r) : "v"((uint32_t)p), "v"((uint32_t)(p >> 32)), "v"(c));
    o[threadIdx.x] = r;
}
__global__ void k2(uint32_t *o, const uint32_t *x)
{
    uint32_t a = x[threadIdx.x];
    unsigned long long p = (unsigned long long)a * 0xD2511F53u;
    o[threadIdx.x] = (uint32_t)p ^ (uint32_t)(p >> 32);
}
