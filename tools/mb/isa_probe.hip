#include <hip/hip_runtime.h>
#include <cstdint>
__global__ void k(uint32_t *o, const uint32_t *x, uint32_t key)
{
    uint32_t a = x[threadIdx.x], b = x[threadIdx.x + 64];
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "s"(key));
    o[threadIdx.x] = r;
}
int main()
{
    uint32_t *d, h[128], out[64];
    for (int i = 0; i < 128; ++i) h[i] = 0x9E3779B9u * (i + 1);
    hipMalloc(&d, 4 * 256);
    hipMemcpy(d + 64, h, 4 * 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, d + 64, 0xDEADBEEFu);
    hipMemcpy(out, d, 4 * 64, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) bad += out[i] != (h[i] ^ h[i + 64] ^ 0xDEADBEEFu);
    printf("bitop3 xor3 mismatches: %d\n", bad);
    return bad != 0;
}
