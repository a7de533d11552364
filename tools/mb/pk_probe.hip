#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(float *o, const float4 *nodes, const float *rays)
{
    const float A = rays[threadIdx.x], C = rays[threadIdx.x + 64], A2 = rays[threadIdx.x + 128], C2 = rays[threadIdx.x + 192];
    float acc = 0.0f;
    for (int i = 0; i < 16; ++i) {
        const float4 q = nodes[i * 64 + threadIdx.x];
        const f2 lo = {q.x, q.y}, hi = {q.z, q.w};
        const f2 a = __builtin_elementwise_fma(lo, (f2){A, A}, (f2){C, C});
        const f2 b = __builtin_elementwise_fma(hi, (f2){A2, A2}, (f2){C2, C2});
        acc += fminf(a.x, b.x) + fmaxf(a.y, b.y);
    }
    o[threadIdx.x] = acc;
}
