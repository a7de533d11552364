#include <hip/hip_runtime.h>
__global__ void k(float *o, const float *x) { o[threadIdx.x] = sqrtf(x[threadIdx.x]); }
__global__ void kd(float *o, const float *x, const float *y) { o[threadIdx.x] = x[threadIdx.x] / y[threadIdx.x]; }
