#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f + 1.f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
  }
  float s = 0;
  for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* d; hipMalloc(&d, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 512, 1024}) {
    int iters = 20000;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 /*waves*/ * iters * 4 * (32.0 * 32 * 2 * 2);
    printf("blocks %d: %.3f ms  %.1f TFLOP/s\n", blocks, ms, flops / ms / 1e9);
  }
  return 0;
}
