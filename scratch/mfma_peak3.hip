#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// per loop body: 16 MFMAs (4 k-pairs x 2x2 tiles).  MODE 1: 16 ds_read_b32 (1/MFMA) ; MODE 2: 8 ds_read_b32 (reuse) ;
// MODE 3: 4 ds_read_b128 (same bytes as mode 1) ; MODE 0: none
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (i % 97) * 1e-3f;
  __syncthreads();
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  const int lane = threadIdx.x & 63;
  const float* pa = lds + (lane >> 5) * 132 + (lane & 31);
  const float4* p4 = reinterpret_cast<const float4*>(lds) + lane;
  float x[2][8], y[2][8];
  for (int q = 0; q < 8; ++q) { x[0][q] = pa[q * 32]; y[0][q] = pa[q * 32 + 1024]; x[1][q] = x[0][q]; y[1][q] = y[0][q]; }
#pragma unroll 2
  for (int i = 0; i < iters; ++i) {
    const int cur = i & 1, nxt = cur ^ 1;
    if (MODE == 1) {
      for (int q = 0; q < 8; ++q) { x[nxt][q] = pa[q * 264 + (i & 3) * 2]; y[nxt][q] = pa[q * 264 + 2100 + (i & 3) * 2]; }
    } else if (MODE == 2) {
      for (int q = 0; q < 4; ++q) { x[nxt][q] = pa[q * 264 + (i & 3) * 2]; y[nxt][q] = pa[q * 264 + 2100 + (i & 3) * 2]; }
    } else if (MODE == 3) {
      float4 u0 = p4[(i & 3) * 64], u1 = p4[(i & 3) * 64 + 256], v0 = p4[(i & 3) * 64 + 512], v1 = p4[(i & 3) * 64 + 768];
      x[nxt][0] = u0.x; x[nxt][1] = u0.y; x[nxt][2] = u0.z; x[nxt][3] = u0.w; x[nxt][4] = u1.x; x[nxt][5] = u1.y; x[nxt][6] = u1.z; x[nxt][7] = u1.w;
      y[nxt][0] = v0.x; y[nxt][1] = v0.y; y[nxt][2] = v0.z; y[nxt][3] = v0.w; y[nxt][4] = v1.x; y[nxt][5] = v1.y; y[nxt][6] = v1.z; y[nxt][7] = v1.w;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[cur][2 * kk], y[cur][2 * kk], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[cur][2 * kk], y[cur][2 * kk + 1], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[cur][2 * kk + 1], y[cur][2 * kk], a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[cur][2 * kk + 1], y[cur][2 * kk + 1], a3, 0, 0, 0);
    }
  }
  float s = 0;
  for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(float* d, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 5000;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 500);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 16 * (32.0 * 32 * 2 * 2);
  printf("mode %d blocks %d: %.3f ms  %.1f TFLOP/s\n", MODE, blocks, ms, flops / ms / 1e9);
}
int main() {
  float* d; hipMalloc(&d, 8192 * 256 * 4);
  for (int b : {512, 1024}) { run<0>(d, b); run<1>(d, b); run<2>(d, b); run<3>(d, b); }
  return 0;
}
