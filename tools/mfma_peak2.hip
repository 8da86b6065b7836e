// Ceiling probe: v_mfma_f32_32x32x2_f32 issued from registers (mode 0) vs with operands re-read from LDS (mode 1), 1-4
// waves per SIMD, with the in-kernel clock.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak2.hip -o /tmp/mfma_peak2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// MODE 0: pure MFMA from registers; MODE 1: operands re-read from LDS each k-pair (2+2 ds_read_b32 per 4 MFMA), pipelined
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (i % 97) * 1e-3f;
  __syncthreads();
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  const int lane = threadIdx.x & 63;
  const float* pa = lds + (lane >> 5) * 129 + (lane & 31);
  const float* pb = lds + 4128 + (lane >> 5) * 128 + (lane & 31);
  float x0 = pa[0], x1 = pa[32], y0 = pb[0], y1 = pb[32];
  for (int i = 0; i < iters; ++i) {
    float nx0 = x0, nx1 = x1, ny0 = y0, ny1 = y1;
    if (MODE == 1) {
      const int o = (i & 15) * 2;
      nx0 = pa[o * 129]; nx1 = pa[o * 129 + 32]; ny0 = pb[o * 128]; ny1 = pb[o * 128 + 32];
      __builtin_amdgcn_sched_barrier(0);
    }
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, a3, 0, 0, 0);
    x0 = nx0; x1 = nx1; y0 = ny0; y1 = ny1;
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
  float s = 0;
  for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(float* d, int blocks) {
  unsigned long long* clk; hipMalloc(&clk, 16); unsigned long long h[2];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 20000;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 2000, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, clk); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 4 * (32.0 * 32 * 2 * 2);
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("mode %d blocks %d (%.0f waves/SIMD): %.3f ms  %.1f TFLOP/s  clock %.3f GHz (cycles/MFMA/wave %.1f)\n", MODE, blocks, blocks / 256.0, ms, flops / ms / 1e9, (double)h[0] / h[1] * 0.1, (double)h[0] / (iters * 4.0));
}
int main() {
  float* d; hipMalloc(&d, 8192 * 256 * 4);
  for (int b : {256, 512, 768, 1024}) run<0>(d, b);
  for (int b : {256, 512, 768, 1024}) run<1>(d, b);
  return 0;
}
