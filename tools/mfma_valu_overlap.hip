// Do VALU instructions of one wave issue under the MFMAs of ANOTHER wave of the same SIMD?  (round 5: the patch kernels' time is close to
// MFMA cycles + VALU cycles of a SIMD's waves, as if the two never overlapped.)  One workgroup of 512 threads per CU = two waves per SIMD.
// mode 1: every even wave runs N MFMAs (4 independent accumulators), odd waves exit; mode 2: odd waves run V independent v_fma_f32 per
// MFMA of the other wave, even waves exit; mode 3: both.   build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o /tmp/ov
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* out, int n, int vper, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool mf = (wave & 1) == 0;      // waves w and w + 4 share a SIMD?  (waves are dealt round-robin: wave w -> SIMD w % 4) - so pair (w, w + 4): use bit 2
  const bool is_m = (wave & 4) == 0;
  (void)mf;
  if (is_m) {
    if (!(mode & 1)) return;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    if (SHAPE == 16) {
      f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
      for (int i = 0; i < n; i += 4) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
      }
      out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
      f32x16 c0, c1;
      for (int i = 0; i < 16; ++i) c0[i] = c1[i] = 0.f;
      for (int i = 0; i < n; i += 2) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
      }
      out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1];
    }
  } else {
    if (!(mode & 2)) return;
    float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
    const float m = 1.0001f, c = 0.5f;
    const long tot = (long)n * vper;
    for (long i = 0; i < tot; i += 8) {
      x0 = __builtin_fmaf(x0, m, c); x1 = __builtin_fmaf(x1, m, c); x2 = __builtin_fmaf(x2, m, c); x3 = __builtin_fmaf(x3, m, c);
      x4 = __builtin_fmaf(x4, m, c); x5 = __builtin_fmaf(x5, m, c); x6 = __builtin_fmaf(x6, m, c); x7 = __builtin_fmaf(x7, m, c);
      asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    }
    out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  }
}
template <int SHAPE>
static float run(float* d, int n, int vper, int mode) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(512), 0, 0, d, n, vper, mode);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(512), 0, 0, d, n, vper, mode);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}
int main() {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  const int n = 1 << 16;
  for (int vper : {1, 2, 4, 6}) {
    printf("16x16x32: %d VALU per MFMA: mfma alone %.0f us, valu alone %.0f us, both %.0f us\n", vper, run<16>(d, n, vper, 1), run<16>(d, n, vper, 2), run<16>(d, n, vper, 3));
    printf("32x32x16: %d VALU per MFMA: mfma alone %.0f us, valu alone %.0f us, both %.0f us\n", 2 * vper, run<32>(d, n / 2, 2 * vper, 1), run<32>(d, n / 2, 2 * vper, 2), run<32>(d, n / 2, 2 * vper, 3));
  }
  return 0;
}
