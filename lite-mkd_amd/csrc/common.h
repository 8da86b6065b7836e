// Shared device/host helpers for the Lite-MKD HIP hot path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define LMKD_OK 0
#define LMKD_EINVAL (-1)
#define LMKD_EHIP (-2)

extern "C" void lmkd_set_error(const char* fmt, ...);

#define LMKD_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      lmkd_set_error(__VA_ARGS__);              \
      return LMKD_EINVAL;                       \
    }                                           \
  } while (0)

#define LMKD_CHECK_LAUNCH(name)                                                  \
  do {                                                                           \
    hipError_t e_ = hipGetLastError();                                           \
    if (e_ != hipSuccess) {                                                      \
      lmkd_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
      return LMKD_EHIP;                                                          \
    }                                                                            \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Unsigned division by a runtime constant: q = (n * mul) >> 32 >> sh  (n < 2^31).
struct FastDiv {
  uint32_t mul, sh, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  if (d == 1) { f.mul = 0; f.sh = 0; return f; }
  uint32_t l = 0;
  while ((1u << l) < d) ++l;
  uint64_t m = ((1ull << (32 + l)) + d - 1) / d;   // ceil(2^(32+l)/d), fits in 33 bits
  f.mul = (uint32_t)(m - (1ull << 32));            // low 32 bits (add-back form)
  f.sh = l;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  if (f.d == 1) return n;
  uint32_t t = __umulhi(n, f.mul);
  return (t + ((n - t) >> 1)) >> (f.sh - 1);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread.
__device__ __forceinline__ float block_sum_256(float v, float* red /*>=4 floats*/) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
