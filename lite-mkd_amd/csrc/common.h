// Shared device/host helpers for the Lite-MKD HIP hot path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

// gfx950 only: the ticketed single-launch reductions (norm_pool.hip colsum_ticket, gemm.hip split-K) hand partial sums from one workgroup
// to another through relaxed agent-scope atomics behind s_waitcnt vmcnt(0), relying on the write-through completion of sc1 stores on
// this target; another target needs release / acquire there.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "liblmkd_hip is written for gfx950 (MI355X): see the note above before porting the ticketed reductions"
#endif

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

// include/lmkd.h: the range bookkeeping of one launch in compute mode 4 (every member nullable; a null struct = none)
struct lmkd_amax_desc {
  const void* x_words;
  const void* dy_words;
  void* out_words;
  const void* ref_words;
  int flags;
};
#define LMKD_AMAX_FENCED 1      // lmkd_amax_desc::flags: the caller withheld an operand's maximum because the range fence flagged its tensor

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

// max |x| of a tensor, folded into a device word by the kernel that writes the tensor (lmkd_amax_next; read by the two-plane fp16
// convolutions, conv_patch16.h): the fp32 bits of a non-negative number order like unsigned integers, so the fold is an atomicMax
__device__ __forceinline__ float amax4(float m, const float4& v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
// A maximum is LMKD_AMAX_SLOTS words, 64 bytes apart (one memory-side atomic request each), per frame segment: every wave of the
// producer folds its own maximum into the slot of its (workgroup, wave) - float atomics execute at the memory side and 4096 of them on
// ONE address cost a 321 MB elementwise pass as much as the pass itself (85 -> 160 us) - and a consumer takes the maximum over the slots
// (one load per lane + a wave reduction, amax_read).  Layout of a tensor's words: [segment][slot][16 words, the first one used].
#define LMKD_AMAX_SLOTS 64
#define LMKD_AMAX_STRIDE 16
#define LMKD_AMAX_SEG_WORDS (LMKD_AMAX_SLOTS * LMKD_AMAX_STRIDE)
// WPB: waves per workgroup where the caller knows it at compile time (blockDim.x is a scalar load from the kernel-argument segment,
// a microsecond on this machine; conv_patch.h)
template <int WPB = 0>
__device__ __forceinline__ void amax_commit(unsigned* __restrict__ word, float m) {      // call with the whole wave active
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f)
    atomicMax(word + ((blockIdx.x * (WPB ? WPB : (int)(blockDim.x >> 6)) + (threadIdx.x >> 6)) & (LMKD_AMAX_SLOTS - 1)) * LMKD_AMAX_STRIDE, __float_as_uint(m));
}
// Range statistics beside a maximum (the range fence of compute mode 4, include/lmkd.h lmkd_amax_desc::ref_words): a producer that is
// given the maximum its result had LAST time (the reference) also counts, per frame segment, the nonzero elements that the two-plane
// split cannot resolve fully against that reference - |x| < thr = 2^-17 reference: each of them carries an absolute error of up to
// 2^-23 thr - and Q = sum of min(|x| / thr, 1024): the fully resolved elements carry up to 2^-23 |x| each, so 2^-23 thr Q bounds their
// share.  n_small / Q is therefore the ratio of the two error masses; lmkd_h2_fence_eval flags a tensor where the under-resolved
// elements would add more than a quarter.  Integers (the wave's float sum is rounded once, then added atomically): the result does
// not depend on the order of the waves.  Slot layout: word 0 maximum, word 1 n_small (u32), words 2-3 Q (u64).
struct AmaxRef { float thr, inv_thr; };      // thr = 0: no reference, nothing is counted
__device__ __forceinline__ AmaxRef amax_ref(unsigned ref_bits) {
  AmaxRef r;
  const float m = __uint_as_float(ref_bits);
  const bool ok = ref_bits != 0u && m * 7.62939453125e-06f > 0.f && m < 3.0e38f;      // 2^-17 m a normal-range number
  r.thr = ok ? m * 7.62939453125e-06f : 0.f;
  r.inv_thr = ok ? 1.f / r.thr : 0.f;
  return r;
}
struct AmaxStat { float m, q; unsigned ns; };
__device__ __forceinline__ void amax_stat4(AmaxStat& s, const float4& v, const AmaxRef& r) {
  s.m = amax4(s.m, v);
  const float a[4] = {fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s.q += fminf(a[j] * r.inv_thr, 1024.f);
    s.ns += (a[j] < r.thr && a[j] > 0.f) ? 1u : 0u;
  }
}
__device__ __forceinline__ void amax_commit_stat(unsigned* __restrict__ word, const AmaxStat& s, bool counted) {      // whole wave active
  amax_commit(word, s.m);
  if (!counted) return;
  float q = s.q;
  unsigned ns = s.ns;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    q += __shfl_xor(q, o, 64);
    ns += (unsigned)__shfl_xor((int)ns, o, 64);
  }
  if ((threadIdx.x & 63) == 0 && (ns | (q > 0.f ? 1u : 0u))) {
    unsigned* slot = word + ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (LMKD_AMAX_SLOTS - 1)) * LMKD_AMAX_STRIDE;
    if (ns) atomicAdd(slot + 1, ns);
    atomicAdd(reinterpret_cast<unsigned long long*>(slot + 2), (unsigned long long)(q + 0.5f));
  }
}
// the fp32 bits of the maximum of frame segment `seg` (wave-uniform; call with the whole wave active)
__device__ __forceinline__ unsigned amax_read(const unsigned* __restrict__ word, int seg) {
  unsigned v = word[seg * LMKD_AMAX_SEG_WORDS + (threadIdx.x & (LMKD_AMAX_SLOTS - 1)) * LMKD_AMAX_STRIDE];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const unsigned u = (unsigned)__shfl_xor((int)v, o, 64);
    v = u > v ? u : v;
  }
  return __builtin_amdgcn_readfirstlane(v);
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

// ---- activation storage type -------------------------------------------------------------------------------------------
// BASELINE configs[2] stores every activation and activation-gradient tensor of the trunk in HBM as bf16 (the reference runs
// under autocast, trainwandb.py:20,126); statistics, accumulators and everything after the pooled head stay fp32.  The
// process-wide switch lmkd_set_activation_dtype selects which instance of the HBM-bound kernels and of the convolution loaders /
// epilogues the entry points launch; in that mode the `float*` activation arguments of the C ABI address bf16 tensors.
extern int g_lmkd_act_bf16;
typedef unsigned short lmkd_bf16_t;      // raw bf16 bits

__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float v) {      // round to nearest even (v_cvt_pk_bf16_f32)
  union { __bf16 h; unsigned short u; } c;
  c.h = (__bf16)v;
  return c.u;
}
__device__ __forceinline__ float round_bf16(float v) { return bf16_to_f32(f32_to_bf16(v)); }

// 4 consecutive elements starting at element index 4*i4
template <typename T> __device__ __forceinline__ float4 ld4(const T* __restrict__ p, long i4);
template <> __device__ __forceinline__ float4 ld4<float>(const float* __restrict__ p, long i4) { return reinterpret_cast<const float4*>(p)[i4]; }
template <> __device__ __forceinline__ float4 ld4<lmkd_bf16_t>(const lmkd_bf16_t* __restrict__ p, long i4) {
  const uint2 u = reinterpret_cast<const uint2*>(p)[i4];
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void st4(T* __restrict__ p, long i4, const float4& v);
template <> __device__ __forceinline__ void st4<float>(float* __restrict__ p, long i4, const float4& v) { reinterpret_cast<float4*>(p)[i4] = v; }
template <> __device__ __forceinline__ void st4<lmkd_bf16_t>(lmkd_bf16_t* __restrict__ p, long i4, const float4& v) {
  uint2 u;
  u.x = (unsigned)f32_to_bf16(v.x) | ((unsigned)f32_to_bf16(v.y) << 16);
  u.y = (unsigned)f32_to_bf16(v.z) | ((unsigned)f32_to_bf16(v.w) << 16);
  reinterpret_cast<uint2*>(p)[i4] = u;
}
template <typename T> __device__ __forceinline__ float ld1(const T* __restrict__ p, long i);
template <> __device__ __forceinline__ float ld1<float>(const float* __restrict__ p, long i) { return p[i]; }
template <> __device__ __forceinline__ float ld1<lmkd_bf16_t>(const lmkd_bf16_t* __restrict__ p, long i) { return bf16_to_f32(p[i]); }
template <typename T> __device__ __forceinline__ void st1(T* __restrict__ p, long i, float v);
template <> __device__ __forceinline__ void st1<float>(float* __restrict__ p, long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st1<lmkd_bf16_t>(lmkd_bf16_t* __restrict__ p, long i, float v) { p[i] = f32_to_bf16(v); }

// U consecutive groups of 4 elements per thread: U = 1 for fp32 (one 16-byte access), U = 2 for bf16 (8 elements = one 16-byte access)
template <typename T> struct ActU { static constexpr int U = 1; };
template <> struct ActU<lmkd_bf16_t> { static constexpr int U = 2; };
template <typename T, int U> __device__ __forceinline__ void ldv(const T* __restrict__ p, long iu, float4 (&v)[U]);
template <> __device__ __forceinline__ void ldv<float, 1>(const float* __restrict__ p, long iu, float4 (&v)[1]) { v[0] = reinterpret_cast<const float4*>(p)[iu]; }
template <> __device__ __forceinline__ void ldv<lmkd_bf16_t, 2>(const lmkd_bf16_t* __restrict__ p, long iu, float4 (&v)[2]) {
  const uint4 u = reinterpret_cast<const uint4*>(p)[iu];
  v[0] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
  v[1] = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u));
}
template <typename T, int U> __device__ __forceinline__ void stv(T* __restrict__ p, long iu, const float4 (&v)[U]);
template <> __device__ __forceinline__ void stv<float, 1>(float* __restrict__ p, long iu, const float4 (&v)[1]) { reinterpret_cast<float4*>(p)[iu] = v[0]; }
template <> __device__ __forceinline__ void stv<lmkd_bf16_t, 2>(lmkd_bf16_t* __restrict__ p, long iu, const float4 (&v)[2]) {
  uint4 u;
  u.x = (unsigned)f32_to_bf16(v[0].x) | ((unsigned)f32_to_bf16(v[0].y) << 16);
  u.y = (unsigned)f32_to_bf16(v[0].z) | ((unsigned)f32_to_bf16(v[0].w) << 16);
  u.z = (unsigned)f32_to_bf16(v[1].x) | ((unsigned)f32_to_bf16(v[1].y) << 16);
  u.w = (unsigned)f32_to_bf16(v[1].z) | ((unsigned)f32_to_bf16(v[1].w) << 16);
  reinterpret_cast<uint4*>(p)[iu] = u;
}

// launch the fp32 or the bf16-activation instance of a kernel template `K<T>` (first template argument = storage type)
#define LMKD_ACT_DISPATCH(K, grid, block, shmem, stream, ...)                                                     \
  do {                                                                                                             \
    if (g_lmkd_act_bf16) hipLaunchKernelGGL((K<lmkd_bf16_t>), grid, block, shmem, stream, __VA_ARGS__);            \
    else hipLaunchKernelGGL((K<float>), grid, block, shmem, stream, __VA_ARGS__);                                  \
  } while (0)
