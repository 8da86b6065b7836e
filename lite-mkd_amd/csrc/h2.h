// fp32 as two fp16 planes: the scale and the split (shared by conv_patch.h, conv_patch16.h, conv_stem.h, wgrad_*.h)
#pragma once

// ---- fp32 as TWO fp16 planes (NPROD == 3; lmkd_conv_set_compute_dtype(4)) ----
// X = x 2^s = h0 + h1 + e with h0 = fp16_rne(X), h1 = fp16_rne(X - h0) (the subtraction is exact in fp32): |e| <= 2^-23 |X| - one fp32
// ulp - and e = 0 for the fp32 values whose residual X - h0 fits 11 bits (at least half of them); e has zero mean (round to nearest).
// That holds wherever fp16's grid is fine enough for h1, |X| >= 1/4 (h1 may be a subnormal fp16 number: the matrix pipe takes those as
// they are, tests/test_gpu_h2.py); below that |e| <= 2^-25 absolutely.  Three products h0 w0 + h0 w1 + h1 w0 on v_mfma_f32_16x16x32_f16
// with fp32 accumulation; the dropped h1 w1 is <= 2^-22 of the product with zero mean (the three-plane bf16 form drops <= 2^-23, but its
// planes are TRUNCATIONS: its dropped terms all have the sign of the product).  2^s is a power of two taken from the tensor's maximum
// (the producer of the tensor folds max |x| into a word, lmkd_amax_next; the weight packs carry max |w|), so the scaling is exact:
// max |X| lies in (2^14, 2^15], elements down to 2^-17 of the maximum keep the full precision, and what an element below that loses is
// < 2^-40 of the tensor's maximum - 2^-16 of the rounding of the fp32 accumulator it is added into.
// Measured against fp64 (tools/h2_error.py, profiles/r04_h2_error.txt): rel-L2 2.7e-7 .. 7.0e-7 on the four 3x3 layers, BELOW the
// three-plane form's 3.5e-7 .. 9.6e-7 and torch's fp32 convolution's 3.1e-7 .. 8.1e-7, at half the MFMA work.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// 2^s for a tensor whose max |x| has the fp32 bits `amax`: max |x| 2^s in (2^14, 2^15]; s clamped to +-126 (2^s and 2^-s stay normal
// fp32 numbers; the epilogue undoes the two operands' scales by two separate multiplications, so no product of scales is ever formed)
__device__ __forceinline__ float h2_scale(unsigned amax) {
  const int e = (int)((amax >> 23) & 0xffu) - 127 + ((amax & 0x7fffffu) ? 1 : 0);      // ceil(log2(max))
  int s = 15 - e;
  s = s < -126 ? -126 : (s > 126 ? 126 : s);
  if (amax == 0u) s = 0;
  return __uint_as_float((unsigned)(s + 127) << 23);
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void h2_split4(const float4& v, float s, uint2& p0, uint2& p1) {
  // on pairs: v_pk_mul_f32, v_cvt_pk_f16_f32, two v_cvt_f32_f16, v_pk_add_f32, v_cvt_pk_f16_f32 - three vector instructions per element
  // (the kernels that call this are bound by vector-instruction issue)
  union { f16x2_t h[2]; uint2 u; } c0, c1;
  const f32x2_t sv = {s, s};
  const f32x2_t xa = f32x2_t{v.x, v.y} * sv, xb = f32x2_t{v.z, v.w} * sv;
  c0.h[0] = __builtin_convertvector(xa, f16x2_t);
  c0.h[1] = __builtin_convertvector(xb, f16x2_t);
  c1.h[0] = __builtin_convertvector(xa - __builtin_convertvector(c0.h[0], f32x2_t), f16x2_t);
  c1.h[1] = __builtin_convertvector(xb - __builtin_convertvector(c0.h[1], f32x2_t), f16x2_t);
  p0 = c0.u;
  p1 = c1.u;
}
__device__ __forceinline__ void h2_split1(float x, unsigned short& p0, unsigned short& p1) {
  union { _Float16 h; unsigned short u; } c0, c1;
  c0.h = (_Float16)x;
  c1.h = (_Float16)(x - (float)c0.h);
  p0 = c0.u;
  p1 = c1.u;
}
