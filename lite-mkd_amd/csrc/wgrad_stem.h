// Weight gradient of the stem (7x7 / stride 2 / pad 3 on the channel-padded NHWC4 input, Cout = 64) in the three-plane arithmetic
// (round 3; before: conv_wgrad_x3_kernel's generic im2col gather, 680 us per 200-frame call = 69 TFLOP/s, VALU-bound: 64 % VALU issue).
//
//   dW[co][(kh, kw, ci)] = sum over output pixels (n, oh, ow) of dy[n][oh][ow][co] * x[n][2 oh - 3 + kh][2 ow - 3 + kw][ci]
//
// As a GEMM: rows = 64 output channels, columns = (kh, kw, ci) with kw padded to 8 (224 columns, the slab layout of the other weight-
// gradient kernels), reduction index k = output pixel.  Both operands are K-outer in memory and are read from LDS with the
// transposing read ds_read_b64_tr_b16 (wgrad_x3.h) - but no im2col copy of x is ever built:
//   * a UNIT is two output rows of one image (2 Wo pixels, contiguous in dy).  Its nine input rows 4 u - 3 .. 4 u + 5 sit in LDS as
//     flat bf16-plane images R_j[(w + 3) * 4 + ci] with zero margins.  For kernel row kh and output row r of the unit the B operand is
//     B[k = ow][n = kw * 4 + ci] = R_{kh + 2 r}[8 ow + n]: row k of the operand is the 32 elements that start 16 bytes after row
//     k - 1's - the transposing read takes a row ADDRESS per lane, so overlapping rows cost nothing.
//   * dy enters in steps of 32 pixels x 64 channels, split into planes when stored (tr_store4), double buffered, one barrier per step.
//   * 16x16x32 MFMAs (conv_patch16.h: the chip holds a higher clock under this shape).  Eight waves = 4 channel blocks x 2 column
//     blocks (kw 0-3 / kw 4-7); a wave keeps one 16 x 16 accumulator per kernel row kh (7 tiles, 28 registers).
//     k order inside a 32-pixel step: lane group q4 = lane / 16, element j  <->  k = 16 (j / 4) + 4 q4 + (j % 4), so that the eight
//     rows a 32-lane half reads at once are CONSECUTIVE pixels: with the 160-byte dy rows (32 bytes x odd) and the 16-byte x rows they
//     fall on distinct banks.
//   * accumulators are flushed to a slab every G units (about 2000 pixels: the rounding error of the MFMA accumulation stays relative
//     to a short partial sum - see wgrad_plan - and the slabs are added by wgrad_reduce_kernel); a workgroup walks slabs
//     blockIdx.x, blockIdx.x + gridDim.x, ... with the next unit's input rows prefetched under the current unit's MFMAs.
#pragma once

struct StemWgradArgs {
  const float* dy;      // [N, Ho, Wo, 64]
  const float* x;       // [N, H, W, 4]
  float* slab;          // [slabs][64][224]
  int N, H, W, Ho, Wo;
  int upi;              // units (output-row pairs) per image
  int units, G, slabs;  // G units per slab
  int RL;               // bytes of one row image of one plane
  const unsigned* h2_xw;      // two-plane fp16 form (NPROD == 3): the maxima of x / dy (lmkd_conv_operand_amax).  ONE scale per operand - the
  const unsigned* h2_dyw;     // larger of the two frame segments' maxima (an upper bound is a valid scale): a slab may sum units of both
};

#define STEM_WG_LDA 160
#define STEM_WG_APLANE (32 * STEM_WG_LDA)

template <int V> struct StemSlot { static constexpr int value = V; };

// D: how many 32-pixel steps ahead the dy tiles are fetched into registers (one workgroup per CU: nothing else hides the HBM latency;
// a step is ~0.7 us of MFMAs).  D <= steps per unit, so the fetch position is never more than one unit ahead.
// RLC: the row-image size as a compile-time constant (0: a.RL) - with it the (kernel row, plane) offsets of the B reads are immediates
template <int NPROD, int D, int RLC>
__global__ __launch_bounds__(512) void stem_wgrad_kernel(StemWgradArgs a) {
  static_assert(NPROD == 6 || NPROD == 9 || NPROD == 3, "three bf16 planes (6 / 9 products) or two fp16 planes (3 products: conv_patch16.h)");
  constexpr int NPL = NPROD == 3 ? 2 : 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int RL = RLC ? RLC : a.RL;
  const int XBUF = 10 * NPL * RL;                     // nine row images (+ one image of slack for the reads past a unit), three planes each
  unsigned char* s_x = smem;                          // [2][10][3][RL]
  unsigned char* s_dy = smem + 2 * XBUF;              // [2][3][32][LDA]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = wave & 3, nb = wave >> 2;
  const int q4 = lane >> 4, idx = lane & 15;
  const int W = a.W, Wo = a.Wo;

  // zero everything once: the margins of the row images are never written again
  for (int i = tid; i < (2 * XBUF) / 16; i += 512) reinterpret_cast<u32x4*>(s_x)[i] = u32x4{0u, 0u, 0u, 0u};

  const __amdgpu_buffer_rsrc_t rs_dy = x3_rsrc(a.dy, (long)a.N * a.Ho * Wo * 64 * 4);
  const __amdgpu_buffer_rsrc_t rs_x = x3_rsrc(a.x, (long)a.N * a.H * W * 4 * 4);

  float h2_sx = 1.f, h2_sdy = 1.f;
  if constexpr (NPROD == 3) {
    const unsigned mx0 = amax_read(a.h2_xw, 0), mx1 = amax_read(a.h2_xw, 1), md0 = amax_read(a.h2_dyw, 0), md1 = amax_read(a.h2_dyw, 1);
    h2_sx = h2_scale(mx0 > mx1 ? mx0 : mx1);
    h2_sdy = h2_scale(md0 > md1 ? md0 : md1);
  }
  const int d_k = tid >> 4, d_c = (tid & 15) * 4;      // dy loader: pixel of the step, 4 channels
  u32x4 rq[D];
  u32x4 rx[9];
  auto as_f4 = [](const u32x4& r) { return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)); };
  struct Pos { int u, ks, n, oh0, limit; };      // a step: unit u (-1: past the end), step ks of it, the unit's geometry
  const int ksteps = (2 * Wo + 31) >> 5;         // >= D (launcher)
  auto set_unit = [&](Pos& p, int u) {
    p.u = u;
    p.ks = 0;
    if (u < 0) return;
    p.n = u / a.upi;
    p.oh0 = 2 * (u - p.n * a.upi);
    p.limit = (a.Ho - p.oh0 >= 2 ? 2 : 1) * Wo;
  };
  // this workgroup's units in order: slabs blockIdx.x, blockIdx.x + gridDim.x, ...; -1 after the last
  auto next_unit = [&](int u) {
    const int s = u / a.G;
    if (u + 1 < a.units && (u + 1) / a.G == s) return u + 1;
    const long ns = (long)s + gridDim.x;
    return ns < a.slabs ? (int)(ns * a.G) : -1;
  };
  auto load_dy = [&](u32x4& r, const Pos& p) {
    const int k = 32 * p.ks + d_k;
    r = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, k < p.limit ? (unsigned)((((p.n * a.Ho + p.oh0) * Wo + k) * 64 + d_c) * 4) : X3_OOB, 0, 0);
  };
  auto load_x = [&](const Pos& p) {      // lane = pixel w of each of the nine input rows (threads >= W idle)
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int ih = 2 * p.oh0 - 3 + j;
      rx[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (tid < W && (unsigned)ih < (unsigned)a.H) ? (unsigned)((((p.n * a.H + ih) * W + tid) * 4) * 4) : X3_OOB, 0, 0);
    }
  };
  auto store_x = [&](int buf) {
    if (tid >= W) return;
    unsigned char* d = s_x + buf * XBUF + (tid + 3) * 8;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      uint2 p0, p1, p2;
      if constexpr (NPROD == 3) h2_split4(as_f4(rx[j]), h2_sx, p0, p1); else x3_split4(as_f4(rx[j]), p0, p1, p2);
      *reinterpret_cast<uint2*>(d + (j * NPL + 0) * RL) = p0;
      *reinterpret_cast<uint2*>(d + (j * NPL + 1) * RL) = p1;
      if constexpr (NPL == 3) *reinterpret_cast<uint2*>(d + (j * NPL + 2) * RL) = p2;
    }
  };

  // fragment addresses
  const int a_off = (4 * q4 + (idx >> 2)) * STEM_WG_LDA + (16 * cb + 4 * (idx & 3)) * 2;      // + 16 * LDA for the upper half of the step
  const int b_col = 32 * nb + 8 * (idx & 3);                                                  // bytes inside a B row
  f32x4 acc[7];
#pragma unroll
  for (int kh = 0; kh < 7; ++kh) acc[kh] = f32x4{0.f, 0.f, 0.f, 0.f};
  if ((int)blockIdx.x >= a.slabs) return;
  Pos cp, pf;
  set_unit(cp, (int)blockIdx.x * a.G);
  pf = cp;
  bool x_pending = false;      // the input rows of the unit after cp's wait in rx[]
  auto advance_pf = [&]() {
    if (++pf.ks < ksteps) return;
    set_unit(pf, next_unit(pf.u));
    if (pf.u >= 0) {
      load_x(pf);
      x_pending = true;
    }
  };
  __syncthreads();      // the zero fill
  load_x(cp);
  store_x(0);
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (pf.u >= 0) {
      load_dy(rq[d], pf);
      advance_pf();
    }
  }
  int xb = 0;           // x buffer of cp's unit
  int gstep = 0;        // running step count: the dy buffer alternates across unit boundaries too
  auto step = [&](auto slot) {
    constexpr int S = decltype(slot)::value;
    if constexpr (NPROD == 3) tr_store4_h2<STEM_WG_LDA, STEM_WG_APLANE>(s_dy + (gstep & 1) * NPL * STEM_WG_APLANE, d_k, d_c, as_f4(rq[S]), h2_sdy);
    else tr_store4<NPL, STEM_WG_LDA, STEM_WG_APLANE>(s_dy + (gstep & 1) * NPL * STEM_WG_APLANE, d_k, d_c, as_f4(rq[S]));
    if (x_pending && cp.ks == ksteps - 1) {      // fetched at least D - 1 steps ago; its buffer was last read one unit ago
      store_x(xb ^ 1);
      x_pending = false;
    }
    __syncthreads();
    if (pf.u >= 0) {
      load_dy(rq[S], pf);
      advance_pf();
    }
    // ---- this lane's two groups of four pixels (lower / upper half of the step): byte offset of their B rows for kh = 0.  Pixels past
    // the unit (k >= limit: the padding of the last step, the missing second row of an odd-height image's last unit) have all-zero dy rows,
    // so their B rows may be ANY finite numbers: they read on in the same buffer (input rows of this unit, at most 26 RL + 16 Wo + 570
    // bytes into the 30 RL of the buffer) - no address select, and the offsets of (kh, plane) are the same for every lane
    const unsigned char* dyb = s_dy + (gstep & 1) * NPL * STEM_WG_APLANE;
    unsigned badr[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int k = 32 * cp.ks + 16 * hh + 4 * q4 + (idx >> 2);
      const int r = k >= Wo ? 1 : 0;
      badr[hh] = (unsigned)(xb * XBUF + b_col + 2 * r * NPL * RL + 16 * (k - r * Wo));
    }
    bf16x8 fa[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(dyb + p * STEM_WG_APLANE + a_off));
      const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(dyb + p * STEM_WG_APLANE + a_off + 16 * STEM_WG_LDA));
      union { s16x4_t s[2]; bf16x8 b; } uu;
      uu.s[0] = lo; uu.s[1] = hi;
      fa[p] = uu.b;
    }
    // the B fragments of kernel row kh + 1 are read BEFORE the MFMAs of row kh are issued (two register sets; the sched_barrier keeps
    // hipcc from sinking the reads next to their use): the LDS latency of a row's six reads otherwise stalls the wave seven times a step
    bf16x8 fb[2][NPL];
    auto read_b = [&](int kh, bf16x8 (&f)[NPL]) {
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(s_x + badr[0] + (kh * NPL + p) * RL));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(s_x + badr[1] + (kh * NPL + p) * RL));
        union { s16x4_t s[2]; bf16x8 b; } uu;
        uu.s[0] = lo; uu.s[1] = hi;
        f[p] = uu.b;
      }
    };
    read_b(0, fb[0]);
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
      if (kh + 1 < 7) read_b(kh + 1, fb[(kh + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8(&b)[NPL] = fb[kh & 1];
      f32x4 c = acc[kh];
      if constexpr (NPROD == 3) {      // two fp16 planes: smallest terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[1]), __builtin_bit_cast(f16x8, b[0]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[0]), __builtin_bit_cast(f16x8, b[1]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[0]), __builtin_bit_cast(f16x8, b[0]), c, 0, 0, 0);
      } else {
        if (NPROD == 9) {
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[NPL - 1], b[NPL - 1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], b[NPL - 1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[NPL - 1], b[1], c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], b[1], c, 0, 0, 0);      // smallest terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], b[NPL - 1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[NPL - 1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], b[0], c, 0, 0, 0);
      }
      acc[kh] = c;
      __builtin_amdgcn_sched_barrier(0);
    }
    ++gstep;
    if (++cp.ks < ksteps) return;
    // ---- end of cp's unit
    const int slab = cp.u / a.G;
    const int un = next_unit(cp.u);
    if (un < 0 || un / a.G != slab) {      // last unit of the slab: rows = channels 16 cb + 4 q4 + e, column = kh * 32 + 16 nb + idx
      float* C = a.slab + (long)slab * 64 * 224;
#pragma unroll
      for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          C[(16 * cb + 4 * q4 + e) * 224 + kh * 32 + 16 * nb + idx] = NPROD == 3 ? acc[kh][e] * (1.f / h2_sx) * (1.f / h2_sdy) : acc[kh][e];
        acc[kh] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    set_unit(cp, un);
    xb ^= 1;
  };
  while (true) {
    step(StemSlot<0>());
    if (cp.u < 0) break;
    if constexpr (D > 1) {
      step(StemSlot<1>());
      if (cp.u < 0) break;
    }
    if constexpr (D > 2) {
      step(StemSlot<2>());
      if (cp.u < 0) break;
    }
    if constexpr (D > 3) {
      step(StemSlot<3>());
      if (cp.u < 0) break;
    }
    if constexpr (D > 4) {
      step(StemSlot<4>());
      if (cp.u < 0) break;
    }
    if constexpr (D > 5) {
      step(StemSlot<5>());
      if (cp.u < 0) break;
    }
  }
}
