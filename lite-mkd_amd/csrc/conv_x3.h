// fp32 convolution on the bf16 matrix pipe: exact 3-way operand split ("x3").
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate.  Every fp32 number is EXACTLY the sum of three bf16 numbers
// (24 significand bits = 8 + 8 + 8, same exponent range):  x = x0 + x1 + x2,  x0 = trunc_bf16(x), x1 = trunc_bf16(x - x0),
// x2 = x - x0 - x1 (both subtractions exact in fp32).  A product a*b is then the sum of nine bf16 x bf16 products, each
// exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  Dropping the three products of order 2^-24 and below
// (a1*b2, a2*b1, a2*b2) leaves six MFMAs per fp32-equivalent MFMA of 8x the K depth: 6 x 32 cycles for what takes
// 8 x 64 cycles on the fp32 pipe, with the same class of error as one fp32 rounding per product (measured against an
// fp64 reference in tests/test_gpu_ops.py: not larger than the native fp32 MFMA kernels').  NPROD = 9 keeps all nine.
//
// Operand tiles live in LDS as three bf16 planes  S[plane][row][32 k + 8 pad]  (80-byte rows: an odd multiple of 16 B, so
// the 16 rows of each ds_read_b128 lane group cover all 64 banks).  A lane's MFMA operand (8 consecutive k of one row) is ONE
// ds_read_b128 per plane.  The activation operand is split when it is stored to LDS (4 VALU + 1.5 v_perm per element, once
// per element per workgroup); the weight operand is split once per optimizer step by lmkd_conv2d_split_weights.
// Eight waves per workgroup, two per SIMD (waves w and w+4 share one), both operand tiles double buffered in LDS, one
// barrier per K-step of 32.  Within a K-step a wave has two independent jobs: the MFMAs of step t (LDS buffer t&1) and the
// split + store of step t+1's tile (other buffer) followed by the global loads of step t+2.  Waves 0-3 run them in the
// order MFMA, store; waves 4-7 store, MFMA: on every SIMD one wave feeds the matrix pipe while its partner uses the VALU,
// LDS-write and memory pipes.
#pragma once

#define X3_LD 40   // bf16 elements per LDS row

// MINW: waves per SIMD the register allocation must leave room for (second __launch_bounds__ argument of hipcc)
template <int BM_, int BN_, int WM_, int WN_, int MINW_ = 1>
struct X3Cfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, MINW = MINW_;
  static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static constexpr int THREADS = 64 * WM * WN;
  static_assert(TM >= 1 && TN >= 1, "wave tile");
};

__device__ __forceinline__ unsigned x3_hi2(unsigned x1, unsigned x0) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }

// 4 consecutive k of one row -> the three planes' packed bf16 quads
__device__ __forceinline__ void x3_split4(const float4& v, uint2& p0, uint2& p1, uint2& p2) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  unsigned b0[4], b1[4], b2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    b0[j] = __float_as_uint(x[j]);
    const float r1 = x[j] - __uint_as_float(b0[j] & 0xffff0000u);
    b1[j] = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(b1[j] & 0xffff0000u);
    b2[j] = __float_as_uint(r2);
  }
  p0 = make_uint2(x3_hi2(b0[1], b0[0]), x3_hi2(b0[3], b0[2]));
  p1 = make_uint2(x3_hi2(b1[1], b1[0]), x3_hi2(b1[3], b1[2]));
  p2 = make_uint2(x3_hi2(b2[1], b2[0]), x3_hi2(b2[3], b2[2]));
}

// plain bf16 mode (one plane): round to nearest even, the arithmetic of BASELINE configs[2]
__device__ __forceinline__ uint2 x3_round4(const float4& v) {
  union { __bf16 h[4]; uint2 u; } c;
  c.h[0] = (__bf16)v.x; c.h[1] = (__bf16)v.y; c.h[2] = (__bf16)v.z; c.h[3] = (__bf16)v.w;
  return c.u;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define X3_OOB 0xfffffff0u   // byte offset past every buffer: the range check of a buffer load returns zeros

__device__ __forceinline__ __amdgpu_buffer_rsrc_t x3_rsrc(const void* base, long bytes) {
  const unsigned long a = (unsigned long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const unsigned n = __builtin_amdgcn_readfirstlane((unsigned)(bytes > 0xffffffe0L ? 0xffffffe0L : bytes));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long)hi << 32) | lo), 0, n, 0x00020000);
}

// which row tiles accumulate -y (X3FragB::init): the second half of the launch's row tiles.  Per-channel sums over the pixels of a
// launch then see the MFMA's directional truncation with both signs in equal measure; and because a workgroup's XCD walks its row
// tiles in order (xcd_decode), every XCD reads ONE copy of the weight planes at a time (alternating the sign from tile to tile
// doubled the weight working set of each 4 MiB L2)
__device__ __forceinline__ bool x3_neg_tile(int tile, int tiles_per_class) { return 2 * tile >= tiles_per_class && tiles_per_class > 1; }

// Weight operand: split once per optimizer step into MFMA FRAGMENT order (lmkd_conv2d_split_weights)
//   Wf[n-tile = col/32][k-group = k/16][plane][lane = col%32 + 32*h][8]  holds plane(W[col][16*kgroup + 8*h + j]), j = 0..7,
// so the B fragments of one K-step (2 k-groups x 3 planes) of one 32-column tile are 6 KiB contiguous and each lane's 16 bytes
// are exactly its MFMA operand: a wave loads them straight into registers (L1/L2 resident), no LDS and no barrier involved.
template <int TN, int NPL>
struct X3FragB {
  static constexpr int NR = TN * 2 * NPL;
  __amdgpu_buffer_rsrc_t rs;
  unsigned off[TN];     // byte offset of (n-tile, k-group 0, plane 0, this lane)
  // neg: fetch the NEGATED copy of the planes (three-plane modes: lmkd_conv2d_split_weights writes it behind the plain copy).
  // The MFMA's addition of products 2^-8 ... 2^-16 smaller than its accumulator truncates toward -inf (measured: a mean error of
  // -4e-11 x K x rms per element in the three-plane modes, none with one plane, none on the fp32 pipe: tools/x3_bias_probe.py) - half an
  // fp32 ulp at most, but of ONE sign, so the per-channel sums over 10^5 .. 10^6 pixels of the BatchNorm backward pick it up
  // coherently.  The workgroups of half the row tiles (x3_neg_tile) therefore accumulate -y (negated weights) and flip the sign in the
  // epilogue: the bias has either sign on half the pixels and cancels in every sum over pixels.
  __device__ __forceinline__ void init(const void* wf, int ncols, int Kp, int col0 /* of this wave */, int lane, bool neg = false) {
    const long copy_bytes = (long)ncols * Kp * 2 * NPL;
    rs = x3_rsrc(reinterpret_cast<const unsigned char*>(wf) + (neg ? copy_bytes : 0), copy_bytes);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nt = (col0 >> 5) + j;
      off[j] = (nt * 32 < ncols) ? (unsigned)(((long)nt * (Kp >> 4) * NPL * 64 + lane) * 16) : X3_OOB;
    }
  }
  // koff: first k of the K-step (multiple of 32), the same for every lane: it travels in the SCALAR offset of the buffer load (the range
  // check looks at the vector offset alone: an out-of-range column block, off = X3_OOB, still reads zeros) - no vector add / select per load
  // (conv_patch16.h: X3FragB16::load, where the counters showed these kernels bound by vector-instruction issue)
  __device__ __forceinline__ void load(int koff, u32x4 (&reg)[NR]) const {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((koff >> 4) * NPL * 1024));
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 2 * NPL; ++q)   // q = group * NPL + plane
        reg[j * 2 * NPL + q] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[j], so + (unsigned)(q * 1024), 0);
  }
};

// implicit-im2col rows (output pixels) x 32 k of one tap: per row a byte offset and one validity bit per tap, so a
// K-step's loads are  offset = valid ? base + tap shift : out of range  (no branches, no 64-bit address arithmetic)
template <int ROWS, bool SMALLC, int THREADS>
struct X3GatherA {
  typedef float4 Reg;
  static constexpr int ESZ = 4;          // bytes per element of the gathered tensor
  static constexpr int RPP = THREADS / 8;
  static constexpr int NI = ROWS / RPP;
  static_assert(ROWS % RPP == 0, "tile rows vs threads");
  __amdgpu_buffer_rsrc_t rs;
  unsigned base[NI], mask[NI];
  int kc4;
  __device__ __forceinline__ void init(const float* src, long elems, int Hs, int Ws, const int* s_src, const int* s_hw,
                                       const Tap* taps, int ntap) {
    rs = x3_rsrc(src, elems * 4);
    const int tid = threadIdx.x;
    kc4 = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = (tid >> 3) + RPP * i;
      const int hw = s_hw[row];
      base[i] = (unsigned)(s_src[row] + kc4) * 4u;   // padded stem (Cs = 4): kc4 floats = kw tap kc4/4, so the same formula
      unsigned m = 0;
      for (int tp = 0; tp < ntap; ++tp) {
        const int h = (hw >> 16) + taps[tp].dh, w = (hw & 0xffff) + taps[tp].dw + (SMALLC ? (kc4 >> 2) : 0);
        if (hw >= 0 && (unsigned)h < (unsigned)Hs && (unsigned)w < (unsigned)Ws) m |= 1u << tp;
      }
      mask[i] = m;
    }
  }
  // rel: byte offset of this K-step's tap and channel chunk relative to the row's own pixel
  __device__ __forceinline__ void load(int tp, int rel, float4 (&reg)[NI]) const {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const unsigned o = ((mask[i] >> tp) & 1u) ? base[i] + (unsigned)rel : X3_OOB;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0);
      reg[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  }
  // PRE (training forward of a convolution fed by relu(BatchNorm(raw))): the tile in `reg` holds RAW values of tap `tp`; rows
  // whose tap falls on a real pixel are normalised + rectified here, at store time (bit-identical to bn_apply_kernel); the
  // zero padding stays zero
  template <int NPL, bool PRE = false>
  __device__ __forceinline__ void store(unsigned char* S, const float4 (&reg)[NI], int tp = 0, float4 psc = float4(), float4 psh = float4()) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      unsigned char* d = S + (((tid >> 3) + RPP * i) * X3_LD + kc4) * 2;
      float4 v = reg[i];
      if (PRE && ((mask[i] >> tp) & 1u)) {
        v.x = fmaxf(fmaf(v.x, psc.x, psh.x), 0.f); v.y = fmaxf(fmaf(v.y, psc.y, psh.y), 0.f);
        v.z = fmaxf(fmaf(v.z, psc.z, psh.z), 0.f); v.w = fmaxf(fmaf(v.w, psc.w, psh.w), 0.f);
      }
      if (NPL == 1) {
        *reinterpret_cast<uint2*>(d) = x3_round4(v);
      } else {
        uint2 p0, p1, p2;
        x3_split4(v, p0, p1, p2);
        *reinterpret_cast<uint2*>(d) = p0;
        *reinterpret_cast<uint2*>(d + ROWS * X3_LD * 2) = p1;
        *reinterpret_cast<uint2*>(d + 2 * ROWS * X3_LD * 2) = p2;
      }
    }
  }
};

// The same gather for a tensor stored as bf16 (lmkd_set_activation_dtype(1), one-plane mode only): a row's 32 k of a K-step are
// 64 bytes = 4 lanes x 16 B (8 channels each), and what is loaded IS the LDS plane - no conversion, one ds_write_b128 per row.
template <int ROWS, int THREADS>
struct X3GatherA16 {
  typedef u32x4 Reg;
  static constexpr int ESZ = 2;
  static constexpr int RPP = THREADS / 4;
  static constexpr int NI = ROWS / RPP;
  static_assert(ROWS % RPP == 0 && NI >= 1, "tile rows vs threads");
  __amdgpu_buffer_rsrc_t rs;
  unsigned base[NI], mask[NI];
  int kc4;      // first channel of this lane's 8 inside the 32-channel chunk (name kept for the shared kernel body)
  __device__ __forceinline__ void init(const float* src, long elems, int Hs, int Ws, const int* s_src, const int* s_hw,
                                       const Tap* taps, int ntap) {
    rs = x3_rsrc(src, elems * 2);
    const int tid = threadIdx.x;
    kc4 = (tid & 3) * 8;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = (tid >> 2) + RPP * i;
      const int hw = s_hw[row];
      base[i] = (unsigned)(s_src[row] + kc4) * 2u;
      unsigned m = 0;
      for (int tp = 0; tp < ntap; ++tp) {
        const int h = (hw >> 16) + taps[tp].dh, w = (hw & 0xffff) + taps[tp].dw;
        if (hw >= 0 && (unsigned)h < (unsigned)Hs && (unsigned)w < (unsigned)Ws) m |= 1u << tp;
      }
      mask[i] = m;
    }
  }
  __device__ __forceinline__ void load(int tp, int rel, Reg (&reg)[NI]) const {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const unsigned o = ((mask[i] >> tp) & 1u) ? base[i] + (unsigned)rel : X3_OOB;
      reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0);
    }
  }
  template <int NPL, bool PRE = false>
  __device__ __forceinline__ void store(unsigned char* S, const Reg (&reg)[NI], int = 0, float4 = float4(), float4 = float4()) const {
    static_assert(NPL == 1 && !PRE, "bf16 activations: one plane, no store-side arithmetic");
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(S + (((tid >> 2) + RPP * i) * X3_LD + kc4) * 2) = reg[i];
  }
};

// All MFMAs of one K-step (two groups of 16 k) of a wave's TM x TN tiles: A fragments from the plane tiles in LDS (group 1
// is read before the MFMAs of group 0 are issued), B fragments from registers.
template <class Cfg, int NPL>
__device__ __forceinline__ void x3_read_group(const unsigned char* __restrict__ ap, int g, bf16x8 (&a)[NPL][Cfg::TM]) {
  constexpr int PA = Cfg::BM * X3_LD * 2;
#pragma unroll
  for (int p = 0; p < NPL; ++p)
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) a[p][i] = *reinterpret_cast<const bf16x8*>(ap + p * PA + i * (32 * X3_LD * 2) + g * 32);
}

template <int N>
__device__ __forceinline__ void x3_landed(float4 (&r)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(r[i].x), "+v"(r[i].y), "+v"(r[i].z), "+v"(r[i].w));
}
template <int N>
__device__ __forceinline__ void x3_landed(u32x4 (&r)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(r[i]));
}

__device__ __forceinline__ bf16x8 x3_as_bf16(const u32x4& v) {
  union { u32x4 u; bf16x8 b; } c;
  c.u = v;
  return c.b;
}

template <class Cfg, int NPROD>
__device__ __forceinline__ void x3_kstep(const unsigned char* __restrict__ As, int a_row, int h,
                                         const u32x4 (&rb)[Cfg::TN * 2 * (NPROD == 1 ? 1 : 3)], f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  constexpr int NPL = NPROD == 1 ? 1 : 3;
  const unsigned char* ap = As + (a_row * X3_LD + 8 * h) * 2;
  bf16x8 a[2][NPL][Cfg::TM];
  x3_read_group<Cfg, NPL>(ap, 0, a[0]);
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    if (g == 0) {
      x3_read_group<Cfg, NPL>(ap, 1, a[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      if constexpr (NPROD == 1) {
        const bf16x8 b0 = x3_as_bf16(rb[j * 2 + g]);
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][0][i], b0, acc[i][j], 0, 0, 0);
      } else {
      const bf16x8 b0 = x3_as_bf16(rb[j * 6 + g * 3 + 0]), b1 = x3_as_bf16(rb[j * 6 + g * 3 + 1]), b2 = x3_as_bf16(rb[j * 6 + g * 3 + 2]);
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) {
        f32x16 c = acc[i][j];
        if (NPROD == 9) {
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][2][i], b2, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][1][i], b2, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][2][i], b1, c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][1][i], b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][0][i], b2, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][2][i], b0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][0][i], b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][1][i], b0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][0][i], b0, c, 0, 0, 0);
        acc[i][j] = c;
      }
      }
    }
  }
}

// Shared epilogue of the bf16-plane convolution kernels: optional out += acc, store (fp32 or bf16 tensor), BatchNorm partial sums.
// s_out[r] = element offset of tile row r's output pixel (channel 0), -1 for rows outside the tensor.
// EP (inference, lmkd_conv2d_fwd_bn): out = relu?( acc * scale[c] + shift[c] (+ res) ) with scale / shift = rows 2 / 3 of the [5][C]
// BatchNorm table a.ep_stats - the operations of bn_apply_kernel in its order, so the fused forward is bit-identical to the
// two-pass form (conv_gemm_kernel's STATS == 2 epilogue does the same for the native fp32 mode).
// H2 (the two-plane instances of conv_patch_x3_kernel, compute mode 4): also fold max |out| of the tile's frame segment `seg` into
// a.amax_out, and - a.bnb_x given: the data gradient that feeds relu + train-mode BatchNorm backward, lmkd_conv2d_bwd_data_seg - leave the
// sums of that backward (sum g, sum g xhat; g = out where fma(x, scale, shift) > 0: bn_bwd_reduce_kernel's terms) in a.stat_partial
// instead of the forward's (sum, sum of squares).  A lane owns ONE channel per 32-column block here, so either pair of sums is local
// to the lane over its 16 TM rows and meets its partner lane (+ 32) in one shuffle.
template <class Cfg, bool STATS, bool OUT16, bool EP = false, bool H2 = false>
__device__ __forceinline__ void x3_epilogue(const ConvGemmArgs& a, f32x16 (&acc)[Cfg::TM][Cfg::TN], const int* s_out, float* s_red, int rt,
                                            int n0, int wm, int wn, int lane, int tid, bool neg = false, int seg = 0) {
  static_assert(!(EP && OUT16), "the BatchNorm epilogue writes fp32 tensors");
  static_assert(!H2 || (!EP && !OUT16), "two-plane instances: fp32 tensors, training / plain epilogue");
  if (neg) {      // this workgroup accumulated -y (X3FragB::init)
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = -acc[i][j][e];
  }
  const int cl0 = wn * (Cfg::TN * 32) + (lane & 31);
  lmkd_bf16_t* out16 = reinterpret_cast<lmkd_bf16_t*>(a.out);
  if constexpr (EP) {
    float esc[Cfg::TN], esh[Cfg::TN];
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const int col = n0 + cl0 + j * 32;
      esc[j] = col < a.Co ? a.ep_stats[2 * a.Co + col] : 0.f;
      esh[j] = col < a.Co ? a.ep_stats[3 * a.Co + col] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ob = s_out[wm * (Cfg::TM * 32) + i * 32 + acc_row(e, lane)];
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) {
          const int col = n0 + cl0 + j * 32;
          if (ob >= 0 && col < a.Co) {
            float v = fmaf(acc[i][j][e], esc[j], esh[j]);
            if (a.ep_res) v += a.ep_res[(long)ob + col];
            if (a.ep_relu) v = fmaxf(v, 0.f);
            a.out[(long)ob + col] = v;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  if (a.accum) {      // out += acc: all previous values first (loads in flight together), then the adds (conv_gemm_kernel)
    float prev[Cfg::TM][Cfg::TN][16];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ob = s_out[wm * (Cfg::TM * 32) + i * 32 + acc_row(e, lane)];
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) {
          const int col = n0 + cl0 + j * 32;
          prev[i][j][e] = (ob >= 0 && col < a.Co) ? (OUT16 ? bf16_to_f32(out16[(long)ob + col]) : a.out[(long)ob + col]) : 0.f;
        }
      }
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = acc[i][j][e] + prev[i][j][e];
  }
  float s1[Cfg::TN], s2[Cfg::TN];
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) s1[j] = s2[j] = 0.f;
  float amo = 0.f;
  const bool bnb = H2 && a.bnb_x != nullptr;      // (uniform)
  float bsc[Cfg::TN], bsh[Cfg::TN], bmean[Cfg::TN], bistd[Cfg::TN];
  if (H2 && bnb) {
    const float* tab = a.bnb_stats + (long)seg * 5 * a.Co;
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const int col = n0 + cl0 + j * 32;
      const bool in = col < a.Co;
      bmean[j] = in ? tab[col] : 0.f;
      bistd[j] = in ? tab[a.Co + col] : 0.f;
      bsc[j] = in ? tab[2 * a.Co + col] : 0.f;
      bsh[j] = in ? tab[3 * a.Co + col] : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rl = wm * (Cfg::TM * 32) + i * 32 + acc_row(e, lane);
      const int ob = s_out[rl];
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
        const int col = n0 + cl0 + j * 32;
        float v = acc[i][j][e];
        if constexpr (H2) {
          if (ob >= 0 && col < a.Co) {
            a.out[(long)ob + col] = v;
            amo = fmaxf(amo, fabsf(v));
            if (bnb) {
              const float xv = a.bnb_x[(long)ob + col];
              const float g = fmaf(xv, bsc[j], bsh[j]) > 0.f ? v : 0.f;
              s1[j] += g;
              s2[j] = fmaf(g, (xv - bmean[j]) * bistd[j], s2[j]);
            }
          }
          if (!bnb) { s1[j] += v; s2[j] = fmaf(v, v, s2[j]); }      // (rows outside the tensor hold exact zeros)
          continue;
        }
        if (OUT16) {      // the tensor in HBM is bf16: the BatchNorm statistics are those of the stored (rounded) values
          const lmkd_bf16_t b = f32_to_bf16(v);
          if (ob >= 0 && col < a.Co) out16[(long)ob + col] = b;
          v = bf16_to_f32(b);
        } else if (ob >= 0 && col < a.Co) {
          a.out[(long)ob + col] = v;
        }
        if (STATS) { s1[j] += v; s2[j] = fmaf(v, v, s2[j]); }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (H2 && a.amax_out) amax_commit(a.amax_out + seg * LMKD_AMAX_SEG_WORDS, amo);      // (every wave of the workgroup reaches this point)
  if (STATS) {
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
      const float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
      if (lane < 32) {
        s_red[(wm * Cfg::BN + cl0 + j * 32) * 2 + 0] = t1;
        s_red[(wm * Cfg::BN + cl0 + j * 32) * 2 + 1] = t2;
      }
    }
    __syncthreads();
    if (tid < Cfg::BN && n0 + tid < a.Co && a.stat_partial) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < Cfg::WM; ++w) {
        t1 += s_red[(w * Cfg::BN + tid) * 2 + 0];
        t2 += s_red[(w * Cfg::BN + tid) * 2 + 1];
      }
      float* p = a.stat_partial + ((long)rt * a.Co + n0 + tid) * 2;
      p[0] = t1;
      p[1] = t2;
    }
  }
}

// IO: bit 0 = the gathered tensor is stored as bf16, bit 1 = the output tensor is stored as bf16 (lmkd_set_activation_dtype(1)),
// bit 2 = inference epilogue (BatchNorm affine + residual + ReLU, x3_epilogue EP)
template <class Cfg, bool SMALLC, bool STATS, int NPROD, bool PRE = false, int IO = 0>
__global__ __launch_bounds__(Cfg::THREADS) void conv_gemm_x3_kernel(ConvGemmArgs a) {
  constexpr bool IN16 = (IO & 1) != 0, OUT16 = (IO & 2) != 0, EP = (IO & 4) != 0;
  static_assert(!(IN16 && SMALLC) && ((IO & 3) == 0 || NPROD == 1), "bf16 activations: one-plane mode; the stem input stays fp32");
  using LA = typename std::conditional<IN16, X3GatherA16<Cfg::BM, Cfg::THREADS>, X3GatherA<Cfg::BM, SMALLC, Cfg::THREADS>>::type;
  constexpr int NPL = NPROD == 1 ? 1 : 3;                 // NPROD == 1: plain bf16 (one RNE-rounded plane, one product)
  constexpr int A_BYTES = NPL * Cfg::BM * X3_LD * 2;
  using LB = X3FragB<Cfg::TN, NPL>;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_BYTES];
  __shared__ int s_src[Cfg::BM], s_hw[Cfg::BM], s_out[Cfg::BM];
  __shared__ int s_tap_rel[LMKD_MAX_TAPS], s_tap_kofs[LMKD_MAX_TAPS];
  __shared__ float s_red[STATS ? Cfg::WM * Cfg::BN * 2 : 1];
  const int tid = threadIdx.x;
  int rt, ct;
  if (!xcd_decode(blockIdx.x, a.n_rt, a.n_ct, a.xcd_mode, rt, ct)) return;
  const int tile = rt / a.nclass;
  const int cls = rt - tile * a.nclass;
  const int ph = cls >> 1, pw = cls & 1;
  if (a.accum && a.ntap[cls] == 0) return;      // out += 0 (see conv_gemm_kernel)
  // per-tap byte shift of the A rows and K offset of the weights, read back as LDS broadcasts: a K-step then needs neither a
  // scalar division nor a kernarg (SMEM) load, whose latency sat in every wave's instruction stream twice per K-step
  if (tid < a.ntap[cls]) {
    const Tap tp = a.taps[cls][tid];
    s_tap_rel[tid] = ((tp.dh * a.Ws + tp.dw) * a.Cs) * LA::ESZ;
    s_tap_kofs[tid] = tp.kofs;
  }
  const SegTile sg = seg_tile(a, tile, Cfg::BM);      // the tile's frame segment (ConvGemmArgs::seg_m0)
  const int row0 = sg.row0, n0 = ct * Cfg::BN;
  for (int r = tid; r < Cfg::BM; r += Cfg::THREADS) {
    const int m = row0 + r;
    bool ok = m < sg.mend;
    int n = 0, aa = 0, bb = 0;
    if (ok) {
      n = fdiv(m, a.div_hw);
      const int rem = m - n * a.Hr * a.Wr;
      aa = fdiv(rem, a.div_w);
      bb = rem - aa * a.Wr;
      ok = (aa * a.omul + ph) < a.Ho && (bb * a.omul + pw) < a.Wo;
    }
    if (ok) {
      s_src[r] = ((n * a.Hs + aa * a.sh) * a.Ws + bb * a.sh) * a.Cs;
      s_hw[r] = ((aa * a.sh) << 16) | (bb * a.sh);
      s_out[r] = ((n * a.Ho + aa * a.omul + ph) * a.Wo + bb * a.omul + pw) * a.Co;
    } else {
      s_src[r] = 0; s_hw[r] = -1; s_out[r] = -1;
    }
  }
  __syncthreads();
  const int nk = a.ntap[cls] * a.cps;
  const Tap* taps = a.taps[cls];
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int h = lane >> 5;
  const int a_row = wm * (Cfg::TM * 32) + (lane & 31);
  LA la;
  LB lb;
  la.init(a.src, (long)a.N * a.Hs * a.Ws * a.Cs, a.Hs, a.Ws, s_src, s_hw, taps, a.ntap[cls]);
  const bool neg = NPL == 3 && !SMALLC && x3_neg_tile(sg.ltile, sg.ltiles);      // the second half of the row tiles accumulates -y (X3FragB::init); the stem (K = 224) does not
  lb.init(a.wpk, a.Co, a.Kp, n0 + wn * (Cfg::TN * 32), lane, neg);
  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // Register prefetch.  A: two sets, the loads of K-step t+3 are issued while those of t+2 are in flight.  B: two sets of
  // fragments.  hipcc places its own s_waitcnt and, across the loop back-edge, falls back to vmcnt(0) - which would also wait
  // for loads issued a moment ago.  So every set is "landed" (x3_landed: an empty asm that consumes and redefines the
  // registers) at ONE point per K-step, right before the split/store, when every outstanding load is at least one MFMA
  // phase old; loads issued after that point never stand between a value and its use.
  typename LA::Reg ra0[LA::NI], ra1[LA::NI];
  u32x4 rb0[LB::NR], rb1[LB::NR];
  // K-steps are issued in increasing order (separately for A and B), channel chunk outer, tap inner (conv.hip)
  const int ntap_c = a.ntap[cls];
  int a_tp = 0, a_cc = 0, b_tp = 0, b_cc = 0;
  auto issue_a = [&](int, typename LA::Reg (&ra)[LA::NI]) {
    la.load(a_tp, s_tap_rel[a_tp] + (SMALLC ? 0 : a_cc * (LMKD_BK * LA::ESZ)), ra);
    if (++a_tp == ntap_c) { a_tp = 0; ++a_cc; }
  };
  auto issue_b = [&](int, u32x4 (&rb)[LB::NR]) {
    lb.load(s_tap_kofs[b_tp] + b_cc * LMKD_BK, rb);
    if (++b_tp == ntap_c) { b_tp = 0; ++b_cc; }
  };
  // PRE: (tap, chunk) of the NEXT tile to be stored, and the BatchNorm scale / shift of this lane's 4 channels for the current
  // chunk and the one after it (fetched a whole chunk = ntap K-steps ahead of their first use)
  int s_tp = 0, s_cc = 0;
  float4 psc_cur = float4(), psh_cur = float4(), psc_nxt = float4(), psh_nxt = float4();
  auto pre_fetch = [&](int cc, float4& sc, float4& sh) {
    const int c = (cc < a.cps ? cc : a.cps - 1) * LMKD_BK + la.kc4;
    sc = *reinterpret_cast<const float4*>(a.pre_stats + (long)sg.seg * 5 * a.Cs + 2 * a.Cs + c);
    sh = *reinterpret_cast<const float4*>(a.pre_stats + (long)sg.seg * 5 * a.Cs + 3 * a.Cs + c);
  };
  if (PRE) {
    pre_fetch(0, psc_cur, psh_cur);
    pre_fetch(1, psc_nxt, psh_nxt);
  }
  auto store_a = [&](unsigned char* dst, typename LA::Reg (&ra)[LA::NI]) {
    if constexpr (PRE) {
      la.template store<NPL, true>(dst, ra, s_tp, psc_cur, psh_cur);
      if (++s_tp == ntap_c) {
        s_tp = 0;
        ++s_cc;
        psc_cur = psc_nxt; psh_cur = psh_nxt;
        pre_fetch(s_cc + 1, psc_nxt, psh_nxt);
      }
    } else {
      la.template store<NPL>(dst, ra);
    }
  };
  const bool store_first = __builtin_amdgcn_readfirstlane(wave) >= Cfg::WM * Cfg::WN / 2;
  // K-step t: MFMAs on LDS buffer t&1 with the B fragments in `rb`; the A set `ra` (step t+1) goes to the other buffer and
  // is refilled with step t+3.  Waves 4-7 (store first) fetch B(t+1) into `rbn` before their MFMAs; waves 0-3 (MFMAs first)
  // land B(t+1) in `rbn` after theirs and refill `rb` with B(t+2).
  auto step = [&](int t, typename LA::Reg (&ra)[LA::NI], u32x4 (&rb)[LB::NR], u32x4 (&rbn)[LB::NR]) {
    const unsigned char* cur = smem + (t & 1) * A_BYTES;
    unsigned char* nxt = smem + ((t + 1) & 1) * A_BYTES;
    if (store_first) {
      x3_landed(ra);
      x3_landed(rb);
      if (t + 1 < nk) {
        store_a(nxt, ra);
        if (t + 3 < nk) issue_a(t + 3, ra);
        issue_b(t + 1, rbn);
      }
    }
    x3_kstep<Cfg, NPROD>(cur, a_row, h, rb, acc);
    if (!store_first && t + 1 < nk) {
      x3_landed(ra);
      x3_landed(rbn);
      store_a(nxt, ra);
      if (t + 3 < nk) issue_a(t + 3, ra);
      if (t + 2 < nk) issue_b(t + 2, rb);
    }
    __syncthreads();
  };
  if (nk > 0) {
    issue_a(0, ra0);
    issue_b(0, rb0);
    if (nk > 1) issue_a(1, ra1);
    x3_landed(ra0);
    store_a(smem, ra0);
    if (nk > 2) issue_a(2, ra0);
    if (!store_first) {
      x3_landed(rb0);
      if (nk > 1) issue_b(1, rb1);
    }
    __syncthreads();
    for (int t = 0; t < nk; t += 2) {
      step(t, ra1, rb0, rb1);
      if (t + 1 < nk) step(t + 1, ra0, rb1, rb0);
    }
  }

  x3_epilogue<Cfg, STATS, OUT16, EP>(a, acc, s_out, s_red, rt, n0, wm, wn, lane, tid, neg);
}

// fp32 K-major packed weights Wp[col][Kp] -> fragment-order bf16 planes (layout: X3FragB); npl = 1: one RNE-rounded plane;
// npl = 3: the three planes of W followed by the three planes of -W (2 x 3 x ncols x Kp bf16)
__global__ void split_weights_kernel(const float* __restrict__ wp, unsigned short* __restrict__ wf, int ncols, int Kp, int npl) {
  const long total = (long)ncols * Kp;
  const int G = Kp >> 4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int col = (int)(idx / Kp), k = (int)(idx - (long)col * Kp);
    const float x = wp[idx];
    const int lane = (col & 31) + 32 * ((k >> 3) & 1);
    const long o = ((((long)(col >> 5) * G + (k >> 4)) * npl) * 64 + lane) * 8 + (k & 7);
    if (npl == 1) {
      union { __bf16 h; unsigned short u; } c;
      c.h = (__bf16)x;
      wf[o] = c.u;
    } else {
      const unsigned b0 = __float_as_uint(x);
      const float r1 = x - __uint_as_float(b0 & 0xffff0000u);
      const unsigned b1 = __float_as_uint(r1);
      const float r2 = r1 - __uint_as_float(b1 & 0xffff0000u);
      const unsigned short p0 = (unsigned short)(b0 >> 16), p1 = (unsigned short)(b1 >> 16), p2 = (unsigned short)(__float_as_uint(r2) >> 16);
      wf[o] = p0;
      wf[o + 512] = p1;
      wf[o + 1024] = p2;
      // second copy: the planes of -W (sign bit flipped: the truncation split is symmetric), read by the odd row tiles (X3FragB::init)
      const long o2 = o + total * 3;
      wf[o2] = p0 ^ 0x8000u;
      wf[o2 + 512] = p1 ^ 0x8000u;
      wf[o2 + 1024] = p2 ^ 0x8000u;
    }
  }
}
