// Weight gradient of a convolution on the bf16 matrix pipe: plain bf16 (one RNE plane, BASELINE configs[2]) or fp32 as an
// exact 3-way bf16 split with 6 (or 9) products per fp32 product (conv_x3.h).
//
//   dW[co][(tap, ci)] = sum_pix dy[pix][co] * x[pix shifted by tap][ci]
//
// Both GEMM operands are K-OUTER in memory (K = pixels: dy[pix][Cout], x[pix][Cin]), while an MFMA 32x32x16 lane wants 8
// consecutive k of ONE row / column.  The fp32 kernel (conv_wgrad_kernel) reads such fragments element by element
// (ds_read_b32); a register transpose costs more VALU than the bf16 MFMAs take.  gfx950's ds_read_b64_tr_b16 does the
// transpose inside the LDS read: the operand tiles are stored as they arrive, as bf16 planes  img[plane][k][col]
// (ds_write_b64 of 4 channels of one pixel), and a 16-lane group reads a 4 (k) x 16 (col) block column-major, i.e. every lane
// receives 4 consecutive k of its own column.  Row stride = cols * 2 + 64 bytes: an odd multiple of 64 B, so the four k rows of
// a block fall on four different quarters of the 64 banks (conflict free for the 2 x 4-row blocks of a 32-lane half).
//
// The operands are converted when a tile is stored to LDS (after the MFMAs of the current K-step, like every loader here):
// one v_cvt_pk per pair (bf16) or the exact three-way split (x3); the optional BatchNorm+ReLU of the x operand
// (lmkd_conv2d_bwd_weight_pre) runs at the same place.  One 256-thread workgroup, wave tiles of up to 64x64 (TM = TN = 2:
// one tr read per MFMA and product plane), double-buffered LDS, one barrier per K-step of 32 pixels, split-K slabs and the
// slab reduction of conv_wgrad_kernel.
#pragma once

typedef short s16x4_t __attribute__((ext_vector_type(4)));

template <int BM_, int BN_, int WM_, int WN_>
struct WgCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static constexpr int THREADS = 64 * WM * WN;
  static_assert(TM >= 1 && TN >= 1 && WM * WN == 4, "4 waves");
};

template <int COLS>
struct TrImg {
  static constexpr int LDB = COLS * 2 + 64;            // bytes per k row
  static constexpr int PLANE = LMKD_BK * LDB;          // bytes per plane
  static_assert((LDB / 64) % 2 == 1, "row stride must be an odd multiple of 64 bytes");
};

// 8 consecutive k (k0 + 8h .. +7, h = lane >> 5) of column col0 + (lane & 31) of a K-outer image: two transposed reads
template <int LDB>
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* __restrict__ img, int lane_off) {
  const unsigned char* a = img + lane_off;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)a);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a + 4 * LDB));
  union { s16x4_t s[2]; bf16x8 b; } u;
  u.s[0] = lo; u.s[1] = hi;
  return u.b;
}
// byte offset of this lane's block row inside an image (before adding k-group and tile offsets):
// lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of the group's 16 columns
__device__ __forceinline__ int tr_lane_off(int ldb, int lane) {
  const int idx = lane & 15;
  return (8 * (lane >> 5) + (idx >> 2)) * ldb + (16 * ((lane >> 4) & 1) + 4 * (idx & 3)) * 2;
}

// store one float4 (4 consecutive columns of k row `k`) as NPL bf16 planes
template <int NPL, int LDB, int PLANE>
__device__ __forceinline__ void tr_store4(unsigned char* __restrict__ img, int k, int col, const float4& v) {
  unsigned char* d = img + k * LDB + col * 2;
  if (NPL == 1) {
    *reinterpret_cast<uint2*>(d) = x3_round4(v);
  } else {
    uint2 p0, p1, p2;
    x3_split4(v, p0, p1, p2);
    *reinterpret_cast<uint2*>(d) = p0;
    *reinterpret_cast<uint2*>(d + PLANE) = p1;
    *reinterpret_cast<uint2*>(d + 2 * PLANE) = p2;
  }
}

// the same as two fp16 planes of v * s (conv_patch16.h: h2_split4)
template <int LDB, int PLANE>
__device__ __forceinline__ void tr_store4_h2(unsigned char* __restrict__ img, int k, int col, const float4& v, float s) {
  unsigned char* d = img + k * LDB + col * 2;
  uint2 p0, p1;
  h2_split4(v, s, p0, p1);
  *reinterpret_cast<uint2*>(d) = p0;
  *reinterpret_cast<uint2*>(d + PLANE) = p1;
}

__device__ __forceinline__ void tr_store8(unsigned char* __restrict__ img, int ldb, int k, int col, const u32x4& v) {
  *reinterpret_cast<u32x4*>(img + k * ldb + col * 2) = v;      // 8 bf16 columns of k row `k`, already in plane format
}

// Operands stored as bf16 in HBM (lmkd_set_activation_dtype(1)): what is loaded IS the LDS plane (8 columns = 16 B per lane).
template <int ROWS, int THREADS>
struct LoaderMMajorDense16 {      // elem(k, row) = base[k*ld + row]
  static constexpr int CPR = ROWS / 8, KPP = THREADS / CPR, NI = LMKD_BK / KPP;
  static_assert(THREADS % CPR == 0 && LMKD_BK % KPP == 0 && NI >= 1, "tile rows vs threads");
  const lmkd_bf16_t* base;
  long ld;
  u32x4 reg[NI];
  int K, r4;
  bool rin;
  __device__ __forceinline__ void init(const float* base_, long ld_, int row0, int nrows, int K_) {
    r4 = (threadIdx.x % CPR) * 8;
    rin = (row0 + r4) < nrows;
    base = reinterpret_cast<const lmkd_bf16_t*>(base_) + row0 + r4;
    ld = ld_; K = K_;
  }
  __device__ __forceinline__ void load(int koff) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k = koff + threadIdx.x / CPR + KPP * i;
      reg[i] = (rin && k < K) ? *reinterpret_cast<const u32x4*>(base + (long)k * ld) : u32x4{0u, 0u, 0u, 0u};
    }
  }
};
template <int ROWS, int THREADS>
struct LoaderWgradGather16 {      // implicit im2col of a bf16 NHWC tensor, K-outer (Cs % 32 == 0: a lane's 8 columns lie in one tap)
  static constexpr int CPR = ROWS / 8, KPP = THREADS / CPR, NI = LMKD_BK / KPP;
  const lmkd_bf16_t* x;
  FastDiv div_hw, div_w;
  int Mpix, HoWo, Wo, Hs, Ws, Cs, stride;
  u32x4 reg[NI];
  int dh, dw, coff, r4;
  bool tap_ok;
  unsigned inb;             // members the kernel body shares with the fp32 loader
  float4 psc, psh;
  __device__ __forceinline__ void init(const WgradArgs& a, int j0) {
    x = reinterpret_cast<const lmkd_bf16_t*>(a.x); div_hw = a.div_hw; div_w = a.div_w;
    Mpix = a.Mpix; HoWo = a.Ho * a.Wo; Wo = a.Wo; Hs = a.Hs; Ws = a.Ws; Cs = a.Cs; stride = a.stride;
    inb = 0u;
    r4 = (threadIdx.x % CPR) * 8;
    const int col = j0 + r4;
    const int tap = col / a.Cs;
    coff = col - tap * a.Cs;
    const int kh = tap / a.KWp, kw = tap - kh * a.KWp;
    dh = kh - a.pad; dw = kw - a.pad;
    tap_ok = col < a.Kp && kw < a.KW;
  }
  __device__ __forceinline__ void load(int koff) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = koff + threadIdx.x / CPR + KPP * i;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (tap_ok && p < Mpix) {
        const int n = fdiv(p, div_hw);
        const int rem = p - n * HoWo;
        const int oh = fdiv(rem, div_w);
        const int ow = rem - oh * Wo;
        const int h = oh * stride + dh, w = ow * stride + dw;
        if ((unsigned)h < (unsigned)Hs && (unsigned)w < (unsigned)Ws)
          v = *reinterpret_cast<const u32x4*>(x + ((long)(n * Hs + h) * Ws + w) * Cs + coff);
      }
      reg[i] = v;
    }
  }
};

// SMALLC: the channel-padded (NHWC4) stem input, columns = (kh, kw, 4 channels).  ACT16: dy (and, except for the fp32 stem input,
// x) are bf16 tensors in HBM (one-plane mode only).
template <class Cfg, bool SMALLC, int NPROD, bool PRE, bool ACT16 = false>
__global__ __launch_bounds__(Cfg::THREADS) void conv_wgrad_x3_kernel(WgradArgs a) {
  // NPROD == 3: two fp16 planes per operand, three products (conv_patch16.h: h2_split4; lmkd_conv_set_compute_dtype(4))
  constexpr int NPL = NPROD == 1 ? 1 : (NPROD == 3 ? 2 : 3);
  static_assert(!ACT16 || (NPROD == 1 && !PRE), "bf16 activations: one plane, no store-side arithmetic");
  static_assert(NPROD != 3 || (!PRE && !SMALLC), "two-plane form: operands whose maxima their producers recorded");
  constexpr bool B16 = ACT16 && !SMALLC;
  using LA = typename std::conditional<ACT16, LoaderMMajorDense16<Cfg::BM, Cfg::THREADS>, LoaderMMajorDense<Cfg::BM, Cfg::THREADS>>::type;
  using LB = typename std::conditional<B16, LoaderWgradGather16<Cfg::BN, Cfg::THREADS>, LoaderWgradGather<Cfg::BN, SMALLC, Cfg::THREADS, PRE>>::type;
  using IA = TrImg<Cfg::BM>;
  using IB = TrImg<Cfg::BN>;
  constexpr int A_BYTES = NPL * IA::PLANE, B_BYTES = NPL * IB::PLANE;
  // ONE LDS buffer and two barriers per K-step: at 3 planes a double-buffered 128x128 tile (120 KB) leaves one workgroup (one
  // wave per SIMD) per CU, and nothing then runs beside the conversion / split VALU work of a wave; with a single buffer two
  // workgroups share the CU and one's MFMAs overlap the other's loads, conversion and LDS stores (the next tile is still
  // prefetched into registers while the current one is multiplied)
  __shared__ __attribute__((aligned(16))) unsigned char smem[A_BYTES + B_BYTES];
  int z, tz;
  const int tiles_per_z = a.n_mt * a.n_jt;
  if (a.xcd_mode) {
    const int xj = blockIdx.x >> 3;
    z = (blockIdx.x & 7) + 8 * (xj / tiles_per_z);
    if (z >= a.splits) return;
    tz = xj - (xj / tiles_per_z) * tiles_per_z;
  } else {
    z = blockIdx.x / tiles_per_z;
    tz = blockIdx.x - z * tiles_per_z;
  }
  const int m0 = (tz % a.n_mt) * Cfg::BM, j0 = (tz / a.n_mt) * Cfg::BN;
  LA la;
  LB lb;
  // two frame segments (WgradArgs::seg_pix0): the slab's pixel range lies inside ONE segment, whose BatchNorm table the PRE loader uses
  const WinSlab sl = win_slab(z, a.Mpix, a.steps_per_split, a.seg_pix0, a.seg_splits0, a.sps1);
  const int nk = sl.nk, kb = sl.kfirst;
  la.init(a.dy, a.Co, m0, a.Co, sl.kend);
  lb.init(a, j0);
  lb.Mpix = sl.kend;
  if constexpr (PRE) {
    if (sl.seg && lb.tap_ok) {
      lb.psc = *reinterpret_cast<const float4*>(a.pre_stats + 5 * a.Cs + 2 * a.Cs + lb.coff);
      lb.psh = *reinterpret_cast<const float4*>(a.pre_stats + 5 * a.Cs + 3 * a.Cs + lb.coff);
    }
  }
  // (The directional truncation of the MFMA's mixed-magnitude additions - conv_x3.h, X3FragB::init - is left alone here: a slab is a
  // sum over ~2.5 K pixels, the bias of a weight-gradient element is ~1e-6 of the tensor's rms and nothing sums weight gradients
  // coherently over 10^5 terms afterwards.  Alternating the accumulation sign per pixel split was measured: it removes the bias, but
  // the negation of dy at store time cost the rolling-window kernel 25 % - its register allocation - and was not kept.)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float h2_sx = 1.f, h2_sdy = 1.f;
  if constexpr (NPROD == 3) {
    h2_sx = h2_scale(amax_read(a.h2_xw, sl.seg));      // the maxima over this slab's frame segment
    h2_sdy = h2_scale(amax_read(a.h2_dyw, sl.seg));
  }
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int offA = tr_lane_off(IA::LDB, lane) + wm * (Cfg::TM * 32) * 2;
  const int offB = tr_lane_off(IB::LDB, lane) + wn * (Cfg::TN * 32) * 2;
  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto store_tiles = [&](unsigned char* sa, unsigned char* sb) {
#pragma unroll
    for (int i = 0; i < LA::NI; ++i) {
      if constexpr (ACT16) tr_store8(sa, IA::LDB, tid / LA::CPR + LA::KPP * i, la.r4, la.reg[i]);
      else if constexpr (NPROD == 3) tr_store4_h2<IA::LDB, IA::PLANE>(sa, tid / LA::CPR + LA::KPP * i, la.r4, la.reg[i], h2_sdy);
      else tr_store4<NPL, IA::LDB, IA::PLANE>(sa, tid / LA::CPR + LA::KPP * i, la.r4, la.reg[i]);
    }
    if constexpr (B16) {
#pragma unroll
      for (int i = 0; i < LB::NI; ++i) tr_store8(sb, IB::LDB, tid / LB::CPR + LB::KPP * i, lb.r4, lb.reg[i]);
    } else {
#pragma unroll
    for (int i = 0; i < LB::NI; ++i) {
      float4 v = lb.reg[i];
      if (PRE && ((lb.inb >> i) & 1u)) {
        v.x = fmaxf(fmaf(v.x, lb.psc.x, lb.psh.x), 0.f); v.y = fmaxf(fmaf(v.y, lb.psc.y, lb.psh.y), 0.f);
        v.z = fmaxf(fmaf(v.z, lb.psc.z, lb.psh.z), 0.f); v.w = fmaxf(fmaf(v.w, lb.psc.w, lb.psh.w), 0.f);
      }
      if constexpr (NPROD == 3) tr_store4_h2<IB::LDB, IB::PLANE>(sb, tid / LB::CPR + LB::KPP * i, lb.r4, v, h2_sx);
      else tr_store4<NPL, IB::LDB, IB::PLANE>(sb, tid / LB::CPR + LB::KPP * i, lb.r4, v);
    }
    }
  };
  auto kstep = [&](const unsigned char* sa, const unsigned char* sb) {
    bf16x8 fa[2][NPL][Cfg::TM], fb[2][NPL][Cfg::TN];
    auto read_group = [&](int g, bf16x8 (&xa)[NPL][Cfg::TM], bf16x8 (&xb)[NPL][Cfg::TN]) {
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i) xa[p][i] = tr_frag<IA::LDB>(sa, offA + p * IA::PLANE + g * 16 * IA::LDB + i * 64);
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) xb[p][j] = tr_frag<IB::LDB>(sb, offB + p * IB::PLANE + g * 16 * IB::LDB + j * 64);
      }
    };
    read_group(0, fa[0], fb[0]);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (g == 0) {
        read_group(1, fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) {
          f32x16 c = acc[i][j];
          if constexpr (NPROD == 1) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][0][i], fb[g][0][j], c, 0, 0, 0);
          } else if constexpr (NPROD == 3) {      // two fp16 planes: smallest terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[g][1][i]), __builtin_bit_cast(f16x8, fb[g][0][j]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[g][0][i]), __builtin_bit_cast(f16x8, fb[g][1][j]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[g][0][i]), __builtin_bit_cast(f16x8, fb[g][0][j]), c, 0, 0, 0);
          } else {
            if (NPROD == 9) {
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][NPL - 1][i], fb[g][NPL - 1][j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][1][i], fb[g][NPL - 1][j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][NPL - 1][i], fb[g][1][j], c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][1][i], fb[g][1][j], c, 0, 0, 0);     // smallest terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][0][i], fb[g][NPL - 1][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][NPL - 1][i], fb[g][0][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][0][i], fb[g][1][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][1][i], fb[g][0][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][0][i], fb[g][0][j], c, 0, 0, 0);
          }
          acc[i][j] = c;
        }
    }
  };

  if (nk > 0) {
    unsigned char* A0 = smem;
    unsigned char* B0 = smem + A_BYTES;
    la.load(kb);
    lb.load(kb);
    for (int t = 0; t < nk; ++t) {
      store_tiles(A0, B0);                    // waits for the prefetched registers of step t
      __syncthreads();
      if (t + 1 < nk) {
        la.load(kb + (t + 1) * LMKD_BK);
        lb.load(kb + (t + 1) * LMKD_BK);
      }
      kstep(A0, B0);
      __syncthreads();                        // every wave has read the tile before it is overwritten
    }
  }

  float* C = a.slab + (long)z * a.Co * a.Kp;
  const float h2_ix = 1.f / h2_sx, h2_idy = 1.f / h2_sdy;      // exact (powers of two), applied one after the other
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const int col = j0 + wn * (Cfg::TN * 32) + j * 32 + (lane & 31);
    if (col >= a.Kp) continue;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * (Cfg::TM * 32) + i * 32 + acc_row(e, lane);
        if (row < a.Co) C[(long)row * a.Kp + col] = NPROD == 3 ? acc[i][j][e] * h2_ix * h2_idy : acc[i][j][e];
      }
  }
}
