// The generic strided-batched GEMM of gemm.hip on the bf16 matrix pipe (round 4): fp32 operands, fp32 accumulation, every fp32
// product formed from the exact 3-way bf16 split of BOTH operands (six of the nine cross products - the arithmetic of the
// convolutions, conv_x3.h), or one RNE-rounded plane per operand (NPL = 1).
//
// Why: the heads of an episode - fc1 / fc2 (resnet18_2fc.py:56-64), the TRX k / v projections, scores and prototypes
// (TRX_2fcsup.py:97-133) and their backward - are 72 GFLOP of small-M GEMMs (M = 400 frames / 700 tuples) that ran on
// v_mfma_f32_32x32x2_f32 at 25 - 65 TFLOP/s: 1.5 ms per episode, most of it in the stretch between the trunk's forward and backward
// where nothing else is queued.  The bf16 pipe does six products in 3/8 of the time of one fp32 MFMA.
//
// Both operands are split when a tile is stored to LDS (after the MFMAs of the current K-step, like every loader here):
//   * an operand whose K is contiguous in memory ('K' layouts: x[M, K], W[N, K]) is stored as planes S[plane][row][32 k + 8 pad]
//     (80-byte rows) and a lane's fragment - 8 consecutive k of its row - is ONE ds_read_b128 per plane (conv_x3.h);
//   * a K-outer operand ('M' / 'N' layouts: dy[k][m], W^T) is stored as it arrives, img[plane][k][col] with rows of 2 * cols + 64
//     bytes, and gfx950's transposing read ds_read_b64_tr_b16 delivers the same fragment (wgrad_x3.h).
// Either way a lane (r = lane % 32, h = lane / 32) holds k = 16 g + 8 h .. + 7 of row r for k-group g: v_mfma_f32_32x32x16_bf16.
// One 256-thread workgroup, ONE LDS buffer and two barriers per K-step (two or three workgroups per CU hide each other's
// conversion work; the next tile is prefetched into registers under the current one's MFMAs), tiles 128 x 128 (wave tile 64 x 64)
// or 64 x 128 (small M: 400 rows are 7 x 64) chosen by how many tiles the launch has.
#pragma once

typedef short g3_s16x4 __attribute__((ext_vector_type(4)));
#define G3_LD 40      // bf16 elements per RowK LDS row

__device__ __forceinline__ unsigned g3_hi2(unsigned x1, unsigned x0) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }

// 4 consecutive elements -> NPL packed bf16 quads (truncation split: the residuals are exactly representable)
template <int NPL>
__device__ __forceinline__ void g3_split_store(unsigned char* d, int plane_bytes, const float4& v) {
  if (NPL == 1) {
    union { __bf16 h[4]; uint2 u; } c;
    c.h[0] = (__bf16)v.x; c.h[1] = (__bf16)v.y; c.h[2] = (__bf16)v.z; c.h[3] = (__bf16)v.w;
    *reinterpret_cast<uint2*>(d) = c.u;
    return;
  }
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
  *reinterpret_cast<uint2*>(d) = make_uint2(g3_hi2(b0[1], b0[0]), g3_hi2(b0[3], b0[2]));
  *reinterpret_cast<uint2*>(d + plane_bytes) = make_uint2(g3_hi2(b1[1], b1[0]), g3_hi2(b1[3], b1[2]));
  *reinterpret_cast<uint2*>(d + 2 * plane_bytes) = make_uint2(g3_hi2(b2[1], b2[0]), g3_hi2(b2[3], b2[2]));
}

// LDS image of one operand tile (ROWS rows / columns x 32 k), NPL planes
template <int ROWS, bool KMAJOR, int NPL>
struct G3Img {
  static constexpr int LDB = KMAJOR ? G3_LD * 2 : ROWS * 2 + 64;      // bytes per row (RowK: per tile row; KOuter: per k)
  static constexpr int PLANE = KMAJOR ? ROWS * LDB : LMKD_BK * LDB;
  static constexpr int BYTES = NPL * PLANE;
  static_assert(KMAJOR || (LDB / 64) % 2 == 1, "K-outer row stride must be an odd multiple of 64 bytes");
  // store the loader's registers (LoaderKMajorDense / LoaderMMajorDense of gemm_core.h)
  template <class L>
  static __device__ __forceinline__ void store(unsigned char* img, const L& l) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < L::NI; ++i) {
      if constexpr (KMAJOR) g3_split_store<NPL>(img + (((tid >> 3) + L::RPP * i) * G3_LD + l.kc4) * 2, PLANE, l.reg[i]);
      else g3_split_store<NPL>(img + (tid / L::CPR + L::KPP * i) * LDB + l.r4 * 2, PLANE, l.reg[i]);
    }
  }
  // byte offset of this lane's fragment of rows row0 + (lane & 31), k-group 0, plane 0
  static __device__ __forceinline__ int lane_off(int row0, int lane) {
    if (KMAJOR) return ((row0 + (lane & 31)) * G3_LD + 8 * (lane >> 5)) * 2;
    const int idx = lane & 15;      // lane 4q + p of a 16-lane group supplies k row q, columns 4p .. 4p + 3 of the group's 16 columns
    return (8 * (lane >> 5) + (idx >> 2)) * LDB + (row0 + 16 * ((lane >> 4) & 1) + 4 * (idx & 3)) * 2;
  }
  // fragment of 32-row block i, k-group g, plane p
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* img, int off, int i, int g, int p) {
    if (KMAJOR) return *reinterpret_cast<const bf16x8*>(img + off + p * PLANE + i * (32 * LDB) + g * 32);
    const unsigned char* a = img + off + p * PLANE + g * 16 * LDB + i * 64;
    const g3_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g3_s16x4 __attribute__((address_space(3)))*)a);
    const g3_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g3_s16x4 __attribute__((address_space(3)))*)(a + 4 * LDB));
    union { g3_s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo; u.s[1] = hi;
    return u.b;
  }
};

template <int BM_, int BN_, int WM_, int WN_>
struct G3Cfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves");
};

template <class Cfg, bool A_KMAJOR, bool B_KMAJOR, int NPL>
__global__ __launch_bounds__(LMKD_THREADS) void gemm_x3_kernel(GemmArgs g) {
  using LA = typename std::conditional<A_KMAJOR, LoaderKMajorDense<Cfg::BM>, LoaderMMajorDense<Cfg::BM>>::type;
  using LB = typename std::conditional<B_KMAJOR, LoaderKMajorDense<Cfg::BN>, LoaderMMajorDense<Cfg::BN>>::type;
  using IA = G3Img<Cfg::BM, A_KMAJOR, NPL>;
  using IB = G3Img<Cfg::BN, B_KMAJOR, NPL>;
  __shared__ __attribute__((aligned(16))) unsigned char smem[IA::BYTES + IB::BYTES];
  unsigned char* sa = smem;
  unsigned char* sb = smem + IA::BYTES;
  const int m0 = blockIdx.x * Cfg::BM, n0 = blockIdx.y * Cfg::BN;
  const long bz = blockIdx.z;
  LA la;
  LB lb;
  la.init(g.A + bz * g.sA, g.lda, m0, g.M, g.K);
  lb.init(g.B + bz * g.sB, g.ldb, n0, g.N, g.K);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int offA = IA::lane_off(wm * (Cfg::TM * 32), lane), offB = IB::lane_off(wn * (Cfg::TN * 32), lane);
  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // The MFMA's addition of products 2^-8 .. 2^-16 smaller than its accumulator truncates toward -inf (conv_x3.h, X3FragB::init: a mean
  // error of -4e-11 x K x rms per element in the three-plane arithmetic - half an ulp at most, but of ONE sign, and the consumers of these
  // GEMMs sum their outputs over hundreds of rows: the last BatchNorm's bias gradient sums the fc input gradient over 400 x 49 pixels).
  // So the 32-row blocks of the output alternate in sign, checkerboard-wise with the wave's column block: a negated block accumulates
  // -(A B) from sign-flipped A fragments (one v_xor per register) and flips back in the epilogue - the truncation then has either sign
  // on half the rows and half the columns and cancels in any sum over them.
  unsigned sgn[Cfg::TM];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
    sgn[i] = (NPL == 3 && (((m0 >> 5) + wm * Cfg::TM + i + (n0 >> 5) + wn * Cfg::TN) & 1)) ? 0x80008000u : 0u;
  const int nk = (g.K + LMKD_BK - 1) / LMKD_BK;
  // TWO register sets: the tile of step t + 2 is requested while step t is multiplied (a launch of these GEMMs is one or two workgroups
  // per CU - M = 400 rows - so nothing but the prefetch distance hides the ~1 us of an L2 / HBM round trip: with one set a K-step
  // took 2700 cycles for 770 cycles of MFMAs)
  LA la2 = la;
  LB lb2 = lb;
  auto step = [&](int t, LA& xa, LB& xb) {
      IA::store(sa, xa);                      // waits for the prefetched registers of step t
      IB::store(sb, xb);
      __syncthreads();
      if (t + 2 < nk) {
        xa.load((t + 2) * LMKD_BK);
        xb.load((t + 2) * LMKD_BK);
      }
      bf16x8 fa[2][NPL][Cfg::TM], fb[2][NPL][Cfg::TN];
      auto read_group = [&](int gq) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
          for (int i = 0; i < Cfg::TM; ++i) {
            union { bf16x8 b; unsigned u[4]; } c;
            c.b = IA::frag(sa, offA, i, gq, p);
            if (NPL == 3) { c.u[0] ^= sgn[i]; c.u[1] ^= sgn[i]; c.u[2] ^= sgn[i]; c.u[3] ^= sgn[i]; }
            fa[gq][p][i] = c.b;
          }
#pragma unroll
          for (int j = 0; j < Cfg::TN; ++j) fb[gq][p][j] = IB::frag(sb, offB, j, gq, p);
        }
      };
      read_group(0);
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        if (gq == 0) {
          read_group(1);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
          for (int j = 0; j < Cfg::TN; ++j) {
            f32x16 c = acc[i][j];
            if constexpr (NPL == 1) {
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[gq][0][i], fb[gq][0][j], c, 0, 0, 0);
            } else {
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[gq][1][i], fb[gq][1][j], c, 0, 0, 0);     // smallest terms first
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[gq][0][i], fb[gq][2][j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[gq][2][i], fb[gq][0][j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[gq][0][i], fb[gq][1][j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[gq][1][i], fb[gq][0][j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[gq][0][i], fb[gq][0][j], c, 0, 0, 0);
            }
            acc[i][j] = c;
          }
      }
      __syncthreads();                        // every wave has read the tile before it is overwritten
  };
  if (nk > 0) {
    la.load(0);
    lb.load(0);
    if (nk > 1) {
      la2.load(LMKD_BK);
      lb2.load(LMKD_BK);
    }
    int t = 0;
    for (; t + 1 < nk; t += 2) {
      step(t, la, lb);
      step(t + 1, la2, lb2);
    }
    if (t < nk) step(t, la, lb);
  }
  float* C = g.C + bz * g.sC;
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const int col = n0 + wn * (Cfg::TN * 32) + j * 32 + (lane & 31);
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * (Cfg::TM * 32) + i * 32 + acc_row(e, lane);
        if (row < g.M) {
          float* c = C + (long)row * g.ldc + col;
          const float a = sgn[i] ? -acc[i][j][e] : acc[i][j][e];
          float v = g.alpha * a + bv;      // the operations of gemm_kernel's epilogue, in its order
          if (g.beta != 0.f) v += g.beta * *c;
          if (g.relu) v = fmaxf(v, 0.f);
          *c = v;
        }
      }
    }
  }
}

template <class Cfg, int NPL>
static int launch_gemm_x3(const GemmArgs& g, int batch, int ak, int bk, hipStream_t s) {
  dim3 grid(cdiv(g.M, Cfg::BM), cdiv(g.N, Cfg::BN), batch);
  if (ak && bk) hipLaunchKernelGGL((gemm_x3_kernel<Cfg, true, true, NPL>), grid, dim3(LMKD_THREADS), 0, s, g);
  else if (ak && !bk) hipLaunchKernelGGL((gemm_x3_kernel<Cfg, true, false, NPL>), grid, dim3(LMKD_THREADS), 0, s, g);
  else if (!ak && bk) hipLaunchKernelGGL((gemm_x3_kernel<Cfg, false, true, NPL>), grid, dim3(LMKD_THREADS), 0, s, g);
  else hipLaunchKernelGGL((gemm_x3_kernel<Cfg, false, false, NPL>), grid, dim3(LMKD_THREADS), 0, s, g);
  LMKD_CHECK_LAUNCH("lmkd_gemm_x3");
  return LMKD_OK;
}
