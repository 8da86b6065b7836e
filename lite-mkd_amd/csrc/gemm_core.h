// fp32 MFMA tile engine for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, 64 cycles/SIMD).
//
// One 256-thread workgroup (4 waves, one per SIMD) owns a BM x BN output tile; both operand tiles are double
// buffered in LDS, BK = 32.  Two LDS tile forms, chosen by how the operand lies in memory:
//   RowK   S[row][36]  (K contiguous in memory: x[M,K], W[N,K], NHWC im2col rows).  Filled with one ds_write_b128 per
//          16-B global chunk; a lane (r = lane&31, h = lane>>5) reads its fragments with ONE ds_read_b128 per 8 k:
//          S[row0+r][8t+4h .. 8t+4h+3].  Row stride 36 floats = 9 x 16 B and 9 is odd, so the 16 rows of a b128 lane
//          group cover all 64 banks: conflict free.
//   KOuter S[k][rows]  (K-outer in memory: dy[pix][C], W[N,K] read as B[k=N][K]).  Filled with ds_write_b128 along the
//          row index; fragments are ds_read_b32 of 32 consecutive floats per half wave.
// An MFMA 32x32x2 consumes k = {k0, k1} on lane halves h = 0/1; any assignment of the 32 k of a K-step to (MFMA, half)
// works as long as A and B use the same one.  Here MFMA (t, j) of a K-step takes k = 8t + 4h + j, which is what the
// b128 RowK read delivers in its 4 components.
// At 64 cycles per MFMA the LDS and VALU pipes have large slack; the fragment reads of group t+1 are issued before
// the MFMAs of group t so their latency hides behind 16 x 64 MFMA cycles.
//
// Loaders (global -> registers -> LDS, software prefetch of the next K-step while the
// MFMAs of the current one run):
//   KMajorDense  rows have K contiguous  (x[M,K], W[N,K], packed conv weights)
//   MMajorDense  K-outer source          (dy[pix][C] for wgrad, W[N,K] read as B[k=N][K])
//   ConvGather   implicit im2col of an NHWC tensor, K-major (conv fwd / dgrad)
//   WgradGather  implicit im2col, K-outer (pixel-major) for wgrad
#pragma once
#include "common.h"
#include <type_traits>

#define LMKD_BK 32
#define LMKD_THREADS 256

struct Tap {
  int dh, dw, kofs;
};
#define LMKD_MAX_TAPS 9
#define LMKD_MAX_CLASSES 4

template <int BM_, int BN_, int WM_, int WN_>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static constexpr int THREADS = 64 * WM * WN;     // 4 waves (one per SIMD) or 8 waves (two per SIMD)
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
  static_assert(TM >= 1 && TN >= 1, "wave tile");
};

// ---------------------------------------------------------------------------------
// loaders
// ---------------------------------------------------------------------------------
#define LMKD_LDK 36   // RowK row stride in floats

// K-major dense: elem(row, k) = base[row*ld + k].  LDS tile RowK.
template <int ROWS, int THREADS = LMKD_THREADS>
struct LoaderKMajorDense {
  static constexpr bool ROWK = true;
  static constexpr int RPP = THREADS / 8;          // rows per pass (8 lanes x 16 B cover one 32-float row)
  static constexpr int NI = ROWS / RPP;
  static_assert(ROWS % RPP == 0, "tile rows vs threads");
  static constexpr int LD = LMKD_LDK;
  static constexpr int LDS_FLOATS = ROWS * LMKD_LDK;
  const float* p[NI];
  float4 reg[NI];
  int kc4, K;
  __device__ __forceinline__ void init(const float* base, long ld, int row0, int nrows, int K_) {
    const int tid = threadIdx.x;
    kc4 = (tid & 7) * 4;
    K = K_;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int r = row0 + (tid >> 3) + RPP * i;
      p[i] = (r < nrows) ? base + (long)r * ld + kc4 : nullptr;
    }
  }
  __device__ __forceinline__ void load(int koff) {
    const bool kin = (koff + kc4) < K;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (p[i] != nullptr && kin) reg[i] = *reinterpret_cast<const float4*>(p[i] + koff);
      else reg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(float* S) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<float4*>(S + ((tid >> 3) + RPP * i) * LMKD_LDK + kc4) = reg[i];
  }
};

// K-outer dense: elem(k, row) = base[k*ld + row].  LDS tile KOuter, LD = ROWS.
template <int ROWS, int THREADS = LMKD_THREADS>
struct LoaderMMajorDense {
  static constexpr bool ROWK = false;
  static constexpr int LD = ROWS;
  static constexpr int LDS_FLOATS = LMKD_BK * LD;
  static constexpr int CPR = ROWS / 4;  // float4 chunks per k row
  static constexpr int KPP = THREADS / CPR;         // k rows per pass
  static constexpr int NI = LMKD_BK / KPP;
  static_assert(THREADS % CPR == 0 && LMKD_BK % KPP == 0, "tile rows vs threads");
  const float* base;
  long ld;
  float4 reg[NI];
  int K, r4;
  bool rin;
  __device__ __forceinline__ void init(const float* base_, long ld_, int row0, int nrows, int K_) {
    const int tid = threadIdx.x;
    r4 = (tid % CPR) * 4;
    rin = (row0 + r4) < nrows;
    base = base_ + row0 + r4;
    ld = ld_;
    K = K_;
  }
  __device__ __forceinline__ void load(int koff) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int k = koff + tid / CPR + KPP * i;
      if (rin && k < K) reg[i] = *reinterpret_cast<const float4*>(base + (long)k * ld);
      else reg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(float* S) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int k = tid / CPR + KPP * i;
      *reinterpret_cast<float4*>(S + k * LD + r4) = reg[i];
    }
  }
};

// ---------------------------------------------------------------------------------
// MFMA main loop
// ---------------------------------------------------------------------------------
// Fragments of k-group t (8 k: MFMA j takes k = 8t + 4h + j) of one operand: N tiles of 32 rows starting at `row`.
template <bool ROWK, int LD, int N>
__device__ __forceinline__ void read_frags(const float* __restrict__ S, int row, int h, int t, float (&f)[N][4]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (ROWK) {
      const float4 q = *reinterpret_cast<const float4*>(S + (row + 32 * i) * LD + 8 * t + 4 * h);
      f[i][0] = q.x; f[i][1] = q.y; f[i][2] = q.z; f[i][3] = q.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) f[i][j] = S[(8 * t + 4 * h + j) * LD + row + 32 * i];
    }
  }
}

// One K-step (BK = 32 = 4 groups of 8 k, 4 MFMAs per output tile per group) from the LDS tiles.  The fragments of group
// t+1 are read BEFORE the MFMAs of group t are issued (two register sets, fully unrolled; the sched_barrier keeps hipcc
// from sinking the reads next to their use), so the LDS latency hides behind the MFMAs.
template <class Cfg, class LA, class LB>
__device__ __forceinline__ void mfma_kstep(const float* __restrict__ As, const float* __restrict__ Bs, int a_row, int b_row,
                                           int h, f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  constexpr int NG = LMKD_BK / 8;
  float a[2][Cfg::TM][4], b[2][Cfg::TN][4];
  read_frags<LA::ROWK, LA::LD, Cfg::TM>(As, a_row, h, 0, a[0]);
  read_frags<LB::ROWK, LB::LD, Cfg::TN>(Bs, b_row, h, 0, b[0]);
#pragma unroll
  for (int t = 0; t < NG; ++t) {
    const int cur = t & 1, nxt = cur ^ 1;
    if (t + 1 < NG) {
      read_frags<LA::ROWK, LA::LD, Cfg::TM>(As, a_row, h, t + 1, a[nxt]);
      read_frags<LB::ROWK, LB::LD, Cfg::TN>(Bs, b_row, h, t + 1, b[nxt]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int jj = 0; jj < Cfg::TN; ++jj)
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][j], b[cur][jj][j], acc[i][jj], 0, 0, 0);
  }
}

// ---- bf16 MFMA variant (BASELINE configs[2]) -------------------------------------------------------------------------
// Same loaders and the same fp32 LDS tiles; the fragments are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) when they are read
// and fed to v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  MFMA (m) of a K-step takes k = 16m + 8h + j (j = 0..7) on
// lane half h for both operands: 8 consecutive k of a RowK row (two ds_read_b128) or 8 rows of a KOuter tile (8 ds_read_b32).
// The matrix pipe is 16x faster than in fp32, so these kernels are bound by the LDS reads and the global loads instead.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool ROWK, int LD, int N>
__device__ __forceinline__ void read_frags_bf16(const float* __restrict__ S, int row, int h, int m, bf16x8 (&f)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float v[8];
    if (ROWK) {
      const float4 q0 = *reinterpret_cast<const float4*>(S + (row + 32 * i) * LD + 16 * m + 8 * h);
      const float4 q1 = *reinterpret_cast<const float4*>(S + (row + 32 * i) * LD + 16 * m + 8 * h + 4);
      v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = S[(16 * m + 8 * h + j) * LD + row + 32 * i];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) f[i][j] = (__bf16)v[j];
  }
}

template <class Cfg, class LA, class LB>
__device__ __forceinline__ void mfma_kstep_bf16(const float* __restrict__ As, const float* __restrict__ Bs, int a_row, int b_row,
                                                int h, f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  constexpr int NM = LMKD_BK / 16;
  bf16x8 a[2][Cfg::TM], b[2][Cfg::TN];
  read_frags_bf16<LA::ROWK, LA::LD, Cfg::TM>(As, a_row, h, 0, a[0]);
  read_frags_bf16<LB::ROWK, LB::LD, Cfg::TN>(Bs, b_row, h, 0, b[0]);
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int cur = m & 1, nxt = cur ^ 1;
    if (m + 1 < NM) {
      read_frags_bf16<LA::ROWK, LA::LD, Cfg::TM>(As, a_row, h, m + 1, a[nxt]);
      read_frags_bf16<LB::ROWK, LB::LD, Cfg::TN>(Bs, b_row, h, m + 1, b[nxt]);
    }
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int jj = 0; jj < Cfg::TN; ++jj)
        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][i], b[cur][jj], acc[i][jj], 0, 0, 0);
  }
}

// Runs nk K-steps.  koffA(t)/koffB(t) give each loader its K offset for step t.
template <class Cfg, class LA, class LB, class FA, class FB, bool BF16 = false, bool FENCE_STORE = false>
__device__ __forceinline__ void gemm_mainloop(LA& la, LB& lb, int nk, FA koffA, FB koffB, float* smem,
                                              f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  constexpr int SA = LA::LDS_FLOATS, SB = LB::LDS_FLOATS;
  float* As0 = smem;
  float* Bs0 = smem + 2 * SA;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int h = lane >> 5;
  const int a_row = wm * (Cfg::TM * 32) + (lane & 31);
  const int b_row = wn * (Cfg::TN * 32) + (lane & 31);
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  if (nk <= 0) return;
  la.load(koffA(0));
  lb.load(koffB(0));
  la.store(As0);
  lb.store(Bs0);
  __syncthreads();
  // last K-step peeled (see conv_gemm_kernel): no condition between a tile's prefetch and its LDS store inside the loop
  for (int t = 0; t + 1 < nk; ++t) {
    const int cur = t & 1;
    la.load(koffA(t + 1));
    lb.load(koffB(t + 1));
    if (BF16) mfma_kstep_bf16<Cfg, LA, LB>(As0 + cur * SA, Bs0 + cur * SB, a_row, b_row, h, acc);
    else mfma_kstep<Cfg, LA, LB>(As0 + cur * SA, Bs0 + cur * SB, a_row, b_row, h, acc);
    __builtin_amdgcn_sched_barrier(0);      // the LDS stores (and store-side loader arithmetic) stay behind the MFMAs
    (void)FENCE_STORE;
    la.store(As0 + (cur ^ 1) * SA);
    lb.store(Bs0 + (cur ^ 1) * SB);
    __syncthreads();
  }
  const int last = (nk - 1) & 1;
  if (BF16) mfma_kstep_bf16<Cfg, LA, LB>(As0 + last * SA, Bs0 + last * SB, a_row, b_row, h, acc);
  else mfma_kstep<Cfg, LA, LB>(As0 + last * SA, Bs0 + last * SB, a_row, b_row, h, acc);
}

// Accumulator element e of lane -> (row, col) inside a 32x32 MFMA tile (C/D map, guide §3).
__device__ __forceinline__ int acc_row(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }
