// fp32 MFMA tile engine for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, 64 cycles/SIMD).
//
// One 256-thread workgroup (4 waves, one per SIMD) owns a BM x BN output tile.  Both
// operand tiles live in LDS in "K-outer" form  S[k][row]  (row = M index for A, N index
// for B), double buffered, BK = 32.  A wave's MFMA fragments are then single ds_read_b32
// per operand per k-pair: lane (r = lane&31, h = lane>>5) reads S[2*kk + h][row0 + r]
// (32 consecutive floats per half wave -> conflict free).  At 64 cycles per MFMA the LDS
// and VALU pipes have ~10x slack, so the engine is MFMA-issue bound by construction; the
// loaders only have to keep HBM/L2 requests 128-B coalesced.
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
  static_assert(WM * WN == 4, "4 waves");
  static_assert(TM >= 1 && TN >= 1, "wave tile");
};

// ---------------------------------------------------------------------------------
// loaders
// ---------------------------------------------------------------------------------
// K-major dense: elem(row, k) = base[row*ld + k].  LDS tile S[k][row], LD = ROWS+1 (odd:
// the transposing ds_write_b32 of a half wave then hits 32 distinct banks).
template <int ROWS>
struct LoaderKMajorDense {
  static constexpr int NI = ROWS / 32;
  static constexpr int LD = ROWS + 1;
  static constexpr int LDS_FLOATS = LMKD_BK * LD;
  const float* p[NI];
  float4 reg[NI];
  int kc4, K;
  __device__ __forceinline__ void init(const float* base, long ld, int row0, int nrows, int K_) {
    const int tid = threadIdx.x;
    kc4 = (tid & 7) * 4;
    K = K_;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int r = row0 + (tid >> 3) + 32 * i;
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
    for (int i = 0; i < NI; ++i) {
      float* d = S + kc4 * LD + (tid >> 3) + 32 * i;
      d[0] = reg[i].x;
      d[LD] = reg[i].y;
      d[2 * LD] = reg[i].z;
      d[3 * LD] = reg[i].w;
    }
  }
};

// K-outer dense: elem(k, row) = base[k*ld + row].  LDS tile S[k][row], LD = ROWS.
template <int ROWS>
struct LoaderMMajorDense {
  static constexpr int NI = ROWS / 32;
  static constexpr int LD = ROWS;
  static constexpr int LDS_FLOATS = LMKD_BK * LD;
  static constexpr int CPR = ROWS / 4;  // float4 chunks per k row
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
      int k = koff + tid / CPR + (LMKD_THREADS / CPR) * i;
      if (rin && k < K) reg[i] = *reinterpret_cast<const float4*>(base + (long)k * ld);
      else reg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(float* S) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int k = tid / CPR + (LMKD_THREADS / CPR) * i;
      *reinterpret_cast<float4*>(S + k * LD + r4) = reg[i];
    }
  }
};

// ---------------------------------------------------------------------------------
// MFMA main loop
// ---------------------------------------------------------------------------------
// One K-step (BK/2 k-pairs) of MFMAs from the LDS tiles.  The fragments of k-pair kk+1 are read from LDS BEFORE the
// MFMAs of k-pair kk are issued (two register sets, fully unrolled), so the ~100-cycle ds_read latency hides behind the
// 4 x 64-cycle MFMAs of the previous pair; hipcc otherwise emits read -> wait lgkmcnt(0) -> 4 MFMAs per pair and the
// matrix pipe idles at every pair.
template <class Cfg, int LDA, int LDB>
__device__ __forceinline__ void mfma_kstep(const float* __restrict__ As, const float* __restrict__ Bs,
                                           int a_off, int b_off, f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  constexpr int NK = LMKD_BK / 2;
  float a[2][Cfg::TM], b[2][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i) a[0][i] = As[a_off + 32 * i];
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) b[0][j] = Bs[b_off + 32 * j];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const int cur = kk & 1, nxt = cur ^ 1;
    if (kk + 1 < NK) {
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) a[nxt][i] = As[a_off + 2 * (kk + 1) * LDA + 32 * i];
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) b[nxt][j] = Bs[b_off + 2 * (kk + 1) * LDB + 32 * j];
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this pair's MFMAs (hipcc would sink it)
    }
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
  }
}

// Runs nk K-steps.  koffA(t)/koffB(t) give each loader its K offset for step t.
template <class Cfg, class LA, class LB, class FA, class FB>
__device__ __forceinline__ void gemm_mainloop(LA& la, LB& lb, int nk, FA koffA, FB koffB, float* smem,
                                              f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  constexpr int SA = LA::LDS_FLOATS, SB = LB::LDS_FLOATS;
  float* As0 = smem;
  float* Bs0 = smem + 2 * SA;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int r = lane & 31, h = lane >> 5;
  const int a_off = h * LA::LD + wm * (Cfg::TM * 32) + r;
  const int b_off = h * LB::LD + wn * (Cfg::TN * 32) + r;
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
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    if (t + 1 < nk) {
      la.load(koffA(t + 1));
      lb.load(koffB(t + 1));
    }
    mfma_kstep<Cfg, LA::LD, LB::LD>(As0 + cur * SA, Bs0 + cur * SB, a_off, b_off, acc);
    if (t + 1 < nk) {
      la.store(As0 + (cur ^ 1) * SA);
      lb.store(Bs0 + (cur ^ 1) * SB);
    }
    __syncthreads();
  }
}

// Accumulator element e of lane -> (row, col) inside a 32x32 MFMA tile (C/D map, guide §3).
__device__ __forceinline__ int acc_row(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }
