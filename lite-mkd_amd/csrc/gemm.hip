// Generic strided-batched fp32 GEMM on the MFMA tile engine:
//   C[b] = alpha * opA(A[b]) * opB(B[b]) + beta * C[b] + bias[col]
// A layouts: 'K' = A[m*lda + k] (K contiguous), 'M' = A[k*lda + m] (K-outer)
// B layouts: 'K' = B[n*ldb + k] (K contiguous), 'N' = B[k*ldb + n] (K-outer)
// Used for the fc heads, the TRX projections/attention GEMMs and the MFM encoder.
#include "gemm_core.h"

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  long lda, ldb, ldc;
  long sA, sB, sC;
  int M, N, K;
  float alpha, beta;
  int relu;
};

template <class Cfg, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(LMKD_THREADS) void gemm_kernel(GemmArgs g) {
  using LA = typename std::conditional<A_KMAJOR, LoaderKMajorDense<Cfg::BM>, LoaderMMajorDense<Cfg::BM>>::type;
  using LB = typename std::conditional<B_KMAJOR, LoaderKMajorDense<Cfg::BN>, LoaderMMajorDense<Cfg::BN>>::type;
  __shared__ __attribute__((aligned(16))) float smem[2 * (LA::LDS_FLOATS + LB::LDS_FLOATS)];
  const int m0 = blockIdx.x * Cfg::BM, n0 = blockIdx.y * Cfg::BN;
  const long bz = blockIdx.z;
  LA la;
  LB lb;
  la.init(g.A + bz * g.sA, g.lda, m0, g.M, g.K);
  lb.init(g.B + bz * g.sB, g.ldb, n0, g.N, g.K);
  f32x16 acc[Cfg::TM][Cfg::TN];
  const int nk = (g.K + LMKD_BK - 1) / LMKD_BK;
  auto koff = [](int t) { return t * LMKD_BK; };
  gemm_mainloop<Cfg>(la, lb, nk, koff, koff, smem, acc);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
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
          float v = g.alpha * acc[i][j][e] + bv;
          if (g.beta != 0.f) v += g.beta * *c;
          if (g.relu) v = fmaxf(v, 0.f);
          *c = v;
        }
      }
    }
  }
}

template <class Cfg>
static int launch_gemm(const GemmArgs& g, int batch, int ak, int bk, hipStream_t s) {
  dim3 grid(cdiv(g.M, Cfg::BM), cdiv(g.N, Cfg::BN), batch);
  if (ak && bk) hipLaunchKernelGGL((gemm_kernel<Cfg, true, true>), grid, dim3(LMKD_THREADS), 0, s, g);
  else if (ak && !bk) hipLaunchKernelGGL((gemm_kernel<Cfg, true, false>), grid, dim3(LMKD_THREADS), 0, s, g);
  else if (!ak && bk) hipLaunchKernelGGL((gemm_kernel<Cfg, false, true>), grid, dim3(LMKD_THREADS), 0, s, g);
  else hipLaunchKernelGGL((gemm_kernel<Cfg, false, false>), grid, dim3(LMKD_THREADS), 0, s, g);
  LMKD_CHECK_LAUNCH("lmkd_gemm_f32");
  return LMKD_OK;
}

extern "C" int lmkd_gemm_f32(char layA, char layB, int M, int N, int K, float alpha, const float* A, long lda,
                             long sA, const float* B, long ldb, long sB, float beta, float* C, long ldc, long sC,
                             const float* bias, int relu, int batch, void* stream) {
  LMKD_REQUIRE(layA == 'K' || layA == 'M', "lmkd_gemm_f32: layA must be 'K' or 'M'");
  LMKD_REQUIRE(layB == 'K' || layB == 'N', "lmkd_gemm_f32: layB must be 'K' or 'N'");
  LMKD_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "lmkd_gemm_f32: empty problem M=%d N=%d K=%d batch=%d", M, N, K, batch);
  LMKD_REQUIRE(A && B && C, "lmkd_gemm_f32: null operand");
  LMKD_REQUIRE(aligned16(A) && aligned16(B), "lmkd_gemm_f32: operands must be 16-byte aligned");
  // float4 loads along the contiguous dim of each operand
  LMKD_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && sA % 4 == 0 && sB % 4 == 0, "lmkd_gemm_f32: lda/ldb/strides must be multiples of 4");
  if (layA == 'K') LMKD_REQUIRE(K % 4 == 0, "lmkd_gemm_f32: K %% 4 != 0 with K-major A");
  else LMKD_REQUIRE(M % 4 == 0, "lmkd_gemm_f32: M %% 4 != 0 with K-outer A");
  if (layB == 'K') LMKD_REQUIRE(K % 4 == 0, "lmkd_gemm_f32: K %% 4 != 0 with K-major B");
  else LMKD_REQUIRE(N % 4 == 0, "lmkd_gemm_f32: N %% 4 != 0 with K-outer B");
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = sA; g.sB = sB; g.sC = sC;
  g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta; g.relu = relu;
  hipStream_t s = (hipStream_t)stream;
  const int ak = layA == 'K', bk = layB == 'K';
  // tile choice: big tiles only when they still give >= ~1.5 waves of workgroups over 256 CUs
  const long t128 = (long)cdiv(M, 128) * cdiv(N, 128) * batch;
  if (t128 >= 384) return launch_gemm<TileCfg<128, 128, 2, 2>>(g, batch, ak, bk, s);
  return launch_gemm<TileCfg<64, 64, 2, 2>>(g, batch, ak, bk, s);
}
