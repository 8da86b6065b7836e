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
  // split-K (lmkd_gemm_f32_splitk): gridDim.z = batch * splits; split ks sums k in [ks * k_per, min(K, (ks + 1) * k_per)), leaves its
  // raw tile in ws[(b * splits + ks)][M][N] and takes a ticket of its output tile; the block that arrives last adds the `splits`
  // partial tiles in split order (the result does not depend on the arrival order) and runs the epilogue.
  int splits, k_per;
  float* ws;
  unsigned* tickets;
};

template <class Cfg, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(LMKD_THREADS) void gemm_kernel(GemmArgs g) {
  using LA = typename std::conditional<A_KMAJOR, LoaderKMajorDense<Cfg::BM>, LoaderMMajorDense<Cfg::BM>>::type;
  using LB = typename std::conditional<B_KMAJOR, LoaderKMajorDense<Cfg::BN>, LoaderMMajorDense<Cfg::BN>>::type;
  __shared__ __attribute__((aligned(16))) float smem[2 * (LA::LDS_FLOATS + LB::LDS_FLOATS)];
  __shared__ int s_last;
  const int m0 = blockIdx.x * Cfg::BM, n0 = blockIdx.y * Cfg::BN;
  const int S = g.splits;
  const long bz = S > 1 ? blockIdx.z / S : blockIdx.z;
  const int ks = S > 1 ? (int)(blockIdx.z - bz * S) : 0;
  const int k0 = ks * g.k_per;
  const int k1 = S > 1 ? (k0 + g.k_per < g.K ? k0 + g.k_per : g.K) : g.K;
  LA la;
  LB lb;
  la.init(g.A + bz * g.sA, g.lda, m0, g.M, k1);
  lb.init(g.B + bz * g.sB, g.ldb, n0, g.N, k1);
  f32x16 acc[Cfg::TM][Cfg::TN];
  const int nk = (k1 - k0 + LMKD_BK - 1) / LMKD_BK;
  auto koff = [k0](int t) { return k0 + t * LMKD_BK; };
  gemm_mainloop<Cfg>(la, lb, nk, koff, koff, smem, acc);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  if (S > 1) {
    // partial tile -> device-coherent memory (relaxed agent-scope atomic stores = write-through stores; norm_pool.hip colsum_ticket has
    // the reasoning for doing this without __threadfence()), wait for the stores, then the ticket
    float* wp = g.ws + (bz * S + ks) * (long)g.M * g.N;
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const int col = n0 + wn * (Cfg::TN * 32) + j * 32 + (lane & 31);
      if (col >= g.N) continue;
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm * (Cfg::TM * 32) + i * 32 + acc_row(e, lane);
          if (row < g.M) __hip_atomic_store(wp + (long)row * g.N + col, acc[i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* tk = g.tickets + (bz * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) s_last = atomicAdd(tk, 1u) == (unsigned)(S - 1);
    __syncthreads();
    if (!s_last) return;
    asm volatile("" ::: "memory");
    const float* w0 = g.ws + bz * S * (long)g.M * g.N;
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const int col = n0 + wn * (Cfg::TN * 32) + j * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
      if (col >= g.N) continue;
      for (int q = 0; q < S; ++q) {      // split order; the 16 * TM loads of a split are independent (in flight together)
        const float* wq = w0 + (long)q * g.M * g.N;
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = m0 + wm * (Cfg::TM * 32) + i * 32 + acc_row(e, lane);
            if (row < g.M) acc[i][j][e] += __hip_atomic_load(wq + (long)row * g.N + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
      }
    }
    if (threadIdx.x == 0) *tk = 0u;      // zero again for the next launch that uses this ticket buffer
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
          float v = g.alpha * acc[i][j][e] + bv;
          if (g.beta != 0.f) v += g.beta * *c;
          if (g.relu) v = fmaxf(v, 0.f);
          *c = v;
        }
      }
    }
  }
}

#include "gemm_x3.h"

// Which arithmetic the GEMMs run in: -1 (default) follows the convolutions' (lmkd_conv_set_compute_dtype: native fp32 MFMA in mode 0, the
// exact 3-way bf16 split - six products, fp32-class error - in modes 1-3: the heads stay fp32-class also when the trunk runs in bf16),
// 0 = always the native fp32 MFMA kernel, 1 = always the three-plane kernel, 2 = ONE bf16 plane per operand (RNE), fp32 accumulation.
static int g_gemm_mode = -1;
static int g_gemm_tile = 0;      // tuning: 0 auto, 1 = 128 x 128 where it fills the chip, 2 = 64 x 128, 3 = 64 x 64
extern "C" int lmkd_gemm_set_tile(int t) { g_gemm_tile = t; return LMKD_OK; }
extern "C" int lmkd_gemm_set_mode(int mode) {
  LMKD_REQUIRE(mode >= -1 && mode <= 2, "lmkd_gemm_set_mode: -1 follow the convolutions, 0 fp32 MFMA, 1 fp32 as 3 x bf16, 2 bf16");
  g_gemm_mode = mode;
  return LMKD_OK;
}
extern "C" int lmkd_conv_get_compute_dtype(void);
static inline int gemm_planes(int M, int N, int K) {
  int m = g_gemm_mode;
  if (m < 0) m = lmkd_conv_get_compute_dtype() == 0 ? 0 : 1;
  if (m == 0 || (long)M * N * K < (1L << 18)) return 0;      // tiny products (the 5 x 4 / 25 x 5 matcher tables) stay where they are
  return m == 2 ? 1 : 3;
}

template <class Cfg>
static int launch_gemm(const GemmArgs& g, int batch, int ak, int bk, hipStream_t s) {
  dim3 grid(cdiv(g.M, Cfg::BM), cdiv(g.N, Cfg::BN), batch * (g.splits > 1 ? g.splits : 1));
  if (ak && bk) hipLaunchKernelGGL((gemm_kernel<Cfg, true, true>), grid, dim3(LMKD_THREADS), 0, s, g);
  else if (ak && !bk) hipLaunchKernelGGL((gemm_kernel<Cfg, true, false>), grid, dim3(LMKD_THREADS), 0, s, g);
  else if (!ak && bk) hipLaunchKernelGGL((gemm_kernel<Cfg, false, true>), grid, dim3(LMKD_THREADS), 0, s, g);
  else hipLaunchKernelGGL((gemm_kernel<Cfg, false, false>), grid, dim3(LMKD_THREADS), 0, s, g);
  LMKD_CHECK_LAUNCH("lmkd_gemm_f32");
  return LMKD_OK;
}

// GEMMs of this path are small (M = 200 / 400 frames or 700 tuples): 32 - 250 output tiles for 256 CUs.  With a workspace the K range
// of the launches with at most 64 output tiles (the fc layers' 200-row calls: 32 tiles, K = 2048) is split over gridDim.z; the partial
// tiles are added in split order by the last-arriving block of each tile (deterministic).  Measured (tools/gemm_splitk_bench.py):
// 200 x 512 x 8192 194 -> 103 us; with 121 - 252 tiles the split LOSES (projections 400 x 1152 x 2048 x 2: 59 -> 68 us, 700 x 700 x
// 1152: 31 -> 47 us - the write-through partial tiles and the finishing block's dependent reads cost more than the idle SIMDs).
#define LMKD_GEMM_TICKET_WORDS 4096
extern "C" long lmkd_gemm_ticket_words(void) { return LMKD_GEMM_TICKET_WORDS; }

static void gemm_split_plan(int M, int N, int K, int batch, long ws_bytes, int* splits, int* k_per) {
  *splits = 1;
  *k_per = K;
  const long tiles = (long)cdiv(M, 64) * cdiv(N, 64) * batch;
  if (tiles > 64 || K < 256) return;
  int S = (int)cdiv(768, tiles);
  if (S > 16) S = 16;
  if (S > K / 128) S = K / 128;
  const long per = (long)M * N * batch * sizeof(float);
  if ((long)S * per > ws_bytes) S = (int)(ws_bytes / per);
  if (S < 2) return;
  int kp = cdiv(cdiv(K, S), LMKD_BK) * LMKD_BK;
  *k_per = kp;
  *splits = cdiv(K, kp);
  if (*splits < 2) { *splits = 1; *k_per = K; }
}

static int gemm_impl(char layA, char layB, int M, int N, int K, float alpha, const float* A, long lda, long sA, const float* B, long ldb,
                     long sB, float beta, float* C, long ldc, long sC, const float* bias, int relu, int batch, void* workspace,
                     long ws_bytes, unsigned* tickets, void* stream) {
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
  g.splits = 1; g.k_per = K; g.ws = (float*)workspace; g.tickets = tickets;
  if (workspace && tickets) gemm_split_plan(M, N, K, batch, ws_bytes, &g.splits, &g.k_per);
  hipStream_t s = (hipStream_t)stream;
  const int ak = layA == 'K', bk = layB == 'K';
  // tile choice: big tiles only when they still give >= ~1.5 waves of workgroups over 256 CUs
  const long t128 = (long)cdiv(M, 128) * cdiv(N, 128) * batch;
  if (g.splits == 1) {
    if (const int npl = gemm_planes(M, N, K)) {      // the bf16 matrix pipe (gemm_x3.h)
      // 64 x 64 tiles throughout: measured on every head GEMM of the episode (tools/gemm_x3_bench.py, profiles/r04_gemm_x3_bench.txt) they
      // beat 64 x 128 and 128 x 128 - these launches have 100 - 600 tiles, and five small workgroups per CU hide the load latency of
      // each other where one or two big ones expose it (projection dW: 40 us against 66 with 128 x 128; prototypes 20 against 40)
      if (g_gemm_tile == 1) return npl == 3 ? launch_gemm_x3<G3Cfg<128, 128, 2, 2>, 3>(g, batch, ak, bk, s) : launch_gemm_x3<G3Cfg<128, 128, 2, 2>, 1>(g, batch, ak, bk, s);
      if (g_gemm_tile == 2) return npl == 3 ? launch_gemm_x3<G3Cfg<64, 128, 1, 4>, 3>(g, batch, ak, bk, s) : launch_gemm_x3<G3Cfg<64, 128, 1, 4>, 1>(g, batch, ak, bk, s);
      return npl == 3 ? launch_gemm_x3<G3Cfg<64, 64, 2, 2>, 3>(g, batch, ak, bk, s) : launch_gemm_x3<G3Cfg<64, 64, 2, 2>, 1>(g, batch, ak, bk, s);
    }
  }
  if (t128 >= 384) return launch_gemm<TileCfg<128, 128, 2, 2>>(g, batch, ak, bk, s);
  return launch_gemm<TileCfg<64, 64, 2, 2>>(g, batch, ak, bk, s);
}

extern "C" int lmkd_gemm_f32(char layA, char layB, int M, int N, int K, float alpha, const float* A, long lda,
                             long sA, const float* B, long ldb, long sB, float beta, float* C, long ldc, long sC,
                             const float* bias, int relu, int batch, void* stream) {
  return gemm_impl(layA, layB, M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, bias, relu, batch, nullptr, 0, nullptr, stream);
}

// the same GEMM with a split-K workspace (any size; the split count adapts to it) and a zeroed ticket buffer of lmkd_gemm_ticket_words()
// words (left zeroed), both private to the stream for as long as launches on it may be in flight
extern "C" int lmkd_gemm_f32_splitk(char layA, char layB, int M, int N, int K, float alpha, const float* A, long lda, long sA,
                                    const float* B, long ldb, long sB, float beta, float* C, long ldc, long sC, const float* bias,
                                    int relu, int batch, void* workspace, long ws_bytes, unsigned* tickets, void* stream) {
  LMKD_REQUIRE(workspace && tickets && ws_bytes > 0, "lmkd_gemm_f32_splitk: workspace / tickets missing");
  LMKD_REQUIRE(aligned16(workspace), "lmkd_gemm_f32_splitk: workspace must be 16-byte aligned");
  return gemm_impl(layA, layB, M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, bias, relu, batch, workspace, ws_bytes, tickets, stream);
}
