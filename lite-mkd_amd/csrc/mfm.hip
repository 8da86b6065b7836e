// MFM teacher-feature fusion (teacher/code/model.py:1135-1151,1300-1331,1361-1392,1648-1664), inference only:
// the pieces of nn.TransformerEncoderLayer (post-norm, ReLU FFN) that are not GEMMs.  The GEMMs (in_proj, out_proj,
// linear1+ReLU, linear2, f1) run on lmkd_gemm_f32.
#include "common.h"

#define MF_THREADS 256

// y[r, col_off : col_off+D] (row stride ldy) = LayerNorm(x[r,:] (+ res[r % res_rows, :])) * gamma + beta
__global__ void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res, long res_rows,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
                                     long ldy, int D, float eps) {
  __shared__ float red[4];
  extern __shared__ float row[];
  const long r = blockIdx.x;
  const float* xr = x + r * D;
  const float* rr = res ? res + (r % res_rows) * D : nullptr;
  float s = 0.f;
  for (int d = threadIdx.x; d < D; d += MF_THREADS) {
    const float v = xr[d] + (rr ? rr[d] : 0.f);
    row[d] = v;
    s += v;
  }
  const float mean = block_sum_256(s, red) / (float)D;
  float q = 0.f;
  for (int d = threadIdx.x; d < D; d += MF_THREADS) { const float c = row[d] - mean; q += c * c; }
  const float rstd = 1.f / sqrtf(block_sum_256(q, red) / (float)D + eps);
  float* yr = y + r * ldy;
  for (int d = threadIdx.x; d < D; d += MF_THREADS) yr[d] = (row[d] - mean) * rstd * gamma[d] + beta[d];
}

extern "C" int lmkd_layernorm_fwd(const float* x, const float* res, long res_rows, const float* gamma, const float* beta, float* y,
                                  long ldy, long rows, int D, float eps, void* stream) {
  LMKD_REQUIRE(x && gamma && beta && y && rows > 0 && D > 0 && ldy >= D, "lmkd_layernorm_fwd: bad arguments");
  LMKD_REQUIRE(!res || res_rows > 0, "lmkd_layernorm_fwd: res_rows must be positive");
  LMKD_REQUIRE((size_t)D * sizeof(float) <= 96 * 1024, "lmkd_layernorm_fwd: row of %d floats does not fit in LDS", D);
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)rows), dim3(MF_THREADS), (size_t)D * sizeof(float), (hipStream_t)stream, x,
                     res, res_rows, gamma, beta, y, ldy, D, eps);
  LMKD_CHECK_LAUNCH("layernorm_fwd_kernel");
  return LMKD_OK;
}

// Self-attention over L <= 8 tokens per (sequence, head): qkv [B*L, 3*D] (q | k | v), out [B*L, D], head dim hd = D/H.
__global__ void mha_small_kernel(const float* __restrict__ qkv, float* __restrict__ out, int L, int D, int H, float scale) {
  __shared__ float part[4][64];
  __shared__ float P[64];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int hd = D / H;
  const float* base = qkv + (long)b * L * 3 * D + h * hd;
  float acc[64];
#pragma unroll
  for (int e = 0; e < 64; ++e) acc[e] = 0.f;
  for (int d = threadIdx.x; d < hd; d += MF_THREADS) {
    float q[8], k[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      q[i] = i < L ? base[(long)i * 3 * D + d] : 0.f;
      k[i] = i < L ? base[(long)i * 3 * D + D + d] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i * 8 + j] = fmaf(q[i], k[j], acc[i * 8 + j]);
  }
#pragma unroll
  for (int e = 0; e < 64; ++e) {
    const float v = wave_sum(acc[e]);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][e] = v;
  }
  __syncthreads();
  if (threadIdx.x < 64) P[threadIdx.x] = (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]) * scale;
  __syncthreads();
  if (threadIdx.x < L) {
    const int i = threadIdx.x;
    float m = -INFINITY;
    for (int j = 0; j < L; ++j) m = fmaxf(m, P[i * 8 + j]);
    float z = 0.f;
    for (int j = 0; j < L; ++j) { const float e = expf(P[i * 8 + j] - m); P[i * 8 + j] = e; z += e; }
    for (int j = 0; j < L; ++j) P[i * 8 + j] /= z;
  }
  __syncthreads();
  float* ob = out + (long)b * L * D + h * hd;
  for (int d = threadIdx.x; d < hd; d += MF_THREADS) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = j < L ? base[(long)j * 3 * D + 2 * D + d] : 0.f;
    for (int i = 0; i < L; ++i) {
      float o = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) o = fmaf(P[i * 8 + j], v[j], o);
      ob[(long)i * D + d] = o;
    }
  }
}

extern "C" int lmkd_mha_small(const float* qkv, float* out, int B, int L, int D, int H, void* stream) {
  LMKD_REQUIRE(qkv && out && B > 0 && L > 0 && L <= 8 && H > 0 && D % H == 0, "lmkd_mha_small: bad arguments (L <= 8)");
  hipLaunchKernelGGL(mha_small_kernel, dim3(B * H), dim3(MF_THREADS), 0, (hipStream_t)stream, qkv, out, L, D, H,
                     1.f / sqrtf((float)(D / H)));
  LMKD_CHECK_LAUNCH("mha_small_kernel");
  return LMKD_OK;
}
