// Support/query matchers: the pieces of TRX (TemporalCrossTransformer), SupportDK and the
// Euclidean matcher that are not GEMMs.  All reductions are wavefront (64-lane) shuffles
// + one LDS hop per 256-thread workgroup; results are deterministic.
//
// TRX data flow (host side in ops.py):  X -> (+PE, dropout) -> P = X * [Wk_a|Wk_b|Wv_a|Wv_b]^T
//   (per-frame projections; the reference's 4096-wide tuple Linear is the sum of two 2048-wide
//   halves) -> tuple combine + LayerNorm(K) -> scores GEMM -> per-class softmax -> P*V GEMM ->
//   squared distance.
#include "common.h"

#define TX_THREADS 256
#define LMKD_MAX_SEG 16
struct Segs {
  int n;
  int off[LMKD_MAX_SEG];   // first column (in units of columns)
  int cnt[LMKD_MAX_SEG];   // number of columns
  int col[LMKD_MAX_SEG];   // logit column of the segment (class label value)
};

static inline int tx_grid(long n_items) {
  long g = (n_items + TX_THREADS - 1) / TX_THREADS;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------------------------
// dropout mask: counter-based hash RNG, mask[i] in {0, 1/(1-p)}
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hash_u32(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (uint32_t)(x >> 16);
}
__global__ void dropout_mask_kernel(float* __restrict__ mask, long n, float p, uint64_t seed) {
  const float keep = 1.f / (1.f - p);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float u = (float)(hash_u32(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)i) >> 8) * (1.f / 16777216.f);
    mask[i] = u >= p ? keep : 0.f;
  }
}
// the same mask with the seed read from DEVICE memory at run time: a launch captured into a hipGraph draws a new mask on every replay
// once the host has written that replay's seed into *seed_dev (trainloop.GraphedEpisode)
__global__ void dropout_mask_dev_kernel(float* __restrict__ mask, long n, float p, const unsigned long long* __restrict__ seed_dev) {
  const uint64_t seed = *seed_dev;
  const float keep = 1.f / (1.f - p);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float u = (float)(hash_u32(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)i) >> 8) * (1.f / 16777216.f);
    mask[i] = u >= p ? keep : 0.f;
  }
}
extern "C" int lmkd_dropout_mask_dev(float* mask, long n, float p, const unsigned long long* seed_dev, void* stream) {
  LMKD_REQUIRE(mask && seed_dev && n > 0 && p >= 0.f && p < 1.f, "lmkd_dropout_mask_dev: bad arguments");
  hipLaunchKernelGGL(dropout_mask_dev_kernel, dim3(tx_grid(n)), dim3(TX_THREADS), 0, (hipStream_t)stream, mask, n, p, seed_dev);
  LMKD_CHECK_LAUNCH("dropout_mask_dev_kernel");
  return LMKD_OK;
}
extern "C" int lmkd_dropout_mask(float* mask, long n, float p, unsigned long long seed, void* stream) {
  LMKD_REQUIRE(mask && n > 0 && p >= 0.f && p < 1.f, "lmkd_dropout_mask: bad arguments");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(tx_grid(n)), dim3(TX_THREADS), 0, (hipStream_t)stream, mask, n, p, (uint64_t)seed);
  LMKD_CHECK_LAUNCH("dropout_mask_kernel");
  return LMKD_OK;
}

// y[r, :] = (x[r, :] + pe[r % L, :]) * mask[r, :]      (mask optional)
__global__ void add_pe_kernel(const float4* __restrict__ x, const float4* __restrict__ pe, const float4* __restrict__ mask,
                              float4* __restrict__ y, long n4, int D4, int L) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long r = i / D4;
    const int d = (int)(i - r * D4);
    const float4 a = x[i], b = pe[(r % L) * D4 + d];
    float4 v = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    if (mask) { const float4 m = mask[i]; v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w; }
    y[i] = v;
  }
}
extern "C" int lmkd_add_pe(const float* x, const float* pe, const float* mask, float* y, long rows, int D, int L, void* stream) {
  LMKD_REQUIRE(x && pe && y && rows > 0 && D % 4 == 0 && L > 0, "lmkd_add_pe: bad arguments");
  const long n4 = rows * D / 4;
  hipLaunchKernelGGL(add_pe_kernel, dim3(tx_grid(n4)), dim3(TX_THREADS), 0, (hipStream_t)stream, (const float4*)x, (const float4*)pe,
                     (const float4*)mask, (float4*)y, n4, D / 4, L);
  LMKD_CHECK_LAUNCH("add_pe_kernel");
  return LMKD_OK;
}

// out = a*b (b optional -> copy) ; out = alpha*a + beta*out variants used by the host glue
__global__ void mul_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ o, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 u = a[i], v = b[i];
    o[i] = make_float4(u.x * v.x, u.y * v.y, u.z * v.z, u.w * v.w);
  }
}
extern "C" int lmkd_mul(const float* a, const float* b, float* out, long n, void* stream) {
  LMKD_REQUIRE(a && b && out && n > 0 && n % 4 == 0, "lmkd_mul: bad arguments");
  hipLaunchKernelGGL(mul_kernel, dim3(tx_grid(n / 4)), dim3(TX_THREADS), 0, (hipStream_t)stream, (const float4*)a, (const float4*)b,
                     (float4*)out, n / 4);
  LMKD_CHECK_LAUNCH("mul_kernel");
  return LMKD_OK;
}
__global__ void axpby_kernel(const float* __restrict__ x, float* __restrict__ y, float alpha, float beta, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = alpha * x[i] + (beta != 0.f ? beta * y[i] : 0.f);
}
extern "C" int lmkd_axpby(const float* x, float* y, float alpha, float beta, long n, void* stream) {
  LMKD_REQUIRE(x && y && n > 0, "lmkd_axpby: bad arguments");
  hipLaunchKernelGGL(axpby_kernel, dim3(tx_grid(n)), dim3(TX_THREADS), 0, (hipStream_t)stream, x, y, alpha, beta, n);
  LMKD_CHECK_LAUNCH("axpby_kernel");
  return LMKD_OK;
}

// ---------------------------------------------------------------------------------
// tuple combine + LayerNorm on keys.  P: [NV*L, 4*D] = [k_a | k_b | v_a | v_b] per frame.
// tuple t = (i<j) in combinations(range(L),2) order.  Output row = rowmap[n]*T + t.
//   Kraw = P[n,i].k_a + P[n,j].k_b + bk ; Khat = (Kraw-mean)*rstd ; Kn = Khat*gamma+beta
//   V    = P[n,i].v_a + P[n,j].v_b + bv
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void tuple_ij(int t, int L, int& i, int& j) {
  int ii = 0, rem = t;
  while (rem >= L - 1 - ii) { rem -= L - 1 - ii; ++ii; }
  i = ii; j = ii + 1 + rem;
}

__global__ void trx_tuple_ln_fwd_kernel(const float* __restrict__ P, const float* __restrict__ bk, const float* __restrict__ bv,
                                        const float* __restrict__ gamma, const float* __restrict__ beta, const int* __restrict__ rowmap,
                                        float* __restrict__ Kn, float* __restrict__ Khat, float* __restrict__ V, float* __restrict__ rstd_out,
                                        int L, int T, int D, float eps) {
  __shared__ float red[4];
  extern __shared__ float kraw[];  // D floats
  const int n = blockIdx.x / T, t = blockIdx.x % T;
  int i, j;
  tuple_ij(t, L, i, j);
  const float* pi = P + (long)(n * L + i) * 4 * D;
  const float* pj = P + (long)(n * L + j) * 4 * D;
  const long orow = (long)(rowmap ? rowmap[n] : n) * T + t;
  float s = 0.f;
  for (int d = threadIdx.x; d < D; d += TX_THREADS) {
    const float k = pi[d] + pj[D + d] + bk[d];
    kraw[d] = k;
    s += k;
    V[orow * D + d] = pi[2 * D + d] + pj[3 * D + d] + bv[d];
  }
  const float mean = block_sum_256(s, red) / (float)D;
  float q = 0.f;
  for (int d = threadIdx.x; d < D; d += TX_THREADS) { const float c = kraw[d] - mean; q += c * c; }
  const float var = block_sum_256(q, red) / (float)D;
  const float rstd = 1.f / sqrtf(var + eps);
  for (int d = threadIdx.x; d < D; d += TX_THREADS) {
    const float h = (kraw[d] - mean) * rstd;
    if (Khat) Khat[orow * D + d] = h;
    Kn[orow * D + d] = h * gamma[d] + beta[d];
  }
  if (threadIdx.x == 0 && rstd_out) rstd_out[orow] = rstd;
}

extern "C" int lmkd_trx_tuple_ln_fwd(const float* P, const float* bk, const float* bv, const float* gamma, const float* beta,
                                     const int* rowmap, float* Kn, float* Khat, float* V, float* rstd, int NV, int L, int D, float eps,
                                     void* stream) {
  LMKD_REQUIRE(P && bk && bv && gamma && beta && Kn && V && NV > 0 && L >= 2 && D > 0, "lmkd_trx_tuple_ln_fwd: bad arguments");
  const int T = L * (L - 1) / 2;
  hipLaunchKernelGGL(trx_tuple_ln_fwd_kernel, dim3(NV * T), dim3(TX_THREADS), D * sizeof(float), (hipStream_t)stream, P, bk, bv, gamma,
                     beta, rowmap, Kn, Khat, V, rstd, L, T, D, eps);
  LMKD_CHECK_LAUNCH("trx_tuple_ln_fwd_kernel");
  return LMKD_OK;
}

// LayerNorm backward per row (in place: dKn -> dKraw):  dxh = dKn*gamma ;
//   dKraw = rstd*(dxh - mean(dxh) - Khat*mean(dxh*Khat))
__global__ void ln_bwd_rows_kernel(float* __restrict__ dK, const float* __restrict__ Khat, const float* __restrict__ rstd,
                                   const float* __restrict__ gamma, int D) {
  __shared__ float red[4];
  const long r = blockIdx.x;
  float* g = dK + r * D;
  const float* h = Khat + r * D;
  float s1 = 0.f, s2 = 0.f;
  for (int d = threadIdx.x; d < D; d += TX_THREADS) {
    const float v = g[d] * gamma[d];
    s1 += v;
    s2 += v * h[d];
  }
  const float m1 = block_sum_256(s1, red) / (float)D;
  const float m2 = block_sum_256(s2, red) / (float)D;
  const float rs = rstd[r];
  for (int d = threadIdx.x; d < D; d += TX_THREADS) g[d] = rs * (g[d] * gamma[d] - m1 - h[d] * m2);
}
extern "C" int lmkd_layernorm_bwd_rows(float* dK_inout, const float* Khat, const float* rstd, const float* gamma, long rows, int D,
                                       void* stream) {
  LMKD_REQUIRE(dK_inout && Khat && rstd && gamma && rows > 0 && D > 0, "lmkd_layernorm_bwd_rows: bad arguments");
  hipLaunchKernelGGL(ln_bwd_rows_kernel, dim3((unsigned)rows), dim3(TX_THREADS), 0, (hipStream_t)stream, dK_inout, Khat, rstd, gamma, D);
  LMKD_CHECK_LAUNCH("ln_bwd_rows_kernel");
  return LMKD_OK;
}

// dP[n*L+f] = [ sum_{t:i_t=f} dKraw | sum_{t:j_t=f} dKraw | sum_{t:i_t=f} dV | sum_{t:j_t=f} dV ]
__global__ void trx_tuple_bwd_gather_kernel(const float* __restrict__ dKraw, const float* __restrict__ dV, const int* __restrict__ rowmap,
                                            float* __restrict__ dP, int L, int T, int D) {
  const int n = blockIdx.x / L, f = blockIdx.x % L;
  const long base = (long)(rowmap ? rowmap[n] : n) * T;
  float* o = dP + (long)blockIdx.x * 4 * D;
  for (int d = threadIdx.x; d < D; d += TX_THREADS) {
    float ka = 0.f, kb = 0.f, va = 0.f, vb = 0.f;
    int t = 0;
    for (int i = 0; i < L; ++i)
      for (int j = i + 1; j < L; ++j, ++t) {
        if (i == f) { ka += dKraw[(base + t) * D + d]; va += dV[(base + t) * D + d]; }
        else if (j == f) { kb += dKraw[(base + t) * D + d]; vb += dV[(base + t) * D + d]; }
      }
    o[d] = ka; o[D + d] = kb; o[2 * D + d] = va; o[3 * D + d] = vb;
  }
}
extern "C" int lmkd_trx_tuple_bwd_gather(const float* dKraw, const float* dV, const int* rowmap, float* dP, int NV, int L, int D,
                                         void* stream) {
  LMKD_REQUIRE(dKraw && dV && dP && NV > 0 && L >= 2, "lmkd_trx_tuple_bwd_gather: bad arguments");
  hipLaunchKernelGGL(trx_tuple_bwd_gather_kernel, dim3(NV * L), dim3(TX_THREADS), 0, (hipStream_t)stream, dKraw, dV, rowmap, dP, L,
                     L * (L - 1) / 2, D);
  LMKD_CHECK_LAUNCH("trx_tuple_bwd_gather_kernel");
  return LMKD_OK;
}

// ---------------------------------------------------------------------------------
// per-class softmax over column segments of S [R, ld] (in place) and its backward
// ---------------------------------------------------------------------------------
__global__ void segment_softmax_fwd_kernel(float* __restrict__ S, long ld, Segs sg) {
  __shared__ float red[4];
  float* row = S + (long)blockIdx.x * ld;
  for (int s = 0; s < sg.n; ++s) {
    float* p = row + sg.off[s];
    const int n = sg.cnt[s];
    float m = -INFINITY;
    for (int c = threadIdx.x; c < n; c += TX_THREADS) m = fmaxf(m, p[c]);
    m = block_max_256(m, red);
    float sum = 0.f;
    for (int c = threadIdx.x; c < n; c += TX_THREADS) { const float e = expf(p[c] - m); p[c] = e; sum += e; }
    sum = block_sum_256(sum, red);
    const float inv = 1.f / sum;
    for (int c = threadIdx.x; c < n; c += TX_THREADS) p[c] *= inv;
  }
}
// dS = P * (dP - sum_seg(P*dP)), written over dP
__global__ void segment_softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, long ld, Segs sg) {
  __shared__ float red[4];
  const float* prow = P + (long)blockIdx.x * ld;
  float* drow = dP + (long)blockIdx.x * ld;
  for (int s = 0; s < sg.n; ++s) {
    const float* p = prow + sg.off[s];
    float* d = drow + sg.off[s];
    const int n = sg.cnt[s];
    float dot = 0.f;
    for (int c = threadIdx.x; c < n; c += TX_THREADS) dot += p[c] * d[c];
    dot = block_sum_256(dot, red);
    for (int c = threadIdx.x; c < n; c += TX_THREADS) d[c] = p[c] * (d[c] - dot);
  }
}
static int fill_segs(Segs* sg, int nseg, const int* off, const int* cnt, const int* col) {
  if (nseg < 1 || nseg > LMKD_MAX_SEG) return -1;
  memset(sg, 0, sizeof(*sg));
  sg->n = nseg;
  for (int i = 0; i < nseg; ++i) { sg->off[i] = off[i]; sg->cnt[i] = cnt[i]; sg->col[i] = col ? col[i] : i; }
  return 0;
}
extern "C" int lmkd_segment_softmax_fwd(float* S, long rows, long ld, int nseg, const int* seg_off, const int* seg_cnt, void* stream) {
  Segs sg;
  LMKD_REQUIRE(S && rows > 0 && seg_off && seg_cnt && fill_segs(&sg, nseg, seg_off, seg_cnt, nullptr) == 0,
               "lmkd_segment_softmax_fwd: bad arguments (1..%d segments)", LMKD_MAX_SEG);
  hipLaunchKernelGGL(segment_softmax_fwd_kernel, dim3((unsigned)rows), dim3(TX_THREADS), 0, (hipStream_t)stream, S, ld, sg);
  LMKD_CHECK_LAUNCH("segment_softmax_fwd_kernel");
  return LMKD_OK;
}
extern "C" int lmkd_segment_softmax_bwd(const float* P, float* dP_inout, long rows, long ld, int nseg, const int* seg_off,
                                        const int* seg_cnt, void* stream) {
  Segs sg;
  LMKD_REQUIRE(P && dP_inout && rows > 0 && seg_off && seg_cnt && fill_segs(&sg, nseg, seg_off, seg_cnt, nullptr) == 0,
               "lmkd_segment_softmax_bwd: bad arguments");
  hipLaunchKernelGGL(segment_softmax_bwd_kernel, dim3((unsigned)rows), dim3(TX_THREADS), 0, (hipStream_t)stream, P, dP_inout, ld, sg);
  LMKD_CHECK_LAUNCH("segment_softmax_bwd_kernel");
  return LMKD_OK;
}

// ---------------------------------------------------------------------------------
// logits[q, col(seg)] = -(1/T) * sum_{t,d} (Qv[q*T+t, d] - proto[seg][q*T+t, d])^2
// ---------------------------------------------------------------------------------
__global__ void trx_dist_fwd_kernel(const float4* __restrict__ Qv, const float4* __restrict__ proto, float* __restrict__ logits, int way,
                                    int T, int D4, long seg_stride4, Segs sg) {
  __shared__ float red[4];
  const int q = blockIdx.x, s = blockIdx.y;
  const float4* a = Qv + (long)q * T * D4;
  const float4* b = proto + (long)s * seg_stride4 + (long)q * T * D4;
  float acc = 0.f;
  for (int i = threadIdx.x; i < T * D4; i += TX_THREADS) {
    const float4 u = a[i], v = b[i];
    const float dx = u.x - v.x, dy = u.y - v.y, dz = u.z - v.z, dw = u.w - v.w;
    acc += dx * dx + dy * dy + dz * dz + dw * dw;
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) logits[(long)q * way + sg.col[s]] = -acc / (float)T;
}
extern "C" int lmkd_trx_dist_fwd(const float* Qv, const float* proto, float* logits, int Nq, int way, int T, int D, int nseg,
                                 const int* seg_col, void* stream) {
  Segs sg;
  int zeros[LMKD_MAX_SEG] = {0};
  LMKD_REQUIRE(Qv && proto && logits && Nq > 0 && D % 4 == 0 && seg_col && nseg <= LMKD_MAX_SEG &&
                   fill_segs(&sg, nseg, zeros, zeros, seg_col) == 0, "lmkd_trx_dist_fwd: bad arguments");
  for (int i = 0; i < nseg; ++i) LMKD_REQUIRE(seg_col[i] >= 0 && seg_col[i] < way, "lmkd_trx_dist_fwd: class label %d outside [0,%d)", seg_col[i], way);
  hipLaunchKernelGGL(trx_dist_fwd_kernel, dim3(Nq, nseg), dim3(TX_THREADS), 0, (hipStream_t)stream, (const float4*)Qv,
                     (const float4*)proto, logits, way, T, D / 4, (long)Nq * T * D / 4, sg);
  LMKD_CHECK_LAUNCH("trx_dist_fwd_kernel");
  return LMKD_OK;
}

// proto[seg] <- dproto[seg] = (2/T) g[q,col] (Qv - proto[seg]) ;  dQv = -sum_seg dproto[seg]
__global__ void trx_dist_bwd_kernel(const float4* __restrict__ Qv, float4* __restrict__ proto, const float* __restrict__ g,
                                    float4* __restrict__ dQv, int way, int T, int D4, long seg_stride4, long n4, Segs sg) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i / ((long)T * D4));
    const float4 u = Qv[i];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < sg.n; ++s) {
      const float cf = g[(long)q * way + sg.col[s]] * (2.f / (float)T);
      float4 v = proto[(long)s * seg_stride4 + i];
      v = make_float4(cf * (u.x - v.x), cf * (u.y - v.y), cf * (u.z - v.z), cf * (u.w - v.w));
      proto[(long)s * seg_stride4 + i] = v;
      acc.x -= v.x; acc.y -= v.y; acc.z -= v.z; acc.w -= v.w;
    }
    dQv[i] = acc;
  }
}
extern "C" int lmkd_trx_dist_bwd(const float* Qv, float* proto_inout, const float* g, float* dQv, int Nq, int way, int T, int D,
                                 int nseg, const int* seg_col, void* stream) {
  Segs sg;
  int zeros[LMKD_MAX_SEG] = {0};
  LMKD_REQUIRE(Qv && proto_inout && g && dQv && D % 4 == 0 && seg_col && nseg <= LMKD_MAX_SEG &&
                   fill_segs(&sg, nseg, zeros, zeros, seg_col) == 0, "lmkd_trx_dist_bwd: bad arguments");
  const long n4 = (long)Nq * T * D / 4;
  hipLaunchKernelGGL(trx_dist_bwd_kernel, dim3(tx_grid(n4)), dim3(TX_THREADS), 0, (hipStream_t)stream, (const float4*)Qv,
                     (float4*)proto_inout, g, (float4*)dQv, way, T, D / 4, n4, n4, sg);
  LMKD_CHECK_LAUNCH("trx_dist_bwd_kernel");
  return LMKD_OK;
}

// ---------------------------------------------------------------------------------
// SupportDK: prototypes = support.reshape(way, shot, E).mean(1) (labels ignored, as the reference);
//   out[i, m] = -|P_i - P_j|^2 / L,  j = m + (m >= i)
// ---------------------------------------------------------------------------------
// Stage 1: every workgroup takes a slice of the E = L*D feature elements, forms the `way` prototype values of each element
// once and the squared differences of all way*(way-1)/2 pairs, and writes one partial sum per pair (grid-wide instead of the
// one-workgroup-per-output form: 20 workgroups walking 16 K elements each took 85 us for a 1.6 MB input).  Stage 2: one wave
// adds the partials in a fixed order (deterministic) and fills both (i,j) and (j,i).
// WAY > 0: compile-time way (all loops unrolled, accumulators in registers); WAY == 0: run-time way (generic fallback)
template <int WAY>
__global__ void supportdk_partial_kernel(const float* __restrict__ sup, float* __restrict__ partial, int way_rt, int shot, long E) {
  __shared__ float red[4];
  const int way = WAY > 0 ? WAY : way_rt;
  constexpr int MAXW = WAY > 0 ? WAY : LMKD_MAX_SEG;
  const float inv = 1.f / (float)shot;
  float acc[MAXW * (MAXW - 1) / 2];
  const int npair = way * (way - 1) / 2;
#pragma unroll
  for (int p = 0; p < MAXW * (MAXW - 1) / 2; ++p) acc[p] = 0.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
    float P[MAXW];
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
      if (i >= way) break;
      float sacc = 0.f;
      for (int k = 0; k < shot; ++k) sacc += sup[((long)i * shot + k) * E + e];
      P[i] = sacc;
    }
    int p = 0;
#pragma unroll
    for (int i = 0; i < MAXW; ++i)
#pragma unroll
      for (int j = i + 1; j < MAXW; ++j) {
        if (j < way) {
          const float d = (P[i] - P[j]) * inv;
          acc[p] += d * d;
          ++p;
        }
      }
  }
#pragma unroll
  for (int p = 0; p < MAXW * (MAXW - 1) / 2; ++p) {
    if (p >= npair) break;
    const float v = block_sum_256(acc[p], red);
    if (threadIdx.x == 0) partial[(long)blockIdx.x * npair + p] = v;
    __syncthreads();
  }
}
__global__ void supportdk_finish_kernel(const float* __restrict__ partial, float* __restrict__ out, int way, int nblk, float inv_len) {
  const int npair = way * (way - 1) / 2;            // up to 120 pairs (way <= LMKD_MAX_SEG = 16): every pair is visited whatever the block size
  for (int p = threadIdx.x; p < npair; p += blockDim.x) {
    float v = 0.f;
    for (int b = 0; b < nblk; ++b) v += partial[(long)b * npair + p];
    int i = 0, rem = p;                               // pair index -> (i, j), i < j
    while (rem >= way - 1 - i) { rem -= way - 1 - i; ++i; }
    const int j = i + 1 + rem;
    out[i * (way - 1) + (j - 1)] = -v * inv_len;      // row i lists j != i ascending: column j-1 for j > i
    out[j * (way - 1) + i] = -v * inv_len;            // row j: column i for i < j
  }
}
// dsup[i*shot+s, e] = sum_{j != i} (G[i][j] + G[j][i]) * (-2/L) * (P_i - P_j) / shot
__global__ void supportdk_bwd_kernel(const float* __restrict__ sup, const float* __restrict__ g, float* __restrict__ dsup, int way,
                                     int shot, long E, float inv_len) {
  const float inv = 1.f / (float)shot;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
    float P[LMKD_MAX_SEG];
    for (int i = 0; i < way; ++i) {
      float s = 0.f;
      for (int k = 0; k < shot; ++k) s += sup[((long)i * shot + k) * E + e];
      P[i] = s * inv;
    }
    for (int i = 0; i < way; ++i) {
      float acc = 0.f;
      for (int j = 0; j < way; ++j) {
        if (j == i) continue;
        const float gij = g[i * (way - 1) + (j > i ? j - 1 : j)];
        const float gji = g[j * (way - 1) + (i > j ? i - 1 : i)];
        acc += (gij + gji) * (P[i] - P[j]);
      }
      acc *= -2.f * inv_len * inv;
      for (int k = 0; k < shot; ++k) dsup[((long)i * shot + k) * E + e] = acc;
    }
  }
}
extern "C" long lmkd_supportdk_workspace(int way) { return 256L * way * (way - 1) / 2 * (long)sizeof(float); }
extern "C" int lmkd_supportdk_fwd(const float* support, float* out, int way, int shot, int seq_len, int D, void* workspace, void* stream) {
  LMKD_REQUIRE(support && out && workspace && way >= 2 && way <= LMKD_MAX_SEG && shot > 0, "lmkd_supportdk_fwd: bad arguments");
  const long E = (long)seq_len * D;
  int nblk = (int)((E + TX_THREADS - 1) / TX_THREADS);
  if (nblk > 256) nblk = 256;
  if (way == 5) hipLaunchKernelGGL(supportdk_partial_kernel<5>, dim3(nblk), dim3(TX_THREADS), 0, (hipStream_t)stream, support, (float*)workspace, way, shot, E);
  else hipLaunchKernelGGL(supportdk_partial_kernel<0>, dim3(nblk), dim3(TX_THREADS), 0, (hipStream_t)stream, support, (float*)workspace, way, shot, E);
  hipLaunchKernelGGL(supportdk_finish_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, (const float*)workspace, out, way, nblk,
                     1.f / (float)seq_len);
  LMKD_CHECK_LAUNCH("supportdk kernels");
  return LMKD_OK;
}
extern "C" int lmkd_supportdk_bwd(const float* support, const float* g, float* dsupport, int way, int shot, int seq_len, int D,
                                  void* stream) {
  LMKD_REQUIRE(support && g && dsupport && way >= 2 && way <= LMKD_MAX_SEG && shot > 0, "lmkd_supportdk_bwd: bad arguments");
  const long E = (long)seq_len * D;
  hipLaunchKernelGGL(supportdk_bwd_kernel, dim3(tx_grid(E)), dim3(TX_THREADS), 0, (hipStream_t)stream, support, g, dsupport, way, shot, E,
                     1.f / (float)seq_len);
  LMKD_CHECK_LAUNCH("supportdk_bwd_kernel");
  return LMKD_OK;
}

// ---------------------------------------------------------------------------------
// Euclidean matcher (e_dist): frame-mean embeddings, cdist(p=2) to every shot, class mean
//   mean_frames: [NV, L, D] -> [NV, D]
//   dist[q, s] = |qm[q] - sm[s]|_2 ;  logits[q, c] = -mean_{s in class c} dist[q, s]
// ---------------------------------------------------------------------------------
__global__ void mean_frames_kernel(const float* __restrict__ x, float* __restrict__ y, long nv, int L, int D) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv * D; i += (long)gridDim.x * blockDim.x) {
    const long v = i / D;
    const int d = (int)(i - v * D);
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += x[(v * L + l) * D + d];
    y[i] = s / (float)L;
  }
}
extern "C" int lmkd_mean_frames(const float* x, float* y, long nv, int L, int D, void* stream) {
  LMKD_REQUIRE(x && y && nv > 0 && L > 0 && D > 0, "lmkd_mean_frames: bad arguments");
  hipLaunchKernelGGL(mean_frames_kernel, dim3(tx_grid(nv * D)), dim3(TX_THREADS), 0, (hipStream_t)stream, x, y, nv, L, D);
  LMKD_CHECK_LAUNCH("mean_frames_kernel");
  return LMKD_OK;
}
// dx[v,l,d] = dy[v,d]/L
__global__ void mean_frames_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long nv, int L, int D) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv * L * D; i += (long)gridDim.x * blockDim.x) {
    const long v = i / ((long)L * D);
    const int d = (int)(i % D);
    dx[i] = dy[v * D + d] / (float)L;
  }
}
extern "C" int lmkd_mean_frames_bwd(const float* dy, float* dx, long nv, int L, int D, void* stream) {
  LMKD_REQUIRE(dy && dx && nv > 0, "lmkd_mean_frames_bwd: bad arguments");
  hipLaunchKernelGGL(mean_frames_bwd_kernel, dim3(tx_grid(nv * L * D)), dim3(TX_THREADS), 0, (hipStream_t)stream, dy, dx, nv, L, D);
  LMKD_CHECK_LAUNCH("mean_frames_bwd_kernel");
  return LMKD_OK;
}

__global__ void cdist_kernel(const float* __restrict__ qm, const float* __restrict__ sm, float* __restrict__ dist, int Ns, int D) {
  __shared__ float red[4];
  const int q = blockIdx.x, s = blockIdx.y;
  float acc = 0.f;
  for (int d = threadIdx.x; d < D; d += TX_THREADS) { const float t = qm[(long)q * D + d] - sm[(long)s * D + d]; acc += t * t; }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) dist[(long)q * Ns + s] = sqrtf(acc);
}
// labels: class id per support video (int), counts per class
__global__ void edist_logits_kernel(const float* __restrict__ dist, const int* __restrict__ cls, float* __restrict__ logits, int Nq,
                                    int Ns, int way) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Nq * way) return;
  const int q = i / way, c = i % way;
  float s = 0.f;
  int n = 0;
  for (int k = 0; k < Ns; ++k)
    if (cls[k] == c) { s += dist[(long)q * Ns + k]; ++n; }
  logits[i] = n > 0 ? -s / (float)n : 0.f;
}
extern "C" int lmkd_edist_fwd(const float* qm, const float* sm, const int* sup_class, float* dist, float* logits, int Nq, int Ns, int way,
                              int D, void* stream) {
  LMKD_REQUIRE(qm && sm && sup_class && dist && logits && Nq > 0 && Ns > 0 && way > 0, "lmkd_edist_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cdist_kernel, dim3(Nq, Ns), dim3(TX_THREADS), 0, st, qm, sm, dist, Ns, D);
  hipLaunchKernelGGL(edist_logits_kernel, dim3(cdiv(Nq * way, 64)), dim3(64), 0, st, (const float*)dist, sup_class, logits, Nq, Ns, way);
  LMKD_CHECK_LAUNCH("edist kernels");
  return LMKD_OK;
}
// w[q,s] = -g[q, cls[s]] / count(cls[s]) / dist[q,s]   (0 where dist == 0, like torch.cdist backward)
//   dqm[q] = sum_s w (qm[q]-sm[s]) ; dsm[s] = -sum_q w (qm[q]-sm[s])
__global__ void edist_bwd_kernel(const float* __restrict__ qm, const float* __restrict__ sm, const int* __restrict__ cls,
                                 const float* __restrict__ dist, const float* __restrict__ g, float* __restrict__ dqm,
                                 float* __restrict__ dsm, int Nq, int Ns, int way, int D) {
  // blockIdx.y == 0: query rows, == 1: support rows; blockIdx.x: row
  const bool is_q = blockIdx.y == 0;
  const int r = blockIdx.x;
  if (is_q ? r >= Nq : r >= Ns) return;
  __shared__ float w[256];
  __shared__ int cnt[LMKD_MAX_SEG * 4];
  if (threadIdx.x < way) {
    int n = 0;
    for (int k = 0; k < Ns; ++k) n += cls[k] == threadIdx.x;
    cnt[threadIdx.x] = n;
  }
  __syncthreads();
  const int other = is_q ? Ns : Nq;
  for (int k = threadIdx.x; k < other; k += TX_THREADS) {
    const int q = is_q ? r : k, s = is_q ? k : r;
    const float dd = dist[(long)q * Ns + s];
    w[k] = dd > 0.f ? -g[(long)q * way + cls[s]] / (float)cnt[cls[s]] / dd : 0.f;
  }
  __syncthreads();
  const float* self = is_q ? qm + (long)r * D : sm + (long)r * D;
  const float* oth = is_q ? sm : qm;
  float* out = is_q ? dqm + (long)r * D : dsm + (long)r * D;
  for (int d = threadIdx.x; d < D; d += TX_THREADS) {
    float acc = 0.f;
    for (int k = 0; k < other; ++k) acc += w[k] * (self[d] - oth[(long)k * D + d]);
    out[d] = acc;   // for support rows: d/dsm of |qm-sm| is -(qm-sm)/dist = (sm-qm)/dist = (self-oth)/dist
  }
}
extern "C" int lmkd_edist_bwd(const float* qm, const float* sm, const int* sup_class, const float* dist, const float* g, float* dqm,
                              float* dsm, int Nq, int Ns, int way, int D, void* stream) {
  LMKD_REQUIRE(qm && sm && sup_class && dist && g && dqm && dsm, "lmkd_edist_bwd: null pointer");
  LMKD_REQUIRE(Nq <= 256 && Ns <= 256 && way <= LMKD_MAX_SEG * 4, "lmkd_edist_bwd: episode too large (Nq,Ns <= 256)");
  const int rows = Nq > Ns ? Nq : Ns;
  hipLaunchKernelGGL(edist_bwd_kernel, dim3(rows, 2), dim3(TX_THREADS), 0, (hipStream_t)stream, qm, sm, sup_class, dist, g, dqm, dsm, Nq,
                     Ns, way, D);
  LMKD_CHECK_LAUNCH("edist_bwd_kernel");
  return LMKD_OK;
}


// ---------------------------------------------------------------------------------
// TRX_sup (TRX_sup.py:117-172): cosine similarity between the query-specific class prototypes of one query,
//   sim[q][a][b] = <P_qa, P_qb> / max(|P_qa| |P_qb|, eps),  P_qc = proto[seg(c)][q*T .. q*T+T-1][:]  (T*D values; classes without
// support keep the reference's all-zero prototype -> similarity 0).  One workgroup per query.
// ---------------------------------------------------------------------------------
#define TRXS_MAXW 8
struct TrxsCols { int c[TRXS_MAXW]; };   // class of each support segment, passed by value (host array at the C ABI)
__global__ __launch_bounds__(256) void trx_sup_sim_fwd_kernel(const float* __restrict__ proto, TrxsCols seg_col,
                                                              float* __restrict__ sim, float* __restrict__ gram, int way, int nseg,
                                                              long Dp, long seg_stride) {
  const int q = blockIdx.x;
  float g[TRXS_MAXW * (TRXS_MAXW + 1) / 2];
#pragma unroll
  for (int i = 0; i < TRXS_MAXW * (TRXS_MAXW + 1) / 2; ++i) g[i] = 0.f;
  const float* base = proto + (long)q * Dp;
  for (long k = threadIdx.x; k < Dp; k += 256) {
    float v[TRXS_MAXW];
#pragma unroll
    for (int a = 0; a < TRXS_MAXW; ++a) v[a] = a < nseg ? base[a * seg_stride + k] : 0.f;
    int i = 0;
#pragma unroll
    for (int a = 0; a < TRXS_MAXW; ++a)
#pragma unroll
      for (int b = a; b < TRXS_MAXW; ++b, ++i) g[i] = fmaf(v[a], v[b], g[i]);
  }
  __shared__ float G[TRXS_MAXW][TRXS_MAXW];
  __shared__ float red[4];
  {
    int i = 0;
    for (int a = 0; a < TRXS_MAXW; ++a)
      for (int b = a; b < TRXS_MAXW; ++b, ++i) {
        const float t = block_sum_256(g[i], red);
        if (threadIdx.x == 0) { G[a][b] = t; G[b][a] = t; }
      }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < way * way; i += 256) sim[(long)q * way * way + i] = 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < nseg * nseg; i += 256) {
    const int a = i / nseg, b = i - a * nseg;
    const float den = fmaxf(sqrtf(G[a][a]) * sqrtf(G[b][b]), 1e-8f);
    sim[((long)q * way + seg_col.c[a]) * way + seg_col.c[b]] = G[a][b] / den;
    gram[((long)q * nseg + a) * nseg + b] = G[a][b];
  }
}

// dP_a = sum_{b != a} (dsim[a][b] + dsim[b][a]) * (P_b / (n_a n_b) - G_ab P_a / (n_a^3 n_b))
__global__ __launch_bounds__(256) void trx_sup_sim_bwd_kernel(const float* __restrict__ proto, TrxsCols seg_col,
                                                              const float* __restrict__ gram, const float* __restrict__ dsim,
                                                              float* __restrict__ dproto, int way, int nseg, long Dp, long seg_stride) {
  const int q = blockIdx.x;
  __shared__ float c1[TRXS_MAXW][TRXS_MAXW], c2[TRXS_MAXW];   // dP_a = sum_b c1[a][b] P_b - c2[a] P_a
  if (threadIdx.x < nseg) {
    const int a = threadIdx.x;
    const float* G = gram + (long)q * nseg * nseg;
    const float na = sqrtf(G[a * nseg + a]);
    float acc2 = 0.f;
    for (int b = 0; b < nseg; ++b) {
      float w = 0.f, cb = 0.f;
      if (b != a) {
        const float nb = sqrtf(G[b * nseg + b]);
        const float den = fmaxf(na * nb, 1e-8f);
        w = dsim[((long)q * way + seg_col.c[a]) * way + seg_col.c[b]] + dsim[((long)q * way + seg_col.c[b]) * way + seg_col.c[a]];
        cb = w / den;
        acc2 += na > 0.f ? w * G[a * nseg + b] / (den * na * na) : 0.f;
      }
      c1[a][b] = cb;
    }
    c2[a] = acc2;
  }
  __syncthreads();
  const float* base = proto + (long)q * Dp;
  float* out = dproto + (long)q * Dp;
  for (long k = threadIdx.x; k < Dp; k += 256) {
    float v[TRXS_MAXW];
#pragma unroll
    for (int a = 0; a < TRXS_MAXW; ++a) v[a] = a < nseg ? base[a * seg_stride + k] : 0.f;
#pragma unroll
    for (int a = 0; a < TRXS_MAXW; ++a) {
      if (a >= nseg) break;
      float d = -c2[a] * v[a];
#pragma unroll
      for (int b = 0; b < TRXS_MAXW; ++b)
        if (b < nseg) d = fmaf(c1[a][b], v[b], d);
      out[a * seg_stride + k] = d;
    }
  }
}

extern "C" int lmkd_trx_sup_sim_fwd(const float* proto, const int* seg_col, float* sim, float* gram, int Nq, int way, int nseg, int T,
                                    int D, void* stream) {
  LMKD_REQUIRE(proto && seg_col && sim && gram && Nq > 0 && T > 0 && D > 0, "lmkd_trx_sup_sim_fwd: bad arguments");
  LMKD_REQUIRE(nseg >= 1 && nseg <= way && way <= TRXS_MAXW, "lmkd_trx_sup_sim_fwd: 1 <= classes with support <= way <= %d", TRXS_MAXW);
  TrxsCols cols;
  for (int i = 0; i < TRXS_MAXW; ++i) cols.c[i] = i < nseg ? seg_col[i] : 0;
  for (int i = 0; i < nseg; ++i) LMKD_REQUIRE(seg_col[i] >= 0 && seg_col[i] < way, "lmkd_trx_sup_sim_fwd: class label %d outside [0,%d)", seg_col[i], way);
  hipLaunchKernelGGL(trx_sup_sim_fwd_kernel, dim3(Nq), dim3(256), 0, (hipStream_t)stream, proto, cols, sim, gram, way, nseg,
                     (long)T * D, (long)Nq * T * D);
  LMKD_CHECK_LAUNCH("trx_sup_sim_fwd_kernel");
  return LMKD_OK;
}

extern "C" int lmkd_trx_sup_sim_bwd(const float* proto, const int* seg_col, const float* gram, const float* dsim, float* dproto, int Nq,
                                    int way, int nseg, int T, int D, void* stream) {
  LMKD_REQUIRE(proto && seg_col && gram && dsim && dproto && Nq > 0 && T > 0 && D > 0, "lmkd_trx_sup_sim_bwd: bad arguments");
  LMKD_REQUIRE(nseg >= 1 && nseg <= way && way <= TRXS_MAXW, "lmkd_trx_sup_sim_bwd: 1 <= classes with support <= way <= %d", TRXS_MAXW);
  TrxsCols cols;
  for (int i = 0; i < TRXS_MAXW; ++i) cols.c[i] = i < nseg ? seg_col[i] : 0;
  for (int i = 0; i < nseg; ++i) LMKD_REQUIRE(seg_col[i] >= 0 && seg_col[i] < way, "lmkd_trx_sup_sim_bwd: class label %d outside [0,%d)", seg_col[i], way);
  hipLaunchKernelGGL(trx_sup_sim_bwd_kernel, dim3(Nq), dim3(256), 0, (hipStream_t)stream, proto, cols, gram, dsim, dproto, way, nseg,
                     (long)T * D, (long)Nq * T * D);
  LMKD_CHECK_LAUNCH("trx_sup_sim_bwd_kernel");
  return LMKD_OK;
}
