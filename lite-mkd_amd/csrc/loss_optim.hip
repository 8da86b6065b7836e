// D2M distillation loss (temperature-KL + inter-class Pearson relation + cross entropy),
// forward values and logits gradients in ONE single-workgroup launch (the tensors are
// [25,5] / [5,4]: launch latency is the only cost), plus the accuracy (argmax) kernel and
// the flat-buffer optimizer steps.
#include "common.h"

#define LO_THREADS 256
#define LO_MAXC 64

// --- row primitives: one thread handles one row of C <= LO_MAXC logits ----------------------
// temperature KL: returns sum_c pt*(log pt - log_softmax(s/T))_c ; grad (unscaled) = softmax(s/T) - pt
__device__ float kd_row(const float* s, const float* t, int C, float T, float* gs) {
  float ms = -INFINITY, mt = -INFINITY;
  for (int c = 0; c < C; ++c) { ms = fmaxf(ms, s[c] / T); mt = fmaxf(mt, t[c] / T); }
  float zs = 0.f, zt = 0.f;
  for (int c = 0; c < C; ++c) { zs += expf(s[c] / T - ms); zt += expf(t[c] / T - mt); }
  const float lzs = logf(zs), lzt = logf(zt);
  float loss = 0.f;
  for (int c = 0; c < C; ++c) {
    const float ls = s[c] / T - ms - lzs;
    const float lt = t[c] / T - mt - lzt;
    const float pt = expf(lt);
    if (pt > 0.f) loss += pt * (lt - ls);
    if (gs) gs[c] = expf(ls) - pt;
  }
  return loss;
}

// Pearson/cosine relation of softmaxed rows: returns cos ; grad (d cos / d ys) in gy
__device__ float icr_row(const float* ys, const float* yt, int C, float* gy) {
  float a[LO_MAXC], b[LO_MAXC];
  float ma = -INFINITY, mb = -INFINITY;
  for (int c = 0; c < C; ++c) { ma = fmaxf(ma, ys[c]); mb = fmaxf(mb, yt[c]); }
  float za = 0.f, zb = 0.f;
  for (int c = 0; c < C; ++c) { a[c] = expf(ys[c] - ma); za += a[c]; b[c] = expf(yt[c] - mb); zb += b[c]; }
  float mean_a = 0.f, mean_b = 0.f;
  for (int c = 0; c < C; ++c) { a[c] /= za; b[c] /= zb; mean_a += a[c]; mean_b += b[c]; }
  mean_a /= (float)C; mean_b /= (float)C;
  float dot = 0.f, na2 = 0.f, nb2 = 0.f;
  for (int c = 0; c < C; ++c) {
    const float x = a[c] - mean_a, y = b[c] - mean_b;
    dot += x * y; na2 += x * x; nb2 += y * y;
  }
  const float na = sqrtf(na2), nb = sqrtf(nb2);
  const float den = na * nb + 1e-8f;
  const float cosv = dot / den;
  if (gy) {
    // d cos / d ac = bc/den - dot*nb*ac/(na*den^2)  (second term 0 when na == 0) ; then centre, then softmax-bwd
    float dac[LO_MAXC];
    float mean_d = 0.f;
    const float k2 = na > 0.f ? dot * nb / (na * den * den) : 0.f;
    for (int c = 0; c < C; ++c) { dac[c] = (b[c] - mean_b) / den - k2 * (a[c] - mean_a); mean_d += dac[c]; }
    mean_d /= (float)C;
    float sdot = 0.f;
    for (int c = 0; c < C; ++c) { dac[c] -= mean_d; sdot += a[c] * dac[c]; }
    for (int c = 0; c < C; ++c) gy[c] = a[c] * (dac[c] - sdot);
  }
  return cosv;
}

// cross entropy row: returns logsumexp - s[label] ; grad (unscaled) = softmax - onehot
__device__ float ce_row(const float* s, long long label, int C, float* gs) {
  float m = -INFINITY;
  for (int c = 0; c < C; ++c) m = fmaxf(m, s[c]);
  float z = 0.f;
  for (int c = 0; c < C; ++c) z += expf(s[c] - m);
  const float lz = logf(z) + m;
  if (gs) for (int c = 0; c < C; ++c) gs[c] = expf(s[c] - lz) - (c == (int)label ? 1.f : 0.f);
  return lz - s[label];
}

struct D2MArgs {
  const float* s_kl; const float* t_kl; const float* s_ce; const long long* labels;
  const float* s_sup; const float* t_sup;
  int Rq, C, Rs, Cs;
  float T, w_kl, w_sup, w_ce;
  float* out;      // [4]: total, kl (kd_loss value), sup (inter_class_relation value), ce (mean CE)
  float* g_kl; float* g_ce; float* g_sup;
};

__global__ void d2m_loss_kernel(D2MArgs a) {
  __shared__ float red[4];
  float kl = 0.f, ce = 0.f, cs = 0.f;
  float g[LO_MAXC];
  if (a.s_kl) {
    for (int r = threadIdx.x; r < a.Rq; r += LO_THREADS) {
      kl += kd_row(a.s_kl + (long)r * a.C, a.t_kl + (long)r * a.C, a.C, a.T, a.g_kl ? g : nullptr);
      if (a.g_kl) for (int c = 0; c < a.C; ++c) a.g_kl[(long)r * a.C + c] = a.w_kl * (a.T / (float)a.Rq) * g[c];
    }
  }
  if (a.s_ce) {
    for (int r = threadIdx.x; r < a.Rq; r += LO_THREADS) {
      ce += ce_row(a.s_ce + (long)r * a.C, a.labels[r], a.C, a.g_ce ? g : nullptr);
      if (a.g_ce) for (int c = 0; c < a.C; ++c) a.g_ce[(long)r * a.C + c] = a.w_ce / (float)a.Rq * g[c];
    }
  }
  if (a.s_sup) {
    for (int r = threadIdx.x; r < a.Rs; r += LO_THREADS) {
      cs += icr_row(a.s_sup + (long)r * a.Cs, a.t_sup + (long)r * a.Cs, a.Cs, a.g_sup ? g : nullptr);
      if (a.g_sup) for (int c = 0; c < a.Cs; ++c) a.g_sup[(long)r * a.Cs + c] = -a.w_sup / (float)a.Rs * g[c];
    }
  }
  kl = block_sum_256(kl, red);
  ce = block_sum_256(ce, red);
  cs = block_sum_256(cs, red);
  if (threadIdx.x == 0) {
    const float v_kl = a.s_kl ? kl / (float)a.Rq * a.T * a.T : 0.f;
    const float v_ce = a.s_ce ? ce / (float)a.Rq : 0.f;
    const float v_sup = a.s_sup ? 1.f - cs / (float)a.Rs : 0.f;
    a.out[0] = a.w_kl * v_kl + a.w_sup * v_sup + a.w_ce * v_ce;
    a.out[1] = v_kl;
    a.out[2] = v_sup;
    a.out[3] = v_ce;
  }
}

// loss = w_kl*kd_loss(s_kl,t_kl,T) + w_sup*inter_class_relation(s_sup,t_sup) + w_ce*CE(s_ce,labels).
// Any of the three terms is skipped when its student pointer is null.  Gradients (d loss / d logits)
// are written when the g_* pointer is non-null.
extern "C" int lmkd_d2m_loss(const float* s_kl, const float* t_kl, const float* s_ce, const long long* labels, const float* s_sup,
                             const float* t_sup, int Rq, int C, int Rs, int Cs, float T, float w_kl, float w_sup, float w_ce,
                             float* out4, float* g_kl, float* g_ce, float* g_sup, void* stream) {
  LMKD_REQUIRE(out4, "lmkd_d2m_loss: out4 is null");
  LMKD_REQUIRE(!s_kl || (t_kl && Rq > 0 && C > 0 && C <= LO_MAXC && T > 0.f), "lmkd_d2m_loss: bad KL arguments");
  LMKD_REQUIRE(!s_ce || (labels && Rq > 0 && C > 0 && C <= LO_MAXC), "lmkd_d2m_loss: bad CE arguments");
  LMKD_REQUIRE(!s_sup || (t_sup && Rs > 0 && Cs > 0 && Cs <= LO_MAXC), "lmkd_d2m_loss: bad relation arguments");
  D2MArgs a;
  a.s_kl = s_kl; a.t_kl = t_kl; a.s_ce = s_ce; a.labels = labels; a.s_sup = s_sup; a.t_sup = t_sup;
  a.Rq = Rq; a.C = C; a.Rs = Rs; a.Cs = Cs; a.T = T; a.w_kl = w_kl; a.w_sup = w_sup; a.w_ce = w_ce;
  a.out = out4; a.g_kl = g_kl; a.g_ce = g_ce; a.g_sup = g_sup;
  hipLaunchKernelGGL(d2m_loss_kernel, dim3(1), dim3(LO_THREADS), 0, (hipStream_t)stream, a);
  LMKD_CHECK_LAUNCH("d2m_loss_kernel");
  return LMKD_OK;
}

// accuracy: pred = argmax(l1 (+ l2)) (first maximal index), acc = mean(pred == label)
__global__ void accuracy_kernel(const float* __restrict__ l1, const float* __restrict__ l2, const long long* __restrict__ labels,
                                long long* __restrict__ pred, float* __restrict__ acc, int R, int C) {
  __shared__ float red[4];
  float hit = 0.f;
  for (int r = threadIdx.x; r < R; r += LO_THREADS) {
    int am = 0;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) {
      const float v = l1[(long)r * C + c] + (l2 ? l2[(long)r * C + c] : 0.f);
      if (v > m) { m = v; am = c; }
    }
    if (pred) pred[r] = am;
    hit += (labels[r] == am) ? 1.f : 0.f;
  }
  hit = block_sum_256(hit, red);
  if (threadIdx.x == 0) acc[0] = hit / (float)R;
}
extern "C" int lmkd_accuracy(const float* l1, const float* l2, const long long* labels, long long* pred, float* acc, int R, int C,
                             void* stream) {
  LMKD_REQUIRE(l1 && labels && acc && R > 0 && C > 0, "lmkd_accuracy: bad arguments");
  hipLaunchKernelGGL(accuracy_kernel, dim3(1), dim3(LO_THREADS), 0, (hipStream_t)stream, l1, l2, labels, pred, acc, R, C);
  LMKD_CHECK_LAUNCH("accuracy_kernel");
  return LMKD_OK;
}

// ---------------------------------------------------------------------------------
// F.mse_loss(student_feature, teacher_feature) of Distiller.KL_feature (distillers.py:138): mean squared difference and its
// gradient 2 (s - t) / n wrt the student, one pass; deterministic two-level reduction (per-workgroup partials, then one workgroup)
// ---------------------------------------------------------------------------------
__global__ void mse_partial_kernel(const float4* __restrict__ s, const float4* __restrict__ t, float4* __restrict__ grad, long n4,
                                   float gscale, double* __restrict__ partial) {
  __shared__ double red[LO_THREADS / 64];
  double acc = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 a = s[i], b = t[i];
    const float4 d = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    acc += (double)d.x * d.x + (double)d.y * d.y + (double)d.z * d.z + (double)d.w * d.w;
    if (grad) grad[i] = make_float4(gscale * d.x, gscale * d.y, gscale * d.z, gscale * d.w);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
    for (int w = 0; w < LO_THREADS / 64; ++w) v += red[w];
    partial[blockIdx.x] = v;
  }
}
__global__ void mse_finish_kernel(const double* __restrict__ partial, int nb, double inv_n, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double v = 0.0;
    for (int b = 0; b < nb; ++b) v += partial[b];
    out[0] = (float)(v * inv_n);
  }
}
extern "C" long lmkd_mse_loss_workspace(void) { return 1024 * (long)sizeof(double); }
extern "C" int lmkd_mse_loss(const float* s, const float* t, long n, float* out1, float* grad, void* workspace, void* stream) {
  LMKD_REQUIRE(s && t && out1 && workspace && n > 0 && n % 4 == 0 && aligned16(s) && aligned16(t) && (!grad || aligned16(grad)),
               "lmkd_mse_loss: n %% 4 == 0 and 16-byte aligned buffers required");
  long gsz = (n / 4 + LO_THREADS - 1) / LO_THREADS;
  if (gsz > 1024) gsz = 1024;
  hipLaunchKernelGGL(mse_partial_kernel, dim3((unsigned)gsz), dim3(LO_THREADS), 0, (hipStream_t)stream, (const float4*)s, (const float4*)t,
                     (float4*)grad, n / 4, 2.f / (float)n, (double*)workspace);
  hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)workspace, (int)gsz, 1.0 / (double)n, out1);
  LMKD_CHECK_LAUNCH("mse_loss kernels");
  return LMKD_OK;
}

// ---------------------------------------------------------------------------------
// optimizers over the flat parameter / gradient buffers
// ---------------------------------------------------------------------------------
__global__ void sgd_kernel(float4* __restrict__ p, const float4* __restrict__ g, float lr, long n4, int zero_grad, float4* gz) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 w = p[i];
    const float4 d = g[i];
    w.x -= lr * d.x; w.y -= lr * d.y; w.z -= lr * d.z; w.w -= lr * d.w;
    p[i] = w;
    if (zero_grad) gz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
extern "C" int lmkd_sgd_step(float* param, float* grad, float lr, long n, int zero_grad, void* stream) {
  LMKD_REQUIRE(param && grad && n > 0 && n % 4 == 0 && aligned16(param) && aligned16(grad), "lmkd_sgd_step: buffers must be 16-byte aligned, n %% 4 == 0");
  long gsz = (n / 4 + LO_THREADS - 1) / LO_THREADS;
  if (gsz > 2048) gsz = 2048;
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)gsz), dim3(LO_THREADS), 0, (hipStream_t)stream, (float4*)param, (const float4*)grad, lr, n / 4,
                     zero_grad, (float4*)grad);
  LMKD_CHECK_LAUNCH("sgd_kernel");
  return LMKD_OK;
}

__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, float lr, float b1,
                            float b2, float eps, float bc1, float bc2_sqrt, long n, int zero_grad) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    if (zero_grad) g[i] = 0.f;
  }
}
// torch.optim.Adam defaults (no amsgrad, no weight decay); step is the 1-based step count
extern "C" int lmkd_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                              long step, long n, int zero_grad, void* stream) {
  LMKD_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step > 0, "lmkd_adam_step: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  long gsz = (n + LO_THREADS - 1) / LO_THREADS;
  if (gsz > 2048) gsz = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)gsz), dim3(LO_THREADS), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, lr, beta1,
                     beta2, eps, bc1, bc2s, n, zero_grad);
  LMKD_CHECK_LAUNCH("adam_kernel");
  return LMKD_OK;
}

__global__ void fill_kernel(float* __restrict__ p, float v, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
extern "C" int lmkd_fill(float* p, float v, long n, void* stream) {
  LMKD_REQUIRE(p && n > 0, "lmkd_fill: bad arguments");
  long gsz = (n + LO_THREADS - 1) / LO_THREADS;
  if (gsz > 2048) gsz = 2048;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)gsz), dim3(LO_THREADS), 0, (hipStream_t)stream, p, v, n);
  LMKD_CHECK_LAUNCH("fill_kernel");
  return LMKD_OK;
}
