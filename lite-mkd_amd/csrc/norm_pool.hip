// HBM-bound kernels around the convolutions: layout change, train-mode BatchNorm
// (finalize / apply / backward), the stem's BN+ReLU+maxpool, the adaptive-max 4x4 + patch
// mean head, column reductions (bias / LayerNorm parameter grads).  All NHWC, float4 per
// lane along the channel dimension, 256-thread workgroups, grid-stride, deterministic
// two-level reductions (no float atomics).
#include "common.h"
#include <vector>
#include <cmath>
#include <algorithm>

#define NP_THREADS 256
static int g_ew_wg_per_cu = 4;
extern "C" int lmkd_set_elementwise_wg_per_cu(int n) {
  LMKD_REQUIRE(n >= 1 && n <= 8, "lmkd_set_elementwise_wg_per_cu: %d outside 1..8 (the BatchNorm partial buffers hold 2048 rows)", n);
  g_ew_wg_per_cu = n;
  return LMKD_OK;
}
static inline int ew_grid(long n_items) {
  long g = (n_items + NP_THREADS - 1) / NP_THREADS;
  if (g > 256 * g_ew_wg_per_cu) g = 256 * g_ew_wg_per_cu;   // default 4: 4 workgroups (16 waves) per CU saturate HBM and leave the other 16 wave slots to a co-running MFMA kernel
  if (g < 1) g = 1;
  return (int)g;
}

// Compute mode 4 (two fp16 planes): the kernels that WRITE a trunk tensor - lmkd_bn_apply_seg, lmkd_bn_relu_maxpool_fwd_seg,
// lmkd_bn_backward_seg, lmkd_bn_backward_part_seg, lmkd_stem_unpool_bn_bwd(_seg), lmkd_nchw3_to_nhwc4 - also fold max |result| into the
// words lmkd_amax_desc::out_words names: 2 frame segments x LMKD_AMAX_SLOTS slots x 16 words (common.h: amax_commit; segment 1's elements
// go to the second half: the two trunk calls of an episode keep the scales they would have as two launches), fp32 bits, atomic max; the
// caller zeroes all lmkd_amax_words() of them.  The two-plane convolutions scale their operands by a power of two taken from it
// (lmkd_amax_desc::x_words / dy_words).  With lmkd_amax_desc::ref_words (the words the same tensor had in the previous episode) they also
// leave the range statistics of common.h (amax_commit_stat) beside the maximum: the range fence, lmkd_h2_fence_eval.
extern "C" long lmkd_amax_words(void) { return 2 * LMKD_AMAX_SEG_WORDS; }
static inline unsigned* amd_out(const lmkd_amax_desc* d) { return d ? (unsigned*)d->out_words : nullptr; }
static inline const unsigned* amd_ref(const lmkd_amax_desc* d) { return (d && d->out_words) ? (const unsigned*)d->ref_words : nullptr; }

// The range fence of compute mode 4.  entry e: words[e] = the words a tensor's producer has just written with lmkd_amax_desc::ref_words =
// refs[e] (its maximum + range statistics, common.h amax_commit_stat); refs[e] = the PERSISTENT reference words of that tensor's site (one
// set per BatchNorm output / gradient of the network, owned by the caller, zero before the first episode).  flags[e] bit s is set when
// frame segment s of the tensor has too much of its magnitude in elements the two-plane split does not resolve fully: 4 n_small > Q, i.e.
// the under-resolved elements would add more than a quarter to the rounding error of the resolved ones - judged only where the reference
// was adequate (current maximum >= reference / 4: against a stale, much larger reference everything looks small; such a launch is simply
// not judged).  Then the reference becomes this episode's maximum (slot 0 of each segment; the other slots stay zero), for the next
// episode's producer.  The caller stops naming a flagged tensor's maximum to the convolutions (LMKD_AMAX_FENCED): they run the three-plane
// form, which has no range to speak of.  One wave per entry.
#define LMKD_FENCE_MAX 128
struct FenceArgs { const unsigned* w[LMKD_FENCE_MAX]; unsigned* r[LMKD_FENCE_MAX]; };
__global__ void h2_fence_eval_kernel(FenceArgs a, int* __restrict__ flags) {
  const unsigned* w = a.w[blockIdx.x];
  unsigned* r = a.r[blockIdx.x];
  int f = 0;
  for (int seg = 0; seg < 2; ++seg) {
    const unsigned* slot = w + seg * LMKD_AMAX_SEG_WORDS + (threadIdx.x & (LMKD_AMAX_SLOTS - 1)) * LMKD_AMAX_STRIDE;
    unsigned long long ns = slot[1], q = *reinterpret_cast<const unsigned long long*>(slot + 2);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      ns += (unsigned long long)__shfl_xor((long long)ns, o, 64);
      q += (unsigned long long)__shfl_xor((long long)q, o, 64);
    }
    const unsigned cb = amax_read(w, seg), rb = amax_read(r, seg);
    const float cur = __uint_as_float(cb), ref = __uint_as_float(rb);
    if (ref > 0.f && cur >= 0.25f * ref && ns > 0 && 4 * ns > q) f |= 1 << seg;
    if (threadIdx.x == 0) r[seg * LMKD_AMAX_SEG_WORDS] = cb;
  }
  if (threadIdx.x == 0) flags[blockIdx.x] = f;
}
// words / refs: HOST arrays of n device pointers (lmkd_amax_words() words each); flags: n device ints
extern "C" int lmkd_h2_fence_eval(const void* const* words, void* const* refs, int n, int* flags, void* stream) {
  LMKD_REQUIRE(words && refs && flags && n > 0, "lmkd_h2_fence_eval: bad arguments");
  for (int i0 = 0; i0 < n; i0 += LMKD_FENCE_MAX) {
    FenceArgs a;
    const int m = std::min(LMKD_FENCE_MAX, n - i0);
    for (int j = 0; j < m; ++j) {
      LMKD_REQUIRE(words[i0 + j] && refs[i0 + j], "lmkd_h2_fence_eval: null entry %d", i0 + j);
      a.w[j] = (const unsigned*)words[i0 + j];
      a.r[j] = (unsigned*)refs[i0 + j];
    }
    for (int j = m; j < LMKD_FENCE_MAX; ++j) { a.w[j] = nullptr; a.r[j] = nullptr; }
    hipLaunchKernelGGL(h2_fence_eval_kernel, dim3(m), dim3(64), 0, (hipStream_t)stream, a, flags + i0);
    LMKD_CHECK_LAUNCH("h2_fence_eval_kernel");
  }
  return LMKD_OK;
}

// ---------------------------------------------------------------------------------
// [N,3,H,W] -> [N,H,W,4] (4th channel zero): the stem conv runs on NHWC4
// ---------------------------------------------------------------------------------
__global__ void nchw3_to_nhwc4_kernel(const float* __restrict__ x, float4* __restrict__ y, long npix_total, long hw, unsigned* __restrict__ amax) {
  float am = 0.f;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix_total; p += (long)gridDim.x * blockDim.x) {
    const long n = p / hw, r = p - n * hw;
    const float* b = x + n * 3 * hw + r;
    const float4 v = make_float4(b[0], b[hw], b[2 * hw], 0.f);
    y[p] = v;
    am = amax4(am, v);
  }
  if (amax) amax_commit(amax, am);      // (one frame segment per call: the slots at `amax`)
}

extern "C" int lmkd_nchw3_to_nhwc4(const float* x, float* y, int N, int H, int W, void* stream, void* amax_words) {
  unsigned* amax = (unsigned*)amax_words;      // nullable: fold max |y| into the slots of ONE frame segment at this address
  LMKD_REQUIRE(x && y && N > 0 && H > 0 && W > 0, "lmkd_nchw3_to_nhwc4: bad arguments");
  LMKD_REQUIRE(aligned16(y), "lmkd_nchw3_to_nhwc4: output must be 16-byte aligned");
  const long hw = (long)H * W, total = hw * N;
  hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(ew_grid(total)), dim3(NP_THREADS), 0, (hipStream_t)stream, x, (float4*)y, total, hw, amax);
  LMKD_CHECK_LAUNCH("lmkd_nchw3_to_nhwc4");
  return LMKD_OK;
}

// ---------------------------------------------------------------------------------
// GPU side of the frame transform (video_reader.py:92-112 after the PIL Resize): per-video crop + horizontal flip +
// ToTensor (/255) of uint8 HWC frames, written straight in the stem's NHWC4 layout.
//   src [F, Hs, Ws, 3] uint8 ; crop_y/crop_x/flip: one entry per video (frames_per_video consecutive frames share it)
// ---------------------------------------------------------------------------------
__global__ void frames_u8_to_nhwc4_kernel(const unsigned char* __restrict__ src, float4* __restrict__ dst, const int* __restrict__ crop_y,
                                          const int* __restrict__ crop_x, const int* __restrict__ flip, long total, int Hs, int Ws,
                                          int H, int W, int frames_per_video) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const int w = (int)(p % W);
    long r = p / W;
    const int h = (int)(r % H);
    const int f = (int)(r / H);
    const int v = f / frames_per_video;
    const int sx = crop_x[v] + (flip[v] ? (W - 1 - w) : w);
    const int sy = crop_y[v] + h;
    const unsigned char* s = src + (((long)f * Hs + sy) * Ws + sx) * 3;
    dst[p] = make_float4((float)s[0] / 255.f, (float)s[1] / 255.f, (float)s[2] / 255.f, 0.f);   // ToTensor: x.div(255)
  }
}

extern "C" int lmkd_frames_u8_to_nhwc4(const unsigned char* src, float* dst, const int* crop_y, const int* crop_x, const int* flip,
                                       int F, int Hs, int Ws, int H, int W, int frames_per_video, void* stream) {
  LMKD_REQUIRE(src && dst && crop_y && crop_x && flip && F > 0 && H > 0 && W > 0 && H <= Hs && W <= Ws && frames_per_video > 0,
               "lmkd_frames_u8_to_nhwc4: bad arguments");
  const long total = (long)F * H * W;
  hipLaunchKernelGGL(frames_u8_to_nhwc4_kernel, dim3(ew_grid(total)), dim3(NP_THREADS), 0, (hipStream_t)stream, src, (float4*)dst, crop_y,
                     crop_x, flip, total, Hs, Ws, H, W, frames_per_video);
  LMKD_CHECK_LAUNCH("frames_u8_to_nhwc4_kernel");
  return LMKD_OK;
}

// ---------------------------------------------------------------------------------
// Resize(256) of the frame transform (video_reader.py:92-112 -> functional.resize_clip, functional.py:44-59, which calls
// PIL.Image.resize(size, BILINEAR)): Pillow's 8-bit resampler, bit exact.  Triangle filter of support max(scale, 1),
// coefficients normalised in double and rounded to 22-bit fixed point (host, lmkd_resize_plan), then one pass per axis
//   out = clip8((2^21 + sum_x pixel[x] * k[x]) >> 22),  horizontal pass first, uint8 in between - exactly Pillow's order.
// A pass sees the tensor as [outer][n_in][inner] uint8 -> [outer][n_out][inner]  (horizontal: inner = C; vertical: inner = W*C).
// ---------------------------------------------------------------------------------
#define LMKD_PIL_PRECISION_BITS 22

extern "C" int lmkd_resize_plan(int in_size, int out_size, int* bounds /*[out][2] or null*/, int* coeffs /*[out][ksize] or null*/) {
  LMKD_REQUIRE(in_size > 0 && out_size > 0, "lmkd_resize_plan: sizes must be positive");
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  if (!bounds || !coeffs) return ksize;
  const double ss = 1.0 / filterscale;
  std::vector<double> w(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < ksize; ++x) w[x] = 0.0;
    for (int x = 0; x < xmax; ++x) {
      double a = (x + xmin - center + 0.5) * ss;
      if (a < 0.0) a = -a;
      w[x] = a < 1.0 ? 1.0 - a : 0.0;
      ww += w[x];
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) w[x] /= ww;
    for (int x = 0; x < ksize; ++x)
      coeffs[xx * ksize + x] = w[x] < 0 ? (int)(-0.5 + w[x] * (1 << LMKD_PIL_PRECISION_BITS)) : (int)(0.5 + w[x] * (1 << LMKD_PIL_PRECISION_BITS));
    bounds[xx * 2] = xmin;
    bounds[xx * 2 + 1] = xmax;
  }
  return ksize;
}

__global__ void resize_pass_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, const int* __restrict__ bounds,
                                      const int* __restrict__ coeffs, int ksize, long total, int n_in, int n_out, long inner) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const long i = p % inner;
    const long r = p / inner;
    const int xx = (int)(r % n_out);
    const long o = r / n_out;
    const int x0 = bounds[xx * 2], n = bounds[xx * 2 + 1];
    const unsigned char* s = src + (o * n_in + x0) * inner + i;
    const int* k = coeffs + (long)xx * ksize;
    int acc = 1 << (LMKD_PIL_PRECISION_BITS - 1);
    for (int x = 0; x < n; ++x) acc += (int)s[(long)x * inner] * k[x];
    acc >>= LMKD_PIL_PRECISION_BITS;
    dst[p] = (unsigned char)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
  }
}

extern "C" int lmkd_resize_pass_u8(const unsigned char* src, unsigned char* dst, const int* bounds_dev, const int* coeffs_dev, int ksize,
                                   long outer, int n_in, int n_out, long inner, void* stream) {
  LMKD_REQUIRE(src && dst && bounds_dev && coeffs_dev && ksize > 0 && outer > 0 && n_in > 0 && n_out > 0 && inner > 0,
               "lmkd_resize_pass_u8: bad arguments");
  const long total = outer * n_out * inner;
  hipLaunchKernelGGL(resize_pass_u8_kernel, dim3(ew_grid(total)), dim3(NP_THREADS), 0, (hipStream_t)stream, src, dst, bounds_dev, coeffs_dev,
                     ksize, total, n_in, n_out, inner);
  LMKD_CHECK_LAUNCH("resize_pass_u8_kernel");
  return LMKD_OK;
}

// The whole frame transform of an episode in ONE launch (round 5): Resize -> crop -> flip -> ToTensor for frames that share one resolution.
// A thread owns one pixel of the cropped 224 x 224 output and evaluates Pillow's two passes for it alone: the vertical pass over the (at
// most ksize_v) rows of the horizontally resized image it needs, each of those values the horizontal pass at its own column - with the
// uint8 rounding and clipping BETWEEN the passes that Pillow's intermediate image has, so the result is the two-pass one bit for bit
// (tests/test_gpu_ops.py) - and writes the ToTensor value.  Nothing is computed for the 43 % of the resized frame the crop discards and
// neither intermediate image exists: 1.08 ms of a streamed episode's copy-stream work (two passes + crop kernel, a 64-bit division pair
// per output BYTE in the passes) -> see profiles/r05_frame_transform.txt.
__global__ void frames_resize_crop_nhwc4_kernel(const unsigned char* __restrict__ src, float4* __restrict__ dst,
                                                const int* __restrict__ bh, const int* __restrict__ kh, int ksh,
                                                const int* __restrict__ bv, const int* __restrict__ kv, int ksv,
                                                const int* __restrict__ crop_y, const int* __restrict__ crop_x, const int* __restrict__ flip,
                                                int Hs, int Ws, int H, int W, int frames_per_video) {
  const int f = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= H * W) return;
  const int h = p / W, w = p - h * W;
  const int v = f / frames_per_video;
  const int sx = crop_x[v] + (flip[v] ? (W - 1 - w) : w);      // column / row of the resized frame
  const int sy = crop_y[v] + h;
  const int x0 = bh[sx * 2], nx = bh[sx * 2 + 1];
  const int y0 = bv[sy * 2], ny = bv[sy * 2 + 1];
  const int* ch = kh + (long)sx * ksh;
  const int* cv = kv + (long)sy * ksv;
  int a0 = 1 << (LMKD_PIL_PRECISION_BITS - 1), a1 = a0, a2 = a0;
  for (int yy = 0; yy < ny; ++yy) {
    const unsigned char* row = src + (((long)f * Hs + y0 + yy) * Ws + x0) * 3;
    int t0 = 1 << (LMKD_PIL_PRECISION_BITS - 1), t1 = t0, t2 = t0;
    for (int xx = 0; xx < nx; ++xx) {
      const int k = ch[xx];
      t0 += (int)row[xx * 3 + 0] * k;
      t1 += (int)row[xx * 3 + 1] * k;
      t2 += (int)row[xx * 3 + 2] * k;
    }
    t0 >>= LMKD_PIL_PRECISION_BITS; t1 >>= LMKD_PIL_PRECISION_BITS; t2 >>= LMKD_PIL_PRECISION_BITS;
    t0 = t0 < 0 ? 0 : (t0 > 255 ? 255 : t0); t1 = t1 < 0 ? 0 : (t1 > 255 ? 255 : t1); t2 = t2 < 0 ? 0 : (t2 > 255 ? 255 : t2);
    const int k = cv[yy];
    a0 += t0 * k; a1 += t1 * k; a2 += t2 * k;
  }
  a0 >>= LMKD_PIL_PRECISION_BITS; a1 >>= LMKD_PIL_PRECISION_BITS; a2 >>= LMKD_PIL_PRECISION_BITS;
  a0 = a0 < 0 ? 0 : (a0 > 255 ? 255 : a0); a1 = a1 < 0 ? 0 : (a1 > 255 ? 255 : a1); a2 = a2 < 0 ? 0 : (a2 > 255 ? 255 : a2);
  dst[(long)f * H * W + p] = make_float4((float)a0 / 255.f, (float)a1 / 255.f, (float)a2 / 255.f, 0.f);   // ToTensor: x.div(255)
}

extern "C" int lmkd_frames_resize_crop_nhwc4(const unsigned char* src, float* dst, const int* bounds_h, const int* coeffs_h, int ksize_h,
                                             const int* bounds_v, const int* coeffs_v, int ksize_v, const int* crop_y, const int* crop_x,
                                             const int* flip, int F, int Hs, int Ws, int Hr, int Wr, int H, int W, int frames_per_video,
                                             void* stream) {
  LMKD_REQUIRE(src && dst && bounds_h && coeffs_h && bounds_v && coeffs_v && crop_y && crop_x && flip && ksize_h > 0 && ksize_v > 0 && F > 0 &&
               F <= 65535 && Hs > 0 && Ws > 0 && H > 0 && W > 0 && H <= Hr && W <= Wr && frames_per_video > 0, "lmkd_frames_resize_crop_nhwc4: bad arguments");
  hipLaunchKernelGGL(frames_resize_crop_nhwc4_kernel, dim3(cdiv(H * W, NP_THREADS), F), dim3(NP_THREADS), 0, (hipStream_t)stream, src,
                     (float4*)dst, bounds_h, coeffs_h, ksize_h, bounds_v, coeffs_v, ksize_v, crop_y, crop_x, flip, Hs, Ws, H, W, frames_per_video);
  LMKD_CHECK_LAUNCH("frames_resize_crop_nhwc4_kernel");
  return LMKD_OK;
}

// ---------------------------------------------------------------------------------
// Column sums of a tall partial-sum buffer in ONE launch (round 3; before: a 64-slice stage-1 kernel for buffers taller than
// 2048 rows plus a finishing kernel in which ONE workgroup per 64 channels walked up to 2048 rows: 11 - 19 us per BatchNorm).
// grid = (column groups of CS_COLS, S row slices), block = CS_COLS columns x CS_LANES row lanes.  Every block sums its slice of
// rows for its columns in fp64 (fixed order: row lane q takes rows q, q + 8, ... with four interleaved accumulators, lanes added
// in lane order) and writes one fp64 row of scratch[S][CV]; a ticket counter per column group tells the block that arrives last,
// which adds the S slices in slice order - the result does not depend on the arrival order - and goes on to the consumer's
// arithmetic (BatchNorm finalize / backward coefficients).  tickets[] must be zero on entry and is zero again on exit; launches
// that may run concurrently (two HIP streams) need separate ticket and scratch buffers.
// ---------------------------------------------------------------------------------
#define CS_COLS 32
#define CS_LANES 8
#define CS_MAX_SLICES 64
#define LMKD_TICKET_WORDS 512      // two segments (blockIdx.z) x cdiv(2 C, CS_COLS) words: C <= 4096
extern "C" long lmkd_ticket_words(void) { return LMKD_TICKET_WORDS; }

static inline int cs_slices(int T) { return T <= 64 ? 1 : std::min(CS_MAX_SLICES, cdiv(T, 64)); }

// -> true in the block that holds the totals of its CS_COLS columns in tot[] (every thread of the block gets the same answer)
template <bool PROD = false>      // PROD: sum in[r][c] * in2[r][c] (the float product, as a separate multiply)
__device__ __forceinline__ bool colsum_ticket(const float* __restrict__ in, int T, int CV, double* __restrict__ scratch,
                                              unsigned* __restrict__ tickets, double (*sm)[CS_COLS], double* tot, int* s_last,
                                              const float* __restrict__ in2 = nullptr, int S_ = 0) {
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int col = blockIdx.x * CS_COLS + tx;
  const bool act = col < CV;
  const int S = S_ > 0 ? S_ : gridDim.y;      // S_: this segment's slice count (<= gridDim.y); blocks past it have nothing to do
  if ((int)blockIdx.y >= S) return false;
  const int per = (T + S - 1) / S;
  const int t0 = blockIdx.y * per;
  const int t1 = t0 + per < T ? t0 + per : T;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (act) {
    int t = t0 + ty;
    auto at = [&](int r) -> double {
      const long i = (long)r * CV + col;
      return PROD ? (double)(in[i] * in2[i]) : (double)in[i];
    };
    for (; t + 3 * CS_LANES < t1; t += 4 * CS_LANES) {
      s0 += at(t);
      s1 += at(t + CS_LANES);
      s2 += at(t + 2 * CS_LANES);
      s3 += at(t + 3 * CS_LANES);
    }
    for (; t < t1; t += CS_LANES) s0 += at(t);
  }
  sm[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  double v = 0.0;
  if (ty == 0)
    for (int q = 0; q < CS_LANES; ++q) v += sm[q][tx];
  if (S == 1) {
    if (ty == 0) tot[tx] = v;
    __syncthreads();
    return true;
  }
  // Hand-off WITHOUT device-scope fences.  __threadfence() = L2 write-back (release) / L2 invalidate (acquire) of the whole XCD: 15 us
  // per launch here and - worse - it evicts what the convolution kernels running beside this one on the other streams keep in L2
  // (measured: the serial kernel sum fell by 1.4 ms per episode while the three-stream wall time ROSE by 0.9 ms).  Instead the slice row
  // is written through to device-coherent memory (relaxed agent-scope atomic store = an sc1 store), the wave waits until those stores
  // have completed, and only then takes its ticket; the finishing block reads the rows with relaxed agent-scope atomic loads, which
  // bypass the non-coherent caches.
  if (ty == 0 && act) __hip_atomic_store(&scratch[(long)blockIdx.y * CV + col], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tx == 0 && ty == 0) *s_last = atomicAdd(&tickets[blockIdx.x], 1u) == (unsigned)(S - 1);
  __syncthreads();
  if (!*s_last) return false;
  asm volatile("" ::: "memory");
  // row lane q adds the slices q, q + 8, ... (independent loads: in flight together), the lanes are added in lane order
  {
    double a0 = 0.0, a1 = 0.0;
    if (act) {
      const double* sp = scratch + col;
      int q = ty;
      for (; q + CS_LANES < S; q += 2 * CS_LANES) {
        a0 += __hip_atomic_load(&sp[(long)q * CV], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a1 += __hip_atomic_load(&sp[(long)(q + CS_LANES) * CV], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (q < S) a0 += __hip_atomic_load(&sp[(long)q * CV], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    sm[ty][tx] = a0 + a1;
  }
  __syncthreads();
  if (ty == 0) {
    double a = 0.0;
    for (int q = 0; q < CS_LANES; ++q) a += sm[q][tx];
    tot[tx] = a;
  }
  if (tx == 0 && ty == 0) tickets[blockIdx.x] = 0u;   // zero again for the next launch that uses this ticket buffer
  __syncthreads();
  return true;
}

// ---------------------------------------------------------------------------------
// BatchNorm finalize: sums -> mean / invstd / fused scale+shift, running-stat update
//   stats layout out: [5][C] = mean, invstd, scale (= gamma*invstd), shift (= beta - mean*scale), unbiased variance
// ---------------------------------------------------------------------------------
// Two frame segments (blockIdx.z = 1: the second trunk call of an episode, lmkd_bn_finalize_seg): rows [T0, T) of `part`, count1, the
// second [5][C] table, scratch and ticket words of its own; the slices of a segment are those a launch of its own would use.
__global__ void bn_finalize_kernel(const float* __restrict__ part, int T, int C, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float momentum, float eps, float* __restrict__ stats, double* __restrict__ scratch,
                                   unsigned* __restrict__ tickets, int T0 = 0, double count1 = 0.0, int S0 = 0, int S1 = 0,
                                   const unsigned* __restrict__ c_words = nullptr, unsigned* __restrict__ bound_words = nullptr,
                                   const unsigned* __restrict__ ref_words = nullptr) {
  __shared__ double sm[CS_LANES][CS_COLS];
  __shared__ double tot[CS_COLS];
  __shared__ int s_last;
  int S = 0;
  if (gridDim.z > 1) {
    if (blockIdx.z) {
      part += (long)T0 * 2 * C; T -= T0; count = count1; stats += 5 * C; scratch += (long)CS_MAX_SLICES * 2 * C; tickets += LMKD_TICKET_WORDS / 2; S = S1;
      if (c_words) { c_words += LMKD_AMAX_SEG_WORDS; bound_words += LMKD_AMAX_SEG_WORDS; }
      if (ref_words) ref_words += LMKD_AMAX_SEG_WORDS;
    } else { T = T0; S = S0; }
  }
  if (!colsum_ticket(part, T, 2 * C, scratch, tickets, sm, tot, &s_last, nullptr, S)) return;
  const int c = blockIdx.x * (CS_COLS / 2) + threadIdx.x;
  if (threadIdx.y != 0 || threadIdx.x >= CS_COLS / 2 || c >= C) return;
  const double s1 = tot[2 * threadIdx.x], s2 = tot[2 * threadIdx.x + 1];
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  stats[c] = (float)mean;
  stats[C + c] = invstd;
  stats[2 * C + c] = g * invstd;
  stats[3 * C + c] = b - (float)mean * g * invstd;
  const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
  stats[4 * C + c] = (float)unb;
  if (c_words) {
    // lmkd_amax_desc (x_words -> out_words): max over the elements of channel c of relu(x scale + shift) <= |scale| max |x| + |shift|, with max |x| over
    // the whole tensor (the convolution's epilogue recorded it: lmkd_amax_desc::out_words of its launch) - an upper bound of max |relu(BatchNorm(x))|, which
    // is all the two-plane kernels need from a maximum (a bound that is 2^k too large costs k of the 17 binades of full precision).
    // A recorded maximum of zero means "not recorded" (the convolution ran another kernel): the bound stays zero = unknown.
    unsigned gm = 0u;
    for (int s = 0; s < LMKD_AMAX_SLOTS; ++s) {
      const unsigned u = c_words[s * LMKD_AMAX_STRIDE];
      gm = u > gm ? u : gm;
    }
    if (gm) {
      const float bnd = fabsf(g * invstd) * __uint_as_float(gm) + fabsf(b - (float)mean * g * invstd);
      unsigned* slot = bound_words + (c & (LMKD_AMAX_SLOTS - 1)) * LMKD_AMAX_STRIDE;
      atomicMax(slot, __float_as_uint(bnd));
      if (ref_words) {
        // the range statistics of a tensor that is never written (common.h amax_commit_stat), per CHANNEL: the normalised activation of
        // channel c has unit variance times gamma around beta, so sqrt(gamma^2 + beta^2) is its typical magnitude; a channel whose typical
        // magnitude lies below 2^-17 of the (previous) bound is under-resolved as a whole
        unsigned rm = 0u;
        for (int q = 0; q < LMKD_AMAX_SLOTS; ++q) {
          const unsigned u = ref_words[q * LMKD_AMAX_STRIDE];
          rm = u > rm ? u : rm;
        }
        const AmaxRef rf = amax_ref(rm);
        const float typ = sqrtf(g * g + b * b);
        if (rf.thr > 0.f && typ > 0.f) {
          if (typ < rf.thr) atomicAdd(slot + 1, 1u);
          atomicAdd(reinterpret_cast<unsigned long long*>(slot + 2), (unsigned long long)(fminf(typ * rf.inv_thr, 1024.f) + 0.5f));
        }
      }
    }
  }
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

// deferred running-statistics update (two trunk calls overlapped on two streams update in program order afterwards):
//   running = (1-m)*running + m*batch  with batch mean = stats[0], unbiased variance = stats[4]
__global__ void bn_running_update_kernel(float* __restrict__ running_mean, float* __restrict__ running_var,
                                         const float* __restrict__ stats, int C, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * stats[c];
  running_var[c] = (1.f - momentum) * running_var[c] + momentum * stats[4 * C + c];
}
extern "C" int lmkd_bn_running_update(float* running_mean, float* running_var, const float* stats, int C, float momentum,
                                      void* stream) {
  LMKD_REQUIRE(running_mean && running_var && stats && C > 0, "lmkd_bn_running_update: bad arguments");
  hipLaunchKernelGGL(bn_running_update_kernel, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, running_mean, running_var,
                     stats, C, momentum);
  LMKD_CHECK_LAUNCH("bn_running_update_kernel");
  return LMKD_OK;
}

// All deferred running-statistics updates of an episode in ONE launch (round 3; before: one launch per BatchNorm and trunk call, 40 per
// episode).  Block b owns BatchNorm layer b and applies its (up to) two updates in order - support-frame call first, then the
// query-frame call, the order of the reference's two sequential trunk calls (resnet18_2fc.py:41-42).
#define BN_MULTI_MAX 64
struct BnMultiArgs {
  float* rm[BN_MULTI_MAX];
  float* rv[BN_MULTI_MAX];
  const float* st0[BN_MULTI_MAX];
  const float* st1[BN_MULTI_MAX];
  long long* nbt[BN_MULTI_MAX];      // num_batches_tracked of the layer (nullable): += one per update applied
  int C[BN_MULTI_MAX];
};
__global__ void bn_running_update_multi_kernel(BnMultiArgs a, float momentum) {
  const int b = blockIdx.x;
  const int C = a.C[b];
  if (threadIdx.x == 0 && a.nbt[b]) *a.nbt[b] += a.st1[b] ? 2 : 1;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float m = a.rm[b][c], v = a.rv[b][c];
    m = (1.f - momentum) * m + momentum * a.st0[b][c];
    v = (1.f - momentum) * v + momentum * a.st0[b][4 * C + c];
    if (a.st1[b]) {
      m = (1.f - momentum) * m + momentum * a.st1[b][c];
      v = (1.f - momentum) * v + momentum * a.st1[b][4 * C + c];
    }
    a.rm[b][c] = m;
    a.rv[b][c] = v;
  }
}
// rm / rv / stats_first / stats_second / C: HOST arrays of n entries (device pointers inside); stats_second[i] may be null
extern "C" int lmkd_bn_running_update_multi_nbt(float* const* running_mean, float* const* running_var, const float* const* stats_first,
                                                const float* const* stats_second, long long* const* num_batches_tracked, const int* C, int n,
                                                float momentum, void* stream);
extern "C" int lmkd_bn_running_update_multi(float* const* running_mean, float* const* running_var, const float* const* stats_first,
                                            const float* const* stats_second, const int* C, int n, float momentum, void* stream) {
  return lmkd_bn_running_update_multi_nbt(running_mean, running_var, stats_first, stats_second, nullptr, C, n, momentum, stream);
}
// the same + nn.BatchNorm2d's num_batches_tracked counters (int64 scalars; array or entries nullable): += 1 per update applied to the layer
extern "C" int lmkd_bn_running_update_multi_nbt(float* const* running_mean, float* const* running_var, const float* const* stats_first,
                                                const float* const* stats_second, long long* const* num_batches_tracked, const int* C, int n,
                                                float momentum, void* stream) {
  LMKD_REQUIRE(running_mean && running_var && stats_first && stats_second && C && n > 0, "lmkd_bn_running_update_multi: bad arguments");
  for (int i0 = 0; i0 < n; i0 += BN_MULTI_MAX) {
    BnMultiArgs a;
    memset(&a, 0, sizeof(a));
    const int m = std::min(BN_MULTI_MAX, n - i0);
    for (int i = 0; i < m; ++i) {
      LMKD_REQUIRE(running_mean[i0 + i] && running_var[i0 + i] && stats_first[i0 + i] && C[i0 + i] > 0, "lmkd_bn_running_update_multi: null entry %d", i0 + i);
      a.rm[i] = running_mean[i0 + i]; a.rv[i] = running_var[i0 + i];
      a.st0[i] = stats_first[i0 + i]; a.st1[i] = stats_second[i0 + i];
      a.nbt[i] = num_batches_tracked ? num_batches_tracked[i0 + i] : nullptr;
      a.C[i] = C[i0 + i];
    }
    hipLaunchKernelGGL(bn_running_update_multi_kernel, dim3(m), dim3(256), 0, (hipStream_t)stream, a, momentum);
    LMKD_CHECK_LAUNCH("bn_running_update_multi_kernel");
  }
  return LMKD_OK;
}

// eval mode: stats from running estimates
__global__ void bn_eval_stats_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ running_mean, const float* __restrict__ running_var, float eps,
                                     float* __restrict__ stats) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(running_var[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  stats[c] = running_mean[c];
  stats[C + c] = invstd;
  stats[2 * C + c] = g * invstd;
  stats[3 * C + c] = b - running_mean[c] * g * invstd;
  stats[4 * C + c] = running_var[c];
}

// lmkd_amax_desc on lmkd_bn_finalize(_seg) (mode 4): the launch also writes, into out_words (lmkd_amax_words() zeroed words), an upper
// bound of max |relu(BatchNorm(x))| per frame segment, from its scale / shift tables and max |x| in x_words (the words the convolution
// that wrote x recorded: lmkd_amax_desc::out_words of lmkd_conv2d_fwd_seg).  This is the maximum to name as x_words for a convolution
// that applies the BatchNorm + ReLU in its loader (pre_stats given).
// partial: [T][C][2] (sum, sumsq) from the conv epilogue; scratch: >= 64*2*C doubles; tickets: lmkd_ticket_words() zeroed words
extern "C" int lmkd_bn_finalize(const float* partial, int T, int C, long count, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, float* stats,
                                double* scratch, unsigned* tickets, void* stream, const lmkd_amax_desc* amd) {
  const unsigned* bin = amd ? (const unsigned*)amd->x_words : nullptr;
  unsigned* bout = amd ? (unsigned*)amd->out_words : nullptr;
  if (!bin || !bout) { bin = nullptr; bout = nullptr; }
  LMKD_REQUIRE(partial && stats && scratch && tickets && T > 0 && C > 0 && count > 0, "lmkd_bn_finalize: bad arguments");
  LMKD_REQUIRE(cdiv(2 * C, CS_COLS) <= LMKD_TICKET_WORDS, "lmkd_bn_finalize: C=%d exceeds the ticket buffer", C);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(2 * C, CS_COLS), cs_slices(T)), dim3(CS_COLS, CS_LANES), 0, (hipStream_t)stream, partial, T,
                     C, (double)count, gamma, beta, running_mean, running_var, momentum, eps, stats, scratch, tickets, 0, 0.0, 0, 0, bin, bout,
                     bout && amd ? (const unsigned*)amd->ref_words : nullptr);
  LMKD_CHECK_LAUNCH("bn_finalize_kernel");
  return LMKD_OK;
}

// lmkd_bn_finalize for two frame segments in one launch: partial rows [0, T0) / [T0, T) -> stats[0] / stats[1] ([2][5][C]) with
// count0 / count1 elements per channel; bit-identical to two lmkd_bn_finalize launches.  The running statistics are NOT updated here
// (two segments = two sequential updates: lmkd_bn_running_update_multi applies them in order).  scratch: >= 2 * 64 * 2 * C doubles.
extern "C" int lmkd_bn_finalize_seg(const float* partial, int T, int T0, int C, long count0, long count1, const float* gamma, const float* beta,
                                    float eps, float* stats, double* scratch, unsigned* tickets, void* stream, const lmkd_amax_desc* amd) {
  const unsigned* bin = amd ? (const unsigned*)amd->x_words : nullptr;
  unsigned* bout = amd ? (unsigned*)amd->out_words : nullptr;
  if (!bin || !bout) { bin = nullptr; bout = nullptr; }
  LMKD_REQUIRE(partial && stats && scratch && tickets && T0 > 0 && T > T0 && C > 0 && count0 > 0 && count1 > 0, "lmkd_bn_finalize_seg: bad arguments");
  LMKD_REQUIRE(cdiv(2 * C, CS_COLS) <= LMKD_TICKET_WORDS / 2, "lmkd_bn_finalize_seg: C=%d exceeds the ticket buffer", C);
  const int S0 = cs_slices(T0), S1 = cs_slices(T - T0);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(2 * C, CS_COLS), std::max(S0, S1), 2), dim3(CS_COLS, CS_LANES), 0, (hipStream_t)stream, partial, T,
                     C, (double)count0, gamma, beta, (float*)nullptr, (float*)nullptr, 0.f, eps, stats, scratch, tickets, T0, (double)count1, S0, S1, bin, bout,
                     bout && amd ? (const unsigned*)amd->ref_words : nullptr);
  LMKD_CHECK_LAUNCH("bn_finalize_kernel");
  return LMKD_OK;
}

extern "C" int lmkd_bn_eval_stats(int C, const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* stats, void* stream) {
  LMKD_REQUIRE(running_mean && running_var && stats && C > 0, "lmkd_bn_eval_stats: bad arguments");
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, C, gamma, beta, running_mean,
                     running_var, eps, stats);
  LMKD_CHECK_LAUNCH("bn_eval_stats_kernel");
  return LMKD_OK;
}

// ---------------------------------------------------------------------------------
// BN apply:  y = act( x*scale + shift  [+ r]  [+ r*rscale + rshift] )
// res_mode: 0 none, 1 plain residual, 2 residual with its own BN affine (downsample branch)
// ---------------------------------------------------------------------------------
// mask_bits (nullable, training): bit e of the array = (y[e] > 0) for flat element e, i.e. the ReLU mask the BatchNorm backward
// needs, at 1/32 of the bytes of y.  A lane owns U groups of 4 consecutive elements (U = 1 fp32, 2 bf16: 16 bytes either way) = U
// nibbles; 8 / U lanes assemble a word with shuffles (n4 % 8 == 0 because C % 32 == 0, so such a lane group is active or inactive
// as a whole).
template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ stats, const T* __restrict__ res,
                                const float* __restrict__ rstats, T* __restrict__ y, long n4, int C, int relu, int res_mode,
                                unsigned* __restrict__ mask_bits, long n4_0, unsigned* __restrict__ amax, const unsigned* __restrict__ aref) {
  // n4_0: groups of 4 elements in frame segment 0 (rows0 * C / 4; = n4 for one segment): elements past it use the second [5][C] table
  constexpr int U = ActU<T>::U;
  const int C4 = C >> 2;
  const long nu = n4 / U;
  AmaxStat am = {0.f, 0.f, 0u}, am1 = {0.f, 0.f, 0u};      // max |y| (+ range statistics) of segment 0 / segment 1
  const AmaxRef rf0 = amax_ref(aref ? amax_read(aref, 0) : 0u), rf1 = amax_ref(aref ? amax_read(aref, 1) : 0u);
  for (long iu = (long)blockIdx.x * blockDim.x + threadIdx.x; iu < nu; iu += (long)gridDim.x * blockDim.x) {
    float4 v[U], r[U];
    ldv<T, U>(x, iu, v);
    if (res_mode) ldv<T, U>(res, iu, r);
    unsigned nibs = 0;
    const int so = iu * U >= n4_0 ? 5 * C : 0;      // (a thread's U groups lie in one row: C4 is a multiple of U)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = iu * U + u;
      const int c = (int)(i % C4) * 4 + so;
      const float4 sc = *reinterpret_cast<const float4*>(stats + 2 * C + c);
      const float4 sh = *reinterpret_cast<const float4*>(stats + 3 * C + c);
      float4& w = v[u];
      w.x = fmaf(w.x, sc.x, sh.x); w.y = fmaf(w.y, sc.y, sh.y); w.z = fmaf(w.z, sc.z, sh.z); w.w = fmaf(w.w, sc.w, sh.w);
      if (res_mode == 1) {
        w.x += r[u].x; w.y += r[u].y; w.z += r[u].z; w.w += r[u].w;
      } else if (res_mode == 2) {
        const float4 rs = *reinterpret_cast<const float4*>(rstats + 2 * C + c);
        const float4 rh = *reinterpret_cast<const float4*>(rstats + 3 * C + c);
        w.x += fmaf(r[u].x, rs.x, rh.x); w.y += fmaf(r[u].y, rs.y, rh.y); w.z += fmaf(r[u].z, rs.z, rh.z); w.w += fmaf(r[u].w, rs.w, rh.w);
      }
      if (relu) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
      nibs |= ((w.x > 0.f ? 1u : 0u) | (w.y > 0.f ? 2u : 0u) | (w.z > 0.f ? 4u : 0u) | (w.w > 0.f ? 8u : 0u)) << (4 * u);
      if (so) amax_stat4(am1, w, rf1); else amax_stat4(am, w, rf0);
    }
    stv<T, U>(y, iu, v);      // bf16 storage: a positive value never rounds to zero, so the mask equals (stored y > 0)
    if (mask_bits) {
      constexpr int LPW = 8 / U;      // lanes per 32-bit word
      unsigned wbits = nibs << (4 * U * (threadIdx.x & (LPW - 1)));
      wbits |= __shfl_xor(wbits, 1, 64);
      wbits |= __shfl_xor(wbits, 2, 64);
      if (LPW == 8) wbits |= __shfl_xor(wbits, 4, 64);
      if ((threadIdx.x & (LPW - 1)) == 0) mask_bits[iu / LPW] = wbits;
    }
  }
  if (amax) {
    amax_commit_stat(amax, am, aref != nullptr);
    if (n4_0 < n4) amax_commit_stat(amax + LMKD_AMAX_SEG_WORDS, am1, aref != nullptr);
  }
}

extern "C" int lmkd_bn_apply_seg(const float* x, const float* stats, const float* res, const float* rstats, float* y, long rows, long rows0,
                                 int C, int relu, int res_mode, unsigned* mask_bits, void* stream, const lmkd_amax_desc* amd);
extern "C" int lmkd_bn_apply(const float* x, const float* stats, const float* res, const float* rstats, float* y, long rows,
                             int C, int relu, int res_mode, unsigned* mask_bits, void* stream) {
  return lmkd_bn_apply_seg(x, stats, res, rstats, y, rows, rows, C, relu, res_mode, mask_bits, stream, nullptr);
}
// two frame segments: rows [0, rows0) use stats[0] (and rstats[0]), rows [rows0, rows) stats[1] / rstats[1] ([2][5][C] tables)
extern "C" int lmkd_bn_apply_seg(const float* x, const float* stats, const float* res, const float* rstats, float* y, long rows, long rows0,
                                 int C, int relu, int res_mode, unsigned* mask_bits, void* stream, const lmkd_amax_desc* amd) {
  unsigned* amax = amd_out(amd);
  const unsigned* aref = amd_ref(amd);
  LMKD_REQUIRE(x && stats && y && rows > 0 && C > 0 && C % 4 == 0, "lmkd_bn_apply: bad arguments (C=%d)", C);
  if (rows0 <= 0 || rows0 > rows) rows0 = rows;
  LMKD_REQUIRE(res_mode == 0 || res, "lmkd_bn_apply: residual pointer missing");
  LMKD_REQUIRE(res_mode != 2 || rstats, "lmkd_bn_apply: residual stats missing");
  LMKD_REQUIRE(!mask_bits || C % 32 == 0, "lmkd_bn_apply: the ReLU bit mask needs C %% 32 == 0 (C=%d)", C);
  LMKD_REQUIRE(!g_lmkd_act_bf16 || C % 8 == 0, "lmkd_bn_apply: bf16 tensors need C %% 8 == 0 (C=%d)", C);
  const long n4 = rows * C / 4;
  if (g_lmkd_act_bf16)
    hipLaunchKernelGGL(bn_apply_kernel<lmkd_bf16_t>, dim3(ew_grid(n4)), dim3(NP_THREADS), 0, (hipStream_t)stream, (const lmkd_bf16_t*)x, stats,
                       (const lmkd_bf16_t*)res, rstats, (lmkd_bf16_t*)y, n4, C, relu, res_mode, mask_bits, rows0 * C / 4, amax, aref);
  else
    hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(ew_grid(n4)), dim3(NP_THREADS), 0, (hipStream_t)stream, x, stats, res, rstats, y, n4, C,
                       relu, res_mode, mask_bits, rows0 * C / 4, amax, aref);
  LMKD_CHECK_LAUNCH("bn_apply_kernel");
  return LMKD_OK;
}

// ---------------------------------------------------------------------------------
// BN backward.  g = dy * mask, mask_mode: 0 none, 1 (yact > 0), 2 (x*scale+shift > 0), 3 (bit mask written by bn_apply:
//   `yact` then points at the packed words)
//   reduce: partial[b][C][2] = (sum g, sum g*xhat)     xhat = (x-mean)*invstd
//   apply : dx = gamma*invstd*(g - sum_g/M - xhat*sum_gx/M);  optional g_out = g
// ---------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float4 bn_masked_grad(const float4 dy, const float4 xv, const T* yact, long i, const float* stats,
                                                 int C, int c, int mask_mode) {
  float4 g = dy;
  if (mask_mode == 3) {
    const unsigned nib = (reinterpret_cast<const unsigned*>(yact)[i >> 3] >> (4 * (int)(i & 7))) & 15u;
    g.x = (nib & 1u) ? g.x : 0.f; g.y = (nib & 2u) ? g.y : 0.f; g.z = (nib & 4u) ? g.z : 0.f; g.w = (nib & 8u) ? g.w : 0.f;
  } else if (mask_mode == 1) {
    const float4 yv = ld4<T>(yact, i);
    g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
  } else if (mask_mode == 2) {
    const float4 sc = *reinterpret_cast<const float4*>(stats + 2 * C + c);
    const float4 sh = *reinterpret_cast<const float4*>(stats + 3 * C + c);
    g.x = fmaf(xv.x, sc.x, sh.x) > 0.f ? g.x : 0.f; g.y = fmaf(xv.y, sc.y, sh.y) > 0.f ? g.y : 0.f;
    g.z = fmaf(xv.z, sc.z, sh.z) > 0.f ? g.z : 0.f; g.w = fmaf(xv.w, sc.w, sh.w) > 0.f ? g.w : 0.f;
  }
  return g;
}

// block: 256 threads = RL row lanes x (CC / 4U) channel groups of one channel chunk (chunk = min(C, 1024) channels, blockIdx.y);
// a thread reads U groups of 4 consecutive channels (16 bytes) per row.  The rows are dealt to nbv "virtual blocks" of RL1 = 256 /
// (CC / 4) row lanes whatever U is - a bf16 workgroup (U = 2) carries two of them - so that the partial sums, their order and with
// them dgamma / dbeta are bit-identical between the fp32 and the bf16 tensor instances on the same values.
// Two frame segments (BnSeg, blockIdx.z = 1): rows [rows0, rows) of every tensor with the second [5][C] table, nb1 virtual blocks and
// the partial rows [2048, 2048 + nb1) - exactly the launch the segment would get on its own.
struct BnSeg { long rows0; int nb1; };
template <typename T>
__global__ void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ yact,
                                     const float* __restrict__ stats, float* __restrict__ partial, long rows, int C, int CC,
                                     int mask_mode, int nbv, BnSeg sg) {
  extern __shared__ float sm[];  // [RL][CC][2]
  constexpr int U = ActU<T>::U;
  if (gridDim.z > 1) {
    if (blockIdx.z) {
      const long eo = sg.rows0 * C;      // element offset of segment 1
      dy += eo; x += eo;
      if (yact) yact = mask_mode == 3 ? reinterpret_cast<const T*>(reinterpret_cast<const unsigned*>(yact) + eo / 32) : yact + eo;
      stats += 5 * C; partial += (long)2048 * 2 * C; rows -= sg.rows0; nbv = sg.nb1;
    } else {
      rows = sg.rows0;
    }
  }
  const int C4 = C >> 2, CCV = CC / (4 * U);
  const int RL = NP_THREADS / CCV, RL1 = RL / U;
  const int cq = threadIdx.x % CCV, rl = threadIdx.x / CCV;
  const int bv = blockIdx.x * U + rl / RL1, vl = rl % RL1;      // virtual block, lane in it
  const int c0 = blockIdx.y * CC;
  const int c = c0 + cq * 4 * U;
  float4 s1[U], s2[U], mean[U], istd[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    s1[u] = make_float4(0, 0, 0, 0); s2[u] = make_float4(0, 0, 0, 0);
    mean[u] = *reinterpret_cast<const float4*>(stats + c + 4 * u);
    istd[u] = *reinterpret_cast<const float4*>(stats + C + c + 4 * u);
  }
  for (long r = bv < nbv ? (long)bv * RL1 + vl : rows; r < rows; r += (long)nbv * RL1) {
    const long i0 = r * C4 + (c >> 2);      // index of the first group of 4
    float4 xv[U], dv[U];
    ldv<T, U>(x, i0 / U, xv);
    ldv<T, U>(dy, i0 / U, dv);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float4 g = bn_masked_grad<T>(dv[u], xv[u], yact, i0 + u, stats, C, c + 4 * u, mask_mode);
      s1[u].x += g.x; s1[u].y += g.y; s1[u].z += g.z; s1[u].w += g.w;
      s2[u].x = fmaf(g.x, (xv[u].x - mean[u].x) * istd[u].x, s2[u].x); s2[u].y = fmaf(g.y, (xv[u].y - mean[u].y) * istd[u].y, s2[u].y);
      s2[u].z = fmaf(g.z, (xv[u].z - mean[u].z) * istd[u].z, s2[u].z); s2[u].w = fmaf(g.w, (xv[u].w - mean[u].w) * istd[u].w, s2[u].w);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float* d = sm + ((long)rl * CC + cq * 4 * U + 4 * u) * 2;
    d[0] = s1[u].x; d[1] = s2[u].x; d[2] = s1[u].y; d[3] = s2[u].y; d[4] = s1[u].z; d[5] = s2[u].z; d[6] = s1[u].w; d[7] = s2[u].w;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * CC; j += NP_THREADS) {
#pragma unroll
    for (int h = 0; h < U; ++h) {
      if (blockIdx.x * U + h >= nbv) break;
      float s = 0.f;
      for (int q = h * RL1; q < (h + 1) * RL1; ++q) s += sm[(long)q * 2 * CC + j];
      partial[(long)(blockIdx.x * U + h) * 2 * C + 2 * c0 + j] = s;
    }
  }
}

// accumulate != 0: dgamma / dbeta point at the parameters' .grad (or at a per-stream shadow of it): += instead of = (gradient
// accumulation over trunk calls / episodes without an ATen add per BatchNorm parameter)
// Two frame segments (gridDim.z = 2): segment z sums ITS partial rows (part + z * pstride floats, Tz rows) into coef[z] ([2][5][C]: A,
// mean g, mean g xhat, sum g xhat, sum g) with its own count, table, scratch and tickets.  The parameter gradients are NOT written here
// (two blocks would add to the same address): the apply kernel that follows adds rows 3 / 4 of both segments (bn_param_grads_seg).
struct CoefSeg { int T1; long pstride; double count1; int S0, S1; };
__global__ void bn_bwd_coef_kernel(const float* __restrict__ part, int T, int C, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ stats, float* __restrict__ coef, float* __restrict__ dgamma,
                                   float* __restrict__ dbeta, int accumulate, double* __restrict__ scratch, unsigned* __restrict__ tickets,
                                   CoefSeg cs) {
  __shared__ double sm[CS_LANES][CS_COLS];
  __shared__ double tot[CS_COLS];
  __shared__ int s_last;
  int S = 0;
  const bool seg = gridDim.z > 1;
  if (seg) {
    if (blockIdx.z) { part += cs.pstride; T = cs.T1; count = cs.count1; stats += 5 * C; coef += 5 * C; scratch += (long)(CS_MAX_SLICES + 2) * 2 * C; tickets += LMKD_TICKET_WORDS / 2; S = cs.S1; }
    else S = cs.S0;
  }
  if (!colsum_ticket(part, T, 2 * C, scratch, tickets, sm, tot, &s_last, nullptr, S)) return;
  const int c = blockIdx.x * (CS_COLS / 2) + threadIdx.x;
  if (threadIdx.y != 0 || threadIdx.x >= CS_COLS / 2 || c >= C) return;
  const double sg = tot[2 * threadIdx.x], sgx = tot[2 * threadIdx.x + 1];
  const float g = gamma ? gamma[c] : 1.f;
  const float invstd = stats[C + c];
  coef[c] = g * invstd;                        // A
  coef[C + c] = (float)(sg / count);           // mean(g)
  coef[2 * C + c] = (float)(sgx / count);      // mean(g*xhat)
  if (seg) {
    coef[3 * C + c] = (float)sgx;
    coef[4 * C + c] = (float)sg;
    return;
  }
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)sgx : (float)sgx;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)sg : (float)sg;
}

// parameter gradients of a two-segment BatchNorm backward: dgamma (+)= sum g xhat of segment 0 + that of segment 1, dbeta likewise - by
// block 0 of the apply pass that follows the coefficient kernel (one writer: deterministic, no launch of its own)
struct ParamGradSeg { float* dgamma; float* dbeta; int accumulate; };
__device__ __forceinline__ void bn_param_grads_seg(const float* __restrict__ coef, int C, const ParamGradSeg& pg) {
  if (blockIdx.x != 0 || !pg.dgamma) return;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float gx = coef[3 * C + c] + coef[5 * C + 3 * C + c], g = coef[4 * C + c] + coef[5 * C + 4 * C + c];
    pg.dgamma[c] = pg.accumulate ? pg.dgamma[c] + gx : gx;
    if (pg.dbeta) pg.dbeta[c] = pg.accumulate ? pg.dbeta[c] + g : g;
  }
}

template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ yact,
                                    const float* __restrict__ stats, const float* __restrict__ coef, T* __restrict__ dx,
                                    T* __restrict__ g_out, long n4, int C, int mask_mode, long n4_0, ParamGradSeg pg,
                                    unsigned* __restrict__ amax, const unsigned* __restrict__ aref) {
  // n4_0 < n4: two frame segments - elements from n4_0 on use stats[1] / coef[1] ([2][5][C] tables), and pg carries the parameter gradients
  constexpr int U = ActU<T>::U;
  const int C4 = C >> 2;
  const long nu = n4 / U;
  if (n4_0 < n4) bn_param_grads_seg(coef, C, pg);
  AmaxStat am = {0.f, 0.f, 0u}, am1 = {0.f, 0.f, 0u};
  const AmaxRef rf0 = amax_ref(aref ? amax_read(aref, 0) : 0u), rf1 = amax_ref(aref ? amax_read(aref, 1) : 0u);
  for (long iu = (long)blockIdx.x * blockDim.x + threadIdx.x; iu < nu; iu += (long)gridDim.x * blockDim.x) {
    float4 xv[U], dv[U], o[U], gq[U];
    ldv<T, U>(x, iu, xv);
    ldv<T, U>(dy, iu, dv);
    const int so = iu * U >= n4_0 ? 5 * C : 0;
    const float* st = stats + so;
    const float* cf = coef + so;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = iu * U + u;
      const int c = (int)(i % C4) * 4;
      const float4 g = bn_masked_grad<T>(dv[u], xv[u], yact, i, st, C, c, mask_mode);
      const float4 mean = *reinterpret_cast<const float4*>(st + c);
      const float4 istd = *reinterpret_cast<const float4*>(st + C + c);
      const float4 A = *reinterpret_cast<const float4*>(cf + c);
      const float4 mg = *reinterpret_cast<const float4*>(cf + C + c);
      const float4 mgx = *reinterpret_cast<const float4*>(cf + 2 * C + c);
      o[u].x = A.x * (g.x - mg.x - (xv[u].x - mean.x) * istd.x * mgx.x);
      o[u].y = A.y * (g.y - mg.y - (xv[u].y - mean.y) * istd.y * mgx.y);
      o[u].z = A.z * (g.z - mg.z - (xv[u].z - mean.z) * istd.z * mgx.z);
      o[u].w = A.w * (g.w - mg.w - (xv[u].w - mean.w) * istd.w * mgx.w);
      gq[u] = g;
      if (so) amax_stat4(am1, o[u], rf1); else amax_stat4(am, o[u], rf0);
    }
    stv<T, U>(dx, iu, o);
    if (g_out) stv<T, U>(g_out, iu, gq);
  }
  if (amax) {
    amax_commit_stat(amax, am, aref != nullptr);
    if (n4_0 < n4) amax_commit_stat(amax + LMKD_AMAX_SEG_WORDS, am1, aref != nullptr);
  }
}

extern "C" long lmkd_bn_bwd_workspace(int C) { return (long)(2048 * 2 * C) * sizeof(float) + (long)(66 * 2 * C) * sizeof(double) + 64; }

// reduce + coefficient launches of the BatchNorm backward: partial sums of (g, g * xhat) over `rows` rows of (dy, x), then
// coef[3][C] = A, mean(g), mean(g * xhat) with the means taken over `count` elements per channel (count = rows except for the stem,
// whose sums run over the POOLED tensor while the BatchNorm normalised the pre-pooling one: lmkd_bn_backward_stats)
// rows0 in (0, rows): two frame segments [0, rows0) | [rows0, rows) with counts count0 | count - count0; stats / coef are then [2][5][C],
// the workspace is 2 * lmkd_bn_bwd_workspace(C) bytes and dgamma / dbeta are left to the apply pass (bn_param_grads_seg)
static int bn_bwd_stats_impl(const float* dy, const float* x, const float* yact, const float* stats, const float* gamma, float* dgamma,
                             float* dbeta, float* coef, void* workspace, unsigned* tickets, long rows, long count, int C, int mask_mode,
                             int accumulate_param_grads, hipStream_t s, const char* who, long rows0 = 0, long count0 = 0) {
  LMKD_REQUIRE(dy && x && stats && coef && workspace && tickets, "%s: null pointer", who);
  const bool seg = rows0 > 0 && rows0 < rows;
  LMKD_REQUIRE(!seg || cdiv(2 * C, CS_COLS) <= LMKD_TICKET_WORDS / 2, "%s: C=%d exceeds the ticket buffer", who, C);
  const int CC = C > 1024 ? 1024 : C;   // channel chunk handled by one workgroup column
  const int U = g_lmkd_act_bf16 ? 2 : 1;      // groups of 4 channels per thread (16-byte accesses)
  LMKD_REQUIRE(C % (4 * U) == 0 && C % CC == 0 && 256 % (CC / 4) == 0, "%s: unsupported channel count %d", who, C);
  LMKD_REQUIRE((mask_mode != 1 && mask_mode != 3) || yact, "%s: mask_mode 1 / 3 needs the activation output / its bit mask", who);
  LMKD_REQUIRE(mask_mode != 3 || C % 32 == 0, "%s: bit masks need C %% 32 == 0", who);
  LMKD_REQUIRE(cdiv(2 * C, CS_COLS) <= LMKD_TICKET_WORDS, "%s: C=%d exceeds the ticket buffer", who, C);
  const int RL1 = NP_THREADS / (CC / 4), RL = RL1 * U;
  auto blocks = [&](long r) {
    int nb = cdiv(r, (long)RL1 * 8);      // virtual blocks of RL1 row lanes (U of them per workgroup)
    // cap: twice the elementwise kernels' workgroup count, for BOTH element types (the partial sums stay bit-identical between them).  A
    // bf16 workgroup carries two virtual blocks: capped at the workgroup count itself, the bf16 launch had half the workgroups and ran
    // as long as the fp32 one on half the bytes (27.0 vs 28.9 us)
    if (nb > 256 * g_ew_wg_per_cu * 2) nb = 256 * g_ew_wg_per_cu * 2;
    if (nb > 2048) nb = 2048;      // rows of the partial buffer (lmkd_bn_bwd_workspace)
    return nb < 1 ? 1 : nb;
  };
  const int nb = blocks(seg ? rows0 : rows), nb1 = seg ? blocks(rows - rows0) : 0;
  const int nbx = std::max(nb, nb1);
  float* partial = (float*)workspace;
  // (two segments: partial rows [0, 2048) | [2048, 4096), then the fp64 scratch of both)
  double* dscr = (double*)((char*)workspace + (((long)(seg ? 2 : 1) * 2048 * 2 * C * sizeof(float) + 63) / 64) * 64);
  BnSeg bs;
  bs.rows0 = seg ? rows0 : rows; bs.nb1 = nb1;
  const int gz = seg ? 2 : 1;
  if (g_lmkd_act_bf16)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<lmkd_bf16_t>, dim3(cdiv(nbx, 2), C / CC, gz), dim3(NP_THREADS), (size_t)RL * CC * 2 * sizeof(float), s,
                       (const lmkd_bf16_t*)dy, (const lmkd_bf16_t*)x, (const lmkd_bf16_t*)yact, stats, partial, rows, C, CC, mask_mode, nb, bs);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(nbx, C / CC, gz), dim3(NP_THREADS), (size_t)RL * CC * 2 * sizeof(float), s, dy, x, yact,
                       stats, partial, rows, C, CC, mask_mode, nb, bs);
  LMKD_CHECK_LAUNCH("bn_bwd_reduce_kernel");
  CoefSeg cs;
  cs.T1 = nb1; cs.pstride = (long)2048 * 2 * C; cs.count1 = (double)(count - count0); cs.S0 = cs_slices(nb); cs.S1 = seg ? cs_slices(nb1) : 0;
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(cdiv(2 * C, CS_COLS), std::max(cs.S0, cs.S1), gz), dim3(CS_COLS, CS_LANES), 0, s, (const float*)partial, nb, C,
                     (double)(seg ? count0 : count), gamma, stats, coef, dgamma, dbeta, accumulate_param_grads, dscr, tickets, cs);
  LMKD_CHECK_LAUNCH("bn_bwd_coef_kernel");
  return LMKD_OK;
}

// dy, x, (yact) : [rows, C];  stats from the forward;  outputs dx (may alias dy), g_out (optional), dgamma, dbeta
// coef: [3][C] floats scratch;  workspace: lmkd_bn_bwd_workspace(C) bytes;  tickets: lmkd_ticket_words() zeroed words (zero again on
// return; one buffer per stream that may run this concurrently);  accumulate_param_grads: dgamma / dbeta += (see bn_bwd_coef_kernel)
extern "C" int lmkd_bn_backward_seg(const float* dy, const float* x, const float* yact, const float* stats, const float* gamma, float* dx,
                                    float* g_out, float* dgamma, float* dbeta, float* coef, void* workspace, unsigned* tickets, long rows,
                                    long rows0, int C, int mask_mode, int accumulate_param_grads, void* stream, const lmkd_amax_desc* amd);
extern "C" int lmkd_bn_backward(const float* dy, const float* x, const float* yact, const float* stats, const float* gamma,
                                float* dx, float* g_out, float* dgamma, float* dbeta, float* coef, void* workspace, unsigned* tickets,
                                long rows, int C, int mask_mode, int accumulate_param_grads, void* stream) {
  return lmkd_bn_backward_seg(dy, x, yact, stats, gamma, dx, g_out, dgamma, dbeta, coef, workspace, tickets, rows, rows, C, mask_mode,
                              accumulate_param_grads, stream, nullptr);
}
// lmkd_bn_backward over two frame segments [0, rows0) | [rows0, rows) of one tensor (both trunk calls of an episode, resnet18_2fc.py:41-42,
// each with its own batch statistics): stats = [2][5][C], coef = [2][5][C] floats of scratch, workspace = 2 * lmkd_bn_bwd_workspace(C)
// bytes; dx / g_out of a segment are bit-identical to a launch on that segment alone; dgamma / dbeta (+)= segment 0's sums + segment 1's.
// rows0 = rows (or 0): one segment, coef [3][C], the plain lmkd_bn_backward.
extern "C" int lmkd_bn_backward_seg(const float* dy, const float* x, const float* yact, const float* stats, const float* gamma, float* dx,
                                    float* g_out, float* dgamma, float* dbeta, float* coef, void* workspace, unsigned* tickets, long rows,
                                    long rows0, int C, int mask_mode, int accumulate_param_grads, void* stream, const lmkd_amax_desc* amd) {
  unsigned* amax = amd_out(amd);
  const unsigned* aref = amd_ref(amd);
  LMKD_REQUIRE(dx, "lmkd_bn_backward: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (rows0 <= 0 || rows0 >= rows) rows0 = rows;
  const int rc = bn_bwd_stats_impl(dy, x, yact, stats, gamma, dgamma, dbeta, coef, workspace, tickets, rows, rows, C, mask_mode,
                                   accumulate_param_grads, s, "lmkd_bn_backward", rows0, rows0);
  if (rc) return rc;
  const long n4 = rows * C / 4, n4_0 = rows0 * C / 4;
  ParamGradSeg pg;
  pg.dgamma = dgamma; pg.dbeta = dbeta; pg.accumulate = accumulate_param_grads;
  if (g_lmkd_act_bf16)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<lmkd_bf16_t>, dim3(ew_grid(n4)), dim3(NP_THREADS), 0, s, (const lmkd_bf16_t*)dy, (const lmkd_bf16_t*)x,
                       (const lmkd_bf16_t*)yact, stats, (const float*)coef, (lmkd_bf16_t*)dx, (lmkd_bf16_t*)g_out, n4, C, mask_mode, n4_0, pg, amax, aref);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(ew_grid(n4)), dim3(NP_THREADS), 0, s, dy, x, yact, stats, (const float*)coef, dx, g_out,
                       n4, C, mask_mode, n4_0, pg, amax, aref);
  LMKD_CHECK_LAUNCH("bn_bwd_apply_kernel");
  return LMKD_OK;
}

// lmkd_bn_backward whose reduction pass already happened in the epilogue of the data gradient that produced dy
// (lmkd_conv2d_bwd_data_bn): part = [T][C][2] per-row-tile sums (sum g, sum g * xhat) of mask mode 2.  Coefficient kernel + apply pass.
extern "C" int lmkd_bn_backward_part_seg(const float* part, int T, int T0, const float* dy, const float* x, const float* stats, const float* gamma,
                                         float* dx, float* dgamma, float* dbeta, float* coef, void* workspace, unsigned* tickets, long rows,
                                         long rows0, int C, int accumulate_param_grads, void* stream, const lmkd_amax_desc* amd);
extern "C" int lmkd_bn_backward_part(const float* part, int T, const float* dy, const float* x, const float* stats, const float* gamma,
                                     float* dx, float* dgamma, float* dbeta, float* coef, void* workspace, unsigned* tickets, long rows, int C,
                                     int accumulate_param_grads, void* stream) {
  return lmkd_bn_backward_part_seg(part, T, T, dy, x, stats, gamma, dx, dgamma, dbeta, coef, workspace, tickets, rows, rows, C,
                                   accumulate_param_grads, stream, nullptr);
}
// two frame segments: partial rows [0, T0) | [T0, T) of lmkd_conv2d_bwd_data_seg, tensor rows [0, rows0) | [rows0, rows); tables as
// lmkd_bn_backward_seg.  T0 = T (rows0 = rows): one segment.
extern "C" int lmkd_bn_backward_part_seg(const float* part, int T, int T0, const float* dy, const float* x, const float* stats, const float* gamma,
                                         float* dx, float* dgamma, float* dbeta, float* coef, void* workspace, unsigned* tickets, long rows,
                                         long rows0, int C, int accumulate_param_grads, void* stream, const lmkd_amax_desc* amd) {
  unsigned* amax = amd_out(amd);
  const unsigned* aref = amd_ref(amd);
  LMKD_REQUIRE(part && T > 0 && dy && x && stats && dx && coef && workspace && tickets, "lmkd_bn_backward_part: null pointer");
  LMKD_REQUIRE(!g_lmkd_act_bf16 && C % 4 == 0, "lmkd_bn_backward_part: fp32 tensors, C %% 4 == 0");
  const bool seg = T0 > 0 && T0 < T && rows0 > 0 && rows0 < rows;
  LMKD_REQUIRE(seg || (T0 == T && rows0 == rows), "lmkd_bn_backward_part_seg: T0 / rows0 must both name a split or both the whole");
  LMKD_REQUIRE(cdiv(2 * C, CS_COLS) <= LMKD_TICKET_WORDS / (seg ? 2 : 1), "lmkd_bn_backward_part: C=%d exceeds the ticket buffer", C);
  hipStream_t s = (hipStream_t)stream;
  double* dscr = (double*)((char*)workspace + (((long)2048 * 2 * C * sizeof(float) + 63) / 64) * 64);
  CoefSeg cs;
  cs.T1 = T - T0; cs.pstride = (long)T0 * 2 * C; cs.count1 = (double)(rows - rows0); cs.S0 = cs_slices(T0); cs.S1 = seg ? cs_slices(T - T0) : 0;
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(cdiv(2 * C, CS_COLS), std::max(cs.S0, cs.S1), seg ? 2 : 1), dim3(CS_COLS, CS_LANES), 0, s, part, T0, C,
                     (double)rows0, gamma, stats, coef, dgamma, dbeta, accumulate_param_grads, dscr, tickets, cs);
  LMKD_CHECK_LAUNCH("bn_bwd_coef_kernel");
  const long n4 = rows * C / 4;
  ParamGradSeg pg;
  pg.dgamma = dgamma; pg.dbeta = dbeta; pg.accumulate = accumulate_param_grads;
  hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(ew_grid(n4)), dim3(NP_THREADS), 0, s, dy, x, (const float*)nullptr, stats, (const float*)coef,
                     dx, (float*)nullptr, n4, C, 2, rows0 * C / 4, pg, amax, aref);
  LMKD_CHECK_LAUNCH("bn_bwd_apply_kernel");
  return LMKD_OK;
}

// The stem's BatchNorm backward statistics from the POOLED side (round 3).  The gradient that reaches the stem's BatchNorm is the
// max-pool backward of dy: non-zero only at the arg-max position of each pooling window, so
//   sum g = sum over windows of dy * [bn(c_max) > 0],   sum g * xhat = sum over windows of dy * [..] * xhat(c_max)
// with c_max = the convolution output at the window's arg-max (saved by lmkd_bn_relu_maxpool_fwd).  One pass over two 160 MB tensors
// instead of materialising the 642 MB pre-pooling gradient and reducing it together with the 642 MB convolution output.
//   dy, cmax: [pooled_rows, C];  count = N * H * W of the pre-pooling tensor (the means of the BatchNorm backward run over it)
extern "C" int lmkd_bn_backward_stats(const float* dy, const float* cmax, const float* stats, const float* gamma, float* dgamma, float* dbeta,
                                      float* coef, void* workspace, unsigned* tickets, long pooled_rows, long count, int C,
                                      int accumulate_param_grads, void* stream) {
  LMKD_REQUIRE(pooled_rows > 0 && count >= pooled_rows, "lmkd_bn_backward_stats: bad row counts");
  return bn_bwd_stats_impl(dy, cmax, nullptr, stats, gamma, dgamma, dbeta, coef, workspace, tickets, pooled_rows, count, C, 2,
                           accumulate_param_grads, (hipStream_t)stream, "lmkd_bn_backward_stats");
}
// two frame segments: pooled rows [0, pooled_rows0) | [.., pooled_rows) with count0 | count - count0 pre-pooling elements per channel;
// stats / coef [2][5][C], workspace 2 * lmkd_bn_bwd_workspace(C).  The parameter gradients are written by lmkd_stem_unpool_bn_bwd_seg.
extern "C" int lmkd_bn_backward_stats_seg(const float* dy, const float* cmax, const float* stats, const float* gamma, float* coef, void* workspace,
                                          unsigned* tickets, long pooled_rows, long pooled_rows0, long count, long count0, int C, void* stream) {
  LMKD_REQUIRE(pooled_rows0 > 0 && pooled_rows0 < pooled_rows && count0 >= pooled_rows0 && count - count0 >= pooled_rows - pooled_rows0,
               "lmkd_bn_backward_stats_seg: bad row counts");
  return bn_bwd_stats_impl(dy, cmax, nullptr, stats, gamma, nullptr, nullptr, coef, workspace, tickets, pooled_rows, count, C, 2, 0,
                           (hipStream_t)stream, "lmkd_bn_backward_stats_seg", pooled_rows0, count0);
}

// plain ReLU backward (eval-mode / no-BN paths): g = dy * (y > 0)
template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ g, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 d = ld4<T>(dy, i), v = ld4<T>(y, i);
    st4<T>(g, i, make_float4(v.x > 0.f ? d.x : 0.f, v.y > 0.f ? d.y : 0.f, v.z > 0.f ? d.z : 0.f, v.w > 0.f ? d.w : 0.f));
  }
}
extern "C" int lmkd_relu_backward(const float* dy, const float* y, float* g, long n, void* stream) {
  LMKD_REQUIRE(dy && y && g && n > 0 && n % 4 == 0, "lmkd_relu_backward: bad arguments");
  if (g_lmkd_act_bf16)
    hipLaunchKernelGGL(relu_bwd_kernel<lmkd_bf16_t>, dim3(ew_grid(n / 4)), dim3(NP_THREADS), 0, (hipStream_t)stream, (const lmkd_bf16_t*)dy,
                       (const lmkd_bf16_t*)y, (lmkd_bf16_t*)g, n / 4);
  else
    hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(ew_grid(n / 4)), dim3(NP_THREADS), 0, (hipStream_t)stream, dy, y, g, n / 4);
  LMKD_CHECK_LAUNCH("relu_bwd_kernel");
  return LMKD_OK;
}

// ---------------------------------------------------------------------------------
// stem: y = maxpool3x3/s2/p1( relu( x*scale + shift ) ), argmax (0..8, first max in scan order)
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void bn_relu_maxpool_kernel(const T* __restrict__ x, const float* __restrict__ stats, T* __restrict__ y,
                                       uchar4* __restrict__ idx, T* __restrict__ cmax, int N, int H, int W, int C, int OH, int OW, int N0,
                                       unsigned* __restrict__ amax, const unsigned* __restrict__ aref) {
  // a thread owns U groups of 4 consecutive channels of one output pixel (U = 2 with bf16 tensors: every access 16 bytes; with 8-byte
  // accesses the bf16 instance took as long as the fp32 one on half the bytes - 260 vs 280 us at 200 frames)
  constexpr int U = ActU<T>::U;
  const int CU = C / (4 * U);
  const long total = (long)N * OH * OW * CU;
  AmaxStat amx = {0.f, 0.f, 0u}, amx1 = {0.f, 0.f, 0u};
  const AmaxRef rf0 = amax_ref(aref ? amax_read(aref, 0) : 0u), rf1 = amax_ref(aref ? amax_read(aref, 1) : 0u);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % CU);
    long r = i / CU;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const int n = (int)(r / OH);
    float4 sc[U], sh[U], m[U], cm[U];      // cm: the raw convolution output at the arg-max (backward: lmkd_bn_backward_stats)
    uchar4 am[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float* st = stats + (n >= N0 ? 5 * C : 0);      // frames from N0 on: the second segment's table
      sc[u] = *reinterpret_cast<const float4*>(st + 2 * C + (cq * U + u) * 4);
      sh[u] = *reinterpret_cast<const float4*>(st + 3 * C + (cq * U + u) * 4);
      m[u] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      cm[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      am[u] = make_uchar4(255, 255, 255, 255);
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = oh * 2 - 1 + kh;
      if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w = ow * 2 - 1 + kw;
        if ((unsigned)w >= (unsigned)W) continue;
        float4 rr[U];
        ldv<T, U>(x, ((long)(n * H + h) * W + w) * CU + cq, rr);
        const unsigned char t = (unsigned char)(kh * 3 + kw);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float4 r4 = rr[u];
          float4 v;
          v.x = fmaxf(fmaf(r4.x, sc[u].x, sh[u].x), 0.f); v.y = fmaxf(fmaf(r4.y, sc[u].y, sh[u].y), 0.f);
          v.z = fmaxf(fmaf(r4.z, sc[u].z, sh[u].z), 0.f); v.w = fmaxf(fmaf(r4.w, sc[u].w, sh[u].w), 0.f);
          if (v.x > m[u].x || am[u].x == 255) { m[u].x = v.x; am[u].x = t; cm[u].x = r4.x; }
          if (v.y > m[u].y || am[u].y == 255) { m[u].y = v.y; am[u].y = t; cm[u].y = r4.y; }
          if (v.z > m[u].z || am[u].z == 255) { m[u].z = v.z; am[u].z = t; cm[u].z = r4.z; }
          if (v.w > m[u].w || am[u].w == 255) { m[u].w = v.w; am[u].w = t; cm[u].w = r4.w; }
        }
      }
    }
    stv<T, U>(y, i, m);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      idx[i * U + u] = am[u];
      if (n >= N0) amax_stat4(amx1, m[u], rf1); else amax_stat4(amx, m[u], rf0);
    }
    if (cmax) stv<T, U>(cmax, i, cm);
  }
  if (amax) {
    amax_commit_stat(amax, amx, aref != nullptr);
    if (N0 < N) amax_commit_stat(amax + LMKD_AMAX_SEG_WORDS, amx1, aref != nullptr);
  }
}

extern "C" int lmkd_bn_relu_maxpool_fwd_seg(const float* x, const float* stats, float* y, unsigned char* idx, float* cmax, int N, int N0, int H,
                                            int W, int C, void* stream, const lmkd_amax_desc* amd);
extern "C" int lmkd_bn_relu_maxpool_fwd(const float* x, const float* stats, float* y, unsigned char* idx, float* cmax, int N, int H, int W,
                                        int C, void* stream) {
  return lmkd_bn_relu_maxpool_fwd_seg(x, stats, y, idx, cmax, N, N, H, W, C, stream, nullptr);
}
// two frame segments: frames [0, N0) use stats[0], frames [N0, N) stats[1] ([2][5][C])
extern "C" int lmkd_bn_relu_maxpool_fwd_seg(const float* x, const float* stats, float* y, unsigned char* idx, float* cmax, int N, int N0, int H,
                                            int W, int C, void* stream, const lmkd_amax_desc* amd) {
  unsigned* amax = amd_out(amd);
  const unsigned* aref = amd_ref(amd);
  LMKD_REQUIRE(x && stats && y && idx && C % (g_lmkd_act_bf16 ? 8 : 4) == 0, "lmkd_bn_relu_maxpool_fwd: bad arguments");
  if (N0 <= 0 || N0 > N) N0 = N;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * OH * OW * C / (g_lmkd_act_bf16 ? 8 : 4);
  if (g_lmkd_act_bf16)
    hipLaunchKernelGGL(bn_relu_maxpool_kernel<lmkd_bf16_t>, dim3(ew_grid(total)), dim3(NP_THREADS), 0, (hipStream_t)stream, (const lmkd_bf16_t*)x,
                       stats, (lmkd_bf16_t*)y, (uchar4*)idx, (lmkd_bf16_t*)cmax, N, H, W, C, OH, OW, N0, amax, aref);
  else
    hipLaunchKernelGGL(bn_relu_maxpool_kernel<float>, dim3(ew_grid(total)), dim3(NP_THREADS), 0, (hipStream_t)stream, x, stats, y, (uchar4*)idx,
                       cmax, N, H, W, C, OH, OW, N0, amax, aref);
  LMKD_CHECK_LAUNCH("bn_relu_maxpool_kernel");
  return LMKD_OK;
}

// gather form of the maxpool backward: g[n,h,w,c] = sum over the (<=4) windows covering (h,w) whose argmax is (h,w)
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const uchar4* __restrict__ idx, T* __restrict__ g, int N, int H,
                                   int W, int C, int OH, int OW) {
  const int C4 = C >> 2;
  const long total = (long)N * H * W * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % C4);
    long r = i / C4;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // windows oh with 2*oh-1 <= h <= 2*oh+1
    const int oh_lo = h >> 1, oh_hi = (h + 1) >> 1;
    const int ow_lo = w >> 1, ow_hi = (w + 1) >> 1;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      if (oh >= OH) continue;
      const int kh = h - (2 * oh - 1);
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        if (ow >= OW) continue;
        const int kw = w - (2 * ow - 1);
        const unsigned char t = (unsigned char)(kh * 3 + kw);
        const long o = ((long)(n * OH + oh) * OW + ow) * C4 + cq;
        const uchar4 am = idx[o];
        const float4 d = ld4<T>(dy, o);
        if (am.x == t) acc.x += d.x;
        if (am.y == t) acc.y += d.y;
        if (am.z == t) acc.z += d.z;
        if (am.w == t) acc.w += d.w;
      }
    }
    st4<T>(g, i, acc);
  }
}

// Max-pool backward + BatchNorm backward apply of the stem in ONE pass (round 3; before: maxpool_bwd wrote the 642 MB pre-pooling
// gradient g, bn_bwd_apply read g and c and wrote dc).  A thread owns a 2 x 2 block of pre-pooling pixels (rows 2a, 2a + 1, columns
// 2b, 2b + 1) x 4 channels: exactly the pixels the four windows (a .. a + 1, b .. b + 1) can select in that block - even rows /
// columns belong to one window, odd ones to two - so it loads four (dy, arg-max) pairs instead of up to four per PIXEL (the
// gather form above), forms g for its four pixels in the order of maxpool_bwd_kernel (window row, then window column) and writes
//   dc = A * (m * g - mean(g) - xhat * mean(g * xhat)),   m = (c * scale + shift > 0),   xhat = (c - mean) * invstd
// - the arithmetic of bn_bwd_apply_kernel with mask_mode 2.
template <typename T>
__global__ void stem_unpool_bn_bwd_kernel(const T* __restrict__ dy, const uchar4* __restrict__ idx, const T* __restrict__ c,
                                          const float* __restrict__ stats, const float* __restrict__ coef, T* __restrict__ dc, int N,
                                          int H, int W, int C, int OH, int OW, int N0, ParamGradSeg pg, unsigned* __restrict__ amax,
                                          const unsigned* __restrict__ aref) {
  const int C4 = C >> 2;
  const int HB = (H + 1) >> 1, WB = (W + 1) >> 1;
  const long total = (long)N * HB * WB * C4;
  AmaxStat amx = {0.f, 0.f, 0u}, amx1 = {0.f, 0.f, 0u};      // max |dc| (+ range statistics) of frame segment 0 / 1
  const AmaxRef rf0 = amax_ref(aref ? amax_read(aref, 0) : 0u), rf1 = amax_ref(aref ? amax_read(aref, 1) : 0u);
  if (N0 < N) bn_param_grads_seg(coef, C, pg);      // two frame segments: tables [2][5][C], parameter gradients from both (bn_bwd_coef_kernel)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % C4);
    long r = i / C4;
    const int b = (int)(r % WB); r /= WB;
    const int a = (int)(r % HB);
    const int n = (int)(r / HB);
    const int ch = cq * 4 + (n >= N0 ? 5 * C : 0);
    const float4 sc = *reinterpret_cast<const float4*>(stats + 2 * C + ch);
    const float4 sh = *reinterpret_cast<const float4*>(stats + 3 * C + ch);
    const float4 mean = *reinterpret_cast<const float4*>(stats + ch);
    const float4 istd = *reinterpret_cast<const float4*>(stats + C + ch);
    const float4 A = *reinterpret_cast<const float4*>(coef + ch);
    const float4 mg = *reinterpret_cast<const float4*>(coef + C + ch);
    const float4 mgx = *reinterpret_cast<const float4*>(coef + 2 * C + ch);
    // the four windows (a + i, b + j)
    float4 d[2][2];
    uchar4 am[2][2];
#pragma unroll
    for (int wi = 0; wi < 2; ++wi)
#pragma unroll
      for (int wj = 0; wj < 2; ++wj) {
        const int oh = a + wi, ow = b + wj;
        if (oh < OH && ow < OW) {
          const long o = ((long)(n * OH + oh) * OW + ow) * C4 + cq;
          d[wi][wj] = ld4<T>(dy, o);
          am[wi][wj] = idx[o];
        } else {
          d[wi][wj] = make_float4(0.f, 0.f, 0.f, 0.f);
          am[wi][wj] = make_uchar4(254, 254, 254, 254);
        }
      }
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const int h = 2 * a + pi;
      if (h >= H) continue;
#pragma unroll
      for (int pj = 0; pj < 2; ++pj) {
        const int w = 2 * b + pj;
        if (w >= W) continue;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        // windows that contain (h, w): rows oh = a (kh = pi + 1) and, for the odd row, oh = a + 1 (kh = 0); columns likewise
#pragma unroll
        for (int wi = 0; wi <= pi; ++wi) {
          const int kh = wi == 0 ? pi + 1 : 0;
#pragma unroll
          for (int wj = 0; wj <= pj; ++wj) {
            const int kw = wj == 0 ? pj + 1 : 0;
            const unsigned char t = (unsigned char)(kh * 3 + kw);
            if (am[wi][wj].x == t) g.x += d[wi][wj].x;
            if (am[wi][wj].y == t) g.y += d[wi][wj].y;
            if (am[wi][wj].z == t) g.z += d[wi][wj].z;
            if (am[wi][wj].w == t) g.w += d[wi][wj].w;
          }
        }
        const long p = ((long)(n * H + h) * W + w) * C4 + cq;
        const float4 xv = ld4<T>(c, p);
        g.x = fmaf(xv.x, sc.x, sh.x) > 0.f ? g.x : 0.f; g.y = fmaf(xv.y, sc.y, sh.y) > 0.f ? g.y : 0.f;
        g.z = fmaf(xv.z, sc.z, sh.z) > 0.f ? g.z : 0.f; g.w = fmaf(xv.w, sc.w, sh.w) > 0.f ? g.w : 0.f;
        float4 o;
        o.x = A.x * (g.x - mg.x - (xv.x - mean.x) * istd.x * mgx.x);
        o.y = A.y * (g.y - mg.y - (xv.y - mean.y) * istd.y * mgx.y);
        o.z = A.z * (g.z - mg.z - (xv.z - mean.z) * istd.z * mgx.z);
        o.w = A.w * (g.w - mg.w - (xv.w - mean.w) * istd.w * mgx.w);
        st4<T>(dc, p, o);
        if (n >= N0) amax_stat4(amx1, o, rf1); else amax_stat4(amx, o, rf0);
      }
    }
  }
  if (amax) {
    amax_commit_stat(amax, amx, aref != nullptr);
    if (N0 < N) amax_commit_stat(amax + LMKD_AMAX_SEG_WORDS, amx1, aref != nullptr);
  }
}

// dy, idx: [N, OH, OW, C] (pooled gradient, arg-max bytes of lmkd_bn_relu_maxpool_fwd); c: [N, H, W, C] convolution output;
// stats: its [5][C] table; coef: [3][C] from lmkd_bn_backward_stats; dc: [N, H, W, C] gradient w.r.t. the convolution output
extern "C" int lmkd_stem_unpool_bn_bwd_seg(const float* dy, const unsigned char* idx, const float* c, const float* stats, const float* coef,
                                           float* dc, float* dgamma, float* dbeta, int accumulate_param_grads, int N, int N0, int H, int W, int C,
                                           void* stream, const lmkd_amax_desc* amd);
extern "C" int lmkd_stem_unpool_bn_bwd(const float* dy, const unsigned char* idx, const float* c, const float* stats, const float* coef,
                                       float* dc, int N, int H, int W, int C, void* stream, const lmkd_amax_desc* amd) {
  return lmkd_stem_unpool_bn_bwd_seg(dy, idx, c, stats, coef, dc, nullptr, nullptr, 0, N, N, H, W, C, stream, amd);
}
// two frame segments [0, N0) | [N0, N): stats / coef = the [2][5][C] tables of lmkd_bn_backward_stats_seg; dgamma / dbeta (nullable) (+)= the
// sums of both segments (rows 3 / 4 of coef).  N0 = N: one segment, the plain form (coef [3][C], dgamma / dbeta ignored).
extern "C" int lmkd_stem_unpool_bn_bwd_seg(const float* dy, const unsigned char* idx, const float* c, const float* stats, const float* coef,
                                           float* dc, float* dgamma, float* dbeta, int accumulate_param_grads, int N, int N0, int H, int W, int C,
                                           void* stream, const lmkd_amax_desc* amd) {
  unsigned* amax = amd_out(amd);
  const unsigned* aref = amd_ref(amd);
  LMKD_REQUIRE(dy && idx && c && stats && coef && dc && C % 4 == 0 && N > 0 && H > 0 && W > 0, "lmkd_stem_unpool_bn_bwd: bad arguments");
  if (N0 <= 0 || N0 > N) N0 = N;
  ParamGradSeg pg;
  pg.dgamma = dgamma; pg.dbeta = dbeta; pg.accumulate = accumulate_param_grads;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * C / 4;
  if (g_lmkd_act_bf16)
    hipLaunchKernelGGL(stem_unpool_bn_bwd_kernel<lmkd_bf16_t>, dim3(ew_grid(total)), dim3(NP_THREADS), 0, (hipStream_t)stream,
                       (const lmkd_bf16_t*)dy, (const uchar4*)idx, (const lmkd_bf16_t*)c, stats, coef, (lmkd_bf16_t*)dc, N, H, W, C, OH, OW, N0, pg, amax, aref);
  else
    hipLaunchKernelGGL(stem_unpool_bn_bwd_kernel<float>, dim3(ew_grid(total)), dim3(NP_THREADS), 0, (hipStream_t)stream, dy, (const uchar4*)idx, c,
                       stats, coef, dc, N, H, W, C, OH, OW, N0, pg, amax, aref);
  LMKD_CHECK_LAUNCH("stem_unpool_bn_bwd_kernel");
  return LMKD_OK;
}

extern "C" int lmkd_maxpool_bwd(const float* dy, const unsigned char* idx, float* g, int N, int H, int W, int C, void* stream) {
  LMKD_REQUIRE(dy && idx && g && C % 4 == 0, "lmkd_maxpool_bwd: bad arguments");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * H * W * C / 4;
  if (g_lmkd_act_bf16)
    hipLaunchKernelGGL(maxpool_bwd_kernel<lmkd_bf16_t>, dim3(ew_grid(total)), dim3(NP_THREADS), 0, (hipStream_t)stream, (const lmkd_bf16_t*)dy,
                       (const uchar4*)idx, (lmkd_bf16_t*)g, N, H, W, C, OH, OW);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(ew_grid(total)), dim3(NP_THREADS), 0, (hipStream_t)stream, dy, (const uchar4*)idx, g, N, H,
                       W, C, OH, OW);
  LMKD_CHECK_LAUNCH("maxpool_bwd_kernel");
  return LMKD_OK;
}

// ---------------------------------------------------------------------------------
// head: AdaptiveMaxPool2d((4,4)) + mean over the 16 patches.  x [F,H,W,C] -> y [F,C]
// window i of an axis of length L: [floor(i*L/4), ceil((i+1)*L/4))
// ---------------------------------------------------------------------------------
template <typename T>      // x: activation storage type; y (pooled frame features) fp32
__global__ void adaptive_maxpool_mean_kernel(const T* __restrict__ x, float* __restrict__ y, int F, int H, int W, int C) {
  const long total = (long)F * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int f = (int)(i / C);
    const T* b = x + (long)f * H * W * C + c;
    float s = 0.f;
    for (int ph = 0; ph < 4; ++ph) {
      const int h0 = (ph * H) / 4, h1 = ((ph + 1) * H + 3) / 4;
      for (int pw = 0; pw < 4; ++pw) {
        const int w0 = (pw * W) / 4, w1 = ((pw + 1) * W + 3) / 4;
        float m = -INFINITY;
        for (int h = h0; h < h1; ++h)
          for (int w = w0; w < w1; ++w) m = fmaxf(m, ld1<T>(b, (long)(h * W + w) * C));
        s += m;
      }
    }
    y[i] = s * (1.f / 16.f);
  }
}

// dx[f,h,w,c] = sum over windows whose (first) argmax is (h,w) of dy[f,c]/16.  H*W <= 64.
template <typename T>      // x, dx: activation storage type; dy fp32
__global__ void adaptive_maxpool_mean_bwd_kernel(const T* __restrict__ x, const float* __restrict__ dy, T* __restrict__ dx,
                                                 int F, int H, int W, int C) {
  const long total = (long)F * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int f = (int)(i / C);
    const T* b = x + (long)f * H * W * C + c;
    T* d = dx + (long)f * H * W * C + c;
    const float gv = dy[i] * (1.f / 16.f);
    unsigned long long cnt_lo = 0, cnt_hi = 0, cnt_4 = 0;   // 3 bit planes: a position is the argmax of at most 4 windows
    for (int ph = 0; ph < 4; ++ph) {
      const int h0 = (ph * H) / 4, h1 = ((ph + 1) * H + 3) / 4;
      for (int pw = 0; pw < 4; ++pw) {
        const int w0 = (pw * W) / 4, w1 = ((pw + 1) * W + 3) / 4;
        float m = -INFINITY;
        int am = h0 * W + w0;
        for (int h = h0; h < h1; ++h)
          for (int w = w0; w < w1; ++w) {
            const float v = ld1<T>(b, (long)(h * W + w) * C);
            if (v > m) { m = v; am = h * W + w; }
          }
        // ripple-add 1 into the per-position counter held as bit planes
        const unsigned long long bit = 1ull << am;
        const unsigned long long carry = cnt_lo & bit;
        const unsigned long long carry2 = cnt_hi & carry;
        cnt_lo ^= bit;
        cnt_hi ^= carry;
        cnt_4 ^= carry2;
      }
    }
    for (int p = 0; p < H * W; ++p) {
      const int k = (int)((cnt_lo >> p) & 1ull) + 2 * (int)((cnt_hi >> p) & 1ull) + 4 * (int)((cnt_4 >> p) & 1ull);
      st1<T>(d, (long)p * C, gv * (float)k);
    }
  }
}

extern "C" int lmkd_adaptive_maxpool_mean_fwd(const float* x, float* y, int F, int H, int W, int C, void* stream) {
  LMKD_REQUIRE(x && y && F > 0 && H >= 1 && W >= 1 && C > 0, "lmkd_adaptive_maxpool_mean_fwd: bad arguments");
  if (g_lmkd_act_bf16)
    hipLaunchKernelGGL(adaptive_maxpool_mean_kernel<lmkd_bf16_t>, dim3(ew_grid((long)F * C)), dim3(NP_THREADS), 0, (hipStream_t)stream,
                       (const lmkd_bf16_t*)x, y, F, H, W, C);
  else
    hipLaunchKernelGGL(adaptive_maxpool_mean_kernel<float>, dim3(ew_grid((long)F * C)), dim3(NP_THREADS), 0, (hipStream_t)stream, x, y, F, H, W, C);
  LMKD_CHECK_LAUNCH("adaptive_maxpool_mean_kernel");
  return LMKD_OK;
}
extern "C" int lmkd_adaptive_maxpool_mean_bwd(const float* x, const float* dy, float* dx, int F, int H, int W, int C, void* stream) {
  LMKD_REQUIRE(x && dy && dx && F > 0 && H * W <= 64, "lmkd_adaptive_maxpool_mean_bwd: bad arguments (H*W must be <= 64)");
  LMKD_REQUIRE(H >= 2 && W >= 2 && H <= 8 && W <= 8, "lmkd_adaptive_maxpool_mean_bwd: H, W must be in [2,8] (at most 2 overlapping windows per axis)");
  if (g_lmkd_act_bf16)
    hipLaunchKernelGGL(adaptive_maxpool_mean_bwd_kernel<lmkd_bf16_t>, dim3(ew_grid((long)F * C)), dim3(NP_THREADS), 0, (hipStream_t)stream,
                       (const lmkd_bf16_t*)x, dy, (lmkd_bf16_t*)dx, F, H, W, C);
  else
    hipLaunchKernelGGL(adaptive_maxpool_mean_bwd_kernel<float>, dim3(ew_grid((long)F * C)), dim3(NP_THREADS), 0, (hipStream_t)stream, x, dy, dx, F,
                       H, W, C);
  LMKD_CHECK_LAUNCH("adaptive_maxpool_mean_bwd_kernel");
  return LMKD_OK;
}

// ---------------------------------------------------------------------------------
// column sums of a [rows, C] matrix (Linear bias grads, LayerNorm parameter grads):
//   out[c] = sum_r a[r,c] * (b ? b[r,c] : 1)
// ---------------------------------------------------------------------------------
template <bool PROD>
__global__ void colsum_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int rows, int C,
                              int accumulate, double* __restrict__ scratch, unsigned* __restrict__ tickets) {
  __shared__ double sm[CS_LANES][CS_COLS];
  __shared__ double tot[CS_COLS];
  __shared__ int s_last;
  if (!colsum_ticket<PROD>(a, rows, C, scratch, tickets, sm, tot, &s_last, b)) return;
  const int c = blockIdx.x * CS_COLS + threadIdx.x;
  if (threadIdx.y == 0 && c < C) out[c] = (accumulate ? out[c] : 0.f) + (float)tot[threadIdx.x];
}

extern "C" long lmkd_colsum_workspace(int C) { return (long)CS_MAX_SLICES * C * sizeof(double) + 64; }

// ONE launch (round 3; before: row slices, a reduction of the slices and a finishing kernel).  tickets: as lmkd_bn_finalize
extern "C" int lmkd_colsum(const float* a, const float* b, float* out, long rows, int C, int accumulate, void* workspace, unsigned* tickets,
                           void* stream) {
  LMKD_REQUIRE(a && out && workspace && tickets && rows > 0 && rows < 2147483647L && C > 0, "lmkd_colsum: bad arguments");
  LMKD_REQUIRE(cdiv(C, CS_COLS) <= LMKD_TICKET_WORDS, "lmkd_colsum: C=%d exceeds the ticket buffer", C);
  const dim3 grid(cdiv(C, CS_COLS), cs_slices((int)rows)), block(CS_COLS, CS_LANES);
  if (b) hipLaunchKernelGGL(colsum_kernel<true>, grid, block, 0, (hipStream_t)stream, a, b, out, (int)rows, C, accumulate, (double*)workspace, tickets);
  else hipLaunchKernelGGL(colsum_kernel<false>, grid, block, 0, (hipStream_t)stream, a, b, out, (int)rows, C, accumulate, (double*)workspace, tickets);
  LMKD_CHECK_LAUNCH("colsum_kernel");
  return LMKD_OK;
}
