// Implicit-GEMM 2-D convolution on the fp32 MFMA tile engine, NHWC activations.
//
//   forward      y[n,oh,ow,co]   = sum_{kh,kw,ci} x[n,oh*s+kh-p,ow*s+kw-p,ci] * W[co,ci,kh,kw]
//   data grad    dx[n,ih,iw,ci]  = sum_{kh,kw,co} dy[n,(ih+p-kh)/s,(iw+p-kw)/s,co] * W[co,ci,kh,kw]
//   weight grad  dW[co,ci,kh,kw] = sum_{n,oh,ow}  dy[n,oh,ow,co] * x[n,oh*s+kh-p,ow*s+kw-p,ci]
//
// forward and data-grad share one kernel: GEMM rows = output pixels, GEMM cols = output
// channels, K = (tap, channel) of the gathered tensor; a per-class tap table {dh,dw,kofs}
// describes which source pixel / packed-weight slice each K-step reads.  Stride-2 data-grad
// enumerates its rows per input-parity class (4 classes), so every tile only runs the taps
// that are non-zero for its parity (no wasted MFMAs on the zero-stuffed positions).
// The 7x7 stem (Cin=3) runs on a channel-padded NHWC4 input: one K-step = one kernel row
// (8 taps x 4 channels = 32 contiguous floats of the input).
// Train-mode BatchNorm statistics (sum, sum of squares per channel) are reduced from the
// accumulators in the forward epilogue into per-row-tile partials (deterministic).
#include "gemm_core.h"
#include <atomic>
#include <mutex>

struct ConvGemmArgs {
  const float* src;
  const float* wpk;
  float* out;
  float* stat_partial;
  int N, Hs, Ws, Cs;
  int Hr, Wr, rows_per_class, tiles_per_class;
  int sh;
  int Ho, Wo, Co, omul;
  int Kp, cps, nclass;
  int n_rt, n_ct, xcd_mode;   // row tiles (all classes), column tiles, tile order (XCD-aware 1-D grid, see xcd_decode)
  int taps_pm1;               // conv_patch.h: every tap of every class lies within +-1 pixel (3x3 / 1x1 launches): the per-row tap masks come from a table
  int grid_step;              // persistent launches: gridDim.x (read from the argument block's LDS copy; gridDim itself is another scalar load from the segment)
  int grid_total;             // persistent launches (conv_patch.h): slots of the tile order, xcd_grid(n_rt, n_ct, xcd_mode); a workgroup walks slots blockIdx.x + k gridDim.x
  int same;    // same-size convolution (conv_same_size): the tile height may be one only the patch kernel has
  int halo;    // conv_patch_x3_kernel: largest |dh * Ws + dw| over the taps (pixels either side of the tile's own range)
  int accum;   // epilogue: out = acc + out (residual-branch gradient already sits in the output buffer)
  // STATS == 2 (inference): out = relu?( acc * scale[c] + shift[c] (+ res) ), scale/shift = rows 2/3 of the [5][C] BatchNorm table
  const float* ep_stats;
  const float* ep_res;
  int ep_relu;
  // PRE (training): src holds the RAW output of the previous convolution; the loader applies that layer's train-mode BatchNorm
  // and ReLU, relu(fmaf(v, scale[c], shift[c])) with scale/shift = rows 2/3 of its [5][C] table, while it fills LDS - the
  // normalised activation never exists in HBM.  Zero padding stays zero (it pads the activation, not the raw tensor).
  const float* pre_stats;
  // data gradient whose result feeds relu + train-mode BatchNorm backward (lmkd_conv2d_bwd_data_bn): bnb_x = that BatchNorm's input (the
  // raw convolution output, shaped like `out`), bnb_stats = its [5][Co] table; stat_partial then receives per row tile
  // (sum g, sum g * xhat), g = out * [fma(x, scale, shift) > 0], xhat = (x - mean) * invstd - what bn_bwd_reduce_kernel sums
  const float* bnb_x;
  const float* bnb_stats;
  // SRC2 (conv_patch16.h): stride-2 forward convolution as four same-size convolutions over the input's parity classes.  Hs / Ws then
  // hold the CLASS grid (= the output grid), s2_wfull the input's width, s2_off[c] the element offset of class c's origin
  // ((sph * W + spw) * Cs) and taps[0][s2_t0[c] .. s2_t0[c + 1] - 1] the taps of class c (dh / dw in the class grid)
  int src2, s2_ncls, s2_wfull;
  int s2_t0[5], s2_off[4];
  // Two frame segments in one launch (round 4: the support-frame and the query-frame trunk call of an episode, resnet18_2fc.py:41-42, as
  // ONE tensor [F0 + F1, H, W, C]): rows [0, seg_m0) of a class belong to segment 0, the rest to segment 1.  The row tiles are dealt per
  // segment - tiles [0, seg_t0) cover segment 0, tile seg_t0 starts at row seg_m0 - so no tile straddles the boundary: every tile's
  // BatchNorm partial sums belong to one segment, the [2][5][C] tables (pre_stats, bnb_stats) are indexed by the tile's segment, and the
  // tiles (with their -W halves, x3_neg_tile) are exactly those of two separate launches: bit-identical results.
  // Unsegmented launch: seg_m0 = rows_per_class, seg_t0 = tiles_per_class (conv_set_tiles).
  int seg_m0, seg_t0;
  const unsigned* h2_xw;      // two-plane fp16 arithmetic (compute dtype 4): the word holding max |operand| (lmkd_conv_operand_amax); null: three bf16 planes
  unsigned* amax_out;         // conv_patch16_x3_kernel: fold max |out| into these words (lmkd_conv_output_amax); null: not recorded
  FastDiv div_hw, div_w;
  int ntap[LMKD_MAX_CLASSES];
  Tap taps[LMKD_MAX_CLASSES][LMKD_MAX_TAPS];
};

// tile -> first row, row limit, segment, index / count of the tile inside its segment (ConvGemmArgs::seg_m0)
struct SegTile { int row0, mend, seg, ltile, ltiles; };
__device__ __forceinline__ SegTile seg_tile(const ConvGemmArgs& a, int tile, int BM) {
  SegTile s;
  if (tile < a.seg_t0) { s.row0 = tile * BM; s.mend = a.seg_m0; s.seg = 0; s.ltile = tile; s.ltiles = a.seg_t0; }
  else { s.row0 = a.seg_m0 + (tile - a.seg_t0) * BM; s.mend = a.rows_per_class; s.seg = 1; s.ltile = tile - a.seg_t0; s.ltiles = a.tiles_per_class - a.seg_t0; }
  return s;
}
// host: row tiles of a launch with BM rows per tile (per class), segment-aware
static inline void conv_set_tiles(ConvGemmArgs& a, int BM) {
  if (a.seg_m0 <= 0 || a.seg_m0 >= a.rows_per_class) a.seg_m0 = a.rows_per_class;
  a.seg_t0 = (a.seg_m0 + BM - 1) / BM;
  a.tiles_per_class = a.seg_t0 + (a.rows_per_class - a.seg_m0 + BM - 1) / BM;
  a.n_rt = a.nclass * a.tiles_per_class;
}

// K-major implicit-im2col loader (rows = output pixels of this tile).
template <int ROWS, bool SMALLC, int THREADS = LMKD_THREADS, bool PRE = false>
struct LoaderConvGather {
  static constexpr bool ROWK = true;
  static constexpr int RPP = THREADS / 8;
  static constexpr int NI = ROWS / RPP;
  static constexpr int LD = LMKD_LDK;
  static constexpr int LDS_FLOATS = ROWS * LMKD_LDK;
  int base[NI], hw[NI];
  float4 reg[NI];
  const float* src;
  int Hs, Ws, Cs, kc4;
  float4 psc, psh;      // PRE: BatchNorm scale / shift of this lane's 4 channels of the current 32-channel chunk
  unsigned inb;         // PRE: which of the NI prefetched rows are real pixels (zero padding must stay zero)
  // the chunk's table entries: fetched once per chunk (the K loop runs the taps of one chunk back to back).  Like the tile
  // loads they are only CONSUMED in store(), after the MFMAs of the current K-step: nothing waits for memory in between.
  __device__ __forceinline__ void load_pre(const float* __restrict__ stats, int coff) {
    psc = *reinterpret_cast<const float4*>(stats + 2 * Cs + coff + kc4);
    psh = *reinterpret_cast<const float4*>(stats + 3 * Cs + coff + kc4);
  }
  __device__ __forceinline__ void init(const float* src_, int Hs_, int Ws_, int Cs_, const int* s_src, const int* s_hw) {
    const int tid = threadIdx.x;
    src = src_; Hs = Hs_; Ws = Ws_; Cs = Cs_;
    kc4 = (tid & 7) * 4;
    inb = 0u;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      base[i] = s_src[(tid >> 3) + RPP * i];
      hw[i] = s_hw[(tid >> 3) + RPP * i];
    }
  }
  // tap: this step's {dh,dw}; coff: channel offset inside the tap (multiple of 32)
  __device__ __forceinline__ void load(int dh, int dw, int coff) {
    const int dwe = SMALLC ? dw + (kc4 >> 2) : dw;
    const int rel = (dh * Ws + dwe) * Cs + (SMALLC ? 0 : coff + kc4);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int h = (hw[i] >> 16) + dh, w = (hw[i] & 0xffff) + dwe;
      if (hw[i] >= 0 && (unsigned)h < (unsigned)Hs && (unsigned)w < (unsigned)Ws) {
        reg[i] = *reinterpret_cast<const float4*>(src + (long)(base[i] + rel));
        if (PRE) inb |= 1u << i;
      } else {
        reg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (PRE) inb &= ~(1u << i);
      }
    }
  }
  __device__ __forceinline__ void store(float* S) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float4 v = reg[i];
      if (PRE && ((inb >> i) & 1u)) {      // the same operations as bn_apply_kernel: bit-identical to the materialised activation
        v.x = fmaxf(fmaf(v.x, psc.x, psh.x), 0.f); v.y = fmaxf(fmaf(v.y, psc.y, psh.y), 0.f);
        v.z = fmaxf(fmaf(v.z, psc.z, psh.z), 0.f); v.w = fmaxf(fmaf(v.w, psc.w, psh.w), 0.f);
      }
      *reinterpret_cast<float4*>(S + ((tid >> 3) + RPP * i) * LMKD_LDK + kc4) = v;
    }
  }
};

// Workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2) in linear id order.  The tile order is
// chosen so that what one XCD touches at a time fits its L2 (PMC: the naive order re-fetched every input row ~6x):
//   mode 0 (row bands): XCD x owns a contiguous band of row tiles and walks it in order, column tiles fastest.  The 3x3
//          halo rows shared by neighbouring row tiles and the A tile shared by the column tiles are then L2 hits.
//   mode 1 (column slices, n_ct a multiple of 8): XCD x owns column tiles x, x+8, ...: it keeps re-reading ONE slice of
//          the packed weights (layer 4: 9.4 MB of weights, 1.2 MB per 64-column slice) while streaming the small A.
// Pure speed hint: any placement computes the same tiles.
static int g_xcd_mode = -1;   // -1 auto
extern "C" int lmkd_conv_set_xcd_mode(int m) { g_xcd_mode = m; return LMKD_OK; }

__device__ __forceinline__ bool xcd_decode(int L, int n_rt, int n_ct, int mode, int& rt, int& ct) {
  const int x = L & 7, j = L >> 3;
  if (mode == 1) {
    ct = x + 8 * (j / n_rt);
    rt = j - (j / n_rt) * n_rt;
    return ct < n_ct;
  }
  const int chunk = (n_rt + 7) >> 3;
  ct = j % n_ct;
  const int rl = j / n_ct;
  rt = x * chunk + rl;
  return rl < chunk && rt < n_rt;
}
static inline int xcd_grid(int n_rt, int n_ct, int mode) {
  if (mode == 1) return n_rt * n_ct;
  return 8 * cdiv(n_rt, 8) * n_ct;
}

template <class Cfg, bool SMALLC, int STATS, bool BF16, bool PRE = false>   // STATS: 0 plain store, 1 + BatchNorm partial sums, 2 BatchNorm-affine (+residual, ReLU) epilogue
__global__ __launch_bounds__(Cfg::THREADS) void conv_gemm_kernel(ConvGemmArgs a) {
  using LA = LoaderConvGather<Cfg::BM, SMALLC, Cfg::THREADS, PRE>;
  using LB = LoaderKMajorDense<Cfg::BN, Cfg::THREADS>;   // packed weights are K-major: Wp[col][k]
  __shared__ __attribute__((aligned(16))) float smem[2 * (LA::LDS_FLOATS + LB::LDS_FLOATS)];
  __shared__ int s_src[Cfg::BM], s_hw[Cfg::BM], s_out[Cfg::BM];
  __shared__ float s_red[STATS == 1 ? Cfg::WM * Cfg::BN * 2 : 1];

  const int tid = threadIdx.x;
  int rt, ct;
  if (!xcd_decode(blockIdx.x, a.n_rt, a.n_ct, a.xcd_mode, rt, ct)) return;
  // parity classes interleaved (stride-2 data gradient: the classes run 1, 2, 2 and 4 taps): every XCD band and every
  // stretch of the dispatch order then carries the same mix of light and heavy tiles
  const int tile = rt / a.nclass;
  const int cls = rt - tile * a.nclass;
  const int ph = cls >> 1, pw = cls & 1;
  if (a.accum && a.ntap[cls] == 0) return;      // out += 0: a parity class no tap reaches (1x1 stride-2 data gradient)
  const int row0 = tile * Cfg::BM, n0 = ct * Cfg::BN;
  for (int r = tid; r < Cfg::BM; r += Cfg::THREADS) {
    const int m = row0 + r;
    bool ok = m < a.rows_per_class;
    int n = 0, aa = 0, bb = 0;
    if (ok) {
      n = fdiv(m, a.div_hw);
      const int rem = m - n * a.Hr * a.Wr;
      aa = fdiv(rem, a.div_w);
      bb = rem - aa * a.Wr;
      // odd H or W in the stride-2 data gradient: the odd-parity classes have one row/column fewer
      ok = (aa * a.omul + ph) < a.Ho && (bb * a.omul + pw) < a.Wo;
    }
    if (ok) {
      s_src[r] = ((n * a.Hs + aa * a.sh) * a.Ws + bb * a.sh) * a.Cs;
      s_hw[r] = ((aa * a.sh) << 16) | (bb * a.sh);
      s_out[r] = ((n * a.Ho + aa * a.omul + ph) * a.Wo + bb * a.omul + pw) * a.Co;
    } else {
      s_src[r] = 0;
      s_hw[r] = -1;
      s_out[r] = -1;
    }
  }
  __syncthreads();
  LA la;
  LB lb;
  la.init(a.src, a.Hs, a.Ws, a.Cs, s_src, s_hw);
  lb.init(a.wpk, a.Kp, n0, a.Co, a.Kp);

  f32x16 acc[Cfg::TM][Cfg::TN];
  const int nk = a.ntap[cls] * a.cps;
  {
    // hand-inlined gemm_mainloop: the two loaders take different per-step arguments
    constexpr int SA = LA::LDS_FLOATS, SB = LB::LDS_FLOATS;
    float* As0 = smem;
    float* Bs0 = smem + 2 * SA;
    const int lane = tid & 63, wave = tid >> 6;
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
    const Tap* taps = a.taps[cls];
    const int ntap_c = a.ntap[cls];
    auto issue = [&](int t) {
      // K order: channel chunk outer, tap inner.  The taps of one 32-channel chunk read the same 128-byte pixel segments (shifted
      // by a pixel), so the 9x reuse of a 3x3 conv sits in consecutive K-steps and stays in L1/L2; tap-major order re-fetched
      // the wide rows of layer 4 (2 KB per pixel) once per tap: PMC FETCH_SIZE 790-1080 MB per launch for 50 MB of operands.
      const int cc = t / ntap_c, tp = t - cc * ntap_c;
      const Tap tap = taps[tp];
      if (PRE && tp == 0) la.load_pre(a.pre_stats, cc * LMKD_BK);
      la.load(tap.dh, tap.dw, cc * LMKD_BK);
      lb.load(tap.kofs + cc * LMKD_BK);
    };
    if (nk > 0) {
      issue(0);
      la.store(As0);
      lb.store(Bs0);
      __syncthreads();
      // last K-step peeled: no condition sits between a tile's prefetch and its LDS store inside the loop (with one, hipcc
      // copies the prefetched registers in front of the MFMAs - and waits for the loads there - in the PRE instances)
      for (int t = 0; t + 1 < nk; ++t) {
        const int cur = t & 1;
        issue(t + 1);
        if (BF16) mfma_kstep_bf16<Cfg, LA, LB>(As0 + cur * SA, Bs0 + cur * SB, a_row, b_row, h, acc);
        else mfma_kstep<Cfg, LA, LB>(As0 + cur * SA, Bs0 + cur * SB, a_row, b_row, h, acc);
        // keep the store side (incl. the PRE loader's BatchNorm+ReLU arithmetic) BEHIND the MFMAs: hoisted above them it drags the
        // s_waitcnt vmcnt(0) of the prefetched tile in front of the matrix work and the prefetch hides nothing
        __builtin_amdgcn_sched_barrier(0);
        la.store(As0 + (cur ^ 1) * SA);
        lb.store(Bs0 + (cur ^ 1) * SB);
        __syncthreads();
      }
      const int last = (nk - 1) & 1;
      if (BF16) mfma_kstep_bf16<Cfg, LA, LB>(As0 + last * SA, Bs0 + last * SB, a_row, b_row, h, acc);
      else mfma_kstep<Cfg, LA, LB>(As0 + last * SA, Bs0 + last * SB, a_row, b_row, h, acc);
    }
  }

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int cl0 = wn * (Cfg::TN * 32) + (lane & 31);
  if (STATS == 0 && a.accum) {
    // out += acc: fetch the whole tile's previous values first (independent loads in flight together), then add; the
    // per-element "load, wait, add, store" form serialised 16 dependent global round trips per wave
    float prev[Cfg::TM][Cfg::TN][16];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ob = s_out[wm * (Cfg::TM * 32) + i * 32 + acc_row(e, lane)];
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) {
          const int col = n0 + cl0 + j * 32;
          prev[i][j][e] = (ob >= 0 && col < a.Co) ? a.out[(long)ob + col] : 0.f;
        }
      }
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = acc[i][j][e] + prev[i][j][e];
  }
  float s1[Cfg::TN], s2[Cfg::TN];
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) s1[j] = s2[j] = 0.f;
  float esc[Cfg::TN], esh[Cfg::TN];
  if (STATS == 2) {
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const int col = n0 + cl0 + j * 32;
      esc[j] = col < a.Co ? a.ep_stats[2 * a.Co + col] : 0.f;
      esh[j] = col < a.Co ? a.ep_stats[3 * a.Co + col] : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rl = wm * (Cfg::TM * 32) + i * 32 + acc_row(e, lane);
      const int ob = s_out[rl];
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
        const int col = n0 + cl0 + j * 32;
        float v = acc[i][j][e];
        if (ob >= 0 && col < a.Co) {
          float* o = a.out + (long)ob + col;
          if (STATS == 2) {   // same operations, in the same order, as bn_apply_kernel: bit-identical to the two-pass form
            v = fmaf(v, esc[j], esh[j]);
            if (a.ep_res) v += a.ep_res[(long)ob + col];
            if (a.ep_relu) v = fmaxf(v, 0.f);
            *o = v;
          } else {
            *o = (STATS == 1 && a.accum) ? v + *o : v;      // STATS == 0: the previous values were added above
          }
        }
        if (STATS == 1) { s1[j] += v; s2[j] = fmaf(v, v, s2[j]); }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (STATS == 1) {
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
      const float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
      if (lane < 32) {
        s_red[(wm * Cfg::BN + cl0 + j * 32) * 2 + 0] = t1;
        s_red[(wm * Cfg::BN + cl0 + j * 32) * 2 + 1] = t2;
      }
    }
  }
  if (STATS == 1) {
    __syncthreads();
    if (tid < Cfg::BN && n0 + tid < a.Co) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < Cfg::WM; ++w) {
        s1 += s_red[(w * Cfg::BN + tid) * 2 + 0];
        s2 += s_red[(w * Cfg::BN + tid) * 2 + 1];
      }
      float* p = a.stat_partial + ((long)rt * a.Co + n0 + tid) * 2;
      p[0] = s1;
      p[1] = s2;
    }
  }
}

#include "conv_x3.h"
#include "h2.h"
#include "conv_patch.h"
#include "conv_patch16.h"
#include "conv_stem.h"

// ---------------------------------------------------------------------------------
// weight gradient: rows = co (dy, K-outer), cols = packed (tap, ci), K = pixels, split-K slabs
// ---------------------------------------------------------------------------------
struct WgradArgs {
  const float* dy;
  const float* x;
  float* slab;
  int N, Hs, Ws, Cs, Ho, Wo, Co;
  int stride, pad, KH, KW, KWp;
  int Kp, Mpix, steps_total, steps_per_split;
  int n_mt, n_jt, splits, xcd_mode;   // 1-D grid decode (xcd_mode 1: all tiles of pixel split z run on XCD z % 8)
  const float* pre_stats;             // PRE: x is a raw conv output; the loader applies relu(BatchNorm(x)) (see ConvGemmArgs)
  int seg_pix0, seg_splits0, sps1;    // two frame segments (wgrad_win.h WgradWinArgs): conv_wgrad_x3_kernel only; one segment: Mpix, splits, 0
  FastDiv div_hw, div_w;
  const unsigned* h2_xw;              // two-plane fp16 form (conv_wgrad_x3_kernel<.., 3, ..>): the maxima of x / dy (lmkd_conv_operand_amax)
  const unsigned* h2_dyw;
};

// split z -> first pixel, pixel limit, number of 32-pixel steps, segment
struct WinSlab { int kfirst, kend, nk, seg; };
__device__ __forceinline__ WinSlab win_slab(int z, int Mpix, int sps0, int seg_pix0, int seg_splits0, int sps1) {
  WinSlab w;
  int sps;
  if (z < seg_splits0) { w.kfirst = z * sps0 * LMKD_BK; w.kend = seg_pix0; w.seg = 0; sps = sps0; }
  else { w.kfirst = seg_pix0 + (z - seg_splits0) * sps1 * LMKD_BK; w.kend = Mpix; w.seg = 1; sps = sps1; }
  const int left = (w.kend - w.kfirst + LMKD_BK - 1) / LMKD_BK;
  w.nk = left < sps ? (left < 0 ? 0 : left) : sps;
  return w;
}

template <int ROWS, bool SMALLC, int THREADS = LMKD_THREADS, bool PRE = false>
struct LoaderWgradGather {
  static constexpr bool ROWK = false;
  static constexpr int LD = ROWS;
  static constexpr int LDS_FLOATS = LMKD_BK * LD;
  static constexpr int CPR = ROWS / 4;
  static constexpr int KPP = THREADS / CPR;
  static constexpr int NI = LMKD_BK / KPP;
  const float* x;
  FastDiv div_hw, div_w;
  int Mpix, HoWo, Wo, Hs, Ws, Cs, stride;
  float4 reg[NI];
  int dh, dw, coff, r4;
  bool tap_ok;
  float4 psc, psh;
  unsigned inb;      // PRE: rows of the prefetched slab that are real pixels; the transform runs in store(), behind the MFMAs
  __device__ __forceinline__ void init(const WgradArgs& a, int j0) {
    inb = 0u;
    x = a.x; div_hw = a.div_hw; div_w = a.div_w;
    Mpix = a.Mpix; HoWo = a.Ho * a.Wo; Wo = a.Wo; Hs = a.Hs; Ws = a.Ws; Cs = a.Cs; stride = a.stride;
    const int tid = threadIdx.x;
    r4 = (tid % CPR) * 4;
    const int col = j0 + r4;
    if (SMALLC) {  // col -> (kh = col/32, kw = (col%32)/4), 4 channels per tap
      const int kh = col >> 5, kw = (col & 31) >> 2;
      dh = kh - a.pad; dw = kw - a.pad; coff = 0;
      tap_ok = kh < a.KH && kw < a.KW && col < a.Kp;
    } else {
      const int tap = col / a.Cs;
      coff = col - tap * a.Cs;
      const int kh = tap / a.KWp, kw = tap - kh * a.KWp;
      dh = kh - a.pad; dw = kw - a.pad;
      tap_ok = col < a.Kp && kw < a.KW;
    }
    if (PRE && tap_ok) {      // this thread's 4 channels never change over the pixel loop
      psc = *reinterpret_cast<const float4*>(a.pre_stats + 2 * a.Cs + coff);
      psh = *reinterpret_cast<const float4*>(a.pre_stats + 3 * a.Cs + coff);
    }
  }
  __device__ __forceinline__ void load(int koff) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = koff + tid / CPR + KPP * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tap_ok && p < Mpix) {
        const int n = fdiv(p, div_hw);
        const int rem = p - n * HoWo;
        const int oh = fdiv(rem, div_w);
        const int ow = rem - oh * Wo;
        const int h = oh * stride + dh, w = ow * stride + dw;
        if ((unsigned)h < (unsigned)Hs && (unsigned)w < (unsigned)Ws) {
          v = *reinterpret_cast<const float4*>(x + ((long)(n * Hs + h) * Ws + w) * Cs + coff);
          if (PRE) inb |= 1u << i;
        } else if (PRE) inb &= ~(1u << i);
      } else if (PRE) inb &= ~(1u << i);
      reg[i] = v;
    }
  }
  __device__ __forceinline__ void store(float* S) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int k = tid / CPR + KPP * i;
      float4 v = reg[i];
      if (PRE && ((inb >> i) & 1u)) {
        v.x = fmaxf(fmaf(v.x, psc.x, psh.x), 0.f); v.y = fmaxf(fmaf(v.y, psc.y, psh.y), 0.f);
        v.z = fmaxf(fmaf(v.z, psc.z, psh.z), 0.f); v.w = fmaxf(fmaf(v.w, psc.w, psh.w), 0.f);
      }
      *reinterpret_cast<float4*>(S + k * LD + r4) = v;
    }
  }
};

template <class Cfg, bool SMALLC, bool BF16, bool PRE = false>
__global__ __launch_bounds__(Cfg::THREADS) void conv_wgrad_kernel(WgradArgs a) {
  using LA = LoaderMMajorDense<Cfg::BM, Cfg::THREADS>;
  using LB = LoaderWgradGather<Cfg::BN, SMALLC, Cfg::THREADS, PRE>;
  __shared__ __attribute__((aligned(16))) float smem[2 * (LA::LDS_FLOATS + LB::LDS_FLOATS)];
  // Tile order.  The 9 column tiles (taps) of one pixel split read the same x rows and the same dy rows; in plain order
  // they land on different XCDs and every XCD fetches those rows again (PMC: 2.6-3.1 GB per launch on the 160 MB layers).
  // With many splits, all tiles of split z are given ids congruent mod 8 so that one XCD's L2 serves them.
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
  la.init(a.dy, a.Co, m0, a.Co, a.Mpix);
  lb.init(a, j0);
  const int s0 = z * a.steps_per_split;
  int nk = a.steps_total - s0;
  if (nk > a.steps_per_split) nk = a.steps_per_split;
  f32x16 acc[Cfg::TM][Cfg::TN];
  auto koff = [s0](int t) { return (s0 + t) * LMKD_BK; };
  gemm_mainloop<Cfg, LA, LB, decltype(koff), decltype(koff), BF16, PRE>(la, lb, nk, koff, koff, smem, acc);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  float* C = a.slab + (long)z * a.Co * a.Kp;
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const int col = j0 + wn * (Cfg::TN * 32) + j * 32 + (lane & 31);
    if (col >= a.Kp) continue;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * (Cfg::TM * 32) + i * 32 + acc_row(e, lane);
        if (row < a.Co) C[(long)row * a.Kp + col] = acc[i][j][e];
      }
  }
}

// dW (OIHW) = sum over split slabs of slab[z][co][(kh*KWp+kw)*Cs + ci]  (coalesced slab reads, 8 loads in flight per
// thread; the scattered 4-byte OIHW writes are ~11 M floats per trunk call in total)
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splits, int Co, int Cin,
                                    int Cs, int KH, int KW, int KWp, int Kp, int accumulate) {
  const long total = (long)Co * KH * KW * Cin;
  const long zs = (long)Co * Kp;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int ci = idx % Cin;
    long r = idx / Cin;
    const int kw = r % KW; r /= KW;
    const int kh = r % KH;
    const int co = r / KH;
    const float* p = slab + (long)co * Kp + (long)(kh * KWp + kw) * Cs + ci;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 8 <= splits; z += 8) {
      const float a0 = p[(z + 0) * zs], a1 = p[(z + 1) * zs], a2 = p[(z + 2) * zs], a3 = p[(z + 3) * zs];
      const float a4 = p[(z + 4) * zs], a5 = p[(z + 5) * zs], a6 = p[(z + 6) * zs], a7 = p[(z + 7) * zs];
      s0 += a0; s1 += a1; s2 += a2; s3 += a3; s0 += a4; s1 += a5; s2 += a6; s3 += a7;
    }
    for (; z < splits; ++z) s0 += p[z * zs];
    float* o = dw + (((long)co * Cin + ci) * KH + kh) * KW + kw;
    const float v = (s0 + s1) + (s2 + s3);
    *o = accumulate ? *o + v : v;      // accumulate: dw is weight.grad itself (gradient accumulation over trunk calls / episodes)
  }
}

// The same reduction for the stem (round 5; Kp = 7 x 8 x 4 = 224 <= 256, several hundred slabs - one per workgroup of stem_wgrad_kernel): one workgroup per
// output channel, thread (k, q) adds the slabs q, q + 4, ... of K position k (coalesced rows of Kp floats), the four partial sums meet in
// LDS.  wgrad_reduce_kernel gave the 9 408 weights one thread each for a walk over every slab: 184 us.
__global__ __launch_bounds__(1024) void wgrad_reduce_stem_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splits, int Co, int Cin,
                                                                 int Cs, int KH, int KW, int KWp, int Kp, int accumulate) {
  __shared__ float sm[4][256];
  const int k = threadIdx.x, q = threadIdx.y, co = blockIdx.x;
  const long zs = (long)Co * Kp;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (k < Kp) {
    const float* p = slab + (long)co * Kp + k;
    int z = q;
    for (; z + 28 < splits; z += 32) {
      const float a0 = p[(z + 0) * zs], a1 = p[(z + 4) * zs], a2 = p[(z + 8) * zs], a3 = p[(z + 12) * zs];
      const float a4 = p[(z + 16) * zs], a5 = p[(z + 20) * zs], a6 = p[(z + 24) * zs], a7 = p[(z + 28) * zs];
      s0 += a0; s1 += a1; s2 += a2; s3 += a3; s0 += a4; s1 += a5; s2 += a6; s3 += a7;
    }
    for (; z < splits; z += 4) s0 += p[z * zs];
  }
  sm[q][k] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0 && k < Kp) {
    const int ci = k % Cs, r = k / Cs;
    const int kw = r % KWp, kh = r / KWp;
    if (ci < Cin && kw < KW && kh < KH) {
      const float v = (sm[0][k] + sm[1][k]) + (sm[2][k] + sm[3][k]);
      float* o = dw + (((long)co * Cin + ci) * KH + kh) * KW + kw;
      *o = accumulate ? *o + v : v;
    }
  }
}
// (a form that transposed 64 x 9 results through LDS into contiguous OIHW runs, one workgroup per output channel and 64 input channels,
// measured 48 % SLOWER than this kernel's scattered 4-byte stores - 9.87 vs 6.66 ms over the 247 non-stem launches of 13 episodes,
// profiles/r05_kernel_stats_serial_f32.csv of that build: the time is the slabs' read, and a third of its lanes idled in the last pass)
static void launch_wgrad_reduce(const float* slab, float* dw, int splits, int Co, int Cin, int Cs, int KH, int KW, int KWp, int Kp,
                                int accumulate, hipStream_t s) {
  const long total = (long)Co * KH * KW * Cin;
  int rg = cdiv(total, 256);
  if (rg > 4096) rg = 4096;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rg), dim3(256), 0, s, slab, dw, splits, Co, Cin, Cs, KH, KW, KWp, Kp, accumulate);
}

#include "wgrad_x3.h"
#include "wgrad_win.h"
#include "wgrad_stem.h"

// K-major packed weights (the GEMM's B operand: one row of K per output column):
// mode 0 (forward): Wp[co][(kh*KWp+kw)*Cs + ci] = W[co][ci][kh][kw]   (zero for padded kw / ci)
// mode 1 (dgrad)  : Wd[ci][(kh*KW+kw)*Co + co] = W[co][ci][kh][kw]
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Cin, int Cs, int KH,
                                    int KW, int KWp, int mode) {
  if (mode == 0) {
    const long total = (long)Co * KH * KWp * Cs;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
      const int ci = idx % Cs;
      long r = idx / Cs;
      const int kw = r % KWp; r /= KWp;
      const int kh = r % KH;
      const int co = r / KH;
      wp[idx] = (ci < Cin && kw < KW) ? w[(((long)co * Cin + ci) * KH + kh) * KW + kw] : 0.f;
    }
  } else {
    const long total = (long)Cin * KH * KW * Co;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
      const int co = idx % Co;
      long r = idx / Co;
      const int kw = r % KW; r /= KW;
      const int kh = r % KH;
      const int ci = r / KH;
      wp[idx] = w[(((long)co * Cin + ci) * KH + kh) * KW + kw];
    }
  }
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
static inline int conv_out(int H, int K, int s, int p) { return (H + 2 * p - K) / s + 1; }
static inline int kw_padded(int Cs, int KW) { return Cs >= 32 ? KW : ((KW * Cs + 31) / 32) * 32 / Cs; }

extern "C" long lmkd_conv2d_packed_weight_elems(int Cout, int Cin, int Cs, int KH, int KW, int mode) {
  if (mode == 0) return (long)Cout * KH * kw_padded(Cs, KW) * Cs;
  return (long)Cin * KH * KW * Cout;
}

extern "C" int lmkd_conv2d_pack_weights(const float* w_oihw, float* wp, int Cout, int Cin, int Cs, int KH, int KW,
                                        int mode, void* stream) {
  LMKD_REQUIRE(w_oihw && wp, "lmkd_conv2d_pack_weights: null pointer");
  LMKD_REQUIRE(mode == 0 || mode == 1, "lmkd_conv2d_pack_weights: mode must be 0 (fwd) or 1 (dgrad)");
  LMKD_REQUIRE(Cs >= Cin && (Cs % 32 == 0 || Cs == 4), "lmkd_conv2d_pack_weights: Cs=%d must be 4 or a multiple of 32", Cs);
  const long total = lmkd_conv2d_packed_weight_elems(Cout, Cin, Cs, KH, KW, mode);
  int grid = cdiv(total, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, wp, Cout, Cin, Cs, KH, KW,
                     kw_padded(Cs, KW), mode);
  LMKD_CHECK_LAUNCH("lmkd_conv2d_pack_weights");
  return LMKD_OK;
}

// ---- compute dtype of the convolutions: 0 = exact fp32 MFMA, 1 = bf16 MFMA inputs with fp32 accumulation, 2 (DEFAULT since round 3:
// the arithmetic bench.py's headline line runs) = fp32 as three bf16 planes, six products, 3 = nine products ----
static int g_conv_bf16 = 0;
static int g_conv_patch = 1;   // same-size convolutions of the bf16-plane modes from an LDS-resident patch (conv_patch.h); 0 = im2col gather
extern "C" int lmkd_conv_set_patch(int on) { g_conv_patch = on ? 1 : 0; return LMKD_OK; }
static int g_conv_x3 = 6;   // 0 | 6 | 9 bf16 MFMA products per fp32 product (conv_x3.h); library default: 6
// mode 4: fp32 as TWO fp16 planes, three products (conv_patch16.h: h2_split4) in the trunk's convolution kernels - the 3x3 forward / data
// gradient (conv_patch16_x3_kernel), the weight gradients (conv_wgrad_win16_kernel, conv_wgrad_x3_kernel, stem_wgrad_kernel), the stem's
// forward (conv_stem_patch_kernel) - wherever the caller names the operands' maxima (lmkd_conv_operand_amax); every other launch (the
// 1x1 forward, the inference epilogues, the loader-side BatchNorm), and every launch without them, runs mode 2.
static int g_conv_h2 = 0;
static std::atomic<long> g_h2_launches{0};       // launches that took the two-plane form (tests; the forward's thread and the autograd engine's both launch)
static std::atomic<long> g_h2_fallbacks{0};      // mode-4 launches whose caller withheld a maximum because the range fence flagged the tensor (LMKD_AMAX_FENCED)
extern "C" long lmkd_conv_h2_launches(void) { return g_h2_launches.load(); }
extern "C" long lmkd_conv_h2_fallbacks(void) { return g_h2_fallbacks.load(); }
static inline void note_fenced(const lmkd_amax_desc* am) { if (g_conv_h2 && am && (am->flags & LMKD_AMAX_FENCED)) ++g_h2_fallbacks; }
extern "C" int lmkd_conv_set_compute_dtype(int mode) {
  LMKD_REQUIRE(mode >= 0 && mode <= 4, "lmkd_conv_set_compute_dtype: 0 fp32 MFMA, 1 bf16, 2 fp32 as 3xbf16 (6 products), 3 (9 products), 4 fp32 as 2xfp16 (3 products) over mode 2");
  g_conv_bf16 = mode == 1;
  g_conv_x3 = (mode == 2 || mode == 4) ? 6 : (mode == 3 ? 9 : 0);
  g_conv_h2 = mode == 4;
  return LMKD_OK;
}
extern "C" int lmkd_conv_get_compute_dtype(void) { return g_conv_bf16 ? 1 : (g_conv_h2 ? 4 : (g_conv_x3 == 6 ? 2 : (g_conv_x3 == 9 ? 3 : 0))); }
// The maxima of a launch's operands travel as explicit arguments (lmkd_amax_desc on the *_seg entry points; round 5 - before: one-shot
// thread-local channels).  x_words / dy_words: device words holding the fp32 bits of max |x| / max |dy| per frame segment (upper bounds are
// valid: a larger word costs range, not correctness), complete in stream order before the launch; either may be null; only mode 4 reads
// them.  out_words (forward): the launch also folds max |y| into these words - in the epilogue of conv_patch16_x3_kernel, or by a
// reduction pass of its own behind any other kernel - so that lmkd_bn_finalize_seg can bound relu(BatchNorm(y)) for a consumer that
// applies the BatchNorm in its loader.
static thread_local bool t_amax_recorded = false;      // (inside one call) set by the launch macros whose kernel folds ConvGemmArgs::amax_out
// elements (16-bit) of the plane buffer of a packed weight of ncols x Kp in the current mode
extern "C" long lmkd_conv2d_plane_elems(int ncols, int Kp) {
  const long n = (long)ncols * Kp;
  return g_conv_bf16 ? n : (g_conv_h2 ? 18 * n + 32 : 12 * n);
}
// max |x| of n floats: slots = 0: into the ONE word at `word` (weight packs); slots = 1: into the slot words of one frame segment of an
// activation maximum (lmkd_amax_next's layout: segment s at word + s * lmkd_amax_words() / 2).  The words are zeroed here first.
static int amax_impl(const float* x, long n, void* word, int slots, void* stream, bool zero = true) {
  LMKD_REQUIRE(x && word && n > 0, "lmkd_amax: bad arguments");
  if (zero && hipMemsetAsync(word, 0, slots ? LMKD_AMAX_SEG_WORDS * 4 : 4, (hipStream_t)stream) != hipSuccess) {
    lmkd_set_error("lmkd_amax: hipMemsetAsync failed");
    return LMKD_EHIP;
  }
  int grid = cdiv(n, 256 * 16);
  if (grid > 2048) grid = 2048;
  if (slots) hipLaunchKernelGGL(amax_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, (unsigned*)word);
  else hipLaunchKernelGGL(amax_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, (unsigned*)word);
  LMKD_CHECK_LAUNCH("amax_kernel");
  return LMKD_OK;
}
extern "C" int lmkd_amax(const float* x, long n, void* word, void* stream) { return amax_impl(x, n, word, 1, stream); }

extern "C" int lmkd_conv2d_split_weights(const float* wp, void* wf, int ncols, int Kp, void* stream) {
  LMKD_REQUIRE(wp && wf, "lmkd_conv2d_split_weights: null pointer");
  LMKD_REQUIRE(ncols > 0 && ncols % 32 == 0 && Kp > 0 && Kp % 32 == 0, "lmkd_conv2d_split_weights: ncols=%d and Kp=%d must be multiples of 32", ncols, Kp);
  LMKD_REQUIRE((long)ncols * Kp * 24 < 0xffffffe0L, "lmkd_conv2d_split_weights: weights exceed the 4 GiB buffer range");
  int grid = cdiv((long)ncols * Kp, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(split_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, wp, (unsigned short*)wf, ncols, Kp,
                     g_conv_bf16 ? 1 : 3);
  LMKD_CHECK_LAUNCH("split_weights_kernel");
  if (!g_conv_bf16) {      // three-plane modes: the 16x16x32 fragment order behind the two 32x32x16 copies (conv_patch16.h)
    hipLaunchKernelGGL(split_weights16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, wp, (unsigned short*)wf + (long)ncols * Kp * 6, ncols, Kp);
    LMKD_CHECK_LAUNCH("split_weights16_kernel");
  }
  if (g_conv_h2) {         // + the two fp16 planes (16x16x32 order) and max |w| behind them
    unsigned short* wh = (unsigned short*)wf + (long)ncols * Kp * 12;
    const int rc = amax_impl(wp, (long)ncols * Kp, wh + (long)ncols * Kp * 4, 0, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(split_weights16_h2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, wp, wh, ncols, Kp);
    LMKD_CHECK_LAUNCH("split_weights16_h2_kernel");
    hipLaunchKernelGGL(split_weights_h2_32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, wp, wh, ncols, Kp);
    LMKD_CHECK_LAUNCH("split_weights_h2_32_kernel");
  }
  return LMKD_OK;
}

// Re-pack MANY convolution weights in one launch (after an optimizer step: 40 cached packs = 120 pack / split launches before).
// Entry e: OIHW weight -> the fragment-order planes lmkd_conv2d_pack_weights + lmkd_conv2d_split_weights would write for (Cs, mode) in
// the current arithmetic (modes 1-3), bit for bit; blockIdx.y = entry.
struct RepackEntry {
  const float* w;
  unsigned short* wf;
  int Co, Cin, Cs, KH, KW, KWp, mode, ncols, Kp;
};
#define LMKD_REPACK_MAX 40
struct RepackArgs {
  RepackEntry e[LMKD_REPACK_MAX];
  int npl;
};
// element (col, k) of the packed weight of entry e (pack_weights_kernel)
__device__ __forceinline__ float repack_fetch(const RepackEntry& e, int col, int k) {
  if (e.mode == 0) {      // mode 0: col = co, k = (kh * KWp + kw) * Cs + ci
    const int ci = k % e.Cs;
    const int r = k / e.Cs;
    const int kw = r % e.KWp, kh = r / e.KWp;
    return (ci < e.Cin && kw < e.KW) ? e.w[(((long)col * e.Cin + ci) * e.KH + kh) * e.KW + kw] : 0.f;
  }
  // mode 1: col = ci, k = (kh * KW + kw) * Co + co
  const int co = k % e.Co;
  const int r = k / e.Co;
  const int kw = r % e.KW, kh = r / e.KW;
  return e.w[(((long)co * e.Cin + col) * e.KH + kh) * e.KW + kw];
}
// mode 4: the fp16 planes behind the bf16 ones, in three launches: phase 0 zeroes the entries' max |w| words, 1 folds the maxima, 2 splits
__global__ void repack_h2_kernel(RepackArgs a, int phase) {
  const RepackEntry& e = a.e[blockIdx.y];
  const long total = (long)e.ncols * e.Kp;
  unsigned short* wh = e.wf + total * 12;
  unsigned* word = reinterpret_cast<unsigned*>(wh + total * 4);
  if (phase == 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *word = 0u;
    return;
  }
  if (phase == 1) {
    float m = 0.f;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
      const int col = (int)(idx / e.Kp), k = (int)(idx - (long)col * e.Kp);
      m = fmaxf(m, fabsf(repack_fetch(e, col, k)));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(word, __float_as_uint(m));
    return;
  }
  const float s = h2_scale(*word);
  const int G16 = e.Kp >> 5, G = e.Kp >> 4;
  unsigned short* wh32 = wh + total * 4 + 32;      // split_weights_h2_32_kernel
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int col = (int)(idx / e.Kp), k = (int)(idx - (long)col * e.Kp);
    const int lane16 = (col & 15) + 16 * patch16_kslot((k >> 3) & 3);
    const long q = ((((long)(col >> 4) * G16 + (k >> 5)) * 2) * 64 + lane16) * 8 + (k & 7);
    unsigned short p0, p1;
    h2_split1(repack_fetch(e, col, k) * s, p0, p1);
    wh[q] = p0; wh[q + 512] = p1;
    const long q2 = q + total * 2;
    wh[q2] = p0 ^ 0x8000u; wh[q2 + 512] = p1 ^ 0x8000u;
    const long o = ((((long)(col >> 5) * G + (k >> 4)) * 2) * 64 + (col & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7);
    wh32[o] = p0; wh32[o + 512] = p1;
  }
}
__global__ void repack_multi_kernel(RepackArgs a) {
  const RepackEntry& e = a.e[blockIdx.y];
  const long total = (long)e.ncols * e.Kp;
  const int G = e.Kp >> 4, G16 = e.Kp >> 5;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int col = (int)(idx / e.Kp), k = (int)(idx - (long)col * e.Kp);
    const float x = repack_fetch(e, col, k);
    const int lane = (col & 31) + 32 * ((k >> 3) & 1);
    const long o = ((((long)(col >> 5) * G + (k >> 4)) * a.npl) * 64 + lane) * 8 + (k & 7);
    if (a.npl == 1) {
      union { __bf16 h; unsigned short u; } c;
      c.h = (__bf16)x;
      e.wf[o] = c.u;
      continue;
    }
    const unsigned b0 = __float_as_uint(x);
    const float r1 = x - __uint_as_float(b0 & 0xffff0000u);
    const unsigned b1 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(b1 & 0xffff0000u);
    const unsigned short p0 = (unsigned short)(b0 >> 16), p1 = (unsigned short)(b1 >> 16), p2 = (unsigned short)(__float_as_uint(r2) >> 16);
    const unsigned short n0 = p0 ^ 0x8000u, n1 = p1 ^ 0x8000u, n2 = p2 ^ 0x8000u;
    e.wf[o] = p0; e.wf[o + 512] = p1; e.wf[o + 1024] = p2;                        // split_weights_kernel: W, then -W
    const long o2 = o + total * 3;
    e.wf[o2] = n0; e.wf[o2 + 512] = n1; e.wf[o2 + 1024] = n2;
    unsigned short* w16 = e.wf + total * 6;                                         // split_weights16_kernel behind them
    const int lane16 = (col & 15) + 16 * patch16_kslot((k >> 3) & 3);
    const long q = ((((long)(col >> 4) * G16 + (k >> 5)) * 3) * 64 + lane16) * 8 + (k & 7);
    w16[q] = p0; w16[q + 512] = p1; w16[q + 1024] = p2;
    const long q2 = q + total * 3;
    w16[q2] = n0; w16[q2 + 512] = n1; w16[q2 + 1024] = n2;
  }
}

// ws[i], wfs[i]: OIHW weight and destination planes of entry i; dims[i * 6 ..] = Cout, Cin, Cs, KH, KW, mode.  Modes 1-3 only.
extern "C" int lmkd_conv2d_repack_multi(const float* const* ws, void* const* wfs, const int* dims, int n, void* stream) {
  LMKD_REQUIRE(ws && wfs && dims && n > 0, "lmkd_conv2d_repack_multi: null pointer");
  LMKD_REQUIRE(g_conv_x3 || g_conv_bf16, "lmkd_conv2d_repack_multi: fragment-order planes exist in the bf16-plane modes only");
  for (int i0 = 0; i0 < n; i0 += LMKD_REPACK_MAX) {
    RepackArgs a;
    memset(&a, 0, sizeof(a));
    a.npl = g_conv_bf16 ? 1 : 3;
    const int m = std::min(LMKD_REPACK_MAX, n - i0);
    long most = 0;
    for (int j = 0; j < m; ++j) {
      const int* d = dims + (long)(i0 + j) * 6;
      RepackEntry& e = a.e[j];
      e.w = ws[i0 + j]; e.wf = (unsigned short*)wfs[i0 + j];
      e.Co = d[0]; e.Cin = d[1]; e.Cs = d[2]; e.KH = d[3]; e.KW = d[4]; e.mode = d[5];
      LMKD_REQUIRE(e.w && e.wf && (e.mode == 0 || e.mode == 1), "lmkd_conv2d_repack_multi: bad entry %d", i0 + j);
      LMKD_REQUIRE(e.Cs >= e.Cin && (e.Cs % 32 == 0 || e.Cs == 4), "lmkd_conv2d_repack_multi: Cs=%d must be 4 or a multiple of 32", e.Cs);
      e.KWp = kw_padded(e.Cs, e.KW);
      e.ncols = e.mode == 0 ? e.Co : e.Cin;
      e.Kp = e.mode == 0 ? e.KH * e.KWp * e.Cs : e.KH * e.KW * e.Co;
      LMKD_REQUIRE(e.ncols % 32 == 0 && e.Kp % 32 == 0, "lmkd_conv2d_repack_multi: entry %d: ncols=%d and Kp=%d must be multiples of 32", i0 + j, e.ncols, e.Kp);
      most = std::max(most, (long)e.ncols * e.Kp);
    }
    int gx = cdiv(most, 256 * 4);
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(repack_multi_kernel, dim3(gx, m), dim3(256), 0, (hipStream_t)stream, a);
    LMKD_CHECK_LAUNCH("repack_multi_kernel");
    if (g_conv_h2) {
      for (int phase = 0; phase < 3; ++phase) {
        hipLaunchKernelGGL(repack_h2_kernel, dim3(phase == 0 ? 1 : gx, m), dim3(256), 0, (hipStream_t)stream, a, phase);
        LMKD_CHECK_LAUNCH("repack_h2_kernel");
      }
    }
  }
  return LMKD_OK;
}

// ---- tile configuration -------------------------------------------------------------------------
// id: 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 64x128 (rows x cols, 4 waves), 5 = 128x128 and 6 = 128x64 with 8 waves (2 per SIMD).
// Workgroups per CU by LDS: 2 / 2 / 4 / 2 / 2.
static int g_tile_override = 0;
extern "C" int lmkd_conv_set_tile(int id) {
  LMKD_REQUIRE(id >= 0 && id <= 14, "lmkd_conv_set_tile: id must be 0 (auto) .. 14");
  g_tile_override = id;
  return LMKD_OK;
}
// 7..10: the bf16-plane kernels (conv_x3.h, conv_patch.h), 8 waves: 7 = 256x64, 8 = 128x128, 9 = 128x64, 10 = 256x128 (patch kernel only)
// 11 / 12 (patch kernel only): 128x64 / 128x128 with FOUR waves, three / two workgroups per CU
// 13 / 14 (patch kernel, one-plane modes only): 256x128 / 256x64 with FOUR waves (wave tiles 128x64 / 128x32)
static inline int cfg_bm(int id) { return (id == 7 || id == 10 || id == 13 || id == 14) ? 256 : ((id <= 2 || id >= 5) ? 128 : 64); }
static inline int cfg_bn(int id) { return (id == 1 || id == 4 || id == 5 || id == 8 || id == 10 || id == 12 || id == 13) ? 128 : 64; }
static inline int cfg_wg_per_cu(int id) { return id == 1 ? 2 : (id == 3 ? 4 : 3); }

// Pick the tile that minimises ceil(tiles / 256 CUs) * work per tile: at 64 cycles per fp32 MFMA every configuration is
// matrix-pipe bound, so what differs is how evenly the launch's tiles divide over the CUs (the tail).
// same: same-size convolution served by the LDS-patch kernel (patch_eligible)
static int pick_conv_cfg(long rows_per_class, int nclass, int ncols, bool same = false) {
  if (g_conv_x3 || g_conv_bf16) {
    if (g_tile_override >= 7) {
      int id = g_tile_override;
      if (id == 10 && !same) id = 8;
      if (id == 11 && !same) id = 9;
      if (id == 12 && !same) id = 8;
      if ((id == 13 || id == 14) && (!same || !g_conv_bf16)) id = id == 13 ? 10 : 7;
      if (id == 10 && !same) id = 8;
      if (ncols <= 64 && cfg_bn(id) == 128) id = id == 10 ? 7 : (id == 12 ? 11 : (id == 13 ? 14 : 9));
      return id;
    }
    if (same) {
      // patch kernel, measured at 200 frames (tools/patch_bench.py): four-wave workgroups, two or three per CU, beat one eight-wave
      // workgroup - a workgroup's prologue / epilogue / chunk refill runs under the other workgroups' MFMAs.  128x128 (wave tile
      // 64x64: half the LDS reads per MFMA) in the three-plane modes once it gives every CU at least two rounds of tiles, 128x64 otherwise.
      if (ncols <= 64) return 11;
      const long t128 = cdiv(rows_per_class, 128) * cdiv(ncols, 128);
      return (g_conv_x3 && t128 >= 1024) ? 12 : 11;      // one-plane modes: 128x64 throughout (three or four workgroups per CU)
    }
    if (ncols <= 64) return 9;
    // measured (tools/conv_bench_x3.py): 128x128 (one workgroup per CU) wins once the launch has a tile per CU
    return (long)nclass * cdiv(rows_per_class, 128) * cdiv(ncols, 128) >= 256 ? 8 : 9;
  }
  if (g_tile_override) {
    if (ncols <= 64 && cfg_bn(g_tile_override) == 128) return g_tile_override == 5 ? 6 : (cfg_bm(g_tile_override) == 128 ? 2 : 3);
    return g_tile_override;
  }
  int best = 0;
  double best_cost = 1e30;
  const int cand[4] = {5, 2, 4, 3};
  for (int c = 0; c < 4; ++c) {
    const int id = cand[c];
    if (ncols <= 64 && cfg_bn(id) == 128) continue;
    const long tiles = (long)nclass * cdiv(rows_per_class, cfg_bm(id)) * cdiv(ncols, cfg_bn(id));
    // matrix-pipe-bound model: a CU's time is the MFMA work of the tiles it receives, whatever their concurrency.
    // Measured per layer (profiles/): what matters is 4 waves per SIMD.  128x128 with 8 waves (2 workgroups per CU)
    // sustains 115-119 TFLOP/s when the launch has many tiles, 64x64 with 4 waves (4 per CU) 104-110, the 4-wave 128-wide
    // tiles (2 waves per SIMD) 92-103.
    double cost = (double)cdiv(tiles, 256) * cfg_bm(id) * cfg_bn(id);
    cost *= (id == 5 ? 0.92 : (id == 3 ? 1.00 : 1.07));
    // stride-2 data gradient: a parity class runs only 1-4 of the 9 taps, i.e. 4-64 K-steps per tile; the big tiles' longer
    // prologue / epilogue then dominates (measured, 200 frames: layer3.0.conv1 dgrad 79 TFLOP/s with 64x64 vs 59 with 128x128)
    if (nclass == 4 && id != 3) cost *= 1.5;
    if (cost < best_cost) { best_cost = cost; best = id; }
  }
  return best;
}

template <class Cfg, bool SMALLC, int STATS, bool BF16 = false>
static void launch_conv_cfg(ConvGemmArgs a, int ncols, hipStream_t s) {
  conv_set_tiles(a, Cfg::BM);
  a.n_ct = cdiv(ncols, Cfg::BN);
  a.xcd_mode = (a.n_ct >= 8 && (a.n_ct & 7) == 0) ? 1 : 0;
  if (g_xcd_mode == 0) a.xcd_mode = 0;
  const dim3 grid(xcd_grid(a.n_rt, a.n_ct, a.xcd_mode));
  if constexpr (STATS == 1 && !SMALLC && !BF16) {      // the training forward of a conv that follows a BatchNorm + ReLU
    if (a.pre_stats) {
      hipLaunchKernelGGL((conv_gemm_kernel<Cfg, SMALLC, STATS, BF16, true>), grid, dim3(Cfg::THREADS), 0, s, a);
      return;
    }
  }
  hipLaunchKernelGGL((conv_gemm_kernel<Cfg, SMALLC, STATS, BF16>), grid, dim3(Cfg::THREADS), 0, s, a);
}

// hipFuncAttributeMaxDynamicSharedMemorySize of a kernel instance, once per (instance, device), from whichever host thread launches
// first (the forward's thread and the autograd engine's both launch): `mask` is the instance's static bit set of devices done.
static inline void lmkd_lds_attr_once(std::atomic<unsigned long long>& mask, const void* fn, int bytes) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) d = 0;
  const unsigned long long bit = 1ull << (d & 63);
  if (mask.load(std::memory_order_acquire) & bit) return;
  const hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (err != hipSuccess) {      // the launch that follows fails with its own message; leave the bit clear so that the next launch retries
    lmkd_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d): %s", bytes, hipGetErrorString(err));
    return;
  }
  mask.fetch_or(bit, std::memory_order_release);
}

// Grid of a PERSISTENT launch (conv_patch.h): no more workgroups than the chip holds at once (occupancy of this instance with `lds` bytes of
// dynamic LDS x CUs), each walking `total` / grid slots of the tile order.  The grid is the smallest multiple of 8 that needs the same
// number of rounds as all resident slots would - every workgroup then has (nearly) the same number of tiles - and a multiple of 8 keeps a
// workgroup's slots on one XCD (xcd_decode: slot & 7).  The occupancy is asked once per (instance, LDS size).
struct PersistOcc { std::atomic<int> n{0}; int lds[8]; int occ[8]; std::mutex mu; };
static int lmkd_cu_count() {
  static std::atomic<int> n{0};
  int v = n.load(std::memory_order_relaxed);
  if (v) return v;
  int d = 0;
  hipDeviceProp_t pr;
  if (hipGetDevice(&d) != hipSuccess || hipGetDeviceProperties(&pr, d) != hipSuccess || pr.multiProcessorCount < 1) return 256;
  n.store(pr.multiProcessorCount, std::memory_order_relaxed);
  return pr.multiProcessorCount;
}
static int g_conv_persist = 1;      // 0: one workgroup per slot of the tile order (the form before round 5; for A/B measurements)
extern "C" int lmkd_conv_set_persistent(int on) { g_conv_persist = on ? 1 : 0; return LMKD_OK; }
template <class K>
static int persistent_grid(PersistOcc& c, K kernel, int threads, size_t lds, int total) {
  if (!g_conv_persist) return total;
  int occ = 0;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    const int n = c.n.load();
    for (int i = 0; i < n; ++i) if (c.lds[i] == (int)lds) occ = c.occ[i];
    if (!occ) {
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, lds) != hipSuccess || occ < 1) occ = 1;
      if (n < 8) { c.lds[n] = (int)lds; c.occ[n] = occ; c.n.store(n + 1); }
    }
  }
  const int slots = occ * lmkd_cu_count();
  if (total <= slots) return total;
  const int rounds = cdiv(total, slots);
  const int g = (cdiv(total, rounds) + 7) & ~7;
  return g < total ? g : total;
}

// Same-size convolutions (3x3 / stride 1 forward and data gradient, 1x1 / stride 1) of the bf16-plane modes read an LDS-resident
// input patch (conv_patch.h).  Returns the halo (largest |dh * Ws + dw| over the taps), or -1 when the launch is not of that kind.
static int patch_halo(const ConvGemmArgs& a) {
  if (!g_conv_patch || !(g_conv_x3 || g_conv_bf16) || a.Cs % 32 != 0 || a.sh != 1) return -1;
  // one class: output grid = input grid; four parity classes (stride-2 data gradient): each class runs on the dy grid
  const bool plain = a.nclass == 1 && a.Hs == a.Ho && a.Ws == a.Wo;
  const bool classes = a.nclass == 4 && a.Hr == a.Hs && a.Wr == a.Ws;
  if (!plain && !classes) return -1;
  int halo = 0;
  for (int c = 0; c < a.nclass; ++c)
    for (int t = 0; t < a.ntap[c]; ++t) {
      const int sft = a.taps[c][t].dh * a.Ws + a.taps[c][t].dw;
      halo = std::max(halo, sft < 0 ? -sft : sft);
    }
  return halo <= PATCH_HALO_MAX ? halo : -1;
}
static inline bool conv_same_size(int H, int W, int KH, int KW, int stride, int pad) {
  return g_conv_patch && stride == 1 && conv_out(H, KH, stride, pad) == H && conv_out(W, KW, stride, pad) == W && (KH / 2) * W + KW / 2 <= PATCH_HALO_MAX;
}

static int g_conv_s2_patch = 1;      // stride-2 3x3 forward on conv_patch16_x3_kernel's SRC2 form; 0 = im2col-gather kernel
extern "C" int lmkd_conv_set_s2_patch(int on) { g_conv_s2_patch = on ? 1 : 0; return LMKD_OK; }
// three-plane modes, fp32 tensors, the 4-wave tiles: conv_patch16_x3_kernel (v_mfma_f32_16x16x32_*); 0 = conv_patch_x3_kernel (32x32x16).
// 1 (default) = automatic: the 16x16x32 kernel, except the TWO-PLANE launches with Cout <= 64 (layer 1; NOT the 64-column tile of a wider layer: the
// two-call and the merged form of an episode pick different tiles there and must stay bit-identical), which run
// conv_patch_x3_kernel<.., 3, ..> on v_mfma_f32_32x32x16_f16 - with half the MFMAs of the three-plane form these launches are bound by
// vector-instruction issue (5.6 vector instructions per 16x16x32 MFMA, which holds the SIMD's vector issue for 8 of its 16 cycles; a
// 32x32x16 MFMA does the work of two and holds it for 8 of 32), and a lane of the 32x32 result owns ONE channel, so the BatchNorm sums
// need one shuffle instead of a 16-lane reduction per channel: 431 -> 350 us forward + sums, 378 -> 347 us data gradient at 400 frames
// (profiles/r05_patch32_vs_16.txt; the 128-column tiles of layers 2-4 stay 5-10 % faster on 16x16x32).  2 = 16x16x32 everywhere.
static int g_patch16 = 1;
extern "C" int lmkd_conv_set_patch16(int on) { g_patch16 = on < 0 ? 0 : (on > 2 ? 2 : on); return LMKD_OK; }

#ifdef LMKD_STAMPS
extern "C" int lmkd_debug_stamps(unsigned long long* host, int zero) {
  if (zero) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_lmkd_stamps)) != hipSuccess) return LMKD_EHIP;
    return hipMemset(p, 0, sizeof(g_lmkd_stamps)) == hipSuccess ? LMKD_OK : LMKD_EHIP;
  }
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lmkd_stamps), sizeof(g_lmkd_stamps)) == hipSuccess ? LMKD_OK : LMKD_EHIP;
}
#endif
static inline int taps_within_one(const ConvGemmArgs& a) {
  for (int c = 0; c < a.nclass; ++c)
    for (int t = 0; t < a.ntap[c]; ++t)
      if (a.taps[c][t].dh < -1 || a.taps[c][t].dh > 1 || a.taps[c][t].dw < -1 || a.taps[c][t].dw > 1) return 0;
  return 1;
}
template <class Cfg>
static void launch_conv_patch(ConvGemmArgs a, int ncols, int halo, hipStream_t s) {
  conv_set_tiles(a, Cfg::BM);
  a.taps_pm1 = taps_within_one(a);
  a.n_ct = cdiv(ncols, Cfg::BN);
  a.xcd_mode = (a.n_ct >= 8 && (a.n_ct & 7) == 0) ? 1 : 0;
  if (g_xcd_mode == 0) a.xcd_mode = 0;
  const dim3 grid(xcd_grid(a.n_rt, a.n_ct, a.xcd_mode));
  a.halo = halo;
  const int npl = (g_conv_bf16 || g_lmkd_act_bf16) ? 1 : 3;
  size_t lds = patch_lds_bytes(Cfg::BM, halo, npl);
#define LMKD_PATCH(NPROD, PRE, IO)                                                                                             \
  do {                                                                                                                         \
    static std::atomic<unsigned long long> attr_done{0};                                                                       \
    static PersistOcc pocc;                                                                                                    \
    lmkd_lds_attr_once(attr_done, reinterpret_cast<const void*>(&conv_patch_x3_kernel<Cfg, NPROD, PRE, IO>),                   \
                       (int)patch_lds_bytes(Cfg::BM, PATCH_HALO_MAX, NPROD == 1 ? 1 : 3));                                     \
    a.grid_total = (int)grid.x;                                                                                                \
    const dim3 pgrid(NPROD == 3 ? persistent_grid(pocc, conv_patch_x3_kernel<Cfg, NPROD, PRE, IO>, Cfg::THREADS, lds, a.grid_total) : a.grid_total); \
    a.grid_step = (int)pgrid.x;                                                                                                \
    hipLaunchKernelGGL((conv_patch_x3_kernel<Cfg, NPROD, PRE, IO>), pgrid, dim3(Cfg::THREADS), lds, s, a);                     \
  } while (0)
  if constexpr (Cfg::THREADS == 256 && Cfg::BM == 128) {      // the benchmark's tiles (ids 11 / 12) on the 16x16x32 MFMA (conv_patch16.h)
    // (lmkd_conv_set_patch16; the inference epilogue - residual loads, no sums - stays on the 16x16x32 kernel and its 16-byte accesses: 168.7 vs 160.6 inference episodes/s)
    const bool h2_on_32 = g_patch16 == 1 && Cfg::BN == 64 && a.Co <= 64 && g_conv_h2 && a.h2_xw && !a.src2 && !a.ep_stats;
    if (g_patch16 && g_conv_x3 && !g_lmkd_act_bf16 && !h2_on_32) {
#define LMKD_PATCH16(NPROD, PRE, EP)                                                                                           \
  do {                                                                                                                         \
    t_amax_recorded = true;                                                                                                    \
    static std::atomic<unsigned long long> attr_done{0};                                                                       \
    lmkd_lds_attr_once(attr_done, reinterpret_cast<const void*>(&conv_patch16_x3_kernel<Cfg, NPROD, PRE, EP>),                 \
                       (int)patch_lds_bytes(Cfg::BM, PATCH_HALO_MAX, 3));                                                      \
    hipLaunchKernelGGL((conv_patch16_x3_kernel<Cfg, NPROD, PRE, EP>), grid, dim3(Cfg::THREADS), lds, s, a);                    \
  } while (0)
      if (a.src2) {
#define LMKD_PATCH16S(NPROD, EP)                                                                                               \
  do {                                                                                                                         \
    t_amax_recorded = true;                                                                                                    \
    static std::atomic<unsigned long long> attr_done{0};                                                                       \
    lmkd_lds_attr_once(attr_done, reinterpret_cast<const void*>(&conv_patch16_x3_kernel<Cfg, NPROD, false, EP, true>),         \
                       (int)patch_lds_bytes(Cfg::BM, PATCH_HALO_MAX, 3));                                                      \
    hipLaunchKernelGGL((conv_patch16_x3_kernel<Cfg, NPROD, false, EP, true>), grid, dim3(Cfg::THREADS), lds, s, a);            \
  } while (0)
        if (a.ep_stats) {
          if (g_conv_x3 == 9) LMKD_PATCH16S(9, true);
          else if (g_conv_h2 && a.h2_xw) { LMKD_PATCH16S(3, true); ++g_h2_launches; }      // inference in the two-plane arithmetic (round 5)
          else LMKD_PATCH16S(6, true);
        }
        else if (g_conv_h2 && a.h2_xw) { LMKD_PATCH16S(3, false); ++g_h2_launches; }
        else { if (g_conv_x3 == 9) LMKD_PATCH16S(9, false); else LMKD_PATCH16S(6, false); }
#undef LMKD_PATCH16S
        return;
      }
      if (a.ep_stats) {
        if (g_conv_x3 == 9) LMKD_PATCH16(9, false, true);
        else if (g_conv_h2 && a.h2_xw) { LMKD_PATCH16(3, false, true); ++g_h2_launches; }      // inference: BatchNorm affine (+ residual, ReLU) on the scaled accumulators; max |y| recorded in the same epilogue
        else LMKD_PATCH16(6, false, true);
      }
      else if (a.pre_stats) {
        if (g_conv_x3 == 9) LMKD_PATCH16(9, true, false);
        else if (g_conv_h2 && a.h2_xw) { LMKD_PATCH16(3, true, false); ++g_h2_launches; }      // h2_xw: a bound of relu(BatchNorm(x)) (lmkd_bn_finalize_bound)
        else LMKD_PATCH16(6, true, false);
      }
      else if (a.bnb_x) {      // data gradient with the BatchNorm-backward sums in its epilogue: instances of their own (conv_patch16.h, BNB)
#define LMKD_PATCH16B(NPROD)                                                                                                   \
  do {                                                                                                                         \
    t_amax_recorded = true;                                                                                                    \
    static std::atomic<unsigned long long> attr_done{0};                                                                       \
    lmkd_lds_attr_once(attr_done, reinterpret_cast<const void*>(&conv_patch16_x3_kernel<Cfg, NPROD, false, false, false, true>), \
                       (int)patch_lds_bytes(Cfg::BM, PATCH_HALO_MAX, 3));                                                      \
    hipLaunchKernelGGL((conv_patch16_x3_kernel<Cfg, NPROD, false, false, false, true>), grid, dim3(Cfg::THREADS), lds, s, a);  \
  } while (0)
        if (g_conv_h2 && a.h2_xw) { LMKD_PATCH16B(3); ++g_h2_launches; }
        else if (g_conv_x3 == 9) LMKD_PATCH16B(9);
        else LMKD_PATCH16B(6);
#undef LMKD_PATCH16B
      }
      else if (g_conv_h2 && a.h2_xw) { LMKD_PATCH16(3, false, false); ++g_h2_launches; }
      else { if (g_conv_x3 == 9) LMKD_PATCH16(9, false, false); else LMKD_PATCH16(6, false, false); }
#undef LMKD_PATCH16
      return;
    }
  }
  if (g_conv_h2 && a.h2_xw && !a.ep_stats && !g_lmkd_act_bf16) {      // two fp16 planes on v_mfma_f32_32x32x16_f16: the 64-column tile (lmkd_conv_set_patch16), or a tile conv_patch16_x3_kernel does not have
    t_amax_recorded = true;      // (x3_epilogue<.., H2> folds ConvGemmArgs::amax_out)
    lds = patch_lds_bytes(Cfg::BM, halo, 2);
    if (a.pre_stats) LMKD_PATCH(3, true, 0); else LMKD_PATCH(3, false, 0);
    ++g_h2_launches;
    return;
  }
  if (a.ep_stats) {      // inference: BatchNorm affine (+ residual, ReLU) in the epilogue, fp32 tensors
    if (g_conv_bf16) LMKD_PATCH(1, false, 4);
    else if (g_conv_x3 == 9) LMKD_PATCH(9, false, 4);
    else LMKD_PATCH(6, false, 4);
  } else if (g_lmkd_act_bf16) LMKD_PATCH(1, false, 3);
  else if (a.pre_stats) {
    if (g_conv_bf16) LMKD_PATCH(1, true, 0);
    else if (g_conv_x3 == 9) LMKD_PATCH(9, true, 0);
    else LMKD_PATCH(6, true, 0);
  } else {
    if (g_conv_bf16) LMKD_PATCH(1, false, 0);
    else if (g_conv_x3 == 9) LMKD_PATCH(9, false, 0);
    else LMKD_PATCH(6, false, 0);
  }
#undef LMKD_PATCH
}

// (round 4: conv_patch16_x3_kernel on 256 x 64 tiles - four waves stacked over the rows, wave tiles 64 x 64, two workgroups per CU - was
// built as tile id 15 and measured SLOWER than id 11 on every layer, 443 vs 383 us on layer 1 in fp32h2: profiles/r04_tile_ab.txt; removed)
// one-plane modes only (tile ids 13 / 14: 256-row tiles with four waves)
template <class Cfg>
static void launch_conv_patch_1p(ConvGemmArgs a, int ncols, int halo, hipStream_t s) {
  conv_set_tiles(a, Cfg::BM);
  a.taps_pm1 = taps_within_one(a);
  a.n_ct = cdiv(ncols, Cfg::BN);
  a.xcd_mode = (a.n_ct >= 8 && (a.n_ct & 7) == 0) ? 1 : 0;
  if (g_xcd_mode == 0) a.xcd_mode = 0;
  const dim3 grid(xcd_grid(a.n_rt, a.n_ct, a.xcd_mode));
  a.halo = halo;
  const size_t lds = patch_lds_bytes(Cfg::BM, halo, 1);
#define LMKD_PATCH1(PRE, IO)                                                                                                   \
  do {                                                                                                                         \
    static std::atomic<unsigned long long> attr_done{0};                                                                       \
    lmkd_lds_attr_once(attr_done, reinterpret_cast<const void*>(&conv_patch_x3_kernel<Cfg, 1, PRE, IO>),                       \
                       (int)patch_lds_bytes(Cfg::BM, PATCH_HALO_MAX, 1));                                                      \
    a.grid_total = (int)grid.x;                                                                                                \
    const dim3 pgrid(a.grid_total);      /* (one-plane instances: one workgroup per tile, conv_patch.h PERSIST) */            \
    a.grid_step = (int)pgrid.x;                                                                                                \
    hipLaunchKernelGGL((conv_patch_x3_kernel<Cfg, 1, PRE, IO>), pgrid, dim3(Cfg::THREADS), lds, s, a);                         \
  } while (0)
  if (a.ep_stats) LMKD_PATCH1(false, 4);
  else if (g_lmkd_act_bf16) LMKD_PATCH1(false, 3);
  else if (a.pre_stats) LMKD_PATCH1(true, 0);
  else LMKD_PATCH1(false, 0);
#undef LMKD_PATCH1
}

template <class Cfg, bool SMALLC, bool STATS>
static void launch_conv_x3(ConvGemmArgs a, int ncols, hipStream_t s) {
  conv_set_tiles(a, Cfg::BM);
  a.n_ct = cdiv(ncols, Cfg::BN);
  a.xcd_mode = (a.n_ct >= 8 && (a.n_ct & 7) == 0) ? 1 : 0;
  if (g_xcd_mode == 0) a.xcd_mode = 0;
  const dim3 grid(xcd_grid(a.n_rt, a.n_ct, a.xcd_mode));
  // ONE instance serves launches with and without BatchNorm partials (stat_partial null -> nothing written): hipcc schedules the
  // main loop of the instance without the statistics epilogue worse (register copies in front of the MFMAs: 105 vs 148 TFLOP/s
  // on the same layer), so that instance is not built
  (void)STATS;
  if constexpr (!SMALLC) {
    if (a.ep_stats) {      // inference epilogue (x3_epilogue EP): stride-2 / 1x1-downsample convolutions of an eval forward
      if (g_conv_bf16) hipLaunchKernelGGL((conv_gemm_x3_kernel<Cfg, false, true, 1, false, 4>), grid, dim3(Cfg::THREADS), 0, s, a);
      else if (g_conv_x3 == 9) hipLaunchKernelGGL((conv_gemm_x3_kernel<Cfg, false, true, 9, false, 4>), grid, dim3(Cfg::THREADS), 0, s, a);
      else hipLaunchKernelGGL((conv_gemm_x3_kernel<Cfg, false, true, 6, false, 4>), grid, dim3(Cfg::THREADS), 0, s, a);
      return;
    }
  }
  if (g_lmkd_act_bf16) {      // bf16 tensors in HBM (one-plane mode): the loader copies, the epilogue rounds; the stem input stays fp32
    if constexpr (SMALLC) hipLaunchKernelGGL((conv_gemm_x3_kernel<Cfg, true, true, 1, false, 2>), grid, dim3(Cfg::THREADS), 0, s, a);
    else hipLaunchKernelGGL((conv_gemm_x3_kernel<Cfg, false, true, 1, false, 3>), grid, dim3(Cfg::THREADS), 0, s, a);
    return;
  }
  if constexpr (!SMALLC) {
    if (a.pre_stats) {      // training forward fed by relu(BatchNorm(raw)): normalise + rectify when the tile is stored to LDS
      if (g_conv_bf16) hipLaunchKernelGGL((conv_gemm_x3_kernel<Cfg, false, true, 1, true>), grid, dim3(Cfg::THREADS), 0, s, a);
      else if (g_conv_x3 == 9) hipLaunchKernelGGL((conv_gemm_x3_kernel<Cfg, false, true, 9, true>), grid, dim3(Cfg::THREADS), 0, s, a);
      else hipLaunchKernelGGL((conv_gemm_x3_kernel<Cfg, false, true, 6, true>), grid, dim3(Cfg::THREADS), 0, s, a);
      return;
    }
  }
  if (g_conv_bf16) hipLaunchKernelGGL((conv_gemm_x3_kernel<Cfg, SMALLC, true, 1>), grid, dim3(Cfg::THREADS), 0, s, a);
  else if (g_conv_x3 == 9) hipLaunchKernelGGL((conv_gemm_x3_kernel<Cfg, SMALLC, true, 9>), grid, dim3(Cfg::THREADS), 0, s, a);
  else hipLaunchKernelGGL((conv_gemm_x3_kernel<Cfg, SMALLC, true, 6>), grid, dim3(Cfg::THREADS), 0, s, a);
}

template <bool SMALLC, int STATS>
static int launch_conv_gemm(const ConvGemmArgs& a, int ncols, hipStream_t s) {
  if constexpr (STATS != 2 || !SMALLC) {      // STATS == 2 (inference epilogue) exists on the bf16-plane kernels for Cs % 32 == 0
    if (g_conv_x3 || g_conv_bf16) {      // bf16 planes in LDS, weights in fragment order (conv_x3.h): 1, 6 or 9 products
      const int halo = SMALLC ? -1 : patch_halo(a);
      int id = pick_conv_cfg(a.rows_per_class, a.nclass, ncols, a.same != 0);
      if (halo < 0 && id == 10) id = 7;
      if (halo < 0 && id == 11) id = 9;
      if (halo < 0 && id == 12) id = 8;      // same tile height (the BatchNorm partials' row count) on the gather kernel
      if (halo < 0 && (id == 13 || id == 14)) id = 7;
      if (halo >= 0) {
        switch (id) {
          case 7: launch_conv_patch<X3Cfg<256, 64, 4, 2>>(a, ncols, halo, s); break;
          case 8: launch_conv_patch<X3Cfg<128, 128, 2, 4>>(a, ncols, halo, s); break;
          case 10: launch_conv_patch<X3Cfg<256, 128, 2, 4>>(a, ncols, halo, s); break;
          case 11: launch_conv_patch<X3Cfg<128, 64, 2, 2, 3>>(a, ncols, halo, s); break;
          case 12: launch_conv_patch<X3Cfg<128, 128, 2, 2, 2>>(a, ncols, halo, s); break;
          case 13: launch_conv_patch_1p<X3Cfg<256, 128, 2, 2, 2>>(a, ncols, halo, s); break;
          case 14: launch_conv_patch_1p<X3Cfg<256, 64, 2, 2, 2>>(a, ncols, halo, s); break;
          default: launch_conv_patch<X3Cfg<128, 64, 4, 2>>(a, ncols, halo, s); break;
        }
        LMKD_CHECK_LAUNCH("conv_patch_x3_kernel");
        return LMKD_OK;
      }
      switch (id) {
        case 7: launch_conv_x3<X3Cfg<256, 64, 4, 2>, SMALLC, STATS == 1>(a, ncols, s); break;
        case 8: launch_conv_x3<X3Cfg<128, 128, 2, 4>, SMALLC, STATS == 1>(a, ncols, s); break;
        default: launch_conv_x3<X3Cfg<128, 64, 4, 2>, SMALLC, STATS == 1>(a, ncols, s); break;
      }
      LMKD_CHECK_LAUNCH("conv_gemm_x3_kernel");
      return LMKD_OK;
    }
  }
  switch (pick_conv_cfg(a.rows_per_class, a.nclass, ncols)) {
    case 1: launch_conv_cfg<TileCfg<128, 128, 2, 2>, SMALLC, STATS>(a, ncols, s); break;
    case 2: launch_conv_cfg<TileCfg<128, 64, 2, 2>, SMALLC, STATS>(a, ncols, s); break;
    case 3: launch_conv_cfg<TileCfg<64, 64, 2, 2>, SMALLC, STATS>(a, ncols, s); break;
    case 5: launch_conv_cfg<TileCfg<128, 128, 2, 4>, SMALLC, STATS>(a, ncols, s); break;
    case 6: launch_conv_cfg<TileCfg<128, 64, 4, 2>, SMALLC, STATS>(a, ncols, s); break;
    default: launch_conv_cfg<TileCfg<64, 128, 2, 2>, SMALLC, STATS>(a, ncols, s); break;
  }
  LMKD_CHECK_LAUNCH("conv_gemm_kernel");
  return LMKD_OK;
}

// The stem (7x7 / stride 2 / pad 3 on the NHWC4 input, Cout <= 64) of the bf16-plane modes runs conv_stem_patch_kernel (conv_stem.h):
// one workgroup per pair of output rows.  Returns the patch row pitch in pixels, or 0 when the launch is not of that kind.
// stride-2 3x3 forward convolution as four same-size convolutions over the input's parity classes (conv_patch16.h, SRC2)?
static bool fwd_s2_patch_ok(int N, int H, int W, int Cs, int Cout, int KH, int KW, int stride, int pad) {
  if (!(g_conv_s2_patch && g_patch16 && g_conv_patch && g_conv_x3 && !g_lmkd_act_bf16) || Cs % 32 != 0 || stride != 2 || KH != 3 ||
      KW != 3 || pad != 1 || H % 2 != 0 || W % 2 != 0 || W / 2 + 1 > PATCH_HALO_MAX)
    return false;
  const int id = pick_conv_cfg((long)N * (H / 2) * (W / 2), 1, Cout, true);
  return id == 11 || id == 12;
}

static int g_conv_stem_patch = 1;
extern "C" int lmkd_conv_set_stem_patch(int on) { g_conv_stem_patch = on ? 1 : 0; return LMKD_OK; }
static int stem_patch_pitch(int W, int Cs, int Cout, int KH, int KW, int stride, int pad) {
  if (!g_conv_stem_patch || !(g_conv_x3 || g_conv_bf16) || Cs != 4 || KH != 7 || KW != 7 || stride != 2 || pad != 3 || Cout > 64) return 0;
  const int Wo = conv_out(W, KW, stride, pad);
  int wp = std::max(2 * Wo + 6, W + 3);
  wp += wp & 1;
  return (2 * Wo <= 256 && wp <= STEM_MAX_WP) ? wp : 0;
}

// number of row tiles (= rows of the BN partial-statistics buffer) of a forward conv
extern "C" int lmkd_conv2d_fwd_row_tiles_seg(int N, int H, int W, int Cs, int Cout, int KH, int KW, int stride, int pad, int seg_n0, int* tiles0);
extern "C" int lmkd_conv2d_fwd_row_tiles_cs(int N, int H, int W, int Cs, int Cout, int KH, int KW, int stride, int pad) {
  return lmkd_conv2d_fwd_row_tiles_seg(N, H, W, Cs, Cout, KH, KW, stride, pad, 0, nullptr);
}
// two frame segments in one launch (ConvGemmArgs::seg_m0): frames [0, seg_n0) | [seg_n0, N).  Returns the rows of the partial-sum
// buffer; *tiles0 (host pointer, nullable) receives how many of them - the leading ones - belong to segment 0.  seg_n0 = 0 or N: one segment.
extern "C" int lmkd_conv2d_fwd_row_tiles_seg(int N, int H, int W, int Cs, int Cout, int KH, int KW, int stride, int pad, int seg_n0, int* tiles0) {
  if (seg_n0 <= 0 || seg_n0 >= N) seg_n0 = N;
  const int Ho = conv_out(H, KH, stride, pad), Wo = conv_out(W, KW, stride, pad);
  if (stem_patch_pitch(W, Cs, Cout, KH, KW, stride, pad)) {
    if (tiles0) *tiles0 = seg_n0 * cdiv(Ho, 2);
    return N * cdiv(Ho, 2);
  }
  const long M = (long)N * Ho * Wo, M0 = (long)seg_n0 * Ho * Wo;
  const int bm = cfg_bm(pick_conv_cfg(M, 1, Cout, conv_same_size(H, W, KH, KW, stride, pad)));
  if (tiles0) *tiles0 = cdiv(M0, bm);
  return cdiv(M0, bm) + cdiv(M - M0, bm);
}
extern "C" int lmkd_conv2d_fwd_row_tiles(int N, int H, int W, int Cout, int KH, int KW, int stride, int pad) {
  return lmkd_conv2d_fwd_row_tiles_cs(N, H, W, 32, Cout, KH, KW, stride, pad);
}

static int conv2d_fwd_impl(const float* x, const float* wp, float* y, float* stat_partial, const float* ep_stats,
                           const float* ep_res, int ep_relu, int N, int H, int W, int Cs, int Cout, int KH, int KW, int stride,
                           int pad, void* stream, const float* pre_stats = nullptr, int seg_n0 = 0, const lmkd_amax_desc* am = nullptr) {
  LMKD_REQUIRE(x && wp && y, "lmkd_conv2d_fwd: null pointer");
  if (seg_n0 <= 0 || seg_n0 >= N) seg_n0 = 0;
  LMKD_REQUIRE(!seg_n0 || ((g_conv_x3 || g_conv_bf16) && !ep_stats), "lmkd_conv2d_fwd_seg: two frame segments exist in the bf16-plane modes (training / plain forward)");
  LMKD_REQUIRE(aligned16(x) && aligned16(wp), "lmkd_conv2d_fwd: x / packed weights must be 16-byte aligned");
  LMKD_REQUIRE(Cs % 32 == 0 || Cs == 4, "lmkd_conv2d_fwd: channel count %d must be 4 (padded stem) or a multiple of 32", Cs);
  LMKD_REQUIRE(N > 0 && H > 0 && W > 0 && Cout > 0, "lmkd_conv2d_fwd: empty tensor");
  const bool smallc = Cs == 4;
  const int KWp = kw_padded(Cs, KW);
  LMKD_REQUIRE(!smallc || KWp == 8, "lmkd_conv2d_fwd: padded-stem path needs KW <= 8");
  LMKD_REQUIRE(KH * (smallc ? 1 : KW) <= LMKD_MAX_TAPS, "lmkd_conv2d_fwd: kernel %dx%d has too many taps", KH, KW);
  LMKD_REQUIRE(H < 32768 && W < 32768, "lmkd_conv2d_fwd: spatial dims too large");
  ConvGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.src = x; a.wpk = wp; a.out = y; a.stat_partial = stat_partial;
  a.h2_xw = am ? (const unsigned*)am->x_words : nullptr;
  a.amax_out = am ? (unsigned*)am->out_words : nullptr;
  note_fenced(am);
  a.ep_stats = ep_stats; a.ep_res = ep_res; a.ep_relu = ep_relu;
  a.pre_stats = pre_stats;
  LMKD_REQUIRE(!pre_stats || (stat_partial && !smallc && !g_lmkd_act_bf16),
               "lmkd_conv2d_fwd_pre: the fused BatchNorm+ReLU loader exists for the training forward with fp32 activations (stat_partial given, Cs %% 32 == 0)");
  LMKD_REQUIRE(!g_lmkd_act_bf16 || (g_conv_bf16 && !ep_stats), "bf16 activations need lmkd_conv_set_compute_dtype(1) (training / plain forward)");
  a.N = N; a.Hs = H; a.Ws = W; a.Cs = Cs;
  a.Ho = conv_out(H, KH, stride, pad); a.Wo = conv_out(W, KW, stride, pad); a.Co = Cout;
  LMKD_REQUIRE(a.Ho > 0 && a.Wo > 0, "lmkd_conv2d_fwd: empty output");
  LMKD_REQUIRE((long)N * H * W * Cs < 2147483647L && (long)N * a.Ho * a.Wo * Cout < 2147483647L,
               "lmkd_conv2d_fwd: tensor exceeds 2^31 elements");
  // the bf16-plane kernels (patch / stem / gather) address the gathered tensor with 32-bit BYTE offsets through a buffer resource
  LMKD_REQUIRE(!(g_conv_x3 || g_conv_bf16) || (long)N * H * W * Cs * (g_lmkd_act_bf16 && Cs != 4 ? 2 : 4) < 0xffffffe0L,
               "lmkd_conv2d_fwd: input tensor exceeds the 4 GiB buffer range of the bf16-plane kernels");
  a.Hr = a.Ho; a.Wr = a.Wo; a.rows_per_class = N * a.Ho * a.Wo; a.tiles_per_class = cdiv(a.rows_per_class, 128);
  a.seg_m0 = seg_n0 * a.Ho * a.Wo;      // 0: one segment (conv_set_tiles)
  a.sh = stride; a.omul = 1; a.nclass = 1;
  a.same = conv_same_size(H, W, KH, KW, stride, pad) ? 1 : 0;      // as in lmkd_conv2d_fwd_row_tiles
  a.Kp = KH * KWp * Cs;
  a.div_hw = make_fastdiv(a.Hr * a.Wr); a.div_w = make_fastdiv(a.Wr);
  if (smallc) {
    a.cps = 1;
    a.ntap[0] = KH;
    for (int kh = 0; kh < KH; ++kh) a.taps[0][kh] = Tap{kh - pad, -pad, kh * KWp * Cs};
  } else {
    a.cps = Cs / 32;
    a.ntap[0] = KH * KW;
    for (int kh = 0; kh < KH; ++kh)
      for (int kw = 0; kw < KW; ++kw) a.taps[0][kh * KW + kw] = Tap{kh - pad, kw - pad, (kh * KWp + kw) * Cs};
  }
  hipStream_t s = (hipStream_t)stream;
  // stride-2 3x3 forward on the patch kernel (SRC2): even input, the 16x16x32 instances (tile ids 11 / 12), no BatchNorm loader
  if (!pre_stats && fwd_s2_patch_ok(N, H, W, Cs, Cout, KH, KW, stride, pad)) {
    {
      a.src2 = 1; a.s2_ncls = 4; a.s2_wfull = W;
      a.Hs = a.Ho; a.Ws = a.Wo; a.sh = 1; a.same = 1;
      LMKD_REQUIRE((long)N * H * W * Cs * 4 < 2147483647L, "lmkd_conv2d_fwd: input tensor exceeds the 2 GiB offset range of the stride-2 patch form");
      static const int order[4][2] = {{1, 1}, {1, 0}, {0, 1}, {0, 0}};      // classes by tap count 4, 2, 2, 1
      int nt = 0;
      for (int c = 0; c < 4; ++c) {
        const int sph = order[c][0], spw = order[c][1];
        a.s2_t0[c] = nt;
        a.s2_off[c] = (sph * W + spw) * Cs;
        for (int kh = 0; kh < 3; ++kh) {
          if (((kh - 1) & 1) != sph) continue;
          for (int kw = 0; kw < 3; ++kw) {
            if (((kw - 1) & 1) != spw) continue;
            a.taps[0][nt++] = Tap{(kh - 1 - sph) / 2, (kw - 1 - spw) / 2, (kh * KWp + kw) * Cs};
          }
        }
      }
      a.s2_t0[4] = nt;
      a.ntap[0] = nt;
    }
  }
  if (const int wp = ep_stats ? 0 : stem_patch_pitch(W, Cs, Cout, KH, KW, stride, pad)) {
    a.halo = wp;      // row pitch of the input-row patch
    a.n_rt = N * cdiv(a.Ho, 2); a.n_ct = 1;
    const int npl = (g_conv_bf16 || g_lmkd_act_bf16) ? 1 : 3;
    const size_t lds = (size_t)npl * 9 * wp * 8;
    const dim3 grid(a.n_rt), block(512);
#define LMKD_STEM(OR)                                                                                                  \
  do {                                                                                                                 \
    if (g_lmkd_act_bf16) hipLaunchKernelGGL((conv_stem_patch_kernel<1, true, OR>), grid, block, lds, s, a);            \
    else if (g_conv_bf16) hipLaunchKernelGGL((conv_stem_patch_kernel<1, false, OR>), grid, block, lds, s, a);          \
    else if (g_conv_x3 == 9) hipLaunchKernelGGL((conv_stem_patch_kernel<9, false, OR>), grid, block, lds, s, a);       \
    else if (g_conv_h2 && a.h2_xw) {                                                                                   \
      hipLaunchKernelGGL((conv_stem_patch_kernel<3, false, OR>), grid, block, lds, s, a);                              \
      ++g_h2_launches;                                                                                                 \
    } else hipLaunchKernelGGL((conv_stem_patch_kernel<6, false, OR>), grid, block, lds, s, a);                         \
  } while (0)
    LMKD_STEM(2);      // two output rows per workgroup (one row, four waves: 557 vs 447 us at 200 frames)
#undef LMKD_STEM
    LMKD_CHECK_LAUNCH("conv_stem_patch_kernel");
    return LMKD_OK;
  }
  t_amax_recorded = false;
  int rc;
  if (ep_stats) rc = smallc ? launch_conv_gemm<true, 2>(a, Cout, s) : launch_conv_gemm<false, 2>(a, Cout, s);
  else if (smallc) rc = stat_partial ? launch_conv_gemm<true, 1>(a, Cout, s) : launch_conv_gemm<true, 0>(a, Cout, s);
  else rc = stat_partial ? launch_conv_gemm<false, 1>(a, Cout, s) : launch_conv_gemm<false, 0>(a, Cout, s);
  if (rc == LMKD_OK && a.amax_out && !t_amax_recorded && !g_lmkd_act_bf16) {
    // lmkd_conv_output_amax on a launch whose kernel does not fold the maximum in its epilogue: a reduction pass of its own per frame
    // segment (rare shapes; the trunk's 3x3 convolutions all run conv_patch16_x3_kernel) - the words are never left at "unknown"
    const long per = (long)a.Ho * a.Wo * Cout, n0 = seg_n0 ? seg_n0 * per : (long)N * per;
    rc = amax_impl(y, n0, a.amax_out, 1, stream, false);
    if (rc == LMKD_OK && seg_n0) rc = amax_impl(y + n0, (long)N * per - n0, a.amax_out + LMKD_AMAX_SEG_WORDS, 1, stream, false);
  }
  return rc;
}

extern "C" int lmkd_conv2d_fwd(const float* x, const float* wp, float* y, float* stat_partial, int N, int H, int W, int Cs,
                               int Cout, int KH, int KW, int stride, int pad, void* stream) {
  return conv2d_fwd_impl(x, wp, y, stat_partial, nullptr, nullptr, 0, N, H, W, Cs, Cout, KH, KW, stride, pad, stream);
}

// Training form for a convolution that consumes relu(BatchNorm(x_raw)): x_raw is the previous convolution's raw output and
// pre_stats its [5][Cs] BatchNorm table (lmkd_bn_finalize); normalise + ReLU happen in the loader, bit-identical to running
// lmkd_bn_apply first (torchvision BasicBlock: conv2(relu(bn1(conv1(x)))), resnet18_2fc.py:41-42).
// lmkd_conv2d_fwd / lmkd_conv2d_fwd_pre (pre_stats nullable) over TWO frame segments [0, seg_n0) | [seg_n0, N) of one tensor - the support-
// frame and the query-frame trunk call of an episode (resnet18_2fc.py:41-42) in one launch: pre_stats is a [2][5][Cs] table (one per
// segment), stat_partial has lmkd_conv2d_fwd_row_tiles_seg rows of which the first *tiles0 belong to segment 0.  Results are bit-identical
// to two launches on the two halves.  bf16-plane modes (lmkd_conv_set_compute_dtype 1-3).
extern "C" int lmkd_conv2d_fwd_seg(const float* x, const float* pre_stats, const float* wp, float* y, float* stat_partial, int N, int H, int W,
                                   int Cs, int Cout, int KH, int KW, int stride, int pad, int seg_n0, void* stream, const lmkd_amax_desc* amax) {
  return conv2d_fwd_impl(x, wp, y, stat_partial, nullptr, nullptr, 0, N, H, W, Cs, Cout, KH, KW, stride, pad, stream, pre_stats, seg_n0, amax);
}

extern "C" int lmkd_conv2d_fwd_pre(const float* x_raw, const float* pre_stats, const float* wp, float* y, float* stat_partial, int N,
                                   int H, int W, int Cs, int Cout, int KH, int KW, int stride, int pad, void* stream) {
  LMKD_REQUIRE(pre_stats, "lmkd_conv2d_fwd_pre: BatchNorm table of the input missing");
  return conv2d_fwd_impl(x_raw, wp, y, stat_partial, nullptr, nullptr, 0, N, H, W, Cs, Cout, KH, KW, stride, pad, stream, pre_stats);
}

// Inference form: y = relu?( conv(x) * scale[c] + shift[c] (+ res) ) in the convolution's epilogue (scale/shift = rows 2/3 of
// the [5][C] table of lmkd_bn_eval_stats); the same operations in the same order as lmkd_conv2d_fwd + lmkd_bn_apply.
extern "C" int lmkd_conv2d_fwd_bn(const float* x, const float* wp, float* y, const float* bn_stats, const float* res, int relu,
                                  int N, int H, int W, int Cs, int Cout, int KH, int KW, int stride, int pad, void* stream, const lmkd_amax_desc* amax) {
  LMKD_REQUIRE(bn_stats, "lmkd_conv2d_fwd_bn: BatchNorm table missing");
  LMKD_REQUIRE(!g_lmkd_act_bf16, "lmkd_conv2d_fwd_bn: fp32 tensors only (with bf16 tensors run lmkd_conv2d_fwd + lmkd_bn_apply)");
  LMKD_REQUIRE(!(g_conv_x3 || g_conv_bf16) || Cs % 32 == 0, "lmkd_conv2d_fwd_bn: the bf16-plane modes need Cs %% 32 == 0 (Cs=%d)", Cs);
  return conv2d_fwd_impl(x, wp, y, nullptr, bn_stats, res, relu, N, H, W, Cs, Cout, KH, KW, stride, pad, stream, nullptr, 0, amax);
}

// dx[N,H,W,Cin] (+= when accumulate) from dy[N,Ho,Wo,Cout]; wd = weights packed with mode 1
static int bwd_data_args(ConvGemmArgs& a, const float* dy, const float* wd, float* dx, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                         int stride, int pad, int accumulate, int seg_n0 = 0, const lmkd_amax_desc* am = nullptr) {
  if (seg_n0 <= 0 || seg_n0 >= N) seg_n0 = 0;
  LMKD_REQUIRE(!seg_n0 || g_conv_x3 || g_conv_bf16, "lmkd_conv2d_bwd_data_seg: two frame segments exist in the bf16-plane modes");
  LMKD_REQUIRE(Cout % 32 == 0, "lmkd_conv2d_bwd_data: Cout=%d must be a multiple of 32", Cout);
  LMKD_REQUIRE(stride == 1 || stride == 2, "lmkd_conv2d_bwd_data: stride %d unsupported", stride);
  LMKD_REQUIRE(KH * KW <= LMKD_MAX_TAPS, "lmkd_conv2d_bwd_data: kernel too large");
  LMKD_REQUIRE(!g_lmkd_act_bf16 || g_conv_bf16, "bf16 activations need lmkd_conv_set_compute_dtype(1)");
  const int Ho = conv_out(H, KH, stride, pad), Wo = conv_out(W, KW, stride, pad);
  LMKD_REQUIRE((long)N * H * W * Cin < 2147483647L && (long)N * Ho * Wo * Cout < 2147483647L,
               "lmkd_conv2d_bwd_data: tensor exceeds 2^31 elements");
  LMKD_REQUIRE(!(g_conv_x3 || g_conv_bf16) || (long)N * Ho * Wo * Cout * (g_lmkd_act_bf16 ? 2 : 4) < 0xffffffe0L,
               "lmkd_conv2d_bwd_data: dy exceeds the 4 GiB buffer range of the bf16-plane kernels");
  memset(&a, 0, sizeof(a));
  a.src = dy; a.wpk = wd; a.out = dx; a.stat_partial = nullptr; a.accum = accumulate;
  a.h2_xw = am ? (const unsigned*)am->dy_words : nullptr;
  a.N = N; a.Hs = Ho; a.Ws = Wo; a.Cs = Cout;
  a.Ho = H; a.Wo = W; a.Co = Cin;
  a.Kp = KH * KW * Cout; a.cps = Cout / 32;
  a.sh = 1;
  if (stride == 1) {
    a.nclass = 1; a.omul = 1; a.Hr = H; a.Wr = W;
    int nt = 0;
    for (int kh = 0; kh < KH; ++kh)
      for (int kw = 0; kw < KW; ++kw) a.taps[0][nt++] = Tap{pad - kh, pad - kw, (kh * KW + kw) * Cout};
    a.ntap[0] = nt;
  } else {
    // rows enumerated per input parity (ph,pw): ih = 2a+ph, source oh = (ih + pad - kh)/2 = a + (ph+pad-kh)/2
    a.nclass = 4; a.omul = 2; a.Hr = (H + 1) / 2; a.Wr = (W + 1) / 2;
    for (int c = 0; c < 4; ++c) {
      const int ph = c >> 1, pw = c & 1;
      int nt = 0;
      for (int kh = 0; kh < KH; ++kh) {
        if (((ph + pad - kh) & 1) != 0) continue;
        for (int kw = 0; kw < KW; ++kw) {
          if (((pw + pad - kw) & 1) != 0) continue;
          // floor division (numerator may be negative)
          const int nh = ph + pad - kh, nw = pw + pad - kw;
          a.taps[c][nt++] = Tap{nh >= 0 ? nh / 2 : -((-nh) / 2), nw >= 0 ? nw / 2 : -((-nw) / 2), (kh * KW + kw) * Cout};
        }
      }
      a.ntap[c] = nt;
    }
  }
  a.rows_per_class = N * a.Hr * a.Wr;
  a.seg_m0 = seg_n0 * a.Hr * a.Wr;
  a.tiles_per_class = cdiv(a.rows_per_class, 128);
  a.div_hw = make_fastdiv(a.Hr * a.Wr); a.div_w = make_fastdiv(a.Wr);
  a.same = patch_halo(a) >= 0 ? 1 : 0;
  return LMKD_OK;
}

extern "C" int lmkd_conv2d_bwd_data(const float* dy, const float* wd, float* dx, int N, int H, int W, int Cin, int Cout,
                                    int KH, int KW, int stride, int pad, int accumulate, void* stream) {
  LMKD_REQUIRE(dy && wd && dx, "lmkd_conv2d_bwd_data: null pointer");
  LMKD_REQUIRE(aligned16(dy) && aligned16(wd), "lmkd_conv2d_bwd_data: operands must be 16-byte aligned");
  ConvGemmArgs a;
  if (const int rc = bwd_data_args(a, dy, wd, dx, N, H, W, Cin, Cout, KH, KW, stride, pad, accumulate)) return rc;
  return launch_conv_gemm<false, 0>(a, Cin, (hipStream_t)stream);
}

// The data gradient that feeds relu + train-mode BatchNorm backward (BasicBlock: d a1 = dgrad(conv2), a1 = relu(bn1(c1));
// resnet18_2fc.py:41-42 runs torchvision's block): the epilogue of the patch kernel also reads the BatchNorm's input at the pixels it
// has just produced and leaves the per-row-tile sums (sum g, sum g * xhat) in `part`, so lmkd_bn_backward_part needs no reduction pass
// over (dx, bn_x).  Exists for the launches conv_patch16_x3_kernel serves (stride 1, three-plane arithmetic, fp32 tensors, 128-row
// tiles); lmkd_conv2d_bwd_data_bn_tiles returns the row count of `part` ([tiles][Cin][2] floats) or 0 when the launch has no such form.
static bool bwd_data_bn_ok(const ConvGemmArgs& a, int Cin) {
  if (!(g_patch16 && g_conv_x3 && !g_lmkd_act_bf16) || a.nclass != 1 || Cin % 4 != 0) return false;
  if (patch_halo(a) < 0) return false;
  const int id = pick_conv_cfg(a.rows_per_class, a.nclass, Cin, a.same != 0);
  return id == 11 || id == 12;
}
extern "C" int lmkd_conv2d_bwd_data_bn_tiles_seg(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int seg_n0, int* tiles0) {
  if (stride != 1 || Cout % 32 != 0 || KH * KW > LMKD_MAX_TAPS) return 0;
  ConvGemmArgs a;
  if (bwd_data_args(a, nullptr, nullptr, nullptr, N, H, W, Cin, Cout, KH, KW, stride, pad, 0, seg_n0)) return 0;
  if (!bwd_data_bn_ok(a, Cin)) return 0;
  conv_set_tiles(a, 128);
  if (tiles0) *tiles0 = a.seg_t0;
  return a.tiles_per_class;
}
extern "C" int lmkd_conv2d_bwd_data_bn_tiles(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  return lmkd_conv2d_bwd_data_bn_tiles_seg(N, H, W, Cin, Cout, KH, KW, stride, pad, 0, nullptr);
}
// lmkd_conv2d_bwd_data / lmkd_conv2d_bwd_data_bn (bn_x / bn_stats / part nullable together) over two frame segments (lmkd_conv2d_fwd_seg):
// bn_stats is a [2][5][Cin] table, part has lmkd_conv2d_bwd_data_bn_tiles_seg rows.  Bit-identical to two launches on the halves.
extern "C" int lmkd_conv2d_bwd_data_seg(const float* dy, const float* wd, float* dx, const float* bn_x, const float* bn_stats, float* part, int N,
                                        int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int accumulate, int seg_n0, void* stream,
                                        const lmkd_amax_desc* amax) {
  LMKD_REQUIRE(dy && wd && dx, "lmkd_conv2d_bwd_data_seg: null pointer");
  LMKD_REQUIRE(aligned16(dy) && aligned16(wd), "lmkd_conv2d_bwd_data_seg: operands must be 16-byte aligned");
  LMKD_REQUIRE((bn_x != nullptr) == (bn_stats != nullptr) && (bn_x != nullptr) == (part != nullptr), "lmkd_conv2d_bwd_data_seg: bn_x, bn_stats and part go together");
  ConvGemmArgs a;
  if (const int rc = bwd_data_args(a, dy, wd, dx, N, H, W, Cin, Cout, KH, KW, stride, pad, accumulate, seg_n0, amax)) return rc;
  note_fenced(amax);
  if (bn_x) {
    LMKD_REQUIRE(!accumulate && aligned16(bn_x) && aligned16(bn_stats), "lmkd_conv2d_bwd_data_seg: the fused BatchNorm sums need a plain (non-accumulating) launch and aligned operands");
    LMKD_REQUIRE(bwd_data_bn_ok(a, Cin), "lmkd_conv2d_bwd_data_seg: this launch has no fused form (lmkd_conv2d_bwd_data_bn_tiles_seg returned 0)");
    a.bnb_x = bn_x; a.bnb_stats = bn_stats; a.stat_partial = part;
  }
  return launch_conv_gemm<false, 0>(a, Cin, (hipStream_t)stream);
}
extern "C" int lmkd_conv2d_bwd_data_bn(const float* dy, const float* wd, float* dx, const float* bn_x, const float* bn_stats, float* part,
                                       int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream) {
  LMKD_REQUIRE(dy && wd && dx && bn_x && bn_stats && part, "lmkd_conv2d_bwd_data_bn: null pointer");
  LMKD_REQUIRE(aligned16(dy) && aligned16(wd) && aligned16(bn_x) && aligned16(bn_stats), "lmkd_conv2d_bwd_data_bn: operands must be 16-byte aligned");
  ConvGemmArgs a;
  if (const int rc = bwd_data_args(a, dy, wd, dx, N, H, W, Cin, Cout, KH, KW, stride, pad, 0)) return rc;
  LMKD_REQUIRE(bwd_data_bn_ok(a, Cin), "lmkd_conv2d_bwd_data_bn: this launch has no fused form (lmkd_conv2d_bwd_data_bn_tiles returned 0)");
  a.bnb_x = bn_x; a.bnb_stats = bn_stats; a.stat_partial = part;
  return launch_conv_gemm<false, 0>(a, Cin, (hipStream_t)stream);
}

// Split-K plan of the weight gradient.  Output tiles are few (Cout x 9*Cin is small) and the reduction is long, so the
// pixel range is split over gridDim.z.  The split count is chosen so that the launch is (just under) a whole number of
// rounds of resident workgroups: an arbitrary count leaves a 1.5-round launch that runs as long as a 2-round one.
static int g_wgrad_planes = 1;   // bf16 / 3xbf16 modes: weight gradient on the bf16-plane kernel (wgrad_x3.h); 0 = fp32-tile kernel
extern "C" int lmkd_conv_set_wgrad_planes(int on) { g_wgrad_planes = on ? 1 : 0; return LMKD_OK; }
// 3x3 / stride-1 weight gradients of the bf16-plane modes from a rolling LDS window of x (wgrad_win.h); 0 = im2col-gather kernel
static int g_wgrad_win16 = 1;      // three-plane modes, fp32 tensors: conv_wgrad_win16_kernel (v_mfma_f32_16x16x32_bf16); 0 = the 32x32x16 form
extern "C" int lmkd_conv_set_wgrad_win16(int on) { g_wgrad_win16 = on ? 1 : 0; return LMKD_OK; }
static int g_wgrad_window = 1;
extern "C" int lmkd_conv_set_wgrad_window(int on) { g_wgrad_window = on < 0 ? 0 : on; return LMKD_OK; }      // > 1: workgroup target of the split plan
static inline bool wgrad_win_eligible(int W, int Cs, int Cout, int KH, int KW, int stride, int pad) {
  return g_wgrad_window && g_wgrad_planes && (g_conv_x3 || g_conv_bf16) && KH == 3 && KW == 3 && stride == 1 && pad == 1 &&
         Cs % 32 == 0 && 32 + 2 * (W + 1) <= 256;
}
// pixel splits: ONE round of resident workgroups (two per CU; four of the two-wave workgroups of the one-plane modes - registers
// allow two waves per SIMD, one for the three-plane Cout = 64 instance) - measured at 200 frames (tools/wgrad_bench.py): 277 / 228 /
// 225 / 229 us per layer against 348 / 273 / 244 / 246 with twice as many, shorter slabs (the window warm-up and the slab traffic
// double) - but no slab shorter than 12 steps per warm-up chunk of the window (rows [k0 - halo, k0 + halo) precede the first MFMA)
static void wgrad_win_plan(int Mpix, int W, int Cs, int Cout, int* cob, int* splits, int* steps_per_split) {
  *cob = Cout >= 128 ? 4 : 2;
  const int tiles = cdiv(Cout, 32 * *cob) * (Cs / 32);
  const int steps = cdiv(Mpix, LMKD_BK);
  const int warm = cdiv(2 * (W + 1), LMKD_BK);
  const int resident = 256 * ((g_conv_x3 && *cob == 2) ? 2 : 8 / *cob);
  // Round 5: the number of slabs is proportional to the segment's WORK, one round of resident workgroups for a whole 400-frame episode
  // (steps x tiles = 39 200 x 4 / cob on every 3x3 layer of the trunk), not one round per call: since round 4 the support and the query call are
  // ONE launch over two frame segments, each planned on its own (so that the merged and the two-call form stay bit-identical, the plan may
  // depend on nothing but the segment) - which made every merged launch 1 024 workgroups = two rounds of shorter slabs.  Per launch over
  // 200 + 200 frames, window kernel + slab reduce (tools/wgrad_split_ab.py, profiles/r05_wgrad_split_ab.txt): 386 / 286 / 275 / 268 us ->
  // 288 - 345 / 259 / 252 / 250.  At least 128 workgroups per segment (small episodes).
  int sp = g_wgrad_window > 1 ? cdiv(g_wgrad_window, tiles)
                              : std::max((int)(((long)resident * steps + 19600L * 4 / *cob) / (39200L * 4 / *cob)), cdiv(128, tiles));      // (to nearest)
  const int cap = std::max(1, steps / (12 * warm));
  if (sp > cap) sp = cap;
  *steps_per_split = cdiv(steps, sp);
  *splits = cdiv(steps, *steps_per_split);
}
static inline bool wgrad_uses_planes(int Cs) { (void)Cs; return (g_conv_x3 || g_conv_bf16) && g_wgrad_planes; }

static void wgrad_plan(int Mpix, int Cout, int Kp, int* splits, int* steps_per_split, int* bm, int* bn, bool planes = false) {
  // both operands are K-outer here (b32 fragment reads): the 128-wide tiles (1 read per MFMA) beat 64x64 (2 per MFMA):
  // measured 91.6 vs 74.8 TFLOP/s over the trunk's weight gradients
  *bm = Cout <= 64 ? 64 : 128;
  *bn = (Kp % 128 == 0 && Kp >= 1024) ? 128 : 64;
  int per_cu = (*bm == 128 && *bn == 128) ? 2 : ((*bm == 64 && *bn == 64) ? 4 : 3);
  if (planes) {      // wgrad_x3.h: 4-wave workgroups, LDS = planes x 32 k rows x (2 bytes per column + 64) per operand
    if (*bm == 64 && *bn == 128) *bn = 64;
    const int npl = g_conv_bf16 ? 1 : 3;
    const long lds = 1L * npl * LMKD_BK * ((*bm * 2 + 64) + (*bn * 2 + 64));      // single LDS buffer
    per_cu = (int)(160 * 1024 / lds);
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
  }
  const int tiles = cdiv(Cout, *bm) * cdiv(Kp, *bn);
  const int steps = cdiv(Mpix, LMKD_BK);
  const int slots = 256 * per_cu;
  int best_sp = 1;
  double best_cost = 1e30;
  for (int r = 1; r <= 3; ++r) {            // (just under) r whole rounds of resident workgroups
    int sp = (slots * r) / tiles;
    if (sp > steps / 4) sp = steps / 4;
    if (sp < 1) sp = 1;
    const int sps = cdiv(steps, sp);
    sp = cdiv(steps, sps);
    const double rounds = (double)cdiv((long)tiles * sp, slots);
    // cost: K-steps executed per slot, plus the slab write + reduce traffic expressed in K-step units
    const double cost = rounds * sps + 1.0 * sp / 8.0;
    if (cost < best_cost) { best_cost = cost; best_sp = sp; }
  }
  // Cout <= 64 (stem, layer 1: reductions over 0.6 - 2.5 M pixels) on the bf16-plane kernel: slabs of at most 64 K-steps.
  // A rounding error of the MFMA accumulation is then relative to a 2 K-pixel partial sum instead of a 10 K-pixel one and the
  // slabs are summed by the reduce kernel (blocked summation; measured against fp64 on the 200-frame stem: relative L2 error
  // 8.3e-6 -> 1.9e-6, the fp32 MFMA kernel's level).  The extra slabs are a few tens of MB per launch.
  if (planes && *bm == 64 && *bn == 64 && best_sp < cdiv(steps, 64)) best_sp = cdiv(steps, 64);
  *steps_per_split = cdiv(steps, best_sp);
  *splits = cdiv(steps, *steps_per_split);
}

// the stem's weight gradient on stem_wgrad_kernel (wgrad_stem.h): three-plane modes, fp32 tensors, 64 output channels, rows of 17 - 128
// output pixels.  -> slabs (0: not that kind of launch), units per slab
static int g_wgrad_stem = 1;
extern "C" int lmkd_conv_set_wgrad_stem(int on) { g_wgrad_stem = on ? 1 : 0; return LMKD_OK; }
static int stem_wgrad_plan(int N, int H, int W, int Cs, int Cout, int KH, int KW, int stride, int pad, int* G) {
  if (!g_wgrad_stem || !g_wgrad_planes || !g_conv_x3 || g_lmkd_act_bf16 || Cs != 4 || Cout != 64 || KH != 7 || KW != 7 || stride != 2 || pad != 3) return 0;
  const int Ho = conv_out(H, KH, stride, pad), Wo = conv_out(W, KW, stride, pad);
  if (Wo < 17 || Wo > 128 || W > 2 * Wo || Ho < 1) return 0;      // >= 2 steps of 32 pixels per unit; LDS: 60 * (16 Wo + 64) + 30720 bytes
  const int units = N * cdiv(Ho, 2);
  int g = std::max(1, 2048 / (2 * Wo));                 // about 2048 pixels per slab (wgrad_plan: the accumulation error of a long partial sum)
  g = std::min(g, std::max(1, units / 256));            // but at least one slab per CU where the problem has that many units
  *G = g;
  return cdiv(units, g);
}

// workspace of the two-segment form (lmkd_conv2d_bwd_weight_seg).  Each segment is split into the slabs a launch of its own would
// use - the same partial sums (a slab's length bounds the MFMA accumulation error, wgrad_plan), twice the workgroups - so the
// workspace is the sum of the two plans (or the one-launch plan where the kernel has no segments: the stem's)
extern "C" long lmkd_conv2d_bwd_weight_workspace(int N, int H, int W, int Cs, int Cout, int KH, int KW, int stride, int pad);
extern "C" long lmkd_conv2d_bwd_weight_workspace_seg(int N, int H, int W, int Cs, int Cout, int KH, int KW, int stride, int pad, int seg_n0) {
  const long whole = lmkd_conv2d_bwd_weight_workspace(N, H, W, Cs, Cout, KH, KW, stride, pad);
  if (seg_n0 <= 0 || seg_n0 >= N) return whole;
  const long parts = lmkd_conv2d_bwd_weight_workspace(seg_n0, H, W, Cs, Cout, KH, KW, stride, pad) +
                     lmkd_conv2d_bwd_weight_workspace(N - seg_n0, H, W, Cs, Cout, KH, KW, stride, pad);
  return std::max(whole, parts);
}

extern "C" long lmkd_conv2d_bwd_weight_workspace(int N, int H, int W, int Cs, int Cout, int KH, int KW, int stride, int pad) {
  const int Ho = conv_out(H, KH, stride, pad), Wo = conv_out(W, KW, stride, pad);
  const int Kp = KH * kw_padded(Cs, KW) * Cs;
  {
    int G;
    if (const int slabs = stem_wgrad_plan(N, H, W, Cs, Cout, KH, KW, stride, pad, &G)) return (long)slabs * Cout * Kp * sizeof(float);
  }
  int splits, sps, bm, bn;
  wgrad_plan(N * Ho * Wo, Cout, Kp, &splits, &sps, &bm, &bn, wgrad_uses_planes(Cs));
  if (wgrad_win_eligible(W, Cs, Cout, KH, KW, stride, pad)) {
    int cob, wsplits, wsps;
    wgrad_win_plan(N * Ho * Wo, W, Cs, Cout, &cob, &wsplits, &wsps);
    splits = std::max(splits, wsplits);
  }
  return (long)splits * Cout * Kp * sizeof(float);
}

// dw_oihw[Cout,Cin,KH,KW] from x[N,H,W,Cs] (Cs >= Cin channel-padded) and dy[N,Ho,Wo,Cout]
static int conv2d_bwd_weight_impl(const float* x, const float* pre_stats, const float* dy, float* dw_oihw, float* workspace,
                                  long ws_bytes, int N, int H, int W, int Cs, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                  void* stream, int accumulate = 0, int seg_n0 = 0, const lmkd_amax_desc* am = nullptr) {
  const unsigned* amax_x = am ? (const unsigned*)am->x_words : nullptr;
  const unsigned* amax_dy = am ? (const unsigned*)am->dy_words : nullptr;
  note_fenced(am);
  LMKD_REQUIRE(x && dy && dw_oihw && workspace, "lmkd_conv2d_bwd_weight: null pointer");
  // two frame segments: each is split into the slabs a launch of its own would use (same partial sums, same accumulation error), the
  // PRE loader takes the segment's BatchNorm table, ONE slab reduce sums everything
  if (seg_n0 <= 0 || seg_n0 >= N) seg_n0 = 0;
  LMKD_REQUIRE(!seg_n0 || ((g_conv_x3 || g_conv_bf16) && g_wgrad_planes), "lmkd_conv2d_bwd_weight_seg: two frame segments exist in the bf16-plane modes");
  LMKD_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(workspace), "lmkd_conv2d_bwd_weight: operands must be 16-byte aligned");
  LMKD_REQUIRE(Cs % 32 == 0 || Cs == 4, "lmkd_conv2d_bwd_weight: channel count %d must be 4 or a multiple of 32", Cs);
  LMKD_REQUIRE(Cout % 4 == 0, "lmkd_conv2d_bwd_weight: Cout %% 4 != 0");
  const bool smallc = Cs == 4;
  LMKD_REQUIRE(!pre_stats || (!smallc && (!g_conv_bf16 || g_wgrad_planes) && !g_lmkd_act_bf16), "lmkd_conv2d_bwd_weight_pre: Cs %% 32 == 0, fp32 activations only");
  LMKD_REQUIRE(!g_lmkd_act_bf16 || (g_conv_bf16 && g_wgrad_planes), "bf16 activations need lmkd_conv_set_compute_dtype(1) and the plane weight-gradient kernel");
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.dy = dy; a.x = x; a.slab = workspace; a.pre_stats = pre_stats;
  a.h2_xw = amax_x; a.h2_dyw = amax_dy;
  a.N = N; a.Hs = H; a.Ws = W; a.Cs = Cs;
  a.Ho = conv_out(H, KH, stride, pad); a.Wo = conv_out(W, KW, stride, pad); a.Co = Cout;
  a.stride = stride; a.pad = pad; a.KH = KH; a.KW = KW; a.KWp = kw_padded(Cs, KW);
  a.Kp = KH * a.KWp * Cs;
  a.Mpix = N * a.Ho * a.Wo;
  LMKD_REQUIRE((long)N * H * W * Cs < 2147483647L && (long)a.Mpix * Cout < 2147483647L, "lmkd_conv2d_bwd_weight: tensor too large");
  hipStream_t s = (hipStream_t)stream;
  {
    int G;
    if (const int slabs = stem_wgrad_plan(N, H, W, Cs, Cout, KH, KW, stride, pad, &G)) {
      LMKD_REQUIRE(ws_bytes >= (long)slabs * Cout * a.Kp * (long)sizeof(float), "lmkd_conv2d_bwd_weight: workspace too small");
      LMKD_REQUIRE((long)a.Mpix * 64 * 4 < 0xffffffe0L && (long)N * H * W * 16 < 0xffffffe0L, "lmkd_conv2d_bwd_weight: tensor too large for 32-bit byte offsets");
      StemWgradArgs w;
      memset(&w, 0, sizeof(w));
      w.dy = dy; w.x = x; w.slab = workspace;
      w.h2_xw = amax_x; w.h2_dyw = amax_dy;
      w.N = N; w.H = H; w.W = W; w.Ho = a.Ho; w.Wo = a.Wo;
      w.upi = cdiv(a.Ho, 2); w.units = N * w.upi; w.G = G; w.slabs = slabs;
      w.RL = 16 * a.Wo + 64;
      const size_t lds = (size_t)2 * 10 * 3 * w.RL + 2 * 3 * STEM_WG_APLANE;
      // one workgroup per slab (the kernel can also walk several): a grid of exactly one resident workgroup per CU ran in TWO rounds
      // whenever a CU was not free (583 us alone, erratic beside other streams)
      const dim3 grid(slabs);
      const bool deep = cdiv(2 * a.Wo, 32) >= 6;      // dy tiles fetched six steps ahead (224-pixel images: seven steps per unit), else two
#define LMKD_STEM_WG(NPROD, D, RLC)                                                                                                      \
  do {                                                                                                                                   \
    static std::atomic<unsigned long long> attr_done{0};                                                                                 \
    lmkd_lds_attr_once(attr_done, reinterpret_cast<const void*>(&stem_wgrad_kernel<NPROD, D, RLC>), 160 * 1024);                         \
    hipLaunchKernelGGL((stem_wgrad_kernel<NPROD, D, RLC>), grid, dim3(512), lds, s, w);                                                   \
  } while (0)
      if (g_conv_x3 == 9) {
        if (deep && w.RL == 1856) LMKD_STEM_WG(9, 6, 1856);      // 224-pixel images
        else if (deep) LMKD_STEM_WG(9, 6, 0);
        else LMKD_STEM_WG(9, 2, 0);
      } else if (g_conv_h2 && amax_x && amax_dy) {      // two fp16 planes, three products
        if (deep && w.RL == 1856) LMKD_STEM_WG(3, 6, 1856);
        else if (deep) LMKD_STEM_WG(3, 6, 0);
        else LMKD_STEM_WG(3, 2, 0);
        ++g_h2_launches;
      } else {
        if (deep && w.RL == 1856) LMKD_STEM_WG(6, 6, 1856);
        else if (deep) LMKD_STEM_WG(6, 6, 0);
        else LMKD_STEM_WG(6, 2, 0);
      }
#undef LMKD_STEM_WG
      LMKD_CHECK_LAUNCH("stem_wgrad_kernel");
      if (a.Kp <= 256) {
        hipLaunchKernelGGL(wgrad_reduce_stem_kernel, dim3(Cout), dim3(256, 4), 0, s, (const float*)workspace, dw_oihw, slabs, Cout, Cin, Cs, KH, KW,
                           a.KWp, a.Kp, accumulate);
      } else {
        launch_wgrad_reduce((const float*)workspace, dw_oihw, slabs, Cout, Cin, Cs, KH, KW, a.KWp, a.Kp, accumulate, s);
      }
      LMKD_CHECK_LAUNCH("wgrad_reduce_kernel");
      return LMKD_OK;
    }
  }
  if (wgrad_win_eligible(W, Cs, Cout, KH, KW, stride, pad)) {
    WgradWinArgs w;
    memset(&w, 0, sizeof(w));
    w.dy = dy; w.x = x; w.slab = workspace; w.pre_stats = pre_stats;
    w.h2_xw = amax_x; w.h2_dyw = amax_dy;
    w.N = N; w.H = H; w.W = W; w.Cs = Cs; w.Co = Cout; w.Kp = a.Kp; w.Mpix = a.Mpix;
    w.steps_total = cdiv(a.Mpix, LMKD_BK);
    int cob;
    wgrad_win_plan(a.Mpix, W, Cs, Cout, &cob, &w.splits, &w.steps_per_split);
    w.seg_pix0 = a.Mpix; w.seg_splits0 = w.splits; w.sps1 = 0;
    if (seg_n0) {
      int sp0, sp1, cob1;
      w.seg_pix0 = seg_n0 * a.Ho * a.Wo;
      wgrad_win_plan(w.seg_pix0, W, Cs, Cout, &cob, &sp0, &w.steps_per_split);
      wgrad_win_plan(a.Mpix - w.seg_pix0, W, Cs, Cout, &cob1, &sp1, &w.sps1);
      w.seg_splits0 = sp0; w.splits = sp0 + sp1;
    }
    LMKD_REQUIRE(ws_bytes >= (long)w.splits * Cout * a.Kp * (long)sizeof(float), "lmkd_conv2d_bwd_weight: workspace too small");
    LMKD_REQUIRE((long)a.Mpix * std::max(Cs, Cout) * 4 < 2147483647L, "lmkd_conv2d_bwd_weight: tensor too large for 32-bit byte offsets");
    w.n_ct = cdiv(Cout, 32 * cob); w.n_it = Cs / 32;
    w.div_hw = make_fastdiv(H * W); w.div_w = make_fastdiv(W);
    const dim3 wgrid(8 * cdiv(w.splits, 8) * w.n_ct * w.n_it);
    const bool big = 32 + 2 * (W + 1) > 128;      // ring of 256 pixel rows instead of 128
#define LMKD_WIN(COB, NPROD, A16, PRE)                                                                                  \
  do {                                                                                                                  \
    /* Cout = 64 leaves two channel blocks: in the three-plane modes the nine taps are dealt 5 + 4 to two wave groups (four    \
       waves instead of two: 262 vs 318 us at 200 frames); with one plane the extra dy-fragment reads cost more (119 vs 94) */ \
    constexpr int TPW = (COB == 2 && NPROD != 1) ? 5 : 9;                                                               \
    constexpr int NT = 64 * COB * ((9 + TPW - 1) / TPW);                                                                \
    if (big) hipLaunchKernelGGL((conv_wgrad_win_kernel<COB, NPROD, A16, 256, PRE, TPW>), wgrid, dim3(NT), 0, s, w);     \
    else hipLaunchKernelGGL((conv_wgrad_win_kernel<COB, NPROD, A16, 128, PRE, TPW>), wgrid, dim3(NT), 0, s, w);         \
  } while (0)
#define LMKD_WIN16(COB, NPROD, PRE)                                                                                     \
  do {                                                                                                                  \
    /* Cout = 64 (two channel blocks): the second pair of waves takes the other 16 input channels of all nine taps */   \
    constexpr int JW = COB == 2 ? 1 : 2;                                                                                \
    constexpr int NT = 64 * COB * (2 / JW);                                                                             \
    if (big) hipLaunchKernelGGL((conv_wgrad_win16_kernel<COB, NPROD, 256, PRE, 9, JW>), wgrid, dim3(NT), 0, s, w);      \
    else hipLaunchKernelGGL((conv_wgrad_win16_kernel<COB, NPROD, 128, PRE, 9, JW>), wgrid, dim3(NT), 0, s, w);          \
  } while (0)
#define LMKD_WIN_MODE(COB)                                                                                              \
  do {                                                                                                                  \
    if (g_wgrad_win16 && g_conv_x3 && !g_lmkd_act_bf16) {      /* three-plane modes, fp32 tensors: the 16x16x32 MFMA form */ \
      if (g_conv_x3 == 9) { if (pre_stats) LMKD_WIN16(COB, 9, true); else LMKD_WIN16(COB, 9, false); }                  \
      else if (g_conv_h2 && w.h2_xw && w.h2_dyw) {                                                                      \
        if (pre_stats) LMKD_WIN16(COB, 3, true); else LMKD_WIN16(COB, 3, false);                                        \
        ++g_h2_launches;                                                                                                \
      }                                                                                                                 \
      else { if (pre_stats) LMKD_WIN16(COB, 6, true); else LMKD_WIN16(COB, 6, false); }                                 \
      break;                                                                                                            \
    }                                                                                                                   \
    if (g_lmkd_act_bf16) LMKD_WIN(COB, 1, true, false);                                                                 \
    else if (g_conv_bf16) { if (pre_stats) LMKD_WIN(COB, 1, false, true); else LMKD_WIN(COB, 1, false, false); }        \
    else if (g_conv_x3 == 9) { if (pre_stats) LMKD_WIN(COB, 9, false, true); else LMKD_WIN(COB, 9, false, false); }     \
    else { if (pre_stats) LMKD_WIN(COB, 6, false, true); else LMKD_WIN(COB, 6, false, false); }                         \
  } while (0)
    if (cob == 4) LMKD_WIN_MODE(4);
    else LMKD_WIN_MODE(2);
#undef LMKD_WIN_MODE
#undef LMKD_WIN16
#undef LMKD_WIN
    LMKD_CHECK_LAUNCH("conv_wgrad_win_kernel");
    launch_wgrad_reduce((const float*)workspace, dw_oihw, w.splits, Cout, Cin, Cs, KH, KW, a.KWp, a.Kp, accumulate, s);
    LMKD_CHECK_LAUNCH("wgrad_reduce_kernel");
    return LMKD_OK;
  }
  int splits, bm, bn;
  const bool planes = wgrad_uses_planes(Cs);
  wgrad_plan(a.Mpix, Cout, a.Kp, &splits, &a.steps_per_split, &bm, &bn, planes);
  a.seg_pix0 = a.Mpix; a.seg_splits0 = splits; a.sps1 = 0;
  if (seg_n0) {
    LMKD_REQUIRE(planes, "lmkd_conv2d_bwd_weight_seg: the fp32-tile kernel has no two-segment form");
    int sp0, sp1, bm1, bn1;
    a.seg_pix0 = seg_n0 * a.Ho * a.Wo;
    wgrad_plan(a.seg_pix0, Cout, a.Kp, &sp0, &a.steps_per_split, &bm, &bn, planes);
    wgrad_plan(a.Mpix - a.seg_pix0, Cout, a.Kp, &sp1, &a.sps1, &bm1, &bn1, planes);      // (the tile shape depends on Cout and Kp only)
    a.seg_splits0 = sp0; splits = sp0 + sp1;
  }
  a.steps_total = cdiv(a.Mpix, LMKD_BK);
  LMKD_REQUIRE(ws_bytes >= (long)splits * Cout * a.Kp * (long)sizeof(float), "lmkd_conv2d_bwd_weight: workspace too small");
  a.div_hw = make_fastdiv(a.Ho * a.Wo); a.div_w = make_fastdiv(a.Wo);
  a.n_mt = cdiv(Cout, bm); a.n_jt = cdiv(a.Kp, bn); a.splits = splits;
  // all column tiles of one pixel split read the same x / dy rows: with >= 32 splits they are dealt to ONE XCD (ids congruent
  // mod 8), whose L2 then serves the re-reads.  PMC: FETCH_SIZE 1053 -> 279 MB per launch (algorithmic 160-320 MB); the kernel
  // alone is a few % slower on layer 1, the two-stream episode is unchanged (20.5 episodes/s), so it is on by default;
  // lmkd_conv_set_xcd_mode(1) keeps the plain order
  a.xcd_mode = (g_xcd_mode != 0 && g_xcd_mode != 1 && splits >= 32) ? 1 : 0;
  dim3 grid((a.xcd_mode ? 8 * cdiv(splits, 8) : splits) * a.n_mt * a.n_jt);
#define LMKD_WGRAD_LAUNCH(CFG, SM, THR)                                                                       \
  do {                                                                                                          \
    if (g_conv_bf16) hipLaunchKernelGGL((conv_wgrad_kernel<CFG, SM, true>), grid, dim3(THR), 0, s, a);          \
    else if (!SM && pre_stats) hipLaunchKernelGGL((conv_wgrad_kernel<CFG, false, false, true>), grid, dim3(THR), 0, s, a); \
    else hipLaunchKernelGGL((conv_wgrad_kernel<CFG, SM, false>), grid, dim3(THR), 0, s, a);                     \
  } while (0)
  if (planes) {
#define LMKD_WGX3(CFG)                                                                                                   \
  do {                                                                                                                   \
    if (g_lmkd_act_bf16) {                                                                                               \
      if (smallc) hipLaunchKernelGGL((conv_wgrad_x3_kernel<CFG, true, 1, false, true>), grid, dim3(256), 0, s, a);      \
      else hipLaunchKernelGGL((conv_wgrad_x3_kernel<CFG, false, 1, false, true>), grid, dim3(256), 0, s, a);            \
    } else if (smallc) {                                                                                                 \
      if (g_conv_bf16) hipLaunchKernelGGL((conv_wgrad_x3_kernel<CFG, true, 1, false>), grid, dim3(256), 0, s, a);       \
      else if (g_conv_x3 == 9) hipLaunchKernelGGL((conv_wgrad_x3_kernel<CFG, true, 9, false>), grid, dim3(256), 0, s, a); \
      else hipLaunchKernelGGL((conv_wgrad_x3_kernel<CFG, true, 6, false>), grid, dim3(256), 0, s, a);                    \
    } else if (g_conv_bf16) {                                                                                            \
      if (pre_stats) hipLaunchKernelGGL((conv_wgrad_x3_kernel<CFG, false, 1, true>), grid, dim3(256), 0, s, a);         \
      else hipLaunchKernelGGL((conv_wgrad_x3_kernel<CFG, false, 1, false>), grid, dim3(256), 0, s, a);                  \
    } else if (g_conv_x3 == 9) {                                                                                         \
      if (pre_stats) hipLaunchKernelGGL((conv_wgrad_x3_kernel<CFG, false, 9, true>), grid, dim3(256), 0, s, a);         \
      else hipLaunchKernelGGL((conv_wgrad_x3_kernel<CFG, false, 9, false>), grid, dim3(256), 0, s, a);                  \
    } else {                                                                                                             \
      if (pre_stats) hipLaunchKernelGGL((conv_wgrad_x3_kernel<CFG, false, 6, true>), grid, dim3(256), 0, s, a);         \
      else if (g_conv_h2 && a.h2_xw && a.h2_dyw) {                                                                      \
        hipLaunchKernelGGL((conv_wgrad_x3_kernel<CFG, false, 3, false>), grid, dim3(256), 0, s, a);                      \
        ++g_h2_launches;                                                                                                 \
      } else hipLaunchKernelGGL((conv_wgrad_x3_kernel<CFG, false, 6, false>), grid, dim3(256), 0, s, a);                 \
    }                                                                                                                    \
  } while (0)
    using W128 = WgCfg<128, 128, 2, 2>;
    using W128x64 = WgCfg<128, 64, 2, 2>;
    using W64 = WgCfg<64, 64, 2, 2>;
    LMKD_REQUIRE(!smallc || (bm == 64 && bn == 64), "lmkd_conv2d_bwd_weight: padded-stem path expects Cout <= 64");
    if (bm == 128 && bn == 128) LMKD_WGX3(W128);
    else if (bm == 128) LMKD_WGX3(W128x64);
    else LMKD_WGX3(W64);
#undef LMKD_WGX3
    LMKD_CHECK_LAUNCH("conv_wgrad_x3_kernel");
  } else {
  using C64 = TileCfg<64, 64, 2, 2>;
  using C128x64 = TileCfg<128, 64, 2, 2>;
  using C64x128 = TileCfg<64, 128, 2, 2>;
  using C128 = TileCfg<128, 128, 2, 4>;     // 8 waves: 4 per SIMD at 2 workgroups per CU
  if (smallc) {
    LMKD_REQUIRE(bm == 64 && bn == 64, "lmkd_conv2d_bwd_weight: padded-stem path expects Cout <= 64");
    LMKD_WGRAD_LAUNCH(C64, true, LMKD_THREADS);
  } else if (bm == 64 && bn == 64) {
    LMKD_WGRAD_LAUNCH(C64, false, LMKD_THREADS);
  } else if (bm == 128 && bn == 64) {
    LMKD_WGRAD_LAUNCH(C128x64, false, LMKD_THREADS);
  } else if (bm == 64 && bn == 128) {
    LMKD_WGRAD_LAUNCH(C64x128, false, LMKD_THREADS);
  } else {
    LMKD_WGRAD_LAUNCH(C128, false, 512);
  }
  LMKD_CHECK_LAUNCH("conv_wgrad_kernel");
  }
  launch_wgrad_reduce((const float*)workspace, dw_oihw, splits, Cout, Cin, Cs, KH, KW, a.KWp, a.Kp, accumulate, s);      // (stride-2 3x3 and 1x1 layers)
  LMKD_CHECK_LAUNCH("wgrad_reduce_kernel");
  return LMKD_OK;
}

extern "C" int lmkd_conv2d_bwd_weight(const float* x, const float* dy, float* dw_oihw, float* workspace, long ws_bytes,
                                      int N, int H, int W, int Cs, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                      void* stream) {
  return conv2d_bwd_weight_impl(x, nullptr, dy, dw_oihw, workspace, ws_bytes, N, H, W, Cs, Cin, Cout, KH, KW, stride, pad, stream);
}

// weight gradient of a convolution whose input was relu(BatchNorm(x_raw)) computed in the forward loader (lmkd_conv2d_fwd_pre):
// the same transform is applied to x_raw here, so the normalised activation is never stored
extern "C" int lmkd_conv2d_bwd_weight_pre(const float* x_raw, const float* pre_stats, const float* dy, float* dw_oihw,
                                          float* workspace, long ws_bytes, int N, int H, int W, int Cs, int Cin, int Cout, int KH,
                                          int KW, int stride, int pad, void* stream) {
  LMKD_REQUIRE(pre_stats, "lmkd_conv2d_bwd_weight_pre: BatchNorm table of the input missing");
  return conv2d_bwd_weight_impl(x_raw, pre_stats, dy, dw_oihw, workspace, ws_bytes, N, H, W, Cs, Cin, Cout, KH, KW, stride, pad, stream);
}

// dw_oihw += weight gradient (dw_oihw is the parameter's .grad: gradient accumulation over the two trunk calls of an episode and
// over the episodes between two optimizer steps, trainwandb.py:141-143, without a separate add pass); pre_stats nullable
extern "C" int lmkd_conv2d_bwd_weight_acc(const float* x, const float* pre_stats, const float* dy, float* dw_oihw, float* workspace,
                                          long ws_bytes, int N, int H, int W, int Cs, int Cin, int Cout, int KH, int KW, int stride,
                                          int pad, void* stream) {
  return conv2d_bwd_weight_impl(x, pre_stats, dy, dw_oihw, workspace, ws_bytes, N, H, W, Cs, Cin, Cout, KH, KW, stride, pad, stream, 1);
}


// weight gradient over two frame segments [0, seg_n0) | [seg_n0, N) of one tensor (lmkd_conv2d_fwd_seg): ONE launch + ONE slab reduce for
// both trunk calls of an episode; pre_stats (nullable) = [2][5][Cs] table; accumulate: dw_oihw += ; workspace:
// lmkd_conv2d_bwd_weight_workspace_seg bytes.  The split-K slabs are those of two separate launches; they are summed in one pass.
extern "C" int lmkd_conv2d_bwd_weight_seg(const float* x, const float* pre_stats, const float* dy, float* dw_oihw, float* workspace, long ws_bytes,
                                          int N, int H, int W, int Cs, int Cin, int Cout, int KH, int KW, int stride, int pad, int accumulate,
                                          int seg_n0, void* stream, const lmkd_amax_desc* amax) {
  return conv2d_bwd_weight_impl(x, pre_stats, dy, dw_oihw, workspace, ws_bytes, N, H, W, Cs, Cin, Cout, KH, KW, stride, pad, stream, accumulate, seg_n0, amax);
}

// stride-2 data gradient: does the LDS-patch kernel run its four parity classes (patch_halo() of the launch's arguments >= 0)?
static bool dgrad_s2_patch_ok(int H, int W, int Cout, int KH, int KW, int pad) {
  if (!g_conv_patch || !(g_conv_x3 || g_conv_bf16) || Cout % 32 != 0) return false;
  const int Hr = (H + 1) / 2, Wr = (W + 1) / 2;
  if (conv_out(H, KH, 2, pad) != Hr || conv_out(W, KW, 2, pad) != Wr) return false;
  int halo = 0;
  for (int c = 0; c < 4; ++c)
    for (int kh = 0; kh < KH; ++kh)
      for (int kw = 0; kw < KW; ++kw) {
        const int nh = (c >> 1) + pad - kh, nw = (c & 1) + pad - kw;
        if ((nh & 1) || (nw & 1)) continue;
        const int sft = (nh >= 0 ? nh / 2 : -((-nh) / 2)) * Wr + (nw >= 0 ? nw / 2 : -((-nw) / 2));
        halo = std::max(halo, sft < 0 ? -sft : sft);
      }
  return halo <= PATCH_HALO_MAX;
}

// Launch plan of a convolution, without launching: which kernel instance and tile order the three entry points above would
// pick for these shapes in the current arithmetic mode (info: 5 ints).  info[0] = tile id (lmkd_conv_set_tile numbering; the weight gradient
// reports 1 = 128x128/8 waves, 2 = 128x64, 3 = 64x64, 4 = 64x128), info[1] = XCD tile order (forward / data gradient: 0 row
// bands, 1 column slices; weight gradient: 1 = all tiles of a pixel split on one XCD), info[2] = pixel splits (weight gradient)
// or parity classes (data gradient), info[3] = workgroups launched.  Used by the parity tests to prove that the benchmark's
// kernel instances are the ones under test.  info[4] = 1 when the launch runs on the LDS-patch kernel (conv_patch.h) or, for the
// weight gradient, on the rolling-window kernel (wgrad_win.h; it reports tile id 5 = 128 output channels per workgroup, 6 = 64);
// info[4] = 2: the stem's input-row patch kernel (conv_stem.h).
extern "C" int lmkd_conv2d_plan(int kind, int N, int H, int W, int Cs, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                int* info) {
  LMKD_REQUIRE(info && kind >= 0 && kind <= 2, "lmkd_conv2d_plan: kind must be 0 (forward), 1 (data gradient) or 2 (weight gradient)");
  LMKD_REQUIRE(N > 0 && H > 0 && W > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0, "lmkd_conv2d_plan: bad shape");
  const int Ho = conv_out(H, KH, stride, pad), Wo = conv_out(W, KW, stride, pad);
  if (kind == 2) {
    const int Kp = KH * kw_padded(Cs, KW) * Cs;
    int splits, sps, bm, bn;
    wgrad_plan(N * Ho * Wo, Cout, Kp, &splits, &sps, &bm, &bn, wgrad_uses_planes(Cs));
    info[0] = (bm == 128 && bn == 128) ? 1 : (bm == 128 ? 2 : (bn == 128 ? 4 : 3));
    info[1] = (g_xcd_mode != 0 && g_xcd_mode != 1 && splits >= 32) ? 1 : 0;
    info[2] = splits;
    info[3] = (info[1] ? 8 * cdiv(splits, 8) : splits) * cdiv(Cout, bm) * cdiv(Kp, bn);
    info[4] = 0;
    if (wgrad_win_eligible(W, Cs, Cout, KH, KW, stride, pad)) {      // rolling-window kernel (wgrad_win.h): tile id 5 / 6 = 128 / 64 output channels
      int cob, wsplits, wsps;
      wgrad_win_plan(N * Ho * Wo, W, Cs, Cout, &cob, &wsplits, &wsps);
      info[0] = cob == 4 ? 5 : 6; info[1] = 1; info[2] = wsplits;
      info[3] = 8 * cdiv(wsplits, 8) * cdiv(Cout, 32 * cob) * (Cs / 32);
      info[4] = 1;
    }
    return LMKD_OK;
  }
  if (kind == 0 && stem_patch_pitch(W, Cs, Cout, KH, KW, stride, pad)) {      // conv_stem_patch_kernel: tile id 14, one workgroup per output row pair
    info[0] = 14; info[1] = 0; info[2] = 1; info[3] = N * cdiv(Ho, 2); info[4] = 2;
    return LMKD_OK;
  }
  long rows;
  int nclass = 1, ncols;
  if (kind == 0) { rows = (long)N * Ho * Wo; ncols = Cout; }
  else {
    ncols = Cin;
    if (stride == 1) rows = (long)N * H * W;
    else { nclass = 4; rows = (long)N * ((H + 1) / 2) * ((W + 1) / 2); }
  }
  const bool s2patch = kind == 1 && stride == 2 && dgrad_s2_patch_ok(H, W, Cout, KH, KW, pad);
  const bool same = conv_same_size(H, W, KH, KW, stride, pad) || s2patch || (kind == 0 && fwd_s2_patch_ok(N, H, W, Cs, Cout, KH, KW, stride, pad));
  int id = pick_conv_cfg(rows, nclass, ncols, same);
  const bool patch = same && (g_conv_x3 || g_conv_bf16) && (kind == 0 ? Cs : Cout) % 32 == 0;      // patch_halo() >= 0 for these launches
  if (!patch) id = id == 10 ? 7 : (id == 11 ? 9 : (id == 12 ? 8 : id));
  const int n_rt = nclass * cdiv(rows, cfg_bm(id)), n_ct = cdiv(ncols, cfg_bn(id));
  int xm = (n_ct >= 8 && (n_ct & 7) == 0) ? 1 : 0;
  if (g_xcd_mode == 0) xm = 0;
  info[0] = id; info[1] = xm; info[2] = nclass; info[3] = xcd_grid(n_rt, n_ct, xm); info[4] = patch ? 1 : 0;
  return LMKD_OK;
}
