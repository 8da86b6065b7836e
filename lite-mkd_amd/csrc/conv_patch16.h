// conv_patch_x3_kernel (conv_patch.h) on v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_32x32x16_bf16 (round 3).
//
// Same algorithm, same LDS patch, same loader: only the matrix instruction, the fragment geometry and the epilogue differ.  Why: with
// every other instruction of the kernel left in place, issuing the same FLOPs as 16x16x32 MFMAs ran the four 3x3 layers at 200 frames
// in 227 / 196 / 195 / 220 us against 255 / 224 / 215 / 235 (tools/patch_ablate.py, ablation 32) - the chip holds a higher clock under
// this shape (MI355X_MICROARCH.md: 1.12-1.15x at equal cycles per FLOP on random data).
//
// Operand roles are swapped with respect to conv_patch.h: A (rows of the 16x16 result) = 16 OUTPUT CHANNELS of the weight operand,
// B (columns) = 16 PIXELS of the patch.  A lane then holds, for pixel lane % 16, the four consecutive channels 4 * (lane / 16) + e of a
// 16-channel block: the epilogue stores 16 bytes per lane (64 contiguous bytes per pixel and instruction) instead of 4, and the
// BatchNorm sums of a channel are a 16-lane row reduction.  Fragments: a lane's 8 bf16 of either operand are k = 8 * (lane / 16) .. + 7
// of its row / column, i.e. ONE ds_read_b128 per (16-pixel block, plane) covers the whole 32-channel K-step (conv_patch.h: two reads of a
// 32-pixel block, one per 16-k group - the same number of reads per FLOP), and the weights come in a fragment order of their own
// (lmkd_conv2d_split_weights writes it behind the 32x32x16 order: Wf16[n / 16][k / 32][plane][lane][8]).
// Three-plane modes with fp32 tensors only (NPROD 6 / 9); bf16 tensors and the one-plane mode stay on conv_patch_x3_kernel.
//
// LDS banking of the patch reads.  ds_read_b128 is served in four 16-lane groups that mix two k-quarters: {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS).  With column j = pixel j and k-quarter q at byte 16 q of the
// 208-byte patch row, the eight lanes of quarter q + 1 land 16 bytes behind rows that are 5 rows away from a lane of quarter q in the
// same group (13 x 5 = 65 = 1 mod 16 slots): every read was 2-way conflicted (SQ_LDS_BANK_CONFLICT 49 % of the kernel's LDS cycles).
// Both operands are free to be permuted: the lane group q reads K-SLOT (q & 1) * 2 + (q >> 1) (so the two quarters of a lane group sit
// 32 bytes apart; the weight fragments are written in the same order), and the columns {4..11} carry the EVEN pixels of the 16-block,
// the columns {0..3, 12..15} the ODD ones (patch16_pixel): 13 x + 2 = 13 y has no solution with x odd, y even - no conflicts left.
#pragma once
#ifndef LMKD_ABL
#define LMKD_ABL 0
#endif

__device__ __host__ __forceinline__ constexpr int patch16_kslot(int q) { return ((q & 1) << 1) | (q >> 1); }      // self-inverse
__device__ __forceinline__ int patch16_pixel(int j) { return (j >= 4 && j < 12) ? 2 * (j - 4) : (j < 4 ? 2 * j + 1 : 2 * (j - 8) + 1); }

// v + (v of the lane a DPP row permutation pairs this lane with): quad_perm [1,0,3,2] (0xB1), [2,3,0,1] (0x4E), row_half_mirror (0x141),
// row_mirror (0x140) in turn leave the sum of a 16-lane row in every lane of it - on the VALU, no ds_bpermute (64 of them per wave and
// tile in the BatchNorm-sum epilogue before)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  return dpp_add<0x140>(v);
}

// NPW = 3: the bf16 planes; NPW = 2: the fp16 planes of the two-plane arithmetic (lmkd_conv_set_compute_dtype(4)), a third region of the
// weight buffer:  [32x32x16 order: W x 3 | -W x 3][16x16x32 order: W x 3 | -W x 3][16x16x32 order, fp16: W x 2 | -W x 2][max |w| : 64 bytes]
template <int TN, int NPW = 3>
struct X3FragB16 {
  static constexpr int NT = 2 * TN;      // 16-channel blocks of this wave
  static constexpr int NR = NT * NPW;
  __amdgpu_buffer_rsrc_t rs;
  unsigned off[NT];
  __device__ __forceinline__ void init(const void* wf, int ncols, int Kp, int col0 /* of this wave */, int lane, bool neg) {
    const long plane_bytes = (long)ncols * Kp * 2;
    const long copy_bytes = plane_bytes * NPW;
    const long base = NPW == 3 ? (neg ? 9 : 6) * plane_bytes : (neg ? 14 : 12) * plane_bytes;
    rs = x3_rsrc(reinterpret_cast<const unsigned char*>(wf) + base, copy_bytes);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int nt = (col0 >> 4) + j;
      off[j] = (nt * 16 < ncols) ? (unsigned)(((long)nt * (Kp >> 5) * NPW * 64 + lane) * 16) : X3_OOB;
    }
  }
  __device__ __forceinline__ void load(int koff /* multiple of 32, wave-uniform */, u32x4 (&reg)[NR]) const {
    // the K offset is the same for every lane: it travels in the SCALAR offset of the buffer load (the range check looks at the vector
    // offset alone, so an out-of-range column block - off = X3_OOB - still reads zeros), and the loop-invariant off[j] is the whole
    // vector offset: no vector instruction per load (before: an add and a select each - the kernel is bound by vector-instruction issue)
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((koff >> 5) * NPW * 1024));
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int p = 0; p < NPW; ++p)
        reg[j * NPW + p] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[j], so + (unsigned)(p * 1024), 0);
  }
};

// fp32 K-major packed weights Wp[col][Kp] -> the 16x16x32 fragment order, W planes followed by -W planes (3 x ncols x Kp bf16 each)
__global__ void split_weights16_kernel(const float* __restrict__ wp, unsigned short* __restrict__ wf, int ncols, int Kp) {
  const long total = (long)ncols * Kp;
  const int G = Kp >> 5;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int col = (int)(idx / Kp), k = (int)(idx - (long)col * Kp);
    const float x = wp[idx];
    const int lane = (col & 15) + 16 * patch16_kslot((k >> 3) & 3);      // the lane group that reads this k-slot
    const long o = ((((long)(col >> 4) * G + (k >> 5)) * 3) * 64 + lane) * 8 + (k & 7);
    const unsigned b0 = __float_as_uint(x);
    const float r1 = x - __uint_as_float(b0 & 0xffff0000u);
    const unsigned b1 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(b1 & 0xffff0000u);
    const unsigned short p0 = (unsigned short)(b0 >> 16), p1 = (unsigned short)(b1 >> 16), p2 = (unsigned short)(__float_as_uint(r2) >> 16);
    wf[o] = p0;
    wf[o + 512] = p1;
    wf[o + 1024] = p2;
    const long o2 = o + total * 3;
    wf[o2] = p0 ^ 0x8000u;
    wf[o2 + 512] = p1 ^ 0x8000u;
    wf[o2 + 1024] = p2 ^ 0x8000u;
  }
}

// (the two-plane arithmetic's helpers - h2_scale, h2_split4, h2_split1 - live in h2.h: conv_patch.h uses them too)
// the fp16 planes of the packed weights: split_weights16_kernel's fragment order with two planes per slot (W, then -W), scaled by the
// power of two of the word behind them (max |w|, written by amax_kernel before this launch)
__global__ void split_weights16_h2_kernel(const float* __restrict__ wp, unsigned short* __restrict__ wh, int ncols, int Kp) {
  const long total = (long)ncols * Kp;
  const int G = Kp >> 5;
  const float s = h2_scale(*reinterpret_cast<const unsigned*>(wh + total * 4));
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int col = (int)(idx / Kp), k = (int)(idx - (long)col * Kp);
    const int lane = (col & 15) + 16 * patch16_kslot((k >> 3) & 3);
    const long o = ((((long)(col >> 4) * G + (k >> 5)) * 2) * 64 + lane) * 8 + (k & 7);
    unsigned short p0, p1;
    h2_split1(wp[idx] * s, p0, p1);
    wh[o] = p0;
    wh[o + 512] = p1;
    const long o2 = o + total * 2;
    wh[o2] = p0 ^ 0x8000u;
    wh[o2 + 512] = p1 ^ 0x8000u;
  }
}
// ... and in conv_x3.h's 32x32x16 fragment order (X3FragB with two planes; W only - the stem's forward, the one two-plane kernel on that
// MFMA shape, has no -W tiles), behind the maximum's 64 bytes: wh32 = wh + 4 * total + 32
__global__ void split_weights_h2_32_kernel(const float* __restrict__ wp, unsigned short* __restrict__ wh, int ncols, int Kp) {
  const long total = (long)ncols * Kp;
  const int G = Kp >> 4;
  const float s = h2_scale(*reinterpret_cast<const unsigned*>(wh + total * 4));
  unsigned short* wh32 = wh + total * 4 + 32;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int col = (int)(idx / Kp), k = (int)(idx - (long)col * Kp);
    const int lane = (col & 31) + 32 * ((k >> 3) & 1);
    const long o = ((((long)(col >> 5) * G + (k >> 4)) * 2) * 64 + lane) * 8 + (k & 7);
    unsigned short p0, p1;
    h2_split1(wp[idx] * s, p0, p1);
    wh32[o] = p0;
    wh32[o + 512] = p1;
  }
}
// max |x| over n floats folded into *word (the fp32 bits of a non-negative number order like unsigned integers); *word zeroed by the caller
// SLOTS: into the slots of an activation maximum (amax_commit); else into the one word of a weight pack
template <bool SLOTS>
__global__ void amax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ word) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
  if (SLOTS) { amax_commit(word, m); return; }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(word, __float_as_uint(m));
}

// the offset table s_adp of a tile (conv_patch16_x3_kernel): row r of the tile, every tap -> patch offset in units of 16 bytes (the zero row
// where the tap leaves the image or the row lies past the tile's segment)
template <class Cfg, int NB, int ROWB>
__device__ __forceinline__ void patch16_fill_adp(unsigned short* __restrict__ s_adp, const ConvGemmArgs& a, const Tap* __restrict__ taps, int ntap,
                                                 int tid, int row0, int M, int halo, int P) {
  for (int r = tid; r < Cfg::BM; r += Cfg::THREADS) {
    const int m = row0 + r;
    const int within = r % (Cfg::TM * 32), px = within & 15;
    const int l16 = (px & 1) ? ((px >> 1) + (px < 8 ? 0 : 8)) : (px >> 1) + 4;      // the lane (mod 16) whose pixel this is: patch16_pixel(l16) == px
    unsigned short* dst = s_adp + ((r / (Cfg::TM * 32)) * 16 + l16) * NB + (within >> 4);
    int hh = -0x40000000, ww = 0;      // rows past the tile's segment: every tap reads the zero row
    if (m < M) {
      const int n = fdiv(m, a.div_hw);
      const int rem = m - n * a.Hs * a.Ws;
      hh = fdiv(rem, a.div_w);
      ww = rem - hh * a.Ws;
    }
    for (int tp = 0; tp < ntap; ++tp) {
      const int y = hh + taps[tp].dh, x = ww + taps[tp].dw;
      const bool in = (unsigned)y < (unsigned)a.Hs && (unsigned)x < (unsigned)a.Ws;
      dst[tp * Cfg::BM] = (unsigned short)((in ? (r + halo + taps[tp].dh * a.Ws + taps[tp].dw) : P) * (ROWB / 16));
    }
  }
}

// SRC2: a stride-2 FORWARD convolution (3x3 / pad 1 on an even-sized input).  The input's four parity classes x[:, sph::2, spw::2] are
// images of the OUTPUT's size, and a tap (kh, kw) reads one of them at a shift of -1 or 0 pixels: the convolution is the sum of four
// same-size convolutions with 4 + 2 + 2 + 1 taps.  The K loop runs (chunk, class, tap of the class); the patch of a (chunk, class) is the
// class image's pixel range of the tile - gathered from every second pixel of every second row (a pixel's 32-channel chunk is 128
// contiguous bytes) through a per-tile table of source offsets, s_src - and everything behind the patch store is the same-size kernel.
// (before: conv_gemm_x3_kernel's im2col gather, 125-135 TFLOP/s on the three stride-2 3x3 layers.)
// BNB (round 5): the launch is a data gradient that leaves the sums of the BatchNorm backward it feeds (ConvGemmArgs::bnb_x) - a template
// parameter, not a run-time branch inside the unrolled store loop: there every one of the 16 store blocks carried ~20 register copies for the
// merges of the two sum forms (46 vector instructions per 16-byte store), and the plain instances held the BatchNorm tables' registers for nothing.
template <class Cfg, int NPROD, bool PRE, bool EP, bool SRC2 = false, bool BNB = false>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::MINW) void conv_patch16_x3_kernel(ConvGemmArgs a) {
  static_assert(!BNB || (!PRE && !EP && !SRC2), "the BatchNorm-backward sums belong to plain data-gradient launches");
  static_assert(NPROD == 6 || NPROD == 9 || NPROD == 3, "three bf16 planes (6 / 9 products) or two fp16 planes (3 products)");
  static_assert(!SRC2 || !PRE, "the stride-2 source form has no BatchNorm loader");
  constexpr int NPL = 3;
  constexpr int NPU = NPROD == 3 ? 2 : 3;      // planes stored / read / multiplied (the LDS rows keep their three-plane pitch)
  constexpr int ROWB = PatchRow<NPL>::BYTES;
  constexpr int LPR = 8;                                  // lanes per patch row (16 bytes each)
  constexpr int RPP = Cfg::THREADS / LPR;                 // patch rows per pass
  constexpr int NI = (Cfg::BM + 2 * PATCH_HALO_MAX + RPP - 1) / RPP;
  constexpr int NB = 2 * Cfg::TM, NC = 2 * Cfg::TN;       // 16-pixel blocks / 16-channel blocks of a wave
  using LB = X3FragB16<Cfg::TN, NPU>;
  extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
  __shared__ int s_out[Cfg::BM];
  __shared__ int s_tap_kofs[LMKD_MAX_TAPS];
  // s_adp[tap][wave row][lane % 16][pixel block]: byte offset, inside the patch, of the row a (tap, output pixel) reads - the zero row
  // where the tap leaves the image.  A lane's NB = 4 entries of a tap are one 8-byte read per K-step (before: a mask test, an add and a
  // select per pixel block and step - the kernel is bound by vector-instruction issue)
  static_assert(NB == 4 && PatchRow<NPL>::BYTES % 16 == 0 && PatchRow<NPL>::BYTES / 16 * (Cfg::BM + 2 * PATCH_HALO_MAX + 1) < 65536,
                "four 16-bit patch offsets (in units of 16 bytes) per lane and tap");
  __shared__ __attribute__((aligned(8))) unsigned short s_adp[LMKD_MAX_TAPS * Cfg::BM];
  __shared__ float s_red[Cfg::WM * Cfg::BN * 2];
  __shared__ int s_src[SRC2 ? Cfg::BM + 2 * PATCH_HALO_MAX : 1];      // SRC2: byte offset of patch row j's pixel in class (0, 0), -1 outside
  const int tid = threadIdx.x;
  STAMP(0);
  int rt, ct;
  if (!xcd_decode(blockIdx.x, a.n_rt, a.n_ct, a.xcd_mode, rt, ct)) return;
  const int tile = rt / a.nclass;
  const int cls = rt - tile * a.nclass;
  const int ph = cls >> 1, pw = cls & 1;
  const int ntap = a.ntap[cls];
  if (a.accum && ntap == 0) return;      // out += 0
  const Tap* taps = a.taps[cls];
  const int halo = a.halo, P = Cfg::BM + 2 * halo;
  const SegTile sg = seg_tile(a, tile, Cfg::BM);      // the tile's frame segment (ConvGemmArgs::seg_m0): rows past its end are not this tile's
  const int M = sg.mend;
  const int row0 = sg.row0, n0 = ct * Cfg::BN;
  if (tid < ntap) {
    const Tap tp = taps[tid];
    s_tap_kofs[tid] = tp.kofs;
  }
  for (int r = tid; r < Cfg::BM; r += Cfg::THREADS) {
    const int m = row0 + r;
    int ob = -1;
    if (m < M) {
      if (a.nclass == 1) {
        ob = m * a.Co;
      } else {
        const int n = fdiv(m, a.div_hw);
        const int rem = m - n * a.Hs * a.Ws;
        const int aa = fdiv(rem, a.div_w), bb = rem - aa * a.Ws;
        const int oh = aa * a.omul + ph, ow = bb * a.omul + pw;
        if (oh < a.Ho && ow < a.Wo) ob = ((n * a.Ho + oh) * a.Wo + ow) * a.Co;
      }
    }
    s_out[r] = ob;
  }
  for (int j = tid; j < ROWB / 4; j += Cfg::THREADS) reinterpret_cast<unsigned*>(psm + (long)P * ROWB)[j] = 0u;   // the zero row
  if constexpr (SRC2) {
    for (int j = tid; j < P; j += Cfg::THREADS) {
      const int m = row0 - halo + j;
      int off = -1;
      if (m >= 0 && m < a.rows_per_class) {
        const int n = fdiv(m, a.div_hw);
        const int rem = m - n * a.Hs * a.Ws;
        const int aa = fdiv(rem, a.div_w), bb = rem - aa * a.Ws;
        off = (((n * 2 * a.Hs + 2 * aa) * a.s2_wfull + 2 * bb) * a.Cs) * 4;
      }
      s_src[j] = off;
    }
  }
  const int lane = tid & 63, wave = tid >> 6;
  float h2_sx = 1.f, h2_ix = 1.f, h2_iw = 1.f;
  bool h2_one = true;
  if constexpr (NPROD == 3) {
    h2_sx = h2_scale(amax_read(a.h2_xw, sg.seg));      // the operand's maximum over this tile's frame segment
    const float sw = h2_scale(*reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(a.wpk) + (long)a.Co * a.Kp * 16));
    h2_ix = 1.f / h2_sx;      // exact: powers of two within 2^+-126
    h2_iw = 1.f / sw;
    const int ex = (int)((__float_as_uint(h2_ix) >> 23) & 0xffu) + (int)((__float_as_uint(h2_iw) >> 23) & 0xffu) - 254;
    h2_one = ex >= -126 && ex <= 126;      // (uniform) the product of the two is a normal number: one multiplication in the epilogue
    if (h2_one) h2_ix *= h2_iw;
  }
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int kq = lane >> 4;                               // lane group: result rows 4 kq .. 4 kq + 3; k-slot patch16_kslot(kq) of a 32-k step
  const int ksl = patch16_kslot(kq);
  const int pxl = patch16_pixel(lane & 15);               // this lane's pixel of each 16-pixel block
  // 128 x 64 tiles: the table is written behind the first patch loads, below (layer 1, two chunks per tile: 415-437 -> 399-402 us at 400
  // frames in fp32h2).  The 128 x 128 tiles write it HERE, before the accumulators and the loader's registers are live: written below,
  // the loop's temporaries push that instance (230 registers) over its budget - layer 2: 240 -> 290 us.  SRC2 takes its source offsets
  // from an LDS table, so its loads wait for the barrier anyway.
  constexpr bool EARLY = !SRC2 && Cfg::BN == 64;
  if constexpr (!EARLY) {      // (written out: as a call of patch16_fill_adp the 128 x 128 two-plane instance spills - 256 registers + 224 bytes of scratch)
    for (int r = tid; r < Cfg::BM; r += Cfg::THREADS) {
      const int m = row0 + r;
      const int within = r % (Cfg::TM * 32), px = within & 15;
      const int l16 = (px & 1) ? ((px >> 1) + (px < 8 ? 0 : 8)) : (px >> 1) + 4;
      unsigned short* dst = s_adp + ((r / (Cfg::TM * 32)) * 16 + l16) * NB + (within >> 4);
      int hh = -0x40000000, ww = 0;
      if (m < M) {
        const int n = fdiv(m, a.div_hw);
        const int rem = m - n * a.Hs * a.Ws;
        hh = fdiv(rem, a.div_w);
        ww = rem - hh * a.Ws;
      }
      for (int tp = 0; tp < ntap; ++tp) {
        const int y = hh + taps[tp].dh, x = ww + taps[tp].dw;
        const bool in = (unsigned)y < (unsigned)a.Hs && (unsigned)x < (unsigned)a.Ws;
        dst[tp * Cfg::BM] = (unsigned short)((in ? (r + halo + taps[tp].dh * a.Ws + taps[tp].dw) : P) * (ROWB / 16));
      }
    }
  }
  const unsigned lane_off = (unsigned)(16 * ksl);      // this lane group's k-slot inside a patch row
  const unsigned short* adp_lane = s_adp + (wm * 16 + (lane & 15)) * NB;
  // patch loader (conv_patch.h): LPR lanes x 16 B per pixel row
  const __amdgpu_buffer_rsrc_t prs = x3_rsrc(a.src, (long)a.N * a.Hs * a.Ws * a.Cs * 4 * (SRC2 ? 4 : 1));
  const int pk = (tid & (LPR - 1)) * 4;
  const long pix0 = (long)row0 - halo + tid / LPR;
  const unsigned p_off0 = (unsigned)((pix0 * a.Cs + pk) * 4), p_step = (unsigned)(RPP * a.Cs * 4);
  unsigned p_ok = 0;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const long pix = pix0 + RPP * i;
    if (tid / LPR + RPP * i < P && pix >= 0 && pix < (long)a.N * a.Hs * a.Ws) p_ok |= 1u << i;
  }
  u32x4 rp[NI];
  int k_cc_abl = 0;
  float4 psc = float4(), psh = float4();
  const float* pre_tab = PRE ? a.pre_stats + (long)sg.seg * 5 * a.Cs : nullptr;      // the [5][Cs] BatchNorm table of this tile's segment
  auto issue_patch = [&](int cc, int sc = 0) {
    if ((LMKD_ABL & 32) && cc > 0) return;
    if constexpr (SRC2) {      // patch row j = pixel s_src[j] of class (0, 0), moved to class sc's origin
      const unsigned rel = (unsigned)((a.s2_off[sc] + cc * 32 + pk) * 4);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (i * RPP >= P) continue;
        const int j = tid / LPR + RPP * i;
        const int so = j < P ? s_src[j] : -1;
        rp[i] = __builtin_amdgcn_raw_buffer_load_b128(prs, so >= 0 ? (unsigned)so + rel : X3_OOB, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (i * RPP < P) rp[i] = __builtin_amdgcn_raw_buffer_load_b128(prs, ((p_ok >> i) & 1u) ? p_off0 + i * p_step + (unsigned)(cc * 32 * 4) : X3_OOB, 0, 0);
    if (PRE) {
      psc = *reinterpret_cast<const float4*>(pre_tab + 2 * a.Cs + cc * 32 + pk);
      psh = *reinterpret_cast<const float4*>(pre_tab + 3 * a.Cs + cc * 32 + pk);
    }
  };
  auto store_patch = [&]() {
    if ((LMKD_ABL & 64) && k_cc_abl > 0) return;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int j = tid / LPR + RPP * i;
      if (j >= P) continue;
      unsigned char* d = psm + j * ROWB + pk * 2;
      float4 v = make_float4(__uint_as_float(rp[i].x), __uint_as_float(rp[i].y), __uint_as_float(rp[i].z), __uint_as_float(rp[i].w));
      if (PRE && ((p_ok >> i) & 1u)) {      // relu(BatchNorm(raw)), bit-identical to bn_apply_kernel; padding never reaches here
        v.x = fmaxf(fmaf(v.x, psc.x, psh.x), 0.f); v.y = fmaxf(fmaf(v.y, psc.y, psh.y), 0.f);
        v.z = fmaxf(fmaf(v.z, psc.z, psh.z), 0.f); v.w = fmaxf(fmaf(v.w, psc.w, psh.w), 0.f);
      }
      uint2 q0, q1, q2;
      if constexpr (NPROD == 3) {
        if (LMKD_ABL & 4) { q0 = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y)); q1 = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w)); }
        else h2_split4(v, h2_sx, q0, q1);
      } else x3_split4(v, q0, q1, q2);
      *reinterpret_cast<uint2*>(d) = q0;
      *reinterpret_cast<uint2*>(d + 64) = q1;
      if (NPU == 3) *reinterpret_cast<uint2*>(d + 128) = q2;
    }
  };
  const bool neg = x3_neg_tile(sg.ltile, sg.ltiles);      // half the row tiles (of the segment) accumulate -y: X3FragB::init
  if (NPROD == 3 && neg) h2_ix = -h2_ix;      // the epilogue's one multiplication also undoes the sign of a -W tile
  LB lb;
  lb.init(a.wpk, a.Co, a.Kp, n0 + wn * (Cfg::TN * 32), lane, neg);
  f32x4 acc[NB][NC];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = ntap * a.cps;
  u32x4 rb0[LB::NR], rb1[LB::NR];
  int b_tp = 0, b_cc = 0;      // (tap, chunk) of the next weight fragments to fetch
  auto issue_b = [&](u32x4 (&rb)[LB::NR]) {
    lb.load(s_tap_kofs[b_tp] + b_cc * LMKD_BK, rb);
    if (++b_tp == ntap) { b_tp = 0; ++b_cc; }
  };
  int k_tp = 0, k_cc = 0;      // (tap, chunk) of the current K-step
  int k_sc = 0;                // SRC2: source class of the current K-step (taps are ordered by class: class c = taps s2_t0[c] .. s2_t0[c + 1] - 1)
  bf16x8 av[NPL][NB / 2];
  // K-step t (one tap of one 32-channel chunk = ONE 16x16x32 MFMA deep): see conv_patch.h for the prefetch / landing discipline
  auto step = [&](int t, u32x4 (&rb)[LB::NR], u32x4 (&rbn)[LB::NR]) {
    if (SRC2 ? k_tp == a.s2_t0[k_sc] : k_tp == 0) {
      __syncthreads();                       // every wave has finished reading the previous chunk
      store_patch();
      __syncthreads();
      if constexpr (SRC2) {
        if (k_sc + 1 < a.s2_ncls) issue_patch(k_cc, k_sc + 1);
        else if (k_cc + 1 < a.cps) issue_patch(k_cc + 1, 0);
      } else {
        if (k_cc + 1 < a.cps) issue_patch(k_cc + 1);
      }
    }
    const uint2 pk2 = *reinterpret_cast<const uint2*>(adp_lane + k_tp * Cfg::BM);
    const unsigned ad[NB] = {((pk2.x & 0xffffu) << 4) + lane_off, ((pk2.x >> 16) << 4) + lane_off, ((pk2.y & 0xffffu) << 4) + lane_off,
                             ((pk2.y >> 16) << 4) + lane_off};
    // the pixel blocks in two halves (conv_patch.h's two k-groups): one register set, the second half's fragments are read after the
    // first half's MFMAs (the set that would hold both halves costs the third workgroup per CU)
    constexpr int HB = NB / 2;
    auto read_a = [&](int hf) {
      if ((LMKD_ABL & 2) && t > 0) return;
#pragma unroll
      for (int p = 0; p < NPU; ++p)
#pragma unroll
        for (int b = 0; b < HB; ++b) av[p][b] = *reinterpret_cast<const bf16x8*>(psm + ad[hf * HB + b] + p * 64);
    };
    read_a(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if (hf == 1) {
        __builtin_amdgcn_sched_barrier(0);
        read_a(1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const bf16x8 w0 = x3_as_bf16(rb[c * NPU + 0]), w1 = x3_as_bf16(rb[c * NPU + 1]), w2 = x3_as_bf16(rb[c * NPU + (NPU - 1)]);
#pragma unroll
        for (int b = 0; b < HB; ++b) {
          f32x4 d = acc[hf * HB + b][c];
          if constexpr (NPROD == 3) {      // two fp16 planes, three products, smallest terms first
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w1), __builtin_bit_cast(f16x8, av[0][b]), d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w0), __builtin_bit_cast(f16x8, av[1][b]), d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w0), __builtin_bit_cast(f16x8, av[0][b]), d, 0, 0, 0);
            acc[hf * HB + b][c] = d;
            continue;
          }
          // rows = channels (weights), columns = pixels (patch); smallest terms first, as in conv_patch.h
          if (NPROD == 9) {
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, av[2][b], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, av[1][b], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, av[2][b], d, 0, 0, 0);
          }
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, av[1][b], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, av[0][b], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, av[2][b], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, av[0][b], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, av[1][b], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, av[0][b], d, 0, 0, 0);
          acc[hf * HB + b][c] = d;
        }
      }
    }
    ++k_tp;
    if (SRC2 && k_sc + 1 < a.s2_ncls && k_tp == a.s2_t0[k_sc + 1]) ++k_sc;
    if (k_tp == ntap) { k_tp = 0; ++k_cc; k_sc = 0; k_cc_abl = 1; }
    __builtin_amdgcn_sched_barrier(0);
    x3_landed(rbn);
    x3_landed(rp);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < nk && !(LMKD_ABL & 1)) issue_b(rb);
  };
  if constexpr (EARLY) {
    if (nk > 0) issue_patch(0);
    patch16_fill_adp<Cfg, NB, ROWB>(s_adp, a, taps, ntap, tid, row0, M, halo, P);
  }
  __syncthreads();      // tap tables, s_out, zero row
  STAMP(1);
  if (nk > 0) {
    if (!EARLY) issue_patch(0);
    issue_b(rb0);
    if (nk > 1) issue_b(rb1);
    x3_landed(rp);
    x3_landed(rb0);
    x3_landed(rb1);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(2);
    int t = 0;
    for (; t + 1 < nk; t += 2) {
      step(t, rb0, rb1);
      step(t + 1, rb1, rb0);
    }
    if (t < nk) step(t, rb0, rb1);
  }

  STAMP(3);
  // ---- epilogue: lane = (pixel patch16_pixel(lane % 16) of each 16-pixel block, channels 4 kq .. 4 kq + 3 of each 16-channel block)
  const int ch0 = n0 + wn * (Cfg::TN * 32) + 4 * kq;      // + 16 c
  float4 s1[NC], s2[NC];
  float amo = 0.f;      // max |out| of this lane (ConvGemmArgs::amax_out)
  float4 esc[NC], esh[NC];
  float4 bmean[NC], bistd[NC];      // bnb_x: the BatchNorm table of this lane's channels, once per tile (esc / esh hold scale / shift)
  constexpr bool bnb = BNB;
  const float* bnb_tab = bnb ? a.bnb_stats + (long)sg.seg * 5 * a.Co : nullptr;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    s1[c] = s2[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int col = ch0 + 16 * c;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EP || bnb) {
      const float* st = EP ? a.ep_stats : bnb_tab;
      esc[c] = col < a.Co ? *reinterpret_cast<const float4*>(st + 2 * a.Co + col) : z4;
      esh[c] = col < a.Co ? *reinterpret_cast<const float4*>(st + 3 * a.Co + col) : z4;
    }
    if (bnb) {
      bmean[c] = col < a.Co ? *reinterpret_cast<const float4*>(bnb_tab + col) : z4;
      bistd[c] = col < a.Co ? *reinterpret_cast<const float4*>(bnb_tab + a.Co + col) : z4;
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int ob = s_out[wm * (Cfg::TM * 32) + 16 * b + pxl];
    float4 prev[NC];
    if (a.accum) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int col = ch0 + 16 * c;
        prev[c] = (ob >= 0 && col < a.Co) ? *reinterpret_cast<const float4*>(a.out + (long)ob + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = ch0 + 16 * c;
      float4 v = make_float4(acc[b][c][0], acc[b][c][1], acc[b][c][2], acc[b][c][3]);
      if constexpr (NPROD == 3) {      // 2^-sx, 2^-sw (exact), the sign of the -W tiles folded in
        if (h2_one) { v.x *= h2_ix; v.y *= h2_ix; v.z *= h2_ix; v.w *= h2_ix; }
        else { v.x = v.x * h2_ix * h2_iw; v.y = v.y * h2_ix * h2_iw; v.z = v.z * h2_ix * h2_iw; v.w = v.w * h2_ix * h2_iw; }
      } else if (neg) { v.x = -v.x; v.y = -v.y; v.z = -v.z; v.w = -v.w; }
      const bool ok = ob >= 0 && col < a.Co;
      if (EP) {      // inference: the operations of bn_apply_kernel in its order (x3_epilogue<EP>)
        v.x = fmaf(v.x, esc[c].x, esh[c].x); v.y = fmaf(v.y, esc[c].y, esh[c].y); v.z = fmaf(v.z, esc[c].z, esh[c].z); v.w = fmaf(v.w, esc[c].w, esh[c].w);
        if (a.ep_res && ok) {
          const float4 r = *reinterpret_cast<const float4*>(a.ep_res + (long)ob + col);
          v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (a.ep_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      } else if (a.accum) {
        v.x += prev[c].x; v.y += prev[c].y; v.z += prev[c].z; v.w += prev[c].w;
      }
      if (ok && (!(LMKD_ABL & 8) || v.x == 123456.789f)) {
        *reinterpret_cast<float4*>(a.out + (long)ob + col) = v;
        if (!(LMKD_ABL & 256)) amo = amax4(amo, v);
      }
      if (!EP && !(LMKD_ABL & 16)) {
        if constexpr (bnb) {      // sums of the BatchNorm backward this gradient feeds (ConvGemmArgs::bnb_x): bn_bwd_reduce_kernel's terms, mask mode 2
          if (ok) {
            const float4 xv = *reinterpret_cast<const float4*>(a.bnb_x + (long)ob + col);
            const float4 mean = bmean[c], istd = bistd[c], sc = esc[c], sh = esh[c];
            const float gx = fmaf(xv.x, sc.x, sh.x) > 0.f ? v.x : 0.f, gy = fmaf(xv.y, sc.y, sh.y) > 0.f ? v.y : 0.f;
            const float gz = fmaf(xv.z, sc.z, sh.z) > 0.f ? v.z : 0.f, gw = fmaf(xv.w, sc.w, sh.w) > 0.f ? v.w : 0.f;
            s1[c].x += gx; s1[c].y += gy; s1[c].z += gz; s1[c].w += gw;
            s2[c].x = fmaf(gx, (xv.x - mean.x) * istd.x, s2[c].x); s2[c].y = fmaf(gy, (xv.y - mean.y) * istd.y, s2[c].y);
            s2[c].z = fmaf(gz, (xv.z - mean.z) * istd.z, s2[c].z); s2[c].w = fmaf(gw, (xv.w - mean.w) * istd.w, s2[c].w);
          }
        } else {      // BatchNorm partial sums (rows outside the tensor hold exact zeros: their patch rows were zero)
          s1[c].x += v.x; s1[c].y += v.y; s1[c].z += v.z; s1[c].w += v.w;
          s2[c].x = fmaf(v.x, v.x, s2[c].x); s2[c].y = fmaf(v.y, v.y, s2[c].y); s2[c].z = fmaf(v.z, v.z, s2[c].z); s2[c].w = fmaf(v.w, v.w, s2[c].w);
        }
      }
    }
  }
  if (a.amax_out) amax_commit(a.amax_out + sg.seg * LMKD_AMAX_SEG_WORDS, amo);      // (every wave of the workgroup reaches this point)
  if (!EP && a.stat_partial && !(LMKD_ABL & 16)) {
    // sum over the 16 pixel lanes of each 16-lane row, then over the WM row groups of the workgroup through LDS
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float v1[4] = {s1[c].x, s1[c].y, s1[c].z, s1[c].w}, v2[4] = {s2[c].x, s2[c].y, s2[c].z, s2[c].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v1[e] = row16_sum(v1[e]);
        v2[e] = row16_sum(v2[e]);
        if ((lane & 15) == 0) {
          const int cl = wn * (Cfg::TN * 32) + 16 * c + 4 * kq + e;
          s_red[(wm * Cfg::BN + cl) * 2 + 0] = v1[e];
          s_red[(wm * Cfg::BN + cl) * 2 + 1] = v2[e];
        }
      }
    }
    __syncthreads();
    if (tid < Cfg::BN && n0 + tid < a.Co) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < Cfg::WM; ++w) {
        t1 += s_red[(w * Cfg::BN + tid) * 2 + 0];
        t2 += s_red[(w * Cfg::BN + tid) * 2 + 1];
      }
      float* p = a.stat_partial + ((long)rt * a.Co + n0 + tid) * 2;
      p[0] = t1;
      p[1] = t2;
    }
  }
  STAMP(4);
}
