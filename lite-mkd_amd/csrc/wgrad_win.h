// Weight gradient of a 3x3 / stride 1 / pad 1 convolution from an LDS-resident, rolling WINDOW of x pixels (bf16-plane modes).
//
//   dW[co][(tap, ci)] = sum_m dy[m][co] * x[m + dh * W + dw][ci]        (m = flattened pixel, masked at the image borders)
//
// conv_wgrad_x3_kernel (wgrad_x3.h) treats this as a GEMM whose B operand is an im2col gather: every column tile (one or two taps
// of 64-128 channels) loads its own shifted copy of x, and every workgroup converts / splits its own copies of both operands:
// 8.7 VALU instructions per MFMA (16 in the 64x64 tiles) on a SIMD that issues one instruction at a time - the kernel is bound by
// the instruction issue of its operand handling, not by the matrix pipe (PMC: VALU 61 %, MFMA 42 %).
//
// Here a workgroup owns 32 input channels x (32 * COB) output channels x ALL NINE taps.  The pixel axis is walked in steps of 32;
// the x rows [k0 - halo, k0 + 32 + halo), halo = W + 1, live in a ring of R pixel rows in LDS (R = 128 or 256, a power of two
// > 63 + 2 halo, so the 32 rows that enter at step t + 1 land on slots whose rows left the window before step t: no barrier
// between the reads of step t and those writes is needed beyond the two per step that the single-buffered dy tile takes anyway).
// Each x row is loaded and split ONCE and then serves the nine taps of 32 + 2 halo steps' worth of MFMAs; the dy tile is split once
// per step and serves nine taps.  A tap is a row shift of the window: the transposed LDS read (ds_read_b64_tr_b16, wgrad_x3.h) takes
// its row address per lane, so the shift is an add + wrap on that address, and the image-border mask of (pixel, tap) is an address
// select to an all-zero row - the lane that supplies row q of a 4 x 16 block applies the mask of ITS pixel.
// A wave multiplies one output-channel block (32 rows) with TPW of the nine shifted 32-channel blocks (TPW accumulator tiles).  TPW = 9:
// COB waves, every dy fragment shared by nine taps.  Where Cout = 64 leaves two channel blocks (two waves per workgroup, one per
// SIMD) the three-plane modes deal the taps 5 + 4 to two wave groups (four waves; three taps per wave, six waves, was slower than
// either: three times the dy-fragment reads and a six-wave barrier).
// Output: split-K slabs in the layout of conv_wgrad_kernel ([split][Cout][Kp], column = tap * Cs + ci), reduced by wgrad_reduce_kernel.
#pragma once

struct WgradWinArgs {
  const float* dy;
  const float* x;
  float* slab;
  const float* pre_stats;      // PRE: x is a raw conv output; relu(BatchNorm(x)) is applied when a row enters the window (ConvGemmArgs)
  int N, H, W, Cs, Co, Kp;
  int Mpix, steps_total, steps_per_split, splits;
  int n_ct, n_it;      // output-channel tiles, input-channel tiles (32 channels each)
  // two frame segments (ConvGemmArgs::seg_m0; only needed with PRE, whose [2][5][Cs] table differs per segment): pixels [0, seg_pix0) are
  // covered by the splits [0, seg_splits0) of steps_per_split steps each, pixels [seg_pix0, Mpix) by the remaining splits of sps1 steps,
  // the first of which STARTS at pixel seg_pix0 - no slab straddles the boundary.  One segment: seg_pix0 = Mpix, seg_splits0 = splits.
  int seg_pix0, seg_splits0, sps1;
  FastDiv div_hw, div_w;
  const unsigned* h2_xw;       // two-plane fp16 form (NPROD == 3, conv_patch16.h): max |x| / max |dy| per frame segment (two words each)
  const unsigned* h2_dyw;
};
template <int COB, int NPROD, bool ACT16, int R, bool PRE = false, int TPW = 9>
__global__ __launch_bounds__(64 * COB * ((9 + TPW - 1) / TPW)) void conv_wgrad_win_kernel(WgradWinArgs a) {
  constexpr int NPL = NPROD == 1 ? 1 : 3;
  static_assert(!ACT16 || (NPROD == 1 && !PRE), "bf16 tensors: one plane, no store-side arithmetic");
  constexpr int THREADS = 64 * COB * ((9 + TPW - 1) / TPW), BM = 32 * COB;
  constexpr int LDA = BM * 2 + 64, A_PLANE = LMKD_BK * LDA;      // dy image [k][co], bytes (odd multiple of 64: wgrad_x3.h)
  constexpr int LDX = 64, X_PLANE = (R + 1) * LDX;               // x ring [slot][32 ci]; slot R = the zero row
  constexpr int ESZ = ACT16 ? 2 : 4, EPL = ACT16 ? 8 : 4;        // element size, elements per 16-byte lane access
  constexpr int A_LPR = BM / EPL, A_RPP = THREADS / A_LPR, A_NI = (LMKD_BK + A_RPP - 1) / A_RPP;      // dy loader: lanes per row, rows per pass
  constexpr int X_LPR = 32 / EPL, X_RPP = THREADS / X_LPR, X_NI = (LMKD_BK + X_RPP - 1) / X_RPP;
  static_assert(THREADS % A_LPR == 0 && THREADS % X_LPR == 0 && (LDA / 64) % 2 == 1, "tile layout");
  __shared__ __attribute__((aligned(16))) unsigned char s_dy[NPL * A_PLANE];
  __shared__ __attribute__((aligned(16))) unsigned char s_x[NPL * X_PLANE];
  __shared__ __attribute__((aligned(16))) unsigned s_adr[2][9 * LMKD_BK];      // per step: ring byte offset of (tap, pixel row), see fill_adr
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // all tiles of one pixel split run on one XCD (block ids congruent mod 8): its L2 serves their re-reads of the same x / dy rows
  const int tiles = a.n_ct * a.n_it;
  const int xj = blockIdx.x >> 3;
  const int z = (blockIdx.x & 7) + 8 * (xj / tiles);
  if (z >= a.splits) return;
  const int tz = xj - (xj / tiles) * tiles;
  const int co0 = (tz % a.n_ct) * BM, ci0 = (tz / a.n_ct) * 32;
  const WinSlab sl = win_slab(z, a.Mpix, a.steps_per_split, a.seg_pix0, a.seg_splits0, a.sps1);
  const int nk = sl.nk, kend = sl.kend;      // kend: this slab's segment ends here - pixels past it are not this slab's
  const int halo = a.W + 1;

  const __amdgpu_buffer_rsrc_t rs_dy = x3_rsrc(a.dy, (long)a.Mpix * a.Co * ESZ);
  const __amdgpu_buffer_rsrc_t rs_x = x3_rsrc(a.x, (long)a.Mpix * a.Cs * ESZ);
  // dy loader: row = pixel of the step, EPL consecutive output channels per lane
  const int a_c = (tid % A_LPR) * EPL, a_r = tid / A_LPR;
  const bool a_cin = co0 + a_c < a.Co;
  // x loader: one 16-byte piece of one pixel row
  const int x_c = (tid % X_LPR) * EPL, x_r = tid / X_LPR;
  u32x4 ra[A_NI], rx[X_NI];
  unsigned x_ok = 0;      // PRE: which of the prefetched x rows are real pixels (rows outside the tensor stay zero)
  float4 psc = float4(), psh = float4();
  if (PRE) {
    psc = *reinterpret_cast<const float4*>(a.pre_stats + (long)sl.seg * 5 * a.Cs + 2 * a.Cs + ci0 + x_c);
    psh = *reinterpret_cast<const float4*>(a.pre_stats + (long)sl.seg * 5 * a.Cs + 3 * a.Cs + ci0 + x_c);
  }
  auto load_dy = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_NI; ++i) {
      const int p = k0 + a_r + A_RPP * i;
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (a_cin && a_r + A_RPP * i < LMKD_BK && p < kend) ? (unsigned)((p * a.Co + co0 + a_c) * ESZ) : X3_OOB, 0, 0);
    }
  };
  auto load_x = [&](int q0) {      // the 32 pixel rows q0 .. q0 + 31 (rows outside the tensor read as zeros)
#pragma unroll
    for (int i = 0; i < X_NI; ++i) {
      const int q = q0 + x_r + X_RPP * i;
      const bool ok = x_r + X_RPP * i < LMKD_BK && q >= 0 && q < a.Mpix;
      rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (unsigned)((q * a.Cs + ci0 + x_c) * ESZ) : X3_OOB, 0, 0);
      if (PRE) x_ok = ok ? (x_ok | (1u << i)) : (x_ok & ~(1u << i));
    }
  };
  auto as_f4 = [](const u32x4& r) { return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)); };
  auto store_dy = [&]() {
#pragma unroll
    for (int i = 0; i < A_NI; ++i) {
      if (a_r + A_RPP * i >= LMKD_BK) continue;
      if constexpr (ACT16) tr_store8(s_dy, LDA, a_r + A_RPP * i, a_c, ra[i]);
      else tr_store4<NPL, LDA, A_PLANE>(s_dy, a_r + A_RPP * i, a_c, as_f4(ra[i]));
    }
  };
  auto store_x = [&](int q0) {
#pragma unroll
    for (int i = 0; i < X_NI; ++i) {
      if (x_r + X_RPP * i >= LMKD_BK) continue;
      const int slot = (q0 + x_r + X_RPP * i) & (R - 1);
      if constexpr (ACT16) {
        tr_store8(s_x, LDX, slot, x_c, rx[i]);
      } else {
        float4 v = as_f4(rx[i]);
        if (PRE && ((x_ok >> i) & 1u)) {      // bit-identical to bn_apply_kernel
          v.x = fmaxf(fmaf(v.x, psc.x, psh.x), 0.f); v.y = fmaxf(fmaf(v.y, psc.y, psh.y), 0.f);
          v.z = fmaxf(fmaf(v.z, psc.z, psh.z), 0.f); v.w = fmaxf(fmaf(v.w, psc.w, psh.w), 0.f);
        }
        tr_store4<NPL, LDX, X_PLANE>(s_x, slot, x_c, v);
      }
    }
  };
  // Address table of a step: for each tap and each of the step's 32 pixel rows, the byte offset (plane 0, column 0) of the ring row
  // the tap reads for that pixel - (pixel + tap shift) wrapped - or of the zero row when the tap's input pixel lies outside the image
  // (or the pixel past the tensor).  288 entries, computed ONCE per workgroup and step (every wave and 16 lanes of each wave need the
  // same row address; computing it per lane cost 2 VALU instructions per MFMA).  Entry order: a lane's four rows of a step
  // (k-group g, lo / hi read j; row = 16 g + 8 h + 4 j + q) are adjacent, so one ds_read_b128 per tap fetches them.
  auto fill_adr = [&](unsigned* tab, int k0) {
    for (int e = tid; e < 9 * LMKD_BK; e += THREADS) {
      const int tp = e >> 5, r = e & 31;
      const int p = k0 + r;
      unsigned adr = (unsigned)(R * LDX);
      if (p < kend) {
        const int n = fdiv(p, a.div_hw);
        const int rem = p - n * a.H * a.W;
        const int h = fdiv(rem, a.div_w), w = rem - h * a.W;
        const int dh = tp / 3 - 1, dw = tp - (tp / 3) * 3 - 1;
        if ((unsigned)(h + dh) < (unsigned)a.H && (unsigned)(w + dw) < (unsigned)a.W) adr = (unsigned)(((p + dh * a.W + dw) & (R - 1)) * LDX);
      }
      tab[tp * LMKD_BK + ((((r >> 3) & 1) * 4 + (r & 3)) * 4 + ((r >> 4) * 2 + ((r >> 2) & 1)))] = adr;
    }
  };

  // transposed-read lane geometry (wgrad_x3.h): this lane supplies row 8h + q (and + 4) of each 16-row k-group, 4 columns at cb
  const int cb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const int cw = wave % COB, tap0 = (wave / COB) * TPW;
  const int offA = tr_lane_off(LDA, lane) + cw * 32 * 2;
  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  if (nk > 0) {
    const int kfirst = sl.kfirst;
    for (int j = tid; j < NPL * LDX / 4; j += THREADS)      // the zero row of every plane
      reinterpret_cast<unsigned*>(s_x + (j / (LDX / 4)) * X_PLANE + R * LDX)[j % (LDX / 4)] = 0u;
    // warm-up: rows [k0 - halo, k0 + halo) in chunks of 32 (a chunk may run into the rows of step 0: rewritten with the same data)
    for (int q0 = kfirst - halo; q0 < kfirst + halo; q0 += LMKD_BK) {
      load_x(q0);
      store_x(q0);
    }
    fill_adr(s_adr[0], kfirst);
    load_dy(kfirst);
    load_x(kfirst + halo);
    for (int t = 0; t < nk; ++t) {
      const int k0 = kfirst + t * LMKD_BK;
      store_dy();                             // waits for the prefetched registers of step t
      store_x(k0 + halo);
      __syncthreads();
      if (t + 1 < nk) {
        load_dy(k0 + LMKD_BK);
        load_x(k0 + LMKD_BK + halo);
        fill_adr(s_adr[(t + 1) & 1], k0 + LMKD_BK);
      }
      // ---- MFMAs of step t
      bf16x8 fa[2][NPL];
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int p = 0; p < NPL; ++p) fa[g][p] = tr_frag<LDA>(s_dy, offA + p * A_PLANE + g * 16 * LDA);
#pragma unroll
      for (int ti = 0; ti < TPW; ++ti) {
        const int tp = tap0 + ti;
        if (9 % TPW != 0 && ti >= 9 - TPW * (9 / TPW) && tp >= 9) break;      // only the last wave group's trailing taps can be missing
        // this lane's four ring rows under tap tp: [k-group 0 lo, hi, k-group 1 lo, hi]
        const u32x4 a4 = *reinterpret_cast<const u32x4*>(&s_adr[t & 1][tp * LMKD_BK + ((lane >> 5) * 4 + ((lane & 15) >> 2)) * 4]);
        f32x16 c = acc[ti];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const unsigned ad[2] = {(g ? a4.z : a4.x) + (unsigned)cb, (g ? a4.w : a4.y) + (unsigned)cb};
          bf16x8 fb[NPL];
#pragma unroll
          for (int p = 0; p < NPL; ++p) {
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(s_x + p * X_PLANE + ad[0]));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(s_x + p * X_PLANE + ad[1]));
            union { s16x4_t s[2]; bf16x8 b; } u;
            u.s[0] = lo; u.s[1] = hi;
            fb[p] = u.b;
          }
          if constexpr (NPROD == 1) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][0], fb[0], c, 0, 0, 0);
          } else {
            if (NPROD == 9) {
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][2], fb[2], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][1], fb[2], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][2], fb[1], c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][1], fb[1], c, 0, 0, 0);     // smallest terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][0], fb[2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][2], fb[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][0], fb[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][1], fb[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][0], fb[0], c, 0, 0, 0);
          }
        }
        acc[ti] = c;
      }
      __syncthreads();                        // every wave has read the dy tile (and this step's window) before they are overwritten
    }
  }

  float* C = a.slab + (long)z * a.Co * a.Kp;
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti) {
    if (9 % TPW != 0 && tap0 + ti >= 9) break;
    const int col = (tap0 + ti) * a.Cs + ci0 + (lane & 31);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = co0 + cw * 32 + acc_row(e, lane);
      if (row < a.Co) C[(long)row * a.Kp + col] = acc[ti][e];
    }
  }
}

// ---- the same kernel on v_mfma_f32_16x16x32_bf16 (round 3; three-plane modes, fp32 tensors).  Same window, same loaders, same
// slabs; what changes is the fragment geometry (conv_patch16.h has the reason: the chip holds a higher clock under this shape):
//   * one MFMA takes the whole 32-pixel step; the 32 x 32 (co x ci) tile of a (wave, tap) is four 16 x 16 accumulators.
//   * k order inside a step: lane group q4 = lane / 16, element j <-> pixel 16 (j / 4) + 4 q4 + (j % 4) (wgrad_stem.h): the lower /
//     upper transposed read of a lane group covers rows 4 q4 .. 4 q4 + 3 / 16 + 4 q4 .. + 3, so a 32-lane half reads EIGHT consecutive
//     rows of one 16-column block at once.
//   * banks: dy rows are BM * 2 + 32 bytes (an odd multiple of 32: eight consecutive rows' 32-byte segments tile the 256-byte bank
//     row).  The x ring keeps its 64-byte rows; rows s and s + 4 would collide, so the two 32-byte halves (ci 0-15 | 16-31) of the
//     slots with bit 2 set are SWAPPED when a row is stored: eight consecutive slots then tile the bank row for either half.  The
//     address table carries the swap bit in bit 5 of an entry, and the reader's block select is an XOR with 32.
//   * address table of a step: entry (tap, row slot 4 q4 + q) = {lower row, upper row}: one ds_read_b64 per lane and tap.
// JW: 16-channel blocks of x per wave (2 = the whole 32-channel tile).  Where Cout = 64 leaves two channel blocks, the second pair of
// waves takes the OTHER 16 input channels of all nine taps (TPW = 9, JW = 1: 108 MFMAs per wave and step for every wave) instead of
// four of the nine taps (TPW = 5: 120 / 96 - the 4-tap waves idle a fifth of the time): 244 -> measured in DESIGN 8.8.
template <int COB, int NPROD, int R, bool PRE = false, int TPW = 9, int JW = 2>
__global__ __launch_bounds__(64 * COB * ((9 + TPW - 1) / TPW) * (2 / JW), JW == 2 ? 2 : 3) void conv_wgrad_win16_kernel(WgradWinArgs a) {
  static_assert(JW == 2 || (JW == 1 && TPW == 9), "the channel-block split exists for nine taps per wave");
  static_assert(NPROD == 6 || NPROD == 9 || NPROD == 3, "three bf16 planes (6 / 9 products) or two fp16 planes (3 products: conv_patch16.h)");
  constexpr int NPL = NPROD == 3 ? 2 : 3;
  constexpr int THREADS = 64 * COB * ((9 + TPW - 1) / TPW) * (2 / JW), BM = 32 * COB;
  constexpr int LDA = BM * 2 + 32, A_PLANE = LMKD_BK * LDA;      // dy image [k][co], bytes
  constexpr int LDX = 64, X_PLANE = (R + 1) * LDX;               // x ring [slot][32 ci]; slot R = the zero row
  constexpr int A_LPR = BM / 4, A_RPP = THREADS / A_LPR, A_NI = (LMKD_BK + A_RPP - 1) / A_RPP;      // dy loader: lanes per row, rows per pass
  constexpr int X_LPR = 8, X_RPP = THREADS / X_LPR, X_NI = (LMKD_BK + X_RPP - 1) / X_RPP;
  static_assert(THREADS % A_LPR == 0 && THREADS % X_LPR == 0 && (LDA / 32) % 2 == 1, "tile layout");
  __shared__ __attribute__((aligned(16))) unsigned char s_dy[NPL * A_PLANE];
  __shared__ __attribute__((aligned(16))) unsigned char s_x[NPL * X_PLANE];
  __shared__ __attribute__((aligned(16))) unsigned s_adr[2][9 * LMKD_BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles = a.n_ct * a.n_it;
  const int xj = blockIdx.x >> 3;
  const int z = (blockIdx.x & 7) + 8 * (xj / tiles);
  if (z >= a.splits) return;
  const int tz = xj - (xj / tiles) * tiles;
  const int co0 = (tz % a.n_ct) * BM, ci0 = (tz / a.n_ct) * 32;
  const WinSlab sl = win_slab(z, a.Mpix, a.steps_per_split, a.seg_pix0, a.seg_splits0, a.sps1);
  const int nk = sl.nk, kend = sl.kend;      // kend: this slab's segment ends here - pixels past it are not this slab's
  const int halo = a.W + 1;

  const __amdgpu_buffer_rsrc_t rs_dy = x3_rsrc(a.dy, (long)a.Mpix * a.Co * 4);
  const __amdgpu_buffer_rsrc_t rs_x = x3_rsrc(a.x, (long)a.Mpix * a.Cs * 4);
  const int a_c = (tid % A_LPR) * 4, a_r = tid / A_LPR;
  const bool a_cin = co0 + a_c < a.Co;
  const int x_c = (tid % X_LPR) * 4, x_r = tid / X_LPR;
  u32x4 ra[A_NI], rx[X_NI];
  unsigned x_ok = 0;
  float h2_sx = 1.f, h2_sdy = 1.f;
  if constexpr (NPROD == 3) {
    h2_sx = h2_scale(amax_read(a.h2_xw, sl.seg));      // the maxima over this slab's frame segment
    h2_sdy = h2_scale(amax_read(a.h2_dyw, sl.seg));
  }
  float4 psc = float4(), psh = float4();
  if (PRE) {
    psc = *reinterpret_cast<const float4*>(a.pre_stats + (long)sl.seg * 5 * a.Cs + 2 * a.Cs + ci0 + x_c);
    psh = *reinterpret_cast<const float4*>(a.pre_stats + (long)sl.seg * 5 * a.Cs + 3 * a.Cs + ci0 + x_c);
  }
  auto load_dy = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_NI; ++i) {
      const int p = k0 + a_r + A_RPP * i;
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (a_cin && a_r + A_RPP * i < LMKD_BK && p < kend) ? (unsigned)((p * a.Co + co0 + a_c) * 4) : X3_OOB, 0, 0);
    }
  };
  auto load_x = [&](int q0) {
#pragma unroll
    for (int i = 0; i < X_NI; ++i) {
      const int q = q0 + x_r + X_RPP * i;
      const bool ok = x_r + X_RPP * i < LMKD_BK && q >= 0 && q < a.Mpix;
      rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (unsigned)((q * a.Cs + ci0 + x_c) * 4) : X3_OOB, 0, 0);
      if (PRE) x_ok = ok ? (x_ok | (1u << i)) : (x_ok & ~(1u << i));
    }
  };
  auto as_f4 = [](const u32x4& r) { return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)); };
  auto store_dy = [&]() {
#pragma unroll
    for (int i = 0; i < A_NI; ++i) {
      if (a_r + A_RPP * i >= LMKD_BK) continue;
      if constexpr (NPROD == 3) {
        uint2 p0, p1;
        h2_split4(as_f4(ra[i]), h2_sdy, p0, p1);
        unsigned char* d = s_dy + (a_r + A_RPP * i) * LDA + a_c * 2;
        *reinterpret_cast<uint2*>(d) = p0;
        *reinterpret_cast<uint2*>(d + A_PLANE) = p1;
      } else {
        tr_store4<NPL, LDA, A_PLANE>(s_dy, a_r + A_RPP * i, a_c, as_f4(ra[i]));
      }
    }
  };
  auto store_x = [&](int q0) {
#pragma unroll
    for (int i = 0; i < X_NI; ++i) {
      if (x_r + X_RPP * i >= LMKD_BK) continue;
      const int slot = (q0 + x_r + X_RPP * i) & (R - 1);
      float4 v = as_f4(rx[i]);
      if (PRE && ((x_ok >> i) & 1u)) {      // bit-identical to bn_apply_kernel
        v.x = fmaxf(fmaf(v.x, psc.x, psh.x), 0.f); v.y = fmaxf(fmaf(v.y, psc.y, psh.y), 0.f);
        v.z = fmaxf(fmaf(v.z, psc.z, psh.z), 0.f); v.w = fmaxf(fmaf(v.w, psc.w, psh.w), 0.f);
      }
      uint2 p0, p1, p2;
      if constexpr (NPROD == 3) h2_split4(v, h2_sx, p0, p1); else x3_split4(v, p0, p1, p2);
      unsigned char* d = s_x + slot * LDX + ((x_c * 2) ^ (((slot >> 2) & 1) << 5));      // halves swapped in the slots with bit 2 set
      *reinterpret_cast<uint2*>(d) = p0;
      *reinterpret_cast<uint2*>(d + X_PLANE) = p1;
      if constexpr (NPL == 3) *reinterpret_cast<uint2*>(d + 2 * X_PLANE) = p2;
    }
  };
  auto fill_adr = [&](unsigned* tab, int k0) {
    for (int e = tid; e < 9 * LMKD_BK; e += THREADS) {
      const int tp = e >> 5, r = e & 31;
      const int p = k0 + r;
      unsigned adr = (unsigned)(R * LDX);
      if (p < kend) {
        const int n = fdiv(p, a.div_hw);
        const int rem = p - n * a.H * a.W;
        const int h = fdiv(rem, a.div_w), w = rem - h * a.W;
        const int dh = tp / 3 - 1, dw = tp - (tp / 3) * 3 - 1;
        if ((unsigned)(h + dh) < (unsigned)a.H && (unsigned)(w + dw) < (unsigned)a.W) {
          const int slot = (p + dh * a.W + dw) & (R - 1);
          adr = (unsigned)(slot * LDX + (((slot >> 2) & 1) << 5));
        }
      }
      tab[tp * LMKD_BK + (r & 15) * 2 + (r >> 4)] = adr;
    }
  };

  const int q4 = lane >> 4, idx = lane & 15;
  const int cw = wave % COB, tap0 = JW == 2 ? (wave / COB) * TPW : 0, j0 = JW == 2 ? 0 : wave / COB;
  const int offA = (4 * q4 + (idx >> 2)) * LDA + (cw * 32 + 4 * (idx & 3)) * 2;      // + 32 bytes: second 16-channel block; + 16 LDA: upper rows
  const unsigned cbx = (unsigned)(8 * (idx & 3));
  f32x4 acc[TPW][2][JW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < JW; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    const int kfirst = sl.kfirst;
    for (int j = tid; j < NPL * LDX / 4; j += THREADS)      // the zero row of every plane
      reinterpret_cast<unsigned*>(s_x + (j / (LDX / 4)) * X_PLANE + R * LDX)[j % (LDX / 4)] = 0u;
    for (int q0 = kfirst - halo; q0 < kfirst + halo; q0 += LMKD_BK) {
      load_x(q0);
      store_x(q0);
    }
    fill_adr(s_adr[0], kfirst);
    load_dy(kfirst);
    load_x(kfirst + halo);
    for (int t = 0; t < nk; ++t) {
      const int k0 = kfirst + t * LMKD_BK;
      store_dy();
      store_x(k0 + halo);
      __syncthreads();
      if (t + 1 < nk) {
        load_dy(k0 + LMKD_BK);
        load_x(k0 + LMKD_BK + halo);
        fill_adr(s_adr[(t + 1) & 1], k0 + LMKD_BK);
      }
      bf16x8 fa[2][NPL];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          const unsigned char* ap = s_dy + p * A_PLANE + offA + i * 32;
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)ap);
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(ap + 16 * LDA));
          union { s16x4_t s[2]; bf16x8 b; } u;
          u.s[0] = lo; u.s[1] = hi;
          fa[i][p] = u.b;
        }
      if constexpr (NPROD == 3 && JW == 2) {
        // two planes, three products: half the MFMAs per fragment read of the three-plane form, so the reads no longer hide behind the
        // MFMAs of their own tap - the fragments of tap ti + 1 are read while tap ti multiplies (one read per MFMA gap): 379 -> 284, 352 -> 251,
        // 365 -> 265 us on layers 2-4 at 400 frames.  (JW = 1, layer 1: six MFMAs per tap - the plain loop is faster there, 334 vs 359 us)
        static_assert(NPROD != 3 || TPW == 9, "all nine taps per wave");
        auto read_fb = [&](int tp, bf16x8 (&fb)[JW][NPL]) {
          const uint2 a2 = *reinterpret_cast<const uint2*>(&s_adr[t & 1][tp * LMKD_BK + (4 * q4 + (idx >> 2)) * 2]);
#pragma unroll
          for (int j = 0; j < JW; ++j) {
            const unsigned lo_a = (a2.x ^ (unsigned)((j0 + j) << 5)) + cbx, hi_a = (a2.y ^ (unsigned)((j0 + j) << 5)) + cbx;
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
              const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(s_x + p * X_PLANE + lo_a));
              const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(s_x + p * X_PLANE + hi_a));
              union { s16x4_t s[2]; bf16x8 b; } u;
              u.s[0] = lo; u.s[1] = hi;
              fb[j][p] = u.b;
            }
          }
        };
        auto mult = [&](int ti, const bf16x8 (&fb)[JW][NPL]) {
#pragma unroll
          for (int j = 0; j < JW; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              f32x4 c = acc[ti][i][j];      // smallest terms first
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[i][1]), __builtin_bit_cast(f16x8, fb[j][0]), c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[i][0]), __builtin_bit_cast(f16x8, fb[j][1]), c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[i][0]), __builtin_bit_cast(f16x8, fb[j][0]), c, 0, 0, 0);
              acc[ti][i][j] = c;
            }
        };
        bf16x8 fb0[JW][NPL], fb1[JW][NPL];
        read_fb(tap0, fb0);
#pragma unroll
        for (int ti = 0; ti < TPW; ti += 2) {
          if (ti + 1 < TPW) read_fb(tap0 + ti + 1, fb1);
          mult(ti, fb0);
          if (ti + 1 < TPW) {
#pragma unroll
            for (int q = 0; q < JW * NPL * 2 + 1; ++q) {      // one LDS read behind each of the first MFMAs, the rest back to back
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, JW * 6 - (JW * NPL * 2 + 1), 0);
            if (ti + 2 < TPW) read_fb(tap0 + ti + 2, fb0);
            mult(ti + 1, fb1);
            if (ti + 2 < TPW) {
#pragma unroll
              for (int q = 0; q < JW * NPL * 2 + 1; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              }
              __builtin_amdgcn_sched_group_barrier(0x008, JW * 6 - (JW * NPL * 2 + 1), 0);
            }
          }
        }
      } else
#pragma unroll
      for (int ti = 0; ti < TPW; ++ti) {
        const int tp = tap0 + ti;
        if (9 % TPW != 0 && ti >= 9 - TPW * (9 / TPW) && tp >= 9) break;      // only the last wave group's trailing taps can be missing
        const uint2 a2 = *reinterpret_cast<const uint2*>(&s_adr[t & 1][tp * LMKD_BK + (4 * q4 + (idx >> 2)) * 2]);
#pragma unroll
        for (int j = 0; j < JW; ++j) {      // the 16-channel blocks of x this wave owns
          bf16x8 fb[NPL];
          const unsigned lo_a = (a2.x ^ (unsigned)((j0 + j) << 5)) + cbx, hi_a = (a2.y ^ (unsigned)((j0 + j) << 5)) + cbx;
#pragma unroll
          for (int p = 0; p < NPL; ++p) {
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(s_x + p * X_PLANE + lo_a));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(s_x + p * X_PLANE + hi_a));
            union { s16x4_t s[2]; bf16x8 b; } u;
            u.s[0] = lo; u.s[1] = hi;
            fb[p] = u.b;
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            f32x4 c = acc[ti][i][j];
            if constexpr (NPROD == 3) {      // two fp16 planes: smallest terms first
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[i][1]), __builtin_bit_cast(f16x8, fb[0]), c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[i][0]), __builtin_bit_cast(f16x8, fb[1]), c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[i][0]), __builtin_bit_cast(f16x8, fb[0]), c, 0, 0, 0);
              acc[ti][i][j] = c;
              continue;
            }
            if (NPROD == 9) {
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][NPL - 1], fb[NPL - 1], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[NPL - 1], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][NPL - 1], fb[1], c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[1], c, 0, 0, 0);     // smallest terms first
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[NPL - 1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][NPL - 1], fb[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[0], c, 0, 0, 0);
            acc[ti][i][j] = c;
          }
        }
      }
      __syncthreads();
    }
  }

  float* C = a.slab + (long)z * a.Co * a.Kp;
  const float h2_ix = 1.f / h2_sx, h2_idy = 1.f / h2_sdy;      // exact (powers of two within 2^+-126), applied one after the other
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti) {
    if (9 % TPW != 0 && tap0 + ti >= 9) break;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < JW; ++j) {
        const int col = (tap0 + ti) * a.Cs + ci0 + 16 * (j0 + j) + idx;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = co0 + cw * 32 + 16 * i + 4 * q4 + e;
          if (row < a.Co) C[(long)row * a.Kp + col] = NPROD == 3 ? acc[ti][i][j][e] * h2_ix * h2_idy : acc[ti][i][j][e];
        }
      }
  }
}
