// The stem convolution (7x7 / stride 2 / pad 3 on the channel-padded NHWC4 input, Cout <= 64) from an LDS-resident patch of INPUT
// ROWS (bf16-plane modes).
//
// conv_gemm_x3_kernel<.., SMALLC> gathers, for every output pixel and kernel row, the 8 input pixels x 4 channels under that row
// (32 contiguous floats) and splits them: an input pixel is loaded and split 7 x 4 = 28 times, there are 7 K-steps of one barrier
// each per 128-pixel tile, and the kernel runs at 77-81 TFLOP/s of fp32-equivalent work (157 in bf16) where the 3x3 layers reach 190.
//
// Here a workgroup computes TWO output rows (2 * Wo pixels: 224 = 7 MFMA row blocks for the 224^2 frames) of one image.  It needs the
// input rows 2 * oh0 - 3 .. 2 * oh0 + 5 (9 rows); they are loaded once, split once into the bf16 planes and stored in LDS WITH their
// zero padding (3 + 5 columns, rows outside the image): plane[row][col] of 8 bytes (4 channels).  The A fragment of output pixel
// (r, ow), kernel row kh, k-group g, lane half h is then the 16 bytes at pixel (2 r + kh, 2 ow + 4 g + 2 h) - two input pixels x 4
// channels = 8 consecutive k of the (kh, kw, c) order the packed weights already have - with NO masking: consecutive lanes read
// consecutive 16-byte pieces, and kh / g / plane are instruction immediates.  One barrier per tile instead of one per K-step, the
// split work per output pixel falls from 224 to 36 elements.  Weights: fragment order from global memory straight into registers
// (X3FragB), prefetched one kernel row ahead.  Wave w owns row block w (32 pixels) x both 32-column blocks.
#pragma once

#define STEM_MAX_WP 240      // patch row pitch in pixels: W + 3 + 5 <= 240

// OR: output rows per workgroup (2: eight waves, 1: four waves)
// NPROD == 3: two fp16 planes, three products (conv_patch16.h: h2_split4; lmkd_conv_set_compute_dtype(4)): the weights' fp16 planes in
// this kernel's 32x32x16 fragment order are the last region of the weight buffer (split_weights_h2_32_kernel)
template <int NPROD, bool OUT16, int OR>
__global__ __launch_bounds__(256 * OR) void conv_stem_patch_kernel(ConvGemmArgs a) {
  constexpr int NPL = NPROD == 1 ? 1 : (NPROD == 3 ? 2 : 3);
  constexpr int NTHR = 256 * OR;
  using Cfg = X3Cfg<128 * OR, 64, 4 * OR, 1>;      // 4 OR waves x (32 rows x 64 columns): TM = 1, TN = 2
  using LB = X3FragB<Cfg::TN, NPL>;
  constexpr int KH = 7, PR = 2 * (OR - 1) + KH;      // patch rows
  extern __shared__ __attribute__((aligned(16))) unsigned char psm[];      // [plane][PR][WP] x 8 bytes
  __shared__ int s_out[Cfg::BM];
  __shared__ float s_red[Cfg::WM * Cfg::BN * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rt = blockIdx.x;                       // tile = (image, output row pair)
  const int pairs = (a.Ho + OR - 1) / OR;
  const int n = rt / pairs, oh0 = (rt - n * pairs) * OR;
  const int WP = a.halo;                           // row pitch of the patch in pixels (host: even, >= 2 (Wo - 1) + 8)
  const int plane_bytes = PR * WP * 8;
  const int ih0 = 2 * oh0 - 3;
  float h2_sx = 1.f, h2_ix = 1.f, h2_iw = 1.f;
  const unsigned short* wpk = reinterpret_cast<const unsigned short*>(a.wpk);
  if constexpr (NPROD == 3) {
    const long nw = (long)a.Co * a.Kp;
    h2_sx = h2_scale(amax_read(a.h2_xw, (a.seg_m0 > 0 && (long)n * a.Ho * a.Wo >= a.seg_m0) ? 1 : 0));      // this image's frame segment
    h2_ix = 1.f / h2_sx;
    h2_iw = 1.f / h2_scale(*reinterpret_cast<const unsigned*>(wpk + nw * 16));
    wpk += nw * 16 + 32;
  }
  // ---- patch: input rows ih0 .. ih0 + PR - 1, columns -3 .. WP - 4, zero outside the image.  Thread -> (column, row parity);
  // every thread's loads are issued before the first one is split (a load -> split -> store loop waits one memory latency per trip)
  {
    static_assert(STEM_MAX_WP <= 256, "one thread per patch column");
    constexpr int RSTEP = NTHR / 256, NIT = (PR + RSTEP - 1) / RSTEP;
    const __amdgpu_buffer_rsrc_t rs = x3_rsrc(a.src, (long)a.N * a.Hs * a.Ws * 16);
    const int pc = tid & 255, iw = pc - 3;
    u32x4 rp[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pr = (tid >> 8) + RSTEP * it, ih = ih0 + pr;
      const bool ok = pr < PR && (unsigned)ih < (unsigned)a.Hs && (unsigned)iw < (unsigned)a.Ws;
      rp[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)(((n * a.Hs + ih) * a.Ws + iw) * 16) : X3_OOB, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pr = (tid >> 8) + RSTEP * it;
      if (pr >= PR || pc >= WP) continue;
      const float4 v = make_float4(__uint_as_float(rp[it].x), __uint_as_float(rp[it].y), __uint_as_float(rp[it].z), __uint_as_float(rp[it].w));
      unsigned char* d = psm + (pr * WP + pc) * 8;
      if (NPL == 1) {
        *reinterpret_cast<uint2*>(d) = x3_round4(v);
      } else if constexpr (NPROD == 3) {
        uint2 q0, q1;
        h2_split4(v, h2_sx, q0, q1);
        *reinterpret_cast<uint2*>(d) = q0;
        *reinterpret_cast<uint2*>(d + plane_bytes) = q1;
      } else {
        uint2 q0, q1, q2;
        x3_split4(v, q0, q1, q2);
        *reinterpret_cast<uint2*>(d) = q0;
        *reinterpret_cast<uint2*>(d + plane_bytes) = q1;
        *reinterpret_cast<uint2*>(d + 2 * plane_bytes) = q2;
      }
    }
  }
  const int npix = OR * a.Wo;                      // output pixels of the tile (a second row may lie below the image)
  for (int r = tid; r < Cfg::BM; r += NTHR) {
    const int orow = r / a.Wo, ow = r - orow * a.Wo;
    s_out[r] = (r < npix && oh0 + orow < a.Ho) ? ((n * a.Ho + oh0 + orow) * a.Wo + ow) * a.Co : -1;
  }
  // this lane's A row: output pixel m = 32 wave + (lane & 31) -> patch pixel (2 r, 2 ow) + lane-half offset
  const int m = wave * 32 + (lane & 31);
  const bool active = __builtin_amdgcn_readfirstlane(wave * 32) < npix;      // waves past the tile's pixels only join the barriers
  int a_off = 0;
  {
    const int mm = m < npix ? m : 0;               // rows past the tile read pixel 0 (finite values; never stored: s_out = -1)
    const int orow = mm / a.Wo, ow = mm - orow * a.Wo;
    a_off = ((2 * orow) * WP + 2 * ow + 2 * (lane >> 5)) * 8;
  }
  LB lb;
  lb.init(wpk, a.Co, a.Kp, 0, lane);
  f32x16 acc[1][Cfg::TN];
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.f;
  u32x4 rb0[LB::NR], rb1[LB::NR];
  if (active) lb.load(0, rb0);
  __syncthreads();
  if (active) {
    auto step = [&](int kh, u32x4 (&rb)[LB::NR], u32x4 (&rbn)[LB::NR]) {
      if (kh + 1 < KH) lb.load((kh + 1) * 32, rbn);
      bf16x8 av[2][NPL];
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int p = 0; p < NPL; ++p) av[g][p] = *reinterpret_cast<const bf16x8*>(psm + a_off + (kh * WP + 4 * g) * 8 + p * plane_bytes);
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) {
          f32x16 c = acc[0][j];
          if constexpr (NPROD == 1) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[g][0], x3_as_bf16(rb[j * 2 + g]), c, 0, 0, 0);
          } else if constexpr (NPROD == 3) {      // two fp16 planes: smallest terms first
            const bf16x8 b0 = x3_as_bf16(rb[j * 4 + g * 2 + 0]), b1 = x3_as_bf16(rb[j * 4 + g * 2 + 1]);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[g][1]), __builtin_bit_cast(f16x8, b0), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[g][0]), __builtin_bit_cast(f16x8, b1), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[g][0]), __builtin_bit_cast(f16x8, b0), c, 0, 0, 0);
          } else {
            const bf16x8 b0 = x3_as_bf16(rb[j * 6 + g * 3 + 0]), b1 = x3_as_bf16(rb[j * 6 + g * 3 + 1]), b2 = x3_as_bf16(rb[j * 6 + g * 3 + 2]);
            if (NPROD == 9) {
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[g][2], b2, c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[g][1], b2, c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[g][2], b1, c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[g][1], b1, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[g][0], b2, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[g][2], b0, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[g][0], b1, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[g][1], b0, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[g][0], b0, c, 0, 0, 0);
          }
          acc[0][j] = c;
        }
    };
    step(0, rb0, rb1); step(1, rb1, rb0); step(2, rb0, rb1); step(3, rb1, rb0); step(4, rb0, rb1); step(5, rb1, rb0); step(6, rb0, rb1);
  }
  // rows past the tile (the eighth row block at 224 pixels, a second output row below the image) multiplied real patch pixels:
  // they are not stored (s_out < 0) and must not enter the BatchNorm sums either
#pragma unroll
  for (int e = 0; e < 16; ++e)
    if (s_out[wave * 32 + acc_row(e, lane)] < 0)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) acc[0][j][e] = 0.f;
  if constexpr (NPROD == 3) {      // 2^-sx, 2^-sw: exact, one after the other
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[0][j][e] = acc[0][j][e] * h2_ix * h2_iw;
  }
  x3_epilogue<Cfg, true, OUT16>(a, acc, s_out, s_red, rt, 0, wave, 0, lane, tid);
}
