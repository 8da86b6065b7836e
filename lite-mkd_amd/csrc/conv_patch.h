// Same-size convolutions (3x3 / stride 1 / pad 1 forward and data gradient: 13 of the 20 convolutions of the ResNet-18 trunk,
// 85 % of its FLOPs) from an LDS-resident input PATCH instead of an im2col gather.
//
// For a same-size convolution the input pixel of output pixel m (flattened n, h, w) under tap (dh, dw) is the flattened pixel
// m + dh * W + dw.  A tile of BM consecutive output pixels therefore reads the CONTIGUOUS pixel range [m0 - halo, m0 + BM + halo),
// halo = W + 1, and every tap's A operand is the same LDS image shifted by a whole number of rows.  Per 32-channel chunk a
// workgroup loads that range once (BM + 2 halo rows instead of 9 x BM gathered rows: 1.1x - 1.9x the tile instead of 9x), splits
// it into the bf16 planes once, and runs the nine taps' MFMAs against it: one pair of barriers per CHUNK, not one per K-step, and
// 1/5 - 1/8 of the loader's global loads, split arithmetic and LDS writes.  Zero padding (and rows past the tensor) is an
// address select per (row, tap): the read goes to an all-zero LDS row.
//
// LDS image: row j = pixel m0 - halo + j, ROWB bytes: [plane 0: 32 bf16][plane 1][plane 2][16 B pad] = 208 B in the three-plane
// modes, [32 bf16][16 B pad] = 80 B in one-plane (bf16) mode - odd multiples of 16 B, so the 16 rows of each ds_read_b128 lane
// group cover all 64 banks; plane and k-group offsets are instruction immediates.  Row P = BM + 2 halo is the zero row.
// The weight operand comes from global memory in fragment order, straight into registers, exactly as in conv_gemm_x3_kernel.
// The next chunk's patch is fetched into registers right after the current one is stored, i.e. a whole chunk (9 K-steps) ahead.
#pragma once

#define PATCH_HALO_MAX 57      // W <= 56

template <int NPL> struct PatchRow { static constexpr int BYTES = NPL == 1 ? 80 : 208; };

// dynamic LDS bytes of a launch
static inline size_t patch_lds_bytes(int bm, int halo, int npl) { return (size_t)(bm + 2 * halo + 1) * (npl == 1 ? 80 : 208); }

// DBG (never instantiated in the library; a measurement build passes it by hand): timing ablations that leave a part of the work out -
// results are garbage - to see what the kernel waits for: 1 = no weight-fragment loads after the first two steps, 2 = no A-fragment LDS reads after the
// first step, 4 = no patch split / LDS store (the loads are still issued), 8 = no output stores
template <class Cfg, int NPROD, bool PRE, int IO, int DBG = 0>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::MINW) void conv_patch_x3_kernel(ConvGemmArgs a) {
  constexpr bool IN16 = (IO & 1) != 0, OUT16 = (IO & 2) != 0, EP = (IO & 4) != 0;      // IO bits: conv_gemm_x3_kernel
  constexpr int NPL = NPROD == 1 ? 1 : 3;
  constexpr int NPU = NPROD == 3 ? 2 : NPL;      // NPROD == 3: two fp16 planes (h2.h) in the three-plane row pitch
  static_assert(((IO & 3) == 0 || NPROD == 1) && !(IN16 && PRE), "bf16 tensors: one-plane mode, no load-side arithmetic");
  constexpr int ROWB = PatchRow<NPL>::BYTES;
  constexpr int LPR = IN16 ? 4 : 8;                       // lanes per patch row (16 bytes each)
  constexpr int RPP = Cfg::THREADS / LPR;                 // patch rows per pass
  constexpr int NI = (Cfg::BM + 2 * PATCH_HALO_MAX + RPP - 1) / RPP;
  using LB = X3FragB<Cfg::TN, NPU>;
  extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
  __shared__ int s_out[Cfg::BM];
  __shared__ int s_tap_shift[LMKD_MAX_TAPS], s_tap_kofs[LMKD_MAX_TAPS];
  __shared__ float s_red[Cfg::WM * Cfg::BN * 2];
  const int tid = threadIdx.x;
  int rt, ct;
  if (!xcd_decode(blockIdx.x, a.n_rt, a.n_ct, a.xcd_mode, rt, ct)) return;
  // parity classes of a stride-2 data gradient (conv.hip): each class is a same-size convolution over the dy grid with its own 1-4
  // taps, its outputs scattered to the pixels (2a + ph, 2b + pw); a.nclass == 1: plain same-size convolution (omul 1, ph = pw = 0)
  const int tile = rt / a.nclass;
  const int cls = rt - tile * a.nclass;
  const int ph = cls >> 1, pw = cls & 1;
  const int ntap = a.ntap[cls];
  if (a.accum && ntap == 0) return;      // out += 0
  const Tap* taps = a.taps[cls];
  const int halo = a.halo, P = Cfg::BM + 2 * halo;
  const SegTile sg = seg_tile(a, tile, Cfg::BM);      // the tile's frame segment (ConvGemmArgs::seg_m0)
  const int M = sg.mend;
  const int row0 = sg.row0, n0 = ct * Cfg::BN;
  if (tid < ntap) {
    const Tap tp = taps[tid];
    s_tap_shift[tid] = (tp.dh * a.Ws + tp.dw) * ROWB;
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
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int h = lane >> 5;
  const int a_row = wm * (Cfg::TM * 32) + (lane & 31);
  // per MFMA row of this lane: LDS byte address of its own pixel's row (tap shift 0) and one validity bit per tap
  unsigned a_base[Cfg::TM], a_mask[Cfg::TM];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i) {
    const int r = a_row + 32 * i, m = row0 + r;
    a_base[i] = (unsigned)((r + halo) * ROWB + 16 * h);
    unsigned mk = 0;
    if (m < M) {
      const int n = fdiv(m, a.div_hw);
      const int rem = m - n * a.Hs * a.Ws;
      const int hh = fdiv(rem, a.div_w), ww = rem - hh * a.Ws;
      for (int tp = 0; tp < ntap; ++tp) {
        const int y = hh + taps[tp].dh, x = ww + taps[tp].dw;
        if ((unsigned)y < (unsigned)a.Hs && (unsigned)x < (unsigned)a.Ws) mk |= 1u << tp;
      }
    }
    a_mask[i] = mk;
  }
  const unsigned zero_addr = (unsigned)(P * ROWB + 16 * h);
  // patch loader: LPR lanes x 16 B per pixel row
  constexpr int ESZ = IN16 ? 2 : 4;
  const __amdgpu_buffer_rsrc_t prs = x3_rsrc(a.src, (long)a.N * a.Hs * a.Ws * a.Cs * ESZ);
  const int pk = ((tid & (LPR - 1))) * (IN16 ? 8 : 4);            // first channel of this lane inside the 32-channel chunk
  // byte offset of (patch row j = tid / LPR + RPP * i, channel pk) of chunk 0 is p_off0 + i * p_step where bit i of p_ok is set
  // (row inside the patch and the tensor), else the load is sent out of range (returns zeros)
  const long pix0 = (long)row0 - halo + tid / LPR;
  const unsigned p_off0 = (unsigned)((pix0 * a.Cs + pk) * ESZ), p_step = (unsigned)(RPP * a.Cs * ESZ);
  unsigned p_ok = 0;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const long pix = pix0 + RPP * i;
    if (tid / LPR + RPP * i < P && pix >= 0 && pix < (long)a.N * a.Hs * a.Ws) p_ok |= 1u << i;
  }
  u32x4 rp[NI];
  float h2_sx = 1.f, h2_ix = 1.f, h2_iw = 1.f;
  if constexpr (NPROD == 3) {
    h2_sx = h2_scale(amax_read(a.h2_xw, sg.seg));
    h2_ix = 1.f / h2_sx;
    h2_iw = 1.f / h2_scale(*reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(a.wpk) + (long)a.Co * a.Kp * 16));
  }
  float4 psc = float4(), psh = float4();
  const float* pre_tab = PRE ? a.pre_stats + (long)sg.seg * 5 * a.Cs : nullptr;      // the [5][Cs] BatchNorm table of this tile's segment
  auto issue_patch = [&](int cc) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (i * RPP < P) rp[i] = __builtin_amdgcn_raw_buffer_load_b128(prs, ((p_ok >> i) & 1u) ? p_off0 + i * p_step + (unsigned)(cc * 32 * ESZ) : X3_OOB, 0, 0);
    if (PRE) {
      psc = *reinterpret_cast<const float4*>(pre_tab + 2 * a.Cs + cc * 32 + pk);
      psh = *reinterpret_cast<const float4*>(pre_tab + 3 * a.Cs + cc * 32 + pk);
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int j = tid / LPR + RPP * i;
      if (j >= P) continue;
      unsigned char* d = psm + j * ROWB + pk * 2;
      if constexpr (IN16) {
        *reinterpret_cast<u32x4*>(d) = rp[i];
      } else {
        float4 v = make_float4(__uint_as_float(rp[i].x), __uint_as_float(rp[i].y), __uint_as_float(rp[i].z), __uint_as_float(rp[i].w));
        if (PRE && ((p_ok >> i) & 1u)) {      // relu(BatchNorm(raw)), bit-identical to bn_apply_kernel; padding never reaches here
          v.x = fmaxf(fmaf(v.x, psc.x, psh.x), 0.f); v.y = fmaxf(fmaf(v.y, psc.y, psh.y), 0.f);
          v.z = fmaxf(fmaf(v.z, psc.z, psh.z), 0.f); v.w = fmaxf(fmaf(v.w, psc.w, psh.w), 0.f);
        }
        if (NPL == 1) {
          *reinterpret_cast<uint2*>(d) = x3_round4(v);
        } else if constexpr (NPROD == 3) {
          uint2 q0, q1;
          h2_split4(v, h2_sx, q0, q1);
          *reinterpret_cast<uint2*>(d) = q0;
          *reinterpret_cast<uint2*>(d + 64) = q1;
        } else {
          uint2 q0, q1, q2;
          x3_split4(v, q0, q1, q2);
          *reinterpret_cast<uint2*>(d) = q0;
          *reinterpret_cast<uint2*>(d + 64) = q1;
          *reinterpret_cast<uint2*>(d + 128) = q2;
        }
      }
    }
  };
  LB lb;
  const bool neg = NPL == 3 && NPROD != 3 && x3_neg_tile(sg.ltile, sg.ltiles);      // half the row tiles (of the segment) accumulate -y: X3FragB::init
  if constexpr (NPROD == 3)      // the fp16 planes of W in the 32x32x16 order: behind the bf16 planes, the 16x16x32 fp16 planes and the maximum's 64 bytes
    lb.init(reinterpret_cast<const unsigned short*>(a.wpk) + (long)a.Co * a.Kp * 16 + 32, a.Co, a.Kp, n0 + wn * (Cfg::TN * 32), lane, false);
  else
    lb.init(a.wpk, a.Co, a.Kp, n0 + wn * (Cfg::TN * 32), lane, neg);
  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int nk = ntap * a.cps;
  u32x4 rb0[LB::NR], rb1[LB::NR];
  int b_tp = 0, b_cc = 0;      // (tap, chunk) of the next B fragments to fetch
  int dbg_b = 0;
  auto issue_b = [&](u32x4 (&rb)[LB::NR]) {
    if ((DBG & 1) && ++dbg_b > 2) return;
    lb.load(s_tap_kofs[b_tp] + b_cc * LMKD_BK, rb);
    if (++b_tp == ntap) { b_tp = 0; ++b_cc; }
  };
  bf16x8 dbg_av[2][NPL][Cfg::TM];      // (dead unless DBG & 2)
  int k_tp = 0, k_cc = 0;      // (tap, chunk) of the current K-step
  // One-plane modes (round 4): the A fragments of K-step t + 1 are read from LDS while the MFMAs of step t run (two register sets,
  // passed to step() in alternation like the weight sets).  With one plane a K-step is 4 MFMAs (128 cycles) per wave behind 4
  // ds_read_b128 whose latency (+ bank conflicts) is as long: read-then-multiply left the matrix pipe 31 % busy (PMC, tools/pmc_patch.sh)
  // with every tile shape.  Across a chunk boundary nothing can be prefetched (the next chunk's patch is not in LDS yet).
  constexpr bool PIPE = NPL == 1 && DBG == 0;
  bf16x8 avp0[2][NPL][Cfg::TM], avp1[2][NPL][Cfg::TM];
  auto read_tap = [&](int tp, bf16x8 (&dst)[2][NPL][Cfg::TM]) {
    const unsigned sh = (unsigned)s_tap_shift[tp];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) {
      const unsigned ad = ((a_mask[i] >> tp) & 1u) ? a_base[i] + sh : zero_addr;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int p = 0; p < NPL; ++p) dst[g][p][i] = *reinterpret_cast<const bf16x8*>(psm + ad + p * 64 + g * 32);
    }
  };
  // K-step t: MFMAs with the B set `rb`; afterwards the set fetched before them (`rbn`, step t+1) is landed - every load in flight
  // is then one MFMA phase old - and `rb` is refilled with step t+2.  At a chunk boundary the patch is replaced first.
  auto step = [&](int t, u32x4 (&rb)[LB::NR], u32x4 (&rbn)[LB::NR], bf16x8 (&avc)[2][NPL][Cfg::TM], bf16x8 (&avn)[2][NPL][Cfg::TM]) {
    if (k_tp == 0) {
      __syncthreads();                       // every wave has finished reading the previous chunk
      if (!(DBG & 4) || t == 0) store_patch();
      __syncthreads();
      if (k_cc + 1 < a.cps) issue_patch(k_cc + 1);
      if constexpr (PIPE) read_tap(0, avc);
    }
    if constexpr (PIPE) {
      if (k_tp + 1 < ntap) read_tap(k_tp + 1, avn);      // lands under this step's MFMAs
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) {
          const bf16x8 b0 = x3_as_bf16(rb[j * 2 + g]);
#pragma unroll
          for (int i = 0; i < Cfg::TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avc[g][0][i], b0, acc[i][j], 0, 0, 0);
        }
      if (++k_tp == ntap) { k_tp = 0; ++k_cc; }
      __builtin_amdgcn_sched_barrier(0);
      x3_landed(rbn);
      x3_landed(rp);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 < nk) issue_b(rb);
      return;
    }
    const unsigned sh = (unsigned)s_tap_shift[k_tp];
    unsigned ad[Cfg::TM];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) ad[i] = ((a_mask[i] >> k_tp) & 1u) ? a_base[i] + sh : zero_addr;
    // both k-groups' A fragments are read up front (group 1 lands under group 0's MFMAs) unless that would be 96 registers
    constexpr bool AHEAD = NPL * Cfg::TM < 12 && (Cfg::THREADS == 512 || NPL == 1);
    bf16x8 av_local[2][NPL][Cfg::TM];
    bf16x8 (&av)[2][NPL][Cfg::TM] = (DBG & 2) ? dbg_av : av_local;      // ablation 2: the fragments of the first step stay in registers
    auto read_a = [&](int g) {
      if ((DBG & 2) && (t > 0 || (!AHEAD && g == 1))) return;
#pragma unroll
      for (int p = 0; p < NPU; ++p)
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i) av[AHEAD ? g : 0][p][i] = *reinterpret_cast<const bf16x8*>(psm + ad[i] + p * 64 + g * 32);
    };
    read_a(0);
    if (AHEAD) read_a(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int ga = AHEAD ? g : 0;
      if (!AHEAD && g == 1) {
        __builtin_amdgcn_sched_barrier(0);
        read_a(1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
        if constexpr (NPROD == 1) {
          const bf16x8 b0 = x3_as_bf16(rb[j * 2 + g]);
#pragma unroll
          for (int i = 0; i < Cfg::TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ga][0][i], b0, acc[i][j], 0, 0, 0);
        } else if constexpr (NPROD == 3) {      // two fp16 planes, three products, smallest terms first
          const f16x8 b0 = __builtin_bit_cast(f16x8, rb[j * 4 + g * 2 + 0]), b1 = __builtin_bit_cast(f16x8, rb[j * 4 + g * 2 + 1]);
#pragma unroll
          for (int i = 0; i < Cfg::TM; ++i) {
            f32x16 c = acc[i][j];
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[ga][0][i]), b1, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[ga][1][i]), b0, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[ga][0][i]), b0, c, 0, 0, 0);
            acc[i][j] = c;
          }
        } else {
          const bf16x8 b0 = x3_as_bf16(rb[j * 6 + g * 3 + 0]), b1 = x3_as_bf16(rb[j * 6 + g * 3 + 1]), b2 = x3_as_bf16(rb[j * 6 + g * 3 + 2]);
          if constexpr ((DBG & 32) != 0) {      // power / clock probe: the same FLOPs as v_mfma_f32_16x16x32_bf16 (two per 32x32x16), garbage results
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i) {
              f32x4 c4[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) c4[q] = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
              c4[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ga][1][i], b1, c4[0], 0, 0, 0);
              c4[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ga][1][i], b1, c4[1], 0, 0, 0);
              c4[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ga][0][i], b2, c4[2], 0, 0, 0);
              c4[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ga][0][i], b2, c4[3], 0, 0, 0);
              c4[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ga][2][i], b0, c4[0], 0, 0, 0);
              c4[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ga][2][i], b0, c4[1], 0, 0, 0);
              c4[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ga][0][i], b1, c4[2], 0, 0, 0);
              c4[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ga][0][i], b1, c4[3], 0, 0, 0);
              c4[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ga][1][i], b0, c4[0], 0, 0, 0);
              c4[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ga][1][i], b0, c4[1], 0, 0, 0);
              c4[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ga][0][i], b0, c4[2], 0, 0, 0);
              c4[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ga][0][i], b0, c4[3], 0, 0, 0);
#pragma unroll
              for (int q = 0; q < 4; ++q) { acc[i][j][4 * q] = c4[q][0]; acc[i][j][4 * q + 1] = c4[q][1]; acc[i][j][4 * q + 2] = c4[q][2]; acc[i][j][4 * q + 3] = c4[q][3]; }
            }
          } else
#pragma unroll
          for (int i = 0; i < Cfg::TM; ++i) {
            f32x16 c = acc[i][j];
            if (NPROD == 9) {
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ga][2][i], b2, c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ga][1][i], b2, c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ga][2][i], b1, c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ga][1][i], b1, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ga][0][i], b2, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ga][2][i], b0, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ga][0][i], b1, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ga][1][i], b0, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ga][0][i], b0, c, 0, 0, 0);
            acc[i][j] = c;
          }
        }
      }
    }
    if (++k_tp == ntap) { k_tp = 0; ++k_cc; }
    // unconditional: with the landing under `if (t + 1 < nk)` the compiler has to assume a path on which the sets are still in
    // flight at the next step and puts s_waitcnt vmcnt(..0) between that step's MFMAs - behind the loads issued a moment ago
    // The scheduling fences keep the wait where it is written: after this step's MFMAs and before the next loads are issued.
    __builtin_amdgcn_sched_barrier(0);
    x3_landed(rbn);
    x3_landed(rp);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < nk) issue_b(rb);
  };
  __syncthreads();      // tap tables, s_out, zero row
  if (nk > 0) {
    issue_patch(0);
    issue_b(rb0);
    if (nk > 1) issue_b(rb1);
    x3_landed(rp);
    x3_landed(rb0);
    x3_landed(rb1);
    __builtin_amdgcn_sched_barrier(0);
    // two steps per trip, the odd last step outside the loop: with `if (t + 1 < nk) step(t + 1, ..)` inside it the compiler sees a
    // path (second step skipped, loop continues) on which the first step's B set was re-issued a moment ago, and guards the
    // MFMAs with s_waitcnt vmcnt(5..0) - which on the real path wait for the loads issued just before them
    int t = 0;
    for (; t + 1 < nk; t += 2) {
      step(t, rb0, rb1, avp0, avp1);
      step(t + 1, rb1, rb0, avp1, avp0);
    }
    if (t < nk) step(t, rb0, rb1, avp0, avp1);
  }
  if (DBG & 16) {      // ablation: the same bytes written with 16-byte stores (wrong element mapping): what would a transposed epilogue buy?
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ob = s_out[wm * (Cfg::TM * 32) + i * 32 + (lane & 31)];
          const int col = n0 + wn * (Cfg::TN * 32) + j * 32 + (lane >> 5) * 16 + q * 4;
          if (ob >= 0 && col < a.Co)
            *reinterpret_cast<float4*>(a.out + (long)ob + col) = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        }
    return;
  }
  if (DBG & 8) {      // ablation: keep the accumulators alive without storing them
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc += acc[i][j][e];
    if (sacc == 1.2345e-30f) a.out[0] = sacc;
    return;
  }
  if constexpr (NPROD == 3) {      // 2^-sx, 2^-sw: two exact multiplications (no product of scales is formed)
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = acc[i][j][e] * h2_ix * h2_iw;
    x3_epilogue<Cfg, true, OUT16, EP, true>(a, acc, s_out, s_red, rt, n0, wm, wn, lane, tid, false, sg.seg);
    return;
  }
  x3_epilogue<Cfg, true, OUT16, EP>(a, acc, s_out, s_red, rt, n0, wm, wn, lane, tid, neg);
}
