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

template <int NPL> struct PatchRow { static constexpr int BYTES = NPL == 1 ? 80 : (NPL == 2 ? 144 : 208); };      // 2: the two fp16 planes of conv_patch_x3_kernel<.., 3, ..> (35 KB per 128-row tile: three workgroups per CU instead of two)

// dynamic LDS bytes of a launch
static inline size_t patch_lds_bytes(int bm, int halo, int npl) { return (size_t)(bm + 2 * halo + 1) * (npl == 1 ? 80 : (npl == 2 ? 144 : 208)); }

#ifdef LMKD_STAMPS      // measurement builds only (tools/ab_build.sh stamps -DLMKD_STAMPS): s_memtime at the phases of every 64th workgroup
__device__ unsigned long long g_lmkd_stamps[4096 * 8];
#define STAMP(k) do { if (threadIdx.x == 0 && (blockIdx.x & 63) == 0 && (blockIdx.x >> 6) < 4096) g_lmkd_stamps[(blockIdx.x >> 6) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif


// Epilogue of the two-plane instances (compute mode 4; before: x3_epilogue<.., H2>): undo the two operands' scales - two exact
// multiplications by powers of two, no product of scales is formed -, store, fold max |out| of the tile's frame segment into a.amax_out,
// and leave either the forward's BatchNorm partial sums (sum, sum of squares) or - a.bnb_x given: the data gradient that feeds relu +
// train-mode BatchNorm backward, lmkd_conv2d_bwd_data_seg - the sums of that backward (sum g, sum g xhat; g = out where
// fma(x, scale, shift) > 0: bn_bwd_reduce_kernel's terms) in a.stat_partial.  A lane owns ONE channel per 32-column block, so either
// pair of sums is local to the lane over its 16 TM rows and meets its partner lane (+ 32) in one shuffle.  The output offsets of a plain
// same-size launch (nclass == 1) are computed; the parity-class scatter reads them from s_out
// (one LDS read, one wait, one branch per accumulator row was most of the 10 800 cycles this epilogue took on layer 1).
template <class Cfg>
__device__ __forceinline__ void patch_h2_epilogue(const ConvGemmArgs& a, f32x16 (&acc)[Cfg::TM][Cfg::TN], const int* s_out, float* s_red, int rt,
                                                  int n0, int wm, int wn, int lane, int tid, int seg, int row0, int M, float h2_ix, float h2_iw) {
  const int cl0 = wn * (Cfg::TN * 32) + (lane & 31);
  const bool bnb = a.bnb_x != nullptr;      // (uniform)
  float s1[Cfg::TN], s2[Cfg::TN];
  float amo = 0.f;
  // plain same-size launch: the tile's output rows are contiguous.  Buffer descriptors based at the tile and ending with its last row inside
  // the segment (byte offsets stay below BM x Co x 4 whatever the tensor's size): a lane's 16 rows of a 32-row block differ by a uniform
  // number of rows - one add per element, no 64-bit address - and the range check drops the rows behind the segment and the columns past Co.
  const bool plain = a.nclass == 1;
  const int rows = M - row0 < Cfg::BM ? M - row0 : Cfg::BM;
  const long tile_off = plain ? (long)row0 * a.Co : 0;
  const long tile_bytes = (long)rows * a.Co * 4;
  const __amdgpu_buffer_rsrc_t ro = x3_rsrc(a.out + tile_off, tile_bytes);
  const __amdgpu_buffer_rsrc_t rx = x3_rsrc((bnb ? a.bnb_x : a.out) + tile_off, tile_bytes);
  const int r_lane = wm * (Cfg::TM * 32) + 4 * (lane >> 5);      // acc_row(e, lane) = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    s1[j] = s2[j] = 0.f;
    const int col = n0 + cl0 + j * 32;
    const bool in = col < a.Co;
    float bmean = 0.f, bistd = 0.f, bsc = 0.f, bsh = 0.f;
    if (bnb && in) {
      const float* tab = a.bnb_stats + (long)seg * 5 * a.Co;
      bmean = tab[col]; bistd = tab[a.Co + col]; bsc = tab[2 * a.Co + col]; bsh = tab[3 * a.Co + col];
    }
    if (plain) {
      const unsigned v0 = in ? (unsigned)((r_lane * a.Co + col) * 4) : 0x80000000u;      // (+ a row offset < 2^31: still past every tile)
      int co4 = a.Co * 4;      // (opaque per tile: the persistent loop would otherwise hoist 32 scalar products and carry them through the K loop)
      asm volatile("" : "+s"(co4));
#pragma unroll
      for (int ih = 0; ih < 2 * Cfg::TM; ++ih) {      // eight rows at a time (16 would hold 64 registers of offsets, operands and results)
        const int i = ih >> 1, e0 = (ih & 1) * 8;
        // the row's offset is ADDED to the vector offset (one instruction): the descriptor ends with the tile's last row inside its segment, so
        // the range check drops the rows behind it; those rows' accumulators are exact zeros (their patch rows were), and so are the columns
        // past Co: the maximum and the sums need no validity test at all (before: a compare and a select in front of every store and three
        // more around the sums - ten vector instructions per element, now six)
        unsigned vo[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) vo[q] = v0 + (unsigned)((i * 32 + (q & 3) + 8 * ((e0 + q) >> 2)) * co4);
        float xv[8], prev[8];
        if (bnb) {
#pragma unroll
          for (int q = 0; q < 8; ++q) xv[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vo[q], 0, 0));
        }
        if (a.accum) {      // out += acc (the residual branch's gradient already sits in the output buffer): all previous values first
#pragma unroll
          for (int q = 0; q < 8; ++q) prev[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ro, vo[q], 0, 0));
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float v = acc[i][j][e0 + q] * h2_ix * h2_iw;
          if (a.accum) v = v + prev[q];
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro, vo[q], 0, 0);
          amo = fmaxf(amo, fabsf(v));
          // one form for both kinds of sums (g, t) = (masked gradient, xhat) or (v, v): selects on a uniform flag, no branch to merge
          const float g = bnb ? (fmaf(xv[q], bsc, bsh) > 0.f ? v : 0.f) : v;
          const float t = bnb ? (xv[q] - bmean) * bistd : v;
          s1[j] += g;
          s2[j] = fmaf(g, t, s2[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {      // parity classes of a stride-2 data gradient: the rows scatter, their offsets come from s_out
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ob = s_out[wm * (Cfg::TM * 32) + i * 32 + acc_row(e, lane)];
          float v = acc[i][j][e] * h2_ix * h2_iw;
          if (ob >= 0 && in) {
            if (a.accum) v = v + a.out[(long)ob + col];
            a.out[(long)ob + col] = v;
            amo = fmaxf(amo, fabsf(v));
            if (bnb) {
              const float xv = a.bnb_x[(long)ob + col];
              const float g = fmaf(xv, bsc, bsh) > 0.f ? v : 0.f;
              s1[j] += g;
              s2[j] = fmaf(g, (xv - bmean) * bistd, s2[j]);
            }
          }
          if (!bnb) { s1[j] += v; s2[j] = fmaf(v, v, s2[j]); }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (a.amax_out) amax_commit<Cfg::THREADS / 64>(a.amax_out + seg * LMKD_AMAX_SEG_WORDS, amo);      // (every wave of the workgroup reaches this point)
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
    const float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
    if (lane < 32) {
      s_red[(wm * Cfg::BN + cl0 + j * 32) * 2 + 0] = t1;
      s_red[(wm * Cfg::BN + cl0 + j * 32) * 2 + 1] = t2;
    }
  }
  __syncthreads();
  if (tid < Cfg::BN && n0 + tid < a.Co && a.stat_partial) {
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

// Round 5: the workgroup is PERSISTENT - it walks the tiles L = blockIdx.x, blockIdx.x + gridDim.x, ... of the launch's tile order
// (ConvGemmArgs::grid_total slots, xcd_decode; gridDim.x a multiple of 8, so a workgroup's tiles stay on its XCD's band).  s_memtime stamps
// (tools/stamp_bench.py) of the one-tile-per-workgroup form on layer 1 (64 -> 64 channels, 18 K-steps per tile): of a workgroup's 48 500
// cycles 9 700 went before the first MFMA (kernel arguments, the HBM latency of the first patch chunk, the tables) and 10 800 behind the
// last one (an LDS read, a branch and a store per accumulator row); the K loop itself kept the matrix pipe 80 % busy.  Now the first
// patch chunk of tile i + 1 is requested where a middle chunk would request its successor (it arrives during the last chunk's MFMAs
// and the epilogue), the epilogue's stores are in flight while the next tile's tables are written, and the output offsets of a plain
// same-size launch are computed, not read from LDS.  Tiles, fragment order and summation order are unchanged: results are bit-identical.
struct PatchTile {
  int rt, ct, cls, ntap, seg, ltile, ltiles, row0, mend;
  unsigned p_off0, p_ok;      // per lane: byte offset of its first patch row of chunk 0; which of its NI rows are real pixels
};

#define PSTAMP(k) do { if (it == 5) STAMP(k); } while (0)
template <class Cfg, int NPROD, bool PRE, int IO, int DBG = 0>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::MINW) void conv_patch_x3_kernel(ConvGemmArgs) {
  // The argument block (1.4 KB) is copied into LDS once per workgroup and read from there: a scalar load from the kernel-argument segment
  // takes a microsecond or so on this machine (the segment is host-visible memory), the compiler re-reads arguments wherever scalar
  // registers are short, and a tile's prologue and epilogue had a dozen of those reads in a row behind each other.
  __shared__ ConvGemmArgs s_args;
  {
    const unsigned* ksrc = (const unsigned*)__builtin_amdgcn_kernarg_segment_ptr();      // (a C cast: the builtin returns a constant-address-space pointer)
    unsigned* kdst = reinterpret_cast<unsigned*>(&s_args);
    for (int i = threadIdx.x; i < (int)(sizeof(ConvGemmArgs) / 4); i += Cfg::THREADS) kdst[i] = ksrc[i];
  }
  __syncthreads();
  // ... and its scalar part (everything in front of the tap tables) goes on into scalar registers, once per workgroup; the tap tables
  // are indexed by the tile's parity class and stay in LDS
  ConvGemmArgs a;
  {
    constexpr int NW = (int)(offsetof(ConvGemmArgs, ntap) / 4);
    unsigned* aw = reinterpret_cast<unsigned*>(&a);
    const unsigned* lw = reinterpret_cast<const unsigned*>(&s_args);
#pragma unroll
    for (int i = 0; i < NW; ++i) aw[i] = __builtin_amdgcn_readfirstlane(lw[i]);
  }
  constexpr bool IN16 = (IO & 1) != 0, OUT16 = (IO & 2) != 0, EP = (IO & 4) != 0;      // IO bits: conv_gemm_x3_kernel
  constexpr int NPL = NPROD == 1 ? 1 : 3;
  constexpr int NPU = NPROD == 3 ? 2 : NPL;      // NPROD == 3: two fp16 planes (h2.h)
  static_assert(((IO & 3) == 0 || NPROD == 1) && !(IN16 && PRE), "bf16 tensors: one-plane mode, no load-side arithmetic");
  constexpr int ROWB = PatchRow<NPROD == 3 ? 2 : NPL>::BYTES;
  constexpr int ESZ = IN16 ? 2 : 4;
  // the two-plane instances walk several tiles; the others are launched with one workgroup per tile and leave the loop after the first
  // (their tiles - up to 256 rows, 128 accumulator registers per lane - have no room for what a second iteration keeps alive:
  // 256 registers + 700 bytes of scratch when they were given the loop, bf16 tensors 77 -> 72 episodes/s)
  constexpr bool PERSIST = NPROD == 3;
  constexpr int LPR = IN16 ? 4 : 8;                       // lanes per patch row (16 bytes each)
  constexpr int RPP = Cfg::THREADS / LPR;                 // patch rows per pass
  constexpr int NI = (Cfg::BM + 2 * PATCH_HALO_MAX + RPP - 1) / RPP;
  using LB = X3FragB<Cfg::TN, NPU>;
  extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
  __shared__ int s_out[Cfg::BM];
  __shared__ int s_tap_shift[LMKD_MAX_TAPS], s_tap_kofs[LMKD_MAX_TAPS], s_tap_dhw[LMKD_MAX_TAPS];
  // taps within +-1 pixel (ConvGemmArgs::taps_pm1: every 3x3 / 1x1 launch): the validity bits of a row's taps come out of a 64-entry table indexed by
  // (which of the rows h-1, h, h+1 exist) x (which of the columns w-1, w, w+1 exist) - one LDS read per row instead of two compares per (row, tap)
  __shared__ unsigned short s_mask_lut[64];
  __shared__ float s_red[Cfg::WM * Cfg::BN * 2];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int h = lane >> 5;
  const int a_row = wm * (Cfg::TM * 32) + (lane & 31);
  const int halo = a.halo, P = Cfg::BM + 2 * halo;
  const unsigned zero_addr = (unsigned)(P * ROWB + 16 * h);
  const __amdgpu_buffer_rsrc_t prs = x3_rsrc(a.src, (long)a.N * a.Hs * a.Ws * a.Cs * ESZ);
  const int pk = ((tid & (LPR - 1))) * (IN16 ? 8 : 4);            // first channel of this lane inside the 32-channel chunk
  const unsigned p_step = (unsigned)(RPP * a.Cs * ESZ);
  const long src_pixels = (long)a.N * a.Hs * a.Ws;
  // tile L of the launch's order: false for the padding slots of the XCD row bands and for a class without taps of an accumulating launch
  // (out += 0).  Parity classes of a stride-2 data gradient (conv.hip): each class is a same-size convolution over the dy grid with its
  // own 1-4 taps, its outputs scattered to the pixels (2a + ph, 2b + pw); a.nclass == 1: plain same-size convolution (omul 1, ph = pw = 0)
  auto decode = [&](int L, PatchTile& d) -> bool {
    if (!xcd_decode(L, a.n_rt, a.n_ct, a.xcd_mode, d.rt, d.ct)) return false;
    const int tile = d.rt / a.nclass;
    d.cls = d.rt - tile * a.nclass;
    d.ntap = __builtin_amdgcn_readfirstlane(s_args.ntap[d.cls]);
    if (a.accum && d.ntap == 0) return false;
    const SegTile sg = seg_tile(a, tile, Cfg::BM);      // the tile's frame segment (ConvGemmArgs::seg_m0)
    d.seg = sg.seg; d.ltile = sg.ltile; d.ltiles = sg.ltiles; d.row0 = sg.row0; d.mend = sg.mend;
    // byte offset of (patch row j = tid / LPR + RPP * i, channel pk) of chunk 0 is p_off0 + i * p_step where bit i of p_ok is set
    // (row inside the patch and the tensor), else the load is sent out of range (returns zeros)
    const long pix0 = (long)sg.row0 - halo + tid / LPR;
    d.p_off0 = (unsigned)((pix0 * a.Cs + pk) * ESZ);
    d.p_ok = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const long pix = pix0 + RPP * i;
      if (tid / LPR + RPP * i < P && pix >= 0 && pix < src_pixels) d.p_ok |= 1u << i;
    }
    return true;
  };
  auto next_tile = [&](int L, PatchTile& d) -> int {      // the first tile at or behind slot L of this workgroup's walk; -1: none
    for (; L < a.grid_total; L += a.grid_step)
      if (decode(L, d)) return L;
    return -1;
  };
  PatchTile cur, nxt;
  int Lc = next_tile((int)blockIdx.x, cur);
  if (Lc < 0) return;
  u32x4 rp[NI];
  float4 psc = float4(), psh = float4();
  auto issue_patch = [&](const PatchTile& d, int cc) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (i * RPP < P) rp[i] = __builtin_amdgcn_raw_buffer_load_b128(prs, ((d.p_ok >> i) & 1u) ? d.p_off0 + i * p_step + (unsigned)(cc * 32 * ESZ) : X3_OOB, 0, 0);
    if (PRE) {
      const float* pre_tab = a.pre_stats + (long)d.seg * 5 * a.Cs;      // the [5][Cs] BatchNorm table of the tile's segment
      psc = *reinterpret_cast<const float4*>(pre_tab + 2 * a.Cs + cc * 32 + pk);
      psh = *reinterpret_cast<const float4*>(pre_tab + 3 * a.Cs + cc * 32 + pk);
    }
  };
  for (int j = tid; j < ROWB / 4; j += Cfg::THREADS) reinterpret_cast<unsigned*>(psm + (long)P * ROWB)[j] = 0u;   // the zero row
  int it = 0;      // tiles this workgroup has done (measurement builds: the stamps are those of its sixth tile)
  bool requested = false;      // the first chunk of `cur` is on its way (requested during the previous tile)
  for (;;) {
    PSTAMP(0);
    const int Ln = PERSIST ? next_tile(Lc + a.grid_step, nxt) : -1;
    const int ntap = cur.ntap;
    const Tap* taps = s_args.taps[cur.cls];
    const int ph = cur.cls >> 1, pw = cur.cls & 1;
    const int M = cur.mend, row0 = cur.row0, n0 = cur.ct * Cfg::BN;
    const int nk = ntap * a.cps;
    // Kernel arguments are SLOW to read one after the other (the tap table sits 1 KB into a 1.4 KB argument block; the per-row mask
    // loop used to make nine dependent scalar loads): every wave fetches the taps with ONE vector load, lane = tap
    const Tap my_tap = taps[lane < ntap ? lane : 0];
    if (!requested && nk > 0) issue_patch(cur, 0);
    requested = false;
    unsigned h2_xbits = 0u, h2_wbits = 0u;      // (loads in flight; reduced below)
    if constexpr (NPROD == 3) {
      h2_xbits = a.h2_xw[cur.seg * LMKD_AMAX_SEG_WORDS + (tid & (LMKD_AMAX_SLOTS - 1)) * LMKD_AMAX_STRIDE];
      h2_wbits = *reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(a.wpk) + (long)a.Co * a.Kp * 16);
    }
    LB lb;
    const bool neg = NPL == 3 && NPROD != 3 && x3_neg_tile(cur.ltile, cur.ltiles);      // half the row tiles (of the segment) accumulate -y: X3FragB::init
    if constexpr (NPROD == 3)      // the fp16 planes of W in the 32x32x16 order: behind the bf16 planes, the 16x16x32 fp16 planes and the maximum's 64 bytes
      lb.init(reinterpret_cast<const unsigned short*>(a.wpk) + (long)a.Co * a.Kp * 16 + 32, a.Co, a.Kp, n0 + wn * (Cfg::TN * 32), lane, false);
    else
      lb.init(a.wpk, a.Co, a.Kp, n0 + wn * (Cfg::TN * 32), lane, neg);
    u32x4 rb0[LB::NR], rb1[LB::NR];
    int b_tp = 0, b_cc = 0;      // (tap, chunk) of the next B fragments to fetch
    // the weight fragments of the first two K-steps go out before the tables are written: their K offsets come from the tap lanes
    if (nk > 0) {
      lb.load(__builtin_amdgcn_readlane(my_tap.kofs, 0), rb0);
      if (++b_tp == ntap) { b_tp = 0; ++b_cc; }
      if (nk > 1) {
        lb.load((ntap > 1 ? __builtin_amdgcn_readlane(my_tap.kofs, 1) : __builtin_amdgcn_readlane(my_tap.kofs, 0)) + b_cc * LMKD_BK, rb1);
        if (++b_tp == ntap) { b_tp = 0; ++b_cc; }
      }
    }
    // A plain launch (one class) of a persistent instance has the same taps on every tile: its tables are written once, and three of a tile's
    // seven barriers go - the one behind the tables, the one in front of the first chunk's patch store (every wave that gets there has passed
    // the previous tile's epilogue barrier, i.e. all waves are through that tile's K loop) and the one at the loop's end (what it protected -
    // s_red against the next tile's writers - is ordered by the next tile's chunk barriers: a writer has passed them, the readers read before
    // they arrive there).  PMC: 39 % of the layer-1 kernel's wave cycles were waits (profiles/r05_pmc_3x3_layers_x3_vs_h2.txt).
    const bool inv = PERSIST && a.nclass == 1;
    const bool tables = !inv || it == 0;
    if (tables && tid < ntap) {
      s_tap_shift[tid] = (my_tap.dh * a.Ws + my_tap.dw) * ROWB;
      s_tap_kofs[tid] = my_tap.kofs;
      s_tap_dhw[tid] = (my_tap.dh << 16) | (my_tap.dw & 0xffff);
    }
    if (tables && NPROD == 3 && a.taps_pm1 && tid < 64) {      // (two-plane instances; reads the argument block's LDS copy: needs nothing of this tile's tables)
      const int rb = tid & 7, cb = tid >> 3;
      unsigned m = 0;
      for (int tp = 0; tp < ntap; ++tp) {
        const int dh = taps[tp].dh, dw = taps[tp].dw;
        if (((rb >> (dh + 1)) & 1) && ((cb >> (dw + 1)) & 1)) m |= 1u << tp;
      }
      s_mask_lut[tid] = (unsigned short)m;
    }
    if (a.nclass != 1 || !(NPROD == 3)) {      // (the two-plane epilogue of a plain same-size launch computes its output offsets)
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
    }
    if (tables) __syncthreads();      // tap tables, s_out, zero row
    // per MFMA row of this lane: LDS byte address of its own pixel's row (tap shift 0) and one validity bit per tap
    unsigned a_base[Cfg::TM], a_mask[Cfg::TM];
    {
      int hh[Cfg::TM], ww[Cfg::TM];
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) {
        const int r = a_row + 32 * i, m = row0 + r;
        a_base[i] = (unsigned)((r + halo) * ROWB + 16 * h);
        a_mask[i] = 0;
        hh[i] = -0x4000; ww[i] = 0;      // rows past the tile's segment: every tap reads the zero row
        if (m < M) {
          const int n = fdiv(m, a.div_hw);
          const int rem = m - n * a.Hs * a.Ws;
          hh[i] = fdiv(rem, a.div_w);
          ww[i] = rem - hh[i] * a.Ws;
        }
      }
      if (NPROD == 3 && a.taps_pm1) {
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i) {
          const unsigned Hs = (unsigned)a.Hs, Ws = (unsigned)a.Ws;
          const unsigned rbits = ((unsigned)(hh[i] - 1) < Hs ? 1u : 0u) | ((unsigned)hh[i] < Hs ? 2u : 0u) | ((unsigned)(hh[i] + 1) < Hs ? 4u : 0u);
          const unsigned cbits = ((unsigned)(ww[i] - 1) < Ws ? 1u : 0u) | ((unsigned)ww[i] < Ws ? 2u : 0u) | ((unsigned)(ww[i] + 1) < Ws ? 4u : 0u);
          a_mask[i] = s_mask_lut[rbits | (cbits << 3)];
        }
      } else
      for (int t0 = 0; t0 < ntap; t0 += 9) {      // nine taps per trip: their LDS reads in flight together
        int dv[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) dv[q] = s_tap_dhw[t0 + q < ntap ? t0 + q : 0];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          const int dh = dv[q] >> 16, dw = (int)(short)(dv[q] & 0xffff);
#pragma unroll
          for (int i = 0; i < Cfg::TM; ++i) {
            const int y = hh[i] + dh, x = ww[i] + dw;
            if (t0 + q < ntap && (unsigned)y < (unsigned)a.Hs && (unsigned)x < (unsigned)a.Ws) a_mask[i] |= 1u << (t0 + q);
          }
        }
      }
    }
    float h2_sx = 1.f, h2_ix = 1.f, h2_iw = 1.f;
    if constexpr (NPROD == 3) {
      unsigned v = h2_xbits;      // amax_read: the largest of the segment's 64 slots
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        const unsigned u = (unsigned)__shfl_xor((int)v, o, 64);
        v = u > v ? u : v;
      }
      h2_sx = h2_scale(__builtin_amdgcn_readfirstlane(v));
      h2_ix = 1.f / h2_sx;
      h2_iw = 1.f / h2_scale(h2_wbits);
    }
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
          if (PRE && ((cur.p_ok >> i) & 1u)) {      // relu(BatchNorm(raw)), bit-identical to bn_apply_kernel; padding never reaches here
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
    f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto issue_b = [&](u32x4 (&rb)[LB::NR]) {
      lb.load(s_tap_kofs[b_tp] + b_cc * LMKD_BK, rb);
      if (++b_tp == ntap) { b_tp = 0; ++b_cc; }
    };
    int k_tp = 0, k_cc = 0;      // (tap, chunk) of the current K-step
    // One-plane modes (round 4): the A fragments of K-step t + 1 are read from LDS while the MFMAs of step t run (two register sets,
    // passed to step() in alternation like the weight sets).  With one plane a K-step is 4 MFMAs (128 cycles) per wave behind 4
    // ds_read_b128 whose latency (+ bank conflicts) is as long: read-then-multiply left the matrix pipe 31 % busy (PMC, tools/pmc_patch.sh)
    // with every tile shape.  Across a chunk boundary nothing can be prefetched (the next chunk's patch is not in LDS yet).
    constexpr bool PIPE = NPL == 1;
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
    // is then one MFMA phase old - and `rb` is refilled with step t+2.  At a chunk boundary the patch is replaced first, and the next
    // one requested: the tile's next chunk, or - behind its last chunk - the first chunk of the workgroup's NEXT tile.
    auto step = [&](int t, u32x4 (&rb)[LB::NR], u32x4 (&rbn)[LB::NR], bf16x8 (&avc)[2][NPL][Cfg::TM], bf16x8 (&avn)[2][NPL][Cfg::TM]) {
      if (k_tp == 0) {
        if (!(inv && k_cc == 0)) __syncthreads();      // every wave has finished reading the previous chunk
        store_patch();
        __syncthreads();
        {      // ONE set of loads with selected operands (two calls under an if / else cost a second register set for the merge)
          const bool more = k_cc + 1 < a.cps;
          const bool ahead = PERSIST && !more && Ln >= 0 && nxt.ntap > 0;
          if (more || ahead) {
            PatchTile d;
            d.p_off0 = more ? cur.p_off0 : nxt.p_off0;
            d.p_ok = more ? cur.p_ok : nxt.p_ok;
            d.seg = more ? cur.seg : nxt.seg;
            issue_patch(d, more ? k_cc + 1 : 0);
            requested = ahead;
          }
        }
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
      bf16x8 av[2][NPL][Cfg::TM];
      auto read_a = [&](int g) {
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
    PSTAMP(1);
    if (nk > 0) {
      x3_landed(rp);
      x3_landed(rb0);
      x3_landed(rb1);
      __builtin_amdgcn_sched_barrier(0);
      PSTAMP(2);
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
    PSTAMP(3);
    if constexpr (NPROD == 3) {
      patch_h2_epilogue<Cfg>(a, acc, s_out, s_red, cur.rt, n0, wm, wn, lane, tid, cur.seg, row0, M, h2_ix, h2_iw);
    } else {
      x3_epilogue<Cfg, true, OUT16, EP>(a, acc, s_out, s_red, cur.rt, n0, wm, wn, lane, tid, neg);
    }
    PSTAMP(4);
    if (!PERSIST || Ln < 0) break;
    if (!inv) __syncthreads();      // the tables and s_red are this tile's until every wave is through its epilogue
    cur = nxt;
    Lc = Ln;
    ++it;
  }
}
