/* lmkd.h — C ABI of liblmkd_hip.so, the MI355X (gfx950) implementation of Lite-MKD's
 * per-episode hot path.
 *
 * The reference (HuiGuanLab/Lite-MKD) is 100 % Python on torch/cuDNN: it has no FFI of its own.
 * Each entry point below therefore names the reference *call site* whose ATen/cuDNN work it
 * replaces (file:line under the reference tree).  The Python host layer in lite-mkd_amd/ binds
 * these with ctypes and exposes the reference's model_select / classifier / Distiller surface.
 *
 * Conventions
 *   - plain pointers + sizes, no C++ or torch types; every pointer is a DEVICE pointer unless
 *     marked "host"; tensors are fp32, activations NHWC, all buffers borrowed for the call only;
 *   - `stream` is a hipStream_t; kernels are enqueued on it and the call returns immediately;
 *   - return 0 on success, negative on error; lmkd_last_error() gives the message (thread local);
 *     nothing throws; no entry point allocates, frees or synchronises (hipGraph-capturable).
 */
#ifndef LMKD_H
#define LMKD_H
#ifdef __cplusplus
extern "C" {
#endif

int lmkd_abi_version(void);
const char* lmkd_last_error(void);
int lmkd_device_check(int device); /* 0 iff `device` is gfx950 */
/* storage type of the trunk's activation and activation-gradient tensors in HBM: 0 = fp32 (default), 1 = bf16 (BASELINE configs[2]:
   the reference trains under autocast, trainwandb.py:20,126).  In mode 1 the activation pointers (`float*` below) of the
   convolution, BatchNorm and pooling entry points address bf16 tensors; statistics, weights, weight gradients, workspaces and
   everything after the pooled head stay fp32.  Needs lmkd_conv_set_compute_dtype(1). */
int lmkd_set_activation_dtype(int mode);
int lmkd_get_activation_dtype(void);

/* ---- generic fp32 MFMA GEMM: C = alpha*op(A)*op(B) + beta*C + bias[col] (+relu), strided batched.
 * layA 'K': A[m*lda+k], 'M': A[k*lda+m];  layB 'K': B[n*ldb+k], 'N': B[k*ldb+n].
 * replaces nn.Linear fwd/bwd at model/backbone/resnet18_2fc.py:56-64, model/classifiers/TRX_2fcsup.py:97-100,
 * torch.matmul at TRX_2fcsup.py:121,133 and the nn.TransformerEncoder GEMMs at teacher/code/model.py:1310-1323. */
int lmkd_gemm_f32(char layA, char layB, int M, int N, int K, float alpha, const float* A, long lda, long sA,
                  const float* B, long ldb, long sB, float beta, float* C, long ldc, long sC, const float* bias,
                  int relu, int batch, void* stream);
/* the same with split-K for launches of at most 64 output tiles (the fc layers' 200-row calls): given a workspace the K range is split
   over more workgroups and the last-arriving block of each output tile adds the partial tiles in split order (deterministic).  workspace: any size (the split count adapts), tickets: lmkd_gemm_ticket_words() zeroed words (left zeroed); both
   private to the stream while its launches may be in flight. */
/* arithmetic of lmkd_gemm_f32: -1 (default) = that of the convolutions (lmkd_conv_set_compute_dtype 0: native fp32 MFMA; 1-3: every fp32
   product from the exact 3-way bf16 split of both operands on the bf16 matrix pipe, six products, fp32-class error - csrc/gemm_x3.h),
   0 = always native fp32 MFMA, 1 = always 3 x bf16, 2 = one RNE bf16 plane per operand with fp32 accumulation (the reference's
   autocast arithmetic for nn.Linear / matmul, trainwandb.py:20,126).  Process-wide, like lmkd_conv_set_compute_dtype. */
int lmkd_gemm_set_mode(int mode);
int lmkd_gemm_set_tile(int t); /* tuning (modes 1 / 2): 0 auto (64 x 64 tiles), 1 = 128 x 128, 2 = 64 x 128 */
long lmkd_gemm_ticket_words(void);
int lmkd_gemm_f32_splitk(char layA, char layB, int M, int N, int K, float alpha, const float* A, long lda, long sA,
                         const float* B, long ldb, long sB, float beta, float* C, long ldc, long sC, const float* bias,
                         int relu, int batch, void* workspace, long ws_bytes, unsigned* tickets, void* stream);

/* ---- convolution (torchvision ResNet conv layers called at resnet18_2fc.py:41-42) ----
 * x,y NHWC.  Cs = channels of the NHWC tensor (4 for the channel-padded stem, else multiple of 32). */
long lmkd_conv2d_packed_weight_elems(int Cout, int Cin, int Cs, int KH, int KW, int mode);
int lmkd_conv2d_pack_weights(const float* w_oihw, float* wp, int Cout, int Cin, int Cs, int KH, int KW, int mode /*0 fwd, 1 dgrad*/,
                             void* stream);
/* rows T of the [T][Cout][2] BatchNorm partial-sum buffer a forward launch fills (one row per row tile), for an input with Cs % 32 == 0;
   evaluate it in the arithmetic mode the launch will run in */
int lmkd_conv2d_fwd_row_tiles(int N, int H, int W, int Cout, int KH, int KW, int stride, int pad);
/* the same for any input channel count: REQUIRED for the padded stem input (Cs = 4), whose launch uses a tiling of its own in the
   bf16-plane modes (one row per pair of output rows) */
int lmkd_conv2d_fwd_row_tiles_cs(int N, int H, int W, int Cs, int Cout, int KH, int KW, int stride, int pad);
int lmkd_set_elementwise_wg_per_cu(int n); /* tuning: grid cap of the HBM-bound kernels, workgroups per CU (default 4) */
/* arithmetic of the convolutions (process-wide setting, not thread safe: set it before the first launch, one process per GPU):
   2 (DEFAULT) = fp32 operands split exactly into three bf16 planes, six bf16 MFMA products per fp32 product, fp32
   accumulation (csrc/conv_x3.h; fp32-class error, the arithmetic of the benchmark's headline line), 3 = the same with all nine
   products, 0 = native fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = bf16 MFMA inputs + fp32 accumulation (BASELINE configs[2]).
   In modes 1/2/3 the packed-weight arguments of lmkd_conv2d_fwd / lmkd_conv2d_bwd_data point at the buffer written by
   lmkd_conv2d_split_weights (in that mode).
   4 (round 4) = mode 2 with the trunk's convolution kernels - the 3x3 forward and data gradient (conv_patch16_x3_kernel, also the
   1x1 / stride-2 data gradient), the weight gradients (conv_wgrad_win16_kernel, conv_wgrad_x3_kernel, stem_wgrad_kernel) and the stem's
   forward (conv_stem_patch_kernel); not the 1x1 forward and the head GEMMs - on TWO fp16 planes and THREE
   v_mfma_f32_16x16x32_f16 products per fp32 product: x 2^s = h0 + h1 with h0 = fp16(x 2^s), h1 = fp16(x 2^s - h0) (round to nearest)
   represents x to 2^-23 (one fp32 ulp; exactly for at least half of all fp32 values; zero-mean), the dropped h1 h1' term is <= 2^-22
   of the product with zero mean (mode 2 drops <= 2^-23, all of one sign: its planes are truncations), accumulation stays fp32, and 2^s
   is a power of two taken from the tensor's maximum (lmkd_amax_desc::x_words / dy_words), so the scaling is exact.  Measured error against fp64
   at or below mode 2's on every layer (profiles/r04_h2_error.txt).  A launch whose operand maxima are not named runs mode 2. */
int lmkd_conv_set_compute_dtype(int mode);
/* mode 4: the range bookkeeping of ONE launch, an explicit argument of the *_seg entry points, lmkd_bn_finalize and
   lmkd_stem_unpool_bn_bwd (round 5; before: four one-shot "the next launch of this host thread" channels).  Every member is nullable and a
   null struct pointer means "none"; the other modes ignore it.  A "maximum" is lmkd_amax_words() device words: two frame segments x 64
   slots, 64 bytes apart (float atomics execute at the memory side: thousands of them on one address cost more than the pass itself);
   word 0 of a slot holds fp32 bits folded with an atomic max - the maximum of a segment is the largest of its slots, and any upper bound
   is a valid maximum - word 1 and words 2-3 the range statistics described under ref_words.  The caller zeroes the words before the
   launch that writes them; they must be complete in stream order before a launch that reads them.
     x_words / dy_words  convolutions: the maxima of the gathered operand (forward: x; data gradient: dy; weight gradient: both).  A launch
                         whose operand maxima are not named runs mode 2.  With pre_stats: the BOUND written by lmkd_bn_finalize(_seg).
                         lmkd_bn_finalize(_seg): x_words = max |x| recorded by the convolution that wrote the BatchNorm's input.
     out_words           producers (lmkd_bn_apply_seg, lmkd_bn_relu_maxpool_fwd_seg, lmkd_bn_backward_seg, lmkd_bn_backward_part_seg,
                         lmkd_stem_unpool_bn_bwd(_seg)): also fold max |result| into these words - this is how a trunk tensor gets its
                         maximum without a pass of its own.  lmkd_conv2d_fwd_seg: also fold max |y| (in the epilogue of
                         conv_patch16_x3_kernel, or by a reduction pass behind any other kernel).  lmkd_bn_finalize(_seg): turn x_words and
                         the scale / shift tables into an upper bound of max |relu(BatchNorm(x))| per segment - the x_words of a consumer
                         that applies the BatchNorm + ReLU in its loader (lmkd_conv2d_fwd_seg / lmkd_conv2d_bwd_weight_seg with pre_stats).
     ref_words           producers, with out_words: words holding the maximum the same tensor had the last time it was written (previous
                         episode; lmkd_h2_fence_eval maintains them).  The
                         launch then also counts, per segment, n_small = nonzero elements below 2^-17 of that maximum (the two-plane split
                         resolves those to an absolute 2^-40 of the maximum instead of a relative 2^-23) and Q = sum of min(|x| 2^17 /
                         maximum, 1024) - integers, independent of the order of the waves.  lmkd_h2_fence_eval judges them (the range fence).
     flags               LMKD_AMAX_FENCED: the caller withheld a maximum from this convolution because the fence had flagged its tensor
                         (counted by lmkd_conv_h2_fallbacks). */
typedef struct lmkd_amax_desc {
  const void* x_words;
  const void* dy_words;
  void* out_words;
  const void* ref_words;
  int flags;
} lmkd_amax_desc;
#define LMKD_AMAX_FENCED 1
/* the range fence: entry e of the HOST arrays words / refs = the words a producer has just written and the ref_words it was given - the
   PERSISTENT reference words of that tensor's site (caller-owned, lmkd_amax_words() words, zero before the first episode).  flags[e]
   (device ints) bit s is set when segment s has more than a quarter as much rounding-error mass in under-resolved elements (4 n_small > Q)
   as in resolved ones - judged only where the reference was adequate (current maximum >= reference / 4).  Then refs[e] is overwritten
   with this launch's maxima: the reference of the next episode.  The caller stops naming a flagged tensor's maximum: its convolutions
   run the three-plane form (mode 2), which has no range to speak of. */
int lmkd_h2_fence_eval(const void* const* words, void* const* refs, int n, int* flags, void* stream);
long lmkd_conv_h2_fallbacks(void); /* mode-4 launches that carried LMKD_AMAX_FENCED so far */
/* max |x[0 .. n)| -> the slots of ONE frame segment at `word` (lmkd_amax_words() / 2 words, zeroed here first; segment 1 of a tensor's
   maximum starts lmkd_amax_words() / 2 words in): a pass of its own, for tensors no kernel of this library wrote */
int lmkd_amax(const float* x, long n, void* word, void* stream);
long lmkd_amax_words(void);
/* 16-bit elements of the plane buffer lmkd_conv2d_split_weights fills for a packed weight [ncols][Kp] in the current mode:
   ncols Kp (mode 1), 12 ncols Kp (2 / 3), 18 ncols Kp + 32 (4: + the two fp16 planes of W and of -W in the 16x16x32 order, 64 bytes
   holding max |w|, and the two fp16 planes of W in the 32x32x16 order) */
long lmkd_conv2d_plane_elems(int ncols, int Kp);
long lmkd_conv_h2_launches(void); /* launches that took the two-plane form so far (tests) */
/* re-pack n convolution weights in ONE launch (after an optimizer step; trainwandb.py:142): entry i = OIHW weight ws[i] -> wfs[i], the
   buffer lmkd_conv2d_pack_weights + lmkd_conv2d_split_weights would fill for dims[6 i ..] = (Cout, Cin, Cs, KH, KW, mode), bit for bit.
   Modes 1-4 of lmkd_conv_set_compute_dtype.  The pointer / dims arrays are host memory. */
int lmkd_conv2d_repack_multi(const float* const* ws, void* const* wfs, const int* dims, int n, void* stream);
/* wp: fp32 K-major packed weights [ncols][Kp] (lmkd_conv2d_pack_weights; ncols = Cout forward, Cin data gradient)
   -> wf: bf16 in MFMA fragment order: mode 1: ncols * Kp (one round-to-nearest plane); modes 2/3: 12 * ncols * Kp = the three planes
   of W and the three planes of -W (half the row tiles accumulate -y, csrc/conv_x3.h X3FragB) in the v_mfma 32x32x16 fragment order,
   then the same six planes in the 16x16x32 order (csrc/conv_patch16.h) */
int lmkd_conv2d_split_weights(const float* wp, void* wf, int ncols, int Kp, void* stream);
int lmkd_conv_get_compute_dtype(void);
int lmkd_conv_set_wgrad_planes(int on); /* tuning (modes 1-3): 1 = weight gradient on the bf16-plane kernel with transposed LDS reads (default), 0 = fp32-tile kernel */
int lmkd_conv_set_patch(int on); /* tuning (modes 1-3): 1 = same-size convolutions (3x3 / stride 1 forward and data gradient) read an LDS-resident input patch (default), 0 = im2col gather */
int lmkd_conv_set_stem_patch(int on); /* tuning (modes 1-3): 1 = the 7x7 / stride-2 stem convolution reads an LDS-resident patch of input rows (default), 0 = im2col gather */
int lmkd_conv_set_s2_patch(int on); /* tuning (modes 2-3, fp32 tensors): 1 = stride-2 3x3 forward convolutions on the LDS-patch kernel, as four same-size convolutions over the input's parity classes (default), 0 = im2col gather */
int lmkd_conv_set_wgrad_win16(int on); /* tuning (modes 2-3, fp32 tensors): 1 = the rolling-window weight gradient on v_mfma_f32_16x16x32_bf16 (default), 0 = 32x32x16 */
int lmkd_conv_set_wgrad_stem(int on); /* tuning (modes 2-3, fp32 tensors): 1 = the stem's weight gradient on stem_wgrad_kernel (input rows resident in LDS, no im2col copy; default), 0 = im2col-gather kernel */
int lmkd_conv_set_wgrad_window(int on); /* tuning (modes 1-3): 1 = 3x3 / stride-1 weight gradients read a rolling LDS window of x, all nine taps per workgroup (default), 0 = im2col-gather kernel */
int lmkd_conv_set_patch16(int on); /* tuning (modes 2 - 4, fp32 tensors): which MFMA the 4-wave patch tiles run on: 1 (default) = v_mfma_f32_16x16x32_* (conv_patch16_x3_kernel), except mode 4's two-plane launches on the 64-column tile (Cout <= 64), which run conv_patch_x3_kernel on v_mfma_f32_32x32x16_f16; 2 = 16x16x32 everywhere; 0 = 32x32x16 everywhere */
int lmkd_conv_set_persistent(int on); /* tuning (LDS-patch kernels): 1 (default) = persistent workgroups - no more than the chip holds at once, each walking its share of the tile order and requesting its next tile's first patch chunk during the current tile's last; 0 = one workgroup per tile (same tiles, bit-identical results) */
int lmkd_conv_set_xcd_mode(int mode); /* tuning: -1 auto (XCD-aware tile order + XCD-grouped weight-gradient splits), 0 plain orders, 1 auto without the weight-gradient grouping */
int lmkd_conv_set_tile(int id); /* tuning: 0 auto, 1 128x128, 2 128x64, 3 64x64, 4 64x128 */
/* stat_partial (nullable): [row_tiles][Cout][2] per-tile (sum, sum of squares) for train-mode BatchNorm */
int lmkd_conv2d_fwd(const float* x, const float* wp_fwd, float* y, float* stat_partial, int N, int H, int W, int Cs, int Cout,
                    int KH, int KW, int stride, int pad, void* stream);
/* training forward of a convolution fed by relu(BatchNorm(x_raw)) (torchvision BasicBlock / Bottleneck inner convs,
   resnet18_2fc.py:41-42): x_raw = raw output of the previous convolution, pre_stats = its [5][Cs] table from lmkd_bn_finalize;
   the loader normalises + rectifies while filling LDS (bit-identical to lmkd_bn_apply first), fp32 mode, Cs % 32 == 0 */
int lmkd_conv2d_fwd_pre(const float* x_raw, const float* pre_stats, const float* wp_fwd, float* y, float* stat_partial, int N, int H,
                        int W, int Cs, int Cout, int KH, int KW, int stride, int pad, void* stream);
/* inference (module.eval(), trainwandb.py:366 / test.py): y = relu?(conv(x)*scale[c] + shift[c] (+ res)) in ONE kernel - the eval-mode
   BatchNorm (bn_stats = the [5][C] table of lmkd_bn_eval_stats), the residual add and the ReLU of torchvision's BasicBlock /
   Bottleneck run in the convolution's epilogue.  Bit-identical to lmkd_conv2d_fwd followed by lmkd_bn_apply. */
int lmkd_conv2d_fwd_bn(const float* x, const float* wp_fwd, float* y, const float* bn_stats, const float* res /*nullable*/, int relu,
                       int N, int H, int W, int Cs, int Cout, int KH, int KW, int stride, int pad, void* stream,
                       const lmkd_amax_desc* amax /*nullable; mode 4: x_words -> two planes, out_words <- max |y| after the epilogue*/);
/* accumulate != 0: dx += ... (the block's residual-branch gradient is already in dx) */
int lmkd_conv2d_bwd_data(const float* dy, const float* wp_dgrad, float* dx, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                         int stride, int pad, int accumulate, void* stream);
/* the data gradient that feeds relu + train-mode BatchNorm backward (torchvision BasicBlock: d relu(bn1(c1)) = dgrad(conv2);
   resnet18_2fc.py:41-42 runs that block): besides dx the kernel's epilogue leaves the per-row-tile sums (sum g, sum g * xhat) of
   lmkd_bn_backward(mask_mode 2) over (dx, bn_x) in part[tiles][Cin][2], so lmkd_bn_backward_part needs no reduction pass.
   bn_x: the BatchNorm's input, shaped like dx; bn_stats: its [5][Cin] table.  lmkd_conv2d_bwd_data_bn_tiles: rows of `part`, or 0 when
   the launch has no such form in the current arithmetic (then run lmkd_conv2d_bwd_data + lmkd_bn_backward). */
int lmkd_conv2d_bwd_data_bn_tiles(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int lmkd_conv2d_bwd_data_bn(const float* dy, const float* wp_dgrad, float* dx, const float* bn_x, const float* bn_stats, float* part,
                            int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream);
long lmkd_conv2d_bwd_weight_workspace(int N, int H, int W, int Cs, int Cout, int KH, int KW, int stride, int pad);
int lmkd_conv2d_bwd_weight(const float* x, const float* dy, float* dw_oihw, float* workspace, long ws_bytes, int N, int H, int W,
                           int Cs, int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream);
/* the same for a convolution run with lmkd_conv2d_fwd_pre: relu(BatchNorm(x_raw)) is recomputed in the loader */
int lmkd_conv2d_bwd_weight_pre(const float* x_raw, const float* pre_stats, const float* dy, float* dw_oihw, float* workspace,
                               long ws_bytes, int N, int H, int W, int Cs, int Cin, int Cout, int KH, int KW, int stride, int pad,
                               void* stream);

/* launch plan (no launch) of lmkd_conv2d_fwd (kind 0) / _bwd_data (1) / _bwd_weight (2) for these shapes in the current mode:
   info[5] (HOST) = tile id, XCD tile order, parity classes | pixel splits, workgroups, 1 if the LDS-patch / rolling-window kernel runs it (2: the stem's input-row patch kernel).
   Test/diagnostic aid. */
int lmkd_conv2d_plan(int kind, int N, int H, int W, int Cs, int Cin, int Cout, int KH, int KW, int stride, int pad, int* info_host);

/* dw_oihw += weight gradient (dw_oihw = the parameter's .grad; accumulation over trunk calls / episodes, trainwandb.py:141-143);
   pre_stats nullable (as lmkd_conv2d_bwd_weight_pre) */
int lmkd_conv2d_bwd_weight_acc(const float* x, const float* pre_stats, const float* dy, float* dw_oihw, float* workspace, long ws_bytes,
                               int N, int H, int W, int Cs, int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream);

/* ---- layout / BatchNorm / pooling (torchvision bn1/relu/maxpool/BasicBlock, resnet18_2fc.py:33,41-54) ---- */
/* crop + horizontal flip + ToTensor of uint8 HWC frames into NHWC4 (video_reader.py:92-112 after Resize); crop/flip per video */
int lmkd_frames_u8_to_nhwc4(const unsigned char* src, float* dst, const int* crop_y, const int* crop_x, const int* flip, int F, int Hs,
                            int Ws, int H, int W, int frames_per_video, void* stream);
/* Resize of the frame transform (video_reader.py:96-101 -> videotransforms/functional.py:44-59 = PIL.Image.resize(BILINEAR)):
   Pillow's 8-bit triangle-filter resampler, bit exact.  lmkd_resize_plan fills the HOST tables of one axis (bounds [out][2],
   22-bit fixed-point coefficients [out][ksize]) and returns ksize (> 0; call with null tables to size them); the caller
   uploads them and runs lmkd_resize_pass_u8 over [outer][n_in][inner] uint8 -> [outer][n_out][inner], horizontal axis first. */
int lmkd_resize_plan(int in_size, int out_size, int* bounds_host, int* coeffs_host);
int lmkd_resize_pass_u8(const unsigned char* src, unsigned char* dst, const int* bounds_dev, const int* coeffs_dev, int ksize, long outer,
                        int n_in, int n_out, long inner, void* stream);
/* The same transform for frames that share one resolution, in ONE launch (round 5): a thread evaluates Pillow's two passes for its own
   pixel of the cropped output (uint8 rounding between the passes as in Pillow's intermediate image: bit-identical to
   lmkd_resize_pass_u8 x 2 + lmkd_frames_u8_to_nhwc4), nothing for the part of the resized frame the crop discards.  bounds_h / coeffs_h:
   lmkd_resize_plan(Ws, Wr) uploaded, bounds_v / coeffs_v: lmkd_resize_plan(Hs, Hr); crop_y / crop_x / flip as lmkd_frames_u8_to_nhwc4
   (coordinates in the Hr x Wr resized frame).  video_reader.py:92-112,377-385. */
int lmkd_frames_resize_crop_nhwc4(const unsigned char* src, float* dst, const int* bounds_h, const int* coeffs_h, int ksize_h,
                                  const int* bounds_v, const int* coeffs_v, int ksize_v, const int* crop_y, const int* crop_x,
                                  const int* flip, int F, int Hs, int Ws, int Hr, int Wr, int H, int W, int frames_per_video, void* stream);
/* amax_words (nullable; mode 4): also fold max |y| into the slots of ONE frame segment at this address (lmkd_amax_words() / 2 words) */
int lmkd_nchw3_to_nhwc4(const float* x_nchw, float* y_nhwc4, int N, int H, int W, void* stream, void* amax_words);
/* Ticket words: the single-launch column reductions (lmkd_bn_finalize, lmkd_bn_backward, lmkd_colsum) elect their finishing workgroup
 * through counters in a caller-owned buffer of lmkd_ticket_words() 32-bit words that must be ZERO on entry and is zero again when
 * the launch has finished; launches that may run concurrently (different streams) need different buffers. */
long lmkd_ticket_words(void);
/* stats: [5][C] = mean, invstd, scale, shift, unbiased batch variance.  scratch: >= 64*2*C doubles.
 * running_mean/var may be NULL (update deferred to lmkd_bn_running_update) */
int lmkd_bn_finalize(const float* partial, int T, int C, long count, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, float momentum, float eps, float* stats, double* scratch, unsigned* tickets, void* stream,
                     const lmkd_amax_desc* amax /*nullable*/);
int lmkd_bn_running_update(float* running_mean, float* running_var, const float* stats, int C, float momentum, void* stream);
/* n BatchNorm layers in one launch: HOST arrays of n device pointers / channel counts; layer i applies stats_first[i], then (if not
   null) stats_second[i] - the two trunk calls of an episode in the reference's order */
int lmkd_bn_running_update_multi(float* const* running_mean, float* const* running_var, const float* const* stats_first,
                                 const float* const* stats_second, const int* C, int n, float momentum, void* stream);
/* the same + nn.BatchNorm2d's num_batches_tracked (int64 device scalars; the array and single entries are nullable): += 1 per update applied */
int lmkd_bn_running_update_multi_nbt(float* const* running_mean, float* const* running_var, const float* const* stats_first,
                                     const float* const* stats_second, long long* const* num_batches_tracked, const int* C, int n,
                                     float momentum, void* stream);
int lmkd_bn_eval_stats(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                       float* stats, void* stream);
/* y = act(x*scale+shift [+res | +res*rscale+rshift]); res_mode 0 none, 1 plain, 2 affine.  mask_bits (nullable; needs
   C % 32 == 0): rows*C/32 words, bit e = (y[e] > 0) - the ReLU mask for lmkd_bn_backward(mask_mode 3) at 1/32 of y's bytes */
int lmkd_bn_apply(const float* x, const float* stats, const float* res, const float* rstats, float* y, long rows, int C, int relu,
                  int res_mode, unsigned* mask_bits, void* stream);
long lmkd_bn_bwd_workspace(int C);
/* mask_mode 0 none, 1 (yact>0), 2 (x*scale+shift>0), 3 (yact = the bit mask of lmkd_bn_apply); coef: [3][C] scratch;
   g_out (nullable) = masked dy; accumulate_param_grads != 0: dgamma / dbeta += (they point at the parameters' .grad: gradient
   accumulation over trunk calls and episodes, trainwandb.py:141-143, without a separate add) */
int lmkd_bn_backward(const float* dy, const float* x, const float* yact, const float* stats, const float* gamma, float* dx,
                     float* g_out, float* dgamma, float* dbeta, float* coef, void* workspace, unsigned* tickets, long rows, int C,
                     int mask_mode, int accumulate_param_grads, void* stream);
/* lmkd_bn_backward(mask_mode 2, no g_out) from the partial sums of lmkd_conv2d_bwd_data_bn: part [T][C][2]; dx may alias dy */
int lmkd_bn_backward_part(const float* part, int T, const float* dy, const float* x, const float* stats, const float* gamma, float* dx,
                          float* dgamma, float* dbeta, float* coef, void* workspace, unsigned* tickets, long rows, int C,
                          int accumulate_param_grads, void* stream);
int lmkd_relu_backward(const float* dy, const float* y, float* g, long n, void* stream);
/* stem: y = maxpool3x3/2/1(relu(x * scale + shift)), idx = arg-max byte (0..8) per output element, cmax (nullable) = x at the arg-max
   (torchvision resnet children 1-3, resnet18_2fc.py:33) */
int lmkd_bn_relu_maxpool_fwd(const float* x, const float* stats, float* y, unsigned char* idx, float* cmax, int N, int H, int W, int C,
                             void* stream);
int lmkd_maxpool_bwd(const float* dy, const unsigned char* idx, float* g, int N, int H, int W, int C, void* stream);
/* the stem's BatchNorm backward without the pre-pooling gradient tensor: (1) reduce + coefficient launches over the POOLED tensors
   (dy, cmax: [pooled_rows, C]; count = N*H*W of the pre-pooling tensor) -> coef [3][C], dgamma, dbeta (+= when accumulate_param_grads);
   (2) max-pool backward + BatchNorm backward apply in one pass -> dc [N,H,W,C], the gradient w.r.t. the convolution output c */
int lmkd_bn_backward_stats(const float* dy, const float* cmax, const float* stats, const float* gamma, float* dgamma, float* dbeta,
                           float* coef, void* workspace, unsigned* tickets, long pooled_rows, long count, int C, int accumulate_param_grads,
                           void* stream);
int lmkd_stem_unpool_bn_bwd(const float* dy, const unsigned char* idx, const float* c, const float* stats, const float* coef, float* dc,
                            int N, int H, int W, int C, void* stream, const lmkd_amax_desc* amax /*nullable*/);
/* AdaptiveMaxPool2d((4,4)) + mean over the 16 patches: resnet18_2fc.py:44-54 */
int lmkd_adaptive_maxpool_mean_fwd(const float* x, float* y, int F, int H, int W, int C, void* stream);
int lmkd_adaptive_maxpool_mean_bwd(const float* x, const float* dy, float* dx, int F, int H, int W, int C, void* stream);
long lmkd_colsum_workspace(int C);
int lmkd_colsum(const float* a, const float* b, float* out, long rows, int C, int accumulate, void* workspace, unsigned* tickets,
                void* stream);

/* ---- matchers: TemporalCrossTransformer (TRX_2fcsup.py:74-148), SupportDK (:162-189), e_dist (e_dist_fc2.py:52-91) ---- */
int lmkd_dropout_mask(float* mask, long n, float p, unsigned long long seed, void* stream);
/* the same mask with the seed read from device memory when the kernel runs (hipGraph replays draw new masks) */
int lmkd_dropout_mask_dev(float* mask, long n, float p, const unsigned long long* seed_dev, void* stream);
int lmkd_add_pe(const float* x, const float* pe, const float* mask, float* y, long rows, int D, int L, void* stream); /* :44-48,79-80 */
int lmkd_mul(const float* a, const float* b, float* out, long n, void* stream);
int lmkd_axpby(const float* x, float* y, float alpha, float beta, long n, void* stream);
int lmkd_trx_tuple_ln_fwd(const float* P, const float* bk, const float* bv, const float* gamma, const float* beta, const int* rowmap,
                          float* Kn, float* Khat, float* V, float* rstd, int NV, int L, int D, float eps, void* stream); /* :85-104 */
int lmkd_layernorm_bwd_rows(float* dK_inout, const float* Khat, const float* rstd, const float* gamma, long rows, int D, void* stream);
int lmkd_trx_tuple_bwd_gather(const float* dKraw, const float* dV, const int* rowmap, float* dP, int NV, int L, int D, void* stream);
/* seg_off/seg_cnt/seg_col are HOST int arrays (<= 16 classes) */
int lmkd_segment_softmax_fwd(float* S, long rows, long ld, int nseg, const int* seg_off, const int* seg_cnt, void* stream); /* :124-130 */
int lmkd_segment_softmax_bwd(const float* P, float* dP_inout, long rows, long ld, int nseg, const int* seg_off, const int* seg_cnt,
                             void* stream);
int lmkd_trx_dist_fwd(const float* Qv, const float* proto, float* logits, int Nq, int way, int T, int D, int nseg, const int* seg_col,
                      void* stream); /* :137-144 */
int lmkd_trx_dist_bwd(const float* Qv, float* proto_inout, const float* g, float* dQv, int Nq, int way, int T, int D, int nseg,
                      const int* seg_col, void* stream);
/* TRX_sup (TRX_sup.py:117-172): sim[q][a][b] = cosine similarity of query q's class-a and class-b prototypes; proto is the
   [nseg][Nq*T][D] prototype buffer of the TRX forward, seg_col[s] (host array) the class of segment s, gram [Nq][nseg][nseg] is kept for backward */
int lmkd_trx_sup_sim_fwd(const float* proto, const int* seg_col, float* sim, float* gram, int Nq, int way, int nseg, int T, int D,
                         void* stream);
int lmkd_trx_sup_sim_bwd(const float* proto, const int* seg_col, const float* gram, const float* dsim, float* dproto, int Nq, int way,
                         int nseg, int T, int D, void* stream);
long lmkd_supportdk_workspace(int way);
int lmkd_supportdk_fwd(const float* support, float* out, int way, int shot, int seq_len, int D, void* workspace, void* stream);
int lmkd_supportdk_bwd(const float* support, const float* g, float* dsupport, int way, int shot, int seq_len, int D, void* stream);
int lmkd_mean_frames(const float* x, float* y, long nv, int L, int D, void* stream);
int lmkd_mean_frames_bwd(const float* dy, float* dx, long nv, int L, int D, void* stream);
int lmkd_edist_fwd(const float* qm, const float* sm, const int* sup_class, float* dist, float* logits, int Nq, int Ns, int way, int D,
                   void* stream);
int lmkd_edist_bwd(const float* qm, const float* sm, const int* sup_class, const float* dist, const float* g, float* dqm, float* dsm,
                   int Nq, int Ns, int way, int D, void* stream);

/* ---- MFM teacher fusion (teacher/code/model.py:1135-1151,1300-1331,1361-1392,1648-1664), inference:
 * LayerNorm(x (+res[r mod res_rows])) written at row stride ldy, and the 8-token self-attention of nn.TransformerEncoderLayer ---- */
int lmkd_layernorm_fwd(const float* x, const float* res, long res_rows, const float* gamma, const float* beta, float* y, long ldy,
                       long rows, int D, float eps, void* stream);
int lmkd_mha_small(const float* qkv, float* out, int B, int L, int D, int H, void* stream);

/* ---- D2M loss (distillers.py:7-30, 295-337), accuracy (utils.py:116-121), optimizer (trainwandb.py:101-104,141-143) ---- */
int lmkd_d2m_loss(const float* s_kl, const float* t_kl, const float* s_ce, const long long* labels, const float* s_sup,
                  const float* t_sup, int Rq, int C, int Rs, int Cs, float T, float w_kl, float w_sup, float w_ce, float* out4,
                  float* g_kl, float* g_ce, float* g_sup, void* stream);
/* F.mse_loss(student_feature, teacher_feature) of Distiller.KL_feature (distillers.py:126-150, fed by trainwandb.py:209-226):
   out1[0] = mean((s-t)^2), grad (nullable) = 2 (s-t) / n; workspace: lmkd_mse_loss_workspace() bytes */
long lmkd_mse_loss_workspace(void);
int lmkd_mse_loss(const float* s, const float* t, long n, float* out1, float* grad, void* workspace, void* stream);
int lmkd_accuracy(const float* l1, const float* l2, const long long* labels, long long* pred, float* acc, int R, int C, void* stream);
int lmkd_sgd_step(float* param, float* grad, float lr, long n, int zero_grad, void* stream);
int lmkd_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                   long step, long n, int zero_grad, void* stream);
int lmkd_fill(float* p, float v, long n, void* stream);

/* ---- both trunk calls of an episode in ONE launch per layer (round 4) ----
 * The reference calls the trunk twice per episode with the same weights and shapes - self.resnet(context_images) and
 * self.resnet(target_images), model/backbone/resnet18_2fc.py:41-42 - and train-mode BatchNorm gives each call its own batch statistics.
 * The *_seg entry points take the two calls as ONE tensor [F0 + F1, H, W, C] with a frame split seg_n0 = F0 (rows0 = F0 * H * W for the
 * per-row kernels): "two frame segments".  BatchNorm tables become [2][5][C] (segment-major), partial-sum buffers hold segment 0's row
 * tiles first (no tile straddles the split), and every tensor element is computed by exactly the arithmetic, in the order, that a launch on
 * its own segment would use: outputs and activation gradients are bit-identical to the two-call form; weight / BatchNorm-parameter
 * gradients are summed over both segments in one pass (another summation order than two accumulating launches).
 * seg_n0 = 0 or N (rows0 = rows, T0 = T) = one segment = the plain entry point.  Modes 1-3 of lmkd_conv_set_compute_dtype. */
int lmkd_conv2d_fwd_row_tiles_seg(int N, int H, int W, int Cs, int Cout, int KH, int KW, int stride, int pad, int seg_n0, int* tiles0 /*host, nullable*/);
int lmkd_conv2d_fwd_seg(const float* x, const float* pre_stats /*nullable: [2][5][Cs], lmkd_conv2d_fwd_pre*/, const float* wp, float* y,
                        float* stat_partial, int N, int H, int W, int Cs, int Cout, int KH, int KW, int stride, int pad, int seg_n0, void* stream,
                        const lmkd_amax_desc* amax /*nullable*/);
int lmkd_conv2d_bwd_data_bn_tiles_seg(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int seg_n0, int* tiles0 /*host, nullable*/);
int lmkd_conv2d_bwd_data_seg(const float* dy, const float* wd, float* dx, const float* bn_x /*nullable with bn_stats, part: lmkd_conv2d_bwd_data_bn*/,
                             const float* bn_stats, float* part, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                             int accumulate, int seg_n0, void* stream, const lmkd_amax_desc* amax /*nullable*/);
long lmkd_conv2d_bwd_weight_workspace_seg(int N, int H, int W, int Cs, int Cout, int KH, int KW, int stride, int pad, int seg_n0);
int lmkd_conv2d_bwd_weight_seg(const float* x, const float* pre_stats /*nullable: [2][5][Cs]*/, const float* dy, float* dw_oihw, float* workspace,
                               long ws_bytes, int N, int H, int W, int Cs, int Cin, int Cout, int KH, int KW, int stride, int pad, int accumulate,
                               int seg_n0, void* stream, const lmkd_amax_desc* amax /*nullable*/);
/* BatchNorm family.  lmkd_bn_finalize_seg does not touch the running statistics (lmkd_bn_running_update_multi applies both segments'
   updates in the reference's order); scratch: 2 * 64 * 2 * C doubles.  The backward forms take coef = [2][5][C] floats of scratch and a
   workspace of 2 * lmkd_bn_bwd_workspace(C) bytes; dgamma / dbeta (+)= (segment 0's sums + segment 1's), written by the apply pass. */
int lmkd_bn_finalize_seg(const float* partial, int T, int T0, int C, long count0, long count1, const float* gamma, const float* beta, float eps,
                         float* stats /*[2][5][C]*/, double* scratch, unsigned* tickets, void* stream, const lmkd_amax_desc* amax /*nullable*/);
int lmkd_bn_apply_seg(const float* x, const float* stats, const float* res, const float* rstats, float* y, long rows, long rows0, int C, int relu,
                      int res_mode, unsigned* mask_bits, void* stream, const lmkd_amax_desc* amax /*nullable*/);
int lmkd_bn_backward_seg(const float* dy, const float* x, const float* yact, const float* stats, const float* gamma, float* dx, float* g_out,
                         float* dgamma, float* dbeta, float* coef, void* workspace, unsigned* tickets, long rows, long rows0, int C, int mask_mode,
                         int accumulate_param_grads, void* stream, const lmkd_amax_desc* amax /*nullable*/);
int lmkd_bn_backward_part_seg(const float* part, int T, int T0, const float* dy, const float* x, const float* stats, const float* gamma, float* dx,
                              float* dgamma, float* dbeta, float* coef, void* workspace, unsigned* tickets, long rows, long rows0, int C,
                              int accumulate_param_grads, void* stream, const lmkd_amax_desc* amax /*nullable*/);
/* the stem (resnet children 0-3): BatchNorm + ReLU + max-pool forward, and its backward from the pooled side */
int lmkd_bn_relu_maxpool_fwd_seg(const float* x, const float* stats, float* y, unsigned char* idx, float* cmax, int N, int N0, int H, int W, int C,
                                 void* stream, const lmkd_amax_desc* amax /*nullable*/);
int lmkd_bn_backward_stats_seg(const float* dy, const float* cmax, const float* stats, const float* gamma, float* coef, void* workspace,
                               unsigned* tickets, long pooled_rows, long pooled_rows0, long count, long count0, int C, void* stream);
int lmkd_stem_unpool_bn_bwd_seg(const float* dy, const unsigned char* idx, const float* c, const float* stats, const float* coef, float* dc,
                                float* dgamma, float* dbeta, int accumulate_param_grads, int N, int N0, int H, int W, int C, void* stream,
                                const lmkd_amax_desc* amax /*nullable*/);

#ifdef __cplusplus
}
#endif
#endif
