/* libsgta_hip.so -- C ABI of the MI355X (gfx950) SGTAPose inference hot path.
 *
 * Drop-in boundary.  The reference's only FFI precedent is
 * ``ctypes.cdll.LoadLibrary`` + ``extern "C" LM(double*, ..., int)`` (sgtapose/rf_tools/LM.py:10,
 * :256-266); this library follows it: plain C functions, raw device pointers owned by the caller
 * (inputs, outputs AND workspaces are allocated by the caller, e.g. torch), explicit sizes, the HIP
 * stream as the last argument.  No allocation and no synchronisation inside: every entry point only
 * enqueues kernels on ``stream`` (device memory is cleared by fill kernels, never by hipMemsetAsync), so a
 * sequence of calls can be captured in a hipGraph.  The only host-side state is a per-device cache of
 * one-time kernel attributes (hipFuncSetAttribute for kernels that need more than 64 KB of LDS, the
 * device's CU count): the first call of a kernel on a device sets them, so run every shape once before
 * capturing.  Calls from several host threads on distinct streams are safe once those first calls are
 * done.  The library reads no environment variables.
 *
 * Return value: 0 = ok, < 0 = bad argument (SGTA_EBADARG = -1), > 0 = hipError_t of the launch.
 * All tensors are contiguous float32 NCHW unless a batch stride ``*_bs`` (in ELEMENTS) is given,
 * which lets a tensor be a channel slice of a larger (concatenation) buffer.
 * ``pos`` / ``tok`` matrices are row-major [rows, ld].
 *
 * Each entry point names the reference code it replaces (paths relative to the reference repo).
 */
#ifndef SGTA_HIP_H
#define SGTA_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* sgta_stream_t; /* == hipStream_t */

#define SGTA_OK 0
#define SGTA_EBADARG (-1)
#define SGTA_ETOOBIG (-2) /* a size beyond the stated maximum of an entry point that owns its scratch memory */
#define SGTA_ACT_NONE 0
#define SGTA_ACT_RELU 1
#define SGTA_ACT_SIGMOID 2

/* library version (major*1000 + minor*100 + patch) and device check (0 iff current device is gfx950) */
int sgta_version(void);
int sgta_check_device(void);

/* conv2d (+ folded BatchNorm bias) (+ residual) (+ activation), implicit GEMM on f32 MFMA.
 * Replaces torch Conv2d + BatchNorm2d(eval) + ReLU of the DLA-34 backbone, stems, Root/Tree/BasicBlock,
 * the DCN offset/mask conv and the heads:
 *   sgtapose/lib/model/networks/dla.py:41-69, :157-175, :178-231, :259-270, :302-312;
 *   sgtapose/lib/model/networks/base_model.py:121-135.
 * wt   : packed weights [Kpad/16][Cout_pad][16], Kpad = roundup(Cin*ks*ks, 16), Cout_pad = roundup(Cout, 32),
 *        element (k/16, co, k%16) = W[co][c][ky][kx] * bn_scale[co] with k = c*ks*ks + ky*ks + kx, zero padded
 * bias : [Cout_pad] (BatchNorm shift / conv bias, zero padded)
 * ktab : int32 [Kpad][2] = {c*H*W + ky*W + kx, ky*ks + kx} per k; second field 63 for the K padding
 * res  : tensor with y's geometry added before (res_mode 1) or after (res_mode 2) the activation
 * ks in {1,3,7}; tile 0 = auto (1: 32x256, 2: 64x256, 3: 128x128 couts x pixels per work-group). */
int sgta_conv2d_f32(const float* x, long x_bs, const float* wt, const float* bias, const int* ktab,
                    const float* res, long res_bs, float* y, long y_bs, int B, int Cin, int H, int W,
                    int Cout, int Cout_pad, int Kpad, int ks, int stride, int pad, int act,
                    int res_mode, int tile, sgta_stream_t stream);

/* Same contract as sgta_conv2d_f32 for the dominant shape ks=3, stride=1, pad=1 (Cin % 4 == 0): the
 * contiguous input range covering all nine taps of a run of output pixels is staged in LDS with
 * coalesced loads and the im2col operand is read from LDS (no global gathers).
 * wt : packed [Cin/4][Cout_pad][36], element (c/4, co, (c%4)*9 + ky*3 + kx) = W[co][c][ky][kx]*bn_scale[co].
 * Returns SGTA_EBADARG when (W, tile) is outside the staged-range capacity (W <= 191 for tile 3,
 * <= 255 for tile 2, <= 767 for tile 1); callers then use sgta_conv2d_f32. */
int sgta_conv3x3s1_f32(const float* x, long x_bs, const float* wt, const float* bias, const float* res,
                       long res_bs, float* y, long y_bs, int B, int Cin, int H, int W, int Cout,
                       int Cout_pad, int act, int res_mode, int tile, sgta_stream_t stream);

/* Same contract again (ks=3, stride=1, pad=1, Cin % 8 == 0) computed with Winograd F(2x2,3x3) in fp32:
 * 2.25x fewer MFMA cycles than the direct formulations; results differ from them at the 1e-6 level.
 * u : host-transformed weights U = G g G^T, packed [Cin/8][8][Cout_pad][20]: element (c/8, c%8, co, p) with the 16
 *     positions p = 4*xi + nu first, then 4 pad floats (80-byte rows: conflict-free 16-byte LDS reads).
 * tile in {0,1,2,3} = 2x32 / 4x16 / 8x8 / 16x4 blocks of 2x2-pixel output tiles per work-group (64 tiles x 32
 * output channels; persistent work-groups, two per CU).  One image (Cin*H*W*4 bytes) must fit 31 bits.
 * ws / ws_floats: optional workspace (may be NULL / 0).  With act = 0 and no residual, a layer whose (block,
 * channel group) items leave most CUs idle (the 27-channel DCN offset convs of levels 4-5) is split over up to 8
 * slices of the input channels; the slices' partial sums (<= 8*B*Cout*H*W floats) meet in ws and are reduced in a
 * fixed order -- results do not depend on the split being taken. */
int sgta_conv3x3_wino_f32(const float* x, long x_bs, const float* u, const float* bias, const float* res,
                          long res_bs, float* y, long y_bs, int B, int Cin, int H, int W, int Cout,
                          int Cout_pad, int act, int res_mode, int tile, float* ws, long ws_floats,
                          sgta_stream_t stream);

/* Same contract with Winograd F(4x4,3x3) in fp32 (interpolation points 0, +-1, +-2, inf): 4x fewer MFMA cycles
 * than a direct convolution; results differ from it at the 1e-5 level.  Needs W % 4 == 0, Cin % 4 == 0, 16-byte
 * aligned x / y / res / bias and batch strides % 4 == 0 (else SGTA_EBADARG: use sgta_conv3x3_wino_f32).
 * u : host-transformed weights U = G g G^T (6x6), packed [Cin/4][4][Cout_pad][36]: element (c/4, c%4, co, 6*xi + nu).
 * tile in {0,1,2,3} = 2x32 / 4x16 / 8x8 / 16x4 blocks of 4x4-pixel output tiles per work-group (64 tiles x 32
 * output channels, 8 waves, one persistent work-group per CU; 64 x 16 / 4 waves / two per CU when Cout_pad % 32 != 0).
 * Cout_pad % 16 == 0. */
int sgta_conv3x3_wino4_f32(const float* x, long x_bs, const float* u, const float* bias, const float* res,
                          long res_bs, float* y, long y_bs, int B, int Cin, int H, int W, int Cout,
                          int Cout_pad, int act, int res_mode, int tile, sgta_stream_t stream);

/* 1x1 conv (+ bias, residual, activation) as a GEMM over the flattened pixels, both operands staged by
 * 16-byte LDS-DMA (Root / project convs, dla.py:157-175, :211-216; second conv of the heads,
 * base_model.py:122-123).  wt : packed [Cin_pad/16][Cout_pad][20], element (c/16, co, c%16) = W[co][c]*bn_scale[co]
 * (4 pad floats per row, zero beyond Cin / Cout).  Requires (H*W) % 4 == 0, x_bs % 4 == 0, 16-byte aligned x;
 * returns SGTA_EBADARG otherwise (callers then use sgta_conv2d_f32). */
int sgta_conv1x1_f32(const float* x, long x_bs, const float* wt, const float* bias, const float* res,
                     long res_bs, float* y, long y_bs, int B, int Cin, int H, int W, int Cout, int Cout_pad,
                     int Cin_pad, int act, int res_mode, sgta_stream_t stream);

/* 1x1 conv with Cout <= 8 (the output conv of each head, base_model.py:122-123): HBM-streaming kernel, one
 * thread = 4 consecutive pixels.  w: plain [Cout][Cin] fp32, bias [Cout].  H*W % 4 == 0, x / y 16-byte aligned,
 * batch strides % 4 == 0 (else SGTA_EBADARG: use sgta_conv1x1_f32 / sgta_conv2d_f32).  act: 0 none, 1 relu, 2 sigmoid. */
int sgta_conv1x1_skinny_f32(const float* x, long x_bs, const float* w, const float* bias, float* y, long y_bs,
                            int B, int Cin, int H, int W, int Cout, int act, sgta_stream_t stream);

/* conv3x3 / stride 2 / pad 1 + folded BN + ReLU (first conv of the down-sampling BasicBlocks, level1;
 * dla.py:44-46, :248-249): 2-D output tiles, de-interleaved input window staged by LDS-DMA.
 * wt : packed k-major [Cin/4][36][Cout_pad], k = c_local*9 + tap (the layout of sgta_dcnv2_lds_f32: rows of
 * consecutive output channels, staged by 16-byte LDS-DMA), Cin % 4 == 0, 16-byte aligned.
 * tile = 10*MT + t: MT in {1,2} = 32 / 64 output channels per work-group, t in {0,1,2} = 8x32 / 16x16 / 32x8 tiles. */
int sgta_conv3x3s2_f32(const float* x, long x_bs, const float* wt, const float* bias, float* y, long y_bs,
                       int B, int Cin, int H, int W, int Cout, int Cout_pad, int tile, sgta_stream_t stream);

/* The same convolution that also writes MaxPool2d(2, 2) of its output -- the ``bottom`` the following Tree takes of it
 * (dla.py:218-223) -- to pool [B, Cout, Ho/2, Wo/2] (batch stride pool_bs; NULL: plain conv).  Needs the 8x32 output
 * tile (tile % 10 == 0) and a map that is not stacked (H >= 16), else SGTA_EBADARG. */
int sgta_conv3x3s2_pool_f32(const float* x, long x_bs, const float* wt, const float* bias, float* y, long y_bs,
                            float* pool, long pool_bs, int B, int Cin, int H, int W, int Cout, int Cout_pad,
                            int tile, sgta_stream_t stream);

/* The 16-output-channel convs on the full-resolution maps (MFMA 16x16x4, whole input window + weights in LDS):
 *   mode 0: level0 = relu(bn(conv3x3(x [B,16,H,W])))                        (dla.py:246-247, :302-312); x2 = NULL
 *           wt [144][16] with k = c*9 + ky*3 + kx (BN scale folded), bias [16]
 *   mode 1: fused stems = relu(bn(conv7x7(x = image [B,3,H,W]))) + relu(bn(conv7x7(x2 = heat-map [B,1,H,W])))
 *           (dla.py:259-270, :324-328); wt = [148][16] image matrix (k = ch*49 + ky*7 + kx, row 147 zero)
 *           followed by [52][16] heat-map matrix (k = ky*7 + kx, rows 49..51 zero), bias [32].
 *   sequence mode (the two stems are separate conv+BN+ReLU whose outputs are only summed,
 *   dla.py:260-270, :325-328, so the image stem of frame t is computed once and reused when t becomes the previous frame):
 *   mode 2: image stem only, y = relu(bn(conv7x7(x = image)))                 (x2 ignored; weights / bias as mode 1)
 *   mode 3: y = x (the CACHED image stem [B,16,H,W]) + relu(bn(conv7x7(x2 = heat-map))) */
int sgta_conv_c16_f32(const float* x, long x_bs, const float* x2, long x2_bs, const float* wt,
                      const float* bias, float* y, long y_bs, int B, int H, int W, int mode,
                      sgta_stream_t stream);

/* Modulated deformable 3x3 conv v2 (stride 1, pad 1, dil 1, 1 deformable group) + folded BN + act,
 * gather fused into the GEMM (no column buffer).  Replaces the external lbin/DCNv2 ``dcn_v2_conv``
 * (import dla.py:21-25, constructed dla.py:545, called dla.py:548) and DeformConv's BN+ReLU
 * (dla.py:541-550).  ``om`` is the RAW 27-channel output of conv_offset_mask (channels 2k / 2k+1 =
 * dy / dx of tap k, 18+k = mask logit; the sigmoid is applied here).
 * wt : packed [9*Cin/16][Cout_pad][16] with k = tap*Cin + c (tap-major), Cin % 16 == 0; act in {0,1}. */
int sgta_dcnv2_fwd_f32(const float* x, long x_bs, const float* om, long om_bs, const float* wt,
                       const float* bias, float* y, long y_bs, int B, int Cin, int H, int W,
                       int Cout, int Cout_pad, int act, int tile, sgta_stream_t stream);

/* Same operator as sgta_dcnv2_fwd_f32, input window staged in LDS: a work-group owns a 2-D tile of 128
 * pixels (TH x TW), stages the (TH+12) x (TW+12) window of 4 input channels at a time by LDS-DMA
 * (double buffered) and samples bilinearly from LDS (per-lane global fallback for samples that leave the window,
 * so the result does not depend on the window size).
 * wt : packed [Cin/4][36][Cout_pad], element (c/4, (c%4)*9 + ky*3 + kx, co); Cin % 4 == 0, Cout_pad % 64 == 0.
 * tile = 10*MT + t: MT in {2,4} = 64 / 128 output channels per work-group, t in {0,1,2} = 4x32 / 8x16 / 16x8 tiles. */
int sgta_dcnv2_lds_f32(const float* x, long x_bs, const float* om, long om_bs, const float* wt,
                       const float* bias, float* y, long y_bs, int B, int Cin, int H, int W, int Cout,
                       int Cout_pad, int act, int tile, sgta_stream_t stream);

/* MaxPool2d(2,2)  (Tree.downsample, dla.py:209-210, :220). */
int sgta_maxpool2_f32(const float* x, long x_bs, float* y, long y_bs, int B, int C, int H, int W,
                      sgta_stream_t stream);

/* y = ConvTranspose2d(C,C,2f,stride f,pad f/2,groups=C,bias=False)(x) + skip  (IDAUp, dla.py:561-577);
 * w [C,1,2f,2f]; skip may be NULL; f in {2,4}. */
int sgta_upsample_add_f32(const float* x, long x_bs, const float* w, const float* skip, long s_bs,
                          float* y, long y_bs, int B, int C, int H, int W, int f, sgta_stream_t stream);

/* element-wise sigmoid (SGTADetector._sigmoid_output, sgtapose/lib/sgta_detector.py:854-862). */
int sgta_sigmoid_f32(const float* x, float* y, long n, sgta_stream_t stream);

/* get_topk_index with K=1 (dla.py:898-913): idx_xy float32 [2][B][K][2] = (x, y) of the arg-max of
 * pre_cls / cur_cls [B,K,h,w]; ties and all-zero maps -> lowest flat index. */
int sgta_prior_argmax(const float* pre_cls, const float* cur_cls, float* idx_xy, int B, int K, int h,
                      int w, sgta_stream_t stream);

/* Window pixel ids of get_topk_features_scale (dla.py:915-968) for both frames of one pyramid level:
 * fid int64 [2][B][n], n = K*(1+2*(ks/2))^2, x-major token order, fp32 coordinate math + truncation,
 * both axes clamped to H-1.  win uint8 [B][n]: 1 iff no later token of the CURRENT frame targets the
 * same pixel (ordered scatter of substitute_topk_features_scale, dla.py:1012-1014). */
int sgta_window_index(const float* idx_xy, long long* fid, unsigned char* win, int B, int K,
                      float scale, int ks, int H, int W, sgta_stream_t stream);

/* tok[b*n + t][col0 + c] = feats[b][c][fid[b][t]]   (gather half of dla.py:961-966, :1012). */
int sgta_window_gather(const float* feats, long f_bs, const long long* fid, float* tok, int B, int C,
                       int H, int W, int n, int ld, int col0, sgta_stream_t stream);

/* feats[b][c][fid[b][t]] = tok[b*n + t][c] for tokens with win == 1  (dla.py:1014). */
int sgta_window_scatter(const float* tok, const long long* fid, const unsigned char* win, float* feats,
                        long f_bs, int B, int C, int H, int W, int n, sgta_stream_t stream);

/* y[M,N] = LN?( relu?(x[M,K] . w[N,K]^T + bias) + res ): nn.Linear of MHCA_ein (dla.py:860-864),
 * the FFN (dla.py:721-732) and cat_layer (dla.py:1499-1502); LayerNorm (eps 1e-5) fused when
 * ln_gamma != NULL (requires N <= 64).  K % 16 == 0. */
int sgta_linear_f32(const float* x, int ldx, const float* w, const float* bias, const float* res,
                    int ldr, const float* ln_gamma, const float* ln_beta, float* y, int ldy, int M,
                    int N, int K, int relu, sgta_stream_t stream);

/* Same operator with a caller-owned workspace (ws_floats floats, may be NULL / 0): for the few-hundred-token
 * cat_layer products of levels 3-5 (dla.py:1537-1543: M = B*7 rows, K up to 2048) K is split over work-groups
 * and the partial sums are added in split order by a second kernel (deterministic).  4.2 M floats always suffice. */
int sgta_linear_ws_f32(const float* x, int ldx, const float* w, const float* bias, const float* res,
                       int ldr, const float* ln_gamma, const float* ln_beta, float* y, int ldy, int M,
                       int N, int K, int relu, float* ws, long ws_floats, sgta_stream_t stream);

/* y[M,C] = LayerNorm( x + W2 . relu(W1 . x + b1) + b2 ): the position-wise feed-forward block of
 * TransformerEncoderLayer (linear1 / ReLU / linear2 / residual / norm3, dla.py:721-732, :737-738; eps 1e-5) in ONE
 * kernel -- the [M,F] hidden activations stay in registers (the accumulator of the first MFMA product is the B operand
 * of the second).  x / y row-major with leading dimensions ldx / ldy (multiples of 4, y may not alias x);
 * w1 [F,C], b1 [F], w2 [C,F], b2 [C], ln_gamma / ln_beta [C]; C in {16,32,64}, F % 16 == 0; all pointers 16-byte
 * aligned. */
int sgta_ffn_ln_f32(const float* x, int ldx, const float* w1, const float* b1, const float* w2, const float* b2,
                    const float* ln_gamma, const float* ln_beta, float* y, int ldy, int M, int C, int F,
                    sgta_stream_t stream);

/* out[b,i,h*d:(h+1)*d] = softmax_j(q_i.k_j/sqrt(d) + pos_embed[h,i,j]) . v_j  per head
 * (MHCA_ein.forward, dla.py:874-885); q/k/v/out row-major [B*n, ld*]; pos_embed [n_heads,n,n] or NULL;
 * head_dim in {4,8,16}. */
int sgta_mhca_fwd_f32(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                      const float* pos_embed, float* out, int ldo, int B, int n, int n_heads,
                      int head_dim, sgta_stream_t stream);

/* Same operator with the bias parameter re-packed ONCE (it is a weight) into 16 x 16 tiles, T = ceil(n / 16):
 * pos_tiles[((h*T + it)*T + jt)*256 + li*16 + c] = pos_embed[h][it*16 + li][jt*16 + c], -inf for key columns >= n
 * (the padding mask), any finite value for query rows >= n; 16-byte aligned.  One coalesced 16-byte load per lane and
 * key tile replaces four strided 4-byte loads. */
int sgta_mhca_tiled_fwd_f32(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                            const float* pos_tiles, float* out, int ldo, int B, int n, int n_heads,
                            int head_dim, sgta_stream_t stream);

/* peaks_from_belief_maps + _peaks_info + the gathers of dream_generic_decode
 * (sgtapose/image_proc.py:1032-1143; sgtapose/lib/model/utils.py:207-284; sgtapose/lib/model/decode.py:184-313)
 * for EVERY sample of the batch.  hm [B,K,H,W] post-sigmoid; reg / tracking [B,2,H,W] or NULL;
 * gauss_w13: the 13 float64 taps w[0]=far .. w[12]=centre of scipy's sigma-3 kernel;
 * ws_tmp / ws_filt: float32 workspaces of hm's size.  Outputs: scores f32 [B,K] (-1 = missing),
 * xs / ys int64 [B,K], regs / trks f32 [B,K,2], npeaks int32 [B,K] (may be NULL). */
int sgta_peaks_decode(const float* hm, const float* reg, const float* tracking, const double* gauss_w13,
                      float* ws_tmp, float* ws_filt, float* scores, long long* xs, long long* ys,
                      float* regs, float* trks, int* npeaks, int B, int K, int H, int W,
                      sgta_stream_t stream);

/* The same decode plus the derived entries of dream_generic_decode's dictionary (decode.py:296-312), so that no
 * element-wise glue follows: cts int64 [B,K,2] = (xs, ys); cts_wreg f32 [B,K,2] = float32(cts) + regs; regs_rel f32
 * [B,K,2] = cts_wreg - float32(cts) (each may be NULL; the last two need reg).  three_pass != 0 forces the
 * three-kernel form (separable passes through ws_tmp, one work-group per map); by default both Gaussian passes, the
 * local-maximum test and the candidate selection of 24 rows of a map run in one work-group (LDS tile), and the tiles
 * of a map are merged by a second kernel -- identical results (same arithmetic per value, order-free selection).
 * ws_tmp must be 8-byte aligned (it also carries the per-tile candidates); ws_filt receives the filtered maps. */
int sgta_peaks_decode_full(const float* hm, const float* reg, const float* tracking, const double* gauss_w13,
                           float* ws_tmp, float* ws_filt, float* scores, long long* xs, long long* ys,
                           float* regs, float* trks, int* npeaks, long long* cts, float* cts_wreg,
                           float* regs_rel, int B, int K, int H, int W, int three_pass, sgta_stream_t stream);

/* sgta_peaks_decode_full plus kp f32 [B,K,3] = (x + reg_x, y + reg_y, score) per key-point (x + 0.5, y + 0.5 without
 * a 'reg' head, decode.py:304-306): the tensor that is all-gathered across ranks, written by the decode
 * kernel itself so that the step needs no element-wise glue.  regs / trks, when given, are zero-filled for a missing
 * head.  kp may be NULL. */
int sgta_peaks_decode_kp(const float* hm, const float* reg, const float* tracking, const double* gauss_w13,
                         float* ws_tmp, float* ws_filt, float* scores, long long* xs, long long* ys,
                         float* regs, float* trks, int* npeaks, long long* cts, float* cts_wreg,
                         float* regs_rel, float* kp, int B, int K, int H, int W, int three_pass,
                         sgta_stream_t stream);

/* sgta_peaks_decode_kp for callers that cannot allocate device memory.  It has NO counterpart in the reference: it exists for
 * code under hipGraph capture, tight serving loops and harnesses that forbid hipMalloc.  Same kernels, same outputs; the two
 * hm-sized workspaces (ws_tmp: per-tile peak candidates, ws_filt: the filtered maps) live in a device array owned by the library
 * (36 MiB per GPU the library runs on, held until the process ends) instead of caller buffers, and the call allocates nothing.
 * B*K*H*W must not exceed sgta_peaks_decode_static_max_elems() (4 718 592: 32 x 7 x 120 x 160 fits), else SGTA_ETOOBIG.
 * All calls on one GPU share that array: calls that overlap in time on one GPU race; nothing in it needs clearing. */
int sgta_peaks_decode_kp_static(const float* hm, const float* reg, const float* tracking, const double* gauss_w13,
                                float* scores, long long* xs, long long* ys, float* regs, float* trks, int* npeaks,
                                long long* cts, float* cts_wreg, float* regs_rel, float* kp, int B, int K, int H, int W,
                                int three_pass, sgta_stream_t stream);
long sgta_peaks_decode_static_max_elems(void);

/* SoftArgmaxPavlo.forward (sgtapose/spatial_softmax.py:24-95): out_xy f32 [B,K,2] = (E[col], E[row]). */
int sgta_softargmax(const float* hm, float* out_xy, int B, int K, int H, int W, float beta,
                    float size_mult, sgta_stream_t stream);

/* ---- the steps either side of the network ------------------------------------ */

/* get_prev_hm_wo_noise (per_class = 0: ONE map, key-points max-combined) / get_prev_hm_wo_noise_cls (per_class = 1:
 * one map per key-point) of sgtapose/utilities.py:1045-1057, 1086-1098, incl. affine_transform_and_clip
 * (:943-972) and draw_umich_gaussian (:800-821, radius 4, sigma 2, integer centre, skipped at the border).
 * kps f64 [B,K,2] raw-image coordinates; trans f64 [B,2,3] (meta['trans_input'] / meta['trans_output']);
 * raw_wh int32 [B,2] = (meta['width'], meta['height']); has_prior u8 [B] (0 = the reference's
 * ``kp_projs_raw is None``: zero map); table f32 [81] = exp(-(dx^2+dy^2)/8) evaluated in f64 by the caller.
 * out f32 [B, per_class ? K : 1, Hm, Wm] contiguous; it is zeroed by this call.  Inputs must be finite. */
int sgta_render_priors(const double* kps, const double* trans, const int* raw_wh, const unsigned char* has_prior,
                       const float* table, float* out, int B, int K, int Hm, int Wm, int per_class,
                       sgta_stream_t stream);

/* dream_generic_post_process (sgtapose/lib/utils/post_process.py:93-175) for the keys of this path, and
 * SGTADetector._get_final_kps (sgtapose/lib/sgta_detector.py:608-651, is_ct path).  Inputs = the device dets of
 * sgta_peaks_decode (scores f32 [B,K], cts int64 [B,K,2], cts_wreg / regs / tracking f32 [B,K,2]; tracking may be
 * NULL) and trans_inv f32 [B,2,3] = get_affine_transform(c, s, 0, (w, h), inv=1).astype(float32).
 * Outputs: valid u8 [B,K] (score >= out_thresh), ct / ct_wreg / reg / trk f32 [B,K,2] in raw-image coordinates,
 * final_kps f64 [B,K,2] (-999.999*4 where the class has no detection). */
int sgta_post_process(const float* scores, const long long* cts, const float* cts_wreg, const float* regs,
                      const float* tracking, const float* trans_inv, float out_thresh, unsigned char* valid,
                      float* ct, float* ct_wreg, float* reg, float* trk, double* final_kps, int B, int K,
                      sgta_stream_t stream);

/* has[b] (u8 [B]) = 1 iff final_kps [B,K,2] (f64, sgta_post_process) holds a detected key-point for sample b, i.e. a coordinate
 * above the missing marker -999.999*4: SGTADetector._get_further_dt_pnp_inputs (sgtapose/lib/sgta_detector.py:471-479) feeds
 * all-zero prior maps to the next frame otherwise.  ``has`` is sgta_render_priors' has_prior. */
int sgta_kps_any_valid(const double* final_kps, unsigned char* has, int B, int K, sgta_stream_t stream);

/* SGTADetector.pre_process (sgtapose/lib/sgta_detector.py:368-399, normalize_img :401-402) for a batch of raw frames of
 * one size: cv2.warpAffine(image, trans_input, (Wo, Ho), INTER_LINEAR, constant border 0) + (img / 255 - mean) / std in
 * float32 + HWC -> CHW.  images u8 [B,Hr,Wr,3]; inv_trans f64 [B,2,3] = the INVERSE of meta['trans_input'] (computed by
 * the caller in double exactly as OpenCV's warpAffine does); mean / stdv f32 [3]; out f32 [B,3,Ho,Wo].  The warp restates
 * OpenCV's published 8-bit algorithm (1/32-pixel fixed-point coordinates, 15-bit weights); OpenCV is absent from the
 * image, so it is unpinned against the library and bit-exact against oracle/prepost.py. */
int sgta_pre_process(const unsigned char* images, const double* inv_trans, const float* mean, const float* stdv,
                     float* out, int B, int Hr, int Wr, int Ho, int Wo, sgta_stream_t stream);

/* ---- pose step downstream of the key-points, fp64, one thread per frame ---------------
 * sgta_pnp_batched = solve_pnp (sgtapose/geometric_vision.py:43-116: cv2.solvePnP EPNP + ITERATIVE refinement) and, when
 * next_proj != NULL, is_pnp (:283-310): the pose of each frame applied to ``next_x3d`` and projected with ``camK`` (on a
 * failed solve the detections ``x2d`` themselves are returned, as the reference does).  x3d f64 [B,K,3] object points,
 * x2d f64 [B,K,2] detections in raw-image pixels; valid u8 [B,K] or NULL (NULL: a key-point is used iff one of its
 * coordinates exceeds the detector's missing marker -999.999*4, sgta_detector.py:469-472); camK f64 [9] row-major.
 * Outputs: ok u8 [B], t f64 [B,3], quat_xyzw f64 [B,4] (convert_rvec_to_quaternion), next_proj f64 [B,K,2].  4 <= K <= 16.
 * OpenCV is absent from the image: EPnP and the refinement follow the published algorithms (see pnp.hip); unpinned
 * against the library, checked against oracle/pnp.py and against the poses the test data were generated from. */
int sgta_pnp_batched(const double* x3d, const double* x2d, const unsigned char* valid, const double* camK,
                     const double* next_x3d, unsigned char* ok, double* t_out, double* quat_xyzw, double* next_proj,
                     int B, int K, sgta_stream_t stream);

/* The reference's Gauss-Newton pose refinement: ``extern "C" LM(double*, ..., int)`` of libtestso_final.so with its python
 * twin GN / fun / dfun (sgtapose/rf_tools/LM.py:128-234, :256-266), batched.  value0 f64 [B,7] = (qw,qx,qy,qz,tx,ty,tz);
 * x2d [B,K,2], x3d [B,K,3], valid u8 [B,K] or NULL, weights f64 [B,K,2] or NULL (= 1); out f64 [B,7]; iters int32 [B] or
 * NULL.  Same iteration as the reference: value -= (float32(J^T J) + 1e-4 I)^-1 J^T f while sum|delta| > 1e-4, at most
 * max_iter (200) times, the reference's Jacobian slip (y rows: d/dqz = d/dqy) included; a NaN update ends the loop. */
int sgta_gn_refine(const double* value0, const double* x2d, const double* x3d, const unsigned char* valid,
                   const double* weights, const double* camK, double* out, int* iters, int B, int K, int max_iter,
                   sgta_stream_t stream);

/* ---- bf16 data path (BASELINE configs[2]: bf16 feature maps / weights, fp32 accumulate / softmax / LayerNorm / decode) ----
 * Feature maps are bf16 BLOCKED channels-last [N][C/16][H][W][16]: a tensor may be a range of 16-channel blocks of a larger
 * (concatenation) buffer -- every entry point takes the pointer at the view's first block and ``*_bs`` = elements per
 * image of the underlying buffer; weights are bf16 in the packed LDS-image layouts produced by
 * sgtapose_amd/ops_bf16.py; all matrix products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  What leaves
 * the bf16 domain is fp32: DCN offsets / mask logits, head outputs, attention tokens, decode. */

/* conv (3x3 stride 1 / 2, pad 1; 1x1) + folded BN bias (+ residual before the activation) (+ ReLU): the bf16 twin of
 * sgta_conv2d_f32 (dla.py:41-69, :157-231, :302-312; DCN offset conv; heads base_model.py:121-135).
 * x_bytes / wt_bytes: bytes addressable from x / wt (< 2^31).  wt: [Cout tile][Cin/kc][tap][32*na][kc] bf16, rows
 * XOR-swizzled in 16-byte pieces.  res: bf16 blocked at the output resolution or NULL.  out_mode 0: y bf16 blocked (y_bs
 * elements per image); 2: y f32 NHWC [N,Ho,Wo,y_bs]; 1: y f32 NCHW [N,Cout,Ho,Wo]
 * with sigmoid on channels < nsig (SGTADetector._sigmoid_output, sgta_detector.py:854-862).  kc = 16 (3x3) / 32 (1x1),
 * Cin % kc == 0; na in {1,2,4} (2 max for stride 2); npf = 32-position fragments per wave (2: 512-position tiles, 1: 256;
 * stride 2 needs 1); log_tw = log2 of the tile width (3..6). */
int sgta_conv_bf16(const void* x, long x_bytes, long x_bs, const void* wt, long wt_bytes, const float* bias,
                   const void* res, long res_bs, void* y, long y_bs, int N, int H, int W, int Cin, int Cout,
                   int ks, int stride, int kc, int na, int npf, int log_tw, int act, int out_mode, int nsig,
                   sgta_stream_t stream);

/* The head branches (base_model.py:121-135: conv3x3 -> ReLU -> conv1x1 per head) in ONE kernel: hidden = relu(conv3x3(x) + b1)
 * with the branches' 3x3 convs concatenated (Chid channels; w1 / b1 in sgta_conv_bf16's layout for na = 4) is rounded to bf16
 * in registers and multiplied by the block-diagonal 1x1 weights of all branches (<= 16 output rows) -- the hidden tensor never
 * exists in memory.  out_i: f32 NCHW [N, c_i, H, W] (c_i = 0: unused), sigmoid on rows < nsig.  w2: bf16 A fragments
 * [Chid/128][4][2][64 lanes][8] of the second v_mfma_f32_32x32x16_bf16: element j of lane (m = lane & 31, h = lane >> 5) in
 * k-step t of 32-channel block i of tile ct is W2[m][128 ct + 32 i + 8 (2t + (j >> 2)) + 4h + (j & 3)]; b2 f32 [16]. */
int sgta_heads_bf16(const void* x, long x_bytes, long x_bs, const void* w1, long w1_bytes, const float* b1,
                    const void* w2, const float* b2, float* out0, float* out1, float* out2, float* out3, int c0, int c1,
                    int c2, int c3, int N, int H, int W, int Cin, int Chid, int npf, int log_tw, int nsig,
                    sgta_stream_t stream);

/* DeformConv (dla.py:538-550) = lbin/DCNv2 modulated deformable 3x3 conv + folded BN + ReLU on bf16 blocked maps -- the
 * ``sgta_dcnv2_fwd_bf16``.  om: f32 NHWC [N,H,W,ldo], channels 0..17 = (dy, dx) per tap,
 * 18..26 = mask logits (raw conv_offset_mask output, sigmoid applied here); sampling coordinates and bilinear /
 * mask weights in fp32, the gathered corners are bf16 (16-byte channel runs), blend in fp32, product on bf16 MFMA.
 * ldo >= 28 and ldo % 4 == 0 (a pixel's 27 values are fetched as 7 x 16 bytes).  The sampling neighbourhood of a tile is
 * staged in LDS (offsets up to +-6 pixels; beyond that a wave falls back to gathers from global memory, same arithmetic),
 * converted once per 16-channel chunk to fp16 and blended by v_fma_mix_f32.
 * wt: [Cout tile][Cin/16][9][32*na][16] bf16 swizzled (the 3x3 layout of sgta_conv_bf16); Cin % 16 == 0; na in {2,4}; margin 6 (samples
 * up to +-6 pixels from the tile are gathered from the staged window; log_tw 3..5) or 10 (+-10 pixels plus patch slots for the
 * samples beyond; log_tw 3..4) -- speed only, the result does not depend on it; wt is FP16 (the main product runs on
 * v_mfma_f32_32x32x16_f16 with the blended samples in fp16). */
int sgta_dcnv2_fwd_bf16(const void* x, long x_bs, const float* om, int ldo, const void* wt, long wt_bytes,
                        const float* bias, void* y, long y_bs, int N, int H, int W, int Cin, int Cout,
                        int na, int log_tw, int margin, int act, sgta_stream_t stream);

/* fused 7x7 stems (dla.py:259-270, :324-328): fp32 NCHW image [B,3,H,W] + prior map [B,1,H,W] in, bf16 [B,H,W,ldy]
 * (16 channels; ldy = 16 is the one-block blocked layout [B][1][H][W][16]) out; weights / bias as sgta_conv_c16_f32 mode 1. */
int sgta_stem_bf16(const float* x, long x_bs, const float* x2, long x2_bs, const float* wt, const float* bias,
                   void* y, int ldy, int B, int H, int W, sgta_stream_t stream);

/* The whole input stage in one kernel: both 7x7 stems, their sum (dla.py:259-270, :324-328) and level0 = conv3x3(16 -> 16)
 * + BN + ReLU (dla.py:271-273) -- fp32 NCHW image [B,3,H,W] (img_bs elements per image) + prior map [B,1,H,W] in, bf16
 * blocked level0 map [B][1][H][W][16] (y_bs elements per image) out; the stem map stays in LDS.  w_stem: bf16 A fragments
 * [7 k-steps = ky][2: image stem, prior stem][64 lanes][8] of v_mfma_f32_16x16x32_bf16 (row m = lane & 15 carries channel
 * 4*perm[m>>2] + (m&3), perm = {0,2,1,3}; element j of lane k-group g = lane >> 4 is tap kx = 2g + (j>>2) (kx = 7: zero),
 * input channel j & 3 (0..2 image, 3 prior map; zero in the other stem's fragments)); b_stem f32 [32] (folded BN);
 * w_l0: bf16 [5 k-steps][64 lanes][8]: tap 2s + (g>>1), input channels 8*(g&1)..+7; b_l0 f32 [16].  W % 4 == 0. */
int sgta_stem_level0_bf16(const float* img, long img_bs, const float* hm, long hm_bs, const void* w_stem,
                          const float* b_stem, const void* w_l0, const float* b_l0, void* y, long y_bs, int B, int H,
                          int W, sgta_stream_t stream);

/* MaxPool2d(2,2) (dla.py:218-220) on bf16 blocked maps. */
int sgta_maxpool2_bf16(const void* x, long x_bs, void* y, long y_bs, int N, int C, int H, int W, sgta_stream_t stream);

/* IDAUp's depthwise ConvTranspose2d(C,C,2f,stride f,pad f/2,groups C) + skip add (dla.py:571-577) on bf16 blocked maps;
 * w f32 [2f][2f][C]; skip (bf16 blocked at the output resolution) may be NULL. */
int sgta_upsample_add_bf16(const void* x, long x_bs, const float* w, const void* skip, long skip_bs, void* y, long y_bs,
                           int N, int C, int H, int W, int f, sgta_stream_t stream);

/* get_topk_features_scale's gather (dla.py:961-968) from a bf16 blocked map into fp32 token rows
 * tok[b*n + t][col0 + c], and substitute_topk_features_scale's ordered write-back (dla.py:1012-1014, winners of
 * sgta_window_index only) of fp32 token rows into the bf16 blocked map. */
int sgta_window_gather_bf16(const void* feats, long feats_bs, const long long* fid, float* tok, int B, int C, int H,
                            int W, int n, int ld, int col0, sgta_stream_t stream);
int sgta_window_scatter_bf16(const float* tok, int ldt, const long long* fid, const unsigned char* win,
                             void* feats, long feats_bs, int B, int C, int H, int W, int n, sgta_stream_t stream);

/* ---- the same 16-bit data path on IEEE half (fp16) feature maps and weights ------------------------------------------------
 * Every ``sgta_*_bf16`` entry point above has a twin with identical arguments, layouts and reference citations whose 16-bit
 * tensors are IEEE half instead of bfloat16 (v_mfma_f32_32x32x16_f16 / _16x16x32_f16, fp32 accumulate): 11 significant bits
 * instead of 8, i.e. an eighth of bf16's rounding error on the head outputs, for a range of +-65504 (values beyond are
 * clamped on store, never written as infinity).  sgta_dcnv2_fwd_f16 takes the same fp16 weights as sgta_dcnv2_fwd_bf16. */
int sgta_conv_f16(const void* x, long x_bytes, long x_bs, const void* wt, long wt_bytes, const float* bias,
                  const void* res, long res_bs, void* y, long y_bs, int N, int H, int W, int Cin, int Cout,
                  int ks, int stride, int kc, int na, int npf, int log_tw, int act, int out_mode, int nsig,
                  sgta_stream_t stream);
int sgta_heads_f16(const void* x, long x_bytes, long x_bs, const void* w1, long w1_bytes, const float* b1,
                   const void* w2, const float* b2, float* out0, float* out1, float* out2, float* out3, int c0, int c1,
                   int c2, int c3, int N, int H, int W, int Cin, int Chid, int npf, int log_tw, int nsig,
                   sgta_stream_t stream);
int sgta_dcnv2_fwd_f16(const void* x, long x_bs, const float* om, int ldo, const void* wt, long wt_bytes,
                       const float* bias, void* y, long y_bs, int N, int H, int W, int Cin, int Cout,
                       int na, int log_tw, int margin, int act, sgta_stream_t stream);
int sgta_stem_f16(const float* x, long x_bs, const float* x2, long x2_bs, const float* wt, const float* bias,
                  void* y, int ldy, int B, int H, int W, sgta_stream_t stream);
int sgta_stem_level0_f16(const float* img, long img_bs, const float* hm, long hm_bs, const void* w_stem,
                         const float* b_stem, const void* w_l0, const float* b_l0, void* y, long y_bs, int B, int H,
                         int W, sgta_stream_t stream);
int sgta_maxpool2_f16(const void* x, long x_bs, void* y, long y_bs, int N, int C, int H, int W, sgta_stream_t stream);
int sgta_upsample_add_f16(const void* x, long x_bs, const float* w, const void* skip, long skip_bs, void* y, long y_bs,
                          int N, int C, int H, int W, int f, sgta_stream_t stream);
int sgta_window_gather_f16(const void* feats, long feats_bs, const long long* fid, float* tok, int B, int C, int H,
                           int W, int n, int ld, int col0, sgta_stream_t stream);
int sgta_window_scatter_f16(const float* tok, int ldt, const long long* fid, const unsigned char* win,
                            void* feats, long feats_bs, int B, int C, int H, int W, int n, sgta_stream_t stream);

/* ---- fp32 emulated on the 16-bit matrix pipe by operand splitting ("x3") ------------------------------------------------------
 * Activations are TWO IEEE-half planes of the blocked layout above, value = hi + lo with hi = half(value), lo = half(value -
 * hi) (22 significant bits); ``*_lo_delta`` = elements from a tensor's hi plane to its lo plane (both planes in one
 * allocation).  Weights are split the same way; every product is x_hi w_hi + x_hi w_lo + x_lo w_hi with fp32 accumulation
 * (the dropped x_lo w_lo term is 2^-22 relative), so results agree with the fp32 operators they replace (the same reference
 * code as sgta_conv2d_f32 / the heads) to ~1e-6 relative.  wt: sgta_conv_f16's layout over 3*Cin virtual channels -- per
 * kc-channel chunk [w_hi, w_lo, w_hi]; Cin = the layer's real input channels. */
int sgta_conv_x3(const void* x, long x_bytes, long x_bs, long x_lo_delta, const void* wt, long wt_bytes, const float* bias,
                 const void* res, long res_bs, long res_lo_delta, void* y, long y_bs, long y_lo_delta, int N, int H, int W,
                 int Cin, int Cout, int ks, int stride, int kc, int na, int npf, int log_tw, int act, int out_mode, int nsig,
                 sgta_stream_t stream);
/* sgta_heads_f16 with split operands: w1 as sgta_conv_x3's wt, w2 / w2_lo = hi and lo parts of the second product's fragments. */
int sgta_heads_x3(const void* x, long x_bytes, long x_bs, long x_lo_delta, const void* w1, long w1_bytes, const float* b1,
                  const void* w2, const void* w2_lo, const float* b2, float* out0, float* out1, float* out2, float* out3,
                  int c0, int c1, int c2, int c3, int N, int H, int W, int Cin, int Chid, int npf, int log_tw, int nsig,
                  sgta_stream_t stream);
/* MaxPool2d(2,2) (dla.py:218-220), window gather / ordered scatter (dla.py:915-968, :1006-1018) on x3 maps, and the hand-off
 * between fp32 NCHW tensors and x3 maps (exact in the x3 -> fp32 direction, 22 bits kept in the other). */
int sgta_maxpool2_x3(const void* x, long x_bs, long x_lo_delta, void* y, long y_bs, long y_lo_delta, int N, int C, int H, int W,
                     sgta_stream_t stream);
int sgta_window_gather_x3(const void* feats, long feats_bs, long feats_lo_delta, const long long* fid, float* tok, int B, int C,
                          int H, int W, int n, int ld, int col0, sgta_stream_t stream);
int sgta_window_scatter_x3(const float* tok, int ldt, const long long* fid, const unsigned char* win, void* feats, long feats_bs,
                           long feats_lo_delta, int B, int C, int H, int W, int n, sgta_stream_t stream);
int sgta_nchw_to_x3(const float* x, long x_bs, void* y, long y_bs, long y_lo_delta, int N, int C, int H, int W,
                    sgta_stream_t stream);
int sgta_x3_to_nchw(const void* x, long x_bs, long x_lo_delta, float* y, long y_bs, int N, int C, int H, int W,
                    sgta_stream_t stream);
/* The fused input stage of sgta_stem_level0_f16 (both 7x7 stems, their sum, level0: dla.py:259-273, :324-328) at fp32 accuracy on
 * the 16-bit matrix pipe: the staged input, the stem region and both weight sets are IEEE-half pairs, three MFMA products per
 * term.  w_stem / w_stem_lo and w_l0 / w_l0_lo: the A fragments of sgta_stem_level0_f16 taken from the hi / lo halves of the folded
 * fp32 weights.  out_f32 = 0: y = IEEE-half blocked [B][1][H][W][16] (y_bs elements per image), lo plane y_lo_delta elements
 * behind; out_f32 = 1: y = fp32 NCHW [B,16,H,W] (y_bs elements per image).  W % 4 == 0, 16-byte aligned pointers. */
int sgta_stem_level0_x3(const float* img, long img_bs, const float* hm, long hm_bs, const void* w_stem, const void* w_stem_lo,
                        const float* b_stem, const void* w_l0, const void* w_l0_lo, const float* b_l0, void* y, long y_bs,
                        long y_lo_delta, int out_f32, int B, int H, int W, sgta_stream_t stream);
/* The modulated deformable conv of sgta_dcnv2_lds_f32 (reference: lbin/DCNv2 behind DeformConv, dla.py:538-550) on the same fp32
 * NCHW tensors at fp32 accuracy, with the product on the 16-bit matrix pipe: the bilinear blend runs in fp32, each blended sample
 * and each weight is an IEEE-half pair (hi + lo), three MFMA products per term.  x 16-byte aligned, x_bs % 4 == 0, W % 4 == 0,
 * Cin % 8 == 0; om: raw offset / mask conv output [B,27,H,W]; wt: IEEE half [n_ct][Cin/8][5][2][32*mt][16] with n_ct =
 * ceil(Cout / (32*mt)), k = 8*(tap & 1) + (channel % 8) at step tap / 2, parts hi / lo (wt_bytes = its size, checked); bias f32
 * [n_ct*32*mt] (BN folded); mt in {2, 4}; log_tw in 3..5: a work-group's tile is (128 >> log_tw) rows x (1 << log_tw) columns;
 * act 1 = ReLU.  Results do not depend on mt / log_tw beyond fp32 summation order. */
int sgta_dcnv2_x3(const float* x, long x_bs, const float* om, long om_bs, const void* wt, long wt_bytes, const float* bias,
                  float* y, long y_bs, int B, int Cin, int H, int W, int Cout, int mt, int log_tw, int act, sgta_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SGTA_HIP_H */
