// The steps either side of the network, on the device (gfx950):
//  * prior heat-map rendering -- get_prev_hm_wo_noise / get_prev_hm_wo_noise_cls (reference
//    sgtapose/utilities.py:1045-1057, 1086-1098 on top of affine_transform_and_clip :943-972 and
//    draw_umich_gaussian :800-821): fp64 affine of the projected key-points, clip, truncation to the integer centre,
//    9x9 sigma-2 Gaussian patch, element-wise max -- replaces four numpy renders + host->device copies per frame;
//  * detection post-processing -- dream_generic_post_process (lib/utils/post_process.py:93-175) and
//    SGTADetector._get_final_kps (lib/sgta_detector.py:608-651): inverse affine of the decoded centres / offsets
//    in fp32, score threshold, per-class key-point table in fp64 (missing = -999.999 * 4).
// Latency-bound helpers: one work-group per (sample, key-point) / one thread per detection.
#include "common.h"

// kps [B,K,2] fp64 raw-image coordinates, trans [B,2,3] fp64, raw_wh [B,2] int32, has_prior [B] (0: map stays 0),
// table [81] = float32(exp(-(dx^2+dy^2)/8)) built by the host in fp64 (bit-exact with numpy's).
__global__ __launch_bounds__(128) void render_prior_kernel(const double* __restrict__ kps,
                                                           const double* __restrict__ trans,
                                                           const int* __restrict__ raw_wh,
                                                           const unsigned char* __restrict__ has_prior,
                                                           const float* __restrict__ table, float* __restrict__ out,
                                                           int K, int Hm, int Wm, int per_class) {
  const int b = blockIdx.x / K, k = blockIdx.x - b * K, tap = threadIdx.x;
  if (tap >= 81 || !has_prior[b]) return;
  const double px = kps[(b * K + k) * 2], py = kps[(b * K + k) * 2 + 1];
  const double* t = trans + b * 6;
  // affine_transform_and_clip: np.dot(t, [x y 1]^T), clip to the map, (0, 0) for points outside the raw image
  double nx = (t[0] * px + t[1] * py) + t[2];
  double ny = (t[3] * px + t[4] * py) + t[5];
  nx = fmin(fmax(nx, 0.0), (double)(Wm - 1));
  ny = fmin(fmax(ny, 0.0), (double)(Hm - 1));
  const bool inside = px >= 0.0 && px < (double)raw_wh[2 * b] && py >= 0.0 && py < (double)raw_wh[2 * b + 1];
  if (!inside) { nx = 0.0; ny = 0.0; }
  const int x = (int)nx, y = (int)ny;                     // python int(): truncation
  // draw_umich_gaussian, radius 4: drawn only if the patch and one more column / row fit
  if (!(x - 4 >= 0 && x + 5 < Wm && y - 4 >= 0 && y + 5 < Hm)) return;
  const int dy = tap / 9 - 4, dx = tap - (tap / 9) * 9 - 4;
  const float v = table[tap];
  float* dst = out + ((long)b * (per_class ? K : 1) + (per_class ? k : 0)) * Hm * Wm + (long)(y + dy) * Wm + (x + dx);
  if (per_class) *dst = v;                                 // one key-point per map: a single writer per element
  else atomicMax(reinterpret_cast<int*>(dst), __float_as_int(v));   // positive floats order like their bit patterns
}

// zero fill by a kernel, not hipMemsetAsync: a memset node recorded into a captured graph is not safe to replay
__global__ __launch_bounds__(256) void zero_f32x4_kernel(float* __restrict__ p, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) *reinterpret_cast<f32x4*>(p + i) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  else
    for (long j = i; j < n; ++j) p[j] = 0.0f;
}

extern "C" int sgta_render_priors(const double* kps, const double* trans, const int* raw_wh,
                                  const unsigned char* has_prior, const float* table, float* out, int B, int K,
                                  int Hm, int Wm, int per_class, hipStream_t st) {
  if (!kps || !trans || !raw_wh || !has_prior || !table || !out || B <= 0 || K <= 0 || Hm <= 0 || Wm <= 0)
    return SGTA_EBADARG;
  const long n = (long)B * (per_class ? K : 1) * Hm * Wm;
  if (((uintptr_t)out & 15) != 0) return SGTA_EBADARG;                 // 16-byte stores in the fill
  hipLaunchKernelGGL(zero_f32x4_kernel, dim3(cdiv(n, 1024)), dim3(256), 0, st, out, n);
  SGTA_LAUNCH_CHECK();
  hipLaunchKernelGGL(render_prior_kernel, dim3(B * K), dim3(128), 0, st, kps, trans, raw_wh, has_prior, table, out,
                     K, Hm, Wm, per_class);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// has[b] = 1 iff sample b holds at least one detected key-point (a coordinate above the detector's missing marker
// -999.999 * 4): SGTADetector._get_further_dt_pnp_inputs returns all-zero prior maps otherwise
// (sgtapose/lib/sgta_detector.py:471-479).  Keeps that decision on the device in a closed video loop.
__global__ __launch_bounds__(64) void kps_any_valid_kernel(const double* __restrict__ kps, unsigned char* __restrict__ has,
                                                            int B, int K) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  bool any = false;
  for (int i = 0; i < 2 * K; ++i) any = any || kps[(long)b * 2 * K + i] > -999.999 * 4;
  has[b] = any ? 1 : 0;
}

extern "C" int sgta_kps_any_valid(const double* final_kps, unsigned char* has, int B, int K, hipStream_t st) {
  if (!final_kps || !has || B <= 0 || K <= 0) return SGTA_EBADARG;
  hipLaunchKernelGGL(kps_any_valid_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, final_kps, has, B, K);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// One thread per (sample, class).  trans_inv [B,2,3] fp32 (the reference casts it, post_process.py:103).
__global__ __launch_bounds__(64) void post_process_kernel(const float* __restrict__ scores,
                                                          const long long* __restrict__ cts,
                                                          const float* __restrict__ cts_wreg,
                                                          const float* __restrict__ regs,
                                                          const float* __restrict__ tracking,
                                                          const float* __restrict__ trans_inv, float out_thresh,
                                                          unsigned char* __restrict__ valid, float* __restrict__ o_ct,
                                                          float* __restrict__ o_ct_wreg, float* __restrict__ o_reg,
                                                          float* __restrict__ o_trk, double* __restrict__ final_kps,
                                                          int n, int K) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const float* t = trans_inv + (i / K) * 6;
  const float cx = (float)cts[2 * i], cy = (float)cts[2 * i + 1];
  // transform_preds_with_trans: fp32 [x y 1] . trans^T
  const float ctx = (t[0] * cx + t[1] * cy) + t[2], cty = (t[3] * cx + t[4] * cy) + t[5];
  const float wx = cts_wreg[2 * i], wy = cts_wreg[2 * i + 1];
  const float cwx = (t[0] * wx + t[1] * wy) + t[2], cwy = (t[3] * wx + t[4] * wy) + t[5];
  const float rx = regs[2 * i], ry = regs[2 * i + 1];      // transform_preds_with_trans_offset: no translation
  const float rgx = t[0] * rx + t[1] * ry, rgy = t[3] * rx + t[4] * ry;
  const bool ok = !(scores[i] < out_thresh);
  valid[i] = ok ? 1 : 0;
  o_ct[2 * i] = ctx; o_ct[2 * i + 1] = cty;
  o_ct_wreg[2 * i] = cwx; o_ct_wreg[2 * i + 1] = cwy;
  o_reg[2 * i] = rgx; o_reg[2 * i + 1] = rgy;
  if (tracking) {
    const float sx = tracking[2 * i] + cx, sy = tracking[2 * i + 1] + cy;
    o_trk[2 * i] = ((t[0] * sx + t[1] * sy) + t[2]) - ctx;
    o_trk[2 * i + 1] = ((t[3] * sx + t[4] * sy) + t[5]) - cty;
  }
  // _get_final_kps: the decode emits exactly one detection per class
  final_kps[2 * i] = ok ? (double)cwx : -999.999 * 4;
  final_kps[2 * i + 1] = ok ? (double)cwy : -999.999 * 4;
}

extern "C" int sgta_post_process(const float* scores, const long long* cts, const float* cts_wreg, const float* regs,
                                 const float* tracking, const float* trans_inv, float out_thresh,
                                 unsigned char* valid, float* ct, float* ct_wreg, float* reg, float* trk,
                                 double* final_kps, int B, int K, hipStream_t st) {
  if (!scores || !cts || !cts_wreg || !regs || !trans_inv || !valid || !ct || !ct_wreg || !reg || !final_kps ||
      B <= 0 || K <= 0 || (tracking && !trk))
    return SGTA_EBADARG;
  const int n = B * K;
  hipLaunchKernelGGL(post_process_kernel, dim3(cdiv(n, 64)), dim3(64), 0, st, scores, cts, cts_wreg, regs, tracking,
                     trans_inv, out_thresh, valid, ct, ct_wreg, reg, trk, final_kps, n, K);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// ---- SGTADetector.pre_process (sgta_detector.py:368-399): cv2.warpAffine(image, trans_input, (inp_w, inp_h), INTER_LINEAR,
// border constant 0) followed by normalize_img ((img / 255 - mean) / std in float32, :401-402) and HWC -> CHW.
// The warp restates OpenCV's published 8-bit algorithm (imgwarp.cpp WarpAffineInvoker + remapBilinear): the inverse
// matrix is applied in double, source coordinates are fixed-point with 10 fractional bits (rounded to nearest, ties to
// even) truncated to 1/32 pixel, the four weights are (32-fx)(32-fy)*32 ... (sum 2^15) and the result is
// (sum + 2^14) >> 15.  OpenCV is not in the image: UNPINNED against the library, bit-exact against oracle/prepost.py.
__global__ __launch_bounds__(256) void pre_process_kernel(const unsigned char* __restrict__ img, const double* __restrict__ inv_m,
                                                          const float* __restrict__ mean, const float* __restrict__ stdv,
                                                          float* __restrict__ out, int Hr, int Wr, int Ho, int Wo, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % Wo);
    long t = i / Wo;
    const int y = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const double* m = inv_m + b * 6;
    const int adelta = __double2int_rn(__dmul_rn(__dmul_rn(m[0], (double)x), 1024.0));
    const int bdelta = __double2int_rn(__dmul_rn(__dmul_rn(m[3], (double)x), 1024.0));
    const int X0 = __double2int_rn(__dmul_rn(__dadd_rn(__dmul_rn(m[1], (double)y), m[2]), 1024.0)) + 16;
    const int Y0 = __double2int_rn(__dmul_rn(__dadd_rn(__dmul_rn(m[4], (double)y), m[5]), 1024.0)) + 16;
    const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    int sx = X >> 5, sy = Y >> 5;
    sx = min(max(sx, -32768), 32767);             // saturate_cast<short>
    sy = min(max(sy, -32768), 32767);
    const int fx = X & 31, fy = Y & 31;
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const unsigned char* src = img + (long)b * Hr * Wr * 3;
    const bool x0 = (unsigned)sx < (unsigned)Wr, x1 = (unsigned)(sx + 1) < (unsigned)Wr;
    const bool y0 = (unsigned)sy < (unsigned)Hr, y1 = (unsigned)(sy + 1) < (unsigned)Hr;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int s = 0;
      if (y0 && x0) s += src[((long)sy * Wr + sx) * 3 + c] * w00;
      if (y0 && x1) s += src[((long)sy * Wr + sx + 1) * 3 + c] * w01;
      if (y1 && x0) s += src[((long)(sy + 1) * Wr + sx) * 3 + c] * w10;
      if (y1 && x1) s += src[((long)(sy + 1) * Wr + sx + 1) * 3 + c] * w11;
      const int v = (s + (1 << 14)) >> 15;
      const float q = __fdiv_rn((float)v, 255.0f);
      out[(((long)b * 3 + c) * Ho + y) * Wo + x] = __fdiv_rn(__fsub_rn(q, mean[c]), stdv[c]);
    }
  }
}

extern "C" int sgta_pre_process(const unsigned char* images, const double* inv_trans, const float* mean, const float* stdv,
                                float* out, int B, int Hr, int Wr, int Ho, int Wo, hipStream_t st) {
  if (!images || !inv_trans || !mean || !stdv || !out || B <= 0 || Hr <= 0 || Wr <= 0 || Ho <= 0 || Wo <= 0) return SGTA_EBADARG;
  const long total = (long)B * Ho * Wo;
  const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(pre_process_kernel, dim3(grid), dim3(256), 0, st, images, inv_trans, mean, stdv, out, Hr, Wr, Ho, Wo, total);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}
