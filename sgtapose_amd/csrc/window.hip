// Keypoint-window bookkeeping of the temporal cross-attention (gfx950):
//   * prior arg-max            reference get_topk_index            dla.py:898-913  (K = 1)
//   * window pixel ids          reference get_topk_features_scale   dla.py:915-968  (fp32 coordinate
//     math, clamp of BOTH axes to H-1, truncation, x-major token order)
//   * token gather from NCHW    (the reference permutes the whole map to NHWC first, dla.py:961-966)
//   * ordered scatter-back      reference substitute_topk_features_scale dla.py:1012-1014: with
//     duplicate destination pixels the LAST token in flattened order wins (serial CPU semantics).
#include "common.h"

// one work-group per (which, b, class) map: arg-max with lowest-index tie break (torch.topk CPU).
__global__ __launch_bounds__(256) void prior_argmax_kernel(const float* __restrict__ pre,
                                                           const float* __restrict__ cur,
                                                           float* __restrict__ idx_xy, int BK, int hw,
                                                           int w) {
  const int map = blockIdx.x;                    // [2][B*K]
  const float* src = (map < BK ? pre : cur) + (long)(map % BK) * hw;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < hw; i += 256) {
    float v = src[i];
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  }
  // NaN-free priors assumed (rendered Gaussians); -inf start makes an all-zero map pick index 0
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(best, o);
    int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  __shared__ float sv[4];
  __shared__ int si[4];
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k)
      if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
    if (bi == 0x7fffffff) bi = 0;
    idx_xy[2 * map + 0] = (float)(bi % w);
    idx_xy[2 * map + 1] = (float)(bi / w);
  }
}

extern "C" int sgta_prior_argmax(const float* pre_cls, const float* cur_cls, float* idx_xy, int B,
                                 int K, int h, int w, hipStream_t st) {
  if (!pre_cls || !cur_cls || !idx_xy || B <= 0 || K <= 0 || h <= 0 || w <= 0) return SGTA_EBADARG;
  hipLaunchKernelGGL(prior_argmax_kernel, dim3(2 * B * K), dim3(256), 0, st, pre_cls, cur_cls, idx_xy, B * K, h * w, w);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// S work-groups per (which, b): window ids of one pyramid level + "winner" flags of the scatter.
// fid[which][b][t], t = kp*w*w + dxi*w + dyi ; win[b][t] = no later token targets the same pixel.  Every work-group
// computes all n ids (cheap) and takes every S-th 256-token strip of the O(n^2) winner search: level 0 has n = 1183 tokens, and
// one work-group per sample left 32 CUs busy for 106 us.
__global__ __launch_bounds__(256) void window_index_kernel(const float* __restrict__ idx_xy,
                                                           long long* __restrict__ fid,
                                                           unsigned char* __restrict__ win, int B,
                                                           int K, float scale, int r, int H, int W, int S) {
  extern __shared__ int sfid[];
  const int slice = blockIdx.x % S, wb = blockIdx.x / S;
  const int which = wb / B, b = wb % B;
  if (which == 0 && slice) return;
  const int ww = 2 * r + 1, n = K * ww * ww;
  const float* src = idx_xy + ((long)which * B + b) * K * 2;
  const float hi = (float)(H - 1);
  for (int t = threadIdx.x; t < n; t += 256) {
    int kp = t / (ww * ww), rem = t - kp * ww * ww;
    int dxi = rem / ww, dyi = rem - dxi * ww;
    // fp32, separately rounded mul/add exactly as torch evaluates idx*scale + off (dla.py:943-946)
    float x = __fadd_rn(__fmul_rn(src[2 * kp], scale), (float)(dxi - r));
    float y = __fadd_rn(__fmul_rn(src[2 * kp + 1], scale), (float)(dyi - r));
    x = fminf(fmaxf(x, 0.0f), hi);
    y = fminf(fmaxf(y, 0.0f), hi);                                // both axes clamp to H-1 (dla.py:948-950)
    float f = __fadd_rn(__fmul_rn(y, (float)W), x);
    int id = (int)f;                                              // .type(torch.long): truncation
    sfid[t] = id;
    if (slice == 0) fid[((long)which * B + b) * n + t] = id;
  }
  if (which == 0) return;                                         // winners only for the current frame
  __syncthreads();
  for (int t = slice * 256 + threadIdx.x; t < n; t += S * 256) {
    int me = sfid[t];
    bool w1 = true;
    for (int u = t + 1; u < n; ++u) w1 &= (sfid[u] != me);
    win[(long)b * n + t] = w1;
  }
}

extern "C" int sgta_window_index(const float* idx_xy, long long* fid, unsigned char* win, int B, int K,
                                 float scale, int ks, int H, int W, hipStream_t st) {
  if (!idx_xy || !fid || !win || B <= 0 || K <= 0 || ks < 1 || H <= 0 || W <= 0) return SGTA_EBADARG;
  int r = ks / 2, ww = 2 * r + 1, n = K * ww * ww;
  const int S = n > 512 ? 8 : (n > 256 ? 2 : 1);
  hipLaunchKernelGGL(window_index_kernel, dim3(2 * B * S), dim3(256), n * sizeof(int), st, idx_xy, fid, win, B, K, scale, r, H, W, S);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// tokens[b][t][c] = feats[b][c][fid[b][t]]   (one thread per (b, t, c); c fastest -> coalesced writes)
__global__ __launch_bounds__(256) void window_gather_kernel(const float* __restrict__ feats, long f_bs,
                                                            const long long* __restrict__ fid,
                                                            float* __restrict__ tok, int C, int HW,
                                                            int n, int ld, int col0, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int c = (int)(i % C);
    long bt = i / C;
    int b = (int)(bt / n);
    long p = fid[bt];
    tok[bt * ld + col0 + c] = feats[b * f_bs + (long)c * HW + p];
  }
}

extern "C" int sgta_window_gather(const float* feats, long f_bs, const long long* fid, float* tok,
                                  int B, int C, int H, int W, int n, int ld, int col0, hipStream_t st) {
  if (!feats || !fid || !tok || B <= 0 || C <= 0 || n <= 0 || ld < col0 + C) return SGTA_EBADARG;
  long total = (long)B * n * C;
  int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(window_gather_kernel, dim3(grid), dim3(256), 0, st, feats, f_bs, fid, tok, C, H * W, n, ld, col0, total);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// feats[b][c][fid[b][t]] = tok[b][t][c]  for winner tokens only (deterministic last-wins).
__global__ __launch_bounds__(256) void window_scatter_kernel(const float* __restrict__ tok,
                                                             const long long* __restrict__ fid,
                                                             const unsigned char* __restrict__ win,
                                                             float* __restrict__ feats, long f_bs, int C,
                                                             int HW, int n, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int c = (int)(i % C);
    long bt = i / C;
    if (!win[bt]) continue;
    int b = (int)(bt / n);
    feats[b * f_bs + (long)c * HW + fid[bt]] = tok[i];
  }
}

extern "C" int sgta_window_scatter(const float* tok, const long long* fid, const unsigned char* win,
                                   float* feats, long f_bs, int B, int C, int H, int W, int n,
                                   hipStream_t st) {
  if (!tok || !fid || !win || !feats || B <= 0 || C <= 0 || n <= 0) return SGTA_EBADARG;
  long total = (long)B * n * C;
  int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(window_scatter_kernel, dim3(grid), dim3(256), 0, st, tok, fid, win, feats, f_bs, C, H * W, n, total);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}
