// bf16 data path: the memory-bound glue of the blocked channels-last engine ([N][C/16][H][W][16]; one thread = 8 channels of
// one pixel = one 16-byte access; ``bs`` = elements per image of the underlying buffer, pointers at the view's first block):
// MaxPool2d(2,2) (dla.py:218-220), depthwise ConvTranspose2d up-sampling + skip add (IDAUp, dla.py:571-577),
// window token gather / ordered scatter between bf16 NHWC maps and the fp32 token matrices of the attention stage
// (get_topk_features_scale / substitute_topk_features_scale, dla.py:915-968, :1006-1018).
#include "common.h"

// every kernel is a template on the element format E = H16<false> (bf16) / H16<true> (IEEE half)
template <typename T>
__device__ __forceinline__ f32x8 ld8(const T* p) {
  typedef T x8 __attribute__((ext_vector_type(8)));
  const x8 v = *reinterpret_cast<const x8*>(p);
  return __builtin_convertvector(v, f32x8);
}

template <bool F16>
__global__ __launch_bounds__(256) void maxpool2_bf16_kernel(const typename H16<F16>::T* __restrict__ x, long x_bs, typename H16<F16>::T* __restrict__ y,
                                                            long y_bs, int C8, int H, int W, long total) {
  const int Ho = H / 2, Wo = W / 2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int half = (int)(i & 1);                 // which 8 of the pixel's 16-channel run: adjacent threads -> adjacent 16 bytes
    long p = i >> 1;
    const int xo = (int)(p % Wo);
    p /= Wo;
    const int yo = (int)(p % Ho);
    p /= Ho;
    const int cb = (int)(p % (C8 / 2));
    const long n = p / (C8 / 2);
    const typename H16<F16>::T* s = x + n * x_bs + ((long)cb * H * W + (long)(2 * yo) * W + 2 * xo) * 16 + half * 8;
    const f32x8 a = ld8(s), b = ld8(s + 16), d = ld8(s + (long)W * 16), e = ld8(s + (long)(W + 1) * 16);
    f32x8 m;
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = fmaxf(fmaxf(a[j], b[j]), fmaxf(d[j], e[j]));
    *reinterpret_cast<typename H16<F16>::x8*>(y + n * y_bs + ((long)cb * Ho * Wo + (long)yo * Wo + xo) * 16 + half * 8) = H16<F16>::cvt8(m);
  }
}

template <bool F16>
static int ew_maxpool2(const void* x, long x_bs, void* y, long y_bs, int N, int C, int H, int W, hipStream_t st) {
  typedef typename H16<F16>::T elt;
  if (!x || !y || N <= 0 || C <= 0 || C % 16 || x_bs % 16 || y_bs % 16 || H < 2 || W < 2) return SGTA_EBADARG;
  const long total = (long)N * (H / 2) * (W / 2) * (C / 8);
  const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(maxpool2_bf16_kernel<F16>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const elt*>(x), x_bs,
                     reinterpret_cast<elt*>(y), y_bs, C / 8, H, W, total);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}
extern "C" int sgta_maxpool2_bf16(const void* x, long x_bs, void* y, long y_bs, int N, int C, int H, int W, hipStream_t st) {
  return ew_maxpool2<false>(x, x_bs, y, y_bs, N, C, H, W, st);
}
extern "C" int sgta_maxpool2_f16(const void* x, long x_bs, void* y, long y_bs, int N, int C, int H, int W, hipStream_t st) {
  return ew_maxpool2<true>(x, x_bs, y, y_bs, N, C, H, W, st);
}

// y[n, oy, ox, c] = skip + sum over the (at most) 2 x 2 input pixels that reach the output pixel of
// x[n, iy, ix, c] * w[ky][kx][c]   (ConvTranspose2d(C, C, 2F, stride F, padding F/2, groups C, bias False))
template <int F, bool F16>
__global__ __launch_bounds__(256) void upsample_add_bf16_kernel(const typename H16<F16>::T* __restrict__ x, long x_bs,
                                                                const float* __restrict__ w,      // [2F][2F][C]
                                                                const typename H16<F16>::T* __restrict__ skip, long s_bs,
                                                                typename H16<F16>::T* __restrict__ y, long y_bs, int C, int H, int W,
                                                                long total) {
  constexpr int KS = 2 * F, P = F / 2;
  const int Ho = H * F, Wo = W * F, Cb = C / 16;
  // grid: x = 256-thread blocks over the (pixel, channel half) pairs of one 16-channel block, y = (image, block): 32-bit
  // index arithmetic only (a flat 64-bit index cost four 64-bit divisions per thread, more than the memory traffic)
  const int cb = blockIdx.y % Cb;
  const long n = blockIdx.y / Cb;
  // the block's 16 channels of all KS x KS taps: LDS reads instead of eight 16-byte global loads per thread (the kernel is
  // bound by vector-memory instruction issue, not by bytes)
  __shared__ __attribute__((aligned(16))) float wl[KS * KS * 16];
  for (int e = threadIdx.x; e < KS * KS * 16; e += 256) wl[e] = w[(long)(e >> 4) * C + cb * 16 + (e & 15)];
  __syncthreads();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < (int)total; i += gridDim.x * 256) {
    const int half = i & 1;
    const int p = i >> 1;
    const int oy = p / Wo, ox = p - oy * Wo;
    const int c = cb * 16 + half * 8;
    const int iy1 = (oy + P) / F, ky1 = oy + P - iy1 * F, iy0 = iy1 - 1, ky0 = ky1 + F;
    const int ix1 = (ox + P) / F, kx1 = ox + P - ix1 * F, ix0 = ix1 - 1, kx0 = kx1 + F;
    const long o = ((long)cb * Ho * Wo + (long)oy * Wo + ox) * 16 + half * 8;
    f32x8 acc;
    if (skip) acc = ld8(skip + n * s_bs + o);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    }
    const int iys[2] = {iy0, iy1}, kys[2] = {ky0, ky1}, ixs[2] = {ix0, ix1}, kxs[2] = {kx0, kx1};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (iys[a] < 0 || iys[a] >= H || ixs[b] < 0 || ixs[b] >= W) continue;
        const f32x8 v = ld8(x + n * x_bs + ((long)cb * H * W + (long)iys[a] * W + ixs[b]) * 16 + half * 8);
        const float* wp = wl + (kys[a] * KS + kxs[b]) * 16 + half * 8;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[j] = __builtin_fmaf(v[j], w0[j], acc[j]);
          acc[4 + j] = __builtin_fmaf(v[4 + j], w1[j], acc[4 + j]);
        }
      }
    *reinterpret_cast<typename H16<F16>::x8*>(y + n * y_bs + o) = H16<F16>::cvt8(acc);
  }
}

// w: f32 [2f][2f][C] (the module's [C,1,2f,2f] weight permuted); skip may be NULL.
template <bool F16>
static int ew_upsample_add(const void* x, long x_bs, const float* w, const void* skip, long s_bs, void* y, long y_bs,
                           int N, int C, int H, int W, int f, hipStream_t st) {
  typedef typename H16<F16>::T elt;
  if (!x || !w || !y || N <= 0 || C <= 0 || C % 16 || x_bs % 16 || y_bs % 16 || (skip && s_bs % 16)) return SGTA_EBADARG;
  if (f != 2 && f != 4) return SGTA_EBADARG;
  const long total = (long)H * f * W * f * 2;                  // (pixel, channel half) pairs of one (image, 16-channel block)
  if (total > 0x7fffffffL || (long)N * (C / 16) > 65535) return SGTA_EBADARG;
  const dim3 grid((unsigned)((total + 255) / 256), (unsigned)(N * (C / 16)));
  auto xx = reinterpret_cast<const elt*>(x);
  auto ss = reinterpret_cast<const elt*>(skip);
  auto yy = reinterpret_cast<elt*>(y);
  if (f == 2) hipLaunchKernelGGL((upsample_add_bf16_kernel<2, F16>), grid, dim3(256), 0, st, xx, x_bs, w, ss, s_bs, yy, y_bs, C, H, W, total);
  else hipLaunchKernelGGL((upsample_add_bf16_kernel<4, F16>), grid, dim3(256), 0, st, xx, x_bs, w, ss, s_bs, yy, y_bs, C, H, W, total);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}
extern "C" int sgta_upsample_add_bf16(const void* x, long x_bs, const float* w, const void* skip, long s_bs, void* y, long y_bs,
                                      int N, int C, int H, int W, int f, hipStream_t st) {
  return ew_upsample_add<false>(x, x_bs, w, skip, s_bs, y, y_bs, N, C, H, W, f, st);
}
extern "C" int sgta_upsample_add_f16(const void* x, long x_bs, const float* w, const void* skip, long s_bs, void* y, long y_bs,
                                     int N, int C, int H, int W, int f, hipStream_t st) {
  return ew_upsample_add<true>(x, x_bs, w, skip, s_bs, y, y_bs, N, C, H, W, f, st);
}

// tok[b*n + t][col0 + c] = float(feats[b][c / 16][fid[b][t]][c % 16])      (fid = pixel index y*W + x inside the image)
template <bool F16>
__global__ __launch_bounds__(256) void window_gather_bf16_kernel(const typename H16<F16>::T* __restrict__ feats, long f_bs, long HW,
                                                                 const long long* __restrict__ fid, float* __restrict__ tok,
                                                                 int C8, int n, int ld, int col0, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C8) * 8;
    const long bt = i / C8;
    const long b = bt / n;
    const f32x8 v = ld8(feats + b * f_bs + ((long)(c >> 4) * HW + fid[bt]) * 16 + (c & 15));
    float* d = tok + bt * ld + col0 + c;
    *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
}

template <bool F16>
static int ew_window_gather(const void* feats, long f_bs, const long long* fid, float* tok, int B, int C, int H,
                            int W, int n, int ld, int col0, hipStream_t st) {
  if (!feats || !fid || !tok || B <= 0 || C <= 0 || C % 16 || f_bs % 16 || n <= 0 || ld < col0 + C || ld % 4 || col0 % 4)
    return SGTA_EBADARG;
  const long total = (long)B * n * (C / 8);
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(window_gather_bf16_kernel<F16>, dim3(grid), dim3(256), 0, st,
                     reinterpret_cast<const typename H16<F16>::T*>(feats), f_bs, (long)H * W, fid, tok, C / 8, n, ld, col0, total);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}
extern "C" int sgta_window_gather_bf16(const void* feats, long f_bs, const long long* fid, float* tok, int B, int C, int H,
                                       int W, int n, int ld, int col0, hipStream_t st) {
  return ew_window_gather<false>(feats, f_bs, fid, tok, B, C, H, W, n, ld, col0, st);
}
extern "C" int sgta_window_gather_f16(const void* feats, long f_bs, const long long* fid, float* tok, int B, int C, int H,
                                      int W, int n, int ld, int col0, hipStream_t st) {
  return ew_window_gather<true>(feats, f_bs, fid, tok, B, C, H, W, n, ld, col0, st);
}

// feats[b][c / 16][fid[b][t]][c % 16] = bf16(tok[b*n + t][c]) for winner tokens only (deterministic last-wins, as sgta_window_scatter)
template <bool F16>
__global__ __launch_bounds__(256) void window_scatter_bf16_kernel(const float* __restrict__ tok, int ldt,
                                                                  const long long* __restrict__ fid,
                                                                  const unsigned char* __restrict__ win,
                                                                  typename H16<F16>::T* __restrict__ feats, long f_bs, long HW, int C8, int n,
                                                                  long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C8) * 8;
    const long bt = i / C8;
    if (!win[bt]) continue;
    const long b = bt / n;
    const float* s = tok + bt * ldt + c;
    const f32x4 a = *reinterpret_cast<const f32x4*>(s), d = *reinterpret_cast<const f32x4*>(s + 4);
    const f32x8 v = {a[0], a[1], a[2], a[3], d[0], d[1], d[2], d[3]};
    *reinterpret_cast<typename H16<F16>::x8*>(feats + b * f_bs + ((long)(c >> 4) * HW + fid[bt]) * 16 + (c & 15)) = H16<F16>::cvt8(v);
  }
}

template <bool F16>
static int ew_window_scatter(const float* tok, int ldt, const long long* fid, const unsigned char* win,
                             void* feats, long f_bs, int B, int C, int H, int W, int n, hipStream_t st) {
  if (!tok || !fid || !win || !feats || B <= 0 || C <= 0 || C % 16 || f_bs % 16 || ldt % 4 || n <= 0) return SGTA_EBADARG;
  const long total = (long)B * n * (C / 8);
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(window_scatter_bf16_kernel<F16>, dim3(grid), dim3(256), 0, st, tok, ldt, fid, win,
                     reinterpret_cast<typename H16<F16>::T*>(feats), f_bs, (long)H * W, C / 8, n, total);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}
extern "C" int sgta_window_scatter_bf16(const float* tok, int ldt, const long long* fid, const unsigned char* win,
                                        void* feats, long f_bs, int B, int C, int H, int W, int n, hipStream_t st) {
  return ew_window_scatter<false>(tok, ldt, fid, win, feats, f_bs, B, C, H, W, n, st);
}
extern "C" int sgta_window_scatter_f16(const float* tok, int ldt, const long long* fid, const unsigned char* win,
                                       void* feats, long f_bs, int B, int C, int H, int W, int n, hipStream_t st) {
  return ew_window_scatter<true>(tok, ldt, fid, win, feats, f_bs, B, C, H, W, n, st);
}
