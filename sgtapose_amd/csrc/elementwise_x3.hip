// Glue of the split-operand ("x3") data path: activations are TWO IEEE-half planes of the blocked channels-last layout
// [N][C/16][H][W][16], value = hi + lo with hi = half(value), lo = half(value - hi) (22 significant bits); ``lo_delta`` =
// elements from a tensor's hi plane to its lo plane.  One thread = 8 channels of one pixel = one 16-byte access per plane.
//   MaxPool2d(2,2)                      (dla.py:218-220)
//   window token gather / ordered scatter between x3 maps and the fp32 token matrices (dla.py:915-968, :1006-1018)
//   fp32 NCHW -> x3 blocked and back      (hand-off between the fp32 kernels and the x3 kernels of the hybrid engine)
#include "common.h"

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x8 x3_load8(const _Float16* p, long lo_delta) {
  const h16x8 a = *reinterpret_cast<const h16x8*>(p), b = *reinterpret_cast<const h16x8*>(p + lo_delta);
  return __builtin_convertvector(a, f32x8) + __builtin_convertvector(b, f32x8);        // exact: 22 bits fit fp32's 24
}

__device__ __forceinline__ void x3_store8(_Float16* p, long lo_delta, f32x8 v) {
  const h16x8 hi = H16<true>::cvt8(v);
  *reinterpret_cast<h16x8*>(p) = hi;
  *reinterpret_cast<h16x8*>(p + lo_delta) = H16<true>::cvt8(v - __builtin_convertvector(hi, f32x8));
}

__global__ __launch_bounds__(256) void maxpool2_x3_kernel(const _Float16* __restrict__ x, long x_bs, long x_lo, _Float16* __restrict__ y,
                                                          long y_bs, long y_lo, int C8, int H, int W, long total) {
  const int Ho = H / 2, Wo = W / 2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int half = (int)(i & 1);
    long p = i >> 1;
    const int xo = (int)(p % Wo);
    p /= Wo;
    const int yo = (int)(p % Ho);
    p /= Ho;
    const int cb = (int)(p % (C8 / 2));
    const long n = p / (C8 / 2);
    const _Float16* s = x + n * x_bs + ((long)cb * H * W + (long)(2 * yo) * W + 2 * xo) * 16 + half * 8;
    const f32x8 a = x3_load8(s, x_lo), b = x3_load8(s + 16, x_lo), d = x3_load8(s + (long)W * 16, x_lo),
                e = x3_load8(s + (long)(W + 1) * 16, x_lo);
    f32x8 m;
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = fmaxf(fmaxf(a[j], b[j]), fmaxf(d[j], e[j]));
    x3_store8(y + n * y_bs + ((long)cb * Ho * Wo + (long)yo * Wo + xo) * 16 + half * 8, y_lo, m);      // exact: m is one of the inputs
  }
}

extern "C" int sgta_maxpool2_x3(const void* x, long x_bs, long x_lo_delta, void* y, long y_bs, long y_lo_delta, int N, int C,
                                int H, int W, hipStream_t st) {
  if (!x || !y || N <= 0 || C <= 0 || C % 16 || x_bs % 16 || y_bs % 16 || H < 2 || W < 2 || x_lo_delta % 8 || y_lo_delta % 8 ||
      x_lo_delta <= 0 || y_lo_delta <= 0)
    return SGTA_EBADARG;
  const long total = (long)N * (H / 2) * (W / 2) * (C / 8);
  const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(maxpool2_x3_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const _Float16*>(x), x_bs, x_lo_delta,
                     reinterpret_cast<_Float16*>(y), y_bs, y_lo_delta, C / 8, H, W, total);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// tok[b*n + t][col0 + c] = value(feats[b][c / 16][fid[b][t]][c % 16])
__global__ __launch_bounds__(256) void window_gather_x3_kernel(const _Float16* __restrict__ feats, long f_bs, long f_lo, long HW,
                                                               const long long* __restrict__ fid, float* __restrict__ tok, int C8,
                                                               int n, int ld, int col0, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C8) * 8;
    const long bt = i / C8;
    const long b = bt / n;
    const f32x8 v = x3_load8(feats + b * f_bs + ((long)(c >> 4) * HW + fid[bt]) * 16 + (c & 15), f_lo);
    float* d = tok + bt * ld + col0 + c;
    *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
}

extern "C" int sgta_window_gather_x3(const void* feats, long f_bs, long f_lo_delta, const long long* fid, float* tok, int B, int C,
                                     int H, int W, int n, int ld, int col0, hipStream_t st) {
  if (!feats || !fid || !tok || B <= 0 || C <= 0 || C % 16 || f_bs % 16 || n <= 0 || ld < col0 + C || ld % 4 || col0 % 4 ||
      f_lo_delta <= 0 || f_lo_delta % 8)
    return SGTA_EBADARG;
  const long total = (long)B * n * (C / 8);
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(window_gather_x3_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const _Float16*>(feats), f_bs,
                     f_lo_delta, (long)H * W, fid, tok, C / 8, n, ld, col0, total);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// feats[b][c / 16][fid[b][t]][c % 16] = split(tok[b*n + t][c]) for winner tokens only (deterministic last-wins)
__global__ __launch_bounds__(256) void window_scatter_x3_kernel(const float* __restrict__ tok, int ldt,
                                                                const long long* __restrict__ fid,
                                                                const unsigned char* __restrict__ win, _Float16* __restrict__ feats,
                                                                long f_bs, long f_lo, long HW, int C8, int n, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C8) * 8;
    const long bt = i / C8;
    if (!win[bt]) continue;
    const long b = bt / n;
    const float* s = tok + bt * ldt + c;
    const f32x4 a = *reinterpret_cast<const f32x4*>(s), d = *reinterpret_cast<const f32x4*>(s + 4);
    x3_store8(feats + b * f_bs + ((long)(c >> 4) * HW + fid[bt]) * 16 + (c & 15), f_lo, f32x8{a[0], a[1], a[2], a[3], d[0], d[1], d[2], d[3]});
  }
}

extern "C" int sgta_window_scatter_x3(const float* tok, int ldt, const long long* fid, const unsigned char* win, void* feats,
                                      long f_bs, long f_lo_delta, int B, int C, int H, int W, int n, hipStream_t st) {
  if (!tok || !fid || !win || !feats || B <= 0 || C <= 0 || C % 16 || f_bs % 16 || ldt % 4 || n <= 0 || f_lo_delta <= 0 ||
      f_lo_delta % 8)
    return SGTA_EBADARG;
  const long total = (long)B * n * (C / 8);
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(window_scatter_x3_kernel, dim3(grid), dim3(256), 0, st, tok, ldt, fid, win, reinterpret_cast<_Float16*>(feats),
                     f_bs, f_lo_delta, (long)H * W, C / 8, n, total);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// fp32 NCHW [N][C][H][W] (x_bs elements per image) <-> x3 blocked.  Thread = (pixel, 16-channel block); a wave covers 64
// consecutive pixels of one block: every fp32 channel plane is read / written in 256-byte runs and each 16-bit plane in one
// 2 KB contiguous run (two 16-byte pieces per thread: whole cache lines, no partial-line writes).
template <bool TO_X3>
__global__ __launch_bounds__(256) void x3_convert_kernel(float* __restrict__ f, long f_bs, _Float16* __restrict__ h, long h_bs, long h_lo,
                                                         int Cb, long HW, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long px = i % HW;
    long p = i / HW;
    const int cb = (int)(p % Cb);
    const long n = p / Cb;
    float* fp = f + n * f_bs + (long)cb * 16 * HW + px;
    _Float16* hp = h + n * h_bs + ((long)cb * HW + px) * 16;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (TO_X3) {
        f32x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fp[(long)(8 * half + j) * HW];
        x3_store8(hp + 8 * half, h_lo, v);
      } else {
        const f32x8 v = x3_load8(hp + 8 * half, h_lo);
#pragma unroll
        for (int j = 0; j < 8; ++j) fp[(long)(8 * half + j) * HW] = v[j];
      }
    }
  }
}

extern "C" int sgta_nchw_to_x3(const float* x, long x_bs, void* y, long y_bs, long y_lo_delta, int N, int C, int H, int W,
                               hipStream_t st) {
  if (!x || !y || N <= 0 || C <= 0 || C % 16 || y_bs % 16 || y_lo_delta <= 0 || y_lo_delta % 8 || ((uintptr_t)y & 15)) return SGTA_EBADARG;
  const long total = (long)N * (C / 16) * H * W;
  const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(x3_convert_kernel<true>, dim3(grid), dim3(256), 0, st, const_cast<float*>(x), x_bs, reinterpret_cast<_Float16*>(y),
                     y_bs, y_lo_delta, C / 16, (long)H * W, total);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

extern "C" int sgta_x3_to_nchw(const void* x, long x_bs, long x_lo_delta, float* y, long y_bs, int N, int C, int H, int W,
                               hipStream_t st) {
  if (!x || !y || N <= 0 || C <= 0 || C % 16 || x_bs % 16 || x_lo_delta <= 0 || x_lo_delta % 8 || ((uintptr_t)x & 15)) return SGTA_EBADARG;
  const long total = (long)N * (C / 16) * H * W;
  const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(x3_convert_kernel<false>, dim3(grid), dim3(256), 0, st, y, y_bs,
                     reinterpret_cast<_Float16*>(const_cast<void*>(x)), x_bs, x_lo_delta, C / 16, (long)H * W, total);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}
