// HBM-bound element-wise pieces of the hot path (gfx950): 2x2 max-pool (Tree.downsample,
// reference dla.py:209-210, :220), depthwise transposed-conv up-sampling fused with the skip add
// (IDAUp: ``layers[i] = up(proj(layers[i]))`` then ``node(layers[i] + layers[i-1])``, dla.py:561-577),
// and the in-place sigmoid of the heat-map head (sgta_detector.py:854-862).
#include "common.h"

// ---- MaxPool2d(2, stride 2), floor mode -----------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ x, long x_bs,
                                                       float* __restrict__ y, long y_bs, int C,
                                                       int H, int W, int Ho, int Wo, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int ox = (int)(i % Wo);
    long t = i / Wo;
    int oy = (int)(t % Ho);
    t /= Ho;
    int c = (int)(t % C);
    int b = (int)(t / C);
    const float* p = x + b * x_bs + ((long)c * H + 2 * oy) * W + 2 * ox;
    float2 r0 = *reinterpret_cast<const float2*>(p);
    float2 r1 = *reinterpret_cast<const float2*>(p + W);
    y[b * y_bs + ((long)c * Ho + oy) * Wo + ox] = fmaxf(fmaxf(r0.x, r0.y), fmaxf(r1.x, r1.y));
  }
}

extern "C" int sgta_maxpool2_f32(const float* x, long x_bs, float* y, long y_bs, int B, int C, int H,
                                 int W, hipStream_t st) {
  if (!x || !y || B <= 0 || C <= 0 || H < 2 || W < 2 || (W & 1)) return SGTA_EBADARG;
  int Ho = H / 2, Wo = W / 2;
  long total = (long)B * C * Ho * Wo;
  int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(maxpool2_kernel, dim3(grid), dim3(256), 0, st, x, x_bs, y, y_bs, C, H, W, Ho, Wo, total);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// ---- y = ConvTranspose2d(C, C, 2f, stride f, pad f/2, groups=C, no bias)(x) + skip ----------------
// out[oy][ox] gathers the (at most) 2 x 2 inputs (iy, ix) with oy = iy*f - f/2 + ky, 0 <= ky < 2f (same for x).
// Fast path (Wo % 4 == 0, 16-byte aligned planes): one thread = 4 consecutive output pixels of one (b, c) plane
// (blockIdx.y): they share 4/f + 2 input columns per row, the kernel-column pattern is a compile-time constant
// and the two kernel rows are one or two 16-byte loads -- 8 / 6 input loads per 4 outputs instead of 16.
template <int F>
__global__ __launch_bounds__(256) void upsample_add4_kernel(const float* __restrict__ x, long x_bs,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ skip, long s_bs,
                                                            float* __restrict__ y, long y_bs, int C,
                                                            int H, int W, long total) {
  constexpr int KS = 2 * F, P = F / 2, NIN = 4 / F + 2;
  const int Ho = H * F, Wo = W * F, Wq = Wo / 4;
  // flattened (image, channel, row, pixel quad): one grid row per (image, channel) left the small maps with
  // thousands of nearly empty work-groups (256 channels of 30x40: 16k work-groups, 206 us for 39 MB)
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= total) return;
  const int per = Ho * Wq;
  const int bc = (int)(idx / per), rem = (int)(idx - (long)bc * per);
  const int b = bc / C, c = bc - b * C;
  const int oy = rem / Wq, oxq = rem - oy * Wq;
  const float* xp = x + b * x_bs + (long)c * H * W;
  const float* wp = w + (long)c * KS * KS;
  // candidate input rows: iy1 = floor((oy + P) / F) (ky1 in [0,F)), iy0 = iy1 - 1 (ky0 = ky1 + F)
  const int iy1 = (oy + P) / F, ky1 = oy + P - iy1 * F;
  const int iy0 = iy1 - 1, ky0 = ky1 + F;
  const int bx = (4 * oxq) / F - 1;                  // first of the NIN input columns
  float in0[NIN], in1[NIN], w0[KS], w1[KS];
  const bool r0 = iy0 >= 0, r1 = iy1 < H;           // iy0 < H and iy1 >= 0 always hold
#pragma unroll
  for (int j = 0; j < NIN; ++j) {
    const int ix = bx + j;
    const bool cok = ix >= 0 && ix < W;
    in0[j] = (r0 && cok) ? xp[iy0 * W + ix] : 0.0f;
    in1[j] = (r1 && cok) ? xp[iy1 * W + ix] : 0.0f;
  }
#pragma unroll
  for (int q = 0; q < KS / 4; ++q) {
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(wp + ky0 * KS + 4 * q);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(wp + ky1 * KS + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) { w0[4 * q + e] = a0[e]; w1[4 * q + e] = a1[e]; }
  }
  const long o = ((long)c * Ho + oy) * Wo + oxq * 4;
  f32x4 out = skip ? *reinterpret_cast<const f32x4*>(skip + b * s_bs + o) : f32x4{0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    constexpr int dummy = 0; (void)dummy;
    const int j1 = (e + P) / F + 1, j0 = j1 - 1;      // compile-time after unrolling
    const int kx1 = (e + P) % F, kx0 = kx1 + F;
    float a = in0[j0] * w0[kx0];
    a = fmaf(in0[j1], w0[kx1], a);
    a = fmaf(in1[j0], w1[kx0], a);
    a = fmaf(in1[j1], w1[kx1], a);
    out[e] += a;
  }
  *reinterpret_cast<f32x4*>(y + b * y_bs + o) = out;
}

// generic path: one thread = one output pixel
template <int F>
__global__ __launch_bounds__(256) void upsample_add_kernel(const float* __restrict__ x, long x_bs,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ skip, long s_bs,
                                                           float* __restrict__ y, long y_bs, int C,
                                                           int H, int W, long total) {
  constexpr int KS = 2 * F, P = F / 2;
  const int Ho = H * F, Wo = W * F;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int ox = (int)(i % Wo);
    long t = i / Wo;
    int oy = (int)(t % Ho);
    t /= Ho;
    int c = (int)(t % C);
    int b = (int)(t / C);
    const float* xp = x + b * x_bs + (long)c * H * W;
    const float* wp = w + (long)c * KS * KS;
    const int iy1 = (oy + P) / F, ky1 = oy + P - iy1 * F;
    const int iy0 = iy1 - 1, ky0 = ky1 + F;
    const bool r0 = iy0 >= 0 && iy0 < H, r1 = iy1 >= 0 && iy1 < H;
    const int ix1 = (ox + P) / F, kx1 = ox + P - ix1 * F;
    const int ix0 = ix1 - 1, kx0 = kx1 + F;
    const bool c0 = ix0 >= 0 && ix0 < W, c1 = ix1 >= 0 && ix1 < W;
    float a = 0.0f;
    if (r0 && c0) a = fmaf(xp[iy0 * W + ix0], wp[ky0 * KS + kx0], a);
    if (r0 && c1) a = fmaf(xp[iy0 * W + ix1], wp[ky0 * KS + kx1], a);
    if (r1 && c0) a = fmaf(xp[iy1 * W + ix0], wp[ky1 * KS + kx0], a);
    if (r1 && c1) a = fmaf(xp[iy1 * W + ix1], wp[ky1 * KS + kx1], a);
    const long o = ((long)c * Ho + oy) * Wo + ox;
    y[b * y_bs + o] = a + (skip ? skip[b * s_bs + o] : 0.0f);
  }
}

extern "C" int sgta_upsample_add_f32(const float* x, long x_bs, const float* w, const float* skip,
                                     long s_bs, float* y, long y_bs, int B, int C, int H, int W,
                                     int f, hipStream_t st) {
  if (!x || !w || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return SGTA_EBADARG;
  if (f != 2 && f != 4) return SGTA_EBADARG;
  const bool v4 = (W * f) % 4 == 0 && (s_bs % 4) == 0 && (y_bs % 4) == 0 && ((uintptr_t)y & 15) == 0 &&
                  (!skip || ((uintptr_t)skip & 15) == 0) && ((uintptr_t)w & 15) == 0;
  const long quads = (long)B * C * H * f * (W * f / 4);
  if (v4 && (quads + 255) / 256 <= 0x7fffffffL) {
    const dim3 grid((unsigned)((quads + 255) / 256));
    if (f == 2) hipLaunchKernelGGL(upsample_add4_kernel<2>, grid, dim3(256), 0, st, x, x_bs, w, skip, s_bs, y, y_bs, C, H, W, quads);
    else hipLaunchKernelGGL(upsample_add4_kernel<4>, grid, dim3(256), 0, st, x, x_bs, w, skip, s_bs, y, y_bs, C, H, W, quads);
  } else {
    const long total = (long)B * C * H * f * W * f;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (f == 2) hipLaunchKernelGGL(upsample_add_kernel<2>, dim3(grid), dim3(256), 0, st, x, x_bs, w, skip, s_bs, y, y_bs, C, H, W, total);
    else hipLaunchKernelGGL(upsample_add_kernel<4>, dim3(grid), dim3(256), 0, st, x, x_bs, w, skip, s_bs, y, y_bs, C, H, W, total);
  }
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// ---- sigmoid (out of place or in place) --------------------------------------------------------------
__global__ __launch_bounds__(256) void sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    y[i] = 1.0f / (1.0f + expf(-x[i]));
}

extern "C" int sgta_sigmoid_f32(const float* x, float* y, long n, hipStream_t st) {
  if (!x || !y || n <= 0) return SGTA_EBADARG;
  int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(sigmoid_kernel, dim3(grid), dim3(256), 0, st, x, y, n);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}
