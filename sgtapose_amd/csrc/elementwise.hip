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
// out[oy] gathers the (at most) 2 input rows iy with oy = iy*f - f/2 + ky, 0 <= ky < 2f.
template <int F, int V>
__global__ __launch_bounds__(256) void upsample_add_kernel(const float* __restrict__ x, long x_bs,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ skip, long s_bs,
                                                           float* __restrict__ y, long y_bs, int C,
                                                           int H, int W, long total4) {
  // one thread = V consecutive output pixels of a row (V = 4: 16-byte skip load / output store)
  constexpr int KS = 2 * F, P = F / 2;
  const int Ho = H * F, Wo = W * F, Wq = Wo / V;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    int oxq = (int)(i % Wq);
    long t = i / Wq;
    int oy = (int)(t % Ho);
    t /= Ho;
    int c = (int)(t % C);
    int b = (int)(t / C);
    const float* xp = x + b * x_bs + (long)c * H * W;
    const float* wp = w + (long)c * KS * KS;
    // candidate input rows: iy1 = floor((oy + P) / F) (ky1 in [0,F)), iy0 = iy1 - 1 (ky0 = ky1 + F)
    const int iy1 = (oy + P) / F, ky1 = oy + P - iy1 * F;
    const int iy0 = iy1 - 1, ky0 = ky1 + F;
    const bool r0 = iy0 >= 0 && iy0 < H, r1 = iy1 >= 0 && iy1 < H;
    const long o = ((long)c * Ho + oy) * Wo + oxq * V;
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int ox = oxq * V + e;
      const int ix1 = (ox + P) / F, kx1 = ox + P - ix1 * F;
      const int ix0 = ix1 - 1, kx0 = kx1 + F;
      const bool c0 = ix0 >= 0 && ix0 < W, c1 = ix1 >= 0 && ix1 < W;
      float a = 0.0f;
      if (r0 && c0) a = fmaf(xp[iy0 * W + ix0], wp[ky0 * KS + kx0], a);
      if (r0 && c1) a = fmaf(xp[iy0 * W + ix1], wp[ky0 * KS + kx1], a);
      if (r1 && c0) a = fmaf(xp[iy1 * W + ix0], wp[ky1 * KS + kx0], a);
      if (r1 && c1) a = fmaf(xp[iy1 * W + ix1], wp[ky1 * KS + kx1], a);
      acc[e] = a;
    }
    if (V == 4) {
      f32x4 out = skip ? *reinterpret_cast<const f32x4*>(skip + b * s_bs + o) : f32x4{0, 0, 0, 0};
      out += f32x4{acc[0], acc[1], acc[2], acc[3]};
      *reinterpret_cast<f32x4*>(y + b * y_bs + o) = out;
    } else {
      y[b * y_bs + o] = acc[0] + (skip ? skip[b * s_bs + o] : 0.0f);
    }
  }
}

extern "C" int sgta_upsample_add_f32(const float* x, long x_bs, const float* w, const float* skip,
                                     long s_bs, float* y, long y_bs, int B, int C, int H, int W,
                                     int f, hipStream_t st) {
  if (!x || !w || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return SGTA_EBADARG;
  const int v = ((W * f) % 4 == 0 && (x_bs % 4) == 0 && (s_bs % 4) == 0 && (y_bs % 4) == 0) ? 4 : 1;
  long total = (long)B * C * H * f * W * f / v;
  int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
#define SGTA_UP(F, V) hipLaunchKernelGGL((upsample_add_kernel<F, V>), dim3(grid), dim3(256), 0, st, x, x_bs, w, skip, s_bs, y, y_bs, C, H, W, total)
  if (f == 2 && v == 4) SGTA_UP(2, 4);
  else if (f == 2) SGTA_UP(2, 1);
  else if (f == 4 && v == 4) SGTA_UP(4, 4);
  else if (f == 4) SGTA_UP(4, 1);
  else return SGTA_EBADARG;
#undef SGTA_UP
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
