// Heat-map decode on the device (gfx950): the reference copies each map to the host and runs
// scipy + python loops (peaks_from_belief_maps, sgtapose/image_proc.py:1032-1143; _peaks_info,
// sgtapose/lib/model/utils.py:207-284; dream_generic_decode, sgtapose/lib/model/decode.py:184-313).
//   pass 1/2  separable Gaussian (sigma 3, 25 taps, 'reflect'), evaluated like scipy's
//             NI_Correlate1D symmetric branch: double accumulation in the order
//             c*w0, then (a[-j]+a[+j])*w[j] from the far taps inwards, float32 storage after each axis,
//             un-fused mul/add (scipy's x86-64 build has no FMA contraction) -> bit-exact maps.
//   pass 3    4-neighbour local maxima above 0.01 on the filtered map, 5x5 weighted centroid (+0.4395)
//             on the RAW map in double, the reference's multi-peak rule (stable sort by centroid-y
//             DESC, keep the first iff score gap >= 0.25 in float32), int() truncation, score lookup,
//             and the CenterNet gather of the reg / tracking heads at the peak index.
// Also: SoftArgmaxPavlo (sgtapose/spatial_softmax.py:24-95), the spatial soft-argmax named by the
// north-star (dead in the reference's live decode, decode.py:197).
#include "common.h"

#define GRADIUS 12

__device__ __forceinline__ int reflect_idx(int i, int L) {
  // scipy 'reflect' (d c b a | a b c d | d c b a), any offset
  if (L == 1) return 0;
  int p = 2 * L;
  i %= p;
  if (i < 0) i += p;
  return i < L ? i : p - 1 - i;
}

template <int AXIS>
__global__ __launch_bounds__(256) void gauss1d_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                      const double* __restrict__ w, int H, int W,
                                                      long total) {
  double wr[GRADIUS + 1];
#pragma unroll
  for (int j = 0; j <= GRADIUS; ++j) wr[j] = w[j];            // w[0] = far tap ... w[12] = centre
  const int L = AXIS == 0 ? H : W;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int x = (int)(i % W);
    long t = i / W;
    int y = (int)(t % H);
    const float* base = in + (t / H) * (long)H * W;
    const int c = AXIS == 0 ? y : x;
    auto at = [&](int q) -> double {
      int r = reflect_idx(q, L);
      return (double)(AXIS == 0 ? base[(long)r * W + x] : base[(long)y * W + r]);
    };
    double acc = __dmul_rn(at(c), wr[GRADIUS]);
#pragma unroll
    for (int j = -GRADIUS; j < 0; ++j)
      acc = __dadd_rn(acc, __dmul_rn(__dadd_rn(at(c + j), at(c - j)), wr[GRADIUS + j]));
    out[i] = (float)acc;
  }
}

struct Cand {                 // a peak candidate, ordered by (cy DESC, flat index ASC)
  double cy, cx;
  float score;
  int idx;                    // flat pixel index; -1 = empty
};

__device__ __forceinline__ bool cand_before(const Cand& a, const Cand& b) {
  if (b.idx < 0) return a.idx >= 0;
  if (a.idx < 0) return false;
  return a.cy > b.cy || (a.cy == b.cy && a.idx < b.idx);
}

__device__ __forceinline__ void cand_insert(Cand& c0, Cand& c1, const Cand& n) {
  if (cand_before(n, c0)) { c1 = c0; c0 = n; }
  else if (cand_before(n, c1)) { c1 = n; }
}

// one work-group per (b, class) map
__global__ __launch_bounds__(256) void peaks_select_kernel(
    const float* __restrict__ hm, const float* __restrict__ gf, const float* __restrict__ reg,
    const float* __restrict__ trk, float* __restrict__ scores, long long* __restrict__ xs,
    long long* __restrict__ ys, float* __restrict__ regs, float* __restrict__ trks,
    int* __restrict__ npeaks, int K, int H, int W) {
  const int map = blockIdx.x, b = map / K;
  const int HW = H * W;
  const float* ori = hm + (long)map * HW;
  const float* g = gf + (long)map * HW;
  Cand c0{0, 0, 0, -1}, c1{0, 0, 0, -1};
  int cnt = 0;
  for (int i = threadIdx.x; i < HW; i += 256) {
    int y = i / W, x = i - y * W;
    float v = g[i];
    float up = y > 0 ? g[i - W] : 0.0f, dn = y < H - 1 ? g[i + W] : 0.0f;
    float lf = x > 0 ? g[i - 1] : 0.0f, rt = x < W - 1 ? g[i + 1] : 0.0f;
    if (!(v >= up && v >= dn && v >= lf && v >= rt && v > 0.01f)) continue;
    ++cnt;
    double sw = 0.0, sx = 0.0, sy = 0.0;
    for (int dx = -2; dx <= 2; ++dx)              // numpy flattening: dx is the slow index
      for (int dy = -2; dy <= 2; ++dy) {
        int yy = y + dy, xx = x + dx;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        double wv = (double)ori[yy * W + xx];
        sw += wv;
        sx += __dmul_rn((double)xx, wv);
        sy += __dmul_rn((double)yy, wv);
      }
    Cand n;
    if (sw == 0.0) { n.cx = x + 0.4395; n.cy = y + 0.4395; }
    else { n.cx = sx / sw + 0.4395; n.cy = sy / sw + 0.4395; }
    n.score = ori[i];
    n.idx = i;
    cand_insert(c0, c1, n);
  }
  __shared__ Cand s0[256], s1[256];
  __shared__ int scnt[256];
  s0[threadIdx.x] = c0; s1[threadIdx.x] = c1; scnt[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      Cand a0 = s0[threadIdx.x], a1 = s1[threadIdx.x];
      cand_insert(a0, a1, s0[threadIdx.x + o]);
      cand_insert(a0, a1, s1[threadIdx.x + o]);
      s0[threadIdx.x] = a0; s1[threadIdx.x] = a1;
      scnt[threadIdx.x] += scnt[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int n = scnt[0];
    const Cand best = s0[0], second = s1[0];
    bool have = (n == 1) || (n > 1 && (best.score - second.score) >= 0.25f);
    long long xi = 0, yi = 0;
    float sc = -1.0f;
    if (have) {
      xi = (long long)best.cx;                     // int(): truncation
      yi = (long long)best.cy;
      sc = ori[yi * W + xi];
    }
    scores[map] = sc; xs[map] = xi; ys[map] = yi;
    if (npeaks) npeaks[map] = n;
    const long ind = yi * W + xi;
    for (int c = 0; c < 2; ++c) {
      if (reg) regs[map * 2 + c] = reg[((long)b * 2 + c) * HW + ind];
      if (trk) trks[map * 2 + c] = trk[((long)b * 2 + c) * HW + ind];
    }
  }
}

extern "C" int sgta_peaks_decode(const float* hm, const float* reg, const float* tracking,
                                 const double* gauss_w13, float* ws_tmp, float* ws_filt, float* scores,
                                 long long* xs, long long* ys, float* regs, float* trks, int* npeaks,
                                 int B, int K, int H, int W, hipStream_t st) {
  if (!hm || !gauss_w13 || !ws_tmp || !ws_filt || !scores || !xs || !ys) return SGTA_EBADARG;
  if (B <= 0 || K <= 0 || H <= 0 || W <= 0 || (reg && !regs) || (tracking && !trks)) return SGTA_EBADARG;
  long total = (long)B * K * H * W;
  int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(gauss1d_kernel<0>, dim3(grid), dim3(256), 0, st, hm, ws_tmp, gauss_w13, H, W, total);
  hipLaunchKernelGGL(gauss1d_kernel<1>, dim3(grid), dim3(256), 0, st, (const float*)ws_tmp, ws_filt, gauss_w13, H, W, total);
  hipLaunchKernelGGL(peaks_select_kernel, dim3(B * K), dim3(256), 0, st, hm, (const float*)ws_filt, reg, tracking,
                     scores, xs, ys, regs, trks, npeaks, K, H, W);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// ---- SoftArgmaxPavlo: AvgPool2d(7,1,3) (zero pad, /49) -> -max -> exp(beta*x) -> /(sum+1e-8) -> E[x],E[y]
__device__ __forceinline__ float pool7(const float* m, int y, int x, int H, int W) {
  float s = 0.0f;
  for (int dy = -3; dy <= 3; ++dy) {
    int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
    for (int dx = -3; dx <= 3; ++dx) {
      int xx = x + dx;
      if (xx >= 0 && xx < W) s += m[yy * W + xx];
    }
  }
  return s / 49.0f;
}

__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max) {
  for (int o = 32; o > 0; o >>= 1) {
    float t = __shfl_xor(v, o);
    v = is_max ? fmaxf(v, t) : v + t;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sh[0];
  for (int k = 1; k < 4; ++k) r = is_max ? fmaxf(r, sh[k]) : r + sh[k];
  return r;
}

__global__ __launch_bounds__(256) void softargmax_kernel(const float* __restrict__ hm, float* __restrict__ out,
                                                         int H, int W, float beta, float size_mult) {
  __shared__ float sh[4];
  const float* m = hm + (long)blockIdx.x * H * W;
  const int HW = H * W;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < HW; i += 256) mx = fmaxf(mx, pool7(m, i / W, i % W, H, W));
  mx = block_reduce(mx, sh, true);
  float se = 0.0f, sx = 0.0f, sy = 0.0f;
  for (int i = threadIdx.x; i < HW; i += 256) {
    int y = i / W, x = i - y * W;
    float e = expf(beta * (pool7(m, y, x, H, W) - mx));
    se += e;
    sx += e * ((float)x * size_mult);
    sy += e * ((float)y * size_mult);
  }
  se = block_reduce(se, sh, false);
  sx = block_reduce(sx, sh, false);
  sy = block_reduce(sy, sh, false);
  if (threadIdx.x == 0) {
    float d = se + 1e-8f;
    out[2 * blockIdx.x + 0] = sx / d;
    out[2 * blockIdx.x + 1] = sy / d;
  }
}

extern "C" int sgta_softargmax(const float* hm, float* out_xy, int B, int K, int H, int W, float beta,
                               float size_mult, hipStream_t st) {
  if (!hm || !out_xy || B <= 0 || K <= 0 || H <= 0 || W <= 0) return SGTA_EBADARG;
  hipLaunchKernelGGL(softargmax_kernel, dim3(B * K), dim3(256), 0, st, hm, out_xy, H, W, beta, size_mult);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}
