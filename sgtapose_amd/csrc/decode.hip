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

extern "C" int sgta_peaks_decode_kp(const float* hm, const float* reg, const float* tracking,
                                    const double* gauss_w13, float* ws_tmp, float* ws_filt, float* scores,
                                    long long* xs, long long* ys, float* regs, float* trks, int* npeaks,
                                    long long* cts, float* cts_wreg, float* regs_rel, float* kp, int B, int K, int H,
                                    int W, int three_pass, hipStream_t st);

// shared tail of both selection kernels: the decode dictionary's entries of one map (decode.py:296-312) and the
// [x + reg_x, y + reg_y, score] key-point row that is all-gathered across ranks
__device__ __forceinline__ void emit_map(int map, int b, int HW, int W, long long xi, long long yi, float sc, int n,
                                         const float* reg, const float* trk, float* scores, long long* xs,
                                         long long* ys, float* regs, float* trks, int* npeaks, long long* cts,
                                         float* cts_wreg, float* regs_rel, float* kp) {
  scores[map] = sc; xs[map] = xi; ys[map] = yi;
  if (npeaks) npeaks[map] = n;
  const long ind = yi * W + xi;
  float rg[2] = {0.0f, 0.0f};
  for (int c = 0; c < 2; ++c) {
    if (reg) rg[c] = reg[((long)b * 2 + c) * HW + ind];
    if (regs) regs[map * 2 + c] = rg[c];
    if (trks) trks[map * 2 + c] = trk ? trk[((long)b * 2 + c) * HW + ind] : 0.0f;
  }
  if (cts) { cts[map * 2] = xi; cts[map * 2 + 1] = yi; }
  const float fx = (float)xi, fy = (float)yi;
  // without a 'reg' head the reference adds 0.5 (decode.py:304-306)
  const float wx = __fadd_rn(fx, reg ? rg[0] : 0.5f), wy = __fadd_rn(fy, reg ? rg[1] : 0.5f);
  if (cts_wreg && reg) {
    cts_wreg[map * 2] = wx; cts_wreg[map * 2 + 1] = wy;
    if (regs_rel) { regs_rel[map * 2] = __fsub_rn(wx, fx); regs_rel[map * 2 + 1] = __fsub_rn(wy, fy); }
  }
  if (kp) { kp[map * 3] = wx; kp[map * 3 + 1] = wy; kp[map * 3 + 2] = sc; }
}

// one work-group per (b, class) map
__global__ __launch_bounds__(256) void peaks_select_kernel(
    const float* __restrict__ hm, const float* __restrict__ gf, const float* __restrict__ reg,
    const float* __restrict__ trk, float* __restrict__ scores, long long* __restrict__ xs,
    long long* __restrict__ ys, float* __restrict__ regs, float* __restrict__ trks,
    int* __restrict__ npeaks, long long* __restrict__ cts, float* __restrict__ cts_wreg,
    float* __restrict__ regs_rel, float* __restrict__ kp, int K, int H, int W) {
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
    emit_map(map, b, HW, W, xi, yi, sc, n, reg, trk, scores, xs, ys, regs, trks, npeaks, cts, cts_wreg, regs_rel, kp);
  }
}

// ---- fused form: both Gaussian passes, the local-maximum test and the per-tile candidate selection in ONE kernel ---
// A work-group owns TH rows of one map: it stages the TH + 26 input rows it needs (row indices already reflected),
// runs the vertical pass for TH + 2 rows (one halo row each side for the 4-neighbour test) into LDS as float32 --
// scipy stores float32 after each axis --, extends those rows by 12 reflected columns each side, runs the horizontal
// pass into LDS, writes its TH filtered rows (ws_filt; kept as an output because the filter is pinned bit-exact
// against scipy by the tests), and reduces the tile's local maxima to its two first candidates + their count.
// peaks_merge_kernel then merges the tiles of a map (the first two of a union are among the first two of its parts).
// Arithmetic per output is exactly gauss1d_kernel's; the three-kernel form above stays as the fall-back for maps too
// wide for the LDS tile and as the cross-check in tests.
#define GP_TH 24
struct TileCands { Cand c0, c1; int cnt, pad; };

__global__ __launch_bounds__(256) void gauss_peaks_kernel(const float* __restrict__ hm, float* __restrict__ filt,
                                                         const double* __restrict__ w, TileCands* __restrict__ tc,
                                                         int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  const int map = blockIdx.y, tile = blockIdx.x, y0 = tile * GP_TH;
  const int WP = W + 2 * GRADIUS;
  float* sin = gsm;                                   // [GP_TH + 26][W]   input rows y0 - 13 + k (reflected)
  float* stmp = sin + (GP_TH + 26) * W;               // [GP_TH + 2][WP]   vertical pass, rows y0 - 1 + k
  float* sf = stmp + (GP_TH + 2) * WP;                // [GP_TH + 2][W]    filtered rows y0 - 1 + k
  const float* src = hm + (long)map * H * W;
  double wr[GRADIUS + 1];
#pragma unroll
  for (int j = 0; j <= GRADIUS; ++j) wr[j] = w[j];
  // eight loads in flight per thread (a one-load-per-iteration loop waited a full memory round trip 31 times: 100 of
  // the kernel's 133 us)
  const int nin = (GP_TH + 26) * W;
  for (int e0 = threadIdx.x; e0 < nin; e0 += 8 * 256) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * 256;
      const int k = e / W, x = e - k * W;
      v[u] = e < nin ? src[(long)reflect_idx(y0 - 13 + k, H) * W + x] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + u * 256 < nin) sin[e0 + u * 256] = v[u];
  }
  __syncthreads();
  // vertical pass: output row yy = y0 - 1 + k reads input rows yy - 12 .. yy + 12 = LDS rows k .. k + 24
  for (int e = threadIdx.x; e < (GP_TH + 2) * W; e += 256) {
    const int k = e / W, x = e - k * W;
    const float* col = sin + (k + GRADIUS) * W + x;   // centre row
    double acc = __dmul_rn((double)col[0], wr[GRADIUS]);
#pragma unroll
    for (int j = -GRADIUS; j < 0; ++j)
      acc = __dadd_rn(acc, __dmul_rn(__dadd_rn((double)col[j * W], (double)col[-j * W]), wr[GRADIUS + j]));
    stmp[k * WP + GRADIUS + x] = (float)acc;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < (GP_TH + 2) * 2 * GRADIUS; e += 256) {     // reflected borders of every row
    const int k = e / (2 * GRADIUS), i = e - k * (2 * GRADIUS);
    const int x = i < GRADIUS ? i - GRADIUS : W + (i - GRADIUS);
    stmp[k * WP + GRADIUS + x] = stmp[k * WP + GRADIUS + reflect_idx(x, W)];
  }
  __syncthreads();
  float* out = filt + (long)map * H * W;
  for (int e = threadIdx.x; e < (GP_TH + 2) * W; e += 256) {
    const int k = e / W, x = e - k * W;
    const float* row = stmp + k * WP + GRADIUS + x;
    double acc = __dmul_rn((double)row[0], wr[GRADIUS]);
#pragma unroll
    for (int j = -GRADIUS; j < 0; ++j)
      acc = __dadd_rn(acc, __dmul_rn(__dadd_rn((double)row[j], (double)row[-j]), wr[GRADIUS + j]));
    const float v = (float)acc;
    sf[e] = v;
    const int yy = y0 - 1 + k;
    if (k >= 1 && k <= GP_TH && yy < H) out[(long)yy * W + x] = v;
  }
  __syncthreads();
  // local maxima of this tile's rows (peaks_select_kernel's test and centroid, rows beyond the map count as 0)
  const float* ori = src;
  Cand c0{0, 0, 0, -1}, c1{0, 0, 0, -1};
  int cnt = 0;
  for (int e = threadIdx.x; e < GP_TH * W; e += 256) {
    const int k = e / W + 1, x = e - (k - 1) * W, y = y0 - 1 + k;
    if (y >= H) break;
    const float* g = sf + k * W + x;
    const float v = g[0];
    const float up = y > 0 ? g[-W] : 0.0f, dn = y < H - 1 ? g[W] : 0.0f;
    const float lf = x > 0 ? g[-1] : 0.0f, rt = x < W - 1 ? g[1] : 0.0f;
    if (!(v >= up && v >= dn && v >= lf && v >= rt && v > 0.01f)) continue;
    ++cnt;
    double sw = 0.0, sx = 0.0, sy = 0.0;
    for (int dx = -2; dx <= 2; ++dx)              // numpy flattening: dx is the slow index
      for (int dy = -2; dy <= 2; ++dy) {
        const int yy = y + dy, xx = x + dx;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const double wv = (double)ori[yy * W + xx];
        sw += wv;
        sx += __dmul_rn((double)xx, wv);
        sy += __dmul_rn((double)yy, wv);
      }
    Cand n;
    if (sw == 0.0) { n.cx = x + 0.4395; n.cy = y + 0.4395; }
    else { n.cx = sx / sw + 0.4395; n.cy = sy / sw + 0.4395; }
    n.idx = y * W + x;
    n.score = ori[n.idx];
    cand_insert(c0, c1, n);
  }
  __syncthreads();                                  // the tile arrays are dead: reuse the LDS for the reduction
  Cand* s0 = reinterpret_cast<Cand*>(gsm);
  Cand* s1 = s0 + 256;
  int* scnt = reinterpret_cast<int*>(s1 + 256);
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
    TileCands t;
    t.c0 = s0[0]; t.c1 = s1[0]; t.cnt = scnt[0]; t.pad = 0;
    tc[(long)map * gridDim.x + tile] = t;
  }
}

// one thread per map: merge the tiles, apply the reference's multi-peak rule, gather reg / tracking, and emit the
// derived entries of the decode dictionary (decode.py:296-312: cts, cts_wreg = int64 + float32 -> float32,
// regs = cts_wreg - cts in float32) so that no element-wise glue kernels follow
__global__ __launch_bounds__(64) void peaks_merge_kernel(
    const TileCands* __restrict__ tc, int ntiles, const float* __restrict__ hm, const float* __restrict__ reg,
    const float* __restrict__ trk, float* __restrict__ scores, long long* __restrict__ xs, long long* __restrict__ ys,
    float* __restrict__ regs, float* __restrict__ trks, int* __restrict__ npeaks, long long* __restrict__ cts,
    float* __restrict__ cts_wreg, float* __restrict__ regs_rel, float* __restrict__ kp, int nmaps, int K, int H,
    int W) {
  const int map = blockIdx.x * 64 + threadIdx.x;
  if (map >= nmaps) return;
  const int b = map / K, HW = H * W;
  Cand best{0, 0, 0, -1}, second{0, 0, 0, -1};
  int n = 0;
  for (int t = 0; t < ntiles; ++t) {
    const TileCands v = tc[(long)map * ntiles + t];
    cand_insert(best, second, v.c0);
    cand_insert(best, second, v.c1);
    n += v.cnt;
  }
  const float* ori = hm + (long)map * HW;
  const bool have = (n == 1) || (n > 1 && (best.score - second.score) >= 0.25f);
  long long xi = 0, yi = 0;
  float sc = -1.0f;
  if (have) {
    xi = (long long)best.cx;                     // int(): truncation
    yi = (long long)best.cy;
    sc = ori[yi * W + xi];
  }
  emit_map(map, b, HW, W, xi, yi, sc, n, reg, trk, scores, xs, ys, regs, trks, npeaks, cts, cts_wreg, regs_rel, kp);
}

static bool gauss_peaks_fits(int W, size_t* bytes) {
  const size_t fl = (size_t)(GP_TH + 26) * W + (size_t)(GP_TH + 2) * (W + 2 * GRADIUS) + (size_t)(GP_TH + 2) * W;
  size_t b = fl * sizeof(float);
  const size_t red = 512 * sizeof(Cand) + 256 * sizeof(int);
  if (b < red) b = red;
  *bytes = b;
  return b <= 150 * 1024;
}

extern "C" int sgta_peaks_decode(const float* hm, const float* reg, const float* tracking,
                                 const double* gauss_w13, float* ws_tmp, float* ws_filt, float* scores,
                                 long long* xs, long long* ys, float* regs, float* trks, int* npeaks,
                                 int B, int K, int H, int W, hipStream_t st) {
  return sgta_peaks_decode_kp(hm, reg, tracking, gauss_w13, ws_tmp, ws_filt, scores, xs, ys, regs, trks, npeaks,
                              nullptr, nullptr, nullptr, nullptr, B, K, H, W, 0, st);
}

extern "C" int sgta_peaks_decode_full(const float* hm, const float* reg, const float* tracking,
                                      const double* gauss_w13, float* ws_tmp, float* ws_filt, float* scores,
                                      long long* xs, long long* ys, float* regs, float* trks, int* npeaks,
                                      long long* cts, float* cts_wreg, float* regs_rel, int B, int K, int H, int W,
                                      int three_pass, hipStream_t st) {
  return sgta_peaks_decode_kp(hm, reg, tracking, gauss_w13, ws_tmp, ws_filt, scores, xs, ys, regs, trks, npeaks, cts,
                              cts_wreg, regs_rel, nullptr, B, K, H, W, three_pass, st);
}

extern "C" int sgta_peaks_decode_kp(const float* hm, const float* reg, const float* tracking,
                                    const double* gauss_w13, float* ws_tmp, float* ws_filt, float* scores,
                                    long long* xs, long long* ys, float* regs, float* trks, int* npeaks,
                                    long long* cts, float* cts_wreg, float* regs_rel, float* kp, int B, int K, int H,
                                    int W, int three_pass, hipStream_t st) {
  if (!hm || !gauss_w13 || !ws_tmp || !ws_filt || !scores || !xs || !ys) return SGTA_EBADARG;
  if (B <= 0 || K <= 0 || H <= 0 || W <= 0 || (reg && !regs) || (tracking && !trks)) return SGTA_EBADARG;
  if ((cts_wreg || regs_rel) && !reg) return SGTA_EBADARG;
  const long total = (long)B * K * H * W;
  const int ntiles = (H + GP_TH - 1) / GP_TH;
  size_t lds = 0;
  // the tile candidates live in ws_tmp (hm-sized float workspace): it must hold them and be 8-byte aligned
  const bool fused = !three_pass && gauss_peaks_fits(W, &lds) && ((uintptr_t)ws_tmp & 7) == 0 &&
                     (size_t)ntiles * sizeof(TileCands) <= (size_t)H * W * sizeof(float);
  if (fused) {
    static bool attr_set_by_dev[SGTA_MAX_DEVICES] = {};
    bool& attr_set = attr_set_by_dev[sgta_device_slot()];
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_peaks_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
        return SGTA_EBADARG;
      attr_set = true;
    }
    TileCands* tc = reinterpret_cast<TileCands*>(ws_tmp);
    hipLaunchKernelGGL(gauss_peaks_kernel, dim3(ntiles, B * K), dim3(256), lds, st, hm, ws_filt, gauss_w13, tc, H, W);
    hipLaunchKernelGGL(peaks_merge_kernel, dim3((B * K + 63) / 64), dim3(64), 0, st, (const TileCands*)tc, ntiles, hm,
                       reg, tracking, scores, xs, ys, regs, trks, npeaks, cts, cts_wreg, regs_rel, kp, B * K, K, H, W);
    SGTA_LAUNCH_CHECK();
    return SGTA_OK;
  }
  int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(gauss1d_kernel<0>, dim3(grid), dim3(256), 0, st, hm, ws_tmp, gauss_w13, H, W, total);
  hipLaunchKernelGGL(gauss1d_kernel<1>, dim3(grid), dim3(256), 0, st, (const float*)ws_tmp, ws_filt, gauss_w13, H, W, total);
  hipLaunchKernelGGL(peaks_select_kernel, dim3(B * K), dim3(256), 0, st, hm, (const float*)ws_filt, reg, tracking,
                     scores, xs, ys, regs, trks, npeaks, cts, cts_wreg, regs_rel, kp, K, H, W);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// ---- the decode for callers that cannot allocate device memory (code under hipGraph capture, serving loops, harnesses that
// forbid hipMalloc): sgta_peaks_decode_kp with its two hm-sized workspaces in a file-scope device array instead of caller
// buffers.  Every GPU the library runs on holds its own copy of the array (36 MiB) until the process ends.  In host code
// the array's NAME is only a handle: the device address is taken once per device with hipGetSymbolAddress (that device
// current) and only that pointer is ever passed to kernels.  Nothing has to start at zero (both halves are fully written
// before they are read); the array keeps the last call's contents, and two calls that overlap on one GPU race on it.
#define SGTA_DECODE_STATIC_MAX_ELEMS 4718592L          /* B*K*H*W <= 4.5 Mi: e.g. 32 x 7 x 120 x 160 */
static __device__ __align__(256) float scratch[2 * SGTA_DECODE_STATIC_MAX_ELEMS];

static float* decode_scratch_ptr(hipError_t* err) {
  static float* ptr_by_dev[SGTA_MAX_DEVICES] = {};
  float*& p = ptr_by_dev[sgta_device_slot()];
  *err = hipSuccess;
  if (!p) {
    void* q = nullptr;
    *err = hipGetSymbolAddress(&q, HIP_SYMBOL(scratch));     // the current device's copy of the array
    if (*err != hipSuccess) return nullptr;
    p = static_cast<float*>(q);
  }
  return p;
}

extern "C" long sgta_peaks_decode_static_max_elems(void) { return SGTA_DECODE_STATIC_MAX_ELEMS; }

extern "C" int sgta_peaks_decode_kp_static(const float* hm, const float* reg, const float* tracking,
                                           const double* gauss_w13, float* scores, long long* xs, long long* ys,
                                           float* regs, float* trks, int* npeaks, long long* cts, float* cts_wreg,
                                           float* regs_rel, float* kp, int B, int K, int H, int W, int three_pass,
                                           hipStream_t st) {
  if (B <= 0 || K <= 0 || H <= 0 || W <= 0) return SGTA_EBADARG;
  if ((long)B * K * H * W > SGTA_DECODE_STATIC_MAX_ELEMS) return SGTA_ETOOBIG;
  hipError_t e;
  float* ws = decode_scratch_ptr(&e);
  if (!ws) return e != hipSuccess ? (int)e : SGTA_EBADARG;
  return sgta_peaks_decode_kp(hm, reg, tracking, gauss_w13, ws, ws + SGTA_DECODE_STATIC_MAX_ELEMS, scores, xs, ys, regs,
                              trks, npeaks, cts, cts_wreg, regs_rel, kp, B, K, H, W, three_pass, st);
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
