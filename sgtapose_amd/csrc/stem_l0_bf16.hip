// Fused input stage of the bf16 data path for gfx950: the two 7x7 stems and level0 in ONE kernel
//
//   s  = relu(bn(conv7x7(img))) + relu(bn(conv7x7(hm)))          (DLA.base_layer / pre_hm_layer and their sum:
//   y0 = relu(bn(conv3x3(s)))                                      reference dla.py:259-270, :324-328; level0 :271-273)
//
// fp32 NCHW image [B,3,H,W] + prior map [B,1,H,W] in, bf16 blocked level0 map [B][1][H][W][16] out.  Un-fused, the 16-channel
// stem map is written and read back at full resolution (2 x 32 bytes per pixel against 16 bytes of input) and the 7x7
// products run on the f32 matrix pipe; here the stem of an 18x66-pixel region stays in LDS as bf16 (rounded exactly as
// the un-fused path rounds it on its way to HBM) and both products run on v_mfma_f32_16x16x32_bf16:
//
//   stem    D[cout 16][pixel 16] += W[cout][k] * X[k][pixel],  k = (ky; kx 0..7, 4 channels): 7 k-steps of 32, the 8th tap
//           carries zero weights.  The input window lies in LDS pixel-major ([row][col][image 3 + prior 1] bf16, 8 bytes per
//           pixel), so a lane's 8 k (2 taps x 4 channels) are 16 contiguous, 8-byte-aligned bytes = one ds_read2_b64, no
//           im2col copy (a DS access off its natural alignment is replayed at 64 cycles: a plane-major window whose
//           fragments are 8 taps of one channel at 2-byte alignment ran 4x slower).  The two stems have separate ReLUs:
//           two accumulators, the image stem's A fragments are zero on channel 3 and the prior stem's on channels 0..2
//   level0  k = (tap, 16 channels): 5 k-steps of 2 taps; a pixel's fragment is one aligned ds_read_b128 of the stem region
//           (two 16-byte halves per pixel, XOR-swizzled by bit 3 of the pixel index: conflict-free for 16 consecutive pixels)
//
// MFMA row m = 4 j + i carries output channel 4 perm[j] + i, perm = {0, 2, 1, 3}: after one v_permlane32_swap between two
// pixel groups every lane holds 8 consecutive channels of ONE pixel = a 16-byte store (LDS and HBM alike).
// The input image is rounded to bf16 once (2^-9 relative, like every other activation of this path); accumulation fp32.
#include "common.h"

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct SlArgs {
  const float* img; long img_bs;
  const float* hm;  long hm_bs;
  const void* w_stem;                // [7 k-steps (ky)][2: image stem, prior stem][64 lanes] A fragments (8 elements each)
  const float* b_stem;               // [32]: image stem 0..15, prior stem 16..31 (folded BN)
  const void* w_l0;                  // [5 k-steps][64 lanes]
  const float* b_l0;                 // [16]
  void* y; long y_bs;
  int B, H, W, n_ty, n_tx;
};

namespace sl {
constexpr int TH = 16, TW = 64;                          // level0 pixels per work-group
constexpr int WH = TH + 8, WW = TW + 8, WS = 80;         // input window rows / columns / row stride (pixels of 8 bytes)
constexpr int WQ = WW / 4;                               // float4 loads per window row and channel
constexpr int SH = TH + 2, SW = TW + 2, SPX = SH * SW;   // stem region
constexpr int WIN_BYTES = WH * WS * 8;                   // 15,360
constexpr int STEM_BYTES = SPX * 32;                     // 38,016
constexpr int N_SPAIR = (SPX + 31) / 32;                 // pairs of 16-pixel groups of the stem region (38)
}  // namespace sl

// F16: IEEE half instead of bf16 for the staged input, the stem region, the weights and the output (v_mfma_f32_16x16x32_f16).
template <bool F16>
__global__ __launch_bounds__(256, 3) void stem_l0_bf16_kernel(SlArgs a) {
  using namespace sl;
  using E = H16<F16>;
  typedef typename E::T elt;
  typedef typename E::x8 bf16x8;
  typedef typename E::x4 bf16x4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[WIN_BYTES + STEM_BYTES];
  unsigned char* const win = lds;
  unsigned char* const stem = lds + WIN_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kg = lane >> 4;
  const int bid = blockIdx.x;
  const int tx = bid % a.n_tx, ty = (bid / a.n_tx) % a.n_ty, b = bid / (a.n_tx * a.n_ty);
  const int y0 = ty * TH, x0 = tx * TW;
  const int H = a.H, W = a.W;
  const long HW = (long)H * W;

  // ---- phase 1: fp32 window -> bf16 [row][col][4 channels] in LDS (zero outside the image = the 7x7 convs' padding) ----------
  for (int e = tid; e < WH * WQ; e += 256) {
    const int r = e / WQ, q = e - r * WQ;
    const int gy = y0 - 4 + r, gx = x0 - 4 + 4 * q;
    f32x4 v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
      const float* src = a.img + (long)b * a.img_bs + (long)gy * W + gx;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = *reinterpret_cast<const f32x4*>(src + c * HW);
      v[3] = *reinterpret_cast<const f32x4*>(a.hm + (long)b * a.hm_bs + (long)gy * W + gx);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<bf16x4*>(win + (r * WS + 4 * q + i) * 8) = E::cvt4(f32x4{v[0][i], v[1][i], v[2][i], v[3][i]});
  }
  if (tid < 4 * WH)                                         // columns 72..79: read by the zero-weight 8th tap, must be finite
    *reinterpret_cast<u32x4*>(win + ((tid >> 2) * WS + WW) * 8 + (tid & 3) * 16) = u32x4{0u, 0u, 0u, 0u};

  // ---- weights and biases of this lane (A fragments: row m = lane & 15, k group = lane >> 4) -------------------------------
  bf16x8 ws[7][2], wl[5];
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    ws[s][0] = reinterpret_cast<const bf16x8*>(a.w_stem)[(2 * s) * 64 + lane];
    ws[s][1] = reinterpret_cast<const bf16x8*>(a.w_stem)[(2 * s + 1) * 64 + lane];
  }
#pragma unroll
  for (int s = 0; s < 5; ++s) wl[s] = reinterpret_cast<const bf16x8*>(a.w_l0)[s * 64 + lane];
  const int c4 = 4 * (((kg & 1) << 1) | (kg >> 1));        // first of this lane's 4 output channels (rows 4 kg .. 4 kg + 3)
  const f32x4 bi = *reinterpret_cast<const f32x4*>(a.b_stem + c4);
  const f32x4 bh = *reinterpret_cast<const f32x4*>(a.b_stem + 16 + c4);
  const f32x4 bl = *reinterpret_cast<const f32x4*>(a.b_l0 + c4);

  __syncthreads();

  // ---- phase 2: stem of the (TH+2) x (TW+2) region -> bf16 in LDS ---------------------------------------------------------
  for (int pr = wave; pr < N_SPAIR; pr += 4) {
    int p[2], base[2];
    bool inside[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      p[u] = (2 * pr + u) * 16 + n;
      const int pc = p[u] < SPX ? p[u] : SPX - 1;
      const int ry = pc / SW, rx = pc - ry * SW;
      base[u] = (ry * WS + rx + 2 * kg) * 8;                              // taps kx = 2 kg, 2 kg + 1 of row ky = k-step
      inside[u] = (unsigned)(y0 - 1 + ry) < (unsigned)H && (unsigned)(x0 - 1 + rx) < (unsigned)W;
    }
    f32x4 ai[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, ah[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < 7; ++s)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const unsigned char* src = win + base[u] + s * (WS * 8);
        const u32x2 lo = *reinterpret_cast<const u32x2*>(src), hi = *reinterpret_cast<const u32x2*>(src + 8);
        const bf16x8 x = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
        ai[u] = E::mfma16(ws[s][0], x, ai[u]);
        ah[u] = E::mfma16(ws[s][1], x, ah[u]);
      }
    u32x2 v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x4 r;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        r[i] = fmaxf(ai[u][i] + bi[i], 0.0f) + fmaxf(ah[u][i] + bh[i], 0.0f);
        if (!inside[u]) r[i] = 0.0f;                                      // level0's zero padding
      }
      v[u] = __builtin_bit_cast(u32x2, E::cvt4(r));
    }
    const auto s0 = __builtin_amdgcn_permlane32_swap(v[0][0], v[1][0], false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(v[0][1], v[1][1], false, false);
    const int pst = lane < 32 ? p[0] : p[1];
    if (pst < SPX)
      *reinterpret_cast<u32x4*>(stem + pst * 32 + (((kg & 1) ^ ((pst >> 3) & 1)) << 4)) = u32x4{s0[0], s1[0], s0[1], s1[1]};
  }
  __syncthreads();

  // ---- phase 3: level0 (3x3, 16 -> 16) from the stem region, 16-byte stores of the blocked map ----------------------------
  int toff[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int t = 2 * s + (kg >> 1);
    const int tt = t < 9 ? t : 0;
    toff[s] = (tt / 3) * SW + tt % 3;
  }
  elt* const yb = reinterpret_cast<elt*>(a.y) + (long)b * a.y_bs;
#pragma unroll 2
  for (int j = 0; j < (TH * TW / 32) / 4; ++j) {
    const int pr = wave * ((TH * TW / 32) / 4) + j;
    const int oy = pr >> 1, oxa = (pr & 1) * 32;                            // groups 2 pr, 2 pr + 1: columns oxa + 16 u + n
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int px = oy * SW + oxa + 16 * u + n + toff[s];
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(stem + px * 32 + (((kg & 1) ^ ((px >> 3) & 1)) << 4));
        acc[u] = E::mfma16(wl[s], x, acc[u]);
      }
    u32x2 v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x4 r;
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = fmaxf(acc[u][i] + bl[i], 0.0f);
      v[u] = __builtin_bit_cast(u32x2, E::cvt4(r));
    }
    const auto s0 = __builtin_amdgcn_permlane32_swap(v[0][0], v[1][0], false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(v[0][1], v[1][1], false, false);
    const int gy = y0 + oy, gx = x0 + oxa + (lane < 32 ? 0 : 16) + n;
    if (gy < H && gx < W)
      *reinterpret_cast<u32x4*>(yb + ((long)gy * W + gx) * 16 + 8 * (kg & 1)) = u32x4{s0[0], s1[0], s0[1], s1[1]};
  }
}

template <bool F16>
static int sl_entry(const float* img, long img_bs, const float* hm, long hm_bs, const void* w_stem,
                    const float* b_stem, const void* w_l0, const float* b_l0, void* y, long y_bs, int B,
                    int H, int W, hipStream_t st) {
  if (!img || !hm || !w_stem || !b_stem || !w_l0 || !b_l0 || !y || B <= 0 || H <= 0 || W <= 0) return SGTA_EBADARG;
  // float4 window loads: rows and planes of both inputs must be 16-byte aligned
  if (W % 4 || img_bs % 4 || hm_bs % 4 || ((uintptr_t)img & 15) || ((uintptr_t)hm & 15) || ((uintptr_t)y & 15) || y_bs % 8)
    return SGTA_EBADARG;
  if (((uintptr_t)w_stem & 15) || ((uintptr_t)w_l0 & 15) || ((uintptr_t)b_stem & 15) || ((uintptr_t)b_l0 & 15)) return SGTA_EBADARG;
  SlArgs a{};
  a.img = img; a.img_bs = img_bs; a.hm = hm; a.hm_bs = hm_bs;
  a.w_stem = w_stem; a.b_stem = b_stem;
  a.w_l0 = w_l0; a.b_l0 = b_l0;
  a.y = y; a.y_bs = y_bs; a.B = B; a.H = H; a.W = W;
  a.n_ty = cdiv(H, sl::TH); a.n_tx = cdiv(W, sl::TW);
  const long blocks = (long)B * a.n_ty * a.n_tx;
  if (blocks > 0x7fffffffL) return SGTA_EBADARG;
  hipLaunchKernelGGL(stem_l0_bf16_kernel<F16>, dim3((unsigned)blocks), dim3(256), 0, st, a);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

extern "C" int sgta_stem_level0_bf16(const float* img, long img_bs, const float* hm, long hm_bs, const void* w_stem,
                                     const float* b_stem, const void* w_l0, const float* b_l0, void* y, long y_bs, int B,
                                     int H, int W, hipStream_t st) {
  return sl_entry<false>(img, img_bs, hm, hm_bs, w_stem, b_stem, w_l0, b_l0, y, y_bs, B, H, W, st);
}

extern "C" int sgta_stem_level0_f16(const float* img, long img_bs, const float* hm, long hm_bs, const void* w_stem,
                                    const float* b_stem, const void* w_l0, const float* b_l0, void* y, long y_bs, int B,
                                    int H, int W, hipStream_t st) {
  return sl_entry<true>(img, img_bs, hm, hm_bs, w_stem, b_stem, w_l0, b_l0, y, y_bs, B, H, W, st);
}

// ---- the same fused stage at fp32 ACCURACY on the 16-bit matrix pipe (operand splitting) -----------------------------------------
// Every operand is an IEEE-half pair, value = hi + lo (22 significant bits): the staged input window, the stem region in LDS and
// both weight sets; each product is three MFMAs (w_hi x_hi + w_lo x_hi + w_hi x_lo, fp32 accumulate).  LDS holds two planes of
// the window and two of the stem region, so the tile is 8 x 64 pixels (61 KB, two work-groups per CU).  Output: the level0 map
// as two half planes of the blocked layout (lo plane y_lo_delta elements behind the hi plane) or as fp32 NCHW.
namespace slx {
constexpr int TH = 8, TW = 64;
constexpr int WH = TH + 8, WW = TW + 8, WS = 80;
constexpr int WQ = WW / 4;
constexpr int SH = TH + 2, SW = TW + 2, SPX = SH * SW;
constexpr int WIN_BYTES = WH * WS * 8;                   // 10,240 per plane
constexpr int STEM_BYTES = SPX * 32;                     // 21,120 per plane
constexpr int N_SPAIR = (SPX + 31) / 32;
}  // namespace slx

struct SxArgs {
  const float* img; long img_bs;
  const float* hm;  long hm_bs;
  const void* w_stem; const void* w_stem_lo;   // [7][2][64 lanes] A fragments, hi / lo parts
  const float* b_stem;
  const void* w_l0; const void* w_l0_lo;       // [5][64 lanes]
  const float* b_l0;
  void* y; long y_bs; long y_lo_delta;
  int out_f32;                                 // 0: y = half planes, blocked; 1: y = fp32 NCHW
  int B, H, W, n_ty, n_tx;
};

__global__ __launch_bounds__(256, 2) void stem_l0_x3_kernel(SxArgs a) {
  using namespace slx;
  using E = H16<true>;
  typedef E::x8 h8;
  typedef E::x4 h4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * WIN_BYTES + 2 * STEM_BYTES];
  unsigned char* const win = lds;                        // hi plane, lo plane WIN_BYTES behind
  unsigned char* const stem = lds + 2 * WIN_BYTES;       // hi plane, lo plane STEM_BYTES behind
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kg = lane >> 4;
  const int bid = blockIdx.x;
  const int tx = bid % a.n_tx, ty = (bid / a.n_tx) % a.n_ty, b = bid / (a.n_tx * a.n_ty);
  const int y0 = ty * TH, x0 = tx * TW;
  const int H = a.H, W = a.W;
  const long HW = (long)H * W;
  auto split4 = [](f32x4 v, h4& hi, h4& lo) {
    hi = E::cvt4(v);
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = v[i] - (float)hi[i];
    lo = E::cvt4(r);
  };

  // ---- phase 1: fp32 window -> two half planes [row][col][4 channels] in LDS (zero outside the image) ------------------------
  // (all loads of a thread's items are issued before the first conversion: one memory latency per work-group, not one per item)
  constexpr int NIT = (WH * WQ + 255) / 256;
  f32x4 v[NIT][4];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = tid + it * 256;
    const int r = e / WQ, q = e - r * WQ;
    const int gy = y0 - 4 + r, gx = x0 - 4 + 4 * q;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[it][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (e < WH * WQ && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
      const float* src = a.img + (long)b * a.img_bs + (long)gy * W + gx;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[it][c] = *reinterpret_cast<const f32x4*>(src + c * HW);
      v[it][3] = *reinterpret_cast<const f32x4*>(a.hm + (long)b * a.hm_bs + (long)gy * W + gx);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = tid + it * 256;
    if (e >= WH * WQ) break;
    const int r = e / WQ, q = e - r * WQ;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      h4 hi, lo;
      split4(f32x4{v[it][0][i], v[it][1][i], v[it][2][i], v[it][3][i]}, hi, lo);
      *reinterpret_cast<h4*>(win + (r * WS + 4 * q + i) * 8) = hi;
      *reinterpret_cast<h4*>(win + WIN_BYTES + (r * WS + 4 * q + i) * 8) = lo;
    }
  }
  if (tid < 8 * WH) {                                      // columns 72..79 of both planes: read by the zero-weight 8th tap
    const int pl = tid / (4 * WH), t = tid - pl * 4 * WH;
    *reinterpret_cast<u32x4*>(win + pl * WIN_BYTES + ((t >> 2) * WS + WW) * 8 + (t & 3) * 16) = u32x4{0u, 0u, 0u, 0u};
  }

  const int c4 = 4 * (((kg & 1) << 1) | (kg >> 1));        // first of this lane's 4 output channels (rows 4 kg .. 4 kg + 3)
  __syncthreads();

  // ---- phase 2: stem of the (TH+2) x (TW+2) region -> two half planes in LDS -------------------------------------------------
  {
    h8 wh[7][2], wl[7][2];
#pragma unroll
    for (int s = 0; s < 7; ++s)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        wh[s][k] = reinterpret_cast<const h8*>(a.w_stem)[(2 * s + k) * 64 + lane];
        wl[s][k] = reinterpret_cast<const h8*>(a.w_stem_lo)[(2 * s + k) * 64 + lane];
      }
    const f32x4 bi = *reinterpret_cast<const f32x4*>(a.b_stem + c4);
    const f32x4 bh = *reinterpret_cast<const f32x4*>(a.b_stem + 16 + c4);
    for (int pr = wave; pr < N_SPAIR; pr += 4) {
      int p[2], base[2];
      bool inside[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        p[u] = (2 * pr + u) * 16 + n;
        const int pc = p[u] < SPX ? p[u] : SPX - 1;
        const int ry = pc / SW, rx = pc - ry * SW;
        base[u] = (ry * WS + rx + 2 * kg) * 8;                              // taps kx = 2 kg, 2 kg + 1 of row ky = k-step
        inside[u] = (unsigned)(y0 - 1 + ry) < (unsigned)H && (unsigned)(x0 - 1 + rx) < (unsigned)W;
      }
      f32x4 ai[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, ah[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int s = 0; s < 7; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const unsigned char* src = win + base[u] + s * (WS * 8);
          const u32x2 h0 = *reinterpret_cast<const u32x2*>(src), h1 = *reinterpret_cast<const u32x2*>(src + 8);
          const u32x2 l0 = *reinterpret_cast<const u32x2*>(src + WIN_BYTES), l1 = *reinterpret_cast<const u32x2*>(src + WIN_BYTES + 8);
          const h8 xh = __builtin_bit_cast(h8, u32x4{h0[0], h0[1], h1[0], h1[1]});
          const h8 xl = __builtin_bit_cast(h8, u32x4{l0[0], l0[1], l1[0], l1[1]});
          ai[u] = E::mfma16(wl[s][0], xh, ai[u]);
          ah[u] = E::mfma16(wl[s][1], xh, ah[u]);
          ai[u] = E::mfma16(wh[s][0], xl, ai[u]);
          ah[u] = E::mfma16(wh[s][1], xl, ah[u]);
          ai[u] = E::mfma16(wh[s][0], xh, ai[u]);
          ah[u] = E::mfma16(wh[s][1], xh, ah[u]);
        }
      u32x2 vh[2], vl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          r[i] = fmaxf(ai[u][i] + bi[i], 0.0f) + fmaxf(ah[u][i] + bh[i], 0.0f);
          if (!inside[u]) r[i] = 0.0f;                                      // level0's zero padding
        }
        h4 hi, lo;
        split4(r, hi, lo);
        vh[u] = __builtin_bit_cast(u32x2, hi);
        vl[u] = __builtin_bit_cast(u32x2, lo);
      }
      const auto s0 = __builtin_amdgcn_permlane32_swap(vh[0][0], vh[1][0], false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(vh[0][1], vh[1][1], false, false);
      const auto t0 = __builtin_amdgcn_permlane32_swap(vl[0][0], vl[1][0], false, false);
      const auto t1 = __builtin_amdgcn_permlane32_swap(vl[0][1], vl[1][1], false, false);
      const int pst = lane < 32 ? p[0] : p[1];
      if (pst < SPX) {
        unsigned char* d = stem + pst * 32 + (((kg & 1) ^ ((pst >> 3) & 1)) << 4);
        *reinterpret_cast<u32x4*>(d) = u32x4{s0[0], s1[0], s0[1], s1[1]};
        *reinterpret_cast<u32x4*>(d + STEM_BYTES) = u32x4{t0[0], t1[0], t0[1], t1[1]};
      }
    }
  }
  __syncthreads();

  // ---- phase 3: level0 (3x3, 16 -> 16) from the stem region ------------------------------------------------------------------
  h8 wh[5], wl[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    wh[s] = reinterpret_cast<const h8*>(a.w_l0)[s * 64 + lane];
    wl[s] = reinterpret_cast<const h8*>(a.w_l0_lo)[s * 64 + lane];
  }
  const f32x4 bl = *reinterpret_cast<const f32x4*>(a.b_l0 + c4);
  int toff[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int t = 2 * s + (kg >> 1);
    const int tt = t < 9 ? t : 0;
    toff[s] = (tt / 3) * SW + tt % 3;
  }
#pragma unroll 2
  for (int j = 0; j < (TH * TW / 32) / 4; ++j) {
    const int pr = wave * ((TH * TW / 32) / 4) + j;
    const int oy = pr >> 1, oxa = (pr & 1) * 32;                            // groups 2 pr, 2 pr + 1: columns oxa + 16 u + n
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int px = oy * SW + oxa + 16 * u + n + toff[s];
        const unsigned char* src = stem + px * 32 + (((kg & 1) ^ ((px >> 3) & 1)) << 4);
        const h8 xh = *reinterpret_cast<const h8*>(src), xl = *reinterpret_cast<const h8*>(src + STEM_BYTES);
        acc[u] = E::mfma16(wl[s], xh, acc[u]);
        acc[u] = E::mfma16(wh[s], xl, acc[u]);
        acc[u] = E::mfma16(wh[s], xh, acc[u]);
      }
    const int gy = y0 + oy;
    if (a.out_f32) {
      float* const yb = reinterpret_cast<float*>(a.y) + (long)b * a.y_bs + (long)gy * W;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int gx = x0 + oxa + 16 * u + n;
        if (gy < H && gx < W) {
#pragma unroll
          for (int i = 0; i < 4; ++i) yb[(long)(c4 + i) * HW + gx] = fmaxf(acc[u][i] + bl[i], 0.0f);
        }
      }
      continue;
    }
    u32x2 vh[2], vl[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x4 r;
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = fmaxf(acc[u][i] + bl[i], 0.0f);
      h4 hi, lo;
      split4(r, hi, lo);
      vh[u] = __builtin_bit_cast(u32x2, hi);
      vl[u] = __builtin_bit_cast(u32x2, lo);
    }
    const auto s0 = __builtin_amdgcn_permlane32_swap(vh[0][0], vh[1][0], false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(vh[0][1], vh[1][1], false, false);
    const auto t0 = __builtin_amdgcn_permlane32_swap(vl[0][0], vl[1][0], false, false);
    const auto t1 = __builtin_amdgcn_permlane32_swap(vl[0][1], vl[1][1], false, false);
    const int gx = x0 + oxa + (lane < 32 ? 0 : 16) + n;
    if (gy < H && gx < W) {
      _Float16* d = reinterpret_cast<_Float16*>(a.y) + (long)b * a.y_bs + ((long)gy * W + gx) * 16 + 8 * (kg & 1);
      *reinterpret_cast<u32x4*>(d) = u32x4{s0[0], s1[0], s0[1], s1[1]};
      *reinterpret_cast<u32x4*>(d + a.y_lo_delta) = u32x4{t0[0], t1[0], t0[1], t1[1]};
    }
  }
}

// img [B,3,H,W] / hm [B,1,H,W] fp32 (16-byte aligned rows: W % 4 == 0); w_stem / w_stem_lo, w_l0 / w_l0_lo: the A fragments of
// sgta_stem_level0_f16 computed from the hi / lo halves of the folded fp32 weights; out_f32 = 0: y = IEEE-half blocked
// [B][1][H][W][16] (y_bs elements per image) with its lo plane y_lo_delta elements behind; out_f32 = 1: y = fp32 NCHW
// [B,16,H,W] (y_bs elements per image, y_lo_delta ignored).
extern "C" int sgta_stem_level0_x3(const float* img, long img_bs, const float* hm, long hm_bs, const void* w_stem,
                                   const void* w_stem_lo, const float* b_stem, const void* w_l0, const void* w_l0_lo,
                                   const float* b_l0, void* y, long y_bs, long y_lo_delta, int out_f32, int B, int H, int W,
                                   hipStream_t st) {
  if (!img || !hm || !w_stem || !w_stem_lo || !b_stem || !w_l0 || !w_l0_lo || !b_l0 || !y || B <= 0 || H <= 0 || W <= 0)
    return SGTA_EBADARG;
  if (W % 4 || img_bs % 4 || hm_bs % 4 || ((uintptr_t)img & 15) || ((uintptr_t)hm & 15) || ((uintptr_t)y & 15)) return SGTA_EBADARG;
  if (!out_f32 && (y_bs % 8 || y_lo_delta <= 0 || y_lo_delta % 8)) return SGTA_EBADARG;
  if (((uintptr_t)w_stem & 15) || ((uintptr_t)w_stem_lo & 15) || ((uintptr_t)w_l0 & 15) || ((uintptr_t)w_l0_lo & 15) ||
      ((uintptr_t)b_stem & 15) || ((uintptr_t)b_l0 & 15))
    return SGTA_EBADARG;
  SxArgs a{};
  a.img = img; a.img_bs = img_bs; a.hm = hm; a.hm_bs = hm_bs;
  a.w_stem = w_stem; a.w_stem_lo = w_stem_lo; a.b_stem = b_stem;
  a.w_l0 = w_l0; a.w_l0_lo = w_l0_lo; a.b_l0 = b_l0;
  a.y = y; a.y_bs = y_bs; a.y_lo_delta = y_lo_delta; a.out_f32 = out_f32; a.B = B; a.H = H; a.W = W;
  a.n_ty = cdiv(H, slx::TH); a.n_tx = cdiv(W, slx::TW);
  const long blocks = (long)B * a.n_ty * a.n_tx;
  if (blocks > 0x7fffffffL) return SGTA_EBADARG;
  hipLaunchKernelGGL(stem_l0_x3_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}
