// 16-output-channel convolutions on the full-resolution maps (gfx950, v_mfma_f32_16x16x4_f32):
//   * level0: conv3x3(16->16)+BN+ReLU on [2B,16,H,W]                       (reference dla.py:246-247, :302-312)
//   * the two live stems fused: relu(bn(conv7x7(img 3->16))) + relu(bn(conv7x7(hm 1->16)))
//                                                                         (reference dla.py:259-270, :324-328)
// These layers are balanced between HBM and the matrix cores (16 couts give ~18 MAC per byte), so a
// 32-row MFMA tile would waste half the matrix time.  Here M = 16 couts exactly, one MFMA covers
// 16 pixels x 4 k, and the k-slot index (lane>>4) selects one of 4 input channels, so the k-steps of a
// channel group are simply the KS*KS taps.
// A work-group owns an 8 x 32 pixel tile: it stages the (8+KS-1) x (32+KS-1) window of ALL input
// channels (zero-filled outside the image, so the loop needs no border logic) and all weights in LDS
// once, then runs barrier-free: per (tap, group) one A read, and per 16-pixel block one B read + MFMA.
// The fused stem keeps two accumulators (image / heat-map conv) because each has its own BN+ReLU
// before the sum; the heat-map plane rides in k-slot 3 of the single channel group {r,g,b,hm}.
#include "common.h"

struct C16Args {
  const float* x;  long x_bs;      // main input [B, Cx, H, W]   (level0: 16 ch; stem: image, 3 ch)
  const float* x2; long x2_bs;     // DUAL: heat-map input [B,1,H,W]
  const float* wt;                 // [NG*4*KK][16] (k = ch*KK + tap); DUAL: second matrix follows
  const float* bias;               // [16]; DUAL: [32] (image conv, heat-map conv)
  float* y;        long y_bs;      // [B,16,H,W]
  int B, H, W, tiles_x, tiles_y;
};

template <int KS, int NG, bool DUAL>
__global__ __launch_bounds__(256) void conv_c16_kernel(C16Args a) {
  constexpr int TH = 8, TW = 32, KK = KS * KS, P = KS / 2;
  constexpr int WH = TH + KS - 1, WW = TW + KS - 1;
  constexpr int WSZ0 = WH * WW, WSZ = WSZ0 + ((48 - WSZ0 % 32) % 32);     // plane stride == 16 (mod 32)
  constexpr int NCH = NG * 4, NK = NCH * KK, NA = DUAL ? 2 : 1;
  __shared__ float win[NCH * WSZ];
  __shared__ __attribute__((aligned(16))) float As[NA * NK * 16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const int H = a.H, W = a.W, HW = H * W;
  const int tiles = a.tiles_x * a.tiles_y;
  const int b = blockIdx.x / tiles;
  const int tr = blockIdx.x - b * tiles;
  const int ty0 = (tr / a.tiles_x) * TH, tx0 = (tr % a.tiles_x) * TW;

  // ---- stage window (zero outside the image) and weights -----------------------------------------
  // all window loads of this thread are issued back to back (fixed trip count, registers), then stored
  constexpr int P_PER = (NCH * WSZ0 + 255) / 256;
  float pv[P_PER];
#pragma unroll
  for (int t = 0; t < P_PER; ++t) {
    const int e = tid + t * 256;
    const int ch = e / WSZ0, r = e - ch * WSZ0;
    const int gy = ty0 - P + r / WW, gx = tx0 - P + r % WW;
    const bool ok = e < NCH * WSZ0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    const int chc = ok ? ch : 0;                   // keep the unconditional load inside the tensor
    const float* src = (DUAL && chc == 3) ? a.x2 + (long)b * a.x2_bs : a.x + (long)b * a.x_bs + (long)chc * HW;
    const float v = src[ok ? gy * W + gx : 0];
    pv[t] = ok ? v : 0.0f;
  }
  constexpr int A_PER = (NA * NK * 16 / 4 + 255) / 256;   // fixed trip count: all weight loads in flight together
  f32x4 wv[A_PER];
#pragma unroll
  for (int t = 0; t < A_PER; ++t) {
    const int e = tid + t * 256;
    wv[t] = reinterpret_cast<const f32x4*>(a.wt)[e < NA * NK * 16 / 4 ? e : 0];
  }
#pragma unroll
  for (int t = 0; t < A_PER; ++t) {
    const int e = tid + t * 256;
    if (e < NA * NK * 16 / 4) reinterpret_cast<f32x4*>(As)[e] = wv[t];
  }
#pragma unroll
  for (int t = 0; t < P_PER; ++t) {
    const int e = tid + t * 256;
    const int ch = e / WSZ0, r = e - ch * WSZ0;
    if (e < NCH * WSZ0) win[ch * WSZ + r] = pv[t];
  }
  __syncthreads();

  // wave w owns tile rows 2w, 2w+1; pixel block n: row 2w + n/2, columns (n&1)*16 .. +15
  f32x4 acc[NA][4];
#pragma unroll
  for (int s = 0; s < NA; ++s)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[s][n] = f32x4{0, 0, 0, 0};

#pragma unroll 1
  for (int q = 0; q < NG; ++q) {
    const float* wp = win + (4 * q + g) * WSZ + c16;
    const float* ap = As + ((4 * q + g) * KK) * 16 + c16;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int tap = ky * KS + kx;
        float av[NA];
#pragma unroll
        for (int s = 0; s < NA; ++s) av[s] = ap[s * NK * 16 + tap * 16];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const float bv = wp[(2 * wave + (n >> 1) + ky) * WW + (n & 1) * 16 + kx];
#pragma unroll
          for (int s = 0; s < NA; ++s)
            acc[s][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv, acc[s][n], 0, 0, 0);
        }
      }
  }

  // ---- epilogue: C/D layout col = lane&15 (pixel), row = 4*(lane>>4) + r (cout) ---------------------
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int y = ty0 + 2 * wave + (n >> 1), x = tx0 + (n & 1) * 16 + c16;
    if (y >= H || x >= W) continue;
    float* yb = a.y + (long)b * a.y_bs + y * W + x;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = 4 * g + r;
      float v = fmaxf(acc[0][n][r] + a.bias[co], 0.0f);
      if (DUAL) v += fmaxf(acc[1][n][r] + a.bias[16 + co], 0.0f);
      yb[(long)co * HW] = v;
    }
  }
}

// ---- fused stems, flattened-k form ---------------------------------------------------------------------
// relu(bn(conv7x7(img))) + relu(bn(conv7x7(hm))): the image conv walks k = ch*49 + tap (147 -> 37 k-steps
// of 4), the heat-map conv k = tap (49 -> 13 k-steps); MFMA k-slot g of step s takes k = 4s + g, so each
// lane keeps the window offset of ITS k in a register (37 + 13 registers) -- no zero-padded channel
// slots: 50 MFMAs per 16 pixels instead of 98.
// NHWC_BF16: the bf16 data path's stem -- same arithmetic (fp32 inputs, f32 MFMA), output rounded to bf16 and written
// channels-last [B,H,W,ldy] (a lane holds 4 consecutive output channels of its pixel: one 8-byte store; a.y_bs = ldy)
// MODE 0: both stems fused (default).  Sequence mode (the image stem of frame t, computed when
// t is the current frame, is reused when t is the previous frame): MODE 1 = image stem only, y = relu(bn(conv7x7(img)));
// MODE 2 = heat-map stem only plus the cached image stem, y = a.x (cached [B,16,H,W]) + relu(bn(conv7x7(hm))).
template <bool NHWC_BF16, int MODE = 0, bool F16 = false>
__global__ __launch_bounds__(256) void stem_c16_kernel(C16Args a) {
  constexpr int KS = 7, TH = 8, TW = 32, KK = 49, P = 3;
  constexpr int WH = TH + KS - 1, WW = TW + KS - 1;
  constexpr int WSZ0 = WH * WW, WSZ = WSZ0 + ((48 - WSZ0 % 32) % 32);
  constexpr int NCH = 4, SI = 37, SH = 13, NKI = SI * 4, NKH = SH * 4;     // 148 / 52 padded k
  __shared__ float win[NCH * WSZ];
  __shared__ __attribute__((aligned(16))) float As[(NKI + NKH) * 16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const int H = a.H, W = a.W, HW = H * W;
  const int tiles = a.tiles_x * a.tiles_y;
  const int b = blockIdx.x / tiles;
  const int tr = blockIdx.x - b * tiles;
  const int ty0 = (tr / a.tiles_x) * TH, tx0 = (tr % a.tiles_x) * TW;

  constexpr int P_PER = (NCH * WSZ0 + 255) / 256;
  float pv[P_PER];
#pragma unroll
  for (int t = 0; t < P_PER; ++t) {
    const int e = tid + t * 256;
    const int ch = e / WSZ0, r = e - ch * WSZ0;
    const int gy = ty0 - P + r / WW, gx = tx0 - P + r % WW;
    const bool ok = e < NCH * WSZ0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    const int chc = ok ? ch : 0;
    float v = 0.0f;
    if (MODE == 0) {
      const float* src = chc == 3 ? a.x2 + (long)b * a.x2_bs : a.x + (long)b * a.x_bs + (long)chc * HW;
      v = src[ok ? gy * W + gx : 0];
    } else if (MODE == 1) {                          // image planes only
      if (ch < 3) v = (a.x + (long)b * a.x_bs + (long)chc * HW)[ok ? gy * W + gx : 0];
    } else {                                         // heat-map plane only
      if (ch == 3) v = (a.x2 + (long)b * a.x2_bs)[ok ? gy * W + gx : 0];
    }
    pv[t] = ok ? v : 0.0f;
  }
  constexpr int A_PER = ((NKI + NKH) * 16 / 4 + 255) / 256;
  f32x4 wv[A_PER];
#pragma unroll
  for (int t = 0; t < A_PER; ++t) {
    const int e = tid + t * 256;
    wv[t] = reinterpret_cast<const f32x4*>(a.wt)[e < (NKI + NKH) * 4 ? e : 0];
  }
#pragma unroll
  for (int t = 0; t < P_PER; ++t) {
    const int e = tid + t * 256;
    const int ch = e / WSZ0, r = e - ch * WSZ0;
    if (e < NCH * WSZ0) win[ch * WSZ + r] = pv[t];
  }
#pragma unroll
  for (int t = 0; t < A_PER; ++t) {
    const int e = tid + t * 256;
    if (e < (NKI + NKH) * 4) reinterpret_cast<f32x4*>(As)[e] = wv[t];
  }

  // window offset of every (k-step, k-slot): k = 4s + g.  Built ONCE per work-group by 200 threads and read back from
  // LDS in the loop -- each lane deriving its 50 offsets itself (k / 49, tap / 7, tap % 7) cost ~500 VALU instructions
  // per thread beside the 200 MFMAs of a wave
  __shared__ int offtab[(SI + SH) * 4];
  if (tid < (SI + SH) * 4) {
    const int s2 = tid >> 2, gg = tid & 3;
    int o;
    if (s2 < SI) {
      const int k = 4 * s2 + gg, ch = k / KK, tap = k - ch * KK;
      o = k < 3 * KK ? ch * WSZ + (tap / KS) * WW + tap % KS : 0;           // k = 147: zero weight row
    } else {
      const int k = 4 * (s2 - SI) + gg;
      o = k < KK ? 3 * WSZ + (k / KS) * WW + k % KS : 3 * WSZ;
    }
    offtab[tid] = o;
  }
  int pix[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) pix[n] = (2 * wave + (n >> 1)) * WW + (n & 1) * 16 + c16;
  __syncthreads();

  f32x4 acc[2][4];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[s2][n] = f32x4{0, 0, 0, 0};
  if (MODE != 2) {
#pragma unroll
    for (int s2 = 0; s2 < SI; ++s2) {
      const float av = As[(4 * s2 + g) * 16 + c16];
      const float* wrow = win + offtab[4 * s2 + g];
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wrow[pix[n]], acc[0][n], 0, 0, 0);
    }
  }
  if (MODE != 1)
#pragma unroll
  for (int s2 = 0; s2 < SH; ++s2) {
    const float av = As[(NKI + 4 * s2 + g) * 16 + c16];
    const float* wrow = win + offtab[4 * (SI + s2) + g];
#pragma unroll
    for (int n = 0; n < 4; ++n)
      acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wrow[pix[n]], acc[1][n], 0, 0, 0);
  }
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int y = ty0 + 2 * wave + (n >> 1), x = tx0 + (n & 1) * 16 + c16;
    if (y >= H || x >= W) continue;
    if (NHWC_BF16) {
      typedef typename H16<F16>::T elt;                  // bf16, or IEEE half for the f16 data path
      typedef typename H16<F16>::x4 bf16x4;
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 4 * g + r;
        v[r] = fmaxf(acc[0][n][r] + a.bias[co], 0.0f) + fmaxf(acc[1][n][r] + a.bias[16 + co], 0.0f);
      }
      elt* yh = reinterpret_cast<elt*>(a.y) + (((long)b * H + y) * W + x) * a.y_bs + 4 * g;
      *reinterpret_cast<bf16x4*>(yh) = H16<F16>::cvt4(v);
      continue;
    }
    float* yb = a.y + (long)b * a.y_bs + y * W + x;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = 4 * g + r;
      if (MODE == 0)
        yb[(long)co * HW] = fmaxf(acc[0][n][r] + a.bias[co], 0.0f) + fmaxf(acc[1][n][r] + a.bias[16 + co], 0.0f);
      else if (MODE == 1)
        yb[(long)co * HW] = fmaxf(acc[0][n][r] + a.bias[co], 0.0f);
      else                                           // same operand order as the fused form: image stem + heat-map stem
        yb[(long)co * HW] = (a.x + (long)b * a.x_bs + (long)co * HW)[y * W + x] + fmaxf(acc[1][n][r] + a.bias[16 + co], 0.0f);
    }
  }
}

extern "C" int sgta_conv_c16_f32(const float* x, long x_bs, const float* x2, long x2_bs, const float* wt,
                                 const float* bias, float* y, long y_bs, int B, int H, int W, int mode,
                                 hipStream_t st) {
  if (!x || !wt || !bias || !y || B <= 0 || H <= 0 || W <= 0) return SGTA_EBADARG;
  if ((long)16 * H * W > 0x7fffffffL) return SGTA_EBADARG;
  C16Args a{x, x_bs, x2, x2_bs, wt, bias, y, y_bs, B, H, W, cdiv(W, 32), cdiv(H, 8)};
  const int grid = B * a.tiles_x * a.tiles_y;
  if (mode == 0)                                   // level0: 3x3, 16 -> 16, BN + ReLU
    hipLaunchKernelGGL((conv_c16_kernel<3, 4, false>), dim3(grid), dim3(256), 0, st, a);
  else if (mode == 1 && x2)                        // fused stems: 7x7 (3 -> 16) and (1 -> 16)
    hipLaunchKernelGGL(stem_c16_kernel<false>, dim3(grid), dim3(256), 0, st, a);
  else if (mode == 2)                              // sequence mode: image stem only  (x = image, x2 unused)
    hipLaunchKernelGGL((stem_c16_kernel<false, 1>), dim3(grid), dim3(256), 0, st, a);
  else if (mode == 3 && x2)                        // sequence mode: x = cached image stem [B,16,H,W], x2 = prior map
    hipLaunchKernelGGL((stem_c16_kernel<false, 2>), dim3(grid), dim3(256), 0, st, a);
  else
    return SGTA_EBADARG;
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// bf16 data path: the fused 7x7 stems (dla.py:259-270, :324-328) with fp32 NCHW inputs (the boundary's images and prior
// maps) and a bf16 channels-last output y [B,H,W,ldy] (16 channels written at y).
extern "C" int sgta_stem_bf16(const float* x, long x_bs, const float* x2, long x2_bs, const float* wt, const float* bias,
                              void* y, int ldy, int B, int H, int W, hipStream_t st) {
  if (!x || !x2 || !wt || !bias || !y || B <= 0 || H <= 0 || W <= 0 || ldy < 16 || ldy % 4) return SGTA_EBADARG;
  if ((long)16 * H * W > 0x7fffffffL) return SGTA_EBADARG;
  C16Args a{x, x_bs, x2, x2_bs, wt, bias, reinterpret_cast<float*>(y), (long)ldy, B, H, W, cdiv(W, 32), cdiv(H, 8)};
  hipLaunchKernelGGL(stem_c16_kernel<true>, dim3(B * a.tiles_x * a.tiles_y), dim3(256), 0, st, a);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// The same stems with an IEEE-half output (the f16 data path).
extern "C" int sgta_stem_f16(const float* x, long x_bs, const float* x2, long x2_bs, const float* wt, const float* bias,
                             void* y, int ldy, int B, int H, int W, hipStream_t st) {
  if (!x || !x2 || !wt || !bias || !y || B <= 0 || H <= 0 || W <= 0 || ldy < 16 || ldy % 4) return SGTA_EBADARG;
  if ((long)16 * H * W > 0x7fffffffL) return SGTA_EBADARG;
  C16Args a{x, x_bs, x2, x2_bs, wt, bias, reinterpret_cast<float*>(y), (long)ldy, B, H, W, cdiv(W, 32), cdiv(H, 8)};
  hipLaunchKernelGGL((stem_c16_kernel<true, 0, true>), dim3(B * a.tiles_x * a.tiles_y), dim3(256), 0, st, a);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}
