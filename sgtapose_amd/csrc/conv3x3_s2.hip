// 3x3 / stride 2 / pad 1 convolution + folded BN + ReLU (the first conv of every down-sampling BasicBlock and
// level1 of DLA-34; reference dla.py:44-46, :248-249, :302-312) for gfx950, f32 MFMA 32x32x2.
// 2-D output tile of 256 pixels (TH x TW); the (2TH+1) x (2TW+1) input window of 4 channels at a time is
// staged by buffer LDS-DMA (double buffered; scalar chunk offsets, out-of-image slots carry an out-of-range offset
// that the hardware bounds check turns into zeros) with its columns DE-INTERLEAVED
// ([even | odd], padded so that the tile rows of a 32-lane group hit disjoint banks): the stride-2 reads of
// neighbouring output pixels become unit-stride and conflict-free.  Lane half = channel of a pair, the nine
// k-steps of a pair are the taps (as in conv3x3_patch.hip).
#include "common.h"
#include <cstdlib>

struct S2Args {
  const float* x;  long x_bs;
  const float* wt;                 // [Cin/4][36][Cout_pad], k_local = c_local*9 + tap (k-major rows)
  const float* bias;               // [Cout_pad]
  float* y;        long y_bs;
  int B, Cin, H, W, Ho, Wo, Cout, Cout_pad;
  int nchunks, tiles_x, tiles_y, n_ct, n_pt;
  int ni, thi;                     // ni > 1: a tile stacks ni images of thi output rows each (small maps)
  float* pool; long pool_bs;       // TW == 32 only: MaxPool2d(2,2) of the output, [B, Cout, Ho/2, Wo/2] (nullptr: none)
};

__host__ __device__ constexpr int s2_wwp(int TW) { return TW == 8 ? 20 : (TW == 16 ? 40 : 2 * TW + 2); }

// VEC: the window is staged in its NATURAL column order by 16-byte DMA (origin 4 columns left of the tile: rows of
// 2*TW + 8 floats, W % 4 == 0) -- 3.6x fewer DMA instructions than 4-byte de-interleaved slots; the stride-2 reads of
// neighbouring output pixels then hit each bank twice, which the HBM-bound shallow layers do not notice
template <int MT, int TW, bool VEC>
__global__ __launch_bounds__(256, 2) void conv3x3_s2_kernel(S2Args a) {
  constexpr int NT = 2, BM = MT * 32, TH = 256 / TW, CK = 4;
  constexpr int WH = 2 * TH + 1, WW = 2 * TW + 1, NE = TW + 1, WWP = VEC ? 2 * TW + 8 : s2_wwp(TW), WSZ = WH * WWP;
  constexpr int KC = CK * 9;
  // weights: k-major rows As[k][BM] (a lane's A operand = consecutive output channels: conflict-free), staged by
  // 16-byte LDS-DMA; window: 4-byte slots (de-interleaved columns)
  constexpr int P_TOT = VEC ? CK * WSZ / 4 : CK * WSZ, A_SLOTS = KC * BM / 4;       // window slots: quads / floats
  constexpr int P_PER = (P_TOT + 255) / 256, A_PER = (A_SLOTS + 255) / 256;
  constexpr int P_FL = VEC ? P_PER * 1024 : P_PER * 256;                           // floats of a stage's window part
  static_assert(A_SLOTS >= 256, "weight tile smaller than one DMA round");
  static_assert(!VEC || (CK * WSZ) % 4 == 0, "quad slots");
  constexpr int STAGE = P_FL + KC * BM;
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE + BM];
  float* const sbias = lds + 2 * STAGE;

  int pt, ct;
  xcd_remap(blockIdx.x, a.n_ct, pt, ct);
  if (pt >= a.n_pt) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int H = a.H, W = a.W, HW = H * W, HWo = a.Ho * a.Wo;
  const int tiles = a.tiles_x * a.tiles_y;
  // STACKED small maps (ni > 1, tiles_y == 1, H <= 2*thi - 1): image si of the tile owns window rows si*2*thi ..
  // (si+1)*2*thi; its last row (gy = 2*thi - 1 >= H, zero) doubles as the zero row above image si + 1, so the window
  // keeps its 2*TH + 1 rows and a pixel's LDS base stays (2*ty)*WWP + tx
  const bool stacked = a.ni > 1;
  const int b = stacked ? (pt / a.tiles_x) * a.ni : pt / tiles;
  const int tr = stacked ? pt % a.tiles_x : pt - b * tiles;
  const int ty0 = stacked ? 0 : (tr / a.tiles_x) * TH, tx0 = (tr % a.tiles_x) * TW;    // output tile origin
  const int wy0 = 2 * ty0 - 1, wx0 = 2 * tx0 - 1;                        // input window origin
  const float* xb = a.x + (long)b * a.x_bs;
  if (tid < BM) sbias[tid] = a.bias[ct * BM + tid];

  // per-slot byte offsets are chunk-invariant VGPRs, the chunk advance is the scalar offset of the buffer load
  unsigned poff[P_PER], aoff[A_PER];
#pragma unroll
  for (int t = 0; t < P_PER; ++t) {
    const int e = tid + t * 256;
    int c, wr, gx;
    bool ok;
    if (VEC) {                                       // slot = 16-byte quad of a natural-order row
      constexpr int QR = WWP / 4, QC = WSZ / 4;
      c = e / QC;
      const int r = e - c * QC;
      wr = r / QR;
      gx = 2 * tx0 - 4 + 4 * (r - wr * QR);
      ok = e < P_TOT && (unsigned)gx < (unsigned)W;
    } else {
      c = e / WSZ;
      const int r = e - c * WSZ;
      const int sx = r % WWP;
      const int col = sx < NE ? 2 * sx : 2 * (sx - NE) + 1;
      wr = r / WWP;
      gx = wx0 + col;
      ok = e < P_TOT && sx < WW && (unsigned)gx < (unsigned)W;
    }
    int gy = wy0 + wr;
    long img = 0;
    if (stacked) {
      const int si = wr / (2 * a.thi);
      gy = wr - si * 2 * a.thi - 1;
      img = (long)si * a.x_bs;
      ok = ok && si < a.ni && b + si < a.B;
    }
    ok = ok && (unsigned)gy < (unsigned)H;
    poff[t] = ok ? (unsigned)((img + c * HW + gy * W + gx) * 4) : 0x80000000u;
  }
#pragma unroll
  for (int t = 0; t < A_PER; ++t) {
    // 16-byte slot e = (k, quarter); the last round is shifted back so that every DMA runs with all 256 lanes
    const int e = tid + (t < A_PER - 1 ? t * 256 : A_SLOTS - 256);
    const int k = e / (BM / 4), s4 = e - k * (BM / 4);
    aoff[t] = (unsigned)(k * a.Cout_pad + ct * BM + s4 * 4) * 4u;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t xrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), (short)0,
                                        (int)(((a.ni - 1) * a.x_bs + (long)a.Cin * HW) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.wt), (short)0, (a.Cin / CK) * KC * a.Cout_pad * 4, 0x00020000);
#endif
  auto stage = [&](int kc, int st) {
#if defined(__HIP_DEVICE_COMPILE__)
    float* dst = lds + st * STAGE + wave * 64;
    float* dst4 = lds + st * STAGE + wave * 256;
#pragma unroll
    for (int t = 0; t < P_PER; ++t)
      if (VEC) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(dst4 + t * 1024), 16, poff[t], kc * CK * HW * 4, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(dst + t * 256), 4, poff[t], kc * CK * HW * 4, 0, 0);
    float* adst = lds + st * STAGE + P_FL + wave * 256;
#pragma unroll
    for (int t = 0; t < A_PER; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(adst + (t < A_PER - 1 ? t * 1024 : A_SLOTS * 4 - 1024)), 16,
                                               aoff[t], kc * KC * a.Cout_pad * 4, 0, 0);
#endif
  };

  int tbase[NT], oy[NT], ox[NT], ob[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int pl = (wave * NT + j) * 32 + l31;
    const int ty = pl / TW, tx = pl % TW;
    oy[j] = ty0 + ty; ox[j] = tx0 + tx;
    ob[j] = b;
    if (stacked) {                                   // the tile row selects the image
      const int si = ty / a.thi;
      ob[j] = b + si;
      oy[j] = ob[j] < a.B ? ty - si * a.thi : a.Ho;  // image slots beyond the batch store nothing
      if (ob[j] >= a.B) ob[j] = a.B - 1;
    }
    tbase[j] = VEC ? (2 * ty) * WWP + 2 * tx + 3 : (2 * ty) * WWP + tx;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kc = 0; kc < a.nchunks; ++kc) {
    const int st = kc & 1;
    if (kc + 1 < a.nchunks) stage(kc + 1, st ^ 1);
    const float* win = lds + st * STAGE;
    const float* As = win + P_FL;
#pragma unroll
    for (int q = 0; q < CK / 2; ++q) {
      const float* prow = win + (2 * q + lh) * WSZ;
      const float* arow = As + ((2 * q + lh) * 9) * BM + l31;       // row k = channel*9 + tap, column = output channel
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        constexpr int dummy = 0; (void)dummy;
        const int ky = tap / 3, kx = tap % 3;
        const int toff = VEC ? ky * WWP + kx : ky * WWP + (kx & 1) * NE + (kx >> 1);
        float av[MT], bv[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) av[i] = arow[tap * BM + i * 32];
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[j] = prow[tbase[j] + toff];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  if (TW == 32 && a.pool) {
    // the 2x2 max-pool the next tree takes of this output (dla.py:218-223) rides along: with 32-wide tiles a lane's two
    // accumulators are output rows 2w and 2w + 1 of the same column, the column pair is the neighbouring lane
    const int Hp = a.Ho >> 1, Wp = a.Wo >> 1;
    const bool pok = (l31 & 1) == 0 && oy[1] < a.Ho && ox[0] + 1 < a.Wo && (oy[0] & 1) == 0;
    float* pb = a.pool + (long)ob[0] * a.pool_bs + (oy[0] >> 1) * Wp + (ox[0] >> 1);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cl = i * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
        const int co = ct * BM + cl;
        float v = fmaxf(fmaxf(acc[i][0][r], acc[i][1][r]) + sbias[cl], 0.0f);
        const float nb = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
        v = fmaxf(v, nb);
        if (pok && co < a.Cout) pb[(long)co * Hp * Wp] = v;
      }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (oy[j] >= a.Ho || ox[j] >= a.Wo) continue;
    float* yb = a.y + (long)ob[j] * a.y_bs + oy[j] * a.Wo + ox[j];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cl = i * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
        const int co = ct * BM + cl;
        const float v = fmaxf(acc[i][j][r] + sbias[cl], 0.0f);
        if (co < a.Cout) yb[(long)co * HWo] = v;
      }
  }
}

extern "C" int sgta_conv3x3s2_pool_f32(const float* x, long x_bs, const float* wt, const float* bias, float* y,
                                       long y_bs, float* pool, long pool_bs, int B, int Cin, int H, int W, int Cout,
                                       int Cout_pad, int tile, hipStream_t st);

extern "C" int sgta_conv3x3s2_f32(const float* x, long x_bs, const float* wt, const float* bias, float* y,
                                  long y_bs, int B, int Cin, int H, int W, int Cout, int Cout_pad, int tile,
                                  hipStream_t st) {
  return sgta_conv3x3s2_pool_f32(x, x_bs, wt, bias, y, y_bs, nullptr, 0, B, Cin, H, W, Cout, Cout_pad, tile, st);
}

// Same convolution that also writes MaxPool2d(2, 2) of its output (the ``bottom`` of the following Tree, dla.py:218-223)
// to pool [B, Cout, Ho/2, Wo/2]; needs the 8x32 output tile (tile % 10 == 0) and an un-stacked map, else SGTA_EBADARG.
extern "C" int sgta_conv3x3s2_pool_f32(const float* x, long x_bs, const float* wt, const float* bias, float* y,
                                       long y_bs, float* pool, long pool_bs, int B, int Cin, int H, int W, int Cout,
                                       int Cout_pad, int tile, hipStream_t st) {
  if (!x || !wt || !bias || !y || B <= 0 || Cin <= 0 || H < 2 || W < 2 || Cout <= 0) return SGTA_EBADARG;
  if (pool && tile % 10 != 0) return SGTA_EBADARG;
  if (Cin % 4 || Cout_pad % 32 || Cout_pad < Cout || (long)Cin * H * W > 0x7fffffffL) return SGTA_EBADARG;
  S2Args a{};
  a.x = x; a.x_bs = x_bs; a.wt = wt; a.bias = bias; a.y = y; a.y_bs = y_bs;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
  a.Cout = Cout; a.Cout_pad = Cout_pad; a.nchunks = Cin / 4;
  a.pool = pool; a.pool_bs = pool_bs;
  auto go = [&](auto kern, int BM, int TW) -> int {
    const int TH = 256 / TW;
    a.n_ct = a.Cout_pad / BM;
    a.tiles_x = cdiv(a.Wo, TW);
    a.tiles_y = cdiv(a.Ho, TH);
    a.n_pt = a.B * a.tiles_x * a.tiles_y;
    // small maps: ni images share a tile, thi = TH / ni output rows each (needs H <= 2*thi - 1, see the kernel)
    a.ni = 1;
    a.thi = TH;
    for (int n = 4; n >= 2; n >>= 1)
      if (TH % n == 0 && a.H <= 2 * (TH / n) - 1 && a.B >= 2 && ((n - 1) * a.x_bs + (long)a.Cin * a.H * a.W) * 4 <= 0x7fffffffL) {
        a.ni = n;
        a.thi = TH / n;
        a.n_pt = cdiv(a.B, n) * a.tiles_x;
        break;
      }
    if (a.pool && a.ni > 1) return SGTA_EBADARG;     // pooled rows would straddle stacked images
    int grid = cdiv(a.n_pt, 8) * 8 * a.n_ct;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, a);
    SGTA_LAUNCH_CHECK();
    return SGTA_OK;
  };
  // tile = 10*MT + t : MT in {1,2} = 32 / 64 couts per work-group; t in {0,1,2} = 8x32 / 16x16 / 32x8 output tiles
  const int mt = tile / 10, tw = tile % 10;
  // 16-byte window DMA where rows and planes are 16-byte aligned (SGTA_S2_VEC=0 keeps the 4-byte de-interleaved form)
  static const bool allow_vec = sgta_tool_env("SGTA_S2_VEC", 1) != 0;
  const bool vec = allow_vec && W % 4 == 0 && x_bs % 4 == 0 && ((uintptr_t)x & 15) == 0;
  if (mt == 2 && Cout_pad % 64 == 0) {
    if (tw == 0) return vec ? go(conv3x3_s2_kernel<2, 32, true>, 64, 32) : go(conv3x3_s2_kernel<2, 32, false>, 64, 32);
    if (tw == 1) return vec ? go(conv3x3_s2_kernel<2, 16, true>, 64, 16) : go(conv3x3_s2_kernel<2, 16, false>, 64, 16);
    if (tw == 2) return vec ? go(conv3x3_s2_kernel<2, 8, true>, 64, 8) : go(conv3x3_s2_kernel<2, 8, false>, 64, 8);
  }
  if (mt == 1) {
    if (tw == 0) return vec ? go(conv3x3_s2_kernel<1, 32, true>, 32, 32) : go(conv3x3_s2_kernel<1, 32, false>, 32, 32);
    if (tw == 1) return vec ? go(conv3x3_s2_kernel<1, 16, true>, 32, 16) : go(conv3x3_s2_kernel<1, 16, false>, 32, 16);
    if (tw == 2) return vec ? go(conv3x3_s2_kernel<1, 8, true>, 32, 8) : go(conv3x3_s2_kernel<1, 8, false>, 32, 8);
  }
  return SGTA_EBADARG;
}
