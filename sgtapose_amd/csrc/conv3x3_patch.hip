// 3x3 / stride 1 / pad 1 convolution (+ folded BN bias, residual, activation) for gfx950: the shape of
// ~85 % of the hot path's FLOPs (DLA-34 BasicBlocks, level0, heads, DCN offset/mask conv; reference
// dla.py:41-69, :302-312, base_model.py:121-135).  Same f32 MFMA 32x32x2 GEMM view as conv_igemm.hip
// (D[co, pixel], pixel on the lane), but the im2col operand is never gathered from global memory:
//
//   for a run of BN consecutive output pixels [p0, p0+BN) of one image, the union of all nine taps is
//   the CONTIGUOUS input range [p0-W-1, p0+BN+W+1) of each channel plane (NCHW rows are contiguous).
//   The work-group stages that range ("patch") for CK channels in LDS with fully coalesced loads
//   (BN+2W+2 floats per channel instead of 9*BN scattered 4-byte gathers) and every wave reads its B
//   operand straight out of the patch:  B[k=(c,tap)][pixel] = patch[c][pixel + ky*W + kx], masked at
//   the image borders.  Lane half 0 / 1 of the MFMA k-slot takes channel 2q / 2q+1 of a channel pair,
//   the nine k-steps of a pair are the nine taps.
//
// K is walked in chunks of CK channels (9*CK k values, 18*CK/2*MT*NT MFMAs per wave between barriers);
// the next chunk's patch + weights are prefetched into registers before the MFMAs and written to the
// other LDS stage after them (one barrier per chunk).
#include "common.h"

struct PatchArgs {
  const float* x;  long x_bs;
  const float* wt;                 // [Cin/CK][Cout_pad][CK*9]
  const float* bias;               // [Cout_pad]
  const float* res; long res_bs;
  float* y;        long y_bs;
  int B, Cin, H, W, Cout, Cout_pad;
  int nchunks;                     // Cin / CK
  int tiles_per_img, n_ct, n_pt;
};

template <int MT, int NT, int CK, int PLMAX, int ACT, int RES>
__global__ __launch_bounds__(256, 2) void conv3x3_patch_kernel(PatchArgs a) {
  constexpr int BM = MT * 32, BN = 4 * NT * 32, KC = CK * 9, LDA = KC + 1, PLT = (PLMAX + 255) / 256;
  constexpr int A_F4 = (BM * KC / 4 + 255) / 256;
  __shared__ float patch[2][CK][PLMAX];
  __shared__ float As[2][BM][LDA];
  __shared__ float sbias[BM];

  int pt, ct;
  xcd_remap(blockIdx.x, a.n_ct, pt, ct);
  if (pt >= a.n_pt) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int W = a.W, HW = a.H * a.W;
  const int b = pt / a.tiles_per_img;
  const int p0 = (pt - b * a.tiles_per_img) * BN;
  const int PL = BN + 2 * W + 2;                 // patch length per channel (host guarantees <= PLMAX)
  const int g0 = p0 - W - 1;                     // plane offset of patch[.][0]
  const float* xb = a.x + (long)b * a.x_bs;
  if (tid < BM) sbias[tid] = a.bias[ct * BM + tid];

  // ---- prefetch registers ---------------------------------------------------------------------
  float preg[CK][PLT];
  f32x4 areg[A_F4];
  int poff[PLT];                                 // clamped plane offset of this thread's patch slots
  bool pok[PLT];
#pragma unroll
  for (int t = 0; t < PLT; ++t) {
    const int off = tid + t * 256;
    const int g = g0 + off;
    pok[t] = off < PL && g >= 0 && g < HW;
    poff[t] = pok[t] ? g : 0;
  }
  auto issue = [&](int kc) {
    const float* xc = xb + (long)kc * CK * HW;
#pragma unroll
    for (int c = 0; c < CK; ++c)
#pragma unroll
      for (int t = 0; t < PLT; ++t) preg[c][t] = xc[(long)c * HW + poff[t]];
    const float* src = a.wt + ((long)kc * a.Cout_pad + ct * BM) * KC;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int idx = tid + i * 256;
      if (BM * KC / 4 % 256 == 0 || idx < BM * KC / 4) areg[i] = *reinterpret_cast<const f32x4*>(src + idx * 4);
    }
  };
  auto commit = [&](int st) {
#pragma unroll
    for (int c = 0; c < CK; ++c)
#pragma unroll
      for (int t = 0; t < PLT; ++t) {
        const int off = tid + t * 256;
        if (PLT * 256 <= PLMAX || off < PLMAX) patch[st][c][off] = pok[t] ? preg[c][t] : 0.0f;   // off < PL <= PLMAX when pok
      }
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int idx = tid + i * 256;
      if (BM * KC / 4 % 256 == 0 || idx < BM * KC / 4) {
        const int row = (idx * 4) / KC, col = (idx * 4) % KC;   // KC % 4 == 0: a float4 never straddles rows
#pragma unroll
        for (int e = 0; e < 4; ++e) As[st][row][col + e] = areg[i][e];
      }
    }
  };

  // ---- per-lane pixel state of the MFMA B operand -------------------------------------------------
  int pxl[NT];                                   // patch index of tap (0,0) = pixel's local index
  bool tok[NT][9];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int pl = (wave * NT + j) * 32 + l31;
    const int p = p0 + pl;
    const bool valid = p < HW;
    const int y = p / W, x = p - y * W;
    pxl[j] = pl;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        tok[j][ky * 3 + kx] = valid && (unsigned)(y + ky - 1) < (unsigned)a.H && (unsigned)(x + kx - 1) < (unsigned)W;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  issue(0);
  commit(0);
  __syncthreads();

  for (int kc = 0; kc < a.nchunks; ++kc) {
    const int st = kc & 1;
    const bool more = kc + 1 < a.nchunks;
    if (more) issue(kc + 1);
    __builtin_amdgcn_sched_barrier(0);
    // software-pipelined k-steps (step = channel pair q x tap): operands of step s+1 are fetched from
    // LDS before the MFMAs of step s issue, so LDS latency hides under 4*MT*NT*16 matrix cycles
    constexpr int NSTEP = (CK / 2) * 9;
    float av[2][MT], bv[2][NT];
    auto fetch = [&](int step, int slot) {
      const int q = step / 9, tap = step - q * 9;
      const int toff = (tap / 3) * W + (tap % 3);
      const float* prow = &patch[st][2 * q + lh][0];
      const float* arow = &As[st][l31][(2 * q + lh) * 9];
#pragma unroll
      for (int i = 0; i < MT; ++i) av[slot][i] = arow[i * 32 * LDA + tap];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float v = prow[pxl[j] + toff];
        bv[slot][j] = tok[j][tap] ? v : 0.0f;
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      if (step + 1 < NSTEP) fetch(step + 1, (step + 1) & 1);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[step & 1][i], bv[step & 1][j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) commit(st ^ 1);
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int p = p0 + (wave * NT + j) * 32 + l31;
    if (p >= HW) continue;
    float* yb = a.y + (long)b * a.y_bs + p;
    const float* rb = RES ? a.res + (long)b * a.res_bs + p : nullptr;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float rv[16];
      if (RES) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = ct * BM + i * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
          rv[r] = (co < a.Cout) ? rb[(long)co * HW] : 0.0f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cl = i * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
        const int co = ct * BM + cl;
        float v = acc[i][j][r] + sbias[cl];
        if (RES == 1) v += rv[r];
        if (ACT == 1) v = fmaxf(v, 0.0f);
        if (ACT == 2) v = sigmoidf_(v);
        if (RES == 2) v += rv[r];
        if (co < a.Cout) yb[(long)co * HW] = v;
      }
    }
  }
}

template <int ACT, int RES>
static int launch_patch(PatchArgs& a, int tile, hipStream_t st) {
  constexpr int CK = 4;
  auto go = [&](auto kern, int BM, int BN) -> int {
    a.n_ct = a.Cout_pad / BM;
    a.tiles_per_img = cdiv((long)a.H * a.W, BN);
    a.n_pt = a.B * a.tiles_per_img;
    int grid = cdiv(a.n_pt, 8) * 8 * a.n_ct;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, a);
    SGTA_LAUNCH_CHECK();
    return SGTA_OK;
  };
  const int W = a.W;
  // tile: 0 auto, 1 = 32 couts x 256 px, 2 = 64 x 256, 3 = 128 x 128
  if (tile == 0) tile = (a.Cout_pad % 128 == 0) ? 3 : (a.Cout_pad % 64 == 0 ? 2 : 1);
  if (tile == 3 && a.Cout_pad % 128) tile = 2;
  if (tile == 2 && a.Cout_pad % 64) tile = 1;
  const int pl128 = 128 + 2 * W + 2, pl256 = 256 + 2 * W + 2;
  // PLMAX (floats per channel in LDS) is picked as small as the map width allows: LDS, not registers,
  // limits residency (4 work-groups per CU at <= 40 KB)
  switch (tile) {
    case 1:
      if (pl256 <= 448) return go(conv3x3_patch_kernel<1, 2, CK, 448, ACT, RES>, 32, 256);
      if (pl256 <= 640) return go(conv3x3_patch_kernel<1, 2, CK, 640, ACT, RES>, 32, 256);
      if (pl256 <= 768) return go(conv3x3_patch_kernel<1, 2, CK, 768, ACT, RES>, 32, 256);
      if (pl256 <= 1792) return go(conv3x3_patch_kernel<1, 2, CK, 1792, ACT, RES>, 32, 256);
      break;
    case 2:
      if (pl256 <= 448) return go(conv3x3_patch_kernel<2, 2, CK, 448, ACT, RES>, 64, 256);
      if (pl256 <= 640) return go(conv3x3_patch_kernel<2, 2, CK, 640, ACT, RES>, 64, 256);
      if (pl256 <= 768) return go(conv3x3_patch_kernel<2, 2, CK, 768, ACT, RES>, 64, 256);
      break;
    case 3:
      if (pl128 <= 320) return go(conv3x3_patch_kernel<4, 1, CK, 320, ACT, RES>, 128, 128);
      if (pl128 <= 512) return go(conv3x3_patch_kernel<4, 1, CK, 512, ACT, RES>, 128, 128);
      break;
  }
  return SGTA_EBADARG;                           // geometry not covered: caller uses sgta_conv2d_f32
}

extern "C" int sgta_conv3x3s1_f32(const float* x, long x_bs, const float* wt, const float* bias,
                                  const float* res, long res_bs, float* y, long y_bs, int B, int Cin,
                                  int H, int W, int Cout, int Cout_pad, int act, int res_mode, int tile,
                                  hipStream_t st) {
  if (!x || !wt || !bias || !y || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return SGTA_EBADARG;
  if (Cin % 4 || Cout_pad % 32 || Cout_pad < Cout) return SGTA_EBADARG;
  if ((res_mode != 0) != (res != nullptr) || (long)Cin * H * W > 0x7fffffffL) return SGTA_EBADARG;
  PatchArgs a{};
  a.x = x; a.x_bs = x_bs; a.wt = wt; a.bias = bias; a.res = res; a.res_bs = res_bs; a.y = y; a.y_bs = y_bs;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Cout_pad = Cout_pad; a.nchunks = Cin / 4;
  if (act == 1 && res_mode == 0) return launch_patch<1, 0>(a, tile, st);
  if (act == 0 && res_mode == 0) return launch_patch<0, 0>(a, tile, st);
  if (act == 1 && res_mode == 1) return launch_patch<1, 1>(a, tile, st);
  if (act == 1 && res_mode == 2) return launch_patch<1, 2>(a, tile, st);
  if (act == 2 && res_mode == 0) return launch_patch<2, 0>(a, tile, st);
  return SGTA_EBADARG;
}
