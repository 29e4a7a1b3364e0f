// Implicit-GEMM convolution and fused deformable convolution (DCNv2) for gfx950, fp32 in / fp32
// accumulate on the f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma chain).
//
// Replaces, on the reference's hot path:
//   * every plain conv of the DLA-34 backbone, the DCN offset/mask conv and the heads
//     (torch conv2d + BatchNorm2d(eval) + ReLU [+ residual]:  reference dla.py:41-69, 157-175,
//      259-270, 302-312; base_model.py:121-135)  -> KS in {1,3,7}, BN folded into weight/bias
//   * DCN + BN + ReLU (DeformConv, reference dla.py:538-550; the external lbin/DCNv2
//     modulated_deformable_im2col + GEMM): here the bilinear gather is the B-tile PRODUCER of the
//     GEMM, so the [9*Cin, H*W] column buffer of the reference never exists in HBM.
//
// GEMM view  D[co, m] = sum_k Wt[k, co] * Col[k, m],   m = flattened (b, oy, ox) output pixel.
//   A operand = weights  (LDS As[k][co], co contiguous),  B operand = im2col (LDS Bs[k][m], m contiguous)
//   -> MFMA C/D layout has the pixel on the lane (col = lane&31): every accumulator register stores
//      32 consecutive pixels of one output channel = coalesced 128-B NCHW segments.
// Work-group = 256 threads = 4 waves side by side along the pixel axis; wave tile = MT x NT blocks
// of 32x32.  K is walked in chunks of 16 with a register-staged prefetch of the next chunk and
// double-buffered LDS (one barrier per chunk).  Gather loads are 4 B per lane with lanes on
// consecutive pixels (coalesced rows of the NCHW plane; taps re-hit L1/L2).
#include "common.h"

struct ConvArgs {
  const float* x;  long x_bs;      // input  [B, Cin, H, W], batch stride in elements
  const float* wt;                 // packed weights [Kpad][Cout_pad]
  const float* bias;               // [Cout_pad] or nullptr
  const float* res; long res_bs;   // residual with the geometry of y, or nullptr
  const float* om;  long om_bs;    // DCN only: raw offset/mask conv output [B, 27, H, W]
  float* y;        long y_bs;      // output [B, Cout, Ho, Wo]
  int B, Cin, H, W, Cout, Cout_pad, Ho, Wo, stride, pad;
  int K;                           // Cin * KS * KS (un-padded)
  int nchunks;                     // Kpad / 16
  int act;                         // 0 none, 1 relu, 2 sigmoid
  int res_mode;                    // 0 none, 1 add before act, 2 add after act
  int M;                           // B * Ho * Wo
  int n_ct, n_pt;                  // cout tiles, pixel tiles
};

template <int KS, int MT, int NT, bool DCN>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs a) {
  constexpr int BK = 16, BM = MT * 32, BN = 4 * NT * 32, KK = KS * KS;
  constexpr int TPP = 256 / BN;                  // threads sharing one pixel of the gather (1 or 2)
  constexpr int KPT = BK / TPP;                  // k values gathered per thread per chunk
  constexpr int A_F4 = (BK * BM / 4 + 255) / 256;  // float4 weight loads per thread per chunk
  __shared__ float As[2][BK][BM];
  __shared__ float Bs[2][BK][BN];

  int pt, ct;
  xcd_remap(blockIdx.x, a.n_ct, pt, ct);
  if (pt >= a.n_pt) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HW = a.H * a.W;
  const int HWo = a.Ho * a.Wo;

  // ---- per-thread gather state: one output pixel -------------------------------------------
  const int gpx = tid % BN;
  const int khalf = __builtin_amdgcn_readfirstlane(tid / BN);
  const int gm = pt * BN + gpx;
  const bool gvalid = gm < a.M;
  const int gmm = gvalid ? gm : 0;
  const int gb = gmm / HWo;
  const int gp = gmm - gb * HWo;
  const int goy = gp / a.Wo;
  const int gox = gp - goy * a.Wo;
  const float* xb = a.x + (long)gb * a.x_bs;
  // plain conv: top-left tap position and a validity bit per tap
  const int iy0 = goy * a.stride - a.pad, ix0 = gox * a.stride - a.pad;
  unsigned long long tapmask = 0;
  if (!DCN && gvalid) {
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        bool ok = (unsigned)(iy0 + ky) < (unsigned)a.H && (unsigned)(ix0 + kx) < (unsigned)a.W;
        tapmask |= (unsigned long long)ok << (ky * KS + kx);
      }
  }
  const long xoff0 = (long)iy0 * a.W + ix0;

  float breg[KPT];
  f32x4 areg[A_F4];

  auto load_a = [&](int kc) {
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      int idx = tid + i * 256;                   // float4 index inside the [BK][BM] tile
      if (BK * BM / 4 >= 256 || idx < BK * BM / 4) {
        int row = (idx * 4) / BM, col = (idx * 4) % BM;
        areg[i] = *reinterpret_cast<const f32x4*>(a.wt + (long)(kc * BK + row) * a.Cout_pad + ct * BM + col);
      }
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      int idx = tid + i * 256;
      if (BK * BM / 4 >= 256 || idx < BK * BM / 4) {
        int row = (idx * 4) / BM, col = (idx * 4) % BM;
        *reinterpret_cast<f32x4*>(&As[buf][row][col]) = areg[i];
      }
    }
  };
  auto gather_b = [&](int kc) {
    if (!DCN) {
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        int k = kc * BK + khalf * KPT + i;       // wave-uniform
        int c = k / KK, tap = k - c * KK;
        int ky = tap / KS, kx = tap - ky * KS;
        bool ok = (k < a.K) && ((tapmask >> tap) & 1ull);
        long off = (long)c * HW + ky * a.W + kx;
        breg[i] = ok ? xb[xoff0 + off] : 0.0f;
      }
    } else {
      // chunk = 16 consecutive input channels of ONE tap (host packs k = tap*Cin + c, Cin % 16 == 0)
      int k0 = kc * BK;
      int tap = k0 / a.Cin;
      int c0 = k0 - tap * a.Cin + khalf * KPT;
      int ky = tap / 3, kx = tap - ky * 3;
      const float* omp = a.om + (long)gb * a.om_bs + gp;
      float dy = omp[(long)(2 * tap) * HW];
      float dx = omp[(long)(2 * tap + 1) * HW];
      float mk = sigmoidf_(omp[(long)(18 + tap) * HW]);
      float py = (float)(goy - 1 + ky) + dy;
      float px = (float)(gox - 1 + kx) + dx;
      bool inside = gvalid && py > -1.0f && px > -1.0f && py < (float)a.H && px < (float)a.W;
      float fy = floorf(py), fx = floorf(px);
      float ly = py - fy, lx = px - fx, hy = 1.0f - ly, hx = 1.0f - lx;
      int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
      bool oy0 = y0 >= 0, oy1 = y1 <= a.H - 1, ox0 = x0 >= 0, ox1 = x1 <= a.W - 1;
      float w1 = (inside && oy0 && ox0) ? hy * hx : 0.0f;
      float w2 = (inside && oy0 && ox1) ? hy * lx : 0.0f;
      float w3 = (inside && oy1 && ox0) ? ly * hx : 0.0f;
      float w4 = (inside && oy1 && ox1) ? ly * lx : 0.0f;
      int cy0 = min(max(y0, 0), a.H - 1), cy1 = min(max(y1, 0), a.H - 1);
      int cx0 = min(max(x0, 0), a.W - 1), cx1 = min(max(x1, 0), a.W - 1);
      int o1 = cy0 * a.W + cx0, o2 = cy0 * a.W + cx1, o3 = cy1 * a.W + cx0, o4 = cy1 * a.W + cx1;
      const float* xc = xb + (long)c0 * HW;
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        float v = w1 * xc[o1];
        v = fmaf(w2, xc[o2], v);
        v = fmaf(w3, xc[o3], v);
        v = fmaf(w4, xc[o4], v);
        breg[i] = v * mk;
        xc += HW;
      }
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) Bs[buf][khalf * KPT + i][gpx] = breg[i];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  load_a(0);
  gather_b(0);
  store_a(0);
  store_b(0);
  __syncthreads();

  const int l31 = lane & 31, lh = lane >> 5;
  for (int kc = 0; kc < a.nchunks; ++kc) {
    const int buf = kc & 1;
    const bool more = kc + 1 < a.nchunks;
    if (more) {                                   // prefetch chunk kc+1 into registers
      load_a(kc + 1);
      gather_b(kc + 1);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      const int k = ks * 2 + lh;
      float av[MT], bv[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) av[i] = As[buf][k][i * 32 + l31];
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = Bs[buf][k][(wave * NT + j) * 32 + l31];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      store_a(buf ^ 1);
      store_b(buf ^ 1);
    }
    __syncthreads();
  }

  // ---- epilogue: bias (+BN folded), residual, activation, coalesced NCHW stores ----------------
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int m = pt * BN + (wave * NT + j) * 32 + l31;
    if (m >= a.M) continue;
    const int b = m / HWo;
    const int p = m - b * HWo;
    float* yb = a.y + (long)b * a.y_bs + p;
    const float* rb = a.res ? a.res + (long)b * a.res_bs + p : nullptr;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = ct * BM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < a.Cout) {
          float v = acc[i][j][r];
          if (a.bias) v += a.bias[co];
          if (a.res_mode == 1) v += rb[(long)co * HWo];
          if (a.act == 1) v = fmaxf(v, 0.0f);
          else if (a.act == 2) v = sigmoidf_(v);
          if (a.res_mode == 2) v += rb[(long)co * HWo];
          yb[(long)co * HWo] = v;
        }
      }
    }
  }
}

template <int KS, bool DCN>
static int launch_conv(const ConvArgs& a0, int tile, hipStream_t st) {
  ConvArgs a = a0;
  auto go = [&](auto kern, int BM, int BN) -> int {
    a.n_ct = a.Cout_pad / BM;
    a.n_pt = cdiv(a.M, BN);
    int grid = cdiv(a.n_pt, 8) * 8 * a.n_ct;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, a);
    SGTA_LAUNCH_CHECK();
    return SGTA_OK;
  };
  // tile: 0 auto, 1 = 32x256, 2 = 64x256, 3 = 128x128
  if (tile == 0) tile = (a.Cout_pad % 128 == 0) ? 3 : (a.Cout_pad % 64 == 0 ? 2 : 1);
  if (tile == 3 && a.Cout_pad % 128) tile = 2;
  if (tile == 2 && a.Cout_pad % 64) tile = 1;
  switch (tile) {
    case 1: return go(conv_igemm_kernel<KS, 1, 2, DCN>, 32, 256);
    case 2: return go(conv_igemm_kernel<KS, 2, 2, DCN>, 64, 256);
    case 3: return go(conv_igemm_kernel<KS, 4, 1, DCN>, 128, 128);
  }
  return SGTA_EBADARG;
}

extern "C" int sgta_conv2d_f32(const float* x, long x_bs, const float* wt, const float* bias,
                               const float* res, long res_bs, float* y, long y_bs, int B, int Cin,
                               int H, int W, int Cout, int Cout_pad, int Kpad, int ks, int stride,
                               int pad, int act, int res_mode, int tile, hipStream_t st) {
  if (!x || !wt || !y || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return SGTA_EBADARG;
  if (Cout_pad % 32 || Cout_pad < Cout || Kpad % 16 || Kpad < Cin * ks * ks) return SGTA_EBADARG;
  if ((res_mode != 0) != (res != nullptr) || stride < 1 || act < 0 || act > 2) return SGTA_EBADARG;
  ConvArgs a{};
  a.x = x; a.x_bs = x_bs; a.wt = wt; a.bias = bias; a.res = res; a.res_bs = res_bs;
  a.om = nullptr; a.om_bs = 0; a.y = y; a.y_bs = y_bs;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Cout_pad = Cout_pad;
  a.Ho = (H + 2 * pad - ks) / stride + 1;
  a.Wo = (W + 2 * pad - ks) / stride + 1;
  a.stride = stride; a.pad = pad; a.K = Cin * ks * ks; a.nchunks = Kpad / 16;
  a.act = act; a.res_mode = res_mode;
  long M = (long)B * a.Ho * a.Wo;
  if (M <= 0 || M > 0x7fffffffL) return SGTA_EBADARG;
  a.M = (int)M;
  switch (ks) {
    case 1: return launch_conv<1, false>(a, tile, st);
    case 3: return launch_conv<3, false>(a, tile, st);
    case 7: return launch_conv<7, false>(a, tile, st);
  }
  return SGTA_EBADARG;
}

extern "C" int sgta_dcnv2_fwd_f32(const float* x, long x_bs, const float* om, long om_bs,
                                  const float* wt, const float* bias, float* y, long y_bs, int B,
                                  int Cin, int H, int W, int Cout, int Cout_pad, int act, int tile,
                                  hipStream_t st) {
  if (!x || !om || !wt || !y || B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return SGTA_EBADARG;
  if (Cin % 16 || Cout_pad % 32 || Cout_pad < Cout || act < 0 || act > 2) return SGTA_EBADARG;
  ConvArgs a{};
  a.x = x; a.x_bs = x_bs; a.wt = wt; a.bias = bias; a.res = nullptr; a.res_bs = 0;
  a.om = om; a.om_bs = om_bs; a.y = y; a.y_bs = y_bs;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Cout_pad = Cout_pad;
  a.Ho = H; a.Wo = W; a.stride = 1; a.pad = 1; a.K = Cin * 9; a.nchunks = Cin * 9 / 16;
  a.act = act; a.res_mode = 0;
  long M = (long)B * H * W;
  if (M > 0x7fffffffL) return SGTA_EBADARG;
  a.M = (int)M;
  return launch_conv<3, true>(a, tile, st);
}
