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
// GEMM view  D[co, m] = sum_k W[co, k] * Col[k, m],   m = flattened (b, oy, ox) output pixel.
//   MFMA C/D layout has the pixel on the lane (col = lane&31): every accumulator register stores
//   32 consecutive pixels of one output channel = coalesced 128-B NCHW segments.
// Work-group = 256 threads = 4 waves side by side along the pixel axis; wave tile = MT x NT blocks
// of 32x32.  K is walked in chunks of 16.  LDS tiles are k-CONTIGUOUS ([row][16 k + 4 pad]): a lane
// fetches its 4 k-steps of one operand row with ONE ds_read_b128 (conflict-free at the 80-B row
// stride), so a chunk is 2 x (MT+NT) wide reads followed by 8*MT*NT back-to-back MFMAs.  MFMA k-slot
// 0 / 1 (lane half) takes k 0-7 / 8-15 of the chunk -- any pairing is valid as long as A and B agree.
// The next chunk is prefetched into registers (weights: float4 along k; im2col: one 4-B load per
// (pixel, k), lanes on consecutive pixels = coalesced rows of the NCHW plane, taps re-hit L1/L2)
// and written to the other LDS buffer after the MFMAs: one barrier per chunk.
// The k -> (channel-plane offset, tap) decode is a host-built table read through the scalar cache.
#include "common.h"

struct ConvArgs {
  const float* x;  long x_bs;      // input  [B, Cin, H, W], batch stride in elements
  const float* wt;                 // packed weights [Kpad/16][Cout_pad][16]
  const float* bias;               // [Cout_pad]
  const float* res; long res_bs;   // residual with the geometry of y, or nullptr
  const float* om;  long om_bs;    // DCN only: raw offset/mask conv output [B, 27, H, W]
  const int2* ktab;                // conv only: [Kpad] {c*H*W + ky*W + kx, tap (63 = padding)}
  float* y;        long y_bs;      // output [B, Cout, Ho, Wo]
  int B, Cin, H, W, Cout, Cout_pad, Ho, Wo, stride, pad;
  int nchunks;                     // Kpad / 16
  int M;                           // B * Ho * Wo
  int n_ct, n_pt;                  // cout tiles, pixel tiles
};

struct __attribute__((packed, aligned(4))) F2u { float a, b; };   // 8-byte load at 4-byte alignment

constexpr int BK = 16, LDK = BK + 4;   // 80-B LDS rows: b128 reads and writes are conflict-free

template <int ACT, int RES, int MT, int NT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, const f32x16 (&acc)[MT][NT],
                                              const float* sbias, int ct, int pt, int wave, int lane) {
  constexpr int BM = MT * 32, BN = 4 * NT * 32;
  const int l31 = lane & 31, lh = lane >> 5;
  const int HWo = a.Ho * a.Wo;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int m = pt * BN + (wave * NT + j) * 32 + l31;
    if (m >= a.M) continue;
    const int b = m / HWo;
    const int p = m - b * HWo;
    float* yb = a.y + (long)b * a.y_bs + p;
    const float* rb = RES ? a.res + (long)b * a.res_bs + p : nullptr;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int cbase = ct * BM + i * 32 + 4 * lh;
      float rv[16];
      if (RES) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cbase + (r & 3) + 8 * (r >> 2);
          rv[r] = (co < a.Cout) ? rb[(long)co * HWo] : 0.0f;
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
        if (co < a.Cout) yb[(long)co * HWo] = v;
      }
    }
  }
}

template <int KS, int MT, int NT, bool DCN, int ACT, int RES>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs a) {
  constexpr int BM = MT * 32, BN = 4 * NT * 32, KK = KS * KS;
  constexpr int TPP = 256 / BN;                  // threads sharing one pixel of the gather (1 or 2)
  constexpr int KPT = BK / TPP;                  // k values gathered per thread per chunk (16 or 8)
  constexpr int A_F4 = (BK * BM / 4 + 255) / 256;  // float4 weight loads per thread per chunk
  __shared__ __attribute__((aligned(16))) float As[2][BM][LDK];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN][LDK];
  __shared__ float sbias[BM];

  int pt, ct;
  xcd_remap(blockIdx.x, a.n_ct, pt, ct);
  if (pt >= a.n_pt) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HW = a.H * a.W;
  const int HWo = a.Ho * a.Wo;
  if (tid < BM) sbias[tid] = a.bias[ct * BM + tid];

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
  const float* xb = a.x + (long)gb * a.x_bs;     // always a valid address
  const int iy0 = goy * a.stride - a.pad, ix0 = gox * a.stride - a.pad;
  const int xoff0 = iy0 * a.W + ix0;             // may be negative; only used when the tap is valid
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
  const unsigned tapmask32 = (unsigned)tapmask;  // KK <= 32 variants use the cheap 32-bit shift

  float breg[KPT];
  f32x4 areg[A_F4];
  int2 kte[KPT];                                 // next chunk's k-table entries (wave-uniform -> SGPRs)

  auto load_a = [&](int kc) {
    const float* src = a.wt + ((long)kc * a.Cout_pad + ct * BM) * BK;   // [BM][16] contiguous
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      int idx = tid + i * 256;                   // float4 index: row = idx/4, k-quad = idx%4
      if (BK * BM / 4 >= 256 || idx < BK * BM / 4) areg[i] = *reinterpret_cast<const f32x4*>(src + idx * 4);
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      int idx = tid + i * 256;
      if (BK * BM / 4 >= 256 || idx < BK * BM / 4)
        *reinterpret_cast<f32x4*>(&As[buf][idx >> 2][(idx & 3) * 4]) = areg[i];
    }
  };
  auto tap_ok = [&](int tap) -> int {            // all-ones when the tap of this pixel is in-bounds
    return KK <= 32 ? -(int)((tapmask32 >> tap) & 1u) : -(int)((tapmask >> tap) & 1ull);
  };
  // plain conv, phase 1: table entries (one wide scalar load) + UNCONDITIONAL loads from clamped
  // addresses -- no branches, no waits: everything that needs the loaded data happens in finish_b()
  // AFTER the chunk's MFMAs, so global-load latency hides under them.
  auto issue_b = [&](int kc) {
    const int4* kt4 = reinterpret_cast<const int4*>(a.ktab + kc * BK + khalf * KPT);
#pragma unroll
    for (int i = 0; i < KPT / 2; ++i) {
      const int4 t = kt4[i];
      kte[2 * i] = int2{t.x, t.y};
      kte[2 * i + 1] = int2{t.z, t.w};
    }
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int off = (xoff0 + kte[i].x) & tap_ok(kte[i].y);
      breg[i] = xb[off];
    }
  };
  auto finish_b = [&]() {
#pragma unroll
    for (int i = 0; i < KPT; ++i)
      breg[i] = __int_as_float(__float_as_int(breg[i]) & tap_ok(kte[i].y));
  };
  auto gather_dcn = [&](int kc) {
    {
      // chunk = 16 consecutive input channels of ONE tap (host packs k = tap*Cin + c, Cin % 16 == 0)
      const int k0 = kc * BK;
      const int tap = k0 / a.Cin;
      const int c0 = k0 - tap * a.Cin + khalf * KPT;
      const int ky = tap / 3, kx = tap - ky * 3;
      const float* omp = a.om + (long)gb * a.om_bs + gp;
      const float dy = omp[(long)(2 * tap) * HW];
      const float dx = omp[(long)(2 * tap + 1) * HW];
      const float mk = sigmoidf_(omp[(long)(18 + tap) * HW]);
      const float py = (float)(goy - 1 + ky) + dy;
      const float px = (float)(gox - 1 + kx) + dx;
      const bool inside = gvalid && py > -1.0f && px > -1.0f && py < (float)a.H && px < (float)a.W;
      const float fy = floorf(py), fx = floorf(px);
      const float ly = py - fy, lx = px - fx, hy = 1.0f - ly, hx = 1.0f - lx;
      const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
      const bool oy0 = y0 >= 0, oy1 = y1 <= a.H - 1, ox0 = x0 >= 0, ox1 = x1 <= a.W - 1;
      // The two x-corners of a row are adjacent in memory: ONE 8-byte load (4-byte aligned) per row
      // instead of two 4-byte gathers.  px0 = first element of the pair, clamped so the pair stays
      // inside the row; wa / wb = the bilinear x-weights routed to the pair's two slots.
      const float wx0 = (inside && ox0) ? hx : 0.0f, wx1 = (inside && ox1) ? lx : 0.0f;
      const int px0 = min(max(x0, 0), a.W - 2);
      const float wa = (x0 == px0 ? wx0 : 0.0f) + (x1 == px0 ? wx1 : 0.0f);
      const float wb = (x0 == px0 + 1 ? wx0 : 0.0f) + (x1 == px0 + 1 ? wx1 : 0.0f);
      const float wy0 = oy0 ? hy : 0.0f, wy1 = oy1 ? ly : 0.0f;
      const float w1 = wy0 * wa, w2 = wy0 * wb, w3 = wy1 * wa, w4 = wy1 * wb;
      const int cy0 = min(max(y0, 0), a.H - 1), cy1 = min(max(y1, 0), a.H - 1);
      const int o1 = cy0 * a.W + px0, o3 = cy1 * a.W + px0;
      const float* xc = xb + (long)c0 * HW;
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        const F2u r0 = *reinterpret_cast<const F2u*>(xc + o1);
        const F2u r1 = *reinterpret_cast<const F2u*>(xc + o3);
        float v = w1 * r0.a;
        v = fmaf(w2, r0.b, v);
        v = fmaf(w3, r1.a, v);
        v = fmaf(w4, r1.b, v);
        breg[i] = v * mk;
        xc += HW;
      }
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < KPT; i += 4)
      *reinterpret_cast<f32x4*>(&Bs[buf][gpx][khalf * KPT + i]) =
          f32x4{breg[i], breg[i + 1], breg[i + 2], breg[i + 3]};
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  load_a(0);
  if (DCN) gather_dcn(0);
  else { issue_b(0); finish_b(); }
  store_a(0);
  store_b(0);
  __syncthreads();

  const int l31 = lane & 31, lh = lane >> 5;
  for (int kc = 0; kc < a.nchunks; ++kc) {
    const int buf = kc & 1;
    const bool more = kc + 1 < a.nchunks;
    // this wave's fragments of the whole chunk: 2 x (MT+NT) ds_read_b128
    f32x4 af[2][MT], bf[2][NT];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < MT; ++i) af[h][i] = *reinterpret_cast<const f32x4*>(&As[buf][i * 32 + l31][lh * 8 + h * 4]);
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[h][j] = *reinterpret_cast<const f32x4*>(&Bs[buf][(wave * NT + j) * 32 + l31][lh * 8 + h * 4]);
    }
    if (more) {                                   // issue the next chunk's global loads; no use of them yet
      load_a(kc + 1);
      if (!DCN) issue_b(kc + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[h][i][s], bf[h][j][s], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);            // consumers of the prefetch stay BELOW the MFMAs
    if (more) {
      if (DCN) gather_dcn(kc + 1);
      else finish_b();
      store_a(buf ^ 1);
      store_b(buf ^ 1);
    }
    __syncthreads();
  }

  conv_epilogue<ACT, RES, MT, NT>(a, acc, sbias, ct, pt, wave, lane);
}

template <int KS, bool DCN, int ACT, int RES>
static int launch_tile(ConvArgs& a, int tile, hipStream_t st) {
  auto go = [&](auto kern, int BM, int BN) -> int {
    a.n_ct = a.Cout_pad / BM;
    a.n_pt = cdiv(a.M, BN);
    int grid = cdiv(a.n_pt, 8) * 8 * a.n_ct;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, a);
    SGTA_LAUNCH_CHECK();
    return SGTA_OK;
  };
  // tile: 0 auto, 1 = 32x256, 2 = 64x256, 3 = 128x128 (couts x pixels per work-group)
  if (tile == 0) tile = (a.Cout_pad % 128 == 0) ? 3 : (a.Cout_pad % 64 == 0 ? 2 : 1);
  if (tile == 3 && a.Cout_pad % 128) tile = 2;
  if (tile == 2 && a.Cout_pad % 64) tile = 1;
  switch (tile) {
    case 1: return go(conv_igemm_kernel<KS, 1, 2, DCN, ACT, RES>, 32, 256);
    case 2: return go(conv_igemm_kernel<KS, 2, 2, DCN, ACT, RES>, 64, 256);
    case 3: return go(conv_igemm_kernel<KS, 4, 1, DCN, ACT, RES>, 128, 128);
  }
  return SGTA_EBADARG;
}

// the (activation, residual) combinations the path uses; anything else is a bad argument
template <int KS, bool DCN>
static int launch_conv(ConvArgs& a, int act, int res_mode, int tile, hipStream_t st) {
  if (act == 1 && res_mode == 0) return launch_tile<KS, DCN, 1, 0>(a, tile, st);
  if (act == 0 && res_mode == 0) return launch_tile<KS, DCN, 0, 0>(a, tile, st);
  if (!DCN) {
    if (act == 1 && res_mode == 1) return launch_tile<KS, false, 1, 1>(a, tile, st);
    if (act == 1 && res_mode == 2) return launch_tile<KS, false, 1, 2>(a, tile, st);
    if (act == 2 && res_mode == 0) return launch_tile<KS, false, 2, 0>(a, tile, st);
    if (KS == 1 && act == 0 && res_mode == 1) return launch_tile<KS, false, 0, 1>(a, tile, st);   // T-frame cat-MLP partial sums
  }
  return SGTA_EBADARG;
}

extern "C" int sgta_conv2d_f32(const float* x, long x_bs, const float* wt, const float* bias,
                               const int* ktab, const float* res, long res_bs, float* y, long y_bs,
                               int B, int Cin, int H, int W, int Cout, int Cout_pad, int Kpad, int ks,
                               int stride, int pad, int act, int res_mode, int tile, hipStream_t st) {
  if (!x || !wt || !bias || !ktab || !y || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return SGTA_EBADARG;
  if (Cout_pad % 32 || Cout_pad < Cout || Kpad % 16 || Kpad < Cin * ks * ks) return SGTA_EBADARG;
  if ((res_mode != 0) != (res != nullptr) || stride < 1 || act < 0 || act > 2) return SGTA_EBADARG;
  if ((long)Cin * H * W > 0x7fffffffL) return SGTA_EBADARG;
  ConvArgs a{};
  a.x = x; a.x_bs = x_bs; a.wt = wt; a.bias = bias; a.res = res; a.res_bs = res_bs;
  a.om = nullptr; a.om_bs = 0; a.ktab = reinterpret_cast<const int2*>(ktab); a.y = y; a.y_bs = y_bs;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Cout_pad = Cout_pad;
  a.Ho = (H + 2 * pad - ks) / stride + 1;
  a.Wo = (W + 2 * pad - ks) / stride + 1;
  a.stride = stride; a.pad = pad; a.nchunks = Kpad / 16;
  long M = (long)B * a.Ho * a.Wo;
  if (M <= 0 || M > 0x7fffffffL) return SGTA_EBADARG;
  a.M = (int)M;
  switch (ks) {
    case 1: return launch_conv<1, false>(a, act, res_mode, tile, st);
    case 3: return launch_conv<3, false>(a, act, res_mode, tile, st);
    case 7: return launch_conv<7, false>(a, act, res_mode, tile, st);
  }
  return SGTA_EBADARG;
}

extern "C" int sgta_dcnv2_fwd_f32(const float* x, long x_bs, const float* om, long om_bs,
                                  const float* wt, const float* bias, float* y, long y_bs, int B,
                                  int Cin, int H, int W, int Cout, int Cout_pad, int act, int tile,
                                  hipStream_t st) {
  if (!x || !om || !wt || !bias || !y || B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return SGTA_EBADARG;
  if (Cin % 16 || Cout_pad % 32 || Cout_pad < Cout || act < 0 || act > 1 || W < 2) return SGTA_EBADARG;
  ConvArgs a{};
  a.x = x; a.x_bs = x_bs; a.wt = wt; a.bias = bias; a.res = nullptr; a.res_bs = 0;
  a.om = om; a.om_bs = om_bs; a.ktab = nullptr; a.y = y; a.y_bs = y_bs;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Cout_pad = Cout_pad;
  a.Ho = H; a.Wo = W; a.stride = 1; a.pad = 1; a.nchunks = Cin * 9 / 16;
  long M = (long)B * H * W;
  if (M > 0x7fffffffL) return SGTA_EBADARG;
  a.M = (int)M;
  return launch_conv<3, true>(a, act, 0, tile, st);
}
