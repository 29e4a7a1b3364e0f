// Fused modulated deformable conv v2 (3x3, stride 1, pad 1) + folded BN + activation on fp32 NCHW maps at fp32 ACCURACY with
// the product on the 16-bit matrix pipe (gfx950, v_mfma_f32_32x32x16_f16).  Same operator contract as sgta_dcnv2_lds_f32
// (reference: external lbin/DCNv2 dcn_v2_conv behind DeformConv, dla.py:538-550).
//
// Both operands are split into IEEE-half pairs (22 significant bits): the weights on the host (w = w_hi + w_lo), the blended
// samples in registers (v = v_hi + v_lo, v_hi = half(v) toward zero, v_lo = half(v - v_hi)); the accumulator (fp32) receives
// w_hi*v_hi + w_lo*v_hi + w_hi*v_lo -- the dropped w_lo*v_lo term is 2^-22 relative.
//
// A work-group owns a 2-D tile of 128 output pixels (TH x TW) and BM output channels.  Per chunk of 8 input channels it stages
// the (TH+12) x (TW+16) fp32 input window (channel-planar, 16-byte LDS-DMA, zero fill outside the image) and the chunk's
// weights in LDS; every lane then builds the GEMM B operand of ITS pixel: a 16-deep MFMA k-slot is (2 taps) x (8 channels),
// lane half 0 / 1 blends the even / odd tap of the pair (4 LDS reads + 4-term bilinear blend per channel, fp32), splits the
// 8 values and issues 3 MFMAs per 32 output channels.  9 taps = 5 k-steps (the tenth tap carries zero weights).  Samples
// that leave the window take a per-lane global-memory fallback, so results do not depend on the margin.
#include "common.h"
#include <type_traits>

typedef _Float16 dx_h8 __attribute__((ext_vector_type(8)));
typedef unsigned dx_u4 __attribute__((ext_vector_type(4)));

struct DxArgs {
  const float* x;  long x_bs;
  const float* om; long om_bs;     // raw offset/mask conv output [B,27,H,W]
  const void* wt;                  // fp16 [n_ct][Cin/8][5 steps][2 parts: hi, lo][BM][16]: k = 8*(tap & 1) + channel, tap = 2*step + (k >> 3)
  const float* bias;               // [n_ct*BM]
  float* y;        long y_bs;
  int B, Cin, H, W, Cout;
  int nchunks, tiles_x, tiles_y, n_ct, n_pt;
};

template <int MT, int TW>
struct DxGeom {
  static constexpr int BM = MT * 32, TH = 128 / TW, R = 6, RX = 8, CK = 8;
  static constexpr int PH = TH + 2 * R, PW = TW + 2 * RX, PSZ = PH * PW;
  static constexpr int P_TOT = CK * PSZ / 4, P_PER = (P_TOT + 255) / 256;       // 16-byte window slots, DMA rounds
  static constexpr int P_BYTES = P_PER * 4096;
  static constexpr int WIMG = 5 * 2 * BM * 32, A_PER = WIMG / 4096;             // weight bytes of one (channel tile, chunk)
  static constexpr int LDS = P_BYTES + WIMG + BM * 4;
  static_assert(WIMG % 4096 == 0 && PW % 4 == 0, "whole DMA rounds, quad rows");
};

template <int MT, int TW, int ACT>
__global__ __launch_bounds__(256, 2) void dcn_x3_kernel(DxArgs a) {
  using G = DxGeom<MT, TW>;
  constexpr int BM = G::BM, TH = G::TH, R = G::R, RX = G::RX, CK = G::CK, PH = G::PH, PW = G::PW, PSZ = G::PSZ;
  constexpr int P_TOT = G::P_TOT, P_PER = G::P_PER, A_PER = G::A_PER, WIMG = G::WIMG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const patch = reinterpret_cast<float*>(smem);
  const _Float16* const As = reinterpret_cast<const _Float16*>(smem + G::P_BYTES);
  float* const sbias = reinterpret_cast<float*>(smem + G::P_BYTES + WIMG);

  int pt, ct;
  xcd_remap(blockIdx.x, a.n_ct, pt, ct);
  if (pt >= a.n_pt) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int H = a.H, W = a.W, HW = H * W;
  const int tiles = a.tiles_x * a.tiles_y;
  const int b = pt / tiles;
  const int tr = pt - b * tiles;
  const int ty0 = (tr / a.tiles_x) * TH, tx0 = (tr % a.tiles_x) * TW;
  const int win_y = ty0 - R, win_x = tx0 - RX;   // window origin (may be negative; win_x is a multiple of 4)
  const float* xb = a.x + (long)b * a.x_bs;
  if (tid < BM) sbias[tid] = a.bias[ct * BM + tid];

  // ---- staging of one channel chunk, global -> LDS through buffer loads: the per-slot byte offsets are chunk-invariant
  // VGPRs, the chunk advance is a scalar offset, slots outside the image carry an out-of-range offset (zeros)
  unsigned poff[P_PER];
#pragma unroll
  for (int t = 0; t < P_PER; ++t) {
    const int e = tid + t * 256;
    const int ef = e * 4;                          // first float of the quad
    const int c = ef / PSZ, r = ef - c * PSZ;
    const int gy = win_y + r / PW, gx = win_x + r % PW;
    // gx is a multiple of 4 and W % 4 == 0: a quad lies inside the row or outside it as a whole
    const bool ok = e < P_TOT && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    poff[t] = ok ? (unsigned)(c * HW + gy * W + gx) * 4u : 0x80000000u;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t xrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), (short)0, a.Cin * HW * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.wt), (short)0, a.n_ct * a.nchunks * WIMG, 0x00020000);
#endif
  auto stage = [&](int kc) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned char* dst = smem + wave * 1024;
#pragma unroll
    for (int t = 0; t < P_PER; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(dst + t * 4096), 16, poff[t], kc * CK * HW * 4, 0, 0);
    const int wbase = (ct * a.nchunks + kc) * WIMG;
#pragma unroll
    for (int t = 0; t < A_PER; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(dst + G::P_BYTES + t * 4096), 16, (unsigned)tid * 16u,
                                               wbase + t * 4096, 0, 0);
#endif
  };

  // ---- this lane's pixel and the sampling parameters of its 5 taps (lane half 0: taps 0,2,4,6,8; half 1: 1,3,5,7,none)
  const int pl = wave * 32 + l31;
  const int py = ty0 + pl / TW, px = tx0 + pl % TW;
  const bool pvalid = py < H && px < W;
  stage(0);                                        // chunk 0 flies while the parameters are computed

  unsigned plo[5];                                 // window offsets of the two row pairs: lo0 | lo1 << 16
  float pw[5][4];
  unsigned outmask = 0;                            // bit s: this lane's sample of step s leaves the staged window
  const float* omp = a.om + (long)b * a.om_bs + (pvalid ? py * W + px : 0);
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int tap = 2 * s + lh;
    const bool tap_ok = tap < 9;
    const int tq = tap_ok ? tap : 0;
    const int ky = tq / 3, kx = tq - ky * 3;
    const float dy = omp[(long)(2 * tq) * HW];
    const float dx = omp[(long)(2 * tq + 1) * HW];
    const float mk = sigmoidf_(omp[(long)(18 + tq) * HW]);
    const float sy = (float)(py - 1 + ky) + dy;
    const float sx = (float)(px - 1 + kx) + dx;
    const bool inside = tap_ok && pvalid && sy > -1.0f && sx > -1.0f && sy < (float)H && sx < (float)W;
    const float fy = floorf(sy), fx = floorf(sx);
    const float ly = sy - fy, lx = sx - fx, hy = 1.0f - ly, hx = 1.0f - lx;
    const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
    const bool oy0 = y0 >= 0, oy1 = y1 <= H - 1, ox0 = x0 >= 0, ox1 = x1 <= W - 1;
    const float wx0 = (inside && ox0) ? hx : 0.0f, wx1 = (inside && ox1) ? lx : 0.0f;
    const int qx0 = min(max(x0, 0), W - 2);        // first column of the (x, x+1) pair, kept in-row
    const float wa = (x0 == qx0 ? wx0 : 0.0f) + (x1 == qx0 ? wx1 : 0.0f);
    const float wb = (x0 == qx0 + 1 ? wx0 : 0.0f) + (x1 == qx0 + 1 ? wx1 : 0.0f);
    // the reference multiplies the blended value by the mask; folding it into the row weights changes the result by one
    // rounding (<< 1e-3)
    const float wr0 = oy0 ? hy * mk : 0.0f, wr1 = oy1 ? ly * mk : 0.0f;
    pw[s][0] = wr0 * wa; pw[s][1] = wr0 * wb; pw[s][2] = wr1 * wa; pw[s][3] = wr1 * wb;
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    const int r0 = cy0 - win_y, r1 = cy1 - win_y, c0 = qx0 - win_x;
    const bool inw = !inside || (r0 >= 0 && r1 < PH && c0 >= 0 && c0 + 1 < PW);
    const bool use = inw && inside;
    plo[s] = use ? (unsigned)(r0 * PW + c0) | ((unsigned)(r1 * PW + c0) << 16) : 0u;
    if (!inw) outmask |= 1u << s;
  }
  unsigned wave_out = 0;                           // wave-uniform: bit set if ANY lane's sample leaves the window
#pragma unroll
  for (int s = 0; s < 5; ++s)
    if (__any((outmask >> s) & 1u)) wave_out |= 1u << s;
  const bool any_out = wave_out != 0;              // the slow loop is skipped entirely for most waves

  f32x16 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

  // One channel chunk.  SLOW = some sample of this wave leaves the staged window and is fetched from global memory instead;
  // the two variants are separate code so that the common (fast) loop contains LDS reads only.
  auto chunk = [&](int kc, auto slow_tag) {
    constexpr bool SLOW = decltype(slow_tag)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the chunk has landed
    __syncthreads();
    const float* xc = xb + (long)kc * CK * HW;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const float* p0 = patch + (plo[s] & 0xffffu);
      const float* p1 = patch + (plo[s] >> 16);
      float q[CK][4];
#pragma unroll
      for (int j = 0; j < CK; ++j) {
#if defined(DX_ABLATE) && (DX_ABLATE & 1)
        q[j][0] = pw[s][0] + j; q[j][1] = pw[s][1]; q[j][2] = pw[s][2]; q[j][3] = pw[s][3];      // tools only: no gather reads
#else
        q[j][0] = p0[j * PSZ]; q[j][1] = p0[j * PSZ + 1]; q[j][2] = p1[j * PSZ]; q[j][3] = p1[j * PSZ + 1];
#endif
      }
      if (SLOW) {
        // wave-uniform test first: steps in which no lane of the wave samples outside the window cost nothing
        if ((wave_out >> s) & 1u) {
          if ((outmask >> s) & 1u) {               // rare: sample outside the staged window
            const int tap = 2 * s + lh;
            const float sy = (float)(py - 1 + tap / 3) + omp[(long)(2 * tap) * HW];
            const float sx = (float)(px - 1 + tap % 3) + omp[(long)(2 * tap + 1) * HW];
            const int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
            const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y0 + 1, 0), H - 1);
            const int qx0 = min(max(x0, 0), W - 2);
            // all 16 loads in flight at once; opaque: a plain C++ load here gets merged with the LDS reads above into FLAT loads
            f32x2 g[CK][2];
#pragma unroll
            for (int j = 0; j < CK; ++j) {
              const float* g0 = xc + (long)j * HW + cy0 * W + qx0;
              const float* g1 = xc + (long)j * HW + cy1 * W + qx0;
              asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_dwordx2 %1, %3, off"
                           : "=&v"(g[j][0]), "=&v"(g[j][1]) : "v"(g0), "v"(g1) : "memory");
            }
            // the wait names every destination register, so no use can be scheduled above it
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(g[0][0]), "+v"(g[0][1]), "+v"(g[1][0]), "+v"(g[1][1]), "+v"(g[2][0]), "+v"(g[2][1]), "+v"(g[3][0]), "+v"(g[3][1]),
                           "+v"(g[4][0]), "+v"(g[4][1]), "+v"(g[5][0]), "+v"(g[5][1]), "+v"(g[6][0]), "+v"(g[6][1]), "+v"(g[7][0]), "+v"(g[7][1])
                         :: "memory");
#pragma unroll
            for (int j = 0; j < CK; ++j) {
              q[j][0] = g[j][0][0]; q[j][1] = g[j][0][1]; q[j][2] = g[j][1][0]; q[j][3] = g[j][1][1];
            }
          }
        }
      }
      dx_u4 hi, lo;
#if defined(DX_ABLATE) && (DX_ABLATE & 8)
#pragma unroll
      for (int j2 = 0; j2 < CK / 2; ++j2) {      // tools only: no blend, no split
        hi[j2] = __builtin_bit_cast(unsigned, q[2 * j2][0]);
        lo[j2] = __builtin_bit_cast(unsigned, q[2 * j2 + 1][2]);
      }
#else
#pragma unroll
      for (int j2 = 0; j2 < CK / 2; ++j2) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int j = 2 * j2 + e;
          float t = pw[s][0] * q[j][0];
          t = fmaf(pw[s][1], q[j][1], t);
          t = fmaf(pw[s][2], q[j][2], t);
          v[e] = fmaf(pw[s][3], q[j][3], t);
        }
        // toward-zero conversions: the residual v - hi is exact in fp32 and fits 13 bits, of which lo keeps 11
        const auto hp = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]);
        // residuals v - float(hi) in one instruction each: the half operand is converted inside v_fma_mix_f32
        float r0, r1;
        const unsigned hpu = __builtin_bit_cast(unsigned, hp);
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hpu), "v"(v[0]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hpu), "v"(v[1]));
        const auto lp = __builtin_amdgcn_cvt_pkrtz(r0, r1);
        hi[j2] = __builtin_bit_cast(unsigned, hp);
        lo[j2] = __builtin_bit_cast(unsigned, lp);
      }
#endif
      const dx_h8 bh = __builtin_bit_cast(dx_h8, hi), bl = __builtin_bit_cast(dx_h8, lo);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#if defined(DX_ABLATE) && (DX_ABLATE & 16)
        const dx_h8 ah = __builtin_bit_cast(dx_h8, hi), al = __builtin_bit_cast(dx_h8, lo);         // tools only: no weight reads
#else
        const dx_h8 ah = *reinterpret_cast<const dx_h8*>(As + ((s * 2 + 0) * BM + i * 32 + l31) * 16 + lh * 8);
        const dx_h8 al = *reinterpret_cast<const dx_h8*>(As + ((s * 2 + 1) * BM + i * 32 + l31) * 16 + lh * 8);
#endif
#if defined(DX_ABLATE) && (DX_ABLATE & 2)
        acc[i][0] += (float)ah[0] * (float)bh[0] + (float)al[1] * (float)bl[1];                     // tools only: no MFMA
#else
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i], 0, 0, 0);
#endif
      }
    }
    if (kc + 1 < a.nchunks) {
      __syncthreads();                             // every wave has read this chunk: the single stage may be overwritten
#if !(defined(DX_ABLATE) && (DX_ABLATE & 4))
      stage(kc + 1);
#endif
    }
  };
  if (any_out) {
    for (int kc = 0; kc < a.nchunks; ++kc) chunk(kc, std::true_type{});
  } else {
    for (int kc = 0; kc < a.nchunks; ++kc) chunk(kc, std::false_type{});
  }

  // ---- epilogue ---------------------------------------------------------------------------------------
  if (!pvalid) return;
  float* yb = a.y + (long)b * a.y_bs + py * W + px;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cl = i * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
      const int co = ct * BM + cl;
      float v = acc[i][r] + sbias[cl];
      if (ACT == 1) v = fmaxf(v, 0.0f);
      if (co < a.Cout) yb[(long)co * HW] = v;
    }
}

template <int MT, int TW, int ACT>
static int dx_go(DxArgs& a, hipStream_t st) {
  using G = DxGeom<MT, TW>;
  auto kern = dcn_x3_kernel<MT, TW, ACT>;
  if (G::LDS > 64 * 1024) {
    static bool attr_by_dev[SGTA_MAX_DEVICES] = {};
    bool& done = attr_by_dev[sgta_device_slot()];
    if (!done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess)
        return SGTA_EBADARG;
      done = true;
    }
  }
  a.tiles_x = cdiv(a.W, TW);
  a.tiles_y = cdiv(a.H, G::TH);
  a.n_pt = a.B * a.tiles_x * a.tiles_y;
  const int grid = cdiv(a.n_pt, 8) * 8 * a.n_ct;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), G::LDS, st, a);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

template <int ACT>
static int dx_launch(DxArgs& a, int mt, int log_tw, hipStream_t st) {
  if (mt == 2) {
    if (log_tw == 5) return dx_go<2, 32, ACT>(a, st);
    if (log_tw == 4) return dx_go<2, 16, ACT>(a, st);
    if (log_tw == 3) return dx_go<2, 8, ACT>(a, st);
  }
  if (mt == 4) {
    if (log_tw == 5) return dx_go<4, 32, ACT>(a, st);
    if (log_tw == 4) return dx_go<4, 16, ACT>(a, st);
    if (log_tw == 3) return dx_go<4, 8, ACT>(a, st);
  }
  return SGTA_EBADARG;
}

// x, y fp32 NCHW (x_bs / y_bs elements per image, x 16-byte aligned, W % 4 == 0, x_bs % 4 == 0); om fp32 [B,27,H,W]: the raw
// conv_offset_mask output (channels 0..17 = (dy, dx) per tap, 18..26 = mask logits); wt: IEEE half
// [n_ct][Cin/8][5][2][32*mt][16] (see DxArgs; n_ct = ceil(Cout / (32*mt)), rows beyond Cout zero), wt_bytes its size; bias f32
// [n_ct*32*mt] (BN folded); act 1 = ReLU; mt in {2, 4}; log_tw in 3..5: the tile is (128 >> log_tw) rows x (1 << log_tw) columns.
extern "C" int sgta_dcnv2_x3(const float* x, long x_bs, const float* om, long om_bs, const void* wt, long wt_bytes,
                             const float* bias, float* y, long y_bs, int B, int Cin, int H, int W, int Cout, int mt,
                             int log_tw, int act, hipStream_t st) {
  if (!x || !om || !wt || !bias || !y || B <= 0 || H <= 0 || W < 4 || Cout <= 0 || Cin <= 0) return SGTA_EBADARG;
  if (Cin % 8 || W % 4 || x_bs % 4 || ((uintptr_t)x & 15) || ((uintptr_t)wt & 15) || act < 0 || act > 1 || (mt != 2 && mt != 4))
    return SGTA_EBADARG;
  // buffer descriptors (LDS-DMA with hardware bounds check) address one image / the packed weights in 31 bits
  if ((long)Cin * H * W * 4 > 0x7fffffffL || wt_bytes > 0x7fffffffL) return SGTA_EBADARG;
  DxArgs a{};
  a.x = x; a.x_bs = x_bs; a.om = om; a.om_bs = om_bs; a.wt = wt; a.bias = bias; a.y = y; a.y_bs = y_bs;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.nchunks = Cin / 8;
  a.n_ct = cdiv(Cout, 32 * mt);
  if (wt_bytes != (long)a.n_ct * a.nchunks * 5 * 2 * 32 * mt * 32) return SGTA_EBADARG;
  return act ? dx_launch<1>(a, mt, log_tw, st) : dx_launch<0>(a, mt, log_tw, st);
}
