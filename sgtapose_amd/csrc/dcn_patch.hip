// Fused modulated deformable conv v2 (3x3, stride 1, pad 1) + folded BN + activation with the input
// WINDOW staged in LDS (gfx950, f32 MFMA 32x32x2).  Same operator contract as sgta_dcnv2_fwd_f32 in
// conv_igemm.hip (reference: external lbin/DCNv2 dcn_v2_conv behind DeformConv, dla.py:538-550).
//
// A work-group owns a 2-D tile of 256 output pixels (TH x TW, TW in {32,16,8}) and BM output channels.
// For CK input channels at a time it stages the (TH+2R) x (TW+2R) input window in LDS with one pass of
// coalesced loads (each input element is fetched from L2/HBM once per tile instead of 36 scattered
// 4-byte gathers per pixel), then every lane builds the GEMM B operand for ITS OWN pixels on the fly:
// four LDS reads + a 4-term bilinear blend per (pixel, tap, channel).  There is no im2col tile at all.
// Samples that leave the window (|offset| > R-2) take a per-lane global-memory fallback, so results
// do not depend on R.  Loop order: channel chunk -> tap (sampling parameters computed once per tap) ->
// channel pair (lane half 0/1 = channel 2q/2q+1 of the MFMA k-slot).
#include "common.h"
#include <type_traits>
#include <cstdlib>

struct DcnArgs {
  const float* x;  long x_bs;
  const float* om; long om_bs;     // raw offset/mask conv output [B,27,H,W]
  const float* wt;                 // [Cin/4][36][Cout_pad], k_local = c_local*9 + tap (k-major rows)
  const float* bias;               // [Cout_pad]
  float* y;        long y_bs;
  int B, Cin, H, W, Cout, Cout_pad;
  int nchunks, tiles_x, tiles_y, n_ct, n_pt;
};


// VEC: window origin 8 columns left of the tile (16-byte aligned rows of TW + 16 floats, W % 4 == 0) staged by 16-byte
// DMA -- 3-4 DMA instructions per chunk and wave instead of 9-11; the two extra halo columns per side also widen the
// range of offsets that stay on the fast path
template <int MT, int NT, int TW, int R, int CK, int ACT, bool VEC>
__global__ __launch_bounds__(256, 2) void dcn_patch_kernel(DcnArgs a) {
  constexpr int BM = MT * 32, TH = 128 * NT / TW;
  constexpr int RX = VEC ? 8 : R;
  constexpr int PH = TH + 2 * R, PW = TW + 2 * RX, PSZ = PH * PW;
  constexpr int KC = CK * 9;
  // weights: k-major rows As[k][BM] (a lane's A operand = consecutive output channels: conflict-free), staged
  // by 16-byte LDS-DMA; window: 4-byte LDS-DMA slots (arbitrary alignment of the image rows)
  constexpr int P_TOT = VEC ? CK * PSZ / 4 : CK * PSZ, A_SLOTS = KC * BM / 4;      // window slots: quads / floats
  constexpr int P_PER = (P_TOT + 255) / 256, A_PER = (A_SLOTS + 255) / 256;
  constexpr int P_FL = VEC ? P_PER * 1024 : P_PER * 256;                          // floats of a stage's window part
  static_assert(A_SLOTS >= 256, "weight tile smaller than one DMA round");
  static_assert(!VEC || PW % 4 == 0, "quad rows");
  // ONE shared array (a second object beside an LDS-DMA target makes hipcc drain vmcnt early)
  constexpr int STAGE = P_FL + KC * BM;                       // floats per stage
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE + BM];
  float* const sbias = lds + 2 * STAGE;

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
  const int win_y = ty0 - R, win_x = tx0 - RX;   // window origin (may be negative)
  const float* xb = a.x + (long)b * a.x_bs;
  if (tid < BM) sbias[tid] = a.bias[ct * BM + tid];

  // ---- asynchronous staging of one channel chunk: window + weights, global -> LDS through buffer loads:
  // the per-slot byte offsets are chunk-invariant VGPRs, the chunk advance is a scalar offset, and slots
  // outside the image carry an out-of-range offset that the buffer bounds check turns into zeros.
  unsigned poff[P_PER], aoff[A_PER];
#pragma unroll
  for (int t = 0; t < P_PER; ++t) {
    const int e = tid + t * 256;
    const int ef = VEC ? e * 4 : e;                 // first float of the slot
    const int c = ef / PSZ, r = ef - c * PSZ;
    const int gy = win_y + r / PW, gx = win_x + r % PW;
    // VEC: gx is a multiple of 4 and W % 4 == 0, so a quad is inside the row or outside it as a whole
    const bool ok = e < P_TOT && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    poff[t] = ok ? (unsigned)(c * HW + gy * W + gx) * 4u : 0x80000000u;
  }
#pragma unroll
  for (int t = 0; t < A_PER; ++t) {
    // 16-byte slot e = (k, quarter): the last round is shifted back to end exactly at A_SLOTS (re-writing a few
    // slots with identical data) so that every DMA runs with all 256 lanes
    const int e = tid + (t < A_PER - 1 ? t * 256 : A_SLOTS - 256);
    const int k = e / (BM / 4), s4 = e - k * (BM / 4);
    aoff[t] = (unsigned)(k * a.Cout_pad + ct * BM + s4 * 4) * 4u;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t xrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), (short)0, a.Cin * HW * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.wt), (short)0, (a.Cin / CK) * KC * a.Cout_pad * 4, 0x00020000);
#endif
  // part 0 / 1: first / second half of the window slots, part 2: weights (spread between the k-steps)
  auto stage = [&](int kc, int st, int part) {
#if defined(__HIP_DEVICE_COMPILE__)
    float* dst = lds + st * STAGE + (VEC ? wave * 256 : wave * 64);
    if (part < 2) {
#pragma unroll
      for (int t = part * ((P_PER + 1) / 2); t < (part ? P_PER : (P_PER + 1) / 2); ++t)
        if (VEC) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(dst + t * 1024), 16, poff[t], kc * CK * HW * 4, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(dst + t * 256), 4, poff[t], kc * CK * HW * 4, 0, 0);
    } else {
      float* adst = lds + st * STAGE + P_FL + wave * 256;
#pragma unroll
      for (int t = 0; t < A_PER; ++t)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(adst + (t < A_PER - 1 ? t * 1024 : A_SLOTS * 4 - 1024)),
                                                 16, aoff[t], kc * KC * a.Cout_pad * 4, 0, 0);
    }
#endif
  };

  // ---- this lane's two pixels and their 9 x 2 sampling parameter sets (kept in registers) -----------
  int py[NT], px[NT];
  bool pvalid[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int pl = (wave * NT + j) * 32 + l31;
    py[j] = ty0 + pl / TW;
    px[j] = tx0 + pl % TW;
    pvalid[j] = py[j] < H && px[j] < W;
  }
  stage(0, 0, 0); stage(0, 0, 1); stage(0, 0, 2);  // chunk 0 flies while the parameters are computed

  unsigned plo[9][NT];                             // window offsets of the two row pairs: lo0 | lo1 << 16
  float pw[9][NT][4];
  unsigned outmask = 0;                            // bit tap*2+j: sample leaves the staged window
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float* omp = a.om + (long)b * a.om_bs + (pvalid[j] ? py[j] * W + px[j] : 0);
      const float dy = omp[(long)(2 * tap) * HW];
      const float dx = omp[(long)(2 * tap + 1) * HW];
      const float mk = sigmoidf_(omp[(long)(18 + tap) * HW]);
      const float sy = (float)(py[j] - 1 + ky) + dy;
      const float sx = (float)(px[j] - 1 + kx) + dx;
      const bool inside = pvalid[j] && sy > -1.0f && sx > -1.0f && sy < (float)H && sx < (float)W;
      const float fy = floorf(sy), fx = floorf(sx);
      const float ly = sy - fy, lx = sx - fx, hy = 1.0f - ly, hx = 1.0f - lx;
      const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
      const bool oy0 = y0 >= 0, oy1 = y1 <= H - 1, ox0 = x0 >= 0, ox1 = x1 <= W - 1;
      const float wx0 = (inside && ox0) ? hx : 0.0f, wx1 = (inside && ox1) ? lx : 0.0f;
      const int qx0 = min(max(x0, 0), W - 2);      // first column of the (x, x+1) pair, kept in-row
      const float wa = (x0 == qx0 ? wx0 : 0.0f) + (x1 == qx0 ? wx1 : 0.0f);
      const float wb = (x0 == qx0 + 1 ? wx0 : 0.0f) + (x1 == qx0 + 1 ? wx1 : 0.0f);
      // the reference multiplies the blended value by the mask; folding it into the row weights
      // changes the result by one rounding (<< 1e-3)
      const float wr0 = oy0 ? hy * mk : 0.0f, wr1 = oy1 ? ly * mk : 0.0f;
      pw[tap][j][0] = wr0 * wa; pw[tap][j][1] = wr0 * wb; pw[tap][j][2] = wr1 * wa; pw[tap][j][3] = wr1 * wb;
      const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
      const int r0 = cy0 - win_y, r1 = cy1 - win_y, c0 = qx0 - win_x;
      const bool inw = !inside || (r0 >= 0 && r1 < PH && c0 >= 0 && c0 + 1 < PW);
      const bool use = inw && inside;
      plo[tap][j] = use ? (unsigned)(r0 * PW + c0) | ((unsigned)(r1 * PW + c0) << 16) : 0u;
      if (!inw) outmask |= 1u << (tap * NT + j);
    }
  }
  unsigned wave_out = 0;                           // wave-uniform: bit set if ANY lane's sample leaves the window
#pragma unroll
  for (int bit = 0; bit < 9 * NT; ++bit)
    if (__any((outmask >> bit) & 1u)) wave_out |= 1u << bit;
  const bool any_out = wave_out != 0;              // the slow loop is skipped entirely for most waves

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // One channel chunk.  SLOW = some sample of this wave leaves the staged window and is fetched from global
  // memory instead; the two variants are separate code so that the common (fast) loop contains LDS reads only
  // -- a single merged loop made hipcc turn every gather into a FLAT load behind s_waitcnt vmcnt(0), i.e. every
  // k-step waited for the LDS-DMA of the next chunk.
  auto chunk = [&](int kc, auto slow_tag) {
    constexpr bool SLOW = decltype(slow_tag)::value;
    const int st = kc & 1;
    const bool more = kc + 1 < a.nchunks;
    if (more) stage(kc + 1, st ^ 1, 0);            // async: lands in the other stage during the MFMAs
    const float* patch = lds + st * STAGE;
    const float* Asm = patch + P_FL;
    const float* xc = xb + (long)kc * CK * HW;
    // software-pipelined k-steps (step = tap x channel pair): the LDS gathers + blend of step s+1 are
    // issued before the MFMAs of step s
    constexpr int NSTEP = 9 * (CK / 2);
    float av[2][MT], bv[2][NT];
    auto fetch = [&](int step, int slot) {
      const int tap = step / (CK / 2), q = step - tap * (CK / 2);
      const int ch = 2 * q + lh;
      const float* prow = patch + ch * PSZ;
#pragma unroll
      for (int i = 0; i < MT; ++i) av[slot][i] = Asm[(ch * 9 + tap) * BM + i * 32 + l31];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int lo0 = plo[tap][j] & 0xffffu, lo1 = plo[tap][j] >> 16;
        float v00 = prow[lo0], v01 = prow[lo0 + 1], v10 = prow[lo1], v11 = prow[lo1 + 1];
        if (SLOW) {
          // wave-uniform test first: taps that no lane of the wave samples outside the window cost nothing
          if ((wave_out >> (tap * NT + j)) & 1u) {
            if ((outmask >> (tap * NT + j)) & 1u) {   // rare: sample outside the staged window
              const float* omp = a.om + (long)b * a.om_bs + py[j] * W + px[j];
              const float sy = (float)(py[j] - 1 + tap / 3) + omp[(long)(2 * tap) * HW];
              const float sx = (float)(px[j] - 1 + tap % 3) + omp[(long)(2 * tap + 1) * HW];
              const int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
              const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y0 + 1, 0), H - 1);
              const int qx0 = min(max(x0, 0), W - 2);
              const float* g0 = xc + (long)ch * HW + cy0 * W + qx0;
              const float* g1 = xc + (long)ch * HW + cy1 * W + qx0;
              // opaque loads: a plain C++ load here gets merged with the LDS reads above into FLAT loads
              f32x2 q0, q1;
              asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_dwordx2 %1, %3, off\n\ts_waitcnt vmcnt(0)"
                           : "=&v"(q0), "=&v"(q1) : "v"(g0), "v"(g1) : "memory");
              v00 = q0[0]; v01 = q0[1]; v10 = q1[0]; v11 = q1[1];
            }
          }
        }
        float v = pw[tap][j][0] * v00;
        v = fmaf(pw[tap][j][1], v01, v);
        v = fmaf(pw[tap][j][2], v10, v);
        v = fmaf(pw[tap][j][3], v11, v);
        bv[slot][j] = v;
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      if (step == NSTEP / 4 && more) stage(kc + 1, st ^ 1, 1);
      if (step == NSTEP / 2 && more) stage(kc + 1, st ^ 1, 2);
      if (step + 1 < NSTEP) fetch(step + 1, (step + 1) & 1);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[step & 1][i], bv[step & 1][j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next stage has landed (this wave's DMAs)
    __syncthreads();
  };
  if (any_out) {
    for (int kc = 0; kc < a.nchunks; ++kc) chunk(kc, std::true_type{});
  } else {
    for (int kc = 0; kc < a.nchunks; ++kc) chunk(kc, std::false_type{});
  }

  // ---- epilogue ---------------------------------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (!pvalid[j]) continue;
    float* yb = a.y + (long)b * a.y_bs + py[j] * W + px[j];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cl = i * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
        const int co = ct * BM + cl;
        float v = acc[i][j][r] + sbias[cl];
        if (ACT == 1) v = fmaxf(v, 0.0f);
        if (co < a.Cout) yb[(long)co * HW] = v;
      }
  }
}

template <int ACT>
static int launch_dcn_patch(DcnArgs& a, int tile, hipStream_t st) {
  constexpr int CK = 4, R = 6;
  auto go = [&](auto kern, int BM, int NT, int TW) -> int {
    const int TH = 128 * NT / TW;
    a.n_ct = a.Cout_pad / BM;
    a.tiles_x = cdiv(a.W, TW);
    a.tiles_y = cdiv(a.H, TH);
    a.n_pt = a.B * a.tiles_x * a.tiles_y;
    int grid = cdiv(a.n_pt, 8) * 8 * a.n_ct;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, a);
    SGTA_LAUNCH_CHECK();
    return SGTA_OK;
  };
  // tile = 10*MT + t:  MT = 2 = 64 output channels per work-group (128 measured slower); 128-pixel 2-D tiles,
  // t in {0,1,2} = 4x32 / 8x16 / 16x8 (rows x cols)
  const int mt = tile / 10, tw = tile % 10;
  // 16-byte window DMA where image rows and planes are 16-byte aligned (SGTA_DCN_VEC=0: always the 4-byte form)
  static const bool allow_vec = sgta_tool_env("SGTA_DCN_VEC", 1) != 0;
  const bool vec = allow_vec && a.W % 4 == 0 && a.x_bs % 4 == 0 && ((uintptr_t)a.x & 15) == 0;
  if (mt == 2 && a.Cout_pad % 64 == 0) {
    if (tw == 0) return vec ? go(dcn_patch_kernel<2, 1, 32, R, CK, ACT, true>, 64, 1, 32) : go(dcn_patch_kernel<2, 1, 32, R, CK, ACT, false>, 64, 1, 32);
    if (tw == 1) return vec ? go(dcn_patch_kernel<2, 1, 16, R, CK, ACT, true>, 64, 1, 16) : go(dcn_patch_kernel<2, 1, 16, R, CK, ACT, false>, 64, 1, 16);
    if (tw == 2) return vec ? go(dcn_patch_kernel<2, 1, 8, R, CK, ACT, true>, 64, 1, 8) : go(dcn_patch_kernel<2, 1, 8, R, CK, ACT, false>, 64, 1, 8);
  }
  if (mt == 4 && a.Cout_pad % 128 == 0) {            // 128 output channels per work-group: every gathered value feeds 4 MFMAs
    if (tw == 0) return vec ? go(dcn_patch_kernel<4, 1, 32, R, CK, ACT, true>, 128, 1, 32) : go(dcn_patch_kernel<4, 1, 32, R, CK, ACT, false>, 128, 1, 32);
    if (tw == 1) return vec ? go(dcn_patch_kernel<4, 1, 16, R, CK, ACT, true>, 128, 1, 16) : go(dcn_patch_kernel<4, 1, 16, R, CK, ACT, false>, 128, 1, 16);
    if (tw == 2) return vec ? go(dcn_patch_kernel<4, 1, 8, R, CK, ACT, true>, 128, 1, 8) : go(dcn_patch_kernel<4, 1, 8, R, CK, ACT, false>, 128, 1, 8);
  }
  return SGTA_EBADARG;
}

extern "C" int sgta_dcnv2_lds_f32(const float* x, long x_bs, const float* om, long om_bs,
                                  const float* wt, const float* bias, float* y, long y_bs, int B,
                                  int Cin, int H, int W, int Cout, int Cout_pad, int act, int tile,
                                  hipStream_t st) {
  if (!x || !om || !wt || !bias || !y || B <= 0 || H <= 0 || W < 2 || Cout <= 0) return SGTA_EBADARG;
  if (Cin % 4 || Cout_pad % 64 || Cout_pad < Cout || act < 0 || act > 1) return SGTA_EBADARG;
  // buffer descriptors (LDS-DMA with hardware bounds check) address one image / the packed weights in 31 bits
  if ((long)Cin * H * W * 4 > 0x7fffffffL || (long)Cin * 9 * Cout_pad * 4 > 0x7fffffffL) return SGTA_EBADARG;
  DcnArgs a{};
  a.x = x; a.x_bs = x_bs; a.om = om; a.om_bs = om_bs; a.wt = wt; a.bias = bias; a.y = y; a.y_bs = y_bs;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Cout_pad = Cout_pad; a.nchunks = Cin / 4;
  return act ? launch_dcn_patch<1>(a, tile, st) : launch_dcn_patch<0>(a, tile, st);
}
