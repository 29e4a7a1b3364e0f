// Winograd F(4x4, 3x3) convolution for gfx950 (3x3, stride 1, pad 1, + folded BN bias, residual, activation;
// BasicBlocks, heads, DCN offset conv: reference dla.py:41-69, base_model.py:121-135): 36 multiplies per 4x4
// output tile and channel instead of 144 -- 4x fewer f32 MFMA cycles than a direct convolution and 1.78x fewer
// than F(2x2,3x3) (conv3x3_wino.hip), which remains the kernel for small or oddly sized maps.
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A      per 4x4 output tile (d = 6x6 input tile),
//   interpolation points 0, +-1, +-2, inf  (Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks")
//
// fp32 throughout; the transforms multiply by at most 8 and the end-to-end error on this network stays at the
// 1e-5 level, two orders below the 1e-3 parity bar.
//
// Ownership as in conv3x3_wino.hip: a wave owns 16 tiles x 16 output channels for ALL 36 positions (36
// accumulators of the f32 MFMA 16x16x4 = 144 registers); lane (g = lane / 16, t = lane % 16) reads the raw
// 6x6 window of tile t / channel g, transforms it in registers (144 VALU ops) and feeds 36 MFMAs; the output
// transform, bias, residual, activation and 16-byte row stores run in registers.  Work-groups are persistent:
// 8 waves = 64 tiles x 32 channels, one per CU (two 16-channel groups share the staged window: the bytes moved
// through the memory pipeline per MFMA, not DMA issue or latency, are what this kernel pays for beside its issue
// time), or 4 waves = 64 tiles x 16 channels, two per CU, for layers without a second channel group.
// The window is stored LINEARLY (row-major, origin 4 columns left of the block): with 4-pixel tiles a lane's
// middle four columns are one aligned ds_read_b128, its first / last column comes from the neighbouring lane
// by DPP row shift (LDS only for the lanes at a tile-row edge), and the window arrives by 16-BYTE buffer
// LDS-DMA (image rows are 16-byte aligned: W % 4 == 0 is required).
#include "common.h"
#include <cstdlib>

struct Wino4Args {
  const float* x;  long x_bs;
  const float* u;                  // [Cin/4][4][Cout_pad][36]
  const float* bias;               // [Cout_pad]
  const float* res; long res_bs;
  float* y;        long y_bs;
  int B, Cin, H, W, Cout, Cout_pad;
  int nchunks, blk_r, blk_c, n_ct, n_pt;
};

// window row length in floats: 4 columns of left margin + 4*TC + 2 + alignment; TC = 4: the four tile rows of an
// 8-lane group must start in different 128-byte bank groups
__host__ __device__ constexpr int w4_wwl(int TC) { return TC == 4 ? 28 : 4 * TC + 8; }

__device__ __forceinline__ float w4_from_left(float v) {    // value of lane - 1 inside a row of 16 lanes
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));
}
__device__ __forceinline__ float w4_from_right(float v) {   // value of lane + 1 inside a row of 16 lanes
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xF, 0xF, true));
}

// B^T x for F(4,3):  [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void w4_bt(const float (&x)[6], float (&y)[6]) {
  y[0] = fmaf(4.0f, x[0], fmaf(-5.0f, x[2], x[4]));
  y[5] = fmaf(4.0f, x[1], fmaf(-5.0f, x[3], x[5]));
  const float t1 = fmaf(-4.0f, x[2], x[4]), t2 = fmaf(-4.0f, x[1], x[3]);
  y[1] = t1 + t2;
  y[2] = t1 - t2;
  const float t3 = x[4] - x[2], t4 = x[3] - x[1];
  y[3] = fmaf(2.0f, t4, t3);
  y[4] = fmaf(-2.0f, t4, t3);
}

// A^T m for F(4,3):  [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void w4_at(const float (&m)[6], float (&y)[4]) {
  const float s1 = m[1] + m[2], s2 = m[1] - m[2], s3 = m[3] + m[4], s4 = m[3] - m[4];
  y[0] = (m[0] + s1) + s3;
  y[1] = fmaf(2.0f, s4, s2);
  y[2] = fmaf(4.0f, s3, s1);
  y[3] = fmaf(8.0f, s4, s2) + m[5];
}

// NCG = output-channel groups (of 16) per work-group: 1 -> 4 waves, two work-groups per CU; 2 -> 8 waves, one work-group
// per CU whose two channel groups share the staged window (35 % fewer bytes through the memory pipeline per MFMA).
// URC > 0: the transformed weights of ALL URC chunks (Cin = 4 * URC, one channel-group item stream: n_ct == 1) stay
// resident in LDS beside the window ring -- a shallow layer (level0: 16 -> 16 at full resolution) otherwise re-streams
// its 74 KB of U for every 32x32-pixel item, 44 % of the bytes it moves
template <int TC, int ACT, int RES, int NCG, int URC = 0>
__global__ __launch_bounds__(256 * NCG, NCG == 1 ? 2 : 1) void conv3x3_wino4_kernel(Wino4Args a) {
  constexpr int NT = 256 * NCG, BM = 16 * NCG, CK = 4, TR = 64 / TC, EDGE = TC < 16 ? TC : 16;
  // channel plane stride == 0 (mod 64 floats = one 256-byte bank row): ds_read_b128 is served in four 16-lane groups
  // that each mix tiles of two channel planes (lanes {0-3,12-15,20-27}, ...); with the planes bank-aligned and the
  // tile-row stride WWL == 8 / 12 (mod 16 slots) every group covers 16 distinct 16-byte slots (conflict-free)
  constexpr int WH = 4 * TR + 2, WWL = w4_wwl(TC), WQR = WWL / 4;
  constexpr int WSZ = (WH * WWL + 63) / 64 * 64, WQ = WSZ / 4;
  constexpr int P_SLOTS = CK * WQ, P_PER = (P_SLOTS + NT - 1) / NT;        // 16-byte window slots per stage
  constexpr int LDU = 36, U_F = CK * BM * LDU, U_SLOTS = U_F / 4, U_PER = (U_SLOTS + NT - 1) / NT;
  static_assert(P_SLOTS >= NT && U_SLOTS >= NT, "a DMA round must be full");
  constexpr int P_F = P_SLOTS * 4;
  constexpr int STAGE = URC ? P_F : P_F + U_F;                            // ring slot: window (+ U unless resident)
  __shared__ __attribute__((aligned(16))) float lds[URC ? 2 * P_F + URC * U_F : 2 * STAGE];

  // persistent work-groups, XCD-contiguous logical ids, channel group fastest (see conv3x3_wino.hip)
  const int G = gridDim.x;
  const int wg = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int n_items = a.n_pt * a.n_ct;
  if (wg >= n_items) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave >> 2, tgw = wave & 3;          // channel group / tile group of this wave
  const int g = lane >> 4, tl = lane & 15;
  const int H = a.H, W = a.W, HW = H * W;
  const int blks = a.blk_r * a.blk_c;
  const int t = tgw * 16 + tl;                       // this lane's tile inside the 64-tile block
  const int trow = t / TC, tcol = t % TC;
  const bool edge_l = (tl % EDGE) == 0, edge_r = (tl % EDGE) == EDGE - 1;
  const int poffl = g * WSZ + (4 * trow) * WWL + 4 * tcol + 4;            // middle quad of the lane's first row
  const int uoffl = (URC ? 2 * P_F : P_F) + (g * BM + cg * 16 + tl) * LDU;   // resident: behind both window slots

  // ---- buffer LDS-DMA staging (16-byte slots; out-of-image slots carry an out-of-range offset -> zeros) ------
  unsigned poff[P_PER], uoff[U_PER];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t xrs;
  const __amdgpu_buffer_rsrc_t urs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), (short)0, a.Cin * a.Cout_pad * LDU * 4, 0x00020000);
#endif
  auto setup = [&](int item) {
    const int pt = item / a.n_ct, ct = item - pt * a.n_ct;
    const int b = pt / blks, rem = pt - b * blks;
    const int br = rem / a.blk_c, bc = rem - br * a.blk_c;
    const int oy0 = 4 * br * TR, ox0 = 4 * bc * TC;
#pragma unroll
    for (int q = 0; q < P_PER; ++q) {
      const int e = tid + (q < P_PER - 1 ? q * NT : P_SLOTS - NT);       // last round shifted back (full waves)
      const int c = e / WQ, rq = e - c * WQ;
      const int r = rq / WQR, qx = rq - r * WQR;
      const int gy = oy0 - 1 + r, gx = ox0 - 4 + 4 * qx;
      const bool ok = qx < TC + 2 && r < WH && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      poff[q] = ok ? (unsigned)(c * HW + gy * W + gx) * 4u : 0x80000000u;
    }
#pragma unroll
    for (int q = 0; q < U_PER; ++q) {
      const int e = tid + (q < U_PER - 1 ? q * NT : U_SLOTS - NT);
      const int c = e / (BM * 9), rem = e - c * (BM * 9);                 // 9 x 16 bytes per (channel, output channel)
      uoff[q] = (unsigned)((c * a.Cout_pad + ct * BM) * LDU + rem * 4) * 4u;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (long)b * a.x_bs), (short)0, a.Cin * HW * 4,
                                            0x00020000);
#endif
  };
  // one chunk = window batch(es) + U batch, issued in two parts around the MFMA block
  auto stage = [&](int kc, int st, int part) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (part == 0) {
      float* dst = lds + st * STAGE + wave * 256;
#pragma unroll
      for (int q = 0; q < P_PER; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(dst + (q < P_PER - 1 ? q * NT * 4 : P_SLOTS * 4 - NT * 4)),
                                                 16, poff[q], kc * CK * HW * 4, 0, 0);
    } else {
      float* udst = lds + (URC ? 2 * P_F + kc * U_F : st * STAGE + P_F) + wave * 256;
#pragma unroll
      for (int q = 0; q < U_PER; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(urs, (lds_ptr_t)(udst + (q < U_PER - 1 ? q * NT * 4 : U_SLOTS * 4 - NT * 4)),
                                                 16, uoff[q], kc * CK * a.Cout_pad * LDU * 4, 0, 0);
    }
#endif
  };

  f32x4 acc[36];
  int item = wg, sidx = 0;
  setup(item);
  stage(0, 0, 0); stage(0, 0, 1);
  if (URC)
    for (int kc = 1; kc < URC; ++kc) stage(kc, 0, 1);           // every chunk's U, once per work-group
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  while (true) {
#pragma unroll
    for (int p = 0; p < 36; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ct = item % a.n_ct;
    f32x4 bsv;                                        // bias of this lane's 4 output channels (4g + r)
    bool more = true;
    for (int kc = 0; kc < a.nchunks; ++kc) {
      const bool last = kc + 1 >= a.nchunks;
      int nkc = kc + 1;
      if (last) {
        // last chunk of the item: point the staging registers at the next item, put its first chunk in flight
        more = item + G < n_items;
        nkc = 0;
        if (more) setup(item + G);
        bsv = *reinterpret_cast<const f32x4*>(a.bias + ct * BM + cg * 16 + 4 * g);
      }
      if (more) stage(nkc, sidx ^ 1, 0);
      // resident-U form: a channel group that is all padding (level0: 16 real output channels in a 32-wide group) only
      // stages and synchronises -- its MFMAs would take the matrix pipe from the wave it shares a SIMD with
      if (!(URC && cg * 16 >= a.Cout)) {
      const float* wp = lds + sidx * STAGE + poffl;
      const float* up = lds + (URC ? kc * U_F : sidx * STAGE) + uoffl;
      // raw 6x6 window: middle four columns from LDS, outer two from the neighbouring lanes' quads
      f32x4 mid[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) mid[i] = *reinterpret_cast<const f32x4*>(wp + i * WWL);
      float dl[6], dr[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        dl[i] = w4_from_left(mid[i][3]);
        dr[i] = w4_from_right(mid[i][0]);
      }
      if (edge_l || edge_r) {
        // a lane at a tile-row edge reads the neighbouring quad of the halo itself: ONE 16-byte read per row serves
        // both kinds of edge lane (4-byte reads of these few lanes would all hit one bank)
        const int eo = edge_l ? -4 : 4;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(wp + i * WWL + eo);
          if (edge_l) dl[i] = q[3];
          else dr[i] = q[0];
        }
      }
      // V = B^T d B: transform along i (rows of d) for each column j, then along j
      float T[6][6], V[36];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        float col[6], out[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = j == 0 ? dl[i] : (j == 5 ? dr[i] : mid[i][j - 1]);
        w4_bt(col, out);
#pragma unroll
        for (int i = 0; i < 6; ++i) T[i][j] = out[i];
      }
      __builtin_amdgcn_sched_barrier(0);
      // U is fetched in two batches so that the register peak stays below the 256-register budget:
      // positions 0..19 now (they land under the second transform pass), 20..35 after the first 12 MFMAs
      f32x4 u0[5], u1[4];
#pragma unroll
      for (int v = 0; v < 5; ++v) u0[v] = *reinterpret_cast<const f32x4*>(up + v * 4);
#pragma unroll
      for (int xi = 0; xi < 6; ++xi) {
        float out[6];
        w4_bt(T[xi], out);
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) V[xi * 6 + nu] = out[nu];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < 12; ++p)
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0[p >> 2][p & 3], V[p], acc[p], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < 4; ++v) u1[v] = *reinterpret_cast<const f32x4*>(up + 20 + v * 4);
      if (more && !URC) stage(nkc, sidx ^ 1, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 12; p < 20; ++p)
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0[p >> 2][p & 3], V[p], acc[p], 0, 0, 0);
#pragma unroll
      for (int p = 20; p < 36; ++p)
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1[(p - 20) >> 2][p & 3], V[p], acc[p], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // every LDS read of this stage has returned; the next chunk has landed once all waves pass the barrier
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      sidx ^= 1;
    }

    // ---- output transform in registers: lane = tile t, output channels ct*16 + 4g + r --------------------------
    {
      const int pt = item / a.n_ct;
      const int b = pt / blks, rem = pt - b * blks;
      const int br = rem / a.blk_c, bc = rem - br * a.blk_c;
      const int oy = 4 * (br * TR + trow), ox = 4 * (bc * TC + tcol);
      if (oy < H && ox < W) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = ct * BM + cg * 16 + 4 * g + r;
          if (co >= a.Cout) continue;
          float R[6][4];                             // A^T along nu for every xi
#pragma unroll
          for (int xi = 0; xi < 6; ++xi) {
            float m[6];
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) m[nu] = acc[xi * 6 + nu][r];
            w4_at(m, R[xi]);
          }
          const float bs = bsv[r];
          const long o = (long)co * HW + oy * W + ox;
          float* yp = a.y + (long)b * a.y_bs + o;
          const float* rp = RES ? a.res + (long)b * a.res_bs + o : nullptr;
          float Y[4][4];                             // Y[col b][row a]: A^T along xi for every output column
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            float m[6];
#pragma unroll
            for (int xi = 0; xi < 6; ++xi) m[xi] = R[xi][cb];
            w4_at(m, Y[cb]);
          }
#pragma unroll
          for (int ra = 0; ra < 4; ++ra) {
            if (oy + ra >= H) continue;
            f32x4 o4 = {Y[0][ra] + bs, Y[1][ra] + bs, Y[2][ra] + bs, Y[3][ra] + bs};
            f32x4 rr = {0.f, 0.f, 0.f, 0.f};
            if (RES) rr = *reinterpret_cast<const f32x4*>(rp + ra * W);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (RES == 1) o4[q] += rr[q];
              if (ACT == 1) o4[q] = fmaxf(o4[q], 0.0f);
              if (ACT == 2) o4[q] = sigmoidf_(o4[q]);
              if (RES == 2) o4[q] += rr[q];
            }
            *reinterpret_cast<f32x4*>(yp + ra * W) = o4;
          }
        }
      }
    }
    if (!more) break;
    item += G;
  }
}

template <int ACT, int RES, int NCG>
static int launch_wino4(Wino4Args& a, int tile, hipStream_t st) {
  auto go = [&](auto kern, int TC) -> int {
    const int TR = 64 / TC;
    a.n_ct = a.Cout_pad / (16 * NCG);
    a.blk_r = cdiv(cdiv(a.H, 4), TR);
    a.blk_c = cdiv(cdiv(a.W, 4), TC);
    a.n_pt = a.B * a.blk_r * a.blk_c;
    static int resident_by_tile[SGTA_MAX_DEVICES][4] = {};    // per <ACT, RES> instance, device and tile shape
    int& resident = resident_by_tile[sgta_device_slot()][tile];
    if (!resident) {
      int dev = 0, cus = 0, per_cu = 0;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256 * NCG, 0) != hipSuccess || cus <= 0 || per_cu <= 0)
        return SGTA_EBADARG;
      resident = cus * per_cu;
    }
    const long items = (long)a.n_pt * a.n_ct;
    int grid = (int)(items < resident ? items : resident);
    grid = cdiv(grid, 8) * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * NCG), 0, st, a);
    SGTA_LAUNCH_CHECK();
    return SGTA_OK;
  };
  // tile t in {0,1,2,3} = 2x32 / 4x16 / 8x8 / 16x4 blocks of 4x4 output tiles
  if (NCG == 2 && ACT == 1 && RES == 0 && tile == 0 && a.Cin == 16 && a.Cout_pad == 32) {
    static const int l0 = sgta_tool_env("SGTA_L0_FORM", 1);
    if (l0 == 1 && a.Cout <= 16) {
      // 4-wave work-groups of 16 output channels, two per CU (16x64-pixel blocks keep ring + resident U under 80 KB):
      // no idle channel group, 0.62-0.66 ms against 0.72 for the 8-wave form (SGTA_L0_FORM=0) on level0
      auto go1 = [&](auto kern, int TC) -> int {
        const int TR = 64 / TC;
        a.n_ct = 1;
        a.blk_r = cdiv(cdiv(a.H, 4), TR);
        a.blk_c = cdiv(cdiv(a.W, 4), TC);
        a.n_pt = a.B * a.blk_r * a.blk_c;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
          return SGTA_EBADARG;
        const long items = a.n_pt;
        int grid = (int)(items < 2L * cus ? items : 2L * cus);
        grid = cdiv(grid, 8) * 8;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, a);
        SGTA_LAUNCH_CHECK();
        return SGTA_OK;
      };
      return go1(conv3x3_wino4_kernel<16, 1, 0, 1, 4>, 16);
    }
    return go(conv3x3_wino4_kernel<32, 1, 0, 2, 4>, 32);   // shallow full-resolution layer: U resident in LDS
  }
  switch (tile) {
    case 0: return go(conv3x3_wino4_kernel<32, ACT, RES, NCG>, 32);
    case 1: return go(conv3x3_wino4_kernel<16, ACT, RES, NCG>, 16);
    case 2: return go(conv3x3_wino4_kernel<8, ACT, RES, NCG>, 8);
    case 3: return go(conv3x3_wino4_kernel<4, ACT, RES, NCG>, 4);
  }
  return SGTA_EBADARG;
}

extern "C" int sgta_conv3x3_wino4_f32(const float* x, long x_bs, const float* u, const float* bias,
                                      const float* res, long res_bs, float* y, long y_bs, int B, int Cin,
                                      int H, int W, int Cout, int Cout_pad, int act, int res_mode, int tile,
                                      hipStream_t st) {
  if (!x || !u || !bias || !y || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return SGTA_EBADARG;
  if (Cin % 4 || Cout_pad % 16 || Cout_pad < Cout || tile < 0 || tile > 3) return SGTA_EBADARG;
  if ((res_mode != 0) != (res != nullptr)) return SGTA_EBADARG;
  // 16-byte window DMA, row stores and residual loads: image rows and planes must be 16-byte aligned
  if (W % 4 || x_bs % 4 || y_bs % 4 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)bias & 15))
    return SGTA_EBADARG;
  if (res && (res_bs % 4 || ((uintptr_t)res & 15))) return SGTA_EBADARG;
  // buffer descriptors address one image / the packed U in 31 bits
  if ((long)Cin * H * W * 4 > 0x7fffffffL || (long)Cin * Cout_pad * 36 * 4 > 0x7fffffffL) return SGTA_EBADARG;
  Wino4Args a{};
  a.x = x; a.x_bs = x_bs; a.u = u; a.bias = bias; a.res = res; a.res_bs = res_bs; a.y = y; a.y_bs = y_bs;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Cout_pad = Cout_pad; a.nchunks = Cin / 4;
  // two channel groups per work-group (shared window: +5-10 % measured on every layer shape of the workload)
  // whenever the layer has them; SGTA_WINO4_NCG=1 forces the 4-wave form
  static const bool allow_wide = sgta_tool_env("SGTA_WINO4_NCG", 2) != 1;
  const bool wide = allow_wide && (Cout_pad % 32 == 0);
  if (act == 1 && res_mode == 0) return wide ? launch_wino4<1, 0, 2>(a, tile, st) : launch_wino4<1, 0, 1>(a, tile, st);
  if (act == 0 && res_mode == 0) return wide ? launch_wino4<0, 0, 2>(a, tile, st) : launch_wino4<0, 0, 1>(a, tile, st);
  if (act == 1 && res_mode == 1) return wide ? launch_wino4<1, 1, 2>(a, tile, st) : launch_wino4<1, 1, 1>(a, tile, st);
  if (act == 1 && res_mode == 2) return wide ? launch_wino4<1, 2, 2>(a, tile, st) : launch_wino4<1, 2, 1>(a, tile, st);
  if (act == 2 && res_mode == 0) return wide ? launch_wino4<2, 0, 2>(a, tile, st) : launch_wino4<2, 0, 1>(a, tile, st);
  return SGTA_EBADARG;
}
