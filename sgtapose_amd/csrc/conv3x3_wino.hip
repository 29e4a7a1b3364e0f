// Winograd F(2x2, 3x3) convolution for gfx950 (3x3, stride 1, pad 1, + folded BN bias, residual, activation;
// BasicBlocks, heads, DCN offset conv: reference dla.py:41-69, base_model.py:121-135): 16 multiplies per 2x2
// output tile and channel instead of 36, i.e. 2.25x fewer f32 MFMA cycles than any direct / implicit-GEMM
// formulation -- the f32 matrix pipe (157 TFLOP/s) bounds this path, so the arithmetic itself is the lever.
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A      per 2x2 output tile (d = 4x4 input tile)
//
// The 16 element-wise products are 16 independent GEMMs  M_p[co][tile] = sum_c U_p[co][c] V_p[c][tile]:
//
//   a wave owns 16 tiles x 32 output channels for ALL 16 Winograd positions: 32 accumulators of the
//   f32 MFMA 16x16x4 (128 registers).  Lane (g = lane / 16, t = lane % 16) reads the raw 4x4 input window of
//   tile t / channel 4s+g ONCE (16 LDS reads), forms the 16 transformed values B^T d B in registers (32 adds)
//   and feeds them to 32 MFMAs -- 1 LDS read per position instead of 4.  After the K loop each lane already
//   holds the 16 position values of its (tile, 4 output channels): the output transform A^T m A, bias,
//   residual, activation and the stores happen in registers, with no LDS round trip and no barrier.
//
// A work-group is only 4 waves (64 tiles x 32 output channels, <= 80 KB of LDS, 256 registers), so TWO
// independent, PERSISTENT work-groups share a CU and one's epilogue hides behind the other's MFMA loop.
// U = G g G^T comes host-packed as [Cin/8][8][Cout_pad][16 positions + 4 pad]: a lane's A operands of a
// k-step are 8 conflict-free ds_read_b128 (80-byte rows).  Window (stored de-interleaved [even | odd] columns so
// that the stride-2 tile pattern reads unit-stride) and U chunk arrive by buffer LDS-DMA through a 2-stage ring.
// fp32 throughout: the transforms only add and halve, error stays at the 1e-6 level (<< 1e-3).
#include "common.h"
#include <cstdlib>

struct WinoArgs {
  const float* x;  long x_bs;
  const float* u;                  // [Cin/8][8][Cout_pad][20]
  const float* bias;               // [Cout_pad]
  const float* res; long res_bs;
  float* y;        long y_bs;
  int B, Cin, H, W, Cout, Cout_pad;
  int nchunks, blk_r, blk_c, n_ct, n_pt, vec;      // vec: 4 / 2 / 1 pixels per store
  int ksplit;                      // > 1: item = (block, channel group, K slice); raw partial sums go to ``part``
  float* part;                     // [ksplit][B][Cout][H][W]
  int ni, tri;                     // ni > 1: a block stacks ni images of tri tile rows each (small maps), else tri = TR
};

// padded, de-interleaved window row length: 2*WWP == 32/(32/TC) (mod 32) puts the tile rows of a 16-lane group on
// disjoint banks (TC = 4: WWP = 18, TC = 8: 20, TC = 16: 40, TC = 32: 66)
__host__ __device__ constexpr int wino_wwp(int TC) { return TC == 4 ? 18 : (TC == 8 ? 20 : (TC == 16 ? 40 : 2 * TC + 2)); }
// channel plane stride == 16 (mod 32): the two channels of a 32-lane group use disjoint bank halves
__host__ __device__ constexpr int wino_pad16(int w) { return (w % 32) <= 16 ? w + 16 - (w % 32) : w + 48 - (w % 32); }

// value of the horizontally adjacent lane (lane ^ 1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float wino_swap_adjacent(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// NCG = output-channel groups (of 32) per work-group: 1 -> 4 waves, two work-groups per CU; 2 -> 8 waves, one
// work-group per CU whose two channel groups share the staged window (fewer bytes through the memory pipeline per MFMA:
// the measured limiter beside MFMA issue).
template <int TC, int ACT, int RES, int NCG>
__global__ __launch_bounds__(256 * NCG, NCG == 1 ? 2 : 1) void conv3x3_wino_kernel(WinoArgs a) {
  constexpr int NT = 256 * NCG, BM = 32 * NCG, CK = 8, TR = 64 / TC;
  constexpr int WH = 2 * TR + 2, WW = 2 * TC + 2, NE = WW / 2, WWP = wino_wwp(TC);
  constexpr int WSZ = WH * WWP, WSZP = wino_pad16(WSZ), ROUNDS = (WSZP + 255) / 256;
  constexpr int LDU = 20, U_F = CK * BM * LDU, U_PER = U_F / 4 / NT;
  static_assert(U_F % (NT * 4) == 0, "U stage must be a whole number of NT x 16-byte slots");
  constexpr int CPW = CK / NCG;                      // window channels staged by each 4-wave half of the work-group
  constexpr int STAGE = CK * WSZP + U_F;
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  // PERSISTENT work-groups: gridDim.x = 2 x CUs; logical ids are consecutive inside an XCD, and a work-group
  // walks the items id, id + G, id + 2G ... (item = (pixel block, 32-channel group), channel group fastest: the
  // groups of one pixel block run side by side on one XCD and share the window in its L2).  The (item, chunk)
  // pairs form ONE continuous stream through the 2-stage LDS ring: chunk 0 of the next item is in flight while
  // the last chunk and the register epilogue of the current one run.
  const int G = gridDim.x;
  const int wg = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int n_items = a.n_pt * a.n_ct * a.ksplit;
  if (wg >= n_items) return;
  const int nch = a.nchunks / a.ksplit;              // chunks per item (split-K: a slice of the input channels)
  int kbase = 0;                                     // first chunk of the item being STAGED
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave >> 2, tgw = wave & 3;          // channel group / tile group of this wave
  const int tid4 = tid & 255;                        // thread id inside the 4-wave half (window staging)
  const int g = lane >> 4, tl = lane & 15;
  const int H = a.H, W = a.W, HW = H * W;
  const int blks = a.blk_r * a.blk_c;
  const int t = tgw * 16 + tl;                       // this lane's tile inside the 64-tile block
  const int trow = t / TC, tcol = t % TC;
  const int tbase = (2 * trow) * WWP + tcol;
  const int poffl = tbase + g * WSZP, uoffl = CK * WSZP + (g * BM + cg * 32 + tl) * LDU;

  // ---- LDS-DMA staging through buffer loads: per-slot byte offsets are chunk- and channel-invariant VGPRs,
  // the channel / chunk advance is a scalar offset, and out-of-image slots carry an out-of-range offset that
  // the buffer bounds check turns into zeros: no vector instruction is spent on addressing inside the loop.
  unsigned poff[ROUNDS], uoff[U_PER];
  int slot_r[ROUNDS], slot_c[ROUNDS];                // item-invariant window coordinates of this thread's slots
#pragma unroll
  for (int q = 0; q < ROUNDS; ++q) {
    // the last round is shifted back to end exactly at WSZP (it re-writes a few slots of the previous round
    // with identical data): every DMA runs with all 256 lanes, no divergent control flow around M0
    const int e = tid4 + (q < ROUNDS - 1 ? q * 256 : WSZP - 256);
    const int r = e / WWP, sx = e - r * WWP;
    const bool ok = e < WSZ && sx < WW;
    slot_r[q] = ok ? r - 1 : -0x10000;
    slot_c[q] = (sx < NE ? 2 * sx : 2 * (sx - NE) + 1) - 1;
  }
  int uslot[U_PER];
#pragma unroll
  for (int q = 0; q < U_PER; ++q) {
    const int e = tid + q * NT;                      // 16-byte slot: channel c = e / (BM*5), then (co, quarter)
    const int c = e / (BM * 5), rem = e - c * (BM * 5);
    uslot[q] = (c * a.Cout_pad * LDU + rem * 4) * 4;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t xrs;
  const __amdgpu_buffer_rsrc_t urs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), (short)0, a.Cin * a.Cout_pad * LDU * 4, 0x00020000);
#endif
  // per-item staging parameters (scalar item -> image, block origin, channel group)
  auto setup = [&](int item) {
    const int ks = item % a.ksplit, rest = item / a.ksplit;
    const int pt = rest / a.n_ct, ct = rest - pt * a.n_ct;
    kbase = ks * nch;
    int b, ox0;
    if (a.ni > 1) {
      // STACKED small maps (H <= 2*tri - 1, blk_r == 1): image si of the block owns window rows si*2*tri .. (si+1)*2*tri;
      // its last row (gy = 2*tri - 1 >= H) is the zero padding that image si + 1 needs above its first row, so the
      // window keeps its 2*TR + 2 rows and a tile's LDS base stays (2*trow)*WWP -- only the DMA sources change
      const int bq = pt / a.blk_c, bc = pt - bq * a.blk_c;
      b = bq * a.ni;
      ox0 = 2 * bc * TC;
      const int S = 2 * a.tri;
#pragma unroll
      for (int q = 0; q < ROUNDS; ++q) {
        const int r = slot_r[q] + 1;                  // window row (negative: slot outside the window)
        const int si = r / S, gy = r - si * S - 1, gx = ox0 + slot_c[q];
        const bool ok = r >= 0 && si < a.ni && b + si < a.B && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        poff[q] = ok ? (unsigned)(si * a.x_bs + gy * W + gx) * 4u : 0x80000000u;
      }
    } else {
      b = pt / blks;
      const int rem = pt - b * blks;
      const int br = rem / a.blk_c, bc = rem - br * a.blk_c;
      const int oy0 = 2 * br * TR;
      ox0 = 2 * bc * TC;
#pragma unroll
      for (int q = 0; q < ROUNDS; ++q) {
        const int gy = oy0 + slot_r[q], gx = ox0 + slot_c[q];
        const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        poff[q] = ok ? (unsigned)(gy * W + gx) * 4u : 0x80000000u;
      }
    }
#pragma unroll
    for (int q = 0; q < U_PER; ++q) uoff[q] = (unsigned)(uslot[q] + ct * BM * LDU * 4);
#if defined(__HIP_DEVICE_COMPILE__)
    xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (long)b * a.x_bs), (short)0,
                                            (int)(((a.ni - 1) * a.x_bs + (long)a.Cin * HW) * 4), 0x00020000);
#endif
  };
  // one chunk = three DMA batches (window channels 0-3, window channels 4-7, U) so that the issue can be
  // spread between the MFMA groups of the previous chunk instead of one 21-instruction burst behind the barrier
  auto stage = [&](int kc, int st, int part) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (part < 2) {
      // each 4-wave half of the work-group stages CPW of the chunk's channels (all 8 when NCG == 1)
      float* dst = lds + st * STAGE + tgw * 64;
#pragma unroll
      for (int cc = part * (CPW / 2); cc < (part + 1) * (CPW / 2); ++cc) {
        const int c = cg * CPW + cc;
#pragma unroll
        for (int q = 0; q < ROUNDS; ++q)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(dst + c * WSZP + (q < ROUNDS - 1 ? q * 256 : WSZP - 256)),
                                                   4, poff[q], ((kbase + kc) * CK + c) * HW * 4, 0, 0);
      }
    } else {
      float* udst = lds + st * STAGE + CK * WSZP + wave * 256;
#pragma unroll
      for (int q = 0; q < U_PER; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(urs, (lds_ptr_t)(udst + q * NT * 4), 16, uoff[q],
                                                 (kbase + kc) * CK * a.Cout_pad * LDU * 4, 0, 0);
    }
#endif
  };

  f32x4 acc[16][2];
  // operand registers, software-pipelined by hand: positions 0..7 use uA, 8..15 use uB; d = raw 4x4 window
  f32x4 uA[2][2], uB[2][2];
  float d[4][4], V[16];
  auto load_d = [&](const float* prow) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) d[r][j] = prow[r * WWP + (j & 1) * NE + (j >> 1)];
  };
  auto load_u = [&](f32x4 (&u)[2][2], const float* urow, int half) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int v = 0; v < 2; ++v) u[i][v] = *reinterpret_cast<const f32x4*>(urow + i * 16 * LDU + (half * 2 + v) * 4);
  };
  // V = B^T d B :  rows then columns (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1])
  auto transform = [&]() {
    float tr[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tr[0][j] = d[0][j] - d[2][j];
      tr[1][j] = d[1][j] + d[2][j];
      tr[2][j] = d[2][j] - d[1][j];
      tr[3][j] = d[1][j] - d[3][j];
    }
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      V[xi * 4 + 0] = tr[xi][0] - tr[xi][2];
      V[xi * 4 + 1] = tr[xi][1] + tr[xi][2];
      V[xi * 4 + 2] = tr[xi][2] - tr[xi][1];
      V[xi * 4 + 3] = tr[xi][1] - tr[xi][3];
    }
  };
  auto mfma8 = [&](const f32x4 (&u)[2][2], int half) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc[half * 8 + q][i] =
            __builtin_amdgcn_mfma_f32_16x16x4f32(u[i][q >> 2][q & 3], V[half * 8 + q], acc[half * 8 + q][i], 0, 0, 0);
  };

  int item = wg, sidx = 0;
  setup(item);
  stage(0, 0, 0); stage(0, 0, 1); stage(0, 0, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  load_d(lds + poffl);
  load_u(uA, lds + uoffl, 0);

  while (true) {
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[p][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ks = item % a.ksplit, ct = (item / a.ksplit) % a.n_ct;
    f32x4 bsv[2];                                     // bias of this lane's 8 output channels (i*16 + 4g + r)
    bool more = true;
    for (int kc = 0; kc < nch; ++kc) {
      const bool last = kc + 1 >= nch;
      int nkc = kc + 1;
      if (last) {
        // last chunk of the item: its staging registers are free -> point them at the next item and put that
        // item's first chunk in flight; fetch the bias for the epilogue
        more = item + G < n_items;
        nkc = 0;
        if (more) setup(item + G);
#pragma unroll
        for (int i = 0; i < 2; ++i)
          bsv[i] = a.ksplit > 1 ? f32x4{0.f, 0.f, 0.f, 0.f}
                                : *reinterpret_cast<const f32x4*>(a.bias + ct * BM + cg * 32 + i * 16 + 4 * g);
      }
      const bool dma = more;
      if (dma) stage(nkc, sidx ^ 1, 0);
      const float* P0 = lds + sidx * STAGE + poffl;
      const float* U0 = lds + sidx * STAGE + uoffl;
      // ---- k-step 0 (channels g of the chunk) ----
      load_u(uB, U0, 1);
      transform();
      __builtin_amdgcn_sched_barrier(0);
      mfma8(uA, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (dma) stage(nkc, sidx ^ 1, 1);
      load_d(P0 + 4 * WSZP);                                // k-step 1 operands fly under the next 16 MFMAs
      load_u(uA, U0 + 4 * BM * LDU, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma8(uB, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (dma) stage(nkc, sidx ^ 1, 2);
      // ---- k-step 1 (channels 4 + g) ----
      load_u(uB, U0 + 4 * BM * LDU, 1);
      transform();
      __builtin_amdgcn_sched_barrier(0);
      mfma8(uA, 0);
      __builtin_amdgcn_sched_barrier(0);
      // every LDS read of this stage has returned; the next chunk has landed once all waves pass the barrier
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      sidx ^= 1;
      if (more) {                                           // first operands of the next chunk (or next item)
        load_d(lds + sidx * STAGE + poffl);
        load_u(uA, lds + sidx * STAGE + uoffl, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma8(uB, 1);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- output transform in registers: lane = tile t, output channels i*16 + 4g + r -----------------------
    {
      const int pt = item / (a.ksplit * a.n_ct);
      int b, oy, ox;
      if (a.ni > 1) {                                // stacked small maps: the tile row selects the image
        const int bq = pt / a.blk_c, bc = pt - bq * a.blk_c, si = trow / a.tri;
        b = bq * a.ni + si;
        oy = b < a.B ? 2 * (trow - si * a.tri) : H;  // images beyond the batch: nothing to store
        ox = 2 * bc * TC + 2 * tcol;
        if (b >= a.B) b = a.B - 1;
      } else {
        b = pt / blks;
        const int rem = pt - b * blks;
        const int br = rem / a.blk_c, bc = rem - br * a.blk_c;
        oy = 2 * br * TR + 2 * trow;
        ox = 2 * bc * TC + 2 * tcol;
      }
      // split-K: raw partial sums (no bias; launched as ACT = RES = 0) go to slice ks of the workspace
      float* const ybase = a.ksplit > 1 ? a.part + (long)ks * a.B * a.Cout * HW : a.y;
      const long ybs = a.ksplit > 1 ? (long)a.Cout * HW : a.y_bs;
      if (a.vec == 4) {
        // 16-byte stores: the lanes of two horizontally adjacent tiles trade halves, the even one stores the
        // four pixels of the upper row, the odd one those of the lower row (W % 4 == 0: a pair is inside the
        // image or outside it together)
        const bool odd = tl & 1;
        const int row = odd ? oy + 1 : oy, col = odd ? ox - 2 : ox;
        const bool ok = oy < H && ox < W && row < H;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = ct * BM + cg * 32 + i * 16 + 4 * g + r;
            if (co >= a.Cout) continue;
            float ra[4], rb[4];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
              ra[xi] = (acc[xi * 4 + 0][i][r] + acc[xi * 4 + 1][i][r]) + acc[xi * 4 + 2][i][r];
              rb[xi] = (acc[xi * 4 + 1][i][r] - acc[xi * 4 + 2][i][r]) - acc[xi * 4 + 3][i][r];
            }
            const float bs = bsv[i][r];
            const float v0 = (ra[0] + ra[1]) + ra[2] + bs, v1 = (rb[0] + rb[1]) + rb[2] + bs;
            const float v2 = (ra[1] - ra[2]) - ra[3] + bs, v3 = (rb[1] - rb[2]) - rb[3] + bs;
            const float g0 = wino_swap_adjacent(odd ? v0 : v2), g1 = wino_swap_adjacent(odd ? v1 : v3);
            f32x4 o4 = odd ? f32x4{g0, g1, v2, v3} : f32x4{v0, v1, g0, g1};
            if (!ok) continue;
            const long o = (long)co * HW + row * W + col;
            f32x4 rr = {0.f, 0.f, 0.f, 0.f};
            if (RES) rr = *reinterpret_cast<const f32x4*>(a.res + (long)b * a.res_bs + o);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (RES == 1) o4[q] += rr[q];
              if (ACT == 1) o4[q] = fmaxf(o4[q], 0.0f);
              if (ACT == 2) o4[q] = sigmoidf_(o4[q]);
              if (RES == 2) o4[q] += rr[q];
            }
            *reinterpret_cast<f32x4*>(ybase + (long)b * ybs + o) = o4;
          }
      } else if (oy < H && ox < W) {
        const bool x1 = ox + 1 < W, y1 = oy + 1 < H;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = ct * BM + cg * 32 + i * 16 + 4 * g + r;
            if (co >= a.Cout) continue;
            float ra[4], rb[4];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
              ra[xi] = (acc[xi * 4 + 0][i][r] + acc[xi * 4 + 1][i][r]) + acc[xi * 4 + 2][i][r];
              rb[xi] = (acc[xi * 4 + 1][i][r] - acc[xi * 4 + 2][i][r]) - acc[xi * 4 + 3][i][r];
            }
            const float bs = bsv[i][r];
            float v[4] = {(ra[0] + ra[1]) + ra[2] + bs, (rb[0] + rb[1]) + rb[2] + bs,
                          (ra[1] - ra[2]) - ra[3] + bs, (rb[1] - rb[2]) - rb[3] + bs};
            const long o = (long)co * HW + oy * W + ox;
            float* yp = ybase + (long)b * ybs + o;
            float rr4[4] = {0.f, 0.f, 0.f, 0.f};
            if (RES) {
              const float* rp = a.res + (long)b * a.res_bs + o;
              if (a.vec == 2) {
                const float2 q0 = *reinterpret_cast<const float2*>(rp);
                rr4[0] = q0.x; rr4[1] = q0.y;
                if (y1) { const float2 q1 = *reinterpret_cast<const float2*>(rp + W); rr4[2] = q1.x; rr4[3] = q1.y; }
              } else {
                rr4[0] = rp[0];
                if (x1) rr4[1] = rp[1];
                if (y1) rr4[2] = rp[W];
                if (x1 && y1) rr4[3] = rp[W + 1];
              }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (RES == 1) v[q] += rr4[q];
              if (ACT == 1) v[q] = fmaxf(v[q], 0.0f);
              if (ACT == 2) v[q] = sigmoidf_(v[q]);
              if (RES == 2) v[q] += rr4[q];
            }
            if (a.vec == 2) {
              *reinterpret_cast<float2*>(yp) = float2{v[0], v[1]};
              if (y1) *reinterpret_cast<float2*>(yp + W) = float2{v[2], v[3]};
            } else {
              yp[0] = v[0];
              if (x1) yp[1] = v[1];
              if (y1) yp[W] = v[2];
              if (x1 && y1) yp[W + 1] = v[3];
            }
          }
      }
    }
    if (!more) break;
    item += G;
  }
}

// split-K epilogue: y = bias + sum_s part[s] in a fixed order (deterministic)
__global__ __launch_bounds__(256) void wino_splitk_reduce_kernel(const float* __restrict__ part, int S,
                                                                 const float* __restrict__ bias, float* __restrict__ y,
                                                                 long y_bs, int Cout, int HW, long per_slice) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= per_slice) return;
  const int p = (int)(i % HW);
  const long t = i / HW;
  const int co = (int)(t % Cout), b = (int)(t / Cout);
  float v = bias[co];
  for (int s = 0; s < S; ++s) v += part[s * per_slice + i];
  y[b * y_bs + (long)co * HW + p] = v;
}

template <int ACT, int RES, int NCG>
static int launch_wino(WinoArgs& a, int tile, long a_ws_floats, hipStream_t st) {
  auto go = [&](auto kern, int TC) -> int {
    const int TR = 64 / TC;
    a.n_ct = a.Cout_pad / (32 * NCG);
    a.blk_r = cdiv(cdiv(a.H, 2), TR);
    a.blk_c = cdiv(cdiv(a.W, 2), TC);
    a.n_pt = a.B * a.blk_r * a.blk_c;
    // small maps: ni images share a block, tri = TR / ni tile rows each (needs H <= 2*tri - 1, see the kernel's setup)
    a.ni = 1;
    a.tri = TR;
    for (int n = 4; n >= 2; n >>= 1)
      if (TR % n == 0 && a.H <= 2 * (TR / n) - 1 && a.B >= 2 && ((n - 1) * a.x_bs + (long)a.Cin * a.H * a.W) * 4 <= 0x7fffffffL) {
        a.ni = n;
        a.tri = TR / n;
        a.n_pt = cdiv(a.B, n) * a.blk_c;
        break;
      }
    // persistent: as many work-groups as the device holds at once (2 per CU), never more than items
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
    long items = (long)a.n_pt * a.n_ct;
    // split-K: a layer with few items (small map, <= 32 output channels, many input channels: the DCN offset convs
    // of levels 4-5) leaves most CUs idle; slices of the input channels become extra items, their raw partial sums
    // meet in the caller's workspace and are reduced in a fixed order
    a.ksplit = 1;
    if (ACT == 0 && RES == 0 && a.part) {
      const long per_slice = (long)a.B * a.Cout * a.H * a.W;
      for (int s = 8; s >= 2; s >>= 1)
        if (items * s <= resident && a.nchunks % s == 0 && a.nchunks / s >= 4 && per_slice * s <= a_ws_floats) {
          a.ksplit = s;
          break;
        }
    }
    items *= a.ksplit;
    int grid = (int)(items < resident ? items : resident);
    grid = cdiv(grid, 8) * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * NCG), 0, st, a);
    SGTA_LAUNCH_CHECK();
    if (a.ksplit > 1) {
      const long per_slice = (long)a.B * a.Cout * a.H * a.W;
      hipLaunchKernelGGL(wino_splitk_reduce_kernel, dim3((unsigned)((per_slice + 255) / 256)), dim3(256), 0, st, a.part,
                         a.ksplit, a.bias, a.y, a.y_bs, a.Cout, a.H * a.W, per_slice);
      SGTA_LAUNCH_CHECK();
    }
    return SGTA_OK;
  };
  // tile t in {0,1,2,3} = 2x32 / 4x16 / 8x8 / 16x4 blocks of 2x2 output tiles
  switch (tile) {
    case 0: return go(conv3x3_wino_kernel<32, ACT, RES, NCG>, 32);
    case 1: return go(conv3x3_wino_kernel<16, ACT, RES, NCG>, 16);
    case 2: return go(conv3x3_wino_kernel<8, ACT, RES, NCG>, 8);
    case 3: return go(conv3x3_wino_kernel<4, ACT, RES, NCG>, 4);
  }
  return SGTA_EBADARG;
}

extern "C" int sgta_conv3x3_wino_f32(const float* x, long x_bs, const float* u, const float* bias,
                                      const float* res, long res_bs, float* y, long y_bs, int B, int Cin,
                                      int H, int W, int Cout, int Cout_pad, int act, int res_mode, int tile,
                                      float* ws, long ws_floats, hipStream_t st) {
  if (!x || !u || !bias || !y || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return SGTA_EBADARG;
  if (Cin % 8 || Cout_pad % 32 || Cout_pad < Cout) return SGTA_EBADARG;
  if ((res_mode != 0) != (res != nullptr) || (long)Cin * H * W > 0x7fffffffL) return SGTA_EBADARG;
  // buffer descriptors (LDS-DMA with hardware bounds check) address one image / the packed U in 31 bits
  if ((long)Cin * H * W * 4 > 0x7fffffffL || (long)Cin * Cout_pad * 80 > 0x7fffffffL) return SGTA_EBADARG;
  WinoArgs a{};
  a.x = x; a.x_bs = x_bs; a.u = u; a.bias = bias; a.res = res; a.res_bs = res_bs; a.y = y; a.y_bs = y_bs;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Cout_pad = Cout_pad; a.nchunks = Cin / 8;
  a.ksplit = 1; a.part = ws;
  const long wsf = ws ? ws_floats : 0;
  // 16- / 8-byte stores and residual loads need widths and plane alignments that are multiples of 4 / 2 pixels
  auto aligned = [&](int n) {
    return (W % n == 0) && (y_bs % n == 0) && (((uintptr_t)y & (4 * n - 1)) == 0) &&
           (!res || ((res_bs % n == 0) && (((uintptr_t)res & (4 * n - 1)) == 0)));
  };
  a.vec = aligned(4) ? 4 : (aligned(2) ? 2 : 1);
  // two 32-channel groups per work-group (shared window): measured +7 % on the 120x160 maps but -5 % on the 15x20
  // ones (half the items) and neutral at 30x40 -- the maps this kernel serves -- so it is opt-in (SGTA_WINO_NCG=2)
  static const bool allow_wide = sgta_tool_env("SGTA_WINO_NCG", 1) == 2;
  const bool wide = allow_wide && (Cout_pad % 64 == 0);
  if (act == 1 && res_mode == 0) return wide ? launch_wino<1, 0, 2>(a, tile, wsf, st) : launch_wino<1, 0, 1>(a, tile, wsf, st);
  if (act == 0 && res_mode == 0) return wide ? launch_wino<0, 0, 2>(a, tile, wsf, st) : launch_wino<0, 0, 1>(a, tile, wsf, st);
  if (act == 1 && res_mode == 1) return wide ? launch_wino<1, 1, 2>(a, tile, wsf, st) : launch_wino<1, 1, 1>(a, tile, wsf, st);
  if (act == 1 && res_mode == 2) return wide ? launch_wino<1, 2, 2>(a, tile, wsf, st) : launch_wino<1, 2, 1>(a, tile, wsf, st);
  if (act == 2 && res_mode == 0) return wide ? launch_wino<2, 0, 2>(a, tile, wsf, st) : launch_wino<2, 0, 1>(a, tile, wsf, st);
  return SGTA_EBADARG;
}
