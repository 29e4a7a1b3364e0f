// Winograd F(2x2, 3x3) convolution for gfx950 (3x3, stride 1, pad 1, + folded BN bias, residual,
// activation): 16 multiplies per 2x2 output tile and channel pair instead of 36, i.e. 2.25x fewer
// f32 MFMA cycles than any direct / implicit-GEMM formulation -- the f32 matrix pipe (157 TFLOP/s) is
// what bounds this path, so the arithmetic itself is the lever.  Same layers as conv3x3_patch.hip
// (BasicBlocks, heads, DCN offset conv; reference dla.py:41-69, base_model.py:121-135).
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A      per 2x2 output tile (d = 4x4 input tile)
//
// The 16 element-wise products are 16 independent GEMMs  M_p[co][tile] = sum_c U_p[co][c] V_p[c][tile].
// A work-group = 16 waves = 1024 threads owns 64 tiles (TR x TC block = 2TR x 2TC output pixels) and
// BM output channels; WAVE p OWNS POSITION p = (xi, nu), so every transformed input value V_p is
// computed exactly once, on the fly, by the lane that feeds it to the MFMA: 4 LDS reads of the raw
// input window + 3 adds (B^T has two +-1 entries per row; the (row pair, column pair, signs) of a
// position are wave-uniform).  The lane half of the MFMA k-slot takes channel 2q / 2q+1.
// U = G g G^T is precomputed on the host.  Input window and U chunk are staged by LDS-DMA (double
// buffered, no registers).  After the K loop the 16 M_p tiles meet in LDS (the whole 160 KB is free
// then) and 1024 threads apply A^T . A, bias, residual, activation and store 2x2 pixel tiles.
// fp32 throughout: the transforms only add and halve, error stays at the 1e-6 level (<< 1e-3).
#include "common.h"

struct WinoArgs {
  const float* x;  long x_bs;
  const float* u;                  // [Cin/8][16][Cout_pad][8], channels of a chunk ordered 0,2,4,6,1,3,5,7
  const float* bias;               // [Cout_pad]
  const float* res; long res_bs;
  float* y;        long y_bs;
  int B, Cin, H, W, Cout, Cout_pad;
  int nchunks, blk_r, blk_c, n_ct, n_pt;
};

__device__ float sgta_wino_zero = 0.0f;

// 16-byte LDS-DMA.  The builtin's size-16 form only exists for gfx950; hipcc's HOST pass silently drops a
// kernel template whose body mentions it, so the host pass must not see the call.
__device__ __forceinline__ void glds16(const float* g, float* l) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
#endif
}

// padded, de-interleaved window row length: 2*WWP == 32/(32/TC) (mod 32) spreads the 32/TC tile rows of a
// 32-lane group over disjoint bank ranges (TC = 4: WWP = 18, TC = 8: 20, TC = 16: 40, TC = 32: 66)
__host__ __device__ constexpr int wino_wwp(int TC) { return TC == 4 ? 18 : (TC == 8 ? 20 : (TC == 16 ? 40 : 2 * TC + 2)); }

// One chunk of one position (xi, nu).  The position is wave-uniform but a RUN-TIME value: a 16-way
// switch over compile-time positions made hipcc spill >500 registers; uniform offsets / signs in
// scalar registers cost one add per LDS read instead.
//   xi = 0: d0 - d2 ; 1: d1 + d2 ; 2: d2 - d1 ; 3: d1 - d3      (same for nu over columns)
template <int MT, int NT, int TC, int CK, int P_PER>
__device__ __forceinline__ void wino_chunk(f32x16 (&acc)[MT][NT], const float* patch, const float* ul,
                                           const int (&tbase)[NT], int lh, int o11, int o12, int o21, int o22,
                                           float s12, float s21, float s22) {
  constexpr int TR = 64 / TC, WH = 2 * TR + 2, WSZ = WH * wino_wwp(TC);
  f32x4 a4[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) a4[i] = *reinterpret_cast<const f32x4*>(ul + i * 32 * CK);
#pragma unroll
  for (int q = 0; q < CK / 2; ++q) {
    const float* prow = patch + (2 * q + lh) * WSZ;
    float bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float* pp = prow + tbase[j];
      float v = pp[o11];
      v = fmaf(s12, pp[o12], v);
      v = fmaf(s21, pp[o21], v);
      v = fmaf(s22, pp[o22], v);
      bv[j] = v;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][q], bv[j], acc[i][j], 0, 0, 0);
  }
}

template <int MT, int TC, int CK, int ACT, int RES, int EC, int NSTG>
__device__ __forceinline__ void wino_body(const WinoArgs& a) {
  constexpr int NT = 2, BM = MT * 32, TR = 64 / TC;
  // Window rows are stored DE-INTERLEAVED ([even columns | odd columns], padded to WWP) so that the
  // stride-2 column pattern of neighbouring tiles becomes unit-stride, and WWP is chosen so that the
  // tile rows of a 32-lane group land on disjoint banks: B-operand reads are bank-conflict free.
  constexpr int WH = 2 * TR + 2, WW = 2 * TC + 2, WWP = wino_wwp(TC), WSZ = WH * WWP;
  // U rows are 8 floats = the chunk's channels in the order [0,2,4,6 | 1,3,5,7] (host packing): lane half
  // lh fetches its four k-steps with ONE ds_read_b128, and the rows are staged by 16-byte LDS-DMA
  static_assert(CK == 8, "U row layout assumes 8 channels per chunk");
  constexpr int P_TOT = CK * WSZ, U_TOT4 = 16 * BM * CK / 4;             // U in float4 slots
  constexpr int P_PER = (P_TOT + 1023) / 1024, U_PER = (U_TOT4 + 1023) / 1024;
  constexpr int STAGE = P_PER * 1024 + U_PER * 4096;
  constexpr int MLD = 65, M_TOT = 16 * EC * MLD;    // EC = output channels per epilogue pass (32 or 16)
  // NSTG = LDS-DMA ring depth: 3 = two chunks in flight behind the one in use, 2 = one
  constexpr int LDS_F = (NSTG * STAGE > M_TOT ? NSTG * STAGE : M_TOT);
  __shared__ __attribute__((aligned(16))) float lds[LDS_F + BM];
  float* const sbias = lds + LDS_F;

  int pt, ct;
  xcd_remap(blockIdx.x, a.n_ct, pt, ct);
  if (pt >= a.n_pt) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = Winograd position p
  const int l31 = lane & 31, lh = lane >> 5;
  const int H = a.H, W = a.W, HW = H * W;
  const int blks = a.blk_r * a.blk_c;
  const int b = pt / blks;
  const int br = (pt - b * blks) / a.blk_c, bc = (pt - b * blks) % a.blk_c;
  const int oy0 = 2 * br * TR, ox0 = 2 * bc * TC;                  // output origin of the block
  const float* xb = a.x + (long)b * a.x_bs;
  if (tid < BM) sbias[tid] = a.bias[ct * BM + tid];

  // ---- LDS-DMA staging: per-slot source offsets are chunk-invariant (-1 = reads as zero) ------------
  int poff[P_PER], uoff[U_PER];
#pragma unroll
  for (int t = 0; t < P_PER; ++t) {
    const int e = tid + t * 1024;
    const int ch = e / WSZ, r = e - ch * WSZ;
    const int sx = r % WWP;                        // slot in the de-interleaved row
    const int col = sx < WW / 2 ? 2 * sx : 2 * (sx - WW / 2) + 1;
    const int gy = oy0 - 1 + r / WWP, gx = ox0 - 1 + col;
    const bool ok = e < P_TOT && sx < WW && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    poff[t] = ok ? ch * HW + gy * W + gx : -1;
  }
#pragma unroll
  for (int t = 0; t < U_PER; ++t) {
    const int e = tid + t * 1024;                  // float4 slot: row = e/2 = p*BM + co, half = e&1
    const int row = e >> 1, p = row / BM, co = row - p * BM;
    uoff[t] = e < U_TOT4 ? (p * a.Cout_pad + ct * BM + co) * CK + (e & 1) * 4 : 0;
  }
  auto stage = [&](int kc, int st) {
    float* dst = lds + st * STAGE + wave * 64;
    const float* xc = xb + (long)kc * CK * HW;
#pragma unroll
    for (int t = 0; t < P_PER; ++t)
      __builtin_amdgcn_global_load_lds(poff[t] >= 0 ? xc + poff[t] : &sgta_wino_zero, dst + t * 1024, 4, 0, 0);
    const float* uc = a.u + (long)kc * 16 * a.Cout_pad * CK;
    float* udst4 = lds + st * STAGE + P_PER * 1024 + wave * 256;
#pragma unroll
    for (int t = 0; t < U_PER; ++t)
      if (U_TOT4 % 1024 == 0 || tid + t * 1024 < U_TOT4)
        glds16(uc + uoff[t], udst4 + t * 4096);
  };
  stage(0, 0);
  if (NSTG == 3 && a.nchunks > 1) stage(1, 1);

  const int xi = wave >> 2, nu = wave & 3;          // this wave's Winograd position
  const int i1 = xi == 0 ? 0 : (xi == 2 ? 2 : 1), i2 = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
  const int j1 = nu == 0 ? 0 : (nu == 2 ? 2 : 1), j2 = nu == 0 ? 2 : (nu == 1 ? 2 : (nu == 2 ? 1 : 3));
  const float si2 = xi == 1 ? 1.0f : -1.0f, sj2 = nu == 1 ? 1.0f : -1.0f;     // first entries are +1
  const int c1 = (j1 & 1) * (WW / 2) + (j1 >> 1), c2 = (j2 & 1) * (WW / 2) + (j2 >> 1);
  const int o11 = i1 * WWP + c1, o12 = i1 * WWP + c2, o21 = i2 * WWP + c1, o22 = i2 * WWP + c2;
  const float s12 = sj2, s21 = si2, s22 = si2 * sj2;
  int tbase[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = j * 32 + l31;
    tbase[j] = (2 * (t / TC)) * WWP + (t % TC);
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // counted waits + raw barriers: __syncthreads() would drain every LDS-DMA in flight (vmcnt(0))
  constexpr int G = P_PER + U_PER;                 // LDS-DMA instructions per wave and stage
  if (NSTG == 3 && a.nchunks > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): sbias store
  __builtin_amdgcn_s_barrier();

  int st = 0;
  for (int kc = 0; kc < a.nchunks; ++kc) {
    const bool issue = kc + (NSTG - 1) < a.nchunks;
    if (issue) stage(kc + NSTG - 1, NSTG == 3 ? (st >= 1 ? st - 1 : 2) : (st ^ 1));   // (st + NSTG - 1) % NSTG
    const float* patch = lds + st * STAGE;
    wino_chunk<MT, NT, TC, CK, P_PER>(acc, patch, patch + P_PER * 1024 + (wave * BM + l31) * CK + lh * 4, tbase, lh,
                                      o11, o12, o21, o22, s12, s21, s22);
    // chunk kc+1 must have landed; chunk kc+2 (just issued) may stay in flight
    if (NSTG == 3 && issue) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    st = st == NSTG - 1 ? 0 : st + 1;
  }

  // ---- output transform through LDS: Mbuf[p][co_local 0..31][tile 0..63] ---------------------------
  float* Mbuf = lds;
  constexpr int RPP = EC / 2;                      // accumulator registers per 32-row tile and pass
#pragma unroll
  for (int e = 0; e < BM / EC; ++e) {
    const int i = (e * EC) / 32, r0 = ((e * EC) % 32) / 2;     // MFMA row tile and first register of the pass
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < RPP; ++r) {
        const int rr = r0 + r;
        const int col = (rr & 3) + 8 * ((rr >> 2) - (r0 >> 2)) + 4 * lh;       // row inside this pass
        Mbuf[(wave * EC + col) * MLD + j * 32 + l31] = acc[i][j][rr];
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int k = 0; k < EC * 64 / 1024; ++k) {
      const int item = tid + k * 1024;
      const int t = item & 63, col = item >> 6;
      const int co = ct * BM + e * EC + col;
      float m[4][4];
#pragma unroll
      for (int p = 0; p < 16; ++p) m[p >> 2][p & 3] = Mbuf[(p * EC + col) * MLD + t];
      float ra[4], rb[4];                          // A^T applied along nu
#pragma unroll
      for (int x4 = 0; x4 < 4; ++x4) {
        ra[x4] = (m[x4][0] + m[x4][1]) + m[x4][2];
        rb[x4] = (m[x4][1] - m[x4][2]) - m[x4][3];
      }
      float y00 = (ra[0] + ra[1]) + ra[2], y01 = (rb[0] + rb[1]) + rb[2];
      float y10 = (ra[1] - ra[2]) - ra[3], y11 = (rb[1] - rb[2]) - rb[3];
      const int oy = oy0 + 2 * (t / TC), ox = ox0 + 2 * (t % TC);
      if (co < a.Cout && oy < H && ox < W) {
        const float bsv = sbias[e * EC + col];
        float* yp = a.y + (long)b * a.y_bs + (long)co * HW + oy * W + ox;
        const float* rp = RES ? a.res + (long)b * a.res_bs + (long)co * HW + oy * W + ox : nullptr;
        const bool x1 = ox + 1 < W, y1 = oy + 1 < H;
        float v[4] = {y00 + bsv, y01 + bsv, y10 + bsv, y11 + bsv};
        float rr4[4] = {0, 0, 0, 0};
        if (RES) {
          rr4[0] = rp[0];
          if (x1) rr4[1] = rp[1];
          if (y1) rr4[2] = rp[W];
          if (x1 && y1) rr4[3] = rp[W + 1];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (RES == 1) v[q] += rr4[q];
          if (ACT == 1) v[q] = fmaxf(v[q], 0.0f);
          if (ACT == 2) v[q] = sigmoidf_(v[q]);
          if (RES == 2) v[q] += rr4[q];
        }
        yp[0] = v[0];
        if (x1) yp[1] = v[1];
        if (y1) yp[W] = v[2];
        if (x1 && y1) yp[W + 1] = v[3];
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
  }
}

// 64 output channels per work-group: one 16-wave work-group per CU (<= 128 registers, 3-deep LDS-DMA ring)
template <int TC, int CK, int ACT, int RES>
__global__ __launch_bounds__(1024) void conv3x3_wino_kernel(WinoArgs a) {
  wino_body<2, TC, CK, ACT, RES, 32, 3>(a);
}

// 32 output channels per work-group, sized (<= 64 registers, <= 80 KB LDS) so that TWO 16-wave work-groups
// share a CU: one's prologue / output transform overlaps the other's MFMA loop
template <int TC, int CK, int ACT, int RES>
__global__ __launch_bounds__(1024, 8) void conv3x3_wino32_kernel(WinoArgs a) {
  wino_body<1, TC, CK, ACT, RES, 16, 2>(a);
}

template <int ACT, int RES>
static int launch_wino(WinoArgs& a, int tile, hipStream_t st) {
  constexpr int CK = 8;
  auto go = [&](auto kern, int BM, int TC) -> int {
    const int TR = 64 / TC;
    a.n_ct = a.Cout_pad / BM;
    a.blk_r = cdiv(cdiv(a.H, 2), TR);
    a.blk_c = cdiv(cdiv(a.W, 2), TC);
    a.n_pt = a.B * a.blk_r * a.blk_c;
    int grid = cdiv(a.n_pt, 8) * 8 * a.n_ct;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0, st, a);
    SGTA_LAUNCH_CHECK();
    return SGTA_OK;
  };
  // tile = 10*MT + t : MT in {1,2} = 32 / 64 couts per work-group; t in {0,1,2,3} = 2x32 / 4x16 / 8x8 / 16x4 tile blocks
  const int mt = tile / 10, tc = tile % 10;
  if (mt == 2 && a.Cout_pad % 64 == 0) {
    if (tc == 0) return go(conv3x3_wino_kernel<32, CK, ACT, RES>, 64, 32);
    if (tc == 1) return go(conv3x3_wino_kernel<16, CK, ACT, RES>, 64, 16);
    if (tc == 2) return go(conv3x3_wino_kernel<8, CK, ACT, RES>, 64, 8);
    if (tc == 3) return go(conv3x3_wino_kernel<4, CK, ACT, RES>, 64, 4);
  }
  if (mt == 1) {
    // 32 couts: <= 64 registers and <= 80 KB of LDS, so TWO 16-wave work-groups share a CU and one's
    // prologue / output transform overlaps the other's MFMA loop
    if (tc == 0) return go(conv3x3_wino32_kernel<32, CK, ACT, RES>, 32, 32);
    if (tc == 1) return go(conv3x3_wino32_kernel<16, CK, ACT, RES>, 32, 16);
    if (tc == 2) return go(conv3x3_wino32_kernel<8, CK, ACT, RES>, 32, 8);
    if (tc == 3) return go(conv3x3_wino32_kernel<4, CK, ACT, RES>, 32, 4);
  }
  return SGTA_EBADARG;
}

extern "C" int sgta_conv3x3_wino_f32(const float* x, long x_bs, const float* u, const float* bias,
                                     const float* res, long res_bs, float* y, long y_bs, int B, int Cin,
                                     int H, int W, int Cout, int Cout_pad, int act, int res_mode,
                                     int tile, hipStream_t st) {
  if (!x || !u || !bias || !y || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return SGTA_EBADARG;
  if (Cin % 8 || Cout_pad % 32 || Cout_pad < Cout) return SGTA_EBADARG;
  if ((res_mode != 0) != (res != nullptr) || (long)Cin * H * W > 0x7fffffffL) return SGTA_EBADARG;
  if ((long)16 * Cout_pad * 8 > 0x7fffffffL) return SGTA_EBADARG;
  WinoArgs a{};
  a.x = x; a.x_bs = x_bs; a.u = u; a.bias = bias; a.res = res; a.res_bs = res_bs; a.y = y; a.y_bs = y_bs;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Cout_pad = Cout_pad; a.nchunks = Cin / 8;
  if (act == 1 && res_mode == 0) return launch_wino<1, 0>(a, tile, st);
  if (act == 0 && res_mode == 0) return launch_wino<0, 0>(a, tile, st);
  if (act == 1 && res_mode == 1) return launch_wino<1, 1>(a, tile, st);
  if (act == 1 && res_mode == 2) return launch_wino<1, 2>(a, tile, st);
  if (act == 2 && res_mode == 0) return launch_wino<2, 0>(a, tile, st);
  return SGTA_EBADARG;
}
