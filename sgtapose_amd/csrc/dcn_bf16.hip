// bf16 data path: modulated deformable 3x3 convolution (lbin/DCNv2 semantics) + folded BN +
// ReLU on bf16 blocked channels-last feature maps ([N][C/16][H][W][16]); replaces DeformConv (dla.py:538-550) like sgta_dcnv2_lds_f32 does in fp32.
//
// Channels-last makes the deformable gather coalesced: one sampling corner of one pixel is a run of consecutive
// channels, so a lane fetches 8 channels (16 bytes) per corner, blends the four corners in fp32 with its per-(pixel, tap)
// weights (bilinear weight x validity x sigmoid(mask), computed once per tap from the fp32 offset / mask map) and has
// the bf16 MFMA B operand of that (pixel, tap, 8 channels) in registers -- the [9*Cin, H*W] column buffer of the
// reference never exists.  A = the bf16 weights, staged per 16-channel chunk ([tap][TC][16], XOR-swizzled rows) by 16-byte
// buffer LDS-DMA, shared by the 8 waves of the work-group; every gathered sample feeds all TC = 64 / 128 output
// channels of the work-group.  Accumulation fp32 on v_mfma_f32_32x32x16_f16 (weights and blended samples fp16, see below).
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct DbArgs {
  const void* x;    long x_bs;  long x_bytes;    // bf16 / fp16 blocked input [N][Cb][H][W][16] (pointer at the view's first block), elements per image, extent
  const float* om;  int ldo;                     // f32 NHWC offsets / mask logits [N,H,W,ldo]: ldo >= 28, ldo % 4 == 0 (read as 7 x 16 bytes)
  const void* wt;   long wt_bytes;               // packed [n_ct][Cin/16][9][TC][16] FP16, swizzled
  const float* bias;                             // [n_ct * TC]
  void* y;          long y_bs;                   // bf16 / fp16 blocked output, elements per image
  int N, H, W, Cin, Cout, ngroups, n_ct, n_rt, n_xt, log_tw, act;
  unsigned mH, mWC[3];                           // ceil(2^32 / H), ceil(2^32 / window columns) for log_tw 3..5 (umulhi division)
};

// floor(v / d) for 0 <= v < 2^16 through the launcher's m = ceil(2^32 / d): no integer-division sequence in the prologue
__device__ __forceinline__ int db_div(int v, unsigned m) { return (int)__umulhi((unsigned)v, m); }

// Two earlier forms gathered the four corners of every sample straight from global memory (lane = pixel, then 8 lanes per
// 128-byte line): both ran at ~18 B/clk/CU -- the L2 -> CU fill rate -- because every input pixel is fetched 36 times
// (9 taps x 4 corners) and the 32 KB L1 cannot hold a tile's neighbourhood.  This form stages the neighbourhood ONCE
// per 16-channel chunk in LDS (the conv kernel's window machinery: 16-byte buffer LDS-DMA, XOR-swizzled 32-byte rows,
// double buffered with the chunk's [tap][TC][16] weight image) and gathers from LDS: (TR + 2R + 1) x (TW + 2R + 1) pixels
// with R = 6, i.e. sampling offsets up to +-6 pixels.  A tap whose sample leaves the window for ANY lane of a wave sends
// that wave through a slow path for that tap (the four corner runs by buffer loads from global memory) -- a separate
// code path, selected wave-uniformly, so the compiler never merges LDS and global reads.  Per lane and tap only the window
// address of the top-left corner and the four weights (bilinear x validity x sigmoid(mask)) stay in registers; they are
// computed once per tile.
// VALU issue (one wave64 instruction per ~4 cycles and SIMD) is what bounds the gather: 96 instructions per wave, tap and 8
// channels, a third of them bf16 -> f32 conversions of window elements that are each used ~36 times.  So the staged
// window is converted ONCE per chunk, in place, to fp16 (exact for bf16 values with |x| in [6e-5, 65504]; larger values
// are clamped to +-65504, smaller ones flush towards zero), the main product runs on v_mfma_f32_32x32x16_F16 with the
// weights packed as fp16 (11 significand bits: closer to the fp32 weights than bf16), and the blend is PACKED fp16
// arithmetic on channel pairs -- one v_pk_mul_f16 + three v_pk_fma_f16 per pair with the per-(pixel, tap, corner) weight
// (bilinear x validity x sigmoid(mask), computed in fp32, rounded once to fp16) broadcast to both halves: 16 instructions per
// tap and 8 channels, and the result IS the MFMA B operand (the fp32 form took 32 v_fma_mix_f32 + 4 conversions, the first
// one 68).  The blend rounds to 11 bits three times where the un-deformed conv path rounds its activations to 8 (bf16).
#ifndef DB_SCHED
#define DB_SCHED 1
#endif
#ifndef DB_ABLATE
#define DB_ABLATE 0          // tools only (tools/dcn_ablate.sh): 1 no MFMA, 2 no blend, 3 no corner reads, 4 no conversion, 5 no DMA
#endif

// Work-group shape: 256 threads = 4 waves x 32 pixels, SINGLE-buffered LDS (window 24 KB + weights 20 / 36 KB), so that two
// to three work-groups share a CU and one's prologue (27 offset loads, 9 x sampling parameters), DMA waits and epilogue
// run under the others' gather -- with one 8-wave double-buffered work-group per CU those phases were 45 % of a
// 64-channel layer's time (25 k cycles per work-group against 7.5 k per chunk).
//
// Margin R and far samples.  R = 6 (PR = 0): the 120x160 layers, whose offsets stay within a few pixels.  R = 10 with PR = 4
// PATCH ROWS (the deep levels: on the benched network 8-18 % of their samples lie more than 6 pixels out, 0.1-1.4 % more than
// 10): a sample whose 2x2 corner block leaves even the wider window gets a patch -- a 2x2 block of slots in PR extra window
// rows under the regular ones.  The owner lane takes a patch index from an LDS counter and writes the four global pixel
// offsets into a small LDS table; the DMA lanes that own those slots read their source offsets from the table, so the
// patch pixels arrive with every chunk's window like any other slot, are converted with it, and the gather reads them
// through the same base + {0, 16, WC*16, WC*16 + 16} addressing.  Only a tile with more far samples than patch blocks
// (PR/2 * WC/2 = 28 / 36) falls back to the global-memory path, wave by wave.
template <int R, int PR>
struct DbGeom {
  static constexpr int NT = 256;
  static constexpr int ws(int tr, int tw) { return (tr + 2 * R + 1 + PR) * (tw + 2 * R + 1); }
  static constexpr int mx(int a, int b) { return a > b ? a : b; }
  // R = 6: tiles 4x32, 8x16, 16x8; R = 10: 8x16, 16x8 (a 4x32 tile would need a 53-column window)
  static constexpr int WS_MAX = R <= 6 ? mx(ws(4, 32), mx(ws(8, 16), ws(16, 8))) : mx(ws(8, 16), ws(16, 8));
  static constexpr int WROUNDS = (WS_MAX * 2 + NT - 1) / NT, WIN_BYTES = WROUNDS * NT * 16;
  static constexpr int PT_MAX = PR * (16 + 2 * R + 1);                  // patch-table entries (slots of the patch rows)
  static constexpr int AUX_BYTES = PR ? (PT_MAX * 4 + 16 + 15) / 16 * 16 : 0;
};

// F16: the feature maps are IEEE half instead of bf16 -- the staged window needs no conversion, the rest is unchanged.
template <int NA, int R, int PR, bool F16 = false>
__global__ __launch_bounds__(256, 2) void dcn_bf16_kernel(DbArgs a) {
  using G = DbGeom<R, PR>;
  using E = H16<F16>;
  typedef typename E::T elt;
  typedef typename E::x8 bf16x8;                             // 8 / 4 elements of the maps' format
  typedef typename E::x4 bf16x4;
  constexpr int TC = NA * 32, TP = 128, ROWB = 32, NT = 256;
  constexpr int WIMG = 9 * TC * ROWB, AROUNDS = (WIMG + NT * 16 - 1) / (NT * 16);
  constexpr int WROUNDS = G::WROUNDS, WIN_BYTES = G::WIN_BYTES;
  // the window is stored as two planes (channel halves) of 16-byte slots: the regular pattern (consecutive pixels ->
  // consecutive slots) reads 16 lanes x 16 contiguous bytes (conflict-free without an XOR swizzle), and a sample's four
  // corners are base + {0, 16, WC*16, WC*16 + 16}: one address register per tap
  constexpr int WPL = WROUNDS * NT / 2;
  constexpr int STAGE = WIN_BYTES + AROUNDS * NT * 16;      // + G::AUX_BYTES of patch table behind it
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  int pt, ct;
  xcd_remap(blockIdx.x, a.n_ct, pt, ct);
  if (pt >= a.n_rt * a.n_xt) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int TW = 1 << a.log_tw, TR = TP >> a.log_tw;
  const int WC = TW + 2 * R + 1, WR = TR + 2 * R + 1, WS = WR * WC;
  const int rt = pt / a.n_xt, xt = pt - rt * a.n_xt;                    // (scalar division: once per work-group)
  const int H = a.H, W = a.W;
  const int R0 = rt * TR - R, X0 = xt * TW - R;                         // stacked row / column of window slot (0, 0)
  const unsigned mwc = a.mWC[a.log_tw - 3];
  unsigned* const ptab = reinterpret_cast<unsigned*>(smem + STAGE);     // [PR * WC] global offsets of the patch slots, then the counter
  unsigned* const pcnt = ptab + G::PT_MAX;
  if (PR) {
    for (int i = tid; i < PR * WC; i += NT) ptab[i] = 0x80000000u;      // unused patch slots: zeros
    if (tid == 0) *pcnt = 0u;
    __syncthreads();
  }

  // ---- per-lane DMA offsets (chunk-invariant): window slots in stacked-row space (N*H rows), weight image -------------
  unsigned woff[WROUNDS];
#pragma unroll
  for (int t = 0; t < WROUNDS; ++t) {
    const int pi = t * NT + tid;
    const int pp = pi >= WPL ? 1 : 0, slot = pi - pp * WPL;      // planar window: 8-channel half pp of every slot, then the other
    unsigned off = 0x80000000u;
    if (slot < WS) {
      const int wr = db_div(slot, mwc), wc = slot - wr * WC;
      const int Rs = R0 + wr, x = X0 + wc;
      if (Rs >= 0 && Rs < a.N * H && (unsigned)x < (unsigned)W) {
        const int piece = pp;
        const int ni = db_div(Rs, a.mH), yi = Rs - ni * H;
        off = (unsigned)(((long)ni * a.x_bs + ((long)yi * W + x) * 16 + piece * 8) * 2);
      }
    }
    woff[t] = off;
  }
  unsigned aoffw[AROUNDS];
#pragma unroll
  for (int t = 0; t < AROUNDS; ++t) {
    const int e = (t * NT + tid) * 16;
    aoffw[t] = e < WIMG ? (unsigned)e : 0x80000000u;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wt), (short)0, (int)a.wt_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), (short)0, (int)a.x_bytes, 0x00020000);
#endif
  const int nchunks = a.Cin / 16;
  auto stage = [&](int kc) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned char* wdst = smem + wave * 1024;
    unsigned char* adst = wdst + WIN_BYTES;
#pragma unroll
    for (int t = 0; t < WROUNDS; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(wdst + t * NT * 16), 16, woff[t], kc * H * W * 32, 0, 0);
#pragma unroll
    for (int t = 0; t < AROUNDS; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(adst + t * NT * 16), 16, aoffw[t],
                                               (ct * nchunks + kc) * WIMG, 0, 0);
#endif
  };
  stage(0);

  // ---- this lane's pixel and its 9 samples ------------------------------------------------------------------------
  const int p = wave * 32 + r;
  const int Rp = rt * TR + (p >> a.log_tw), xp = xt * TW + (p & (TW - 1));
  const int n = db_div(Rp, a.mH), yp = Rp - n * H;
  const bool okp = n < a.N && xp < W;
  const long pix = okp ? (long)Rp * W + xp : -1;
  int sbase[9];                                  // LDS byte offset of the sample's top-left corner (y0, x0), this lane's channel half
  f16x2 cw[9][4];                                // corner weights, fp16, broadcast to both halves of a channel pair
  unsigned inmask = 0;                           // bit t: all four corners of tap t lie inside the staged window
  {
    // the pixel's 27 offset / mask values: one 128-byte row, fetched as 7 x 16 bytes
    const f32x4* o4 = reinterpret_cast<const f32x4*>(a.om + (pix >= 0 ? pix : 0) * a.ldo);
    float o[28];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const f32x4 v4 = o4[q];
      o[4 * q] = v4[0]; o[4 * q + 1] = v4[1]; o[4 * q + 2] = v4[2]; o[4 * q + 3] = v4[3];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float dy = o[2 * t], dx = o[2 * t + 1], m = __builtin_amdgcn_rcpf(1.0f + __expf(-o[18 + t]));
      const float py = (float)(yp - 1 + t / 3) + dy, px = (float)(xp - 1 + t % 3) + dx;
      const bool inside = okp && py > -1.0f && px > -1.0f && py < (float)H && px < (float)W;
      const float fy = floorf(py), fx = floorf(px);
      const float ly = py - fy, lx = px - fx, hy = 1.0f - ly, hx = 1.0f - lx;
      const int y0 = (int)fy, x0 = (int)fx;
      const bool oy0 = y0 >= 0, oy1 = y0 + 1 <= H - 1, ox0 = x0 >= 0, ox1 = x0 + 1 <= W - 1;
      const float w0 = (inside && oy0 && ox0) ? hy * hx * m : 0.0f, w1 = (inside && oy0 && ox1) ? hy * lx * m : 0.0f;
      const float w2 = (inside && oy1 && ox0) ? ly * hx * m : 0.0f, w3 = (inside && oy1 && ox1) ? ly * lx * m : 0.0f;
      cw[t][0] = f16x2{(_Float16)w0, (_Float16)w0};
      cw[t][1] = f16x2{(_Float16)w1, (_Float16)w1};
      cw[t][2] = f16x2{(_Float16)w2, (_Float16)w2};
      cw[t][3] = f16x2{(_Float16)w3, (_Float16)w3};
      // window coordinates of (y0, x0); the +1 corners need wr + 1 < WR and wc + 1 < WC.  A sample outside the image has
      // all-zero weights: it is pointed at slot 0 and counts as inside.
      const int wr = n * H + y0 - R0, wc = x0 - X0;
      bool in = !inside || (wr >= 0 && wr + 1 < WR && wc >= 0 && wc + 1 < WC);
      int slot0 = (inside && in) ? wr * WC + wc : 0;
      if (PR) {
        // far sample: the lower lane half takes a patch block and publishes its four source pixels; both halves address it
        int pid = -1;
        if (!in && h == 0) pid = (int)__hip_atomic_fetch_add(pcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        pid = __shfl(pid, r);
        const int ppr = WC >> 1;                       // patch blocks per pair of patch rows
        if (pid >= 0 && pid < (PR / 2) * ppr) {
          const int prow = pid / ppr, pcol = pid - prow * ppr;
          slot0 = (WR + 2 * prow) * WC + 2 * pcol;
          if (h == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
              const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
              ptab[slot0 - WS + (k & 1) + (k >> 1) * WC] =
                  ok ? (unsigned)(((long)n * a.x_bs + ((long)yy * W + xx) * 16) * 2) : 0x80000000u;
            }
          }
          in = true;
        }
      }
      sbase[t] = slot0 * 16 + h * (WPL * 16);
      inmask |= (in ? 1u : 0u) << t;
    }
  }
  unsigned aaddr = (unsigned)(r * ROWB + ((h ^ ((r >> 3) & 1)) << 4));
  if (PR) {
    // the table is complete: the lanes that own patch slots take their DMA source from it (chunk 0's window is already in
    // flight with zeros in those slots: wait for it, then fetch the patch slots of chunk 0; later chunks carry them along)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool any = *pcnt != 0u;
#pragma unroll
    for (int t = 0; t < WROUNDS; ++t) {
      const int pi = t * NT + tid;
      const int pp = pi >= WPL ? 1 : 0, slot = pi - pp * WPL;
      if (slot >= WS && slot < WS + PR * WC) {
        const unsigned src = ptab[slot - WS];
        woff[t] = src == 0x80000000u ? src : src + pp * 16;
      }
    }
    if (any) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int t = 0; t < WROUNDS; ++t) {
        const int lo = t * NT, hi = lo + NT, pw = PR * WC;       // rounds that hold patch slots of either plane (uniform)
        if ((lo < WS + pw && hi > WS) || (lo < WPL + WS + pw && hi > WPL + WS))
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(smem + wave * 1024 + t * NT * 16), 16, woff[t], 0, 0, 0);
      }
#endif
    }
  }

  // the accumulators start at the bias (C/D row = (q & 3) + 8 (q >> 2) + 4 h): no dependent bias loads in the epilogue
  f32x16 acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + ct * TC + i * 32 + 8 * g4 + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][4 * g4 + j] = bv[j];
    }

  auto blend_mma = [&](const bf16x8 (&cv)[4], const float (&w4)[4], const unsigned char* As, int t) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float s0 = w4[0] * (float)cv[0][j];
      s0 = __builtin_fmaf(w4[1], (float)cv[1][j], s0);
      s0 = __builtin_fmaf(w4[2], (float)cv[2][j], s0);
      v[j] = __builtin_fmaf(w4[3], (float)cv[3][j], s0);
    }
    f16x8 bfr;
#pragma unroll
    for (int j = 0; j < 8; ++j) bfr[j] = (_Float16)fminf(fmaxf(v[j], -65504.0f), 65504.0f);
    f16x8 af[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) af[i] = *reinterpret_cast<const f16x8*>(As + (t * TC + i * 32) * ROWB + aaddr);
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bfr, acc[i], 0, 0, 0);
  };

  // wave-uniform choice for the whole tile: every sample of every lane inside the staged window -> LDS gathers; otherwise
  // this wave fetches its corner runs from global memory (rare: offsets beyond +-6 pixels).  Two separate loops with the
  // same barrier structure, so LDS and global reads never share a code path.
  const bool all_in = __builtin_amdgcn_ballot_w64(inmask != 0x1ffu) == 0;
  // in-place bf16 -> fp16 conversion of this thread's OWN window pieces (the ones its DMA lanes wrote): every wave runs it,
  // also the waves on the slow path (they read nothing from the window but share the barriers)
  auto convert_window = [&](unsigned char* Wbuf) {
#pragma unroll
    for (int t = 0; t < WROUNDS; ++t) {
      unsigned char* pp = Wbuf + (t * NT + tid) * 16;
      // bf16 -> f32 is a shift / mask of the packed pair; v_cvt_pkrtz_f16_f32 converts two values per instruction, exactly
      // (8 significant bits fit fp16's 11) and, rounding toward zero, saturates |x| > 65504 to +-65504 instead of infinity
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 b = *reinterpret_cast<const u32x4*>(pp);
      u32x4 hv;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        hv[j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(__builtin_bit_cast(float, b[j] << 16),
                                                                       __builtin_bit_cast(float, b[j] & 0xffff0000u)));
      *reinterpret_cast<u32x4*>(pp) = hv;
    }
  };
  unsigned char* const Ws = smem;
  const unsigned char* const As = smem + WIN_BYTES;
  if (all_in) {
    for (int kc = 0; kc < (DB_ABLATE == 6 ? 0 : (DB_ABLATE == 7 ? 1 : nchunks)); ++kc) {
      if (kc) {
        __builtin_amdgcn_s_barrier();              // every wave is done with chunk kc - 1: the buffers may be overwritten
        if (DB_ABLATE != 5) stage(kc);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (DB_ABLATE != 4 && !F16) convert_window(Ws);      // own pieces only: no barrier needed between the DMA wait and this
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                // window (converted) + weights of chunk kc complete for all waves
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        // (requesting tap t + 1's corner runs before tap t's blend -- a two-deep software pipeline -- measured no gain on the
        // 64-channel form and costs the 128-channel form 24 spilled registers: LDS latency is not what the taps wait for)
        f16x8 cv[4];
        const unsigned char* c0 = Ws + sbase[t];
        const unsigned char* c1 = c0 + WC * 16;        // next window row
        if (DB_ABLATE == 3) {
          const _Float16 cst = (_Float16)(float)(sbase[t] & 7);
#pragma unroll
          for (int k = 0; k < 4; ++k) cv[k] = f16x8{cst, cst, cst, cst, cst, cst, cst, cst};
        } else {
          cv[0] = *reinterpret_cast<const f16x8*>(c0);
          cv[1] = *reinterpret_cast<const f16x8*>(c0 + 16);
          cv[2] = *reinterpret_cast<const f16x8*>(c1);
          cv[3] = *reinterpret_cast<const f16x8*>(c1 + 16);
        }
        f16x8 af[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) af[i] = *reinterpret_cast<const f16x8*>(As + (t * TC + i * 32) * ROWB + aaddr);
        f16x8 bfr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                // channel pair j: v_pk_mul_f16 + 3 v_pk_fma_f16
          f16x2 s0 = cw[t][0] * f16x2{cv[0][2 * j], cv[0][2 * j + 1]};
          s0 = __builtin_elementwise_fma(cw[t][1], f16x2{cv[1][2 * j], cv[1][2 * j + 1]}, s0);
          s0 = __builtin_elementwise_fma(cw[t][2], f16x2{cv[2][2 * j], cv[2][2 * j + 1]}, s0);
          s0 = __builtin_elementwise_fma(cw[t][3], f16x2{cv[3][2 * j], cv[3][2 * j + 1]}, s0);
          bfr[2 * j] = s0[0];
          bfr[2 * j + 1] = s0[1];
        }
        if (DB_ABLATE == 2) bfr = cv[0];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          if (DB_ABLATE == 1) acc[i][t] += (float)bfr[i] + (float)af[i][t & 7];
          else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bfr, acc[i], 0, 0, 0);
        }
#if DB_SCHED
        __builtin_amdgcn_sched_barrier(0);           // one tap's reads in flight at a time (without it the compiler's order is 3-10 % slower)
#endif
      }
    }
  } else {
    const float* o = a.om + (pix >= 0 ? pix : 0) * a.ldo;
    const int nb = okp ? n : 0;
    for (int kc = 0; kc < nchunks; ++kc) {
      if (kc) {
        __builtin_amdgcn_s_barrier();
        stage(kc);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!F16) convert_window(Ws);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll 1
      for (int t = 0; t < 9; ++t) {
        const float py = (float)(yp - 1 + t / 3) + o[2 * t], px = (float)(xp - 1 + t % 3) + o[2 * t + 1];
        const float m = __builtin_amdgcn_rcpf(1.0f + __expf(-o[18 + t]));
        const bool inside = okp && py > -1.0f && px > -1.0f && py < (float)H && px < (float)W;
        const float fy = floorf(py), fx = floorf(px);
        const float ly = py - fy, lx = px - fx, hy = 1.0f - ly, hx = 1.0f - lx;
        const int y0 = (int)fy, x0 = (int)fx;
        const bool oy0 = y0 >= 0, oy1 = y0 + 1 <= H - 1, ox0 = x0 >= 0, ox1 = x0 + 1 <= W - 1;
        const float w4[4] = {(inside && oy0 && ox0) ? hy * hx * m : 0.0f, (inside && oy0 && ox1) ? hy * lx * m : 0.0f,
                             (inside && oy1 && ox0) ? ly * hx * m : 0.0f, (inside && oy1 && ox1) ? ly * lx * m : 0.0f};
        const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y0 + 1, 0), H - 1);
        const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x0 + 1, 0), W - 1);
        bf16x8 cv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int yy = (k >> 1) ? cy1 : cy0, xx = (k & 1) ? cx1 : cx0;
          const unsigned off = (unsigned)(((long)nb * a.x_bs + ((long)yy * W + xx) * 16 + h * 8) * 2);
#if defined(__HIP_DEVICE_COMPILE__)
          cv[k] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, kc * H * W * 32, 0));
#endif
        }
        blend_mma(cv, w4, As, t);
      }
    }
  }

  // epilogue: bias + ReLU, then v_permlane32_swap pairs the two lane halves' channel groups so that every lane stores 8
  // consecutive channels (16 bytes) and a pixel's two lanes 32 contiguous bytes (see conv_bf16.hip)
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto value4 = [&](int i, int g4) {
    const int co0 = ct * TC + i * 32 + 8 * g4 + 4 * h;
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (co0 < a.Cout) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc[i][4 * g4 + j];
        if (a.act == 1) v[j] = fmaxf(v[j], 0.0f);
      }
    }
    return v;
  };
  if (pix >= 0) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const bf16x4 pb = E::cvt4(value4(i, 2 * m));
        const bf16x4 qb = E::cvt4(value4(i, 2 * m + 1));
        const u32x2 P = __builtin_bit_cast(u32x2, pb), Q = __builtin_bit_cast(u32x2, qb);
        const auto s0 = __builtin_amdgcn_permlane32_swap(P[0], Q[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(P[1], Q[1], false, false);
        const int co = ct * TC + i * 32 + 16 * m + 8 * h;
        if (co < a.Cout)
          *reinterpret_cast<u32x4*>(reinterpret_cast<elt*>(a.y) + (long)n * a.y_bs + (long)(co >> 4) * H * W * 16 + ((long)yp * W + xp) * 16 + (co & 15)) =
              u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
  }
}

template <int NA, int R, int PR, bool F16>
static int db_launch(DbArgs& a, hipStream_t st) {
  using G = DbGeom<R, PR>;
  constexpr int NT = 256, WIMG = 9 * NA * 32 * 32;
  constexpr int LDS = G::WIN_BYTES + (WIMG + NT * 16 - 1) / (NT * 16) * NT * 16 + G::AUX_BYTES;
  static_assert(2 * LDS <= 160 * 1024, "two work-groups per CU");
  auto kern = dcn_bf16_kernel<NA, R, PR, F16>;
  if (LDS > 64 * 1024) {
    static bool attr_by_dev[SGTA_MAX_DEVICES] = {};
    bool& done = attr_by_dev[sgta_device_slot()];
    if (!done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
        return SGTA_EBADARG;
      done = true;
    }
  }
  const int TW = 1 << a.log_tw, TR = 128 >> a.log_tw;
  a.mH = (unsigned)((0x100000000ULL + a.H - 1) / a.H);
  for (int l = 3; l <= 5; ++l) a.mWC[l - 3] = (unsigned)((0x100000000ULL + (1 << l) + 2 * R) / ((1 << l) + 2 * R + 1));
  if ((long)a.N * a.H >= 65536 || a.H < 2) return SGTA_EBADARG;          // umulhi division range
  a.n_rt = cdiv((long)a.N * a.H, TR);
  a.n_xt = cdiv(a.W, TW);
  const int grid = cdiv(a.n_rt * a.n_xt, 8) * 8 * a.n_ct;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), LDS, st, a);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// x bf16 blocked [N][Cb][H][W][16] (x_bs elements per image); om f32 NHWC [N,H,W,ldo]: channels 0..17 = (dy, dx) per tap, 18..26 = mask logits (the raw
// conv_offset_mask output); wt: [n_ct][Cin/16][9][32*na][16] bf16 rows XOR-swizzled in 16-byte pieces (the 3x3 layout of
// sgta_conv_bf16, ops_bf16.py); bias f32 [n_ct*32*na] (BN folded); y bf16 blocked (y_bs elements per image); act 1 = ReLU.
// Cin % 16 == 0, Cout % 8 == 0.  margin: 6 (window of +-6 pixels round the tile, log_tw in 3..5) or 10 (+-10 pixels and patch
// slots for the samples beyond, log_tw in 3..4): speed only, every margin gives the same result.
template <bool F16>
static int db_dcn(const void* x, long x_bs, const float* om, int ldo, const void* wt, long wt_bytes,
                  const float* bias, void* y, long y_bs, int N, int H, int W, int Cin, int Cout,
                  int na, int log_tw, int margin, int act, hipStream_t st) {
  if (!x || !om || !wt || !bias || !y || N <= 0 || H <= 0 || W <= 0) return SGTA_EBADARG;
  if (Cin % 16 || Cout % 8 || x_bs % 16 || y_bs % 16 || ((uintptr_t)y & 15) || ((uintptr_t)x & 15) || ldo < 28 || ldo % 4 || ((uintptr_t)om & 15) || log_tw < 3 ||
      log_tw > 5 || (margin != 6 && margin != 10) || (margin == 10 && log_tw > 4))
    return SGTA_EBADARG;
  if (wt_bytes <= 0 || wt_bytes > 0x7fffffffL) return SGTA_EBADARG;
  if ((long)N * x_bs * 2 > 0x7fffffffL) return SGTA_EBADARG;             // 31-bit byte offsets in the gather
  DbArgs a;
  a.x = x; a.x_bs = x_bs;
  a.x_bytes = ((long)(N - 1) * x_bs + (long)(Cin / 16) * H * W * 16) * 2; a.om = om; a.ldo = ldo; a.wt = wt; a.wt_bytes = wt_bytes;
  a.bias = bias; a.y = y; a.y_bs = y_bs; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.ngroups = Cin / 16; a.n_ct = cdiv(Cout, 32 * na); a.log_tw = log_tw; a.act = act;
  if (na == 2) return margin == 6 ? db_launch<2, 6, 0, F16>(a, st) : db_launch<2, 10, 4, F16>(a, st);
  if (na == 4) return margin == 6 ? db_launch<4, 6, 0, F16>(a, st) : db_launch<4, 10, 4, F16>(a, st);
  return SGTA_EBADARG;
}

extern "C" int sgta_dcnv2_fwd_bf16(const void* x, long x_bs, const float* om, int ldo, const void* wt, long wt_bytes,
                                   const float* bias, void* y, long y_bs, int N, int H, int W, int Cin, int Cout,
                                   int na, int log_tw, int margin, int act, hipStream_t st) {
  return db_dcn<false>(x, x_bs, om, ldo, wt, wt_bytes, bias, y, y_bs, N, H, W, Cin, Cout, na, log_tw, margin, act, st);
}

// The same operator on IEEE-half feature maps (x, y fp16 blocked): the staged window is used as it arrives.
extern "C" int sgta_dcnv2_fwd_f16(const void* x, long x_bs, const float* om, int ldo, const void* wt, long wt_bytes,
                                  const float* bias, void* y, long y_bs, int N, int H, int W, int Cin, int Cout,
                                  int na, int log_tw, int margin, int act, hipStream_t st) {
  return db_dcn<true>(x, x_bs, om, ldo, wt, wt_bytes, bias, y, y_bs, N, H, W, Cin, Cout, na, log_tw, margin, act, st);
}
