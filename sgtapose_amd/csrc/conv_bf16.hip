// bf16 data path (BASELINE configs[2]): convolution (+ folded BN bias, residual, activation) on bf16 blocked channels-last feature
// maps with bf16 weights, fp32 accumulation on v_mfma_f32_32x32x16_bf16.  Replaces the same reference code as
// sgta_conv2d_f32 (dla.py:41-69, :157-231, :302-312; DCN offset/mask conv; heads base_model.py:121-135).
//
// Layout: BLOCKED channels-last [N][C/16][H][W][16] (bf16): the MFMA B operand of a pixel (8 consecutive k = 8 consecutive
// input channels of one tap) is ONE 16-byte read, the 16-channel chunk of a window row is ONE contiguous run of global
// memory (the LDS-DMA moves whole cache lines: with plain NHWC a 16-channel chunk was 32 bytes out of every pixel's
// C*2-byte row, a quarter of each line at C = 64, and the window fetch cost 20-60 % of the conv kernels), and a wave's
// output store (32 pixels x 16 channels) is 1 KB contiguous.  The A operand (8 consecutive k of one output channel) is
// one 16-byte read of the packed weights.  D[cout, pixel]: the pixel sits on the lane, 4 consecutive output channels in
// 4 accumulator registers (fp32 NHWC / fp32 NCHW stores for the tensors that leave the bf16 domain).
//
// Tiling: the batch is ONE tall image of N*(H+1) rows (one zero row under every image, so a tile may span images and
// every tap is a constant LDS offset); a 512-thread work-group owns TR x TW = 256 / 512 output positions x TC = 32 / 64 /
// 128 output channels, each of its 8 waves 32 / 64 positions x all TC channels.  Per chunk of KC input channels the
// (S(TR-1)+KS) x (S(TW-1)+KS) input window and the [tap][TC][KC] weight image arrive by 16-byte buffer LDS-DMA
// (out-of-image window slots carry an out-of-range offset: the bounds check writes zeros), double buffered, one barrier
// per chunk.  LDS rows are KC*2 bytes; 16-byte pieces are XOR-swizzled with the row index (on the DMA source side for
// the window, at pack time for the weights) so the ds_read_b128 fragment reads are conflict-free.
#include "common.h"
#include <cstdlib>

struct CbArgs {
  const void* x;    long x_bytes;  long x_bs;    // bf16 blocked input (pointer at the view's first 16-channel block), bytes reachable, elements per image
  const void* wt;   long wt_bytes;               // packed [n_ct][nchunks][taps][TC][KC] bf16, swizzled
  const float* bias;                             // [n_ct * TC]
  const void* res; long res_bs;                  // residual, 16-bit blocked at the output resolution (or null)
  void* y;          long y_bs;                   // bf16 blocked (elements per image) / fp32 NHWC (channels per pixel) / fp32 NCHW (unused)
  int N, H, W, Ho, Wo, Cout, nchunks, n_ct, n_rt, n_xt, log_tw, act, out_mode, nsig;
  // fused heads (FUSE = 1): second product W2 [<= 16 outputs] x relu(conv + bias) over ALL n_ct channel tiles of the work-group
  const void* w2;                                // bf16 A fragments [n_ct][NA][2 k-steps][64 lanes][8]
  const float* b2;                               // [16]
  float* out[4];                                 // f32 NCHW per head
  int hs[5];                                     // first output row of head i (hs[n_heads..4] = total)
  int single;                                    // two-stage loop with ONE LDS stage (see the kernel)
  // X3 (fp32 emulated by operand splitting): every activation tensor is TWO planes of identical layout, value = hi + lo;
  // the lo plane lies a fixed distance behind the hi plane in the same allocation
  int x_lo_delta;                                // bytes from x to its lo plane
  long res_lo_delta, y_lo_delta;                 // elements from res / y to their lo planes
  const void* w2_lo;                             // FUSE: lo part of the second product's weights
};

template <int KS, int S, int KC, int NPF>
struct CbGeom {
  static constexpr int TP = 256 * NPF;                                // output positions per work-group
  static constexpr int TAPS = KS * KS, KSTEPS = KC / 16, P = KC / 8, ROWB = KC * 2;
  static constexpr int LOG_R = KC == 16 ? 3 : (KC == 32 ? 2 : 1);     // rows per 256-byte bank row
  // largest window over the allowed tile shapes (TW = 8 .. 64)
  static constexpr int ws(int tw) { return (S * (TP / tw - 1) + KS) * (S * (tw - 1) + KS); }
  static constexpr int mx(int a, int b) { return a > b ? a : b; }
  static constexpr int WS_MAX = mx(mx(ws(8), ws(16)), mx(ws(32), ws(64)));
  static constexpr int WROUNDS = (WS_MAX * P + 511) / 512;            // 16-byte pieces / 512 lanes
  static constexpr int WIN_BYTES = WROUNDS * 8192;
};

__device__ __forceinline__ float cb_sigmoid(float v) { return 1.0f / (1.0f + __expf(-v)); }

// FUSE = 1 (the three head branches, base_model.py:121-135: conv3x3 -> ReLU -> conv1x1 each, their 3x3 convs concatenated to
// one Cin -> n_ct*TC product): the work-group walks ALL channel tiles of its positions; after the last chunk of a tile the
// hidden activations relu(acc + bias) are rounded to bf16 IN REGISTERS -- accumulator registers 8t..8t+7 of a lane are
// exactly the B operand (k = 8h + j) of a second v_mfma_f32_32x32x16_bf16 whose k order the packed W2 follows -- and
// multiplied by the (block-diagonal) 1x1 weights of the heads: the hidden tensor never exists in memory.
//
// P3 (3x3 stride 1): the chunk loop as a THREE-DEEP pipeline with counted waits.  The two-stage form above waits for
// vmcnt(0) at every chunk barrier -- the "two barriers per K-step" GEMM structure whose ceiling is ~900 TFLOP/s, which is
// where these kernels sat (the DMA instructions of chunk k + 1 are issued during chunk k; the last of them land a
// memory latency after the chunk's MFMAs are done).  Here a chunk is three sub-steps (one tap row = 3 taps = 24-48 MFMAs
// per wave); the weight image is staged per tap row in a ring of three slots, two sub-steps ahead, the window per chunk in
// a ring of two, a whole chunk ahead; every sub-step waits only until the loads issued AFTER the data it needs are the
// only ones outstanding (s_waitcnt vmcnt(N), N = the rounds of one weight image [+ one window]): a load always has at
// least one full sub-step of matrix work to land in, and nothing in the loop drains the memory pipeline.
template <int N>
__device__ __forceinline__ void cb_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// X3: fp32 emulated on the 16-bit matrix pipe.  x = x_hi + x_lo and w = w_hi + w_lo (each part one 16-bit number: 2 x 11
// significant bits on IEEE half), product = x_hi w_hi + x_hi w_lo + x_lo w_hi accumulated in fp32 (the dropped x_lo w_lo term
// is 2^-22 relative).  For the kernel this is a convolution over 3x the chunks: virtual chunk v = 3 kc + p reads the window
// of real chunk kc from the hi plane (p = 0, 1) or the lo plane (p = 2), and the packed weights hold [w_hi, w_lo, w_hi] per real
// chunk (ops_x3.py); the epilogue splits the fp32 result into the two output planes.  Pass p = 1 reads the window that pass
// p = 0 staged (no second fetch of the hi plane).
template <int KS, int S, int KC, int NPF, int NA, int FUSE = 0, bool P3 = false, bool F16 = false, bool X3 = false>
__global__ __launch_bounds__(512) void conv_bf16_kernel(CbArgs a) {
  static_assert(!P3 || (KS == 3 && KC == 16), "the three-deep pipeline is built for 3x3 convolutions");
  using G = CbGeom<KS, S, KC, NPF>;
  using E = H16<F16>;                                                   // element format: bf16 or IEEE half
  typedef typename E::T elt;
  typedef typename E::x8 bf16x8;                                        // 8 / 4 elements of the path's 16-bit format
  typedef typename E::x4 bf16x4;
  constexpr int TC = NA * 32, TAPS = G::TAPS, KSTEPS = G::KSTEPS, P = G::P, ROWB = G::ROWB, LOG_R = G::LOG_R;
  constexpr int PAD = KS / 2;
  constexpr int WIMG = TAPS * TC * ROWB;                               // weight image bytes per (cout tile, chunk)
  constexpr int AROUNDS = (WIMG + 8191) / 8192;
  constexpr int STAGE = G::WIN_BYTES + AROUNDS * 8192;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  int pt, ct;
  xcd_remap(blockIdx.x, FUSE ? 1 : a.n_ct, pt, ct);
  if (pt >= a.n_rt * a.n_xt) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int TW = 1 << a.log_tw, TR = G::TP >> a.log_tw;
  const int WC = S * (TW - 1) + KS, WR = S * (TR - 1) + KS, WS = WR * WC;
  const int rt = pt / a.n_xt, xt = pt - rt * a.n_xt;
  const int R0 = rt * TR, x0 = xt * TW;
  const int Hop = a.Ho + 1, Hp = a.H + 1;
  // input row (stacked space) of the first tap row of output row R:  (R / Hop) * Hp + S * (R % Hop) - PAD
  const int inrow_base = (R0 / Hop) * Hp + S * (R0 % Hop) - PAD;
  const int xin_base = S * x0 - PAD;

  // ---- per-lane window DMA offsets (chunk-invariant) --------------------------------------------------------
  unsigned woff[G::WROUNDS];
#pragma unroll
  for (int t = 0; t < G::WROUNDS; ++t) {
    const int pi = t * 512 + tid;
    const int slot = pi / P, pp = pi - slot * P;
    unsigned off = 0x80000000u;
    if (slot < WS) {
      const int wr = slot / WC, wc = slot - wr * WC;
      const int Rin = inrow_base + wr, x = xin_base + wc;
      if (Rin >= 0 && (unsigned)x < (unsigned)a.W) {
        const int n = Rin / Hp, y = Rin - n * Hp;
        if (n < a.N && y < a.H) {
          const int piece = pp ^ ((slot >> LOG_R) & (P - 1));        // logical 8-channel piece of the chunk
          // block (piece >> 1) of the chunk, half (piece & 1) of the pixel's 16-channel run
          off = (unsigned)(((long)n * a.x_bs + (long)(piece >> 1) * a.H * a.W * 16 + ((long)y * a.W + x) * 16 + (piece & 1) * 8) * 2);
        }
      }
    }
#if defined(CB_ABLATE) && (CB_ABLATE & 1)
    off = 0x80000000u;                             // tools only: no window fetch (the DMA still writes zeros)
#endif
    woff[t] = off;
  }
  unsigned aoffw[AROUNDS];
#pragma unroll
  for (int t = 0; t < AROUNDS; ++t) {
    const int e = (t * 512 + tid) * 16;
    aoffw[t] = e < WIMG ? (unsigned)e : 0x80000000u;
#if defined(CB_ABLATE) && (CB_ABLATE & 2)
    aoffw[t] = 0x80000000u;                        // tools only: no weight fetch
#endif
  }
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), (short)0, (int)a.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wt), (short)0, (int)a.wt_bytes, 0x00020000);
#endif
  // byte offset (scalar) of the window source of (virtual) chunk v relative to a.x
  auto src_off = [&](int v) -> int {
    if (!X3) return v * (KC / 16) * a.H * a.W * 32;
    const int rk = v / 3, pp3 = v - 3 * rk;
    return rk * (KC / 16) * a.H * a.W * 32 + (pp3 == 2 ? a.x_lo_delta : 0);
  };
  // DMA instruction j (0 .. NDMA-1) of chunk kc (weight image wi) into stage st: window rounds first, then the weight image
  // (the window part of stage buffer ``wsl`` and the weight part of stage buffer ``st``: the two advance separately under X3,
  // whose middle pass reads the window of the pass before it)
  constexpr int NDMA = G::WROUNDS + AROUNDS;
  auto dma_one = [&](int j, int kc, int wi, int st, int wsl) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (j < G::WROUNDS)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(smem + wsl * STAGE + wave * 1024 + j * 8192), 16,
                                               woff[j < G::WROUNDS ? j : 0], src_off(kc), 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(smem + st * STAGE + wave * 1024 + G::WIN_BYTES + (j - G::WROUNDS) * 8192), 16,
                                               aoffw[j >= G::WROUNDS ? j - G::WROUNDS : 0], wi * WIMG, 0, 0);
#endif
  };
  auto stage = [&](int kc, int wi, int st, int wsl, bool with_win) {
#pragma unroll
    for (int j = 0; j < NDMA; ++j)
      if (j >= G::WROUNDS || with_win) dma_one(j, kc, wi, st, wsl);
  };

  // ---- per-lane output positions and LDS fragment addresses -----------------------------------------------------
  int pn[NPF], prem[NPF];                       // image index, yo * Wo + xo (or -1)
  unsigned baddr[TAPS * KSTEPS][NPF];
#pragma unroll
  for (int f = 0; f < NPF; ++f) {
    const int i = (wave * NPF + f) * 32 + r;
    const int tr = i >> a.log_tw, tc = i & (TW - 1);
    const int Rout = R0 + tr, xo = x0 + tc;
    const int n = Rout / Hop, yo = Rout - n * Hop;
    pn[f] = n;
    prem[f] = (n < a.N && yo < a.Ho && xo < a.Wo) ? yo * a.Wo + xo : -1;
    const int wr = (n * Hp + S * yo - PAD) - inrow_base, wc = S * tc;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int slot = (wr + t / KS) * WC + wc + t % KS;
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s)
        baddr[t * KSTEPS + s][f] = (unsigned)(slot * ROWB + (((2 * s + h) ^ ((slot >> LOG_R) & (P - 1))) << 4));
    }
  }
  unsigned aaddr[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) aaddr[s] = (unsigned)(r * ROWB + (((2 * s + h) ^ ((r >> LOG_R) & (P - 1))) << 4));

  f32x16 acc[NA][NPF];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int f = 0; f < NPF; ++f)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][f][q] = 0.0f;

  f32x16 acc2[FUSE ? NPF : 1];
  if (FUSE) {
#pragma unroll
    for (int f = 0; f < NPF; ++f)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc2[f][q] = 0.0f;
  }
  // FUSE: one flat sequence of (channel tile, chunk) steps; the window of chunk kc is fetched again for every tile (L2)
  const int n_it = FUSE ? a.n_ct * a.nchunks : a.nchunks;
  const int wi0 = FUSE ? 0 : ct * a.nchunks;
  int kc = 0, ctl = 0;
  auto heads_tile = [&]() {
    // ---- channel tile ctl is complete: hidden = relu(acc + bias) -> bf16 -> second product, accumulators cleared ------
    const bf16x8* w2 = reinterpret_cast<const bf16x8*>(a.w2) + (long)ctl * NA * 2 * 64 + lane;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 a2 = w2[(i * 2 + t) * 64];
        const float* bp = a.bias + ctl * TC + i * 32 + 16 * t + 4 * h;          // rows (j & 3) + 8 (2t + (j >> 2)) + 4h
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 8);
#pragma unroll
        for (int f = 0; f < NPF; ++f) {
          f32x8 hv;
#pragma unroll
          for (int j = 0; j < 8; ++j) hv[j] = fmaxf(acc[i][f][8 * t + j] + (j < 4 ? b0[j & 3] : b1[j & 3]), 0.0f);
          if (X3) {
            const bf16x8 a2l = (reinterpret_cast<const bf16x8*>(a.w2_lo) + (long)ctl * NA * 2 * 64 + lane)[(i * 2 + t) * 64];
            const bf16x8 hh = E::cvt8(hv);
            f32x8 hl;
#pragma unroll
            for (int j = 0; j < 8; ++j) hl[j] = hv[j] - (float)hh[j];
            acc2[FUSE ? f : 0] = E::mfma32(a2, hh, acc2[FUSE ? f : 0]);
            acc2[FUSE ? f : 0] = E::mfma32(a2l, hh, acc2[FUSE ? f : 0]);
            acc2[FUSE ? f : 0] = E::mfma32(a2, E::cvt8(hl), acc2[FUSE ? f : 0]);
          } else
          acc2[FUSE ? f : 0] = E::mfma32(a2, E::cvt8(hv), acc2[FUSE ? f : 0]);
        }
      }
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int f = 0; f < NPF; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][f][q] = 0.0f;
    ++ctl;
  };
  if constexpr (P3) {
    constexpr int WSUBB = 3 * TC * ROWB;                  // bytes of one tap row of the weight image
    constexpr int AW = (WSUBB + 8191) / 8192, WSUB = AW * 8192, WRN = G::WROUNDS;
    unsigned char* const win0 = smem;                      // two window slots, then three weight slots
    unsigned char* const wts0 = smem + 2 * G::WIN_BYTES;
    unsigned aoff3[AW];
#pragma unroll
    for (int t = 0; t < AW; ++t) {
      const int e = (t * 512 + tid) * 16;
      aoff3[t] = e < WSUBB ? (unsigned)e : 0x80000000u;
    }
    auto dma_win = [&](int kcn, int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int j = 0; j < WRN; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(win0 + slot * G::WIN_BYTES + wave * 1024 + j * 8192), 16, woff[j],
                                                 src_off(kcn), 0, 0);
#endif
    };
    auto dma_wts = [&](int row, int slot) {               // row = global tap-row index: (weight image index) * 3 + tap row
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int t = 0; t < AW; ++t)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(wts0 + slot * WSUB + wave * 1024 + t * 8192), 16, aoff3[t],
                                                 row * WSUBB, 0, 0);
#endif
    };
    const int total = 3 * n_it;
    dma_win(0, 0);
    dma_wts(3 * wi0, 0);
    dma_wts(3 * wi0 + 1, 1);
    int wslot = 0, ph = 0;                                 // window slot of this chunk; X3: pass of this chunk (kc % 3)
    for (int it = 0; it < n_it; ++it) {
      const int nkc = kc + 1 < a.nchunks ? kc + 1 : 0;
      const int nph = ph == 2 ? 0 : ph + 1;
      const bool next_new = !X3 || nph != 1;               // the next chunk reads a window of its own
      const bool fetch = it + 1 < n_it && next_new;
      const unsigned char* Ws = win0 + wslot * G::WIN_BYTES;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int s_ = 3 * it + j;
        // everything issued BEFORE the data of this sub-step has landed once only the later loads are outstanding:
        // the weight image of sub-step s + 1 (issued one sub-step ago) and, behind j == 0, the next chunk's window
        if (s_ + 1 >= total) cb_wait_vm<0>();
        else if (j == 1 && fetch) cb_wait_vm<AW + WRN>();
        else cb_wait_vm<AW>();
        __builtin_amdgcn_s_barrier();
        if (j == 0 && fetch) dma_win(nkc, wslot ^ 1);
        if (s_ + 2 < total) dma_wts(3 * wi0 + s_ + 2, (j + 2) % 3);
        const unsigned char* As = wts0 + j * WSUB;
        bf16x8 afr[2][NA], bfr[2][NPF];
        auto frags = [&](int tl, bf16x8 (&af)[NA], bf16x8 (&bf)[NPF]) {
#pragma unroll
          for (int i = 0; i < NA; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + (tl * TC + i * 32) * ROWB + aaddr[0]);
#pragma unroll
          for (int f = 0; f < NPF; ++f) bf[f] = *reinterpret_cast<const bf16x8*>(Ws + baddr[3 * j + tl][f]);
        };
        frags(0, afr[0], bfr[0]);
#pragma unroll
        for (int tl = 0; tl < 3; ++tl) {
          if (tl + 1 < 3) frags(tl + 1, afr[(tl + 1) & 1], bfr[(tl + 1) & 1]);
#pragma unroll
          for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int f = 0; f < NPF; ++f)
              acc[i][f] = E::mfma32(afr[tl & 1][i], bfr[tl & 1][f], acc[i][f]);
        }
      }
      if (FUSE && nkc == 0) heads_tile();
      kc = nkc;
      ph = nph;
      if (next_new) wslot ^= 1;
    }
  } else {
  // a.single: ONE LDS stage (a chunk's DMA is issued when every wave is done with the previous one and waited for at once),
  // half the LDS -> twice the work-groups per CU: for layers of a few chunks the other work-group's matrix work covers the
  // wait better than a second stage of the same work-group does
  stage(0, wi0, 0, 0, true);
  int wslot = 0, ph = 0;                           // stage buffer that holds this chunk's window; X3: pass of this chunk (kc % 3)
  for (int it = 0; it < n_it; ++it) {
    const int st = a.single ? 0 : (it & 1);
    if (a.single && it) {
      __builtin_amdgcn_s_barrier();                // every wave has read chunk it - 1
      stage(kc, wi0 + it, 0, 0, !X3 || ph != 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool more = !a.single && it + 1 < n_it;  // the next chunk's DMA is issued piece by piece between the MFMA groups below
    const int nkc = kc + 1 < a.nchunks ? kc + 1 : 0;
    const int nph = ph == 2 ? 0 : ph + 1;
    const bool next_new = !X3 || nph != 1;         // the next chunk reads a window of its own (X3: not the middle pass)
    const int nwslot = a.single ? 0 : (next_new ? wslot ^ 1 : wslot);
    const unsigned char* Ws = smem + wslot * STAGE;
    const unsigned char* As = smem + st * STAGE + G::WIN_BYTES;
    // software pipeline over the k-steps of the chunk: the fragments of step k + 1 are requested before the MFMAs of step
    // k issue, so the LDS latency sits under matrix work of the SAME wave (the two waves of a SIMD run this loop in
    // lock-step after the barrier and would otherwise wait on LDS together, then contend for the matrix pipe together)
    constexpr int NK = TAPS * KSTEPS;
    bf16x8 afr[2][NA], bfr[2][NPF];
    auto frags = [&](int k, bf16x8 (&af)[NA], bf16x8 (&bf)[NPF]) {
      const int t = k / KSTEPS, s_ = k % KSTEPS;
#pragma unroll
      for (int i = 0; i < NA; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + (t * TC + i * 32) * ROWB + aaddr[s_]);
#pragma unroll
      for (int f = 0; f < NPF; ++f) bf[f] = *reinterpret_cast<const bf16x8*>(Ws + baddr[k][f]);
    };
    // (a burst of NDMA LDS-DMA instructions right behind the barrier stalls the CU's single texture-address path while the
    // matrix pipes idle: one or two per k-step instead)
    constexpr int DPK = (NDMA + NK - 1) / NK;
    frags(0, afr[0], bfr[0]);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      if (k + 1 < NK) frags(k + 1, afr[(k + 1) & 1], bfr[(k + 1) & 1]);
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int f = 0; f < NPF; ++f)
          acc[i][f] = E::mfma32(afr[k & 1][i], bfr[k & 1][f], acc[i][f]);
      if (more) {
#pragma unroll
        for (int d = 0; d < DPK; ++d)
          if (k * DPK + d < NDMA && (k * DPK + d >= G::WROUNDS || next_new)) dma_one(k * DPK + d, nkc, wi0 + it + 1, st ^ 1, nwslot);
      }
    }
    if (FUSE && nkc == 0) heads_tile();
    kc = nkc;
    ph = nph;
    wslot = nwslot;
  }
  }

  if (FUSE) {
    // ---- heads epilogue: rows m = (q & 3) + 8 (q >> 2) + 4h < 16 of acc2 -> f32 NCHW of the head that owns row m -----------
    const long HWo = (long)a.Ho * a.Wo;
#pragma unroll
    for (int f = 0; f < NPF; ++f) {
      if (prem[f] < 0) continue;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int m = (q & 3) + 8 * (q >> 2) + 4 * h;
        if (m >= a.hs[4]) continue;
        const int hd = (m >= a.hs[1]) + (m >= a.hs[2]) + (m >= a.hs[3]);
        const int c = m - a.hs[hd], nc = a.hs[hd + 1] - a.hs[hd];
        float v = acc2[f][q] + a.b2[m];
        if (m < a.nsig) v = cb_sigmoid(v);
        a.out[hd][((long)pn[f] * nc + c) * HWo + prem[f]] = v;
      }
    }
    return;
  }

  // ---- epilogue: C/D layout col = lane & 31 (position), row = (q & 3) + 8 (q >> 2) + 4 h (output channel) ----------
  const long HWo = (long)a.Ho * a.Wo;
  auto value4 = [&](int i, int f, int g4) {                    // bias + residual + activation of 4 consecutive channels
    const int co0 = ct * TC + i * 32 + 8 * g4 + 4 * h;
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (co0 < a.Cout) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + co0);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[i][f][4 * g4 + j] + bv[j];
      if (a.res) {
        const bf16x4 rv = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const elt*>(a.res) + (long)pn[f] * a.res_bs + (long)(co0 >> 4) * HWo * 16 +
                                                           (long)prem[f] * 16 + (co0 & 15));
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += (float)rv[j];
        if (X3) {
          const bf16x4 rl = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const elt*>(a.res) + a.res_lo_delta + (long)pn[f] * a.res_bs +
                                                             (long)(co0 >> 4) * HWo * 16 + (long)prem[f] * 16 + (co0 & 15));
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += (float)rl[j];
        }
      }
      if (a.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
      }
    }
    return v;
  };
#pragma unroll
  for (int f = 0; f < NPF; ++f) {
    if (prem[f] < 0) continue;                                 // both lane halves of a position agree (same r)
    const long pix = (long)pn[f] * HWo + prem[f];
    if (a.out_mode == 0 && (a.Cout & 7) == 0) {
      // bf16 blocked: a lane holds channels {8 g4 + 4 h ..+3}; exchanging the odd group of the lower lane half with the even
      // group of the upper half (v_permlane32_swap) gives every lane 8 consecutive channels = one 16-byte store, and the
      // two halves of a position 32 contiguous bytes (8-byte stores were written back sector by sector: 2.5x the bytes)
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const f32x4 pv = value4(i, f, 2 * m), qv = value4(i, f, 2 * m + 1);
          const bf16x4 pb = E::cvt4(pv);
          const bf16x4 qb = E::cvt4(qv);
          const u32x2 P = __builtin_bit_cast(u32x2, pb), Q = __builtin_bit_cast(u32x2, qb);
          const auto s0 = __builtin_amdgcn_permlane32_swap(P[0], Q[0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(P[1], Q[1], false, false);
          const int co = ct * TC + i * 32 + 16 * m + 8 * h;
          elt* const yp = reinterpret_cast<elt*>(a.y) + (long)pn[f] * a.y_bs + (long)(co >> 4) * HWo * 16 + (long)prem[f] * 16 + (co & 15);
          if (co < a.Cout)                       // blocked: block co >> 4, the lane's half (8 h) of the pixel's 16-channel run
            *reinterpret_cast<u32x4*>(yp) = u32x4{s0[0], s1[0], s0[1], s1[1]};
          if (X3) {                              // the lo plane: what the hi plane's rounding left over
            f32x4 pl, ql;
#pragma unroll
            for (int j = 0; j < 4; ++j) { pl[j] = pv[j] - (float)pb[j]; ql[j] = qv[j] - (float)qb[j]; }
            const u32x2 PL = __builtin_bit_cast(u32x2, E::cvt4(pl)), QL = __builtin_bit_cast(u32x2, E::cvt4(ql));
            const auto t0 = __builtin_amdgcn_permlane32_swap(PL[0], QL[0], false, false);
            const auto t1 = __builtin_amdgcn_permlane32_swap(PL[1], QL[1], false, false);
            if (co < a.Cout) *reinterpret_cast<u32x4*>(yp + a.y_lo_delta) = u32x4{t0[0], t1[0], t0[1], t1[1]};
          }
        }
      continue;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int co0 = ct * TC + i * 32 + 8 * g4 + 4 * h;
        if (co0 >= a.Cout) continue;
        const f32x4 v = value4(i, f, g4);
        if (a.out_mode == 0) {
          elt* const yp = reinterpret_cast<elt*>(a.y) + (long)pn[f] * a.y_bs + (long)(co0 >> 4) * HWo * 16 + (long)prem[f] * 16 + (co0 & 15);
          const bf16x4 vh = E::cvt4(v);
          *reinterpret_cast<bf16x4*>(yp) = vh;
          if (X3) {
            f32x4 vl;
#pragma unroll
            for (int j = 0; j < 4; ++j) vl[j] = v[j] - (float)vh[j];
            *reinterpret_cast<bf16x4*>(yp + a.y_lo_delta) = E::cvt4(vl);
          }
        } else if (a.out_mode == 2) {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + pix * a.y_bs + co0) = v;
        } else {
          float* yb = reinterpret_cast<float*>(a.y) + ((long)pn[f] * a.Cout + co0) * HWo + prem[f];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co0 + j < a.Cout) yb[j * HWo] = (co0 + j < a.nsig) ? cb_sigmoid(v[j]) : v[j];
        }
      }
  }
}

template <int KS, int S, int KC, int NPF, int NA, int FUSE = 0, bool P3 = false, bool F16 = false, bool X3 = false>
static int cb_launch(CbArgs& a, hipStream_t st) {
  using G = CbGeom<KS, S, KC, NPF>;
  constexpr int WIMG = KS * KS * NA * 32 * KC * 2;
  constexpr int LDS = P3 ? 2 * G::WIN_BYTES + 3 * ((3 * NA * 32 * KC * 2 + 8191) / 8192) * 8192
                         : 2 * (G::WIN_BYTES + ((WIMG + 8191) / 8192) * 8192);
  static_assert(LDS <= 160 * 1024, "LDS budget");
  auto kern = conv_bf16_kernel<KS, S, KC, NPF, NA, FUSE, P3, F16, X3>;
  static bool attr_by_dev[SGTA_MAX_DEVICES] = {};
  bool& done = attr_by_dev[sgta_device_slot()];
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return SGTA_EBADARG;
    done = true;
  }
  const int n_pt = a.n_rt * a.n_xt;
  const int grid = cdiv(n_pt, 8) * 8 * (FUSE ? 1 : a.n_ct);
  // a single-chunk layer (level1: Cin = 16) only ever uses stage 0 of the two-stage loop: half the LDS = two work-groups per
  // CU for a kernel that is one DMA round trip, nine MFMAs and an epilogue per work-group
  // stride 2 (40 KB windows: two stages = one work-group per CU): ONE stage is 14 % faster at every depth of the network;
  // SGTA_CONV_SINGLE1 = chunks up to which the 1x1 convolutions use one stage (tools)
  static const int single_s2 = sgta_tool_env("SGTA_CONV_SINGLE", 1 << 30);
  static const int single_1x1 = sgta_tool_env("SGTA_CONV_SINGLE1", 0);
  a.single = (!P3 && !FUSE && ((KS == 3 && S == 2 && a.nchunks <= single_s2) || (KS == 1 && a.nchunks <= single_1x1))) ? 1 : 0;
  const int lds = (!P3 && !FUSE && (a.nchunks == 1 || a.single)) ? LDS / 2 : LDS;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// x: bf16 blocked [N][Cb][H][W][16] (pointer at the first 16-channel block of the view; x_bs = elements per image of the
// underlying buffer; x_bytes = bytes from x to the end of the allocation that may be addressed, < 2^31); wt: ops_bf16.py
// layout for (ks, stride, kc, na); bias f32 [n_ct*32*na]; res: bf16 blocked at the output resolution (res_bs elements
// per image) or NULL (added before the activation); out_mode 0: y bf16 blocked (y_bs elements per image), 2: y f32 NHWC
// [N,Ho,Wo,y_bs], 1: y f32 NCHW [N,Cout,Ho,Wo] with sigmoid on channels < nsig.  act: 0 none, 1 ReLU.
// npf: 32-position fragments per wave (2: 512-position tiles; 1: 256; stride 2 needs 1); log_tw: log2 of the tile width
// (3..6).  Cin % kc == 0, kc in {16 (3x3), 32 (1x1)}.
template <bool F16, bool X3 = false>
static int cb_conv(const void* x, long x_bytes, long x_bs, const void* wt, long wt_bytes, const float* bias,
                   const void* res, long res_bs, void* y, long y_bs, int N, int H, int W, int Cin, int Cout,
                   int ks, int stride, int kc, int na, int npf, int log_tw, int act, int out_mode, int nsig,
                   hipStream_t st, long x_lo_delta = 0, long res_lo_delta = 0, long y_lo_delta = 0) {
  if (X3 && (x_lo_delta <= 0 || x_lo_delta * 2 > 0x7fffffffL || x_lo_delta % 8 || (out_mode == 0 && (y_lo_delta <= 0 || y_lo_delta % 8)) ||
             (res && (res_lo_delta <= 0 || res_lo_delta % 4))))
    return SGTA_EBADARG;
  if (!x || !wt || !bias || !y || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SGTA_EBADARG;
  if (x_bytes <= 0 || x_bytes > 0x7fffffffL || wt_bytes <= 0 || wt_bytes > 0x7fffffffL) return SGTA_EBADARG;
  if (Cin % kc || (x_bs % 16) || ((uintptr_t)x & 15) || log_tw < 3 || log_tw > 6 || act < 0 || act > 1 || out_mode < 0 ||
      out_mode > 2)
    return SGTA_EBADARG;
  if (out_mode == 2 && (y_bs % 4 || Cout % 4)) return SGTA_EBADARG;
  if (out_mode == 0 && (Cout % 4 || y_bs % 16 || ((uintptr_t)y & 15))) return SGTA_EBADARG;        // 16-byte stores
  if (res && (res_bs % 16 || ((uintptr_t)res & 7))) return SGTA_EBADARG;
  CbArgs a{};
  a.x = x; a.x_bytes = x_bytes; a.x_bs = x_bs; a.wt = wt; a.wt_bytes = wt_bytes; a.bias = bias;
  a.res = res; a.res_bs = res_bs; a.y = y; a.y_bs = y_bs;
  a.N = N; a.H = H; a.W = W; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1; a.Cout = Cout;
  a.nchunks = (X3 ? 3 : 1) * (Cin / kc); a.n_ct = cdiv(Cout, 32 * na); a.log_tw = log_tw; a.act = act; a.out_mode = out_mode; a.nsig = nsig;
  a.x_lo_delta = (int)(x_lo_delta * 2); a.res_lo_delta = res_lo_delta; a.y_lo_delta = y_lo_delta;
  if (npf != 1 && npf != 2) return SGTA_EBADARG;
  const int TW = 1 << log_tw, TR = (256 * npf) >> log_tw;
  a.n_rt = cdiv((long)N * (a.Ho + 1), TR);
  a.n_xt = cdiv(a.Wo, TW);
  static const bool p3 = sgta_tool_env("SGTA_CONV_P3", 1) != 0;   // tools: 0 = the two-stage loop
  if (ks == 3 && stride == 1 && kc == 16 && npf == 2) {
    if (na == 1) return p3 ? cb_launch<3, 1, 16, 2, 1, 0, true, F16, X3>(a, st) : cb_launch<3, 1, 16, 2, 1, 0, false, F16, X3>(a, st);
    if (na == 2) return p3 ? cb_launch<3, 1, 16, 2, 2, 0, true, F16, X3>(a, st) : cb_launch<3, 1, 16, 2, 2, 0, false, F16, X3>(a, st);
    if (na == 4) return p3 ? cb_launch<3, 1, 16, 2, 4, 0, true, F16, X3>(a, st) : cb_launch<3, 1, 16, 2, 4, 0, false, F16, X3>(a, st);
  } else if (ks == 3 && stride == 1 && kc == 16 && npf == 1) {       // half-size tiles: small maps (work-group count / tail)
    if (na == 1) return p3 ? cb_launch<3, 1, 16, 1, 1, 0, true, F16, X3>(a, st) : cb_launch<3, 1, 16, 1, 1, 0, false, F16, X3>(a, st);
    if (na == 2) return p3 ? cb_launch<3, 1, 16, 1, 2, 0, true, F16, X3>(a, st) : cb_launch<3, 1, 16, 1, 2, 0, false, F16, X3>(a, st);
    if (na == 4) return p3 ? cb_launch<3, 1, 16, 1, 4, 0, true, F16, X3>(a, st) : cb_launch<3, 1, 16, 1, 4, 0, false, F16, X3>(a, st);
  } else if (ks == 3 && stride == 2 && kc == 16 && npf == 1) {
    static const bool p3s2 = sgta_tool_env("SGTA_CONV_P3S2", 0) == 1;
    if (na == 1) return p3s2 ? cb_launch<3, 2, 16, 1, 1, 0, true, F16, X3>(a, st) : cb_launch<3, 2, 16, 1, 1, 0, false, F16, X3>(a, st);
    if (na == 2) return p3s2 ? cb_launch<3, 2, 16, 1, 2, 0, true, F16, X3>(a, st) : cb_launch<3, 2, 16, 1, 2, 0, false, F16, X3>(a, st);
  } else if (ks == 1 && stride == 1 && kc == 32 && npf == 2) {
    if (na == 1) return cb_launch<1, 1, 32, 2, 1, 0, false, F16, X3>(a, st);
    if (na == 2) return cb_launch<1, 1, 32, 2, 2, 0, false, F16, X3>(a, st);
    if (na == 4) return cb_launch<1, 1, 32, 2, 4, 0, false, F16, X3>(a, st);
  } else if (ks == 1 && stride == 1 && kc == 32 && npf == 1) {
    if (na == 1) return cb_launch<1, 1, 32, 1, 1, 0, false, F16, X3>(a, st);
    if (na == 2) return cb_launch<1, 1, 32, 1, 2, 0, false, F16, X3>(a, st);
    if (na == 4) return cb_launch<1, 1, 32, 1, 4, 0, false, F16, X3>(a, st);
  }
  return SGTA_EBADARG;
}

extern "C" int sgta_conv_bf16(const void* x, long x_bytes, long x_bs, const void* wt, long wt_bytes, const float* bias,
                              const void* res, long res_bs, void* y, long y_bs, int N, int H, int W, int Cin, int Cout,
                              int ks, int stride, int kc, int na, int npf, int log_tw, int act, int out_mode, int nsig,
                              hipStream_t st) {
  return cb_conv<false>(x, x_bytes, x_bs, wt, wt_bytes, bias, res, res_bs, y, y_bs, N, H, W, Cin, Cout, ks, stride, kc, na, npf,
                        log_tw, act, out_mode, nsig, st);
}

// The same convolution on IEEE-half feature maps and weights (v_mfma_f32_32x32x16_f16): identical layouts and arguments.
extern "C" int sgta_conv_f16(const void* x, long x_bytes, long x_bs, const void* wt, long wt_bytes, const float* bias,
                             const void* res, long res_bs, void* y, long y_bs, int N, int H, int W, int Cin, int Cout,
                             int ks, int stride, int kc, int na, int npf, int log_tw, int act, int out_mode, int nsig,
                             hipStream_t st) {
  return cb_conv<true>(x, x_bytes, x_bs, wt, wt_bytes, bias, res, res_bs, y, y_bs, N, H, W, Cin, Cout, ks, stride, kc, na, npf,
                       log_tw, act, out_mode, nsig, st);
}

// The three head branches in one kernel (base_model.py:121-135): hidden = relu(conv3x3(x) + b1) with the branches' 3x3 convs
// concatenated (Chid = n_ct * 128 channels, w1 / b1 as sgta_conv_bf16 with na = 4), out = W2 . hidden + b2 with the
// block-diagonal 1x1 weights of all branches (<= 16 output rows in total), written as f32 NCHW per head: out_i [N, c_i, H, W],
// sigmoid on rows < nsig.  w2: bf16 A fragments [Chid/128][4][2][64 lanes][8] of the second v_mfma_f32_32x32x16_bf16: lane
// (m = lane & 31, h = lane >> 5), element j of k-step t of 32-channel block i holds W2[m][128 ct + 32 i + 8 (2t + (j >> 2)) +
// 4h + (j & 3)] (zero for m >= the number of rows).  npf / log_tw as sgta_conv_bf16.  Cin % 16 == 0, Chid % 128 == 0.
template <bool F16, bool X3 = false>
static int cb_heads(const void* x, long x_bytes, long x_bs, const void* w1, long w1_bytes, const float* b1,
                    const void* w2, const float* b2, float* out0, float* out1, float* out2, float* out3, int c0,
                    int c1, int c2, int c3, int N, int H, int W, int Cin, int Chid, int npf, int log_tw, int nsig,
                    hipStream_t st, long x_lo_delta = 0, const void* w2_lo = nullptr) {
  if (X3 && (x_lo_delta <= 0 || x_lo_delta * 2 > 0x7fffffffL || x_lo_delta % 8 || !w2_lo || ((uintptr_t)w2_lo & 15))) return SGTA_EBADARG;
  if (!x || !w1 || !b1 || !w2 || !b2 || !out0 || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Chid <= 0) return SGTA_EBADARG;
  if (x_bytes <= 0 || x_bytes > 0x7fffffffL || w1_bytes <= 0 || w1_bytes > 0x7fffffffL) return SGTA_EBADARG;
  if (Cin % 16 || Chid % 128 || (x_bs % 16) || ((uintptr_t)x & 15) || ((uintptr_t)w2 & 15) || log_tw < 3 || log_tw > 6) return SGTA_EBADARG;
  if (c0 <= 0 || c1 < 0 || c2 < 0 || c3 < 0 || c0 + c1 + c2 + c3 > 16) return SGTA_EBADARG;
  if ((c1 > 0 && !out1) || (c2 > 0 && !out2) || (c3 > 0 && !out3)) return SGTA_EBADARG;
  CbArgs a{};
  a.x = x; a.x_bytes = x_bytes; a.x_bs = x_bs; a.wt = w1; a.wt_bytes = w1_bytes; a.bias = b1;
  a.N = N; a.H = H; a.W = W; a.Ho = H; a.Wo = W; a.Cout = Chid; a.nchunks = (X3 ? 3 : 1) * (Cin / 16); a.n_ct = Chid / 128;
  a.x_lo_delta = (int)(x_lo_delta * 2); a.w2_lo = w2_lo;
  a.log_tw = log_tw; a.act = 1; a.out_mode = 1; a.nsig = nsig;
  a.w2 = w2; a.b2 = b2; a.out[0] = out0; a.out[1] = out1; a.out[2] = out2; a.out[3] = out3;
  a.hs[0] = 0; a.hs[1] = c0; a.hs[2] = c0 + c1; a.hs[3] = c0 + c1 + c2; a.hs[4] = c0 + c1 + c2 + c3;
  const int TW = 1 << log_tw, TR = (256 * npf) >> log_tw;
  a.n_rt = cdiv((long)N * (H + 1), TR);
  a.n_xt = cdiv(W, TW);
  // (the two-stage loop: with six channel tiles per work-group and no output tensor its waits are already hidden, and the
  // three-deep form's extra barriers cost it 9 % here)
  if (npf == 2) return cb_launch<3, 1, 16, 2, 4, 1, false, F16, X3>(a, st);
  if (npf == 1) return cb_launch<3, 1, 16, 1, 4, 1, false, F16, X3>(a, st);
  return SGTA_EBADARG;
}

extern "C" int sgta_heads_bf16(const void* x, long x_bytes, long x_bs, const void* w1, long w1_bytes, const float* b1,
                               const void* w2, const float* b2, float* out0, float* out1, float* out2, float* out3, int c0,
                               int c1, int c2, int c3, int N, int H, int W, int Cin, int Chid, int npf, int log_tw, int nsig,
                               hipStream_t st) {
  return cb_heads<false>(x, x_bytes, x_bs, w1, w1_bytes, b1, w2, b2, out0, out1, out2, out3, c0, c1, c2, c3, N, H, W, Cin, Chid,
                         npf, log_tw, nsig, st);
}

extern "C" int sgta_heads_f16(const void* x, long x_bytes, long x_bs, const void* w1, long w1_bytes, const float* b1,
                              const void* w2, const float* b2, float* out0, float* out1, float* out2, float* out3, int c0,
                              int c1, int c2, int c3, int N, int H, int W, int Cin, int Chid, int npf, int log_tw, int nsig,
                              hipStream_t st) {
  return cb_heads<true>(x, x_bytes, x_bs, w1, w1_bytes, b1, w2, b2, out0, out1, out2, out3, c0, c1, c2, c3, N, H, W, Cin, Chid,
                        npf, log_tw, nsig, st);
}

// ---- fp32 emulated on the 16-bit matrix pipe by operand splitting (see the kernel's X3 note).  Same layouts as sgta_conv_f16
// with every activation tensor in TWO IEEE-half planes of identical layout (value = hi + lo; *_lo_delta = elements from a
// tensor's hi plane to its lo plane, both inside one allocation), Cin = the layer's real input channels, and wt packed
// over 3 Cin virtual channels: per kc-channel chunk [w_hi, w_lo, w_hi] with w_hi = half(w), w_lo = half(w - w_hi).
// Results agree with an fp32 convolution to ~1e-6 relative.
extern "C" int sgta_conv_x3(const void* x, long x_bytes, long x_bs, long x_lo_delta, const void* wt, long wt_bytes, const float* bias,
                            const void* res, long res_bs, long res_lo_delta, void* y, long y_bs, long y_lo_delta, int N, int H,
                            int W, int Cin, int Cout, int ks, int stride, int kc, int na, int npf, int log_tw, int act,
                            int out_mode, int nsig, hipStream_t st) {
  return cb_conv<true, true>(x, x_bytes, x_bs, wt, wt_bytes, bias, res, res_bs, y, y_bs, N, H, W, Cin, Cout, ks, stride, kc, na, npf,
                             log_tw, act, out_mode, nsig, st, x_lo_delta, res_lo_delta, y_lo_delta);
}

// sgta_heads_f16 with split operands: w1 over 3 Cin virtual channels as sgta_conv_x3, w2 / w2_lo = hi and lo parts of the
// second product's fragments; the hidden activations are split in registers.
extern "C" int sgta_heads_x3(const void* x, long x_bytes, long x_bs, long x_lo_delta, const void* w1, long w1_bytes, const float* b1,
                             const void* w2, const void* w2_lo, const float* b2, float* out0, float* out1, float* out2,
                             float* out3, int c0, int c1, int c2, int c3, int N, int H, int W, int Cin, int Chid, int npf,
                             int log_tw, int nsig, hipStream_t st) {
  return cb_heads<true, true>(x, x_bytes, x_bs, w1, w1_bytes, b1, w2, b2, out0, out1, out2, out3, c0, c1, c2, c3, N, H, W, Cin, Chid,
                              npf, log_tw, nsig, st, x_lo_delta, w2_lo);
}
