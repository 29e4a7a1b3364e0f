// 1x1 convolution (+ folded BN bias, residual, activation) = a plain GEMM D[co, m] = W[co, c] X[c, m] over the
// flattened pixels m = (b, p) of an NCHW tensor (Root / project convs of the DLA trees, second conv of the
// heads; reference dla.py:157-175, :211-216, base_model.py:122-123).  Channel planes are contiguous rows, so
// both operands are staged by 16-byte LDS-DMA with no index math in the loop: X chunk [16 c][256 m] and the
// host-padded weight rows [BM][16+4] land in a ring of LDS stages; f32 MFMA 32x32x2 with the pixel on the lane
// (same C/D mapping and epilogue as conv_igemm.hip).  MFMA k-slot (lane half) lh takes channels lh*8 .. lh*8+7
// of a chunk: the A fragment of a chunk is two conflict-free ds_read_b128 (80-B rows).
#include "common.h"

struct C1Args {
  const float* x;  long x_bs;
  const float* wt;                 // [Cin_pad/16][Cout_pad][20]  (16 k + 4 pad floats per row)
  const float* bias;               // [Cout_pad]
  const float* res; long res_bs;
  float* y;        long y_bs;
  int B, Cin, HW, Cout, Cout_pad, nchunks, M, n_ct, n_pt;
};

__device__ float sgta_c1_zero[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void c1_glds16(const float* g, float* l) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
#endif
}

// BUF: buffer-addressed LDS-DMA (per-slot byte offsets are chunk-invariant registers, the chunk advance is the scalar
// offset, pixel quads beyond M carry an out-of-range offset that the bounds check turns into zeros); needs Cin % 16 == 0
// and the whole activation tensor inside one 31-bit descriptor -- else the per-lane-address form
template <int MT, int ACT, int RES, bool BUF>
__global__ __launch_bounds__(256, 2) void conv1x1_kernel(C1Args a) {
  constexpr int NT = 2, BM = MT * 32, BN = 256, CK = 16, LDA = 20, NSTG = 2;
  constexpr int B_SLOTS = CK * BN / 4, A_SLOTS = BM * LDA / 4;          // float4 slots per stage
  constexpr int B_PER = B_SLOTS / 256, A_PER = (A_SLOTS + 255) / 256;
  constexpr int STAGE = B_SLOTS * 4 + A_PER * 1024;
  __shared__ __attribute__((aligned(16))) float lds[NSTG * STAGE + BM];
  float* const sbias = lds + NSTG * STAGE;

  int pt, ct;
  xcd_remap(blockIdx.x, a.n_ct, pt, ct);
  if (pt >= a.n_pt) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int HW = a.HW;
  if (tid < BM) sbias[tid] = a.bias[ct * BM + tid];

  // X slots: slot e = tid + t*256 <-> channel row e / 64, pixel quad e % 64 (4 consecutive m of one image)
  long xoff[B_PER];
  int xch[B_PER];
#pragma unroll
  for (int t = 0; t < B_PER; ++t) {
    const int e = tid + t * 256;
    const int m = pt * BN + (e & 63) * 4;
    xch[t] = e >> 6;
    if (m < a.M) {
      const int b = m / HW, p = m - b * HW;
      xoff[t] = (long)b * a.x_bs + p;
    } else {
      xoff[t] = -1;
    }
  }
  unsigned boff[B_PER], aoffb[A_PER];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t xrs, wrs;
  if (BUF) {
#pragma unroll
    for (int t = 0; t < B_PER; ++t)
      boff[t] = xoff[t] >= 0 ? (unsigned)((xoff[t] + (long)xch[t] * HW) * 4) : 0x80000000u;
#pragma unroll
    for (int t = 0; t < A_PER; ++t) {
      const int e = tid + t * 256;                     // slots beyond the tile: out of range -> zeros into the stage's padding
      aoffb[t] = e < A_SLOTS ? (unsigned)(ct * BM * LDA + e * 4) * 4u : 0x80000000u;
    }
    xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), (short)0,
                                            (int)((((long)a.B - 1) * a.x_bs + (long)a.Cin * HW) * 4), 0x00020000);
    wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wt), (short)0, a.nchunks * a.Cout_pad * LDA * 4,
                                            0x00020000);
  }
#endif
  auto stage = [&](int kc, int st) {
    float* dst = lds + st * STAGE + wave * 256;
    float* adst = lds + st * STAGE + B_SLOTS * 4 + wave * 256;
    if (BUF) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int t = 0; t < B_PER; ++t)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(dst + t * 1024), 16, boff[t], kc * CK * HW * 4, 0, 0);
#pragma unroll
      for (int t = 0; t < A_PER; ++t)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(adst + t * 1024), 16, aoffb[t], kc * a.Cout_pad * LDA * 4, 0, 0);
#endif
      return;
    }
#pragma unroll
    for (int t = 0; t < B_PER; ++t) {
      const int c = kc * CK + xch[t];
      const bool ok = xoff[t] >= 0 && c < a.Cin;
      c1_glds16(ok ? a.x + xoff[t] + (long)c * HW : sgta_c1_zero, dst + t * 1024);
    }
    const float* src = a.wt + ((long)kc * a.Cout_pad + ct * BM) * LDA;
#pragma unroll
    for (int t = 0; t < A_PER; ++t)
      if (A_SLOTS % 256 == 0 || tid + t * 256 < A_SLOTS) c1_glds16(src + (tid + t * 256) * 4, adst + t * 1024);
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_s_barrier();

  for (int kc = 0; kc < a.nchunks; ++kc) {
    const int st = kc & 1;
    if (kc + 1 < a.nchunks) stage(kc + 1, st ^ 1);
    const float* Bs = lds + st * STAGE;
    const float* As = Bs + B_SLOTS * 4;
    f32x4 af[2][MT];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < MT; ++i) af[h][i] = *reinterpret_cast<const f32x4*>(As + (i * 32 + l31) * LDA + lh * 8 + h * 4);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float bv[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = Bs[(lh * 8 + s) * BN + (wave * NT + j) * 32 + l31];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s >> 2][i][s & 3], bv[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int m = pt * BN + (wave * NT + j) * 32 + l31;
    if (m >= a.M) continue;
    const int b = m / HW, p = m - b * HW;
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
static int launch_c1(C1Args& a, hipStream_t st) {
  auto go = [&](auto kern, int BM) -> int {
    a.n_ct = a.Cout_pad / BM;
    a.n_pt = cdiv(a.M, 256);
    int grid = cdiv(a.n_pt, 8) * 8 * a.n_ct;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, a);
    SGTA_LAUNCH_CHECK();
    return SGTA_OK;
  };
  // buffer-addressed staging: whole channel chunks, every offset (incl. one chunk beyond: the scalar part) in 31 bits
  const bool buf = a.Cin % 16 == 0 && (((long)a.B - 1) * a.x_bs + (long)a.Cin * a.HW) * 4 <= 0x7fffffffL &&
                   (long)a.nchunks * a.Cout_pad * 20 * 4 <= 0x7fffffffL;
  if (a.Cout_pad % 64 == 0) return buf ? go(conv1x1_kernel<2, ACT, RES, true>, 64) : go(conv1x1_kernel<2, ACT, RES, false>, 64);
  return buf ? go(conv1x1_kernel<1, ACT, RES, true>, 32) : go(conv1x1_kernel<1, ACT, RES, false>, 32);
}

extern "C" int sgta_conv1x1_f32(const float* x, long x_bs, const float* wt, const float* bias,
                                const float* res, long res_bs, float* y, long y_bs, int B, int Cin, int H,
                                int W, int Cout, int Cout_pad, int Cin_pad, int act, int res_mode,
                                hipStream_t st) {
  if (!x || !wt || !bias || !y || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return SGTA_EBADARG;
  if (Cout_pad % 32 || Cout_pad < Cout || Cin_pad % 16 || Cin_pad < Cin) return SGTA_EBADARG;
  if ((res_mode != 0) != (res != nullptr)) return SGTA_EBADARG;
  // 16-byte LDS-DMA of pixel quads: planes and batch strides must keep 4-pixel groups 16-byte aligned
  if ((H * W) % 4 || x_bs % 4 || ((uintptr_t)x & 15)) return SGTA_EBADARG;
  long M = (long)B * H * W;
  if (M > 0x7fffffffL) return SGTA_EBADARG;
  C1Args a{x, x_bs, wt, bias, res, res_bs, y, y_bs, B, Cin, H * W, Cout, Cout_pad, Cin_pad / 16, (int)M, 0, 0};
  if (act == 1 && res_mode == 0) return launch_c1<1, 0>(a, st);
  if (act == 0 && res_mode == 0) return launch_c1<0, 0>(a, st);
  if (act == 1 && res_mode == 1) return launch_c1<1, 1>(a, st);
  if (act == 0 && res_mode == 1) return launch_c1<0, 1>(a, st);     // partial sums of the T-frame full-map cat-MLP
  if (act == 1 && res_mode == 2) return launch_c1<1, 2>(a, st);
  if (act == 2 && res_mode == 0) return launch_c1<2, 0>(a, st);
  return SGTA_EBADARG;
}

// ---- "skinny" 1x1 conv: Cout <= 8 (the output conv of every head, base_model.py:122-123) --------------------
// 7 / 2 / 2 output channels cannot fill an MFMA tile and the op is a pure stream over the hidden activations
// (256 x HW floats in, <= 8 x HW out): one thread owns 4 consecutive pixels, walks the channel planes with
// 16-byte loads (8 in flight per lane) and keeps Cout x 4 accumulators; the weights are wave-uniform scalar loads.
template <int CO, int ACT>
__global__ __launch_bounds__(128) void conv1x1_skinny_kernel(const float* __restrict__ x, long x_bs,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ y, long y_bs, int Cin, int HW,
                                                             int quads_per_img, long nquads) {
  constexpr int U = 8;
  const long q = blockIdx.x * 128L + threadIdx.x;
  if (q >= nquads) return;
  const int b = (int)(q / quads_per_img);
  const int p = (int)(q - (long)b * quads_per_img) * 4;
  const float* xp = x + (long)b * x_bs + p;
  f32x4 acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) { const float bo = bias[o]; acc[o] = f32x4{bo, bo, bo, bo}; }
  int c = 0;
  for (; c + U <= Cin; c += U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xp + (long)(c + u) * HW));
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int o = 0; o < CO; ++o) {
        const float wv = w[o * Cin + c + u];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[o][e] = fmaf(wv, v[u][e], acc[o][e]);
      }
  }
  for (; c < Cin; ++c) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xp + (long)c * HW);
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      const float wv = w[o * Cin + c];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[o][e] = fmaf(wv, v[e], acc[o][e]);
    }
  }
  float* yp = y + (long)b * y_bs + p;
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    f32x4 r = acc[o];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (ACT == 1) r[e] = fmaxf(r[e], 0.0f);
      if (ACT == 2) r[e] = sigmoidf_(r[e]);
    }
    *reinterpret_cast<f32x4*>(yp + (long)o * HW) = r;
  }
}

template <int CO>
static int launch_skinny(const float* x, long x_bs, const float* w, const float* bias, float* y, long y_bs, int Cin,
                         int HW, long nquads, int act, hipStream_t st) {
  const int grid = (int)((nquads + 127) / 128);
#define SGTA_SK(ACT) hipLaunchKernelGGL((conv1x1_skinny_kernel<CO, ACT>), dim3(grid), dim3(128), 0, st, x, x_bs, w, bias, y, y_bs, Cin, HW, HW / 4, nquads)
  if (act == 0) SGTA_SK(0);
  else if (act == 1) SGTA_SK(1);
  else if (act == 2) SGTA_SK(2);
  else return SGTA_EBADARG;
#undef SGTA_SK
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

extern "C" int sgta_conv1x1_skinny_f32(const float* x, long x_bs, const float* w, const float* bias, float* y,
                                       long y_bs, int B, int Cin, int H, int W, int Cout, int act,
                                       hipStream_t st) {
  if (!x || !w || !bias || !y || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout > 8) return SGTA_EBADARG;
  const int HW = H * W;
  if (HW % 4 || x_bs % 4 || y_bs % 4 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return SGTA_EBADARG;
  const long nquads = (long)B * (HW / 4);
  if (nquads > 0x7fffffffL * 32) return SGTA_EBADARG;
  switch (Cout) {
    case 1: return launch_skinny<1>(x, x_bs, w, bias, y, y_bs, Cin, HW, nquads, act, st);
    case 2: return launch_skinny<2>(x, x_bs, w, bias, y, y_bs, Cin, HW, nquads, act, st);
    case 3: return launch_skinny<3>(x, x_bs, w, bias, y, y_bs, Cin, HW, nquads, act, st);
    case 4: return launch_skinny<4>(x, x_bs, w, bias, y, y_bs, Cin, HW, nquads, act, st);
    case 5: return launch_skinny<5>(x, x_bs, w, bias, y, y_bs, Cin, HW, nquads, act, st);
    case 6: return launch_skinny<6>(x, x_bs, w, bias, y, y_bs, Cin, HW, nquads, act, st);
    case 7: return launch_skinny<7>(x, x_bs, w, bias, y, y_bs, Cin, HW, nquads, act, st);
    default: return launch_skinny<8>(x, x_bs, w, bias, y, y_bs, Cin, HW, nquads, act, st);
  }
}
