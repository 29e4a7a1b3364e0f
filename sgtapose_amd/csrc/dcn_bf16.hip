// bf16 data path: modulated deformable 3x3 convolution (lbin/DCNv2 semantics) + folded BN +
// ReLU on bf16 NHWC feature maps; replaces DeformConv (dla.py:538-550) like sgta_dcnv2_lds_f32 does in fp32.
//
// Channels-last makes the deformable gather coalesced: one sampling corner of one pixel is a run of consecutive
// channels, so a lane fetches 8 channels (16 bytes) per corner, blends the four corners in fp32 with its per-(pixel, tap)
// weights (bilinear weight x validity x sigmoid(mask), computed once per tap from the fp32 offset / mask map) and has
// the bf16 MFMA B operand of that (pixel, tap, 8 channels) in registers -- the [9*Cin, H*W] column buffer of the
// reference never exists.  A = the bf16 weights [cout][tap][cin], staged per (tap, 64-channel group) by 16-byte buffer
// LDS-DMA (double buffered, XOR-swizzled rows), shared by the 8 waves of the work-group; every gathered sample feeds
// all TC = 64 / 128 output channels of the work-group.  Accumulation fp32 on v_mfma_f32_32x32x16_bf16.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct DbArgs {
  const __bf16* x;  int ldx;                     // bf16 NHWC input [N,H,W,ldx] (pointer at the view's first channel)
  const float* om;  int ldo;                     // f32 NHWC offsets / mask logits [N,H,W,ldo >= 27]
  const void* wt;   long wt_bytes;               // packed [n_ct][9][Cin/64][TC][64] bf16, swizzled
  const float* bias;                             // [n_ct * TC]
  __bf16* y;        int ldy;                     // bf16 NHWC output
  int N, H, W, Cin, Cout, ngroups, n_ct, n_rt, n_xt, log_tw, act;
};

template <int NPF, int NA>
__global__ __launch_bounds__(512) void dcn_bf16_kernel(DbArgs a) {
  constexpr int TC = NA * 32, TP = 256 * NPF, ROWB = 128, SIMG = TC * ROWB, AROUNDS = (SIMG + 8191) / 8192;
  constexpr int STAGE = AROUNDS * 8192;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  int pt, ct;
  xcd_remap(blockIdx.x, a.n_ct, pt, ct);
  if (pt >= a.n_rt * a.n_xt) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int TW = 1 << a.log_tw, TR = TP >> a.log_tw;
  const int rt = pt / a.n_xt, xt = pt - rt * a.n_xt;
  const int H = a.H, W = a.W;

  unsigned aoffw[AROUNDS];
#pragma unroll
  for (int t = 0; t < AROUNDS; ++t) {
    const int e = (t * 512 + tid) * 16;
    aoffw[t] = e < SIMG ? (unsigned)e : 0x80000000u;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wt), (short)0, (int)a.wt_bytes, 0x00020000);
#endif
  const int nstages = 9 * a.ngroups;
  auto stage = [&](int q, int st) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned char* adst = smem + st * STAGE + wave * 1024;
#pragma unroll
    for (int t = 0; t < AROUNDS; ++t)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(adst + t * 8192), 16, aoffw[t],
                                               (ct * nstages + q) * SIMG, 0, 0);
#endif
  };

  // ---- this lane's pixels ------------------------------------------------------------------------------------
  int py0[NPF], px0[NPF];
  long pbase[NPF];                               // image base pixel index n * H * W
  long pix[NPF];                                 // own pixel index or -1
#pragma unroll
  for (int f = 0; f < NPF; ++f) {
    const int i = (wave * NPF + f) * 32 + r;
    const int R = rt * TR + (i >> a.log_tw), x = xt * TW + (i & (TW - 1));
    const int n = R / H, y = R - n * H;
    const bool ok = n < a.N && x < W;
    py0[f] = y; px0[f] = x;
    pbase[f] = (long)(ok ? n : 0) * H * W;
    pix[f] = ok ? pbase[f] + (long)y * W + x : -1;
  }
  unsigned aaddr[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) aaddr[s] = (unsigned)(r * ROWB + (((2 * s + h) ^ ((r >> 1) & 7)) << 4));

  f32x16 acc[NA][NPF];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int f = 0; f < NPF; ++f)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][f][q] = 0.0f;

  // the 27 offset / mask values of this lane's pixels, fetched once (not a dependent load in front of every tap)
  float omv[NPF][27];
#pragma unroll
  for (int f = 0; f < NPF; ++f) {
    const float* o = a.om + (pix[f] >= 0 ? pix[f] : 0) * a.ldo;
#pragma unroll
    for (int j = 0; j < 27; ++j) omv[f][j] = o[j];
  }

  stage(0, 0);
  int q = 0;
#pragma unroll 1
  for (int t = 0; t < 9; ++t) {
    // sampling parameters of tap t for this lane's pixels: 4 corner element offsets + 4 weights
    long coff[NPF][4];
    float cw[NPF][4];
#pragma unroll
    for (int f = 0; f < NPF; ++f) {
      float dy = 0.0f, dx = 0.0f, ml = 0.0f;
#pragma unroll
      for (int j = 0; j < 9; ++j)                     // register array indexed by the (runtime) tap: select chain
        if (j == t) { dy = omv[f][2 * j]; dx = omv[f][2 * j + 1]; ml = omv[f][18 + j]; }
      const float m = 1.0f / (1.0f + __expf(-ml));
      const float py = (float)(py0[f] - 1 + t / 3) + dy, px = (float)(px0[f] - 1 + t % 3) + dx;
      const bool inside = pix[f] >= 0 && py > -1.0f && px > -1.0f && py < (float)H && px < (float)W;
      const float fy = floorf(py), fx = floorf(px);
      const float ly = py - fy, lx = px - fx, hy = 1.0f - ly, hx = 1.0f - lx;
      const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
      const bool oy0 = y0 >= 0, oy1 = y1 <= H - 1, ox0 = x0 >= 0, ox1 = x1 <= W - 1;
      const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
      const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
      coff[f][0] = (pbase[f] + (long)cy0 * W + cx0) * a.ldx;
      coff[f][1] = (pbase[f] + (long)cy0 * W + cx1) * a.ldx;
      coff[f][2] = (pbase[f] + (long)cy1 * W + cx0) * a.ldx;
      coff[f][3] = (pbase[f] + (long)cy1 * W + cx1) * a.ldx;
      cw[f][0] = (inside && oy0 && ox0) ? hy * hx * m : 0.0f;
      cw[f][1] = (inside && oy0 && ox1) ? hy * lx * m : 0.0f;
      cw[f][2] = (inside && oy1 && ox0) ? ly * hx * m : 0.0f;
      cw[f][3] = (inside && oy1 && ox1) ? ly * lx * m : 0.0f;
    }
#pragma unroll 1
    for (int g = 0; g < a.ngroups; ++g, ++q) {
      const int st = q & 1;
      // all 16 corner runs of this 64-channel group in flight together (the blend of k-step s waits only for its own)
      bf16x8 cv[4][NPF][4];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int f = 0; f < NPF; ++f)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            cv[s][f][k] = *reinterpret_cast<const bf16x8*>(a.x + coff[f][k] + g * 64 + s * 16 + h * 8);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16 * NPF) : "memory");          // the weight DMA of this stage (older than the loads)
      __builtin_amdgcn_s_barrier();
      if (q + 1 < nstages) stage(q + 1, st ^ 1);
      const unsigned char* As = smem + st * STAGE;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 bfr[NPF];
#pragma unroll
        for (int f = 0; f < NPF; ++f) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float s0 = cw[f][0] * (float)cv[s][f][0][j];
            s0 = __builtin_fmaf(cw[f][1], (float)cv[s][f][1][j], s0);
            s0 = __builtin_fmaf(cw[f][2], (float)cv[s][f][2][j], s0);
            v[j] = __builtin_fmaf(cw[f][3], (float)cv[s][f][3][j], s0);
          }
          typedef float f32x8 __attribute__((ext_vector_type(8)));
          const f32x8 vv = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
          bfr[f] = __builtin_convertvector(vv, bf16x8);
        }
        bf16x8 af[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * ROWB + aaddr[s]);
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int f = 0; f < NPF; ++f)
            acc[i][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[f], acc[i][f], 0, 0, 0);
      }
    }
  }

  // epilogue: bias + ReLU, then v_permlane32_swap pairs the two lane halves' channel groups so that every lane stores 8
  // consecutive channels (16 bytes) and a pixel's two lanes 32 contiguous bytes (see conv_bf16.hip)
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto value4 = [&](int i, int f, int g4) {
    const int co0 = ct * TC + i * 32 + 8 * g4 + 4 * h;
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (co0 < a.Cout) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + co0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc[i][f][4 * g4 + j] + bv[j];
        if (a.act == 1) v[j] = fmaxf(v[j], 0.0f);
      }
    }
    return v;
  };
#pragma unroll
  for (int f = 0; f < NPF; ++f) {
    if (pix[f] < 0) continue;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const bf16x4 pb = __builtin_convertvector(value4(i, f, 2 * m), bf16x4);
        const bf16x4 qb = __builtin_convertvector(value4(i, f, 2 * m + 1), bf16x4);
        const u32x2 P = __builtin_bit_cast(u32x2, pb), Q = __builtin_bit_cast(u32x2, qb);
        const auto s0 = __builtin_amdgcn_permlane32_swap(P[0], Q[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(P[1], Q[1], false, false);
        const int co = ct * TC + i * 32 + 16 * m + 8 * h;
        if (co < a.Cout) *reinterpret_cast<u32x4*>(a.y + pix[f] * a.ldy + co) = u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
  }
}

template <int NPF, int NA>
static int db_launch(DbArgs& a, hipStream_t st) {
  const int TP = 256 * NPF, TW = 1 << a.log_tw, TR = TP >> a.log_tw;
  a.n_rt = cdiv((long)a.N * a.H, TR);
  a.n_xt = cdiv(a.W, TW);
  const int grid = cdiv(a.n_rt * a.n_xt, 8) * 8 * a.n_ct;
  hipLaunchKernelGGL((dcn_bf16_kernel<NPF, NA>), dim3(grid), dim3(512), 0, st, a);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// x bf16 NHWC [N,H,W,ldx]; om f32 NHWC [N,H,W,ldo]: channels 0..17 = (dy, dx) per tap, 18..26 = mask logits (the raw
// conv_offset_mask output); wt: [n_ct][9][Cin/64][32*na][64] bf16 rows XOR-swizzled in 16-byte pieces (ops_bf16.py);
// bias f32 [n_ct*32*na] (BN folded); y bf16 NHWC [N,H,W,ldy]; act 1 = ReLU.  Cin % 64 == 0, Cout % 8 == 0.
extern "C" int sgta_dcnv2_fwd_bf16(const void* x, int ldx, const float* om, int ldo, const void* wt, long wt_bytes,
                                   const float* bias, void* y, int ldy, int N, int H, int W, int Cin, int Cout,
                                   int na, int log_tw, int act, hipStream_t st) {
  if (!x || !om || !wt || !bias || !y || N <= 0 || H <= 0 || W <= 0) return SGTA_EBADARG;
  if (Cin % 64 || Cout % 8 || ldx % 8 || ldy % 8 || ((uintptr_t)y & 15) || ((uintptr_t)x & 15) || ldo < 27 || log_tw < 3 ||
      log_tw > 6)
    return SGTA_EBADARG;
  if (wt_bytes <= 0 || wt_bytes > 0x7fffffffL) return SGTA_EBADARG;
  DbArgs a;
  a.x = reinterpret_cast<const __bf16*>(x); a.ldx = ldx; a.om = om; a.ldo = ldo; a.wt = wt; a.wt_bytes = wt_bytes;
  a.bias = bias; a.y = reinterpret_cast<__bf16*>(y); a.ldy = ldy; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.ngroups = Cin / 64; a.n_ct = cdiv(Cout, 32 * na); a.log_tw = log_tw; a.act = act;
  if (na == 2) return db_launch<1, 2>(a, st);
  if (na == 4) return db_launch<1, 4>(a, st);
  return SGTA_EBADARG;
}
