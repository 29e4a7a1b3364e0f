// Token-major dense layers of the temporal transformer (gfx950, f32 MFMA 32x32x2):
//   Y[M,N] = epilogue( X[M,K] . W[N,K]^T + bias )
// covering nn.Linear of MHCA_ein (w_q/w_k/w_v, fc: reference dla.py:860-864, :870-872, :886), the FFN
// (linear1/linear2, dla.py:721-724, :728-732) and cat_layer's two Linears (dla.py:1499-1502).
// Epilogues: +bias, ReLU, +residual, and (when one 64-wide tile holds the whole row) the post-norm
// LayerNorm of dla.py:737-738 / :730-731 (eps 1e-5, biased variance), fused so the pre-norm
// activations never go to HBM.
// Layout: rows = tokens on the MFMA row index, output features on the lane (col = lane&31) so each
// accumulator register stores 32 consecutive features of a row: coalesced row-major stores.
#include "common.h"

struct LinArgs {
  const float* x; int ldx;
  const float* w;                // [N, K] row-major (torch Linear)
  const float* bias;             // [N] or nullptr
  const float* res; int ldr;     // residual [M, N] or nullptr
  const float* gamma; const float* beta;   // LayerNorm affine (LN variant)
  float* y; int ldy;
  int M, N, K, relu;
  float* ws; int cps;           // split-K: partial sums [gridDim.z][M][N], BK-chunks per split (0 = no split)
};

template <bool LN>
__global__ __launch_bounds__(256) void linear_kernel(LinArgs a) {
  constexpr int BM = 128, BN = 64, BK = 16, LDA = BM + 2, LDB = BN + 2;
  __shared__ float Xs[2][BK][LDA];
  __shared__ float Ws[2][BK][LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  f32x4 xr[2], wr;
  auto load = [&](int kc) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int idx = tid + i * 256, row = idx >> 2, kq = idx & 3;
      int m = m0 + row;
      xr[i] = (m < a.M) ? *reinterpret_cast<const f32x4*>(a.x + (long)m * a.ldx + kc * BK + kq * 4)
                        : f32x4{0, 0, 0, 0};
    }
    int row = tid >> 2, kq = tid & 3, n = n0 + row;
    wr = (n < a.N) ? *reinterpret_cast<const f32x4*>(a.w + (long)n * a.K + kc * BK + kq * 4)
                   : f32x4{0, 0, 0, 0};
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int idx = tid + i * 256, row = idx >> 2, kq = idx & 3;
#pragma unroll
      for (int j = 0; j < 4; ++j) Xs[buf][kq * 4 + j][row] = xr[i][j];
    }
    int row = tid >> 2, kq = tid & 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) Ws[buf][kq * 4 + j][row] = wr[j];
  };

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

  // split-K (token counts of a few hundred rows leave the 128 x 64 tiles on a handful of CUs, each walking the
  // whole K behind one global-load latency per 16-deep chunk): blockIdx.z owns chunks [kc0, kc1)
  const int nall = a.K / BK;
  const int kc0 = a.cps ? blockIdx.z * a.cps : 0;
  const int kc1 = a.cps ? (kc0 + a.cps < nall ? kc0 + a.cps : nall) : nall;
  load(kc0);
  store(kc0 & 1);
  __syncthreads();
  for (int kc = kc0; kc < kc1; ++kc) {
    const int buf = kc & 1;
    const bool more = kc + 1 < kc1;
    if (more) load(kc + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      const int k = ks * 2 + lh;
      float av = Xs[buf][k][wave * 32 + l31];
      float b0 = Ws[buf][k][l31], b1 = Ws[buf][k][32 + l31];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[1], 0, 0, 0);
    }
    if (more) store(buf ^ 1);
    __syncthreads();
  }

  // epilogue: row m = m0 + wave*32 + (r&3) + 8*(r>>2) + 4*lh ; col n = n0 + j*32 + l31
  if (a.cps) {                                       // raw partial sums; linear_splitk_reduce_kernel finishes
    float* wsz = a.ws + (long)blockIdx.z * a.M * a.N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= a.M) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + j * 32 + l31;
        if (n < a.N) wsz[(long)m * a.N + n] = acc[j][r];
      }
    }
    return;
  }
  float bj[2], gj[2], tj[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int n = n0 + j * 32 + l31;
    bool ok = n < a.N;
    bj[j] = (a.bias && ok) ? a.bias[n] : 0.0f;
    if (LN) { gj[j] = ok ? a.gamma[n] : 0.0f; tj[j] = ok ? a.beta[n] : 0.0f; }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    const bool mok = m < a.M;
    float v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int n = n0 + j * 32 + l31;
      float t = acc[j][r] + bj[j];
      if (a.relu) t = fmaxf(t, 0.0f);
      if (a.res && mok && n < a.N) t += a.res[(long)m * a.ldr + n];
      v[j] = (n < a.N) ? t : 0.0f;
    }
    if (LN) {
      float s = v[0] + v[1];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
      const float mean = s / (float)a.N;
      float d0 = (n0 + l31 < a.N) ? v[0] - mean : 0.0f;
      float d1 = (n0 + 32 + l31 < a.N) ? v[1] - mean : 0.0f;
      float q = d0 * d0 + d1 * d1;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
      const float rstd = 1.0f / sqrtf(q / (float)a.N + 1e-5f);
      v[0] = d0 * rstd * gj[0] + tj[0];
      v[1] = d1 * rstd * gj[1] + tj[1];
    }
    if (mok) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int n = n0 + j * 32 + l31;
        if (n < a.N) a.y[(long)m * a.ldy + n] = v[j];
      }
    }
  }
}

// y = relu?(sum_z ws[z] + bias) + res, summed in split order (deterministic); one thread = 4 features of a row
__global__ __launch_bounds__(256) void linear_splitk_reduce_kernel(const float* __restrict__ ws, int S,
                                                                   const float* __restrict__ bias,
                                                                   const float* __restrict__ res, int ldr,
                                                                   float* __restrict__ y, int ldy, int M, int N,
                                                                   int relu) {
  const int nq = N >> 2;
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= (long)M * nq) return;
  const int m = (int)(i / nq), n = (int)(i - (long)m * nq) * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(ws + (long)m * N + n);
  for (int z = 1; z < S; ++z) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(ws + ((long)z * M + m) * N + n);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += t[q];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float t = v[q] + (bias ? bias[n + q] : 0.0f);
    if (relu) t = fmaxf(t, 0.0f);
    if (res) t += res[(long)m * ldr + n + q];
    y[(long)m * ldy + n + q] = t;
  }
}

extern "C" int sgta_linear_ws_f32(const float* x, int ldx, const float* w, const float* bias, const float* res,
                                  int ldr, const float* ln_gamma, const float* ln_beta, float* y, int ldy, int M,
                                  int N, int K, int relu, float* ws, long ws_floats, hipStream_t st);

extern "C" int sgta_linear_f32(const float* x, int ldx, const float* w, const float* bias,
                               const float* res, int ldr, const float* ln_gamma, const float* ln_beta,
                               float* y, int ldy, int M, int N, int K, int relu, hipStream_t st) {
  if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0 || (K % 16) || (ldx % 4)) return SGTA_EBADARG;
  if ((ln_gamma != nullptr) != (ln_beta != nullptr)) return SGTA_EBADARG;
  if (ln_gamma && N > 64) return SGTA_EBADARG;     // fused LayerNorm needs the whole row in one tile
  return sgta_linear_ws_f32(x, ldx, w, bias, res, ldr, ln_gamma, ln_beta, y, ldy, M, N, K, relu, nullptr, 0, st);
}

// Same operator with a caller-owned workspace of ws_floats floats: when the 128 x 64 tiles alone would leave most
// CUs idle (few hundred tokens) and the workspace is large enough, K is split over blockIdx.z and a second kernel
// sums the partials in split order and applies the epilogue.  No LayerNorm variant (its rows fit one tile: large M).
extern "C" int sgta_linear_ws_f32(const float* x, int ldx, const float* w, const float* bias, const float* res,
                                  int ldr, const float* ln_gamma, const float* ln_beta, float* y, int ldy, int M,
                                  int N, int K, int relu, float* ws, long ws_floats, hipStream_t st) {
  if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0 || (K % 16) || (ldx % 4)) return SGTA_EBADARG;
  if ((ln_gamma != nullptr) != (ln_beta != nullptr)) return SGTA_EBADARG;
  if (ln_gamma && N > 64) return SGTA_EBADARG;     // fused LayerNorm needs the whole row in one tile
  LinArgs a{x, ldx, w, bias, res, ldr, ln_gamma, ln_beta, y, ldy, M, N, K, relu, nullptr, 0};
  dim3 grid((M + 127) / 128, (N + 63) / 64);
  const int tiles = grid.x * grid.y, nchunks = K / 16;
  int S = 1;
  if (ws && !ln_gamma && N % 4 == 0 && ldy % 4 == 0 && tiles < 128) {
    while (S < 32 && tiles * S < 256 && nchunks / (S * 2) >= 4) S *= 2;     // >= 4 chunks per split
    while (S > 1 && (long)S * M * N > ws_floats) S /= 2;
  }
  if (S > 1) {
    a.ws = ws;
    a.cps = (nchunks + S - 1) / S;
    grid.z = (nchunks + a.cps - 1) / a.cps;
    hipLaunchKernelGGL(linear_kernel<false>, grid, dim3(256), 0, st, a);
    const long nthreads = (long)M * (N / 4);
    hipLaunchKernelGGL(linear_splitk_reduce_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, (int)grid.z, bias, res, ldr, y, ldy, M, N, relu);
  } else if (ln_gamma) {
    hipLaunchKernelGGL(linear_kernel<true>, grid, dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(linear_kernel<false>, grid, dim3(256), 0, st, a);
  }
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}
