// Fused position-wise feed-forward block of the temporal encoder layer (gfx950, f32 MFMA 16x16x4):
//   y = LayerNorm( x + W2 . relu(W1 . x + b1) + b2 )         (reference dla.py:721-732, :737-738; eps 1e-5)
// for token-major x [M, C], C in {16, 32, 64}, hidden width F (1024 in the reference, F % 16 == 0).
// The two nn.Linear launches this replaces wrote and re-read the [M, F] hidden activations (155 MB per layer at
// level 0: 37856 tokens) and walked K = F behind one barrier per 16-deep chunk.  Here the hidden activations never
// leave the registers: everything is computed TRANSPOSED so that the accumulator of the first product is, register
// for register, the B operand of the second --
//   H^T[16 hidden x 16 tokens] = W1_tile[16 x C] . X^T[C x 16]      lane (n = lane & 15, g = lane >> 4) holds hidden 4g + r
//   Y^T[C x 16 tokens]        += W2[:, tile]     . H^T              k-slot g of step r = hidden 4g + r: that same register
// -- no LDS, no barrier, no shuffle; a wave owns 16 * TT tokens and streams the weights (identical for every wave:
// L1 / L2 hits) two hidden tiles ahead of the MFMAs.  The contraction index of the first product is permuted
// (k-slot g, step s <-> channel g * C/4 + s) so that every operand fragment is a contiguous 16-byte load.
#include "common.h"

struct FfnArgs {
  const float* x; int ldx;
  const float* w1; const float* b1;      // [F, C], [F]
  const float* w2; const float* b2;      // [C, F], [C]
  const float* gamma; const float* beta; // [C]
  float* y; int ldy;
  int M, F;
};

template <int C, int TT>
__global__ __launch_bounds__(64) void ffn_ln_kernel(FfnArgs a) {
  constexpr int S = C / 4, Q = C / 16;               // k-steps of the first product / 16-byte quads per fragment = row tiles of Y^T
  const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
  const long tok0 = ((long)blockIdx.x * TT) * 16;
  // X^T fragments: lane (n, g) holds x[token n][g*S .. g*S + S - 1] of each of its TT token tiles
  f32x4 xb[TT][Q];
  bool tok_ok[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const long tok = tok0 + t * 16 + n;
    tok_ok[t] = tok < a.M;
    const float* xp = a.x + (tok_ok[t] ? tok : 0) * a.ldx + g * S;
#pragma unroll
    for (int q = 0; q < Q; ++q) xb[t][q] = tok_ok[t] ? *reinterpret_cast<const f32x4*>(xp + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 acc2[TT][Q];
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int q = 0; q < Q; ++q) acc2[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // weight fragments of one 16-wide hidden tile h0: A1 = W1[h0 + n][g*S ..], bias = b1[h0 + 4g ..], A2[mt] = W2[mt*16 + n][h0 + 4g ..]
  struct Frag { f32x4 a1[Q], bs, a2[Q]; };
  auto load = [&](int h0, Frag& f) {
    const float* p1 = a.w1 + (long)(h0 + n) * C + g * S;
#pragma unroll
    for (int q = 0; q < Q; ++q) f.a1[q] = *reinterpret_cast<const f32x4*>(p1 + 4 * q);
    f.bs = *reinterpret_cast<const f32x4*>(a.b1 + h0 + 4 * g);
#pragma unroll
    for (int q = 0; q < Q; ++q) f.a2[q] = *reinterpret_cast<const f32x4*>(a.w2 + (long)(q * 16 + n) * a.F + h0 + 4 * g);
  };
  auto compute = [&](const Frag& f) {
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      f32x4 h = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < S; ++s) h = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a1[s >> 2][s & 3], xb[t][s >> 2][s & 3], h, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = fmaxf(h[r] + f.bs[r], 0.0f);
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc2[t][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2[q][r], h[r], acc2[t][q], 0, 0, 0);
    }
  };
  const int ntiles = a.F / 16;
  Frag f0, f1;
  load(0, f0);
  if (ntiles > 1) load(16, f1);
  for (int ht = 0; ht < ntiles; ht += 2) {
    compute(f0);
    if (ht + 2 < ntiles) load((ht + 2) * 16, f0);
    if (ht + 1 < ntiles) {
      compute(f1);
      if (ht + 3 < ntiles) load((ht + 3) * 16, f1);
    }
  }

  // ---- epilogue: lane (n, g) holds features q*16 + 4g + r of token n: + b2 + x, LayerNorm over the C features ------------
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const long tok = tok0 + t * 16 + n;
    const float* xr = a.x + (tok_ok[t] ? tok : 0) * a.ldx;
    float v[Q][4];
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const f32x4 b2 = *reinterpret_cast<const f32x4*>(a.b2 + q * 16 + 4 * g);
      const f32x4 xr4 = *reinterpret_cast<const f32x4*>(xr + q * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[q][r] = (acc2[t][q][r] + b2[r]) + xr4[r];
        s += v[q][r];
      }
    }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s / (float)C;
    float qs = 0.0f;
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[q][r] -= mean;
        qs += v[q][r] * v[q][r];
      }
    qs += __shfl_xor(qs, 16);
    qs += __shfl_xor(qs, 32);
    const float rstd = 1.0f / sqrtf(qs / (float)C + 1e-5f);
    if (!tok_ok[t]) continue;
    float* yr = a.y + tok * a.ldy;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + q * 16 + 4 * g);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(a.beta + q * 16 + 4 * g);
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = v[q][r] * rstd * gm[r] + bt[r];
      *reinterpret_cast<f32x4*>(yr + q * 16 + 4 * g) = o;
    }
  }
}

template <int C>
static int launch_ffn(const FfnArgs& a, hipStream_t st) {
  // tokens per wave: as many 16-token tiles (weight fragments are reused across them) as still leave >= 512 waves
  const long t16 = (a.M + 15) / 16;
  if (t16 / 4 >= 512) hipLaunchKernelGGL((ffn_ln_kernel<C, 4>), dim3((unsigned)((t16 + 3) / 4)), dim3(64), 0, st, a);
  else if (t16 / 2 >= 512) hipLaunchKernelGGL((ffn_ln_kernel<C, 2>), dim3((unsigned)((t16 + 1) / 2)), dim3(64), 0, st, a);
  else hipLaunchKernelGGL((ffn_ln_kernel<C, 1>), dim3((unsigned)t16), dim3(64), 0, st, a);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

extern "C" int sgta_ffn_ln_f32(const float* x, int ldx, const float* w1, const float* b1, const float* w2,
                               const float* b2, const float* ln_gamma, const float* ln_beta, float* y, int ldy,
                               int M, int C, int F, hipStream_t st) {
  if (!x || !w1 || !b1 || !w2 || !b2 || !ln_gamma || !ln_beta || !y || M <= 0 || F <= 0) return SGTA_EBADARG;
  if ((C != 16 && C != 32 && C != 64) || F % 16 || ldx % 4 || ldy % 4 || ldx < C || ldy < C) return SGTA_EBADARG;
  const uintptr_t al = (uintptr_t)x | (uintptr_t)w1 | (uintptr_t)b1 | (uintptr_t)w2 | (uintptr_t)b2 | (uintptr_t)ln_gamma |
                       (uintptr_t)ln_beta | (uintptr_t)y;
  if (al & 15) return SGTA_EBADARG;                  // every fragment is a 16-byte load / store
  FfnArgs a{x, ldx, w1, b1, w2, b2, ln_gamma, ln_beta, y, ldy, M, F};
  if (C == 16) return launch_ffn<16>(a, st);
  if (C == 32) return launch_ffn<32>(a, st);
  return launch_ffn<64>(a, st);
}
