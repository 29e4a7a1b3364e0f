// Fused multi-head cross-attention core with the learned structure-prior bias (gfx950).
//   out[b,i,h*d:(h+1)*d] = softmax_j( Q[b,i,h]*K[b,j,h] / sqrt(d) + pos_embed[h,i,j] ) . V[b,j,h]
// = the energy / softmax / AV part of MHCA_ein.forward (reference dla.py:874-885); the Q/K/V and
// fc projections are token GEMMs (linear.hip).  The [B,8,n,n] energy tensor of the reference never
// exists: scores live in MFMA accumulators, softmax is online (running max / sum), and P feeds the
// PV product straight from registers.
//
// Mapping (v_mfma_f32_16x16x4_f32, exact f32):
//   S^T tile (16 keys x 16 queries) = Kt-tile . Q^T : the QUERY is on the lane (col = lane&15), a
//   lane holds 4 keys (j = 4*(lane>>4)+r) -> the softmax row statistics need 2 shuffles per tile,
//   and the accumulator registers ARE the B operand of the PV product O^T[d, i] += V^T[d, j] P^T[j, i]
//   (k index = lane>>4) with no data movement.  Head dims 4/8/16 only fill 4/8/16 of the 16 MFMA
//   rows of PV; that padding is inherent to the reference's tiny heads.
// K and V of one (batch, head) are staged once, transposed, in LDS ([d][n_pad], <= 38 KiB) and
// shared by the work-group's 4 waves, each of which walks its own 16-query tiles over all keys.
#include "common.h"

template <int D>
__global__ __launch_bounds__(256) void mhca_kernel(const float* __restrict__ q, int ldq,
                                                   const float* __restrict__ k, int ldk,
                                                   const float* __restrict__ v, int ldv,
                                                   const float* __restrict__ pos,
                                                   const float* __restrict__ post,
                                                   float* __restrict__ out, int ldo, int n, int npad,
                                                   float inv_scale, int nb, int qsplit) {
  extern __shared__ float smem[];
  const int ldt = npad + 4;                        // keeps 16-B alignment of the b128 V reads
  float* Kt = smem;                                // [D][ldt]
  float* Vt = smem + D * ldt;                      // [D][ldt]
  // 1-D grid, SAMPLE fastest: the work-groups that read the same bias tiles (same head, same query range, different
  // samples) are dispatched together -- one per XCD slot in turn, so each XCD's L2 serves B / 8 of them from one fetch (the
  // level-0 bias is 5.6 MB per head: with the sample on the slowest grid axis every work-group streamed it from the
  // Infinity Cache, 1.4 GB per launch)
  const int b = blockIdx.x % nb, h = (blockIdx.x / nb) / qsplit, qblk = (blockIdx.x / nb) % qsplit;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;

  // ---- stage K^T, V^T of (b, h) -----------------------------------------------------------------
  for (int t = tid; t < npad * (D / 4); t += 256) {
    int j = t % npad, c4 = t / npad;
    f32x4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
    if (j < n) {
      kv = *reinterpret_cast<const f32x4*>(k + ((long)b * n + j) * ldk + h * D + c4 * 4);
      vv = *reinterpret_cast<const f32x4*>(v + ((long)b * n + j) * ldv + h * D + c4 * 4);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      Kt[(c4 * 4 + c) * ldt + j] = kv[c];
      Vt[(c4 * 4 + c) * ldt + j] = vv[c];
    }
  }
  __syncthreads();

  const int ntile = npad / 16;
  const float* posh = pos ? pos + (long)h * n * n : nullptr;
  const int vrow = (li < D ? li : D - 1) * ldt;    // MFMA rows >= D are never read back

  for (int it = qblk * 4 + wave; it < ntile; it += qsplit * 4) {
    const int i = it * 16 + li;
    const int ic = i < n ? i : n - 1;
    // Q fragment: B[k = g][col = li] = Q[i][h*D + 4c + g]
    float qf[D / 4];
#pragma unroll
    for (int c = 0; c < D / 4; ++c) qf[c] = q[((long)b * n + ic) * ldq + h * D + c * 4 + g];
    const float* prow = posh ? posh + (long)ic * n : nullptr;
    // tiled bias (sgta_mhca_tiled_fwd_f32): [head][query tile][key tile][16 queries][16 keys], -inf baked in beyond n:
    // one coalesced 16-byte load per lane and key tile instead of four strided 4-byte ones
    const float* ptile = post ? post + (((long)h * ntile + it) * ntile) * 256 + li * 16 + 4 * g : nullptr;

    float m = -INFINITY, lsum = 0.0f;
    f32x4 o = {0, 0, 0, 0};
    float bias[4];
    auto load_bias = [&](int jt, float* dst) {
      if (ptile) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(ptile + (long)jt * 256);
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[r] = t4[r];
        return;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int j = jt * 16 + 4 * g + r;
        dst[r] = (j < n) ? (prow ? prow[j] : 0.0f) : -INFINITY;
      }
    };
    // PF bias tiles in flight per wave (see mhca4_kernel: a one-tile prefetch leaves the loop waiting on L2 / Infinity Cache)
    constexpr int PF = 8;
    float bq[PF][4];
#pragma unroll
    for (int u = 0; u < PF; ++u) load_bias(u < ntile ? u : ntile - 1, bq[u]);
    for (int jt0 = 0; jt0 < ntile; jt0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
      const int jt = jt0 + u;
      if (jt >= ntile) break;
#pragma unroll
      for (int r = 0; r < 4; ++r) bias[r] = bq[u][r];
      load_bias(jt + PF < ntile ? jt + PF : ntile - 1, bq[u]);
      f32x4 s = {0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < D / 4; ++c) {
        float a = Kt[(c * 4 + g) * ldt + jt * 16 + li];     // A[row = key li][k = g]
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qf[c], s, 0, 0, 0);
      }
      float e[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r] = fmaf(s[r], inv_scale, bias[r]);
      float tmax = fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3]));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float mn = fmaxf(m, tmax);
      const float alpha = __expf(m - mn);
      m = mn;
      float p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r] = __expf(e[r] - mn);
      lsum = fmaf(lsum, alpha, (p[0] + p[1]) + (p[2] + p[3]));
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] *= alpha;
      const f32x4 va = *reinterpret_cast<const f32x4*>(&Vt[vrow + jt * 16 + 4 * g]);  // V[j=4g+r][dd=li]
#pragma unroll
      for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_16x16x4f32(va[r], p[r], o, 0, 0, 0);
      }
    }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    // O^T rows dd = 4g + r live in lanes with 4g < D
    if (i < n && 4 * g < D) {
      const float inv = 1.0f / lsum;
      f32x4 r4 = {o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
      *reinterpret_cast<f32x4*>(out + ((long)b * n + i) * ldo + h * D + 4 * g) = r4;
    }
  }
}

// ---- head_dim 4 (level 0: 1183 tokens per sample, the longest attention of the path) --------------------------------
// With d = 4 the PV product fills 4 of the 16 MFMA rows, and a common running maximum across the four key groups of a
// query costs two cross-lane exchanges per key tile.  Here every lane keeps its OWN online-softmax state (m, l, o[4])
// over the keys it sees (j = 4g + r of every tile), PV is 16 plain FMAs on V rows broadcast from LDS, and the four
// partial states of a query are merged once per query tile (log-sum-exp merge).  Same mathematics, different summation
// order.  S^T = K Q^T stays on the MFMA.
__global__ __launch_bounds__(256) void mhca4_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                    int ldk, const float* __restrict__ v, int ldv,
                                                    const float* __restrict__ post, float* __restrict__ out, int ldo,
                                                    int n, int npad, float inv_scale, int nb, int qsplit) {
  constexpr int D = 4;
  extern __shared__ float smem[];
  const int ldt = npad + 4;
  float* Kt = smem;                                // [4][ldt]   K transposed (MFMA A operand rows)
  float* Vr = smem + D * ldt;                      // [npad][4]  V rows (one 16-byte broadcast read per key)
  // 1-D grid, SAMPLE fastest: the work-groups that read the same bias tiles (same head, same query range, different
  // samples) are dispatched together -- one per XCD slot in turn, so each XCD's L2 serves B / 8 of them from one fetch (the
  // level-0 bias is 5.6 MB per head: with the sample on the slowest grid axis every work-group streamed it from the
  // Infinity Cache, 1.4 GB per launch)
  const int b = blockIdx.x % nb, h = (blockIdx.x / nb) / qsplit, qblk = (blockIdx.x / nb) % qsplit;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  for (int j = tid; j < npad; j += 256) {
    f32x4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
    if (j < n) {
      kv = *reinterpret_cast<const f32x4*>(k + ((long)b * n + j) * ldk + h * D);
      vv = *reinterpret_cast<const f32x4*>(v + ((long)b * n + j) * ldv + h * D);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) Kt[c * ldt + j] = kv[c];
    *reinterpret_cast<f32x4*>(Vr + j * 4) = vv;
  }
  __syncthreads();
  const int ntile = npad / 16;
  for (int it = qblk * 4 + wave; it < ntile; it += qsplit * 4) {
    const int i = it * 16 + li;
    const int ic = i < n ? i : n - 1;
    const float qf = q[((long)b * n + ic) * ldq + h * D + g];          // B[k = g][col = li]
    const float* ptile = post + (((long)h * ntile + it) * ntile) * 256 + li * 16 + 4 * g;
    float m = -INFINITY, lsum = 0.0f, o[4] = {0.f, 0.f, 0.f, 0.f};
    // the bias tiles come from L2 / the Infinity Cache (5.6 MB per head at level 0): a tile of scores is ~200 cycles of work,
    // a miss a few thousand -- keep PF tiles of bias in flight per wave (a one-tile prefetch left the loop waiting on them)
    constexpr int PF = 8;
    f32x4 bq[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) bq[u] = *reinterpret_cast<const f32x4*>(ptile + (long)(u < ntile ? u : ntile - 1) * 256);
    for (int jt0 = 0; jt0 < ntile; jt0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int jt = jt0 + u;
        if (jt >= ntile) break;
        const f32x4 bias = bq[u];
        const int jn = jt + PF < ntile ? jt + PF : ntile - 1;
        bq[u] = *reinterpret_cast<const f32x4*>(ptile + (long)jn * 256);
        const float a = Kt[g * ldt + jt * 16 + li];                      // A[row = key li][k = g]
        f32x4 s = {0, 0, 0, 0};
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qf, s, 0, 0, 0);     // lane (li = query, g): keys jt*16 + 4g + r
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = fmaf(s[r], inv_scale, bias[r]);
        const float mn = fmaxf(fmaxf(m, fmaxf(e[0], e[1])), fmaxf(e[2], e[3]));
        // all four keys masked and nothing seen yet: keep the state empty (exp(-inf - -inf) would be NaN)
        const float msafe = mn == -INFINITY ? 0.0f : mn;
        const float alpha = __expf(m - msafe);
        m = mn;
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = __expf(e[r] - msafe);
        lsum = fmaf(lsum, alpha, (p[0] + p[1]) + (p[2] + p[3]));
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) o[dd] *= alpha;
        const float* vp = Vr + (jt * 16 + 4 * g) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 vv = *reinterpret_cast<const f32x4*>(vp + r * 4);  // V[key 4g + r][0..3], same address across li
#pragma unroll
          for (int dd = 0; dd < 4; ++dd) o[dd] = fmaf(p[r], vv[dd], o[dd]);
        }
      }
    }
    // merge the four key groups of every query
    float mall = fmaxf(m, __shfl_xor(m, 16));
    mall = fmaxf(mall, __shfl_xor(mall, 32));
    const float sc = m == -INFINITY ? 0.0f : __expf(m - mall);
    lsum *= sc;
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
      o[dd] *= sc;
      o[dd] += __shfl_xor(o[dd], 16);
      o[dd] += __shfl_xor(o[dd], 32);
    }
    if (i < n && g == 0) {
      const float inv = 1.0f / lsum;
      *reinterpret_cast<f32x4*>(out + ((long)b * n + i) * ldo + h * D) = f32x4{o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
    }
  }
}

static int mhca_launch(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                       const float* pos_embed, const float* pos_tiles, float* out, int ldo, int B, int n, int n_heads,
                       int head_dim, hipStream_t st) {
  if (!q || !k || !v || !out || B <= 0 || n <= 0 || n_heads <= 0) return SGTA_EBADARG;
  if ((ldq | ldk | ldv | ldo) & 3) return SGTA_EBADARG;
  if (pos_tiles && ((uintptr_t)pos_tiles & 15)) return SGTA_EBADARG;
  const int npad = (n + 15) / 16 * 16;
  const int ntile = npad / 16;
  // work-groups along the query axis: one 16-query tile per wave where possible (level 0: 74 tiles -> 19 work-groups per
  // (sample, head); with at most 8 the waves owned 3 / 2 / 2 / 2 tiles and the work-group ran as long as its slowest wave)
  int qsplit = (ntile + 3) / 4;
  if (qsplit > 32) qsplit = 32;
  dim3 grid((unsigned)qsplit * n_heads * B);
  const float inv = 1.0f / sqrtf((float)head_dim);
  size_t lds = (size_t)2 * head_dim * (npad + 4) * sizeof(float);
  if (head_dim == 4 && pos_tiles && !(ldk & 3) && !(ldv & 3) && !(ldo & 3)) {
    hipLaunchKernelGGL(mhca4_kernel, grid, dim3(256), lds, st, q, ldq, k, ldk, v, ldv, pos_tiles, out, ldo, n, npad, inv, B, qsplit);
    SGTA_LAUNCH_CHECK();
    return SGTA_OK;
  }
  switch (head_dim) {
    case 4: hipLaunchKernelGGL(mhca_kernel<4>, grid, dim3(256), lds, st, q, ldq, k, ldk, v, ldv, pos_embed, pos_tiles, out, ldo, n, npad, inv, B, qsplit); break;
    case 8: hipLaunchKernelGGL(mhca_kernel<8>, grid, dim3(256), lds, st, q, ldq, k, ldk, v, ldv, pos_embed, pos_tiles, out, ldo, n, npad, inv, B, qsplit); break;
    case 16: hipLaunchKernelGGL(mhca_kernel<16>, grid, dim3(256), lds, st, q, ldq, k, ldk, v, ldv, pos_embed, pos_tiles, out, ldo, n, npad, inv, B, qsplit); break;
    default: return SGTA_EBADARG;
  }
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

extern "C" int sgta_mhca_fwd_f32(const float* q, int ldq, const float* k, int ldk, const float* v,
                                 int ldv, const float* pos_embed, float* out, int ldo, int B, int n,
                                 int n_heads, int head_dim, hipStream_t st) {
  return mhca_launch(q, ldq, k, ldk, v, ldv, pos_embed, nullptr, out, ldo, B, n, n_heads, head_dim, st);
}

// Same operator with the bias re-packed once by the caller into 16 x 16 tiles:
//   pos_tiles[((h * T + it) * T + jt) * 256 + li * 16 + c] = pos_embed[h][it*16 + li][jt*16 + c]   (T = ceil(n / 16)),
//   -inf where jt*16 + c >= n (the key padding mask), anything finite where it*16 + li >= n.
extern "C" int sgta_mhca_tiled_fwd_f32(const float* q, int ldq, const float* k, int ldk, const float* v,
                                       int ldv, const float* pos_tiles, float* out, int ldo, int B, int n,
                                       int n_heads, int head_dim, hipStream_t st) {
  if (!pos_tiles) return SGTA_EBADARG;
  return mhca_launch(q, ldq, k, ldk, v, ldv, nullptr, pos_tiles, out, ldo, B, n, n_heads, head_dim, st);
}
