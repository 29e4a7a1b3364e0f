/* Oracle twin in plain C: modulated deformable convolution v2 (3x3, stride 1, pad 1, dil 1, one
 * deformable group) -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py); never linked by the product.
 *
 * Restates the published algorithm of the un-vendored lbin/DCNv2 (branch pytorch_1.9; reference
 * README.md:21-28, call site sgtapose/lib/model/networks/dla.py:545-548): modulated_deformable_im2col
 * followed by a GEMM.  PARITY UNPINNED for this operator (no source, no test in the reference tree);
 * this file exists so the torch restatement in oracle/dcnv2.py has an independent scalar witness.
 *
 *   om    [B,27,H,W]  raw conv_offset_mask output: ch 2k = dy, 2k+1 = dx of tap k, 18+k = mask logit
 *   w     [Co,Ci,3,3], bias [Co],  x [B,Ci,H,W],  y [B,Co,H,W]
 */
#include <math.h>
#include <stdlib.h>

static float bilinear(const float* p, int H, int W, float h, float w) {
  int h_low = (int)floorf(h), w_low = (int)floorf(w);
  int h_high = h_low + 1, w_high = w_low + 1;
  float lh = h - (float)h_low, lw = w - (float)w_low, hh = 1.0f - lh, hw = 1.0f - lw;
  float v1 = (h_low >= 0 && w_low >= 0) ? p[h_low * W + w_low] : 0.0f;
  float v2 = (h_low >= 0 && w_high <= W - 1) ? p[h_low * W + w_high] : 0.0f;
  float v3 = (h_high <= H - 1 && w_low >= 0) ? p[h_high * W + w_low] : 0.0f;
  float v4 = (h_high <= H - 1 && w_high <= W - 1) ? p[h_high * W + w_high] : 0.0f;
  float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

int dcnv2_forward_ref(const float* x, const float* om, const float* w, const float* bias, float* y,
                      int B, int Ci, int H, int W, int Co) {
  const int HW = H * W;
  float* col = (float*)malloc(sizeof(float) * (size_t)Ci * 9);
  if (!col) return 1;
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < H; ++h)
      for (int ww = 0; ww < W; ++ww) {
        const float* o = om + (size_t)b * 27 * HW + h * W + ww;
        for (int k = 0; k < 9; ++k) {
          const int ky = k / 3, kx = k % 3;
          const float dy = o[(size_t)(2 * k) * HW], dx = o[(size_t)(2 * k + 1) * HW];
          const float m = 1.0f / (1.0f + expf(-o[(size_t)(18 + k) * HW]));
          const float py = (float)(h - 1 + ky) + dy, px = (float)(ww - 1 + kx) + dx;
          const int inside = py > -1.0f && px > -1.0f && py < (float)H && px < (float)W;
          for (int c = 0; c < Ci; ++c) {
            float v = 0.0f;
            if (inside) v = bilinear(x + ((size_t)b * Ci + c) * HW, H, W, py, px);
            col[c * 9 + k] = v * m;
          }
        }
        for (int co = 0; co < Co; ++co) {
          double acc = 0.0;                       /* double accumulation: a witness, not a bit-twin */
          const float* wr = w + (size_t)co * Ci * 9;
          for (int i = 0; i < Ci * 9; ++i) acc += (double)wr[i] * (double)col[i];
          y[((size_t)b * Co + co) * HW + h * W + ww] = (float)acc + bias[co];
        }
      }
  free(col);
  return 0;
}
