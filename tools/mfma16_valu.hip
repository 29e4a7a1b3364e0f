#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bt(const float (&x)[6], float (&y)[6]) {
  y[0] = fmaf(4.0f, x[0], fmaf(-5.0f, x[2], x[4]));
  y[5] = fmaf(4.0f, x[1], fmaf(-5.0f, x[3], x[5]));
  const float t1 = fmaf(-4.0f, x[2], x[4]), t2 = fmaf(-4.0f, x[1], x[3]);
  y[1] = t1 + t2; y[2] = t1 - t2;
  const float t3 = x[4] - x[2], t4 = x[3] - x[1];
  y[3] = fmaf(2.0f, t4, t3); y[4] = fmaf(-2.0f, t4, t3);
}
template <int NV>   // NV = number of 6-vector transforms per 36 MFMAs (12 = full F(4,3) input transform)
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float a0) {
  f32x4 acc[36];
  for (int p = 0; p < 36; ++p) acc[p] = f32x4{0, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  float d[36], V[36];
  f32x4 u[9];
  for (int v = 0; v < 9; ++v) u[v] = f32x4{a0, a0 + 1, a0 + 2, a0 + 3};
  for (int j = 0; j < 36; ++j) d[j] = a0 + j + lane;
  for (int it = 0; it < iters; ++it) {
    for (int j = 0; j < 36; ++j) V[j] = d[j];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      float x[6], y[6];
      for (int i = 0; i < 6; ++i) x[i] = V[(q % 6) * 6 + i];
      bt(x, y);
      for (int i = 0; i < 6; ++i) V[(q % 6) * 6 + i] = y[i];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 36; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[p >> 2][p & 3], V[p], acc[p], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    for (int j = 0; j < 36; ++j) d[j] = V[j] * 0.5f;
  }
  float s = 0;
  for (int p = 0; p < 36; ++p) for (int r = 0; r < 4; ++r) s += acc[p][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV>
void run(int bpc) {
  float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
  int iters = 4000, grid = 256 * bpc;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NV>, dim3(grid), dim3(256), 0, 0, out, 10, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NV>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 * iters * 36.0 * 2.0 * 16 * 16 * 4;
  printf("NV=%d (%d VALU per 36 MFMA) blocks/CU=%d : %.1f TFLOP/s\n", NV, NV * 12 + 72, bpc, flops / ms / 1e9);
  hipFree(out);
}
int main() { run<0>(2); run<3>(2); run<6>(2); run<12>(2); run<12>(1); run<24>(2); return 0; }
