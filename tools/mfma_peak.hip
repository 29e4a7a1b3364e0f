// Micro-benchmark: f32 MFMA 32x32x2 issue rate vs work-group shape (GPU box only; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int threads, int blocks_per_cu) {
  float* out; hipMalloc(&out, 256 * 8 * 1024 * 4 * 4);
  int iters = 2000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, out, 10, 1.f, 2.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * (threads / 64) * iters * 4.0 * NACC * 2.0 * 32 * 32 * 2;
  printf("threads=%4d blocks/CU=%d nacc=%d : %.1f TFLOP/s\n", threads, blocks_per_cu, NACC, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<4>(256, 1); run<4>(256, 2); run<4>(256, 4); run<4>(512, 1); run<4>(1024, 1); run<2>(1024, 1); run<1>(1024, 1); run<1>(256, 4); run<1>(256, 1);
  return 0;
}
