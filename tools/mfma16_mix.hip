// Micro-benchmark: f32 MFMA 16x16x4 issue rate with the operand traffic of conv3x3_wino.hip mixed in
// (GPU box only; not part of the library).  hipcc --offload-arch=gfx950 -O3 tools/mfma16_mix.hip -o /tmp/mix && /tmp/mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // bit0: VALU transform, bit1: LDS reads of U (8 b128), bit2: LDS reads of window (16 b32)
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float a0) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = a0 + i;
  __syncthreads();
  f32x4 acc[16][2];
  for (int p = 0; p < 16; ++p) for (int i = 0; i < 2; ++i) acc[p][i] = f32x4{0, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  const float* ub = lds + (lane & 15) * 20 + (lane >> 4) * 640;
  const float* pb = lds + 8192 + (lane & 15) + (lane >> 4) * 400;
  f32x4 u[2][4];
  float d[16], V[16];
  for (int i = 0; i < 2; ++i) for (int v = 0; v < 4; ++v) u[i][v] = f32x4{a0, a0 + 1, a0 + 2, a0 + 3};
  for (int j = 0; j < 16; ++j) d[j] = a0 + j + lane;
  for (int it = 0; it < iters; ++it) {
    const int o = (it & 3) * 2560;
    if (MODE & 2)
      for (int i = 0; i < 2; ++i) for (int v = 0; v < 4; ++v) u[i][v] = *reinterpret_cast<const f32x4*>(ub + o + i * 320 + v * 4);
    if (MODE & 4)
      for (int r = 0; r < 4; ++r) for (int j = 0; j < 4; ++j) d[r * 4 + j] = pb[(it & 3) * 1600 + r * 40 + (j & 1) * 17 + (j >> 1)];
    if (MODE & 1) {
      float tr[16];
      for (int j = 0; j < 4; ++j) {
        tr[0 + j] = d[0 + j] - d[8 + j]; tr[4 + j] = d[4 + j] + d[8 + j];
        tr[8 + j] = d[8 + j] - d[4 + j]; tr[12 + j] = d[4 + j] - d[12 + j];
      }
      for (int x = 0; x < 4; ++x) {
        V[x * 4 + 0] = tr[x * 4 + 0] - tr[x * 4 + 2]; V[x * 4 + 1] = tr[x * 4 + 1] + tr[x * 4 + 2];
        V[x * 4 + 2] = tr[x * 4 + 2] - tr[x * 4 + 1]; V[x * 4 + 3] = tr[x * 4 + 1] - tr[x * 4 + 3];
      }
    } else {
      for (int j = 0; j < 16; ++j) V[j] = d[j];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[p][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[i][p >> 2][p & 3], V[p], acc[p][i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0;
  for (int p = 0; p < 16; ++p) for (int i = 0; i < 2; ++i) for (int r = 0; r < 4; ++r) s += acc[p][i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(int blocks_per_cu) {
  float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
  int iters = 4000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 10, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 * iters * 32.0 * 2.0 * 16 * 16 * 4;
  printf("mode=%d blocks/CU=%d : %.1f TFLOP/s (%.3f ms)\n", MODE, blocks_per_cu, flops / ms / 1e9, ms);
  hipFree(out);
}
int main() {
  run<0>(1); run<0>(2); run<1>(2); run<2>(2); run<4>(2); run<6>(2); run<7>(2); run<7>(1);
  return 0;
}
