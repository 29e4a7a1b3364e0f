// Which XCD does work-group b of a 1-D grid run on?  (GPU box only; validates the blockIdx % 8 assumption of
// xcd_remap / the persistent kernels' logical ids.)  hipcc --offload-arch=gfx950 -O3 tools/xcc_map.hip -o /tmp/xcc && /tmp/xcc
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256, 2) void k(int* out, int spin) {
  __shared__ float pad[16000];                        // 64 KB: two work-groups per CU, like the Winograd kernels
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  float acc = 0;
  for (int i = 0; i < spin; ++i) acc += __sinf(i * 0.001f + threadIdx.x);
  pad[threadIdx.x] = acc;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = x & 0xf; out[2 * blockIdx.x + 1] = (hw >> 8) & 0xf; /* cu id */ }
}
int main() {
  const int G = 512;
  int* d; hipMalloc(&d, G * 8);
  hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, d, 20000);
  int h[G * 2]; hipMemcpy(h, d, G * 8, hipMemcpyDeviceToHost);
  int ok = 0;
  for (int b = 0; b < G; ++b) ok += (h[2 * b] == (b & 7));
  printf("blocks on XCD == blockIdx %% 8: %d of %d\n", ok, G);
  printf("first 24 (xcc,cu): ");
  for (int b = 0; b < 24; ++b) printf("(%d,%d) ", h[2 * b], h[2 * b + 1]);
  printf("\nblocks 0,8,16,...,120 cu ids: ");
  for (int b = 0; b < 128; b += 8) printf("%d ", h[2 * b + 1]);
  printf("\n");
  return 0;
}
