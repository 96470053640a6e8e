// Which SIMD does wave w of a 512-thread (-DWAVES=12: 768-thread) block land on?  (profiles/microbench/wave_simd.hip)
// hipcc --offload-arch=gfx950 -O2 wave_simd.hip -o wave_simd && ./wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef WAVES
#define WAVES 8
#endif
__global__ __launch_bounds__(WAVES * 64) void k(int *out) {
  __shared__ float big[36 * 1024];  // 144 KB: one block per CU, like the MLP kernels
  big[threadIdx.x] = 0.f;
  const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID, all 32 bits
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * WAVES + (threadIdx.x >> 6)] = (int)id;
}
int main() {
  int *d; hipMalloc(&d, 256 * WAVES * sizeof(int));
  hipLaunchKernelGGL(k, dim3(256), dim3(WAVES * 64), 0, 0, d);
  static int h[256 * WAVES]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 4; ++b) {
    printf("block %d:", b);
    for (int w = 0; w < WAVES; ++w) printf("  w%d: simd %d", w, (h[b * WAVES + w] >> 4) & 3);
    printf("\n");
  }
  int same = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w + 4 < WAVES; ++w) same += ((h[b * WAVES + w] >> 4) & 3) == ((h[b * WAVES + w + 4] >> 4) & 3);
  printf("waves (w, w+4) on the same SIMD: %d of %d\n", same, 256 * (WAVES - 4));
  return 0;
}
