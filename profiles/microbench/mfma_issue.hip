// Does the f32 matrix pipe lose throughput when TWO waves of a SIMD feed it?  (profiles/microbench/mfma_issue.hip)
// Each wave issues v_mfma_f32_16x16x4_f32 on CH independent accumulator chains; modes:
//   0: 1 wave / SIMD, all MFMAs            1: 2 waves / SIMD, each half of the MFMAs
//   2: 2 waves / SIMD, one all MFMAs, the other independent v_fma_f32 (4 per MFMA of the partner)
// hipcc --offload-arch=gfx950 -O3 mfma_issue.hip -o mfma_issue && ./mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__device__ __forceinline__ void mfmas(int iters, float a, float b, float *out) {
  f32x4 c[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) c[i] = (f32x4)(0.f);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < CH; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[threadIdx.x + blockIdx.x * blockDim.x] = s;
}
__device__ __forceinline__ void fmas(int iters, float a, float b, float *out) {
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], b, a);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[threadIdx.x + blockIdx.x * blockDim.x] = s;
}
template <int MODE, int CH>
__global__ __launch_bounds__(512) void k(int iters, float a, float b, float *out) {
  __shared__ float big[30 * 1024];  // 120 KB: one block per CU
  big[threadIdx.x] = a;
  const int wave = threadIdx.x >> 6;
  if (MODE == 2 && wave >= 4) fmas(iters * CH / 4, a, b, out);  // 8 * CH MFMAs <-> 32 * CH FMAs per iteration of the partner
  else mfmas<CH>(iters, a, b, out);
}
template <int MODE, int CH>
void run(const char *what) {
  float *out; hipMalloc(&out, 256 * 512 * sizeof(float));
  const int waves = MODE == 0 ? 4 : 8;
  const int total_iters = 20000;                       // per SIMD: total_iters * 8 * CH MFMAs in modes 0/1
  const int iters = MODE == 1 ? total_iters / 2 : total_iters;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, CH>), dim3(256), dim3(64 * waves), 0, 0, iters, 1.0f, 0.5f, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)total_iters * 8 * CH;      // MFMAs per SIMD
  printf("%-58s chains %d: %8.3f ms  -> %6.2f ns per MFMA per SIMD (%.1f clocks at 2.4 GHz), %.1f TFLOP/s\n", what, CH, ms,
         ms * 1e6 / mf, ms * 1e6 / mf * 2.4, mf * 1024 * 2048 / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  run<0, 4>("1 wave/SIMD");
  run<0, 2>("1 wave/SIMD");
  run<0, 1>("1 wave/SIMD (dependent chain)");
  run<1, 4>("2 waves/SIMD, MFMAs split");
  run<1, 2>("2 waves/SIMD, MFMAs split");
  run<1, 1>("2 waves/SIMD, MFMAs split (each a dependent chain)");
  run<2, 4>("2 waves/SIMD, one MFMA, one v_fma (4 per MFMA)");
  return 0;
}
