// membench_stepk.hip -- traffic-only ceiling of pds_step_k (K open-loop steps per launch): per env-step one 16-byte
// action read and the D-float observation row + reward, cost and two flag bytes written (non-temporal, the row through
// the wave's LDS tile), the state read once and written once per launch.  No physics: what the memory system gives a
// WRITE-dominated stream of this launch shape (256-thread blocks, one 64-env tile per wave, 3 blocks per CU).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/membench_stepk profiles/microbench/membench_stepk.hip && /tmp/membench_stepk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4nt(float4 *p, float4 v) { v4f t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(p)); }
__device__ __forceinline__ float4 ld4nt(const float4 *p) { v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p)); return make_float4(t.x, t.y, t.z, t.w); }
struct Args { const float4 *act; float4 *st[5]; float *obs, *rew, *cost; unsigned char *term, *trunc; long long n; int K; };
template <int D>
__global__ __launch_bounds__(256, 3) void stepk(Args a) {
  __shared__ __attribute__((aligned(16))) float tile_all[4 * 64 * D];
  __shared__ float pad[1200];  // (the product kernel's queue + scratch: keeps the block at 3 per CU)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *tile = tile_all + wave * 64 * D;
  long long blk = blockIdx.x; const long long nb = gridDim.x, per = nb / 8;
  if (per * 8 == nb) blk = (blk % 8) * per + blk / 8;
  const long long t = blk * 4 + wave;
  if (t * 64 >= a.n) return;
  const long long i = t * 64 + lane;
  float4 s[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) s[j] = a.st[j][i];
  float4 u = ld4nt(a.act + i);
  if (threadIdx.x == 9999) pad[lane] = 0.f;
  for (int k = 0; k < a.K; ++k) {
    float4 un = u;
    if (k + 1 < a.K) un = ld4nt(a.act + (long long)(k + 1) * a.n + i);
    float *row = tile + lane * D;
#pragma unroll
    for (int j = 0; j < D; ++j) row[j] = s[j % 5].x + u.x + (float)j;
    s[0].x += u.y; s[1].y += u.z;
    const long long o1 = (long long)k * a.n;
    __builtin_nontemporal_store(s[0].x, a.rew + o1 + i);
    __builtin_nontemporal_store(s[1].y, a.cost + o1 + i);
    a.term[o1 + i] = (unsigned char)(k & 1);
    a.trunc[o1 + i] = 0;
    __builtin_amdgcn_wave_barrier();
    float4 *dst = reinterpret_cast<float4 *>(a.obs + (o1 + t * 64) * D);
    const float4 *src4 = reinterpret_cast<const float4 *>(tile);
#pragma unroll
    for (int it = 0; it < (16 * D + 63) / 64; ++it) {
      const int idx = it * 64 + lane;
      if (idx < 16 * D) st4nt(dst + idx, src4[idx]);
    }
    __builtin_amdgcn_wave_barrier();
    u = un;
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) a.st[j][i] = s[j];
}
int main() {
  const long long n = 1 << 20; const int D = 42;
  for (int K : {1, 4, 8, 16}) {
    Args a; a.n = n; a.K = K;
    float4 *act; CK(hipMalloc(&act, n * 16 * K)); CK(hipMemset(act, 0, n * 16 * K)); a.act = act;
    for (int j = 0; j < 5; ++j) { CK(hipMalloc(&a.st[j], n * 16)); CK(hipMemset(a.st[j], 0, n * 16)); }
    CK(hipMalloc(&a.obs, n * D * 4 * K)); CK(hipMalloc(&a.rew, n * 4 * K)); CK(hipMalloc(&a.cost, n * 4 * K));
    CK(hipMalloc(&a.term, n * K)); CK(hipMalloc(&a.trunc, n * K));
    const int grid = (int)(n / 256);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((stepk<D>), dim3(grid), dim3(256), 0, 0, a);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    const int iters = 100;
    for (int w = 0; w < iters; ++w) hipLaunchKernelGGL((stepk<D>), dim3(grid), dim3(256), 0, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters / K;
    const double bytes = 16 + 4.0 * D + 10 + 160.0 / K;
    printf("step_k traffic shape, Hover 2^20, K = %2d: %7.2f us per env-step, %5.1f B/env-step -> %6.1f GB/s = %.1f %% of 8 TB/s\n", K, us,
           bytes, bytes * n / us / 1e3, 100.0 * bytes * n / (us * 1e-6) / 8e12);
    CK(hipFree(act)); for (int j = 0; j < 5; ++j) CK(hipFree(a.st[j]));
    CK(hipFree(a.obs)); CK(hipFree(a.rew)); CK(hipFree(a.cost)); CK(hipFree(a.term)); CK(hipFree(a.trunc));
  }
  return 0;
}
