// membench.hip -- traffic-shape ceiling for the fused step kernel (no physics, same bytes).
//
// Moves exactly the bytes pds::step_kernel<Hover> moves per env-step (read 100 B: action, 3 state
// quads, 2 history quads, counter; write 246 B: 3 state quads, history quad, counter, reward, cost,
// 2 flag bytes, 42-float observation row) in several launch shapes, to separate "what the memory
// system gives this read/write mix" from "what the physics costs".  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/membench profiles/microbench/membench.hip && /tmp/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr int D = 42;

struct Arrs {
  const float4 *act; float4 *s0, *s1, *s2, *h0, *h1; unsigned *ctr;
  float *obs, *rew, *cost; unsigned char *term, *trunc; long long n;
};

typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void st4(float4 *p, float4 v) {
  if (NT) { v4f t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(p)); } else *p = v;
}
template <bool NT> __device__ __forceinline__ float4 ld4(const float4 *p) {
  if (NT) { v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p)); return make_float4(t.x, t.y, t.z, t.w); }
  else return *p;
}

__global__ void copy4(const float4 *__restrict__ a, float4 *__restrict__ b, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[i];
}

// MODE 0: per-wave LDS tile, one tile per wave (the shipped structure)
// MODE 1: strided row stores, no LDS
// MODE 2: persistent grid-stride over wave tiles with register prefetch of the next tile's loads
template <int MODE, bool NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void mix(Arrs a) {
  __shared__ __attribute__((aligned(16))) float tile_all[(MODE == 1) ? 4 : BLOCK * D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *tile = tile_all + ((MODE == 1) ? 0 : wave * 64 * D);
  const long long nt = a.n / 64;  // wave tiles
  long long t = (long long)blockIdx.x * (BLOCK / 64) + wave;
  const long long stride = (MODE == 2) ? (long long)gridDim.x * (BLOCK / 64) : nt;
  if (t >= nt) return;
  long long i = t * 64 + lane;
  float4 act = ld4<NT>(a.act + i), q0 = a.s0[i], q1 = a.s1[i], q2 = a.s2[i], g0 = a.h0[i], g1 = a.h1[i];
  unsigned c = a.ctr[i];
  for (; t < nt; t += stride) {
    i = t * 64 + lane;
    float4 nact, n0, n1, n2, m0, m1; unsigned nc = 0;
    const long long tn = t + stride;
    if (MODE == 2 && tn < nt) {
      const long long j = tn * 64 + lane;
      nact = ld4<NT>(a.act + j); n0 = a.s0[j]; n1 = a.s1[j]; n2 = a.s2[j]; m0 = a.h0[j]; m1 = a.h1[j]; nc = a.ctr[j];
    }
    float v[D];
    const float src[24] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w,
                           g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, act.x, act.y, act.z, act.w};
#pragma unroll
    for (int j = 0; j < D; ++j) v[j] = src[j % 24] + (float)j;
    if (MODE == 1) {
      float2 *row = reinterpret_cast<float2 *>(a.obs + i * D);
#pragma unroll
      for (int j = 0; j < D / 2; ++j) row[j] = make_float2(v[2 * j], v[2 * j + 1]);
    } else {
      float *row = tile + lane * D;
#pragma unroll
      for (int j = 0; j < D; ++j) row[j] = v[j];
    }
    a.s0[i] = make_float4(q0.x + 1.f, q0.y, q0.z, q0.w);
    a.s1[i] = make_float4(q1.x + 1.f, q1.y, q1.z, q1.w);
    a.s2[i] = make_float4(q2.x + 1.f, q2.y, q2.z, q2.w);
    a.h1[i] = act;
    a.ctr[i] = c + 1;
    if (NT) { __builtin_nontemporal_store(v[0], a.rew + i); __builtin_nontemporal_store(v[1], a.cost + i); }
    else { a.rew[i] = v[0]; a.cost[i] = v[1]; }
    a.term[i] = (unsigned char)(c & 1);
    a.trunc[i] = (unsigned char)((c >> 1) & 1);
    if (MODE != 1) {
      __builtin_amdgcn_wave_barrier();
      const float4 *src4 = reinterpret_cast<const float4 *>(tile);
      float4 *dst = reinterpret_cast<float4 *>(a.obs + t * 64 * D);
#pragma unroll
      for (int it = 0; it < (16 * D + 63) / 64; ++it) {
        const int idx = it * 64 + lane;
        if (idx < 16 * D) st4<NT>(dst + idx, src4[idx]);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (MODE == 2) { act = nact; q0 = n0; q1 = n1; q2 = n2; g0 = m0; g1 = m1; c = nc; }
  }
}

template <typename F> static float time_ms(F launch, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main(int argc, char **argv) {
  const long long n = argc > 1 ? atoll(argv[1]) : (1 << 20);  // envs
  printf("n = %lld envs, %.1f MB per pass\n", n, 346.0 * n / 1e6);
  const int T = 8;  // action slabs, cycled
  Arrs a; a.n = n;
  float4 *act; CK(hipMalloc(&act, n * 16 * T)); CK(hipMemset(act, 0, n * 16 * T));
  CK(hipMalloc(&a.s0, n * 16)); CK(hipMalloc(&a.s1, n * 16)); CK(hipMalloc(&a.s2, n * 16));
  CK(hipMalloc(&a.h0, n * 16)); CK(hipMalloc(&a.h1, n * 16)); CK(hipMalloc(&a.ctr, n * 4));
  CK(hipMalloc(&a.obs, n * D * 4 * 2)); CK(hipMalloc(&a.rew, n * 4)); CK(hipMalloc(&a.cost, n * 4));
  CK(hipMalloc(&a.term, n)); CK(hipMalloc(&a.trunc, n));
  CK(hipMemset(a.s0, 0, n * 16)); CK(hipMemset(a.s1, 0, n * 16)); CK(hipMemset(a.s2, 0, n * 16));
  CK(hipMemset(a.h0, 0, n * 16)); CK(hipMemset(a.h1, 0, n * 16)); CK(hipMemset(a.ctr, 0, n * 4));
  const double bytes = 346.0 * n;
  int step = 0;
  float *obs0 = a.obs;
  auto upd = [&]() { a.act = act + (step % T) * n; a.obs = obs0 + (step & 1) * n * D; ++step; };
  {
    const long long m = (long long)(bytes / 32);  // float4 elements so that read+write == bytes
    float4 *x, *y; CK(hipMalloc(&x, m * 16)); CK(hipMalloc(&y, m * 16)); CK(hipMemset(x, 0, m * 16));
    float ms = time_ms([&]() { hipLaunchKernelGGL(copy4, dim3((m + 255) / 256), dim3(256), 0, 0, x, y, m); }, 200);
    printf("%-44s %8.2f us  %7.1f GB/s\n", "copy4 (float4 copy, same total bytes)", ms * 1e3, bytes / ms / 1e6);
  }
#define RUN(name, MODE, NT, BLOCK, GRID)                                                            \
  { float ms = time_ms([&]() { upd(); hipLaunchKernelGGL((mix<MODE, NT, BLOCK>), dim3(GRID), dim3(BLOCK), 0, 0, a); }, 200); \
    CK(hipGetLastError());                                                                           \
    printf("%-44s %8.2f us  %7.1f GB/s  (%.1f%% of 8 TB/s)\n", name, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0); }
  const int tiles = (int)(n / 64);
  RUN("mix lds tile, block 256", 0, false, 256, tiles / 4);
  RUN("mix lds tile, block 256, nt", 0, true, 256, tiles / 4);
  RUN("mix lds tile, block 128", 0, false, 128, tiles / 2);
  RUN("mix lds tile, block 64", 0, false, 64, tiles);
  RUN("mix lds tile, block 64, nt", 0, true, 64, tiles);
  RUN("mix strided rows (no lds)", 1, false, 256, tiles / 4);
  RUN("mix persistent+prefetch, 256x3 blocks of 256", 2, false, 256, 256 * 3);
  RUN("mix persistent+prefetch, 256x3, nt", 2, true, 256, 256 * 3);
  RUN("mix persistent+prefetch, 256x14 blocks of 64", 2, false, 64, 256 * 14);
  RUN("mix persistent+prefetch, 256x14 of 64, nt", 2, true, 64, 256 * 14);
  RUN("mix persistent+prefetch, 256x7 blocks of 128", 2, false, 128, 256 * 7);
  RUN("mix persistent+prefetch, 256x6 blocks of 64", 2, false, 64, 256 * 6);
  return 0;
}
