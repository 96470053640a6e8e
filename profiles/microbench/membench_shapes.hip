// membench_shapes.hip -- traffic-only ceiling for every bench configuration (no physics, same bytes,
// same launch geometry as pds::step_kernel: 256-thread blocks, one 64-row LDS observation tile per
// wave, padded tile stride for D % 8 == 0, XCD-contiguous block order, non-temporal row stores).
//
// A shape is (envs, D, read-only quads, read+write quads): every env-step reads the read-only and
// read+write float4 arrays and the 4-byte counter, and writes the read+write arrays, the counter,
// reward, cost, two flag bytes and the D-float observation row.  The shapes below are the physical
// arrays the product kernels of bench.py --config 0/2/3/4/6 touch.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/membench_shapes profiles/microbench/membench_shapes.hip && /tmp/membench_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr int kMaxArr = 8;
struct Shape {
  const float4 *ro[kMaxArr];
  float4 *rw[kMaxArr];
  unsigned *ctr;
  float *obs, *rew, *cost;
  unsigned char *term, *trunc;
  long long n;
};

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4nt(float4 *p, float4 v) {
  v4f t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(p));
}

template <int D> constexpr int stride_of() { return (D % 8 == 0) ? D + 4 : D; }

template <int D, int NRO, int NRW>
__global__ __launch_bounds__(256) void mix(Shape a) {
  constexpr int S = stride_of<D>();
  __shared__ __attribute__((aligned(16))) float tile_all[4 * 64 * S];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *tile = tile_all + wave * 64 * S;
  long long blk = blockIdx.x;
  const long long nb = gridDim.x, per = nb / 8;
  if (per * 8 == nb) blk = (blk % 8) * per + blk / 8;  // one XCD streams one contiguous eighth
  const long long t = blk * 4 + wave;
  if (t * 64 >= a.n) return;
  const long long i = t * 64 + lane;
  float4 r[NRO], w[NRW];
#pragma unroll
  for (int j = 0; j < NRO; ++j) r[j] = a.ro[j][i];
#pragma unroll
  for (int j = 0; j < NRW; ++j) w[j] = a.rw[j][i];
  const unsigned c = a.ctr[i];
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < NRO; ++j) acc += r[j].x + r[j].y + r[j].z + r[j].w;
  float *row = tile + lane * S;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const float4 q = w[j / 4 % NRW];
    row[j] = (j % 4 == 0 ? q.x : j % 4 == 1 ? q.y : j % 4 == 2 ? q.z : q.w) + acc + (float)j;
  }
#pragma unroll
  for (int j = 0; j < NRW; ++j) a.rw[j][i] = make_float4(w[j].x + 1.f, w[j].y + acc, w[j].z, w[j].w);
  a.ctr[i] = c + 1;
  __builtin_nontemporal_store(acc, a.rew + i);
  __builtin_nontemporal_store(acc + 1.f, a.cost + i);
  a.term[i] = (unsigned char)(c & 1);
  a.trunc[i] = (unsigned char)((c >> 1) & 1);
  __builtin_amdgcn_wave_barrier();
  float4 *dst = reinterpret_cast<float4 *>(a.obs + t * 64 * D);
  if (S == D) {
    const float4 *src4 = reinterpret_cast<const float4 *>(tile);
#pragma unroll
    for (int it = 0; it < (16 * D + 63) / 64; ++it) {
      const int idx = it * 64 + lane;
      if (idx < 16 * D) st4nt(dst + idx, src4[idx]);
    }
  } else {  // padded rows: D / 4 quads per row
    constexpr int QR = D / 4;
#pragma unroll
    for (int it = 0; it < (64 * QR + 63) / 64; ++it) {
      const int idx = it * 64 + lane;
      if (idx < 64 * QR) {
        const int rr = idx / QR, qq = idx - rr * QR;
        st4nt(dst + idx, *reinterpret_cast<const float4 *>(tile + rr * S + 4 * qq));
      }
    }
  }
}

template <typename F> static float time_ms(F launch, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

template <int D, int NRO, int NRW>
static void run(const char *name, long long n, int algo_bytes) {
  Shape a; a.n = n;
  const int T = 4;  // input slabs cycled like fresh action batches
  float4 *ro0;
  CK(hipMalloc(&ro0, n * 16 * T)); CK(hipMemset(ro0, 0, n * 16 * T));
  for (int j = 1; j < NRO; ++j) { float4 *p; CK(hipMalloc(&p, n * 16)); CK(hipMemset(p, 0, n * 16)); a.ro[j] = p; }
  for (int j = 0; j < NRW; ++j) { CK(hipMalloc(&a.rw[j], n * 16)); CK(hipMemset(a.rw[j], 0, n * 16)); }
  CK(hipMalloc(&a.ctr, n * 4)); CK(hipMemset(a.ctr, 0, n * 4));
  float *obs0; CK(hipMalloc(&obs0, n * D * 4 * 2));
  CK(hipMalloc(&a.rew, n * 4)); CK(hipMalloc(&a.cost, n * 4)); CK(hipMalloc(&a.term, n)); CK(hipMalloc(&a.trunc, n));
  int step = 0;
  const int grid = (int)((n / 64 + 3) / 4);
  const float ms = time_ms([&]() {
    a.ro[0] = ro0 + (step % T) * n; a.obs = obs0 + (step & 1) * n * D; ++step;
    hipLaunchKernelGGL((mix<D, NRO, NRW>), dim3(grid), dim3(256), 0, 0, a);
  }, 400);
  CK(hipGetLastError());
  const double phys = ((NRO + NRW) * 16 + 4 + NRW * 16 + 4 + 10 + 4.0 * D) * n;
  printf("%-44s n=%8lld  %7.2f us   physical %6.1f MB %6.1f GB/s   algorithmic %d B/env-step -> %.1f %% of 8 TB/s\n",
         name, n, ms * 1e3, phys / 1e6, phys / ms / 1e6, algo_bytes, 100.0 * algo_bytes * n / (ms * 1e-3) / 8e12);
  CK(hipFree(ro0)); for (int j = 1; j < NRO; ++j) CK(hipFree((void *)a.ro[j])); for (int j = 0; j < NRW; ++j) CK(hipFree(a.rw[j]));
  CK(hipFree(a.ctr)); CK(hipFree(obs0)); CK(hipFree(a.rew)); CK(hipFree(a.cost)); CK(hipFree(a.term)); CK(hipFree(a.trunc));
}

// Tile-major state layout (experiment, round 3): the read-only / read+write quads and the counter of one 64-env tile
// are contiguous ((NRO + NRW) KiB + 256 B per tile) instead of one array per field: a wave touches ONE region of the
// state instead of NRO + NRW + 1 regions 16+ MB apart.  Same bytes, same outputs.
template <int D, int NRO, int NRW>
__global__ __launch_bounds__(256) void mix_tm(Shape a, char *slab, const float4 *act) {
  constexpr int S = stride_of<D>();
  constexpr int TB = (NRO - 1 + NRW) * 1024 + 256;  // bytes per tile (the first read-only array is the action: separate)
  __shared__ __attribute__((aligned(16))) float tile_all[4 * 64 * S];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *tile = tile_all + wave * 64 * S;
  long long blk = blockIdx.x;
  const long long nb = gridDim.x, per = nb / 8;
  if (per * 8 == nb) blk = (blk % 8) * per + blk / 8;
  const long long t = blk * 4 + wave;
  if (t * 64 >= a.n) return;
  const long long i = t * 64 + lane;
  char *tb = slab + t * TB;
  float4 r[NRO], w[NRW];
  r[0] = act[i];
#pragma unroll
  for (int j = 1; j < NRO; ++j) r[j] = reinterpret_cast<const float4 *>(tb + (j - 1) * 1024)[lane];
#pragma unroll
  for (int j = 0; j < NRW; ++j) w[j] = reinterpret_cast<const float4 *>(tb + (NRO - 1 + j) * 1024)[lane];
  const unsigned c = reinterpret_cast<const unsigned *>(tb + (NRO - 1 + NRW) * 1024)[lane];
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < NRO; ++j) acc += r[j].x + r[j].y + r[j].z + r[j].w;
  float *row = tile + lane * S;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const float4 q = w[j / 4 % NRW];
    row[j] = (j % 4 == 0 ? q.x : j % 4 == 1 ? q.y : j % 4 == 2 ? q.z : q.w) + acc + (float)j;
  }
#pragma unroll
  for (int j = 0; j < NRW; ++j) reinterpret_cast<float4 *>(tb + (NRO - 1 + j) * 1024)[lane] = make_float4(w[j].x + 1.f, w[j].y + acc, w[j].z, w[j].w);
  reinterpret_cast<unsigned *>(tb + (NRO - 1 + NRW) * 1024)[lane] = c + 1;
  __builtin_nontemporal_store(acc, a.rew + i);
  __builtin_nontemporal_store(acc + 1.f, a.cost + i);
  a.term[i] = (unsigned char)(c & 1);
  a.trunc[i] = (unsigned char)((c >> 1) & 1);
  __builtin_amdgcn_wave_barrier();
  float4 *dst = reinterpret_cast<float4 *>(a.obs + t * 64 * D);
  if (S == D) {
    const float4 *src4 = reinterpret_cast<const float4 *>(tile);
#pragma unroll
    for (int it = 0; it < (16 * D + 63) / 64; ++it) {
      const int idx = it * 64 + lane;
      if (idx < 16 * D) st4nt(dst + idx, src4[idx]);
    }
  } else {
    constexpr int QR = D / 4;
#pragma unroll
    for (int it = 0; it < (64 * QR + 63) / 64; ++it) {
      const int idx = it * 64 + lane;
      if (idx < 64 * QR) {
        const int rr = idx / QR, qq = idx - rr * QR;
        st4nt(dst + idx, *reinterpret_cast<const float4 *>(tile + rr * S + 4 * qq));
      }
    }
  }
}

template <int D, int NRO, int NRW>
static void run_tm(const char *name, long long n, int algo_bytes) {
  Shape a; a.n = n;
  const int T = 4;
  constexpr int TB = (NRO - 1 + NRW) * 1024 + 256;
  float4 *act; CK(hipMalloc(&act, n * 16 * T)); CK(hipMemset(act, 0, n * 16 * T));
  char *slab; CK(hipMalloc(&slab, (n / 64) * TB)); CK(hipMemset(slab, 0, (n / 64) * TB));
  float *obs0; CK(hipMalloc(&obs0, n * D * 4 * 2));
  CK(hipMalloc(&a.rew, n * 4)); CK(hipMalloc(&a.cost, n * 4)); CK(hipMalloc(&a.term, n)); CK(hipMalloc(&a.trunc, n));
  int step = 0;
  const int grid = (int)((n / 64 + 3) / 4);
  const float ms = time_ms([&]() {
    a.obs = obs0 + (step & 1) * n * D; const float4 *ac = act + (step % T) * n; ++step;
    hipLaunchKernelGGL((mix_tm<D, NRO, NRW>), dim3(grid), dim3(256), 0, 0, a, slab, ac);
  }, 400);
  CK(hipGetLastError());
  printf("%-44s n=%8lld  %7.2f us   TILE-MAJOR state   algorithmic %d B/env-step -> %.1f %% of 8 TB/s\n",
         name, n, ms * 1e3, algo_bytes, 100.0 * algo_bytes * n / (ms * 1e-3) / 8e12);
  CK(hipFree(act)); CK(hipFree(slab)); CK(hipFree(obs0)); CK(hipFree(a.rew)); CK(hipFree(a.cost)); CK(hipFree(a.term)); CK(hipFree(a.trunc));
}

int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<42, 2, 4>("config 2  Hover", 65536, 346);
    run<40, 6, 5>("config 3  Circle + PT1 + DR", 262144, 426);
    run<42, 2, 4>("config 0  Hover (headline)", 1 << 20, 346);
    run<48, 2, 4>("config 4  TakeOff + ground effect", 1 << 20, 370);
    run<42, 4, 8>("config 6  Hover, sensor noise + DR", 1 << 20, 498);
    run<42, 2, 4>("          Hover 2^21", 1 << 21, 346);
    run_tm<42, 2, 4>("config 2  Hover", 65536, 346);
    run_tm<40, 6, 5>("config 3  Circle + PT1 + DR", 262144, 426);
    run_tm<42, 2, 4>("config 0  Hover (headline)", 1 << 20, 346);
    run_tm<48, 2, 4>("config 4  TakeOff + ground effect", 1 << 20, 370);
    run_tm<42, 4, 8>("config 6  Hover, sensor noise + DR", 1 << 20, 498);
    run_tm<42, 2, 4>("          Hover 2^21", 1 << 21, 346);
  }
  return 0;
}
