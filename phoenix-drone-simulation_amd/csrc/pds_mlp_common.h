// pds_mlp_common.h -- what the trainer's dense kernels share (csrc/pds_mlp.hip: d_in <= 64, two waves per SIMD;
// csrc/pds_mlp_wide.hip: 64 < d_in <= 192, the first layer K-tiled): argument block, parameter layout, activations,
// LDS access helpers.  See pds_mlp.hip for the design.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pds.h"

namespace pds_mlp_detail {  // named (not anonymous) so that profiler kernel names are readable

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTS = 16;            // samples per wave tile (= N of every activation GEMM)
constexpr int kTW = 16;            // feature tile width
constexpr int kNT = 4;             // 16-wide tiles per 64-wide dimension
constexpr int kS = 68;             // image row stride: 16-B aligned rows (b128 access) and 4 * kS == 16 (mod 32),
                                   // so the sample-slot walk of the dword reads below is bank-conflict free
constexpr int kSY = 20;            // row stride of the [16 samples][16 outputs] output-gradient image (same rule)
constexpr int kMaxDim = 64;        // d_in, h1, h2 <= 64
constexpr int kMaxOut = 8;         // d_out <= 8
constexpr int kWaves = 8;          // waves per block, two per SIMD; 1 block per CU (LDS-bound)
constexpr int kStats = 4;          // loss sum, ratio sum, kl sum, sample count
constexpr int kWaveFloats = 3 * kTS * kS + kTS * kSY;
// csrc/pds_mlp_wide.hip: 64 < d_in <= 192
constexpr int kMaxDimIn = 192;     // d_in of the K-tiled kernels (twelve 16-wide input tiles)
constexpr int kWideWaves = 4;      // waves per block there: one per SIMD, up to 512 registers each
constexpr int kWideMaxBlocks = 256;

enum { LOSS_NONE = 0, LOSS_PPO = 1, LOSS_MSE = 2 };

struct Args {
  pds_mlp m;
  const float *x;            // [rows, d_in]
  const int64_t *index;      // optional gather: sample g reads row index[g]
  long long B;               // samples
  const float *mean, *stdv;  // optional input standardisation (x - mean) / (std + eps)
  float eps;
  float *y;                  // forward output [B, d_out]
  const float *act, *adv, *logp_old, *log_std;  // PPO
  const float *target;                          // MSE
  float clip;
  float *partials;           // [waves of the grid][pstride]
  int pstride;
};

// ACT 0 relu, 1 tanh (branch-free: 1 - 2 / (e^{2v} + 1) on v_exp_f32 / v_rcp_f32, abs error < 3e-7)
template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
  if (ACT == 0) return fmaxf(v, 0.f);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * v) + 1.f);
}
// derivative expressed through the activation's OUTPUT h (relu: h > 0; tanh: 1 - h^2)
template <int ACT>
__device__ __forceinline__ float act_grad(float h) { return ACT == 0 ? (h > 0.f ? 1.f : 0.f) : 1.f - h * h; }

// flat parameter layout == torch's nn.Sequential order: W1 [h1][d_in], b1, W2 [h2][h1], b2, W3 [d_out][h2], b3
struct Offsets {
  int w1, b1, w2, b2, w3, b3, total;
};
__host__ __device__ inline Offsets offsets(const pds_mlp &m) {
  Offsets o;
  o.w1 = 0;
  o.b1 = o.w1 + m.h1 * m.d_in;
  o.w2 = o.b1 + m.h1;
  o.b2 = o.w2 + m.h2 * m.h1;
  o.w3 = o.b2 + m.h2;
  o.b3 = o.w3 + m.d_out * m.h2;
  o.total = o.b3 + m.d_out;
  return o;
}

#define PDS_WAVE_SYNC()                                          \
  do {                                                           \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       \
    __builtin_amdgcn_wave_barrier();                             \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       \
  } while (0)

#define PDS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x4 lds4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void sts4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }

// ---- split-bf16 operands (round 6) ------------------------------------------------------------------------------
// x = hi + mid + lo, three bf16 pieces (round to nearest even; the residuals x - hi and (x - hi) - mid are exact in f32), and
// the six products hi hi, hi mid, mid hi, hi lo, lo hi, mid mid on v_mfma_f32_16x16x32_bf16 (16 cycles per K = 32 against
// 8 x 32 for v_mfma_f32_16x16x4_f32): products exact, one f32 rounding per 32 terms -- max error / sum |products| 2^-24.5 ..
// 2^-23.1, below the f32 MFMA's own 2^-22.6 .. 2^-21.8 (profiles/r06_split_bf16.txt).  4.5 vector instructions per element.
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {  // v_cvt_pk_bf16_f32: low half a, high half b
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_){a, b}, bf16x2_));
}
struct Quad3 { u32x2_ hi, mid, lo; };  // four values in three pieces, two dwords per piece
__device__ __forceinline__ Quad3 split4(const f32x4 x) {
  Quad3 s;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const uint32_t h = pk_bf16(x[2 * p], x[2 * p + 1]);
    const float r0 = x[2 * p] - __uint_as_float(h << 16), r1 = x[2 * p + 1] - __uint_as_float(h & 0xFFFF0000u);
    const uint32_t m = pk_bf16(r0, r1);
    s.hi[p] = h;
    s.mid[p] = m;
    s.lo[p] = pk_bf16(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xFFFF0000u));
  }
  return s;
}
// the 8 k-slots of a lane: slots 0..3 = `a`, slots 4..7 = `b`
__device__ __forceinline__ bf16x8_ cat8(const u32x2_ a, const u32x2_ b) {
  const u32x4_ v = {a[0], a[1], b[0], b[1]};
  return __builtin_bit_cast(bf16x8_, v);
}
#define PDS_MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
// C += A B over the 8 k-slots of every lane group, A = (a0 | a1), B = (b0 | b1) in three pieces each: the six products, smallest first
struct Oct3 { bf16x8_ hi, mid, lo; };
__device__ __forceinline__ Oct3 oct3(const Quad3 &a, const Quad3 &b) { return Oct3{cat8(a.hi, b.hi), cat8(a.mid, b.mid), cat8(a.lo, b.lo)}; }

// csrc/pds_mlp_wide.hip: launches mlp_wide_kernel<loss, activation, input tiles> on `s`; returns the number of partials
// (one per wave of the grid) that the reduce kernel has to sum (gradient calls)
int launch_wide(int loss, const Args &a, hipStream_t s);

}  // namespace pds_mlp_detail
