// pds_mlp_fwd.h -- the forward pass of the trainer's 3-layer MLPs on v_mfma_f32_16x16x4_f32 as device
// functions, for kernels that run network inference next to other work (csrc/pds_rollout.h: the fused rollout).
// Same operand layout and the same k-ordered accumulation as mlp_kernel<LOSS_NONE> of csrc/pds_mlp.hip
// (pds_mlp_forward), so both produce the same bits for the same sample: an all-padding k-step that one of them
// skips adds +0 * x to the accumulator, which changes nothing.
//
// Reference: ActorCritic.step / MLPGaussianActor / MLPCritic, algs/core.py:228-311, 370-393;
// OnlineMeanStd.forward, utils/online_mean_std.py:32-43.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pds.h"

namespace pds_mlpf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTS = 16;      // samples per wave tile (= N of every activation GEMM)
constexpr int kTW = 16;      // feature tile width
constexpr int kNT = 4;       // 16-wide tiles per 64-wide dimension
constexpr int kS = 68;       // row stride of the LDS weight images (see pds_mlp.hip)
constexpr int kMaxDim = 64;  // d_in, h1, h2 <= 64

__device__ __forceinline__ f32x4 lds4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

#define PDS_MLPF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ACT 0 relu, 1 tanh (branch-free: 1 - 2 / (e^{2v} + 1) on v_exp_f32 / v_rcp_f32) -- pds_mlp.hip act_fn
template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
  if (ACT == 0) return fmaxf(v, 0.f);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * v) + 1.f);
}

// LDS image of one network: weights [out][in] zero padded to 64 x 68 (W3: 16 x 68), biases zero padded.
struct NetLds {
  float *W1, *W2, *W3, *b1, *b2, *b3;
};
constexpr int kNetFloats = 2 * kMaxDim * kS + kTW * kS + 2 * kMaxDim + kTW;

__device__ __forceinline__ NetLds net_lds(float *base) {
  NetLds n;
  n.W1 = base;
  n.W2 = n.W1 + kMaxDim * kS;
  n.W3 = n.W2 + kMaxDim * kS;
  n.b1 = n.W3 + kTW * kS;
  n.b2 = n.b1 + kMaxDim;
  n.b3 = n.b2 + kMaxDim;
  return n;
}

// all threads of the block; the caller synchronises afterwards
__device__ __forceinline__ void stage_net(const pds_mlp &m, const NetLds &n, int tid, int nthreads) {
  for (int i = tid; i < kMaxDim * kS; i += nthreads) {
    const int r = i / kS, k = i - r * kS;
    n.W1[i] = (r < m.h1 && k < m.d_in) ? m.w1[r * m.d_in + k] : 0.f;
    n.W2[i] = (r < m.h2 && k < m.h1) ? m.w2[r * m.h1 + k] : 0.f;
    if (i < kTW * kS) n.W3[i] = (r < m.d_out && k < m.h2) ? m.w3[r * m.h2 + k] : 0.f;
  }
  for (int i = tid; i < kMaxDim; i += nthreads) {
    n.b1[i] = i < m.h1 ? m.b1[i] : 0.f;
    n.b2[i] = i < m.h2 ? m.b2[i] : 0.f;
    if (i < kTW) n.b3[i] = i < m.d_out ? m.b3[i] : 0.f;
  }
}

// Z^T tiles `it`, `it + 1` (16 output features x 16 samples each) = W[16 it .. +32][:] * In^T; In^T as NK register
// tiles in the C/D layout (pds_mlp.hip gemm_wt2): two accumulation chains alternate.
// LASTJ: k-steps of the LAST k-tile that carry data (pds_mlp.hip gemm_wt: step j of k-tile kt holds features
// 16 kt + 4 h + j; with 50 hidden units or 34 inputs only j < 2 do) -- the skipped steps would add +0 * x.
template <int NK, int LASTJ = 4>
__device__ __forceinline__ void gemm_wt2(const float *Ws, int it, const f32x4 (&in)[NK], int n, int g, f32x4 &c0, f32x4 &c1) {
  c0 = (f32x4)(0.f);
  c1 = (f32x4)(0.f);
  const float *wp = Ws + (it * kTW + n) * kS + 4 * g;
#pragma unroll
  for (int kt = 0; kt < NK; ++kt) {
    const f32x4 a0 = lds4(wp + kt * kTW), a1 = lds4(wp + kTW * kS + kt * kTW);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (kt < NK - 1 || j < LASTJ) {
        c0 = PDS_MLPF_MFMA(a0[j], in[kt][j], c0);
        c1 = PDS_MLPF_MFMA(a1[j], in[kt][j], c1);
      }
    }
  }
}
template <int NK, int LASTJ = 4>
__device__ __forceinline__ f32x4 gemm_wt(const float *Ws, int it, const f32x4 (&in)[NK], int n, int g) {
  f32x4 c = (f32x4)(0.f);
  const float *wp = Ws + (it * kTW + n) * kS + 4 * g;
#pragma unroll
  for (int kt = 0; kt < NK; ++kt) {
    const f32x4 a = lds4(wp + kt * kTW);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (kt < NK - 1 || j < LASTJ) c = PDS_MLPF_MFMA(a[j], in[kt][j], c);
  }
  return c;
}

// y^T = W3 act(W2 act(W1 x^T + b1) + b2) + b3 for the wave's 16 samples.  Lane (n = lane & 15, g = lane >> 4) passes
// features 16 kt + 4 g + q of sample n in xin[kt][q] (already standardised, padding features 0) and receives
// outputs 4 g + q of sample n (rows >= d_out: 0 + 0).  KJI / KJH: data-carrying k-steps of the last input / hidden
// k-tile (4 = all; 2 for 34 inputs / 50 hidden units: the all-padding steps are not issued, same bits).
template <int ACT, int NIN, int KJI = 4, int KJH = 4>
__device__ __forceinline__ f32x4 forward16(const NetLds &w, const f32x4 (&xin)[NIN], int n, int g) {
  f32x4 h1r[kNT], h2r[kNT], cc[kNT];
#pragma unroll
  for (int it = 0; it < kNT; it += 2) gemm_wt2<NIN, KJI>(w.W1, it, xin, n, g, cc[it], cc[it + 1]);
#pragma unroll
  for (int it = 0; it < kNT; ++it) {
    const f32x4 b = lds4(w.b1 + it * kTW + 4 * g);
#pragma unroll
    for (int q = 0; q < 4; ++q) h1r[it][q] = act_fn<ACT>(cc[it][q] + b[q]);
  }
#pragma unroll
  for (int it = 0; it < kNT; it += 2) gemm_wt2<kNT, KJH>(w.W2, it, h1r, n, g, cc[it], cc[it + 1]);
#pragma unroll
  for (int it = 0; it < kNT; ++it) {
    const f32x4 b = lds4(w.b2 + it * kTW + 4 * g);
#pragma unroll
    for (int q = 0; q < 4; ++q) h2r[it][q] = act_fn<ACT>(cc[it][q] + b[q]);
  }
  const f32x4 c = gemm_wt<kNT, KJH>(w.W3, 0, h2r, n, g);
  const f32x4 b = lds4(w.b3 + 4 * g);
  return c + b;
}

// The same pass for a first layer that is K-tiled over NIN input tiles with an LDS image of row stride S1 = 16 NIN + 4
// (csrc/pds_mlp_wide.hip mlp_wide_kernel<LOSS_NONE>, pds_mlp_forward with more than 64 inputs): same k-order, same bits.
template <int NK, int S>
__device__ __forceinline__ void gemm_wt2s(const float *Ws, int it, const f32x4 (&in)[NK], int n, int g, f32x4 &c0, f32x4 &c1) {
  c0 = (f32x4)(0.f);
  c1 = (f32x4)(0.f);
  const float *wp = Ws + (it * kTW + n) * S + 4 * g;
#pragma unroll
  for (int kt = 0; kt < NK; ++kt) {
    const f32x4 a0 = lds4(wp + kt * kTW), a1 = lds4(wp + kTW * S + kt * kTW);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      c0 = PDS_MLPF_MFMA(a0[j], in[kt][j], c0);
      c1 = PDS_MLPF_MFMA(a1[j], in[kt][j], c1);
    }
  }
}
// w.W1: [64][S1] image; w.W2 / w.W3 / biases as in NetLds.  KJH as in forward16.
template <int ACT, int NIN, int S1, int KJH = 4>
__device__ __forceinline__ f32x4 forward16_wide(const NetLds &w, const f32x4 (&xin)[NIN], int n, int g) {
  f32x4 h1r[kNT], h2r[kNT], cc[kNT];
#pragma unroll
  for (int it = 0; it < kNT; it += 2) gemm_wt2s<NIN, S1>(w.W1, it, xin, n, g, cc[it], cc[it + 1]);
#pragma unroll
  for (int it = 0; it < kNT; ++it) {
    const f32x4 b = lds4(w.b1 + it * kTW + 4 * g);
#pragma unroll
    for (int q = 0; q < 4; ++q) h1r[it][q] = act_fn<ACT>(cc[it][q] + b[q]);
  }
#pragma unroll
  for (int it = 0; it < kNT; it += 2) gemm_wt2<kNT, KJH>(w.W2, it, h1r, n, g, cc[it], cc[it + 1]);
#pragma unroll
  for (int it = 0; it < kNT; ++it) {
    const f32x4 b = lds4(w.b2 + it * kTW + 4 * g);
#pragma unroll
    for (int q = 0; q < 4; ++q) h2r[it][q] = act_fn<ACT>(cc[it][q] + b[q]);
  }
  const f32x4 c = gemm_wt<kNT, KJH>(w.W3, 0, h2r, n, g);
  const f32x4 b = lds4(w.b3 + 4 * g);
  return c + b;
}
// LDS image of a network whose first layer has NIN input tiles (W1 row stride S1 = 16 NIN + 4)
template <int NIN>
constexpr int net_floats_wide() { return kMaxDim * (kTW * NIN + 4) + kMaxDim * kS + kTW * kS + 2 * kMaxDim + kTW; }
template <int NIN>
__device__ __forceinline__ NetLds net_lds_wide(float *base) {
  NetLds n;
  n.W1 = base;
  n.W2 = n.W1 + kMaxDim * (kTW * NIN + 4);
  n.W3 = n.W2 + kMaxDim * kS;
  n.b1 = n.W3 + kTW * kS;
  n.b2 = n.b1 + kMaxDim;
  n.b3 = n.b2 + kMaxDim;
  return n;
}
template <int NIN>
__device__ __forceinline__ void stage_net_wide(const pds_mlp &m, const NetLds &n, int tid, int nthreads) {
  constexpr int S1 = kTW * NIN + 4;
  for (int i = tid; i < kMaxDim * S1; i += nthreads) {
    const int r = i / S1, k = i - r * S1;
    n.W1[i] = (r < m.h1 && k < m.d_in) ? m.w1[r * m.d_in + k] : 0.f;
  }
  for (int i = tid; i < kMaxDim * kS; i += nthreads) {
    const int r = i / kS, k = i - r * kS;
    n.W2[i] = (r < m.h2 && k < m.h1) ? m.w2[r * m.h1 + k] : 0.f;
    if (i < kTW * kS) n.W3[i] = (r < m.d_out && k < m.h2) ? m.w3[r * m.h2 + k] : 0.f;
  }
  for (int i = tid; i < kMaxDim; i += nthreads) {
    n.b1[i] = i < m.h1 ? m.b1[i] : 0.f;
    n.b2[i] = i < m.h2 ? m.b2[i] : 0.f;
    if (i < kTW) n.b3[i] = i < m.d_out ? m.b3[i] : 0.f;
  }
}

// run-time shape -> the instantiation with the fewest k-steps (data steps of the last k-tile of a dimension `dim`
// that spans `tiles` 16-wide tiles; 4 when the tile is full or the dimension ends in an earlier tile)
__device__ __forceinline__ int last_tile_steps(int dim, int tiles) {
  const int rest = dim - 16 * (tiles - 1);  // features in the last tile
  return (rest >= 1 && rest <= 2) ? 2 : 4;  // (1-2 features: lane group 0, steps 0-1; 3-4 would be 4 steps: 4 h + j < rest)
}
template <int ACT, int NIN>
__device__ __forceinline__ f32x4 forward16_shape(const NetLds &w, const pds_mlp &m, const f32x4 (&xin)[NIN], int n, int g) {
  const bool ki2 = last_tile_steps(m.d_in, NIN) == 2;
  const bool kh2 = m.h1 == m.h2 && last_tile_steps(m.h1, kNT) == 2;
  if (kh2) return ki2 ? forward16<ACT, NIN, 2, 2>(w, xin, n, g) : forward16<ACT, NIN, 4, 2>(w, xin, n, g);
  return ki2 ? forward16<ACT, NIN, 2, 4>(w, xin, n, g) : forward16<ACT, NIN, 4, 4>(w, xin, n, g);
}

}  // namespace pds_mlpf
