// pds_mlp.hip -- the trainer's dense work on the f32 matrix cores of gfx950:
// fused 3-layer MLP forward (rollout inference of actor / critic) and fused loss + backward of the
// PPO-clip policy objective and of the value regression, one pass over the batch per call.
//
// Replaces, in the caller of the hot path, the PyTorch op chains of
//   ActorCritic.step / MLPGaussianActor / MLPCritic        algs/core.py:228-311, 370-393
//   ProximalPolicyOptimizationAlgorithm.compute_loss_pi     algs/ppo/ppo.py:22-40
//   IWPGAlgorithm.compute_loss_v / update_value_net         algs/iwpg/iwpg.py:272-275, 487-522
// (80 full-batch policy iterations + 5 x 16 value mini-batches per epoch, algs/ppo/defaults.py:6-19):
// autograd materialises ~30 [B, 50] tensors per iteration in HBM; here a wave keeps a 16-sample
// tile in LDS, runs every GEMM of forward and backward on v_mfma_f32_16x16x4_f32 (exact f32: a
// k-ordered fmaf chain, so results match an fp32 reference to rounding), accumulates the weight
// gradients in registers across its tiles and writes one partial per wave; a second tiny kernel
// sums the partials in a fixed order (deterministic, no atomics).
//
// A wave's GEMM -> epilogue -> GEMM phases depend on each other, so one wave cannot keep the matrix
// core and the vector ALU busy at once: the block runs TWO waves per SIMD (8 waves, 16-sample tiles,
// <= 256 registers each) and the hardware interleaves one wave's epilogues with the other's MFMAs.
//
// GEMM operands always come from LDS in natural [row][col] images with an odd row stride (65), so
// the A map (lane l: A[l&15][l>>4]) and the B map (B[l>>4][l&15]) of the instruction read either
// consecutive words or a conflict-free odd-stride column, whichever way a matrix is walked:
//   forward      Z1 = X W1^T, Z2 = H1 W2^T, Y = H2 W3^T            (M = samples)
//   backward     dH2 = dY W3, dH1 = dZ2 W2                          (M = samples)
//   weight grads dW3 = dY^T H2, dW2 = dZ2^T H1, dW1 = dZ1^T X       (K = the tile's 16 samples)
// Bound: MFMA f32 (157 TFLOP/s dense peak on MI355X = the f32 vector rate; MI355X_MICROARCH.md).
//
// Round 6: from 65 536 samples on, four of ppo_split_kernel's eight GEMMs run on v_mfma_f32_16x16x32_bf16 with every operand in
// three bf16 pieces (x = hi + mid + lo, six exact products per K = 32: more accurate than the f32 MFMA, 2.67 x its rate); see
// PDS_SPLIT_BF16 below and DESIGN.md section 9.
// Round 3: the PPO gradient of the reference's default policy (50-50 relu) runs on ppo_split_kernel below instead --
// the two waves of a SIMD take different ROLES on the same tiles (forward / loss / small GEMMs vs. the large
// weight-gradient GEMMs) so that neither carries 128 accumulator registers through phases that do not need them.
#include <stdlib.h>

#include "pds_mlp_common.h"

namespace pds_mlp_detail {

#ifndef PDS_SPLIT_WRES
#define PDS_SPLIT_WRES 1
#endif
#ifndef PDS_SPLIT_DYNPRIO
#define PDS_SPLIT_DYNPRIO 1
#endif
#if PDS_SPLIT_DYNPRIO  // A/B: the forward role raises its priority for its MFMA bursts only
#define PDS_FPRIO_MFMA() __builtin_amdgcn_s_setprio(2)
#define PDS_FPRIO_VALU() __builtin_amdgcn_s_setprio(0)
#else
#define PDS_FPRIO_MFMA() do { } while (0)
#define PDS_FPRIO_VALU() do { } while (0)
#endif
#ifndef PDS_SPLIT_SIMD_ROLES
#define PDS_SPLIT_SIMD_ROLES 0
#endif
#ifndef PDS_SPLIT_GPRIO
#define PDS_SPLIT_GPRIO 0
#endif
#ifndef PDS_SPLIT_FPRIO
#define PDS_SPLIT_FPRIO 3
#endif
#ifndef PDS_SPLIT_STAMPS
#define PDS_SPLIT_STAMPS 0
#endif
#ifndef PDS_SPLIT_DEBUG
#define PDS_SPLIT_DEBUG 0
#endif
#ifndef PDS_MLP_SPLIT
#define PDS_MLP_SPLIT 1  // weight-gradient roles of ppo_split_kernel; A/B: 2 = three waves per SIMD (measured: 454 us against 341 us at
                         // 1 M samples -- 168 registers per wave undo F's operand prefetch), 0 = the round-2 route through mlp_kernel
#endif
#ifndef PDS_MLP_EDGE
#define PDS_MLP_EDGE 1  // A/B: 0 = the fourth output tile of the 50-wide layers on the matrix cores as well
#endif
// v_mfma_f32_4x4x1_16b_f32: 16 independent 4 x 4 outer products per instruction (block b = lane / 4: D_b[i][j] += A_b[i] B_b[j],
// lane 4 b + i supplies A_b[i], lane 4 b + j supplies B_b[j] and holds D_b[0..3][j] in its four registers), 8 cycles
// against 32 for a 16x16x4 tile at the same flop rate: the right shape for a 4-row or 4-column STRIP of a weight
// gradient (rank-1 update per sample), where a 16-row tile would carry 2 useful rows.
#define PDS_MFMA44(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
#ifndef PDS_MLP_STRIPS
#define PDS_MLP_STRIPS 1  // round 4, ppo_split_kernel: rows / columns 48..51 of dW2, rows 48..51 of dW1 and the whole dW3 as 4x4x1 strips; A/B: 0
#endif
#ifndef PDS_SPLIT_BF16_L1
#define PDS_SPLIT_BF16_L1 1  // the forward role's layer 1 on the bf16 instruction too (W1 as an LDS image of its operand pieces, 18 KB in place of the f32 image); A/B: 0
#endif
#ifndef PDS_SPLIT_BF16_L2
#define PDS_SPLIT_BF16_L2 1  // the forward role's layer 2 on the bf16 instruction too (its A operand in pieces stays in registers, layer 1's then comes from LDS); A/B: 0
#endif
#ifndef PDS_SPLIT_BF16
#define PDS_SPLIT_BF16 1  // round 6, ppo_split_kernel: the weight-gradient role's three GEMMs (dZ1, dW2, dW1) as split-bf16 MFMAs; A/B: 0 = f32 MFMAs
#endif
#if PDS_SPLIT_DEBUG == 3  // profiling: the forward role WITHOUT its MFMAs -- operands stay alive, no instruction is
                           // issued: what the rest of its instruction stream costs the pair (results invalid)
typedef float pds_f32x4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ pds_f32x4_ pds_no_mfma(float a, float b, pds_f32x4_ c) {
  asm volatile("" : "+v"(c) : "v"(a), "v"(b));
  return c;
}
#define PDS_MFMA_F(a, b, c) pds_no_mfma((a), (b), (c))
#else
#define PDS_MFMA_F(a, b, c) PDS_MFMA(a, b, c)
#endif

// Z^T tile `it` (16 output features x 16 samples) = W[16 it .. +16][:] * In^T, with In^T given as NK
// register tiles in the C/D layout: the k-slot (step j, lane group h) of feature tile kt carries
// feature 16 kt + 4 h + j, which is register j of that tile in every lane of group h -- no movement.
// The matching A operands are 4 consecutive floats of a weight row: one ds_read_b128 per (it, kt).
// LASTJ: k-steps of the LAST k-tile that carry data (step j holds features 16 kt + 4 h + j, so with
// dim = 16 (NK - 1) + d, d <= 4, only j < d do; 4 = all).  A compile-time parameter: the specialised
// kernels of the reference's default nets (50 hidden units, 34 inputs) skip their all-padding steps.
template <int NK, int LASTJ>
__device__ __forceinline__ f32x4 gemm_wt(const float *Ws, int it, const f32x4 (&in)[NK], int n, int g) {
  f32x4 c = (f32x4)(0.f);
  const float *wp = Ws + (it * kTW + n) * kS + 4 * g;
#pragma unroll
  for (int kt = 0; kt < NK; ++kt) {
    const f32x4 a = lds4(wp + kt * kTW);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (kt < NK - 1 || j < LASTJ) c = PDS_MFMA(a[j], in[kt][j], c);
  }
  return c;
}

// Two output tiles at once: their accumulation chains alternate, so no MFMA waits for the 40-cycle
// dependent-accumulator latency of v_mfma_f32_16x16x4_f32 (issue interval 32 cycles).
template <int NK, int LASTJ>
__device__ __forceinline__ void gemm_wt2(const float *Ws, int it, const f32x4 (&in)[NK], int n, int g, f32x4 &c0,
                                         f32x4 &c1) {
  c0 = (f32x4)(0.f);
  c1 = (f32x4)(0.f);
  const float *wp = Ws + (it * kTW + n) * kS + 4 * g;
#pragma unroll
  for (int kt = 0; kt < NK; ++kt) {
    const f32x4 a0 = lds4(wp + kt * kTW), a1 = lds4(wp + kTW * kS + kt * kTW);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (kt < NK - 1 || j < LASTJ) {
        c0 = PDS_MFMA(a0[j], in[kt][j], c0);
        c1 = PDS_MFMA(a1[j], in[kt][j], c1);
      }
    }
  }
}

// ---- the two edge features of a 50-wide layer on the vector ALU (round 3) --------------------------------------
// 50 hidden units fill three 16-row MFMA tiles and TWO rows of a fourth: that tile is 12-14 MFMAs (384-448 cycles of
// the matrix pipe) for 2 useful rows of 16.  The gradient kernels of the reference's default policy net (h1 = h2 = 50:
// KJH == 2) compute features 48 and 49 as plain dot products instead -- each lane over the 12-16 input features it
// holds, the four lane groups of a sample added up with two cross-lane exchanges -- ~45 vector instructions that
// run on the OTHER pipe while the co-resident wave issues MFMAs.  Result in the C/D layout of tile 3: lanes of group
// 0 hold (feature 48, feature 49, 0, 0), the other groups zeros (features 52..63 are padding).
// edge_rows: rows 48, 49 of W (forward: z = W in);  edge_cols: columns 48, 49 of W (backward: dz1 = W^T dz2).
template <int NK>
__device__ __forceinline__ f32x4 edge_rows(const float *Ws, const f32x4 (&in)[NK], int g) {
  float e0 = 0.f, e1 = 0.f;
  const float *w0 = Ws + 48 * kS + 4 * g;
#pragma unroll
  for (int kt = 0; kt < NK; ++kt) {
    const f32x4 a0 = lds4(w0 + kt * kTW), a1 = lds4(w0 + kS + kt * kTW);
#pragma unroll
    for (int q = 0; q < 4; ++q) { e0 = fmaf(a0[q], in[kt][q], e0); e1 = fmaf(a1[q], in[kt][q], e1); }
  }
  e0 += __shfl_xor(e0, 16); e1 += __shfl_xor(e1, 16);
  e0 += __shfl_xor(e0, 32); e1 += __shfl_xor(e1, 32);
  f32x4 r = (f32x4)(0.f);
  if (g == 0) { r[0] = e0; r[1] = e1; }
  return r;
}
__device__ __forceinline__ f32x4 edge_cols(const float *Ws, const f32x4 (&in)[kNT], int g) {
  float e0 = 0.f, e1 = 0.f;
#pragma unroll
  for (int kt = 0; kt < kNT; ++kt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float2 w = *reinterpret_cast<const float2 *>(Ws + (kt * kTW + 4 * g + q) * kS + 48);
      e0 = fmaf(w.x, in[kt][q], e0); e1 = fmaf(w.y, in[kt][q], e1);
    }
  }
  e0 += __shfl_xor(e0, 16); e1 += __shfl_xor(e1, 16);
  e0 += __shfl_xor(e0, 32); e1 += __shfl_xor(e1, 32);
  f32x4 r = (f32x4)(0.f);
  if (g == 0) { r[0] = e0; r[1] = e1; }
  return r;
}

// The weight images of a block: [out][in] rows of kS floats, zero padded (W1, W2: 64 rows -- W1 may be cut to W1ROWS --,
// W3: 16 rows).  All loads of a thread are issued before its first LDS store (nine L2 round trips in flight instead of
// one after the other: the prologue is most of a small batch's time).
template <int W1ROWS, int THREADS = kWaves * 64>
__device__ __forceinline__ void stage_weights(const pds_mlp &m, float *W1s, float *W2s, float *W3s, int tid) {
  constexpr int kThreads = THREADS, kIters = (kMaxDim * kS + kThreads - 1) / kThreads;
  float v1[kIters], v2[kIters], v3[kIters];
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const int i = tid + it * kThreads, r = i / kS, k = i - r * kS;
    const bool in = i < kMaxDim * kS;
    v1[it] = (in && r < W1ROWS && r < m.h1 && k < m.d_in) ? m.w1[r * m.d_in + k] : 0.f;
    v2[it] = (in && r < m.h2 && k < m.h1) ? m.w2[r * m.h1 + k] : 0.f;
    v3[it] = (i < kTW * kS && r < m.d_out && k < m.h2) ? m.w3[r * m.h2 + k] : 0.f;
  }
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const int i = tid + it * kThreads;
    if (i < W1ROWS * kS) W1s[i] = v1[it];
    if (i < kMaxDim * kS) W2s[i] = v2[it];
    if (i < kTW * kS) W3s[i] = v3[it];
  }
}

// NINB: 16-wide tiles of the input dimension beyond the first two (1: d_in <= 48, 2: d_in <= 64) -- 16
// accumulator registers that decide whether the gradient kernels fit 256 registers (two waves per SIMD)
// GB: bias gradients as per-lane partial sums (36 registers).  Otherwise they come for free out of
// the weight-gradient GEMMs: the first padding column of each [sample][feature] image (column d_in
// of X, h1 of H1, h2 of H2) is set to 1, so column d_in / h1 / h2 of dW1 / dW2 / dW3 accumulates
// sum_s dZ[s][i] -- possible whenever the dimension leaves a padding column (not a multiple of 16).
// KJI / KJH: data-carrying k-steps of the last input / hidden k-tile (see gemm_wt; 4 = generic)
template <int LOSS, int ACT, int NINB, bool GB, int KJI = 4, int KJH = 4>
__global__ __launch_bounds__(kWaves * 64, 2) void mlp_kernel(const Args a) {
  constexpr int NIN = 2 + NINB;
  constexpr bool EDGE = PDS_MLP_EDGE && LOSS != LOSS_NONE && KJH == 2;  // h1 == h2 == 50: features 48, 49 on the vector ALU
  // ---- LDS images ---------------------------------------------------------------------------------
  __shared__ __attribute__((aligned(16))) float W1s[kMaxDim * kS];  // [out][in], zero padded
  __shared__ __attribute__((aligned(16))) float W2s[kMaxDim * kS];
  __shared__ __attribute__((aligned(16))) float W3s[kTW * kS];
  __shared__ __attribute__((aligned(16))) float b1s[kMaxDim], b2s[kMaxDim], b3s[kTW], mus[kMaxDim], iss[kMaxDim];
  __shared__ float isg[kTW], lsg[kTW];
  __shared__ __attribute__((aligned(16))) float images[kWaves * kWaveFloats];
  const pds_mlp &m = a.m;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;  // C/D layout: column (sample) n, rows 4 g + q
  stage_weights<kMaxDim>(m, W1s, W2s, W3s, tid);
  if (tid < kMaxDim) {
    b1s[tid] = tid < m.h1 ? m.b1[tid] : 0.f;
    b2s[tid] = tid < m.h2 ? m.b2[tid] : 0.f;
    const bool std_on = a.mean != nullptr && tid < m.d_in;
    mus[tid] = std_on ? a.mean[tid] : 0.f;
    iss[tid] = std_on ? 1.0f / (a.stdv[tid] + a.eps) : 1.f;
  }
  if (tid < kTW) {
    b3s[tid] = tid < m.d_out ? m.b3[tid] : 0.f;
    const float ls = (LOSS == LOSS_PPO && tid < m.d_out) ? a.log_std[tid] : 0.f;
    lsg[tid] = ls;
    isg[tid] = expf(-ls);  // 1 / sigma
  }
  float *Ximg = images + wave * kWaveFloats;  // [sample][feature] images for the weight-gradient GEMMs
  float *H1img = Ximg + kTS * kS, *H2img = H1img + kTS * kS, *dYimg = H2img + kTS * kS;
  for (int i = lane; i < kWaveFloats; i += 64) Ximg[i] = 0.f;
  __syncthreads();

  // weight-gradient accumulators of this wave (over all of its tiles), C/D layout; bias gradients as
  // per-lane partial sums over this lane's sample column
  f32x4 gW1[kNT][NIN], gW2[kNT][kNT], gW3[kNT], gb1[GB ? kNT : 1], gb2[GB ? kNT : 1], gb3 = (f32x4)(0.f);
  float st_loss = 0.f, st_ratio = 0.f, st_kl = 0.f, st_cnt = 0.f;
#pragma unroll
  for (int i = 0; i < kNT; ++i) {
    gW3[i] = (f32x4)(0.f);
    if (GB) { gb1[i] = (f32x4)(0.f); gb2[i] = (f32x4)(0.f); }
#pragma unroll
    for (int j = 0; j < kNT; ++j) gW2[i][j] = (f32x4)(0.f);
#pragma unroll
    for (int j = 0; j < NIN; ++j) gW1[i][j] = (f32x4)(0.f);
  }

  const long long ntiles = (a.B + kTS - 1) / kTS;
  const long long wid = (long long)blockIdx.x * kWaves + wave, nw = (long long)gridDim.x * kWaves;
  auto load_index = [&](long long tt) -> long long {  // source row of this lane's sample, -1: none
    const long long s = tt * kTS + n;
    if (tt >= ntiles || s >= a.B) return -1;
    return a.index != nullptr ? a.index[s] : s;
  };
  long long row_next = load_index(wid);
  for (long long t = wid; t < ntiles; t += nw) {
    const long long s0 = t * kTS;
    const long long row = row_next;
    const bool valid = row >= 0;
    row_next = load_index(t + nw);
    // ---- input: lane (n, g) holds features 16 kt + 4 g + q of its sample = the B operands of layer 1 --
    f32x4 xin[NIN];
#pragma unroll
    for (int kt = 0; kt < NIN; ++kt) {
      const int k0 = kt * kTW + 4 * g;
      const f32x4 mu = lds4(mus + k0), is = lds4(iss + k0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v = (valid && k0 + q < m.d_in) ? a.x[row * m.d_in + k0 + q] : mu[q];
        xin[kt][q] = (v - mu[q]) * is[q];
      }
    }
    // loss inputs, consumed after the forward GEMMs
    float c_act[4] = {0.f, 0.f, 0.f, 0.f}, c_adv = 0.f, c_old = 0.f, c_tgt = 0.f;
    if (LOSS == LOSS_PPO && valid) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (4 * g + q < m.d_out) c_act[q] = a.act[(s0 + n) * m.d_out + 4 * g + q];
      c_adv = a.adv[s0 + n]; c_old = a.logp_old[s0 + n];
    }
    if (LOSS == LOSS_MSE && valid) c_tgt = a.target[row];
    if (LOSS != LOSS_NONE) {
#pragma unroll
      for (int kt = 0; kt < NIN; ++kt) {
        f32x4 v = xin[kt];
        if (!GB) {
#pragma unroll
          for (int q = 0; q < 4; ++q) if (kt * kTW + 4 * g + q == m.d_in) v[q] = 1.f;
        }
        sts4(Ximg + n * kS + kt * kTW + 4 * g, v);
      }
    }
    // ---- forward: activations stay in registers from layer to layer -------------------------------
    f32x4 h1r[kNT], h2r[kNT];
    f32x4 cc[kNT];
    if constexpr (EDGE) {
      gemm_wt2<NIN, KJI>(W1s, 0, xin, n, g, cc[0], cc[1]);
      cc[2] = gemm_wt<NIN, KJI>(W1s, 2, xin, n, g);
      cc[3] = edge_rows<NIN>(W1s, xin, g);
    } else {
#pragma unroll
      for (int it = 0; it < kNT; it += 2) gemm_wt2<NIN, KJI>(W1s, it, xin, n, g, cc[it], cc[it + 1]);
    }
#pragma unroll
    for (int it = 0; it < kNT; ++it) {  // H1^T = act(W1 X^T + b1); rows >= h1: act(0) = 0
      const f32x4 c = cc[it];
      const f32x4 b = lds4(b1s + it * kTW + 4 * g);
#pragma unroll
      for (int q = 0; q < 4; ++q) h1r[it][q] = act_fn<ACT>(c[q] + b[q]);
      if (LOSS != LOSS_NONE) {
        f32x4 v = h1r[it];
        if (!GB) {
#pragma unroll
          for (int q = 0; q < 4; ++q) if (it * kTW + 4 * g + q == m.h1) v[q] = 1.f;
        }
        sts4(H1img + n * kS + it * kTW + 4 * g, v);
      }
    }
    if constexpr (EDGE) {
      gemm_wt2<kNT, KJH>(W2s, 0, h1r, n, g, cc[0], cc[1]);
      cc[2] = gemm_wt<kNT, KJH>(W2s, 2, h1r, n, g);
      cc[3] = edge_rows<kNT>(W2s, h1r, g);
    } else {
#pragma unroll
      for (int it = 0; it < kNT; it += 2) gemm_wt2<kNT, KJH>(W2s, it, h1r, n, g, cc[it], cc[it + 1]);
    }
#pragma unroll
    for (int it = 0; it < kNT; ++it) {  // H2^T = act(W2 H1^T + b2)
      const f32x4 c = cc[it];
      const f32x4 b = lds4(b2s + it * kTW + 4 * g);
#pragma unroll
      for (int q = 0; q < 4; ++q) h2r[it][q] = act_fn<ACT>(c[q] + b[q]);
      if (LOSS != LOSS_NONE) {
        f32x4 v = h2r[it];
        if (!GB) {
#pragma unroll
          for (int q = 0; q < 4; ++q) if (it * kTW + 4 * g + q == m.h2) v[q] = 1.f;
        }
        sts4(H2img + n * kS + it * kTW + 4 * g, v);
      }
    }
    f32x4 y;  // Y^T = W3 H2^T + b3: lane (n, g) holds outputs 4 g + q of sample n (rows >= d_out: 0)
    {
      const f32x4 c = gemm_wt<kNT, KJH>(W3s, 0, h2r, n, g);
      const f32x4 b = lds4(b3s + 4 * g);
      y = c + b;
    }
    if (LOSS == LOSS_NONE) {
      if (valid) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * g + q < m.d_out) a.y[(s0 + n) * m.d_out + 4 * g + q] = y[q];
      }
      continue;
    }

    // ---- loss and its gradient with respect to the network output -------------------------------
    f32x4 dy = (f32x4)(0.f);  // dY^T: lane (n, g) holds d loss / d output (4 g + q) of sample n
    if (LOSS == LOSS_PPO) {
      // compute_loss_pi, algs/ppo/ppo.py:22-40 (Normal(mu, sigma).log_prob(act).sum(-1)); every lane
      // of a sample column evaluates the scalar part, the 4 lane groups share the outputs
      float lp = 0.f, kl = 0.f;
      f32x4 zs = (f32x4)(0.f);  // z / sigma
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = 4 * g + q;
        if (j < m.d_out) {
          const float z = (c_act[q] - y[q]) * isg[j];
          lp += -0.5f * z * z - lsg[j] - 0.91893853320467274178f;
          kl += 0.5f * z * z;
          zs[q] = z * isg[j];
        }
      }
      lp += __shfl_xor(lp, 16); lp += __shfl_xor(lp, 32);
      kl += __shfl_xor(kl, 16); kl += __shfl_xor(kl, 32);
      const float ratio = expf(lp - c_old);
      const float lo = 1.f - a.clip, hi = 1.f + a.clip;
      const float obj = fminf(ratio * c_adv, fminf(fmaxf(ratio, lo), hi) * c_adv);
      // d min(r A, clip(r) A) / d r  (torch.min splits ties, clamp passes its range: net A inside
      // the range, A outside it only on the un-clipped branch)
      const bool cut = (c_adv > 0.f && ratio > hi) || (c_adv < 0.f && ratio < lo);
      const float gcoef = (cut || !valid) ? 0.f : -c_adv * ratio;  // d(-obj)/d logp
      dy = gcoef * zs;
      if (valid && g == 0) { st_loss += -obj; st_ratio += ratio; st_kl += kl; st_cnt += 1.f; }
    } else {
      // compute_loss_v: mse_loss(v(obs), target_v), algs/iwpg/iwpg.py:272-275
      if (valid && g == 0) {
        const float d = y[0] - c_tgt;
        st_loss += d * d; st_cnt += 1.f;
        dy[0] = 2.f * d;
      }
    }
    if (GB) gb3 += dy;
    sts4(dYimg + n * kSY + 4 * g, dy);
    PDS_WAVE_SYNC();

    // ---- backward.  Weight-gradient GEMMs take K = the tile's 16 samples: k-slot (j, h) carries
    // sample 4 h + j, both operands are dword reads of [sample][feature] images (conflict free). -----
    const int r = n, h = g;  // A-operand lane roles
    {  // dW3 += dY^T H2 (rows = outputs)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float av = dYimg[(4 * h + j) * kSY + r];
#pragma unroll
        for (int jt = 0; jt < kNT; ++jt) gW3[jt] = PDS_MFMA(av, H2img[(4 * h + j) * kS + jt * kTW + n], gW3[jt]);
      }
    }
    // dZ2^T = (W3^T dY^T) * act'(H2^T); the k-slot (j, h) carries output 4 h + j = register j of dy
    f32x4 dz2[kNT];
#pragma unroll
    for (int it = 0; it < kNT; ++it) cc[it] = (f32x4)(0.f);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int it = 0; it < kNT; ++it) cc[it] = PDS_MFMA(W3s[(4 * h + j) * kS + it * kTW + r], dy[j], cc[it]);
#pragma unroll
    for (int it = 0; it < kNT; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) dz2[it][q] = cc[it][q] * act_grad<ACT>(h2r[it][q]);
      if (GB) gb2[it] += dz2[it];
    }
#pragma unroll
    for (int it = 0; it < kNT; ++it) sts4(H2img + n * kS + it * kTW + 4 * g, dz2[it]);  // after the dW3 reads (in order)
    PDS_WAVE_SYNC();
    // dW2 += dZ2^T H1
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float av[kNT], bv[kNT];
#pragma unroll
      for (int i = 0; i < kNT; ++i) {
        av[i] = H2img[(4 * h + j) * kS + i * kTW + r];
        bv[i] = H1img[(4 * h + j) * kS + i * kTW + n];
      }
#pragma unroll
      for (int it = 0; it < kNT; ++it)
#pragma unroll
        for (int jt = 0; jt < kNT; ++jt) gW2[it][jt] = PDS_MFMA(av[it], bv[jt], gW2[it][jt]);
    }
    // dZ1^T = (W2^T dZ2^T) * act'(H1^T): A = W2^T read column-wise (4 dwords per k-tile)
    f32x4 dz1[kNT];
#pragma unroll
    for (int jt = 0; jt < kNT; ++jt) cc[jt] = (f32x4)(0.f);
#pragma unroll
    for (int kt = 0; kt < kNT; ++kt)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int jt = 0; jt < (EDGE ? kNT - 1 : kNT); ++jt)
          if (kt < kNT - 1 || j < KJH) cc[jt] = PDS_MFMA(W2s[(kt * kTW + 4 * h + j) * kS + jt * kTW + r], dz2[kt][j], cc[jt]);
    if constexpr (EDGE) cc[kNT - 1] = edge_cols(W2s, dz2, g);
#pragma unroll
    for (int jt = 0; jt < kNT; ++jt) {
      const f32x4 c = cc[jt];
      const f32x4 hv = lds4(H1img + n * kS + jt * kTW + 4 * g);  // this lane's own H1 values
#pragma unroll
      for (int q = 0; q < 4; ++q) dz1[jt][q] = c[q] * act_grad<ACT>(hv[q]);
      if (GB) gb1[jt] += dz1[jt];
    }
#pragma unroll
    for (int jt = 0; jt < kNT; ++jt) sts4(H1img + n * kS + jt * kTW + 4 * g, dz1[jt]);  // after the dW2 reads
    PDS_WAVE_SYNC();
    // dW1 += dZ1^T X
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float av[kNT], bv[NIN];
#pragma unroll
      for (int i = 0; i < kNT; ++i) av[i] = H1img[(4 * h + j) * kS + i * kTW + r];
#pragma unroll
      for (int i = 0; i < NIN; ++i) bv[i] = Ximg[(4 * h + j) * kS + i * kTW + n];
#pragma unroll
      for (int it = 0; it < kNT; ++it)
#pragma unroll
        for (int kt = 0; kt < NIN; ++kt) gW1[it][kt] = PDS_MFMA(av[it], bv[kt], gW1[it][kt]);
    }
    PDS_WAVE_SYNC();  // the images are rewritten by the next tile
  }

  if (LOSS == LOSS_NONE) return;
  // ---- the block's 8 waves add their accumulators up through LDS (the images are free now): rounds
  // 6->2 7->3, 4->0 5->1, 2->0 3->1, 1->0, two register images in flight per round (an image is
  // up to 184 floats per lane).  One partial per BLOCK instead of one per wave leaves the second
  // kernel 8 x less to read.
  {
    float *red = images;
    constexpr int kRegs = 4 * (kNT * NIN + kNT * kNT + kNT + (GB ? 2 * kNT + 1 : 0)) + kStats;
    static_assert(2 * kRegs * 64 <= kWaves * kWaveFloats, "two register images must fit in the tile images");
    auto xfer = [&](float *slot, bool add) {
      int r = 0;  // 16-byte slots: one ds_read / ds_write_b128 per 4 registers, consecutive lanes 16 B apart
      auto four = [&](f32x4 &v) {
        float *q = slot + (r * 64 + lane) * 4;
        if (add) v += lds4(q); else sts4(q, v);
        ++r;
      };
#pragma unroll
      for (int i = 0; i < kNT; ++i) {
#pragma unroll
        for (int j = 0; j < NIN; ++j) four(gW1[i][j]);
#pragma unroll
        for (int j = 0; j < kNT; ++j) four(gW2[i][j]);
        four(gW3[i]);
        if (GB) { four(gb1[i]); four(gb2[i]); }
      }
      if (GB) four(gb3);
      f32x4 st = {st_loss, st_ratio, st_kl, st_cnt};
      four(st);
      st_loss = st[0]; st_ratio = st[1]; st_kl = st[2]; st_cnt = st[3];
    };
    __syncthreads();
#pragma unroll
    for (int round = 0; round < 4; ++round) {
      const int src0 = round == 0 ? 6 : (round == 1 ? 4 : (round == 2 ? 2 : 1));
      const int nsrc = round == 3 ? 1 : 2;
      const int dst0 = round == 0 ? 2 : 0;
      if (wave >= src0 && wave < src0 + nsrc) xfer(red + (wave - src0) * kRegs * 64, false);
      __syncthreads();
      if (wave >= dst0 && wave < dst0 + nsrc) xfer(red + (wave - dst0) * kRegs * 64, true);
      __syncthreads();
    }
    if (wave != 0) return;
  }
  // ---- the block's partial sums -> partials[block][...] (flat parameter layout + statistics) ----------
  float *out = a.partials + (long long)blockIdx.x * a.pstride;
  const Offsets o = offsets(m);
#pragma unroll
  for (int it = 0; it < kNT; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = it * kTW + 4 * g + q;  // row of the C/D layout
#pragma unroll
      for (int jt = 0; jt < kNT; ++jt) {
        const int j = jt * kTW + n;
        if (jt < NIN && i < m.h1 && j < m.d_in) out[o.w1 + i * m.d_in + j] = gW1[it][jt < NIN ? jt : 0][q];
        if (i < m.h2 && j < m.h1) out[o.w2 + i * m.h1 + j] = gW2[it][jt][q];
        if (!GB) {  // the ones columns
          if (jt < NIN && i < m.h1 && j == m.d_in) out[o.b1 + i] = gW1[it][jt < NIN ? jt : 0][q];
          if (i < m.h2 && j == m.h1) out[o.b2 + i] = gW2[it][jt][q];
        }
      }
      if (GB) {  // sum of the per-lane partials over the 16 sample columns of the lane group
        float v1 = gb1[it][q], v2 = gb2[it][q];
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) { v1 += __shfl_xor(v1, d); v2 += __shfl_xor(v2, d); }
        if (n == 0) {
          if (i < m.h1) out[o.b1 + i] = v1;
          if (i < m.h2) out[o.b2 + i] = v2;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = 4 * g + q;
#pragma unroll
    for (int jt = 0; jt < kNT; ++jt) {
      const int j = jt * kTW + n;
      if (i < m.d_out && j < m.h2) out[o.w3 + i * m.h2 + j] = gW3[jt][q];
      if (!GB && i < m.d_out && j == m.h2) out[o.b3 + i] = gW3[jt][q];
    }
    if (GB) {
      float v3 = gb3[q];
#pragma unroll
      for (int d = 8; d >= 1; d >>= 1) v3 += __shfl_xor(v3, d);
      if (n == 0 && i < m.d_out) out[o.b3 + i] = v3;
    }
  }
  // statistics: lanes of group 0 hold per-sample sums
  float s4[kStats] = {st_loss, st_ratio, st_kl, st_cnt};
#pragma unroll
  for (int q = 0; q < kStats; ++q) {
    float v = s4[q];
    for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0) out[o.total + q] = v;
  }
}

// ---- round 3: the PPO gradient of the reference's default policy net (50-50 relu, d_in < 48), wave roles ---------
// mlp_kernel keeps 128 weight-gradient accumulator registers live through the forward and backward GEMMs of every
// tile; at two waves per SIMD (256 registers each) that leaves the compiler no room to fetch LDS operands ahead of
// the MFMAs that use them -- its schedule is read -> wait -> 2 MFMAs, the waves sit in s_waitcnt for ~20 % of their
// cycles and the matrix pipe idles a third of the time (profiles/r01_mlp_bench.txt: SQ_VALU_MFMA_BUSY 67 %).
// Here the two waves of a SIMD take different ROLES on the same tile stream instead of different tiles:
//   F (waves 0-3): forward, loss, dZ2 and the small dW3                -- no large accumulators, 112 MFMAs per tile
//                    plus all the vector-ALU work of the activations and of the loss;
//   G (waves 4-7): dZ1, dW2 += dZ2^T H1 and dW1 += dZ1^T X           -- 112 accumulator registers, 154 MFMAs per tile.
// F hands a tile to its G through a set of three [16 samples][52] LDS images (X, H1, dZ2); two sets per pair, so F
// works on tile k + 1 while G consumes tile k.  Hand-over: monotonic counters in LDS (full / empty per set), release
// fence + store after the last image write, acquire load in a sleep loop -- the 8 waves of the block are resident
// together (one block per CU), so the wait cannot deadlock.  Measured alone (profiling builds, PDS_SPLIT_DEBUG) F needs
// ~2 x the cycles of its MFMAs (epilogues, loss, stores), G ~1.15 x: hence dZ1 on G's side.
constexpr int kSI = 52;                       // image row stride: 50 features + ones column, 4 * kSI == 16 (mod 32)
constexpr int kPairs = kWaves / 2;
constexpr int kSetFloats = 3 * kTS * kSI;     // X, H1, dZ2
constexpr int kPrivFloats = kTS * kSI + kTS * kSY;  // F's own H2 and dY images
constexpr int kPrivGFloats = kTS * kSI;             // G's own dZ1 image
constexpr int kW1Rows = 50;

__device__ __forceinline__ void wait_ge(int *flag, int need) {
  while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < need)
    __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void signal(int *flag, int value, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");  // every lane's image accesses are complete
  if (lane == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#if PDS_SPLIT_STAMPS  // profiling build: s_memtime at the phase boundaries of one tile of pair 0 of block 0
__device__ unsigned long long g_split_stamps[32];
#define PDS_SSTAMP(role, i)                                                                   \
  do {                                                                                        \
    if (k == 10 && blockIdx.x == 0 && pair == 0) {                                            \
      const unsigned long long tt_ = __builtin_amdgcn_s_memtime();                            \
      if (lane == 0) g_split_stamps[(role) * 16 + (i)] = tt_;                                 \
    }                                                                                         \
  } while (0)
#else
#define PDS_SSTAMP(role, i) do { } while (0)
#endif

// relu in ONE instruction: fmaxf() costs two (LLVM quiets a possible signalling NaN with v_max x, x first); the
// hardware's v_max_f32 returns the non-NaN operand, i.e. 0 for a NaN input either way
__device__ __forceinline__ float relu1(float v) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
  return r;
}
template <int ACT>
__device__ __forceinline__ float act_fast(float v) { return ACT == 0 ? relu1(v) : act_fn<ACT>(v); }

// gradient through the activation, given its OUTPUT h: relu as a select (one instruction less than c * (0 or 1))
template <int ACT>
__device__ __forceinline__ float act_back(float c, float h) { return ACT == 0 ? (h > 0.f ? c : 0.f) : c * (1.f - h * h); }

// rows 48, 49 of a 50-wide layer on the vector ALU (see edge_rows), bias included: even / odd feature slots accumulate
// in the two halves of v_pk_fma_f32 -- the operand pairs are adjacent registers, no moves.  wp: row 48 at this lane
// group's column offset; result in the C/D layout of tile 3 (lane group 0: features 48, 49, then zeros).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NK>
__device__ __forceinline__ f32x4 edge_pair(const float *wp, const f32x4 (&in)[NK], const float *bias, int g) {
  f32x2 s0 = (f32x2)(0.f), s1 = (f32x2)(0.f);
#pragma unroll
  for (int kt = 0; kt < NK; ++kt) {
    const f32x4 w0 = lds4(wp + kt * kTW), w1 = lds4(wp + kS + kt * kTW);
    s0 = __builtin_elementwise_fma(w0.xy, in[kt].xy, s0);
    s1 = __builtin_elementwise_fma(w1.xy, in[kt].xy, s1);
    s0 = __builtin_elementwise_fma(w0.zw, in[kt].zw, s0);
    s1 = __builtin_elementwise_fma(w1.zw, in[kt].zw, s1);
  }
  float v0 = s0.x + s0.y, v1 = s1.x + s1.y;
  v0 += __shfl_xor(v0, 16); v1 += __shfl_xor(v1, 16);
  v0 += __shfl_xor(v0, 32); v1 += __shfl_xor(v1, 32);
  f32x4 r = (f32x4)(0.f);
  if (g == 0) { r[0] = v0 + bias[0]; r[1] = v1 + bias[1]; }
  return r;
}

// NG = 2 (not the default: slower, see PDS_MLP_SPLIT): the weight-gradient role split once more -- G1 (waves 4-7: dZ1,
// dW1), G2 (waves 8-11: dW2) -- three waves per SIMD (12 per block, <= 168 registers each).
template <int KJI, int NG, bool BFP = false>
__global__ __launch_bounds__((1 + NG) * 256, 1 + NG) void ppo_split_kernel(const Args a) {
  constexpr int NIN = 3, KJH = 2, ACT = 0;
  constexpr int kThreads = (1 + NG) * 256;
  // L1B (PDS_SPLIT_BF16_L1): layer 1 of the forward role on the bf16 instruction -- W1 as an image of its A operands in pieces
  // ([row tile][k step][piece][lane] x 16 bytes, 18 KB) INSTEAD of the f32 image (13.6 KB), of which only rows 48, 49 remain
  constexpr bool L1B = BFP && PDS_SPLIT_BF16 != 0 && PDS_SPLIT_BF16_L1 != 0 && PDS_SPLIT_BF16_L2 != 0 && PDS_SPLIT_WRES != 0 && PDS_MLP_STRIPS != 0 && NG == 1;
  __shared__ __attribute__((aligned(16))) float W1s[(L1B ? 2 : kW1Rows) * kS];  // rows 48, 49: the vector-ALU features
  __shared__ __attribute__((aligned(16))) uint32_t W1b[L1B ? (kNT - 1) * 2 * 3 * 64 * 4 : 4];
  __shared__ __attribute__((aligned(16))) float W2s[kMaxDim * kS];
  __shared__ __attribute__((aligned(16))) float W3s[kTW * kS];
  __shared__ __attribute__((aligned(16))) float b1s[kMaxDim], b2s[kMaxDim], b3s[kTW];
  __shared__ __attribute__((aligned(16))) float isg[kTW], lsg[kTW];
  __shared__ __attribute__((aligned(16))) float sets[kPairs * 2 * kSetFloats];
  __shared__ __attribute__((aligned(16))) float priv[kPairs * kPrivFloats];
  __shared__ __attribute__((aligned(16))) float privg[kPairs * kPrivGFloats];
  __shared__ int flags[kPairs * 8];  // per pair: full[2], empty[2] (G / G1), empty2[2] (G2)
  const pds_mlp &m = a.m;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Which waves share a SIMD: w and w + 4 (profiles/r03_mlp_microbench.txt).  PDS_SPLIT_SIMD_ROLES = 0 (rounds 3-4): F_p = wave p,
  // G_p = wave p + 4 -- every SIMD hosts one F and one G.  1: the two F waves of pairs (p, p + 2) on one SIMD and their G waves
  // on another -- SIMDs 0, 1 run forward roles only, SIMDs 2, 3 weight-gradient roles only (NG == 1).  Same tiles per pair and the
  // same reduction order by pair index: bit-identical results either way.
#if PDS_SPLIT_SIMD_ROLES
  const int pair = NG == 1 ? ((wave & 1) | ((wave >> 2) << 1)) : (wave & 3);
  const int role = NG == 1 ? ((wave >> 1) & 1) : (wave >> 2);
#else
  const int pair = wave & 3;
  const int role = wave >> 2;  // 0: F, 1: G (NG = 1) or G1, 2: G2
#endif
  const int n = lane & 15, g = lane >> 4;
  stage_weights<L1B ? 0 : kW1Rows, kThreads>(m, W1s, W2s, W3s, tid);
  if constexpr (L1B) {
    for (int i = tid; i < 2 * kS; i += kThreads) {
      const int rr = 48 + i / kS, k = i % kS;
      W1s[i] = (rr < m.h1 && k < m.d_in) ? m.w1[rr * m.d_in + k] : 0.f;
    }
    for (int e = tid; e < (kNT - 1) * 2 * 64; e += kThreads) {  // slot (lane group gg, i) of step ks: input feature 32 ks + 4 gg + i (i < 4), + 16 (i >= 4)
      const int it = e >> 7, ks = (e >> 6) & 1, l = e & 63, row = 16 * it + (l & 15), gg = l >> 4;
      f32x4 w0, w1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k0 = 32 * ks + 4 * gg + q, k1 = k0 + 16;
        w0[q] = (row < m.h1 && k0 < m.d_in) ? m.w1[row * m.d_in + k0] : 0.f;
        w1[q] = (row < m.h1 && k1 < m.d_in) ? m.w1[row * m.d_in + k1] : 0.f;
      }
      const Oct3 o3 = oct3(split4(w0), split4(w1));
      uint32_t *dst = W1b + (((it * 2 + ks) * 3) * 64 + l) * 4;
      *reinterpret_cast<bf16x8_ *>(dst) = o3.hi;
      *reinterpret_cast<bf16x8_ *>(dst + 64 * 4) = o3.mid;
      *reinterpret_cast<bf16x8_ *>(dst + 2 * 64 * 4) = o3.lo;
    }
  }
  if (tid < kMaxDim) {
    b1s[tid] = tid < m.h1 ? m.b1[tid] : 0.f;
    b2s[tid] = tid < m.h2 ? m.b2[tid] : 0.f;
  }
  if (tid < kTW) {
    b3s[tid] = tid < m.d_out ? m.b3[tid] : 0.f;
    const float ls = tid < m.d_out ? a.log_std[tid] : 0.f;
    lsg[tid] = ls;
    isg[tid] = expf(-ls);
  }
  if (tid < kPairs * 8) flags[tid] = 0;
  __syncthreads();

  const long long ntiles = (a.B + kTS - 1) / kTS;
  const long long pid = (long long)blockIdx.x * kPairs + pair, np = (long long)gridDim.x * kPairs;
  int *full = flags + pair * 8, *empty = full + 2, *empty2 = full + 4;
  float *pset = sets + pair * 2 * kSetFloats;
  const int r = n, h = g;  // A-operand lane roles of the weight-gradient GEMMs (k-slot (j, h) = sample 4 h + j)
  const int n3 = min(n, 3);  // column tile 3 of a 52-wide image holds columns 48..51 only

  f32x4 gW1[kNT][NIN], gW2[kNT][kNT], gW3[kNT];
  float st_loss = 0.f, st_ratio = 0.f, st_kl = 0.f, st_cnt = 0.f;
#pragma unroll
  for (int i = 0; i < kNT; ++i) {
    gW3[i] = (f32x4)(0.f);
#pragma unroll
    for (int j = 0; j < kNT; ++j) gW2[i][j] = (f32x4)(0.f);
#pragma unroll
    for (int j = 0; j < NIN; ++j) gW1[i][j] = (f32x4)(0.f);
  }

  // Round 4 (NG = 1): the edge of the 50-wide layers as 4x4x1 strips.  50 = 3 x 16 + 2, so row tile 3 / column tile 3 of
  // dW2 and row tile 3 of dW1 were 16-wide tiles with 2 (3 with the bias column) useful rows or columns: 7 of dW2's 16
  // tiles and 3 of dW1's 12, 40 of G's 154 MFMAs per 16 samples.  Now: gW2r = rows 48..51 x all columns (lane l: column
  // l), gW2c = all rows x columns 48..51 (lane l: rows 4 (l / 4) .. + 3, column 48 + l % 4), gW1r = rows 48..51 of dW1
  // x all input columns; 16 rank-1 updates (one per sample) of 8 cycles each per strip instead of 4 k-steps x 32 cycles
  // per tile.  The full tiles cover rows / columns 0..47.  dW3 (4 x 50) is one such strip on F's side (gW3s).
  constexpr bool STRIP = PDS_MLP_STRIPS != 0 && NG == 1;
  constexpr int NTF = STRIP ? kNT - 1 : kNT;  // full 16-wide tiles per hidden dimension
  constexpr bool BF = BFP && PDS_SPLIT_BF16 != 0 && STRIP;  // the weight-gradient role's full-tile GEMMs on v_mfma_f32_16x16x32_bf16 (three pieces per operand)
  f32x4 gW2r = (f32x4)(0.f), gW2c = (f32x4)(0.f), gW1r = (f32x4)(0.f), gW3s = (f32x4)(0.f), gW3s2 = (f32x4)(0.f);
  const int lc = lane < kSI ? lane : kSI - 1;  // column of a 52-wide image row (lanes 52..63: clamped, their results are dropped)

  if (NG == 2 && role == 2) {
    // ================= G2 (NG = 2): dW2 += dZ2^T H1 =========================================================
    const int r3 = min(r, 3);
    int k = 0;
    for (long long t = pid; t < ntiles; t += np, ++k) {
      const int s = k & 1;
      const float *H1img = pset + s * kSetFloats + kTS * kSI, *dZ2img = H1img + kTS * kSI;
      wait_ge(full + s, (k >> 1) + 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float av[kNT], bv[kNT];
        const int row = (4 * h + j) * kSI;
#pragma unroll
        for (int i = 0; i < kNT; ++i) {
          av[i] = dZ2img[row + i * kTW + (i == kNT - 1 ? r3 : r)];
          bv[i] = H1img[row + i * kTW + (i == kNT - 1 ? n3 : n)];
        }
#pragma unroll
        for (int it = 0; it < kNT; ++it)
#pragma unroll
          for (int jt = 0; jt < kNT; ++jt) gW2[it][jt] = PDS_MFMA(av[it], bv[jt], gW2[it][jt]);
      }
      signal(empty2 + s, (k >> 1) + 1, lane);
    }
  } else if (role == 1) {
    // ================= G (NG = 1) / G1: dZ1 and the weight-gradient GEMMs =========================================
    __builtin_amdgcn_s_setprio(PDS_SPLIT_GPRIO);
    float *dZ1img = privg + pair * kPrivGFloats;
    float wz1[kNT][4][kNT - 1];  // W2^T read column-wise: tile invariant, kept in registers
    // BF (PDS_SPLIT_BF16): the same operand for the K = 32 instruction, in three bf16 pieces.  The k index of an MFMA is a dummy
    // index, so its slots can be assigned freely as long as A and B agree: slot (lane group h, i) of step ks carries h2 feature
    // 16 (2 ks) + 4 h + i for i < 4 and 16 (2 ks + 1) + 4 h + (i - 4) for i >= 4 -- the B operand is then the lane's own dZ2
    // registers of two feature tiles (the C/D layout), exactly as in the f32 form; no cross-lane movement.
    // Tile invariant, kept in registers like wz1 (72 instead of 48; an LDS image of it would be 18 KB the block does not have).
    Oct3 wzb[2][kNT - 1];
    if constexpr (BF) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jt = 0; jt < kNT - 1; ++jt) {
          f32x4 w0, w1;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            w0[q] = W2s[(32 * ks + 4 * h + q) * kS + jt * kTW + r];
            w1[q] = W2s[(32 * ks + 16 + 4 * h + q) * kS + jt * kTW + r];
          }
          wzb[ks][jt] = oct3(split4(w0), split4(w1));
        }
    } else {
#pragma unroll
      for (int kt = 0; kt < kNT; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int jt = 0; jt < kNT - 1; ++jt)
            wz1[kt][j][jt] = (kt < kNT - 1 || j < KJH) ? W2s[(kt * kTW + 4 * h + j) * kS + jt * kTW + r] : 0.f;
    }
    const int r3 = min(r, 3);
    int k = 0;
    for (long long t = pid; t < ntiles; t += np, ++k) {
      const int s = k & 1;
      const float *Ximg = pset + s * kSetFloats, *H1img = Ximg + kTS * kSI, *dZ2img = H1img + kTS * kSI;
      PDS_SSTAMP(1, 0);
      wait_ge(full + s, (k >> 1) + 1);
      PDS_SSTAMP(1, 1);
#if PDS_SPLIT_DEBUG != 1  // profiling: 1 = G only acknowledges (F's own rate), 2 = F only signals (G's own rate)
      // this lane's dZ2 values (the B operands of dZ1) and columns 48, 49 of W2 (its vector-ALU part)
      f32x4 dz2[kNT];
#pragma unroll
      for (int kt = 0; kt < kNT; ++kt)
        dz2[kt] = (kt < kNT - 1 || g == 0) ? lds4(dZ2img + n * kSI + kt * kTW + 4 * g) : (f32x4)(0.f);
      float2 we[kNT][4];
#pragma unroll
      for (int kt = 0; kt < kNT; ++kt)
#pragma unroll
        for (int q = 0; q < 4; ++q) we[kt][q] = *reinterpret_cast<const float2 *>(W2s + (kt * kTW + 4 * g + q) * kS + 48);
      // ---- dZ1^T = (W2^T dZ2^T) * act'(H1^T): three chains ----
      f32x4 cc[kNT];
      if constexpr (BF) {
        Oct3 bz[2];
        bz[0] = oct3(split4(dz2[0]), split4(dz2[1]));
        bz[1] = oct3(split4(dz2[2]), split4(dz2[3]));
#pragma unroll
        for (int jt = 0; jt < kNT - 1; ++jt) cc[jt] = (f32x4)(0.f);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {  // six products per step, smallest first; the three chains alternate
#pragma unroll
          for (int jt = 0; jt < kNT - 1; ++jt) cc[jt] = PDS_MFMA_BF(wzb[ks][jt].mid, bz[ks].mid, cc[jt]);
#pragma unroll
          for (int jt = 0; jt < kNT - 1; ++jt) cc[jt] = PDS_MFMA_BF(wzb[ks][jt].hi, bz[ks].lo, cc[jt]);
#pragma unroll
          for (int jt = 0; jt < kNT - 1; ++jt) cc[jt] = PDS_MFMA_BF(wzb[ks][jt].lo, bz[ks].hi, cc[jt]);
#pragma unroll
          for (int jt = 0; jt < kNT - 1; ++jt) cc[jt] = PDS_MFMA_BF(wzb[ks][jt].hi, bz[ks].mid, cc[jt]);
#pragma unroll
          for (int jt = 0; jt < kNT - 1; ++jt) cc[jt] = PDS_MFMA_BF(wzb[ks][jt].mid, bz[ks].hi, cc[jt]);
#pragma unroll
          for (int jt = 0; jt < kNT - 1; ++jt) cc[jt] = PDS_MFMA_BF(wzb[ks][jt].hi, bz[ks].hi, cc[jt]);
        }
      } else {
#pragma unroll
        for (int kt = 0; kt < kNT; ++kt)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int jt = 0; jt < kNT - 1; ++jt)
              if (kt < kNT - 1 || j < KJH)
                cc[jt] = PDS_MFMA(wz1[kt][j][jt], dz2[kt][j], (kt == 0 && j == 0) ? (f32x4)(0.f) : cc[jt]);
      }
      PDS_SSTAMP(1, 2);
      // ---- dW2 += dZ2^T H1 (covers the result latency of dZ1; G2's job when there is one) ----
      // BF: K = the tile's 16 samples x TWO piece combinations -- slot (h, i < 4) = sample 4 h + i with pieces (a, b), slot
      // (h, i >= 4) = the same sample with pieces (a', b'): (hi hi | hi mid), (mid hi | hi lo), (lo hi | mid mid) are the six
      // products in three instructions.  Operands: the same four dword reads per lane and 16 x 16 block as the f32 form
      // (samples 4 h .. 4 h + 3 of one feature), split in registers.
      if constexpr (BF) {
        Quad3 qa[NTF], qb[NTF];
#pragma unroll
        for (int i = 0; i < NTF; ++i) {
          f32x4 va, vb;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            va[j] = dZ2img[(4 * h + j) * kSI + i * kTW + r];
            vb[j] = H1img[(4 * h + j) * kSI + i * kTW + n];
          }
          qa[i] = split4(va);
          qb[i] = split4(vb);
        }
#pragma unroll
        for (int it = 0; it < NTF; ++it)
#pragma unroll
          for (int jt = 0; jt < NTF; ++jt) gW2[it][jt] = PDS_MFMA_BF(cat8(qa[it].lo, qa[it].mid), cat8(qb[jt].hi, qb[jt].mid), gW2[it][jt]);
#pragma unroll
        for (int it = 0; it < NTF; ++it)
#pragma unroll
          for (int jt = 0; jt < NTF; ++jt) gW2[it][jt] = PDS_MFMA_BF(cat8(qa[it].mid, qa[it].hi), cat8(qb[jt].hi, qb[jt].lo), gW2[it][jt]);
#pragma unroll
        for (int it = 0; it < NTF; ++it)
#pragma unroll
          for (int jt = 0; jt < NTF; ++jt) gW2[it][jt] = PDS_MFMA_BF(cat8(qa[it].hi, qa[it].hi), cat8(qb[jt].hi, qb[jt].mid), gW2[it][jt]);
      }
#pragma unroll
      for (int j = 0; j < ((NG == 1 && !BF) ? 4 : 0); ++j) {
        float av[kNT], bv[kNT];
        const int row = (4 * h + j) * kSI;
#pragma unroll
        for (int i = 0; i < NTF; ++i) {
          av[i] = dZ2img[row + i * kTW + (i == kNT - 1 ? r3 : r)];
          bv[i] = H1img[row + i * kTW + (i == kNT - 1 ? n3 : n)];
        }
#pragma unroll
        for (int it = 0; it < NTF; ++it)
#pragma unroll
          for (int jt = 0; jt < NTF; ++jt) gW2[it][jt] = PDS_MFMA(av[it], bv[jt], gW2[it][jt]);
      }
      if constexpr (STRIP) {  // rows and columns 48..51 of dW2: one rank-1 update per sample and strip
#pragma unroll
        for (int sm = 0; sm < kTS; ++sm) {
          const float dzr = dZ2img[sm * kSI + 48 + (lane & 3)], dzc = dZ2img[sm * kSI + lc];
          const float h1r_ = H1img[sm * kSI + lc], h1c = H1img[sm * kSI + 48 + (lane & 3)];
          gW2r = PDS_MFMA44(dzr, h1r_, gW2r);
          gW2c = PDS_MFMA44(dzc, h1c, gW2c);
        }
      }
      PDS_SSTAMP(1, 3);
      {
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int kt = 0; kt < kNT; ++kt)
#pragma unroll
          for (int q = 0; q < 4; ++q) { v0 = fmaf(we[kt][q].x, dz2[kt][q], v0); v1 = fmaf(we[kt][q].y, dz2[kt][q], v1); }
        v0 += __shfl_xor(v0, 16); v1 += __shfl_xor(v1, 16);
        v0 += __shfl_xor(v0, 32); v1 += __shfl_xor(v1, 32);
        cc[kNT - 1] = (f32x4)(0.f);
        if (g == 0) { cc[kNT - 1][0] = v0; cc[kNT - 1][1] = v1; }
      }
#pragma unroll
      for (int jt = 0; jt < kNT; ++jt) {
        // this lane's own H1 values (column 50 of the image is the ones column: its dz1 is 0 * 1)
        const f32x4 hv = (jt < kNT - 1 || g == 0) ? lds4(H1img + n * kSI + jt * kTW + 4 * g) : (f32x4)(0.f);
        f32x4 dz1;
#pragma unroll
        for (int q = 0; q < 4; ++q) dz1[q] = act_back<ACT>(cc[jt][q], hv[q]);
        if (jt < kNT - 1 || g == 0) sts4(dZ1img + n * kSI + jt * kTW + 4 * g, dz1);
      }
      PDS_WAVE_SYNC();
      PDS_SSTAMP(1, 4);
      if constexpr (BF) {  // dW1 += dZ1^T X, as dW2 above
        Quad3 qa[NTF], qb[NIN];
#pragma unroll
        for (int i = 0; i < NTF; ++i) {
          f32x4 va;
#pragma unroll
          for (int j = 0; j < 4; ++j) va[j] = dZ1img[(4 * h + j) * kSI + i * kTW + r];
          qa[i] = split4(va);
        }
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
          f32x4 vb;
#pragma unroll
          for (int j = 0; j < 4; ++j) vb[j] = Ximg[(4 * h + j) * kSI + i * kTW + n];
          qb[i] = split4(vb);
        }
#pragma unroll
        for (int it = 0; it < NTF; ++it)
#pragma unroll
          for (int kt = 0; kt < NIN; ++kt) gW1[it][kt] = PDS_MFMA_BF(cat8(qa[it].lo, qa[it].mid), cat8(qb[kt].hi, qb[kt].mid), gW1[it][kt]);
#pragma unroll
        for (int it = 0; it < NTF; ++it)
#pragma unroll
          for (int kt = 0; kt < NIN; ++kt) gW1[it][kt] = PDS_MFMA_BF(cat8(qa[it].mid, qa[it].hi), cat8(qb[kt].hi, qb[kt].lo), gW1[it][kt]);
#pragma unroll
        for (int it = 0; it < NTF; ++it)
#pragma unroll
          for (int kt = 0; kt < NIN; ++kt) gW1[it][kt] = PDS_MFMA_BF(cat8(qa[it].hi, qa[it].hi), cat8(qb[kt].hi, qb[kt].mid), gW1[it][kt]);
      }
#pragma unroll
      for (int j = 0; j < (BF ? 0 : 4); ++j) {  // dW1 += dZ1^T X
        float av[kNT], bv[NIN];
        const int row = (4 * h + j) * kSI;
#pragma unroll
        for (int i = 0; i < NTF; ++i) av[i] = dZ1img[row + i * kTW + (i == kNT - 1 ? r3 : r)];
#pragma unroll
        for (int i = 0; i < NIN; ++i) bv[i] = Ximg[row + i * kTW + n];
#pragma unroll
        for (int it = 0; it < NTF; ++it)
#pragma unroll
          for (int kt = 0; kt < NIN; ++kt) gW1[it][kt] = PDS_MFMA(av[it], bv[kt], gW1[it][kt]);
      }
      if constexpr (STRIP) {  // rows 48..51 of dW1
#pragma unroll
        for (int sm = 0; sm < kTS; ++sm) gW1r = PDS_MFMA44(dZ1img[sm * kSI + 48 + (lane & 3)], Ximg[sm * kSI + lc], gW1r);
      }
      PDS_SSTAMP(1, 5);
#endif
      signal(empty + s, (k >> 1) + 1, lane);
      PDS_SSTAMP(1, 6);
    }
  } else {
    // ================= F: forward, loss, backward through the activations, dW3 ================================
    // Source order = issue order here (the compiler keeps it when registers allow).  F's weights are tile invariant:
    // the A operands of the three forward GEMMs (PDS_SPLIT_WRES) and W3^T (dZ2's operand) stay in registers; what is
    // left in LDS (the two vector-ALU rows, the images) is read a phase ahead of its use.
    // F's MFMAs come in bursts between vector-ALU phases; it raises its priority for the bursts (PDS_FPRIO_MFMA).
    __builtin_amdgcn_s_setprio(PDS_SPLIT_FPRIO);
    float *H2img = priv + pair * kPrivFloats, *dYimg = H2img + kTS * kSI;
    float wz2[kNT];
#pragma unroll
    for (int it = 0; it < kNT; ++it) wz2[it] = W3s[h * kS + it * kTW + r];
    auto load_x = [&](long long tt, f32x4 (&raw)[NIN]) {
      const long long s = tt * kTS + n;
      const float *xr = a.x + (s < a.B ? s : 0) * m.d_in;
#pragma unroll
      for (int kt = 0; kt < NIN; ++kt)
#pragma unroll
        for (int q = 0; q < 4; ++q) raw[kt][q] = xr[min(kt * kTW + 4 * g + q, m.d_in - 1)];
    };
    const float *w1p = W1s + n * kS + 4 * g, *w2p = W2s + n * kS + 4 * g, *w3p = W3s + n * kS + 4 * g;
    const float *e1p = W1s + (L1B ? 0 : 48) * kS + 4 * g, *e2p = W2s + 48 * kS + 4 * g;
#if PDS_SPLIT_WRES  // the forward GEMMs' weight operands are tile invariant: 100 registers instead of 25 b128 LDS reads per tile
    f32x4 a1[kNT - 1][NIN], a2[kNT - 1][kNT], a3[kNT];
#pragma unroll
    for (int kt = 0; kt < NIN; ++kt)
#pragma unroll
      for (int it = 0; it < kNT - 1; ++it) a1[it][kt] = L1B ? (f32x4)(0.f) : lds4(w1p + it * kTW * kS + kt * kTW);
#pragma unroll
    for (int kt = 0; kt < kNT; ++kt) {
#pragma unroll
      for (int it = 0; it < kNT - 1; ++it) a2[it][kt] = lds4(w2p + it * kTW * kS + kt * kTW);
      a3[kt] = lds4(w3p + kt * kTW);
    }
#endif
#if PDS_SPLIT_BF16_L2 && PDS_SPLIT_WRES  // A/B: layer 2 of the forward role on the bf16 instruction as well (k-slots as in G's dZ1)
    Oct3 a2b[kNT - 1][2];
    if constexpr (BF) {
#pragma unroll
      for (int it = 0; it < kNT - 1; ++it)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a2b[it][ks] = oct3(split4(a2[it][2 * ks]), split4(a2[it][2 * ks + 1]));
    }
#endif
    f32x4 xraw[NIN];
    load_x(pid, xraw);
    int k = 0;
    for (long long t = pid; t < ntiles; t += np, ++k) {
      const int s = k & 1;
      float *Ximg = pset + s * kSetFloats, *H1img = Ximg + kTS * kSI, *dZ2img = H1img + kTS * kSI;
#if PDS_SPLIT_DEBUG == 2
      if (k >= 2) { wait_ge(empty + s, k >> 1); if (NG == 2) wait_ge(empty2 + s, k >> 1); }
      signal(full + s, (k >> 1) + 1, lane);
      continue;
#endif
      PDS_SSTAMP(0, 0);
      const long long s0 = t * kTS;
      const bool valid = s0 + n < a.B;
      // operands of layer 1 (and of its two vector-ALU rows)
#if !PDS_SPLIT_WRES
      f32x4 a1[kNT - 1][NIN];
#pragma unroll
      for (int kt = 0; kt < NIN; ++kt)
#pragma unroll
        for (int it = 0; it < kNT - 1; ++it) a1[it][kt] = lds4(w1p + it * kTW * kS + kt * kTW);
#endif
      // no masking of the input: a sample outside the batch carries gcoef = 0 (its forward pass runs on the clamped
      // row and is discarded), and a feature slot >= d_in (a clamped re-read) meets a zero column of W1 on the way
      // forward and a discarded column of dW1 on the way back
      f32x4 xin[NIN];
#pragma unroll
      for (int kt = 0; kt < NIN; ++kt) xin[kt] = xraw[kt];
      float c_act[4] = {0.f, 0.f, 0.f, 0.f}, c_adv = 0.f, c_old = 0.f;
      if (valid) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * g + q < m.d_out) c_act[q] = a.act[(s0 + n) * m.d_out + 4 * g + q];
        c_adv = a.adv[s0 + n]; c_old = a.logp_old[s0 + n];
      }
      if (k >= 2) {  // the weight-gradient waves are done with the tile that used this set
        wait_ge(empty + s, k >> 1);
        if (NG == 2) wait_ge(empty2 + s, k >> 1);
      }
      PDS_SSTAMP(0, 1);
#pragma unroll
      for (int kt = 0; kt < NIN; ++kt) sts4(Ximg + n * kSI + kt * kTW + 4 * g, xin[kt]);
      if (g == 0) Ximg[n * kSI + m.d_in] = 1.f;  // the bias column of dW1 (after the row's b128 stores: LDS keeps a wave's order)
      // operands of layer 2: in flight during the MFMAs of layer 1
#if !PDS_SPLIT_WRES
      f32x4 a2[kNT - 1][kNT];
#pragma unroll
      for (int kt = 0; kt < kNT; ++kt)
#pragma unroll
        for (int it = 0; it < kNT - 1; ++it) a2[it][kt] = lds4(w2p + it * kTW * kS + kt * kTW);
#endif
      PDS_FPRIO_MFMA();
      // ---- layer 1: three accumulation chains alternate (tiles 0..2), each started from its bias; features 48, 49
      // on the vector ALU (packed pairs: even / odd feature slots) ----
      f32x4 h1r[kNT], h2r[kNT], cc[kNT];
#pragma unroll
      for (int it = 0; it < kNT - 1; ++it) cc[it] = lds4(b1s + it * kTW + 4 * g);
      if constexpr (L1B) {
        static_assert(NIN == 3, "two k steps: input tiles (0, 1) and (2, none)");
        Oct3 bx[2];
        bx[0] = oct3(split4(xin[0]), split4(xin[1]));
        bx[1] = oct3(split4(xin[2]), Quad3{(u32x2_)(0u), (u32x2_)(0u), (u32x2_)(0u)});
        auto w1b = [&](int it, int ks, int piece) -> bf16x8_ {  // piece 0 hi, 1 mid, 2 lo
          return *reinterpret_cast<const bf16x8_ *>(W1b + (((it * 2 + ks) * 3 + piece) * 64 + lane) * 4);
        };
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int it = 0; it < kNT - 1; ++it) cc[it] = PDS_MFMA_BF(w1b(it, ks, 1), bx[ks].mid, cc[it]);
#pragma unroll
          for (int it = 0; it < kNT - 1; ++it) cc[it] = PDS_MFMA_BF(w1b(it, ks, 0), bx[ks].lo, cc[it]);
#pragma unroll
          for (int it = 0; it < kNT - 1; ++it) cc[it] = PDS_MFMA_BF(w1b(it, ks, 2), bx[ks].hi, cc[it]);
#pragma unroll
          for (int it = 0; it < kNT - 1; ++it) cc[it] = PDS_MFMA_BF(w1b(it, ks, 0), bx[ks].mid, cc[it]);
#pragma unroll
          for (int it = 0; it < kNT - 1; ++it) cc[it] = PDS_MFMA_BF(w1b(it, ks, 1), bx[ks].hi, cc[it]);
#pragma unroll
          for (int it = 0; it < kNT - 1; ++it) cc[it] = PDS_MFMA_BF(w1b(it, ks, 0), bx[ks].hi, cc[it]);
        }
      } else {
#if PDS_SPLIT_BF16_L2 && PDS_SPLIT_WRES  // (layer 2's operand is 72 registers in pieces: layer 1's then comes from LDS per tile, 9 b128 reads)
      f32x4 a1t[kNT - 1][NIN];
#pragma unroll
      for (int kt = 0; kt < NIN; ++kt)
#pragma unroll
        for (int it = 0; it < kNT - 1; ++it) a1t[it][kt] = BF ? lds4(w1p + it * kTW * kS + kt * kTW) : a1[it][kt];
#else
      f32x4 (&a1t)[kNT - 1][NIN] = a1;
#endif
#pragma unroll
      for (int kt = 0; kt < NIN; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int it = 0; it < kNT - 1; ++it)
            if (kt < NIN - 1 || j < KJI) cc[it] = PDS_MFMA_F(a1t[it][kt][j], xin[kt][j], cc[it]);
      }
      PDS_FPRIO_VALU();
      PDS_SSTAMP(0, 2);
      load_x(t + np, xraw);  // the next tile's rows: in flight during the rest of this tile
      cc[kNT - 1] = edge_pair<NIN>(e1p, xin, b1s + 48, g);
      // operands of layer 3
#if !PDS_SPLIT_WRES
      f32x4 a3[kNT];
#pragma unroll
      for (int kt = 0; kt < kNT; ++kt) a3[kt] = lds4(w3p + kt * kTW);
#endif
#pragma unroll
      for (int it = 0; it < kNT; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) h1r[it][q] = act_fast<ACT>(cc[it][q]);
        f32x4 v = h1r[it];
        if (it == kNT - 1) v[2] = 1.f;  // column 50: the bias column of dW2
        if (it < kNT - 1 || g == 0) sts4(H1img + n * kSI + it * kTW + 4 * g, v);
      }
      PDS_SSTAMP(0, 3);
      PDS_FPRIO_MFMA();
      // ---- layer 2 ----
#pragma unroll
      for (int it = 0; it < kNT - 1; ++it) cc[it] = lds4(b2s + it * kTW + 4 * g);
#if PDS_SPLIT_BF16_L2 && PDS_SPLIT_WRES
      if constexpr (BF) {
        Oct3 bh[2];
        bh[0] = oct3(split4(h1r[0]), split4(h1r[1]));
        bh[1] = oct3(split4(h1r[2]), split4(h1r[3]));
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int it = 0; it < kNT - 1; ++it) cc[it] = PDS_MFMA_BF(a2b[it][ks].mid, bh[ks].mid, cc[it]);
#pragma unroll
          for (int it = 0; it < kNT - 1; ++it) cc[it] = PDS_MFMA_BF(a2b[it][ks].hi, bh[ks].lo, cc[it]);
#pragma unroll
          for (int it = 0; it < kNT - 1; ++it) cc[it] = PDS_MFMA_BF(a2b[it][ks].lo, bh[ks].hi, cc[it]);
#pragma unroll
          for (int it = 0; it < kNT - 1; ++it) cc[it] = PDS_MFMA_BF(a2b[it][ks].hi, bh[ks].mid, cc[it]);
#pragma unroll
          for (int it = 0; it < kNT - 1; ++it) cc[it] = PDS_MFMA_BF(a2b[it][ks].mid, bh[ks].hi, cc[it]);
#pragma unroll
          for (int it = 0; it < kNT - 1; ++it) cc[it] = PDS_MFMA_BF(a2b[it][ks].hi, bh[ks].hi, cc[it]);
        }
      } else
#endif
      {
#pragma unroll
      for (int kt = 0; kt < kNT; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int it = 0; it < kNT - 1; ++it)
            if (kt < kNT - 1 || j < KJH) cc[it] = PDS_MFMA_F(a2[it][kt][j], h1r[kt][j], cc[it]);
      }
      PDS_FPRIO_VALU();
      PDS_SSTAMP(0, 4);
      cc[kNT - 1] = edge_pair<kNT>(e2p, h1r, b2s + 48, g);
#pragma unroll
      for (int it = 0; it < kNT; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) h2r[it][q] = act_fast<ACT>(cc[it][q]);
        f32x4 v = h2r[it];
        if (it == kNT - 1) v[2] = 1.f;  // the bias column of dW3
        if (it < kNT - 1 || g == 0) sts4(H2img + n * kSI + it * kTW + 4 * g, v);
      }
      PDS_SSTAMP(0, 5);
      PDS_FPRIO_MFMA();
      // ---- layer 3: two chains (even / odd k-tiles) ----
      f32x4 y;
      {
        f32x4 c0, c1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          c0 = PDS_MFMA_F(a3[0][j], h2r[0][j], j == 0 ? lds4(b3s + 4 * g) : c0);
          c1 = PDS_MFMA_F(a3[1][j], h2r[1][j], j == 0 ? (f32x4)(0.f) : c1);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          c0 = PDS_MFMA_F(a3[2][j], h2r[2][j], c0);
          if (j < KJH) c1 = PDS_MFMA_F(a3[3][j], h2r[3][j], c1);
        }
        y = c0 + c1;
      }
      PDS_FPRIO_VALU();
      PDS_SSTAMP(0, 6);
      // ---- loss: compute_loss_pi, algs/ppo/ppo.py:22-40 (see mlp_kernel) ----
      f32x4 dy = (f32x4)(0.f);
      {
        float lp = 0.f, kl = 0.f;
        f32x4 zs = (f32x4)(0.f);
        const f32x4 is4 = lds4(isg + 4 * g), ls4 = lds4(lsg + 4 * g);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool on = 4 * g + q < m.d_out;  // selects, not branches
          const float z = (c_act[q] - y[q]) * is4[q];
          lp += on ? -0.5f * z * z - ls4[q] - 0.91893853320467274178f : 0.f;
          kl += on ? 0.5f * z * z : 0.f;
          zs[q] = on ? z * is4[q] : 0.f;
        }
        if (m.d_out > 4) {  // the sums over the action dimensions span two lane groups (else: group 0 holds all of them, and
                            // the other groups' values only ever multiply their own zs = 0)
          lp += __shfl_xor(lp, 16); lp += __shfl_xor(lp, 32);
          kl += __shfl_xor(kl, 16); kl += __shfl_xor(kl, 32);
        }
        const float ratio = expf(lp - c_old);
        const float lo = 1.f - a.clip, hi = 1.f + a.clip;
        const float obj = fminf(ratio * c_adv, fminf(fmaxf(ratio, lo), hi) * c_adv);
        const bool cut = (c_adv > 0.f && ratio > hi) || (c_adv < 0.f && ratio < lo);
        const float gcoef = (cut || !valid) ? 0.f : -c_adv * ratio;
        dy = gcoef * zs;
        if (valid && g == 0) { st_loss += -obj; st_ratio += ratio; st_kl += kl; st_cnt += 1.f; }
      }
      sts4(dYimg + n * kSY + 4 * g, dy);
      PDS_WAVE_SYNC();
      PDS_SSTAMP(0, 7);
      // operands of dZ2 (B: one dword of dY) and of dW3
      const float dyb = dYimg[n * kSY + h];
      float av3[4], bv3[4][kNT];
      float sa3[kTS], sb3[kTS];  // STRIP: dY[sample][lane % 4], H2[sample][lane]
      if constexpr (STRIP) {
#pragma unroll
        for (int sm = 0; sm < kTS; ++sm) { sa3[sm] = dYimg[sm * kSY + (lane & 3)]; sb3[sm] = H2img[sm * kSI + lc]; }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          av3[j] = dYimg[(4 * h + j) * kSY + r];
#pragma unroll
          for (int jt = 0; jt < kNT; ++jt) bv3[j][jt] = H2img[(4 * h + j) * kSI + jt * kTW + (jt == kNT - 1 ? n3 : n)];
        }
      }
      PDS_FPRIO_MFMA();
      // ---- dZ2^T = (W3^T dY^T) * act'(H2^T): the k-slot (step jj, lane group h) carries output 4 jj + h, so one
      // step covers the 4 action dimensions of the drone (mlp_kernel's slot order needs 4 steps for them) ----
#pragma unroll
      for (int it = 0; it < kNT; ++it) cc[it] = PDS_MFMA_F(wz2[it], dyb, (f32x4)(0.f));
      if (m.d_out > 4) {
        const float dyb2 = dYimg[n * kSY + 4 + h];
#pragma unroll
        for (int it = 0; it < kNT; ++it) cc[it] = PDS_MFMA_F(W3s[(4 + h) * kS + it * kTW + r], dyb2, cc[it]);
      }
      // ---- dW3 += dY^T H2 (its MFMAs cover the result latency of dZ2): a 4-row strip, one rank-1 update per sample ----
      if constexpr (STRIP) {
#pragma unroll
        for (int sm = 0; sm < kTS; ++sm) gW3s = PDS_MFMA44(sa3[sm], sb3[sm], gW3s);
        if (m.d_out > 4) {
#pragma unroll
          for (int sm = 0; sm < kTS; ++sm) gW3s2 = PDS_MFMA44(dYimg[sm * kSY + 4 + (lane & 3)], sb3[sm], gW3s2);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int jt = 0; jt < kNT; ++jt) gW3[jt] = PDS_MFMA_F(av3[j], bv3[j][jt], gW3[jt]);
      }
      PDS_FPRIO_VALU();
      PDS_SSTAMP(0, 8);
#pragma unroll
      for (int it = 0; it < kNT; ++it) {
        f32x4 dz2;
#pragma unroll
        for (int q = 0; q < 4; ++q) dz2[q] = act_back<ACT>(cc[it][q], h2r[it][q]);
        if (it < kNT - 1 || g == 0) sts4(dZ2img + n * kSI + it * kTW + 4 * g, dz2);
      }
      PDS_SSTAMP(0, 9);
      signal(full + s, (k >> 1) + 1, lane);
      PDS_SSTAMP(0, 10);
    }
  }

  // ---- block sums: the 4 waves of each role add up what that role accumulates (F: dW3, statistics; G / G1: dW1 (+ dW2);
  // G2: dW2); rounds 2,3 -> 0,1 and 1 -> 0 through LDS (the images are free now) -------------------------------------
  {
    constexpr int kRegs1 = 4 * (kNT * NIN + (NG == 1 ? kNT * kNT : 0)), kRegs2 = 4 * kNT * kNT, kRegsF = 4 * kNT + kStats;  // (upper bounds with STRIP)
    static_assert(2 * (kRegs1 + (NG == 2 ? kRegs2 : 0)) * 64 <= kPairs * 2 * kSetFloats, "the register images of the weight-gradient roles must fit in the tile sets");
    static_assert(2 * kRegsF * 64 <= kPairs * kPrivFloats, "two F register images must fit in the private images");
    auto xfer = [&](float *slot, bool add) {
      int c = 0;  // 16-byte slots (see mlp_kernel)
      auto four = [&](f32x4 &v) {
        float *q = slot + (c * 64 + lane) * 4;
        if (add) v += lds4(q); else sts4(q, v);
        ++c;
      };
      if (role == 1) {
#pragma unroll
        for (int i = 0; i < NTF; ++i) {
#pragma unroll
          for (int j = 0; j < NIN; ++j) four(gW1[i][j]);
          if (NG == 1) {
#pragma unroll
            for (int j = 0; j < NTF; ++j) four(gW2[i][j]);
          }
        }
        if (STRIP) { four(gW2r); four(gW2c); four(gW1r); }
      } else if (role == 2) {
#pragma unroll
        for (int i = 0; i < kNT; ++i)
#pragma unroll
          for (int j = 0; j < kNT; ++j) four(gW2[i][j]);
      } else {
        if (STRIP) {
          four(gW3s); four(gW3s2);
        } else {
#pragma unroll
          for (int i = 0; i < kNT; ++i) four(gW3[i]);
        }
        f32x4 st = {st_loss, st_ratio, st_kl, st_cnt};
        four(st);
        st_loss = st[0]; st_ratio = st[1]; st_kl = st[2]; st_cnt = st[3];
      }
    };
    float *red = role == 0 ? priv : (role == 1 ? sets : sets + 2 * kRegs1 * 64);
    const int stride = (role == 0 ? kRegsF : (role == 1 ? kRegs1 : kRegs2)) * 64;
    __syncthreads();
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      const int src0 = round == 0 ? 2 : 1, nsrc = round == 0 ? 2 : 1;
      if (pair >= src0 && pair < src0 + nsrc) xfer(red + (pair - src0) * stride, false);
      __syncthreads();
      if (pair < nsrc) xfer(red + pair * stride, true);
      __syncthreads();
    }
    if (pair != 0) return;
  }
  float *out = a.partials + (long long)blockIdx.x * a.pstride;
  const Offsets o = offsets(m);
  if (role != 0) {
#pragma unroll
    for (int it = 0; it < NTF; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = it * kTW + 4 * g + q;
#pragma unroll
        for (int jt = 0; jt < kNT; ++jt) {
          const int j = jt * kTW + n;
          if (role == 1) {
            if (jt < NIN && i < m.h1 && j < m.d_in) out[o.w1 + i * m.d_in + j] = gW1[it][jt < NIN ? jt : 0][q];
            if (jt < NIN && i < m.h1 && j == m.d_in) out[o.b1 + i] = gW1[it][jt < NIN ? jt : 0][q];
          }
          if (role == (NG == 1 ? 1 : 2) && jt < NTF) {
            if (i < m.h2 && j < m.h1) out[o.w2 + i * m.h1 + j] = gW2[it][jt][q];
            if (i < m.h2 && j == m.h1) out[o.b2 + i] = gW2[it][jt][q];
          }
        }
      }
    }
    if (STRIP && role == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // row strips: lane l holds rows 48 + q at column l
        const int i = 48 + q;
        if (i < m.h2 && lane < m.h1) out[o.w2 + i * m.h1 + lane] = gW2r[q];
        if (i < m.h2 && lane == m.h1) out[o.b2 + i] = gW2r[q];
        if (i < m.h1 && lane < m.d_in) out[o.w1 + i * m.d_in + lane] = gW1r[q];
        if (i < m.h1 && lane == m.d_in) out[o.b1 + i] = gW1r[q];
        // column strip: lane l holds rows 4 (l / 4) + q at column 48 + l % 4 (rows 48.. belong to the row strip)
        const int ic = 4 * (lane >> 2) + q, jc = 48 + (lane & 3);
        if (ic < 48 && jc < m.h1) out[o.w2 + ic * m.h1 + jc] = gW2c[q];
        if (ic < 48 && jc == m.h1) out[o.b2 + ic] = gW2c[q];
      }
    }
  } else if (STRIP) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // lane l holds outputs q (gW3s) and 4 + q (gW3s2) at column l
      if (q < m.d_out && lane < m.h2) out[o.w3 + q * m.h2 + lane] = gW3s[q];
      if (q < m.d_out && lane == m.h2) out[o.b3 + q] = gW3s[q];
      if (4 + q < m.d_out && lane < m.h2) out[o.w3 + (4 + q) * m.h2 + lane] = gW3s2[q];
      if (4 + q < m.d_out && lane == m.h2) out[o.b3 + 4 + q] = gW3s2[q];
    }
    float s4[kStats] = {st_loss, st_ratio, st_kl, st_cnt};
#pragma unroll
    for (int q = 0; q < kStats; ++q) {
      float v = s4[q];
      for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d);
      if (lane == 0) out[o.total + q] = v;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 4 * g + q;
#pragma unroll
      for (int jt = 0; jt < kNT; ++jt) {
        const int j = jt * kTW + n;
        if (i < m.d_out && j < m.h2) out[o.w3 + i * m.h2 + j] = gW3[jt][q];
        if (i < m.d_out && j == m.h2) out[o.b3 + i] = gW3[jt][q];
      }
    }
    float s4[kStats] = {st_loss, st_ratio, st_kl, st_cnt};
#pragma unroll
    for (int q = 0; q < kStats; ++q) {
      float v = s4[q];
      for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d);
      if (lane == 0) out[o.total + q] = v;
    }
  }
}

// The optimiser step that may ride on the partial-sum kernel (pds_*_grad_step): torch.optim.Adam, the arithmetic of
// pds_adam_step (csrc/pds_train.hip adam_kernel) expression for expression, so both routes give the same bits.
struct AdamK {
  float *em, *ev;  // exp_avg, exp_avg_sq [total]; em == nullptr: no step
  float lr, b1, b2, eps, bc1, bc2s;
};

// block = 64 outputs x 16 slices of the wave range: 16 x fewer dependent loads per thread
__global__ __launch_bounds__(1024) void reduce_kernel(const float *partials, int pstride, int nwaves, int total,
                                                      float denom_scale, float *grads, float *stats, pds_mlp m, AdamK ad) {
  __shared__ float part[16][64];
  const int px = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + px;
  float s = 0.f;
  if (p < total + kStats) {
    const int per = (nwaves + 15) / 16, w0 = sl * per, w1 = min(nwaves, w0 + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = w0;
    for (; w + 3 < w1; w += 4) {
      s0 += partials[(long long)w * pstride + p];
      s1 += partials[(long long)(w + 1) * pstride + p];
      s2 += partials[(long long)(w + 2) * pstride + p];
      s3 += partials[(long long)(w + 3) * pstride + p];
    }
    for (; w < w1; ++w) s0 += partials[(long long)w * pstride + p];
    s = (s0 + s1) + (s2 + s3);
  }
  part[sl][px] = s;
  __syncthreads();
  if (sl == 0 && p < total + kStats) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += part[q][px];
    if (p < total) {
      const float gr = t * denom_scale;
      grads[p] = gr;
      if (ad.em != nullptr) {
        const Offsets o = offsets(m);
        float *dst;
        if (p < o.b1) dst = const_cast<float *>(m.w1) + p;
        else if (p < o.w2) dst = const_cast<float *>(m.b1) + (p - o.b1);
        else if (p < o.b2) dst = const_cast<float *>(m.w2) + (p - o.w2);
        else if (p < o.w3) dst = const_cast<float *>(m.b2) + (p - o.b2);
        else if (p < o.b3) dst = const_cast<float *>(m.w3) + (p - o.w3);
        else dst = const_cast<float *>(m.b3) + (p - o.b3);
        const float mm = ad.b1 * ad.em[p] + (1.f - ad.b1) * gr;
        const float vv = ad.b2 * ad.ev[p] + (1.f - ad.b2) * gr * gr;
        ad.em[p] = mm; ad.ev[p] = vv;
        const float denom = sqrtf(vv) / ad.bc2s + ad.eps;
        *dst = *dst - (ad.lr / ad.bc1) * (mm / denom);
      }
    } else {
      stats[p - total] = t;
    }
  }
}

// data-carrying k-steps of a partially filled last k-tile (4: the tile is full enough for the generic code)
inline int last_steps(int dim) {
  const int d = dim - 16 * ((dim - 1) / 16);
  return d < 4 ? d : 4;
}
// the shapes of the reference's default nets get the compile-time skipping of all-padding k-steps:
// 2 data steps in the last hidden k-tile (50 units) and / or in the last input k-tile (34 inputs)
inline bool two_hidden_steps(const pds_mlp &m) { return m.h1 == m.h2 && m.h1 > 48 && last_steps(m.h1) == 2; }
inline bool two_input_steps(const pds_mlp &m) { return m.d_in > 32 && m.d_in <= 48 && last_steps(m.d_in) == 2; }

int check(const pds_mlp *m) {
  if (!m || m->d_in < 1 || m->d_in > kMaxDimIn || m->h1 < 1 || m->h1 > kMaxDim || m->h2 < 1 || m->h2 > kMaxDim ||
      m->d_out < 1 || m->d_out > kMaxOut || (m->activation != 0 && m->activation != 1) || !m->w1 || !m->b1 ||
      !m->w2 || !m->b2 || !m->w3 || !m->b3)
    return PDS_EINVAL;
  return PDS_OK;
}

int grid_blocks(long long B) {
  const long long tiles = (B + kTS - 1) / kTS;
  const long long blocks = (tiles + kWaves - 1) / kWaves;
  return (int)(blocks < 256 ? blocks : 256);  // one persistent block per CU
}

constexpr int kMaxGridBlocks = 256;  // one partial per block

}  // namespace pds_mlp_detail
using namespace pds_mlp_detail;

#if PDS_SPLIT_STAMPS
extern "C" int pds_debug_split_stamps(unsigned long long *out32) {
  return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_split_stamps), 32 * sizeof(unsigned long long)) == hipSuccess ? PDS_OK : PDS_EHIP;
}
#endif

extern "C" int pds_mlp_param_count(const pds_mlp *m) {
  if (check(m) != PDS_OK) return PDS_EINVAL;
  return offsets(*m).total;
}

extern "C" int64_t pds_mlp_workspace_floats(const pds_mlp *m) {
  if (check(m) != PDS_OK) return PDS_EINVAL;
  // (more than 64 inputs: one partial per WAVE of csrc/pds_mlp_wide.hip's grid)
  const int64_t parts = m->d_in > kMaxDim ? (int64_t)kWideMaxBlocks * kWideWaves : kMaxGridBlocks;
  return parts * (offsets(*m).total + kStats);
}

extern "C" int pds_mlp_forward(const pds_mlp *m, const float *d_x, const int64_t *d_index, int64_t B,
                               const float *d_mean, const float *d_std, float eps, float *d_y, void *stream) {
  if (check(m) != PDS_OK || !d_x || !d_y || B < 1 || ((d_mean == nullptr) != (d_std == nullptr))) return PDS_EINVAL;
  Args a{};
  a.m = *m; a.x = d_x; a.index = d_index; a.B = B; a.mean = d_mean; a.stdv = d_std; a.eps = eps; a.y = d_y;
  const dim3 g(grid_blocks(B)), b(kWaves * 64);
  hipStream_t s = (hipStream_t)stream;
  if (m->d_in > kMaxDim) {  // the K-tiled first layer (csrc/pds_mlp_wide.hip)
    launch_wide(LOSS_NONE, a, s);
    return hipGetLastError() == hipSuccess ? PDS_OK : PDS_EHIP;
  }
  const bool wide = m->d_in > 3 * kTW;
  const bool ki2 = two_input_steps(*m), kh2 = two_hidden_steps(*m);
  if (m->activation == 0) {
    if (wide) hipLaunchKernelGGL((mlp_kernel<LOSS_NONE, 0, 2, false>), g, b, 0, s, a);
    else if (kh2 && ki2) hipLaunchKernelGGL((mlp_kernel<LOSS_NONE, 0, 1, false, 2, 2>), g, b, 0, s, a);  // default policy, 34 inputs
    else if (kh2) hipLaunchKernelGGL((mlp_kernel<LOSS_NONE, 0, 1, false, 4, 2>), g, b, 0, s, a);         // default policy
    else hipLaunchKernelGGL((mlp_kernel<LOSS_NONE, 0, 1, false>), g, b, 0, s, a);
  } else {
    if (wide) hipLaunchKernelGGL((mlp_kernel<LOSS_NONE, 1, 2, false>), g, b, 0, s, a);
    else if (ki2 && !kh2) hipLaunchKernelGGL((mlp_kernel<LOSS_NONE, 1, 1, false, 2, 4>), g, b, 0, s, a);  // default critic, 34 inputs
    else hipLaunchKernelGGL((mlp_kernel<LOSS_NONE, 1, 1, false>), g, b, 0, s, a);
  }
  return hipGetLastError() == hipSuccess ? PDS_OK : PDS_EHIP;
}

static int launch_grad(int loss, Args &a, float *d_grads, float *d_stats, float *d_workspace, const pds_adam *opt,
                       void *stream) {
  const Offsets o = offsets(a.m);
  int blocks = grid_blocks(a.B);
  a.partials = d_workspace;
  a.pstride = o.total + kStats;
  hipStream_t s = (hipStream_t)stream;
  const dim3 g(blocks), b(kWaves * 64);
  // bias gradients out of the GEMMs' padding columns when every dimension leaves one (d_in = 48 then
  // takes the 4-tile input variant); per-lane partial sums otherwise (e.g. the 64-64 critic)
  const bool gb = a.m.h1 == kMaxDim || a.m.h2 == kMaxDim || a.m.d_in == kMaxDim;
  const bool wide = gb ? a.m.d_in > 3 * kTW : a.m.d_in >= 3 * kTW;
  const bool ktiled = a.m.d_in > kMaxDim;  // csrc/pds_mlp_wide.hip: one partial per wave of its grid
#define PDS_MLP_LAUNCH(L, A)                                                             \
  do {                                                                                   \
    if (gb) {                                                                            \
      if (wide) hipLaunchKernelGGL((mlp_kernel<L, A, 2, true>), g, b, 0, s, a);          \
      else hipLaunchKernelGGL((mlp_kernel<L, A, 1, true>), g, b, 0, s, a);               \
    } else {                                                                             \
      if (wide) hipLaunchKernelGGL((mlp_kernel<L, A, 2, false>), g, b, 0, s, a);         \
      else hipLaunchKernelGGL((mlp_kernel<L, A, 1, false>), g, b, 0, s, a);              \
    }                                                                                    \
  } while (0)
  // the reference's default policy (algs/ppo/defaults.py: 50-50 relu) on 34 (Hover, noisy) / 40 / 42 / 48
  // inputs: hidden k-tile 3 holds features 48, 49 only, input k-tile 2 of the 34-input net 32, 33 only
  if (ktiled) {
    blocks = launch_wide(loss, a, s);
  } else if (PDS_MLP_SPLIT && loss == LOSS_PPO && a.m.activation == 0 && !gb && !wide && two_hidden_steps(a.m) &&
      a.index == nullptr && a.mean == nullptr) {
    // wave roles (ppo_split_kernel): a block takes 4 tiles at a time, not 8
    const long long tiles = (a.B + kTS - 1) / kTS;
    blocks = (int)((tiles + kPairs - 1) / kPairs < kMaxGridBlocks ? (tiles + kPairs - 1) / kPairs : kMaxGridBlocks);
    const dim3 gs(blocks);
    const dim3 bs((1 + PDS_MLP_SPLIT) * 256);  // PDS_MLP_SPLIT = number of weight-gradient roles
    // the split-bf16 form of the weight-gradient role wins where a wave pair streams many tiles (its per-tile LATENCY is longer:
    // the splits sit in front of the MFMAs), the f32 form below ~4 tiles per pair (measured crossover: 32 000 .. 65 536 samples, profiles/r06_bf16_threshold.txt)
    static const long long bf16_min = [] { const char *e = getenv("PDS_BF16_MIN_SAMPLES"); return e ? atoll(e) : 65536ll; }();
    if (PDS_SPLIT_BF16 && a.B >= bf16_min) {
      if (two_input_steps(a.m)) hipLaunchKernelGGL((ppo_split_kernel<2, PDS_MLP_SPLIT, true>), gs, bs, 0, s, a);
      else hipLaunchKernelGGL((ppo_split_kernel<4, PDS_MLP_SPLIT, true>), gs, bs, 0, s, a);
    } else {
      if (two_input_steps(a.m)) hipLaunchKernelGGL((ppo_split_kernel<2, PDS_MLP_SPLIT>), gs, bs, 0, s, a);
      else hipLaunchKernelGGL((ppo_split_kernel<4, PDS_MLP_SPLIT>), gs, bs, 0, s, a);
    }
  } else if (loss == LOSS_PPO && a.m.activation == 0 && !gb && !wide && two_hidden_steps(a.m)) {
    if (two_input_steps(a.m)) hipLaunchKernelGGL((mlp_kernel<LOSS_PPO, 0, 1, false, 2, 2>), g, b, 0, s, a);
    else hipLaunchKernelGGL((mlp_kernel<LOSS_PPO, 0, 1, false, 4, 2>), g, b, 0, s, a);
  } else if (loss == LOSS_MSE && a.m.activation == 1 && gb && !wide && two_input_steps(a.m) && !two_hidden_steps(a.m)) {
    hipLaunchKernelGGL((mlp_kernel<LOSS_MSE, 1, 1, true, 2, 4>), g, b, 0, s, a);  // default critic, 34 inputs
  } else if (loss == LOSS_PPO) {
    if (a.m.activation == 0) PDS_MLP_LAUNCH(LOSS_PPO, 0); else PDS_MLP_LAUNCH(LOSS_PPO, 1);
  } else {
    if (a.m.activation == 0) PDS_MLP_LAUNCH(LOSS_MSE, 0); else PDS_MLP_LAUNCH(LOSS_MSE, 1);
  }
#undef PDS_MLP_LAUNCH
  const int n = o.total + kStats;
  AdamK ad{};
  if (opt != nullptr) {
    ad.em = opt->d_exp_avg; ad.ev = opt->d_exp_avg_sq;
    ad.lr = opt->lr; ad.b1 = opt->beta1; ad.b2 = opt->beta2; ad.eps = opt->eps;
    ad.bc1 = 1.0f - powf(opt->beta1, (float)opt->step);  // as pds_adam_step
    ad.bc2s = sqrtf(1.0f - powf(opt->beta2, (float)opt->step));
  }
  hipLaunchKernelGGL(reduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, s, (const float *)d_workspace, a.pstride,
                     blocks, o.total, 1.0f / (float)a.B, d_grads, d_stats, a.m, ad);
  return hipGetLastError() == hipSuccess ? PDS_OK : PDS_EHIP;
}

static bool adam_ok(const pds_adam *opt) {
  return opt == nullptr || (opt->d_exp_avg && opt->d_exp_avg_sq && opt->step >= 1);
}

extern "C" int pds_ppo_policy_grad_step(const pds_mlp *m, const float *d_x, const float *d_act, const float *d_adv,
                                        const float *d_logp_old, const float *d_log_std, int64_t B, float clip_ratio,
                                        float *d_grads, float *d_stats, float *d_workspace, const pds_adam *opt,
                                        void *stream) {
  if (check(m) != PDS_OK || !d_x || !d_act || !d_adv || !d_logp_old || !d_log_std || !d_grads || !d_stats ||
      !d_workspace || B < 1 || !adam_ok(opt))
    return PDS_EINVAL;
  Args a{};
  a.m = *m; a.x = d_x; a.B = B; a.act = d_act; a.adv = d_adv; a.logp_old = d_logp_old; a.log_std = d_log_std;
  a.clip = clip_ratio;
  return launch_grad(LOSS_PPO, a, d_grads, d_stats, d_workspace, opt, stream);
}

extern "C" int pds_ppo_policy_grad(const pds_mlp *m, const float *d_x, const float *d_act, const float *d_adv,
                                   const float *d_logp_old, const float *d_log_std, int64_t B, float clip_ratio,
                                   float *d_grads, float *d_stats, float *d_workspace, void *stream) {
  return pds_ppo_policy_grad_step(m, d_x, d_act, d_adv, d_logp_old, d_log_std, B, clip_ratio, d_grads, d_stats,
                                  d_workspace, nullptr, stream);
}

extern "C" int pds_value_grad_step(const pds_mlp *m, const float *d_x, const int64_t *d_index, const float *d_target,
                                   int64_t B, float *d_grads, float *d_stats, float *d_workspace, const pds_adam *opt,
                                   void *stream) {
  if (check(m) != PDS_OK || m->d_out != 1 || !d_x || !d_target || !d_grads || !d_stats || !d_workspace || B < 1 ||
      !adam_ok(opt))
    return PDS_EINVAL;
  Args a{};
  a.m = *m; a.x = d_x; a.index = d_index; a.B = B; a.target = d_target;
  return launch_grad(LOSS_MSE, a, d_grads, d_stats, d_workspace, opt, stream);
}

extern "C" int pds_value_grad(const pds_mlp *m, const float *d_x, const int64_t *d_index, const float *d_target,
                              int64_t B, float *d_grads, float *d_stats, float *d_workspace, void *stream) {
  return pds_value_grad_step(m, d_x, d_index, d_target, B, d_grads, d_stats, d_workspace, nullptr, stream);
}
