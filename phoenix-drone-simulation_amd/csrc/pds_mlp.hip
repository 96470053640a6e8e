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
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pds.h"

namespace pds_mlp_detail {  // named (not anonymous) so that profiler kernel names are readable

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTS = 16;            // samples per wave tile (= M of one MFMA tile)
constexpr int kTW = 16;            // tile width
constexpr int kNT = 4;             // 16-wide tiles per 64-wide dimension
constexpr int kLS = 65;            // LDS row stride of every [rows][<= 64] image (odd: conflict-free both ways)
constexpr int kOS = 9;             // row stride of the [16][<= 8] output / output-gradient tile
constexpr int kMaxDim = 64;        // d_in, h1, h2 <= 64
constexpr int kMaxOut = 8;         // d_out <= 8
constexpr int kWaves = 8;          // waves per block, two per SIMD; 1 block per CU (LDS-bound)
constexpr int kStats = 4;          // loss sum, ratio sum, kl sum, sample count
constexpr int kTileFloats = 3 * kTS * kLS + kTS * kOS;

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

// C/D map of v_mfma_f32_16x16x4_f32: lane l holds column l & 15 of rows 4 * (l >> 4) + r, r = 0..3
__device__ __forceinline__ int row_of(int r, int lane) { return (lane >> 4) * 4 + r; }

// c[i][j] (16x16 each) += A_i[16 x K] * B_j[K x 16] for the first na / nb of NA row tiles of A and NB
// column tiles of B.  Element (m, k) of A tile i at A[i * a_toff + m * a_sm + k * a_sk], (k, n) of
// B tile j at B[j * b_toff + k * b_sk + n * b_sn]; lanes whose row index is >= a_rows feed zeros
// (short matrices).  K is walked in chunks of 2 MFMA k-steps (8 k values): the operands of the
// next chunk are read from LDS while the matrix core works on the current one.  The walk may run up
// to 7 k values past K: every image is zero padded to 64 columns / rows, so that adds zeros.
template <int NA, int NB>
__device__ __forceinline__ void mma_block(f32x4 (&c)[NA][NB], int na, int nb, const float *A, int a_sm, int a_sk,
                                          int a_toff, int a_rows, const float *B, int b_sk, int b_sn, int b_toff,
                                          int K, int lane) {
  constexpr int CH = 2;
  const int r = lane & 15, h = lane >> 4;
  const float *ap = A + r * a_sm + h * a_sk;
  const float *bp = B + h * b_sk + r * b_sn;
  const bool a_on = r < a_rows;
  float a0[CH][NA], b0[CH][NB];
#pragma unroll
  for (int q = 0; q < CH; ++q) {
#pragma unroll
    for (int i = 0; i < NA; ++i) a0[q][i] = a_on ? ap[i * a_toff + 4 * q * a_sk] : 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) b0[q][j] = bp[j * b_toff + 4 * q * b_sk];
  }
  for (int k0 = 0; k0 < K; k0 += 4 * CH) {
    float a1[CH][NA], b1[CH][NB];
    const int kn = k0 + 4 * CH;
    if (kn < K) {
#pragma unroll
      for (int q = 0; q < CH; ++q) {
#pragma unroll
        for (int i = 0; i < NA; ++i) a1[q][i] = a_on ? ap[i * a_toff + (kn + 4 * q) * a_sk] : 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) b1[q][j] = bp[j * b_toff + (kn + 4 * q) * b_sk];
      }
    }
#pragma unroll
    for (int q = 0; q < CH; ++q)
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
          c[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q][i], b0[q][j], c[i][j], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < CH; ++q) {
#pragma unroll
      for (int i = 0; i < NA; ++i) a0[q][i] = a1[q][i];
#pragma unroll
      for (int j = 0; j < NB; ++j) b0[q][j] = b1[q][j];
    }
  }
}

// ACT 0 relu, 1 tanh (branch-free: 1 - 2 / (e^{2v} + 1) on v_exp_f32 / v_rcp_f32, abs error < 3e-7)
template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
  if (ACT == 0) return fmaxf(v, 0.f);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * v) + 1.f);
}
// derivative expressed through the activation's OUTPUT h (relu: h > 0; tanh: 1 - h^2)
template <int ACT>
__device__ __forceinline__ float act_grad(float h) { return ACT == 0 ? (h > 0.f ? 1.f : 0.f) : 1.f - h * h; }

__device__ __forceinline__ int tiles_of(int v) { return (v + kTW - 1) / kTW; }

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

// NINB: 16-wide column tiles of dW1 beyond the first two (1: d_in <= 48, 2: d_in <= 64) -- 16 accumulator
// registers that decide whether the gradient kernels fit the 256-register budget of two waves per SIMD
template <int LOSS, int ACT, int NINB>
__global__ __launch_bounds__(kWaves * 64, 2) void mlp_kernel(const Args a) {
  // ---- LDS images ---------------------------------------------------------------------------------
  __shared__ float W1s[kMaxDim * kLS], W2s[kMaxDim * kLS], W3s[kTW * kLS];  // [out][in], zero padded
  __shared__ float b1s[kMaxDim], b2s[kMaxDim], b3s[kMaxOut], isg[kMaxOut], lsg[kMaxOut];
  __shared__ float tiles[kWaves * kTileFloats];
  const pds_mlp &m = a.m;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < kMaxDim * kLS; i += kWaves * 64) {
    const int n = i / kLS, k = i - n * kLS;
    W1s[i] = (n < m.h1 && k < m.d_in) ? m.w1[n * m.d_in + k] : 0.f;
    W2s[i] = (n < m.h2 && k < m.h1) ? m.w2[n * m.h1 + k] : 0.f;
    if (i < kTW * kLS) W3s[i] = (n < m.d_out && k < m.h2) ? m.w3[n * m.h2 + k] : 0.f;
  }
  if (tid < kMaxDim) {
    b1s[tid] = tid < m.h1 ? m.b1[tid] : 0.f;
    b2s[tid] = tid < m.h2 ? m.b2[tid] : 0.f;
  }
  if (tid < kMaxOut) {
    b3s[tid] = tid < m.d_out ? m.b3[tid] : 0.f;
    const float ls = (LOSS == LOSS_PPO && tid < m.d_out) ? a.log_std[tid] : 0.f;
    lsg[tid] = ls;
    isg[tid] = expf(-ls);  // 1 / sigma
  }
  float *X = tiles + wave * kTileFloats;
  float *H1 = X + kTS * kLS, *H2 = H1 + kTS * kLS, *Y = H2 + kTS * kLS;
  for (int i = lane; i < kTileFloats; i += 64) X[i] = 0.f;  // pad columns stay zero
  __syncthreads();

  const int n_in = tiles_of(m.d_in), n_h1 = tiles_of(m.h1), n_h2 = tiles_of(m.h2);
  const int col = lane & 15;
  // weight-gradient accumulators of this wave (over all of its tiles); the column tiles in two halves
  f32x4 gW1a[kNT][2], gW1b[kNT][NINB], gW2a[kNT][2], gW2b[kNT][2], gW3[1][kNT];
  float gb1[kNT], gb2[kNT], gb3 = 0.f;
  float st_loss = 0.f, st_ratio = 0.f, st_kl = 0.f, st_cnt = 0.f;
#pragma unroll
  for (int i = 0; i < kNT; ++i) {
    gb1[i] = 0.f; gb2[i] = 0.f;
    gW3[0][i] = (f32x4)(0.f);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      gW1a[i][j] = (f32x4)(0.f); gW2a[i][j] = (f32x4)(0.f); gW2b[i][j] = (f32x4)(0.f);
      if (j < NINB) gW1b[i][j] = (f32x4)(0.f);
    }
  }

  const long long ntiles = (a.B + kTS - 1) / kTS;
  const long long wid = (long long)blockIdx.x * kWaves + wave, nw = (long long)gridDim.x * kWaves;
  // lane k holds feature k of the tile's 16 rows
  const bool kon = lane < m.d_in;
  float x_mu = 0.f, x_is = 1.f;
  if (a.mean != nullptr && kon) { x_mu = a.mean[lane]; x_is = 1.0f / (a.stdv[lane] + a.eps); }
  float xr[kTS];
  int idx_next = 0;  // lane s < 16: source row of sample s of the tile after next (-1: none)
  auto load_index = [&](long long tt) -> int {
    const long long g = tt * kTS + (lane & (kTS - 1));
    if (tt >= ntiles || g >= a.B) return -1;
    return a.index != nullptr ? (int)a.index[g] : (int)g;
  };
  auto load_rows = [&](int rows_of_tile) {
#pragma unroll
    for (int s = 0; s < kTS; ++s) {
      const int row = __builtin_amdgcn_readlane(rows_of_tile, s);  // scalar
      xr[s] = (kon && row >= 0) ? a.x[(long long)row * m.d_in + lane] : x_mu;
    }
  };
  idx_next = load_index(wid);
  if (LOSS == LOSS_NONE) {  // forward only: registers to spare, the rows travel one tile ahead
    load_rows(idx_next);
    idx_next = load_index(wid + nw);
  }
  for (long long t = wid; t < ntiles; t += nw) {
    const long long s0 = t * kTS;
    // ---- stage the input tile (optionally gathered and standardised) -----------------------------
    // (gradient kernels: the gather indices travel one tile ahead; the rows' HBM latency is covered
    // by the SIMD's other wave -- a register prefetch of the rows does not fit their 256-register budget)
    if (LOSS != LOSS_NONE) load_rows(idx_next);
    const int my_row = idx_next;  // lanes 0..15: source row of this lane's sample
    if (LOSS != LOSS_NONE) idx_next = load_index(t + nw);
    // the loss inputs are requested now and consumed after the three forward GEMMs
    float c_act[4] = {0.f, 0.f, 0.f, 0.f}, c_adv = 0.f, c_old = 0.f, c_tgt = 0.f;
    if (LOSS != LOSS_NONE && lane < kTS && s0 + lane < a.B) {
      const long long g = s0 + lane;
      if (LOSS == LOSS_PPO) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < m.d_out) c_act[j] = a.act[g * m.d_out + j];
        c_adv = a.adv[g]; c_old = a.logp_old[g];
      } else {
        c_tgt = a.target[my_row];
      }
    }
#pragma unroll
    for (int s = 0; s < kTS; ++s) X[s * kLS + lane] = (xr[s] - x_mu) * x_is;
    if (LOSS == LOSS_NONE) {
      load_rows(idx_next);
      idx_next = load_index(t + 2 * nw);
    }
    PDS_WAVE_SYNC();
    // ---- forward ---------------------------------------------------------------------------------
    {  // H1 = act(X W1^T + b1); columns >= h1 come out as act(0) = 0 (zero-padded weights and biases)
      f32x4 c[1][kNT];
#pragma unroll
      for (int j = 0; j < kNT; ++j) c[0][j] = (f32x4)(0.f);
      mma_block<1, kNT>(c, 1, n_h1, X, kLS, 1, 0, kTS, W1s, 1, kLS, kTW * kLS, m.d_in, lane);
#pragma unroll
      for (int nt = 0; nt < kNT; ++nt) {
        const int n = nt * kTW + col;
        const float bias = b1s[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) H1[row_of(r, lane) * kLS + n] = act_fn<ACT>(c[0][nt][r] + bias);
      }
    }
    PDS_WAVE_SYNC();
    {  // H2 = act(H1 W2^T + b2)
      f32x4 c[1][kNT];
#pragma unroll
      for (int j = 0; j < kNT; ++j) c[0][j] = (f32x4)(0.f);
      mma_block<1, kNT>(c, 1, n_h2, H1, kLS, 1, 0, kTS, W2s, 1, kLS, kTW * kLS, m.h1, lane);
#pragma unroll
      for (int nt = 0; nt < kNT; ++nt) {
        const int n = nt * kTW + col;
        const float bias = b2s[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) H2[row_of(r, lane) * kLS + n] = act_fn<ACT>(c[0][nt][r] + bias);
      }
    }
    PDS_WAVE_SYNC();
    {  // Y = H2 W3^T + b3 (columns >= d_out: 0)
      f32x4 c[1][1] = {{(f32x4)(0.f)}};
      mma_block<1, 1>(c, 1, 1, H2, kLS, 1, 0, kTS, W3s, 1, kLS, 0, m.h2, lane);
      if (col < kMaxOut) {
        const float bias = b3s[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[row_of(r, lane) * kOS + col] = c[0][0][r] + bias;
      }
    }
    PDS_WAVE_SYNC();

    if (LOSS == LOSS_NONE) {
      if (lane < kTS && s0 + lane < a.B)
        for (int j = 0; j < m.d_out; ++j) a.y[(s0 + lane) * m.d_out + j] = Y[lane * kOS + j];
      __builtin_amdgcn_wave_barrier();
      continue;
    }

    // ---- loss and its gradient with respect to the network output (lanes 0..15: one sample each) ---
    if (lane < kTS) {
      const long long g = s0 + lane;
      float *yr = Y + lane * kOS;
      if (g < a.B) {
        if (LOSS == LOSS_PPO) {
          // compute_loss_pi, algs/ppo/ppo.py:22-40 (Normal(mu, sigma).log_prob(act).sum(-1))
          float logp = 0.f, kl = 0.f, z[kMaxOut];
          for (int j = 0; j < m.d_out; ++j) {
            const float aj = j < 4 ? c_act[j & 3] : a.act[g * m.d_out + j];
            z[j] = (aj - yr[j]) * isg[j];
            logp += -0.5f * z[j] * z[j] - lsg[j] - 0.91893853320467274178f;
            kl += 0.5f * z[j] * z[j];
          }
          const float ratio = expf(logp - c_old);
          const float adv = c_adv;
          const float lo = 1.f - a.clip, hi = 1.f + a.clip;
          const float obj = fminf(ratio * adv, fminf(fmaxf(ratio, lo), hi) * adv);
          // d min(r A, clip(r) A) / d r  (torch.min splits ties, clamp passes its range: net A inside
          // the range, A outside it only on the un-clipped branch)
          const bool cut = (adv > 0.f && ratio > hi) || (adv < 0.f && ratio < lo);
          const float gcoef = cut ? 0.f : -adv * ratio;  // d(-obj)/d logp
          for (int j = 0; j < m.d_out; ++j) yr[j] = gcoef * z[j] * isg[j];
          st_loss += -obj; st_ratio += ratio; st_kl += kl; st_cnt += 1.f;
        } else {
          // compute_loss_v: mse_loss(v(obs), target_v), algs/iwpg/iwpg.py:272-275
          const float d = yr[0] - c_tgt;
          st_loss += d * d; st_cnt += 1.f;
          yr[0] = 2.f * d;
        }
      } else {
        for (int j = 0; j < m.d_out; ++j) yr[j] = 0.f;
      }
    }
    PDS_WAVE_SYNC();

    // ---- backward --------------------------------------------------------------------------------
    // dW3 += dY^T H2 (rows = outputs, K = samples); db3 += column sums of dY
    mma_block<1, kNT>(gW3, 1, n_h2, Y, 1, kOS, 0, kMaxOut, H2, kLS, 1, kTW, kTS, lane);
    if (lane < kMaxOut) {
      float sacc = 0.f;
#pragma unroll
      for (int s = 0; s < kTS; ++s) sacc += Y[s * kOS + lane];
      gb3 += sacc;
    }
    // dZ2 = (dY W3) * act'(H2), in place over H2 (columns >= h2: W3 columns are zero)
    {
      f32x4 c[1][kNT];
#pragma unroll
      for (int j = 0; j < kNT; ++j) c[0][j] = (f32x4)(0.f);
      mma_block<1, kNT>(c, 1, n_h2, Y, kOS, 1, 0, kTS, W3s, kLS, 1, kTW, m.d_out, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // every dW3 read of H2 is done
#pragma unroll
      for (int nt = 0; nt < kNT; ++nt) {
        const int n = nt * kTW + col;
        float sacc = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float *p = H2 + row_of(r, lane) * kLS + n;
          const float dz = c[0][nt][r] * act_grad<ACT>(*p);
          *p = dz;
          sacc += dz;
        }
        gb2[nt] += sacc;
      }
    }
    PDS_WAVE_SYNC();
    // dW2 += dZ2^T H1
    mma_block<kNT, 2>(gW2a, n_h2, min(n_h1, 2), H2, 1, kLS, kTW, kTS, H1, kLS, 1, kTW, kTS, lane);
    mma_block<kNT, 2>(gW2b, n_h2, n_h1 - 2, H2, 1, kLS, kTW, kTS, H1 + 2 * kTW, kLS, 1, kTW, kTS, lane);
    // dZ1 = (dZ2 W2) * act'(H1), in place over H1
    {
      f32x4 c[1][kNT];
#pragma unroll
      for (int j = 0; j < kNT; ++j) c[0][j] = (f32x4)(0.f);
      mma_block<1, kNT>(c, 1, n_h1, H2, kLS, 1, 0, kTS, W2s, kLS, 1, kTW, m.h2, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // every dW2 read of H1 is done
#pragma unroll
      for (int nt = 0; nt < kNT; ++nt) {
        const int n = nt * kTW + col;
        float sacc = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float *p = H1 + row_of(r, lane) * kLS + n;
          const float dz = c[0][nt][r] * act_grad<ACT>(*p);
          *p = dz;
          sacc += dz;
        }
        gb1[nt] += sacc;
      }
    }
    PDS_WAVE_SYNC();
    // dW1 += dZ1^T X
    mma_block<kNT, 2>(gW1a, n_h1, min(n_in, 2), H1, 1, kLS, kTW, kTS, X, kLS, 1, kTW, kTS, lane);
    mma_block<kNT, NINB>(gW1b, n_h1, n_in - 2, H1, 1, kLS, kTW, kTS, X + 2 * kTW, kLS, 1, kTW, kTS, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the tile images are restaged by the next iteration
  }

  if (LOSS == LOSS_NONE) return;
  // ---- this wave's partial sums -> partials[wid][...] (flat parameter layout + statistics) -----------
  float *out = a.partials + wid * a.pstride;
  const Offsets o = offsets(m);
#pragma unroll
  for (int it = 0; it < kNT; ++it)
#pragma unroll
    for (int jt = 0; jt < kNT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = it * kTW + row_of(r, lane), j = jt * kTW + col;
        const float w1v = jt < 2 ? gW1a[it][jt & 1][r] : (jt - 2 < NINB ? gW1b[it][(jt - 2) < NINB ? (jt - 2) : 0][r] : 0.f);
        const float w2v = jt < 2 ? gW2a[it][jt & 1][r] : gW2b[it][jt & 1][r];
        if (i < m.h1 && j < m.d_in) out[o.w1 + i * m.d_in + j] = w1v;
        if (i < m.h2 && j < m.h1) out[o.w2 + i * m.h1 + j] = w2v;
      }
#pragma unroll
  for (int jt = 0; jt < kNT; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = row_of(r, lane), j = jt * kTW + col;
      if (i < m.d_out && j < m.h2) out[o.w3 + i * m.h2 + j] = gW3[0][jt][r];
    }
  // bias gradients: lanes l, l + 16, l + 32, l + 48 hold the four row groups of the same column
#pragma unroll
  for (int nt = 0; nt < kNT; ++nt) {
    float v1 = gb1[nt], v2 = gb2[nt];
    v1 += __shfl_xor(v1, 16); v1 += __shfl_xor(v1, 32);
    v2 += __shfl_xor(v2, 16); v2 += __shfl_xor(v2, 32);
    const int n = nt * kTW + col;
    if (lane < kTW) {
      if (n < m.h1) out[o.b1 + n] = v1;
      if (n < m.h2) out[o.b2 + n] = v2;
    }
  }
  if (lane < m.d_out) out[o.b3 + lane] = gb3;
  // statistics: lanes 0..15 hold per-sample sums
  float s4[kStats] = {st_loss, st_ratio, st_kl, st_cnt};
#pragma unroll
  for (int q = 0; q < kStats; ++q) {
    float v = s4[q];
    for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0) out[o.total + q] = v;
  }
}

// block = 64 outputs x 16 slices of the wave range: 16 x fewer dependent loads per thread
__global__ __launch_bounds__(1024) void reduce_kernel(const float *partials, int pstride, int nwaves, int total,
                                                      float denom_scale, float *grads, float *stats) {
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
    if (p < total) grads[p] = t * denom_scale;
    else stats[p - total] = t;
  }
}

int check(const pds_mlp *m) {
  if (!m || m->d_in < 1 || m->d_in > kMaxDim || m->h1 < 1 || m->h1 > kMaxDim || m->h2 < 1 || m->h2 > kMaxDim ||
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

constexpr int kMaxGridWaves = 256 * kWaves;

}  // namespace pds_mlp_detail
using namespace pds_mlp_detail;

extern "C" int pds_mlp_param_count(const pds_mlp *m) {
  if (check(m) != PDS_OK) return PDS_EINVAL;
  return offsets(*m).total;
}

extern "C" int64_t pds_mlp_workspace_floats(const pds_mlp *m) {
  if (check(m) != PDS_OK) return PDS_EINVAL;
  return (int64_t)kMaxGridWaves * (offsets(*m).total + kStats);
}

extern "C" int pds_mlp_forward(const pds_mlp *m, const float *d_x, const int64_t *d_index, int64_t B,
                               const float *d_mean, const float *d_std, float eps, float *d_y, void *stream) {
  if (check(m) != PDS_OK || !d_x || !d_y || B < 1 || ((d_mean == nullptr) != (d_std == nullptr))) return PDS_EINVAL;
  Args a{};
  a.m = *m; a.x = d_x; a.index = d_index; a.B = B; a.mean = d_mean; a.stdv = d_std; a.eps = eps; a.y = d_y;
  if (m->activation == 0) hipLaunchKernelGGL((mlp_kernel<LOSS_NONE, 0, 1>), dim3(grid_blocks(B)), dim3(kWaves * 64), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((mlp_kernel<LOSS_NONE, 1, 1>), dim3(grid_blocks(B)), dim3(kWaves * 64), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? PDS_OK : PDS_EHIP;
}

static int launch_grad(int loss, Args &a, float *d_grads, float *d_stats, float *d_workspace, void *stream) {
  const Offsets o = offsets(a.m);
  const int blocks = grid_blocks(a.B);
  a.partials = d_workspace;
  a.pstride = o.total + kStats;
  hipStream_t s = (hipStream_t)stream;
  const dim3 g(blocks), b(kWaves * 64);
  const bool wide = a.m.d_in > 3 * kTW;
#define PDS_MLP_LAUNCH(L, A)                                                             \
  do {                                                                                   \
    if (wide) hipLaunchKernelGGL((mlp_kernel<L, A, 2>), g, b, 0, s, a);                  \
    else hipLaunchKernelGGL((mlp_kernel<L, A, 1>), g, b, 0, s, a);                       \
  } while (0)
  if (loss == LOSS_PPO) {
    if (a.m.activation == 0) PDS_MLP_LAUNCH(LOSS_PPO, 0); else PDS_MLP_LAUNCH(LOSS_PPO, 1);
  } else {
    if (a.m.activation == 0) PDS_MLP_LAUNCH(LOSS_MSE, 0); else PDS_MLP_LAUNCH(LOSS_MSE, 1);
  }
#undef PDS_MLP_LAUNCH
  const int n = o.total + kStats;
  hipLaunchKernelGGL(reduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, s, (const float *)d_workspace, a.pstride,
                     blocks * kWaves, o.total, 1.0f / (float)a.B, d_grads, d_stats);
  return hipGetLastError() == hipSuccess ? PDS_OK : PDS_EHIP;
}

extern "C" int pds_ppo_policy_grad(const pds_mlp *m, const float *d_x, const float *d_act, const float *d_adv,
                                   const float *d_logp_old, const float *d_log_std, int64_t B, float clip_ratio,
                                   float *d_grads, float *d_stats, float *d_workspace, void *stream) {
  if (check(m) != PDS_OK || !d_x || !d_act || !d_adv || !d_logp_old || !d_log_std || !d_grads || !d_stats ||
      !d_workspace || B < 1)
    return PDS_EINVAL;
  Args a{};
  a.m = *m; a.x = d_x; a.B = B; a.act = d_act; a.adv = d_adv; a.logp_old = d_logp_old; a.log_std = d_log_std;
  a.clip = clip_ratio;
  return launch_grad(LOSS_PPO, a, d_grads, d_stats, d_workspace, stream);
}

extern "C" int pds_value_grad(const pds_mlp *m, const float *d_x, const int64_t *d_index, const float *d_target,
                              int64_t B, float *d_grads, float *d_stats, float *d_workspace, void *stream) {
  if (check(m) != PDS_OK || m->d_out != 1 || !d_x || !d_target || !d_grads || !d_stats || !d_workspace || B < 1)
    return PDS_EINVAL;
  Args a{};
  a.m = *m; a.x = d_x; a.index = d_index; a.B = B; a.target = d_target;
  return launch_grad(LOSS_MSE, a, d_grads, d_stats, d_workspace, stream);
}
