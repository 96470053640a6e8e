// pds_mlp_wide.hip -- the trainer's dense kernels for networks with MORE than 64 inputs (64 < d_in <= 192): the first layer
// K-tiled over up to twelve 16-wide input tiles.
//
// Who needs it: `observation_history_size` >= 4 (envs/base.py:303-319: the observation is the last H [o, u] halves) --
// Hover 68 / 102 / 136 inputs at H = 4 / 6 / 8, Circle 80 / 120 / 160, TakeOff 96 / 144 / 192; what the reference's
// experiments/04_history_of_state_action_inputs/04_train_with_history.py:34 trains (H in {1, 2, 4, 6, 8}, policy 32-32 / 48-48 /
// 64-64 relu, critic 64-64 tanh).  Rounds 1-5 dropped both networks to PyTorch ops there (7 launches per rollout step +
// autograd: the round-1 path).
//
// Same design as mlp_kernel of csrc/pds_mlp.hip (read that first): a wave owns a 16-sample tile, every GEMM of forward and
// backward runs on v_mfma_f32_16x16x4_f32 with the transposed chain (activations stay in registers from layer to layer),
// the weight gradients accumulate in registers over the wave's tiles, a second kernel sums the partials in a fixed order.
// What changes with the width:
//   * the accumulators of dW1 are 4 x NIN tiles = up to 192 registers (mlp_kernel: 48), next to 64 of dW2, 16 of dW3 and 36
//     of the bias gradients: ONE wave per SIMD (4 per block, up to 512 registers each) instead of two -- the matrix pipe and the
//     vector ALU no longer overlap between waves; this is the route for shapes off the reference's default, not the hot one;
//   * W1's LDS image and the wave's X image have a row stride of 16 NIN + 4 floats (an odd multiple of 16 B, and 4 x stride ==
//     16 mod 32: the same two bank rules as the 68-float stride of the narrow kernels); 162.1 of 160 x 1024 = 163.8 KB of LDS
//     at NIN = 12 with four waves -- which is why there is no block-level reduction here (its staging area would not fit):
//     every WAVE writes its partial, and the reduce kernel sums 4 x more of them;
//   * bias gradients are always per-lane partial sums (the ones-column trick of the narrow kernels needs a padding column).
// Bound: MFMA f32, as the narrow kernels.
#include "pds_mlp_common.h"

#ifndef PDS_WIDE_BF16
#define PDS_WIDE_BF16 1  // round 6: dW2 and dW1 (K = the tile's 16 samples) on split-bf16 MFMAs, see pds_mlp_common.h / pds_mlp.hip; A/B: 0
#endif

namespace pds_mlp_detail {

// C[it][jt] += A_it^T B_jt over the 16 samples of a tile, operands as three bf16 pieces of the four values (samples 4 h .. 4 h + 3
// of one feature) a lane holds per 16 x 16 block: slot (h, i < 4) = sample 4 h + i with pieces (a, b), slot (h, i >= 4) = the
// same sample with (a', b') -- (lo | mid)(hi | mid), (mid | hi)(hi | lo), (hi | hi)(hi | mid) are the six products.
template <int NA, int NB, class Acc>
__device__ __forceinline__ void outer_bf16(const Quad3 (&qa)[NA], const Quad3 (&qb)[NB], Acc &&acc) {
#pragma unroll
  for (int it = 0; it < NA; ++it)
#pragma unroll
    for (int jt = 0; jt < NB; ++jt) acc(it, jt) = PDS_MFMA_BF(cat8(qa[it].lo, qa[it].mid), cat8(qb[jt].hi, qb[jt].mid), acc(it, jt));
#pragma unroll
  for (int it = 0; it < NA; ++it)
#pragma unroll
    for (int jt = 0; jt < NB; ++jt) acc(it, jt) = PDS_MFMA_BF(cat8(qa[it].mid, qa[it].hi), cat8(qb[jt].hi, qb[jt].lo), acc(it, jt));
#pragma unroll
  for (int it = 0; it < NA; ++it)
#pragma unroll
    for (int jt = 0; jt < NB; ++jt) acc(it, jt) = PDS_MFMA_BF(cat8(qa[it].hi, qa[it].hi), cat8(qb[jt].hi, qb[jt].mid), acc(it, jt));
}

template <int NIN>
constexpr int wide_stride() { return kTW * NIN + 4; }

// Z^T tiles `it`, `it + 1` = W[16 it .. +32][:] In^T for an LDS weight image with row stride S (pds_mlp.hip gemm_wt2)
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
      c0 = PDS_MFMA(a0[j], in[kt][j], c0);
      c1 = PDS_MFMA(a1[j], in[kt][j], c1);
    }
  }
}
template <int NK, int S>
__device__ __forceinline__ f32x4 gemm_wts(const float *Ws, int it, const f32x4 (&in)[NK], int n, int g) {
  f32x4 c = (f32x4)(0.f);
  const float *wp = Ws + (it * kTW + n) * S + 4 * g;
#pragma unroll
  for (int kt = 0; kt < NK; ++kt) {
    const f32x4 a = lds4(wp + kt * kTW);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = PDS_MFMA(a[j], in[kt][j], c);
  }
  return c;
}

template <int LOSS, int ACT, int NIN>
__global__ __launch_bounds__(kWideWaves * 64, 1) void mlp_wide_kernel(const Args a) {
  constexpr int S1 = wide_stride<NIN>();
  constexpr int kImg = (LOSS == LOSS_NONE) ? 0 : kTS * S1 + 2 * kTS * kS + kTS * kSY;  // X, H1, H2, dY per wave
  __shared__ __attribute__((aligned(16))) float W1s[kMaxDim * S1];  // [out][in], zero padded
  __shared__ __attribute__((aligned(16))) float W2s[kMaxDim * kS];
  // W3: the 8 rows d_out <= kMaxOut can fill (the narrow kernels keep 16): rows 8..15 of the 16-row MFMA tile ALIAS rows 0..7
  // (`& 7` below) -- outputs 8..15 are never read, and their gradient dY is zero, so the aliased rows only ever meet zeros
  __shared__ __attribute__((aligned(16))) float W3s[kMaxOut * kS];
  __shared__ __attribute__((aligned(16))) float b1s[kMaxDim], b2s[kMaxDim], b3s[kTW], mus[kTW * NIN], iss[kTW * NIN];
  __shared__ float isg[kTW], lsg[kTW];
  __shared__ __attribute__((aligned(16))) float images[kImg > 0 ? kWideWaves * kImg : 4];
  const pds_mlp &m = a.m;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;  // C/D layout: column (sample) n, rows 4 g + q
  constexpr int kThreads = kWideWaves * 64;
  for (int i = tid; i < kMaxDim * S1; i += kThreads) {
    const int r = i / S1, k = i - r * S1;
    W1s[i] = (r < m.h1 && k < m.d_in) ? m.w1[r * m.d_in + k] : 0.f;
  }
  for (int i = tid; i < kMaxDim * kS; i += kThreads) {
    const int r = i / kS, k = i - r * kS;
    W2s[i] = (r < m.h2 && k < m.h1) ? m.w2[r * m.h1 + k] : 0.f;
    if (i < kMaxOut * kS) W3s[i] = (r < m.d_out && k < m.h2) ? m.w3[r * m.h2 + k] : 0.f;
  }
  for (int i = tid; i < kTW * NIN; i += kThreads) {
    const bool std_on = a.mean != nullptr && i < m.d_in;
    mus[i] = std_on ? a.mean[i] : 0.f;
    iss[i] = std_on ? 1.0f / (a.stdv[i] + a.eps) : 1.f;
  }
  if (tid < kMaxDim) {
    b1s[tid] = tid < m.h1 ? m.b1[tid] : 0.f;
    b2s[tid] = tid < m.h2 ? m.b2[tid] : 0.f;
  }
  if (tid < kTW) {
    b3s[tid] = tid < m.d_out ? m.b3[tid] : 0.f;
    const float ls = (LOSS == LOSS_PPO && tid < m.d_out) ? a.log_std[tid] : 0.f;
    lsg[tid] = ls;
    isg[tid] = expf(-ls);  // 1 / sigma
  }
  float *Ximg = images + wave * kImg;  // [sample][feature] images for the weight-gradient GEMMs
  float *H1img = Ximg + kTS * S1, *H2img = H1img + kTS * kS, *dYimg = H2img + kTS * kS;
  if (LOSS != LOSS_NONE)
    for (int i = lane; i < kImg; i += 64) Ximg[i] = 0.f;
  __syncthreads();

  // weight-gradient accumulators of this wave (over all of its tiles), C/D layout; bias gradients as per-lane partial sums
  constexpr int NG1 = (LOSS == LOSS_NONE) ? 1 : NIN;
  f32x4 gW1[kNT][NG1], gW2[kNT][kNT], gW3[kNT], gb1[kNT], gb2[kNT], gb3 = (f32x4)(0.f);
  float st_loss = 0.f, st_ratio = 0.f, st_kl = 0.f, st_cnt = 0.f;
#pragma unroll
  for (int i = 0; i < kNT; ++i) {
    gW3[i] = (f32x4)(0.f); gb1[i] = (f32x4)(0.f); gb2[i] = (f32x4)(0.f);
#pragma unroll
    for (int j = 0; j < kNT; ++j) gW2[i][j] = (f32x4)(0.f);
#pragma unroll
    for (int j = 0; j < NG1; ++j) gW1[i][j] = (f32x4)(0.f);
  }

  const long long ntiles = (a.B + kTS - 1) / kTS;
  const long long wid = (long long)blockIdx.x * kWideWaves + wave, nw = (long long)gridDim.x * kWideWaves;
  for (long long t = wid; t < ntiles; t += nw) {
    const long long s0 = t * kTS;
    long long row = -1;  // source row of this lane's sample, -1: none
    if (s0 + n < a.B) row = a.index != nullptr ? a.index[s0 + n] : s0 + n;
    const bool valid = row >= 0;
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
      if (LOSS != LOSS_NONE) sts4(Ximg + n * S1 + kt * kTW + 4 * g, xin[kt]);
    }
    float c_act[4] = {0.f, 0.f, 0.f, 0.f}, c_adv = 0.f, c_old = 0.f, c_tgt = 0.f;
    if (LOSS == LOSS_PPO && valid) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (4 * g + q < m.d_out) c_act[q] = a.act[(s0 + n) * m.d_out + 4 * g + q];
      c_adv = a.adv[s0 + n]; c_old = a.logp_old[s0 + n];
    }
    if (LOSS == LOSS_MSE && valid) c_tgt = a.target[row];
    // ---- forward: activations stay in registers from layer to layer -------------------------------
    f32x4 h1r[kNT], h2r[kNT], cc[kNT];
#pragma unroll
    for (int it = 0; it < kNT; it += 2) gemm_wt2s<NIN, S1>(W1s, it, xin, n, g, cc[it], cc[it + 1]);
#pragma unroll
    for (int it = 0; it < kNT; ++it) {  // H1^T = act(W1 X^T + b1); rows >= h1: act(0) = 0
      const f32x4 b = lds4(b1s + it * kTW + 4 * g);
#pragma unroll
      for (int q = 0; q < 4; ++q) h1r[it][q] = act_fn<ACT>(cc[it][q] + b[q]);
      if (LOSS != LOSS_NONE) sts4(H1img + n * kS + it * kTW + 4 * g, h1r[it]);
    }
#pragma unroll
    for (int it = 0; it < kNT; it += 2) gemm_wt2s<kNT, kS>(W2s, it, h1r, n, g, cc[it], cc[it + 1]);
#pragma unroll
    for (int it = 0; it < kNT; ++it) {  // H2^T = act(W2 H1^T + b2)
      const f32x4 b = lds4(b2s + it * kTW + 4 * g);
#pragma unroll
      for (int q = 0; q < 4; ++q) h2r[it][q] = act_fn<ACT>(cc[it][q] + b[q]);
      if (LOSS != LOSS_NONE) sts4(H2img + n * kS + it * kTW + 4 * g, h2r[it]);
    }
    f32x4 y;  // Y^T = W3 H2^T + b3: lane (n, g) holds outputs 4 g + q of sample n (rows >= d_out: 0)
    {
      const f32x4 c = gemm_wts<kNT, kS>(W3s, 0, h2r, n & (kMaxOut - 1), g);
      y = c + lds4(b3s + 4 * g);
    }
    if constexpr (LOSS == LOSS_NONE) {
      if (valid) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * g + q < m.d_out) a.y[(s0 + n) * m.d_out + 4 * g + q] = y[q];
      }
    } else {
      // ---- loss and its gradient with respect to the network output (pds_mlp.hip mlp_kernel, same expressions) -------
      f32x4 dy = (f32x4)(0.f);
      if (LOSS == LOSS_PPO) {  // compute_loss_pi, algs/ppo/ppo.py:22-40
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
        const bool cut = (c_adv > 0.f && ratio > hi) || (c_adv < 0.f && ratio < lo);
        const float gcoef = (cut || !valid) ? 0.f : -c_adv * ratio;  // d(-obj)/d logp
        dy = gcoef * zs;
        if (valid && g == 0) { st_loss += -obj; st_ratio += ratio; st_kl += kl; st_cnt += 1.f; }
      } else {  // compute_loss_v: mse_loss(v(obs), target_v), algs/iwpg/iwpg.py:272-275
        if (valid && g == 0) {
          const float d = y[0] - c_tgt;
          st_loss += d * d; st_cnt += 1.f;
          dy[0] = 2.f * d;
        }
      }
      gb3 += dy;
      sts4(dYimg + n * kSY + 4 * g, dy);
      PDS_WAVE_SYNC();

      // ---- backward.  Weight-gradient GEMMs take K = the tile's 16 samples: k-slot (j, h) carries sample 4 h + j, both
      // operands are dword reads of [sample][feature] images (conflict free). ---------------------------------------
      const int r = n, h = g;  // A-operand lane roles
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // dW3 += dY^T H2 (rows = outputs)
        const float av = dYimg[(4 * h + j) * kSY + r];
#pragma unroll
        for (int jt = 0; jt < kNT; ++jt) gW3[jt] = PDS_MFMA(av, H2img[(4 * h + j) * kS + jt * kTW + n], gW3[jt]);
      }
      // dZ2^T = (W3^T dY^T) * act'(H2^T); the k-slot (j, h) carries output 4 h + j = register j of dy
      f32x4 dz2[kNT];
#pragma unroll
      for (int it = 0; it < kNT; ++it) cc[it] = (f32x4)(0.f);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int it = 0; it < kNT; ++it) cc[it] = PDS_MFMA(W3s[((4 * h + j) & (kMaxOut - 1)) * kS + it * kTW + r], dy[j], cc[it]);
#pragma unroll
      for (int it = 0; it < kNT; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dz2[it][q] = cc[it][q] * act_grad<ACT>(h2r[it][q]);
        gb2[it] += dz2[it];
      }
#pragma unroll
      for (int it = 0; it < kNT; ++it) sts4(H2img + n * kS + it * kTW + 4 * g, dz2[it]);  // after the dW3 reads (in order)
      PDS_WAVE_SYNC();
      if constexpr (PDS_WIDE_BF16 != 0) {  // dW2 += dZ2^T H1
        Quad3 qa[kNT], qb[kNT];
#pragma unroll
        for (int i = 0; i < kNT; ++i) {
          f32x4 va, vb;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            va[j] = H2img[(4 * h + j) * kS + i * kTW + r];
            vb[j] = H1img[(4 * h + j) * kS + i * kTW + n];
          }
          qa[i] = split4(va);
          qb[i] = split4(vb);
        }
        outer_bf16<kNT, kNT>(qa, qb, [&](int it, int jt) -> f32x4 & { return gW2[it][jt]; });
      }
#pragma unroll
      for (int j = 0; j < (PDS_WIDE_BF16 != 0 ? 0 : 4); ++j) {  // dW2 += dZ2^T H1
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
          for (int jt = 0; jt < kNT; ++jt) cc[jt] = PDS_MFMA(W2s[(kt * kTW + 4 * h + j) * kS + jt * kTW + r], dz2[kt][j], cc[jt]);
#pragma unroll
      for (int jt = 0; jt < kNT; ++jt) {
        const f32x4 hv = lds4(H1img + n * kS + jt * kTW + 4 * g);  // this lane's own H1 values
#pragma unroll
        for (int q = 0; q < 4; ++q) dz1[jt][q] = cc[jt][q] * act_grad<ACT>(hv[q]);
        gb1[jt] += dz1[jt];
      }
#pragma unroll
      for (int jt = 0; jt < kNT; ++jt) sts4(H1img + n * kS + jt * kTW + 4 * g, dz1[jt]);  // after the dW2 reads
      PDS_WAVE_SYNC();
      if constexpr (PDS_WIDE_BF16 != 0) {  // dW1 += dZ1^T X
        Quad3 qa[kNT];
#pragma unroll
        for (int i = 0; i < kNT; ++i) {
          f32x4 va;
#pragma unroll
          for (int j = 0; j < 4; ++j) va[j] = H1img[(4 * h + j) * kS + i * kTW + r];
          qa[i] = split4(va);
        }
        // two input tiles at a time: the pieces of all NIN of them (6 registers each) next to 16 NIN accumulators do not fit 512
        static_assert(NG1 % 2 == 0, "input tiles come in pairs");
#pragma unroll
        for (int k0 = 0; k0 < NG1; k0 += 2) {
          Quad3 qb2[2];
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            f32x4 vb;
#pragma unroll
            for (int j = 0; j < 4; ++j) vb[j] = Ximg[(4 * h + j) * S1 + (k0 + kk) * kTW + n];
            qb2[kk] = split4(vb);
          }
          outer_bf16<kNT, 2>(qa, qb2, [&](int it, int kk) -> f32x4 & { return gW1[it][k0 + kk]; });
        }
      }
#pragma unroll
      for (int j = 0; j < (PDS_WIDE_BF16 != 0 ? 0 : 4); ++j) {  // dW1 += dZ1^T X
        float av[kNT];
#pragma unroll
        for (int i = 0; i < kNT; ++i) av[i] = H1img[(4 * h + j) * kS + i * kTW + r];
#pragma unroll
        for (int kt = 0; kt < NG1; ++kt) {
          const float bv = Ximg[(4 * h + j) * S1 + kt * kTW + n];
#pragma unroll
          for (int it = 0; it < kNT; ++it) gW1[it][kt] = PDS_MFMA(av[it], bv, gW1[it][kt]);
        }
      }
      PDS_WAVE_SYNC();  // the images are rewritten by the next tile
    }
  }

  if constexpr (LOSS != LOSS_NONE) {
    // ---- this WAVE's partial sums -> partials[wave of the grid][...] (flat parameter layout + statistics) ----------
    float *out = a.partials + wid * a.pstride;
    const Offsets o = offsets(m);
#pragma unroll
    for (int it = 0; it < kNT; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = it * kTW + 4 * g + q;  // row of the C/D layout
#pragma unroll
        for (int jt = 0; jt < NG1; ++jt) {
          const int j = jt * kTW + n;
          if (i < m.h1 && j < m.d_in) out[o.w1 + i * m.d_in + j] = gW1[it][jt][q];
        }
#pragma unroll
        for (int jt = 0; jt < kNT; ++jt) {
          const int j = jt * kTW + n;
          if (i < m.h2 && j < m.h1) out[o.w2 + i * m.h1 + j] = gW2[it][jt][q];
        }
        float v1 = gb1[it][q], v2 = gb2[it][q];  // sum of the per-lane partials over the 16 sample columns of the lane group
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) { v1 += __shfl_xor(v1, d); v2 += __shfl_xor(v2, d); }
        if (n == 0) {
          if (i < m.h1) out[o.b1 + i] = v1;
          if (i < m.h2) out[o.b2 + i] = v2;
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
      }
      float v3 = gb3[q];
#pragma unroll
      for (int d = 8; d >= 1; d >>= 1) v3 += __shfl_xor(v3, d);
      if (n == 0 && i < m.d_out) out[o.b3 + i] = v3;
    }
    float s4[kStats] = {st_loss, st_ratio, st_kl, st_cnt};  // lanes of group 0 hold per-sample sums
#pragma unroll
    for (int q = 0; q < kStats; ++q) {
      float v = s4[q];
      for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d);
      if (lane == 0) out[o.total + q] = v;
    }
  }
}

template <int LOSS, int ACT>
static void launch_nin(int nin, dim3 g, hipStream_t s, const Args &a) {
  const dim3 b(kWideWaves * 64);
  // input tiles in steps of two (d_in <= 96 / 128 / 160 / 192): an all-padding tile costs 4 x 4 MFMAs per layer-1 GEMM
  if (nin <= 6) hipLaunchKernelGGL((mlp_wide_kernel<LOSS, ACT, 6>), g, b, 0, s, a);
  else if (nin <= 8) hipLaunchKernelGGL((mlp_wide_kernel<LOSS, ACT, 8>), g, b, 0, s, a);
  else if (nin <= 10) hipLaunchKernelGGL((mlp_wide_kernel<LOSS, ACT, 10>), g, b, 0, s, a);
  else hipLaunchKernelGGL((mlp_wide_kernel<LOSS, ACT, 12>), g, b, 0, s, a);
}

int wide_grid_blocks(long long B) {
  const long long tiles = (B + kTS - 1) / kTS;
  const long long blocks = (tiles + kWideWaves - 1) / kWideWaves;
  return (int)(blocks < kWideMaxBlocks ? blocks : kWideMaxBlocks);  // one persistent block per CU
}

// -> number of partials written (one per wave of the grid)
int launch_wide(int loss, const Args &a, hipStream_t s) {
  const int nin = (a.m.d_in + kTW - 1) / kTW;
  const int blocks = wide_grid_blocks(a.B);
  const dim3 g(blocks);
  if (loss == LOSS_NONE) { if (a.m.activation == 0) launch_nin<LOSS_NONE, 0>(nin, g, s, a); else launch_nin<LOSS_NONE, 1>(nin, g, s, a); }
  else if (loss == LOSS_PPO) { if (a.m.activation == 0) launch_nin<LOSS_PPO, 0>(nin, g, s, a); else launch_nin<LOSS_PPO, 1>(nin, g, s, a); }
  else { if (a.m.activation == 0) launch_nin<LOSS_MSE, 0>(nin, g, s, a); else launch_nin<LOSS_MSE, 1>(nin, g, s, a); }
  return blocks * kWideWaves;
}

}  // namespace pds_mlp_detail
