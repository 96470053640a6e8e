// pds_rollout.h -- ONE launch per rollout (gfx950): the T closed-loop steps
//     o -> standardise -> actor MLP -> a = mu + sigma z, log p -> critic V(o) -> env.step(a) -> buffers
// of the caller's roll_out (algs/iwpg/iwpg.py:350-385, ActorCritic.step algs/core.py:370-393) for a 64-env tile
// per 256-thread block, the env state in registers for the whole rollout (like pds_step_k), the observation tile in
// LDS between the env step and the networks, both networks' weights in LDS.
//
// Round 2 ran a rollout step as 7 launches (critic, actor, sample, env step, V(final_obs), record, + copies), replayed
// from a hipGraph: 63 us per step at 8 192 envs, i.e. launch-bound (the env step itself is 7 us).
// Here a block's four waves split the two network passes of its 64 envs (16 samples each on
// v_mfma_f32_16x16x4_f32, csrc/pds_mlp_fwd.h: the code path of pds_mlp_forward, same bits), wave 0 then steps the 64
// envs (step_once of csrc/pds_step.h: the code path of pds_step / pds_step_k, same bits), and the waves whose rows
// hold a finished env evaluate V(final_obs) for the TimeLimit bootstrap (algs/iwpg/iwpg.py:375-385) out of an
// LDS copy of those rows.  Two block barriers per step; no tensor of the step round-trips through HBM except
// the rollout buffers themselves.
//
// Envs are independent, the policy is frozen during a rollout and the running observation statistics are only
// updated after it (ppo.py), so a block needs nothing from another block for all T steps.
#pragma once
#include "pds_mlp_fwd.h"
#include "pds_step.h"

namespace pds {

#ifndef PDS_ROLLOUT_SKIP
#define PDS_ROLLOUT_SKIP 0  // profiling builds only: 1 = no network passes, 2 = no env step, 4 = no V(final_obs)
#endif
constexpr int kRolloutThreads = 256;

// network input of this lane: features 16 kt + 4 g + q of row `r` of an LDS image with row stride `stride`
template <int NIN>
PDS_DEV void gather_input(const float *img, int stride, int r, int d_in, const float *mus, const float *iss, int g,
                          pds_mlpf::f32x4 (&xin)[NIN]) {
#pragma unroll
  for (int kt = 0; kt < NIN; ++kt) {
    const int k0 = kt * 16 + 4 * g;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = k0 + q;
      const float mu = mus[k], is = iss[k];  // (16 NIN <= 64)
      const float v = (k < d_in) ? img[r * stride + k] : mu;
      xin[kt][q] = (v - mu) * is;  // (pds_mlp.hip: (v - mu) * (1 / (std + eps)); padding features 0)
    }
  }
}

template <class V>
__global__ __launch_bounds__(kRolloutThreads, 1) void rollout_kernel(const RolloutArgs ra) {
  using namespace pds_mlpf;
  constexpr int D = V::D;
  constexpr int TS = tile_stride<D>();
  constexpr int NIN = (D + 15) / 16;
  constexpr int RM = merged_reset_variant<V>() ? RM_MERGED : RM_INLINE;
  constexpr int kScratchU4_ = (RM == RM_MERGED) ? kMergedScratchU4 : (inline_coop_variant<V>() ? inline_envs_per_pass<V>() * kScratchBlocks : 0);
  static_assert(D <= 64, "network input <= 64 features");
  __shared__ __attribute__((aligned(16))) float net_pi[kNetFloats];
  __shared__ __attribute__((aligned(16))) float net_vf[kNetFloats];
  __shared__ __attribute__((aligned(16))) float mus[64], iss[64];
  __shared__ __attribute__((aligned(16))) float tile[kWave * TS];
  __shared__ __attribute__((aligned(16))) float fin[kWave * D];
  __shared__ __attribute__((aligned(16))) float4 act_lds[kWave];
  __shared__ uint32_t done_lds[kWave];
  __shared__ uint32_t queue[kQueueCap];
  __shared__ U4 scratch_all[kScratchU4_ > 0 ? kScratchU4_ : 1];
#ifdef PDS_STAMPS
  unsigned long long stamp_[kStampSlots];
#endif
  prefetch_kernargs();
  const StepArgs &a = ra.s;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, g = lane >> 4;
  const NetLds wpi = net_lds(net_pi), wvf = net_lds(net_vf);
  const long long t = blockIdx.x;  // one 64-env tile per block
  const long long wave_base = t * kWave;
  const long long rem_ = a.n - wave_base;
  const bool active = rem_ >= kWave || lane < (int)rem_;
  const EnvIdx ix{wave_base, active ? (uint32_t)lane : (uint32_t)rem_ - 1u};
  const int rows = rem_ >= kWave ? kWave : (int)rem_;  // envs of this tile
  const int T = ra.T;
  const int d_out = ra.pi.d_out;

  // ---- prologue: networks, statistics and o(0) into LDS; env state into wave 0's registers ---------------
  stage_net(ra.pi, wpi, tid, kRolloutThreads);
  stage_net(ra.vf, wvf, tid, kRolloutThreads);
  if (tid < 64) {
    const bool on = ra.mean != nullptr && tid < D;
    mus[tid] = on ? ra.mean[tid] : 0.f;
    iss[tid] = on ? 1.0f / (ra.stdv[tid] + ra.eps) : 1.f;
  }
  for (int idx = tid; idx < kWave * D; idx += kRolloutThreads) {
    const int r = idx / D, c = idx - r * D;
    tile[r * TS + c] = (r < rows) ? ra.obs0[(wave_base + r) * D + c] : 0.f;
  }
  Loaded cur;
  RngKey rk{a.seed_lo, a.seed_hi, 0u, 0u};
  int parity = 0;
  EnvState S;
  float ep_ret = 0.f, ep_len = 0.f, st0 = 0.f, st1 = 0.f, st2 = 0.f;
  if (wave == 0) {
    load_env<V>(a, ix, t, cur);
    rk.tick_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.clk.x);
    rk.tick_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.clk.y);
    parity = __builtin_amdgcn_readfirstlane((int)cur.clk.z) & 1;
    unpack_state<V>(a.k, cur, parity, S);
    ep_ret = *at(ra.ep_ret, ix);
    ep_len = *at(ra.ep_len, ix);
  }
  const RngKey rk0 = rk;
  unsigned long long call0 = ra.call_offset;
  if (ra.call_base != nullptr) call0 += *ra.call_base;
  __syncthreads();

  const int r16 = wave * 16 + n16;             // this lane's sample row in the tile (network phases)
  const bool row_ok = r16 < rows;
  const long long env16 = wave_base + r16;     // its env
  int qcount = 0;
  for (int s = 0; s < T; ++s) {
    const RolloutArgs &rl = *reinterpret_cast<const RolloutArgs *>(&reload_args<201, true>(s));
    const long long o1 = (long long)s * rl.s.n;
    // ---- networks on o(s): 16 samples per wave --------------------------------------------------------
    if (!(PDS_ROLLOUT_SKIP & 1)) {
      f32x4 xin[NIN];
      gather_input<NIN>(tile, TS, r16, D, mus, iss, g, xin);
      const f32x4 v = (rl.vf.activation == 0) ? forward16<0, NIN>(wvf, xin, n16, g) : forward16<1, NIN>(wvf, xin, n16, g);
      const f32x4 mu = (rl.pi.activation == 0) ? forward16<0, NIN>(wpi, xin, n16, g) : forward16<1, NIN>(wpi, xin, n16, g);
      if (g == 0) {  // lane n16 owns sample r16: outputs 0..3 of the actor, output 0 of the critic
        // pds_gaussian_sample (csrc/pds_train.hip sample_kernel): counter = (sample id lo, id hi << 8 | block, call lo, call hi)
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (!rl.deterministic) {
          const unsigned long long gid = rl.s.env_id_base + (unsigned long long)env16;
          const unsigned long long call = call0 + (unsigned long long)s + 1ull;
          const U4 r = philox4x32_10((uint32_t)gid, ((uint32_t)(gid >> 32) << 8) | 0u, (uint32_t)call, (uint32_t)(call >> 32),
                                     (uint32_t)rl.seed, (uint32_t)(rl.seed >> 32));
          box_muller(r.x, r.y, z[0], z[1]);
          box_muller(r.z, r.w, z[2], z[3]);
        }
        float av[4], lp = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float ls = (j < d_out) ? rl.log_std[j] : 0.f;
          av[j] = fmaf(expf(ls), z[j], mu[j]);
          if (j < d_out) lp += -0.5f * z[j] * z[j] - ls - 0.91893853320467274178f;
        }
        act_lds[r16] = make_float4(av[0], av[1], av[2], av[3]);
        if (row_ok) {
          *reinterpret_cast<float4 *>(rl.act_buf + (o1 + env16) * 4) = make_float4(av[0], av[1], av[2], av[3]);
          rl.logp_buf[o1 + env16] = lp;
          rl.val_buf[o1 + env16] = v[0];
        }
      }
    }
    __syncthreads();  // actions in LDS; every wave is done reading the tile
    // ---- env.step by wave 0 (state in registers) --------------------------------------------------------
    if (wave == 0 && !(PDS_ROLLOUT_SKIP & 2)) {
      const float4 act = act_lds[lane];
      StepOut so;
      step_once<V, kWave, RM, false>(rl.s, o1, rk, parity, nullptr, tile, nullptr, queue, scratch_all, lane, wave_base, ix, active, act,
                                     S, qcount, fin, &so PDS_STAMP_ARG);
      parity ^= 1;
      rk.tick_lo += 1u;
      if (rk.tick_lo == 0u) rk.tick_hi += 1u;
      // pds_rollout_record (csrc/pds_train.hip record_kernel)
      const bool dn = (so.done || so.trunc) && active;
      const float er = ep_ret + so.reward, el = ep_len + 1.f;
      if (dn) { st0 += er; st1 += el; st2 += 1.f; }
      ep_ret = dn ? 0.f : er;
      ep_len = dn ? 0.f : el;
      done_lds[lane] = dn ? 1u : 0u;
    }
    __syncthreads();  // o(s + 1) in the tile, the finished envs' last rows in `fin`
    // ---- V(final_obs) where an env finished (the other rows of fval_buf are never read: pds_gae) -----
    if (!(PDS_ROLLOUT_SKIP & 4)) {
      const bool dn = done_lds[r16] != 0u;
      if (__ballot(dn) != 0ull) {  // wave-uniform: one of this wave's 16 envs finished
        f32x4 xin[NIN];
        gather_input<NIN>(fin, D, r16, D, mus, iss, g, xin);
        const f32x4 v = (rl.vf.activation == 0) ? forward16<0, NIN>(wvf, xin, n16, g) : forward16<1, NIN>(wvf, xin, n16, g);
        if (g == 0 && dn && row_ok) rl.fval_buf[o1 + env16] = v[0];
      }
    }
    // (no barrier: `fin` / done_lds are rewritten by wave 0 only after the next step's first barrier)
  }
  // ---- epilogue: V(o(T)), env state and episode bookkeeping back to HBM -----------------------------------
  {
    const RolloutArgs &rl = *reinterpret_cast<const RolloutArgs *>(&reload_args<202, true>(T));
    f32x4 xin[NIN];
    gather_input<NIN>(tile, TS, r16, D, mus, iss, g, xin);
    const f32x4 v = (rl.vf.activation == 0) ? forward16<0, NIN>(wvf, xin, n16, g) : forward16<1, NIN>(wvf, xin, n16, g);
    if (g == 0 && row_ok) rl.last_val[env16] = v[0];
    if (wave == 0) {
      if (active) {
        store_state<V>(rl.s, ix, parity, S, true);
        *at(rl.ep_ret, ix) = ep_ret;
        *at(rl.ep_len, ix) = ep_len;
      }
      advance_clock(rl.s.st.clk, t, rk0, parity, (uint32_t)T, lane);
      for (int d = 32; d >= 1; d >>= 1) { st0 += __shfl_xor(st0, d); st1 += __shfl_xor(st1, d); st2 += __shfl_xor(st2, d); }
      if (lane == 0 && st2 != 0.f) {
        atomicAdd(rl.stats + 0, st0);
        atomicAdd(rl.stats + 1, st1);
        atomicAdd(rl.stats + 2, st2);
      }
    }
  }
}

// The variants the fused rollout is built for: control_mode PWM, no latency, no Kalman hold, no ground effect;
// {lean, reference default (DR + thrust noise + observation noise)} x {with, without motor dynamics}.
template <int TASK>
inline bool launch_rollout_task(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra) {
  if (f.ctrl != 0 || f.lat || f.hold || f.ge) return false;
  const bool lean = !f.dr && !f.tn && !f.on, full = f.dr && f.tn && f.on;
  if (!lean && !full) return false;
  if (TASK == PDS_TASK_TAKEOFF && f.motor) return false;
#define PDS_ROLLOUT_LAUNCH(M, X) hipLaunchKernelGGL((rollout_kernel<Variant<TASK, M, X, false, X, X, 0, false, false>>), grid, dim3(kRolloutThreads), 0, s, ra)
  if constexpr (TASK == PDS_TASK_TAKEOFF) {
    if (full) PDS_ROLLOUT_LAUNCH(false, true); else PDS_ROLLOUT_LAUNCH(false, false);
  } else {
    if (f.motor) { if (full) PDS_ROLLOUT_LAUNCH(true, true); else PDS_ROLLOUT_LAUNCH(true, false); }
    else { if (full) PDS_ROLLOUT_LAUNCH(false, true); else PDS_ROLLOUT_LAUNCH(false, false); }
  }
#undef PDS_ROLLOUT_LAUNCH
  return true;
}

}  // namespace pds
