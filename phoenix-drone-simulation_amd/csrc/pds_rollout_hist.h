// pds_rollout_hist.h -- ONE launch per rollout for observation histories other than 2 (observation_history_size = H,
// envs/base.py:44, 303-319, 417-431; the reference's experiments/04_history_of_state_action_inputs trains H = 1, 2, 4, 6, 8):
// the actor reads the last H [o, u] halves of every env, H x half <= 192 floats.
//
// Rounds 2-5 ran such a rollout as 8 launches per step (critic, actor, sample, env step, pds_history_advance, V(final_obs),
// record, + copies); csrc/pds_rollout.h -- the one-launch form for H = 2 -- keeps both networks' weights and the observation
// tile in LDS, which at 160 inputs would take 233 KB.  Here:
//   * the ACTOR, the sampling, the env step (step_once of csrc/pds_step.h: the code path of pds_step, same bits), the history
//     update of pds_history_advance (csrc/pds_history.hip, same values) and the episode bookkeeping run in the kernel: one
//     block per 64-env tile, four network waves (16 rows each, csrc/pds_mlp_fwd.h forward16_wide: the code path of
//     pds_mlp_forward, same bits) + one env wave, hand-over through LDS counters as in csrc/pds_rollout.h;
//   * the CRITIC does not: V(o(t)) for all t is one pds_mlp_forward over obs_buf after the kernel (it is off the step's
//     critical path; at H = 2 the in-kernel critic measured 6 % ahead of this, csrc/pds_rollout.h header), and the final
//     histories of the envs the TimeLimit cut (or that finished on the rollout's last step: algs/iwpg/iwpg.py:374-379) go to a
//     small per-env slot list `fin_rows` / `fin_step` that the caller evaluates the same way;
//   * LDS: the actor's image (W1 [64][16 HN + 4]), the history tile [64][16 HN + 4] (the network input, shifted in place by
//     the env wave), the kernel's own [o(k), o(k + 1)] row tile and the finished envs' last rows: 152 KB at HN = 12.
// obs_buf [T + 1, N, H half]: row 0 = the histories on entry (the env's current observation), row t + 1 written by step t.
#pragma once
#include "pds_rollout.h"

namespace pds {

// network input of this lane from a history row: features 16 kt + 4 g + q (one b128 read per input tile)
template <int NIN>
PDS_DEV void gather_hist(const float *row, int d_in, const float *mus, const float *iss, int g, pds_mlpf::f32x4 (&xin)[NIN]) {
#pragma unroll
  for (int kt = 0; kt < NIN; ++kt) {
    const int k0 = kt * 16 + 4 * g;
    const pds_mlpf::f32x4 v = pds_mlpf::lds4(row + k0), mu = pds_mlpf::lds4(mus + k0), is = pds_mlpf::lds4(iss + k0);
#pragma unroll
    for (int q = 0; q < 4; ++q) xin[kt][q] = ((k0 + q < d_in ? v[q] : mu[q]) - mu[q]) * is[q];
  }
}

template <class V_, int HN>
__global__ __launch_bounds__(kRolloutThreads, 1) void rollout_hist_kernel(const RolloutHistArgs ra) {
  using namespace pds_mlpf;
  using V = std::conditional_t<regen_obs_variant<V_>(), StoredOh<V_>, V_>;
  constexpr int D = V::D;                    // the kernel's own row: two halves
  constexpr int HALF = D / 2;
  constexpr int TS = tile_stride<D>();
  constexpr int S1 = kTW * HN + 4;           // row stride of the history tile and of W1's image
  constexpr int RM = merged_reset_variant<V>() ? RM_MERGED : RM_INLINE;
  constexpr int kScratchU4_ = (RM == RM_MERGED) ? kMergedScratchU4 : (inline_coop_variant<V>() ? inline_envs<V, false>() * scratch_stride<V>() : 0);
  __shared__ __attribute__((aligned(16))) float net_pi[net_floats_wide<HN>()];
  __shared__ __attribute__((aligned(16))) float mus[kTW * HN], iss[kTW * HN];
  __shared__ __attribute__((aligned(16))) float hist[kWave * S1];
  __shared__ __attribute__((aligned(16))) float tile[kWave * TS];
  __shared__ __attribute__((aligned(16))) float fin[kWave * D];
  __shared__ __attribute__((aligned(16))) float4 act_all[kWave];
  __shared__ uint32_t queue[kQueueCap];
  __shared__ U4 scratch[kScratchU4_ > 0 ? kScratchU4_ : 1];
  __shared__ int obs_ready, act_ready;
#ifdef PDS_STAMPS
  unsigned long long stamp_[kStampSlots];
#endif
  prefetch_kernargs();
  const StepArgs &a = ra.s;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_env = wave >= kRolloutMlpWaves;
  const int n16 = lane & 15, g = lane >> 4;
  const NetLds wpi = net_lds_wide<HN>(net_pi);
  const long long t = blockIdx.x;  // this block's 64-env tile
  const int T = ra.T, H = ra.H;
  const int HS = H * HALF;         // floats per history row (= the actor's d_in)
  const int d_out = ra.pi.d_out;
  const long long rem0 = a.n - t * kWave;
  const int rows = rem0 >= kWave ? kWave : (int)rem0;

  // ---- prologue: the actor, the statistics and the histories into LDS; env state into the env wave's registers ----
  stage_net_wide<HN>(ra.pi, wpi, tid, kRolloutThreads);
  for (int i = tid; i < kTW * HN; i += kRolloutThreads) {
    const bool on = ra.mean != nullptr && i < HS;
    mus[i] = on ? ra.mean[i] : 0.f;
    iss[i] = on ? 1.0f / (ra.stdv[i] + ra.eps) : 1.f;
  }
  for (int i = tid; i < kWave * S1; i += kRolloutThreads) {
    const int r = i / S1, c = i - r * S1;
    hist[i] = (r < rows && c < HS) ? ra.obs_buf[(t * kWave + r) * HS + c] : 0.f;
  }
  if (tid == 0) { obs_ready = 0; act_ready = 0; }
  unsigned long long call0 = ra.call_offset;
  if (ra.call_base != nullptr) call0 += *ra.call_base;
  __syncthreads();  // (the only block barrier: from here on the roles meet through the counters)

  if (is_env) {
    // ================================ env wave: the tile's 64 envs in registers =================================
#if PDS_ROLLOUT_ENV_PRIO
    __builtin_amdgcn_s_setprio(PDS_ROLLOUT_ENV_PRIO);
#endif
    const long long wave_base = t * kWave;
    const bool active = rem0 >= kWave || lane < (int)rem0;
    const Idx<V> ix{wave_base, active ? (uint32_t)lane : (uint32_t)rem0 - 1u};
    Loaded cur;
    RngKey rk{a.seed_lo, a.seed_hi, 0u, 0u};
    load_env<V>(a, ix, t, cur);
    rk.tick_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.clk.x);
    rk.tick_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.clk.y);
    int parity = __builtin_amdgcn_readfirstlane((int)cur.clk.z) & 1;
    const RngKey rk0 = rk;
    EnvState S;
    unpack_state<V>(a.k, cur, parity, S);
    init_kept_obs<V>(a, rk, ix, S);
    float ep_ret = *at(ra.ep_ret, ix), ep_len = *at(ra.ep_len, ix), st0 = 0.f, st1 = 0.f, st2 = 0.f;
    int qcount = 0, nfin = 0;
    float *hrow = hist + lane * S1;
    for (int s = 0; s < T; ++s) {
      const RolloutHistArgs &rl = *reinterpret_cast<const RolloutHistArgs *>(&reload_args<211, true>(ra.s, s));
      const long long o1 = (long long)s * rl.s.n;
      rollout_wait_ge(&act_ready, kRolloutMlpWaves * (s + 1));  // the network waves have read the histories and written a(s)
      RngKey rks = rk;
      int lane_s = lane;
      if (PDS_STEPK_OPAQUE_KEY) asm volatile("" : "+s"(rks.seed_lo), "+s"(rks.seed_hi), "+v"(lane_s));
      const float4 act = act_all[lane_s];
      StepOut so;
      step_once<V, kWave, RM, false>(rl.s, o1, rks, parity, nullptr, tile, nullptr, queue, scratch, lane_s, wave_base, ix, active, act, S,
                                     qcount, fin, &so PDS_STAMP_ARG);
      parity ^= 1;
      rk.tick_lo += 1u;
      if (rk.tick_lo == 0u) rk.tick_hi += 1u;
      // pds_rollout_record (csrc/pds_train.hip record_kernel)
      const bool dn = (so.done || so.trunc) && active;
      const float er = ep_ret + so.reward, el = ep_len + 1.f;
      if (dn) { st0 += er; st1 += el; st2 += 1.f; }
      ep_ret = dn ? 0.f : er;
      ep_len = dn ? 0.f : el;
      // pds_history_advance (csrc/pds_history.hip), this lane's own row: shift by one half, append the newest half -- of the
      // env's LAST observation where it finished (`fin`: the row that goes to final_obs), of the step's row otherwise
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // (step_once's wave-cooperative writes of `tile` and `fin` are complete)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float *krow = tile + lane * TS;
      const float *newest = dn ? fin + lane * D + HALF : krow + HALF;
      const int keep = HS - HALF;
      // (blocks of reads, then the block's writes: a read-then-write loop over one array serialises on the LDS latency of every
      //  element -- the compiler must assume the two alias --, ~130 cycles x 51 moves for Hover H = 4: 18.7 -> 13 us per step)
      if constexpr (HALF % 4 == 0) {  // 16-byte aligned halves: b128 moves (8 consecutive lanes cover all banks)
        for (int j0 = 0; j0 < keep; j0 += 32) {
          float4 buf[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) buf[u] = (j0 + 4 * u < keep) ? *reinterpret_cast<const float4 *>(hrow + j0 + 4 * u + HALF) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int u = 0; u < 8; ++u) if (j0 + 4 * u < keep) *reinterpret_cast<float4 *>(hrow + j0 + 4 * u) = buf[u];
        }
        float4 nb[HALF / 4];
#pragma unroll
        for (int c = 0; c < HALF / 4; ++c) nb[c] = *reinterpret_cast<const float4 *>(newest + 4 * c);
#pragma unroll
        for (int c = 0; c < HALF / 4; ++c) *reinterpret_cast<float4 *>(hrow + keep + 4 * c) = nb[c];
      } else {
        for (int j0 = 0; j0 < keep; j0 += 16) {
          float buf[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) buf[u] = (j0 + u < keep) ? hrow[j0 + u + HALF] : 0.f;
#pragma unroll
          for (int u = 0; u < 16; ++u) if (j0 + u < keep) hrow[j0 + u] = buf[u];
        }
        float nb[HALF];
#pragma unroll
        for (int c = 0; c < HALF; ++c) nb[c] = newest[c];
#pragma unroll
        for (int c = 0; c < HALF; ++c) hrow[keep + c] = nb[c];
      }
      // the final history of an env the TimeLimit cut bootstraps its path with V (also when it terminated on that step, and
      // for every env that finished on the rollout's last step: algs/iwpg/iwpg.py:374-379) -> the caller's slot list
      const bool want = dn && (so.trunc || s == T - 1);
      unsigned long long wm = __ballot(want);
      if (wm != 0ull) {  // wave-uniform, rare
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        while (wm != 0ull) {
          const int src = __builtin_ctzll(wm);
          wm &= wm - 1ull;
          const int slot = __builtin_amdgcn_readlane(nfin, src);
          if (slot < rl.slots) {
            float *dst = rl.fin_rows + ((long long)slot * rl.s.n + wave_base + src) * HS;
            for (int c = lane; c < HS; c += kWave) dst[c] = hist[src * S1 + c];
            if (lane == 0) rl.fin_step[(long long)slot * rl.s.n + wave_base + src] = s;
          }
        }
        if (want) ++nfin;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // (the rows are overwritten below)
      }
      if (dn) {  // restart: H - 1 copies of the reset row's first half, then its second half (envs/base.py:417-431)
        float k0[HALF], k1[HALF];
#pragma unroll
        for (int c = 0; c < HALF; ++c) { k0[c] = krow[c]; k1[c] = krow[HALF + c]; }
        for (int j = 0; j < H - 1; ++j)
#pragma unroll
          for (int c = 0; c < HALF; ++c) hrow[j * HALF + c] = k0[c];
#pragma unroll
        for (int c = 0; c < HALF; ++c) hrow[keep + c] = k1[c];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // o(s + 1) -> obs_buf[s + 1]: the tile's rows are one contiguous piece of it (wave-cooperative, 256 B per pass)
      {
        float *dst = rl.obs_buf + ((long long)(s + 1) * rl.s.n + wave_base) * HS;
        const int total = rows * HS;
        // 16-byte stores where the rows allow it (H half a multiple of 4: Circle, TakeOff, Hover at H = 4 / 8; the piece of obs_buf
        // 16-byte aligned): the first form of this copy, 4 bytes per lane and store, was 68 store instructions per step at H = 4 --
        // the release fence behind them waits for every one of them
        if ((HS & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
          const int q4 = HS >> 2, total4 = total >> 2;  // float4 per row / in all
          int r = lane / q4, c4 = lane - r * q4;
          for (int idx0 = lane; idx0 < total4; idx0 += 4 * kWave) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              v[u] = (idx0 + u * kWave < total4) ? *reinterpret_cast<const float4 *>(hist + r * S1 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
              c4 += kWave;
              while (c4 >= q4) { c4 -= q4; ++r; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (idx0 + u * kWave < total4) nt_store4(reinterpret_cast<float4 *>(dst) + idx0 + u * kWave, v[u]);
          }
        } else {
          int r = lane / HS, c = lane - r * HS;
          for (int idx0 = lane; idx0 < total; idx0 += 8 * kWave) {  // eight LDS reads in flight, then their stores
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              v[u] = (idx0 + u * kWave < total) ? hist[r * S1 + c] : 0.f;
              c += kWave;
              while (c >= HS) { c -= HS; ++r; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (idx0 + u * kWave < total) nt_store(dst + idx0 + u * kWave, v[u]);
          }
        }
      }
      rollout_post(&obs_ready, lane);  // the histories of step s + 1 are in LDS
    }
    const RolloutHistArgs &rl = *reinterpret_cast<const RolloutHistArgs *>(&reload_args<212, true>(ra.s, T));
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
    return;
  }

  // ================================ network waves: the actor for 16 rows of the tile each ========================
  const int own = wave * 16 + n16;  // this lane's sample row
  const bool own_ok = own < rows;
  const long long env_own = t * kWave + own;
  for (int s = 0; s < T; ++s) {
    const RolloutHistArgs &rl = *reinterpret_cast<const RolloutHistArgs *>(&reload_args<213, true>(ra.s, s));
    const long long o1 = (long long)s * rl.s.n;
    f32x4 x_own[HN];
    // the action noise of step s depends on (env, call) only: drawn while the env wave is still stepping
    // pds_gaussian_sample (csrc/pds_train.hip sample_kernel): counter = (sample id lo, id hi << 8 | block, call lo, call hi)
    float z[4] = {0.f, 0.f, 0.f, 0.f}, sig[4], lsd[4];
    if (g == 0) {
      if (!rl.deterministic) {
        const unsigned long long gid = rl.s.env_id_base + (unsigned long long)env_own;
        const unsigned long long call = call0 + (unsigned long long)s + 1ull;
        const U4 r = philox4x32_10((uint32_t)gid, ((uint32_t)(gid >> 32) << 8) | 0u, (uint32_t)call, (uint32_t)(call >> 32),
                                   (uint32_t)rl.seed, (uint32_t)(rl.seed >> 32));
        box_muller(r.x, r.y, z[0], z[1]);
        box_muller(r.z, r.w, z[2], z[3]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        lsd[q] = (q < d_out) ? rl.log_std[q] : 0.f;
        sig[q] = expf(lsd[q]);
      }
    }
    rollout_wait_ge(&obs_ready, s);  // the histories of step s are in LDS
    gather_hist<HN>(hist + own * S1, HS, mus, iss, g, x_own);
    const bool kh2 = rl.pi.h1 == rl.pi.h2 && last_tile_steps(rl.pi.h1, kNT) == 2;
    f32x4 mu;
    if (rl.pi.activation == 0) mu = kh2 ? forward16_wide<0, HN, S1, 2>(wpi, x_own, n16, g) : forward16_wide<0, HN, S1, 4>(wpi, x_own, n16, g);
    else mu = kh2 ? forward16_wide<1, HN, S1, 2>(wpi, x_own, n16, g) : forward16_wide<1, HN, S1, 4>(wpi, x_own, n16, g);
    if (g == 0) {  // lane n16 owns sample `own`: outputs 0..3 of the actor
      float av[4], lp = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        av[q] = fmaf(sig[q], z[q], mu[q]);
        if (q < d_out) lp += -0.5f * z[q] * z[q] - lsd[q] - 0.91893853320467274178f;
      }
      act_all[own] = make_float4(av[0], av[1], av[2], av[3]);
      if (own_ok) {
        *reinterpret_cast<float4 *>(rl.act_buf + (o1 + env_own) * 4) = make_float4(av[0], av[1], av[2], av[3]);
        rl.logp_buf[o1 + env_own] = lp;
      }
    }
    rollout_post(&act_ready, lane);  // this wave is done with the histories of step s
  }
}

template <class RV_>
inline void launch_rollout_hist_variant(int hn, dim3 grid, hipStream_t s, const RolloutHistArgs &ra) {
  if (hn == 4) hipLaunchKernelGGL((rollout_hist_kernel<RV_, 4>), grid, dim3(kRolloutThreads), 0, s, ra);
  else if (hn == 6) hipLaunchKernelGGL((rollout_hist_kernel<RV_, 6>), grid, dim3(kRolloutThreads), 0, s, ra);
  else if (hn == 8) hipLaunchKernelGGL((rollout_hist_kernel<RV_, 8>), grid, dim3(kRolloutThreads), 0, s, ra);
  else hipLaunchKernelGGL((rollout_hist_kernel<RV_, 12>), grid, dim3(kRolloutThreads), 0, s, ra);
}
template <int TASK>
inline bool launch_rollout_hist_task(const LaunchFlags &f, int hn, dim3 grid, hipStream_t s, const RolloutHistArgs &ra) {
  if (!rollout_hist_supported(TASK, f)) return false;
  const bool full = f.on;
  if constexpr (TASK != PDS_TASK_TAKEOFF) {
    if (f.motor) {
      if (full) launch_rollout_hist_variant<Variant<TASK, true, true, false, true, true, 0, false, false>>(hn, grid, s, ra);
      else launch_rollout_hist_variant<Variant<TASK, true, false, false, false, false, 0, false, false>>(hn, grid, s, ra);
      return true;
    }
  }
  if (full) launch_rollout_hist_variant<Variant<TASK, false, true, false, true, true, 0, false, false>>(hn, grid, s, ra);
  else launch_rollout_hist_variant<Variant<TASK, false, false, false, false, false, 0, false, false>>(hn, grid, s, ra);
  return true;
}

}  // namespace pds
