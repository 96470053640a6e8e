// pds_rollout.h -- ONE launch per rollout (gfx950): the T closed-loop steps
//     o -> standardise -> actor MLP -> a = mu + sigma z, log p -> critic V(o) -> env.step(a) -> buffers
// of the caller's roll_out (algs/iwpg/iwpg.py:350-385, ActorCritic.step algs/core.py:370-393), the env state in
// registers for the whole rollout (like pds_step_k), the observation tile in LDS between the env step and the
// networks, both networks' weights in LDS.
//
// Round 2 ran a rollout step as 7 launches (critic, actor, sample, env step, V(final_obs), record, + copies), replayed
// from a hipGraph: 63 us per step at 8 192 envs, i.e. launch-bound (the env step itself is 7 us).
// Round 3 gave every 64-env tile one 256-thread block whose four waves ran the networks, then WAITED while wave 0
// stepped the envs, then evaluated V(final_obs): 17 us per step, 7.5 of them one wave's env step with three waves idle
// at the barrier, and one tile per CU (101 KB of LDS, most of it the two weight images).
//
// Round 4 -- wave roles, no block barrier inside the loop.  320 threads per tile:
//   M0..M3 (network waves, rows 16 w .. 16 w + 15, v_mfma_f32_16x16x4_f32 through csrc/pds_mlp_fwd.h -- the code path of
//           pds_mlp_forward, same bits):   wait for o(t) -> read their rows into registers -> actor -> sample ->
//           action into LDS, SIGNAL; then, while the env wave steps: the critic V(o(t)) and V(final_obs) of the envs that
//           finished in step t - 1;
//   E      (wave 4: the 64 envs of the tile in registers, step_once of csrc/pds_step.h -- the code path of pds_step /
//           pds_step_k, same bits):        wait for the four action signals -> env.step -> o(t + 1), the finished
//           envs' last rows and flags into LDS, SIGNAL.
// Only the actor is on the critical path of a step (action -> env step -> next observation); the critic, the TimeLimit
// bootstrap V(final_obs) (algs/iwpg/iwpg.py:375-385), the action-noise draws of the NEXT step and all buffer writes of
// the network waves run while the env wave steps.  Measured and kept out: the critic as one pds_mlp_forward over
// obs_buf after the kernel (0.97 vs 0.91 ms at 8 192 x 64), the actor's weight operands resident in registers
// (a forward16 that keeps its 29 b128 weight operands in 116 registers: no change, 14.0 us per step either way -- the pass is a dependent chain of
// 110 MFMAs + epilogues, not LDS-bound).
// Waves w and w + 4 of a block share a SIMD (profiles/r03_mlp_microbench.txt), and a wave that
// streams MFMAs halves the vector-ALU rate of its SIMD-mate (a first version with two tiles per block, every env wave
// next to a busy network wave: 23 us per step against 17.7 for round 3's kernel), so M0 -- the env wave's mate -- only
// works while E waits: its rows' critic pass is taken by M1, their V(final_obs) by M2.
// Hand-over through monotonic counters in LDS (release / acquire, s_sleep loop; all five waves of a block are resident
// together, so the waits cannot deadlock); every LDS image has exactly one writer role, and every reader finishes with
// it before it posts the signal its writer waits for.
//
// Envs are independent, the policy is frozen during a rollout and the running observation statistics are only
// updated after it (ppo.py), so a tile needs nothing from another tile for all T steps.
#pragma once
#include "pds_mlp_fwd.h"
#include "pds_step.h"

namespace pds {

#ifndef PDS_ROLLOUT_TWO_TEAMS_ABOVE
#define PDS_ROLLOUT_TWO_TEAMS_ABOVE 256  // tiles (one per CU)
#endif
constexpr int kRolloutTwoTeamsAbove = PDS_ROLLOUT_TWO_TEAMS_ABOVE;
// Network wave 0 shares its SIMD with the env wave.  Where the env wave's step is long (observation noise: 20 k cycles per step,
// 14 k of them the in-place reset of the finished envs) wave 0 stays quiet while it runs and waves 1 / 2 take its critic
// passes; where it is short (lean variants: 8 k cycles) the network waves' shadow work is what bounds the step, and wave 0
// does its own share (same box, 8 192 x 64 lean: 0.91 -> 0.76 ms; default config 0.96 -> 1.05 the other way;
// profiles/r04_rollout_timing.txt).
#ifndef PDS_ROLLOUT_M0_QUIET
#define PDS_ROLLOUT_M0_QUIET (V::ON)
#endif
#ifndef PDS_ROLLOUT_ENV_PRIO
#define PDS_ROLLOUT_ENV_PRIO 3
#endif
#ifndef PDS_ROLLOUT_ACTOR_PRIO
#define PDS_ROLLOUT_ACTOR_PRIO 0
#endif
#ifndef PDS_ROLLOUT_SKIP
#define PDS_ROLLOUT_SKIP 0  // profiling builds only: 1 = no actor / critic passes, 2 = no env step, 4 = no V(final_obs)
#endif
constexpr int kRolloutMlpWaves = 4;               // network waves per block (16 rows of the tile each)
constexpr int kRolloutTeamWaves = kRolloutMlpWaves + 1;    // + the env wave: one TEAM per 64-env tile
constexpr int kRolloutThreads = kWave * kRolloutTeamWaves;  // threads per team

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

// hand-over counters (LDS, monotonic)
PDS_DEV void rollout_wait_ge(int *flag, int need) {
  while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < need)
    __builtin_amdgcn_s_sleep(1);
}
PDS_DEV void rollout_post(int *flag, int lane) {  // +1, after every lane's LDS accesses of this phase
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// TEAMS: tiles (= teams of 4 network waves + 1 env wave) per block.  One team per block is the latency form (8 192 envs:
// 128 blocks on 256 CUs); with more tiles than CUs the network waves' idle time -- they work 5-9 of a step's 14 us -- is
// what a second team on the same CU fills: the two teams share nothing but the weight images (84 of the block's 124 KB
// of LDS) and the CU.  (Ten waves per CU = three per SIMD on two SIMDs: the 168-register cap, which the env wave only
// meets because it reads the kept noisy observation from memory, StoredOh, like the K-step kernel.)
template <class V_, int TEAMS>
__global__ __launch_bounds__(kRolloutThreads * TEAMS, 1) void rollout_kernel(const RolloutArgs ra) {
  using namespace pds_mlpf;
  using V = std::conditional_t<regen_obs_variant<V_>(), StoredOh<V_>, V_>;
  constexpr int D = V::D;
  constexpr int TS = tile_stride<D>();
  constexpr int NIN = (D + 15) / 16;
  constexpr int RM = merged_reset_variant<V>() ? RM_MERGED : RM_INLINE;
  constexpr int kScratchU4_ = (RM == RM_MERGED) ? kMergedScratchU4 : (inline_coop_variant<V>() ? inline_envs<V, false>() * scratch_stride<V>() : 0);
  static_assert(D <= 64, "network input <= 64 features");
  __shared__ __attribute__((aligned(16))) float net_pi[kNetFloats];
  __shared__ __attribute__((aligned(16))) float net_vf[kNetFloats];
  __shared__ __attribute__((aligned(16))) float mus[64], iss[64];
  __shared__ __attribute__((aligned(16))) float tile_all[TEAMS][kWave * TS];
  __shared__ __attribute__((aligned(16))) float fin_all[TEAMS][kWave * D];
  __shared__ __attribute__((aligned(16))) float4 act_all[TEAMS][kWave];
  __shared__ uint32_t done_all[TEAMS][kWave];
  __shared__ uint32_t queue_all[TEAMS][kQueueCap];
  __shared__ U4 scratch_all[TEAMS][kScratchU4_ > 0 ? kScratchU4_ : 1];
  __shared__ int obs_ready[TEAMS], act_ready[TEAMS];
#ifdef PDS_STAMPS
  unsigned long long stamp_[kStampSlots];
#endif
  prefetch_kernargs();
  const StepArgs &a = ra.s;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int team = __builtin_amdgcn_readfirstlane(tid / kRolloutThreads);                   // wave-uniform
  const int wave = __builtin_amdgcn_readfirstlane((tid - team * kRolloutThreads) >> 6);     // wave within its team
  const bool is_env = wave >= kRolloutMlpWaves;
  constexpr int kThreads = kRolloutThreads * TEAMS;
  const int n16 = lane & 15, g = lane >> 4;
  const NetLds wpi = net_lds(net_pi), wvf = net_lds(net_vf);
  const long long ntiles = (a.n + kWave - 1) / kWave;
  const long long tile0 = (long long)blockIdx.x * TEAMS;  // first 64-env tile of this block
  const int T = ra.T;
  const int d_out = ra.pi.d_out;
  auto tile_rows = [&](int j) -> int {  // envs of tile j of this block (0: the last block of an odd tile count)
    const long long tt = tile0 + j;
    if (tt >= ntiles) return 0;
    const long long rem = a.n - tt * kWave;
    return rem >= kWave ? kWave : (int)rem;
  };

  // ---- prologue: networks, statistics and o(0) into LDS; env state into the env waves' registers --------------
  stage_net(ra.pi, wpi, tid, kThreads);
  stage_net(ra.vf, wvf, tid, kThreads);
  if (tid < 64) {
    const bool on = ra.mean != nullptr && tid < D;
    mus[tid] = on ? ra.mean[tid] : 0.f;
    iss[tid] = on ? 1.0f / (ra.stdv[tid] + ra.eps) : 1.f;
  }
  if (tid < TEAMS) { obs_ready[tid] = 0; act_ready[tid] = 0; }
#pragma unroll
  for (int j = 0; j < TEAMS; ++j) {
    const int rows = tile_rows(j);
    for (int idx = tid; idx < kWave * D; idx += kThreads) {
      const int r = idx / D, c = idx - r * D;
      tile_all[j][r * TS + c] = (r < rows) ? ra.obs0[((tile0 + j) * kWave + r) * D + c] : 0.f;
    }
    if (tid < kWave) done_all[j][tid] = 0u;
  }
  unsigned long long call0 = ra.call_offset;
  if (ra.call_base != nullptr) call0 += *ra.call_base;
  __syncthreads();  // (the only block barrier: from here on the roles meet through the counters)

  if (is_env) {
    // ================================ env wave: one tile's 64 envs in registers =================================
    const int grp = team;
#if PDS_ROLLOUT_ENV_PRIO
    __builtin_amdgcn_s_setprio(PDS_ROLLOUT_ENV_PRIO);  // the env wave's instructions before its SIMD-mates' (network waves)
#endif
    const long long t = tile0 + grp;
    if (t >= ntiles) return;
    const long long wave_base = t * kWave;
    const long long rem_ = a.n - wave_base;
    const bool active = rem_ >= kWave || lane < (int)rem_;
    const Idx<V> ix{wave_base, active ? (uint32_t)lane : (uint32_t)rem_ - 1u};
    float *tile = tile_all[grp], *fin = fin_all[grp];
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
    int qcount = 0;
#ifdef PDS_ROLLOUT_TIMING
    unsigned long long tw = 0, ts = 0, tp = 0, tph = 0, tout = 0, trst = 0, tfin = 0, tev = 0, tfill = 0, teval = 0;
#endif
    for (int s = 0; s < T; ++s) {
      const RolloutArgs &rl = *reinterpret_cast<const RolloutArgs *>(&reload_args<201, true>(ra.s, s));
      const long long o1 = (long long)s * rl.s.n;
#ifdef PDS_ROLLOUT_TIMING
      const unsigned long long q0 = __builtin_amdgcn_s_memtime();
#endif
      rollout_wait_ge(&act_ready[grp], kRolloutMlpWaves * (s + 1));  // the network waves have read o(s) and written a(s)
#ifdef PDS_ROLLOUT_TIMING
      const unsigned long long q1 = __builtin_amdgcn_s_memtime();
#endif
      if (!(PDS_ROLLOUT_SKIP & 2)) {
        // (opaque per-iteration copies of the seed and the lane index: see step_k_kernel -- their loop-invariant
        //  derivatives would otherwise be formed in the loop header and spilled)
        RngKey rks = rk;
        int lane_s = lane;
        if (PDS_STEPK_OPAQUE_KEY) asm volatile("" : "+s"(rks.seed_lo), "+s"(rks.seed_hi), "+v"(lane_s));
        const float4 act = act_all[grp][lane_s];
        StepOut so;
#ifdef PDS_STAMPS_RESET
        stamp_[8] = 0; stamp_[9] = 0;
#endif
        step_once<V, kWave, RM, false>(rl.s, o1, rks, parity, nullptr, tile, nullptr, queue_all[grp], scratch_all[grp], lane_s,
                                       wave_base, ix, active, act, S, qcount, fin, &so PDS_STAMP_ARG);
        parity ^= 1;
        rk.tick_lo += 1u;
        if (rk.tick_lo == 0u) rk.tick_hi += 1u;
        // pds_rollout_record (csrc/pds_train.hip record_kernel)
        const bool dn = (so.done || so.trunc) && active;
        const float er = ep_ret + so.reward, el = ep_len + 1.f;
        if (dn) { st0 += er; st1 += el; st2 += 1.f; }
        ep_ret = dn ? 0.f : er;
        ep_len = dn ? 0.f : el;
        // V(final_obs) is the bootstrap of an episode the TimeLimit cut (algs/iwpg/iwpg.py:374-379: also when it terminated
        // on the same step); one that only terminated bootstraps with 0 and pds_gae never reads its fval entry: only the
        // truncated envs ask the network waves for a pass (a young policy ends ~5 % of its episodes per step, nearly all of
        // them by termination).  The LAST step of the rollout hands over every finished env: a caller that mirrors the
        // reference's epoch-end cut (`epoch_ended` takes V(o) for a terminated path too; ppo.py reset_each_rollout) reads it.
        done_all[grp][lane] = ((so.trunc || (so.done && s == T - 1)) && active) ? 1u : 0u;
      }
#ifdef PDS_ROLLOUT_TIMING
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long q2 = __builtin_amdgcn_s_memtime();
#endif
      rollout_post(&obs_ready[grp], lane);  // o(s + 1) in the tile, the finished envs' last rows in `fin`, flags
#ifdef PDS_ROLLOUT_TIMING
      const unsigned long long q3 = __builtin_amdgcn_s_memtime();
      tw += q1 - q0; ts += q2 - q1; tp += q3 - q2;
#ifdef PDS_STAMPS
      if (!(PDS_ROLLOUT_SKIP & 2)) { tph += stamp_[3] - q1; tout += stamp_[4] - stamp_[3]; trst += stamp_[5] - stamp_[4]; tfin += stamp_[6] - stamp_[4]; tev += stamp_[7] - stamp_[6]; }
#ifdef PDS_STAMPS_RESET
      tfill += stamp_[8]; teval += stamp_[9];
#endif
#endif
#endif
    }
#ifdef PDS_ROLLOUT_TIMING
    if (blockIdx.x == 0 && team == 0 && lane == 0) { ra.stats[8] = (float)tw / T; ra.stats[9] = (float)ts / T; ra.stats[10] = (float)tp / T;
      ra.stats[11] = (float)tph / T; ra.stats[12] = (float)tout / T; ra.stats[13] = (float)trst / T; ra.stats[14] = (float)tfin / T; ra.stats[15] = (float)tev / T; ra.stats[48] = (float)tfill / T; ra.stats[49] = (float)teval / T; }
#endif
    const RolloutArgs &rl = *reinterpret_cast<const RolloutArgs *>(&reload_args<202, true>(ra.s, T));
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

  // ================================ network waves: 16 rows of the tile each ====================================
  // rows 16 w .. 16 w + 15 are wave w's; the critic of wave 0's rows runs on wave 1 and their V(final_obs) on wave 2
  // (wave 0 shares its SIMD with the env wave: it only works while that one waits).
  const int j = team;
  const int rows = tile_rows(j);
  if (rows == 0) return;  // (the last block of an odd tile count: this team has no env wave to wait for)
  const int own = wave * 16 + n16;  // this lane's sample row
  const bool own_ok = own < rows, r0_ok = n16 < rows;
  const long long env_own = (tile0 + j) * kWave + own, env_r0 = (tile0 + j) * kWave + n16;
#ifdef PDS_ROLLOUT_TIMING
  unsigned long long mw = 0, mg = 0, ma = 0, mp = 0, mc = 0, mpre = 0;
#endif
  for (int s = 0; s <= T; ++s) {  // s == T: only V(o(T)) and the last step's V(final_obs)
#ifdef PDS_ROLLOUT_TIMING
    const unsigned long long r00 = __builtin_amdgcn_s_memtime();
#endif
    const RolloutArgs &rl = *reinterpret_cast<const RolloutArgs *>(&reload_args<203, true>(ra.s, s));
    const long long o1 = (long long)s * rl.s.n;
    f32x4 x_own[NIN], x_r0[NIN], f_own[NIN], f_r0[NIN];
    // the action noise of step s depends on (env, call) only: drawn while the env wave is still stepping
    // pds_gaussian_sample (csrc/pds_train.hip sample_kernel): counter = (sample id lo, id hi << 8 | block, call lo, call hi)
    float z[4] = {0.f, 0.f, 0.f, 0.f}, sig[4], lsd[4];
    if (s < T && g == 0) {
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
#ifdef PDS_ROLLOUT_TIMING
    const unsigned long long r0 = __builtin_amdgcn_s_memtime();
#endif
    rollout_wait_ge(&obs_ready[j], s);  // o(s) and the outcome of step s - 1 are in LDS
#ifdef PDS_ROLLOUT_TIMING
    const unsigned long long r1 = __builtin_amdgcn_s_memtime();
#endif
#if PDS_ROLLOUT_ACTOR_PRIO
    __builtin_amdgcn_s_setprio(PDS_ROLLOUT_ACTOR_PRIO);  // the actor pass is on the step's critical path, the critic passes are not
#endif
    gather_input<NIN>(tile_all[j], TS, own, D, mus, iss, g, x_own);
    constexpr bool kQuiet = PDS_ROLLOUT_M0_QUIET;
    if (kQuiet && wave == 1) gather_input<NIN>(tile_all[j], TS, n16, D, mus, iss, g, x_r0);
    const bool skip_fin = (PDS_ROLLOUT_SKIP & 4) != 0;
    const bool dn_own = done_all[j][own] != 0u;                                   // (step s - 1; zeros before the first step)
    const bool any_own = (wave != 0 || !kQuiet) && __ballot(dn_own) != 0ull && !skip_fin;      // wave-uniform: one of these 16 envs finished
    const bool dn_r0 = done_all[j][n16] != 0u;
    const bool any_r0 = kQuiet && wave == 2 && __ballot(dn_r0) != 0ull && !skip_fin;
    if (any_own) gather_input<NIN>(fin_all[j], D, own, D, mus, iss, g, f_own);
    if (any_r0) gather_input<NIN>(fin_all[j], D, n16, D, mus, iss, g, f_r0);
#ifdef PDS_ROLLOUT_TIMING
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long r2 = __builtin_amdgcn_s_memtime();
    unsigned long long r3 = r2, r4 = r2;
#endif
    if (s < T) {
      if (!(PDS_ROLLOUT_SKIP & 1)) {
        const f32x4 mu = (rl.pi.activation == 0) ? forward16_shape<0, NIN>(wpi, rl.pi, x_own, n16, g) : forward16_shape<1, NIN>(wpi, rl.pi, x_own, n16, g);
        if (g == 0) {  // lane n16 owns sample `own`: outputs 0..3 of the actor
          float av[4], lp = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            av[q] = fmaf(sig[q], z[q], mu[q]);
            if (q < d_out) lp += -0.5f * z[q] * z[q] - lsd[q] - 0.91893853320467274178f;
          }
          act_all[j][own] = make_float4(av[0], av[1], av[2], av[3]);
          if (own_ok) {
            *reinterpret_cast<float4 *>(rl.act_buf + (o1 + env_own) * 4) = make_float4(av[0], av[1], av[2], av[3]);
            rl.logp_buf[o1 + env_own] = lp;
          }
        }
      }
#ifdef PDS_ROLLOUT_TIMING
      r3 = __builtin_amdgcn_s_memtime();
#endif
      rollout_post(&act_ready[j], lane);  // this wave is done with the tile, `fin` and the flags of step s - 1
#ifdef PDS_ROLLOUT_TIMING
      r4 = __builtin_amdgcn_s_memtime();
#endif
    }
#if PDS_ROLLOUT_ACTOR_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    // ---- off the critical path: the env wave is stepping (wave 0, its SIMD-mate, stays quiet) ------------------
    auto critic = [&](const f32x4 (&x)[NIN]) -> float {
      const f32x4 v = (rl.vf.activation == 0) ? forward16_shape<0, NIN>(wvf, rl.vf, x, n16, g) : forward16_shape<1, NIN>(wvf, rl.vf, x, n16, g);
      return v[0];
    };
    if ((wave != 0 || !kQuiet) && !(PDS_ROLLOUT_SKIP & 1)) {
      const float v = critic(x_own);
      if (g == 0 && own_ok) {
        if (s < T) rl.val_buf[o1 + env_own] = v;
        else rl.last_val[env_own] = v;
      }
    }
    if (kQuiet && wave == 1 && !(PDS_ROLLOUT_SKIP & 1)) {
      const float v = critic(x_r0);
      if (g == 0 && r0_ok) {
        if (s < T) rl.val_buf[o1 + env_r0] = v;
        else rl.last_val[env_r0] = v;
      }
    }
    // V(final_obs) where an env finished in step s - 1 (the other rows of fval_buf are never read: pds_gae)
    if (any_own) {
      const float v = critic(f_own);
      if (g == 0 && dn_own && own_ok) rl.fval_buf[o1 - rl.s.n + env_own] = v;
    }
    if (any_r0) {
      const float v = critic(f_r0);
      if (g == 0 && dn_r0 && r0_ok) rl.fval_buf[o1 - rl.s.n + env_r0] = v;
    }
#ifdef PDS_ROLLOUT_TIMING
    const unsigned long long r5 = __builtin_amdgcn_s_memtime();
    if (s > 0 && s < T) { mpre += r0 - r00; mw += r1 - r0; mg += r2 - r1; ma += r3 - r2; mp += r4 - r3; mc += r5 - r4; }
#endif
  }
#ifdef PDS_ROLLOUT_TIMING
  // (diagnostic builds: the caller's `stats` must have room for 64 floats -- profiles/tools/rollout_timing.py)
  if (blockIdx.x == 0 && team == 0 && lane == 0) {
    float *o = ra.stats + 16 + 8 * wave;
    o[0] = (float)mpre / (T - 1); o[1] = (float)mw / (T - 1); o[2] = (float)mg / (T - 1); o[3] = (float)ma / (T - 1); o[4] = (float)mp / (T - 1); o[5] = (float)mc / (T - 1);
  }
#endif
}

// The variants the fused rollout is built for (rollout_supported in csrc/pds_types.h states the same rule for the host):
//   control_mode PWM, no latency ring, no Kalman hold: every combination of domain randomisation / thrust noise /
//     observation noise (the reference's ctor arguments are independent: envs/base.py:26-48) x {with, without motor dynamics};
//   the PID control modes (what the reference's exp-07 trains: experiments/07_control_structure_hypothesis/
//     run_control_structures.py:53-61, envs/control.py:120-287) and the latency ring (envs/agents.py:267-276), also together:
//     {lean, reference default (DR + thrust noise + observation noise)} x {with, without motor dynamics};
//   the Kalman hold (observation_frequency < sim_freq, envs/hover.py:134-156) with control_mode PWM: observation noise with
//     {none, both} of DR + thrust noise x {with, without motor dynamics};
//   the ground effect: TakeOff with control_mode PWM, every noise setting (round 6); nowhere else.
// `grid.x` = number of 64-env tiles; more tiles than CUs: two teams per block.
template <class RV_>
inline void launch_rollout_variant(dim3 grid, hipStream_t s, const RolloutArgs &ra) {
  if (grid.x > (unsigned)kRolloutTwoTeamsAbove)
    hipLaunchKernelGGL((rollout_kernel<RV_, 2>), dim3((grid.x + 1) / 2), dim3(2 * kRolloutThreads), 0, s, ra);
  else
    hipLaunchKernelGGL((rollout_kernel<RV_, 1>), grid, dim3(kRolloutThreads), 0, s, ra);
}
// motor dynamics x {lean, full}
template <int TASK, int CTRL, bool LAT>
inline bool launch_rollout_lean_or_full(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra) {
  const bool lean = !f.dr && !f.tn && !f.on, full = f.dr && f.tn && f.on;
  if (!lean && !full) return false;
  if (f.motor) {
    if (full) launch_rollout_variant<Variant<TASK, true, true, false, true, true, CTRL, LAT, false>>(grid, s, ra);
    else launch_rollout_variant<Variant<TASK, true, false, false, false, false, CTRL, LAT, false>>(grid, s, ra);
  } else {
    if (full) launch_rollout_variant<Variant<TASK, false, true, false, true, true, CTRL, LAT, false>>(grid, s, ra);
    else launch_rollout_variant<Variant<TASK, false, false, false, false, false, CTRL, LAT, false>>(grid, s, ra);
  }
  return true;
}
// control_mode PWM without latency / hold: all eight noise settings (GE: the ground-effect extension, TakeOff only)
template <int TASK, bool MOTOR, bool GE = false>
inline void launch_rollout_pwm(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra) {
#define PDS_ROLLOUT_CASE(DR, TN, ON) \
  if (f.dr == DR && f.tn == TN && f.on == ON) return launch_rollout_variant<Variant<TASK, MOTOR, DR, GE, TN, ON, 0, false, false>>(grid, s, ra)
  PDS_ROLLOUT_CASE(false, false, false); PDS_ROLLOUT_CASE(true, true, true);
  PDS_ROLLOUT_CASE(true, false, false); PDS_ROLLOUT_CASE(false, true, false); PDS_ROLLOUT_CASE(false, false, true);
  PDS_ROLLOUT_CASE(true, true, false); PDS_ROLLOUT_CASE(true, false, true); PDS_ROLLOUT_CASE(false, true, true);
#undef PDS_ROLLOUT_CASE
}
template <int TASK, bool MOTOR>
inline bool launch_rollout_hold(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra) {
  if (!f.on || f.dr != f.tn) return false;
  if (f.dr) launch_rollout_variant<Variant<TASK, MOTOR, true, false, true, true, 0, false, true>>(grid, s, ra);
  else launch_rollout_variant<Variant<TASK, MOTOR, false, false, false, true, 0, false, true>>(grid, s, ra);
  return true;
}
// The families are instantiated in translation units of their own (csrc/pds_rollout_<task>[_pwm|_lat].hip: the build compiles
// them in parallel); launch_rollout_task is the dispatcher in csrc/pds_rollout_<task>.hip.
template <int TASK>
inline bool launch_rollout_pwm_family(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra) {
  constexpr bool kMotor = TASK != PDS_TASK_TAKEOFF;  // (TakeOff + motor dynamics: only with the latency ring)
  if (f.ge) {  // round 6: BasePhysics.calculate_ground_effect (envs/physics.py:27-58) on the task it matters for (BASELINE config 4)
    if constexpr (TASK == PDS_TASK_TAKEOFF) { launch_rollout_pwm<TASK, false, true>(f, grid, s, ra); return true; }
    return false;
  }
  if constexpr (kMotor) { if (f.motor) { launch_rollout_pwm<TASK, true>(f, grid, s, ra); return true; } }
  launch_rollout_pwm<TASK, false>(f, grid, s, ra);
  return true;
}
template <int TASK>
inline bool launch_rollout_lat_family(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra) {
  if (f.ctrl == 0) return launch_rollout_lean_or_full<TASK, 0, true>(f, grid, s, ra);
  if constexpr (TASK != PDS_TASK_TAKEOFF) {  // TakeOff fixes control_mode = 'PWM' (envs/takeoff.py:225)
    if (f.ctrl == 1) return launch_rollout_lean_or_full<TASK, 1, true>(f, grid, s, ra);
    return launch_rollout_lean_or_full<TASK, 2, true>(f, grid, s, ra);
  }
  return false;
}
template <int TASK>
inline bool launch_rollout_pid_hold_family(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra) {
  constexpr bool kMotor = TASK != PDS_TASK_TAKEOFF;
  if (f.hold) {
    if constexpr (kMotor) { if (f.motor) return launch_rollout_hold<TASK, true>(f, grid, s, ra); }
    return launch_rollout_hold<TASK, false>(f, grid, s, ra);
  }
  if constexpr (TASK != PDS_TASK_TAKEOFF) {
    if (f.ctrl == 1) return launch_rollout_lean_or_full<TASK, 1, false>(f, grid, s, ra);
    if (f.ctrl == 2) return launch_rollout_lean_or_full<TASK, 2, false>(f, grid, s, ra);
  }
  return false;
}

}  // namespace pds
