// pds_types.h -- data layout shared by the kernels and the host side of libpds_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/pds.h"
#include "pds_device.h"

namespace pds {

#ifndef PDS_BLOCK
#define PDS_BLOCK 256
#endif
constexpr int kBlock = PDS_BLOCK;  // 4 waves; each wave owns a private LDS tile (no block barrier needed)
#ifndef PDS_MERGED_RESET
#define PDS_MERGED_RESET 1  // A/B: 0 = deferred drain everywhere
#endif
constexpr int kWave = 64;
// Rows of the per-wave LDS observation tile: the step kernel exists with a full tile (64 rows,
// 43 KB LDS per block for Hover => 3 blocks per CU) and, for the variants without observation
// noise, with a half tile (32 rows: the wave stages and flushes its 64 rows in two passes, each
// still one contiguous 16 B-aligned region of the [N, D] tensor => 4 blocks per CU).  Measured on
// MI355X (profiles/r01_tile_rows.txt): the half tile wins when the grid is between one and about
// 2.7 rounds of the 3-blocks-per-CU residency (Circle 262 144 + PT1 + DR: 20.8 vs 24.9 us); the full
// tile wins by 1-2 % from 2^20 envs up (fewer registers, one flush).  pds_step picks per launch.
constexpr int kHalfTileRows = 32;
constexpr int kFullTileBlocksPerCU = 3;  // x hipDeviceProp.multiProcessorCount (256 on MI355X), queried in pds_create
constexpr int kQueueCap = 64;    // deferred-reset queue entries per wave (LDS): one tile
constexpr int kRefPoints = 300;  // envs/circle.py:48, envs/takeoff.py:43
constexpr int kStaggerBytes = 4352;  // 17 x 256 B between consecutive state arrays in the slab

constexpr int kStampSlots = 10;  // diagnostic builds (-DPDS_STAMPS): s_memtime stamps per wave
constexpr int kMaxLatSteps = PDS_MAX_LATENCY_STEPS;  // rows of the latency action ring (envs/agents.py:180-182)

// ---- packing of the per-env counter word --------------------------------------------------------
// bits 0..15 env.step calls since reset | bit 16 quaternion == -Q(rpy) | bits 17..25 Circle ref_offset
// | bits 26..28 action_idx of the latency ring (envs/agents.py:183,273) | bit 29 kCtrOhBit (below).
// Circle keeps the PHASE (env.step calls + ref_offset) mod num_ref_points in bits 17..25, i.e. the index
// of the current reference point (envs/circle.py:130): it advances by one per step with a wrap, so the
// step needs no modulo; ref_offset itself is recovered where it is asked for (circle_ref_offset).
PDS_DEV uint32_t ctr_pack(uint32_t step, uint32_t sign, uint32_t off, uint32_t lat_idx = 0u) {
  return step | (sign << 16) | (off << 17) | (lat_idx << 26);
}
PDS_DEV uint32_t ctr_step(uint32_t c) { return c & 0xFFFFu; }
PDS_DEV uint32_t ctr_sign(uint32_t c) { return (c >> 16) & 1u; }
PDS_DEV uint32_t ctr_off(uint32_t c) { return (c >> 17) & 0x1FFu; }
PDS_DEV uint32_t ctr_lat(uint32_t c) { return (c >> 26) & 0x7u; }
// bit 29 (observation-noise variants without the Kalman hold): "the kept noisy observation o(k) of this env is in
// oh0-2".  Clear = it is REGENERATED from the stored true state and the previous tick's Philox blocks (round 4: the
// Philox-driven single-step kernel neither reads nor writes oh0-2, -80 B per env-step).  Set by whatever produced o(k) from
// draws that Philox cannot replay -- pds_reset / pds_reset_from_samples, pds_step_with_variates, pds_set_state(NOISY_OBS), the
// deferred-drain resets --, by the kernels that keep it in memory (StoredOh: pds_step_k, pds_rollout, pds_step with two or more
// physics sub-steps) and by materialize_oh_kernel, which csrc/pds_api.hip runs in front of everything that would otherwise
// invalidate the regeneration's inputs: a masked reset (the clock of every tile advances), pds_set_tick, any pds_set_state.
#ifndef PDS_REGEN_OBS
#define PDS_REGEN_OBS 1  // A/B: 0 = the kept observation always lives in oh0-2 (rounds 1-3)
#endif
constexpr uint32_t kCtrOhBit = 1u << 29;
PDS_DEV uint32_t ctr_oh(uint32_t c) { return (c >> 29) & 1u; }
PDS_DEV uint32_t circle_ref_offset(uint32_t c, int ref_points) {  // not on the hot path
  const int d = (int)ctr_off(c) - (int)(ctr_step(c) % (uint32_t)ref_points);
  return (uint32_t)(d < 0 ? d + ref_points : d);
}

// ---- addressing of the per-env streams -------------------------------------------------------------
// One wave owns the 64 consecutive envs of a tile, so every per-env access of the step is
//   (wave-uniform 64-bit base: array + first env of the tile, in SGPRs)  +  (32-bit per-lane byte offset)
// which is the SADDR form of the global instructions (global_load_dwordx4 v, v_off, s[base:base+1]): ONE
// offset VGPR per element size is shared by all arrays, and no 64-bit vector address (2 VGPRs + a
// v_lshl_add_u64 each) is formed per stream.  `lc` is the lane, clamped to the last env of the batch for the
// lanes past the end (they recompute that env; their stores are masked).
// Round 4: the form is a per-variant trait (saddr_variant<V>(), csrc/pds_step.h).  The lean variants -- no PT1 / DR /
// noise / PID / latency: the headline, configs 2 and 4 -- had no spilled register with the plain 64-bit per-lane
// addresses of rounds 1-2 and ran 0.5-1 % faster with them on the same box (profiles/r03_ab_final_vs_round2.txt);
// EnvIdxT<false> is that form: `array + (wb + lc)`, one 64-bit vector address per stream.
template <bool SADDR>
struct EnvIdxT {
  long long wb;  // first env of the tile (wave-uniform)
  uint32_t lc;   // lane within the tile, clamped
  PDS_DEV long long global() const { return wb + (long long)lc; }
};
// 64-bit form: the env index is formed ONCE and made opaque, so that every stream's address is one shift-add on it
// (array + (gi << 4)); left to itself the compiler re-associates to (array + (wb << 4)) + (lc << 4) -- two 64-bit adds
// per stream, ~17 more vector instructions in front of the kernel's first loads (the round-2 kernels had the single
// form; same-box A/B of the headline: profiles/r04_ab_lib_r2.txt).
template <>
struct EnvIdxT<false> {
  long long wb;
  uint32_t lc;
  long long gi;
  PDS_DEV EnvIdxT(long long wb_, uint32_t lc_) : wb(wb_), lc(lc_), gi(wb_ + (long long)lc_) { asm("" : "+v"(gi)); }
  PDS_DEV long long global() const { return gi; }
};
typedef EnvIdxT<true> EnvIdx;
template <class T>
PDS_DEV T *lane_ptr(T *uniform_base, uint32_t index) {  // uniform_base + index, with a 32-bit byte offset
  return reinterpret_cast<T *>(reinterpret_cast<char *>(uniform_base) + index * (uint32_t)sizeof(T));
}
template <class T>
PDS_DEV const T *lane_ptr(const T *uniform_base, uint32_t index) {
  return reinterpret_cast<const T *>(reinterpret_cast<const char *>(uniform_base) + index * (uint32_t)sizeof(T));
}
template <class T>
PDS_DEV T *at(T *array, const EnvIdxT<true> x) { return lane_ptr(array + x.wb, x.lc); }
template <class T>
PDS_DEV T *at(T *array, const EnvIdxT<false> x) { return array + x.global(); }
// The instruction selector matches the SADDR form per BASIC BLOCK: it has to see the zero-extension of the
// 32-bit offset next to the access.  An offset that was extended in an earlier block (common subexpression
// with the loads at kernel entry) arrives as a 64-bit register and the access falls back to a 64-bit vector
// address.  `fresh()` makes the lane index opaque (no instruction) at the top of a block of stores.
// NOT volatile (an asm with unmodelled side effects is a barrier for the scheduler's memory operations): each
// call site passes its own ID as an unused immediate operand instead, so that two sites are never merged.
template <int ID>
PDS_DEV uint32_t fresh(uint32_t lane_index) {
  asm("" : "+v"(lane_index) : "n"(ID));
  return lane_index;
}
template <int ID>
PDS_DEV EnvIdxT<true> fresh(EnvIdxT<true> x) {
  x.lc = fresh<ID>(x.lc);
  return x;
}
template <int ID>
PDS_DEV EnvIdxT<false> fresh(EnvIdxT<false> x) { return x; }  // 64-bit vector addresses: nothing to re-derive
// lane-indexed access into a wave-uniform region (observation tile flush, final_obs rows)
template <bool SADDR, int ID, class T>
PDS_DEV T *lane_at(T *uniform_base, uint32_t lane_index) {
  if constexpr (SADDR) return lane_ptr(uniform_base, fresh<ID>(lane_index));
  else return uniform_base + lane_index;
}

// ---- per-wave clock word (device memory) ----------------------------------------------------------
// x, y: tick (counts pds_reset* / pds_step calls; words 1 and 2 of the Philox counter), z: parity of
// the action ring.  One word per 64-env tile, read by the wave that owns the tile when it starts
// and advanced by its lane 0 when it ends: no kernel argument changes from one step to the next, so
// a stream of pds_step calls can be captured in a hipGraph and replayed (a wave reads its word
// before it writes it; no other wave touches it; the next launch on the stream sees the update).
typedef uint4 WaveClock;

// ---- SoA state in HBM (one float4 "quad" per env and array => 16 B/lane coalesced streams) ------
struct DevState {
  float4 *s0;       // px py pz vx
  float4 *s1;       // vy vz roll pitch
  float4 *s2;       // yaw wx wy wz
  float4 *hist[2];  // action ring: slot `parity` = u(k-1), slot `parity^1` = u(k-2)
  uint32_t *ctr;
  float4 *mx;       // motor state x[4]                 (use_motor_dynamics)
  float4 *par0;     // dt m Jxx Jyy                     (domain_randomization)
  float2 *par1;     // Jzz ftf1                         (domain_randomization)
  float4 *mA;       // per-motor A (B = 1-A)            (domain_randomization & motor)
  float4 *mK;       // per-motor K                      (domain_randomization & motor)
  float4 *ou;       // OU thrust-noise state            (motor_thrust_noise > 0)
  float4 *nz0;      // gyro_bias xyz, gyro_lpf x        (observation_noise > 0)
  float2 *nz1;      // gyro_lpf y z                     (observation_noise > 0)
  float4 *oh0;      // noisy o(k): x y z qx             (observation_noise > 0; a noisy observation
  float4 *oh1;      //             qy qz qw vx           cannot be rebuilt from the true state, so the
  float2 *oh2;      //             vy vz                 history half is kept)
  float4 *pid0;     // rate-PID integral xyz, last_error x  (control_mode != PWM)
  float2 *pid1;     // rate-PID last_error y z
  float4 *pid2;     // attitude-PID integral xyz, last_error x (control_mode == Attitude)
  float2 *pid3;     // attitude-PID last_error y z
  float4 *lat;      // [lat_steps][N] delayed-action ring (use_latency; envs/agents.py:267-273)
  WaveClock *clk;   // [ceil(N / 64)] tick + parity of each tile
  const float2 *circle_ref;  // [300] (x, y) of the reference circle, z = 1
};

struct Consts {
  // model (envs/assets/cf21x_sys_eq.urdf:10,16-17; envs/agents.py:138-156)
  float K, G, m, Jx, Jy, Jz, ftf1, Lq, dt, A, hover_x, hover_action;
  float gec, prop_r, h_clip, t2w, mtc, M_nom, Jx_nom, Jy_nom, Jz_nom, ftf1_nom, dt_nom;
  // task
  float pa, pang, pspin, pterm, pvel, arp;
  float target[3];
  float init_xyz[3], init_rpy[3], init_vel[3], init_w[3];
  float dr;
  // noise (envs/utils.py:85-108 OUNoise; envs/sensors.py:14-27 SensorNoise defaults; rotorS gyro model
  // constants of sensors.py:124-128 evaluated on the host in double for dt = 1/sim_freq)
  float ou_sigma, pos_std, pos_unif, vel_std, q_std, q_unif, gyro_pi, gyro_sb, gyro_rw, gyro_to;
  int agg, max_steps, reset_dist;
  int lat_steps;   // buf_size of the latency ring (0: use_latency False)
  int lat_own1, lat_own2;  // step whose action action_buffer[-1] holds after env.step 1 / 2 (0: still the reset row)
  int ref_points;  // Circle: circle_time * observation_frequency (envs/circle.py:49), <= kRefPoints
  float ref_dth_hi, ref_dth_lo;  // 2 pi / ref_points split in two floats (the angle of reference point t is t x that)
  int obs_rate;    // sim_freq // observation_frequency (envs/base.py:108); > 1: Kalman-hold branch of compute_observation
};

struct StepArgs {
  DevState st;
  Consts k;
  const float4 *actions;
  float *obs;
  float *reward;
  uint8_t *term;
  uint8_t *trunc;
  float *cost;
  float *final_obs;
  const uint8_t *mask;   // reset kernel only
  const float *samples;  // reset kernel only (injected draws) or nullptr
  const float *noise;    // step kernel: injected standard variates [N, PDS_NOISE_FLOATS] or nullptr
  long long n;
  unsigned long long env_id_base;
  uint32_t seed_lo, seed_hi;
  int auto_reset;
  int k_steps;                // step_k_kernel: number of env.step()s per launch
  unsigned long long *stamps;  // diagnostic builds (-DPDS_STAMPS): s_memtime stamps per wave
};

// A second view of the kernel arguments, for the LATE phases of a kernel.  The kernarg segment is read with
// scalar loads that the compiler places where a value is first used -- and then keeps: a pointer that is used by
// the loads at kernel entry and by the stores at the end occupies two SGPRs for the whole kernel, and with ~25
// arrays + ~60 constants the 102 SGPRs overflow into v_writelane / v_readlane spills (up to 114 per kernel in
// round 2).  Reading the late phases' arguments through an OPAQUE copy of the kernarg pointer makes those
// values new loads (scalar-cache hits: the segment was prefetched at entry) instead of live ranges.
// (Kernels only: the calling kernel's first parameter must be the StepArgs struct.)
#ifndef PDS_RELOAD_ARGS
#define PDS_RELOAD_ARGS 1
#endif
// ID: one per call site (the asm is not volatile, see fresh(); equal calls would be merged); `tag`: a loop
// counter when the view has to be renewed per iteration (otherwise the asm is loop-invariant and hoisted).
// ENABLE = false: the lean variants (no PT1 / DR / noise / PID / latency) fit the SGPR file as they are, and on
// a latency-bound launch (65 536 envs: one wave per SIMD) every extra scalar-load round trip shows: 7.2 vs 7.0 us.
// `passed`: the arguments as the caller holds them -- returned as they are when the re-read is off, so a caller that
// hands step_once a StepArgs that is NOT the head of the kernarg segment is only wrong for ENABLE = true, which is
// why every kernel that reaches step_once / drain_reset_queue has the StepArgs as its FIRST kernel argument (RolloutArgs.s:
// static_assert below).
template <int ID, bool ENABLE = true>
PDS_DEV const StepArgs &reload_args(const StepArgs &passed, int tag = 0) {
  if (!ENABLE || !PDS_RELOAD_ARGS) return passed;
  auto kp = __builtin_amdgcn_kernarg_segment_ptr();
  asm("" : "+s"(kp) : "n"(ID), "s"(tag));
  return *(const StepArgs *)kp;  // (C cast: address space 4 -> generic; the loads are inferred back to scalar loads)
}

// tick + seed of the current call: words of the Philox counter / key
struct RngKey {
  uint32_t seed_lo, seed_hi, tick_lo, tick_hi;
};

// Compile-time variant of the fused step: task and feature flags.
template <int TASK_, bool MOTOR_, bool DR_, bool GE_, bool TN_, bool ON_, int CTRL_ = 0, bool LAT_ = false, bool HOLD_ = false>
struct Variant {
  // obs_rate = sim_freq // observation_frequency > 1: Kalman-hold branch of compute_observation (envs/hover.py:150-156).
  // A template flag, not a run-time branch: its registers (held state, full first-call noise) pushed the
  // observation-noise variants to the 168-VGPR cap with spills (Hover 2^20 noise + DR: 95 vs 90.7 us).
  static constexpr bool HOLD = HOLD_;
  static constexpr int CTRL = CTRL_;     // 0 PWM, 1 AttitudeRate PID, 2 cascaded Attitude PID (envs/control.py)
  static constexpr bool LAT = LAT_;      // delayed actions through the latency ring (envs/agents.py:267-276)
  static constexpr int TASK = TASK_;
  static constexpr bool MOTOR = MOTOR_;  // first-order motor model (envs/agents.py:284-288)
  static constexpr bool DR = DR_;        // per-env dt, m, J, ftf1 (, A, K)
  static constexpr bool GE = GE_;        // ground effect (envs/physics.py:27-58, opt-in)
  static constexpr bool TN = TN_;        // OU thrust noise (envs/utils.py:104-108)
  static constexpr bool ON = ON_;        // SensorNoise + gyro low-pass (envs/sensors.py:75-134)
  // |o|: envs/hover.py:131-163, envs/circle.py:128-177, envs/takeoff.py:107-149
  static constexpr int O = ON_ ? (TASK_ == PDS_TASK_HOVER ? 13 : (TASK_ == PDS_TASK_CIRCLE ? 16 : 20))
                               : (TASK_ == PDS_TASK_HOVER ? 17 : (TASK_ == PDS_TASK_CIRCLE ? 16 : 20));
  static constexpr int D = 2 * (O + 4);
  static constexpr bool OH_STORED = false;  // see StoredOh
  static constexpr bool SPLIT_RESET = false;  // see SplitReset
};

// The same variant for a kernel that finds the kept noisy observation of EVERY env in oh0-2 (flagged kCtrOhBit by
// materialize_oh_kernel, csrc/pds_api.hip) and leaves it there: the K-step kernel, which reads and writes the state once per
// K steps -- regenerating in its prologue cost it 40-50 VGPRs over the whole loop (csrc/pds_step.h, step_k_kernel).
template <class V>
struct StoredOh : V {
  static constexpr bool OH_STORED = true;
};

// The same variant for a single-step kernel that does NOT reset the envs that finish (round 6): it stores their terminal
// state, leaves their last observation in the obs row and sets the flags; post_reset_kernel (csrc/pds_reset.h), launched behind it
// on the same stream, compacts the finished envs of 1024-env blocks and resets them DENSELY, one env per lane.  Why: an env that
// finishes costs the in-place reset ~1750 vector instructions of its whole 64-lane wave (cooperative Philox fill + the evaluation
// by 1-3 owner lanes) -- at 2 % finished envs per step that is 3 of 4 waves, a third of the step kernel's vector work spent at
// 2-5 % lane utilisation; densely the same resets cost 1/25 of it.
template <class V>
struct SplitReset : V {
  static constexpr bool SPLIT_RESET = true;
};

struct EnvRegs {
  float px, py, pz, vx, vy, vz, roll, pitch, yaw, wx, wy, wz;
};

struct Params {  // per-env physical parameters (constants unless domain randomisation is on)
  float dt, m, Jx, Jy, Jz, ftf1;
  float A[4], K[4];
};

// standard variates of one SensorNoise.add_noise call that reach the observation
// (envs/sensors.py:75-134); layout == PDS_N_OBS_* / the B part of a noise row
struct ObsNoise {
  float pos_z[3], pos_u[3], vel_z[3], bias_z[3], rw_z[3], to_z[3], th_z[3], th_u[3];
};

// what a noisy observation keeps besides the filtered gyro (which lives in the low-pass state)
struct NoisyObs {
  float x, y, z, qx, qy, qz, qw, vx, vy, vz;
};

struct PidState {  // envs/control.py:133-134, 227-228
  float rate_int[3], rate_err[3], att_int[3], att_err[3];
};

struct NoiseState {
  float ou[4];
  float bias[3];
  float lpf[3];
};

struct Sample {  // one reset() worth of draws, reference order (see include/pds.h PDS_S_*)
  float pos[3], rpy[3], vel[3], w[3], mx[4], act[4];
  float dt, m, J[3], ftf1, T[4], t2w[4];
  int ref_offset;
  float abuf[kMaxLatSteps - 1][4];  // LAT: rows 0..B-2 of action_buffer (row B-1 is `act`), hover.py:226-229
};

PDS_DEV void default_params(const Consts &k, Params &p) {
  p.dt = k.dt; p.m = k.m; p.Jx = k.Jx; p.Jy = k.Jy; p.Jz = k.Jz; p.ftf1 = k.ftf1;
#pragma unroll
  for (int i = 0; i < 4; ++i) { p.A[i] = k.A; p.K[i] = k.K; }
}

// Reference trajectories: envs/circle.py:45-56, envs/takeoff.py:43-47 (z = k/300).  The circle
// (0.25 (1 - cos th_t), 0.25 sin th_t, 1), th_t = 2 pi t / P, is evaluated, not looked up (round 3): the table of rounds
// 1-2 cost every block a global-memory round trip and a barrier before its first instruction of the step, and
// 2.4 KB of LDS; ~27 vector instructions per point with the step's own sincos (abs error 1e-7 x 0.25).  The angle
// t x (2 pi / P) is formed from a two-float 2 pi / P with the rounding error of the leading product recovered by an
// fma (p = t hi; err = fma(t, hi, -p) exactly; th = p + (t lo + err)): th is the correctly rounded float of the exact
// angle up to 1 ulp (2.4e-7 rad near 2 pi), i.e. a target error of <= 0.25 x (2.4e-7 + 1e-7) = 9e-8 m against the
// reference's float64 table (tests/test_gpu_properties.py compares all t < ref_points).  `ref_lds` is unused.
template <int TASK>
PDS_DEV void target_at(const Consts &k, const float2 *ref_lds, int t, float &tx, float &ty, float &tz) {
  if (TASK == PDS_TASK_CIRCLE) {
    const float tf = (float)t;
    const float p = __fmul_rn(tf, k.ref_dth_hi);          // (pinned: not contracted with the fma below)
    const float err = fmaf(tf, k.ref_dth_hi, -p);         // exact rounding error of p
    const float th = p + fmaf(tf, k.ref_dth_lo, err);
    float sn, cs;
    fast_sincos(th, sn, cs);
    tx = 0.25f * (1.0f - cs); ty = 0.25f * sn; tz = 1.0f;
  } else if (TASK == PDS_TASK_TAKEOFF) {
    tx = 0.f; ty = 0.f; tz = (float)t / 300.0f;
  } else {
    tx = k.target[0]; ty = k.target[1]; tz = k.target[2];
  }
}

template <int TASK>
PDS_DEV int target_index(int step, int agg, int phase) {
  if (TASK == PDS_TASK_CIRCLE) return phase;  // (iteration // aggregate_phy_steps + ref_offset) % num_ref_points, envs/circle.py:130
  if (TASK == PDS_TASK_TAKEOFF) return min(step * agg, kRefPoints - 1);  // envs/takeoff.py:108
  return 0;
}

// ---- host-side launch interface of the per-task translation units -------------------------------
struct LaunchFlags {
  bool motor, dr, ge, tn, on;
  int ctrl;
  bool lat;
  bool hold;
  bool half_tile;  // per launch: use the 32-row observation tile (variants without observation noise)
};
// kLaunchStepStored: the single-step kernel of an observation-noise variant in its StoredOh form (the kept noisy observation read
// from and written to oh0-2 instead of regenerated) for pds_step with aggregate_phy_steps >= PDS_STORED_OH_FROM_AGG.  Built and
// measured in round 5, NOT adopted (the macro is 0: the kernels are not instantiated): the regenerating form is the faster one at
// every sub-step count -- same box, Hover default 2^20: 2 sub-steps 113.5 vs 120.9 us, 4 sub-steps 197.5 vs 205.4
// (profiles/r05_ab_stored_vs_regen.txt): it runs four blocks per CU (csrc/pds_step.h four_block_variant), the stored form three.
#ifndef PDS_STORED_OH_FROM_AGG
#define PDS_STORED_OH_FROM_AGG 0  // A/B builds: 2 = instantiate the stored single-step kernels and use them from 2 sub-steps on
#endif
// Which single-step kernels reset finished envs in registers BEHIND their stores (RM_INLINE, csrc/pds_step.h) -- the ones that
// have a SplitReset form: observation noise or the latency ring (no merged form), not TakeOff (its envs only finish by the
// 500-step truncation: deferred drain), not Circle with the latency ring or a PID mode (measured 3-7 % slower than its drain).
#ifndef PDS_INLINE_SINGLE_STEP
#define PDS_INLINE_SINGLE_STEP 1  // A/B: 0 = deferred drain
#endif
constexpr bool inline_single_step_rule(int task, bool on, bool lat, int ctrl) {
  return PDS_INLINE_SINGLE_STEP && (on || lat) && task != PDS_TASK_TAKEOFF && !(task == PDS_TASK_CIRCLE && (lat || ctrl != 0));
}
inline bool split_reset_supported(int task, const LaunchFlags &f) { return inline_single_step_rule(task, f.on, f.lat, f.ctrl); }
// kLaunchStepSplit / kLaunchPostReset: the single-step kernel without its in-place reset (SplitReset<V>) and the dense reset
// launched behind it.
enum LaunchKind { kLaunchStep = 0, kLaunchStepK = 1, kLaunchReset = 2, kLaunchStepStored = 3, kLaunchStepSplit = 4, kLaunchPostReset = 5 };
#ifndef PDS_POST_RESET_ENVS
#define PDS_POST_RESET_ENVS 1024  // envs per wave of post_reset_kernel (A/B: 512, 256)
#endif
constexpr int kPostResetEnvsPerBlock = PDS_POST_RESET_ENVS;
// one translation unit per (task, family) keeps the build parallel: pds_task_*.hip
// Arguments of the fused rollout (csrc/pds_rollout.h).
struct RolloutArgs {
  StepArgs s;  // FIRST member (reload_args reads the kernarg segment as a StepArgs); reward / term / trunc / cost point
               // at the [T, N] rollout buffers, obs at obs_buf + N D (step t writes o(t + 1) into row t + 1)
  pds_mlp pi, vf;
  const float *mean, *stdv;  // optional standardisation of the network inputs (both or none)
  float eps;
  const float *log_std;      // [d_out]
  unsigned long long seed;   // Philox key of the action noise (pds_gaussian_sample)
  const unsigned long long *call_base;  // device word added to call_offset (hipGraph replays), or nullptr
  unsigned long long call_offset;       // step t samples with call = *call_base + call_offset + t + 1
  int deterministic, T;
  const float *obs0;         // [N, D] o(0) (= obs_buf row 0)
  float *act_buf, *logp_buf, *val_buf, *fval_buf, *last_val;
  float *ep_ret, *ep_len, *stats;
};

static_assert(offsetof(RolloutArgs, s) == 0, "reload_args() reads the head of the kernarg segment as a StepArgs");

// The env configurations the fused rollout has a kernel for (csrc/pds_rollout.h launch_rollout_task / _family decide by the same
// rule; pds_rollout asks BEFORE it touches the handle).
inline bool rollout_supported(int task, const LaunchFlags &f) {
  if (f.ge) return task == PDS_TASK_TAKEOFF && f.ctrl == 0 && !f.lat && !f.hold && !f.motor;
  const bool lean = !f.dr && !f.tn && !f.on, full = f.dr && f.tn && f.on;
  if (f.hold) return f.on && f.dr == f.tn && f.ctrl == 0 && !f.lat && !(task == PDS_TASK_TAKEOFF && f.motor);
  if (f.lat) return (lean || full) && (f.ctrl == 0 || task != PDS_TASK_TAKEOFF);
  if (f.ctrl != 0) return (lean || full) && task != PDS_TASK_TAKEOFF;
  return !(task == PDS_TASK_TAKEOFF && f.motor);  // control_mode PWM: every noise setting
}
// ---- the one-launch rollout for observation histories other than 2 (csrc/pds_rollout_hist.h) ----
struct RolloutHistArgs {
  StepArgs s;  // FIRST member (reload_args); reward / term / trunc / cost point at the [T, N] rollout buffers; obs unused
  pds_mlp pi;
  const float *mean, *stdv;
  float eps;
  const float *log_std;
  unsigned long long seed;
  const unsigned long long *call_base;
  unsigned long long call_offset;
  int deterministic, T, H, half, slots;
  float *obs_buf;            // [T + 1, N, H half]
  float *act_buf, *logp_buf;
  float *fin_rows;           // [slots, N, H half]
  int *fin_step;             // [slots, N]: step whose fval entry the row's value is, -1 = unused (set by the caller)
  float *ep_ret, *ep_len, *stats;
};
static_assert(offsetof(RolloutHistArgs, s) == 0, "reload_args() reads the head of the kernarg segment as a StepArgs");

// The env configurations the history rollout is built for: control_mode PWM, no latency ring, no Kalman hold, no ground effect;
// noise {none, reference default (DR + thrust noise + observation noise)} x {with, without motor dynamics; TakeOff: without}.
inline bool rollout_hist_supported(int task, const LaunchFlags &f) {
  if (f.ge || f.hold || f.lat || f.ctrl != 0) return false;
  const bool lean = !f.dr && !f.tn && !f.on, full = f.dr && f.tn && f.on;
  if (!lean && !full) return false;
  return !(task == PDS_TASK_TAKEOFF && f.motor);
}
// input tiles the kernels are instantiated for (d_in <= 64 / 96 / 128 / 192)
inline int rollout_hist_tiles(int d_in) { return d_in <= 64 ? 4 : (d_in <= 96 ? 6 : (d_in <= 128 ? 8 : 12)); }

bool launch_rollout_hist_hover(const LaunchFlags &f, int hn, dim3 grid, hipStream_t s, const RolloutHistArgs &ra);
bool launch_rollout_hist_circle(const LaunchFlags &f, int hn, dim3 grid, hipStream_t s, const RolloutHistArgs &ra);
bool launch_rollout_hist_takeoff(const LaunchFlags &f, int hn, dim3 grid, hipStream_t s, const RolloutHistArgs &ra);
bool launch_rollout_hover(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra);
bool launch_rollout_circle(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra);
bool launch_rollout_takeoff(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra);
// (their families, one translation unit each: control_mode PWM with every noise setting / the latency ring / PID modes + Kalman hold)
bool launch_rollout_hover_pwm(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra);
bool launch_rollout_hover_lat(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra);
bool launch_rollout_circle_pwm(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra);
bool launch_rollout_circle_lat(const LaunchFlags &f, dim3 grid, hipStream_t s, const RolloutArgs &ra);
void launch_hover(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a);
void launch_circle(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a);
void launch_takeoff(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a);
void launch_hover_pid(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a);
void launch_circle_pid(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a);
void launch_hover_pid_ge(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a);
void launch_circle_pid_ge(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a);
void launch_hover_lat(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a);
void launch_circle_lat(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a);
void launch_takeoff_lat(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a);
void launch_hover_hold(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a);
void launch_circle_hold(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a);
void launch_takeoff_hold(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a);

}  // namespace pds
