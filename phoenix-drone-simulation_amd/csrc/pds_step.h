// pds_step.h -- the fused lockstep CrazyFlie SimplePhysics step for MI355X (gfx950, wave64).
//
// One thread per environment.  Per step a thread streams its SoA state quads (16 B/lane, fully
// coalesced), advances PWM->thrust (optionally through the latency ring, the PID cascade, the PT1
// motor model), Newton-Euler force/torque, semi-implicit Euler and Euler->quaternion in registers,
// evaluates the task's reward / cost / termination, resets finished envs from a counter-based Philox
// stream (in registers before its stores, or in a drain after them), and stages its observation row in
// a per-wave LDS tile so that the row-major [N, D] observation tensor is written with contiguous
// 1 KiB wave stores instead of 64 strided rows.  HBM-bound by design: no MFMA (there is no dense
// contraction on this path).  step_once = one env.step() in registers; step_kernel = one step per
// launch, step_k_kernel = K open-loop steps per launch.  Tick and action-ring parity live in device
// memory (one clock word per tile), so launches are hipGraph-capturable.
//
// Reference (paths relative to phoenix_drone_simulation/): envs/physics.py:130-200,
// envs/agents.py:259-298, envs/control.py:94-100, envs/base.py:303-319,433-475, envs/hover.py,
// envs/circle.py, envs/takeoff.py, envs/sensors.py:75-134, envs/utils.py:59-108.
#pragma once
#include <type_traits>
#include "pds_reset.h"

namespace pds {

// Row stride of the per-wave LDS observation tile.  The lanes write their rows side by side, so the
// lane-to-lane stride decides the bank pattern of those writes (ds_write*: bank = (addr / 4) mod 32):
// D = 42 / 34 (Hover) is conflict-free for the 8-byte writes the compiler emits, but D = 40 (Circle) and
// D = 48 (TakeOff) put 8 / 16 lanes of every 32 on the same bank (SQ_LDS_BANK_CONFLICT 230-400 and
// 860-1100 cycles per wave, profiles/r02_pmc_sq_configs_round1_kernels.txt).  Those rows (D % 8 == 0: both
// halves 16-byte aligned) are therefore written as float4 (ds_write_b128: groups of 8 lanes) with the stride
// padded to D + 4 floats, an odd multiple of 16 B: 8 consecutive lanes then cover all 32 banks exactly once.
template <int D>
constexpr int tile_stride() {
  return (D % 8 == 0) ? D + 4 : D;
}

// Row stride of the half tile's parking area (first row halves of lanes 32-63): float4 rows want an odd multiple of
// 16 B (20 floats: yes; 24: padded to 28), scalar rows an odd number of floats (21: yes).
template <int HW>
constexpr int park_stride() {
  return (HW % 4 == 0) ? ((HW / 4) % 2 == 1 ? HW : HW + 4) : (HW % 2 == 1 ? HW : HW + 1);
}

template <int N>
PDS_DEV void lds_store_row(float *dst, const float *src) {  // dst 16-byte aligned
  static_assert(N % 4 == 0, "float4 rows");
#pragma unroll
  for (int j = 0; j < N / 4; ++j)
    reinterpret_cast<float4 *>(dst)[j] = make_float4(src[4 * j], src[4 * j + 1], src[4 * j + 2], src[4 * j + 3]);
}

// Variants whose kernel arguments do not fit the SGPR file for the whole step: they re-read them (reload_args).
template <class V>
constexpr bool heavy_variant() { return V::MOTOR || V::DR || V::TN || V::ON || V::CTRL != 0 || V::LAT; }

// Addressing form and half-tile staging are per-variant traits (round 4).  Round 3 moved every variant to the SADDR
// form + a half tile that parks the first row half in LDS; both exist to get the heavy variants under their
// register caps (SGPR spills, the merged reset of PT1 + DR on the half tile).  The lean variants had no spill to
// remove and paid for it: same box, round-2 vs round-3 library, headline 55.6 -> 56.1 us, Hover 65 536 (half tile)
// 6.98 -> 7.2 us, TakeOff + ground effect 57.8 -> 58.5 us (profiles/r03_ab_final_vs_round2.txt).  They keep the
// round-2 forms: 64-bit per-lane addresses, and on the half tile the whole row in registers until its pass.
#ifndef PDS_SADDR_LEAN
#define PDS_SADDR_LEAN 0  // A/B: 1 = round-3 form for every variant
#endif
template <class V>
constexpr bool saddr_variant() { return PDS_SADDR_LEAN || heavy_variant<V>(); }
template <class V>
constexpr bool park_variant() { return PDS_SADDR_LEAN || heavy_variant<V>(); }
template <class V>
using Idx = EnvIdxT<saddr_variant<V>()>;

// Observation-noise variants without the Kalman hold do not keep the noisy o(k) in memory: it is regenerated
// (regen_kept_obs, csrc/pds_reset.h) unless the env's counter word says it was stored (kCtrOhBit; PDS_REGEN_OBS in pds_types.h).
template <class V>
constexpr bool regen_obs_variant() { return PDS_REGEN_OBS && V::ON && !V::HOLD && !V::OH_STORED; }
// ... and a kernel that leaves every env's kept observation in oh0-2 says so in the counter word (StoredOh)
template <class V>
constexpr bool flags_oh_variant() { return PDS_REGEN_OBS && V::ON && !V::HOLD && V::OH_STORED; }

// Wave-cooperative copy of this wave's [rows, D] LDS tile to global memory (contiguous region).
template <int D, int TR, bool SADDR>
PDS_DEV void flush_tile(const float *tile, float *gdst, int rows, int lane) {
  constexpr int TS = tile_stride<D>();
  // (an [N, D] slice of a [K, N, D] tensor is only 8-byte aligned when N D is not a multiple of 4)
  const bool fast = rows == TR && (reinterpret_cast<uintptr_t>(gdst) & 15u) == 0;  // wave-uniform
  if constexpr (TS == D) {
    if (fast) {
      constexpr int NV = TR * D / 4;  // float4 count (D is even, 32*D divisible by 4)
      const float4 *src = reinterpret_cast<const float4 *>(tile);
      float4 *dst = reinterpret_cast<float4 *>(gdst);
      if constexpr (SADDR) {
        const uint32_t ln = fresh<1>((uint32_t)lane);
#pragma unroll
        for (int it = 0; it < (NV + kWave - 1) / kWave; ++it) {
          const int idx = it * kWave + lane;
          if (idx < NV) nt_store4(lane_ptr(dst + it * kWave, ln), src[idx]);  // (uniform base + constant, lane offset)
        }
      } else {
#pragma unroll
        for (int it = 0; it < (NV + kWave - 1) / kWave; ++it) {
          const int idx = it * kWave + lane;
          if (idx < NV) nt_store4(dst + idx, src[idx]);
        }
      }
      return;
    }
  } else {
    if (fast) {
      // padded rows: Q float4 per row, RP whole rows per pass (6 x 10 or 5 x 12 lanes); the region a pass
      // stores is still contiguous in the [N, D] tensor (rows follow each other without padding there)
      constexpr int Q = D / 4, RP = kWave / Q, NP = (TR + RP - 1) / RP;
      const int r0 = lane / Q, c = lane - r0 * Q;
      const float4 *src = reinterpret_cast<const float4 *>(tile + r0 * TS + 4 * c);
      float4 *dst = lane_at<SADDR, 2>(reinterpret_cast<float4 *>(gdst), (uint32_t)(r0 * Q + c));
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (r0 < RP && p * RP + r0 < TR) nt_store4(dst + p * RP * Q, src[p * RP * (TS / 4)]);  // (+ compile-time constant: the instruction offset)
      }
      return;
    }
  }
  const int n = rows * D;
  for (int idx = lane; idx < n; idx += kWave) {
    const int r = idx / D;
    gdst[idx] = tile[r * TS + (idx - r * D)];
  }
}

// Order of the action load among the step's loads (same-box A/B, profiles/r02_load_order.txt):
//  1  action LAST, no wait: loads return in order and the action slice is the one stream that is cold
//     every step (fresh 16 B x N region), so nothing queues behind it; best for the lean variants
//     (Hover 65 536: 7.2 vs 7.4 us, Hover 2^20: 57.9 vs 58.3 us);
//  2  action FIRST and waited for before the other loads are issued: paces the 12-20 streams of the
//     heavy variants when the whole grid starts at once (Circle 262 144 + PT1 + DR: 20.7 vs 22.8 us;
//     Hover 2^20 with noise + DR: 89.4 vs 90.0 us).
// 0 = action first without the wait (A/B only).  Default: by variant.
#ifndef PDS_ACT_LOAD_ORDER
#define PDS_ACT_LOAD_ORDER ((V::MOTOR || V::DR || V::TN || V::ON || V::CTRL != 0 || V::LAT) ? 2 : 1)
#endif
// Inputs of one env-step, loaded 16 B/lane.
struct Loaded {
  float4 act, q0, q1, q2, hA, hB, mx, p0, mA, mK, ou, nz0, oh0, oh1, pid0, pid2;
  float2 p1, nz1, oh2, pid1, pid3;
  uint32_t ctr;
  WaveClock clk;
};

// Both slots of the action ring are requested by index (hA = slot 0, hB = slot 1): which of them is
// u(k-1) is decided by the parity bit of the wave's clock word, which arrives with the same batch of
// loads -- no load address depends on another load.
template <class V>
PDS_DEV void load_env(const StepArgs &a, const Idx<V> ix, long long tile, Loaded &L) {
  constexpr int kOrder = PDS_ACT_LOAD_ORDER;
  if (kOrder == 0 || kOrder == 2) L.act = nt_load4(at(a.actions, ix));  // read once per step: keep it out of the caches
  if (kOrder == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  L.q0 = st_load4(at(a.st.s0, ix));
  L.q1 = st_load4(at(a.st.s1, ix));
  L.q2 = st_load4(at(a.st.s2, ix));
  L.hA = st_load4(at(a.st.hist[0], ix));
  L.hB = st_load4(at(a.st.hist[1], ix));
  L.ctr = *at(a.st.ctr, ix);
  L.clk = a.st.clk[tile];
  if (V::MOTOR) L.mx = *at(a.st.mx, ix);
  if (V::DR) {
    L.p0 = *at(a.st.par0, ix);
    L.p1 = *at(a.st.par1, ix);
    if (V::MOTOR) { L.mA = *at(a.st.mA, ix); L.mK = *at(a.st.mK, ix); }
  }
  if (V::TN) L.ou = *at(a.st.ou, ix);
  if (V::CTRL >= 1) { L.pid0 = *at(a.st.pid0, ix); L.pid1 = *at(a.st.pid1, ix); }
  if (V::CTRL == 2) { L.pid2 = *at(a.st.pid2, ix); L.pid3 = *at(a.st.pid3, ix); }
  if (V::ON) {
    L.nz0 = *at(a.st.nz0, ix); L.nz1 = *at(a.st.nz1, ix);
    if (!regen_obs_variant<V>()) { L.oh0 = *at(a.st.oh0, ix); L.oh1 = *at(a.st.oh1, ix); L.oh2 = *at(a.st.oh2, ix); }
  }
  if (kOrder == 1) L.act = nt_load4(at(a.actions, ix));
}

// Everything one env carries from step to step, in registers (members a variant does not use are
// never touched and cost nothing).
struct EnvState {
  EnvRegs e;
  float4 h1, h2;  // u(k-1), u(k-2)
  uint32_t ctr;
  float xm[4];    // MOTOR
  Params par;     // DR (otherwise the constants)
  NoiseState ns;  // TN / ON
  PidState ps;    // CTRL
  NoisyObs oh;    // ON: the previous noisy observation
};

template <class V>
PDS_DEV void unpack_state(const Consts &k, const Loaded &cur, int parity, EnvState &S) {
  S.h1 = parity ? cur.hB : cur.hA;
  S.h2 = parity ? cur.hA : cur.hB;
  S.ctr = cur.ctr;
#pragma unroll
  for (int j = 0; j < 4; ++j) S.xm[j] = 0.f;
  if (V::MOTOR) { S.xm[0] = cur.mx.x; S.xm[1] = cur.mx.y; S.xm[2] = cur.mx.z; S.xm[3] = cur.mx.w; }
  default_params(k, S.par);
  if (V::DR) {
    S.par.dt = cur.p0.x; S.par.m = cur.p0.y; S.par.Jx = cur.p0.z; S.par.Jy = cur.p0.w; S.par.Jz = cur.p1.x; S.par.ftf1 = cur.p1.y;
    if (V::MOTOR) {
      S.par.A[0] = cur.mA.x; S.par.A[1] = cur.mA.y; S.par.A[2] = cur.mA.z; S.par.A[3] = cur.mA.w;
      S.par.K[0] = cur.mK.x; S.par.K[1] = cur.mK.y; S.par.K[2] = cur.mK.z; S.par.K[3] = cur.mK.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) S.ns.ou[j] = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) { S.ns.bias[j] = 0.f; S.ns.lpf[j] = 0.f; }
  if (V::TN) { S.ns.ou[0] = cur.ou.x; S.ns.ou[1] = cur.ou.y; S.ns.ou[2] = cur.ou.z; S.ns.ou[3] = cur.ou.w; }
  if (V::ON) {
    S.ns.bias[0] = cur.nz0.x; S.ns.bias[1] = cur.nz0.y; S.ns.bias[2] = cur.nz0.z;
    S.ns.lpf[0] = cur.nz0.w; S.ns.lpf[1] = cur.nz1.x; S.ns.lpf[2] = cur.nz1.y;
    if (!regen_obs_variant<V>())
      S.oh = NoisyObs{cur.oh0.x, cur.oh0.y, cur.oh0.z, cur.oh0.w, cur.oh1.x, cur.oh1.y, cur.oh1.z,
                      cur.oh1.w, cur.oh2.x, cur.oh2.y};
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) { S.ps.rate_int[j] = S.ps.rate_err[j] = S.ps.att_int[j] = S.ps.att_err[j] = 0.f; }
  if (V::CTRL >= 1) {
    S.ps.rate_int[0] = cur.pid0.x; S.ps.rate_int[1] = cur.pid0.y; S.ps.rate_int[2] = cur.pid0.z;
    S.ps.rate_err[0] = cur.pid0.w; S.ps.rate_err[1] = cur.pid1.x; S.ps.rate_err[2] = cur.pid1.y;
  }
  if (V::CTRL == 2) {
    S.ps.att_int[0] = cur.pid2.x; S.ps.att_int[1] = cur.pid2.y; S.ps.att_int[2] = cur.pid2.z;
    S.ps.att_err[0] = cur.pid2.w; S.ps.att_err[1] = cur.pid3.x; S.ps.att_err[2] = cur.pid3.y;
  }
  S.e = EnvRegs{cur.q0.x, cur.q0.y, cur.q0.z, cur.q0.w, cur.q1.x, cur.q1.y, cur.q1.z, cur.q1.w,
                cur.q2.x, cur.q2.y, cur.q2.z, cur.q2.w};
}

// S.oh of a freshly loaded env (regen_obs_variant: see there): regenerated, or -- rarely: the first step after an
// explicit reset, injected variates or pds_set_state -- read from oh0-2 by the lanes whose counter word says so.
template <class V>
PDS_DEV void init_kept_obs(const StepArgs &a, const RngKey &rk, const Idx<V> ix, EnvState &S) {
  if constexpr (regen_obs_variant<V>()) {
    const uint32_t env_id = (uint32_t)(a.env_id_base + (unsigned long long)ix.wb) + ix.lc;
    regen_kept_obs(a.k, env_id, rk, ctr_step(S.ctr) == 0u, S.e, S.oh);
    const bool stored = ctr_oh(S.ctr) != 0u;
    if (__ballot(stored) != 0ull) {  // wave-uniform
      if (stored) {
        const float4 o0 = *at(a.st.oh0, ix), o1 = *at(a.st.oh1, ix);
        const float2 o2 = *at(a.st.oh2, ix);
        S.oh = NoisyObs{o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w, o2.x, o2.y};
      }
    }
  }
}

// Coalesced 16 B/lane stores of the env state.  `new_parity` = parity of the action ring AFTER the
// stored step: slot new_parity holds u(k-1) = S.h1.  The single-step kernel leaves the other slot
// alone (it already holds S.h2) unless the env was reset in registers (`both`); the K-step kernel
// rewrites both slots and the randomised parameters.
template <class V>
PDS_DEV void store_state(const StepArgs &a, const Idx<V> i_, int new_parity, const EnvState &S, bool both) {
  const EnvRegs &e = S.e;
  Idx<V> i = fresh<3>(i_);
  st_store4(at(a.st.s0, i), make_float4(e.px, e.py, e.pz, e.vx));
  st_store4(at(a.st.s1, i), make_float4(e.vy, e.vz, e.roll, e.pitch));
  st_store4(at(a.st.s2, i), make_float4(e.yaw, e.wx, e.wy, e.wz));
  st_store4(at(a.st.hist[new_parity], i), S.h1);
  *at(a.st.ctr, i) = flags_oh_variant<V>() ? (S.ctr | kCtrOhBit) : S.ctr;
  if (V::MOTOR) *at(a.st.mx, i) = make_float4(S.xm[0], S.xm[1], S.xm[2], S.xm[3]);
  if (V::TN) *at(a.st.ou, i) = make_float4(S.ns.ou[0], S.ns.ou[1], S.ns.ou[2], S.ns.ou[3]);
  if (V::CTRL >= 1) {
    *at(a.st.pid0, i) = make_float4(S.ps.rate_int[0], S.ps.rate_int[1], S.ps.rate_int[2], S.ps.rate_err[0]);
    *at(a.st.pid1, i) = make_float2(S.ps.rate_err[1], S.ps.rate_err[2]);
  }
  if (V::CTRL == 2) {
    *at(a.st.pid2, i) = make_float4(S.ps.att_int[0], S.ps.att_int[1], S.ps.att_int[2], S.ps.att_err[0]);
    *at(a.st.pid3, i) = make_float2(S.ps.att_err[1], S.ps.att_err[2]);
  }
  if (V::ON) {
    *at(a.st.nz0, i) = make_float4(S.ns.bias[0], S.ns.bias[1], S.ns.bias[2], S.ns.lpf[0]);
    *at(a.st.nz1, i) = make_float2(S.ns.lpf[1], S.ns.lpf[2]);
    if (!regen_obs_variant<V>() || a.noise != nullptr) {  // (injected variates cannot be replayed: S.ctr carries kCtrOhBit)
      *at(a.st.oh0, i) = make_float4(S.oh.x, S.oh.y, S.oh.z, S.oh.qx);
      *at(a.st.oh1, i) = make_float4(S.oh.qy, S.oh.qz, S.oh.qw, S.oh.vx);
      *at(a.st.oh2, i) = make_float2(S.oh.vy, S.oh.vz);
    }
  }
  if (both) {  // written only by resets (no write-after-write with this step's stores)
    i = fresh<4>(i_);
    *at(a.st.hist[new_parity ^ 1], i) = S.h2;
    if (V::DR) {
      *at(a.st.par0, i) = make_float4(S.par.dt, S.par.m, S.par.Jx, S.par.Jy);
      *at(a.st.par1, i) = make_float2(S.par.Jz, S.par.ftf1);
      if (V::MOTOR) {
        *at(a.st.mA, i) = make_float4(S.par.A[0], S.par.A[1], S.par.A[2], S.par.A[3]);
        *at(a.st.mK, i) = make_float4(S.par.K[0], S.par.K[1], S.par.K[2], S.par.K[3]);
      }
    }
  }
}

// envs/control.py:120-191 AttitudeRate.compute_output: PID on the body rates in deg/s with the
// firmware gains (control.py:12-27); dt is the controller's construction-time step 1/sim_freq.
PDS_DEV void rate_pid(float dt, const EnvRegs &e, const float target[3], PidState &ps, float out[3]) {
  const float kp[3] = {250.f, 250.f, 120.f}, ki[3] = {500.f, 500.f, 16.7f}, kd[3] = {2.5f, 2.5f, 0.f};
  const float lim[3] = {33.3f, 33.3f, 166.7f};
  const float w[3] = {e.wx, e.wy, e.wz};
  const float inv_dt = 1.0f / dt;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float error = (target[i] - w[i]) * 180.f / kPi;
    const float derivative = (error - ps.rate_err[i]) * inv_dt;
    ps.rate_err[i] = error;
    ps.rate_int[i] = clampf(ps.rate_int[i] + error * dt, -lim[i], lim[i]);
    out[i] = kp[i] * error + ki[i] * ps.rate_int[i] + kd[i] * derivative;
  }
}

// envs/control.py:194-287 Attitude.compute_output: outer loop on the Euler angles, output in rad/s
PDS_DEV void att_pid(float dt, const EnvRegs &e, const float target[3], PidState &ps, float out[3]) {
  const float kp[3] = {6.f, 6.f, 6.f}, ki[3] = {3.f, 3.f, 1.f}, kd[3] = {0.f, 0.f, 0.35f};
  const float lim[3] = {20.f, 20.f, 360.f};
  const float r[3] = {e.roll, e.pitch, e.yaw};
  const float inv_dt = 1.0f / dt;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float error = (target[i] - r[i]) * 180.f / kPi;
    const float derivative = (error - ps.att_err[i]) * inv_dt;
    ps.att_err[i] = error;
    ps.att_int[i] = clampf(ps.att_int[i] + error * dt, -lim[i], lim[i]);
    out[i] = (kp[i] * error + ki[i] * ps.att_int[i] + kd[i] * derivative) / 180.f * kPi;
  }
}

// action -> PWM for the PID control modes: AttitudeRate.act (control.py:151-160) / Attitude.act
// (control.py:244-259) + rpy_control_factors_to_PWM (control.py:34-50)
template <int CTRL>
PDS_DEV void control_pwm(float dt, const EnvRegs &e, const float av[4], PidState &ps, float pwm[4]) {
  const float c0 = clampf(av[0], -1.f, 1.f), c1 = clampf(av[1], -1.f, 1.f), c2 = clampf(av[2], -1.f, 1.f),
              c3 = clampf(av[3], -1.f, 1.f);
  float factors[3], thrust;
  if (CTRL == 1) {
    thrust = 30000.f + c0 * 30000.f;
    const float tgt[3] = {c1 * kPi / 3.f, c2 * kPi / 3.f, c3 * kPi / 3.f};
    rate_pid(dt, e, tgt, ps, factors);
  } else {
    const float tgt[3] = {c1 * kPi / 18.f, c2 * kPi / 18.f, c3 * kPi / 18.f};
    thrust = 45000.f + c0 * 10000.f;
    float rates[3];
    att_pid(dt, e, tgt, ps, rates);
    rate_pid(dt, e, rates, ps, factors);
  }
  const float r = factors[0] * 0.5f, p = factors[1] * 0.5f, y = factors[2];
  pwm[0] = clampf(thrust - r - p - y, 0.f, 60000.f);
  pwm[1] = clampf(thrust - r + p + y, 0.f, 60000.f);
  pwm[2] = clampf(thrust + r + p - y, 0.f, 60000.f);
  pwm[3] = clampf(thrust + r - p + y, 0.f, 60000.f);
}

// a row half into the LDS tile: scalar stores in place, or (padded float4 rows) built in registers
// and stored as ds_write_b128
template <int TASK, int N, bool VEC>
PDS_DEV void put_obs_half(float *dst, const EnvRegs &e, const Quat &q, const float4 &la, float tx, float ty, float tz,
                          const float4 &ha) {
  if constexpr (VEC) {
    float h[N];
    write_obs_half<TASK>(h, e, q, la, tx, ty, tz, ha);
    lds_store_row<N>(dst, h);
  } else {
    write_obs_half<TASK>(dst, e, q, la, tx, ty, tz, ha);
  }
}
template <int TASK, int N, bool VEC>
PDS_DEV void put_noisy_half(float *dst, const NoisyObs &o, const float lpf[3], const float4 &la, float tx, float ty,
                            float tz, const float4 &ha) {
  if constexpr (VEC) {
    float h[N];
    write_noisy_half<TASK>(h, o, lpf, la, tx, ty, tz, ha);
    lds_store_row<N>(dst, h);
  } else {
    write_noisy_half<TASK>(dst, o, lpf, la, tx, ty, tz, ha);
  }
}

// Standard variates of one physics sub-step: OUNoise.noise (4 z) and the gyro part of the
// add_noise call whose observation is discarded (envs/base.py:464): 9 z.
struct SubNoise {
  float ou[4], bias_z[3], rw_z[3], to_z[3];
  ObsNoise full;  // obs_rate > 1: the whole call (its gyro members alias the three above)
};

template <class V>
PDS_DEV void sub_noise(const StepArgs &a, const RngKey &rk, uint32_t env_id, const Idx<V> ix, int sub, SubNoise &n) {
  if (a.noise != nullptr) {  // injected (parity tests): one PDS_NOISE_FLOATS block per physics sub-step
    const float *p = a.noise + (ix.global() * a.k.agg + sub) * PDS_NOISE_FLOATS;
#pragma unroll
    for (int j = 0; j < 4; ++j) n.ou[j] = p[PDS_N_OU + j];
#pragma unroll
    for (int j = 0; j < 3; ++j) { n.bias_z[j] = p[PDS_N_A_BIAS + j]; n.rw_z[j] = p[PDS_N_A_RW + j]; n.to_z[j] = p[PDS_N_A_TO + j]; }
    if (V::ON && V::HOLD) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        n.full.pos_z[j] = p[PDS_N_A_POS_Z + j]; n.full.pos_u[j] = p[PDS_N_A_POS_U + j]; n.full.vel_z[j] = p[PDS_N_A_VEL_Z + j];
        n.full.th_z[j] = p[PDS_N_A_TH_Z + j]; n.full.th_u[j] = p[PDS_N_A_TH_U + j];
        n.full.bias_z[j] = n.bias_z[j]; n.full.rw_z[j] = n.rw_z[j]; n.full.to_z[j] = n.to_z[j];
      }
    }
    return;
  }
  // words 0,1 -> OU z[0..3]; words 2..6 -> bias, random walk, turn-on z[4..12] (one pair per word)
  const uint32_t b0 = kBlkSubNoise + 2u * (uint32_t)sub;
  float z[14];
  const U4 r = philox4x32_7(env_id, rk.tick_lo, rk.tick_hi, b0, rk.seed_lo, rk.seed_hi);
  box_muller_word(r.x, z[0], z[1]);
  box_muller_word(r.y, z[2], z[3]);
  if (V::ON) {
    const U4 r1 = philox4x32_7(env_id, rk.tick_lo, rk.tick_hi, b0 + 1u, rk.seed_lo, rk.seed_hi);
    box_muller_word(r.z, z[4], z[5]);
    box_muller_word(r.w, z[6], z[7]);
    box_muller_word(r1.x, z[8], z[9]);
    box_muller_word(r1.y, z[10], z[11]);
    box_muller_word(r1.z, z[12], z[13]);
  } else {
#pragma unroll
    for (int j = 4; j < 14; ++j) z[j] = 0.f;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) n.ou[j] = z[j];
#pragma unroll
  for (int j = 0; j < 3; ++j) { n.bias_z[j] = z[4 + j]; n.rw_z[j] = z[7 + j]; n.to_z[j] = z[10 + j]; }
  if (V::ON && V::HOLD) {  // the held state may be refreshed by this call
    const uint32_t bx = kBlkSubNoiseX + 2u * (uint32_t)sub;
    const U4 x0 = philox4x32_7(env_id, rk.tick_lo, rk.tick_hi, bx, rk.seed_lo, rk.seed_hi);
    const U4 x1 = philox4x32_7(env_id, rk.tick_lo, rk.tick_hi, bx + 1u, rk.seed_lo, rk.seed_hi);
    float y[10];
    box_muller_word(x0.x, y[0], y[1]);
    box_muller_word(x0.y, y[2], y[3]);
    box_muller_word(x0.z, y[4], y[5]);
    box_muller_word(x0.w, y[6], y[7]);
    box_muller_word(x1.x, y[8], y[9]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      n.full.pos_z[j] = y[j]; n.full.vel_z[j] = y[3 + j]; n.full.th_z[j] = y[6 + j];
      n.full.bias_z[j] = n.bias_z[j]; n.full.rw_z[j] = n.rw_z[j]; n.full.to_z[j] = n.to_z[j];
    }
    n.full.pos_u[0] = u01_lo16(x1.y); n.full.pos_u[1] = u01_hi16(x1.y);
    n.full.pos_u[2] = u01_lo16(x1.z); n.full.th_u[0] = u01_hi16(x1.z);
    n.full.th_u[1] = u01_lo16(x1.w); n.full.th_u[2] = u01_hi16(x1.w);
  }
}

// How an env that finished is reset inside the step (all three produce the same bits):
//  RM_MERGED   before the wave stores, 8 lanes per finished env, results through ds_bpermute
//              (reset_in_registers): state and observation leave through the ordinary coalesced stores;
//  RM_DEFERRED after the wave's stores (drain_reset_queue, cooperative Philox through LDS, scattered
//              stores): the variants whose merged form would cost a wave of occupancy (single step only);
//  RM_INLINE   by the finished lane itself after the final_obs copy (K-step kernel for the variants
//              that are not merged: the fresh state has to come back into registers).
enum { RM_MERGED = 0, RM_DEFERRED = 1, RM_INLINE = 2 };

#ifdef PDS_STAMPS  // diagnostic build: s_memtime stamps of one wave's phases (profiles/tools/stamps.py)
#define PDS_STAMP(j)                                                  \
  do {                                                                \
    __builtin_amdgcn_sched_barrier(0);                                \
    stamp_[j] = __builtin_amdgcn_s_memtime();                         \
    __builtin_amdgcn_sched_barrier(0);                                \
  } while (0)
#define PDS_STAMP_WAIT(j)                                             \
  do {                                                                \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       \
    PDS_STAMP(j);                                                     \
  } while (0)
#else
#define PDS_STAMP(j) do { } while (0)
#define PDS_STAMP_WAIT(j) do { } while (0)
#endif

// One env.step() of the env in `S` (registers in, registers out) + this step's output streams, which
// start `o1` envs into the output tensors (0 for the single-step kernel, step * N for the K-step one;
// the pointers are formed where they are used so that the kernel arguments are not held in SGPRs
// across the whole step).  Returns true for a lane whose env was reset in registers (RM_MERGED /
// RM_INLINE).
// Inline reset of the K-step kernel: cooperative Philox through an LDS scratch (see step_once), except TakeOff.
template <class V>
constexpr bool inline_coop_variant() { return V::TASK != PDS_TASK_TAKEOFF; }
// envs per pass: the scratch of the noise-free latency variants has to leave room for 3 blocks per CU
template <class V>
constexpr int inline_envs_per_pass() {
  // a wave holds 1.3 finished envs on average when it holds any.  Round 3 used 8 per pass for Hover's 34-float noisy rows
  // because 3 blocks per CU still fitted; round 4 wants FOUR blocks per CU there (four_block_variant), with a scratch of
  // CONVERTED variates (336 B per env): 3 envs per pass -- 4 x (8704 + 3 x 336 + 256) = 39 872 B per block
  return (V::ON && !V::LAT && V::TASK == PDS_TASK_HOVER) ? 3 : kResetsPerPass / 2;
}
// ... in the single-step kernel.  The K-step and rollout kernels (STORE == false) have the LDS for 8: a training rollout under a
// young policy finishes ~5 % of its envs per step, 3.2 per tile, and every pass beyond the first repeats the whole evaluation.
#ifndef PDS_INLINE_ENVS_KSTEP
#define PDS_INLINE_ENVS_KSTEP 8
#endif
template <class V, bool STORE>
constexpr int inline_envs() {  // (Circle's 46 KB of observation tiles per block leave no room for more at three blocks per CU)
  return (STORE || V::TASK != PDS_TASK_HOVER) ? inline_envs_per_pass<V>() : PDS_INLINE_ENVS_KSTEP;
}
// U4 slots per env of the in-register resets' scratch (LdsVariates layout: floats, see fill_reset_variates)
template <class V>
constexpr int scratch_stride() { return (variates_floats<V>() + 3) / 4; }
// Hover with observation noise (D = 34: an 8.7 KB tile per wave): since the kept observation is regenerated instead
// of loaded (regen_obs_variant) these kernels need 107-127 VGPRs, so FOUR blocks per CU fit the register file -- and the
// LDS, with the scratch above: 39 872 B per block.  Not with PT1 + DR (127-139 VGPRs:
// spills under the 128 cap), the PID modes or the latency ring; Circle / TakeOff rows (44 / 52 floats) do not fit 40 KB.
template <class V>
constexpr bool four_block_variant() {
  return regen_obs_variant<V>() && V::TASK == PDS_TASK_HOVER && !(V::MOTOR && V::DR) && V::CTRL == 0 && !V::LAT;
}

// What a caller that goes on with the step's results in registers gets back (csrc/pds_rollout.h).
struct StepOut {
  float reward;
  bool done, trunc;
};

// `fin_lds` (optional): [64][D] LDS image that receives the last observation row of every env that finished (the
// rows that go to final_obs); `so` (optional): reward / flags of this lane's env.
template <class V, int TR, int RM, bool STORE>
PDS_DEV bool step_once(const StepArgs &a, const long long o1, const RngKey &rk, int parity, const float2 *ref_lds,
                       float *tile, float *park, uint32_t *queue, U4 *scratch, int lane, long long wave_base, const Idx<V> ix,
                       bool active, const float4 act, EnvState &S, int &qcount, float *fin_lds, StepOut *so
#ifdef PDS_STAMPS
                       , unsigned long long *stamp_
#endif
) {
  constexpr int TASK = V::TASK;
  constexpr int D = V::D;
  constexpr int O = V::O;
  constexpr int TS = tile_stride<D>();
  constexpr bool VEC = (TS != D) && TR == kWave;  // padded rows written in place as float4 halves
  static_assert(RM != RM_MERGED || (!V::ON && !V::LAT), "merged reset: variants without observation noise / latency");
  static_assert(RM != RM_INLINE || TR == kWave, "inline reset: full tile only");
  static_assert(RM != RM_DEFERRED || STORE, "deferred drain: the state must be in HBM before it");
  const Consts &k = a.k;
  const StepArgs &a_in = a;  // (the late phases shadow `a` with the re-read view)
  // Full tile: the row is built in place in LDS.  Half tile (the wave stages and flushes its 64 rows in two passes
  // of 32): the FIRST half of the row -- o(k), known before the physics -- goes straight to LDS as well, into the
  // tile for lanes 0-31 and into a parking area for lanes 32-63, so that only the second half (o(k+1), written last)
  // waits in registers for its pass: 20-24 VGPRs instead of a whole 40-48 float row held across the step.  That is
  // what lets the PT1 + DR variants reset in registers on the half tile (round 2: 61 spilled VGPRs, deferred drain).
  // The lean variants (park_variant<V>() false) keep the round-2 form of the half tile: the whole row in registers
  // until its pass (it costs them spills but no LDS round trip on the latency-bound 65 536-env launch: 6.98 vs 7.2 us).
  constexpr int HW = O + 4;           // floats per row half (D == 2 HW)
  constexpr int PS = park_stride<HW>();
  constexpr bool PARK = (TR != kWave) && park_variant<V>();
  constexpr bool VEC1 = (TR == kWave) ? VEC : (PARK && TS != D);  // first half as float4 (rows 16-byte aligned in both places)
  float rowbuf[(TR == kWave) ? 1 : (PARK ? HW : D)];
  float *row = (TR == kWave) ? tile + lane * TS : (PARK ? (lane < TR ? tile + lane * TS : park + (lane - TR) * PS) : rowbuf);
  float *row2 = (TR == kWave) ? row + HW : (PARK ? rowbuf : rowbuf + HW);
  const uint32_t env_id = (uint32_t)(a.env_id_base + (unsigned long long)ix.wb) + ix.lc;  // (uniform part in SGPRs)
  const float4 h1 = S.h1, h2 = S.h2;
  const uint32_t ctr = S.ctr;
  EnvRegs &e = S.e;
  Params &par = S.par;
  NoiseState &ns = S.ns;
  PidState &ps = S.ps;
  const int step = (int)ctr_step(ctr);
  const int phase = (int)ctr_off(ctr);  // Circle: index of the current reference point
  const int phase1 = (TASK == PDS_TASK_CIRCLE) ? ((phase + 1 == k.ref_points) ? 0 : phase + 1) : 0;
  // ref_offset survives a reset only without the reset distribution (envs/circle.py:222-226)
  int ref_offset = 0;
  if (TASK == PDS_TASK_CIRCLE && !k.reset_dist) ref_offset = (int)circle_ref_offset(ctr, k.ref_points);

  // paired actions of the two row halves: u(k-2) and u(k-1).  With the latency model the reference's
  // action_history still holds VIEWS of action_buffer[-1] for the first two steps after a reset
  // (agents.py:386, base.py:425-426), so those entries show whatever apply_action has written into
  // that row meanwhile (k.lat_own1/2 = the step whose action the row holds after step 1 / 2).
  float4 pa1 = h2, pa2 = h1;
  if (V::LAT) {
    if (step == 0) {
      if (k.lat_own1 == 1) { pa1 = act; pa2 = act; }
    } else if (step == 1) {
      if (k.lat_own2 == 2) pa1 = act;
      else if (k.lat_own2 == 1) pa1 = h1;
    }
  }

  // ---- o(k): first half of the row ------------------------------------------------------------
  // noise-free: rebuilt from the pre-step state instead of being re-read from HBM;
  // noisy: the stored noisy observation + the filtered gyro (== the low-pass state)
  Quat q = quat_from_euler(e.roll, e.pitch, e.yaw);
  {
    float tx, ty, tz;
    target_at<TASK>(k, ref_lds, target_index<TASK>(step, k.agg, phase), tx, ty, tz);
    if (V::ON) {
      put_noisy_half<TASK, O + 4, VEC1>(row, S.oh, ns.lpf, h1, tx, ty, tz, pa1);
    } else {
      Quat qk = q;
      if (ctr_sign(ctr)) { qk.x = -q.x; qk.y = -q.y; qk.z = -q.z; qk.w = -q.w; }
      put_obs_half<TASK, O + 4, VEC1>(row, e, qk, h1, tx, ty, tz, pa1);
    }
  }

  // ---- aggregate_phy_steps x SimplePhysics.step_forward (envs/base.py:457-465) ---------------
  float *xm = S.xm;
  NoisyObs held = S.oh;  // ON, obs_rate > 1: self.state[0:10] of the reference
  uint32_t lat_idx = ctr_lat(ctr);
  // divisions by the per-env mass / inertia become multiplications by v_rcp_f32 results (1 ulp)
  const float inv_m = fast_rcp(par.m), inv_Jx = fast_rcp(par.Jx), inv_Jy = fast_rcp(par.Jy), inv_Jz = fast_rcp(par.Jz);
  for (int sub = 0; sub < k.agg; ++sub) {
    SubNoise sn;
    if (V::TN || V::ON) sub_noise<V>(a, rk, env_id, ix, sub, sn);
    // CrazyFlieAgent.apply_action, envs/agents.py:259-298 (+ PWM.act envs/control.py:94-100)
    float av[4] = {act.x, act.y, act.z, act.w};
    if (V::LAT) {  // agents.py:267-276: the controller sees the action of buf_size physics steps ago
      float4 *slot = a.st.lat + (long long)lat_idx * a.n + ix.global();  // (the ring index is per env)
      const float4 d = *slot;
      if (active) *slot = act;
      lat_idx = (lat_idx + 1u == (uint32_t)k.lat_steps) ? 0u : lat_idx + 1u;
      av[0] = d.x; av[1] = d.y; av[2] = d.z; av[3] = d.w;
    }
    float f[4], pwmv[4];
    if (V::CTRL == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) pwmv[j] = 30000.f + clampf(av[j], -1.f, 1.f) * 30000.f;
    } else {
      control_pwm<V::CTRL>(k.dt_nom, e, av, ps, pwmv);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float un = pwmv[j] * (1.0f / 60000.f);
      float noise1 = 1.0f;
      if (V::TN) {  // OUNoise.noise, envs/utils.py:104-108 (theta .15, mu 0); never reset
        ns.ou[j] = ns.ou[j] + (0.15f * (0.f - ns.ou[j]) + k.ou_sigma * sn.ou[j]);
        noise1 = 1.0f + ns.ou[j];
      }
      float n;
      if (V::MOTOR) {
        xm[j] = par.A[j] * xm[j] + (1.0f - par.A[j]) * fast_sqrt(un);
        n = noise1 * (xm[j] * xm[j]);
      } else {
        n = noise1 * un;
      }
      f[j] = par.K[j] * clampf(n, 0.f, 1.f);
    }
    // yaw torque: sum of +-(ftf1*f_i + ftf0); ftf0 cancels (envs/agents.py:295-297)
    const float tz_ = par.ftf1 * (-f[0] + f[1] - f[2] + f[3]);
    float R[9];
    matrix_from_quat(q, R);  // envs/physics.py:160 (quaternion of the PREVIOUS step)
    if (V::GE) {
      // BasePhysics.calculate_ground_effect, envs/physics.py:27-58, applied as extra per-motor
      // thrust (envs/physics.py:117-120); branch-free per-env scale
      const float ok = (fabsf(e.roll) < kHalfPi && fabsf(e.pitch) < kHalfPi) ? 1.f : 0.f;
      const float ox[4] = {0.028f, -0.028f, -0.028f, 0.028f};
      const float oy[4] = {-0.028f, -0.028f, 0.028f, 0.028f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float hz = fmaxf(e.pz + (R[6] * ox[j] + R[7] * oy[j]), k.h_clip);
        const float qq = k.prop_r * fast_rcp(4.f * hz);
        f[j] = f[j] + ok * (f[j] * k.gec * (qq * qq));
      }
    }
    const float thrust = ((f[0] + f[1]) + f[2]) + f[3];
    const float Fx = R[2] * thrust, Fy = R[5] * thrust, Fz = R[8] * thrust - k.G * par.m;
    const float tx_ = (-f[0] - f[1] + f[2] + f[3]) * k.Lq;  // envs/physics.py:167
    const float ty_ = (-f[0] + f[1] + f[2] - f[3]) * k.Lq;  // envs/physics.py:168
    const float Jwx = par.Jx * e.wx, Jwy = par.Jy * e.wy, Jwz = par.Jz * e.wz;
    const float t0 = tx_ - (e.wy * Jwz - e.wz * Jwy);  // tau - w x (J w), envs/physics.py:170-171
    const float t1 = ty_ - (e.wz * Jwx - e.wx * Jwz);
    const float t2 = tz_ - (e.wx * Jwy - e.wy * Jwx);
    const float dt = par.dt;
    e.vx += dt * (Fx * inv_m); e.vy += dt * (Fy * inv_m); e.vz += dt * (Fz * inv_m);    // :173,175
    e.wx += dt * (t0 * inv_Jx); e.wy += dt * (t1 * inv_Jy); e.wz += dt * (t2 * inv_Jz);  // :172,176
    e.px += dt * e.vx; e.py += dt * e.vy; e.pz += dt * e.vz;                             // :177
    e.roll += dt * e.wx; e.pitch += dt * e.wy; e.yaw += dt * e.wz;                       // :178
    // :179 -- with observation noise nothing reads the TRUE quaternion after the last sub-step (the observation carries
    // Q(noisy rpy), the next env.step rebuilds Q(rpy) from the stored angles): skip its three sincos there
    if (!V::ON || sub + 1 < k.agg) q = quat_from_euler(e.roll, e.pitch, e.yaw);
    e.pz = fmaxf(e.pz, 0.f);                                                             // :182
    // envs/base.py:464: compute_observation() whose result is dropped still advances the gyro
    // bias random walk and the low-pass filter
    if (V::ON) {
      if (!V::HOLD) {
        gyro_update(k, e, sn.bias_z, sn.rw_z, sn.to_z, ns);
      } else {
        // ... and, at an iteration that is a multiple of obs_rate, refreshes the held position / attitude /
        // velocity that later calls re-use (envs/hover.py:134-156)
        const int it = step * k.agg + sub;
        const bool fresh = (it % k.obs_rate) == 0;
        NoisyObs cand;
        sensor_observe(k, e, sn.full, ns, cand);  // (its gyro part == gyro_update with the same variates)
        held.x = fresh ? cand.x : held.x; held.y = fresh ? cand.y : held.y; held.z = fresh ? cand.z : held.z;
        held.qx = fresh ? cand.qx : held.qx; held.qy = fresh ? cand.qy : held.qy; held.qz = fresh ? cand.qz : held.qz;
        held.qw = fresh ? cand.qw : held.qw;
        held.vx = fresh ? cand.vx : held.vx; held.vy = fresh ? cand.vy : held.vy; held.vz = fresh ? cand.vz : held.vz;
      }
    }
  }
  PDS_STAMP(3);

  // ---- task: target, done, reward, cost (all on the TRUE state) -------------------------------
  float tx, ty, tz;
  target_at<TASK>(k, ref_lds, target_index<TASK>(step + 1, k.agg, phase1), tx, ty, tz);
  const float dx = e.px - tx, dy = e.py - ty, dz = e.pz - tz;
  const float dist = fast_sqrt(dx * dx + dy * dy + dz * dz);
  bool done = false;
  if (TASK == PDS_TASK_HOVER) {  // envs/hover.py:89-101
    constexpr float lim = 60.f * kPi / 180.f;
    constexpr float r2d = 180.f / kPi;
    done = (e.pz < 0.2f) || (fabsf(e.roll) > lim) || (fabsf(e.pitch) > lim) ||
           (fabsf(e.wx) * r2d > 300.f) || (fabsf(e.wy) * r2d > 300.f) || (fabsf(e.wz) * r2d > 300.f);
  } else if (TASK == PDS_TASK_CIRCLE) {  // envs/circle.py:116-120
    done = dist > 0.25f;
  }
  float reward;
  {  // envs/hover.py:169-187, envs/circle.py:183-204, envs/takeoff.py:155-174
    const float n0 = 0.5f * (clampf(act.x, -1.f, 1.f) + 1.f), n1 = 0.5f * (clampf(act.y, -1.f, 1.f) + 1.f);
    const float n2 = 0.5f * (clampf(act.z, -1.f, 1.f) + 1.f), n3 = 0.5f * (clampf(act.w, -1.f, 1.f) + 1.f);
    const float pen_act = k.pa * fast_sqrt(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3);
    float pen_rate = 0.f;
    if (TASK == PDS_TASK_CIRCLE) {  // a - env.last_action (previous action, envs/circle.py:186)
      const float d0 = act.x - h1.x, d1 = act.y - h1.y, d2 = act.z - h1.z, d3 = act.w - h1.w;
      pen_rate = k.arp * fast_sqrt(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
    }
    float pen_rpy = 0.f;
    if (k.pang != 0.f) pen_rpy = k.pang * fast_sqrt(e.roll * e.roll + e.pitch * e.pitch + e.yaw * e.yaw);
    const float pen_spin = k.pspin * fast_sqrt(e.wx * e.wx + e.wy * e.wy + e.wz * e.wz);
    const float pen_term = done ? k.pterm : 0.f;
    // envs/takeoff.py:165 multiplies the velocity norm by penalty_ACTION
    const float pv = (TASK == PDS_TASK_TAKEOFF) ? k.pa : k.pvel;
    float pen_vel = 0.f;
    if (pv != 0.f) pen_vel = pv * fast_sqrt(e.vx * e.vx + e.vy * e.vy + e.vz * e.vz);
    const float penalties = ((((pen_rpy + pen_rate) + pen_spin) + pen_vel) + pen_act) + pen_term;
    reward = -dist - penalties;
    if (TASK == PDS_TASK_TAKEOFF && e.pz < 0.08f) reward -= 1.f;  // envs/takeoff.py:172-173
  }
  float cost = 0.f;
  if (TASK == PDS_TASK_HOVER) {
    // envs/hover.py:103-129: state[10:13] is rpy_dot and state[13:16] is last_action[0:3] in the
    // get_state layout -- reproduced as is
    constexpr float rp_lim = 10.f * kPi / 180.f, dot_lim = 200.f * kPi / 180.f;
    const bool c = (fabsf(e.px) > 0.10f) || (fabsf(e.py) > 0.10f) || (e.pz > 1.20f) ||
                   (fabsf(e.roll) > rp_lim) || (fabsf(e.pitch) > rp_lim) ||
                   (fabsf(e.wx) > 0.25f) || (fabsf(e.wy) > 0.25f) || (fabsf(e.wz) > 0.25f) ||
                   (fabsf(act.x) > dot_lim) || (fabsf(act.y) > dot_lim) || (fabsf(act.z) > dot_lim);
    cost = c ? 1.f : 0.f;
  }
  const bool trunc = (step + 1) >= k.max_steps;  // gymnasium TimeLimit, __init__.py:11

  // ---- o(k+1) and u(k-1) -> second half of the row (envs/base.py:303-319) --------------------
  if (V::ON) {
    ObsNoise n;
    if (a.noise != nullptr) obs_noise_load(a.noise + ix.global() * k.agg * PDS_NOISE_FLOATS + PDS_N_OBS, n);  // (block of sub-step 0)
    else obs_noise_philox(env_id, rk, kBlkObsNoise, n);
    sensor_observe(k, e, n, ns, S.oh);
    if (V::HOLD) {  // per env: fresh observation or the held one + the fresh gyro
      const bool fresh = (((step + 1) * k.agg) % k.obs_rate) == 0;
      S.oh.x = fresh ? S.oh.x : held.x; S.oh.y = fresh ? S.oh.y : held.y; S.oh.z = fresh ? S.oh.z : held.z;
      S.oh.qx = fresh ? S.oh.qx : held.qx; S.oh.qy = fresh ? S.oh.qy : held.qy; S.oh.qz = fresh ? S.oh.qz : held.qz;
      S.oh.qw = fresh ? S.oh.qw : held.qw;
      S.oh.vx = fresh ? S.oh.vx : held.vx; S.oh.vy = fresh ? S.oh.vy : held.vy; S.oh.vz = fresh ? S.oh.vz : held.vz;
    }
    put_noisy_half<TASK, O + 4, VEC>(row2, S.oh, ns.lpf, act, tx, ty, tz, pa2);
  } else {
    put_obs_half<TASK, O + 4, VEC>(row2, e, q, act, tx, ty, tz, pa2);
  }

  S.ctr = ctr_pack((uint32_t)(step + 1), 0u, (uint32_t)phase1, lat_idx);
  if (regen_obs_variant<V>() && a.noise != nullptr) S.ctr |= kCtrOhBit;  // o(k+1) from injected variates: kept in oh0-2
  S.h2 = h1;   // u(k-1) becomes u(k-2)
  S.h1 = act;  // -> the ring slot that held u(k-2)
  // ---- auto-reset.  ~2 % of the envs finish per step under random actions, i.e. 3 of 4 waves
  // hold one or two finished envs.  Their last observation goes to final_obs (below, out of the
  // LDS tile); the reset itself: see RM_* above.
  // SplitReset<V> (round 6): the finished envs are reset by post_reset_kernel behind this launch, nothing of it is compiled in
  const bool need_reset = (V::SPLIT_RESET && STORE) ? false : (a.auto_reset && (done || trunc) && active);
  const unsigned long long reset_mask = __ballot(need_reset);  // wave-uniform
  const unsigned long long done_mask = (a.final_obs != nullptr || fin_lds != nullptr) ? reset_mask : 0ull;  // -> final_obs
  if (so != nullptr) { so->reward = reward; so->done = done; so->trunc = trunc; }
  bool was_reset = false;
  if (reset_mask != 0ull) {  // wave-uniform
    const int pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(reset_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)reset_mask, 0u));
    if constexpr (RM == RM_DEFERRED) {
      qcount = __popcll(reset_mask);
      if (need_reset) queue[pos] = (uint32_t)lane | ((uint32_t)ref_offset << 6);
    } else if constexpr (RM == RM_MERGED) {
      const StepArgs &a = reload_args<101, heavy_variant<V>()>(a_in, (int)o1);  // (shadows the parameter: see "coalesced stores" below)
      const int count = __popcll(reset_mask);
      if (need_reset) queue[pos] = (uint32_t)lane;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float4 u0 = act, mxr = make_float4(xm[0], xm[1], xm[2], xm[3]);
      reset_in_registers<V>(a, rk, ref_lds, queue, count, need_reset, pos, lane, wave_base, ref_offset, scratch, e, q,
                            u0, mxr, par, S.ctr);
      if (need_reset) {
        was_reset = true;
        S.h1 = u0; S.h2 = u0;
        xm[0] = mxr.x; xm[1] = mxr.y; xm[2] = mxr.z; xm[3] = mxr.w;
#pragma unroll
        for (int j = 0; j < 3; ++j) { ps.rate_int[j] = ps.rate_err[j] = ps.att_int[j] = ps.att_err[j] = 0.f; }
      }
    }
  }

  // ---- coalesced stores ----------------------------------------------------------------------
  // From here on the kernel arguments are read through a second, opaque view (reload_args): the pointers and
  // constants of the first half do not stay live in SGPRs across the step.
  {
  const StepArgs &a = reload_args<102, heavy_variant<V>()>(a_in, (int)o1);  // (o1: renewed per iteration of the K-step loop)
  const Consts &k = a.k;
  if (active) {
    // (RM_INLINE: a finished env's lane stores the FRESH state from inside the reset below, nothing here: no
    //  write-after-write between the two, hence no wait for these stores)
    if constexpr (STORE && RM != RM_INLINE) store_state<V>(a, ix, parity ^ 1, S, was_reset);
    if constexpr (STORE && RM == RM_INLINE) {
      if (!need_reset) store_state<V>(a, ix, parity ^ 1, S, false);
    }
    const Idx<V> jx = fresh<5>(ix);
    nt_store(at(a.reward + o1, jx), reward);
    nt_store(at(a.cost + o1, jx), cost);
    nt_store(at(a.term + o1, jx), (uint8_t)(done ? 1 : 0));
    nt_store(at(a.trunc + o1, jx), (uint8_t)(trunc ? 1 : 0));
  }
  PDS_STAMP(4);
  // LDS rows of this wave were written by its own lanes only: wave-synchronous, no block barrier
  const long long rem = a.n - wave_base;
#pragma unroll
  for (int pass = 0; pass < kWave / TR; ++pass) {
    if (TR != kWave) {
      if ((lane / TR) == pass) {
        float *dst = tile + (lane % TR) * TS;
        if constexpr (PARK) {
          if (pass == 1) {  // the parked first half of lanes 32-63 moves into the (flushed) tile
            const float *src = park + (lane - TR) * PS;
            if constexpr (TS != D) {
#pragma unroll
              for (int j = 0; j < HW / 4; ++j) reinterpret_cast<float4 *>(dst)[j] = reinterpret_cast<const float4 *>(src)[j];
            } else {
#pragma unroll
              for (int j = 0; j < HW; ++j) dst[j] = src[j];
            }
          }
          if constexpr (TS != D) {
            lds_store_row<HW>(dst + HW, rowbuf);
          } else {
#pragma unroll
            for (int j = 0; j < HW; ++j) dst[HW + j] = rowbuf[j];
          }
        } else {  // whole row from registers
          if constexpr (TS != D) {
            lds_store_row<D>(dst, rowbuf);
          } else {
#pragma unroll
            for (int j = 0; j < D; ++j) dst[j] = rowbuf[j];
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // last observation of each finished env -> final_obs: the whole wave copies one row
    // (D <= 64 contiguous floats) per finished env straight out of the LDS tile
    unsigned long long m = done_mask;
    if (TR != kWave) m &= (pass == 0) ? 0x00000000FFFFFFFFull : 0xFFFFFFFF00000000ull;
    while (m != 0ull) {
      const int src_lane = __builtin_ctzll(m);
      m &= m - 1ull;
      // non-temporal like the other streamed outputs (same box: 57.7 vs 58.4 us on Hover 2^20)
      if (lane < D) {
        const float v = tile[(src_lane % TR) * TS + lane];
        if (a.final_obs != nullptr) nt_store(lane_at<saddr_variant<V>(), 6>(a.final_obs + (o1 + wave_base + src_lane) * D, (uint32_t)lane), v);
        if (fin_lds != nullptr) fin_lds[src_lane * D + lane] = v;
      }
    }
    if constexpr (!STORE) { PDS_STAMP(6); }  // (K-step / rollout kernels: the single-step kernel uses slot 6 itself)
    if (RM != RM_DEFERRED && reset_mask != 0ull) {  // wave-uniform: the reset envs' rows become [o0, u0, o0', u0]
      if constexpr (RM == RM_INLINE) {
        // TakeOff ends episodes by truncation only, i.e. all 64 envs of a wave at once every 500 steps: there every
        // lane computes its own Philox blocks (DirectWords) and the reset is evaluated once for the whole wave.
        // Elsewhere 1-3 % of the envs finish per step: the Philox blocks of the finished envs (5-22 per env, ~650
        // cycles each when one lane computes them in a row) are computed side by side by the whole wave into an
        // LDS scratch, kInlineEnvs envs per pass like the drain; the reset itself is evaluated by the lanes that
        // own the envs, so the fresh state stays in registers.
        auto evaluate = [&](const auto &dw) {
          was_reset = true;
          const float stale_w[3] = {e.wx, e.wy, e.wz};
          const float bias[3] = {ns.bias[0], ns.bias[1], ns.bias[2]};
          ResetOut r;
          reset_compute<V>(a, ref_lds, dw, ctr_pack(0u, 0u, (uint32_t)ref_offset), nullptr, stale_w, bias, r);
          if constexpr (STORE) {
            // single-step kernel: the finished env's lane has not stored its (terminal) state -- see "coalesced
            // stores" -- and stores the fresh one here, straight out of the reset's result: the step's own state
            // registers are dead by now, so the result does not have to be merged into them (the K-step kernel's
            // form below costs 9-88 spilled VGPRs under the single-step kernel's 168-register cap)
            EnvState F;
            F.e = r.e; F.ctr = r.ctr | ((regen_obs_variant<V>() && a.noise != nullptr) ? kCtrOhBit : 0u); F.h1 = r.u0; F.h2 = r.u0;
            F.xm[0] = r.mx.x; F.xm[1] = r.mx.y; F.xm[2] = r.mx.z; F.xm[3] = r.mx.w;
            F.par = r.par;
#pragma unroll
            for (int j = 0; j < 4; ++j) F.ns.ou[j] = ns.ou[j];  // the OU state is never reset (envs/agents.py:377-386)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              F.ns.bias[j] = r.ns.bias[j]; F.ns.lpf[j] = r.ns.lpf[j];
              F.ps.rate_int[j] = F.ps.rate_err[j] = F.ps.att_int[j] = F.ps.att_err[j] = 0.f;
            }
            F.oh = r.ob;
            store_state<V>(a, ix, parity ^ 1, F, true);
          } else {
            e = r.e;
            S.ctr = r.ctr;
            S.h1 = r.u0; S.h2 = r.u0;
            xm[0] = r.mx.x; xm[1] = r.mx.y; xm[2] = r.mx.z; xm[3] = r.mx.w;
            if (V::DR) par = r.par;
#pragma unroll
            for (int j = 0; j < 3; ++j) { ps.rate_int[j] = ps.rate_err[j] = ps.att_int[j] = ps.att_err[j] = 0.f; }
          }
          float tx0, ty0, tz0;
          target_at<TASK>(k, ref_lds, target_index<TASK>(0, k.agg, (int)ctr_off(r.ctr)), tx0, ty0, tz0);
          float *dst = tile + lane * TS;
          if (V::ON) {
            if constexpr (!STORE) {
#pragma unroll
              for (int j = 0; j < 3; ++j) { ns.bias[j] = r.ns.bias[j]; ns.lpf[j] = r.ns.lpf[j]; }
              S.oh = r.ob;
            }
            put_noisy_half<TASK, O + 4, (TS != D)>(dst, r.oa, r.lpf_a, r.u0, tx0, ty0, tz0, r.u0);
            put_noisy_half<TASK, O + 4, (TS != D)>(dst + O + 4, r.ob, r.ns.lpf, r.u0, tx0, ty0, tz0, r.u0);
          } else {
            put_obs_half<TASK, O + 4, (TS != D)>(dst, r.e, r.q, r.u0, tx0, ty0, tz0, r.u0);
            put_obs_half<TASK, O + 4, (TS != D)>(dst + O + 4, r.e, r.q, r.u0, tx0, ty0, tz0, r.u0);
          }
          if constexpr (V::LAT) {
#pragma unroll
            for (int b = 0; b < kMaxLatSteps; ++b)
              if (b < k.lat_steps) {
                // (by value: a conditional between the two lvalues would select a POINTER and push `r` into scratch memory)
                float4 v = r.lat.r[b < kMaxLatSteps - 1 ? b : 0];
                if (b == k.lat_steps - 1 || b == kMaxLatSteps - 1) v = r.u0;
                *at(a.st.lat + (long long)b * a.n, ix) = v;
              }
          }
        };
        if constexpr (!inline_coop_variant<V>()) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();  // every final_obs row has left the tile
          if (need_reset) evaluate(DirectWords(env_id, rk));
        } else {
          constexpr int IE = inline_envs<V, STORE>();
          const int count = __popcll(reset_mask);
          const int pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(reset_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)reset_mask, 0u));
          if (need_reset) queue[pos] = (uint32_t)lane;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();  // every final_obs row has left the tile; the queue is complete
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          for (int base = 0; base < count; base += IE) {
            const int cnt = min(IE, count - base);  // wave-uniform
#ifdef PDS_STAMPS_RESET
            const unsigned long long z0 = __builtin_amdgcn_s_memtime();
#endif
            fill_reset_variates<V>(a, rk, queue + base, cnt, lane, wave_base, reinterpret_cast<float *>(scratch));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef PDS_STAMPS_RESET
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long z1 = __builtin_amdgcn_s_memtime();
#endif
            if (need_reset && pos >= base && pos < base + cnt)
              evaluate(LdsVariates{reinterpret_cast<const float *>(scratch) + (pos - base) * variates_floats<V>()});
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // the scratch is refilled by the next pass
#ifdef PDS_STAMPS_RESET
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long z2 = __builtin_amdgcn_s_memtime();
            stamp_[8] += z1 - z0; stamp_[9] += z2 - z1;
#endif
          }
        }
      } else {
        if (need_reset && (TR == kWave || (lane / TR) == pass)) {
          float tx0, ty0, tz0;
          target_at<TASK>(k, ref_lds, target_index<TASK>(0, k.agg, (int)ctr_off(S.ctr)), tx0, ty0, tz0);
          float *dst = tile + (lane % TR) * TS;
          put_obs_half<TASK, O + 4, (TS != D)>(dst, e, q, S.h1, tx0, ty0, tz0, S.h1);
          put_obs_half<TASK, O + 4, (TS != D)>(dst + O + 4, e, q, S.h1, tx0, ty0, tz0, S.h1);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if constexpr (!STORE) { PDS_STAMP(7); }
    const long long left = rem - pass * TR;
    // (the history rollout, csrc/pds_rollout_hist.h, keeps the row in LDS only: obs == nullptr; compile-time true for pds_step)
    if (left > 0 && (STORE || a.obs != nullptr))
      flush_tile<D, TR, saddr_variant<V>()>(tile, a.obs + (o1 + wave_base + pass * TR) * D, left >= TR ? TR : (int)left, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the tile is rewritten by the next pass / the reset drain / the next step
  }
  PDS_STAMP(5);
  }
  return was_reset;
}

// One 64-env tile per wave, one launch covers all tiles (the hardware dispatcher balances blocks
// whose deferred-reset drains differ in length).  A persistent grid-stride variant (256 x 3 resident
// blocks, register prefetch of the next tile) measured 66.7 us vs 62.7 us for this shape, doubled the
// input registers, and -- because the compiler hoists every loop-invariant kernel argument out of
// the tile loop -- pushed the kernel over the SGPR budget (137 v_writelane/v_readlane spill
// instructions in the main path), so there is no tile loop here.
// __launch_bounds__(256, 3): LDS admits 3 blocks (12 waves) per CU, so cap VGPRs at 168.
// The lean variants are additionally held to 128 VGPRs (min 4 waves/SIMD): measured 2-3 % faster
// (62.7 vs 64.3 us Hover, 63.5 vs 64.7 us TakeOff+GE on the same box); the observation-noise
// variants spill badly under that cap (154 vs 106 us) and keep 168.
#ifndef PDS_MIN_WAVES_LEAN
#define PDS_MIN_WAVES_LEAN 4
#endif
#ifndef PDS_MIN_WAVES
#define PDS_MIN_WAVES (V::ON ? (four_block_variant<V>() ? 4 : 3) : PDS_MIN_WAVES_LEAN)
#endif

// Auto-reset in registers before the stores, or deferred drain after them.  Measured on one MI355X
// box, merged vs deferred: Hover 2^20 58.45 vs 60.0 us, Hover 2^21 108.5 vs 111.0 us; but Circle +
// PT1 + DR at 2^20 82.9 vs 78.6 us (181 VGPRs => 2 waves/SIMD), TakeOff (resets only by the 500-step
// truncation) 61.4 vs 60.7 us, and under the half tile's 128-VGPR cap it spills (Circle 262 144: 39
// vs 21 us) -- so those keep the deferred drain.
// Round 3: the observation-noise and latency variants (no merged form) reset in registers too in the single-step
// kernel -- the inline reset of the K-step kernel, with the state stored AFTER it -- instead of the deferred drain,
// whose `s_waitcnt vmcnt(0)` put a full store round trip, a second evaluation pass and ~25 scattered stores behind
// the wave's own stores on more than half of the waves.  (TakeOff, whose envs only finish by the 500-step
// truncation, keeps the drain.)
template <class V, int TR>
constexpr bool inline_reset_single_step() {  // (the rule itself: csrc/pds_types.h inline_single_step_rule, shared with the host)
  return TR == kWave && inline_single_step_rule(V::TASK, V::ON, V::LAT, V::CTRL);
}

#ifndef PDS_MERGED_HALF_PT1DR
#define PDS_MERGED_HALF_PT1DR 1  // round 3: fits since the half tile parks the first row half in LDS (A/B: 0 = deferred drain)
#endif
template <class V, int TR = kWave>
constexpr bool merged_reset_variant() {
  // (the half tile keeps the whole observation row in registers: PT1 + DR would spill 61 VGPRs there)
  return PDS_MERGED_RESET && !V::ON && !V::LAT && V::TASK != PDS_TASK_TAKEOFF && (TR == kWave || !(V::MOTOR && V::DR) || (PDS_MERGED_HALF_PT1DR && V::CTRL == 0 && !V::TN));  // (PID / thrust-noise + PT1 + DR: 6-22 spilled VGPRs under the half tile's cap)
}

// The kernel arguments (StepArgs, ~10 cache lines) are read with scalar loads that the compiler
// places next to their uses -- about 30 of them along the step, each a scalar-cache miss (the
// kernarg segment is rewritten by the host for every launch) whose latency the wave waits out on
// the spot: measured with the s_memtime stamps, ~250 cycles each, a third of a wave's lifetime when
// one wave runs per SIMD (65 536 envs).  Touching every line once at kernel entry (all misses in
// flight together, one wait) turns the later loads into scalar-cache hits.
#ifndef PDS_KERNARG_PREFETCH
#define PDS_KERNARG_PREFETCH 1
#endif
PDS_DEV void prefetch_kernargs() {
#if PDS_KERNARG_PREFETCH
  // one dword from every 64-byte line that starts inside the kernarg segment
  static_assert(sizeof(StepArgs) > 8 * 64 && sizeof(StepArgs) <= 9 * 64, "prefetch_kernargs touches lines 0..8: adjust");
  const auto p = __builtin_amdgcn_kernarg_segment_ptr();
  uint32_t t0, t1, t2, t3, t4, t5, t6, t7, t8;
  asm volatile(
      "s_load_dword %0, %9, 0x0\n\ts_load_dword %1, %9, 0x40\n\ts_load_dword %2, %9, 0x80\n\t"
      "s_load_dword %3, %9, 0xc0\n\ts_load_dword %4, %9, 0x100\n\ts_load_dword %5, %9, 0x140\n\t"
      "s_load_dword %6, %9, 0x180\n\ts_load_dword %7, %9, 0x1c0\n\ts_load_dword %8, %9, 0x200\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8)
      : "s"(p)
      : "memory");
#endif
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one; MI355X_MICROARCH.md,
// workgroup dispatch).  With the identity mapping every XCD's L2 / fabric port sees every 8th 4-KiB piece of
// each array; with this remap the blocks of one XCD take CONSECUTIVE tiles, i.e. every XCD streams one
// contiguous eighth of each array.  Speed only (any permutation of blocks over tiles is correct; grids that are
// not a multiple of 8 keep the identity).  Same-box A/B: Hover 2^20 56.4 -> 55.5 us, TakeOff + GE 58.6 -> 57.8,
// Hover 2^21 107.7 -> 105.3 (86 % of the HBM peak), neutral at 2^19 and on the single-round configs.
#ifndef PDS_XCD_REMAP
#define PDS_XCD_REMAP 1
#endif
#define PDS_WAVE_LDS(V, TR, RM, ST)                                                                      \
  __shared__ __attribute__((aligned(16))) float tile_all[(kBlock / kWave) * TR * tile_stride<V::D>()]; \
  constexpr int kParkFloats_ = (TR == kWave || !park_variant<V>()) ? 0 : (kWave - TR) * park_stride<V::O + 4>(); \
  __shared__ __attribute__((aligned(16))) float park_all[kParkFloats_ > 0 ? (kBlock / kWave) * kParkFloats_ : 4]; \
  const float2 *ref_lds = nullptr; /* (the Circle table of rounds 1-2: the reference point is evaluated now) */ \
  __shared__ uint32_t queue_all[(kBlock / kWave) * kQueueCap];                                        \
  constexpr int kScratchU4_ = (RM == RM_MERGED) ? kMergedScratchU4 : ((RM == RM_INLINE && inline_coop_variant<V>() && !(V::SPLIT_RESET && ST)) ? inline_envs<V, ST>() * scratch_stride<V>() : 0); \
  __shared__ U4 scratch_all[kScratchU4_ > 0 ? (kBlock / kWave) * kScratchU4_ : 1];                     \
  const int tid = threadIdx.x;                                                                         \
  const int lane = tid & (kWave - 1);                                                                  \
  const int wave = saddr_variant<V>() ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6); /* SADDR form: wave-uniform, everything derived from it lives in SGPRs */ \
  U4 *scratch = scratch_all + wave * kScratchU4_;                                                      \
  uint32_t *queue = queue_all + wave * kQueueCap;                                                      \
  float *tile = tile_all + wave * (TR * tile_stride<V::D>());                                          \
  float *park = park_all + wave * kParkFloats_;

#define PDS_WAVE_SETUP(V, TR, RM, ST)                                                                    \
  PDS_WAVE_LDS(V, TR, RM, ST)                                                                            \
  const long long ntiles = (a.n + kWave - 1) / kWave;                                                  \
  long long blk_ = blockIdx.x;                                                                         \
  if (PDS_XCD_REMAP) { /* blocks b, b + 8, ... (one XCD) take consecutive tiles */                     \
    const long long nb_ = gridDim.x, per_ = nb_ / 8;                                                   \
    if (per_ * 8 == nb_) blk_ = (blk_ % 8) * per_ + blk_ / 8;                                          \
  }                                                                                                    \
  const long long t = blk_ * (kBlock / kWave) + wave;                                                  \
  if (t >= ntiles) return; /* wave-uniform */                                                          \
  const long long wave_base = t * kWave;                                                               \
  const long long rem_ = a.n - wave_base; /* envs of this tile and beyond: > 0, wave-uniform */         \
  const bool active = rem_ >= kWave || lane < (int)rem_;                                               \
  /* tail lanes recompute the last env, stores masked */                                               \
  const Idx<V> ix{wave_base, active ? (uint32_t)lane : (uint32_t)rem_ - 1u};

#ifdef PDS_STAMPS
#define PDS_STAMP_ARG , stamp_
#define PDS_STAMP_DECL unsigned long long stamp_[kStampSlots] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; stamp_[0] = __builtin_amdgcn_s_memtime(); stamp_[8] = __builtin_amdgcn_s_memrealtime();
#define PDS_STAMP_FLUSH                                                                                \
  do {                                                                                                 \
    PDS_STAMP_WAIT(7);                                                                                 \
    stamp_[9] = __builtin_amdgcn_s_memrealtime();                                                      \
    if (a.stamps != nullptr && lane == 0)                                                              \
      for (int j_ = 0; j_ < kStampSlots; ++j_) a.stamps[t * kStampSlots + j_] = stamp_[j_];            \
  } while (0)
#else
#define PDS_STAMP_ARG
#define PDS_STAMP_DECL
#define PDS_STAMP_FLUSH do { } while (0)
#endif

// One 64-env tile per wave, one block per 4 tiles, no tile loop.  Round 3 re-tried persistent waves (at most the
// resident number of blocks, each wave walking over several tiles with a static XCD-aware schedule; the kernel
// arguments re-read per tile through reload_args(), so the loop no longer costs SGPR spills as it did in rounds 1-2):
// Hover 2^20 64.0 us against 55.6 us for this form, 62.6 us with the next tile's loads software-pipelined in front
// of the current tile's stores (gfx9 counts loads and stores on one in-order vmcnt, so a load issued behind a
// tile's stores waits for their acknowledgement), config 6 99.6 / 122 (spills) vs 88.9, 2^21 114.7 vs 104.3
// (profiles/r03_ab_persistent.txt).  Waves that all start together run their load / compute / store phases in
// lockstep across the chip; the dispatcher's staggered block starts are what overlaps one block's memory phases
// with another's arithmetic.
template <class V, int TR>
__global__ __launch_bounds__(kBlock, (PDS_MIN_WAVES) * (256 / kBlock)) void step_kernel(const StepArgs a) {
  PDS_STAMP_DECL
  prefetch_kernargs();
  constexpr int RM = merged_reset_variant<V, TR>() ? RM_MERGED : (inline_reset_single_step<V, TR>() ? RM_INLINE : RM_DEFERRED);
  PDS_WAVE_SETUP(V, TR, RM, true)
  // The loads are issued before anything else so that the scalar preamble of the kernel
  // (kernel-argument loads, uniform constants) overlaps with their latency.
  Loaded cur;
  load_env<V>(a, ix, t, cur);
  __builtin_amdgcn_sched_barrier(0);
  PDS_STAMP(1);
  PDS_STAMP_WAIT(2);
  RngKey rk{a.seed_lo, a.seed_hi, 0u, 0u};
  int parity;
  rk.tick_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.clk.x);
  rk.tick_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.clk.y);
  parity = __builtin_amdgcn_readfirstlane((int)cur.clk.z) & 1;
  EnvState S;
  unpack_state<V>(a.k, cur, parity, S);
  init_kept_obs<V>(a, rk, ix, S);
  int qcount = 0;  // wave-uniform
  step_once<V, TR, RM, true>(a, 0ll, rk, parity, ref_lds, tile, park, queue, scratch, lane, wave_base, ix, active, cur.act, S, qcount, nullptr, nullptr PDS_STAMP_ARG);
  if (RM == RM_DEFERRED && qcount > 0) {
    const float own_w[3] = {S.e.wx, S.e.wy, S.e.wz};
    drain_reset_queue<V>(reload_args<104, heavy_variant<V>()>(a), rk, ref_lds, queue, qcount, lane, wave_base, tile, own_w, S.ns.bias);
  }
  PDS_STAMP(6);
  advance_clock(reload_args<105, heavy_variant<V>()>(a).st.clk, t, rk, parity ^ 1, 1u, lane);
  PDS_STAMP_FLUSH;
}

// K env.step()s per launch for open-loop action sequences (pds_step_k): the env state stays in
// registers, per step only the action (16 B) comes in and the observation row, reward, cost and flags
// go out -- 4 D + 26 B per env-step instead of 4 D + 178 B, and one launch instead of K.
// K-step kernel: the state of the env stays in registers across the loop.  Rounds 2-3: the noise variants needed more than
// the 168 VGPRs of 3 blocks per CU (27-74 spilled registers inside the loop at that cap) and were built for 2
// (profiles/r02_variant_timings_stepk_minwaves.txt).  Round 4: with the kept observation read from memory (StoredOh, below)
// they need 144-186, and at the 168 cap 45 of the 288 kernels spill 1-15 registers: 3 blocks per CU for all of them.  Same
// box, 2^20 envs, K = 8, us per env-step, cap 168 vs 256: Hover latency ring + noise 69.0 vs 83.6, Kalman hold 74.2 vs 86.9, Circle
// PT1 + noise 61.5 vs 70.7, Circle latency 77.9 vs 81.6, everything else equal (profiles/r04_ab_stepk_regen.txt).
#ifndef PDS_STEPK_OPAQUE_KEY
#define PDS_STEPK_OPAQUE_KEY 1  // A/B: 0 = round-3 form (key schedule hoisted out of the K loop and spilled)
#endif
#ifndef PDS_STEPK_MIN_WAVES_OF
#define PDS_STEPK_MIN_WAVES_OF(V) 3
#endif
template <class V_>
__global__ __launch_bounds__(kBlock, (PDS_STEPK_MIN_WAVES_OF(V_)) * (256 / kBlock)) void step_k_kernel(const StepArgs a) {
  // (round 4) the kept noisy observation comes out of oh0-2 and goes back there: regenerating it in the prologue (the
  // single-step kernels' way) took the register allocation of the whole loop from 152 to 200 VGPRs, i.e. from three blocks
  // per CU to two -- Hover noise + DR, K = 8: 79.1 vs 64.7 us per env-step, 2 sub-steps 114.3 vs 95.9
  // (profiles/r04_ab_stepk_regen.txt).  pds_step_k launches materialize_oh_kernel in front of this kernel.
  using W = std::conditional_t<regen_obs_variant<V_>(), StoredOh<V_>, V_>;
  constexpr int TR = kWave;
  constexpr int RM = merged_reset_variant<W>() ? RM_MERGED : RM_INLINE;
#ifdef PDS_STAMPS
  unsigned long long stamp_[kStampSlots];
#endif
  prefetch_kernargs();
  PDS_WAVE_SETUP(W, TR, RM, false)
  Loaded cur;
  load_env<W>(a, ix, t, cur);
  RngKey rk{a.seed_lo, a.seed_hi, 0u, 0u};
  int parity;
  rk.tick_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.clk.x);
  rk.tick_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.clk.y);
  parity = __builtin_amdgcn_readfirstlane((int)cur.clk.z) & 1;
  const RngKey rk0 = rk;
  EnvState S;
  unpack_state<W>(a.k, cur, parity, S);
  init_kept_obs<W>(a, rk, ix, S);
  const int K = a.k_steps;
  float4 act = cur.act;  // actions[0]
  int qcount = 0;
  for (int s = 0; s < K; ++s) {
    // a fresh view of the kernel arguments per iteration: what the loop body needs is re-read (scalar-cache
    // hits) instead of being hoisted out of the loop into SGPRs that do not exist (41-79 spills in round 2)
    const StepArgs &al = reload_args<106, PDS_STEPK_OPAQUE_KEY || heavy_variant<W>()>(a, s);
    // ... and the Philox key: the 2 x 10 round keys (seed + r x Weyl constant) are loop-invariant, so the compiler forms
    // them in the loop header and -- with the 102 SGPRs taken -- spills them to VGPR lanes there and reads them back in
    // every round of every Philox call of every iteration (2-45 spilled SGPRs per step_k kernel in round 3).  An opaque
    // copy of the seed per iteration makes the schedule part of the iteration: ~20 scalar adds, live only where used.
    RngKey rks = rk;
    if (PDS_STEPK_OPAQUE_KEY) asm volatile("" : "+s"(rks.seed_lo), "+s"(rks.seed_hi));
    // ... and the lane index: every lane predicate of the body (tile flush bounds, row ownership, `lane < D`) is
    // loop-invariant too and would be kept as a 64-bit mask in an SGPR pair from the loop header on
    int lane_s = lane;
    if (PDS_STEPK_OPAQUE_KEY) asm volatile("" : "+v"(lane_s));
    float4 act_next = act;
    if (s + 1 < K) act_next = nt_load4(at(al.actions + (long long)(s + 1) * al.n, ix));  // in flight during step s
    step_once<W, TR, RM, false>(al, (long long)s * al.n, rks, parity, ref_lds, tile, park, queue, scratch, lane_s, wave_base, ix, active, act, S, qcount, nullptr, nullptr PDS_STAMP_ARG);
    act = act_next;
    parity ^= 1;
    rk.tick_lo += 1u;
    if (rk.tick_lo == 0u) rk.tick_hi += 1u;
  }
  const StepArgs &az = reload_args<107, heavy_variant<W>()>(a);
  if (active) store_state<W>(az, ix, parity, S, true);
  advance_clock(az.st.clk, t, rk0, parity, (uint32_t)K, lane);
}

// ---- host-side dispatch: runtime flags -> template instantiation ----------------------------------
template <class V>
inline void launch_variant(int kind, bool half_tile, dim3 grid, hipStream_t s, const StepArgs &a) {
  if (kind == kLaunchReset || kind == kLaunchPostReset) {
    // the reset kernels do not depend on GE / TN / CTRL / HOLD: the launch_* families fold those flags before
    // they get here, so only the folded variants are instantiated (48 kernels instead of 472)
    if constexpr (!V::GE && !V::TN && V::CTRL == 0 && !V::HOLD) {
      if (kind == kLaunchReset) hipLaunchKernelGGL((reset_kernel<V>), grid, dim3(kBlock), 0, s, a);
      else if constexpr (V::ON || V::LAT) hipLaunchKernelGGL((post_reset_kernel<V>), grid, dim3(kWave), 0, s, a);
      else abort();
    } else abort();
  } else if (kind == kLaunchStepK) {
    // (the PID control modes have no K-step kernel: pds_step_k loops over pds_step for them)
    if constexpr (V::CTRL == 0) hipLaunchKernelGGL((step_k_kernel<V>), grid, dim3(kBlock), 0, s, a);
  } else {
#if PDS_STORED_OH_FROM_AGG > 0
    if constexpr (regen_obs_variant<V>()) {
      // stored vs regenerated kept observation as a per-launch choice (same bits either way): see kLaunchStepStored
      if (kind == kLaunchStepStored) {
        hipLaunchKernelGGL((step_kernel<StoredOh<V>, kWave>), grid, dim3(kBlock), 0, s, a);
        return;
      }
    }
#endif
    if constexpr (!V::ON && !V::LAT) {
      if (half_tile) {
        hipLaunchKernelGGL((step_kernel<V, kHalfTileRows>), grid, dim3(kBlock), 0, s, a);
        return;
      }
    }
    if constexpr (inline_reset_single_step<V, kWave>()) {
      if (kind == kLaunchStepSplit) {  // the host launches post_reset_kernel behind it (csrc/pds_api.hip)
        hipLaunchKernelGGL((step_kernel<SplitReset<V>, kWave>), grid, dim3(kBlock), 0, s, a);
        return;
      }
    }
    hipLaunchKernelGGL((step_kernel<V, kWave>), grid, dim3(kBlock), 0, s, a);
  }
}

template <int TASK, int CTRL, bool LAT, bool HOLD, bool MOTOR, bool DR, bool GE, bool TN, bool ON>
struct MakeVariant {
  using type = Variant<TASK, MOTOR, DR, GE, TN, ON, CTRL, LAT, HOLD>;
};

// binds the boolean flags one by one: Bs... = MOTOR, DR, GE, TN, ON
template <int TASK, int CTRL, bool LAT, bool HOLD, bool... Bs>
struct VariantDispatch {
  template <typename... Rest>
  static void run(int kind, bool half, dim3 grid, hipStream_t s, const StepArgs &a, bool first, Rest... rest) {
    if (first) VariantDispatch<TASK, CTRL, LAT, HOLD, Bs..., true>::run(kind, half, grid, s, a, rest...);
    else VariantDispatch<TASK, CTRL, LAT, HOLD, Bs..., false>::run(kind, half, grid, s, a, rest...);
  }
  static void run(int kind, bool half, dim3 grid, hipStream_t s, const StepArgs &a) {
    static_assert(sizeof...(Bs) == 5, "MOTOR, DR, GE, TN, ON");
    launch_variant<typename MakeVariant<TASK, CTRL, LAT, HOLD, Bs...>::type>(kind, half, grid, s, a);
  }
};

// Families (one translation unit each, see pds_task_*.hip):
//  base: control_mode PWM, no latency: motor x DR x GE x TN x ON  (32 step variants, half + full tile)
//  pid:  AttitudeRate / Attitude, no ground effect, no latency     (2 x 16)
//  lat:  use_latency, any control mode; ground effect with PWM only  (32 + 2 x 16; TakeOff: PWM only)
// The reset kernel does not depend on GE / TN / CTRL: those flags are folded to false / 0 for it.
template <int TASK>
inline void launch_base(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a) {
  if (kind == kLaunchReset || kind == kLaunchPostReset) VariantDispatch<TASK, 0, false, false>::run(kind, false, grid, s, a, f.motor, f.dr, false, false, f.on);
  else VariantDispatch<TASK, 0, false, false>::run(kind, f.half_tile, grid, s, a, f.motor, f.dr, f.ge, f.tn, f.on);
}
template <int TASK>
inline void launch_pid(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a) {
  if (f.ctrl == 1) VariantDispatch<TASK, 1, false, false>::run(kind, f.half_tile, grid, s, a, f.motor, f.dr, false, f.tn, f.on);
  else VariantDispatch<TASK, 2, false, false>::run(kind, f.half_tile, grid, s, a, f.motor, f.dr, false, f.tn, f.on);
}
// round 5: the PID modes with the ground-effect extension (envs/physics.py:27-58 x envs/control.py:120-287), a family (and a
// translation unit, pds_task_*_pid_ge.hip) of its own: 2 x 16 variants; not with the latency ring or the Kalman hold
template <int TASK, int CTRL>
inline void launch_pid_ge_ctrl(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a) {
#define PDS_PGE(M, R, T, O) launch_variant<Variant<TASK, M, R, true, T, O, CTRL, false, false>>(kind, f.half_tile, grid, s, a)
#define PDS_PGE_O(M, R, T) do { if (f.on) PDS_PGE(M, R, T, true); else PDS_PGE(M, R, T, false); } while (0)
#define PDS_PGE_T(M, R) do { if (f.tn) PDS_PGE_O(M, R, true); else PDS_PGE_O(M, R, false); } while (0)
  if (f.motor) { if (f.dr) PDS_PGE_T(true, true); else PDS_PGE_T(true, false); }
  else { if (f.dr) PDS_PGE_T(false, true); else PDS_PGE_T(false, false); }
#undef PDS_PGE_T
#undef PDS_PGE_O
#undef PDS_PGE
}
template <int TASK>
inline void launch_pid_ge(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a) {
  if (f.ctrl == 1) launch_pid_ge_ctrl<TASK, 1>(kind, f, grid, s, a);
  else launch_pid_ge_ctrl<TASK, 2>(kind, f, grid, s, a);
}
template <int TASK>
inline void launch_lat(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a) {
  if (kind == kLaunchReset || kind == kLaunchPostReset) { VariantDispatch<TASK, 0, true, false>::run(kind, false, grid, s, a, f.motor, f.dr, false, false, f.on); return; }
  if (TASK == PDS_TASK_TAKEOFF || f.ctrl == 0) VariantDispatch<TASK, 0, true, false>::run(kind, false, grid, s, a, f.motor, f.dr, f.ge, f.tn, f.on);
  else if constexpr (TASK != PDS_TASK_TAKEOFF) {  // TakeOff fixes control_mode='PWM', envs/takeoff.py:225
    if (f.ctrl == 1) VariantDispatch<TASK, 1, true, false>::run(kind, false, grid, s, a, f.motor, f.dr, false, f.tn, f.on);
    else VariantDispatch<TASK, 2, true, false>::run(kind, false, grid, s, a, f.motor, f.dr, false, f.tn, f.on);
  }
}

// hold: observation noise with obs_rate > 1 (the Kalman-hold branch): motor x DR x GE x TN for control_mode PWM with and
// without the latency ring; round 4: also the PID control modes (no ground effect there), with and without latency
template <int TASK, int CTRL, bool LAT>
inline void launch_hold_family(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a) {
  // ON = true, HOLD = true
#define PDS_HOLD(M, R, G, T) launch_variant<Variant<TASK, M, R, G, T, true, CTRL, LAT, true>>(kind, false, grid, s, a)
#define PDS_HOLD_T(M, R, G) do { if (f.tn) PDS_HOLD(M, R, G, true); else PDS_HOLD(M, R, G, false); } while (0)
#define PDS_HOLD_G(M, R) do { if (CTRL == 0 && f.ge) { if constexpr (CTRL == 0) PDS_HOLD_T(M, R, true); } else PDS_HOLD_T(M, R, false); } while (0)
  if (f.motor) { if (f.dr) PDS_HOLD_G(true, true); else PDS_HOLD_G(true, false); }
  else { if (f.dr) PDS_HOLD_G(false, true); else PDS_HOLD_G(false, false); }
#undef PDS_HOLD_G
#undef PDS_HOLD_T
#undef PDS_HOLD
}
template <int TASK>
inline void launch_hold(int kind, const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a) {
  if constexpr (TASK == PDS_TASK_TAKEOFF) {  // TakeOff fixes control_mode = 'PWM' (envs/takeoff.py:225)
    if (f.lat) launch_hold_family<TASK, 0, true>(kind, f, grid, s, a);
    else launch_hold_family<TASK, 0, false>(kind, f, grid, s, a);
  } else {
    if (f.lat) {
      if (f.ctrl == 0) launch_hold_family<TASK, 0, true>(kind, f, grid, s, a);
      else if (f.ctrl == 1) launch_hold_family<TASK, 1, true>(kind, f, grid, s, a);
      else launch_hold_family<TASK, 2, true>(kind, f, grid, s, a);
    } else {
      if (f.ctrl == 0) launch_hold_family<TASK, 0, false>(kind, f, grid, s, a);
      else if (f.ctrl == 1) launch_hold_family<TASK, 1, false>(kind, f, grid, s, a);
      else launch_hold_family<TASK, 2, false>(kind, f, grid, s, a);
    }
  }
}

}  // namespace pds
