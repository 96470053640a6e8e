// pds_step.h -- the fused lockstep CrazyFlie SimplePhysics step for MI355X (gfx950, wave64).
//
// One thread per environment.  Per step a thread streams its SoA state quads (16 B/lane, fully
// coalesced), advances PWM->thrust, Newton-Euler force/torque, semi-implicit Euler and
// Euler->quaternion in registers, evaluates the task's reward / cost / termination, queues finished
// envs for a deferred reset from a counter-based Philox stream, and stages its observation row in a
// per-wave LDS tile so that the row-major [N, D] observation tensor is written with contiguous
// 1 KiB wave stores instead of 64 strided rows.  HBM-bound by design: no MFMA (there is no dense
// contraction on this path).
//
// Reference (paths relative to phoenix_drone_simulation/): envs/physics.py:130-200,
// envs/agents.py:259-298, envs/control.py:94-100, envs/base.py:303-319,433-475, envs/hover.py,
// envs/circle.py, envs/takeoff.py, envs/sensors.py:75-134, envs/utils.py:59-108.
#pragma once
#include "pds_reset.h"

namespace pds {

// Wave-cooperative copy of this wave's [rows, D] LDS tile to global memory (contiguous region).
template <int D, int TR>
PDS_DEV void flush_tile(const float *tile, float *gdst, int rows, int lane) {
  if (rows == TR) {
    constexpr int NV = TR * D / 4;  // float4 count (D is even, 32*D divisible by 4)
    const float4 *src = reinterpret_cast<const float4 *>(tile);
    float4 *dst = reinterpret_cast<float4 *>(gdst);
#pragma unroll
    for (int it = 0; it < (NV + kWave - 1) / kWave; ++it) {
      const int idx = it * kWave + lane;
      if (idx < NV) nt_store4(dst + idx, src[idx]);
    }
  } else {
    const int n = rows * D;
    for (int idx = lane; idx < n; idx += kWave) gdst[idx] = tile[idx];
  }
}

// Inputs of one env-step, loaded 16 B/lane.
struct Loaded {
  float4 act, q0, q1, q2, h1, h2, mx, p0, mA, mK, ou, nz0, oh0, oh1, pid0, pid2;
  float2 p1, nz1, oh2, pid1, pid3;
  uint32_t ctr;
};

template <class V>
PDS_DEV void load_env(const StepArgs &a, long long ii, Loaded &L) {
  L.act = nt_load4(a.actions + ii);  // read once per step: keep it out of the caches
  L.q0 = st_load4(a.st.s0 + ii);
  L.q1 = st_load4(a.st.s1 + ii);
  L.q2 = st_load4(a.st.s2 + ii);
  L.h1 = st_load4(a.st.hist[a.parity] + ii);      // u(k-1)
  L.h2 = st_load4(a.st.hist[a.parity ^ 1] + ii);  // u(k-2)
  L.ctr = a.st.ctr[ii];
  if (V::MOTOR) L.mx = a.st.mx[ii];
  if (V::DR) {
    L.p0 = a.st.par0[ii];
    L.p1 = a.st.par1[ii];
    if (V::MOTOR) { L.mA = a.st.mA[ii]; L.mK = a.st.mK[ii]; }
  }
  if (V::TN) L.ou = a.st.ou[ii];
  if (V::CTRL >= 1) { L.pid0 = a.st.pid0[ii]; L.pid1 = a.st.pid1[ii]; }
  if (V::CTRL == 2) { L.pid2 = a.st.pid2[ii]; L.pid3 = a.st.pid3[ii]; }
  if (V::ON) {
    L.nz0 = a.st.nz0[ii]; L.nz1 = a.st.nz1[ii];
    L.oh0 = a.st.oh0[ii]; L.oh1 = a.st.oh1[ii]; L.oh2 = a.st.oh2[ii];
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

// Standard variates of one physics sub-step: OUNoise.noise (4 z) and the gyro part of the
// add_noise call whose observation is discarded (envs/base.py:464): 9 z.
struct SubNoise {
  float ou[4], bias_z[3], rw_z[3], to_z[3];
};

template <class V>
PDS_DEV void sub_noise(const StepArgs &a, uint32_t env_id, long long ii, int sub, SubNoise &n) {
  if (a.noise != nullptr) {  // injected (parity tests; aggregate_phy_steps == 1)
    const float *p = a.noise + ii * PDS_NOISE_FLOATS;
#pragma unroll
    for (int j = 0; j < 4; ++j) n.ou[j] = p[PDS_N_OU + j];
#pragma unroll
    for (int j = 0; j < 3; ++j) { n.bias_z[j] = p[PDS_N_A_BIAS + j]; n.rw_z[j] = p[PDS_N_A_RW + j]; n.to_z[j] = p[PDS_N_A_TO + j]; }
    return;
  }
  // words 0,1 -> OU z[0..3]; words 2..6 -> bias, random walk, turn-on z[4..12] (one pair per word)
  const uint32_t b0 = kBlkSubNoise + 2u * (uint32_t)sub;
  float z[14];
  const U4 r = philox4x32_7(env_id, a.tick_lo, a.tick_hi, b0, a.seed_lo, a.seed_hi);
  box_muller_word(r.x, z[0], z[1]);
  box_muller_word(r.y, z[2], z[3]);
  if (V::ON) {
    const U4 r1 = philox4x32_7(env_id, a.tick_lo, a.tick_hi, b0 + 1u, a.seed_lo, a.seed_hi);
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
}

// One 64-env tile per wave, one launch covers all tiles (the hardware dispatcher balances blocks
// whose deferred-reset drains differ in length).  A persistent grid-stride variant (256 x 3 resident
// blocks, register prefetch of the next tile) measured 66.7 us vs 62.7 us for this shape, doubled the
// input registers, and -- because the compiler hoists every loop-invariant kernel argument out of
// the tile loop -- pushed the kernel over the SGPR budget (137 v_writelane/v_readlane spill
// instructions in the main path), so there is no tile loop here.
// __launch_bounds__(256, 3): LDS admits 3 blocks (12 waves) per CU, so cap VGPRs at 168.
template <class V, int TR>
// The lean variants are additionally held to 128 VGPRs (min 4 waves/SIMD): measured 2-3 % faster
// (62.7 vs 64.3 us Hover, 63.5 vs 64.7 us TakeOff+GE on the same box); the observation-noise
// variants spill badly under that cap (154 vs 106 us) and keep 168.
#ifndef PDS_MIN_WAVES_LEAN
#define PDS_MIN_WAVES_LEAN 4
#endif
#ifndef PDS_MIN_WAVES
#define PDS_MIN_WAVES (V::ON ? 3 : PDS_MIN_WAVES_LEAN)
#endif
__global__ __launch_bounds__(kBlock, PDS_MIN_WAVES) void step_kernel(const StepArgs a) {
  constexpr int TASK = V::TASK;
  constexpr int D = V::D;
  constexpr int O = V::O;
  // Auto-reset in registers before the stores, or deferred drain after them (see below).  Measured
  // on one MI355X box, merged vs deferred: Hover 2^20 58.45 vs 60.0 us, Hover 2^21 108.5 vs 111.0 us;
  // but Circle + PT1 + DR at 2^20 82.9 vs 78.6 us (181 VGPRs => 2 waves/SIMD), TakeOff (resets only by
  // the 500-step truncation) 61.4 vs 60.7 us, and under the half tile's 128-VGPR cap it spills
  // (Circle 262 144: 39 vs 21 us) -- so those keep the deferred drain.
  constexpr bool MERGED = PDS_MERGED_RESET && !V::ON && TR == kWave && !(V::MOTOR && V::DR) && TASK != PDS_TASK_TAKEOFF;
  __shared__ __attribute__((aligned(16))) float tile_all[(kBlock / kWave) * TR * D];
  __shared__ float2 ref_lds[(TASK == PDS_TASK_CIRCLE) ? kRefPoints : 1];
  __shared__ uint32_t queue_all[(kBlock / kWave) * kQueueCap];
  const Consts &k = a.k;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;
  if (TASK == PDS_TASK_CIRCLE) {
    for (int t = tid; t < kRefPoints; t += kBlock) ref_lds[t] = a.st.circle_ref[t];
    __syncthreads();
  }
  uint32_t *queue = queue_all + wave * kQueueCap;
  int qcount = 0;  // wave-uniform
  float *tile = tile_all + wave * (TR * D);
  // full tile: the row is built in place in LDS; half tile: in registers, staged pass by pass
  float rowbuf[(TR == kWave) ? 1 : D];
  float *row = (TR == kWave) ? tile + lane * D : rowbuf;
  const long long ntiles = (a.n + kWave - 1) / kWave;
  const long long t = (long long)blockIdx.x * (kBlock / kWave) + wave;
  if (t >= ntiles) return;  // wave-uniform

  // The loads are issued before anything else so that the scalar preamble of the kernel
  // (kernel-argument loads, uniform constants) overlaps with their latency.
  const long long wave_base = t * kWave;
  const long long i = wave_base + lane;
  const bool active = i < a.n;
  const long long ii = active ? i : (a.n - 1);  // tail lanes recompute the last env, stores masked
  Loaded cur;
  load_env<V>(a, ii, cur);
  __builtin_amdgcn_sched_barrier(0);
  {
    const uint32_t env_id = (uint32_t)(a.env_id_base + (unsigned long long)ii);

    const float4 act = cur.act, h1 = cur.h1, h2 = cur.h2;
    const uint32_t ctr = cur.ctr;
    float4 mx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (V::MOTOR) mx = cur.mx;
    Params par;
    default_params(k, par);
    if (V::DR) {
      par.dt = cur.p0.x; par.m = cur.p0.y; par.Jx = cur.p0.z; par.Jy = cur.p0.w; par.Jz = cur.p1.x; par.ftf1 = cur.p1.y;
      if (V::MOTOR) {
        par.A[0] = cur.mA.x; par.A[1] = cur.mA.y; par.A[2] = cur.mA.z; par.A[3] = cur.mA.w;
        par.K[0] = cur.mK.x; par.K[1] = cur.mK.y; par.K[2] = cur.mK.z; par.K[3] = cur.mK.w;
      }
    }
    NoiseState ns;
#pragma unroll
    for (int j = 0; j < 4; ++j) ns.ou[j] = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) { ns.bias[j] = 0.f; ns.lpf[j] = 0.f; }
    if (V::TN) { ns.ou[0] = cur.ou.x; ns.ou[1] = cur.ou.y; ns.ou[2] = cur.ou.z; ns.ou[3] = cur.ou.w; }
    if (V::ON) {
      ns.bias[0] = cur.nz0.x; ns.bias[1] = cur.nz0.y; ns.bias[2] = cur.nz0.z;
      ns.lpf[0] = cur.nz0.w; ns.lpf[1] = cur.nz1.x; ns.lpf[2] = cur.nz1.y;
    }
    PidState ps;
#pragma unroll
    for (int j = 0; j < 3; ++j) { ps.rate_int[j] = ps.rate_err[j] = ps.att_int[j] = ps.att_err[j] = 0.f; }
    if (V::CTRL >= 1) {
      ps.rate_int[0] = cur.pid0.x; ps.rate_int[1] = cur.pid0.y; ps.rate_int[2] = cur.pid0.z;
      ps.rate_err[0] = cur.pid0.w; ps.rate_err[1] = cur.pid1.x; ps.rate_err[2] = cur.pid1.y;
    }
    if (V::CTRL == 2) {
      ps.att_int[0] = cur.pid2.x; ps.att_int[1] = cur.pid2.y; ps.att_int[2] = cur.pid2.z;
      ps.att_err[0] = cur.pid2.w; ps.att_err[1] = cur.pid3.x; ps.att_err[2] = cur.pid3.y;
    }
    EnvRegs e{cur.q0.x, cur.q0.y, cur.q0.z, cur.q0.w, cur.q1.x, cur.q1.y, cur.q1.z, cur.q1.w,
              cur.q2.x, cur.q2.y, cur.q2.z, cur.q2.w};
    const int step = (int)ctr_step(ctr);
    const int ref_offset = (int)ctr_off(ctr);

    // ---- o(k): first half of the row ------------------------------------------------------------
    // noise-free: rebuilt from the pre-step state instead of being re-read from HBM;
    // noisy: the stored noisy observation + the filtered gyro (== the low-pass state)
    Quat q = quat_from_euler(e.roll, e.pitch, e.yaw);
    {
      float tx, ty, tz;
      target_at<TASK>(k, ref_lds, target_index<TASK>(step, k.agg, ref_offset), tx, ty, tz);
      if (V::ON) {
        const NoisyObs ok{cur.oh0.x, cur.oh0.y, cur.oh0.z, cur.oh0.w, cur.oh1.x, cur.oh1.y, cur.oh1.z,
                          cur.oh1.w, cur.oh2.x, cur.oh2.y};
        write_noisy_half<TASK>(row, ok, ns.lpf, h1, tx, ty, tz, h2);
      } else {
        Quat qk = q;
        if (ctr_sign(ctr)) { qk.x = -q.x; qk.y = -q.y; qk.z = -q.z; qk.w = -q.w; }
        write_obs_half<TASK>(row, e, qk, h1, tx, ty, tz, h2);
      }
    }

    // ---- aggregate_phy_steps x SimplePhysics.step_forward (envs/base.py:457-465) ---------------
    float xm[4] = {mx.x, mx.y, mx.z, mx.w};
    const float av[4] = {act.x, act.y, act.z, act.w};
    // divisions by the per-env mass / inertia become multiplications by v_rcp_f32 results (1 ulp)
    const float inv_m = fast_rcp(par.m), inv_Jx = fast_rcp(par.Jx), inv_Jy = fast_rcp(par.Jy), inv_Jz = fast_rcp(par.Jz);
    for (int sub = 0; sub < k.agg; ++sub) {
      SubNoise sn;
      if (V::TN || V::ON) sub_noise<V>(a, env_id, ii, sub, sn);
      // CrazyFlieAgent.apply_action, envs/agents.py:259-298 (+ PWM.act envs/control.py:94-100)
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
      q = quat_from_euler(e.roll, e.pitch, e.yaw);                                         // :179
      e.pz = fmaxf(e.pz, 0.f);                                                             // :182
      // envs/base.py:464: compute_observation() whose result is dropped still advances the gyro
      // bias random walk and the low-pass filter
      if (V::ON) gyro_update(k, e, sn.bias_z, sn.rw_z, sn.to_z, ns);
    }

    // ---- task: target, done, reward, cost (all on the TRUE state) -------------------------------
    float tx, ty, tz;
    target_at<TASK>(k, ref_lds, target_index<TASK>(step + 1, k.agg, ref_offset), tx, ty, tz);
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
    NoisyObs on_new;
    if (V::ON) {
      ObsNoise n;
      if (a.noise != nullptr) obs_noise_load(a.noise + ii * PDS_NOISE_FLOATS + PDS_N_OBS, n);
      else obs_noise_philox(env_id, a, kBlkObsNoise, n);
      sensor_observe(k, e, n, ns, on_new);
      write_noisy_half<TASK>(row + O + 4, on_new, ns.lpf, act, tx, ty, tz, h1);
    } else {
      write_obs_half<TASK>(row + O + 4, e, q, act, tx, ty, tz, h1);
    }

    uint32_t ctr_new = ctr_pack((uint32_t)(step + 1), 0u, (uint32_t)ref_offset);
    float4 hist_new = act;  // -> hist[parity ^ 1]: overwrites u(k-2); next step's parity makes it u(k-1)
    // ---- auto-reset.  ~2 % of the envs finish per step under random actions, i.e. 3 of 4 waves
    // hold one or two finished envs.  Their last observation goes to final_obs (below, out of the
    // LDS tile); the reset itself is done densely, 8 lanes per finished env:
    //  * without observation noise: now, in registers (reset_in_registers), so the fresh state and
    //    observation leave through the wave's ordinary coalesced stores;
    //  * with observation noise (the reset needs much more state): deferred to a drain after the
    //    stores (drain_reset_queue).
    const bool need_reset = a.auto_reset && (done || trunc) && active;
    const unsigned long long reset_mask = __ballot(need_reset);  // wave-uniform
    const unsigned long long done_mask = (a.final_obs != nullptr) ? reset_mask : 0ull;  // -> final_obs
    if (reset_mask != 0ull) {  // wave-uniform
      const int pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(reset_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)reset_mask, 0u));
      qcount = __popcll(reset_mask);
      if (!MERGED) {
        if (need_reset) queue[pos] = (uint32_t)lane | ((uint32_t)ref_offset << 6);
      } else {
        if (need_reset) queue[pos] = (uint32_t)lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float4 u0 = act, mxr = make_float4(xm[0], xm[1], xm[2], xm[3]);
        if constexpr (MERGED)
          reset_in_registers<V>(a, ref_lds, queue, qcount, need_reset, pos, lane, wave_base, ref_offset, e, q, u0,
                                mxr, par, ctr_new);
        if (need_reset) {
          hist_new = u0;
          xm[0] = mxr.x; xm[1] = mxr.y; xm[2] = mxr.z; xm[3] = mxr.w;
#pragma unroll
          for (int j = 0; j < 3; ++j) { ps.rate_int[j] = ps.rate_err[j] = ps.att_int[j] = ps.att_err[j] = 0.f; }
          // written only by resets (no write-after-write with this step's stores)
          a.st.hist[a.parity][i] = u0;
          if (V::DR) {
            a.st.par0[i] = make_float4(par.dt, par.m, par.Jx, par.Jy);
            a.st.par1[i] = make_float2(par.Jz, par.ftf1);
            if (V::MOTOR) {
              a.st.mA[i] = make_float4(par.A[0], par.A[1], par.A[2], par.A[3]);
              a.st.mK[i] = make_float4(par.K[0], par.K[1], par.K[2], par.K[3]);
            }
          }
        }
      }
    }

    // ---- coalesced stores ----------------------------------------------------------------------
    if (active) {
      st_store4(a.st.s0 + i, make_float4(e.px, e.py, e.pz, e.vx));
      st_store4(a.st.s1 + i, make_float4(e.vy, e.vz, e.roll, e.pitch));
      st_store4(a.st.s2 + i, make_float4(e.yaw, e.wx, e.wy, e.wz));
      st_store4(a.st.hist[a.parity ^ 1] + i, hist_new);
      a.st.ctr[i] = ctr_new;
      if (V::MOTOR) a.st.mx[i] = make_float4(xm[0], xm[1], xm[2], xm[3]);
      if (V::TN) a.st.ou[i] = make_float4(ns.ou[0], ns.ou[1], ns.ou[2], ns.ou[3]);
      if (V::CTRL >= 1) {
        a.st.pid0[i] = make_float4(ps.rate_int[0], ps.rate_int[1], ps.rate_int[2], ps.rate_err[0]);
        a.st.pid1[i] = make_float2(ps.rate_err[1], ps.rate_err[2]);
      }
      if (V::CTRL == 2) {
        a.st.pid2[i] = make_float4(ps.att_int[0], ps.att_int[1], ps.att_int[2], ps.att_err[0]);
        a.st.pid3[i] = make_float2(ps.att_err[1], ps.att_err[2]);
      }
      if (V::ON) {
        a.st.nz0[i] = make_float4(ns.bias[0], ns.bias[1], ns.bias[2], ns.lpf[0]);
        a.st.nz1[i] = make_float2(ns.lpf[1], ns.lpf[2]);
        a.st.oh0[i] = make_float4(on_new.x, on_new.y, on_new.z, on_new.qx);
        a.st.oh1[i] = make_float4(on_new.qy, on_new.qz, on_new.qw, on_new.vx);
        a.st.oh2[i] = make_float2(on_new.vy, on_new.vz);
      }
      nt_store(a.reward + i, reward);
      nt_store(a.cost + i, cost);
      nt_store(a.term + i, (uint8_t)(done ? 1 : 0));
      nt_store(a.trunc + i, (uint8_t)(trunc ? 1 : 0));
    }
    // LDS rows of this wave were written by its own lanes only: wave-synchronous, no block barrier
    const long long rem = a.n - wave_base;
#pragma unroll
    for (int pass = 0; pass < kWave / TR; ++pass) {
      if (TR != kWave) {
        if ((lane / TR) == pass) {
          float *dst = tile + (lane % TR) * D;
#pragma unroll
          for (int j = 0; j < D; ++j) dst[j] = rowbuf[j];
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
        if (lane < D) nt_store(a.final_obs + (wave_base + src_lane) * D + lane, tile[(src_lane % TR) * D + lane]);
      }
      if (MERGED && reset_mask != 0ull) {  // wave-uniform: the reset envs' rows become [o0, u0, o0, u0]
        if (need_reset && (TR == kWave || (lane / TR) == pass)) {
          float tx0, ty0, tz0;
          target_at<TASK>(k, ref_lds, target_index<TASK>(0, k.agg, (int)ctr_off(ctr_new)), tx0, ty0, tz0);
          float *dst = tile + (lane % TR) * D;
          write_obs_half<TASK>(dst, e, q, hist_new, tx0, ty0, tz0, hist_new);
          write_obs_half<TASK>(dst + O + 4, e, q, hist_new, tx0, ty0, tz0, hist_new);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      const long long left = rem - pass * TR;
      if (left > 0)
        flush_tile<D, TR>(tile, a.obs + (wave_base + pass * TR) * D, left >= TR ? TR : (int)left, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // the tile is rewritten by the next pass / the reset drain
    }
  }
  if (!MERGED && qcount > 0) drain_reset_queue<V>(a, ref_lds, queue, qcount, lane, wave_base, tile);
}

// ---- per-task instantiation (one translation unit per task keeps the build parallel) -----------
#define PDS_DISPATCH5(FN, TASK, f, ...)                                                              \
  do {                                                                                                \
    const int key_ = (f.motor ? 16 : 0) | (f.dr ? 8 : 0) | (f.ge ? 4 : 0) | (f.tn ? 2 : 0) | (f.on ? 1 : 0); \
    switch (key_) {                                                                                   \
      PDS_CASES16(FN, TASK, 0, false, __VA_ARGS__)                                                    \
      PDS_CASES16(FN, TASK, 16, true, __VA_ARGS__)                                                    \
    }                                                                                                 \
  } while (0)
#define PDS_CASES16(FN, TASK, base, M, ...)                                                          \
  PDS_CASES8(FN, TASK, base, M, false, __VA_ARGS__) PDS_CASES8(FN, TASK, base + 8, M, true, __VA_ARGS__)
#define PDS_CASES8(FN, TASK, base, M, R, ...)                                                        \
  PDS_CASES4(FN, TASK, base, M, R, false, __VA_ARGS__) PDS_CASES4(FN, TASK, base + 4, M, R, true, __VA_ARGS__)
#define PDS_CASES4(FN, TASK, base, M, R, G, ...)                                                     \
  case base + 0: FN((Variant<TASK, M, R, G, false, false>), __VA_ARGS__); break;                     \
  case base + 1: FN((Variant<TASK, M, R, G, false, true>), __VA_ARGS__); break;                      \
  case base + 2: FN((Variant<TASK, M, R, G, true, false>), __VA_ARGS__); break;                      \
  case base + 3: FN((Variant<TASK, M, R, G, true, true>), __VA_ARGS__); break;

#define PDS_UNPAREN(...) __VA_ARGS__
template <class V>
inline void launch_step_variant(bool half_tile, dim3 grid, hipStream_t s, const StepArgs &a) {
  if constexpr (!V::ON) {
    if (half_tile) {
      hipLaunchKernelGGL((step_kernel<V, kHalfTileRows>), grid, dim3(kBlock), 0, s, a);
      return;
    }
  }
  hipLaunchKernelGGL((step_kernel<V, kWave>), grid, dim3(kBlock), 0, s, a);
}
#define PDS_LAUNCH_STEP(V, grid, s, a) launch_step_variant<PDS_UNPAREN V>(f.half_tile, grid, s, a)
#define PDS_LAUNCH_RESET(V, grid, s, a) hipLaunchKernelGGL((reset_kernel<PDS_UNPAREN V>), grid, dim3(kBlock), 0, s, a)

// PID control modes: 16 variants each (motor x DR x thrust noise x observation noise), no ground effect
#define PDS_PID_CASES(TASK, C, grid, s, a)                                                            \
  switch ((f.motor ? 8 : 0) | (f.dr ? 4 : 0) | (f.tn ? 2 : 0) | (f.on ? 1 : 0)) {                    \
    case 0: PDS_LAUNCH_STEP((Variant<TASK, false, false, false, false, false, C>), grid, s, a); break; \
    case 1: PDS_LAUNCH_STEP((Variant<TASK, false, false, false, false, true, C>), grid, s, a); break;  \
    case 2: PDS_LAUNCH_STEP((Variant<TASK, false, false, false, true, false, C>), grid, s, a); break;  \
    case 3: PDS_LAUNCH_STEP((Variant<TASK, false, false, false, true, true, C>), grid, s, a); break;   \
    case 4: PDS_LAUNCH_STEP((Variant<TASK, false, true, false, false, false, C>), grid, s, a); break;  \
    case 5: PDS_LAUNCH_STEP((Variant<TASK, false, true, false, false, true, C>), grid, s, a); break;   \
    case 6: PDS_LAUNCH_STEP((Variant<TASK, false, true, false, true, false, C>), grid, s, a); break;   \
    case 7: PDS_LAUNCH_STEP((Variant<TASK, false, true, false, true, true, C>), grid, s, a); break;    \
    case 8: PDS_LAUNCH_STEP((Variant<TASK, true, false, false, false, false, C>), grid, s, a); break;  \
    case 9: PDS_LAUNCH_STEP((Variant<TASK, true, false, false, false, true, C>), grid, s, a); break;   \
    case 10: PDS_LAUNCH_STEP((Variant<TASK, true, false, false, true, false, C>), grid, s, a); break;  \
    case 11: PDS_LAUNCH_STEP((Variant<TASK, true, false, false, true, true, C>), grid, s, a); break;   \
    case 12: PDS_LAUNCH_STEP((Variant<TASK, true, true, false, false, false, C>), grid, s, a); break;  \
    case 13: PDS_LAUNCH_STEP((Variant<TASK, true, true, false, false, true, C>), grid, s, a); break;   \
    case 14: PDS_LAUNCH_STEP((Variant<TASK, true, true, false, true, false, C>), grid, s, a); break;   \
    default: PDS_LAUNCH_STEP((Variant<TASK, true, true, false, true, true, C>), grid, s, a); break;    \
  }

#define PDS_DEFINE_TASK_LAUNCHERS(NAME, TASK)                                                        \
  void launch_step_##NAME(const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a) {       \
    if (f.ctrl == 0) {                                                                                \
      PDS_DISPATCH5(PDS_LAUNCH_STEP, TASK, f, grid, s, a);                                           \
    } else if (TASK != PDS_TASK_TAKEOFF) { /* TakeOff fixes control_mode='PWM', envs/takeoff.py:225 */ \
      if (f.ctrl == 1) { PDS_PID_CASES(TASK, 1, grid, s, a) } else { PDS_PID_CASES(TASK, 2, grid, s, a) } \
    }                                                                                                 \
  }                                                                                                   \
  void launch_reset_##NAME(const LaunchFlags &f, dim3 grid, hipStream_t s, const StepArgs &a) {      \
    /* the reset code does not depend on the GE / TN flags: 8 variants */                            \
    switch ((f.motor ? 4 : 0) | (f.dr ? 2 : 0) | (f.on ? 1 : 0)) {                                    \
      case 0: PDS_LAUNCH_RESET((Variant<TASK, false, false, false, false, false>), grid, s, a); break; \
      case 1: PDS_LAUNCH_RESET((Variant<TASK, false, false, false, false, true>), grid, s, a); break;  \
      case 2: PDS_LAUNCH_RESET((Variant<TASK, false, true, false, false, false>), grid, s, a); break;  \
      case 3: PDS_LAUNCH_RESET((Variant<TASK, false, true, false, false, true>), grid, s, a); break;   \
      case 4: PDS_LAUNCH_RESET((Variant<TASK, true, false, false, false, false>), grid, s, a); break;  \
      case 5: PDS_LAUNCH_RESET((Variant<TASK, true, false, false, false, true>), grid, s, a); break;   \
      case 6: PDS_LAUNCH_RESET((Variant<TASK, true, true, false, false, false>), grid, s, a); break;   \
      default: PDS_LAUNCH_RESET((Variant<TASK, true, true, false, false, true>), grid, s, a); break;   \
    }                                                                                                 \
  }

}  // namespace pds
