// pds_kernels.hip -- fused lockstep CrazyFlie SimplePhysics step for MI355X (gfx950, wave64).
//
// One thread per environment.  Per step a thread streams its SoA state quads (16 B/lane, fully
// coalesced), advances PWM->thrust, Newton-Euler force/torque, semi-implicit Euler and
// Euler->quaternion in registers, evaluates the task's reward / cost / termination, optionally
// resets the env from a counter-based Philox stream, and stages its observation row in a
// per-wave LDS tile so that the row-major [N, D] observation tensor is written with contiguous
// 1 KiB wave stores instead of 64 strided rows.  HBM-bound by design: no MFMA (there is no dense
// contraction on this path).
//
// Reference (paths relative to phoenix_drone_simulation/): envs/physics.py:130-200,
// envs/agents.py:259-298, envs/control.py:94-100, envs/base.py:239-319,382-475, envs/hover.py,
// envs/circle.py, envs/takeoff.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/pds.h"
#include "pds_device.h"

namespace pds {

constexpr int kBlock = 256;  // 4 waves; each wave owns a private LDS tile (no block barrier needed)
constexpr int kWave = 64;
constexpr int kQueueCap = 128;   // deferred-reset queue entries per wave (LDS)
constexpr int kBlocksPerCU = 3;  // 160 KiB LDS / (256 x 48 x 4 B + circle table)
constexpr int kRefPoints = 300;  // envs/circle.py:48, envs/takeoff.py:43

// ---- packing of the per-env counter word --------------------------------------------------------
// bits 0..15 env.step calls since reset | bit 16 quaternion == -Q(rpy) | bits 17..25 Circle ref_offset
PDS_DEV uint32_t ctr_pack(uint32_t step, uint32_t sign, uint32_t off) { return step | (sign << 16) | (off << 17); }
PDS_DEV uint32_t ctr_step(uint32_t c) { return c & 0xFFFFu; }
PDS_DEV uint32_t ctr_sign(uint32_t c) { return (c >> 16) & 1u; }
PDS_DEV uint32_t ctr_off(uint32_t c) { return (c >> 17) & 0x1FFu; }

// ---- SoA state in HBM (one float4 "quad" per env and array => 16 B/lane coalesced streams) ------
struct DevState {
  float4 *s0;      // px py pz vx
  float4 *s1;      // vy vz roll pitch
  float4 *s2;      // yaw wx wy wz
  float4 *hist[2]; // action ring: slot `parity` = u(k-1), slot `parity^1` = u(k-2)
  uint32_t *ctr;
  float4 *mx;      // motor state x[4]                (use_motor_dynamics)
  float4 *par0;    // dt m Jxx Jyy                    (domain_randomization)
  float2 *par1;    // Jzz ftf1                        (domain_randomization)
  float4 *mA;      // per-motor A (B = 1-A)           (domain_randomization & motor)
  float4 *mK;      // per-motor K                     (domain_randomization & motor)
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
  int agg, max_steps, reset_dist;
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
  long long n;
  unsigned long long env_id_base;
  uint32_t seed_lo, seed_hi, tick_lo, tick_hi;
  int parity;
  int auto_reset;
};

template <int TASK, bool NOISY>
struct ObsLayout {
  static constexpr int O = NOISY ? (TASK == PDS_TASK_HOVER ? 13 : (TASK == PDS_TASK_CIRCLE ? 16 : 20))
                                 : (TASK == PDS_TASK_HOVER ? 17 : (TASK == PDS_TASK_CIRCLE ? 16 : 20));
  static constexpr int D = 2 * (O + 4);
};

struct EnvRegs {
  float px, py, pz, vx, vy, vz, roll, pitch, yaw, wx, wy, wz;
};

struct Params {  // per-env physical parameters (constants unless domain randomisation is on)
  float dt, m, Jx, Jy, Jz, ftf1;
  float A[4], K[4];
};

struct Sample {  // one reset() worth of draws, reference order (see include/pds.h PDS_S_*)
  float pos[3], rpy[3], vel[3], w[3], mx[4], act[4];
  float dt, m, J[3], ftf1, T[4], t2w[4];
  int ref_offset;
};

// In-kernel reset sampler; restated draw for draw by oracle/phoenix_oracle.c
// po_philox_reset_sample.  Ranges: envs/hover.py:201-228, envs/circle.py:225-257,
// envs/takeoff.py:186-191, envs/base.py:250-287.
template <int TASK, bool MOTOR, bool DR>
PDS_DEV void sample_philox(const Consts &k, uint32_t env_id, const StepArgs &a, Sample &s) {
  constexpr float D2R = kPi / 180.f;
  float pos_lim, rp_lim, yaw_lim, vel_lim, w_lim, wz_lim;
  if (TASK == PDS_TASK_HOVER) {
    pos_lim = 0.25f; rp_lim = kPi / 6.f; yaw_lim = 2.f * kPi; vel_lim = 0.1f; w_lim = 200.f * D2R; wz_lim = 20.f * D2R;
  } else if (TASK == PDS_TASK_CIRCLE) {
    pos_lim = 0.05f; rp_lim = 20.f * D2R; yaw_lim = 0.1f * kPi; vel_lim = 0.1f; w_lim = 50.f * D2R; wz_lim = 20.f * D2R;
  } else {
    pos_lim = 0.25f; rp_lim = 0.f; yaw_lim = kPi; vel_lim = 0.f; w_lim = 0.f; wz_lim = 0.f;
  }
  const U4 r0 = philox4x32_10(env_id, a.tick_lo, a.tick_hi, 0u, a.seed_lo, a.seed_hi);
  const U4 r1 = philox4x32_10(env_id, a.tick_lo, a.tick_hi, 1u, a.seed_lo, a.seed_hi);
  s.pos[0] = urange(r0.x, -pos_lim, pos_lim);
  s.pos[1] = urange(r0.y, -pos_lim, pos_lim);
  s.pos[2] = (TASK == PDS_TASK_TAKEOFF) ? 0.f : urange(r0.z, -pos_lim, pos_lim);
  s.rpy[0] = urange(r0.w, -rp_lim, rp_lim);
  s.rpy[1] = urange(r1.x, -rp_lim, rp_lim);
  s.rpy[2] = urange(r1.y, -yaw_lim, yaw_lim);
  s.vel[0] = urange(r1.z, -vel_lim, vel_lim);
  s.vel[1] = urange(r1.w, -vel_lim, vel_lim);
  if (TASK != PDS_TASK_TAKEOFF) {
    const U4 r2 = philox4x32_10(env_id, a.tick_lo, a.tick_hi, 2u, a.seed_lo, a.seed_hi);
    const U4 r3 = philox4x32_10(env_id, a.tick_lo, a.tick_hi, 3u, a.seed_lo, a.seed_hi);
    const U4 r4 = philox4x32_10(env_id, a.tick_lo, a.tick_hi, 4u, a.seed_lo, a.seed_hi);
    s.vel[2] = urange(r2.x, -vel_lim, vel_lim);
    s.w[0] = urange(r2.y, -w_lim, w_lim);
    s.w[1] = urange(r2.z, -w_lim, w_lim);
    s.w[2] = urange(r2.w, -wz_lim, wz_lim);
    float z[8];
    box_muller(r3.x, r3.y, z[0], z[1]);
    box_muller(r3.z, r3.w, z[2], z[3]);
    box_muller(r4.x, r4.y, z[4], z[5]);
    box_muller(r4.z, r4.w, z[6], z[7]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s.mx[i] = k.hover_x + 0.02f * z[i];
      s.act[i] = k.hover_action + 0.02f * z[4 + i];
    }
  } else {
    s.vel[2] = 0.f; s.w[0] = s.w[1] = s.w[2] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { s.mx[i] = 0.f; s.act[i] = 0.f; }
  }
  s.ref_offset = 0;
  if (DR || TASK == PDS_TASK_CIRCLE) {
    const float f = k.dr;
    const U4 r5 = philox4x32_10(env_id, a.tick_lo, a.tick_hi, 5u, a.seed_lo, a.seed_hi);
    const U4 r6 = philox4x32_10(env_id, a.tick_lo, a.tick_hi, 6u, a.seed_lo, a.seed_hi);
#define PDS_DRV(x, d) urange((x), (d) - f * (d), (d) + f * (d))
    s.dt = PDS_DRV(r5.x, k.dt_nom);
    s.m = PDS_DRV(r5.y, k.M_nom);
    s.J[0] = PDS_DRV(r5.z, k.Jx_nom);
    s.J[1] = PDS_DRV(r5.w, k.Jy_nom);
    s.J[2] = PDS_DRV(r6.x, k.Jz_nom);
    s.ftf1 = PDS_DRV(r6.z, k.ftf1_nom);
    s.ref_offset = (int)__umulhi(r6.w, 300u);
    if (MOTOR && DR) {
      const U4 r7 = philox4x32_10(env_id, a.tick_lo, a.tick_hi, 7u, a.seed_lo, a.seed_hi);
      const U4 r8 = philox4x32_10(env_id, a.tick_lo, a.tick_hi, 8u, a.seed_lo, a.seed_hi);
      s.T[0] = PDS_DRV(r7.x, k.mtc); s.T[1] = PDS_DRV(r7.y, k.mtc);
      s.T[2] = PDS_DRV(r7.z, k.mtc); s.T[3] = PDS_DRV(r7.w, k.mtc);
      s.t2w[0] = PDS_DRV(r8.x, k.t2w); s.t2w[1] = PDS_DRV(r8.y, k.t2w);
      s.t2w[2] = PDS_DRV(r8.z, k.t2w); s.t2w[3] = PDS_DRV(r8.w, k.t2w);
    }
#undef PDS_DRV
  }
}

PDS_DEV void sample_load(const float *row, Sample &s) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    s.pos[i] = row[PDS_S_POS_OFFSET + i]; s.rpy[i] = row[PDS_S_RPY + i];
    s.vel[i] = row[PDS_S_VEL + i]; s.w[i] = row[PDS_S_OMEGA + i]; s.J[i] = row[PDS_S_DR_J + i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s.mx[i] = row[PDS_S_MOTOR_X + i]; s.act[i] = row[PDS_S_ACTION + i];
    s.T[i] = row[PDS_S_DR_T + i]; s.t2w[i] = row[PDS_S_DR_T2W + i];
  }
  s.dt = row[PDS_S_DR_DT]; s.m = row[PDS_S_DR_M]; s.ftf1 = row[PDS_S_DR_FTF1];
  s.ref_offset = (int)row[PDS_S_REF_OFFSET];
}

PDS_DEV void default_params(const Consts &k, Params &p) {
  p.dt = k.dt; p.m = k.m; p.Jx = k.Jx; p.Jy = k.Jy; p.Jz = k.Jz; p.ftf1 = k.ftf1;
#pragma unroll
  for (int i = 0; i < 4; ++i) { p.A[i] = k.A; p.K[i] = k.K; }
}

// Reference trajectories: envs/circle.py:45-56 (table in HBM/L2, staged per block in LDS),
// envs/takeoff.py:43-47 (z = k/300).
template <int TASK>
PDS_DEV void target_at(const Consts &k, const float2 *ref_lds, int t, float &tx, float &ty, float &tz) {
  if (TASK == PDS_TASK_CIRCLE) {
    const float2 r = ref_lds[t];
    tx = r.x; ty = r.y; tz = 1.0f;
  } else if (TASK == PDS_TASK_TAKEOFF) {
    tx = 0.f; ty = 0.f; tz = (float)t / 300.0f;
  } else {
    tx = k.target[0]; ty = k.target[1]; tz = k.target[2];
  }
}

// DroneBaseEnv.reset (envs/base.py:382-431) for one env: task_specific_reset, domain
// randomisation, the Bullet pose/velocity round trip of update_information
// (envs/agents.py:434-453: rpy = Euler(quat), omega = R^T R^T omega_sampled).
template <int TASK, bool MOTOR, bool DR>
PDS_DEV void reset_env(const Consts &k, const float2 *ref_lds, const Sample &s, EnvRegs &e, Quat &q,
                       float4 &u0, float4 &mx, Params &par, uint32_t &ctr) {
  float px = k.init_xyz[0], py = k.init_xyz[1], pz = k.init_xyz[2];
  float vx = k.init_vel[0], vy = k.init_vel[1], vz = k.init_vel[2];
  float w0 = k.init_w[0], w1 = k.init_w[1], w2 = k.init_w[2];
  float r0 = k.init_rpy[0], r1 = k.init_rpy[1], r2 = k.init_rpy[2];
  int ref_offset = (TASK == PDS_TASK_CIRCLE) ? (int)ctr_off(ctr) : 0;  // kept when no reset distribution
  u0 = make_float4(0.f, 0.f, 0.f, 0.f);  // drone.reset(): envs/agents.py:380-386
  mx = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k.reset_dist) {
    if (TASK == PDS_TASK_HOVER) {  // envs/hover.py:201-229
      px += s.pos[0]; py += s.pos[1]; pz += s.pos[2];
      r0 = s.rpy[0]; r1 = s.rpy[1]; r2 = s.rpy[2];
      vx += s.vel[0]; vy += s.vel[1]; vz += s.vel[2];
      w0 += s.w[0]; w1 += s.w[1]; w2 = s.w[2];
    } else if (TASK == PDS_TASK_CIRCLE) {  // envs/circle.py:225-257
      ref_offset = s.ref_offset;
      float tx, ty, tz;
      target_at<TASK>(k, ref_lds, ref_offset, tx, ty, tz);
      px = tx + s.pos[0]; py = ty + s.pos[1]; pz = tz + s.pos[2];
      r0 = s.rpy[0]; r1 = s.rpy[1]; r2 = s.rpy[2];
      vx += s.vel[0]; vy += s.vel[1]; vz += s.vel[2];
      w0 = s.w[0]; w1 = s.w[1]; w2 = s.w[2];
    } else {  // envs/takeoff.py:186-191
      px += s.pos[0]; py += s.pos[1];
      r0 = 0.f; r1 = 0.f; r2 = s.rpy[2];
    }
    if (TASK != PDS_TASK_TAKEOFF) {
      mx = make_float4(s.mx[0], s.mx[1], s.mx[2], s.mx[3]);
      u0 = make_float4(clampf(s.act[0], -1.f, 1.f), clampf(s.act[1], -1.f, 1.f),
                       clampf(s.act[2], -1.f, 1.f), clampf(s.act[3], -1.f, 1.f));
    }
  }
  if (TASK == PDS_TASK_TAKEOFF) {  // envs/takeoff.py:209-212 (unconditional)
    mx = make_float4(0.f, 0.f, 0.f, 0.f);
    u0 = make_float4(-1.f, -1.f, -1.f, -1.f);
  }
  default_params(k, par);
  if (DR) {  // envs/base.py:259-287
    par.dt = s.dt; par.m = s.m; par.Jx = s.J[0]; par.Jy = s.J[1]; par.Jz = s.J[2]; par.ftf1 = s.ftf1;
    if (MOTOR) {  // envs/agents.py:208-224 (K uses the hard-coded 0.028)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float T = fmaxf(s.T[i], par.dt);
        par.A[i] = 1.0f - par.dt / T;
        par.K[i] = 0.028f * k.G * s.t2w[i] / 4.0f;
      }
    }
  }
  q = quat_from_euler(r0, r1, r2);
  float R[9];
  matrix_from_quat(q, R);
  // bc.resetBaseVelocity(R^T w) then update_information: R^T (R^T w)
  const float a0 = R[0] * w0 + R[3] * w1 + R[6] * w2;
  const float a1 = R[1] * w0 + R[4] * w1 + R[7] * w2;
  const float a2 = R[2] * w0 + R[5] * w1 + R[8] * w2;
  e.wx = R[0] * a0 + R[3] * a1 + R[6] * a2;
  e.wy = R[1] * a0 + R[4] * a1 + R[7] * a2;
  e.wz = R[2] * a0 + R[5] * a1 + R[8] * a2;
  e.px = px; e.py = py; e.pz = pz; e.vx = vx; e.vy = vy; e.vz = vz;
  // rpy = Euler(quat) (envs/agents.py:446).  The quaternion keeps the sign of Q(sampled rpy) until
  // the first step while the state stores the wrapped Euler angles, so remember whether Q(wrapped)
  // has the opposite sign.
  uint32_t sign;
  if (fabsf(r0) < 1.55f && fabsf(r1) < 1.55f) {
    // every sampled attitude lands here: roll/pitch inside the principal range, so
    // Euler(Q(r,p,y)) == (r, p, y - 2 pi k) and Q flips sign once per 2 pi of yaw
    const float kk = rintf(r2 * 0.15915494309189533577f);
    float yw = fmaf(-kk, 6.2831854820251465f, r2);
    yw = fmaf(kk, 1.7484555e-7f, yw);
    e.roll = r0; e.pitch = r1; e.yaw = yw;
    sign = ((int)kk) & 1;
  } else {  // init_rpy overrides near / beyond gimbal lock: the general pybullet formulas
    euler_from_quat(q, e.roll, e.pitch, e.yaw);
    const Quat qw = quat_from_euler(e.roll, e.pitch, e.yaw);
    sign = (qw.x * q.x + qw.y * q.y + qw.z * q.z + qw.w * q.w) < 0.f ? 1u : 0u;
  }
  ctr = ctr_pack(0u, sign, (uint32_t)ref_offset);
}

// one observation o (envs/agents.py:339-348 get_state; envs/circle.py:173-177; envs/takeoff.py:146-147)
template <int TASK>
PDS_DEV void write_obs_half(float *row, const EnvRegs &e, const Quat &q, const float4 &last_action,
                            float tx, float ty, float tz, const float4 &hist_action) {
  int n = 0;
  row[n++] = e.px; row[n++] = e.py; row[n++] = e.pz;
  row[n++] = q.x; row[n++] = q.y; row[n++] = q.z; row[n++] = q.w;
  row[n++] = e.vx; row[n++] = e.vy; row[n++] = e.vz;
  row[n++] = e.wx; row[n++] = e.wy; row[n++] = e.wz;
  if (TASK != PDS_TASK_CIRCLE) {
    row[n++] = last_action.x; row[n++] = last_action.y; row[n++] = last_action.z; row[n++] = last_action.w;
  }
  if (TASK != PDS_TASK_HOVER) {
    row[n++] = tx - e.px; row[n++] = ty - e.py; row[n++] = tz - e.pz;
  }
  row[n++] = hist_action.x; row[n++] = hist_action.y; row[n++] = hist_action.z; row[n++] = hist_action.w;
}

template <int TASK>
PDS_DEV int target_index(int step, int agg, int ref_offset) {
  if (TASK == PDS_TASK_CIRCLE) {
    int t = step + ref_offset;  // (iteration // agg + ref_offset) % 300, envs/circle.py:130
    t = t % kRefPoints;
    return t;
  }
  if (TASK == PDS_TASK_TAKEOFF) return min(step * agg, kRefPoints - 1);  // envs/takeoff.py:108
  return 0;
}

// Wave-cooperative copy of this wave's [rows, D] LDS tile to global memory (contiguous region).
template <int D>
PDS_DEV void flush_tile(const float *tile, float *gdst, int rows, int lane) {
  if (rows == kWave) {
    constexpr int NV = kWave * D / 4;  // float4 count (D is even, 64*D divisible by 4)
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

template <int TASK, bool MOTOR, bool DR>
PDS_DEV void drain_reset_queue(const StepArgs &a, const float2 *ref_lds, const uint32_t *queue, int qcount, int lane);

// Inputs of one env-step, loaded 16 B/lane.  Kept in a struct so the persistent loop can prefetch the
// next tile's inputs into a second register set while the current tile is being computed.
struct Loaded {
  float4 act, q0, q1, q2, h1, h2, mx, p0, mA, mK;
  float2 p1;
  uint32_t ctr;
};

template <bool MOTOR, bool DR>
PDS_DEV void load_env(const StepArgs &a, long long ii, Loaded &L) {
  L.act = nt_load4(a.actions + ii);  // read once per step: keep it out of the caches
  L.q0 = st_load4(a.st.s0 + ii);
  L.q1 = st_load4(a.st.s1 + ii);
  L.q2 = st_load4(a.st.s2 + ii);
  L.h1 = st_load4(a.st.hist[a.parity] + ii);      // u(k-1)
  L.h2 = st_load4(a.st.hist[a.parity ^ 1] + ii);  // u(k-2)
  L.ctr = a.st.ctr[ii];
  if (MOTOR) L.mx = a.st.mx[ii];
  if (DR) {
    L.p0 = a.st.par0[ii];
    L.p1 = a.st.par1[ii];
    if (MOTOR) { L.mA = a.st.mA[ii]; L.mK = a.st.mK[ii]; }
  }
}

// Persistent kernel: grid = (blocks that fit on the chip), every wave walks over 64-env tiles with
// a grid stride and prefetches the next tile's inputs before computing the current one.
template <int TASK, bool MOTOR, bool DR, bool GE>
__global__ __launch_bounds__(kBlock) void step_kernel(const StepArgs a) {
  using L = ObsLayout<TASK, false>;
  constexpr int D = L::D;
  __shared__ __attribute__((aligned(16))) float tile_all[kBlock * D];
  __shared__ float2 ref_lds[(TASK == PDS_TASK_CIRCLE) ? kRefPoints : 1];
  const Consts &k = a.k;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;
  if (TASK == PDS_TASK_CIRCLE) {
    for (int t = tid; t < kRefPoints; t += kBlock) ref_lds[t] = a.st.circle_ref[t];
    __syncthreads();
  }
  __shared__ uint32_t queue_all[(kBlock / kWave) * kQueueCap];
  uint32_t *queue = queue_all + wave * kQueueCap;
  int qcount = 0;  // wave-uniform
  float *tile = tile_all + wave * (kWave * D);
  float *row = tile + lane * D;
  const long long ntiles = (a.n + kWave - 1) / kWave;
  const long long tstride = (long long)gridDim.x * (kBlock / kWave);
  long long t = (long long)blockIdx.x * (kBlock / kWave) + wave;
  if (t >= ntiles) return;  // wave-uniform

  Loaded cur, nxt;
  {
    const long long i0 = t * kWave + lane;
    load_env<MOTOR, DR>(a, i0 < a.n ? i0 : a.n - 1, cur);
  }
  for (; t < ntiles; t += tstride) {
    const long long wave_base = t * kWave;
    const long long i = wave_base + lane;
    const bool active = i < a.n;
    const long long ii = active ? i : (a.n - 1);  // tail lanes recompute the last env, stores masked
    if (t + tstride < ntiles) {  // prefetch (wave-uniform branch)
      const long long j = (t + tstride) * kWave + lane;
      load_env<MOTOR, DR>(a, j < a.n ? j : a.n - 1, nxt);
    }

    const float4 act = cur.act, h1 = cur.h1, h2 = cur.h2;
    const uint32_t ctr = cur.ctr;
    float4 mx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MOTOR) mx = cur.mx;
    Params par;
    default_params(k, par);
    if (DR) {
      par.dt = cur.p0.x; par.m = cur.p0.y; par.Jx = cur.p0.z; par.Jy = cur.p0.w; par.Jz = cur.p1.x; par.ftf1 = cur.p1.y;
      if (MOTOR) {
        par.A[0] = cur.mA.x; par.A[1] = cur.mA.y; par.A[2] = cur.mA.z; par.A[3] = cur.mA.w;
        par.K[0] = cur.mK.x; par.K[1] = cur.mK.y; par.K[2] = cur.mK.z; par.K[3] = cur.mK.w;
      }
    }
    EnvRegs e{cur.q0.x, cur.q0.y, cur.q0.z, cur.q0.w, cur.q1.x, cur.q1.y, cur.q1.z, cur.q1.w,
              cur.q2.x, cur.q2.y, cur.q2.z, cur.q2.w};
    const int step = (int)ctr_step(ctr);
    const int ref_offset = (int)ctr_off(ctr);

    // ---- o(k): rebuilt from the pre-step state instead of being re-read from HBM ----------------
    Quat q = quat_from_euler(e.roll, e.pitch, e.yaw);
    if (ctr_sign(ctr)) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    {
      float tx, ty, tz;
      target_at<TASK>(k, ref_lds, target_index<TASK>(step, k.agg, ref_offset), tx, ty, tz);
      write_obs_half<TASK>(row, e, q, h1, tx, ty, tz, h2);
    }

    // ---- aggregate_phy_steps x SimplePhysics.step_forward (envs/base.py:457-465) ---------------
    float xm[4] = {mx.x, mx.y, mx.z, mx.w};
    const float av[4] = {act.x, act.y, act.z, act.w};
    // divisions by the per-env mass / inertia become multiplications by v_rcp_f32 results (1 ulp)
    const float inv_m = fast_rcp(par.m), inv_Jx = fast_rcp(par.Jx), inv_Jy = fast_rcp(par.Jy), inv_Jz = fast_rcp(par.Jz);
    for (int sub = 0; sub < k.agg; ++sub) {
      // CrazyFlieAgent.apply_action, envs/agents.py:259-298 (+ PWM.act envs/control.py:94-100)
      float f[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pwm = 30000.f + clampf(av[j], -1.f, 1.f) * 30000.f;
        const float un = pwm * (1.0f / 60000.f);
        float n;
        if (MOTOR) {
          xm[j] = par.A[j] * xm[j] + (1.0f - par.A[j]) * fast_sqrt(un);
          n = xm[j] * xm[j];
        } else {
          n = un;
        }
        f[j] = par.K[j] * clampf(n, 0.f, 1.f);
      }
      // yaw torque: sum of +-(ftf1*f_i + ftf0); ftf0 cancels (envs/agents.py:295-297)
      const float tz_ = par.ftf1 * (-f[0] + f[1] - f[2] + f[3]);
      float R[9];
      matrix_from_quat(q, R);  // envs/physics.py:160 (quaternion of the PREVIOUS step)
      if (GE) {
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
    }

    // ---- task: target, done, reward, cost ---------------------------------------------------
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

    // ---- o(k+1) and u(k-1) -> second half of the row (envs/base.py:303-319) ------------------
    write_obs_half<TASK>(row + L::O + 4, e, q, act, tx, ty, tz, h1);

    const uint32_t ctr_new = ctr_pack((uint32_t)(step + 1), 0u, (uint32_t)ref_offset);
    // ---- auto-reset: DEFERRED.  ~2 % of the envs finish per step under random actions, i.e. 3 of
    // 4 waves would run the (long, transcendental-heavy) reset path for one or two live lanes.  A
    // finished env only hands its last observation to final_obs here and queues its index in LDS;
    // the wave resets its queued envs densely after its last tile (or when the queue fills up).
    const bool need_reset = a.auto_reset && (done || trunc) && active;
    {
      unsigned long long m = __ballot(need_reset);
      if (m != 0ull) {  // wave-uniform
        const int pos = qcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (need_reset) queue[pos] = (uint32_t)i | ((uint32_t)ref_offset << 23);
        qcount += __popcll(m);
        if (a.final_obs != nullptr) {
          // last observation of each finished env -> final_obs: the whole wave copies one row
          // (D <= 64 contiguous floats) per finished env straight out of the LDS tile
          do {
            const int src_lane = __builtin_ctzll(m);
            m &= m - 1ull;
            if (lane < D) a.final_obs[(wave_base + src_lane) * D + lane] = tile[src_lane * D + lane];
          } while (m != 0ull);
        }
      }
    }

    // ---- coalesced stores ----------------------------------------------------------------------
    if (active) {
      st_store4(a.st.s0 + i, make_float4(e.px, e.py, e.pz, e.vx));
      st_store4(a.st.s1 + i, make_float4(e.vy, e.vz, e.roll, e.pitch));
      st_store4(a.st.s2 + i, make_float4(e.yaw, e.wx, e.wy, e.wz));
      st_store4(a.st.hist[a.parity ^ 1] + i, act);  // overwrites u(k-2); next step's parity makes it u(k-1)
      a.st.ctr[i] = ctr_new;
      if (MOTOR) a.st.mx[i] = make_float4(xm[0], xm[1], xm[2], xm[3]);
      nt_store(a.reward + i, reward);
      nt_store(a.cost + i, cost);
      nt_store(a.term + i, (uint8_t)(done ? 1 : 0));
      nt_store(a.trunc + i, (uint8_t)(trunc ? 1 : 0));
    }
    // LDS rows of this wave were written by its own lanes only: wave-synchronous, no block barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const long long rem = a.n - wave_base;
    flush_tile<D>(tile, a.obs + wave_base * D, rem >= kWave ? kWave : (int)rem, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the tile is rewritten by the next iteration
    if (qcount > kQueueCap - kWave) {  // wave-uniform: next tile could overflow the queue
      drain_reset_queue<TASK, MOTOR, DR>(a, ref_lds, queue, qcount, lane);
      qcount = 0;
    }
    cur = nxt;
  }
  if (qcount > 0) drain_reset_queue<TASK, MOTOR, DR>(a, ref_lds, queue, qcount, lane);
}

// Reset of one env, split in a pure-compute half and a store half so that the deferred auto-reset
// drain can do its arithmetic while the wave's earlier stores are still draining.
struct ResetOut {
  EnvRegs e;
  Quat q;
  float4 u0, mx;
  Params par;
  uint32_t ctr;
};

template <int TASK, bool MOTOR, bool DR>
PDS_DEV void reset_compute(const StepArgs &a, const float2 *ref_lds, long long i, uint32_t ctr_old,
                           const float *sample_row, ResetOut &r) {
  Sample s;
  if (sample_row != nullptr) sample_load(sample_row, s);
  else sample_philox<TASK, MOTOR, DR>(a.k, (uint32_t)(a.env_id_base + (unsigned long long)i), a, s);
  r.ctr = ctr_old;
  reset_env<TASK, MOTOR, DR>(a.k, ref_lds, s, r.e, r.q, r.u0, r.mx, r.par, r.ctr);
}

// Writes the complete state, parameters and (optionally) the observation row [o0,u0,o0,u0]
// (envs/base.py:417-431) of a reset env straight to HBM.  Rows are strided -> only for the explicit
// reset kernel and the deferred auto-reset drain, never on the per-step stream.
template <int TASK, bool MOTOR, bool DR>
PDS_DEV void reset_store(const StepArgs &a, const float2 *ref_lds, long long i, const ResetOut &r) {
  using L = ObsLayout<TASK, false>;
  constexpr int D = L::D;
  const EnvRegs &e = r.e;
  a.st.s0[i] = make_float4(e.px, e.py, e.pz, e.vx);
  a.st.s1[i] = make_float4(e.vy, e.vz, e.roll, e.pitch);
  a.st.s2[i] = make_float4(e.yaw, e.wx, e.wy, e.wz);
  a.st.hist[0][i] = r.u0;
  a.st.hist[1][i] = r.u0;
  a.st.ctr[i] = r.ctr;
  if (MOTOR) a.st.mx[i] = r.mx;
  if (DR) {
    a.st.par0[i] = make_float4(r.par.dt, r.par.m, r.par.Jx, r.par.Jy);
    a.st.par1[i] = make_float2(r.par.Jz, r.par.ftf1);
    if (MOTOR) {
      a.st.mA[i] = make_float4(r.par.A[0], r.par.A[1], r.par.A[2], r.par.A[3]);
      a.st.mK[i] = make_float4(r.par.K[0], r.par.K[1], r.par.K[2], r.par.K[3]);
    }
  }
  if (a.obs != nullptr) {
    float rowbuf[D];
    float tx, ty, tz;
    target_at<TASK>(a.k, ref_lds, target_index<TASK>(0, a.k.agg, (int)ctr_off(r.ctr)), tx, ty, tz);
    write_obs_half<TASK>(rowbuf, e, r.q, r.u0, tx, ty, tz, r.u0);
    write_obs_half<TASK>(rowbuf + L::O + 4, e, r.q, r.u0, tx, ty, tz, r.u0);
    float2 *dst = reinterpret_cast<float2 *>(a.obs + i * D);
#pragma unroll
    for (int j = 0; j < D / 2; ++j) dst[j] = make_float2(rowbuf[2 * j], rowbuf[2 * j + 1]);
  }
}

// Dense pass over the envs a wave queued for auto-reset (queue entry = env index | ref_offset << 23).
// The arithmetic touches no memory, so it overlaps with the wave's outstanding stores; those must
// have completed (s_waitcnt vmcnt(0)) before the same addresses are overwritten.
template <int TASK, bool MOTOR, bool DR>
PDS_DEV void drain_reset_queue(const StepArgs &a, const float2 *ref_lds, const uint32_t *queue, int qcount, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int base = 0; base < qcount; base += kWave) {
    const int idx = base + lane;
    const bool on = idx < qcount;
    ResetOut r;
    long long i = 0;
    if (on) {
      const uint32_t ent = queue[idx];
      i = (long long)(ent & 0x7FFFFFu);
      reset_compute<TASK, MOTOR, DR>(a, ref_lds, i, ctr_pack(0u, 0u, ent >> 23), nullptr, r);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (on) reset_store<TASK, MOTOR, DR>(a, ref_lds, i, r);
  }
  __builtin_amdgcn_wave_barrier();
}

// Explicit reset (pds_reset / pds_reset_from_samples): not a hot path.
template <int TASK, bool MOTOR, bool DR>
__global__ __launch_bounds__(kBlock) void reset_kernel(const StepArgs a) {
  __shared__ float2 ref_lds[(TASK == PDS_TASK_CIRCLE) ? kRefPoints : 1];
  if (TASK == PDS_TASK_CIRCLE) {
    for (int t = threadIdx.x; t < kRefPoints; t += kBlock) ref_lds[t] = a.st.circle_ref[t];
    __syncthreads();
  }
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.n) return;
  if (a.mask != nullptr && a.mask[i] == 0) return;
  ResetOut r;
  reset_compute<TASK, MOTOR, DR>(a, ref_lds, i, a.st.ctr[i], a.samples != nullptr ? a.samples + i * PDS_SAMPLE_FLOATS : nullptr, r);
  reset_store<TASK, MOTOR, DR>(a, ref_lds, i, r);
}

// ---- state access (parity injection, checkpointing) -------------------------------------------
struct FieldArgs {
  DevState st;
  Consts k;
  void *user;
  long long n;
  int field;
  int parity;
  int set;
  int has_motor, has_dr;
};

__global__ __launch_bounds__(kBlock) void field_kernel(const FieldArgs a) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.n) return;
  float *uf = reinterpret_cast<float *>(a.user);
  int32_t *ui = reinterpret_cast<int32_t *>(a.user);
  float4 q0 = a.st.s0[i], q1 = a.st.s1[i], q2 = a.st.s2[i];
  uint32_t c = a.st.ctr[i];
  switch (a.field) {
    case PDS_F_POS:
      if (a.set) { q0.x = uf[3 * i]; q0.y = uf[3 * i + 1]; q0.z = uf[3 * i + 2]; a.st.s0[i] = q0; }
      else { uf[3 * i] = q0.x; uf[3 * i + 1] = q0.y; uf[3 * i + 2] = q0.z; }
      break;
    case PDS_F_VEL:
      if (a.set) { q0.w = uf[3 * i]; q1.x = uf[3 * i + 1]; q1.y = uf[3 * i + 2]; a.st.s0[i] = q0; a.st.s1[i] = q1; }
      else { uf[3 * i] = q0.w; uf[3 * i + 1] = q1.x; uf[3 * i + 2] = q1.y; }
      break;
    case PDS_F_RPY:
      if (a.set) { q1.z = uf[3 * i]; q1.w = uf[3 * i + 1]; q2.x = uf[3 * i + 2]; a.st.s1[i] = q1; a.st.s2[i] = q2; }
      else { uf[3 * i] = q1.z; uf[3 * i + 1] = q1.w; uf[3 * i + 2] = q2.x; }
      break;
    case PDS_F_OMEGA:
      if (a.set) { q2.y = uf[3 * i]; q2.z = uf[3 * i + 1]; q2.w = uf[3 * i + 2]; a.st.s2[i] = q2; }
      else { uf[3 * i] = q2.y; uf[3 * i + 1] = q2.z; uf[3 * i + 2] = q2.w; }
      break;
    case PDS_F_QUAT:
      if (!a.set) {
        Quat q = quat_from_euler(q1.z, q1.w, q2.x);
        const float sgn = ctr_sign(c) ? -1.f : 1.f;
        uf[4 * i] = sgn * q.x; uf[4 * i + 1] = sgn * q.y; uf[4 * i + 2] = sgn * q.z; uf[4 * i + 3] = sgn * q.w;
      }
      break;
    case PDS_F_MOTOR_X:
    case PDS_F_LAST_ACTION:
    case PDS_F_PREV_ACTION:
    case PDS_F_MOTOR_A:
    case PDS_F_MOTOR_K: {
      float4 *arr = nullptr;
      float4 dflt = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.field == PDS_F_MOTOR_X) arr = a.has_motor ? a.st.mx : nullptr;
      else if (a.field == PDS_F_LAST_ACTION) arr = a.st.hist[a.parity];
      else if (a.field == PDS_F_PREV_ACTION) arr = a.st.hist[a.parity ^ 1];
      else if (a.field == PDS_F_MOTOR_A) { arr = (a.has_motor && a.has_dr) ? a.st.mA : nullptr; dflt = make_float4(a.k.A, a.k.A, a.k.A, a.k.A); }
      else { arr = (a.has_motor && a.has_dr) ? a.st.mK : nullptr; dflt = make_float4(a.k.K, a.k.K, a.k.K, a.k.K); }
      if (a.set) { if (arr) arr[i] = make_float4(uf[4 * i], uf[4 * i + 1], uf[4 * i + 2], uf[4 * i + 3]); }
      else { const float4 v = arr ? arr[i] : dflt; uf[4 * i] = v.x; uf[4 * i + 1] = v.y; uf[4 * i + 2] = v.z; uf[4 * i + 3] = v.w; }
      break;
    }
    case PDS_F_STEP_COUNT:
      if (a.set) a.st.ctr[i] = ctr_pack((uint32_t)ui[i] & 0xFFFFu, ctr_sign(c), ctr_off(c));
      else ui[i] = (int32_t)ctr_step(c);
      break;
    case PDS_F_QUAT_SIGN:
      if (a.set) a.st.ctr[i] = ctr_pack(ctr_step(c), ui[i] ? 1u : 0u, ctr_off(c));
      else ui[i] = (int32_t)ctr_sign(c);
      break;
    case PDS_F_REF_OFFSET:
      if (a.set) a.st.ctr[i] = ctr_pack(ctr_step(c), ctr_sign(c), (uint32_t)ui[i] % 300u);
      else ui[i] = (int32_t)ctr_off(c);
      break;
    case PDS_F_PARAMS:
      if (a.has_dr) {
        if (a.set) {
          a.st.par0[i] = make_float4(uf[6 * i], uf[6 * i + 1], uf[6 * i + 2], uf[6 * i + 3]);
          a.st.par1[i] = make_float2(uf[6 * i + 4], uf[6 * i + 5]);
        } else {
          const float4 p0 = a.st.par0[i]; const float2 p1 = a.st.par1[i];
          uf[6 * i] = p0.x; uf[6 * i + 1] = p0.y; uf[6 * i + 2] = p0.z; uf[6 * i + 3] = p0.w; uf[6 * i + 4] = p1.x; uf[6 * i + 5] = p1.y;
        }
      } else if (!a.set) {
        uf[6 * i] = a.k.dt; uf[6 * i + 1] = a.k.m; uf[6 * i + 2] = a.k.Jx; uf[6 * i + 3] = a.k.Jy; uf[6 * i + 4] = a.k.Jz; uf[6 * i + 5] = a.k.ftf1;
      }
      break;
    default: break;
  }
}

}  // namespace pds

// =================================================================================================
// host side: the C ABI of include/pds.h
// =================================================================================================
using namespace pds;

struct pds_handle {
  pds_config cfg;
  DevState st;
  Consts k;
  float2 *d_circle_ref;
  int obs_dim;
  int num_cus;
  long long grid_override;
  int parity;
  uint64_t tick;
  bool was_reset;
  char err[512];
};

static int fail(pds_handle *h, int code, const char *fmt, ...) {
  if (h) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(h->err, sizeof(h->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

static thread_local char g_create_err[512] = "";

#define PDS_HIP(h, call)                                                                        \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) return fail(h, PDS_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

extern "C" int pds_version(void) { return PDS_VERSION; }

// ctor defaults: envs/base.py:26-48, envs/hover.py:7-45, envs/circle.py:7-61, envs/takeoff.py:13-56
extern "C" int pds_default_config(int task, pds_config *c) {
  if (!c || task < 0 || task > 2) return PDS_EINVAL;
  memset(c, 0, sizeof(*c));
  c->struct_size = (int32_t)sizeof(pds_config);
  c->task = task;
  c->num_envs = 1;
  c->env_id_base = 0;
  c->seed = 0;
  c->device = 0;
  c->use_motor_dynamics = 0;
  c->use_ground_effect = 0;
  c->observation_noise = 1;
  c->aggregate_phy_steps = 1;
  c->enable_reset_distribution = 1;
  c->max_episode_steps = 500;
  c->auto_reset = 1;
  c->domain_randomization = 0.10;
  c->motor_thrust_noise = 0.05;
  c->time_step = 1.0 / 100.0;
  c->motor_time_constant = 0.080;
  c->penalty_action = 1e-4;
  c->penalty_angle = 0.0;
  c->penalty_spin = (task == PDS_TASK_CIRCLE) ? 1e-3 : 1e-4;
  c->penalty_terminal = 100.0;
  c->penalty_velocity = (task == PDS_TASK_CIRCLE) ? 1e-4 : 0.0;
  c->ARP = (task == PDS_TASK_CIRCLE) ? 1e-3 : 0.0;
  c->target_pos[2] = 1.0;
  c->init_xyz[2] = (task == PDS_TASK_TAKEOFF) ? (double)0.0125f : 1.0;
  return PDS_OK;
}

static void fill_consts(const pds_config &c, Consts &k) {
  // envs/assets/cf21x_sys_eq.urdf:10,16-17 and envs/agents.py:138-156
  const double M = 0.027, L = 0.0397, T2W = 2.25, IXX = 1.7e-5, IYY = 1.7e-5, IZZ = 2.9e-5;
  const double KF = 3.16e-10, GEC = 11.36859, PR = 2.31348e-2, FTF1 = 5.96e-3, G = 9.81;
  const double MAX_THRUST = G * M * T2W / 4;
  const double MAX_RPM = sqrt((T2W * G * M) / (4 * MAX_THRUST));
  memset(&k, 0, sizeof(k));
  k.K = (float)MAX_THRUST; k.G = (float)G; k.m = (float)M;
  k.Jx = (float)IXX; k.Jy = (float)IYY; k.Jz = (float)IZZ; k.ftf1 = (float)FTF1;
  k.Lq = (float)(L / sqrt(2.0));
  k.dt = (float)c.time_step;
  k.A = (float)(1.0 - c.time_step / c.motor_time_constant);
  k.hover_x = (float)sqrt(1 / T2W);
  k.hover_action = (float)(2 * 1 / T2W - 1);
  k.gec = (float)GEC; k.prop_r = (float)PR;
  k.h_clip = (float)(0.25 * PR * sqrt((15 * MAX_RPM * MAX_RPM * KF * GEC) / MAX_THRUST));
  k.t2w = (float)T2W; k.mtc = (float)c.motor_time_constant;
  k.M_nom = (float)M; k.Jx_nom = (float)IXX; k.Jy_nom = (float)IYY; k.Jz_nom = (float)IZZ;
  k.ftf1_nom = (float)FTF1; k.dt_nom = (float)c.time_step;
  k.pa = (float)c.penalty_action; k.pang = (float)c.penalty_angle; k.pspin = (float)c.penalty_spin;
  k.pterm = (float)c.penalty_terminal; k.pvel = (float)c.penalty_velocity; k.arp = (float)c.ARP;
  for (int i = 0; i < 3; ++i) {
    k.target[i] = (float)c.target_pos[i];
    k.init_xyz[i] = (float)c.init_xyz[i]; k.init_rpy[i] = (float)c.init_rpy[i];
    k.init_vel[i] = (float)c.init_xyz_dot[i]; k.init_w[i] = (float)c.init_rpy_dot[i];
  }
  k.dr = (float)(c.domain_randomization > 0 ? c.domain_randomization : 0.0);
  k.agg = c.aggregate_phy_steps; k.max_steps = c.max_episode_steps;
  k.reset_dist = c.enable_reset_distribution ? 1 : 0;
}

static int obs_dim_of(const pds_config &c) {
  const bool noisy = c.observation_noise > 0;
  const int o = noisy ? (c.task == PDS_TASK_HOVER ? 13 : (c.task == PDS_TASK_CIRCLE ? 16 : 20))
                      : (c.task == PDS_TASK_HOVER ? 17 : (c.task == PDS_TASK_CIRCLE ? 16 : 20));
  return 2 * (o + 4);
}

extern "C" int pds_create(const pds_config *cfg, pds_handle **out) {
  if (!cfg || !out) return PDS_EINVAL;
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(pds_config)) { snprintf(g_create_err, sizeof(g_create_err), "pds_config size mismatch"); return PDS_EINVAL; }
  if (cfg->task < 0 || cfg->task > 2 || cfg->num_envs < 1 || cfg->aggregate_phy_steps < 1 ||
      cfg->max_episode_steps < 1 || cfg->max_episode_steps > 65535 || cfg->time_step <= 0 ||
      cfg->num_envs > (1ll << 23)) {  // env index + ref_offset share a 32-bit reset-queue word
    snprintf(g_create_err, sizeof(g_create_err), "invalid pds_config");
    return PDS_EINVAL;
  }
  if (cfg->observation_noise > 0 || cfg->motor_thrust_noise > 0) {
    snprintf(g_create_err, sizeof(g_create_err),
             "stochastic sensor/thrust noise is not built yet: pass observation_noise<=0 and motor_thrust_noise=0");
    return PDS_EUNSUPPORTED;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device >= ndev) {
    snprintf(g_create_err, sizeof(g_create_err), "no HIP device %d (found %d); there is no CPU fallback", cfg->device, ndev);
    return PDS_ENODEVICE;
  }
  pds_handle *h = new (std::nothrow) pds_handle();
  if (!h) return PDS_ENOMEM;
  memset(h, 0, sizeof(*h));
  h->cfg = *cfg;
  fill_consts(*cfg, h->k);
  h->obs_dim = obs_dim_of(*cfg);
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess || cus <= 0) cus = 256;
    h->num_cus = cus;
    const char *g = getenv("PDS_GRID_BLOCKS");
    h->grid_override = g ? atoll(g) : 0;
  }
  const size_t n = (size_t)cfg->num_envs;
  const bool motor = cfg->use_motor_dynamics != 0, dr = cfg->domain_randomization > 0;
  hipError_t e = hipSetDevice(cfg->device);
  auto alloc = [&](void **p, size_t bytes) { if (e == hipSuccess) { e = hipMalloc(p, bytes); if (e == hipSuccess) e = hipMemset(*p, 0, bytes); } };
  alloc((void **)&h->st.s0, n * 16); alloc((void **)&h->st.s1, n * 16); alloc((void **)&h->st.s2, n * 16);
  alloc((void **)&h->st.hist[0], n * 16); alloc((void **)&h->st.hist[1], n * 16);
  alloc((void **)&h->st.ctr, n * 4);
  if (motor) alloc((void **)&h->st.mx, n * 16);
  if (dr) { alloc((void **)&h->st.par0, n * 16); alloc((void **)&h->st.par1, n * 8); }
  if (dr && motor) { alloc((void **)&h->st.mA, n * 16); alloc((void **)&h->st.mK, n * 16); }
  alloc((void **)&h->d_circle_ref, kRefPoints * sizeof(float2));
  if (e == hipSuccess) {
    float2 ref[kRefPoints];  // envs/circle.py:45-56
    for (int t = 0; t < kRefPoints; ++t) {
      const double ts = 2 * M_PI * (double)t / kRefPoints;
      ref[t].x = (float)(0.25 * (1 - cos(ts)));
      ref[t].y = (float)(0.25 * sin(ts));
    }
    e = hipMemcpy(h->d_circle_ref, ref, sizeof(ref), hipMemcpyHostToDevice);
  }
  h->st.circle_ref = h->d_circle_ref;
  if (e != hipSuccess) {
    snprintf(g_create_err, sizeof(g_create_err), "allocation of %zu envs failed: %s", n, hipGetErrorString(e));
    pds_destroy(h);
    return e == hipErrorOutOfMemory ? PDS_ENOMEM : PDS_EHIP;
  }
  *out = h;
  return PDS_OK;
}

extern "C" int pds_destroy(pds_handle *h) {
  if (!h) return PDS_OK;
  (void)hipSetDevice(h->cfg.device);
  void *ptrs[] = {h->st.s0, h->st.s1, h->st.s2, h->st.hist[0], h->st.hist[1], h->st.ctr, h->st.mx,
                  h->st.par0, h->st.par1, h->st.mA, h->st.mK, h->d_circle_ref};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  delete h;
  return PDS_OK;
}

extern "C" int pds_obs_dim(const pds_handle *h) { return h ? h->obs_dim : PDS_EINVAL; }
extern "C" int64_t pds_num_envs(const pds_handle *h) { return h ? h->cfg.num_envs : PDS_EINVAL; }
extern "C" uint64_t pds_tick(const pds_handle *h) { return h ? h->tick : 0; }
extern "C" const char *pds_last_error(const pds_handle *h) { return h ? h->err : g_create_err; }

// read action 16 + dyn state 48 + action history 32 + counter 4; write dyn state 48
// + newest history slot 16 + counter 4 + reward 4 + cost 4 + terminated 1 + truncated 1; obs 4*D;
// DR params +24; motor PT1: x R+W 32 (+ A, K 32 when randomised).
extern "C" int pds_bytes_per_env_step(const pds_handle *h) {
  if (!h) return PDS_EINVAL;
  int b = 100 + 78 + 4 * h->obs_dim;
  const bool motor = h->cfg.use_motor_dynamics != 0, dr = h->cfg.domain_randomization > 0;
  if (dr) b += 24;
  if (motor) b += 32;
  if (motor && dr) b += 32;
  return b;
}

template <int TASK, bool MOTOR, bool DR>
static void launch_step_ge(bool ge, dim3 grid, hipStream_t s, const StepArgs &a) {
  if (ge) hipLaunchKernelGGL((step_kernel<TASK, MOTOR, DR, true>), grid, dim3(kBlock), 0, s, a);
  else hipLaunchKernelGGL((step_kernel<TASK, MOTOR, DR, false>), grid, dim3(kBlock), 0, s, a);
}
template <int TASK>
static void launch_step_task(bool motor, bool dr, bool ge, dim3 grid, hipStream_t s, const StepArgs &a) {
  if (motor) { if (dr) launch_step_ge<TASK, true, true>(ge, grid, s, a); else launch_step_ge<TASK, true, false>(ge, grid, s, a); }
  else { if (dr) launch_step_ge<TASK, false, true>(ge, grid, s, a); else launch_step_ge<TASK, false, false>(ge, grid, s, a); }
}
template <int TASK>
static void launch_reset_task(bool motor, bool dr, dim3 grid, hipStream_t s, const StepArgs &a) {
  if (motor) { if (dr) hipLaunchKernelGGL((reset_kernel<TASK, true, true>), grid, dim3(kBlock), 0, s, a);
               else hipLaunchKernelGGL((reset_kernel<TASK, true, false>), grid, dim3(kBlock), 0, s, a); }
  else { if (dr) hipLaunchKernelGGL((reset_kernel<TASK, false, true>), grid, dim3(kBlock), 0, s, a);
         else hipLaunchKernelGGL((reset_kernel<TASK, false, false>), grid, dim3(kBlock), 0, s, a); }
}

static void base_args(pds_handle *h, StepArgs &a) {
  memset(&a, 0, sizeof(a));
  a.st = h->st;
  a.k = h->k;
  a.n = h->cfg.num_envs;
  a.env_id_base = (unsigned long long)h->cfg.env_id_base;
  a.seed_lo = (uint32_t)h->cfg.seed; a.seed_hi = (uint32_t)(h->cfg.seed >> 32);
  a.tick_lo = (uint32_t)h->tick; a.tick_hi = (uint32_t)(h->tick >> 32);
  a.parity = h->parity;
  a.auto_reset = h->cfg.auto_reset;
}

static int do_reset(pds_handle *h, const uint8_t *d_mask, const float *d_samples, float *d_obs, void *stream) {
  if (!h) return PDS_EINVAL;
  PDS_HIP(h, hipSetDevice(h->cfg.device));
  StepArgs a;
  base_args(h, a);
  a.mask = d_mask; a.samples = d_samples; a.obs = d_obs;
  const dim3 grid((unsigned)((a.n + kBlock - 1) / kBlock));
  const bool motor = h->cfg.use_motor_dynamics != 0, dr = h->cfg.domain_randomization > 0;
  hipStream_t s = (hipStream_t)stream;
  switch (h->cfg.task) {
    case PDS_TASK_HOVER: launch_reset_task<PDS_TASK_HOVER>(motor, dr, grid, s, a); break;
    case PDS_TASK_CIRCLE: launch_reset_task<PDS_TASK_CIRCLE>(motor, dr, grid, s, a); break;
    default: launch_reset_task<PDS_TASK_TAKEOFF>(motor, dr, grid, s, a); break;
  }
  PDS_HIP(h, hipGetLastError());
  h->tick += 1;
  h->was_reset = true;
  return PDS_OK;
}

extern "C" int pds_reset(pds_handle *h, const uint8_t *d_mask, float *d_obs, void *stream) {
  return do_reset(h, d_mask, nullptr, d_obs, stream);
}

extern "C" int pds_reset_from_samples(pds_handle *h, const uint8_t *d_mask, const float *d_samples,
                                      float *d_obs, void *stream) {
  if (h && !d_samples) return fail(h, PDS_EINVAL, "pds_reset_from_samples: d_samples is NULL");
  return do_reset(h, d_mask, d_samples, d_obs, stream);
}

extern "C" int pds_step(pds_handle *h, const float *d_actions, float *d_obs, float *d_reward,
                        uint8_t *d_terminated, uint8_t *d_truncated, float *d_cost, float *d_final_obs,
                        void *stream) {
  if (!h) return PDS_EINVAL;
  if (!d_actions || !d_obs || !d_reward || !d_terminated || !d_truncated || !d_cost)
    return fail(h, PDS_EINVAL, "pds_step: NULL tensor pointer");
  if ((((uintptr_t)d_actions) | ((uintptr_t)d_obs)) & 15u)
    return fail(h, PDS_EINVAL, "pds_step: d_actions and d_obs must be 16-byte aligned");
  if (!h->was_reset) return fail(h, PDS_EINVAL, "pds_step before pds_reset");
  PDS_HIP(h, hipSetDevice(h->cfg.device));
  StepArgs a;
  base_args(h, a);
  a.actions = reinterpret_cast<const float4 *>(d_actions);
  a.obs = d_obs; a.reward = d_reward; a.term = d_terminated; a.trunc = d_truncated; a.cost = d_cost;
  a.final_obs = d_final_obs;
  // persistent launch: at most kBlocksPerCU resident blocks per CU (LDS-bound: 4 wave tiles of
  // 64 x D floats per block), every wave strides over the 64-env tiles
  const long long blocks_needed = (a.n + kBlock - 1) / kBlock;
  // default: one 256-env block per 4 tiles (the hardware dispatcher balances blocks whose
  // deferred-reset drains have different lengths; measured faster than a resident grid)
  long long blocks_resident = blocks_needed;
  if (h->grid_override > 0) blocks_resident = h->grid_override;  // PDS_GRID_BLOCKS: tuning knob
  const dim3 grid((unsigned)(blocks_needed < blocks_resident ? blocks_needed : blocks_resident));
  (void)kBlocksPerCU;
  const bool motor = h->cfg.use_motor_dynamics != 0, dr = h->cfg.domain_randomization > 0;
  const bool ge = h->cfg.use_ground_effect != 0;
  hipStream_t s = (hipStream_t)stream;
  switch (h->cfg.task) {
    case PDS_TASK_HOVER: launch_step_task<PDS_TASK_HOVER>(motor, dr, ge, grid, s, a); break;
    case PDS_TASK_CIRCLE: launch_step_task<PDS_TASK_CIRCLE>(motor, dr, ge, grid, s, a); break;
    default: launch_step_task<PDS_TASK_TAKEOFF>(motor, dr, ge, grid, s, a); break;
  }
  PDS_HIP(h, hipGetLastError());
  h->parity ^= 1;
  h->tick += 1;
  return PDS_OK;
}

extern "C" int pds_field_width(int field) {
  switch (field) {
    case PDS_F_POS: case PDS_F_RPY: case PDS_F_VEL: case PDS_F_OMEGA: case PDS_F_GYRO_BIAS: case PDS_F_GYRO_LPF: return 3;
    case PDS_F_QUAT: case PDS_F_MOTOR_X: case PDS_F_LAST_ACTION: case PDS_F_PREV_ACTION: case PDS_F_MOTOR_A:
    case PDS_F_MOTOR_K: case PDS_F_OU: return 4;
    case PDS_F_STEP_COUNT: case PDS_F_QUAT_SIGN: case PDS_F_REF_OFFSET: return 1;
    case PDS_F_PARAMS: return 6;
    default: return PDS_EINVAL;
  }
}

static int do_field(pds_handle *h, int field, void *d_ptr, int set, void *stream) {
  if (!h) return PDS_EINVAL;
  if (!d_ptr || pds_field_width(field) < 0) return fail(h, PDS_EINVAL, "bad field %d or NULL pointer", field);
  if (field == PDS_F_OU || field == PDS_F_GYRO_BIAS || field == PDS_F_GYRO_LPF)
    return fail(h, PDS_EUNSUPPORTED, "noise state fields are not built yet");
  if (set && field == PDS_F_QUAT) return fail(h, PDS_EINVAL, "PDS_F_QUAT is derived (set PDS_F_RPY / PDS_F_QUAT_SIGN)");
  PDS_HIP(h, hipSetDevice(h->cfg.device));
  FieldArgs a;
  memset(&a, 0, sizeof(a));
  a.st = h->st; a.k = h->k; a.user = d_ptr; a.n = h->cfg.num_envs; a.field = field; a.parity = h->parity; a.set = set;
  a.has_motor = h->cfg.use_motor_dynamics != 0; a.has_dr = h->cfg.domain_randomization > 0;
  const dim3 grid((unsigned)((a.n + kBlock - 1) / kBlock));
  hipLaunchKernelGGL(field_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, a);
  PDS_HIP(h, hipGetLastError());
  if (set) h->was_reset = true;
  return PDS_OK;
}

extern "C" int pds_get_state(pds_handle *h, int field, void *d_out, void *stream) { return do_field(h, field, d_out, 0, stream); }
extern "C" int pds_set_state(pds_handle *h, int field, const void *d_in, void *stream) { return do_field(h, field, const_cast<void *>(d_in), 1, stream); }
