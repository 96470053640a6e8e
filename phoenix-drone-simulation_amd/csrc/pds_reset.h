// pds_reset.h -- sampling, sensor-noise and reset device functions (gfx950 only).
//
// Reference (paths relative to phoenix_drone_simulation/): envs/base.py:239-296,382-431,
// envs/hover.py:192-243, envs/circle.py:213-277, envs/takeoff.py:179-212, envs/agents.py:208-224,
// 377-386,434-453, envs/sensors.py:75-134, envs/utils.py:59-108.
#pragma once
#include "pds_types.h"

namespace pds {

// Philox block ids (counter word 3) -- the RNG contract restated by oracle/phoenix_oracle.c
constexpr uint32_t kBlkReset = 0;        // 0..8   reset distribution + domain randomisation (10 rounds)
constexpr uint32_t kBlkLatRows = 9;      // 9..15  rows 0..6 of the latency action buffer (10 rounds); 16 = ctor noise
constexpr uint32_t kBlkResetNoise = 32;  // 32..37 two add_noise calls inside reset (7 rounds)
constexpr uint32_t kBlkObsNoise = 64;    // 64..66 the add_noise call that produces o(k+1) (7 rounds)
constexpr uint32_t kBlkSubNoise = 128;   // 128+2*sub+{0,1}: OU + the discarded add_noise call (7 rounds)
constexpr uint32_t kBlkSubNoiseX = 256;  // 256+2*sub+{0,1}: position / velocity / angle draws of that call (obs_rate > 1)

// Where the Philox words of a reset come from: computed on the spot by the resetting thread
// (explicit reset kernel), or read back from an LDS scratch that the whole wave filled
// cooperatively (deferred auto-reset drain: one block per lane instead of 5-15 blocks in a row on
// the one or two lanes that own a finished env).
constexpr int kResetBlocks = 9;        // reset distribution + domain randomisation
constexpr int kObsCallBlocks = 3;     // one add_noise call that reaches the observation
constexpr int kResetNoiseBlocks = 2 * kObsCallBlocks;  // two add_noise calls
constexpr int kLatRowBlocks = kMaxLatSteps - 1;        // one block (4 normals) per extra action-buffer row
constexpr int kScratchBlocks = kResetBlocks + kResetNoiseBlocks + kLatRowBlocks;

// The standard variates of one Philox block, by the block's role.  One set of conversion functions serves every
// source below, so a reset draws the same floats whether its blocks are converted by the resetting thread or, side by
// side, by the lanes that computed them (LdsVariates).
PDS_DEV void words_to_uniforms4(const U4 &w, float (&u)[4]) {  // 24-bit uniforms in [0, 1)
  u[0] = u01(w.x); u[1] = u01(w.y); u[2] = u01(w.z); u[3] = u01(w.w);
}
PDS_DEV void words_to_normals4(const U4 &w, float (&z)[4]) {   // two full-precision Box-Muller pairs
  box_muller(w.x, w.y, z[0], z[1]);
  box_muller(w.z, w.w, z[2], z[3]);
}
// block b (0..2) of one add_noise call (round 5 layout): words 0..4 of the call -> the normals n[0..9] (one Box-Muller pair
// per word), words 5..7 -> 6 uniforms (16 bits each), words 8..11 -> the normals n[10..17]; f[2 i], f[2 i + 1] = the pair of
// word i of the block.  What the kept observation is regenerated from -- position, velocity and angle noise: n[0..8] and
// the uniforms -- sits in blocks 0 and 1, so regen_kept_obs computes two Philox blocks, not three (obs_noise_from_blocks).
PDS_DEV void words_to_noise8(const U4 &w, int b, float (&f)[8]) {
  const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float z0, z1;
    box_muller_word(ww[i], z0, z1);
    const bool uni = b == 1 && i > 0;
    f[2 * i] = uni ? u01_lo16(ww[i]) : z0;
    f[2 * i + 1] = uni ? u01_hi16(ww[i]) : z1;
  }
}

struct DirectWords {
  uint32_t env_id, tick_lo, tick_hi, seed_lo, seed_hi;
  PDS_DEV DirectWords(uint32_t id, const RngKey &r) : env_id(id), tick_lo(r.tick_lo), tick_hi(r.tick_hi), seed_lo(r.seed_lo), seed_hi(r.seed_hi) {}
  PDS_DEV U4 reset_block(uint32_t b) const { return philox4x32_10(env_id, tick_lo, tick_hi, kBlkReset + b, seed_lo, seed_hi); }
  PDS_DEV U4 noise_block(uint32_t b) const { return philox4x32_7(env_id, tick_lo, tick_hi, kBlkResetNoise + b, seed_lo, seed_hi); }
  PDS_DEV U4 lat_block(uint32_t r) const { return philox4x32_10(env_id, tick_lo, tick_hi, kBlkLatRows + r, seed_lo, seed_hi); }
  // scratch slot j of the cooperative fill (LdsWords layout): reset / latency-row blocks take 10 rounds, the
  // noise blocks 7 -- evaluated in one instruction stream for all lanes of the wave
  template <bool X3 = true>
  PDS_DEV U4 scratch_block(int j) const {
    const bool noise = j >= kResetBlocks && j < kResetBlocks + kResetNoiseBlocks;
    const uint32_t blk = j < kResetBlocks ? kBlkReset + (uint32_t)j
                         : (noise ? kBlkResetNoise + (uint32_t)(j - kResetBlocks)
                                  : kBlkLatRows + (uint32_t)(j - kResetBlocks - kResetNoiseBlocks));
    return philox4x32_10_or_7<X3>(env_id, tick_lo, tick_hi, blk, seed_lo, seed_hi, noise);
  }
  PDS_DEV void uniforms4(uint32_t b, float (&u)[4]) const { words_to_uniforms4(reset_block(b), u); }
  PDS_DEV U4 raw4(uint32_t b) const { return reset_block(b); }
  PDS_DEV void normals4(uint32_t b, float (&z)[4]) const { words_to_normals4(reset_block(b), z); }
  PDS_DEV void lat_normals4(uint32_t r, float (&z)[4]) const { words_to_normals4(lat_block(r), z); }
  PDS_DEV void noise8(uint32_t nb, float (&f)[8]) const { words_to_noise8(noise_block(nb), (int)(nb % (uint32_t)kObsCallBlocks), f); }
};
struct LdsWords {
  const U4 *slot;  // [kScratchBlocks]
  PDS_DEV U4 reset_block(uint32_t b) const { return slot[b]; }
  PDS_DEV U4 noise_block(uint32_t b) const { return slot[kResetBlocks + b]; }
  PDS_DEV U4 lat_block(uint32_t r) const { return slot[kResetBlocks + kResetNoiseBlocks + r]; }
  PDS_DEV void uniforms4(uint32_t b, float (&u)[4]) const { words_to_uniforms4(reset_block(b), u); }
  PDS_DEV U4 raw4(uint32_t b) const { return reset_block(b); }
  PDS_DEV void normals4(uint32_t b, float (&z)[4]) const { words_to_normals4(reset_block(b), z); }
  PDS_DEV void lat_normals4(uint32_t r, float (&z)[4]) const { words_to_normals4(lat_block(r), z); }
  PDS_DEV void noise8(uint32_t nb, float (&f)[8]) const { words_to_noise8(noise_block(nb), (int)(nb % (uint32_t)kObsCallBlocks), f); }
};
// Round 4: the blocks of a reset already CONVERTED, by the lanes that computed them (fill_reset_variates): the thread
// that evaluates the reset reads floats.  Per env: reset blocks 0..8 four floats each (block 6 -- it carries the
// integer draw of ref_offset -- as raw words), the six noise blocks eight floats each, the latency rows four each.
constexpr int kVarResetFloats = 4 * kResetBlocks;
constexpr int kVarNoiseFloats = 8 * kResetNoiseBlocks;
struct LdsVariates {
  const float *p;
  PDS_DEV void uniforms4(uint32_t b, float (&u)[4]) const {
    const float4 v = *reinterpret_cast<const float4 *>(p + 4 * b);
    u[0] = v.x; u[1] = v.y; u[2] = v.z; u[3] = v.w;
  }
  PDS_DEV U4 raw4(uint32_t b) const {
    const uint4 v = *reinterpret_cast<const uint4 *>(p + 4 * b);
    return U4{v.x, v.y, v.z, v.w};
  }
  PDS_DEV void normals4(uint32_t b, float (&z)[4]) const { uniforms4(b, z); }
  PDS_DEV void lat_normals4(uint32_t r, float (&z)[4]) const {
    const float4 v = *reinterpret_cast<const float4 *>(p + kVarResetFloats + kVarNoiseFloats + 4 * r);
    z[0] = v.x; z[1] = v.y; z[2] = v.z; z[3] = v.w;
  }
  PDS_DEV void noise8(uint32_t nb, float (&f)[8]) const {
    const float4 a = *reinterpret_cast<const float4 *>(p + kVarResetFloats + 8 * nb);
    const float4 b = *reinterpret_cast<const float4 *>(p + kVarResetFloats + 8 * nb + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  }
};
constexpr int kRawWordsBlock = 6;  // ref_offset = mulhi(word 3, num_ref_points): kept as integers

// number of leading scratch slots a variant can touch (bounds the cooperative fill loop)
template <class V>
PDS_DEV constexpr int scratch_blocks_used() {
  return V::LAT ? kScratchBlocks : (V::ON ? kResetBlocks + kResetNoiseBlocks : kResetBlocks);
}

// which of the scratch blocks a variant consumes (the cooperative fill skips the others)
template <class V>
PDS_DEV constexpr bool block_needed(int j) {
  if (j >= kResetBlocks + kResetNoiseBlocks) return V::LAT && V::TASK != PDS_TASK_TAKEOFF;
  if (j >= kResetBlocks) return V::ON;
  if (j <= 1) return true;
  if (j <= 4) return V::TASK != PDS_TASK_TAKEOFF;
  if (j <= 6) return V::DR || V::TASK == PDS_TASK_CIRCLE;
  return V::MOTOR && V::DR;
}

// In-kernel reset sampler; restated draw for draw by oracle/phoenix_oracle.c
// po_philox_reset_sample.  Ranges: envs/hover.py:201-228, envs/circle.py:225-257,
// envs/takeoff.py:186-191, envs/base.py:250-287.
template <class V, class SRC>
PDS_DEV void sample_philox(const Consts &k, const SRC &src, Sample &s) {
  constexpr int TASK = V::TASK;
  constexpr float D2R = kPi / 180.f;
  float pos_lim, rp_lim, yaw_lim, vel_lim, w_lim, wz_lim;
  if (TASK == PDS_TASK_HOVER) {
    pos_lim = 0.25f; rp_lim = kPi / 6.f; yaw_lim = 2.f * kPi; vel_lim = 0.1f; w_lim = 200.f * D2R; wz_lim = 20.f * D2R;
  } else if (TASK == PDS_TASK_CIRCLE) {
    pos_lim = 0.05f; rp_lim = 20.f * D2R; yaw_lim = 0.1f * kPi; vel_lim = 0.1f; w_lim = 50.f * D2R; wz_lim = 20.f * D2R;
  } else {
    pos_lim = 0.25f; rp_lim = 0.f; yaw_lim = kPi; vel_lim = 0.f; w_lim = 0.f; wz_lim = 0.f;
  }
  float u0[4], u1[4];
  src.uniforms4(0u, u0);
  src.uniforms4(1u, u1);
  s.pos[0] = urange_u(u0[0], -pos_lim, pos_lim);
  s.pos[1] = urange_u(u0[1], -pos_lim, pos_lim);
  s.pos[2] = (TASK == PDS_TASK_TAKEOFF) ? 0.f : urange_u(u0[2], -pos_lim, pos_lim);
  s.rpy[0] = urange_u(u0[3], -rp_lim, rp_lim);
  s.rpy[1] = urange_u(u1[0], -rp_lim, rp_lim);
  s.rpy[2] = urange_u(u1[1], -yaw_lim, yaw_lim);
  s.vel[0] = urange_u(u1[2], -vel_lim, vel_lim);
  s.vel[1] = urange_u(u1[3], -vel_lim, vel_lim);
  if (TASK != PDS_TASK_TAKEOFF) {
    float u2[4], z3[4], z4[4];
    src.uniforms4(2u, u2);
    src.normals4(3u, z3);
    src.normals4(4u, z4);
    s.vel[2] = urange_u(u2[0], -vel_lim, vel_lim);
    s.w[0] = urange_u(u2[1], -w_lim, w_lim);
    s.w[1] = urange_u(u2[2], -w_lim, w_lim);
    s.w[2] = urange_u(u2[3], -wz_lim, wz_lim);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s.mx[i] = k.hover_x + 0.02f * z3[i];
      s.act[i] = k.hover_action + 0.02f * z4[i];
    }
    if constexpr (V::LAT) {  // the other rows of np.random.normal(HOVER_ACTION, 0.02, (buf_size, 4)), hover.py:226-228
#pragma unroll
      for (int r = 0; r < kMaxLatSteps - 1; ++r) {
        if (r < k.lat_steps - 1) {
          float y[4];
          src.lat_normals4((uint32_t)r, y);
#pragma unroll
          for (int i = 0; i < 4; ++i) s.abuf[r][i] = k.hover_action + 0.02f * y[i];
        }
      }
    }
  } else {
    s.vel[2] = 0.f; s.w[0] = s.w[1] = s.w[2] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { s.mx[i] = 0.f; s.act[i] = 0.f; }
  }
  s.ref_offset = 0;
  if (V::DR || TASK == PDS_TASK_CIRCLE) {
    const float f = k.dr;
    float u5[4];
    src.uniforms4(5u, u5);
    const U4 r6 = src.raw4((uint32_t)kRawWordsBlock);
#define PDS_DRV(u, d) urange_u((u), (d) - f * (d), (d) + f * (d))
    s.dt = PDS_DRV(u5[0], k.dt_nom);
    s.m = PDS_DRV(u5[1], k.M_nom);
    s.J[0] = PDS_DRV(u5[2], k.Jx_nom);
    s.J[1] = PDS_DRV(u5[3], k.Jy_nom);
    s.J[2] = PDS_DRV(u01(r6.x), k.Jz_nom);
    s.ftf1 = PDS_DRV(u01(r6.z), k.ftf1_nom);
    s.ref_offset = (int)__umulhi(r6.w, (uint32_t)k.ref_points);  // randint(0, num_ref_points), circle.py:225
    if (V::MOTOR && V::DR) {
      float u7[4], u8[4];
      src.uniforms4(7u, u7);
      src.uniforms4(8u, u8);
      s.T[0] = PDS_DRV(u7[0], k.mtc); s.T[1] = PDS_DRV(u7[1], k.mtc);
      s.T[2] = PDS_DRV(u7[2], k.mtc); s.T[3] = PDS_DRV(u7[3], k.mtc);
      s.t2w[0] = PDS_DRV(u8[0], k.t2w); s.t2w[1] = PDS_DRV(u8[1], k.t2w);
      s.t2w[2] = PDS_DRV(u8[2], k.t2w); s.t2w[3] = PDS_DRV(u8[3], k.t2w);
    }
#undef PDS_DRV
  }
}

template <bool LAT>
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
  if constexpr (LAT) {
#pragma unroll
    for (int r = 0; r < kMaxLatSteps - 1; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) s.abuf[r][i] = row[PDS_S_ACTION_BUF + 4 * r + i];
  }
}

// ---- sensor noise -------------------------------------------------------------------------------
// 24 standard variates of one add_noise call from three Philox blocks (layout: words_to_noise8): normals n[0..17] =
// position 3, velocity 3, angle 3, gyro bias walk 3, gyro random walk 3, gyro turn-on 3; uniforms = position 3, angle 3.
// (f0, f1, f2: the three blocks of the call as words_to_noise8 converts them; NBLK == 2: the gyro normals are not wanted)
template <int NBLK = kObsCallBlocks>
PDS_DEV void obs_noise_from_blocks(const float (&f0)[8], const float (&f1)[8], const float (&f2)[8], ObsNoise &n) {
  float z[18];
#pragma unroll
  for (int i = 0; i < 8; ++i) { z[i] = f0[i]; z[10 + i] = NBLK > 2 ? f2[i] : 0.f; }
  z[8] = f1[0]; z[9] = f1[1];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    n.pos_z[i] = z[i]; n.vel_z[i] = z[3 + i]; n.th_z[i] = z[6 + i];
    n.bias_z[i] = z[9 + i]; n.rw_z[i] = z[12 + i]; n.to_z[i] = z[15 + i];
  }
  n.pos_u[0] = f1[2]; n.pos_u[1] = f1[3];
  n.pos_u[2] = f1[4]; n.th_u[0] = f1[5];
  n.th_u[1] = f1[6]; n.th_u[2] = f1[7];
}

template <int NBLK = kObsCallBlocks>
PDS_DEV void obs_noise_philox(uint32_t env_id, const RngKey &a, uint32_t blk0, ObsNoise &n) {
  float f[kObsCallBlocks][8];
#pragma unroll
  for (int b = 0; b < kObsCallBlocks; ++b) {
    if (b < NBLK) {
      words_to_noise8(philox4x32_7(env_id, a.tick_lo, a.tick_hi, blk0 + (uint32_t)b, a.seed_lo, a.seed_hi), b, f[b]);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) f[b][i] = 0.f;
    }
  }
  obs_noise_from_blocks<NBLK>(f[0], f[1], f[2], n);
}

template <class SRC>
PDS_DEV void obs_noise_reset(const SRC &src, int call, ObsNoise &n) {
  float f[kObsCallBlocks][8];
#pragma unroll
  for (int b = 0; b < kObsCallBlocks; ++b) src.noise8((uint32_t)(kObsCallBlocks * call + b), f[b]);
  obs_noise_from_blocks(f[0], f[1], f[2], n);
}

PDS_DEV void obs_noise_load(const float *p, ObsNoise &n) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    n.pos_z[i] = p[PDS_N_OBS_POS_Z + i]; n.pos_u[i] = p[PDS_N_OBS_POS_U + i];
    n.vel_z[i] = p[PDS_N_OBS_VEL_Z + i]; n.bias_z[i] = p[PDS_N_OBS_BIAS + i];
    n.rw_z[i] = p[PDS_N_OBS_RW + i]; n.to_z[i] = p[PDS_N_OBS_TO + i];
    n.th_z[i] = p[PDS_N_OBS_TH_Z + i]; n.th_u[i] = p[PDS_N_OBS_TH_U + i];
  }
}

// SensorNoise.add_noise_to_omega (envs/sensors.py:121-134) followed by the gyro low-pass
// LowPassFilter.apply with T_s/T = 0.5 (envs/base.py:109-110, envs/utils.py:76-79).
PDS_DEV void gyro_update(const Consts &k, const EnvRegs &e, const float bz[3], const float rz[3],
                         const float tz[3], NoiseState &ns) {
  const float w[3] = {e.wx, e.wy, e.wz};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    ns.bias[i] = k.gyro_pi * ns.bias[i] + k.gyro_sb * bz[i];
    const float om = ((w[i] + ns.bias[i]) + k.gyro_rw * rz[i]) + k.gyro_to * tz[i];
    ns.lpf[i] = 0.5f * ns.lpf[i] + 0.5f * om;
  }
}

// SensorNoise.add_noise (envs/sensors.py:75-118) on the true state + quaternion of the noisy
// Euler angles (envs/hover.py:146); updates gyro bias / low-pass, returns the noisy observation.
PDS_DEV void sensor_observe(const Consts &k, const EnvRegs &e, const ObsNoise &n, NoiseState &ns, NoisyObs &o) {
  const float p[3] = {e.px, e.py, e.pz}, v[3] = {e.vx, e.vy, e.vz}, r[3] = {e.roll, e.pitch, e.yaw};
  float pn[3], vn[3], rn[3];
  const float lo[3] = {-kPi, -kHalfPi, -kPi}, hi[3] = {kPi, kHalfPi, kPi};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    pn[i] = p[i] + (k.pos_std * n.pos_z[i] + (-k.pos_unif + (2.f * k.pos_unif) * n.pos_u[i]));
    vn[i] = v[i] + k.vel_std * n.vel_z[i];
    rn[i] = clampf(r[i] + (k.q_std * n.th_z[i] + (-k.q_unif + (2.f * k.q_unif) * n.th_u[i])), lo[i], hi[i]);
  }
  gyro_update(k, e, n.bias_z, n.rw_z, n.to_z, ns);
  const Quat q = quat_from_euler(rn[0], rn[1], rn[2]);
  o.x = pn[0]; o.y = pn[1]; o.z = pn[2];
  o.qx = q.x; o.qy = q.y; o.qz = q.z; o.qw = q.w;
  o.vx = vn[0]; o.vy = vn[1]; o.vz = vn[2];
}

// The noisy observation o(k) that the previous env.step (or the reset) returned, REGENERATED instead of kept in
// memory (observation-noise variants without the Kalman hold): its position / attitude / velocity part is a pure
// function of the true state the env had then -- which is the state the next step loads -- and of the standard
// variates of that call, which are Philox blocks of the PREVIOUS tick: kBlkObsNoise.. for a step, the second
// add_noise call of the reset (kBlkResetNoise + 3..5) when the env was reset in that tick (step counter == 0).  Same
// device functions, same inputs => the same bits the previous launch wrote into the second half of its row.  (The
// filtered gyro of o(k) is the low-pass state, which stays in memory.)  ~400 vector instructions per 64 envs for 80 B
// per env-step less traffic: Hover 2^20 noise + DR 85.7 -> 79.9 us (profiles/r04_ab_regen.txt).
PDS_DEV void regen_kept_obs(const Consts &k, uint32_t env_id, const RngKey &now, bool after_reset, const EnvRegs &e,
                            NoisyObs &o) {
  RngKey prev = now;
  prev.tick_lo = now.tick_lo - 1u;
  if (now.tick_lo == 0u) prev.tick_hi = now.tick_hi - 1u;
  ObsNoise n;
  obs_noise_philox<2>(env_id, prev, after_reset ? kBlkResetNoise + (uint32_t)kObsCallBlocks : kBlkObsNoise, n);  // (no gyro normals)
  NoiseState unused;
#pragma unroll
  for (int j = 0; j < 3; ++j) { unused.bias[j] = 0.f; unused.lpf[j] = 0.f; }
  sensor_observe(k, e, n, unused, o);  // (its gyro part is dead code here)
}

// ---- observation rows ---------------------------------------------------------------------------
// noise-free o (envs/agents.py:339-348 get_state; envs/circle.py:173-177; envs/takeoff.py:146-147)
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

// noisy o (envs/hover.py:133-159, envs/circle.py:135-171, envs/takeoff.py:113-143):
// [xyz_n, Q(rpy_n), vel_n, lpf(omega_n) (, last_action: TakeOff) (, target - xyz_n: Circle/TakeOff)]
template <int TASK>
PDS_DEV void write_noisy_half(float *row, const NoisyObs &o, const float lpf[3], const float4 &last_action,
                              float tx, float ty, float tz, const float4 &hist_action) {
  int n = 0;
  row[n++] = o.x; row[n++] = o.y; row[n++] = o.z;
  row[n++] = o.qx; row[n++] = o.qy; row[n++] = o.qz; row[n++] = o.qw;
  row[n++] = o.vx; row[n++] = o.vy; row[n++] = o.vz;
  row[n++] = lpf[0]; row[n++] = lpf[1]; row[n++] = lpf[2];
  if (TASK == PDS_TASK_TAKEOFF) {
    row[n++] = last_action.x; row[n++] = last_action.y; row[n++] = last_action.z; row[n++] = last_action.w;
  }
  if (TASK != PDS_TASK_HOVER) {
    row[n++] = tx - o.x; row[n++] = ty - o.y; row[n++] = tz - o.z;
  }
  row[n++] = hist_action.x; row[n++] = hist_action.y; row[n++] = hist_action.z; row[n++] = hist_action.w;
}

// ---- reset ----------------------------------------------------------------------------------------
// DroneBaseEnv.reset (envs/base.py:382-431) for one env: task_specific_reset, domain
// randomisation, the Bullet pose/velocity round trip of update_information
// (envs/agents.py:434-453: rpy = Euler(quat), omega = R^T R^T omega_sampled).
struct LatRows {  // rows 0..B-2 of drone.action_buffer after a reset (row B-1 == u0 == drone.last_action)
  float4 r[kMaxLatSteps - 1];
};

template <class V>
PDS_DEV void reset_env(const Consts &k, const float2 *ref_lds, const Sample &s, EnvRegs &e, Quat &q,
                       float4 &u0, float4 &mx, Params &par, uint32_t &ctr, LatRows &rows) {
  constexpr int TASK = V::TASK;
  float px = k.init_xyz[0], py = k.init_xyz[1], pz = k.init_xyz[2];
  float vx = k.init_vel[0], vy = k.init_vel[1], vz = k.init_vel[2];
  float w0 = k.init_w[0], w1 = k.init_w[1], w2 = k.init_w[2];
  float r0 = k.init_rpy[0], r1 = k.init_rpy[1], r2 = k.init_rpy[2];
  int ref_offset = (TASK == PDS_TASK_CIRCLE) ? (int)ctr_off(ctr) : 0;  // kept when no reset distribution
  u0 = make_float4(0.f, 0.f, 0.f, 0.f);  // drone.reset(): envs/agents.py:380-386
  mx = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (V::LAT) {
#pragma unroll
    for (int r = 0; r < kMaxLatSteps - 1; ++r) rows.r[r] = make_float4(0.f, 0.f, 0.f, 0.f);  // agents.py:385
  }
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
      if constexpr (V::LAT) {  // action_buffer = clip(normal(HOVER_ACTION, .02, (B, 4))), last_action = its last row
#pragma unroll
        for (int r = 0; r < kMaxLatSteps - 1; ++r)
          rows.r[r] = make_float4(clampf(s.abuf[r][0], -1.f, 1.f), clampf(s.abuf[r][1], -1.f, 1.f),
                                  clampf(s.abuf[r][2], -1.f, 1.f), clampf(s.abuf[r][3], -1.f, 1.f));
      }
    }
  }
  if (TASK == PDS_TASK_TAKEOFF) {  // envs/takeoff.py:209-212 (unconditional)
    mx = make_float4(0.f, 0.f, 0.f, 0.f);
    u0 = make_float4(-1.f, -1.f, -1.f, -1.f);
    if constexpr (V::LAT) {
#pragma unroll
      for (int r = 0; r < kMaxLatSteps - 1; ++r) rows.r[r] = u0;  // action_buffer[:] = -1
    }
  }
  default_params(k, par);
  if (V::DR) {  // envs/base.py:259-287
    par.dt = s.dt; par.m = s.m; par.Jx = s.J[0]; par.Jy = s.J[1]; par.Jz = s.J[2]; par.ftf1 = s.ftf1;
    if (V::MOTOR) {  // envs/agents.py:208-224 (K uses the hard-coded 0.028)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float T = fmaxf(s.T[i], par.dt);
        par.A[i] = 1.0f - par.dt * fast_rcp(T);  // v_rcp_f32: 1 ulp (an IEEE division is ~12 instructions)
        par.K[i] = (0.028f * k.G * s.t2w[i]) * 0.25f;
      }
    }
  }
  q = quat_from_euler(r0, r1, r2);
  float R[9];
  matrix_from_quat(q, R);
  // update_information reads the pose back from Bullet (envs/agents.py:443): the quaternion comes back
  // through btTransform's 3x3 basis (btMatrix3x3::setRotation / getRotation, see oracle/phoenix_oracle.c
  // po_bullet_readback_quat), i.e. as +-Q(sampled rpy) with the sign Bullet's extraction gives it:
  // w > 0 when the trace is positive, else the component of the largest diagonal element positive.
  // Only the sign is taken from that rule (the f32 value of Q is closer to the reference's f64 quaternion
  // than an f32 matrix round trip would be).
  uint32_t sign;
  {
    // on the unit quaternion itself: trace = 4 w^2 - 1, and diag_i - diag_j = 2 (q_i^2 - q_j^2)
    const float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
    float lead = q.w;
    if (!(4.f * q.w * q.w > 1.f)) lead = xx < yy ? (yy < zz ? q.z : q.y) : (xx < zz ? q.z : q.x);
    sign = lead < 0.f ? 1u : 0u;
    if (sign) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
  }
  // bc.resetBaseVelocity(R^T w) then update_information: R^T (R^T w)
  const float a0 = R[0] * w0 + R[3] * w1 + R[6] * w2;
  const float a1 = R[1] * w0 + R[4] * w1 + R[7] * w2;
  const float a2 = R[2] * w0 + R[5] * w1 + R[8] * w2;
  e.wx = R[0] * a0 + R[3] * a1 + R[6] * a2;
  e.wy = R[1] * a0 + R[4] * a1 + R[7] * a2;
  e.wz = R[2] * a0 + R[5] * a1 + R[8] * a2;
  e.px = px; e.py = py; e.pz = pz; e.vx = vx; e.vy = vy; e.vz = vz;
  // rpy = Euler(quat) (envs/agents.py:446).  The state stores the wrapped Euler angles and every later
  // quaternion is Q(rpy) (envs/physics.py:179); until the first step the observation carries the
  // read-back quaternion, so remember whether Q(wrapped rpy) has the opposite sign.
  if (fabsf(r0) < 1.55f && fabsf(r1) < 1.55f) {
    // every sampled attitude lands here: roll/pitch inside the principal range, so
    // Euler(+-Q(r,p,y)) == (r, p, y - 2 pi k) and Q flips sign once per 2 pi of yaw
    const float kk = rintf(r2 * 0.15915494309189533577f);
    float yw = fmaf(-kk, 6.2831854820251465f, r2);
    yw = fmaf(kk, 1.7484555e-7f, yw);
    e.roll = r0; e.pitch = r1; e.yaw = yw;
    sign ^= ((uint32_t)(int)kk) & 1u;
  } else {  // init_rpy overrides near / beyond gimbal lock: the general pybullet formulas
    euler_from_quat(q, e.roll, e.pitch, e.yaw);
    const Quat qw = quat_from_euler(e.roll, e.pitch, e.yaw);
    sign = (qw.x * q.x + qw.y * q.y + qw.z * q.z + qw.w * q.w) < 0.f ? 1u : 0u;
  }
  ctr = ctr_pack(0u, sign, (uint32_t)ref_offset, 0u);  // iteration = 0, action_idx = 0
}

// Result of one reset, split in a pure-compute half and a store half so that the deferred
// auto-reset drain can do its arithmetic while the wave's earlier stores are still draining.
struct ResetOut {
  EnvRegs e;
  Quat q;
  float4 u0, mx;
  Params par;
  uint32_t ctr;
  LatRows lat;       // LAT
  NoiseState ns;     // ON: gyro bias / low-pass after the two observation calls of reset
  NoisyObs oa, ob;   // ON: the two noisy observations (history fill / compute_history)
  float lpf_a[3];    // ON: filtered gyro of the first one
};

// `stale_w`: drone.rpy_dot BEFORE the reset -- the reference re-initialises the gyro low-pass with
// it (envs/base.py:411 runs before update_information); `bias`: persisting gyro bias.
template <class V, class SRC>
PDS_DEV void reset_compute(const StepArgs &a, const float2 *ref_lds, const SRC &src, uint32_t ctr_old,
                           const float *sample_row, const float stale_w[3], const float bias[3], ResetOut &r) {
  Sample s;
  if (sample_row != nullptr) sample_load<V::LAT>(sample_row, s);
  else sample_philox<V>(a.k, src, s);
  r.ctr = ctr_old;
  reset_env<V>(a.k, ref_lds, s, r.e, r.q, r.u0, r.mx, r.par, r.ctr, r.lat);
  if (V::ON) {
    ObsNoise n0, n1;
    if (sample_row != nullptr) {
      obs_noise_load(sample_row + PDS_S_NOISE_CALL0, n0);
      obs_noise_load(sample_row + PDS_S_NOISE_CALL1, n1);
    } else {
      obs_noise_reset(src, 0, n0);
      obs_noise_reset(src, 1, n1);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { r.ns.lpf[j] = stale_w[j]; r.ns.bias[j] = bias[j]; }
    sensor_observe(a.k, r.e, n0, r.ns, r.oa);   // obs = compute_observation(), envs/base.py:419
#pragma unroll
    for (int j = 0; j < 3; ++j) r.lpf_a[j] = r.ns.lpf[j];
    sensor_observe(a.k, r.e, n1, r.ns, r.ob);   // compute_history(), envs/base.py:429
  }
}

// Writes the complete state, parameters and (optionally) the observation row [o0,u0,o0',u0]
// (envs/base.py:417-431) of a reset env straight to HBM.  Rows are strided -> only for the explicit
// reset kernel and the deferred auto-reset drain, never on the per-step stream.
// `oh_mode` 1: the kept noisy observation goes to oh0-2 and the counter word says so (kCtrOhBit); post_reset_kernel, mirroring what
// the in-place reset of the step kernel in front of it leaves behind: 0 (a REGENERATING step kernel, csrc/pds_step.h
// regen_obs_variant): neither -- the next step regenerates it; 2 (Kalman-hold kernels: they always keep it in oh0-2): stored, no flag
template <class V>
PDS_DEV void reset_store(const StepArgs &a, const float2 *ref_lds, long long i, const ResetOut &r, int oh_mode = 1) {
  const bool keep_oh = oh_mode != 0;
  constexpr int TASK = V::TASK;
  constexpr int D = V::D;
  const EnvRegs &e = r.e;
  a.st.s0[i] = make_float4(e.px, e.py, e.pz, e.vx);
  a.st.s1[i] = make_float4(e.vy, e.vz, e.roll, e.pitch);
  a.st.s2[i] = make_float4(e.yaw, e.wx, e.wy, e.wz);
  a.st.hist[0][i] = r.u0;
  a.st.hist[1][i] = r.u0;
  a.st.ctr[i] = (V::ON && oh_mode == 1) ? (r.ctr | kCtrOhBit) : r.ctr;  // (ON: the kept observation is written below, not regenerated)
  if (a.st.pid0 != nullptr) {  // control.reset(): envs/agents.py:379, envs/control.py:178-187, 279-287
    a.st.pid0[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    a.st.pid1[i] = make_float2(0.f, 0.f);
    if (a.st.pid2 != nullptr) {
      a.st.pid2[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      a.st.pid3[i] = make_float2(0.f, 0.f);
    }
  }
  if (V::MOTOR) a.st.mx[i] = r.mx;
  if constexpr (V::LAT) {
#pragma unroll
    for (int b = 0; b < kMaxLatSteps; ++b)
      if (b < a.k.lat_steps) {
        // (by value: a conditional between the two lvalues would select a POINTER and push `r` into scratch memory)
        float4 v = r.lat.r[b < kMaxLatSteps - 1 ? b : 0];
        if (b == a.k.lat_steps - 1 || b == kMaxLatSteps - 1) v = r.u0;
        a.st.lat[(long long)b * a.n + i] = v;
      }
  }
  if (V::DR) {
    a.st.par0[i] = make_float4(r.par.dt, r.par.m, r.par.Jx, r.par.Jy);
    a.st.par1[i] = make_float2(r.par.Jz, r.par.ftf1);
    if (V::MOTOR) {
      a.st.mA[i] = make_float4(r.par.A[0], r.par.A[1], r.par.A[2], r.par.A[3]);
      a.st.mK[i] = make_float4(r.par.K[0], r.par.K[1], r.par.K[2], r.par.K[3]);
    }
  }
  if (V::ON) {
    a.st.nz0[i] = make_float4(r.ns.bias[0], r.ns.bias[1], r.ns.bias[2], r.ns.lpf[0]);
    a.st.nz1[i] = make_float2(r.ns.lpf[1], r.ns.lpf[2]);
    if (keep_oh) {
      a.st.oh0[i] = make_float4(r.ob.x, r.ob.y, r.ob.z, r.ob.qx);
      a.st.oh1[i] = make_float4(r.ob.qy, r.ob.qz, r.ob.qw, r.ob.vx);
      a.st.oh2[i] = make_float2(r.ob.vy, r.ob.vz);
    }
  }
  if (a.obs != nullptr) {
    float rowbuf[D];
    float tx, ty, tz;
    target_at<TASK>(a.k, ref_lds, target_index<TASK>(0, a.k.agg, (int)ctr_off(r.ctr)), tx, ty, tz);
    if (V::ON) {
      write_noisy_half<TASK>(rowbuf, r.oa, r.lpf_a, r.u0, tx, ty, tz, r.u0);
      write_noisy_half<TASK>(rowbuf + V::O + 4, r.ob, r.ns.lpf, r.u0, tx, ty, tz, r.u0);
    } else {
      write_obs_half<TASK>(rowbuf, e, r.q, r.u0, tx, ty, tz, r.u0);
      write_obs_half<TASK>(rowbuf + V::O + 4, e, r.q, r.u0, tx, ty, tz, r.u0);
    }
    float2 *dst = reinterpret_cast<float2 *>(a.obs + i * D);
#pragma unroll
    for (int j = 0; j < D / 2; ++j) dst[j] = make_float2(rowbuf[2 * j], rowbuf[2 * j + 1]);
  }
}

// Pass over the envs a wave queued for auto-reset (queue entry = lane of the env in the wave's tile | ref_offset << 6).
// Groups of 8 lanes serve one queued env: every lane computes one (or a few) of the env's Philox
// blocks into an LDS scratch (the wave's observation tile, free after its flush), then the group's
// first lane assembles the sample from the scratch and finishes the reset.  The arithmetic touches no
// global memory, so it overlaps with the wave's outstanding stores; those must have completed
// (s_waitcnt vmcnt(0)) before the same addresses are overwritten.  (With observation noise the reset
// needs the terminal body rates and the gyro bias of the env: they come out of the registers of the
// lane that stepped it, by ds_bpermute, not back from memory.)
constexpr int kLanesPerReset = 8;
constexpr int kResetsPerPass = kWave / kLanesPerReset;

// Philox blocks of up to kResetsPerPass queued envs (lane ids in bits 0..5 of entries[0..cnt)), computed L lanes
// per env into `scratch` (LdsWords layout, slot s = entry s).  Wave-uniform control flow; the caller fences.
// STRIDE: U4 slots per env in `scratch` (>= scratch_blocks_used<V>(): the drain lays its scratch over the free tile with
// kScratchBlocks, the in-register resets size theirs with the blocks the variant really uses)
template <class V, int STRIDE = kScratchBlocks>
PDS_DEV void fill_reset_scratch(const StepArgs &a, const RngKey &rk, const uint32_t *entries, int cnt, int lane,
                                long long wave_base, U4 *scratch) {
  static_assert(STRIDE >= scratch_blocks_used<V>(), "scratch slots per env");
  constexpr int NB = scratch_blocks_used<V>();
  constexpr int L = NB <= 8 ? 8 : (NB <= 16 ? 16 : 32);  // lanes per env in a Philox round
  constexpr int EPR = kWave / L;                           // envs per Philox round
  const int g = lane / L, j = lane % L;
  for (int sub = 0; sub < cnt; sub += EPR) {
    const int slot = sub + g;  // env of the pass this lane computes a block for
    const bool on = slot < cnt;
    uint32_t ent = 0;
    if (on) ent = entries[slot];
    const uint32_t env_id = (uint32_t)(a.env_id_base + (unsigned long long)(wave_base + (long long)(ent & 63u)));
    const DirectWords dw(env_id, rk);
    bool need = false;
#pragma unroll
    for (int c = 0; c < NB; ++c) need = need || (c == j && block_needed<V>(c));
    if (on && need) scratch[slot * STRIDE + j] = dw.template scratch_block<V::ON || V::LAT>(j);  // (see philox4x32_10_or_7)
  }
}

// Floats per env of the converted scratch (LdsVariates layout)
template <class V>
constexpr int variates_floats() {
  return V::LAT ? kVarResetFloats + kVarNoiseFloats + 4 * kLatRowBlocks : (V::ON ? kVarResetFloats + kVarNoiseFloats : kVarResetFloats);
}

// Philox block j of a reset (DirectWords::scratch_block order) -> the standard variates of the block's role, at their place in an
// env's LdsVariates image `dst`
PDS_DEV void store_block_variates(const U4 &w, int j, float *dst) {
  if (j < kResetBlocks) {
    float f[4];
    if (j == 3 || j == 4) {
      words_to_normals4(w, f);
    } else if (j == kRawWordsBlock) {
      f[0] = __uint_as_float(w.x); f[1] = __uint_as_float(w.y); f[2] = __uint_as_float(w.z); f[3] = __uint_as_float(w.w);
    } else {
      words_to_uniforms4(w, f);
    }
    *reinterpret_cast<float4 *>(dst + 4 * j) = make_float4(f[0], f[1], f[2], f[3]);
  } else if (j < kResetBlocks + kResetNoiseBlocks) {
    const int nb = j - kResetBlocks;
    float f[8];
    words_to_noise8(w, nb % kObsCallBlocks, f);
    *reinterpret_cast<float4 *>(dst + kVarResetFloats + 8 * nb) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4 *>(dst + kVarResetFloats + 8 * nb + 4) = make_float4(f[4], f[5], f[6], f[7]);
  } else {
    float f[4];
    words_to_normals4(w, f);
    *reinterpret_cast<float4 *>(dst + kVarResetFloats + kVarNoiseFloats + 4 * (j - kResetBlocks - kResetNoiseBlocks)) =
        make_float4(f[0], f[1], f[2], f[3]);
  }
}

// fill_reset_scratch + conversion: lane j of an env's group computes Philox block j AND turns its four words into the
// standard variates of the block's role (words_to_*), so that the one or two lanes that evaluate the reset read floats
// instead of running ~20 Box-Muller pairs and ~30 uniform conversions in a row (~260 of the ~1100 vector instructions
// of a reset with observation noise).  `var`: [envs of the pass][variates_floats<V>()].
template <class V>
PDS_DEV void fill_reset_variates(const StepArgs &a, const RngKey &rk, const uint32_t *entries, int cnt, int lane,
                                 long long wave_base, float *var) {
  constexpr int NB = scratch_blocks_used<V>();
  constexpr int L = NB <= 8 ? 8 : (NB <= 16 ? 16 : 32);  // lanes per env in a Philox round
  constexpr int EPR = kWave / L;                           // envs per Philox round
  constexpr int VF = variates_floats<V>();
  const int g = lane / L, j = lane % L;
  for (int sub = 0; sub < cnt; sub += EPR) {
    const int slot = sub + g;  // env of the pass this lane computes a block for
    const bool on = slot < cnt;
    uint32_t ent = 0;
    if (on) ent = entries[slot];
    const uint32_t env_id = (uint32_t)(a.env_id_base + (unsigned long long)(wave_base + (long long)(ent & 63u)));
    const DirectWords dw(env_id, rk);
    bool need = false;
#pragma unroll
    for (int c = 0; c < NB; ++c) need = need || (c == j && block_needed<V>(c));
    if (on && need) store_block_variates(dw.scratch_block(j), j, var + slot * VF);
  }
}

template <class V>
PDS_DEV void drain_reset_queue(const StepArgs &a, const RngKey &rk, const float2 *ref_lds, const uint32_t *queue,
                               int qcount, int lane, long long wave_base, float *tile, const float (&own_w)[3],
                               const float (&own_bias)[3]) {
  // Two costs per pass, both paid by the whole wave whatever the number of active lanes: one Philox (~650
  // cycles: 10 rounds x ~14 vector instructions) per round of block computations, and one evaluation of the
  // reset (~2000 cycles) by the owner lanes.  So: up to 8 envs per pass share ONE reset evaluation, and their
  // Philox blocks are computed L lanes per env (8, 16 or 32 for the 9 / 15 / 22 blocks a variant can need),
  // i.e. in as few rounds as the number of queued envs allows (one round for up to 64 / L envs).
  static_assert(kResetsPerPass * kScratchBlocks * 16 <= kHalfTileRows * V::D * 4, "scratch must fit in the wave's tile");
  U4 *scratch = reinterpret_cast<U4 *>(tile);
  const int og = lane / kLanesPerReset;  // owner lanes: the first of every 8
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int base = 0; base < qcount; base += kResetsPerPass) {
    const int cnt = min(kResetsPerPass, qcount - base);  // wave-uniform
    fill_reset_scratch<V>(a, rk, queue + base, cnt, lane, wave_base, scratch);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool owner = og < cnt && (lane % kLanesPerReset) == 0;
    uint32_t ent = 0;
    if (owner) ent = queue[base + og];
    const long long i = wave_base + (long long)(ent & 63u);
    ResetOut r;
    float stale_w[3] = {0.f, 0.f, 0.f}, bias[3] = {0.f, 0.f, 0.f};
    if (V::ON) {
      // the gyro rates and the gyro bias the finished env had after this step (the reference re-initialises the
      // low-pass with them): out of the registers of the lane that stepped it -- they are what that lane has
      // just stored, and reading them back from memory would put a wait for the wave's stores and a load in
      // front of the evaluation
      const int src = (int)(ent & 63u);
#pragma unroll
      for (int j = 0; j < 3; ++j) { stale_w[j] = __shfl(own_w[j], src); bias[j] = __shfl(own_bias[j], src); }
    }
    if (owner) {
      const LdsWords lw{scratch + og * kScratchBlocks};
      reset_compute<V>(a, ref_lds, lw, ctr_pack(0u, 0u, ent >> 6), nullptr, stale_w, bias, r);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (owner) reset_store<V>(a, ref_lds, i, r);
    __builtin_amdgcn_wave_barrier();  // scratch is refilled by the next pass
  }
}

// ---- auto-reset merged into the step's own stores (variants without observation noise) -----------
// The finished envs of a wave are reset BEFORE the wave stores: the fresh state replaces the
// terminal one in the registers of the lane that owns the env, so it leaves through the same
// coalesced stores and the same observation-tile flush as every other env -- no scattered partial-line
// writes over lines the wave has just written, no wait for those stores.  (Measured on Hover 2^20:
// the scattered observation-row rewrite of the deferred drain alone cost 1.0 us of 60.)
// Groups of 8 lanes serve one finished env: lane b of the group computes Philox block b into a small
// per-wave LDS scratch (8 envs x 9 blocks x 16 B next to the observation tile, which is still
// occupied); then each finished lane reads ITS env's words back (ds_read_b128) and evaluates the
// reset under its own exec mask.  Round 1 moved the words and the results through ~57 dependent
// ds_bpermute per pass with the reset evaluated redundantly by all 8 lanes of a group; the s_memtime
// stamps (profiles/r02_stamps_*.txt) showed ~3000 cycles per wave with a finished env for that,
// on the critical path of every launch that fits the chip in one round.
constexpr int kMergedScratchBlocks = kResetBlocks;                                   // per env
constexpr int kMergedScratchU4 = kResetsPerPass * kMergedScratchBlocks;              // per wave

// `mine`: this lane's env finished; `pos`: its rank among the wave's `count` finished lanes, whose
// lane ids are in `lanes[0..count)`.  Must be called in wave-uniform control flow.
template <class V>
PDS_DEV void reset_in_registers(const StepArgs &a, const RngKey &rk, const float2 *ref_lds, const uint32_t *lanes,
                                int count, bool mine, int pos, int lane, long long wave_base, int ref_offset,
                                U4 *scratch, EnvRegs &e, Quat &q, float4 &u0, float4 &mx, Params &par, uint32_t &ctr) {
  static_assert(!V::ON && !V::LAT, "observation-noise / latency variants use the deferred drain or the inline reset");
  // One Philox4x32-10 block costs ~650 cycles of the wave's VALU time (10 rounds x ~14 vector instructions),
  // however many lanes compute one -- so all blocks of a pass are computed side by side: 8 envs x 8 blocks,
  // or, for the variants that need a ninth block (PT1 + DR), 4 envs x 8 blocks on lanes 0..31 and the ninth
  // blocks of those 4 envs on lanes 32, 40, 48, 56 (a second round if more than 4 envs finished); the reset
  // itself (~2000 cycles) is then evaluated ONCE for up to 8 finished envs.
  constexpr bool NINE = V::MOTOR && V::DR;
  constexpr int EPR = NINE ? kResetsPerPass / 2 : kResetsPerPass;  // envs per Philox round
  const int g = lane / kLanesPerReset, b = lane % kLanesPerReset;
  const int ge = NINE ? (g & 3) : g;                          // env of the round this lane works for
  const uint32_t blk = (NINE && g >= 4) ? 8u : (uint32_t)b;  // block it computes
  bool need = false;
#pragma unroll
  for (int c = 0; c < kMergedScratchBlocks; ++c) need = need || (c == (int)blk && block_needed<V>(c));
  if (NINE && g >= 4 && b != 0) need = false;
  for (int base = 0; base < count; base += kResetsPerPass) {
    const int cnt = min(kResetsPerPass, count - base);  // wave-uniform
    for (int sub = 0; sub < cnt; sub += EPR) {
      const int slot = sub + ge;
      const bool on = slot < cnt;
      const int src = on ? (int)lanes[base + slot] : lane;
      const uint32_t env_id = (uint32_t)(a.env_id_base + (unsigned long long)(wave_base + src));
      const DirectWords dw(env_id, rk);
      if (on && need) scratch[slot * kMergedScratchBlocks + blk] = dw.reset_block(blk);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (mine && pos >= base && pos < base + kResetsPerPass) {  // one evaluation for up to 8 finished envs
      const LdsWords lw{scratch + (pos - base) * kMergedScratchBlocks};
      Sample s;
      sample_philox<V>(a.k, lw, s);
      ctr = ctr_pack(0u, 0u, (uint32_t)ref_offset);
      LatRows no_rows;
      reset_env<V>(a.k, ref_lds, s, e, q, u0, mx, par, ctr, no_rows);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the scratch is refilled by the next pass
  }
}

// tick / parity of the tile this wave owns (see WaveClock in pds_types.h)
PDS_DEV void read_clock(const WaveClock *clk, long long tile, RngKey &rk, int &parity) {
  const WaveClock c = clk[tile];
  rk.tick_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.x);
  rk.tick_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.y);
  parity = __builtin_amdgcn_readfirstlane((int)c.z) & 1;
}
PDS_DEV void advance_clock(WaveClock *clk, long long tile, const RngKey &rk, int parity, uint32_t ticks, int lane) {
  if (lane == 0) {
    const unsigned long long t = (((unsigned long long)rk.tick_hi << 32) | rk.tick_lo) + ticks;
    clk[tile] = make_uint4((uint32_t)t, (uint32_t)(t >> 32), (uint32_t)parity, 0u);
  }
}

// Explicit reset (pds_reset / pds_reset_from_samples): not a hot path.
template <class V>
__global__ __launch_bounds__(kBlock) void reset_kernel(const StepArgs a) {
  const float2 *ref_lds = nullptr;  // (unused: target_at evaluates the reference circle)
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  const long long tile = i / kWave;  // wave-uniform
  if (tile * kWave >= a.n) return;
  RngKey rk{a.seed_lo, a.seed_hi, 0u, 0u};
  int parity;
  read_clock(a.st.clk, tile, rk, parity);
  if (i < a.n && !(a.mask != nullptr && a.mask[i] == 0)) {
    float stale_w[3] = {0.f, 0.f, 0.f}, bias[3] = {0.f, 0.f, 0.f};
    if (V::ON) {
      const float4 q2 = a.st.s2[i];
      const float4 nz = a.st.nz0[i];
      stale_w[0] = q2.y; stale_w[1] = q2.z; stale_w[2] = q2.w;
      bias[0] = nz.x; bias[1] = nz.y; bias[2] = nz.z;
    }
    ResetOut r;
    const DirectWords dw((uint32_t)(a.env_id_base + (unsigned long long)i), rk);
    const uint32_t ctr_old = ctr_pack(0u, 0u, circle_ref_offset(a.st.ctr[i], a.k.ref_points));  // reset_env keeps ref_offset
    reset_compute<V>(a, ref_lds, dw, ctr_old, a.samples != nullptr ? a.samples + i * PDS_SAMPLE_FLOATS : nullptr,
                     stale_w, bias, r);
    reset_store<V>(a, ref_lds, i, r);
  }
  advance_clock(a.st.clk, tile, rk, parity, 1u, threadIdx.x & (kWave - 1));  // a reset consumes one tick
}

// The auto-reset of a SplitReset<V> step kernel (csrc/pds_types.h), launched behind it on the same stream: that kernel stored the
// finished envs' TERMINAL state, left their last observation in the obs row and set the flags.  One WAVE (= block) per 1024 envs
// -- a reset is a ~3 500-instruction dependent chain for the lane that evaluates it, whatever else its block does, so what
// matters is that all 1024 waves of a 2^20-env launch are resident at once (a 256-thread block with one working wave held
// four waves' registers: 20.7 us; the Philox blocks computed side by side by four waves through LDS + block barriers: 30.2):
//   1. every lane reads the flags of 16 consecutive envs and appends the finished ones to an LDS queue;
//   2. the queue is served one env per lane: the env's last observation row is copied to final_obs, then the reset of the
//      explicit reset kernel above (every lane computes its own Philox blocks: DirectWords) with the inputs the in-place reset
//      takes from its registers read back from the stored state -- the terminal body rates (the reference re-initialises the gyro
//      low-pass with them, envs/base.py:411), the gyro bias, Circle's ref_offset -- and the step's tick (the clock word has
//      already been advanced: tick - 1), the fresh state and the row [o0, u0, o0', u0] written straight to HBM (strided: 2 % of
//      the envs).  Same functions, same draws, same bits as the in-place reset (tests/test_gpu_properties.py).
// a.k_steps = reset_store's oh_mode (what the step kernel in front keeps of the noisy observation: 0 regenerated, 2 Kalman hold).
template <class V>
__global__ __launch_bounds__(kWave) void post_reset_kernel(const StepArgs a) {
  constexpr int D = V::D;
  constexpr int kPer = kPostResetEnvsPerBlock / kWave;  // flags per lane
  static_assert(kPer == 16 || kPer == 8 || kPer == 4, "one 16- / 8- / 4-byte load of each flag array per lane");
  __shared__ uint32_t queue[kPostResetEnvsPerBlock];
  __shared__ int qn;
  const float2 *ref_lds = nullptr;
  const int lane = threadIdx.x;
  const long long base = (long long)blockIdx.x * kPostResetEnvsPerBlock;
  if (lane == 0) qn = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  {
    const long long e0 = base + (long long)kPer * lane;
    uint32_t fl[4] = {0u, 0u, 0u, 0u};
    if (e0 + kPer <= a.n && ((reinterpret_cast<uintptr_t>(a.term) | reinterpret_cast<uintptr_t>(a.trunc)) & (uintptr_t)(kPer - 1)) == 0) {
      if constexpr (kPer == 16) {
        const uint4 t4 = *reinterpret_cast<const uint4 *>(a.term + e0), u4 = *reinterpret_cast<const uint4 *>(a.trunc + e0);
        fl[0] = t4.x | u4.x; fl[1] = t4.y | u4.y; fl[2] = t4.z | u4.z; fl[3] = t4.w | u4.w;
      } else if constexpr (kPer == 8) {
        const uint2 t2 = *reinterpret_cast<const uint2 *>(a.term + e0), u2 = *reinterpret_cast<const uint2 *>(a.trunc + e0);
        fl[0] = t2.x | u2.x; fl[1] = t2.y | u2.y;
      } else {
        fl[0] = *reinterpret_cast<const uint32_t *>(a.term + e0) | *reinterpret_cast<const uint32_t *>(a.trunc + e0);
      }
    } else {
      for (int j = 0; j < kPer; ++j)
        if (e0 + j < a.n && (a.term[e0 + j] | a.trunc[e0 + j])) fl[j >> 2] |= 1u << (8 * (j & 3));
    }
    if ((fl[0] | fl[1] | fl[2] | fl[3]) != 0u) {
      int cnt = 0;
      for (int j = 0; j < kPer; ++j) cnt += ((fl[j >> 2] >> (8 * (j & 3))) & 0xFFu) != 0u;
      int at_ = atomicAdd(&qn, cnt);
      for (int j = 0; j < kPer; ++j)
        if (((fl[j >> 2] >> (8 * (j & 3))) & 0xFFu) != 0u) queue[at_++] = (uint32_t)(kPer * lane + j);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const int n_fin = __builtin_amdgcn_readfirstlane(qn);
  for (int idx = lane; idx < n_fin; idx += kWave) {
    const long long i = base + queue[idx];
    // the step's tick: the step kernel has advanced the tile's clock word by one
    const WaveClock clk = a.st.clk[i / kWave];
    RngKey rk{a.seed_lo, a.seed_hi, 0u, 0u};
    {
      const unsigned long long t = (((unsigned long long)clk.y << 32) | clk.x) - 1ull;
      rk.tick_lo = (uint32_t)t; rk.tick_hi = (uint32_t)(t >> 32);
    }
    if (a.final_obs != nullptr) {  // the last observation of the finished episode (D even: 8-byte aligned rows)
      const float2 *src = reinterpret_cast<const float2 *>(a.obs + i * D);
      float2 *dst = reinterpret_cast<float2 *>(a.final_obs + i * D);
#pragma unroll
      for (int j = 0; j < D / 2; ++j) dst[j] = src[j];
    }
    float stale_w[3] = {0.f, 0.f, 0.f}, bias[3] = {0.f, 0.f, 0.f};
    if (V::ON) {
      const float4 q2 = a.st.s2[i];
      const float4 nz = a.st.nz0[i];
      stale_w[0] = q2.y; stale_w[1] = q2.z; stale_w[2] = q2.w;
      bias[0] = nz.x; bias[1] = nz.y; bias[2] = nz.z;
    }
    uint32_t ref_offset = 0u;
    if (V::TASK == PDS_TASK_CIRCLE && !a.k.reset_dist) ref_offset = circle_ref_offset(a.st.ctr[i], a.k.ref_points);
    ResetOut r;
    const DirectWords dw((uint32_t)(a.env_id_base + (unsigned long long)i), rk);
    reset_compute<V>(a, ref_lds, dw, ctr_pack(0u, 0u, ref_offset), nullptr, stale_w, bias, r);
    reset_store<V>(a, ref_lds, i, r, a.k_steps);
  }
}

}  // namespace pds
