// pds_device.h -- device-side math of the fused CrazyFlie step (gfx950 only).
//
// Reference citations are relative to phoenix_drone_simulation/ in SvenGronauer/phoenix-drone-simulation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PDS_DEV __device__ __forceinline__

namespace pds {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kHalfPi = 1.57079632679489661923f;

struct Quat { float x, y, z, w; };

// sin/cos for the half-angle arguments of the step: 3-term Cody-Waite reduction by pi/2 with FMA
// (valid far beyond any angle an episode can reach) + cephes-style minimax kernels on [-pi/4, pi/4].
// Max abs error 9.3e-8 for |x| <= 1e5 (checked against double on 28 M samples) and the reduction
// stays exact up to |x| ~ 2.6e7 (n = rint(x 2/pi) < 2^24); a half angle that large needs body rates
// above 1e4 rad/s for a whole episode, i.e. a state that has already overflowed.  Non-finite
// arguments give NaN like the library.  Branch-free, ~25 VALU instructions; ocml's sincosf with its
// Payne-Hanek path cost 12 branches and ~3000 lines of code per kernel.
PDS_DEV void fast_sincos(float x, float &s, float &c) {
  const float n = rintf(__fmul_rn(x, 0.636619772367581343f));
  float r = fmaf(n, -1.57079637050628662109375f, x);
  r = fmaf(n, 4.37113900018624283e-8f, r);
  r = fmaf(n, 1.7151245100059521e-15f, r);
  // beyond |x| ~ 2.6e7 the reduction is no longer exact: keep the polynomial argument in range so
  // that a state that has already diverged still yields a point on the unit circle (1 instruction)
  r = __builtin_amdgcn_fmed3f(r, -0.7855f, 0.7855f);
  const int q = (int)n;
  const float r2 = __fmul_rn(r, r);
  const float sp = fmaf(__fmul_rn(r, r2), fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  const float cp = fmaf(__fmul_rn(r2, r2), fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                        fmaf(r2, -0.5f, 1.0f));
  const float ss = (q & 1) ? cp : sp;
  const float cc = (q & 1) ? sp : cp;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
}

PDS_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }   // v_rcp_f32, 1 ulp
PDS_DEV float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); } // v_sqrt_f32, 1 ulp

// pybullet.getQuaternionFromEuler (ZYX half-angle products, [x,y,z,w], normalised); the same
// formula is restated in the reference at envs/utils.py:32-56.  Call sites envs/physics.py:179.
PDS_DEV Quat quat_from_euler(float roll, float pitch, float yaw) {
  float sr, cr, sp, cp, sy, cy;
  fast_sincos(__fmul_rn(roll, 0.5f), sr, cr);
  fast_sincos(__fmul_rn(pitch, 0.5f), sp, cp);
  fast_sincos(__fmul_rn(yaw, 0.5f), sy, cy);
  // The operation sequence is pinned (explicit FMAs, products that must not be contracted go
  // through __fmul_rn) so that every inlined copy produces the same bits: the o(k) half of an
  // observation is REBUILT from the stored Euler angles in the next step and has to equal the
  // o(k+1) half written by the previous step exactly, like the reference's history deque.
  const float a = __fmul_rn(sr, cp), b = __fmul_rn(cr, sp), c = __fmul_rn(cr, cp), d = __fmul_rn(sr, sp);
  Quat q;
  q.x = fmaf(a, cy, -__fmul_rn(b, sy));
  q.y = fmaf(b, cy, __fmul_rn(a, sy));
  q.z = fmaf(c, sy, -__fmul_rn(d, cy));
  q.w = fmaf(c, cy, __fmul_rn(d, sy));
  const float n2 = fmaf(q.w, q.w, fmaf(q.z, q.z, fmaf(q.y, q.y, __fmul_rn(q.x, q.x))));
  const float inv = __builtin_amdgcn_rsqf(n2);
  q.x = __fmul_rn(q.x, inv); q.y = __fmul_rn(q.y, inv); q.z = __fmul_rn(q.z, inv); q.w = __fmul_rn(q.w, inv);
  return q;
}

// pybullet.getMatrixFromQuaternion (b3Matrix3x3::setRotation), row-major R[9].
PDS_DEV void matrix_from_quat(const Quat q, float R[9]) {
  const float d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  const float s = 2.0f * fast_rcp(d);
  const float xs = q.x * s, ys = q.y * s, zs = q.z * s;
  const float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
  const float xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
  const float yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
  R[0] = 1.0f - (yy + zz); R[1] = xy - wz; R[2] = xz + wy;
  R[3] = xy + wz; R[4] = 1.0f - (xx + zz); R[5] = yz - wx;
  R[6] = xz - wy; R[7] = yz + wx; R[8] = 1.0f - (xx + yy);
}

// pybullet.getEulerFromQuaternion (gimbal guard at |sarg| >= 0.99999); envs/agents.py:446.
PDS_DEV void euler_from_quat(const Quat q, float &roll, float &pitch, float &yaw) {
  const float sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, squ = q.w * q.w;
  const float sarg = -2.0f * (q.x * q.z - q.w * q.y);
  if (sarg <= -0.99999f) {
    roll = 0.f; pitch = -kHalfPi; yaw = 2.f * atan2f(q.x, -q.y);
  } else if (sarg >= 0.99999f) {
    roll = 0.f; pitch = kHalfPi; yaw = 2.f * atan2f(-q.x, q.y);
  } else {
    roll = atan2f(2.f * (q.y * q.z + q.w * q.x), squ - sqx - sqy + sqz);
    pitch = asinf(sarg);
    yaw = atan2f(2.f * (q.x * q.y + q.w * q.z), squ + sqx - sqy - sqz);
  }
}

// ---- counter-based RNG: Philox4x32-10 (Salmon et al., SC'11) ---------------------------------
// counter = (global env id, tick lo, tick hi, block), key = (seed lo, seed hi).  Restated on the
// CPU by oracle/phoenix_oracle.c po_philox4x32_10 / po_philox_reset_sample.
struct U4 { uint32_t x, y, z, w; };

// a ^ b ^ c in ONE vector instruction (gfx950 v_bitop3_b32, truth table 0x96): the compiler emits two v_xor_b32 for the
// three-input xor of a Philox round (it reserves BITOP3 for mixed and/or/xor trees) -- 2 of the ~14 vector instructions of a round
#ifndef PDS_XOR3
#define PDS_XOR3 1
#endif
PDS_DEV uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if PDS_XOR3
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}

// X3: see philox4x32_10_or_7 below (default: the 7-round per-step noise blocks use v_bitop3_b32, the 10-round reset / sampling
// blocks plain xors)
template <int ROUNDS, bool X3 = (ROUNDS == 7)>
PDS_DEV U4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    // separate hi / lo multiplies: measured faster than one v_mad_u64_u32 per product on gfx950
    // (noisy Hover step 138 us vs 170 us)
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = X3 ? xor3(hi1, c1, k0) : (hi1 ^ c1 ^ k0);
    const uint32_t n2 = X3 ? xor3(hi0, c3, k1) : (hi0 ^ c3 ^ k1);
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

// Both variants in ONE instruction stream: lanes with `seven` keep the state after round 7, the others
// after round 10.  (A wave whose lanes compute different blocks side by side would otherwise run the
// 10-round and the 7-round code one after the other: a round is ~14 vector instructions, ~60 cycles of the
// wave's VALU time whatever the number of active lanes; v_mul_lo/hi_u32 issue at the v_fma_f32 rate on gfx950,
// profiles/r02_valu_issue.txt.)
// X3: the three-input xors as v_bitop3_b32 (xor3 above) -- what the observation-noise / latency variants gained 1.5 % from in round 5.
// false (plain `^`, two v_xor_b32) for the merged in-register reset of the variants WITHOUT noise or latency: there v_bitop3's
// register operands cost the half-tile PT1 + DR kernel (BASELINE config 3, 128-VGPR cap) 4 spilled VGPRs and 1.7 % (same box,
// round-4 library 19.74 us, round 5 / 6 with xor3 20.1; profiles/r06_ab_vs_round4.txt).  Same bits either way.
template <bool X3 = true>
PDS_DEV U4 philox4x32_10_or_7(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, bool seven) {
  U4 at7{0u, 0u, 0u, 0u};
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    if (r == 7) at7 = U4{c0, c1, c2, c3};
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = X3 ? xor3(hi1, c1, k0) : (hi1 ^ c1 ^ k0);
    const uint32_t n2 = X3 ? xor3(hi0, c3, k1) : (hi0 ^ c3 ^ k1);
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return seven ? at7 : U4{c0, c1, c2, c3};
}

// reset sampling: the standard 10 rounds
PDS_DEV U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  return philox4x32<10>(c0, c1, c2, c3, k0, k1);
}
// per-step sensor / thrust noise (5 blocks per env-step): 7 rounds, the smallest Philox4x32 variant
// that passes BigCrush (Salmon et al., SC'11, table 2)
PDS_DEV U4 philox4x32_7(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  return philox4x32<7>(c0, c1, c2, c3, k0, k1);
}

PDS_DEV float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
PDS_DEV float urange_u(float u, float lo, float hi) { return lo + (hi - lo) * u; }  // u in [0, 1)
PDS_DEV float urange(uint32_t x, float lo, float hi) { return urange_u(u01(x), lo, hi); }
// Box-Muller on two Philox words, evaluated with the hardware transcendental units (v_log_f32,
// v_sqrt_f32, v_sin_f32 / v_cos_f32 take their argument in revolutions): abs error ~1e-6 in z,
// which is noise on a random variate.  The oracle restates the same formula with libm.
PDS_DEV void box_muller(uint32_t a, uint32_t b, float &z0, float &z1) {
  const float u1 = (float)((a >> 8) + 1u) * (1.0f / 16777216.0f);
  const float u2 = u01(b);
  const float r = fast_sqrt(-1.38629436111989061883f * __log2f(u1));  // -2 ln u1 = -2 ln2 log2 u1
  z0 = r * __builtin_amdgcn_cosf(u2);
  z1 = r * __builtin_amdgcn_sinf(u2);
}

// Per-step noise: ONE Philox word per Box-Muller pair -- radius from the high 20 bits (|z| <= 5.26),
// angle from the low 12 bits (a 4096-point rule integrates the smooth periodic angle dependence of
// the marginal to float precision, so each z is N(0,1) up to the 1.4e-7 tail mass beyond 5.26 sigma)
// -- and one 16-bit half word per uniform.  Halves the Philox work of the noisy variants.
// Round 5: the angles are the MIDPOINTS (j + 1/2) / 4096 of a turn.  The grid j / 4096 contains 0, 1/4, 1/2 and 3/4 of
// a turn, where cos or sin is exactly 0: every normal had an atom of mass 1/2048 at z = 0, which 2^26 samples show as a
// Kolmogorov-Smirnov distance of 3.4e-4 (tests/test_gpu_noise.py; 1.95 / sqrt(2^26) = 2.4e-4 is the 0.1 % critical value).
PDS_DEV void box_muller_word(uint32_t w, float &z0, float &z1) {
  const float u1 = (float)((w >> 12) + 1u) * (1.0f / 1048576.0f);
  const float u2 = fmaf((float)(w & 0xFFFu), 1.0f / 4096.0f, 0.5f / 4096.0f);
  const float r = fast_sqrt(-1.38629436111989061883f * __log2f(u1));
  z0 = r * __builtin_amdgcn_cosf(u2);
  z1 = r * __builtin_amdgcn_sinf(u2);
}
PDS_DEV float u01_lo16(uint32_t w) { return (float)(w & 0xFFFFu) * (1.0f / 65536.0f); }
PDS_DEV float u01_hi16(uint32_t w) { return (float)(w >> 16) * (1.0f / 65536.0f); }

PDS_DEV float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// Streamed-once tensors (actions in; observation, reward, cost, flags out) use the non-temporal
// cache policy so they do not evict the state quads that the next step re-reads (measured on the
// traffic-shape microbenchmark: +1.5 %).  -DPDS_NT=0 builds the plain-policy variant for A/B runs.
#ifndef PDS_NT
#define PDS_NT 1
#endif
typedef float pds_v4f __attribute__((ext_vector_type(4)));
PDS_DEV float4 nt_load4(const float4 *p) {
#if PDS_NT
  const pds_v4f t = __builtin_nontemporal_load(reinterpret_cast<const pds_v4f *>(p));
  return make_float4(t.x, t.y, t.z, t.w);
#else
  return *p;
#endif
}
PDS_DEV void nt_store4(float4 *p, const float4 v) {
#if PDS_NT
  const pds_v4f t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<pds_v4f *>(p));
#else
  *p = v;
#endif
}
// cache policy of the per-env STATE quads (re-read by the next step): tuning knob
#ifndef PDS_NT_STATE
#define PDS_NT_STATE 0
#endif
PDS_DEV float4 st_load4(const float4 *p) {
#if PDS_NT_STATE >= 2
  return nt_load4(p);
#else
  return *p;
#endif
}
PDS_DEV void st_store4(float4 *p, const float4 v) {
#if PDS_NT_STATE >= 1
  nt_store4(p, v);
#else
  *p = v;
#endif
}
template <typename T>
PDS_DEV void nt_store(T *p, const T v) {
#if PDS_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

}  // namespace pds
