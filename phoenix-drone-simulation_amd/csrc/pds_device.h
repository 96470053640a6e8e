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

// pybullet.getQuaternionFromEuler (ZYX half-angle products, [x,y,z,w], normalised); the same
// formula is restated in the reference at envs/utils.py:32-56.  Call sites envs/physics.py:179.
PDS_DEV Quat quat_from_euler(float roll, float pitch, float yaw) {
  float sr, cr, sp, cp, sy, cy;
  sincosf(roll * 0.5f, &sr, &cr);
  sincosf(pitch * 0.5f, &sp, &cp);
  sincosf(yaw * 0.5f, &sy, &cy);
  Quat q;
  q.x = sr * cp * cy - cr * sp * sy;
  q.y = cr * sp * cy + sr * cp * sy;
  q.z = cr * cp * sy - sr * sp * cy;
  q.w = cr * cp * cy + sr * sp * sy;
  const float inv = rsqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x *= inv; q.y *= inv; q.z *= inv; q.w *= inv;
  return q;
}

// pybullet.getMatrixFromQuaternion (b3Matrix3x3::setRotation), row-major R[9].
PDS_DEV void matrix_from_quat(const Quat q, float R[9]) {
  const float d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  const float s = 2.0f / d;
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

PDS_DEV U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0;
    const uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

PDS_DEV float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
PDS_DEV float urange(uint32_t x, float lo, float hi) { return lo + (hi - lo) * u01(x); }
PDS_DEV void box_muller(uint32_t a, uint32_t b, float &z0, float &z1) {
  const float u1 = (float)((a >> 8) + 1u) * (1.0f / 16777216.0f);
  const float u2 = u01(b);
  const float r = sqrtf(-2.0f * logf(u1));
  float s, c;
  sincosf(6.28318530717958647692f * u2, &s, &c);
  z0 = r * c;
  z1 = r * s;
}

PDS_DEV float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

}  // namespace pds
