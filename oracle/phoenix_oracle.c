/*
 * phoenix_oracle.c -- CPU ORACLE (test infrastructure, NOT the product; see phoenix_oracle.h).
 *
 * Restates, function by function and in the reference's operation order, the SimplePhysics
 * reset()/step() path of SvenGronauer/phoenix-drone-simulation.  Every function cites the reference
 * file:line (relative to /root/reference/phoenix_drone_simulation/) it follows.
 * Compiled twice by oracle/Makefile: -DPO_F64 (double, suffix _f64) and -DPO_F32 (float, suffix _f32).
 */
#include "phoenix_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#if defined(PO_F32)
typedef float REAL;
#define SUF(name) name##_f32
#define R_SIN sinf
#define R_COS cosf
#define R_SQRT sqrtf
#define R_ATAN2 atan2f
#define R_ASIN asinf
#define R_FABS fabsf
#define R_LOG logf
#define R_EXP expf
typedef po_env_f32 ENV;
#else
typedef double REAL;
#define SUF(name) name##_f64
#define R_SIN sin
#define R_COS cos
#define R_SQRT sqrt
#define R_ATAN2 atan2
#define R_ASIN asin
#define R_FABS fabs
#define R_LOG log
#define R_EXP exp
typedef po_env_f64 ENV;
#endif

#define PO_PI 3.14159265358979323846

/* ---- model constants: envs/assets/cf21x_sys_eq.urdf:10 (properties), :16-17 (mass, inertia);
 *      derived in envs/agents.py:138-156 ------------------------------------------------------- */
#define C_M 0.027
#define C_L 0.0397
#define C_T2W 2.25
#define C_IXX 1.7e-5
#define C_IYY 1.7e-5
#define C_IZZ 2.9e-5
#define C_KF 3.16e-10
#define C_KM 7.94e-12
#define C_GND_EFF_COEFF 11.36859
#define C_PROP_RADIUS 2.31348e-2
#define C_FTF0 1.56e-5 /* envs/agents.py:139 */
#define C_FTF1 5.96e-3 /* envs/agents.py:140 */
#define C_G 9.81       /* envs/agents.py:142 and envs/physics.py:16 */
/* motor link offsets, links 0..3: envs/assets/cf21x_sys_eq.urdf:47,59,71,83 */
static const double C_MOTOR_X[4] = {0.028, -0.028, -0.028, 0.028};
static const double C_MOTOR_Y[4] = {-0.028, -0.028, 0.028, 0.028};

#ifdef PO_F64 /* -------- precision independent helpers, emitted once -------------------------- */
void po_get_constants(po_constants *k) {
  k->M = C_M; k->L = C_L; k->THRUST2WEIGHT_RATIO = C_T2W;
  k->IXX = C_IXX; k->IYY = C_IYY; k->IZZ = C_IZZ; k->KF = C_KF; k->KM = C_KM;
  k->GND_EFF_COEFF = C_GND_EFF_COEFF; k->PROP_RADIUS = C_PROP_RADIUS;
  k->FORCE_TORQUE_FACTOR_0 = C_FTF0; k->FORCE_TORQUE_FACTOR_1 = C_FTF1;
  k->G = C_G;
  k->GRAVITY = k->G * k->M;                                   /* agents.py:145 */
  k->MAX_THRUST = k->GRAVITY * k->THRUST2WEIGHT_RATIO / 4;    /* agents.py:146 */
  k->MAX_TORQUE = k->FORCE_TORQUE_FACTOR_1 * k->MAX_THRUST;   /* agents.py:147 */
  k->HOVER_X = sqrt(1 / k->THRUST2WEIGHT_RATIO);              /* agents.py:149 */
  k->HOVER_ACTION = 2 * 1 / k->THRUST2WEIGHT_RATIO - 1;       /* agents.py:150 */
  k->MAX_RPM = sqrt((k->THRUST2WEIGHT_RATIO * k->GRAVITY) / (4 * k->MAX_THRUST)); /* :152 */
  k->GND_EFF_H_CLIP = 0.25 * k->PROP_RADIUS *
      sqrt((15 * k->MAX_RPM * k->MAX_RPM * k->KF * k->GND_EFF_COEFF) / k->MAX_THRUST); /* :153 */
}

/* ctor defaults: envs/hover.py:7-45, envs/circle.py:7-61, envs/takeoff.py:13-56,
 * *SimpleEnv classes hover.py:253-266, circle.py:286-299, takeoff.py:221-231, base.py:26-48 */
void po_default_config(int task, po_config *c) {
  memset(c, 0, sizeof(*c));
  c->task = task;
  c->use_motor_dynamics = 0;
  c->use_ground_effect = 0;
  c->observation_noise = 1;
  c->aggregate_phy_steps = 1;
  c->enable_reset_distribution = 1;
  c->max_episode_steps = 500;
  c->obs_rate = 1;
  c->domain_randomization = 0.10;
  c->motor_thrust_noise = 0.05;
  c->time_step = 1.0 / 100.0;
  c->motor_time_constant = 0.080;
  c->penalty_action = 1e-4;
  c->penalty_angle = 0;
  c->penalty_spin = (task == PO_TASK_CIRCLE) ? 1e-3 : 1e-4;
  c->penalty_terminal = 100;
  c->penalty_velocity = (task == PO_TASK_CIRCLE) ? 1e-4 : 0;
  c->ARP = (task == PO_TASK_CIRCLE) ? 1e-3 : 0;
  c->target_pos[0] = 0; c->target_pos[1] = 0; c->target_pos[2] = 1.0;
  c->init_xyz[0] = 0; c->init_xyz[1] = 0;
  c->init_xyz[2] = (task == PO_TASK_TAKEOFF) ? (double)0.0125f : 1.0; /* float32 literal, takeoff.py:51 */
  c->use_latency = 0;  /* CrazyFlieSimpleAgent passes use_latency=False, agents.py:492 */
  c->latency = 0.015;  /* envs/base.py:40 */
  c->ref_points = 300; /* 3 s * observation_frequency (100), circle.py:47-49; TakeOff: 300 fixed, takeoff.py:43 */
}

/* Philox4x32-10 (Salmon et al., SC'11).  This is the in-kernel RNG of the NEW framework (the
 * reference uses the global numpy MT19937 stream, which cannot be reproduced for 2^20 lockstep
 * envs); the oracle restates it so the GPU reset sampling can be checked draw for draw. */
void po_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < rounds; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void po_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  po_philox4x32(ctr, key, 10, out);
}

int po_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
#endif /* PO_F64 */

/* ------------------------------------------------------------------------------------------------
 * pybullet pure functions (third party, absent; restated from Bullet3 -- see header)
 * ---------------------------------------------------------------------------------------------- */

/* pybullet.getQuaternionFromEuler; same formula in envs/utils.py:32-56. Call sites physics.py:179,
 * agents.py:55, hover.py:146,209. */
void SUF(po_quat_from_euler)(const REAL rpy[3], REAL q[4]) {
  REAL phi = rpy[0] * (REAL)0.5, the = rpy[1] * (REAL)0.5, psi = rpy[2] * (REAL)0.5;
  REAL sphi = R_SIN(phi), cphi = R_COS(phi);
  REAL sthe = R_SIN(the), cthe = R_COS(the);
  REAL spsi = R_SIN(psi), cpsi = R_COS(psi);
  REAL x = sphi * cthe * cpsi - cphi * sthe * spsi;
  REAL y = cphi * sthe * cpsi + sphi * cthe * spsi;
  REAL z = cphi * cthe * spsi - sphi * sthe * cpsi;
  REAL w = cphi * cthe * cpsi + sphi * sthe * spsi;
  REAL n = R_SQRT(x * x + y * y + z * z + w * w);
  q[0] = x / n; q[1] = y / n; q[2] = z / n; q[3] = w / n;
}

/* pybullet.getMatrixFromQuaternion (b3Matrix3x3::setRotation). Call sites physics.py:160,
 * agents.py:452, hover.py:237. */
void SUF(po_matrix_from_quat)(const REAL q[4], REAL R[9]) {
  REAL x = q[0], y = q[1], z = q[2], w = q[3];
  REAL d = x * x + y * y + z * z + w * w;
  REAL s = (REAL)2.0 / d;
  REAL xs = x * s, ys = y * s, zs = z * s;
  REAL wx = w * xs, wy = w * ys, wz = w * zs;
  REAL xx = x * xs, xy = x * ys, xz = x * zs;
  REAL yy = y * ys, yz = y * zs, zz = z * zs;
  R[0] = (REAL)1.0 - (yy + zz); R[1] = xy - wz; R[2] = xz + wy;
  R[3] = xy + wz; R[4] = (REAL)1.0 - (xx + zz); R[5] = yz - wx;
  R[6] = xz - wy; R[7] = yz + wx; R[8] = (REAL)1.0 - (xx + yy);
}

/* pybullet.getEulerFromQuaternion. Call site agents.py:446 (reset only). */
void SUF(po_euler_from_quat)(const REAL q[4], REAL rpy[3]) {
  REAL x = q[0], y = q[1], z = q[2], w = q[3];
  REAL sqx = x * x, sqy = y * y, sqz = z * z, squ = w * w;
  REAL sarg = (REAL)-2.0 * (x * z - w * y);
  if (sarg <= (REAL)-0.99999) {
    rpy[0] = 0; rpy[1] = (REAL)(-0.5 * PO_PI); rpy[2] = 2 * R_ATAN2(x, -y);
  } else if (sarg >= (REAL)0.99999) {
    rpy[0] = 0; rpy[1] = (REAL)(0.5 * PO_PI); rpy[2] = 2 * R_ATAN2(-x, y);
  } else {
    rpy[0] = R_ATAN2(2 * (y * z + w * x), squ - sqx - sqy + sqz);
    rpy[1] = R_ASIN(sarg);
    rpy[2] = R_ATAN2(2 * (x * y + w * z), squ + sqx - sqy - sqz);
  }
}

/* Bullet's pose read-back: bc.getBasePositionAndOrientation (agents.py:443) does not return the
 * quaternion that resetBasePositionAndOrientation (hover.py:231-235, circle.py:259-263,
 * takeoff.py:193-197) stored.  Bullet3 as published (third party, unpinned -- setup.py:32):
 * PhysicsServerCommandProcessor.cpp processRequestActualStateCommand answers with
 * `btTransform tr; tr.setRotation(mb->getWorldToBaseRot().inverse()); ... tr.getRotation()`, and
 * btTransform keeps a 3x3 basis, i.e. btMatrix3x3::setRotation followed by btMatrix3x3::getRotation
 * (LinearMath/btMatrix3x3.h; double precision in pybullet): trace > 0 -> w = sqrt(trace + 1) / 2 > 0,
 * otherwise the component belonging to the largest diagonal element is the positive square root.
 * pybullet.c pybullet_internalGetBasePositionAndOrientation copies the seven numbers unchanged. */
void SUF(po_bullet_readback_quat)(const REAL qin[4], REAL q[4]) {
  REAL m[3][3];
  {
    REAL R[9];
    SUF(po_matrix_from_quat)(qin, R); /* btMatrix3x3::setRotation == the getMatrixFromQuaternion formula */
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) m[r][c] = R[3 * r + c];
  }
  REAL trace = m[0][0] + m[1][1] + m[2][2];
  REAL temp[4];
  if (trace > (REAL)0.0) {
    REAL s = R_SQRT(trace + (REAL)1.0);
    temp[3] = s * (REAL)0.5;
    s = (REAL)0.5 / s;
    temp[0] = (m[2][1] - m[1][2]) * s;
    temp[1] = (m[0][2] - m[2][0]) * s;
    temp[2] = (m[1][0] - m[0][1]) * s;
  } else {
    int i = m[0][0] < m[1][1] ? (m[1][1] < m[2][2] ? 2 : 1) : (m[0][0] < m[2][2] ? 2 : 0);
    int j = (i + 1) % 3, k = (i + 2) % 3;
    REAL s = R_SQRT(m[i][i] - m[j][j] - m[k][k] + (REAL)1.0);
    temp[i] = s * (REAL)0.5;
    s = (REAL)0.5 / s;
    temp[3] = (m[k][j] - m[j][k]) * s;
    temp[j] = (m[j][i] + m[i][j]) * s;
    temp[k] = (m[k][i] + m[i][k]) * s;
  }
#ifdef PO_F64
  for (int i = 0; i < 4; ++i) q[i] = temp[i];
#else
  /* f32 build (the stand-in for what the f32 HIP kernel may do): a float matrix round trip adds ~3e-7 of
   * rounding that the reference's double-precision round trip (~1e-16) does not have, so only the SIGN the
   * extraction produces is applied to the input quaternion -- csrc/pds_reset.h reset_env does the same. */
  {
    REAL dot = temp[0] * qin[0] + temp[1] * qin[1] + temp[2] * qin[2] + temp[3] * qin[3];
    for (int i = 0; i < 4; ++i) q[i] = dot < 0 ? -qin[i] : qin[i];
  }
#endif
}

/* ------------------------------------------------------------------------------------------------
 * variate source
 * ---------------------------------------------------------------------------------------------- */
static REAL rng_normal(po_rng *rng) { /* standard normal; numpy: loc + scale*z */
  if (!rng || rng->iz >= rng->nz) return 0;
  return (REAL)rng->z[rng->iz++];
}
static REAL rng_uniform01(po_rng *rng) { /* numpy uniform: low + (high-low)*u */
  if (!rng || rng->iu >= rng->nu) return (REAL)0.5;
  return (REAL)rng->u[rng->iu++];
}

/* Python's `a // b` for floats (Objects/floatobject.c float_divmod): NOT floor(a / b) -- 0.03 // 0.01 == 2.0
 * while 0.03 / 0.01 == 3.0 -- because the remainder is taken exactly (fmod) first. */
static double SUF(py_float_floordiv)(double vx, double wx) {
  double mod = fmod(vx, wx);
  double div = (vx - mod) / wx;
  if (mod != 0 && ((wx < 0) != (mod < 0))) div -= 1.0;
  if (div == 0) return 0.0;
  double fl = floor(div);
  if (div - fl > 0.5) fl += 1.0;
  return fl;
}


/* ------------------------------------------------------------------------------------------------
 * agent construction: envs/agents.py:114-206 (CrazyFlieAgent.__init__), AgentBase :21-79
 * ---------------------------------------------------------------------------------------------- */
void SUF(po_env_init)(const po_config *c, ENV *e) {
  po_constants k;
  po_get_constants(&k);
  memset(e, 0, sizeof(*e));
  e->xyz[2] = 1; /* AgentBase default xyz, agents.py:33 */
  e->quat[3] = 1;
  e->m = (REAL)k.M;
  e->J[0] = (REAL)k.IXX; e->J[1] = (REAL)k.IYY; e->J[2] = (REAL)k.IZZ;
  e->ftf0 = (REAL)k.FORCE_TORQUE_FACTOR_0;
  e->ftf1 = (REAL)k.FORCE_TORQUE_FACTOR_1;
  e->dt = (REAL)c->time_step;        /* physics.time_step, base.py:229 */
  e->T_s = (REAL)c->time_step;       /* agents.py:197 */
  for (int i = 0; i < 4; ++i) {
    e->T[i] = (REAL)c->motor_time_constant;                       /* agents.py:195 */
    e->t2w[i] = (REAL)k.THRUST2WEIGHT_RATIO;
    e->A[i] = (REAL)1 - e->T_s / e->T[i];                         /* agents.py:201 */
    e->B[i] = e->T_s / e->T[i];                                   /* agents.py:202 */
    e->K[i] = (REAL)k.MAX_THRUST;                                 /* agents.py:198 */
  }
  e->target_pos[0] = (REAL)c->target_pos[0];
  e->target_pos[1] = (REAL)c->target_pos[1];
  e->target_pos[2] = (REAL)c->target_pos[2];
  /* agents.py:165,179-183 */
  e->use_latency = (c->use_latency && c->latency >= c->time_step) ? 1 : 0;
  {
    int b = (int)SUF(py_float_floordiv)(c->latency, c->time_step); /* int(self.LATENCY // time_step) */
    e->buf_size = b < 1 ? 1 : b;
    if (e->buf_size > PO_MAX_LAT) e->buf_size = PO_MAX_LAT;
  }
  e->action_idx = 0;
}

/* envs/agents.py:388-404 CrazyFlieAgent.set_latency (called by simopt/pybullet.py:248) */
void SUF(po_set_latency)(const po_config *c, ENV *e, double new_latency) {
  if (new_latency < c->time_step) {
    e->use_latency = 0;
  } else {
    e->use_latency = 1;
    e->buf_size = (int)(new_latency / c->time_step);
    if (e->buf_size > PO_MAX_LAT) e->buf_size = PO_MAX_LAT;
    memset(e->action_buffer, 0, sizeof(e->action_buffer));
    e->action_idx = 0;
    /* last_action / action_history keep pointing at the OLD buffer object, which nothing writes any more */
    if (e->last_action_alias) e->last_action_alias = 0;
    for (int h = 0; h < PO_HIST; ++h) e->hist_alias[h] = 0;
  }
}

/* envs/agents.py:208-224 update_motor_dynamics (note the hard-coded 0.028 in K, :224) */
void SUF(po_update_motor_dynamics)(ENV *e, const REAL *T_new, const REAL *Ts_new, const REAL *t2w_new) {
  if (Ts_new) e->T_s = *Ts_new;
  for (int i = 0; i < 4; ++i) {
    if (T_new) e->T[i] = (T_new[i] < e->T_s) ? e->T_s : T_new[i]; /* np.clip(T, T_s, inf) :218 */
    if (t2w_new) e->t2w[i] = t2w_new[i];
    e->A[i] = (REAL)1 - e->T_s / e->T[i];
    e->B[i] = e->T_s / e->T[i];
    e->K[i] = (REAL)0.028 * (REAL)C_G * e->t2w[i] / (REAL)4;
  }
}

static REAL clipr(REAL v, REAL lo, REAL hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* envs/control.py:120-191 AttitudeRate, :194-287 Attitude (cascaded), :34-50 mixer.  The
 * controllers are built with time_step = 1/sim_freq (envs/agents.py:72-77) and keep it under
 * domain randomisation. */
static void rate_pid(const po_config *c, ENV *e, const REAL target[3], REAL out[3]) {
  const REAL kp[3] = {250, 250, 120}, ki[3] = {500, 500, (REAL)16.7}, kd[3] = {(REAL)2.5, (REAL)2.5, 0};
  const REAL lim[3] = {(REAL)33.3, (REAL)33.3, (REAL)166.7};
  const REAL dt = (REAL)c->time_step;
  for (int i = 0; i < 3; ++i) {
    REAL error = (target[i] - e->rpy_dot[i]) * (REAL)180. / (REAL)PO_PI;   /* control.py:166 */
    REAL derivative = (error - e->rate_err[i]) / dt;
    e->rate_err[i] = error;
    e->rate_int[i] = clipr(e->rate_int[i] + error * dt, -lim[i], lim[i]);
    out[i] = kp[i] * error + ki[i] * e->rate_int[i] + kd[i] * derivative;   /* :174-176 */
  }
}

static void att_pid(const po_config *c, ENV *e, const REAL target[3], REAL out[3]) {
  const REAL kp[3] = {6, 6, 6}, ki[3] = {3, 3, 1}, kd[3] = {0, 0, (REAL)0.35};
  const REAL lim[3] = {20, 20, 360};
  const REAL dt = (REAL)c->time_step;
  for (int i = 0; i < 3; ++i) {
    REAL error = (target[i] - e->rpy[i]) * (REAL)180. / (REAL)PO_PI;       /* control.py:268 */
    REAL derivative = (error - e->att_err[i]) / dt;
    e->att_err[i] = error;
    e->att_int[i] = clipr(e->att_int[i] + error * dt, -lim[i], lim[i]);
    REAL o = kp[i] * error + ki[i] * e->att_int[i] + kd[i] * derivative;
    out[i] = o / (REAL)180. * (REAL)PO_PI;                                  /* degree_to_rad :277 */
  }
}

static void control_pid(const po_config *c, ENV *e, const REAL a[4]) {
  REAL ca[4], factors[3], thrust;
  for (int i = 0; i < 4; ++i) ca[i] = clipr(a[i], -1, 1);
  if (c->control_mode == 1) {              /* AttitudeRate.act, control.py:151-160 */
    thrust = (REAL)30000 + ca[0] * (REAL)30000;
    REAL tgt[3] = {ca[1] * (REAL)PO_PI / 3, ca[2] * (REAL)PO_PI / 3, ca[3] * (REAL)PO_PI / 3};
    rate_pid(c, e, tgt, factors);
  } else {                                 /* Attitude.act, control.py:244-259 */
    REAL tgt[3] = {ca[1] * (REAL)PO_PI / 18, ca[2] * (REAL)PO_PI / 18, ca[3] * (REAL)PO_PI / 18};
    thrust = (REAL)45000 + ca[0] * (REAL)10000;
    REAL rates[3];
    att_pid(c, e, tgt, rates);
    rate_pid(c, e, rates, factors);
  }
  /* rpy_control_factors_to_PWM, control.py:34-50 */
  const REAL r = factors[0] / (REAL)2.0, p = factors[1] / (REAL)2.0, y = factors[2];
  e->pwm[0] = clipr(thrust - r - p - y, 0, 60000);
  e->pwm[1] = clipr(thrust - r + p + y, 0, 60000);
  e->pwm[2] = clipr(thrust + r + p - y, 0, 60000);
  e->pwm[3] = clipr(thrust + r - p + y, 0, 60000);
}

/* envs/agents.py:259-298 apply_action (use_latency False for the Simple agent, :492), with
 * envs/control.py:94-100 PWM.act and envs/utils.py:104-108 OUNoise.noise inlined. */
void SUF(po_apply_action)(const po_config *c, ENV *e, const REAL a[4], po_rng *rng, REAL forces[4],
                          REAL *z_torque) {
  REAL torques[4];
  const REAL sigma = (REAL)(0.2 * c->motor_thrust_noise); /* agents.py:206 */
  for (int i = 0; i < 4; ++i) e->last_action[i] = a[i];   /* :264 last_action = action.copy() */
  e->last_action_alias = 0;
  REAL applied[4] = {a[0], a[1], a[2], a[3]};
  if (e->use_latency) {                                    /* :267-274 */
    for (int i = 0; i < 4; ++i) applied[i] = e->action_buffer[e->action_idx][i];   /* delayed action */
    for (int i = 0; i < 4; ++i) e->action_buffer[e->action_idx][i] = a[i];
    e->action_idx = (e->action_idx + 1) % e->buf_size;
  }
  if (c->control_mode == 0) {
    for (int i = 0; i < 4; ++i)                            /* PWM.act, control.py:98-99 */
      e->pwm[i] = (REAL)30000 + clipr(applied[i], -1, 1) * (REAL)30000;
  } else {
    control_pid(c, e, applied);                            /* AttitudeRate.act / Attitude.act */
  }
  for (int i = 0; i < 4; ++i) { /* utils.py:105-107: dx = theta*(mu-x) + sigma*randn; theta .15 mu 0 */
    REAL x = e->ou[i];
    REAL dx = (REAL)0.15 * ((REAL)0 - x) + sigma * rng_normal(rng);
    e->ou[i] = x + dx;
  }
  for (int i = 0; i < 4; ++i) {
    REAL thrust_normed = e->pwm[i] / (REAL)60000; /* :279 */
    REAL noisy_x;
    if (c->use_motor_dynamics) {                  /* :284-288 */
      REAL rot_normed = R_SQRT(thrust_normed);
      e->x[i] = e->A[i] * e->x[i] + e->B[i] * rot_normed;
      noisy_x = ((REAL)1 + e->ou[i]) * (e->x[i] * e->x[i]);
    } else {
      noisy_x = ((REAL)1 + e->ou[i]) * thrust_normed; /* :290 */
    }
    REAL n = clipr(noisy_x, 0, 1);                /* :291 */
    e->y[i] = e->K[i] * n;                        /* :292 */
    forces[i] = e->y[i];
    torques[i] = e->ftf1 * forces[i] + e->ftf0;   /* :295-296 */
  }
  *z_torque = (-torques[0] + torques[1] - torques[2] + torques[3]); /* :297 */
}

/* envs/physics.py:27-58 BasePhysics.calculate_ground_effect.  NOT executed by any Simple env in
 * the reference (use_ground_effect defaults False, physics.py:18, and only PyBulletPhysics
 * consults it, :117-120).  EXTENSION used by BASELINE config (4): the propeller link height is
 * p_z + (R . offset_i)_z with the URDF link offsets; ge_i is added to motor force i (the way
 * PyBulletPhysics applies it, physics.py:119-120); yaw torque is unchanged. */
void SUF(po_ground_effect)(const ENV *e, const REAL forces[4], REAL ge[4]) {
  po_constants k;
  po_get_constants(&k);
  REAL R[9];
  SUF(po_matrix_from_quat)(e->quat, R);
  const REAL gec = (REAL)k.GND_EFF_COEFF, r = (REAL)k.PROP_RADIUS, hclip = (REAL)k.GND_EFF_H_CLIP;
  int ok = (R_FABS(e->rpy[0]) < (REAL)(PO_PI / 2)) && (R_FABS(e->rpy[1]) < (REAL)(PO_PI / 2)); /* :54 */
  for (int i = 0; i < 4; ++i) {
    REAL pz = e->xyz[2] + (R[6] * (REAL)C_MOTOR_X[i] + R[7] * (REAL)C_MOTOR_Y[i]);
    if (pz < hclip) pz = hclip;                    /* :51 */
    REAL q = r / ((REAL)4 * pz);
    REAL g = forces[i] * gec * (q * q);            /* :53 */
    ge[i] = ok ? g : (REAL)0;
  }
}

/* envs/physics.py:130-200 SimplePhysics.step_forward */
void SUF(po_step_forward)(const po_config *c, ENV *e, const REAL a[4], po_rng *rng) {
  REAL forces[4], z_torque;
  SUF(po_apply_action)(c, e, a, rng, forces, &z_torque); /* :149 */
  if (c->use_ground_effect) {
    REAL ge[4];
    SUF(po_ground_effect)(e, forces, ge);
    for (int i = 0; i < 4; ++i) forces[i] = forces[i] + ge[i];
  }
  REAL pos[3], rpy[3], vel[3], w[3], R[9];
  for (int i = 0; i < 3; ++i) { pos[i] = e->xyz[i]; rpy[i] = e->rpy[i]; vel[i] = e->xyz_dot[i]; w[i] = e->rpy_dot[i]; }
  REAL thrust = (((REAL)0 + forces[0]) + forces[1] + forces[2]) + forces[3]; /* np.sum :159 */
  SUF(po_matrix_from_quat)(e->quat, R);                                        /* :160 */
  REAL Fw[3] = {R[2] * thrust, R[5] * thrust, R[8] * thrust};                  /* :161 */
  Fw[0] = Fw[0] - (REAL)0 * e->m;
  Fw[1] = Fw[1] - (REAL)0 * e->m;
  Fw[2] = Fw[2] - (REAL)C_G * e->m;                                            /* :162-163 */
  const REAL sqrt2 = R_SQRT((REAL)2);
  REAL x_torque = (-forces[0] - forces[1] + forces[2] + forces[3]) * (REAL)C_L / sqrt2; /* :167 */
  REAL y_torque = (-forces[0] + forces[1] + forces[2] - forces[3]) * (REAL)C_L / sqrt2; /* :168 */
  REAL Jw[3] = {e->J[0] * w[0], e->J[1] * w[1], e->J[2] * w[2]};
  REAL tq[3] = {x_torque - (w[1] * Jw[2] - w[2] * Jw[1]),                     /* :170-171 */
                y_torque - (w[2] * Jw[0] - w[0] * Jw[2]),
                z_torque - (w[0] * Jw[1] - w[1] * Jw[0])};
  REAL wdd[3] = {tq[0] * ((REAL)1 / e->J[0]), tq[1] * ((REAL)1 / e->J[1]), tq[2] * ((REAL)1 / e->J[2])}; /* :172 */
  REAL acc[3] = {Fw[0] / e->m, Fw[1] / e->m, Fw[2] / e->m};                   /* :173 */
  const REAL dt = e->dt;
  for (int i = 0; i < 3; ++i) vel[i] += dt * acc[i];                           /* :175 */
  for (int i = 0; i < 3; ++i) w[i] += dt * wdd[i];                             /* :176 */
  for (int i = 0; i < 3; ++i) pos[i] += dt * vel[i];                           /* :177 */
  for (int i = 0; i < 3; ++i) rpy[i] += dt * w[i];                             /* :178 */
  SUF(po_quat_from_euler)(rpy, e->quat);                                       /* :179 */
  if (pos[2] < 0) pos[2] = 0;                                                  /* :182 */
  for (int i = 0; i < 3; ++i) { e->xyz[i] = pos[i]; e->rpy[i] = rpy[i]; e->xyz_dot[i] = vel[i]; e->rpy_dot[i] = w[i]; } /* :185-189 */
}

/* ------------------------------------------------------------------------------------------------
 * sensors: envs/sensors.py:75-134 SensorNoise, envs/utils.py:59-82 LowPassFilter
 * ---------------------------------------------------------------------------------------------- */
static void add_noise_to_omega(ENV *e, const REAL omega[3], REAL dt, po_rng *rng, REAL out[3]) {
  /* sensors.py:121-134; defaults :18-21 */
  const double gyro_noise_density = 0.000175, gyro_random_walk = 0.0105, corr = 1000.0;
  const double turn_on = PO_PI * 5 / 180;
  double sigma_g_d = gyro_noise_density / sqrt((double)dt);
  double sigma_b_g_d = sqrt(-(sigma_g_d * sigma_g_d) * (corr / 2) * (exp(-2 * (double)dt / corr) - 1));
  double pi_g_d = exp(-(double)dt / corr);
  for (int i = 0; i < 3; ++i) e->gyro_bias[i] = (REAL)pi_g_d * e->gyro_bias[i] + (REAL)sigma_b_g_d * rng_normal(rng);
  REAL n1[3], n2[3];
  for (int i = 0; i < 3; ++i) n1[i] = rng_normal(rng);
  for (int i = 0; i < 3; ++i) n2[i] = rng_normal(rng);
  for (int i = 0; i < 3; ++i)
    out[i] = omega[i] + e->gyro_bias[i] + (REAL)gyro_random_walk * n1[i] + (REAL)turn_on * n2[i];
}

static void sensor_add_noise(ENV *e, REAL dt, po_rng *rng, REAL pos[3], REAL vel[3], REAL rot[3], REAL omega[3]) {
  /* sensors.py:75-118; defaults :14-19 */
  const REAL pos_std = (REAL)0.002, pos_unif = (REAL)0.001, vel_std = (REAL)0.01, vel_unif = 0;
  const REAL q_std = (REAL)(PO_PI * 0.1 / 180), q_unif = (REAL)(PO_PI * 0.05 / 180);
  REAL g[3], u[3];
  for (int i = 0; i < 3; ++i) g[i] = (REAL)0 + pos_std * rng_normal(rng);
  for (int i = 0; i < 3; ++i) u[i] = -pos_unif + (pos_unif - (-pos_unif)) * rng_uniform01(rng);
  for (int i = 0; i < 3; ++i) pos[i] = e->xyz[i] + (g[i] + u[i]);                 /* :84-88 */
  for (int i = 0; i < 3; ++i) g[i] = (REAL)0 + vel_std * rng_normal(rng);
  for (int i = 0; i < 3; ++i) u[i] = -vel_unif + (vel_unif - (-vel_unif)) * rng_uniform01(rng);
  for (int i = 0; i < 3; ++i) vel[i] = e->xyz_dot[i] + g[i] + u[i];               /* :91-95 */
  add_noise_to_omega(e, e->rpy_dot, dt, rng, omega);                               /* :98 */
  for (int i = 0; i < 3; ++i) g[i] = (REAL)0 + q_std * rng_normal(rng);
  for (int i = 0; i < 3; ++i) u[i] = -q_unif + (q_unif - (-q_unif)) * rng_uniform01(rng);
  const REAL lo[3] = {(REAL)-PO_PI, (REAL)(-PO_PI / 2), (REAL)-PO_PI};
  const REAL hi[3] = {(REAL)PO_PI, (REAL)(PO_PI / 2), (REAL)PO_PI};
  for (int i = 0; i < 3; ++i) rot[i] = clipr(e->rpy[i] + (g[i] + u[i]), lo[i], hi[i]); /* :101-109 */
  /* accelerometer noise :111-116 draws 2x3 normals whose result is unused by the envs */
  for (int i = 0; i < 6; ++i) (void)rng_normal(rng);
}

/* target update shared by Circle/TakeOff compute_observation: circle.py:130-131, takeoff.py:108-109;
 * reference tables circle.py:45-56 (300 points, radius .25), takeoff.py:43-47 (z = k/300). */
static void update_target(const po_config *c, ENV *e) {
  if (c->task == PO_TASK_CIRCLE) {
    int t = (e->iteration / c->aggregate_phy_steps + e->ref_offset) % c->ref_points;
    double ts = 2 * PO_PI * (double)t / c->ref_points;
    e->target_pos[0] = (REAL)(0.25 * (1 - cos(ts)));
    e->target_pos[1] = (REAL)(0.25 * sin(ts));
    e->target_pos[2] = (REAL)1.;
  } else if (c->task == PO_TASK_TAKEOFF) {
    int t = e->iteration < 299 ? e->iteration : 299;
    e->target_pos[0] = 0; e->target_pos[1] = 0;
    e->target_pos[2] = (REAL)((double)t / 300);
  }
}

/* compute_observation: envs/hover.py:131-163, envs/circle.py:128-177, envs/takeoff.py:107-149;
 * get_state envs/agents.py:339-348.  Returns |o|. */
int SUF(po_compute_observation)(const po_config *c, ENV *e, po_rng *rng, REAL *o) {
  int n = 0;
  update_target(c, e);
  if (c->observation_noise > 0) {
    REAL xyz[3], vel[3], rpy[3], omega[3], quat[4];
    if (e->iteration % c->obs_rate == 0) {
      sensor_add_noise(e, (REAL)c->time_step, rng, xyz, vel, rpy, omega);
      SUF(po_quat_from_euler)(rpy, quat);
      for (int i = 0; i < 3; ++i) e->kf_state[i] = xyz[i];
      for (int i = 0; i < 4; ++i) e->kf_state[3 + i] = quat[i];
      for (int i = 0; i < 3; ++i) e->kf_state[7 + i] = vel[i];
      for (int i = 0; i < 3; ++i) e->kf_state[10 + i] = omega[i];
      for (int i = 0; i < 4; ++i) e->kf_state[13 + i] = e->last_action[i];
    } else {
      for (int i = 0; i < 3; ++i) xyz[i] = e->kf_state[i];
      for (int i = 0; i < 4; ++i) quat[i] = e->kf_state[3 + i];
      for (int i = 0; i < 3; ++i) vel[i] = e->kf_state[7 + i];
      add_noise_to_omega(e, e->rpy_dot, (REAL)c->time_step, rng, omega);
    }
    /* gyro low-pass: LowPassFilter(gain 1, T = 2/sim_freq, T_s = 1/sim_freq) base.py:109-110,
     * utils.py:76-79 */
    const REAL ratio = (REAL)0.5;
    for (int i = 0; i < 3; ++i) {
      e->lpf[i] = ((REAL)1 - ratio) * e->lpf[i] + (REAL)1 * ratio * omega[i];
      omega[i] = e->lpf[i];
    }
    for (int i = 0; i < 3; ++i) o[n++] = xyz[i];
    for (int i = 0; i < 4; ++i) o[n++] = quat[i];
    for (int i = 0; i < 3; ++i) o[n++] = vel[i];
    for (int i = 0; i < 3; ++i) o[n++] = omega[i];
    if (c->task == PO_TASK_TAKEOFF)
      for (int i = 0; i < 4; ++i) o[n++] = e->last_action[i];
    if (c->task != PO_TASK_HOVER)
      for (int i = 0; i < 3; ++i) o[n++] = e->target_pos[i] - xyz[i];
  } else {
    for (int i = 0; i < 3; ++i) o[n++] = e->xyz[i];
    for (int i = 0; i < 4; ++i) o[n++] = e->quat[i];
    for (int i = 0; i < 3; ++i) o[n++] = e->xyz_dot[i];
    for (int i = 0; i < 3; ++i) o[n++] = e->rpy_dot[i];
    if (c->task != PO_TASK_CIRCLE)
      for (int i = 0; i < 4; ++i) o[n++] = e->last_action[i];
    if (c->task != PO_TASK_HOVER)
      for (int i = 0; i < 3; ++i) o[n++] = e->target_pos[i] - e->xyz[i];
  }
  return n;
}

int SUF(po_obs_dim)(const po_config *c) {
  int o;
  if (c->observation_noise > 0) o = (c->task == PO_TASK_HOVER) ? 13 : (c->task == PO_TASK_CIRCLE ? 16 : 20);
  else o = (c->task == PO_TASK_HOVER) ? 17 : (c->task == PO_TASK_CIRCLE ? 16 : 20);
  return PO_HIST * (o + 4);
}

static REAL norm3(const REAL *v) { return R_SQRT(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* compute_done: envs/hover.py:89-101, envs/circle.py:116-120, envs/takeoff.py:96-100 */
int SUF(po_compute_done)(const po_config *c, const ENV *e) {
  if (c->task == PO_TASK_HOVER) {
    const REAL d = (REAL)(PO_PI * 60 / 180);
    int z_limit = e->xyz[2] < (REAL)0.2;
    int rpy_limit = (R_FABS(e->rpy[0]) > d) || (R_FABS(e->rpy[1]) > d);
    int dot_limit = 0;
    for (int i = 0; i < 3; ++i)
      if ((REAL)180 * R_FABS(e->rpy_dot[i]) / (REAL)PO_PI > (REAL)300) dot_limit = 1; /* rad2deg utils.py:17 */
    return rpy_limit || dot_limit || z_limit;
  } else if (c->task == PO_TASK_CIRCLE) {
    REAL d[3] = {e->xyz[0] - e->target_pos[0], e->xyz[1] - e->target_pos[1], e->xyz[2] - e->target_pos[2]};
    return norm3(d) > (REAL)0.25;
  }
  return 0;
}

/* compute_reward: envs/hover.py:169-187, envs/circle.py:183-204, envs/takeoff.py:155-174 */
REAL SUF(po_compute_reward)(const po_config *c, const ENV *e, const REAL a[4]) {
  REAL act_diff[4], nca[4];
  for (int i = 0; i < 4; ++i) {
    /* hover.py:171 / takeoff.py:157 use drone.last_action (== a, just set); circle.py:186 uses
     * env.last_action (the previous action, base.py:474) */
    act_diff[i] = a[i] - ((c->task == PO_TASK_CIRCLE) ? e->env_last_action[i] : e->last_action[i]);
    nca[i] = (REAL)0.5 * (clipr(a[i], -1, 1) + (REAL)1);
  }
  REAL penalty_action = (REAL)c->penalty_action * R_SQRT(nca[0] * nca[0] + nca[1] * nca[1] + nca[2] * nca[2] + nca[3] * nca[3]);
  REAL penalty_action_rate = (REAL)c->ARP * R_SQRT(act_diff[0] * act_diff[0] + act_diff[1] * act_diff[1] + act_diff[2] * act_diff[2] + act_diff[3] * act_diff[3]);
  REAL penalty_rpy = (REAL)c->penalty_angle * norm3(e->rpy);
  REAL penalty_spin = (REAL)c->penalty_spin * norm3(e->rpy_dot);
  REAL penalty_terminal = SUF(po_compute_done)(c, e) ? (REAL)c->penalty_terminal : (REAL)0;
  /* takeoff.py:165 multiplies by penalty_ACTION (quirk) */
  REAL penalty_velocity = (REAL)((c->task == PO_TASK_TAKEOFF) ? c->penalty_action : c->penalty_velocity) * norm3(e->xyz_dot);
  REAL penalties = ((((((REAL)0 + penalty_rpy) + penalty_action_rate) + penalty_spin) + penalty_velocity) + penalty_action) + penalty_terminal;
  REAL d[3] = {e->xyz[0] - e->target_pos[0], e->xyz[1] - e->target_pos[1], e->xyz[2] - e->target_pos[2]};
  REAL reward = -norm3(d) - penalties;
  if (c->task == PO_TASK_TAKEOFF && e->xyz[2] < (REAL)0.08) reward -= (REAL)1.; /* takeoff.py:172-173 */
  return reward;
}

/* compute_info -> cost: envs/hover.py:103-129 (state[10:13] is rpy_dot and state[13:16] is
 * last_action[0:3] in the get_state layout -- reproduced as is); circle.py:122-126 and
 * takeoff.py:102-105 return 0. */
REAL SUF(po_compute_cost)(const po_config *c, const ENV *e) {
  if (c->task != PO_TASK_HOVER) return 0;
  REAL cost = 0;
  const REAL rp_lim = (REAL)(PO_PI * 10 / 180), dot_lim = (REAL)(PO_PI * 200 / 180);
  if (R_FABS(e->xyz[0]) > (REAL)0.10 || R_FABS(e->xyz[1]) > (REAL)0.10 || e->xyz[2] > (REAL)1.20) cost = 1;
  if (R_FABS(e->rpy[0]) > rp_lim || R_FABS(e->rpy[1]) > rp_lim) cost = 1;
  for (int i = 0; i < 3; ++i) if (R_FABS(e->rpy_dot[i]) > (REAL)0.25) cost = 1;
  for (int i = 0; i < 3; ++i) if (R_FABS(e->last_action[i]) > dot_lim) cost = 1;
  return cost;
}

/* envs/base.py:303-319 compute_history */
static void compute_history(const po_config *c, ENV *e, po_rng *rng, REAL *obs) {
  REAL o[PO_MAX_OBS];
  int n = SUF(po_compute_observation)(c, e, rng, o);
  e->obs_len = n;
  memcpy(e->obs_hist[0], e->obs_hist[1], sizeof(REAL) * PO_MAX_OBS); /* deque(maxlen=2).append */
  memcpy(e->obs_hist[1], o, sizeof(REAL) * n);
  int k = 0;
  for (int h = 0; h < PO_HIST; ++h) {
    for (int i = 0; i < n; ++i) obs[k++] = e->obs_hist[h][i];
    /* an entry that is still the view action_buffer[-1, :] shows the buffer's CURRENT content */
    const REAL *ah = e->hist_alias[h] ? e->action_buffer[e->buf_size - 1] : e->act_hist[h];
    for (int i = 0; i < 4; ++i) obs[k++] = ah[i];
  }
  memcpy(e->act_hist[0], e->act_hist[1], sizeof(REAL) * 4);
  e->hist_alias[0] = e->hist_alias[1];
  memcpy(e->act_hist[1], e->last_action, sizeof(REAL) * 4); /* append(drone.last_action): the object itself */
  e->hist_alias[1] = e->last_action_alias;
}

/* envs/base.py:433-475 DroneBaseEnv.step + gymnasium TimeLimit (__init__.py:8-50) */
void SUF(po_step)(const po_config *c, ENV *e, const REAL a[4], po_rng *rng, REAL *obs, REAL *reward,
                  int32_t *terminated, int32_t *truncated, REAL *cost) {
  REAL scratch[PO_MAX_OBS];
  for (int s = 0; s < c->aggregate_phy_steps; ++s) {
    SUF(po_step_forward)(c, e, a, rng);                    /* :461 */
    (void)SUF(po_compute_observation)(c, e, rng, scratch); /* :464 (advances noise/LPF state) */
    e->iteration += 1;                                     /* :465 */
  }
  compute_history(c, e, rng, obs);                         /* :468 */
  *reward = SUF(po_compute_reward)(c, e, a);               /* :470 */
  *cost = SUF(po_compute_cost)(c, e);                      /* :471 */
  *terminated = SUF(po_compute_done)(c, e);                /* :472 */
  for (int i = 0; i < 4; ++i) e->env_last_action[i] = a[i]; /* :474 */
  e->elapsed_steps += 1;
  *truncated = e->elapsed_steps >= c->max_episode_steps;
}

/* envs/base.py:382-431 reset, envs/agents.py:377-386 drone.reset, task_specific_reset
 * (hover.py:192-243, circle.py:213-277, takeoff.py:179-212), apply_domain_randomization
 * (base.py:239-296), update_information (agents.py:434-453).  `s` holds the values the reference
 * drew from np.random in that order. */
void SUF(po_reset)(const po_config *c, ENV *e, const po_reset_sample *s, po_rng *rng, REAL *obs) {
  po_constants k;
  po_get_constants(&k);
  e->iteration = 0;            /* base.py:397 */
  e->elapsed_steps = 0;        /* TimeLimit.reset */
  /* drone.reset(): agents.py:380-386 */
  for (int i = 0; i < 4; ++i) { e->x[i] = 0; e->y[i] = 0; e->last_action[i] = 0; }
  for (int i = 0; i < 3; ++i) { e->rate_int[i] = e->rate_err[i] = e->att_int[i] = e->att_err[i] = 0; } /* control.reset() */
  e->action_idx = 0;                                        /* :384 */
  memset(e->action_buffer, 0, sizeof(e->action_buffer));    /* :385 */
  e->last_action_alias = 1;                                 /* :386 last_action = action_buffer[-1, :] (a view) */

  /* ---- task_specific_reset ---- */
  REAL pos[3] = {(REAL)c->init_xyz[0], (REAL)c->init_xyz[1], (REAL)c->init_xyz[2]};
  REAL vel[3] = {(REAL)c->init_xyz_dot[0], (REAL)c->init_xyz_dot[1], (REAL)c->init_xyz_dot[2]};
  REAL w_s[3] = {(REAL)c->init_rpy_dot[0], (REAL)c->init_rpy_dot[1], (REAL)c->init_rpy_dot[2]};
  REAL quat[4], init_rpy[3] = {(REAL)c->init_rpy[0], (REAL)c->init_rpy[1], (REAL)c->init_rpy[2]};
  SUF(po_quat_from_euler)(init_rpy, quat); /* init_quaternion = Q(init_rpy) base.py:89 */
  if (c->enable_reset_distribution) {
    REAL rpy_s[3];
    if (c->task == PO_TASK_HOVER) {
      /* float32 array += float64 draws: result rounded to float32 (hover.py:44,195,203) */
      for (int i = 0; i < 3; ++i) pos[i] = (REAL)(float)((double)(float)c->init_xyz[i] + s->pos_offset[i]);
      for (int i = 0; i < 3; ++i) rpy_s[i] = (REAL)s->rpy[i];
      SUF(po_quat_from_euler)(rpy_s, quat);                        /* :209 */
      for (int i = 0; i < 3; ++i) vel[i] = vel[i] + (REAL)s->vel[i]; /* :213 */
      for (int i = 0; i < 3; ++i) w_s[i] = (i < 2) ? w_s[i] + (REAL)s->omega[i] : (REAL)s->omega[i]; /* :216-217 */
    } else if (c->task == PO_TASK_CIRCLE) {
      e->ref_offset = s->ref_offset;                               /* circle.py:225 */
      double ts = 2 * PO_PI * (double)s->ref_offset / c->ref_points;
      e->target_pos[0] = (REAL)(0.25 * (1 - cos(ts)));
      e->target_pos[1] = (REAL)(0.25 * sin(ts));
      e->target_pos[2] = (REAL)1.;
      for (int i = 0; i < 3; ++i) pos[i] = e->target_pos[i] + (REAL)s->pos_offset[i]; /* :227-230 */
      for (int i = 0; i < 3; ++i) rpy_s[i] = (REAL)s->rpy[i];
      SUF(po_quat_from_euler)(rpy_s, quat);                        /* :236 */
      for (int i = 0; i < 3; ++i) vel[i] = vel[i] + (REAL)s->vel[i];
      for (int i = 0; i < 3; ++i) w_s[i] = (REAL)s->omega[i]; /* circle.py:246-247 overwrite */
    } else {
      for (int i = 0; i < 2; ++i) pos[i] = (REAL)(float)((double)(float)c->init_xyz[i] + s->pos_offset[i]); /* takeoff.py:188 */
      rpy_s[0] = 0; rpy_s[1] = 0; rpy_s[2] = (REAL)s->rpy[2];      /* :191 */
      SUF(po_quat_from_euler)(rpy_s, quat);
    }
    if (c->task != PO_TASK_TAKEOFF) {
      /* hover.py:223-229 / circle.py:251-257 */
      for (int i = 0; i < 4; ++i) e->x[i] = (REAL)s->motor_x[i];
      for (int i = 0; i < 4; ++i) e->y[i] = e->K[i] * e->x[i];
      /* action_buffer = clip(normal(HOVER_ACTION, .02, (buf_size, 4))); last_action = its last row (a view) */
      for (int r = 0; r < e->buf_size - 1; ++r)
        for (int i = 0; i < 4; ++i) e->action_buffer[r][i] = clipr((REAL)s->action_buf[r][i], -1, 1);
      for (int i = 0; i < 4; ++i) e->action_buffer[e->buf_size - 1][i] = clipr((REAL)s->action[i], -1, 1);
      for (int i = 0; i < 4; ++i) e->last_action[i] = e->action_buffer[e->buf_size - 1][i];
    }
  } else if (c->task == PO_TASK_CIRCLE) {
    /* target_pos / ref_offset keep their previous values (circle.py:222-226 not executed) */
  }
  if (c->task == PO_TASK_TAKEOFF) { /* takeoff.py:209-212, unconditional */
    for (int i = 0; i < 4; ++i) { e->x[i] = 0; e->y[i] = e->K[i] * e->x[i]; e->last_action[i] = -1; }
    for (int r = 0; r < e->buf_size; ++r)
      for (int i = 0; i < 4; ++i) e->action_buffer[r][i] = -1;      /* action_buffer[:] = -1 */
  }
  /* bc.resetBaseVelocity(angularVelocity = R.T @ rpy_dot): hover.py:237-243 */
  REAL R[9], w_world[3];
  SUF(po_matrix_from_quat)(quat, R);
  for (int i = 0; i < 3; ++i) w_world[i] = R[0 + i] * w_s[0] + R[3 + i] * w_s[1] + R[6 + i] * w_s[2];

  /* ---- apply_domain_randomization: base.py:259-296 ---- */
  if (c->domain_randomization > 0) {
    e->dt = (REAL)s->dr_dt;                        /* :261-265 */
    e->m = (REAL)s->dr_m;                          /* :268 */
    for (int i = 0; i < 3; ++i) e->J[i] = (REAL)s->dr_J[i]; /* :269-272 */
    e->ftf0 = (REAL)s->dr_ftf0;                    /* :274-275 */
    e->ftf1 = (REAL)s->dr_ftf1;                    /* :276-277 */
    if (c->use_motor_dynamics) {                   /* :279-287 */
      REAL T_new[4], t2w_new[4], Ts = e->dt;
      for (int i = 0; i < 4; ++i) { T_new[i] = (REAL)s->dr_T[i]; t2w_new[i] = (REAL)s->dr_t2w[i]; }
      SUF(po_update_motor_dynamics)(e, T_new, &Ts, t2w_new);
    }
  }
  /* gyro_lpf.set(drone.rpy_dot) with the STALE rpy_dot of the previous episode: base.py:411 */
  for (int i = 0; i < 3; ++i) e->lpf[i] = e->rpy_dot[i];

  /* ---- drone.update_information(): agents.py:434-453 ---- */
  for (int i = 0; i < 3; ++i) e->xyz[i] = pos[i];
  SUF(po_bullet_readback_quat)(quat, e->quat);                     /* :443 the quaternion as Bullet returns it */
  SUF(po_euler_from_quat)(e->quat, e->rpy);                        /* :446 */
  for (int i = 0; i < 3; ++i) e->xyz_dot[i] = vel[i];
  SUF(po_matrix_from_quat)(e->quat, R);                            /* :452 */
  for (int i = 0; i < 3; ++i) e->rpy_dot[i] = R[0 + i] * w_world[0] + R[3 + i] * w_world[1] + R[6 + i] * w_world[2]; /* :453 */

  /* ---- history fill: base.py:417-431 ---- */
  REAL o[PO_MAX_OBS];
  int n = SUF(po_compute_observation)(c, e, rng, o);               /* :419 */
  e->obs_len = n;
  for (int h = 0; h < PO_HIST; ++h) memcpy(e->obs_hist[h], o, sizeof(REAL) * n);
  for (int h = 0; h < PO_HIST; ++h) memcpy(e->act_hist[h], e->last_action, sizeof(REAL) * 4);
  for (int h = 0; h < PO_HIST; ++h) e->hist_alias[h] = e->last_action_alias;       /* :425-426 the view itself */
  for (int i = 0; i < 4; ++i) e->env_last_action[i] = e->last_action[i]; /* :428 */
  compute_history(c, e, rng, obs);                                 /* :429 */
}

/* ------------------------------------------------------------------------------------------------
 * In-kernel reset sampler of the NEW framework, restated (see DESIGN.md "RNG contract").
 * counter = (env_id, tick_lo, tick_hi, block), key = (seed_lo, seed_hi).
 * ---------------------------------------------------------------------------------------------- */
static REAL u01(uint32_t x) { return (REAL)(x >> 8) * (REAL)(1.0 / 16777216.0); }
static REAL urange(uint32_t x, REAL lo, REAL hi) { return lo + (hi - lo) * u01(x); }
static void box_muller(uint32_t a, uint32_t b, REAL *z0, REAL *z1) {
  REAL u1 = (REAL)((a >> 8) + 1u) * (REAL)(1.0 / 16777216.0);
  REAL u2 = u01(b);
  REAL r = R_SQRT((REAL)-2 * R_LOG(u1));
  REAL ang = (REAL)(2 * PO_PI) * u2;
  *z0 = r * R_COS(ang);
  *z1 = r * R_SIN(ang);
}

/* per-step noise: one word per Box-Muller pair (radius: high 20 bits, angle: low 12 bits at the MIDPOINTS
 * (j + 1/2) / 4096 -- an angle grid that contains 0, 1/4, 1/2, 3/4 of a turn puts an atom of mass 1/2048 at z = 0), one
 * 16-bit half word per uniform -- csrc/pds_device.h box_muller_word / u01_lo16 / u01_hi16 */
static void box_muller_word(uint32_t w, REAL *z0, REAL *z1) {
  REAL u1 = (REAL)((w >> 12) + 1u) * (REAL)(1.0 / 1048576.0);
  REAL u2 = ((REAL)(w & 0xFFFu) + (REAL)0.5) * (REAL)(1.0 / 4096.0);
  REAL r = R_SQRT((REAL)-2 * R_LOG(u1));
  REAL ang = (REAL)(2 * PO_PI) * u2;
  *z0 = r * R_COS(ang);
  *z1 = r * R_SIN(ang);
}
static REAL u01_lo16(uint32_t w) { return (REAL)(w & 0xFFFFu) * (REAL)(1.0 / 65536.0); }
static REAL u01_hi16(uint32_t w) { return (REAL)(w >> 16) * (REAL)(1.0 / 65536.0); }

void SUF(po_philox_reset_sample)(const po_config *c, uint64_t seed, uint64_t env_id, uint64_t tick,
                                 po_reset_sample *s) {
  po_constants k;
  po_get_constants(&k);
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t r[9][4];
  for (uint32_t b = 0; b < 9; ++b) {
    uint32_t ctr[4] = {(uint32_t)env_id, (uint32_t)tick, (uint32_t)(tick >> 32), b};
    po_philox4x32_10(ctr, key, r[b]);
  }
  memset(s, 0, sizeof(*s));
  const REAL D2R = (REAL)(PO_PI / 180);
  REAL pos_lim, rp_lim, yaw_lim, vel_lim, w_lim, wz_lim;
  if (c->task == PO_TASK_HOVER) {
    pos_lim = (REAL)0.25; rp_lim = (REAL)(PO_PI / 6); yaw_lim = (REAL)(2 * PO_PI);
    vel_lim = (REAL)0.1; w_lim = 200 * D2R; wz_lim = 20 * D2R;
  } else if (c->task == PO_TASK_CIRCLE) {
    pos_lim = (REAL)0.05; rp_lim = 20 * D2R; yaw_lim = (REAL)(0.1 * PO_PI);
    vel_lim = (REAL)0.1; w_lim = 50 * D2R; wz_lim = 20 * D2R;
  } else {
    pos_lim = (REAL)0.25; rp_lim = 0; yaw_lim = (REAL)PO_PI; vel_lim = 0; w_lim = 0; wz_lim = 0;
  }
  s->pos_offset[0] = urange(r[0][0], -pos_lim, pos_lim);
  s->pos_offset[1] = urange(r[0][1], -pos_lim, pos_lim);
  s->pos_offset[2] = (c->task == PO_TASK_TAKEOFF) ? 0 : urange(r[0][2], -pos_lim, pos_lim);
  s->rpy[0] = urange(r[0][3], -rp_lim, rp_lim);
  s->rpy[1] = urange(r[1][0], -rp_lim, rp_lim);
  s->rpy[2] = urange(r[1][1], -yaw_lim, yaw_lim);
  s->vel[0] = urange(r[1][2], -vel_lim, vel_lim);
  s->vel[1] = urange(r[1][3], -vel_lim, vel_lim);
  s->vel[2] = urange(r[2][0], -vel_lim, vel_lim);
  s->omega[0] = urange(r[2][1], -w_lim, w_lim);
  s->omega[1] = urange(r[2][2], -w_lim, w_lim);
  s->omega[2] = urange(r[2][3], -wz_lim, wz_lim);
  REAL z[8];
  box_muller(r[3][0], r[3][1], &z[0], &z[1]);
  box_muller(r[3][2], r[3][3], &z[2], &z[3]);
  box_muller(r[4][0], r[4][1], &z[4], &z[5]);
  box_muller(r[4][2], r[4][3], &z[6], &z[7]);
  for (int i = 0; i < 4; ++i) s->motor_x[i] = (REAL)k.HOVER_X + (REAL)0.02 * z[i];
  for (int i = 0; i < 4; ++i) s->action[i] = (REAL)k.HOVER_ACTION + (REAL)0.02 * z[4 + i];
  const REAL f = (REAL)c->domain_randomization;
#define DRV(x, d) urange((x), (REAL)(d) - f * (REAL)(d), (REAL)(d) + f * (REAL)(d))
  s->dr_dt = DRV(r[5][0], c->time_step);
  s->dr_m = DRV(r[5][1], k.M);
  s->dr_J[0] = DRV(r[5][2], k.IXX);
  s->dr_J[1] = DRV(r[5][3], k.IYY);
  s->dr_J[2] = DRV(r[6][0], k.IZZ);
  s->dr_ftf0 = DRV(r[6][1], k.FORCE_TORQUE_FACTOR_0);
  s->dr_ftf1 = DRV(r[6][2], k.FORCE_TORQUE_FACTOR_1);
  s->ref_offset = (int32_t)(((uint64_t)r[6][3] * (uint32_t)c->ref_points) >> 32);
  for (int i = 0; i < 4; ++i) s->dr_T[i] = DRV(r[7][i], c->motor_time_constant);
  for (int i = 0; i < 4; ++i) s->dr_t2w[i] = DRV(r[8][i], k.THRUST2WEIGHT_RATIO);
#undef DRV
  /* rows 0..B-2 of the latency action buffer: blocks 9.. (csrc/pds_reset.h kBlkLatRows) */
  if (c->use_latency && c->latency >= c->time_step) {
    int B = (int)SUF(py_float_floordiv)(c->latency, c->time_step);
    if (B < 1) B = 1;
    if (B > PO_MAX_LAT) B = PO_MAX_LAT;
    for (int row = 0; row < B - 1; ++row) {
      uint32_t ctr[4] = {(uint32_t)env_id, (uint32_t)tick, (uint32_t)(tick >> 32), 9u + (uint32_t)row};
      uint32_t w[4];
      po_philox4x32_10(ctr, key, w);
      REAL y[4];
      box_muller(w[0], w[1], &y[0], &y[1]);
      box_muller(w[2], w[3], &y[2], &y[3]);
      for (int i = 0; i < 4; ++i) s->action_buf[row][i] = (REAL)k.HOVER_ACTION + (REAL)0.02 * y[i];
    }
  }
}

/* In-kernel NOISE streams of the new framework (Philox4x32-7; block ids as in csrc/pds_reset.h),
 * laid out in the reference's numpy draw order so that the restated SensorNoise / OUNoise code
 * above consumes them unchanged.  Draws whose value never reaches the state or the observation
 * (position / velocity / angle noise of the discarded add_noise call, accelerometer noise) are not
 * generated by the kernel; they are filled with neutral values here. */
#define PO_BLK_RESET_NOISE 32u
#define PO_BLK_OBS_NOISE 64u
#define PO_BLK_SUB_NOISE 128u
#define PO_BLK_SUB_NOISE_X 256u

static void philox_words(uint64_t seed, uint64_t env_id, uint64_t tick, uint32_t blk0, int nblk, uint32_t *w) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (int b = 0; b < nblk; ++b) {
    uint32_t ctr[4] = {(uint32_t)env_id, (uint32_t)tick, (uint32_t)(tick >> 32), blk0 + (uint32_t)b};
    po_philox4x32(ctr, key, 7, w + 4 * b);
  }
}

/* one add_noise call that reaches the observation: 3 blocks -> z[24] (numpy order: pos3 vel3 bias3
 * rw3 to3 theta3 acc6), u[9] (pos3 vel3 theta3).  Kernel layout (csrc/pds_reset.h words_to_noise8, round 5): words 0..4 ->
 * normals n[0..9], words 5..7 -> six 16-bit uniforms, words 8..11 -> normals n[10..17]; n = pos3 vel3 theta3 bias3 rw3 to3. */
static void obs_call_streams(uint64_t seed, uint64_t env_id, uint64_t tick, uint32_t blk0, double *z, double *u) {
  uint32_t w[12];
  philox_words(seed, env_id, tick, blk0, 3, w);
  REAL n[18];
  for (int p = 0; p < 5; ++p) box_muller_word(w[p], &n[2 * p], &n[2 * p + 1]);
  for (int p = 5; p < 9; ++p) box_muller_word(w[p + 3], &n[2 * p], &n[2 * p + 1]);
  for (int i = 0; i < 6; ++i) z[i] = (double)n[i];           /* pos3 vel3 */
  for (int i = 0; i < 9; ++i) z[6 + i] = (double)n[9 + i];   /* bias3 rw3 to3 */
  for (int i = 0; i < 3; ++i) z[15 + i] = (double)n[6 + i];  /* theta3 */
  for (int i = 18; i < 24; ++i) z[i] = 0;
  for (int i = 0; i < 3; ++i) u[3 + i] = 0.5;
  u[0] = (double)u01_lo16(w[5]); u[1] = (double)u01_hi16(w[5]);
  u[2] = (double)u01_lo16(w[6]); u[6] = (double)u01_hi16(w[6]);
  u[7] = (double)u01_lo16(w[7]); u[8] = (double)u01_hi16(w[7]);
}

/* streams of one env.step() (aggregate_phy_steps sub-steps): per sub-step OU z4 + discarded call, then
 * the observing call.  A call at an iteration that is a multiple of obs_rate is a full add_noise
 * (z24, u9); otherwise only add_noise_to_omega draws (z9) -- envs/hover.py:134-156.  `iteration` is
 * the env's iteration counter before the step. */
static void step_streams(const po_config *c, uint64_t seed, uint64_t env_id, uint64_t tick, int iteration, double *z,
                         double *u, int *nz, int *nu) {
  int iz = 0, iu = 0;
  const int on = c->observation_noise > 0;
  const int R = c->obs_rate > 0 ? c->obs_rate : 1;
  for (int sub = 0; sub < c->aggregate_phy_steps; ++sub) {
    uint32_t w[8];
    philox_words(seed, env_id, tick, PO_BLK_SUB_NOISE + 2u * (uint32_t)sub, on ? 2 : 1, w);
    REAL n[14];
    for (int i = 0; i < 14; ++i) n[i] = 0;
    box_muller_word(w[0], &n[0], &n[1]);
    box_muller_word(w[1], &n[2], &n[3]);
    if (on) for (int p = 2; p < 7; ++p) box_muller_word(w[p], &n[2 * p], &n[2 * p + 1]);
    for (int i = 0; i < 4; ++i) z[iz++] = (double)n[i];             /* OUNoise randn(4) */
    if (on) {
      const int fresh = ((iteration + sub) % R) == 0;
      if (R == 1) {
        for (int i = 0; i < 6; ++i) z[iz++] = 0;                    /* pos, vel (discarded) */
        for (int i = 0; i < 9; ++i) z[iz++] = (double)n[4 + i];     /* bias, rw, turn-on */
        for (int i = 0; i < 9; ++i) z[iz++] = 0;                    /* theta, acc (discarded) */
        for (int i = 0; i < 9; ++i) u[iu++] = 0.5;
      } else if (fresh) {  /* the call refreshes the held state: its position / velocity / angle draws count */
        uint32_t x[8];
        philox_words(seed, env_id, tick, PO_BLK_SUB_NOISE_X + 2u * (uint32_t)sub, 2, x);
        REAL y[10];
        for (int p = 0; p < 5; ++p) box_muller_word(x[p], &y[2 * p], &y[2 * p + 1]);
        for (int i = 0; i < 6; ++i) z[iz++] = (double)y[i];         /* pos3, vel3 */
        for (int i = 0; i < 9; ++i) z[iz++] = (double)n[4 + i];     /* bias, rw, turn-on */
        for (int i = 0; i < 3; ++i) z[iz++] = (double)y[6 + i];     /* theta3 */
        for (int i = 0; i < 6; ++i) z[iz++] = 0;                    /* acc (unused) */
        u[iu++] = (double)u01_lo16(x[5]); u[iu++] = (double)u01_hi16(x[5]); u[iu++] = (double)u01_lo16(x[6]); /* pos */
        for (int i = 0; i < 3; ++i) u[iu++] = 0.5;                  /* vel (range 0) */
        u[iu++] = (double)u01_hi16(x[6]); u[iu++] = (double)u01_lo16(x[7]); u[iu++] = (double)u01_hi16(x[7]); /* theta */
      } else {
        for (int i = 0; i < 9; ++i) z[iz++] = (double)n[4 + i];     /* add_noise_to_omega only */
      }
    }
  }
  if (on) {
    const int fresh = ((iteration + c->aggregate_phy_steps) % R) == 0;
    double zz[24], uu[9];
    obs_call_streams(seed, env_id, tick, PO_BLK_OBS_NOISE, zz, uu);
    if (fresh) {
      for (int i = 0; i < 24; ++i) z[iz++] = zz[i];
      for (int i = 0; i < 9; ++i) u[iu++] = uu[i];
    } else {
      for (int i = 0; i < 9; ++i) z[iz++] = zz[6 + i];              /* bias, rw, turn-on of that call */
    }
  }
  *nz = iz; *nu = iu;
}

/* DroneBaseEnv.__init__ evaluates compute_observation() once to size the observation space
 * (envs/base.py:142): with sensor noise that call advances the gyro bias random walk by one draw
 * (envs/sensors.py:130-131) before the first reset; the bias is never reset afterwards.  In-kernel
 * counterpart: block PO_BLK_CTOR of tick 0 (csrc/pds_api.hip ctor_noise_kernel). */
#define PO_BLK_CTOR 16u
void SUF(po_ctor_noise_batch)(const po_config *c, ENV *envs, int64_t n, uint64_t seed) {
  if (c->observation_noise <= 0) return;
  const double gyro_noise_density = 0.000175, corr = 1000.0, dt = (double)c->time_step;
  const double sigma_g_d = gyro_noise_density / sqrt(dt);
  const REAL sb = (REAL)sqrt(-(sigma_g_d * sigma_g_d) * (corr / 2) * (exp(-2 * dt / corr) - 1));
  for (int64_t i = 0; i < n; ++i) {
    uint32_t w[4];
    philox_words(seed, (uint64_t)i, 0, PO_BLK_CTOR, 1, w);
    REAL z[4];
    box_muller_word(w[0], &z[0], &z[1]);
    box_muller_word(w[1], &z[2], &z[3]);
    for (int j = 0; j < 3; ++j) envs[i].gyro_bias[j] = sb * z[j];
  }
}

/* Batched drivers (OpenMP over envs) -- the `cpu_baseline` leg of bench.py and the lockstep
 * auto-reset semantics the HIP path is compared with: an env that terminates or truncates in this
 * step hands its last observation to final_obs and is reset in the same call. */
void SUF(po_reset_batch)(const po_config *c, ENV *envs, int64_t n, REAL *obs, uint64_t seed,
                         uint64_t tick, int nthreads) {
  const int D = SUF(po_obs_dim)(c);
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
  for (int64_t i = 0; i < n; ++i) {
    po_reset_sample s;
    SUF(po_philox_reset_sample)(c, seed, (uint64_t)i, tick, &s);
    double z[48], u[18];
    po_rng rng = {z, u, 0, 0, 0, 0};
    if (c->observation_noise > 0) {
      obs_call_streams(seed, (uint64_t)i, tick, PO_BLK_RESET_NOISE, z, u);
      obs_call_streams(seed, (uint64_t)i, tick, PO_BLK_RESET_NOISE + 3u, z + 24, u + 9);
      rng.nz = 48; rng.nu = 18;
    }
    SUF(po_reset)(c, &envs[i], &s, &rng, obs + i * D);
  }
  (void)nthreads;
}

void SUF(po_step_batch)(const po_config *c, ENV *envs, int64_t n, const REAL *actions, REAL *obs,
                        REAL *reward, uint8_t *terminated, uint8_t *truncated, REAL *cost,
                        REAL *final_obs, uint64_t seed, uint64_t tick, int auto_reset, int nthreads) {
  const int D = SUF(po_obs_dim)(c);
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
  for (int64_t i = 0; i < n; ++i) {
    int32_t term, trunc;
    double z[64 * 28 + 24], u[64 * 9 + 9]; /* aggregate_phy_steps <= 64 */
    po_rng rng = {z, u, 0, 0, 0, 0};
    if (c->observation_noise > 0 || c->motor_thrust_noise > 0) {
      int nz = 0, nu = 0;
      step_streams(c, seed, (uint64_t)i, tick, envs[i].iteration, z, u, &nz, &nu);
      rng.nz = nz; rng.nu = nu;
    }
    SUF(po_step)(c, &envs[i], actions + 4 * i, &rng, obs + i * D, &reward[i], &term, &trunc, &cost[i]);
    terminated[i] = (uint8_t)term;
    truncated[i] = (uint8_t)trunc;
    if (auto_reset && (term || trunc)) {
      if (final_obs) memcpy(final_obs + i * D, obs + i * D, sizeof(REAL) * D);
      po_reset_sample s;
      SUF(po_philox_reset_sample)(c, seed, (uint64_t)i, tick, &s);
      double zr[48], ur[18];
      po_rng rr = {zr, ur, 0, 0, 0, 0};
      if (c->observation_noise > 0) {
        obs_call_streams(seed, (uint64_t)i, tick, PO_BLK_RESET_NOISE, zr, ur);
        obs_call_streams(seed, (uint64_t)i, tick, PO_BLK_RESET_NOISE + 3u, zr + 24, ur + 9);
        rr.nz = 48; rr.nu = 18;
      }
      SUF(po_reset)(c, &envs[i], &s, &rr, obs + i * D);
    }
  }
  (void)nthreads;
}

/* Same static schedule as po_step_batch / po_reset_batch: the thread that will step an env is the one that
 * touches its struct first (first-touch NUMA placement; the caller passes untouched memory). */
void SUF(po_env_init_batch_mt)(const po_config *c, ENV *envs, int64_t n, int nthreads) {
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
  for (int64_t i = 0; i < n; ++i) SUF(po_env_init)(c, &envs[i]);
}
void SUF(po_env_init_batch)(const po_config *c, ENV *envs, int64_t n) { SUF(po_env_init_batch_mt)(c, envs, n, 1); }

int SUF(po_sizeof_env)(void) { return (int)sizeof(ENV); }
