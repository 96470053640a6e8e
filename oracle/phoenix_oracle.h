/*
 * phoenix_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference's SimplePhysics hot path
 * (SvenGronauer/phoenix-drone-simulation, files under phoenix_drone_simulation/envs/).
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it, and there
 * only as the checker / reported CPU baseline -- the product path (libpds_hip.so) never links,
 * loads or falls back to it.
 *
 * Parity status: PINNED against the reference itself -- tests/golden/ was generated in the build
 * container by importing the reference package from /root/reference (oracle/refgen/gen_golden.py;
 * the absent third-party modules pybullet / pybullet_utils / pybullet_data / gymnasium are replaced by
 * the stand-ins in oracle/refgen/standins/).  The only arithmetic that lives in an absent third-party
 * dependency is pybullet's getQuaternionFromEuler / getMatrixFromQuaternion / getEulerFromQuaternion
 * (pybullet is UNPINNED in the reference's setup.py:32); those three are restated from Bullet3's
 * published algorithm and cross-checked against scipy.spatial.transform.Rotation and against the
 * reference's own in-repo restatement envs/utils.py:32-56 and its test tests/test_quaternion.py:35-43.
 *
 * The file is compiled twice: PO_REAL=double (symbols *_f64, must match the reference to <=1e-12) and
 * PO_REAL=float (symbols *_f32, defines what the fp32 GPU kernel is compared with).
 */
#ifndef PHOENIX_ORACLE_H
#define PHOENIX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PO_TASK_HOVER 0
#define PO_TASK_CIRCLE 1
#define PO_TASK_TAKEOFF 2
#define PO_MAX_OBS 24 /* largest single observation o (TakeOff noise-free: 20) rounded up */
#define PO_HIST 2     /* observation_history_size default (envs/base.py:44) */
#define PO_MAX_LAT 8  /* rows of drone.action_buffer the oracle can hold (envs/agents.py:180-182) */

/* Mirror of the env ctor kwargs on the path (envs/base.py:26-48, envs/hover.py:7-24). */
typedef struct po_config {
  int32_t task;
  int32_t use_motor_dynamics;        /* envs/agents.py:284 branch; Simple agent default 0 (:493) */
  int32_t use_ground_effect;         /* extension: envs/physics.py:27-58 formula, default 0 */
  int32_t observation_noise;         /* >0: SensorNoise path (envs/hover.py:133) */
  int32_t aggregate_phy_steps;       /* envs/base.py:457 */
  int32_t enable_reset_distribution; /* envs/base.py:37 */
  int32_t max_episode_steps;         /* gymnasium TimeLimit, __init__.py:11 */
  int32_t obs_rate;                  /* sim_freq // observation_frequency (envs/base.py:108) */
  double domain_randomization;       /* <=0: off (envs/base.py:259) */
  double motor_thrust_noise;         /* OU sigma = 0.2*this (envs/agents.py:206) */
  double time_step;                  /* TIME_STEP = 1/sim_freq (envs/base.py:98) */
  double motor_time_constant;        /* envs/base.py:41 */
  double penalty_action, penalty_angle, penalty_spin, penalty_terminal, penalty_velocity, ARP;
  double target_pos[3];
  double init_xyz[3];
  double init_rpy[3], init_xyz_dot[3], init_rpy_dot[3]; /* envs/base.py:84-91; mutated by simopt callers */
  int32_t control_mode;              /* 0 PWM, 1 AttitudeRate, 2 Attitude (envs/control.py:91-287) */
  int32_t use_latency;               /* CrazyFlieAgent(use_latency=...), envs/agents.py:125,165; Simple agent: False (:492) */
  double latency;                    /* [s] envs/base.py:40; buf_size = max(1, int(latency // time_step)), agents.py:180 */
  int32_t ref_points;                /* Circle: num_ref_points = circle_time (3 s) * observation_frequency, circle.py:47-49 */
  int32_t pad3_;
} po_config;

/* Values drawn by one reset() in the reference's draw order (the *sampled values*, i.e. what
 * np.random.uniform/normal/randint returned).  envs/hover.py:192-243, envs/circle.py:213-277,
 * envs/takeoff.py:179-212, envs/base.py:239-296. */
typedef struct po_reset_sample {
  double pos_offset[3]; /* U(-lim,lim)^3 added to init / ref point (TakeOff: [0:2] only) */
  double rpy[3];        /* sampled Euler angles, yaw unwrapped */
  double vel[3];
  double omega[3];      /* sampled body rates before the R^T.R^T round trip */
  double motor_x[4];    /* N(HOVER_X, .02)^4 */
  double action[4];     /* N(HOVER_ACTION,.02)^4 before clipping */
  double dr_dt, dr_m, dr_J[3], dr_ftf0, dr_ftf1, dr_T[4], dr_t2w[4];
  int32_t ref_offset;   /* Circle: randint(0,300) */
  int32_t pad_;
  /* rows 0..B-2 of np.random.normal(HOVER_ACTION, .02, size=(buf_size, 4)) before clipping (hover.py:226-228);
   * the last row is `action` above (it becomes drone.last_action) */
  double action_buf[PO_MAX_LAT - 1][4];
} po_reset_sample;

/* Source of standard variates for the stochastic parts (OU thrust noise, SensorNoise).
 * mode 0: replay arrays recorded from the reference run (z ~ N(0,1) stream, u ~ U[0,1) stream). */
typedef struct po_rng {
  const double *z;
  const double *u;
  int64_t iz, iu, nz, nu;
} po_rng;

#define PO_DECL(SUF, REAL)                                                                          \
  typedef struct po_env##SUF {                                                                      \
    REAL xyz[3], rpy[3], quat[4], xyz_dot[3], rpy_dot[3];                                           \
    REAL x[4], y[4], last_action[4], env_last_action[4], pwm[4];                                    \
    REAL act_hist[PO_HIST][4];                                                                      \
    REAL obs_hist[PO_HIST][PO_MAX_OBS];                                                             \
    REAL target_pos[3];                                                                             \
    REAL dt, m, J[3], ftf0, ftf1, A[4], B[4], K[4], T[4], t2w[4], T_s;                             \
    REAL ou[4], gyro_bias[3], lpf[3], kf_state[17];                                                 \
    REAL rate_int[3], rate_err[3], att_int[3], att_err[3]; /* PID integrals / last errors */        \
    REAL action_buffer[PO_MAX_LAT][4]; /* drone.action_buffer, rows [0, buf_size) */                \
    int32_t iteration, ref_offset, elapsed_steps, obs_len;                                          \
    int32_t use_latency, buf_size, action_idx; /* envs/agents.py:165,180,183 */                    \
    /* action_history[h] / drone.last_action still ARE the numpy view action_buffer[-1, :] that    \
     * drone.reset() / task_specific_reset hand out (agents.py:386, hover.py:229) */                \
    int32_t hist_alias[PO_HIST], last_action_alias;                                                 \
  } po_env##SUF;                                                                                    \
  void po_set_latency##SUF(const po_config *c, po_env##SUF *e, double new_latency);                 \
  int po_sizeof_env##SUF(void);                                                                     \
  void po_quat_from_euler##SUF(const REAL rpy[3], REAL q[4]);                                       \
  void po_matrix_from_quat##SUF(const REAL q[4], REAL R[9]);                                        \
  void po_euler_from_quat##SUF(const REAL q[4], REAL rpy[3]);                                       \
  void po_bullet_readback_quat##SUF(const REAL qin[4], REAL q[4]);                                  \
  void po_env_init##SUF(const po_config *c, po_env##SUF *e);                                        \
  void po_apply_action##SUF(const po_config *c, po_env##SUF *e, const REAL a[4], po_rng *rng,      \
                            REAL forces[4], REAL *z_torque);                                        \
  void po_ground_effect##SUF(const po_env##SUF *e, const REAL forces[4], REAL ge[4]);              \
  void po_step_forward##SUF(const po_config *c, po_env##SUF *e, const REAL a[4], po_rng *rng);     \
  int po_compute_observation##SUF(const po_config *c, po_env##SUF *e, po_rng *rng, REAL *o);       \
  int po_compute_done##SUF(const po_config *c, const po_env##SUF *e);                               \
  REAL po_compute_reward##SUF(const po_config *c, const po_env##SUF *e, const REAL a[4]);           \
  REAL po_compute_cost##SUF(const po_config *c, const po_env##SUF *e);                              \
  void po_update_motor_dynamics##SUF(po_env##SUF *e, const REAL *T_new, const REAL *Ts_new,         \
                                     const REAL *t2w_new);                                          \
  int po_obs_dim##SUF(const po_config *c);                                                          \
  void po_reset##SUF(const po_config *c, po_env##SUF *e, const po_reset_sample *s, po_rng *rng,     \
                     REAL *obs);                                                                    \
  void po_step##SUF(const po_config *c, po_env##SUF *e, const REAL a[4], po_rng *rng, REAL *obs,    \
                    REAL *reward, int32_t *terminated, int32_t *truncated, REAL *cost);             \
  void po_philox_reset_sample##SUF(const po_config *c, uint64_t seed, uint64_t env_id,              \
                                   uint64_t tick, po_reset_sample *s);                              \
  void po_step_batch##SUF(const po_config *c, po_env##SUF *envs, int64_t n, const REAL *actions,    \
                          REAL *obs, REAL *reward, uint8_t *terminated, uint8_t *truncated,         \
                          REAL *cost, REAL *final_obs, uint64_t seed, uint64_t tick,                \
                          int auto_reset, int nthreads);                                            \
  void po_reset_batch##SUF(const po_config *c, po_env##SUF *envs, int64_t n, REAL *obs,             \
                           uint64_t seed, uint64_t tick, int nthreads);                             \
  void po_ctor_noise_batch##SUF(const po_config *c, po_env##SUF *envs, int64_t n, uint64_t seed);           \
  void po_env_init_batch##SUF(const po_config *c, po_env##SUF *envs, int64_t n);                  \
  void po_env_init_batch_mt##SUF(const po_config *c, po_env##SUF *envs, int64_t n, int nthreads);

PO_DECL(_f64, double)
PO_DECL(_f32, float)

void po_default_config(int task, po_config *c);
void po_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void po_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]);
int po_max_threads(void);

/* model constants (envs/assets/cf21x_sys_eq.urdf:10,16-17; envs/agents.py:142-156) */
typedef struct po_constants {
  double M, L, THRUST2WEIGHT_RATIO, IXX, IYY, IZZ, KF, KM, GND_EFF_COEFF, PROP_RADIUS;
  double FORCE_TORQUE_FACTOR_0, FORCE_TORQUE_FACTOR_1, G, GRAVITY, MAX_THRUST, MAX_TORQUE;
  double HOVER_X, HOVER_ACTION, MAX_RPM, GND_EFF_H_CLIP;
} po_constants;
void po_get_constants(po_constants *k);

#ifdef __cplusplus
}
#endif
#endif
