/*
 * pds.h -- C ABI of libpds_hip.so, the MI355X (gfx950) batched CrazyFlie SimplePhysics stepper.
 *
 * Drop-in boundary for ONE hot path of SvenGronauer/phoenix-drone-simulation: the per-env
 * `env.reset()` / `env.step(action)` loop of DroneHoverSimpleEnv-v0, DroneCircleSimpleEnv-v0 and
 * DroneTakeOffSimpleEnv-v0, replaced by a lockstep step over N independent environments whose state
 * lives SoA in HBM.  Reference interfaces each entry point replaces (paths relative to
 * phoenix_drone_simulation/ in the reference):
 *
 *   pds_default_config / pds_create  <- gym.make(id, **kwargs) -> DroneBaseEnv.__init__
 *                                       (envs/base.py:26-153, envs/hover.py:7-63, envs/circle.py:7-77,
 *                                        envs/takeoff.py:13-70, ids in __init__.py:8-50)
 *   pds_reset                        <- DroneBaseEnv.reset (envs/base.py:382-431) incl.
 *                                       task_specific_reset (hover.py:192-243, circle.py:213-277,
 *                                       takeoff.py:179-212) and apply_domain_randomization
 *                                       (base.py:239-296)
 *   pds_reset_from_samples           <- same, with the np.random draws supplied by the caller
 *                                       (parity injection; the reference draws from the global
 *                                       numpy stream)
 *   pds_step                         <- DroneBaseEnv.step (envs/base.py:433-475) =
 *                                       SimplePhysics.step_forward (envs/physics.py:130-200) +
 *                                       CrazyFlieAgent.apply_action (envs/agents.py:259-298) +
 *                                       compute_history/reward/info/done + TimeLimit truncation
 *   pds_get_state / pds_set_state    <- direct attribute access env.drone.{xyz,rpy,xyz_dot,rpy_dot,x,
 *                                       last_action,...} used by simopt/ and debug/ callers
 *   pds_step_k                       <- the open-loop replay loop `for i in range(T-1): sim_env.step(acs[i])`
 *                                       of simopt (simopt/pybullet.py:163-176), K steps per launch
 *   pds_set_latency                  <- CrazyFlieAgent.set_latency (envs/agents.py:388-404)
 *   pds_destroy                      <- env.close()
 *
 * All pointers named `d_*` are DEVICE pointers on the handle's device; tensors are row-major fp32.
 * Every entry point returns 0 on success or a negative PDS_E* code; pds_last_error() gives the text.
 * Launches are asynchronous on the caller's stream (`stream` is a hipStream_t passed as void*).
 * A handle is not thread-safe; different handles are independent; there is no global state.
 * There is NO CPU fallback: without a HIP device pds_create fails with PDS_ENODEVICE.
 */
#ifndef PDS_H
#define PDS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDS_VERSION 2

#define PDS_TASK_HOVER 0   /* DroneHoverSimpleEnv-v0   */
#define PDS_TASK_CIRCLE 1  /* DroneCircleSimpleEnv-v0  */
#define PDS_TASK_TAKEOFF 2 /* DroneTakeOffSimpleEnv-v0 */

#define PDS_CTRL_PWM 0           /* envs/control.py:91-100  */
#define PDS_CTRL_ATTITUDE_RATE 1 /* envs/control.py:120-191 */
#define PDS_CTRL_ATTITUDE 2      /* envs/control.py:194-287 */

#define PDS_MAX_LATENCY_STEPS 8 /* rows of the delayed-action ring: int(latency / time_step) <= 8 */
#define PDS_MAX_REF_POINTS 300  /* Circle: circle_time * observation_frequency <= 300 (envs/circle.py:49) */

#define PDS_OK 0
#define PDS_EINVAL -1
#define PDS_ENODEVICE -2
#define PDS_EHIP -3
#define PDS_ENOMEM -4
#define PDS_EUNSUPPORTED -5

/* Mirror of the reference's env kwargs on this path (same names, same defaults). */
typedef struct pds_config {
  int32_t struct_size; /* = sizeof(pds_config), set by pds_default_config */
  int32_t task;
  int64_t num_envs;    /* N envs stepped in lockstep on this device */
  int64_t env_id_base; /* global id of local env 0 (multi-GPU sharding; keys the in-kernel RNG) */
  uint64_t seed;
  int32_t device;                    /* HIP device ordinal */
  int32_t use_motor_dynamics;        /* first-order motor model, envs/agents.py:284-288; default 0 */
  int32_t use_ground_effect;         /* envs/physics.py:27-58 formula as opt-in; default 0 */
  int32_t observation_noise;         /* >0: SensorNoise path; reference default 1 */
  int32_t aggregate_phy_steps;       /* default 1 for the *Simple envs */
  int32_t enable_reset_distribution; /* default 1 */
  int32_t max_episode_steps;         /* TimeLimit, default 500 */
  int32_t auto_reset;                /* 1: envs that terminate/truncate are reset inside pds_step */
  double domain_randomization;       /* default 0.10; <=0 disables */
  double motor_thrust_noise;         /* default 0.05; OU sigma = 0.2*value */
  double time_step;                  /* 1/sim_freq = 0.01 */
  double motor_time_constant;        /* 0.080 s */
  double penalty_action, penalty_angle, penalty_spin, penalty_terminal, penalty_velocity, ARP;
  double target_pos[3];
  double init_xyz[3], init_rpy[3], init_xyz_dot[3], init_rpy_dot[3];
  int32_t control_mode;              /* PDS_CTRL_*: 'PWM' (default), 'AttitudeRate', 'Attitude' (envs/control.py) */
  int32_t use_latency;               /* CrazyFlieAgent(use_latency=...), envs/agents.py:125,165; the Simple agent passes
                                        False (agents.py:492), simopt flips it through set_latency; default 0 */
  double latency;                    /* [s] envs/base.py:40 (0.015); buf_size = max(1, latency // time_step) */
  int32_t observation_frequency;     /* envs/base.py:42 (100): obs_rate = sim_freq // observation_frequency
                                        (base.py:108) and Circle num_ref_points = 3 * observation_frequency */
  int32_t reserved_;
} pds_config;

typedef struct pds_handle pds_handle;

/* State fields for pds_get_state / pds_set_state: [N, width] row-major, fp32 unless noted. */
enum pds_field {
  PDS_F_POS = 0,         /* 3  drone.xyz */
  PDS_F_RPY = 1,         /* 3  drone.rpy */
  PDS_F_VEL = 2,         /* 3  drone.xyz_dot */
  PDS_F_OMEGA = 3,       /* 3  drone.rpy_dot (body rates) */
  PDS_F_QUAT = 4,        /* 4  drone.quaternion (derived: sign * Q(rpy); read-only) */
  PDS_F_MOTOR_X = 5,     /* 4  drone.x */
  PDS_F_LAST_ACTION = 6, /* 4  u(k-1): drone.last_action == action_history[-1] */
  PDS_F_PREV_ACTION = 7, /* 4  u(k-2): action_history[-2] */
  PDS_F_STEP_COUNT = 8,  /* 1  int32: env.step calls since reset (iteration / aggregate_phy_steps) */
  PDS_F_QUAT_SIGN = 9,   /* 1  int32 0/1: quaternion == -Q(rpy) (only right after reset) */
  PDS_F_REF_OFFSET = 10, /* 1  int32: Circle ref_offset */
  PDS_F_PARAMS = 11,     /* 6  dt, m, Jxx, Jyy, Jzz, force_torque_factor_1 */
  PDS_F_MOTOR_A = 12,    /* 4  drone.A (B = 1 - A) */
  PDS_F_MOTOR_K = 13,    /* 4  drone.K */
  PDS_F_OU = 14,         /* 4  thrust_noise.state */
  PDS_F_GYRO_BIAS = 15,  /* 3  sensor_noise.gyro_bias */
  PDS_F_GYRO_LPF = 16,   /* 3  gyro_lpf._x */
  PDS_F_NOISY_OBS = 17,  /* 10 observation_history[-1][0:10]: noisy xyz, quaternion, velocity */
  PDS_F_PID = 18,        /* 12 rate integral3, rate last_error3, attitude integral3, attitude last_error3 */
  PDS_F_ACTION_BUFFER = 19, /* 32 drone.action_buffer, PDS_MAX_LATENCY_STEPS rows x 4 (rows >= buf_size: 0) */
  PDS_F_ACTION_IDX = 20,    /* 1  int32: drone.action_idx */
  PDS_F_COUNT_ = 21
};

/* Layout of one row of `d_samples` for pds_reset_from_samples (the values np.random returned in
 * the reference's draw order; see oracle/phoenix_oracle.h po_reset_sample). */
#define PDS_SAMPLE_FLOATS 112
#define PDS_S_POS_OFFSET 0 /* 3 */
#define PDS_S_RPY 3        /* 3 */
#define PDS_S_VEL 6        /* 3 */
#define PDS_S_OMEGA 9      /* 3 */
#define PDS_S_MOTOR_X 12   /* 4 */
#define PDS_S_ACTION 16    /* 4 */
#define PDS_S_DR_DT 20
#define PDS_S_DR_M 21
#define PDS_S_DR_J 22      /* 3 */
#define PDS_S_DR_FTF0 25
#define PDS_S_DR_FTF1 26
#define PDS_S_DR_T 27      /* 4 */
#define PDS_S_DR_T2W 31    /* 4 */
#define PDS_S_REF_OFFSET 35
/* standard variates of the two SensorNoise.add_noise calls inside reset() (envs/base.py:419,429);
 * each call: PDS_N_OBS_* layout below (24 floats).  Only read when observation_noise > 0. */
#define PDS_S_NOISE_CALL0 36
#define PDS_S_NOISE_CALL1 60
/* use_latency with buf_size B > 1: rows 0..B-2 of np.random.normal(HOVER_ACTION, 0.02, size=(B, 4))
 * (envs/hover.py:226-228) before clipping; row B-1 is PDS_S_ACTION (it becomes drone.last_action). */
#define PDS_S_ACTION_BUF 84 /* (PDS_MAX_LATENCY_STEPS - 1) x 4 */

/* Layout of one row of `d_variates` for pds_step_with_variates: the STANDARD variates (z ~ N(0,1),
 * u ~ U[0,1)) one env.step() consumes, in the reference's draw order restricted to the draws whose
 * value reaches the state or the observation:
 *   OUNoise.noise (envs/utils.py:106) | first add_noise call (envs/base.py:464, only its gyro part
 *   survives at obs_rate 1) | second add_noise call (envs/base.py:468 -> compute_history) | the rest of the
 *   first call.  With obs_rate > 1 a call at an iteration that is not a multiple of obs_rate only draws the
 *   gyro part (add_noise_to_omega, envs/sensors.py:121-134); the unused entries are ignored.
 * With aggregate_phy_steps = A > 1 (envs/base.py:457-465: A x {step_forward; compute_observation}) a row is
 * A consecutive blocks of PDS_NOISE_FLOATS: block `sub` holds PDS_N_OU and the PDS_N_A_* entries of physics
 * sub-step `sub`; the one observing call (PDS_N_OBS) is read from block 0. */
#define PDS_NOISE_FLOATS 52
#define PDS_N_OU 0        /* 4 z */
#define PDS_N_A_BIAS 4    /* 3 z  gyro bias random walk   (envs/sensors.py:130) */
#define PDS_N_A_RW 7      /* 3 z  gyro_random_walk term   (envs/sensors.py:133) */
#define PDS_N_A_TO 10     /* 3 z  turn-on-bias term       (envs/sensors.py:134) */
#define PDS_N_OBS 13      /* second call, 24 floats: */
#define PDS_N_OBS_POS_Z 0  /* 3 z */
#define PDS_N_OBS_POS_U 3  /* 3 u */
#define PDS_N_OBS_VEL_Z 6  /* 3 z */
#define PDS_N_OBS_BIAS 9   /* 3 z */
#define PDS_N_OBS_RW 12    /* 3 z */
#define PDS_N_OBS_TO 15    /* 3 z */
#define PDS_N_OBS_TH_Z 18  /* 3 z */
#define PDS_N_OBS_TH_U 21  /* 3 u */
/* position / velocity / angle draws of the FIRST call: they reach the observation only through the held
 * "Kalman" state when obs_rate = sim_freq // observation_frequency > 1 (envs/hover.py:134-156) */
#define PDS_N_A_POS_Z 37  /* 3 z */
#define PDS_N_A_POS_U 40  /* 3 u */
#define PDS_N_A_VEL_Z 43  /* 3 z */
#define PDS_N_A_TH_Z 46   /* 3 z */
#define PDS_N_A_TH_U 49   /* 3 u */

int pds_version(void);

/* Fill `cfg` with the reference ctor defaults of `task` (N = 1, device 0, auto_reset = 1). */
int pds_default_config(int task, pds_config *cfg);

/* Allocate the SoA state for cfg->num_envs envs on cfg->device.  State is undefined until the first
 * pds_reset*.  *out receives the handle. */
int pds_create(const pds_config *cfg, pds_handle **out);
int pds_destroy(pds_handle *h);

/* Observation width D = 2 * (|o| + 4): 42/40/48 noise-free, 34/40/48 with observation_noise. */
int pds_obs_dim(const pds_handle *h);
int64_t pds_num_envs(const pds_handle *h);

/* Reset the envs with d_mask[i] != 0 (d_mask == NULL: all) from the in-kernel Philox stream keyed by
 * (seed, env_id_base + i, reset tick); writes the reset observation [o0,u0,o0,u0] into d_obs rows of
 * the reset envs (d_obs: [N, D]). */
int pds_reset(pds_handle *h, const uint8_t *d_mask, float *d_obs, void *stream);

/* Same, with the sampled values supplied per env: d_samples [N, PDS_SAMPLE_FLOATS]. */
int pds_reset_from_samples(pds_handle *h, const uint8_t *d_mask, const float *d_samples,
                           float *d_obs, void *stream);

/* One lockstep env.step() for all N envs.
 *   d_actions   [N,4] in      d_obs        [N,D] out (reset obs for envs auto-reset in this step)
 *   d_reward    [N]   out     d_terminated [N] u8 out      d_truncated [N] u8 out
 *   d_cost      [N]   out (info['cost'])
 *   d_final_obs [N,D] or NULL: rows of envs that finished in this step receive their last obs. */
int pds_step(pds_handle *h, const float *d_actions, float *d_obs, float *d_reward,
             uint8_t *d_terminated, uint8_t *d_truncated, float *d_cost, float *d_final_obs,
             void *stream);

/* pds_step with the noise variates supplied by the caller (parity injection for the stochastic
 * parts: OU thrust noise, SensorNoise): d_variates [N, aggregate_phy_steps * PDS_NOISE_FLOATS].  The reference draws them
 * from the global numpy stream (envs/utils.py:106, envs/sensors.py:84-134). */
int pds_step_with_variates(pds_handle *h, const float *d_actions, const float *d_variates, float *d_obs,
                           float *d_reward, uint8_t *d_terminated, uint8_t *d_truncated, float *d_cost,
                           float *d_final_obs, void *stream);

/* K lockstep env.step()s in ONE launch for open-loop action sequences (the replay of recorded actions
 * in simopt, simopt/pybullet.py:127-183: `for i in range(T-1): sim_env.step(acs[i])`): the env state
 * stays in registers between the K steps, only actions (in) and observations / rewards / flags (out)
 * stream through HBM.  Bitwise identical to K pds_step calls.  (Two launches on `stream` for the observation-noise
 * configurations: the kept noisy observation, which pds_step regenerates instead of storing, is materialised first.)
 *   d_actions [K,N,4]   d_obs [K,N,D]   d_reward, d_cost [K,N]   d_terminated, d_truncated [K,N] u8
 *   d_final_obs [K,N,D] or NULL */
int pds_step_k(pds_handle *h, int k_steps, const float *d_actions, float *d_obs, float *d_reward,
               uint8_t *d_terminated, uint8_t *d_truncated, float *d_cost, float *d_final_obs, void *stream);

/* CrazyFlieAgent.set_latency (envs/agents.py:388-404; called by simopt/pybullet.py:248): latency <
 * time_step disables the delay, otherwise buf_size = int(latency / time_step) and the action buffer and
 * its index are zeroed for every env.  Synchronises the device (not a hot path). */
int pds_set_latency(pds_handle *h, double latency);
int pds_latency_steps(const pds_handle *h); /* current buf_size, 0 when use_latency is off */

int pds_field_width(int field);
int pds_get_state(pds_handle *h, int field, void *d_out, void *stream);
/* Edits ONE field of every env.  What an env has observed is not edited with it: with observation noise the history half of
 * the next observation row stays the (noisy) observation the previous step returned -- the kernels regenerate it from the
 * state and the previous tick's draws, so pds_set_state (like a masked pds_reset and pds_set_tick) first writes it to memory,
 * one extra launch on `stream`; pds_set_state(PDS_F_NOISY_OBS) replaces it.  Without observation noise o(k) is rebuilt from
 * the state the step loads, i.e. it follows an edit of PDS_F_POS / PDS_F_RPY / PDS_F_VEL / PDS_F_OMEGA. */
int pds_set_state(pds_handle *h, int field, const void *d_in, void *stream);

/* Number of reset/step ticks issued so far (the Philox counter word). */
/* Diagnostic (synchronises the stream): number of envs whose position / attitude / velocity / body
 * rates hold a NaN or an Inf.  The reference has no such guard; its explicit Euler
 * step can overflow on envs that never terminate (TakeOff with domain randomisation). */
int pds_count_nonfinite(pds_handle *h, int64_t *count, void *stream);

/* The tick and the parity of the action ring live in DEVICE memory (one word per 64-env tile, advanced
 * by the kernels themselves), so no kernel argument changes between two pds_step calls with the same
 * pointers: a sequence of pds_step / pds_reset calls can be captured into a hipGraph and replayed.
 * pds_tick returns the host's mirror (exact unless a captured graph was replayed);
 * pds_sync_tick(h, stream) synchronises `stream`, re-reads the device word and returns it. */
uint64_t pds_tick(const pds_handle *h);
uint64_t pds_sync_tick(pds_handle *h, void *stream);
/* Restore the tick of a checkpoint: a handle created with the same config whose fields were all set
 * with pds_set_state and whose tick was set to the saved one continues the saved run bit for bit
 * (there is no other hidden state: the reference offers no checkpointing of its envs; its trainer
 * checkpoints are model.pt / state.pkl, utils/loggers.py:382-407). */
int pds_set_tick(pds_handle *h, uint64_t tick);

/* Algorithmic HBM bytes one pds_step moves per env for this configuration. */
int pds_bytes_per_env_step(const pds_handle *h);
/* the same for pds_step_k with k_steps per launch (state traffic amortised over the K steps) */
int pds_bytes_per_env_step_k(const pds_handle *h, int k_steps);

const char *pds_last_error(const pds_handle *h);

/* The generator of the RNG contract, exposed for verification: d_out[i] =
 * Philox4x32-<rounds>(counter d_ctr[i][0..3], key d_key[i][0..1]) for i < n, computed by the same device
 * function the step / reset kernels use (rounds 10: reset sampling, 7: per-step noise).  tests/ check it
 * against the Random123 known-answer vectors.  Runs on the current device. */
int pds_philox4x32(const uint32_t *d_ctr, const uint32_t *d_key, int rounds, int64_t n, uint32_t *d_out, void *stream);

/* The standard normals of the per-step noise, exposed for verification: d_out[i][0..7] = the four
 * one-word Box-Muller pairs of Philox4x32-7(counter (env_id_base + i, tick lo, tick hi, block), key = seed) for i < n --
 * the device functions the step kernels draw the OU / gyro / sensor normals with (envs/sensors.py:75-134, envs/base.py:457-468).
 * tests/ hold 2^26 of them against N(0, 1): Kolmogorov-Smirnov distance, moments, tail mass.  Runs on the current device. */
int pds_noise_normals(uint64_t seed, uint64_t tick, uint32_t block, uint64_t env_id_base, int64_t n, float *d_out, void *stream);

/* observation_history_size = H other than 2 (envs/base.py:44, 303-319, 417-431): advances the [N, H, half]
 * history of every env by the step's new row `d_obs2` [N, 2 * half] (the pds_step output) in one launch.
 * Running envs: hist' = [hist[1:], newest half].  Finished envs (auto_reset != 0): their final history
 * [hist[1:], newest half of d_final_obs2] goes to d_final_hist (rows of other envs untouched; may be NULL) and
 * hist' = [H - 1 copies of the reset row's first half, its second half].  d_hist_out must not alias d_hist_in. */
int pds_history_advance(int64_t n, int half, int history, const float *d_obs2, const uint8_t *d_terminated,
                        const uint8_t *d_truncated, const float *d_final_obs2, int auto_reset, const float *d_hist_in,
                        float *d_hist_out, float *d_final_hist, void *stream);

/* ---- caller-side helper: GAE over a lockstep rollout ----------------------
 * Replaces core.Buffer.finish_path / calculate_adv_and_value_targets (algs/core.py:461-533, one
 * scipy lfilter per finished path) for a [T, N] rollout: d_rew, d_val [T,N] f32; d_terminated,
 * d_truncated [T,N] u8 (the flags pds_step returned); d_final_val [T,N] = V(final_obs) read where
 * truncated -- the cut wins over a termination on the same step, algs/iwpg/iwpg.py:374-379 -- (may be NULL); d_last_val [N] = V(o_T).  rew_scale = 1/(ret_std + eps) with clipping to
 * +-rew_clip (use_reward_scaling) or 0 for raw rewards.  Outputs [T,N]: advantages, value targets,
 * discounted returns.  Runs on the current device, asynchronously on `stream`. */
int pds_gae(const float *d_rew, const float *d_val, const uint8_t *d_terminated, const uint8_t *d_truncated,
            const float *d_final_val, const float *d_last_val, float gamma, float lam, float rew_scale,
            float rew_clip, int64_t T, int64_t N, float *d_adv, float *d_target_v, float *d_disc_ret,
            void *stream);

/* ---- caller-side dense kernels on the f32 matrix cores ----------------------
 * A 3-layer MLP in torch's nn.Linear layout (weights [out][in] row-major, device pointers):
 * y = W3 act(W2 act(W1 x + b1) + b2) + b3; d_in <= 192 (more than 64 inputs -- observation_history_size >= 4,
 * envs/base.py:303-319 -- run the K-tiled kernels of csrc/pds_mlp_wide.hip), h1, h2 <= 64, d_out <= 8; activation 0 relu, 1 tanh.
 * Mirrors build_mlp_network / MLPGaussianActor.net / MLPCritic.net (algs/core.py:65-103, 228-311). */
typedef struct pds_mlp {
  int32_t d_in, h1, h2, d_out, activation;
  const float *w1, *b1, *w2, *b2, *w3, *b3;
} pds_mlp;

/* ONE launch per rollout (csrc/pds_rollout.h): the T closed-loop steps of the caller's roll_out
 * (algs/iwpg/iwpg.py:350-385; ActorCritic.step algs/core.py:370-393) --
 *   V(o) -> d_val_buf[t];  a = mu(o) + exp(log_std) z, log p -> d_act_buf[t], d_logp_buf[t] (the draws of
 *   pds_gaussian_sample with call = *d_call_base + call_offset + t + 1);  env.step(a) (bitwise pds_step) ->
 *   d_rew_buf[t], d_term_buf[t], d_trunc_buf[t], d_cost_buf[t], next observation -> d_obs_buf[t + 1];
 *   V(final observation) of the envs whose episode the TimeLimit cut at step t (truncated, terminated or not: the
 *   bootstrap value of algs/iwpg/iwpg.py:374-379; at t = T - 1 also of the envs that only terminated, for a caller that
 *   mirrors the reference's epoch-end cut) -> d_fval_buf[t] (other entries are left alone: pds_gae never reads them);  episode return / length bookkeeping of pds_rollout_record -> d_ep_ret, d_ep_len, d_stats[3]
 * -- and V(o(T)) -> d_last_val.  d_obs_buf is [T + 1, N, D]: row 0 holds o(0) on entry, rows 1..T are written.
 * Networks: actor d_in = D, d_out = 4; critic d_in = D, d_out = 1; hidden <= 64; inputs standardised with
 * d_mean / d_std / eps when given (OnlineMeanStd.forward).  All other buffers are [T, N] ([T, N, 4] actions).
 * PDS_EUNSUPPORTED for env configurations the kernel is not built for (the caller falls back to the per-step
 * entry points, which give the same bits). */
int pds_rollout(pds_handle *h, int T, const pds_mlp *pi, const pds_mlp *vf, const float *d_mean, const float *d_std, float eps,
                const float *d_log_std, uint64_t seed, const uint64_t *d_call_base, uint64_t call_offset, int deterministic,
                float *d_obs_buf, float *d_act_buf, float *d_logp_buf, float *d_val_buf, float *d_rew_buf,
                uint8_t *d_term_buf, uint8_t *d_trunc_buf, float *d_cost_buf, float *d_fval_buf, float *d_last_val,
                float *d_ep_ret, float *d_ep_len, float *d_stats, void *stream);

/* The same for observation histories other than 2 (observation_history_size = H, envs/base.py:44, 303-319, 417-431;
 * csrc/pds_rollout_hist.h): the actor reads the last `history` [o, u] halves of every env, history x half <= 192 inputs.
 * In the kernel: actor, sampling, env.step (bitwise pds_step), the history update of pds_history_advance, the episode
 * bookkeeping.  NOT in the kernel: the critic -- the caller evaluates V over d_obs_buf afterwards (pds_mlp_forward), and
 * over d_fin_rows: the final histories of the envs whose path bootstraps with V (the TimeLimit cut it, or it finished on
 * step T - 1: algs/iwpg/iwpg.py:374-379), one slot list per env -- d_fin_rows [slots, N, history x half], d_fin_step
 * [slots, N] = the step t whose d_fval_buf[t] entry the row's value is; the caller presets d_fin_step to -1 (unused);
 * slots >= T / max_episode_steps + 2.  d_obs_buf [T + 1, N, history x half]: row 0 = the histories on entry, rows 1..T
 * written.  Other buffers as pds_rollout.  PDS_EUNSUPPORTED where no kernel is built (the per-step path gives the same bits). */
int pds_rollout_history(pds_handle *h, int T, int history, const pds_mlp *pi, const float *d_mean, const float *d_std, float eps,
                        const float *d_log_std, uint64_t seed, const uint64_t *d_call_base, uint64_t call_offset,
                        int deterministic, float *d_obs_buf, float *d_act_buf, float *d_logp_buf, float *d_rew_buf,
                        uint8_t *d_term_buf, uint8_t *d_trunc_buf, float *d_cost_buf, float *d_fin_rows, int32_t *d_fin_step,
                        int slots, float *d_ep_ret, float *d_ep_len, float *d_stats, void *stream);

/* number of parameters; flat gradient layout = [W1, b1, W2, b2, W3, b3] (torch parameter order) */
int pds_mlp_param_count(const pds_mlp *m);
/* floats of scratch the *_grad entry points need (per-wave partial sums) */
int64_t pds_mlp_workspace_floats(const pds_mlp *m);

/* d_y[B, d_out] = MLP(x'), x' = row d_index[g] (or g when d_index is NULL) of d_x[rows, d_in],
 * standardised as (x - mean) / (std + eps) when d_mean / d_std are given (OnlineMeanStd.forward,
 * utils/online_mean_std.py:32-43).  Replaces ActorCritic.step's network calls (algs/core.py:370-393). */
int pds_mlp_forward(const pds_mlp *m, const float *d_x, const int64_t *d_index, int64_t B, const float *d_mean,
                    const float *d_std, float eps, float *d_y, void *stream);

/* Gradient of the PPO-clip policy loss  -mean(min(r A, clip(r, 1-c, 1+c) A)),  r = exp(logp - logp_old),
 * logp = Normal(MLP(x), exp(log_std)).log_prob(act).sum(-1)  (compute_loss_pi, algs/ppo/ppo.py:22-40)
 * with respect to the MLP parameters: d_grads[param_count]; d_stats[4] = {sum of -min(..), sum of r,
 * sum over samples and actions of 0.5 z^2 (approx_kl numerator), sample count}.  d_x is the
 * already standardised observation batch [B, d_in]. */
int pds_ppo_policy_grad(const pds_mlp *m, const float *d_x, const float *d_act, const float *d_adv,
                        const float *d_logp_old, const float *d_log_std, int64_t B, float clip_ratio, float *d_grads,
                        float *d_stats, float *d_workspace, void *stream);

/* Gradient of mse_loss(MLP(x[index]), target[index]) (compute_loss_v, algs/iwpg/iwpg.py:272-275) for a
 * critic (d_out == 1); d_index selects the mini-batch rows (NULL: rows 0..B-1); d_stats[0] = sum of
 * squared errors, d_stats[3] = sample count. */
int pds_value_grad(const pds_mlp *m, const float *d_x, const int64_t *d_index, const float *d_target, int64_t B,
                   float *d_grads, float *d_stats, float *d_workspace, void *stream);

/* a[n, d_out] = mu + exp(log_std) z with z ~ N(0, 1) (Philox4x32-10, key = seed, counter = (global
 * sample id = id_base + row, call)), logp[n] = Normal(mu, sigma).log_prob(a).sum(-1); deterministic != 0
 * gives a = mu (evaluation mode).  dist.sample() + log_prob of ActorCritic.step, algs/core.py:370-393. */
int pds_gaussian_sample(const float *d_mu, const float *d_log_std, int64_t n, int d_out, uint64_t seed, uint64_t call,
                        uint64_t id_base, int deterministic, float *d_act, float *d_logp, void *stream);

/* The same with the call counter split into a DEVICE word and a by-value offset: call = *d_call_base + call_offset.
 * A rollout captured into a hipGraph passes the step index as the offset and advances the device word once
 * per replay with pds_counter_add (*d_counter += inc, one thread), so every replay draws fresh variates. */
int pds_gaussian_sample_dev(const float *d_mu, const float *d_log_std, int64_t n, int d_out, uint64_t seed,
                            const uint64_t *d_call_base, uint64_t call_offset, uint64_t id_base, int deterministic,
                            float *d_act, float *d_logp, void *stream);
int pds_counter_add(uint64_t *d_counter, uint64_t inc, void *stream);

/* d_out[i] = p(i), i < n: a pseudo-random permutation of 0 .. n-1 keyed by (seed, call) -- a 6-round Feistel network
 * (round keys from Philox4x32-10) over the next even-width power of two, cycle-walked into [0, n).  One elementwise
 * launch; stands in for the index shuffle of the value net's mini-batches (np.random.shuffle in
 * IWPGAlgorithm.update_value_net, algs/iwpg/iwpg.py:487-522), whose stream the GPU trainer does not share anyway. */
int pds_permutation(int64_t *d_out, int64_t n, uint64_t seed, uint64_t call, void *stream);

/* One rollout step's bookkeeping (buf.store + episode statistics of IWPGAlgorithm.roll_out,
 * algs/iwpg/iwpg.py:350-385): copies reward / terminated / truncated [n] into their [T, N] slices, adds
 * the reward to the running episode return and 1 to the length, and for finished envs adds
 * (return, length, 1) to d_stats[0..2] and zeroes their running values. */
int pds_rollout_record(const float *d_rew, const uint8_t *d_term, const uint8_t *d_trunc, int64_t n, float *d_rew_buf,
                       uint8_t *d_term_buf, uint8_t *d_trunc_buf, float *d_ep_ret, float *d_ep_len, float *d_stats,
                       void *stream);

/* torch.optim.Adam.step (no weight decay / amsgrad) for the six tensors of `m` from their flat gradient;
 * d_exp_avg / d_exp_avg_sq [param_count] are the optimiser state, step counts from 1. */
int pds_adam_step(const pds_mlp *m, const float *d_grads, float *d_exp_avg, float *d_exp_avg_sq, int64_t step, float lr,
                  float beta1, float beta2, float eps, void *stream);

/* Gradient + optimiser step in the SAME two launches as the gradient alone: the partial-sum kernel of
 * pds_ppo_policy_grad / pds_value_grad applies torch.optim.Adam.step to each parameter right after it has summed its
 * gradient (same arithmetic as pds_adam_step: the two routes give the same bits; d_grads and d_stats are written as
 * before).  opt == NULL: no step (= the plain entry points).  For single-process training without gradient clipping --
 * with several ranks the gradient all-reduce sits between the two. */
typedef struct pds_adam {
  float *d_exp_avg, *d_exp_avg_sq; /* optimiser state [param_count] */
  int64_t step;                    /* counts from 1 */
  float lr, beta1, beta2, eps;
} pds_adam;
int pds_ppo_policy_grad_step(const pds_mlp *m, const float *d_x, const float *d_act, const float *d_adv,
                             const float *d_logp_old, const float *d_log_std, int64_t B, float clip_ratio, float *d_grads,
                             float *d_stats, float *d_workspace, const pds_adam *opt, void *stream);
int pds_value_grad_step(const pds_mlp *m, const float *d_x, const int64_t *d_index, const float *d_target, int64_t B,
                        float *d_grads, float *d_stats, float *d_workspace, const pds_adam *opt, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PDS_H */
