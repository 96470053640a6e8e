// pds_api.hip -- host side of libpds_hip.so: the C ABI of include/pds.h (+ the state pack/unpack
// kernel behind pds_get_state / pds_set_state).  The fused step / reset kernels live in pds_step.h /
// pds_reset.h and are instantiated per task in pds_task_{hover,circle,takeoff}.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

#include "pds_reset.h"  // (pds_types.h + regen_kept_obs for pds_get_state)

namespace pds {

// ---- state access (parity injection, checkpointing) -------------------------------------------
struct FieldArgs {
  DevState st;
  Consts k;
  void *user;
  long long n;
  int field;
  int task;
  int set;
  int has_motor, has_dr, has_tn, has_on, ctrl;
  int regen_obs;  // observation noise without the Kalman hold: the kept observation is regenerated unless kCtrOhBit (pds_types.h)
  unsigned long long env_id_base;
  uint32_t seed_lo, seed_hi;
};

__global__ __launch_bounds__(kBlock) void field_kernel(const FieldArgs a) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.n) return;
  float *uf = reinterpret_cast<float *>(a.user);
  int32_t *ui = reinterpret_cast<int32_t *>(a.user);
  float4 q0 = a.st.s0[i], q1 = a.st.s1[i], q2 = a.st.s2[i];
  const uint32_t c = a.st.ctr[i];
  const int parity = (int)(a.st.clk[0].z & 1u);  // every tile's clock holds the same value between launches
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
        const Quat q = quat_from_euler(q1.z, q1.w, q2.x);
        const float sgn = ctr_sign(c) ? -1.f : 1.f;
        uf[4 * i] = sgn * q.x; uf[4 * i + 1] = sgn * q.y; uf[4 * i + 2] = sgn * q.z; uf[4 * i + 3] = sgn * q.w;
      }
      break;
    case PDS_F_MOTOR_X:
    case PDS_F_LAST_ACTION:
    case PDS_F_PREV_ACTION:
    case PDS_F_MOTOR_A:
    case PDS_F_MOTOR_K:
    case PDS_F_OU: {
      float4 *arr = nullptr;
      float4 dflt = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.field == PDS_F_MOTOR_X) arr = a.has_motor ? a.st.mx : nullptr;
      else if (a.field == PDS_F_LAST_ACTION) arr = a.st.hist[parity];
      else if (a.field == PDS_F_PREV_ACTION) arr = a.st.hist[parity ^ 1];
      else if (a.field == PDS_F_OU) arr = a.has_tn ? a.st.ou : nullptr;
      else if (a.field == PDS_F_MOTOR_A) { arr = (a.has_motor && a.has_dr) ? a.st.mA : nullptr; dflt = make_float4(a.k.A, a.k.A, a.k.A, a.k.A); }
      else { arr = (a.has_motor && a.has_dr) ? a.st.mK : nullptr; dflt = make_float4(a.k.K, a.k.K, a.k.K, a.k.K); }
      if (a.set) { if (arr) arr[i] = make_float4(uf[4 * i], uf[4 * i + 1], uf[4 * i + 2], uf[4 * i + 3]); }
      else { const float4 v = arr ? arr[i] : dflt; uf[4 * i] = v.x; uf[4 * i + 1] = v.y; uf[4 * i + 2] = v.z; uf[4 * i + 3] = v.w; }
      break;
    }
    case PDS_F_STEP_COUNT:
      if (a.set) {
        uint32_t off = ctr_off(c);
        const uint32_t s_new = (uint32_t)ui[i] & 0xFFFFu;
        if (a.task == PDS_TASK_CIRCLE)  // keep ref_offset: the stored phase is (steps + ref_offset) mod num_ref_points
          off = (circle_ref_offset(c, a.k.ref_points) + s_new % (uint32_t)a.k.ref_points) % (uint32_t)a.k.ref_points;
        a.st.ctr[i] = ctr_pack(s_new, ctr_sign(c), off, ctr_lat(c)) | (c & kCtrOhBit);
      }
      else ui[i] = (int32_t)ctr_step(c);
      break;
    case PDS_F_QUAT_SIGN:
      if (a.set) a.st.ctr[i] = ctr_pack(ctr_step(c), ui[i] ? 1u : 0u, ctr_off(c), ctr_lat(c)) | (c & kCtrOhBit);
      else ui[i] = (int32_t)ctr_sign(c);
      break;
    case PDS_F_REF_OFFSET:
      if (a.task == PDS_TASK_CIRCLE) {
        const uint32_t P = (uint32_t)a.k.ref_points;
        if (a.set) a.st.ctr[i] = ctr_pack(ctr_step(c), ctr_sign(c), ((uint32_t)ui[i] % P + ctr_step(c) % P) % P, ctr_lat(c)) | (c & kCtrOhBit);
        else ui[i] = (int32_t)circle_ref_offset(c, a.k.ref_points);
      } else if (!a.set) {
        ui[i] = 0;
      }
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
    case PDS_F_GYRO_BIAS:
    case PDS_F_GYRO_LPF:
      if (a.has_on) {
        float4 n0 = a.st.nz0[i]; float2 n1 = a.st.nz1[i];
        if (a.field == PDS_F_GYRO_BIAS) {
          if (a.set) { n0.x = uf[3 * i]; n0.y = uf[3 * i + 1]; n0.z = uf[3 * i + 2]; a.st.nz0[i] = n0; }
          else { uf[3 * i] = n0.x; uf[3 * i + 1] = n0.y; uf[3 * i + 2] = n0.z; }
        } else {
          if (a.set) { n0.w = uf[3 * i]; n1.x = uf[3 * i + 1]; n1.y = uf[3 * i + 2]; a.st.nz0[i] = n0; a.st.nz1[i] = n1; }
          else { uf[3 * i] = n0.w; uf[3 * i + 1] = n1.x; uf[3 * i + 2] = n1.y; }
        }
      } else if (!a.set) { uf[3 * i] = 0.f; uf[3 * i + 1] = 0.f; uf[3 * i + 2] = 0.f; }
      break;
    case PDS_F_NOISY_OBS:
      if (a.has_on) {
        if (a.set) {
          a.st.oh0[i] = make_float4(uf[10 * i], uf[10 * i + 1], uf[10 * i + 2], uf[10 * i + 3]);
          a.st.oh1[i] = make_float4(uf[10 * i + 4], uf[10 * i + 5], uf[10 * i + 6], uf[10 * i + 7]);
          a.st.oh2[i] = make_float2(uf[10 * i + 8], uf[10 * i + 9]);
          if (a.regen_obs) a.st.ctr[i] = c | kCtrOhBit;  // from now on it is what the next step reads
        } else if (a.regen_obs && !ctr_oh(c)) {
          // not in memory: what the next step will regenerate (csrc/pds_reset.h regen_kept_obs), from the same inputs
          const WaveClock ck = a.st.clk[i / kWave];
          const RngKey now{a.seed_lo, a.seed_hi, ck.x, ck.y};
          const EnvRegs e{q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
          NoisyObs o;
          regen_kept_obs(a.k, (uint32_t)(a.env_id_base + (unsigned long long)i), now, ctr_step(c) == 0u, e, o);
          uf[10 * i] = o.x; uf[10 * i + 1] = o.y; uf[10 * i + 2] = o.z; uf[10 * i + 3] = o.qx;
          uf[10 * i + 4] = o.qy; uf[10 * i + 5] = o.qz; uf[10 * i + 6] = o.qw; uf[10 * i + 7] = o.vx;
          uf[10 * i + 8] = o.vy; uf[10 * i + 9] = o.vz;
        } else {
          const float4 o0 = a.st.oh0[i], o1 = a.st.oh1[i]; const float2 o2 = a.st.oh2[i];
          uf[10 * i] = o0.x; uf[10 * i + 1] = o0.y; uf[10 * i + 2] = o0.z; uf[10 * i + 3] = o0.w;
          uf[10 * i + 4] = o1.x; uf[10 * i + 5] = o1.y; uf[10 * i + 6] = o1.z; uf[10 * i + 7] = o1.w;
          uf[10 * i + 8] = o2.x; uf[10 * i + 9] = o2.y;
        }
      } else if (!a.set) {
        for (int j = 0; j < 10; ++j) uf[10 * i + j] = 0.f;
      }
      break;
    case PDS_F_PID: {
      float v[12];
      for (int j = 0; j < 12; ++j) v[j] = 0.f;
      if (a.set) {
        for (int j = 0; j < 12; ++j) v[j] = uf[12 * i + j];
        if (a.ctrl >= 1) { a.st.pid0[i] = make_float4(v[0], v[1], v[2], v[3]); a.st.pid1[i] = make_float2(v[4], v[5]); }
        if (a.ctrl == 2) { a.st.pid2[i] = make_float4(v[6], v[7], v[8], v[9]); a.st.pid3[i] = make_float2(v[10], v[11]); }
      } else {
        if (a.ctrl >= 1) { const float4 p0 = a.st.pid0[i]; const float2 p1 = a.st.pid1[i]; v[0] = p0.x; v[1] = p0.y; v[2] = p0.z; v[3] = p0.w; v[4] = p1.x; v[5] = p1.y; }
        if (a.ctrl == 2) { const float4 p2 = a.st.pid2[i]; const float2 p3 = a.st.pid3[i]; v[6] = p2.x; v[7] = p2.y; v[8] = p2.z; v[9] = p2.w; v[10] = p3.x; v[11] = p3.y; }
        for (int j = 0; j < 12; ++j) uf[12 * i + j] = v[j];
      }
      break;
    }
    case PDS_F_ACTION_BUFFER:  // drone.action_buffer, rows >= buf_size read as 0 / are ignored
      for (int b = 0; b < kMaxLatSteps; ++b) {
        const bool live = a.st.lat != nullptr && b < a.k.lat_steps;
        if (a.set) {
          if (live) a.st.lat[(long long)b * a.n + i] = make_float4(uf[32 * i + 4 * b], uf[32 * i + 4 * b + 1], uf[32 * i + 4 * b + 2], uf[32 * i + 4 * b + 3]);
        } else {
          const float4 v = live ? a.st.lat[(long long)b * a.n + i] : make_float4(0.f, 0.f, 0.f, 0.f);
          uf[32 * i + 4 * b] = v.x; uf[32 * i + 4 * b + 1] = v.y; uf[32 * i + 4 * b + 2] = v.z; uf[32 * i + 4 * b + 3] = v.w;
        }
      }
      break;
    case PDS_F_ACTION_IDX:
      if (a.set) a.st.ctr[i] = ctr_pack(ctr_step(c), ctr_sign(c), ctr_off(c), a.k.lat_steps > 0 ? (uint32_t)ui[i] % (uint32_t)a.k.lat_steps : 0u) | (c & kCtrOhBit);
      else ui[i] = (int32_t)ctr_lat(c);
      break;
    default: break;
  }
}

// pds_set_tick / pds_create: every tile's clock word <- (tick, parity kept or 0)
__global__ __launch_bounds__(256) void clock_fill_kernel(WaveClock *clk, long long ntiles, unsigned long long tick, int keep_parity) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntiles) return;
  const uint32_t par = keep_parity ? (clk[t].z & 1u) : 0u;
  clk[t] = make_uint4((uint32_t)tick, (uint32_t)(tick >> 32), par, 0u);
}

// pds_set_latency: action buffer and index of every env <- 0 (envs/agents.py:403-404)
__global__ __launch_bounds__(256) void latency_clear_kernel(DevState st, long long n, int rows) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int b = 0; b < rows; ++b) st.lat[(long long)b * n + i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const uint32_t c = st.ctr[i];
  st.ctr[i] = ctr_pack(ctr_step(c), ctr_sign(c), ctr_off(c), 0u) | (c & kCtrOhBit);
}

// The kept noisy observation of every env that does not have it in oh0-2 (i.e. every env after a pds_step, none after a
// pds_step_k / pds_rollout / reset) is regenerated there and flagged in the counter word -- what the single-step kernels do in
// their prologue (init_kept_obs), as a pass of its own: in front of the K-step and rollout kernels (so that their loops are not
// compiled around it, csrc/pds_step.h step_k_kernel) and in front of every call that changes the regeneration's inputs without
// stepping the env (materialize_kept_obs below).  Same function, same inputs, same bits.  4 B per env when every env is
// flagged, 52 + 44 B otherwise.
__global__ __launch_bounds__(kBlock) void materialize_oh_kernel(DevState st, Consts k, long long n, unsigned long long env_id_base,
                                                                uint32_t seed_lo, uint32_t seed_hi) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint32_t c = st.ctr[i];
  if (ctr_oh(c) != 0u) return;
  const WaveClock ck = st.clk[i / kWave];
  const RngKey now{seed_lo, seed_hi, ck.x, ck.y};
  const float4 q0 = st.s0[i], q1 = st.s1[i], q2 = st.s2[i];
  const EnvRegs e{q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
  NoisyObs o;
  regen_kept_obs(k, (uint32_t)(env_id_base + (unsigned long long)i), now, ctr_step(c) == 0u, e, o);
  st.oh0[i] = make_float4(o.x, o.y, o.z, o.qx);
  st.oh1[i] = make_float4(o.qy, o.qz, o.qw, o.vx);
  st.oh2[i] = make_float2(o.vy, o.vz);
  st.ctr[i] = c | kCtrOhBit;
}


}  // namespace pds

// =================================================================================================
// the C ABI of include/pds.h
// =================================================================================================
using namespace pds;

struct pds_handle {
  pds_config cfg;
  DevState st;
  Consts k;
  LaunchFlags flags;
  int force_tile;  // PDS_FORCE_TILE=half|full (tests / A-B runs): 1 half, 2 full, 0 pick per launch
  float2 *d_circle_ref;
  void *slab;  // one allocation holds every state array (staggered, see pds_create)
  float4 *lat_buf;  // [PDS_MAX_LATENCY_STEPS][N] delayed-action ring, allocated when latency is first enabled
  unsigned long long *d_count;  // pds_count_nonfinite result word (in the slab)
  unsigned long long *d_stamps;  // diagnostic builds: pds_debug_stamps
  int obs_dim;
  int num_cus;    // hipDeviceProp.multiProcessorCount (256 on MI355X): drives the half-tile rule
  uint64_t tick;  // host mirror of the device clock words (pds_sync_tick refreshes it)
  bool was_reset;
  int stored_from_agg;  // (set in pds_create) aggregate_phy_steps from which pds_step keeps the noisy observation in memory
  bool split_reset;   // pds_step launches the SplitReset form + post_reset_kernel where the variant has one (PDS_SPLIT_RESET != 0)
  bool stored_ready;  // kLaunchStepStored handles: the one materialize_oh_kernel pass in front of their first step is done
  char err[512];
};

// Every entry point runs on the handle's device and leaves the caller's current device as it found it.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      switched = err == hipSuccess;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
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

// The kept noisy observation of every env into oh0-2 (flagged in the counter word) where the variant regenerates it from the
// previous tick's Philox blocks and the env's state: in front of everything that moves a tile's clock, an env's state or its step
// counter WITHOUT stepping the env -- a masked reset (the reset kernel advances the clock of every tile, also for the envs outside
// the mask), pds_set_tick, the pds_set_state edits -- and in front of the K-step / rollout kernels, which read it from memory.
// The history half of the next observation is then the observation that was returned, as in the reference (envs/base.py:303-319).
static int materialize_kept_obs(pds_handle *h, hipStream_t s) {
  if (!(PDS_REGEN_OBS && h->flags.on && !h->flags.hold)) return PDS_OK;
  const long long n = h->cfg.num_envs;
  hipLaunchKernelGGL(pds::materialize_oh_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, h->st, h->k, n,
                     (unsigned long long)h->cfg.env_id_base, (uint32_t)h->cfg.seed, (uint32_t)(h->cfg.seed >> 32));
  PDS_HIP(h, hipGetLastError());
  return PDS_OK;
}

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
  c->control_mode = PDS_CTRL_PWM;
  c->use_latency = 0;          // CrazyFlieSimpleAgent: use_latency=False (envs/agents.py:492)
  c->latency = 0.015;          // envs/base.py:40
  c->observation_frequency = 100;  // envs/base.py:42
  return PDS_OK;
}

/* Python's `a // b` for floats (Objects/floatobject.c float_divmod): NOT floor(a / b) -- 0.03 // 0.01 == 2.0
 * while 0.03 / 0.01 == 3.0 -- because the remainder is taken exactly (fmod) first. */
static double py_float_floordiv(double vx, double wx) {
  double mod = fmod(vx, wx);
  double div = (vx - mod) / wx;
  if (mod != 0 && ((wx < 0) != (mod < 0))) div -= 1.0;
  if (div == 0) return 0.0;
  double fl = floor(div);
  if (div - fl > 0.5) fl += 1.0;
  return fl;
}

// buf_size of the delayed-action ring: ctor form max(1, int(LATENCY // time_step)) (envs/agents.py:180),
// set_latency form int(latency / TIME_STEP) (envs/agents.py:401); 0 = no delay
static int latency_steps_ctor(double latency, double time_step) {
  if (!(latency >= time_step)) return 0;  // use_latency if latency >= time_step else False, agents.py:165
  const int b = (int)py_float_floordiv(latency, time_step);
  return b < 1 ? 1 : b;
}

static void set_latency_consts(Consts &k, int lat_steps, int agg) {
  k.lat_steps = lat_steps;
  k.lat_own1 = k.lat_own2 = 0;
  if (lat_steps > 0) {
    // the row action_buffer[-1] is rewritten by every lat_steps-th apply_action; after env.step s
    // (agg sub-steps each) it holds the action of step ceil(m / agg), m = floor(s agg / B) B (0: reset row)
    for (int s = 1; s <= 2; ++s) {
      const int m = (s * agg / lat_steps) * lat_steps;
      (s == 1 ? k.lat_own1 : k.lat_own2) = (m + agg - 1) / agg;
    }
  }
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
  // OUNoise(sigma = 0.2 * motor_thrust_noise), envs/agents.py:206
  k.ou_sigma = (float)(0.2 * (c.motor_thrust_noise > 0 ? c.motor_thrust_noise : 0.0));
  // SensorNoise defaults, envs/sensors.py:14-23; gyro model envs/sensors.py:124-128 with
  // dt = 1/sim_freq (envs/hover.py:143)
  const double D2R = M_PI / 180.0, dt = c.time_step;
  k.pos_std = 0.002f; k.pos_unif = 0.001f; k.vel_std = 0.01f;
  k.q_std = (float)(0.1 * D2R); k.q_unif = (float)(0.05 * D2R);
  const double gnd = 0.000175, corr = 1000.0;
  const double sigma_g_d = gnd / sqrt(dt);
  k.gyro_sb = (float)sqrt(-(sigma_g_d * sigma_g_d) * (corr / 2) * (exp(-2 * dt / corr) - 1));
  k.gyro_pi = (float)exp(-dt / corr);
  k.gyro_rw = 0.0105f;
  k.gyro_to = (float)(5 * D2R);
  k.agg = c.aggregate_phy_steps; k.max_steps = c.max_episode_steps;
  k.reset_dist = c.enable_reset_distribution ? 1 : 0;
  k.ref_points = (c.task == PDS_TASK_CIRCLE) ? 3 * c.observation_frequency : kRefPoints;  // envs/circle.py:47-49
  {
    const double dth = 2.0 * M_PI / (double)k.ref_points;
    k.ref_dth_hi = (float)dth;
    k.ref_dth_lo = (float)(dth - (double)k.ref_dth_hi);
  }
  k.obs_rate = 1;
  if (c.observation_noise > 0) {  // obs_rate = sim_freq // observation_frequency, envs/base.py:108
    const int r = (int)llround(1.0 / c.time_step) / c.observation_frequency;
    k.obs_rate = r < 1 ? 1 : r;
  }
  set_latency_consts(k, c.use_latency ? latency_steps_ctor(c.latency, c.time_step) : 0, c.aggregate_phy_steps);
}

static int obs_dim_of(const pds_config &c) {
  const bool noisy = c.observation_noise > 0;
  const int o = noisy ? (c.task == PDS_TASK_HOVER ? 13 : (c.task == PDS_TASK_CIRCLE ? 16 : 20))
                      : (c.task == PDS_TASK_HOVER ? 17 : (c.task == PDS_TASK_CIRCLE ? 16 : 20));
  return 2 * (o + 4);
}

// DroneBaseEnv.__init__ evaluates compute_observation() once to size the observation space
// (envs/base.py:142): with sensor noise that advances the gyro bias random walk by one draw
// (envs/sensors.py:130-131) before the first reset, and the bias is never reset afterwards.
constexpr uint32_t kBlkCtor = 16;  // Philox block of tick 0 (oracle: PO_BLK_CTOR)
__global__ __launch_bounds__(256) void ctor_noise_kernel(DevState st, long long n, unsigned long long id_base, float gyro_sb,
                                                         uint32_t seed_lo, uint32_t seed_hi) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const U4 r = philox4x32_7((uint32_t)(id_base + (unsigned long long)i), 0u, 0u, kBlkCtor, seed_lo, seed_hi);
  float z0, z1, z2, z3;
  box_muller_word(r.x, z0, z1);
  box_muller_word(r.y, z2, z3);
  st.nz0[i] = make_float4(gyro_sb * z0, gyro_sb * z1, gyro_sb * z2, 0.f);
}

#define PDS_CREATE_FAIL(code, ...)                                   \
  do {                                                               \
    snprintf(g_create_err, sizeof(g_create_err), __VA_ARGS__);       \
    return code;                                                     \
  } while (0)

static int ensure_latency_buffer(pds_handle *h) {
  if (h->lat_buf) return PDS_OK;
  const size_t bytes = (size_t)kMaxLatSteps * (size_t)h->cfg.num_envs * sizeof(float4);
  hipError_t e = hipMalloc((void **)&h->lat_buf, bytes);
  if (e == hipSuccess) e = hipMemset(h->lat_buf, 0, bytes);
  if (e != hipSuccess) {
    if (h->lat_buf) (void)hipFree(h->lat_buf);
    h->lat_buf = nullptr;
    return fail(h, e == hipErrorOutOfMemory ? PDS_ENOMEM : PDS_EHIP, "latency buffer (%zu bytes): %s", bytes, hipGetErrorString(e));
  }
  h->st.lat = h->lat_buf;
  return PDS_OK;
}

extern "C" int pds_create(const pds_config *cfg, pds_handle **out) {
  if (!cfg || !out) return PDS_EINVAL;
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(pds_config)) PDS_CREATE_FAIL(PDS_EINVAL, "pds_config size mismatch (built against another pds.h?)");
  if (cfg->task < 0 || cfg->task > 2 || cfg->num_envs < 1 || cfg->aggregate_phy_steps < 1 ||
      cfg->max_episode_steps < 1 || cfg->max_episode_steps > 65535 || cfg->time_step <= 0)
    PDS_CREATE_FAIL(PDS_EINVAL, "invalid pds_config (task, num_envs, aggregate_phy_steps, max_episode_steps or time_step)");
  if (cfg->num_envs > (1ll << 30)) PDS_CREATE_FAIL(PDS_EINVAL, "num_envs %lld exceeds 2^30 per handle", (long long)cfg->num_envs);
  // the Philox counter carries a 32-bit global env id
  if (cfg->env_id_base < 0 || cfg->env_id_base + cfg->num_envs > (1ll << 32))
    PDS_CREATE_FAIL(PDS_EINVAL, "env_id_base %lld + num_envs %lld leaves the 32-bit global env id range [0, 2^32)",
                    (long long)cfg->env_id_base, (long long)cfg->num_envs);
  if (cfg->device < 0) PDS_CREATE_FAIL(PDS_EINVAL, "device %d is negative", cfg->device);
  if (cfg->control_mode < 0 || cfg->control_mode > 2) PDS_CREATE_FAIL(PDS_EINVAL, "control_mode %d", cfg->control_mode);
  if (cfg->control_mode != PDS_CTRL_PWM && cfg->task == PDS_TASK_TAKEOFF)
    PDS_CREATE_FAIL(PDS_EUNSUPPORTED, "control_mode %d: the PID modes exist for Hover/Circle (TakeOff fixes control_mode PWM, envs/takeoff.py:225)", cfg->control_mode);
  if (cfg->control_mode != PDS_CTRL_PWM && cfg->use_ground_effect) {
    // round 5: PID + ground effect is built; not together with the latency ring or the Kalman hold
    const int sf = (int)llround(1.0 / cfg->time_step);
    const bool hold = cfg->observation_noise > 0 && cfg->observation_frequency > 0 && sf / cfg->observation_frequency != 1;
    if (cfg->use_latency || hold)
      PDS_CREATE_FAIL(PDS_EUNSUPPORTED, "control_mode %d with the ground-effect extension: not with use_latency or observation_frequency < sim_freq", cfg->control_mode);
  }
  if (cfg->observation_frequency < 1) PDS_CREATE_FAIL(PDS_EINVAL, "observation_frequency %d", cfg->observation_frequency);
  if (cfg->task == PDS_TASK_CIRCLE && 3 * cfg->observation_frequency > PDS_MAX_REF_POINTS)
    PDS_CREATE_FAIL(PDS_EUNSUPPORTED, "Circle with observation_frequency %d needs %d reference points (limit %d)",
                    cfg->observation_frequency, 3 * cfg->observation_frequency, PDS_MAX_REF_POINTS);
  if (cfg->observation_noise > 0) {
    // obs_rate = sim_freq // observation_frequency (envs/base.py:108)
    const int sim_freq = (int)llround(1.0 / cfg->time_step);
    const int obs_rate = sim_freq / cfg->observation_frequency;
    if (obs_rate < 1) PDS_CREATE_FAIL(PDS_EINVAL, "observation_frequency %d above sim_freq %d: obs_rate 0 (the reference divides by it)", cfg->observation_frequency, sim_freq);
  }
  // (observation noise with obs_rate > 1 -- the Kalman-hold branch -- together with a PID control mode or the latency
  //  ring: built since round 4)
  const int lat_steps = cfg->use_latency ? latency_steps_ctor(cfg->latency, cfg->time_step) : 0;
  if (lat_steps > kMaxLatSteps) PDS_CREATE_FAIL(PDS_EUNSUPPORTED, "latency %g s = %d steps (limit %d)", cfg->latency, lat_steps, kMaxLatSteps);
  // (use_latency + the ground-effect extension: built for control_mode PWM since round 3; the PID modes have no
  //  ground-effect instantiation at all, refused above)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device >= ndev)
    PDS_CREATE_FAIL(PDS_ENODEVICE, "no HIP device %d (found %d); there is no CPU fallback", cfg->device, ndev);
  pds_handle *h = new (std::nothrow) pds_handle();
  if (!h) return PDS_ENOMEM;
  memset((void *)h, 0, sizeof(*h));
  h->cfg = *cfg;
  fill_consts(*cfg, h->k);
  h->obs_dim = obs_dim_of(*cfg);
  h->flags.motor = cfg->use_motor_dynamics != 0;
  h->flags.dr = cfg->domain_randomization > 0;
  h->flags.ge = cfg->use_ground_effect != 0;
  h->flags.tn = cfg->motor_thrust_noise > 0;
  h->flags.on = cfg->observation_noise > 0;
  h->flags.ctrl = cfg->control_mode;
  h->flags.lat = lat_steps > 0;
  h->flags.hold = h->k.obs_rate != 1;
  h->flags.half_tile = false;
  if (const char *ft = getenv("PDS_FORCE_TILE")) h->force_tile = (ft[0] == 'h') ? 1 : ((ft[0] == 'f') ? 2 : 0);
  // Round 6 built the reset OUT of the single-step kernel (SplitReset<V> + post_reset_kernel, csrc/pds_types.h) and measured it
  // SLOWER than the in-place reset on every configuration (profiles/r06_ab_split_reset.txt: config 6 85 -> 108-117 us), so it is
  // off unless PDS_SPLIT_RESET=1 asks for it (same bits either way: tests/test_gpu_properties.py)
  h->split_reset = false;
  if (const char *sr = getenv("PDS_SPLIT_RESET")) h->split_reset = sr[0] == '1';
  h->stored_from_agg = PDS_STORED_OH_FROM_AGG;  // (the memset above wiped the member initialisers)
  if (PDS_STORED_OH_FROM_AGG > 0)
    if (const char *sa = getenv("PDS_STORED_OH_FROM_AGG")) h->stored_from_agg = atoi(sa);  // (A/B builds: 0 = always regenerate)
  const size_t n = (size_t)cfg->num_envs;
  const size_t ntiles = (n + kWave - 1) / kWave;
  const LaunchFlags &f = h->flags;
  DeviceGuard guard(cfg->device);
  hipError_t e = guard.err;
  if (e == hipSuccess) {
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, cfg->device);
    h->num_cus = (e == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  // All state arrays live in ONE slab (one allocation, one free, contiguous pages); consecutive arrays
  // are staggered by an odd multiple of 256 B so that the 6-21 concurrent streams of the step kernel do
  // not start at the same offset modulo a power of two (measured neutral on MI355X: 62.4-63.7 us for
  // staggers of 0 B ... 1 MiB, so this is hygiene, not a lever).
  std::vector<std::pair<void **, size_t>> req;
  auto alloc = [&](void **p, size_t bytes) { req.emplace_back(p, bytes); };
  alloc((void **)&h->st.s0, n * 16); alloc((void **)&h->st.s1, n * 16); alloc((void **)&h->st.s2, n * 16);
  alloc((void **)&h->st.hist[0], n * 16); alloc((void **)&h->st.hist[1], n * 16);
  alloc((void **)&h->st.ctr, n * 4);
  alloc((void **)&h->st.clk, ntiles * sizeof(WaveClock));
  alloc((void **)&h->d_count, 256);
  if (f.motor) alloc((void **)&h->st.mx, n * 16);
  if (f.dr) { alloc((void **)&h->st.par0, n * 16); alloc((void **)&h->st.par1, n * 8); }
  if (f.dr && f.motor) { alloc((void **)&h->st.mA, n * 16); alloc((void **)&h->st.mK, n * 16); }
  if (f.tn) alloc((void **)&h->st.ou, n * 16);
  if (f.on) {
    alloc((void **)&h->st.nz0, n * 16); alloc((void **)&h->st.nz1, n * 8);
    alloc((void **)&h->st.oh0, n * 16); alloc((void **)&h->st.oh1, n * 16); alloc((void **)&h->st.oh2, n * 8);
  }
  if (f.ctrl >= 1) { alloc((void **)&h->st.pid0, n * 16); alloc((void **)&h->st.pid1, n * 8); }
  if (f.ctrl == 2) { alloc((void **)&h->st.pid2, n * 16); alloc((void **)&h->st.pid3, n * 8); }
  alloc((void **)&h->d_circle_ref, kRefPoints * sizeof(float2));
  {
    const size_t stagger = (size_t)kStaggerBytes;
    size_t total = 0;
    std::vector<size_t> off;
    for (size_t j = 0; j < req.size(); ++j) {
      total = (total + 255) / 256 * 256 + stagger;
      off.push_back(total);
      total += req[j].second;
    }
    if (e == hipSuccess) e = hipMalloc(&h->slab, total + 256);
    if (e == hipSuccess) e = hipMemset(h->slab, 0, total + 256);  // tick 0, parity 0, OU / PID / stale-rate state 0
    if (e == hipSuccess)
      for (size_t j = 0; j < req.size(); ++j) *req[j].first = (char *)h->slab + off[j];
  }
  if (e == hipSuccess) {
    float2 ref[kRefPoints];  // envs/circle.py:45-56
    const int np = h->k.ref_points;
    for (int t = 0; t < kRefPoints; ++t) {
      const double ts = 2 * M_PI * (double)(t % np) / np;
      ref[t].x = (float)(0.25 * (1 - cos(ts)));
      ref[t].y = (float)(0.25 * sin(ts));
    }
    e = hipMemcpy(h->d_circle_ref, ref, sizeof(ref), hipMemcpyHostToDevice);
  }
  h->st.circle_ref = h->d_circle_ref;
  if (e == hipSuccess && h->flags.on) {
    hipLaunchKernelGGL(ctor_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, h->st, (long long)n,
                       (unsigned long long)cfg->env_id_base, h->k.gyro_sb, (uint32_t)cfg->seed, (uint32_t)(cfg->seed >> 32));
    e = hipGetLastError();
  }
  // The zero fill and the constructor draw ran on the null stream; the caller's stream may be a
  // non-blocking one that is not ordered behind it, so the state is made visible here, always.
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    snprintf(g_create_err, sizeof(g_create_err), "allocation of %zu envs failed: %s", n, hipGetErrorString(e));
    pds_destroy(h);
    return e == hipErrorOutOfMemory ? PDS_ENOMEM : PDS_EHIP;
  }
  if (h->flags.lat) {
    const int rc = ensure_latency_buffer(h);
    if (rc != PDS_OK) {
      snprintf(g_create_err, sizeof(g_create_err), "%s", h->err);
      pds_destroy(h);
      return rc;
    }
  }
  *out = h;
  return PDS_OK;
}

extern "C" int pds_destroy(pds_handle *h) {
  if (!h) return PDS_OK;
  DeviceGuard guard(h->cfg.device);
  if (h->slab) (void)hipFree(h->slab);
  if (h->lat_buf) (void)hipFree(h->lat_buf);
  if (h->d_stamps) (void)hipFree(h->d_stamps);
  delete h;
  return PDS_OK;
}

extern "C" int pds_obs_dim(const pds_handle *h) { return h ? h->obs_dim : PDS_EINVAL; }
extern "C" int64_t pds_num_envs(const pds_handle *h) { return h ? h->cfg.num_envs : PDS_EINVAL; }
extern "C" uint64_t pds_tick(const pds_handle *h) { return h ? h->tick : 0; }
extern "C" int pds_latency_steps(const pds_handle *h) { return h ? h->k.lat_steps : PDS_EINVAL; }
extern "C" const char *pds_last_error(const pds_handle *h) { return h ? h->err : g_create_err; }

extern "C" uint64_t pds_sync_tick(pds_handle *h, void *stream) {
  if (!h) return 0;
  DeviceGuard guard(h->cfg.device);
  WaveClock c;
  if (hipStreamSynchronize((hipStream_t)stream) == hipSuccess &&
      hipMemcpy(&c, h->st.clk, sizeof(c), hipMemcpyDeviceToHost) == hipSuccess)
    h->tick = ((uint64_t)c.y << 32) | c.x;
  return h->tick;
}

extern "C" int pds_set_tick(pds_handle *h, uint64_t tick) {
  if (!h) return PDS_EINVAL;
  DeviceGuard guard(h->cfg.device);
  PDS_HIP(h, guard.err);
  PDS_HIP(h, hipDeviceSynchronize());  // no stream argument: order behind everything in flight
  if (const int rc = materialize_kept_obs(h, 0)) return rc;  // (regenerated from the OLD tick's blocks)
  const long long ntiles = (h->cfg.num_envs + kWave - 1) / kWave;
  hipLaunchKernelGGL(clock_fill_kernel, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, 0, h->st.clk, ntiles,
                     (unsigned long long)tick, 1);
  PDS_HIP(h, hipGetLastError());
  PDS_HIP(h, hipDeviceSynchronize());
  h->tick = tick;
  return PDS_OK;
}

// CrazyFlieAgent.set_latency, envs/agents.py:388-404
extern "C" int pds_set_latency(pds_handle *h, double latency) {
  if (!h) return PDS_EINVAL;
  if (h->flags.ge && h->flags.ctrl != 0) return fail(h, PDS_EUNSUPPORTED, "use_latency with the ground-effect extension: control_mode PWM only");
  int steps = 0;
  if (!(latency < h->cfg.time_step)) {
    steps = (int)(latency / h->cfg.time_step);  // int(self.latency / self.TIME_STEP)
    if (steps < 1) return fail(h, PDS_EINVAL, "latency %g: buf_size 0 (the reference asserts buf_size > 0)", latency);
    if (steps > kMaxLatSteps) return fail(h, PDS_EUNSUPPORTED, "latency %g s = %d steps (limit %d)", latency, steps, kMaxLatSteps);
  }
  // every check and the allocation come BEFORE the handle is touched: a refused call leaves it as it was
  DeviceGuard guard(h->cfg.device);
  PDS_HIP(h, guard.err);
  PDS_HIP(h, hipDeviceSynchronize());
  if (steps > 0) {
    const int rc = ensure_latency_buffer(h);
    if (rc != PDS_OK) return rc;
  }
  h->cfg.latency = latency;
  h->cfg.use_latency = steps > 0;
  h->flags.lat = steps > 0;
  set_latency_consts(h->k, steps, h->cfg.aggregate_phy_steps);
  if (steps > 0) {
    const long long n = h->cfg.num_envs;
    hipLaunchKernelGGL(latency_clear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, h->st, n, kMaxLatSteps);
    PDS_HIP(h, hipGetLastError());
    PDS_HIP(h, hipDeviceSynchronize());
  }
  return PDS_OK;
}

// read action 16 + dyn state 48 + action history 32 + counter 4; write dyn state 48
// + newest history slot 16 + counter 4 + reward 4 + cost 4 + terminated 1 + truncated 1; obs 4*D;
// DR params +24; motor PT1: x R+W 32 (+ A, K 32 when randomised); OU state R+W 32; gyro bias +
// low-pass R+W 48; kept noisy observation (10 floats) R+W 80 only with the Kalman hold (obs_rate > 1) -- otherwise it is
// regenerated from the previous tick's Philox blocks, not kept (csrc/pds_reset.h regen_kept_obs; pds_step_with_variates
// and the first step after an explicit reset still move those 80 B); latency ring: one row R+W per physics
// sub-step.  (The per-tile clock word adds 0.5 B per env-step and is not counted.)
extern "C" int pds_bytes_per_env_step(const pds_handle *h) {
  if (!h) return PDS_EINVAL;
  int b = 100 + 78 + 4 * h->obs_dim;
  const LaunchFlags &f = h->flags;
  if (f.dr) b += 24;
  if (f.motor) b += 32;
  if (f.motor && f.dr) b += 32;
  if (f.tn) b += 32;
  if (f.on) b += 48 + ((f.hold || !PDS_REGEN_OBS || (h->stored_from_agg > 0 && h->cfg.aggregate_phy_steps >= h->stored_from_agg)) ? 80 : 0);
  if (f.ctrl >= 1) b += 48;  // rate-PID integral + last error, R+W
  if (f.ctrl == 2) b += 48;  // attitude-PID integral + last error, R+W
  if (f.lat) b += 32 * h->cfg.aggregate_phy_steps;
  return b;
}

// Algorithmic bytes per env-step of pds_step_k: the env state is read and written once per launch.
extern "C" int pds_bytes_per_env_step_k(const pds_handle *h, int k_steps) {
  if (!h || k_steps < 1) return PDS_EINVAL;
  const LaunchFlags &f = h->flags;
  const int full = pds_bytes_per_env_step(h);
  // the PID control modes have no K-step kernel: pds_step_k loops over pds_step for them (below), every step moves the
  // single-step kernel's bytes (rounds 2-5 priced those rows with the K-step formula: 0.26-0.36 "of their own roofline"
  // for what is pds_step at 0.57-0.86)
  if (f.ctrl != 0) return full;
  const int stream = 16 + 4 * h->obs_dim + 10 + (f.lat ? 32 * h->cfg.aggregate_phy_steps : 0);
  // once per launch: the state read + written, both ring slots and the randomised parameters written back
  // (+ the kept noisy observation, which the K-step kernel reads from and leaves in oh0-2: materialize_oh_kernel)
  const int state = full - stream + 16 + (f.dr ? 24 : 0) + (f.dr && f.motor ? 32 : 0) + ((PDS_REGEN_OBS && f.on && !f.hold) ? 80 : 0);
  return stream + (state + k_steps - 1) / k_steps;
}

static void base_args(pds_handle *h, StepArgs &a) {
  memset(&a, 0, sizeof(a));
  a.st = h->st;
  a.k = h->k;
  a.n = h->cfg.num_envs;
  a.env_id_base = (unsigned long long)h->cfg.env_id_base;
  a.seed_lo = (uint32_t)h->cfg.seed; a.seed_hi = (uint32_t)(h->cfg.seed >> 32);
  a.auto_reset = h->cfg.auto_reset;
  a.k_steps = 1;
  a.stamps = h->d_stamps;
}

static void launch_family(pds_handle *h, int kind, const LaunchFlags &lf, dim3 grid, hipStream_t s, const StepArgs &a) {
  const int task = h->cfg.task;
  const bool reset_kind = kind == kLaunchReset || kind == kLaunchPostReset;
  if (lf.hold && !reset_kind) {  // (a reset observes at iteration 0: always a fresh observation)
    if (task == PDS_TASK_HOVER) launch_hover_hold(kind, lf, grid, s, a);
    else if (task == PDS_TASK_CIRCLE) launch_circle_hold(kind, lf, grid, s, a);
    else launch_takeoff_hold(kind, lf, grid, s, a);
  } else if (lf.lat) {
    if (task == PDS_TASK_HOVER) launch_hover_lat(kind, lf, grid, s, a);
    else if (task == PDS_TASK_CIRCLE) launch_circle_lat(kind, lf, grid, s, a);
    else launch_takeoff_lat(kind, lf, grid, s, a);
  } else if (lf.ctrl != 0 && !reset_kind) {
    if (lf.ge) {
      if (task == PDS_TASK_HOVER) launch_hover_pid_ge(kind, lf, grid, s, a);
      else launch_circle_pid_ge(kind, lf, grid, s, a);
    } else if (task == PDS_TASK_HOVER) launch_hover_pid(kind, lf, grid, s, a);
    else launch_circle_pid(kind, lf, grid, s, a);
  } else {
    if (task == PDS_TASK_HOVER) launch_hover(kind, lf, grid, s, a);
    else if (task == PDS_TASK_CIRCLE) launch_circle(kind, lf, grid, s, a);
    else launch_takeoff(kind, lf, grid, s, a);
  }
}

static int do_reset(pds_handle *h, const uint8_t *d_mask, const float *d_samples, float *d_obs, void *stream) {
  if (!h) return PDS_EINVAL;
  DeviceGuard guard(h->cfg.device);
  PDS_HIP(h, guard.err);
  StepArgs a;
  base_args(h, a);
  a.mask = d_mask; a.samples = d_samples; a.obs = d_obs;
  const dim3 grid((unsigned)((a.n + kBlock - 1) / kBlock));
  if (d_mask != nullptr)  // (the envs outside the mask keep their episode: see materialize_kept_obs)
    if (const int rc = materialize_kept_obs(h, (hipStream_t)stream)) return rc;
  launch_family(h, kLaunchReset, h->flags, grid, (hipStream_t)stream, a);
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

static int check_step_pointers(pds_handle *h, const float *d_actions, float *d_obs, float *d_reward, uint8_t *d_terminated,
                               uint8_t *d_truncated, float *d_cost) {
  if (!d_actions || !d_obs || !d_reward || !d_terminated || !d_truncated || !d_cost)
    return fail(h, PDS_EINVAL, "pds_step: NULL tensor pointer");
  if (((uintptr_t)d_actions) & 15u) return fail(h, PDS_EINVAL, "pds_step: d_actions must be 16-byte aligned");
  if (((uintptr_t)d_obs) & 3u) return fail(h, PDS_EINVAL, "pds_step: d_obs must be 4-byte aligned (16 for full speed)");
  if (!h->was_reset) return fail(h, PDS_EINVAL, "pds_step before pds_reset");
  return PDS_OK;
}

extern "C" int pds_step_with_variates(pds_handle *h, const float *d_actions, const float *d_variates,
                                      float *d_obs, float *d_reward, uint8_t *d_terminated,
                                      uint8_t *d_truncated, float *d_cost, float *d_final_obs, void *stream) {
  if (!h) return PDS_EINVAL;
  if (const int rc = check_step_pointers(h, d_actions, d_obs, d_reward, d_terminated, d_truncated, d_cost)) return rc;
  DeviceGuard guard(h->cfg.device);  // (no hipSetDevice when the caller is already on the device)
  PDS_HIP(h, guard.err);
  StepArgs a;
  base_args(h, a);
  a.actions = reinterpret_cast<const float4 *>(d_actions);
  a.obs = d_obs; a.reward = d_reward; a.term = d_terminated; a.trunc = d_truncated; a.cost = d_cost;
  a.final_obs = d_final_obs;
  a.noise = d_variates;
  // one 256-env block per 4 tiles
  const dim3 grid((unsigned)((a.n + kBlock - 1) / kBlock));
  // half observation tile (4 resident blocks per CU instead of 3) while the grid is between one and
  // about 2.7 rounds of the full-tile residency; see pds_types.h kHalfTileRows
  LaunchFlags lf = h->flags;
  // (variants that reset in registers on the full tile -- pds_step.h merged_reset_variant -- gain less
  // from the half tile: it wins up to 5 blocks per CU there, up to 8 for the others; profiles/r01_tile_rows.txt)
  const bool merged = !lf.on && !lf.lat && !(lf.motor && lf.dr) && h->cfg.task != PDS_TASK_TAKEOFF && h->cfg.auto_reset;
  const unsigned cus = (unsigned)h->num_cus;
  lf.half_tile = grid.x > (unsigned)kFullTileBlocksPerCU * cus * (256 / kBlock) && grid.x <= (merged ? 5u : 8u) * cus * (256 / kBlock);
  if (h->force_tile) lf.half_tile = h->force_tile == 1;
  if (lf.on || lf.lat) lf.half_tile = false;  // (no half-tile instantiation)
  // observation noise with two or more physics sub-steps: the StoredOh form of the same kernel (kLaunchStepStored); such a
  // handle never runs the regenerating form, so every env's kept observation is in memory from the first call on
  int kind = kLaunchStep;
  if (PDS_REGEN_OBS && h->stored_from_agg > 0 && lf.on && !lf.hold && h->cfg.aggregate_phy_steps >= h->stored_from_agg &&
      d_variates == nullptr) {
    kind = kLaunchStepStored;
    if (!h->stored_ready) {
      if (const int rc = materialize_kept_obs(h, (hipStream_t)stream)) return rc;
      h->stored_ready = true;
    }
  }
  // Round 6 (opt-in, PDS_SPLIT_RESET=1: measured slower, see pds_create): where the single-step kernel resets finished envs IN
  // PLACE (observation noise / latency ring: no merged form), it is launched in its SplitReset form instead and post_reset_kernel
  // behind it resets them densely (csrc/pds_types.h SplitReset: same draws, same bits).  Not with injected variates (parity
  // replays) -- their reset rows carry the kCtrOhBit bookkeeping of the in-place path.
  const bool split = h->split_reset && h->cfg.auto_reset && d_variates == nullptr && kind == kLaunchStep &&
                     split_reset_supported(h->cfg.task, lf);
  launch_family(h, split ? kLaunchStepSplit : kind, lf, grid, (hipStream_t)stream, a);
  if (split) {
    a.k_steps = lf.hold ? 2 : ((PDS_REGEN_OBS && lf.on) ? 0 : 1);  // reset_store's oh_mode
    const dim3 pgrid((unsigned)((a.n + kPostResetEnvsPerBlock - 1) / kPostResetEnvsPerBlock));
    launch_family(h, kLaunchPostReset, lf, pgrid, (hipStream_t)stream, a);
  }
  PDS_HIP(h, hipGetLastError());
  h->tick += 1;
  return PDS_OK;
}

extern "C" int pds_step(pds_handle *h, const float *d_actions, float *d_obs, float *d_reward,
                        uint8_t *d_terminated, uint8_t *d_truncated, float *d_cost, float *d_final_obs,
                        void *stream) {
  return pds_step_with_variates(h, d_actions, nullptr, d_obs, d_reward, d_terminated, d_truncated, d_cost,
                                d_final_obs, stream);
}

extern "C" int pds_step_k(pds_handle *h, int k_steps, const float *d_actions, float *d_obs, float *d_reward,
                          uint8_t *d_terminated, uint8_t *d_truncated, float *d_cost, float *d_final_obs, void *stream) {
  if (!h) return PDS_EINVAL;
  if (k_steps < 1) return fail(h, PDS_EINVAL, "pds_step_k: k_steps %d", k_steps);
  if (const int rc = check_step_pointers(h, d_actions, d_obs, d_reward, d_terminated, d_truncated, d_cost)) return rc;
  const long long n = h->cfg.num_envs;
  const int D = h->obs_dim;
  if (h->flags.ctrl != 0) {
    // PID control modes (no K-step kernel): K single-step launches, same results
    for (int s = 0; s < k_steps; ++s) {
      const long long o1 = (long long)s * n;
      const int rc = pds_step(h, d_actions + o1 * 4, d_obs + o1 * D, d_reward + o1, d_terminated + o1, d_truncated + o1,
                              d_cost + o1, d_final_obs ? d_final_obs + o1 * D : nullptr, stream);
      if (rc != PDS_OK) return rc;
    }
    return PDS_OK;
  }
  DeviceGuard guard(h->cfg.device);
  PDS_HIP(h, guard.err);
  StepArgs a;
  base_args(h, a);
  a.actions = reinterpret_cast<const float4 *>(d_actions);
  a.obs = d_obs; a.reward = d_reward; a.term = d_terminated; a.trunc = d_truncated; a.cost = d_cost;
  a.final_obs = d_final_obs;
  a.k_steps = k_steps;
  const dim3 grid((unsigned)((a.n + kBlock - 1) / kBlock));
  LaunchFlags lf = h->flags;
  lf.half_tile = false;
  if (const int rc = materialize_kept_obs(h, (hipStream_t)stream)) return rc;  // (the K-step kernel reads oh0-2: StoredOh)
  launch_family(h, kLaunchStepK, lf, grid, (hipStream_t)stream, a);
  PDS_HIP(h, hipGetLastError());
  h->tick += (uint64_t)k_steps;
  return PDS_OK;
}

extern "C" int pds_field_width(int field) {
  switch (field) {
    case PDS_F_POS: case PDS_F_RPY: case PDS_F_VEL: case PDS_F_OMEGA: case PDS_F_GYRO_BIAS: case PDS_F_GYRO_LPF: return 3;
    case PDS_F_QUAT: case PDS_F_MOTOR_X: case PDS_F_LAST_ACTION: case PDS_F_PREV_ACTION: case PDS_F_MOTOR_A:
    case PDS_F_MOTOR_K: case PDS_F_OU: return 4;
    case PDS_F_STEP_COUNT: case PDS_F_QUAT_SIGN: case PDS_F_REF_OFFSET: case PDS_F_ACTION_IDX: return 1;
    case PDS_F_PARAMS: return 6;
    case PDS_F_NOISY_OBS: return 10;
    case PDS_F_PID: return 12;
    case PDS_F_ACTION_BUFFER: return 4 * kMaxLatSteps;
    default: return PDS_EINVAL;
  }
}

// number of envs whose dynamic state holds a NaN or an Inf (diagnostic, see pds_count_nonfinite)
// One launch per rollout: csrc/pds_rollout.h (the caller's roll_out, algs/iwpg/iwpg.py:350-385).
extern "C" int pds_rollout(pds_handle *h, int T, const pds_mlp *pi, const pds_mlp *vf, const float *d_mean, const float *d_std,
                           float eps, const float *d_log_std, uint64_t seed, const uint64_t *d_call_base, uint64_t call_offset,
                           int deterministic, float *d_obs_buf, float *d_act_buf, float *d_logp_buf, float *d_val_buf,
                           float *d_rew_buf, uint8_t *d_term_buf, uint8_t *d_trunc_buf, float *d_cost_buf, float *d_fval_buf,
                           float *d_last_val, float *d_ep_ret, float *d_ep_len, float *d_stats, void *stream) {
  if (!h) return PDS_EINVAL;
  if (T < 1) return fail(h, PDS_EINVAL, "pds_rollout: T %d", T);
  if (!pi || !vf || !d_log_std || !d_obs_buf || !d_act_buf || !d_logp_buf || !d_val_buf || !d_rew_buf || !d_term_buf ||
      !d_trunc_buf || !d_cost_buf || !d_fval_buf || !d_last_val || !d_ep_ret || !d_ep_len || !d_stats)
    return fail(h, PDS_EINVAL, "pds_rollout: NULL pointer");
  if ((d_mean == nullptr) != (d_std == nullptr)) return fail(h, PDS_EINVAL, "pds_rollout: mean and std come together");
  if (!h->was_reset) return fail(h, PDS_EINVAL, "pds_rollout before pds_reset");
  // the rollout bootstraps finished episodes from V(final_obs) and restarts them in place (roll_out of the reference
  // calls env.reset() itself, algs/iwpg/iwpg.py:382-385): without auto-reset there is no final observation to evaluate
  if (!h->cfg.auto_reset) return fail(h, PDS_EUNSUPPORTED, "pds_rollout needs a handle created with auto_reset = 1");
  const int D = h->obs_dim;
  for (const pds_mlp *m : {pi, vf})
    if (m->d_in != D || m->h1 < 1 || m->h1 > 64 || m->h2 < 1 || m->h2 > 64 || (m->activation != 0 && m->activation != 1) ||
        !m->w1 || !m->b1 || !m->w2 || !m->b2 || !m->w3 || !m->b3)
      return fail(h, PDS_EINVAL, "pds_rollout: network shape (d_in must be the observation width %d, hidden <= 64)", D);
  if (pi->d_out != 4 || vf->d_out != 1) return fail(h, PDS_EINVAL, "pds_rollout: actor d_out 4, critic d_out 1");
  if ((((uintptr_t)d_act_buf) & 15u) || (((uintptr_t)d_obs_buf) & 3u)) return fail(h, PDS_EINVAL, "pds_rollout: alignment");
  DeviceGuard guard(h->cfg.device);
  PDS_HIP(h, guard.err);
  RolloutArgs ra;
  memset(&ra, 0, sizeof(ra));
  base_args(h, ra.s);
  const long long n = h->cfg.num_envs;
  ra.s.actions = reinterpret_cast<const float4 *>(d_act_buf);  // (load_env's action slot: valid memory, value unused)
  ra.s.obs = d_obs_buf + n * D;                                 // step t writes o(t + 1) into row t + 1
  ra.s.reward = d_rew_buf; ra.s.term = d_term_buf; ra.s.trunc = d_trunc_buf; ra.s.cost = d_cost_buf;
  ra.s.final_obs = nullptr;                                     // (the finished rows stay in LDS)
  ra.s.k_steps = T;
  ra.pi = *pi; ra.vf = *vf;
  ra.mean = d_mean; ra.stdv = d_std; ra.eps = eps; ra.log_std = d_log_std;
  ra.seed = seed; ra.call_base = reinterpret_cast<const unsigned long long *>(d_call_base); ra.call_offset = call_offset;
  ra.deterministic = deterministic; ra.T = T;
  ra.obs0 = d_obs_buf;
  ra.act_buf = d_act_buf; ra.logp_buf = d_logp_buf; ra.val_buf = d_val_buf; ra.fval_buf = d_fval_buf; ra.last_val = d_last_val;
  ra.ep_ret = d_ep_ret; ra.ep_len = d_ep_len; ra.stats = d_stats;
  const long long tiles = (n + kWave - 1) / kWave;
  const dim3 grid((unsigned)tiles);  // (the number of tiles: the launchers pick one or two tiles per block, csrc/pds_rollout.h)
  // support is decided BEFORE the handle is touched (a refused call leaves it as it was)
  if (!rollout_supported(h->cfg.task, h->flags))
    return fail(h, PDS_EUNSUPPORTED, "pds_rollout: no kernel for this env configuration (not built: the ground effect except on TakeOff with control_mode PWM; the Kalman hold or "
                                     "partial noise settings together with a PID mode or the latency ring; TakeOff with motor dynamics "
                                     "without the latency ring) -- the per-step kernels give the same bits");
  // the env waves read the kept noisy observation from oh0-2 (StoredOh, like the K-step kernel)
  if (const int rc = materialize_kept_obs(h, (hipStream_t)stream)) return rc;
  bool ok;
  if (h->cfg.task == PDS_TASK_HOVER) ok = launch_rollout_hover(h->flags, grid, (hipStream_t)stream, ra);
  else if (h->cfg.task == PDS_TASK_CIRCLE) ok = launch_rollout_circle(h->flags, grid, (hipStream_t)stream, ra);
  else ok = launch_rollout_takeoff(h->flags, grid, (hipStream_t)stream, ra);
  if (!ok) return fail(h, PDS_EHIP, "pds_rollout: rollout_supported() and the launchers disagree");
  PDS_HIP(h, hipGetLastError());
  h->tick += (uint64_t)T;
  return PDS_OK;
}

// One launch per rollout for observation histories other than 2: csrc/pds_rollout_hist.h.
extern "C" int pds_rollout_history(pds_handle *h, int T, int history, const pds_mlp *pi, const float *d_mean, const float *d_std,
                                   float eps, const float *d_log_std, uint64_t seed, const uint64_t *d_call_base,
                                   uint64_t call_offset, int deterministic, float *d_obs_buf, float *d_act_buf,
                                   float *d_logp_buf, float *d_rew_buf, uint8_t *d_term_buf, uint8_t *d_trunc_buf,
                                   float *d_cost_buf, float *d_fin_rows, int32_t *d_fin_step, int slots, float *d_ep_ret,
                                   float *d_ep_len, float *d_stats, void *stream) {
  if (!h) return PDS_EINVAL;
  if (T < 1 || history < 1 || slots < 1) return fail(h, PDS_EINVAL, "pds_rollout_history: T %d, history %d, slots %d", T, history, slots);
  if (!pi || !d_log_std || !d_obs_buf || !d_act_buf || !d_logp_buf || !d_rew_buf || !d_term_buf || !d_trunc_buf || !d_cost_buf ||
      !d_fin_rows || !d_fin_step || !d_ep_ret || !d_ep_len || !d_stats)
    return fail(h, PDS_EINVAL, "pds_rollout_history: NULL pointer");
  if ((d_mean == nullptr) != (d_std == nullptr)) return fail(h, PDS_EINVAL, "pds_rollout_history: mean and std come together");
  if (!h->was_reset) return fail(h, PDS_EINVAL, "pds_rollout_history before pds_reset");
  if (!h->cfg.auto_reset) return fail(h, PDS_EUNSUPPORTED, "pds_rollout_history needs a handle created with auto_reset = 1");
  const int half = h->obs_dim / 2, HS = history * half;
  if (HS > 192) return fail(h, PDS_EUNSUPPORTED, "pds_rollout_history: %d x %d = %d network inputs (<= 192)", history, half, HS);
  if (pi->d_in != HS || pi->h1 < 1 || pi->h1 > 64 || pi->h2 < 1 || pi->h2 > 64 || (pi->activation != 0 && pi->activation != 1) ||
      !pi->w1 || !pi->b1 || !pi->w2 || !pi->b2 || !pi->w3 || !pi->b3 || pi->d_out != 4)
    return fail(h, PDS_EINVAL, "pds_rollout_history: actor shape (d_in must be history x half = %d, hidden <= 64, d_out 4)", HS);
  // max_episode_steps bounds how often the TimeLimit can cut one env within T steps; + the rollout's last step
  if (slots < T / h->cfg.max_episode_steps + 2)
    return fail(h, PDS_EINVAL, "pds_rollout_history: slots %d < T / max_episode_steps + 2 = %d", slots, T / h->cfg.max_episode_steps + 2);
  if ((((uintptr_t)d_act_buf) & 15u) || (((uintptr_t)d_obs_buf) & 3u)) return fail(h, PDS_EINVAL, "pds_rollout_history: alignment");
  // support is decided BEFORE the handle is touched (a refused call leaves it as it was)
  if (!rollout_hist_supported(h->cfg.task, h->flags))
    return fail(h, PDS_EUNSUPPORTED, "pds_rollout_history: no kernel for this env configuration (built: control_mode PWM without latency ring, "
                                     "Kalman hold or ground effect; noise off or the reference's default; TakeOff without motor dynamics) "
                                     "-- the per-step kernels give the same bits");
  DeviceGuard guard(h->cfg.device);
  PDS_HIP(h, guard.err);
  RolloutHistArgs ra;
  memset(&ra, 0, sizeof(ra));
  base_args(h, ra.s);
  ra.s.actions = reinterpret_cast<const float4 *>(d_act_buf);  // (load_env's action slot: valid memory, value unused)
  ra.s.obs = nullptr;                                           // (the kernel's own [o(k), o(k + 1)] row stays in LDS)
  ra.s.reward = d_rew_buf; ra.s.term = d_term_buf; ra.s.trunc = d_trunc_buf; ra.s.cost = d_cost_buf;
  ra.s.final_obs = nullptr;
  ra.s.k_steps = T;
  ra.pi = *pi;
  ra.mean = d_mean; ra.stdv = d_std; ra.eps = eps; ra.log_std = d_log_std;
  ra.seed = seed; ra.call_base = reinterpret_cast<const unsigned long long *>(d_call_base); ra.call_offset = call_offset;
  ra.deterministic = deterministic; ra.T = T; ra.H = history; ra.half = half; ra.slots = slots;
  ra.obs_buf = d_obs_buf; ra.act_buf = d_act_buf; ra.logp_buf = d_logp_buf;
  ra.fin_rows = d_fin_rows; ra.fin_step = d_fin_step;
  ra.ep_ret = d_ep_ret; ra.ep_len = d_ep_len; ra.stats = d_stats;
  const long long n = h->cfg.num_envs;
  const dim3 grid((unsigned)((n + kWave - 1) / kWave));
  if (const int rc = materialize_kept_obs(h, (hipStream_t)stream)) return rc;  // (StoredOh, like pds_rollout)
  const int hn = rollout_hist_tiles(HS);
  bool ok;
  if (h->cfg.task == PDS_TASK_HOVER) ok = launch_rollout_hist_hover(h->flags, hn, grid, (hipStream_t)stream, ra);
  else if (h->cfg.task == PDS_TASK_CIRCLE) ok = launch_rollout_hist_circle(h->flags, hn, grid, (hipStream_t)stream, ra);
  else ok = launch_rollout_hist_takeoff(h->flags, hn, grid, (hipStream_t)stream, ra);
  if (!ok) return fail(h, PDS_EHIP, "pds_rollout_history: rollout_hist_supported() and the launchers disagree");
  PDS_HIP(h, hipGetLastError());
  h->tick += (uint64_t)T;
  return PDS_OK;
}

__global__ __launch_bounds__(256) void nonfinite_kernel(DevState st, long long n, unsigned long long *count) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool bad = false;
  if (i < n) {
    const float4 a = st.s0[i], b = st.s1[i], c = st.s2[i];
    const float s = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w));
    bad = !(fabsf(s) <= 3.4e38f);  // NaN or Inf anywhere poisons the sum (Inf - Inf = NaN included)
  }
  const unsigned long long m = __ballot(bad);
  if ((threadIdx.x & 63) == 0 && m != 0ull) atomicAdd(count, (unsigned long long)__popcll(m));
}

static int do_field(pds_handle *h, int field, void *d_ptr, int set, void *stream) {
  if (!h) return PDS_EINVAL;
  if (!d_ptr || pds_field_width(field) < 0) return fail(h, PDS_EINVAL, "bad field %d or NULL pointer", field);
  if (set && field == PDS_F_QUAT) return fail(h, PDS_EINVAL, "PDS_F_QUAT is derived (set PDS_F_RPY / PDS_F_QUAT_SIGN)");
  DeviceGuard guard(h->cfg.device);
  PDS_HIP(h, guard.err);
  FieldArgs a;
  memset(&a, 0, sizeof(a));
  a.st = h->st; a.k = h->k; a.user = d_ptr; a.n = h->cfg.num_envs; a.field = field; a.task = h->cfg.task; a.set = set;
  a.has_motor = h->flags.motor; a.has_dr = h->flags.dr; a.has_tn = h->flags.tn; a.has_on = h->flags.on; a.ctrl = h->flags.ctrl;
  a.regen_obs = PDS_REGEN_OBS && h->flags.on && !h->flags.hold;
  // an edit of the state / step counter does not change what the env has OBSERVED: the kept observation is regenerated from the
  // unedited state first and stays in oh0-2 (the reference's history keeps the row that was returned, envs/base.py:303-319)
  if (set && field != PDS_F_NOISY_OBS)
    if (const int rc = materialize_kept_obs(h, (hipStream_t)stream)) return rc;
  a.env_id_base = (unsigned long long)h->cfg.env_id_base;
  a.seed_lo = (uint32_t)h->cfg.seed; a.seed_hi = (uint32_t)(h->cfg.seed >> 32);
  const dim3 grid((unsigned)((a.n + kBlock - 1) / kBlock));
  hipLaunchKernelGGL(field_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, a);
  PDS_HIP(h, hipGetLastError());
  if (set) h->was_reset = true;
  return PDS_OK;
}

extern "C" int pds_count_nonfinite(pds_handle *h, int64_t *count, void *stream) {
  if (!h || !count) return PDS_EINVAL;
  DeviceGuard guard(h->cfg.device);
  PDS_HIP(h, guard.err);
  hipStream_t s = (hipStream_t)stream;
  PDS_HIP(h, hipMemsetAsync(h->d_count, 0, sizeof(*h->d_count), s));  // handle-owned word: nothing to free
  const long long n = h->cfg.num_envs;
  hipLaunchKernelGGL(nonfinite_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h->st, n, h->d_count);
  PDS_HIP(h, hipGetLastError());
  unsigned long long host = 0;
  PDS_HIP(h, hipMemcpyAsync(&host, h->d_count, sizeof(host), hipMemcpyDeviceToHost, s));
  PDS_HIP(h, hipStreamSynchronize(s));
  *count = (int64_t)host;
  return PDS_OK;
}

extern "C" int pds_get_state(pds_handle *h, int field, void *d_out, void *stream) { return do_field(h, field, d_out, 0, stream); }
extern "C" int pds_set_state(pds_handle *h, int field, const void *d_in, void *stream) { return do_field(h, field, const_cast<void *>(d_in), 1, stream); }

template <int ROUNDS>
__global__ __launch_bounds__(256) void philox_kernel(const uint32_t *ctr, const uint32_t *key, long long n, uint32_t *out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const U4 r = philox4x32<ROUNDS>(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], key[2 * i], key[2 * i + 1]);
  out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

__global__ __launch_bounds__(256) void noise_normals_kernel(uint32_t seed_lo, uint32_t seed_hi, uint32_t tick_lo, uint32_t tick_hi,
                                                            uint32_t block, unsigned long long id_base, long long n, float *out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const U4 r = philox4x32_7((uint32_t)(id_base + (unsigned long long)i), tick_lo, tick_hi, block, seed_lo, seed_hi);
  float z[8];
  box_muller_word(r.x, z[0], z[1]);
  box_muller_word(r.y, z[2], z[3]);
  box_muller_word(r.z, z[4], z[5]);
  box_muller_word(r.w, z[6], z[7]);
  float4 *o = reinterpret_cast<float4 *>(out + 8 * i);
  o[0] = make_float4(z[0], z[1], z[2], z[3]);
  o[1] = make_float4(z[4], z[5], z[6], z[7]);
}

extern "C" int pds_noise_normals(uint64_t seed, uint64_t tick, uint32_t block, uint64_t env_id_base, int64_t n, float *d_out,
                                 void *stream) {
  if (!d_out || n < 0 || (((uintptr_t)d_out) & 15u)) return PDS_EINVAL;
  if (n == 0) return PDS_OK;
  hipLaunchKernelGGL(noise_normals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint32_t)seed,
                     (uint32_t)(seed >> 32), (uint32_t)tick, (uint32_t)(tick >> 32), block, (unsigned long long)env_id_base,
                     (long long)n, d_out);
  return hipGetLastError() == hipSuccess ? PDS_OK : PDS_EHIP;
}

extern "C" int pds_philox4x32(const uint32_t *d_ctr, const uint32_t *d_key, int rounds, int64_t n, uint32_t *d_out, void *stream) {
  if (!d_ctr || !d_key || !d_out || n < 0 || (rounds != 7 && rounds != 10)) return PDS_EINVAL;
  if (n == 0) return PDS_OK;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (rounds == 10) hipLaunchKernelGGL(philox_kernel<10>, grid, dim3(256), 0, (hipStream_t)stream, d_ctr, d_key, (long long)n, d_out);
  else hipLaunchKernelGGL(philox_kernel<7>, grid, dim3(256), 0, (hipStream_t)stream, d_ctr, d_key, (long long)n, d_out);
  return hipGetLastError() == hipSuccess ? PDS_OK : PDS_EHIP;
}

// Diagnostic builds (-DPDS_STAMPS): per-wave s_memtime stamps of the step kernel's phases, kStampSlots
// words per 64-env tile; NULL / 0 in regular builds.  Not part of include/pds.h.
extern "C" int pds_debug_stamps(pds_handle *h, unsigned long long *host_out, long long max_words) {
#ifdef PDS_STAMPS
  if (!h) return PDS_EINVAL;
  DeviceGuard guard(h->cfg.device);
  const long long ntiles = (h->cfg.num_envs + kWave - 1) / kWave;
  const long long words = ntiles * kStampSlots;
  if (!h->d_stamps) {
    PDS_HIP(h, hipMalloc((void **)&h->d_stamps, (size_t)words * 8));
    PDS_HIP(h, hipMemset(h->d_stamps, 0, (size_t)words * 8));
    return 0;
  }
  PDS_HIP(h, hipDeviceSynchronize());
  const long long w = words < max_words ? words : max_words;
  if (host_out && w > 0) PDS_HIP(h, hipMemcpy(host_out, h->d_stamps, (size_t)w * 8, hipMemcpyDeviceToHost));
  return (int)(w / kStampSlots);
#else
  (void)h; (void)host_out; (void)max_words;
  return PDS_EUNSUPPORTED;
#endif
}
