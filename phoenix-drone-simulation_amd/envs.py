"""Batched `*SimpleEnv-v0` environments with the reference's gymnasium reset()/step() surface.

Mirrors (same ids, kwarg names, defaults, observation layout, reward/termination/cost semantics):
  DroneBaseEnv            phoenix_drone_simulation/envs/base.py:23-475
  DroneHoverSimpleEnv     phoenix_drone_simulation/envs/hover.py:253-266
  DroneCircleSimpleEnv    phoenix_drone_simulation/envs/circle.py:286-299
  DroneTakeOffSimpleEnv   phoenix_drone_simulation/envs/takeoff.py:221-231
  registration            phoenix_drone_simulation/__init__.py:8-50 (max_episode_steps=500)

Differences that follow from batching (documented in INTEGRATION.md):
  * `num_envs` environments advance in lockstep; tensors live on the HIP device:
      actions [N,4] f32 -> obs [N,D] f32, reward [N] f32, terminated [N] bool, truncated [N] bool,
      info = {'cost': [N] f32, 'final_obs': [N,D] f32 (rows valid where terminated|truncated)}
  * the 500-step TimeLimit of the gymnasium registration is part of the env (`truncated`), and envs
    that finish are reset inside the same step (`auto_reset=True`) from a counter-based Philox
    stream keyed by (seed, global env id, tick) -- the reference draws from the global numpy stream.
All computation happens in libpds_hip.so (HIP, gfx950); there is no CPU path here.
"""
import contextlib
import ctypes as C
import sys

import numpy as np
import torch

from . import native

_NULL_CTX = contextlib.nullcontext()

try:  # gymnasium is optional (absent in the build image); spaces are duck-typed otherwise
    from gymnasium.spaces import Box as _GymBox
except Exception:  # pragma: no cover
    _GymBox = None


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # (device index) -> stream handle as int

class Box:
    """Minimal stand-in for gymnasium.spaces.Box (shape/low/high/dtype/sample/contains)."""

    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


def _box(low, high):
    if _GymBox is not None:
        return _GymBox(low, high, dtype=np.float32)
    return Box(low, high, dtype=np.float32)


_TASKS = {"hover": native.TASK_HOVER, "circle": native.TASK_CIRCLE, "takeoff": native.TASK_TAKEOFF}


class DroneVecEnv:
    """N CrazyFlie SimplePhysics environments stepped in lockstep on one MI355X.

    Keyword arguments are the reference's (envs/base.py:26-48, envs/hover.py:7-24); additional ones:
    num_envs, device, seed, env_id_base (global id of local env 0 for multi-GPU sharding),
    auto_reset, use_motor_dynamics (agents.py:284-288 branch), use_ground_effect
    (physics.py:27-58 formula as opt-in extension), init_xyz / init_rpy / init_xyz_dot / init_rpy_dot
    (the env.init_* attributes of envs/base.py:84-91), fresh_outputs (step() returns newly allocated
    tensors like the reference's fresh arrays, envs/base.py:311, instead of two alternating buffer sets
    owned by the env: ~20 us of host time per step for callers that keep observations around).
    """
    metadata = {'render.modes': []}
    task = None

    def __init__(self, num_envs=1, device=None, seed=0, env_id_base=0, auto_reset=True,
                 use_motor_dynamics=False, use_ground_effect=False, use_latency=False,
                 init_xyz=None, init_rpy=None, init_xyz_dot=None, init_rpy_dot=None,
                 # --- reference kwargs ---
                 aggregate_phy_steps=1, control_mode='PWM', observation_noise=1,
                 domain_randomization=0.10, target_pos=(0., 0., 1.0), penalty_action=1e-4,
                 penalty_angle=0., penalty_spin=None, penalty_terminal=100., penalty_velocity=None,
                 enable_reset_distribution=True, latency=0.015, motor_time_constant=0.080,
                 motor_thrust_noise=0.05, observation_frequency=100, observation_history_size=2,
                 render_mode=None, debug=False, max_episode_steps=500, fresh_outputs=False):
        if control_mode not in native.CONTROL_MODES:
            raise AssertionError(f'Control={control_mode} not found.')  # envs/agents.py:70-71
        if int(observation_history_size) < 1:
            raise AssertionError("observation_history_size >= 1")  # envs/base.py:135
        self.observation_history_size = int(observation_history_size)
        if use_latency and self.observation_history_size != 2:
            # the history entries that alias drone.action_buffer (envs/agents.py:386) are resolved inside
            # the kernel for the default history of 2 only
            raise NotImplementedError("use_latency with observation_history_size != 2")
        if render_mode not in (None, 'rgb_array'):
            raise NotImplementedError("rendering is out of scope (no Bullet world on this path)")
        if int(aggregate_phy_steps) < 1:
            raise AssertionError("aggregate_phy_steps >= 1")  # envs/base.py:104
        if not torch.cuda.is_available():
            raise RuntimeError("phoenix-drone-simulation_amd needs a HIP device (MI355X); there is no CPU fallback")
        self.lib = native.load()
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(f"device {dev} is not a HIP device")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        cfg = native.default_config(_TASKS[self.task])
        cfg.num_envs = int(num_envs)
        cfg.env_id_base = int(env_id_base)
        self.env_id_base = int(env_id_base)
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        cfg.device = self.device.index
        cfg.use_motor_dynamics = int(bool(use_motor_dynamics))
        cfg.use_ground_effect = int(bool(use_ground_effect))
        cfg.control_mode = native.CONTROL_MODES[control_mode]
        # CrazyFlieAgent(use_latency=..., latency=...), envs/agents.py:125,165,179-183.  The Simple agent of the
        # reference passes use_latency=False (agents.py:492); sim-opt callers switch it on with set_latency().
        cfg.use_latency = int(bool(use_latency))
        cfg.latency = float(latency)
        # obs_rate = sim_freq // observation_frequency (envs/base.py:108); Circle: 3 * observation_frequency
        # reference points (envs/circle.py:49).  Unsupported combinations are refused by pds_create.
        cfg.observation_frequency = int(observation_frequency)
        cfg.observation_noise = 1 if observation_noise > 0 else 0
        cfg.aggregate_phy_steps = int(aggregate_phy_steps)
        cfg.enable_reset_distribution = int(bool(enable_reset_distribution))
        cfg.max_episode_steps = int(max_episode_steps)
        cfg.auto_reset = int(bool(auto_reset))
        cfg.domain_randomization = float(domain_randomization)
        cfg.motor_thrust_noise = float(motor_thrust_noise)
        cfg.motor_time_constant = float(motor_time_constant)
        cfg.penalty_action = float(penalty_action)
        cfg.penalty_angle = float(penalty_angle)
        if penalty_spin is not None:
            cfg.penalty_spin = float(penalty_spin)
        cfg.penalty_terminal = float(penalty_terminal)
        if penalty_velocity is not None:
            cfg.penalty_velocity = float(penalty_velocity)
        for i in range(3):
            cfg.target_pos[i] = float(target_pos[i])
        # env.init_* attributes (envs/base.py:84-91) are mutated by simopt-style callers
        # (simopt/pybullet.py:233-248); here they are ctor kwargs
        for name, val in (("init_xyz", init_xyz), ("init_rpy", init_rpy),
                          ("init_xyz_dot", init_xyz_dot), ("init_rpy_dot", init_rpy_dot)):
            if val is not None:
                for i in range(3):
                    getattr(cfg, name)[i] = float(val[i])
        self.cfg = cfg
        self.num_envs = int(num_envs)
        self.domain_randomization = float(domain_randomization)
        self.observation_noise = observation_noise
        self.enable_reset_distribution = bool(enable_reset_distribution)
        self.aggregate_phy_steps = int(aggregate_phy_steps)
        self._max_episode_steps = int(max_episode_steps)
        self.render_mode = render_mode
        self.debug = debug
        self._handle = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.pds_create(C.byref(cfg), C.byref(self._handle))
        native.check(None, rc, "pds_create")
        self.obs_dim = self.lib.pds_obs_dim(self._handle)
        # The kernel produces the reference's default history of 2: [o(k-1), u(k-2), o(k), u(k-1)].  Other
        # sizes H (experiments/04_*: 1, 2, 4, 6, 8; envs/base.py:303-319) are composed on the device from
        # the kernel's newest half [o(k), u(k-1)]: `_hist` keeps the last H halves of every env.
        self._half = self.obs_dim // 2
        self._hist = None
        if self.observation_history_size != 2:
            self.obs_dim = self.observation_history_size * self._half
            self._hist = torch.zeros(self.num_envs, self.observation_history_size, self._half,
                                     dtype=torch.float32, device=self.device)
        self._hist_sets = None
        self._auto_reset = bool(auto_reset)
        self.act_dim = 4
        o_lim = 1000 * np.ones((self.obs_dim,), dtype=np.float32)  # envs/base.py:147-150
        a_lim = np.ones((self.act_dim,), dtype=np.float32)
        self.observation_space = self.single_observation_space = _box(-o_lim, o_lim)
        self.action_space = self.single_action_space = _box(-a_lim, a_lim)
        N, D = self.num_envs, 2 * self._half
        f32 = dict(dtype=torch.float32, device=self.device)
        # two output sets so that `o` and `next_o` of a rollout loop can be alive together
        self._bufs = [dict(obs=torch.zeros(N, D, **f32), reward=torch.zeros(N, **f32),
                           cost=torch.zeros(N, **f32),
                           terminated=torch.zeros(N, dtype=torch.uint8, device=self.device),
                           truncated=torch.zeros(N, dtype=torch.uint8, device=self.device),
                           final_obs=torch.zeros(N, D, **f32)) for _ in range(2)]
        self._flip = 0
        # per buffer set: the ctypes argument tuple of pds_step and the value returned by step(), built
        # once (the Python side of a step is ~10 calls otherwise; it matters when a 65 536-env step
        # takes 4 us on the GPU)
        for b in self._bufs:
            b["_args"] = tuple(C.c_void_p(b[k].data_ptr()) for k in
                               ("obs", "reward", "terminated", "truncated", "cost", "final_obs"))
            b["_ret"] = (b["obs"], b["reward"], b["terminated"].view(torch.bool), b["truncated"].view(torch.bool),
                         {"cost": b["cost"], "final_obs": b["final_obs"],
                          "final_observation": b["final_obs"]})  # gymnasium's VectorEnv key, same tensor
        self._shape = (self.num_envs, 4)
        self._kbufs = {}
        self._fresh = bool(fresh_outputs)
        self._last_obs = self._bufs[0]["obs"]

    def _fresh_bufs(self):
        """One newly allocated output set (fresh_outputs=True); final_obs rows are defined where an env finished."""
        N, D = self.num_envs, 2 * self._half
        f32 = dict(dtype=torch.float32, device=self.device)
        u8 = dict(dtype=torch.uint8, device=self.device)
        b = dict(obs=torch.empty(N, D, **f32), reward=torch.empty(N, **f32), cost=torch.empty(N, **f32),
                 terminated=torch.empty(N, **u8), truncated=torch.empty(N, **u8), final_obs=torch.zeros(N, D, **f32))
        b["_args"] = tuple(C.c_void_p(b[k].data_ptr()) for k in ("obs", "reward", "terminated", "truncated", "cost", "final_obs"))
        b["_ret"] = (b["obs"], b["reward"], b["terminated"].view(torch.bool), b["truncated"].view(torch.bool),
                     {"cost": b["cost"], "final_obs": b["final_obs"], "final_observation": b["final_obs"]})
        return b

    # ------------------------------------------------------------------ gymnasium surface ----
    @property
    def unwrapped(self):
        return self

    def _stream(self):
        return C.c_void_p(self._raw_stream())

    def _raw_stream(self):
        """hipStream_t of torch's CURRENT stream on the env's device, as an int (looked up on every call: the
        caller may have switched streams).  torch's raw-stream getter is ~3x cheaper than building a Stream
        object, which matters below ~10^4 envs where a step is host-bound."""
        get = _RAW_STREAM
        if get is not None:
            return get(self.device.index)
        return torch.cuda.current_stream(self.device).cuda_stream

    def _next_buf(self):
        if self._fresh:
            return self._fresh_bufs()
        self._flip ^= 1
        return self._bufs[self._flip]

    def reset(self, *, seed=None, options=None, mask=None):
        """Reset all envs (or those with mask != 0).  Like the reference (envs/base.py:382-431),
        `seed` and `options` are accepted and ignored: randomness is keyed by the ctor `seed`."""
        b = self._next_buf()
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            b["obs"].copy_(self._last_obs)  # envs outside the mask keep their last observation
        self._last_obs = b["obs"]
        rc = self.lib.pds_reset(self._handle, C.c_void_p(m.data_ptr()) if m is not None else None,
                                C.c_void_p(b["obs"].data_ptr()), self._stream())
        native.check(self._handle, rc, "pds_reset")
        if self._hist is not None:
            return self._refill_history(b["obs"], m), {}
        return b["obs"], {}

    def reset_from_samples(self, samples, mask=None):
        """Reset with caller-supplied draws: `samples` [N, native.SAMPLE_FLOATS] (layout native.SAMPLE_LAYOUT)."""
        s = torch.as_tensor(samples, dtype=torch.float32, device=self.device).contiguous()
        assert s.shape == (self.num_envs, native.SAMPLE_FLOATS)
        b = self._next_buf()
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            b["obs"].copy_(self._last_obs)  # envs outside the mask keep their last observation
        self._last_obs = b["obs"]
        rc = self.lib.pds_reset_from_samples(
            self._handle, C.c_void_p(m.data_ptr()) if m is not None else None,
            C.c_void_p(s.data_ptr()), C.c_void_p(b["obs"].data_ptr()), self._stream())
        native.check(self._handle, rc, "pds_reset_from_samples")
        if self._hist is not None:
            return self._refill_history(b["obs"], m), {}
        return b["obs"], {}

    # ---- observation_history_size != 2 -------------------------------------------------------
    def _reset_rows(self, row2):
        """History right after a reset (envs/base.py:417-431): H - 1 copies of the first reset
        observation, then the observation compute_history() appends -- the two halves of the kernel's
        reset row [o0, u0, o0', u0]."""
        H, half = self.observation_history_size, self._half
        old, new = row2[:, :half], row2[:, half:]
        return torch.cat([old[:, None].expand(-1, H - 1, -1), new[:, None]], 1)

    def _refill_history(self, row2, mask):
        fresh = self._reset_rows(row2)
        self._hist = fresh if mask is None else torch.where(mask.bool()[:, None, None], fresh, self._hist)
        return self._hist.reshape(self.num_envs, -1)

    def adopt_history(self, flat):
        """Take `flat` [N, H * half] as the current observation histories (the one-launch rollout, pds_rollout_history, advanced
        them in the kernel)."""
        if self._hist is None:
            raise ValueError("observation_history_size == 2: the handle keeps the observation itself")
        own = getattr(self, "_hist_own", None)
        if own is None:
            own = self._hist_own = torch.empty(self.num_envs, self.observation_history_size, self._half,
                                               dtype=torch.float32, device=self.device)
        own.copy_(flat.view_as(own))
        self._hist = own

    def _advance_history(self, ret):
        """One launch of pds_history_advance (csrc/pds_history.hip): shift the [N, H, half] history by the
        step's newest half, write the final history of the envs that finished and restart theirs from the reset
        row.  Like step()'s other outputs the returned tensors alternate between two env-owned sets (or are
        newly allocated with fresh_outputs=True); info['final_obs'] rows are defined where an env finished
        (without auto-reset it is the returned observation itself)."""
        obs2, reward, term, trunc, info = ret
        N, H, half = self.num_envs, self.observation_history_size, self._half
        if self._fresh:
            out = torch.empty_like(self._hist)
            fin = torch.zeros_like(self._hist) if self._auto_reset else None
        else:
            if self._hist_sets is None:
                self._hist_sets = [(torch.empty_like(self._hist), torch.zeros_like(self._hist) if self._auto_reset else None)
                                   for _ in range(2)]
                self._hist_flip = 0
            self._hist_flip ^= 1
            out, fin = self._hist_sets[self._hist_flip]
            if out.data_ptr() == self._hist.data_ptr():  # (after a reset() replaced self._hist with one of the sets)
                self._hist_flip ^= 1
                out, fin = self._hist_sets[self._hist_flip]
        # pds_history_advance has no handle argument: it launches on the CURRENT device, which must be the env's
        with (_NULL_CTX if self.device.index == torch.cuda.current_device() else torch.cuda.device(self.device)):
            rc = self.lib.pds_history_advance(N, half, H, obs2.data_ptr(), term.data_ptr(), trunc.data_ptr(),
                                              info["final_obs"].data_ptr() if self._auto_reset else None,
                                              int(self._auto_reset), self._hist.data_ptr(), out.data_ptr(),
                                              fin.data_ptr() if fin is not None else None, self._raw_stream())
        if rc != 0:
            native.check(self._handle, rc, "pds_history_advance")
        self._hist = out
        flat = out.reshape(N, -1)
        final = fin.reshape(N, -1) if fin is not None else flat
        return (flat, reward, term, trunc, {"cost": info["cost"], "final_obs": final, "final_observation": final})

    def _advance_history_torch(self, hist, ret):
        """The same update written with torch ops (the pre-round-2 implementation; tests compare the kernel with it).
        Returns (new history [N, H, half], final history [N, H, half])."""
        obs2, reward, term, trunc, info = ret
        half = self._half
        done = (term | trunc)[:, None]
        new = obs2[:, half:]
        newest = torch.where(done, info["final_obs"][:, half:], new) if self._auto_reset else new
        final_hist = torch.cat([hist[:, 1:], newest[:, None]], 1)
        if self._auto_reset:
            return torch.where(done[:, :, None], self._reset_rows(obs2), final_hist), final_hist
        return final_hist, final_hist

    def step(self, action, noise_variates=None):
        """env.step(action).  The returned tensors are OWNED by the env: two buffer sets alternate, so the
        result of a step stays valid during the next one (`o` and `next_o` of a rollout loop) and is
        overwritten by the one after -- clone() what must live longer (the reference returns fresh arrays,
        envs/base.py:311).  `noise_variates` [N, aggregate_phy_steps * 52] (native.STEP_NOISE_LAYOUT per physics sub-step) replaces the in-kernel
        Philox draws of the OU thrust noise / SensorNoise with caller-supplied standard variates
        (parity tests replay the reference's numpy draws this way)."""
        a = action
        if not (isinstance(a, torch.Tensor) and a.dtype == torch.float32 and a.device == self.device and a.is_contiguous()):
            a = torch.as_tensor(np.asarray(action) if not isinstance(action, torch.Tensor) else action,
                                dtype=torch.float32).to(self.device).contiguous()
        if a.shape != self._shape:
            raise ValueError(f"actions must have shape ({self.num_envs}, 4), got {tuple(a.shape)}")
        b = self._next_buf()
        self._last_obs = b["obs"]
        nv = None
        if noise_variates is not None:
            nv = torch.as_tensor(noise_variates, dtype=torch.float32).to(self.device).contiguous()
            assert nv.shape == (self.num_envs, native.NOISE_FLOATS * self.aggregate_phy_steps)  # one block per physics sub-step
            nv = C.c_void_p(nv.data_ptr())
        rc = self.lib.pds_step_with_variates(self._handle, a.data_ptr(), nv, *b["_args"],
                                             self._raw_stream())
        if rc != 0:
            native.check(self._handle, rc, "pds_step")
        if self._hist is not None:
            return self._advance_history(b["_ret"])
        return b["_ret"]

    def step_k(self, actions, out=None):
        """K open-loop env.step()s in ONE launch (`pds_step_k`): `actions` [K, N, 4] -> (obs [K, N, D],
        reward [K, N], terminated [K, N], truncated [K, N], info with cost [K, N] and final_obs [K, N, D]).
        Counterpart of the recorded-action replay loop of the reference's sim-opt
        (simopt/pybullet.py:163-176); bitwise identical to K calls of step().  The returned tensors are
        owned by the env and reused by the next step_k call with the same K (pass `out`, a dict of
        preallocated tensors with the same keys, to keep them)."""
        a = actions
        if not (isinstance(a, torch.Tensor) and a.dtype == torch.float32 and a.device == self.device and a.is_contiguous()):
            a = torch.as_tensor(np.asarray(actions) if not isinstance(actions, torch.Tensor) else actions,
                                dtype=torch.float32).to(self.device).contiguous()
        if a.dim() != 3 or tuple(a.shape[1:]) != self._shape:
            raise ValueError(f"actions must have shape (K, {self.num_envs}, 4), got {tuple(a.shape)}")
        K, N, D = int(a.shape[0]), self.num_envs, 2 * self._half
        if self._hist is not None and self._hist is not getattr(self, "_hist_own", None) and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("step_k with observation_history_size != 2: call it once eagerly before capturing it "
                               "(the history buffer is adopted at the first call)")
        b = out if out is not None else self._kbufs.get(K)
        if b is None:
            f32 = dict(dtype=torch.float32, device=self.device)
            u8 = dict(dtype=torch.uint8, device=self.device)
            b = dict(obs=torch.empty(K, N, D, **f32), reward=torch.empty(K, N, **f32), cost=torch.empty(K, N, **f32),
                     terminated=torch.empty(K, N, **u8), truncated=torch.empty(K, N, **u8),
                     final_obs=torch.zeros(K, N, D, **f32))
            self._kbufs = {K: b}  # one cached set
        rc = self.lib.pds_step_k(self._handle, K, a.data_ptr(), b["obs"].data_ptr(), b["reward"].data_ptr(),
                                 b["terminated"].data_ptr(), b["truncated"].data_ptr(), b["cost"].data_ptr(),
                                 b["final_obs"].data_ptr(), self._raw_stream())
        if rc != 0:
            native.check(self._handle, rc, "pds_step_k")
        obs, final = b["obs"], b["final_obs"]
        self._last_obs = obs[K - 1]  # (what a masked reset() copies for the envs outside the mask)
        if self._hist is not None:
            # observation_history_size != 2: the K rows of the launch run through pds_history_advance one after the other
            # (K more launches; the env state itself advanced in one)
            H, half = self.observation_history_size, self._half
            if "obs_hist" not in b:
                b["obs_hist"] = torch.empty(K, N, H, half, dtype=torch.float32, device=self.device)
                b["final_hist"] = torch.zeros(K, N, H, half, dtype=torch.float32, device=self.device)
            # the history lives in ONE env-owned buffer across step_k calls (a hipGraph that captured a call re-reads and
            # re-writes that address at every replay, so the history carries from replay to replay)
            own = getattr(self, "_hist_own", None)
            if own is None:
                own = self._hist_own = torch.empty_like(self._hist)
            if self._hist is not own:  # (never while capturing: refused above)
                own.copy_(self._hist)
                self._hist = own
            hist = own
            with (_NULL_CTX if self.device.index == torch.cuda.current_device() else torch.cuda.device(self.device)):
                for k in range(K):
                    rc = self.lib.pds_history_advance(N, half, H, obs[k].data_ptr(), b["terminated"][k].data_ptr(),
                                                      b["truncated"][k].data_ptr(),
                                                      final[k].data_ptr() if self._auto_reset else None, int(self._auto_reset),
                                                      hist.data_ptr(), b["obs_hist"][k].data_ptr(),
                                                      b["final_hist"][k].data_ptr() if self._auto_reset else None,
                                                      self._raw_stream())
                    if rc != 0:
                        native.check(self._handle, rc, "pds_history_advance")
                    hist = b["obs_hist"][k]
            own.copy_(hist)  # (the cached set is rewritten by the next call)
            obs = b["obs_hist"].reshape(K, N, H * half)
            final = b["final_hist"].reshape(K, N, H * half) if self._auto_reset else obs
        return (obs, b["reward"], b["terminated"].view(torch.bool), b["truncated"].view(torch.bool),
                {"cost": b["cost"], "final_obs": final, "final_observation": final})

    def set_latency(self, new_latency):
        """CrazyFlieAgent.set_latency (envs/agents.py:388-404): below one time step the delay is switched
        off, otherwise buf_size = int(latency / time_step); the action buffer of every env is zeroed."""
        if self._hist is not None and float(new_latency) >= float(self.cfg.time_step):
            # the aliased action-history entries of the first two steps after a reset (agents.py:386, base.py:425-426)
            # are resolved in-kernel for observation_history_size == 2 only -- same refusal as the constructor's
            raise NotImplementedError("set_latency >= one time step with observation_history_size != 2 is not built")
        rc = self.lib.pds_set_latency(self._handle, float(new_latency))
        native.check(self._handle, rc, "pds_set_latency")

    @property
    def latency_steps(self):
        return int(self.lib.pds_latency_steps(self._handle))

    def sync_tick(self):
        """Re-read the device-side tick after replaying a captured hipGraph of steps (synchronises)."""
        return int(self.lib.pds_sync_tick(self._handle, self._stream()))

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle:
            self.lib.pds_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        # At interpreter shutdown the HIP runtime (and a profiler's tool library) may already be tearing down:
        # freeing device memory from here was seen to hang a process under rocprofv3.  The driver reclaims the
        # memory with the process; only a collected env of a live interpreter is destroyed here.
        try:
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass

    def render(self):
        if self.render_mode == "rgb_array":
            return np.array([])  # envs/base.py:377
        raise NotImplementedError

    # ------------------------------------------------------------------ state access ---------
    def get_state(self, name):
        fid = native.FIELDS[name]
        w = self.lib.pds_field_width(fid)
        dt = torch.int32 if name in native.INT_FIELDS else torch.float32
        out = torch.zeros(self.num_envs, w, dtype=dt, device=self.device)
        rc = self.lib.pds_get_state(self._handle, fid, C.c_void_p(out.data_ptr()), self._stream())
        native.check(self._handle, rc, f"pds_get_state({name})")
        return out

    def set_state(self, name, value):
        fid = native.FIELDS[name]
        w = self.lib.pds_field_width(fid)
        dt = torch.int32 if name in native.INT_FIELDS else torch.float32
        v = torch.as_tensor(value).to(device=self.device, dtype=dt).reshape(self.num_envs, w).contiguous()
        rc = self.lib.pds_set_state(self._handle, fid, C.c_void_p(v.data_ptr()), self._stream())
        native.check(self._handle, rc, f"pds_set_state({name})")
        torch.cuda.current_stream(self.device).synchronize()  # `v` may be a temporary

    def count_nonfinite(self):
        """Number of envs whose dynamic state holds a NaN / Inf (diagnostic; synchronises)."""
        out = C.c_int64(0)
        rc = self.lib.pds_count_nonfinite(self._handle, C.byref(out), self._stream())
        native.check(self._handle, rc, "pds_count_nonfinite")
        return int(out.value)

    def state_dict(self):
        """Everything a bit-exact continuation needs: every state field (`pds_get_state`) + the RNG tick.
        Load it into an env created with the same kwargs (`load_state_dict`)."""
        sd = {name: self.get_state(name) for name in native.FIELDS if name != "quat"}  # quat is derived
        sd["tick"] = int(self.tick)
        if self._hist is not None:
            sd["observation_history"] = self._hist.clone()
        return sd

    def load_state_dict(self, sd):
        for name, v in sd.items():
            if name == "observation_history":
                self._hist = v.to(self.device).clone()
            elif name != "tick":
                self.set_state(name, v)
        rc = self.lib.pds_set_tick(self._handle, int(sd["tick"]))
        native.check(self._handle, rc, "pds_set_tick")

    @property
    def bytes_per_env_step(self):
        return self.lib.pds_bytes_per_env_step(self._handle)

    def bytes_per_env_step_k(self, k_steps):
        return self.lib.pds_bytes_per_env_step_k(self._handle, int(k_steps))

    @property
    def tick(self):
        return int(self.lib.pds_tick(self._handle))


class DroneHoverSimpleEnv(DroneVecEnv):
    """envs/hover.py:253-266 (penalty_spin 1e-4, penalty_velocity 0, ARP 0)."""
    task = "hover"


class DroneCircleSimpleEnv(DroneVecEnv):
    """envs/circle.py:286-299 (penalty_spin 1e-3, penalty_velocity 1e-4, ARP 1e-3)."""
    task = "circle"


class DroneTakeOffSimpleEnv(DroneVecEnv):
    """envs/takeoff.py:221-231."""
    task = "takeoff"

    def __init__(self, **kwargs):
        if kwargs.get("aggregate_phy_steps", 1) != 1 or kwargs.get("control_mode", "PWM") != "PWM":
            raise TypeError("DroneTakeOffSimpleEnv fixes aggregate_phy_steps=1, control_mode='PWM' (envs/takeoff.py:224-225)")
        super().__init__(**kwargs)


registry = {}


def register(id, entry_point, max_episode_steps=500):
    registry[id] = (entry_point, max_episode_steps)


# phoenix_drone_simulation/__init__.py:8-50 (the Bullet ids are out of scope)
register('DroneHoverSimpleEnv-v0', DroneHoverSimpleEnv, 500)
register('DroneCircleSimpleEnv-v0', DroneCircleSimpleEnv, 500)
register('DroneTakeOffSimpleEnv-v0', DroneTakeOffSimpleEnv, 500)


def make(id, **kwargs):
    """Counterpart of gym.make(id, **kwargs) (algs/iwpg/iwpg.py:72-75) returning a batched env."""
    if id not in registry:
        raise KeyError(f"unknown env id {id!r}; accelerated ids: {sorted(registry)}")
    cls, max_steps = registry[id]
    kwargs.setdefault("max_episode_steps", max_steps)
    return cls(**kwargs)
