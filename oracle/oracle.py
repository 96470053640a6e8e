"""ctypes binding of the CPU ORACLE (oracle/libphoenix_oracle.so).

TEST INFRASTRUCTURE -- NOT the product.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module (see oracle/phoenix_oracle.h).  The product path
(phoenix-drone-simulation_amd -> libpds_hip.so) never imports it and has no CPU fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libphoenix_oracle.so")

TASK_HOVER, TASK_CIRCLE, TASK_TAKEOFF = 0, 1, 2
TASK_IDS = {"hover": 0, "circle": 1, "takeoff": 2}
MAX_OBS, HIST, MAX_LAT = 24, 2, 8


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


class Config(C.Structure):
    _fields_ = [
        ("task", C.c_int32), ("use_motor_dynamics", C.c_int32), ("use_ground_effect", C.c_int32),
        ("observation_noise", C.c_int32), ("aggregate_phy_steps", C.c_int32),
        ("enable_reset_distribution", C.c_int32), ("max_episode_steps", C.c_int32),
        ("obs_rate", C.c_int32),
        ("domain_randomization", C.c_double), ("motor_thrust_noise", C.c_double),
        ("time_step", C.c_double), ("motor_time_constant", C.c_double),
        ("penalty_action", C.c_double), ("penalty_angle", C.c_double), ("penalty_spin", C.c_double),
        ("penalty_terminal", C.c_double), ("penalty_velocity", C.c_double), ("ARP", C.c_double),
        ("target_pos", C.c_double * 3), ("init_xyz", C.c_double * 3),
        ("init_rpy", C.c_double * 3), ("init_xyz_dot", C.c_double * 3), ("init_rpy_dot", C.c_double * 3),
        ("control_mode", C.c_int32), ("use_latency", C.c_int32), ("latency", C.c_double),
        ("ref_points", C.c_int32), ("pad3_", C.c_int32),
    ]


class ResetSample(C.Structure):
    _fields_ = [
        ("pos_offset", C.c_double * 3), ("rpy", C.c_double * 3), ("vel", C.c_double * 3),
        ("omega", C.c_double * 3), ("motor_x", C.c_double * 4), ("action", C.c_double * 4),
        ("dr_dt", C.c_double), ("dr_m", C.c_double), ("dr_J", C.c_double * 3),
        ("dr_ftf0", C.c_double), ("dr_ftf1", C.c_double), ("dr_T", C.c_double * 4),
        ("dr_t2w", C.c_double * 4), ("ref_offset", C.c_int32), ("pad_", C.c_int32),
        ("action_buf", (C.c_double * 4) * (MAX_LAT - 1)),
    ]


class Rng(C.Structure):
    _fields_ = [("z", C.POINTER(C.c_double)), ("u", C.POINTER(C.c_double)),
                ("iz", C.c_int64), ("iu", C.c_int64), ("nz", C.c_int64), ("nu", C.c_int64)]


class Constants(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "M", "L", "THRUST2WEIGHT_RATIO", "IXX", "IYY", "IZZ", "KF", "KM", "GND_EFF_COEFF",
        "PROP_RADIUS", "FORCE_TORQUE_FACTOR_0", "FORCE_TORQUE_FACTOR_1", "G", "GRAVITY",
        "MAX_THRUST", "MAX_TORQUE", "HOVER_X", "HOVER_ACTION", "MAX_RPM", "GND_EFF_H_CLIP")]


def _env_struct(real):
    class Env(C.Structure):
        _fields_ = [
            ("xyz", real * 3), ("rpy", real * 3), ("quat", real * 4), ("xyz_dot", real * 3),
            ("rpy_dot", real * 3),
            ("x", real * 4), ("y", real * 4), ("last_action", real * 4),
            ("env_last_action", real * 4), ("pwm", real * 4),
            ("act_hist", (real * 4) * HIST), ("obs_hist", (real * MAX_OBS) * HIST),
            ("target_pos", real * 3),
            ("dt", real), ("m", real), ("J", real * 3), ("ftf0", real), ("ftf1", real),
            ("A", real * 4), ("B", real * 4), ("K", real * 4), ("T", real * 4), ("t2w", real * 4),
            ("T_s", real),
            ("ou", real * 4), ("gyro_bias", real * 3), ("lpf", real * 3), ("kf_state", real * 17),
            ("rate_int", real * 3), ("rate_err", real * 3), ("att_int", real * 3), ("att_err", real * 3),
            ("action_buffer", (real * 4) * MAX_LAT),
            ("iteration", C.c_int32), ("ref_offset", C.c_int32), ("elapsed_steps", C.c_int32),
            ("obs_len", C.c_int32),
            ("use_latency", C.c_int32), ("buf_size", C.c_int32), ("action_idx", C.c_int32),
            ("hist_alias", C.c_int32 * HIST), ("last_action_alias", C.c_int32),
        ]
    return Env


EnvF64 = _env_struct(C.c_double)
EnvF32 = _env_struct(C.c_float)

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.po_max_threads.restype = C.c_int
        for suf, real in (("_f64", C.c_double), ("_f32", C.c_float)):
            getattr(_lib, "po_compute_reward" + suf).restype = real
            getattr(_lib, "po_compute_cost" + suf).restype = real
            getattr(_lib, "po_sizeof_env" + suf).restype = C.c_int
        assert _lib.po_sizeof_env_f64() == C.sizeof(EnvF64), "EnvF64 layout drifted from po_env_f64"
        assert _lib.po_sizeof_env_f32() == C.sizeof(EnvF32), "EnvF32 layout drifted from po_env_f32"
    return _lib


def constants():
    k = Constants()
    lib().po_get_constants(C.byref(k))
    return {n: getattr(k, n) for n, _ in Constants._fields_}


def default_config(task, **overrides):
    """Config with the reference's ctor defaults; `overrides` use the reference kwarg names."""
    c = Config()
    lib().po_default_config(int(TASK_IDS.get(task, task)), C.byref(c))
    for k, v in overrides.items():
        if k in ("target_pos", "init_xyz", "init_rpy", "init_xyz_dot", "init_rpy_dot"):
            for i in range(3):
                getattr(c, k)[i] = float(v[i])
        elif k == "observation_noise":
            c.observation_noise = 1 if v > 0 else 0
        elif k == "use_latency":
            c.use_latency = int(bool(v))
        elif k == "observation_frequency":
            # obs_rate = sim_freq // observation_frequency (envs/base.py:108), sim_freq = 100 on the Simple envs
            # (envs/hover.py:262); Circle: num_ref_points = 3 * observation_frequency (envs/circle.py:49)
            c.obs_rate = int(100 // int(v))
            if c.task == TASK_CIRCLE:
                c.ref_points = 3 * int(v)
        elif k == "control_mode":
            c.control_mode = {"PWM": 0, "AttitudeRate": 1, "Attitude": 2}[v] if isinstance(v, str) else int(v)
        else:
            assert hasattr(c, k), k
            setattr(c, k, v)
    return c


def philox4x32_10(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().po_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    return [int(v) for v in out]


def _arr(vals, real):
    return (real * len(vals))(*[float(v) for v in vals])


class OracleEnv:
    """One env of the oracle (`precision` 'f64' or 'f32'); mirrors the reference's gymnasium surface."""

    def __init__(self, task, precision="f64", **kwargs):
        self.L = lib()
        self.suf = "_" + precision
        self.real = C.c_double if precision == "f64" else C.c_float
        self.np_real = np.float64 if precision == "f64" else np.float32
        self.cfg = task if isinstance(task, Config) else default_config(task, **kwargs)
        self.env = (EnvF64 if precision == "f64" else EnvF32)()
        self._f("po_env_init")(C.byref(self.cfg), C.byref(self.env))
        self.obs_dim = self._f("po_obs_dim")(C.byref(self.cfg))
        self.rng = None

    def _f(self, name):
        return getattr(self.L, name + self.suf)

    def set_streams(self, z=None, u=None):
        """Replay recorded standard-normal / uniform[0,1) streams (reference draw order)."""
        self._z = np.ascontiguousarray(z if z is not None else [], dtype=np.float64)
        self._u = np.ascontiguousarray(u if u is not None else [], dtype=np.float64)
        self.rng = Rng(self._z.ctypes.data_as(C.POINTER(C.c_double)),
                       self._u.ctypes.data_as(C.POINTER(C.c_double)), 0, 0, self._z.size, self._u.size)

    def _rng_ref(self):
        return C.byref(self.rng) if self.rng is not None else None

    # -- state access by field name (numpy in / out) --
    def get(self, name):
        v = getattr(self.env, name)
        return np.array(v, dtype=self.np_real) if hasattr(v, "__len__") else v

    def set(self, name, val):
        cur = getattr(self.env, name)
        if hasattr(cur, "__len__"):
            flat = np.asarray(val, dtype=np.float64).reshape(-1)
            dst = np.ctypeslib.as_array(cur).reshape(-1)
            dst[:] = flat
        else:
            setattr(self.env, name, type(cur)(val))

    def reset(self, sample=None):
        s = sample if isinstance(sample, ResetSample) else make_reset_sample(**(sample or {}))
        obs = np.zeros(self.obs_dim, dtype=self.np_real)
        self._f("po_reset")(C.byref(self.cfg), C.byref(self.env), C.byref(s), self._rng_ref(),
                            obs.ctypes.data_as(C.POINTER(self.real)))
        return obs

    def step(self, action):
        a = _arr(action, self.real)
        obs = np.zeros(self.obs_dim, dtype=self.np_real)
        r, cost = self.real(), self.real()
        term, trunc = C.c_int32(), C.c_int32()
        self._f("po_step")(C.byref(self.cfg), C.byref(self.env), a, self._rng_ref(),
                           obs.ctypes.data_as(C.POINTER(self.real)), C.byref(r), C.byref(term),
                           C.byref(trunc), C.byref(cost))
        return obs, r.value, bool(term.value), bool(trunc.value), cost.value

    def step_forward(self, action):
        self._f("po_step_forward")(C.byref(self.cfg), C.byref(self.env), _arr(action, self.real),
                                   self._rng_ref())

    def set_latency(self, new_latency):
        self._f("po_set_latency")(C.byref(self.cfg), C.byref(self.env), C.c_double(new_latency))

    def philox_reset_sample(self, seed, env_id, tick):
        s = ResetSample()
        self._f("po_philox_reset_sample")(C.byref(self.cfg), C.c_uint64(seed), C.c_uint64(env_id),
                                          C.c_uint64(tick), C.byref(s))
        return s


def make_reset_sample(**kw):
    s = ResetSample()
    for k, v in kw.items():
        cur = getattr(s, k)
        if hasattr(cur, "__len__"):
            flat = np.asarray(v, dtype=np.float64).reshape(-1)
            dst = np.ctypeslib.as_array(cur).reshape(-1)  # (1-D and 2-D members alike)
            dst[:flat.size] = flat
        else:
            setattr(s, k, int(v) if k == "ref_offset" else float(v))
    return s


def sample_to_dict(s):
    out = {}
    for name, _ in ResetSample._fields_:
        v = getattr(s, name)
        out[name] = np.array(v, dtype=np.float64) if hasattr(v, "__len__") else v
    return out


def quat_from_euler(rpy, precision="f64"):
    real = C.c_double if precision == "f64" else C.c_float
    q = (real * 4)()
    getattr(lib(), "po_quat_from_euler_" + precision)(_arr(rpy, real), q)
    return np.array(q)


def matrix_from_quat(q, precision="f64"):
    real = C.c_double if precision == "f64" else C.c_float
    R = (real * 9)()
    getattr(lib(), "po_matrix_from_quat_" + precision)(_arr(q, real), R)
    return np.array(R).reshape(3, 3)


def euler_from_quat(q, precision="f64"):
    real = C.c_double if precision == "f64" else C.c_float
    e = (real * 3)()
    getattr(lib(), "po_euler_from_quat_" + precision)(_arr(q, real), e)
    return np.array(e)


def bullet_readback_quat(q, precision="f64"):
    """The quaternion getBasePositionAndOrientation returns for a stored `q` (btMatrix3x3 round trip)."""
    real = C.c_double if precision == "f64" else C.c_float
    out = (real * 4)()
    getattr(lib(), "po_bullet_readback_quat_" + precision)(_arr(q, real), out)
    return np.array(out)


class OracleBatch:
    """N oracle envs stepped with OpenMP (the timed `cpu_baseline`, and the lockstep auto-reset
    semantics the HIP path is checked against)."""

    def __init__(self, task, n, precision="f32", nthreads=0, **kwargs):
        self.L = lib()
        self.suf = "_" + precision
        self.real = C.c_double if precision == "f64" else C.c_float
        self.np_real = np.float64 if precision == "f64" else np.float32
        self.cfg = task if isinstance(task, Config) else default_config(task, **kwargs)
        self.n = int(n)
        Env = EnvF64 if precision == "f64" else EnvF32
        self.nthreads = nthreads or self.L.po_max_threads()
        # untouched memory (a ctypes array would be zero-filled by THIS thread): po_env_init_batch_mt touches every
        # struct from the OpenMP thread that later steps it (same static schedule) -> first-touch NUMA placement
        self._env_mem = np.empty(self.n * C.sizeof(Env), np.uint8)
        self.envs = (Env * self.n).from_buffer(self._env_mem)
        getattr(self.L, "po_env_init_batch_mt" + self.suf)(C.byref(self.cfg), self.envs, C.c_int64(self.n), C.c_int(self.nthreads))
        self.obs_dim = getattr(self.L, "po_obs_dim" + self.suf)(C.byref(self.cfg))
        self.obs = np.zeros((self.n, self.obs_dim), self.np_real)
        self.final_obs = np.zeros((self.n, self.obs_dim), self.np_real)
        self.reward = np.zeros(self.n, self.np_real)
        self.cost = np.zeros(self.n, self.np_real)
        self.terminated = np.zeros(self.n, np.uint8)
        self.truncated = np.zeros(self.n, np.uint8)

    def _p(self, a, t=None):
        return a.ctypes.data_as(C.POINTER(t or self.real))

    def reset(self, seed, tick):
        if not getattr(self, "_ctor_done", False):  # the constructor's compute_observation() (envs/base.py:142)
            getattr(self.L, "po_ctor_noise_batch" + self.suf)(C.byref(self.cfg), self.envs, C.c_int64(self.n), C.c_uint64(seed))
            self._ctor_done = True
        getattr(self.L, "po_reset_batch" + self.suf)(
            C.byref(self.cfg), self.envs, C.c_int64(self.n), self._p(self.obs), C.c_uint64(seed),
            C.c_uint64(tick), C.c_int(self.nthreads))
        return self.obs

    def step(self, actions, seed=0, tick=0, auto_reset=True):
        a = np.ascontiguousarray(actions, dtype=self.np_real)
        assert a.shape == (self.n, 4)
        getattr(self.L, "po_step_batch" + self.suf)(
            C.byref(self.cfg), self.envs, C.c_int64(self.n), self._p(a), self._p(self.obs),
            self._p(self.reward), self._p(self.terminated, C.c_uint8),
            self._p(self.truncated, C.c_uint8), self._p(self.cost), self._p(self.final_obs),
            C.c_uint64(seed), C.c_uint64(tick), C.c_int(1 if auto_reset else 0),
            C.c_int(self.nthreads))
        return self.obs, self.reward, self.terminated, self.truncated, self.cost

    def field(self, name):
        """Gather one state field of all envs into an [N, k] array."""
        first = getattr(self.envs[0], name)
        if hasattr(first, "__len__"):
            return np.array([np.array(getattr(e, name)).reshape(-1) for e in self.envs], dtype=self.np_real)
        return np.array([getattr(e, name) for e in self.envs])
