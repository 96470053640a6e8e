"""Helpers shared by the parity tests: load tests/golden/*.npz (vectors generated from the
reference by oracle/refgen/gen_golden.py) and replay them through the CPU oracle."""
import json
import os

import numpy as np

from oracle import oracle as po

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SAMPLE_FIELDS = ["pos_offset", "rpy", "vel", "omega", "motor_x", "action", "dr_dt", "dr_m", "dr_J",
                 "dr_ftf0", "dr_ftf1", "dr_T", "dr_t2w", "ref_offset"]
DYN_FIELDS = ["xyz", "rpy", "quat", "xyz_dot", "rpy_dot"]


def scenario_names():
    with open(os.path.join(GOLDEN, "INDEX.json")) as f:
        return sorted(json.load(f).keys())


class Golden:
    def __init__(self, name):
        self.name = name
        self.d = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(str(self.d["meta"]))
        self.task = self.meta["task"]
        self.kwargs = dict(self.meta["kwargs"])
        self.motor = self.meta["motor"]
        self.E, self.T, self.D = self.meta["episodes"], self.meta["steps"], self.meta["obs_dim"]

    def __getitem__(self, k):
        return self.d[k]

    def oracle_kwargs(self):
        kw = dict(self.kwargs)
        kw["use_motor_dynamics"] = 1 if self.motor else 0
        if "enable_reset_distribution" in kw:
            kw["enable_reset_distribution"] = int(kw["enable_reset_distribution"])
        return kw

    def sample(self, ep):
        return {k: self.d["sample_" + k][ep] for k in SAMPLE_FIELDS}

    def n_valid(self, ep):
        return int(self.d["valid"][ep].sum())


def make_oracle(g, precision="f64"):
    return po.OracleEnv(g.task, precision=precision, **g.oracle_kwargs())


def begin_episode(env, g, ep):
    """Put the oracle env in the state the reference env had just before reset() of episode `ep`
    (persisting noise states + the stale rpy_dot that base.py:411 feeds into the gyro filter),
    install the recorded variate streams and the init_* overrides, then reset."""
    for k in ("init_xyz", "init_rpy", "init_xyz_dot", "init_rpy_dot"):
        for i in range(3):
            getattr(env.cfg, k)[i] = float(g[k][ep][i])
    env.set("ou", g["pre_ou"][ep])
    env.set("gyro_bias", g["pre_gyro_bias"][ep])
    env.set("rpy_dot", g["pre_rpy_dot"][ep])
    # the streams also hold the draws of the reset distribution itself (injected as `sample`)
    z0, z1 = int(g["z_off"][ep]) + int(g["z_skip"][ep]), int(g["z_off"][ep + 1])
    u0, u1 = int(g["u_off"][ep]) + int(g["u_skip"][ep]), int(g["u_off"][ep + 1])
    env.set_streams(g["z"][z0:z1], g["u"][u0:u1])
    return env.reset(g.sample(ep))


def assert_close(a, b, rtol, atol, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    if not np.all(err <= tol):
        i = int(np.argmax(err - tol))
        raise AssertionError(f"{what}: max violation at flat index {i}: got {a.reshape(-1)[i]!r} "
                             f"want {b.reshape(-1)[i]!r} (err {err.reshape(-1)[i]:.3e}, tol {tol.reshape(-1)[i]:.3e})")
