#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

Imports `phoenix_drone_simulation` from /root/reference (read-only) with the stand-in modules of
oracle/refgen/standins/ on sys.path for the absent third-party imports (pybullet, pybullet_utils,
pybullet_data, gymnasium).  Only DATA (inputs + expected outputs) is written to tests/golden/ --
no reference source or bytecode.  /root/reference does not exist on the GPU box; nothing at test time
reads it.

All randomness of the reference comes from the global np.random functions; they are patched here by
recording wrappers that implement numpy's own transformations (normal = loc + scale*z,
uniform = low + (high-low)*u) on top of recorded standard streams, so that the oracle can replay
the very same variates ("treat RNG outputs as inputs captured by the oracle").

Usage: python oracle/refgen/gen_golden.py [--out tests/golden]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "standins"))
sys.path.insert(0, "/root/reference")

import gymnasium as gym  # noqa: E402  (stand-in)
import phoenix_drone_simulation  # noqa: E402,F401  (the reference; registers the env ids)
from phoenix_drone_simulation.envs.physics import SimplePhysics  # noqa: E402

ENV_IDS = {"hover": "DroneHoverSimpleEnv-v0", "circle": "DroneCircleSimpleEnv-v0",
           "takeoff": "DroneTakeOffSimpleEnv-v0"}


# ------------------------------------------------------------------------------------------------
# recording RNG
# ------------------------------------------------------------------------------------------------
class Recorder:
    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.z, self.u, self.calls = [], [], []

    def clear(self):
        self.z, self.u, self.calls = [], [], []

    def _std_normal(self, size):
        z = self.rs.standard_normal(size)
        self.z.extend(np.asarray(z, dtype=np.float64).reshape(-1).tolist())
        return z

    def _std_uniform(self, size):
        u = self.rs.random_sample(size)
        self.u.extend(np.asarray(u, dtype=np.float64).reshape(-1).tolist())
        return u

    def normal(self, loc=0.0, scale=1.0, size=None):
        shape = size if size is not None else np.broadcast(np.asarray(loc), np.asarray(scale)).shape
        out = np.asarray(loc) + np.asarray(scale) * self._std_normal(shape if shape != () else None)
        self.calls.append(("normal", np.array(out, dtype=np.float64).reshape(-1)))
        return out

    def randn(self, *shape):
        out = self._std_normal(shape if shape else None)
        self.calls.append(("randn", np.array(out, dtype=np.float64).reshape(-1)))
        return out

    def uniform(self, low=0.0, high=1.0, size=None):
        low, high = np.asarray(low), np.asarray(high)
        shape = size if size is not None else np.broadcast(low, high).shape
        out = low + (high - low) * self._std_uniform(shape if shape != () else None)
        self.calls.append(("uniform", np.array(out, dtype=np.float64).reshape(-1)))
        return out

    def randint(self, low, high=None, size=None):
        out = self.rs.randint(low, high, size)
        self.calls.append(("randint", np.array(out, dtype=np.float64).reshape(-1)))
        return out

    def install(self):
        np.random.normal = self.normal
        np.random.randn = self.randn
        np.random.uniform = self.uniform
        np.random.randint = self.randint


# ------------------------------------------------------------------------------------------------
# state capture
# ------------------------------------------------------------------------------------------------
STATE_FIELDS = ["xyz", "rpy", "quat", "xyz_dot", "rpy_dot", "x", "last_action", "env_last_action",
                "act_hist", "obs_hist", "target_pos", "dt", "m", "J", "ftf0", "ftf1", "A", "B", "K",
                "ou", "gyro_bias", "lpf", "kf_state", "iteration", "ref_offset",
                "rate_int", "rate_err", "att_int", "att_err",
                "action_buffer", "action_idx", "buf_size", "use_latency", "hist_alias", "last_action_alias"]
MAX_LAT = 8


def capture(env):
    e = env.unwrapped
    d = e.drone
    oh = np.zeros((2, 24))
    for i, o in enumerate(e.observation_history):
        oh[i, :len(o)] = o
    ah = np.zeros((2, 4))
    alias = np.zeros(2, np.int32)
    for i, a in enumerate(e.action_history):
        ah[i] = np.array(a, dtype=np.float64)
        # the deque entry may still BE the view action_buffer[-1, :] handed out by drone.reset() (agents.py:386)
        alias[i] = int(isinstance(a, np.ndarray) and np.shares_memory(a, d.action_buffer))
    ab = np.zeros((MAX_LAT, 4))
    ab[:d.action_buffer.shape[0]] = d.action_buffer
    st = dict(
        xyz=np.array(d.xyz, dtype=np.float64), rpy=np.array(d.rpy, dtype=np.float64),
        quat=np.array(d.quaternion, dtype=np.float64), xyz_dot=np.array(d.xyz_dot, dtype=np.float64),
        rpy_dot=np.array(d.rpy_dot, dtype=np.float64), x=np.array(d.x, dtype=np.float64),
        last_action=np.array(d.last_action, dtype=np.float64),
        env_last_action=np.array(e.last_action, dtype=np.float64),
        act_hist=ah,
        obs_hist=oh, target_pos=np.array(e.target_pos, dtype=np.float64),
        dt=float(e.physics.time_step), m=float(d.m), J=np.diag(d.J).astype(np.float64),
        ftf0=float(d.force_torque_factor_0), ftf1=float(d.force_torque_factor_1),
        A=np.ones(4) * d.A, B=np.ones(4) * d.B, K=np.ones(4) * d.K,
        ou=np.array(d.thrust_noise.state, dtype=np.float64),
        gyro_bias=np.array(e.sensor_noise.gyro_bias, dtype=np.float64),
        lpf=np.ones(3) * np.asarray(e.gyro_lpf._x, dtype=np.float64),
        kf_state=np.array(e.state, dtype=np.float64).reshape(-1)[:17],
        iteration=int(e.iteration), ref_offset=int(getattr(e, "ref_offset", 0)),
        action_buffer=ab, action_idx=int(d.action_idx), buf_size=int(d.action_buffer.shape[0]),
        use_latency=int(bool(d.use_latency)), hist_alias=alias,
        last_action_alias=int(isinstance(d.last_action, np.ndarray) and np.shares_memory(d.last_action, d.action_buffer)),
    )
    # PID controller state (envs/control.py:133-134, 227-228, 239-241)
    ctl = d.control
    z3 = np.zeros(3)
    rate = getattr(ctl, "attitude_rate_controller", ctl if hasattr(ctl, "kp_att_rate") else None)
    st["rate_int"] = np.array(rate.integral, dtype=np.float64) if rate is not None else z3
    st["rate_err"] = np.array(rate.last_error, dtype=np.float64) if rate is not None else z3
    att = ctl if hasattr(ctl, "kps") else None
    st["att_int"] = np.array(att.integral, dtype=np.float64) if att is not None else z3
    st["att_err"] = np.array(att.last_error, dtype=np.float64) if att is not None else z3
    return st


def _action_rows(s, v):
    """np.random.normal(HOVER_ACTION, 0.02, size=action_buffer.shape): all rows but the last go to
    action_buf, the last one (drone.last_action) to action."""
    rows = v.reshape(-1, 4)
    s["action"] = rows[-1]
    s["action_buf"][:rows.shape[0] - 1] = rows[:-1]


def parse_reset_sample(task, calls, dr_on, motor_on, reset_dist, buf_size=1):
    """Map the recorded np.random calls of one reset() onto po_reset_sample fields
    (draw order: hover.py:203-228, circle.py:225-257, takeoff.py:188-191, base.py:261-287)."""
    s = dict(pos_offset=np.zeros(3), rpy=np.zeros(3), vel=np.zeros(3), omega=np.zeros(3),
             motor_x=np.zeros(4), action=np.zeros(4), dr_dt=0.0, dr_m=0.0, dr_J=np.zeros(3),
             dr_ftf0=0.0, dr_ftf1=0.0, dr_T=np.zeros(4), dr_t2w=np.zeros(4), ref_offset=0,
             action_buf=np.zeros((MAX_LAT - 1, 4)))
    it = iter(calls)

    used = dict(normal=0, uniform=0, randint=0)

    def nxt(kind, n):
        k, v = next(it)
        assert k == kind and v.size == n, (k, v.size, kind, n)
        used[kind] += n
        return v

    if reset_dist:
        if task == "hover":
            s["pos_offset"] = nxt("uniform", 3)
            s["rpy"] = nxt("uniform", 3).copy()
            s["rpy"][2] = nxt("uniform", 1)[0]
            s["vel"] = nxt("uniform", 3)
            s["omega"] = nxt("uniform", 3).copy()
            s["omega"][2] = nxt("uniform", 1)[0]
            s["motor_x"] = nxt("normal", 4)
            _action_rows(s, nxt("normal", 4 * buf_size))
        elif task == "circle":
            s["ref_offset"] = int(nxt("randint", 1)[0])
            s["pos_offset"] = nxt("uniform", 3)
            s["rpy"] = nxt("uniform", 3).copy()
            s["rpy"][2] = nxt("uniform", 1)[0]
            s["vel"] = nxt("uniform", 3)
            s["omega"][:2] = nxt("uniform", 2)
            s["omega"][2] = nxt("uniform", 1)[0]
            s["motor_x"] = nxt("normal", 4)
            _action_rows(s, nxt("normal", 4 * buf_size))
        else:
            s["pos_offset"][:2] = nxt("uniform", 2)
            s["rpy"][2] = nxt("uniform", 1)[0]
    if dr_on:
        s["dr_dt"] = nxt("uniform", 1)[0]
        s["dr_m"] = nxt("uniform", 1)[0]
        s["dr_J"] = nxt("uniform", 3)
        s["dr_ftf0"] = nxt("uniform", 1)[0]
        s["dr_ftf1"] = nxt("uniform", 1)[0]
        if motor_on:
            s["dr_T"] = nxt("uniform", 4)
            s["dr_t2w"] = nxt("uniform", 4)
    s["_z_skip"], s["_u_skip"] = used["normal"], used["uniform"]
    return s


SAMPLE_FIELDS = ["pos_offset", "rpy", "vel", "omega", "motor_x", "action", "dr_dt", "dr_m", "dr_J",
                 "dr_ftf0", "dr_ftf1", "dr_T", "dr_t2w", "ref_offset", "action_buf"]


# ------------------------------------------------------------------------------------------------
# scenario runner
# ------------------------------------------------------------------------------------------------
def run_scenario(name, task, kwargs, episodes, steps, action_fn, seed, motor=False,
                 init_override=None, full_episode=False, latency_on=False, set_latency=None):
    rec = Recorder(seed)
    rec.install()
    env = gym.make(ENV_IDS[task], **kwargs)
    e = env.unwrapped
    if motor:
        e.drone.use_motor_dynamics = True  # debug/compare_system_equations_with_PyBullet.py:26-30
    if latency_on:
        # the Simple agent is built with use_latency=False (agents.py:492); the flag is flipped the way
        # use_motor_dynamics is, the buffer keeps its ctor size max(1, int(latency // time_step)) (agents.py:180)
        e.drone.use_latency = True
    if set_latency is not None:
        e.drone.set_latency(set_latency)  # the sim-opt route, simopt/pybullet.py:248
    reset_dist = kwargs.get("enable_reset_distribution", True)
    dr_on = kwargs.get("domain_randomization", 0.10) > 0
    act_rs = np.random.RandomState(seed + 12345)
    D = env.observation_space.shape[0]
    E, T = episodes, steps
    out = {k: [] for k in ("reset_obs", "pre_ou", "pre_gyro_bias", "pre_rpy_dot", "z_off", "u_off",
                           "init_xyz", "init_rpy", "init_xyz_dot", "init_rpy_dot", "z_skip", "u_skip")}
    samples = {k: [] for k in SAMPLE_FIELDS}
    reset_state = {k: [] for k in STATE_FIELDS}
    step_state = {k: np.zeros((E, T) + np.shape(capture(env)[k])) for k in STATE_FIELDS}
    actions = np.zeros((E, T, 4))
    obs = np.zeros((E, T, D))
    reward = np.zeros((E, T))
    cost = np.zeros((E, T))
    terminated = np.zeros((E, T), np.uint8)
    truncated = np.zeros((E, T), np.uint8)
    valid = np.zeros((E, T), np.uint8)
    z_all, u_all = [], []
    for ep in range(E):
        if init_override is not None:
            ov = init_override(ep, act_rs)
            for k, v in ov.items():
                setattr(e, k, v)
            if "init_rpy" in ov:  # what DroneBaseEnv.__init__ does (envs/base.py:88-89)
                from phoenix_drone_simulation.envs.utils import get_quaternion_from_euler
                e.init_quaternion = get_quaternion_from_euler(np.asarray(ov["init_rpy"], dtype=np.float64))
        pre = capture(env)
        for k in ("init_xyz", "init_rpy", "init_xyz_dot", "init_rpy_dot"):
            out[k].append(np.array(getattr(e, k), dtype=np.float64))
        out["pre_ou"].append(pre["ou"]); out["pre_gyro_bias"].append(pre["gyro_bias"])
        out["pre_rpy_dot"].append(pre["rpy_dot"])
        rec.clear()
        out["z_off"].append(len(z_all)); out["u_off"].append(len(u_all))
        o, _ = env.reset()
        smp = parse_reset_sample(task, rec.calls, dr_on, motor, reset_dist, buf_size=e.drone.action_buffer.shape[0])
        for k in SAMPLE_FIELDS:
            samples[k].append(smp[k])
        out["z_skip"].append(smp["_z_skip"]); out["u_skip"].append(smp["_u_skip"])
        st = capture(env)
        for k in STATE_FIELDS:
            reset_state[k].append(st[k])
        out["reset_obs"].append(np.array(o, dtype=np.float64))
        for t in range(T):
            a = action_fn(ep, t, act_rs, e)
            o, r, term, trunc, info = env.step(a)
            actions[ep, t] = a
            obs[ep, t] = o
            reward[ep, t] = r
            cost[ep, t] = info["cost"]
            terminated[ep, t] = term
            truncated[ep, t] = trunc
            valid[ep, t] = 1
            st = capture(env)
            for k in STATE_FIELDS:
                step_state[k][ep, t] = st[k]
            if (term or trunc) and not full_episode:
                break
            if trunc:
                break
        z_all.extend(rec.z); u_all.extend(rec.u)
    out["z_off"].append(len(z_all)); out["u_off"].append(len(u_all))
    meta = dict(name=name, task=task, kwargs={k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in kwargs.items()},
                motor=bool(motor), latency_on=bool(latency_on), set_latency=set_latency,
                episodes=E, steps=T, obs_dim=int(D), seed=seed,
                generator="oracle/refgen/gen_golden.py", reference="SvenGronauer/phoenix-drone-simulation v1.1")
    arrays = dict(meta=np.array(json.dumps(meta)), actions=actions, obs=obs, reward=reward, cost=cost,
                  terminated=terminated, truncated=truncated, valid=valid,
                  z=np.array(z_all), u=np.array(u_all))
    for k, v in out.items():
        arrays[k] = np.array(v)
    for k in SAMPLE_FIELDS:
        arrays["sample_" + k] = np.array(samples[k])
    for k in STATE_FIELDS:
        arrays["reset_" + k] = np.array(reset_state[k])
        arrays["step_" + k] = step_state[k]
    return arrays


def act_random(scale=0.3, center=None):
    def fn(ep, t, rs, e):
        c = e.drone.HOVER_ACTION if center is None else center
        a = c + scale * rs.standard_normal(4)
        if ep % 4 == 3:  # some actions outside [-1, 1]: clipped in PWM.act only (control.py:98)
            a = a * 4.0
        return a
    return fn


def act_hover(ep, t, rs, e):
    return np.ones(4) * e.drone.HOVER_ACTION


DET = dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0)


def edge_hover(ep, rs):
    """Initial states next to every Hover termination / cost threshold (hover.py:89-129) and the
    z clip (physics.py:182); injected the way simopt does, through env.init_* with the reset
    distribution disabled."""
    d2r = np.pi / 180
    cases = [
        dict(init_xyz=np.array([0, 0, 0.2004], np.float32)),                 # z just above 0.2
        dict(init_xyz=np.array([0, 0, 0.1996], np.float32)),                 # just below
        dict(init_xyz=np.array([0, 0, 0.0003], np.float32), init_xyz_dot=np.array([0, 0, -1.0])),  # z clip
        dict(init_rpy=np.array([59.9 * d2r, 0, 0])), dict(init_rpy=np.array([60.2 * d2r, 0, 0.3])),
        dict(init_rpy=np.array([0, -59.9 * d2r, 1.0])), dict(init_rpy=np.array([0, -60.2 * d2r, -2.0])),
        dict(init_rpy_dot=np.array([299.0 * d2r, 0, 0])), dict(init_rpy_dot=np.array([0, 301.5 * d2r, 0])),
        dict(init_rpy_dot=np.array([0, 0, -300.5 * d2r])),
        dict(init_xyz=np.array([0.099, 0, 1], np.float32)), dict(init_xyz=np.array([0.101, 0, 1], np.float32)),
        dict(init_xyz=np.array([0, -0.101, 1], np.float32)), dict(init_xyz=np.array([0, 0, 1.201], np.float32)),
        dict(init_rpy=np.array([9.9 * d2r, 0, 0])), dict(init_rpy=np.array([0, 10.1 * d2r, 0])),
        dict(init_rpy_dot=np.array([0.2, 0, 0])), dict(init_rpy_dot=np.array([0, 0, 0.3])),
        dict(init_rpy=np.array([0.2, -0.3, 3.5])), dict(init_rpy=np.array([0.1, 0.2, -5.0])),  # yaw beyond +-pi
    ]
    base = dict(init_xyz=np.array([0, 0, 1], np.float32), init_rpy=np.zeros(3),
                init_xyz_dot=np.zeros(3), init_rpy_dot=np.zeros(3))
    base.update(cases[ep % len(cases)])
    return base


def edge_takeoff(ep, rs):
    cases = [dict(init_xyz=np.array([0, 0, 0.0795], np.float32)),
             dict(init_xyz=np.array([0, 0, 0.0805], np.float32)),
             dict(init_xyz=np.array([0.1, -0.2, 0.0125], np.float32))]
    base = dict(init_xyz=np.array([0, 0, 0.0125], np.float32), init_rpy=np.zeros(3),
                init_xyz_dot=np.zeros(3), init_rpy_dot=np.zeros(3))
    base.update(cases[ep % len(cases)])
    return base


def edge_circle(ep, rs):
    cases = [dict(init_xyz=np.array([0.2495, 0, 1], np.float32)),  # dist to ref[1] around 0.25
             dict(init_xyz=np.array([0.2560, 0, 1], np.float32)),
             dict(init_xyz=np.array([0, 0.18, 1.17], np.float32))]
    base = dict(init_xyz=np.array([0, 0, 1], np.float32), init_rpy=np.zeros(3),
                init_xyz_dot=np.zeros(3), init_rpy_dot=np.zeros(3))
    base.update(cases[ep % len(cases)])
    return base


def ground_effect_vectors(seed=7, n=128):
    """G7: BasePhysics.calculate_ground_effect (envs/physics.py:27-58) evaluated by the reference on
    random low-altitude states; link heights come from the stand-in getLinkStates."""
    rs = np.random.RandomState(seed)
    env = gym.make(ENV_IDS["takeoff"], **DET, enable_reset_distribution=False)
    e = env.unwrapped
    env.reset()
    from pybullet import getQuaternionFromEuler
    xyz, rpy, forces, ge, ok = [], [], [], [], []
    for i in range(n):
        p = np.array([rs.uniform(-.3, .3), rs.uniform(-.3, .3), rs.uniform(0.0, 0.3) if i % 8 else 0.0])
        a = np.array([rs.uniform(-.6, .6), rs.uniform(-.6, .6), rs.uniform(-3, 3)])
        if i % 16 == 5:
            a[0] = rs.choice([-1, 1]) * rs.uniform(np.pi / 2 + 0.01, 3.0)
        f = rs.uniform(0, e.drone.MAX_THRUST, size=4)
        e.drone.xyz, e.drone.rpy = p, a
        e.bc.resetBasePositionAndOrientation(e.drone.body_unique_id, p, getQuaternionFromEuler(a))
        flag, g = e.physics.calculate_ground_effect(f)
        xyz.append(p); rpy.append(a); forces.append(f); ge.append(np.array(g)); ok.append(flag)
    return dict(xyz=np.array(xyz), rpy=np.array(rpy), forces=np.array(forces), ge=np.array(ge),
                ok=np.array(ok, np.uint8), meta=np.array(json.dumps(dict(
                    name="ground_effect", source="BasePhysics.calculate_ground_effect envs/physics.py:27-58"))))


def constants_vectors():
    env = gym.make(ENV_IDS["hover"], **DET)
    d = env.unwrapped.drone
    names = ["M", "L", "THRUST2WEIGHT_RATIO", "IXX", "IYY", "IZZ", "KF", "KM", "GND_EFF_COEFF",
             "PROP_RADIUS", "FORCE_TORQUE_FACTOR_0", "FORCE_TORQUE_FACTOR_1", "G", "GRAVITY",
             "MAX_THRUST", "MAX_TORQUE", "HOVER_X", "HOVER_ACTION", "MAX_RPM", "GND_EFF_H_CLIP"]
    out = {n: float(getattr(d, n)) for n in names}
    out["A0"], out["B0"], out["K0"] = float(d.A[0]), float(d.B[0]), float(d.K)
    out["obs_dim"] = {}
    for task in ENV_IDS:
        out["obs_dim"][task] = {
            "noise_free": int(gym.make(ENV_IDS[task], **DET).observation_space.shape[0]),
            "noisy": int(gym.make(ENV_IDS[task]).observation_space.shape[0])}
    # circle / takeoff reference tables (circle.py:45-56, takeoff.py:43-47)
    out["circle_ref"] = gym.make(ENV_IDS["circle"], **DET).unwrapped.ref.tolist()
    out["takeoff_ref_z"] = gym.make(ENV_IDS["takeoff"], **DET).unwrapped.ref[:, 2].tolist()
    return out


def quaternion_vectors(seed=3, n=256):
    """pybullet-boundary convention (tests/test_quaternion.py:35-43) evaluated with the in-repo
    formula envs/utils.py:32-56 of the reference."""
    from phoenix_drone_simulation.envs.utils import get_quaternion_from_euler
    rs = np.random.RandomState(seed)
    rpy = rs.uniform(-2, 2, size=(n, 3))
    q = np.array([get_quaternion_from_euler(r) for r in rpy])
    return dict(rpy=rpy, quat=q)


def reference_cpu_rate(steps=20000):
    """Reference Python step-loop rate in THIS container (upper bound: Bullet calls are no-ops)."""
    res = {}
    for task in ENV_IDS:
        for label, kw in (("det", DET), ("defaults", {})):
            env = gym.make(ENV_IDS[task], **kw)
            env.reset()
            rs = np.random.RandomState(0)
            acts = env.unwrapped.drone.HOVER_ACTION + 0.1 * rs.standard_normal((steps, 4))
            t0 = time.perf_counter()
            for i in range(steps):
                _, _, term, trunc, _ = env.step(acts[i])
                if term or trunc:
                    env.reset()
            res[f"{task}_{label}"] = steps / (time.perf_counter() - t0)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "..", "tests", "golden"))
    ap.add_argument("--rate", action="store_true", help="also time the reference python loop")
    ap.add_argument("--only", default=None, help="comma-separated name fragments: regenerate only those scenarios")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    _orig = (np.random.normal, np.random.randn, np.random.uniform, np.random.randint)

    scen = []
    for ti, task in enumerate(ENV_IDS):
        scen.append((f"{task}_det", task, dict(DET), 24, 12, act_random(0.3), 100 + ti, False, None, False))
        scen.append((f"{task}_det_long", task, dict(DET), 2, 500, act_random(0.05), 110 + ti, False, None, True))
        scen.append((f"{task}_hoveract", task, dict(DET, enable_reset_distribution=False), 1, 40, act_hover, 120 + ti, False, None, True))
        scen.append((f"{task}_dr", task, dict(DET, domain_randomization=0.1), 16, 10, act_random(0.3), 130 + ti, False, None, False))
        scen.append((f"{task}_motor", task, dict(DET), 12, 12, act_random(0.3), 140 + ti, True, None, False))
        scen.append((f"{task}_motor_dr", task, dict(DET, domain_randomization=0.1), 16, 12, act_random(0.3), 150 + ti, True, None, False))
        scen.append((f"{task}_defaults", task, dict(), 12, 10, act_random(0.2), 160 + ti, False, None, False))
        scen.append((f"{task}_noise_only", task, dict(domain_randomization=-1, motor_thrust_noise=0.05), 6, 10, act_random(0.2), 170 + ti, False, None, False))
        scen.append((f"{task}_agg2", task, dict(DET, aggregate_phy_steps=2) if task != "takeoff" else None, 6, 10, act_random(0.3), 180 + ti, False, None, False))
    # PID control modes on the Simple physics (envs/control.py:120-287)
    pid_act = lambda ep, t, rs, e: np.clip(0.5 * rs.standard_normal(4), -1.3, 1.3) * np.array([0.3, 1, 1, 1]) + np.array([-0.1, 0, 0, 0])
    scen.append(("hover_rate", "hover", dict(DET, control_mode="AttitudeRate"), 10, 10, pid_act, 200, False, None, False))
    scen.append(("hover_rate_agg4_dr", "hover", dict(DET, control_mode="AttitudeRate", aggregate_phy_steps=4, domain_randomization=0.1), 8, 8, pid_act, 201, False, None, False))
    scen.append(("hover_att_agg2", "hover", dict(DET, control_mode="Attitude", aggregate_phy_steps=2), 10, 10, pid_act, 202, False, None, False))
    scen.append(("circle_att_motor", "circle", dict(DET, control_mode="Attitude"), 8, 10, pid_act, 203, True, None, False))
    scen.append(("circle_rate_noise_only", "circle", dict(control_mode="AttitudeRate", domain_randomization=-1), 6, 8, pid_act, 204, False, None, False))
    scen.append(("hover_edge", "hover", dict(DET, enable_reset_distribution=False), 20, 3, act_random(0.2), 190, False, edge_hover, True))
    scen.append(("takeoff_edge", "takeoff", dict(DET, enable_reset_distribution=False), 3, 4, act_random(0.2, center=0.3), 191, False, edge_takeoff, True))
    scen.append(("circle_edge", "circle", dict(DET, enable_reset_distribution=False), 3, 4, act_random(0.2), 192, False, edge_circle, True))
    scen.append(("hover_bigact", "hover", dict(DET), 4, 6, lambda ep, t, rs, e: rs.uniform(-5, 5, 4), 193, False, None, True))
    # delayed actions through drone.action_buffer (envs/agents.py:179-183, 267-276, 384-404)
    lat = {}
    def lat_scen(name, task, kw, E, T, fn, seed, motor=False, set_latency=None, latency_on=True):
        scen.append((name, task, kw, E, T, fn, seed, motor, None, False))
        lat[name] = dict(latency_on=latency_on, set_latency=set_latency)
    lat_scen("hover_lat1", "hover", dict(DET), 10, 8, act_random(0.3), 300)                             # default 0.015 s: 1 row
    lat_scen("hover_lat2_motor", "hover", dict(DET, latency=0.025), 10, 8, act_random(0.3), 301, motor=True)
    lat_scen("circle_lat2_dr", "circle", dict(DET, latency=0.02, domain_randomization=0.1), 10, 8, act_random(0.3), 302)
    lat_scen("takeoff_lat3", "takeoff", dict(DET, latency=0.035), 6, 8, act_random(0.3, center=0.2), 303)
    lat_scen("hover_lat3_agg2", "hover", dict(DET, latency=0.035, aggregate_phy_steps=2), 8, 8, act_random(0.3), 304)
    lat_scen("hover_lat2_noresetdist", "hover", dict(DET, latency=0.025, enable_reset_distribution=False), 4, 8, act_random(0.2), 305)
    lat_scen("circle_rate_lat2", "circle", dict(DET, control_mode="AttitudeRate", latency=0.02), 8, 8, pid_act, 306)
    lat_scen("hover_setlat4", "hover", dict(DET), 8, 8, act_random(0.3), 307, set_latency=0.045, latency_on=False)  # int(.045/.01) = 4
    lat_scen("hover_lat2_defaults", "hover", dict(latency=0.025), 8, 8, act_random(0.2), 308)           # noise + DR + latency
    lat_scen("hover_lat2_floordiv", "hover", dict(DET, latency=0.03), 4, 6, act_random(0.3), 309)       # 0.03 // 0.01 == 2.0 (0.03 / 0.01 == 3.0)
    # observation_frequency != 100: obs_rate = 100 // f > 1 takes the Kalman-hold branch of compute_observation
    # (hover.py:150-156 and the circle / takeoff equivalents); Circle sizes its reference from it (circle.py:47-49)
    scen.append(("hover_obsf50", "hover", dict(observation_frequency=50, domain_randomization=-1), 8, 10, act_random(0.2), 320, False, None, False))
    scen.append(("hover_obsf50_agg2_defaults", "hover", dict(observation_frequency=50, aggregate_phy_steps=2), 8, 8, act_random(0.2), 321, False, None, False))
    scen.append(("circle_obsf50", "circle", dict(observation_frequency=50, domain_randomization=-1), 8, 10, act_random(0.2), 322, False, None, False))
    scen.append(("takeoff_obsf25", "takeoff", dict(observation_frequency=25, domain_randomization=-1), 6, 10, act_random(0.2, center=0.2), 323, False, None, False))
    scen.append(("circle_obsf50_det", "circle", dict(DET, observation_frequency=50), 8, 10, act_random(0.3), 324, False, None, False))
    scen.append(("hover_obsf33_agg3", "hover", dict(observation_frequency=33, aggregate_phy_steps=3, domain_randomization=-1), 6, 8, act_random(0.2), 325, False, None, False))
    # sensor noise with aggregate_phy_steps > 1 on CIRCLE: the reference point is indexed by iteration // aggregate_phy_steps
    # (circle.py:143-146), so a wrong env.step counter shows in target, reward and observation
    scen.append(("circle_noise_agg2", "circle", dict(aggregate_phy_steps=2, domain_randomization=-1), 6, 10, act_random(0.2), 326, False, None, False))
    scen.append(("circle_defaults_agg3", "circle", dict(aggregate_phy_steps=3), 6, 8, act_random(0.2), 327, False, None, False))

    # round 4: the Kalman-hold branch together with a PID control mode / the latency ring (refused until then)
    scen.append(("hover_obsf50_rate", "hover", dict(observation_frequency=50, domain_randomization=-1, control_mode="AttitudeRate"), 8, 10, pid_act, 328, False, None, False))
    scen.append(("circle_obsf50_att_agg2", "circle", dict(observation_frequency=50, control_mode="Attitude", aggregate_phy_steps=2), 6, 8, pid_act, 329, False, None, False))
    lat_scen("hover_obsf50_lat2", "hover", dict(observation_frequency=50, latency=0.025, domain_randomization=-1), 8, 8, act_random(0.2), 330)
    lat_scen("hover_obsf50_rate_lat2_defaults", "hover", dict(observation_frequency=50, latency=0.02, control_mode="AttitudeRate"), 6, 8, pid_act, 331)

    only = set(args.only.split(",")) if args.only else None
    index = {}
    if only is not None and os.path.exists(os.path.join(args.out, "INDEX.json")):
        index = json.load(open(os.path.join(args.out, "INDEX.json")))
    for (name, task, kw, E, T, fn, seed, motor, ov, full) in scen:
        if kw is None:
            continue
        if only is not None and not any(name.startswith(o) or o in name for o in only):
            continue
        arrays = run_scenario(name, task, kw, E, T, fn, seed, motor=motor, init_override=ov, full_episode=full,
                              **lat.get(name, {}))
        path = os.path.join(args.out, name + ".npz")
        np.savez_compressed(path, **arrays)
        index[name] = dict(task=task, episodes=E, steps=T, bytes=os.path.getsize(path))
        print(f"{name:24s} {os.path.getsize(path) / 1024:8.1f} kB")

    (np.random.normal, np.random.randn, np.random.uniform, np.random.randint) = _orig
    if only is None:
        np.savez_compressed(os.path.join(args.out, "ground_effect.npz"), **ground_effect_vectors())
        np.savez_compressed(os.path.join(args.out, "quaternion.npz"), **quaternion_vectors())
        with open(os.path.join(args.out, "constants.json"), "w") as f:
            json.dump(constants_vectors(), f, indent=1)
    if args.rate:
        rate = reference_cpu_rate()
        rate["_method"] = ("reference python step loop, 1 process, build container (8 cores visible), "
                           "PyBullet replaced by the no-op stand-in => upper bound; "
                           "oracle/refgen/gen_golden.py --rate")
        with open(os.path.join(args.out, "reference_cpu_rate.json"), "w") as f:
            json.dump(rate, f, indent=1)
        print(rate)
    with open(os.path.join(args.out, "INDEX.json"), "w") as f:
        json.dump(index, f, indent=1)


if __name__ == "__main__":
    main()
