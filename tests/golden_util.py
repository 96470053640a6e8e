"""Helpers shared by the parity tests: load tests/golden/*.npz (vectors generated from the
reference by oracle/refgen/gen_golden.py) and replay them through the CPU oracle."""
import json
import os

import numpy as np

from oracle import oracle as po

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SAMPLE_FIELDS = ["pos_offset", "rpy", "vel", "omega", "motor_x", "action", "dr_dt", "dr_m", "dr_J",
                 "dr_ftf0", "dr_ftf1", "dr_T", "dr_t2w", "ref_offset", "action_buf"]
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
        self.latency_on = bool(self.meta.get("latency_on", False))  # drone.use_latency flipped after construction
        self.set_latency = self.meta.get("set_latency")             # drone.set_latency(x) called after construction
        self.E, self.T, self.D = self.meta["episodes"], self.meta["steps"], self.meta["obs_dim"]

    def __getitem__(self, k):
        return self.d[k]

    def oracle_kwargs(self):
        kw = dict(self.kwargs)
        kw["use_motor_dynamics"] = 1 if self.motor else 0
        kw["use_latency"] = 1 if self.latency_on else 0
        # control_mode is passed through as a string (oracle.default_config maps it)
        if "enable_reset_distribution" in kw:
            kw["enable_reset_distribution"] = int(kw["enable_reset_distribution"])
        return kw

    def sample(self, ep):
        return {k: self.d["sample_" + k][ep] for k in SAMPLE_FIELDS}

    def n_valid(self, ep):
        return int(self.d["valid"][ep].sum())


def make_oracle(g, precision="f64"):
    env = po.OracleEnv(g.task, precision=precision, **g.oracle_kwargs())
    if g.set_latency is not None:
        env.set_latency(g.set_latency)
    return env


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


# ---- stochastic scenarios: map the reference's recorded numpy streams onto the kernel's layouts ----
# Per add_noise call the reference draws z: pos3 vel3 bias3 rw3 turn_on3 theta3 acc6 (24) and
# u: pos3 vel3 theta3 (9)  (envs/sensors.py:75-134); per env.step(): OU randn(4) + two calls
# (envs/base.py:461-468); per reset(): two calls (envs/base.py:419,429).
Z_CALL, U_CALL = 24, 9


def _obs_call_variates(z, u):
    """24 kernel-order variates (PDS_N_OBS_*) of one add_noise call from its numpy-order draws."""
    out = np.zeros(24)
    out[0:3] = z[0:3]      # pos z
    out[3:6] = u[0:3]      # pos u
    out[6:9] = z[3:6]      # vel z
    out[9:12] = z[6:9]     # gyro bias z
    out[12:15] = z[9:12]   # random-walk z
    out[15:18] = z[12:15]  # turn-on z
    out[18:21] = z[15:18]  # theta z
    out[21:24] = u[6:9]    # theta u
    return out


def noisy(g):
    return g.kwargs.get("observation_noise", 1) > 0


def episode_streams(g, ep):
    z0, z1 = int(g["z_off"][ep]) + int(g["z_skip"][ep]), int(g["z_off"][ep + 1])
    u0, u1 = int(g["u_off"][ep]) + int(g["u_skip"][ep]), int(g["u_off"][ep + 1])
    return g["z"][z0:z1], g["u"][u0:u1]


def reset_noise_variates(g, ep):
    """[48]: the two add_noise calls of reset() (PDS_S_NOISE_CALL0/1)."""
    z, u = episode_streams(g, ep)
    if not noisy(g):
        return np.zeros(48)
    return np.concatenate([_obs_call_variates(z[0:24], u[0:9]), _obs_call_variates(z[24:48], u[9:18])])


def obs_rate(g):
    """sim_freq // observation_frequency (envs/base.py:108; sim_freq = 100 on the Simple envs)."""
    return int(100 // int(g.kwargs.get("observation_frequency", 100)))


def _agg(g):
    return int(g.kwargs.get("aggregate_phy_steps", 1))


def _stream_offsets(g, t):
    """(z, u) offsets of env.step() number t inside the episode's recorded streams.  One env.step() draws, per
    physics sub-step (envs/base.py:457-465), OUNoise's randn(4) and one add_noise call at iteration
    t * A + sub, then the observing call at iteration (t + 1) * A (base.py:468); a call at an iteration that
    is not a multiple of obs_rate only draws the 9 gyro normals (envs/hover.py:134-156)."""
    R, A = obs_rate(g), _agg(g)
    zs, us = 2 * Z_CALL, 2 * U_CALL  # reset(): two full calls at iteration 0
    for k in range(t):
        for it in list(range(k * A, (k + 1) * A)) + [(k + 1) * A]:
            full = (it % R) == 0
            zs += Z_CALL if full else 9
            us += U_CALL if full else 0
        zs += 4 * A  # OUNoise, once per sub-step
    return zs, us


def step_noise_variates(g, ep, t):
    """[A * 52], A = aggregate_phy_steps: per physics sub-step a block of OU z4 | gyro part of that sub-step's
    (discarded) add_noise call (bias3 rw3 to3) | [block 0 only: the observing call (24)] | position / velocity /
    angle draws of the discarded call (pos_z3 pos_u3 vel_z3 th_z3 th_u3; they matter when obs_rate > 1)."""
    z, u = episode_streams(g, ep)
    R, A = obs_rate(g), _agg(g)
    out = np.zeros((A, 52))
    if noisy(g):
        zs, us = _stream_offsets(g, t)
        for sub in range(A):
            o = out[sub]
            o[0:4] = z[zs:zs + 4]
            zs += 4
            if ((t * A + sub) % R) == 0:
                za, ua = z[zs:zs + Z_CALL], u[us:us + U_CALL]
                o[4:13] = za[6:15]
                o[37:40], o[40:43], o[43:46] = za[0:3], ua[0:3], za[3:6]
                o[46:49], o[49:52] = za[15:18], ua[6:9]
                zs += Z_CALL; us += U_CALL
            else:
                o[4:13] = z[zs:zs + 9]
                zs += 9
        if (((t + 1) * A) % R) == 0:
            out[0, 13:37] = _obs_call_variates(z[zs:zs + Z_CALL], u[us:us + U_CALL])
        else:
            out[0, 13 + 9:13 + 18] = z[zs:zs + 9]  # bias, random walk, turn-on of add_noise_to_omega
    else:
        for sub in range(A):
            out[sub, 0:4] = z[4 * (A * t + sub):4 * (A * t + sub) + 4]
    return out.reshape(-1)


def tolerances(name):
    """(rtol, atol) of the float32 CPU ORACLE's single-step bar (tests/test_oracle_golden.py; the oracle's f32 build
    uses libm, not the kernel's arithmetic).  The HIP kernel is held to the per-field bars below."""
    if any(t in name for t in ("_rate", "_att")):
        return 1e-5, 1e-5
    return 1e-6, 2e-6


# ---- single-step bars of the HIP kernel (1e-6 relative + 1e-7 absolute) ---------------------------------
# Every field is held to RTOL = 1e-6 relative + 1e-7 absolute, with three stated exceptions, each a quantity whose
# float32 INPUTS already carry more than 1e-7 (measured margins: profiles/r04_parity_margins.txt):
#   * body rates (rpy_dot and the observation columns that carry it): + 2e-6 absolute -- near-cancelling torques
#     (equal thrusts) are amplified by dt / J ~ 600; 5e-6 in the PID control modes, whose 1/dt derivative and gains up to
#     250 (envs/control.py:166-176, 268-277) amplify the float32 rounding of the attitude; 4e-6 for the NOISY filtered
#     gyro, a sum of O(3 rad/s) terms (R^T R^T omega, turn-on bias, random walk, stale low-pass state);
#   * quaternion: + 0.5e-6 x (|roll| + |pitch| + |yaw|) -- it is derived from the Euler angles (physics.py:179), which
#     are themselves only good to 1e-6 RELATIVE (yaw is never wrapped: at 30 rad one float32 ulp is 1.9e-6), and
#     d q / d angle <= 1/2;
#   * target - position (Circle / TakeOff observation): + 1e-6 x |position| -- a difference of two O(1) numbers of
#     which the position carries the 1e-6 relative bar.
RTOL = 1e-6
ATOL = 1e-7


def is_pid(name):
    return any(t in name for t in ("_rate", "_att"))


def rate_atol(name, noisy_gyro=False):
    if noisy_gyro:
        return 4e-6
    return 5e-6 if is_pid(name) else 2e-6


def quat_atol(rpy):
    """[B, 3] Euler angles the quaternion is derived from -> [B, 1] absolute bar."""
    return ATOL + 0.5e-6 * np.abs(np.asarray(rpy, dtype=np.float64)).sum(-1, keepdims=True)


def obs_groups(task, D, noisy_obs):
    """column -> content of one observation row (two history halves; envs/hover.py:131-163, circle.py:128-177,
    takeoff.py:107-149, base.py:303-319): p position, q quaternion, v velocity, w body rates, a last action,
    e target - position, u paired action of the history."""
    if noisy_obs and task == "hover":
        half = ["p"] * 3 + ["q"] * 4 + ["v"] * 3 + ["w"] * 3
    elif task == "hover":
        half = ["p"] * 3 + ["q"] * 4 + ["v"] * 3 + ["w"] * 3 + ["a"] * 4
    elif task == "circle":
        half = ["p"] * 3 + ["q"] * 4 + ["v"] * 3 + ["w"] * 3 + ["e"] * 3
    else:
        half = ["p"] * 3 + ["q"] * 4 + ["v"] * 3 + ["w"] * 3 + ["a"] * 4 + ["e"] * 3
    half = half + ["u"] * 4
    assert 2 * len(half) == D, (task, D, noisy_obs, len(half))
    return half + half


def obs_atol(g, name, want_obs, rpy_first, rpy_second, noisy_obs=False):
    """Per-element absolute bar of an observation row [B, D]; rpy_first / rpy_second: the Euler angles behind the
    quaternion of the first / second history half (o(k) and o(k+1))."""
    want = np.asarray(want_obs, dtype=np.float64)
    B, D = want.shape
    groups = obs_groups(g.task, D, noisy_obs)
    half = D // 2
    atol = np.full((B, D), ATOL)
    qa = (quat_atol(rpy_first)[:, 0], quat_atol(rpy_second)[:, 0])
    e_seen = [0, 0]
    for c, grp in enumerate(groups):
        h = 0 if c < half else 1
        if grp == "q":
            atol[:, c] = qa[h]
        elif grp == "w":
            atol[:, c] = rate_atol(name, noisy_obs)
        elif grp == "e":
            atol[:, c] = ATOL + 1e-6 * np.abs(want[:, h * half + e_seen[h]])  # |position component| of the same half
            e_seen[h] += 1
    return atol


# ---- learning curves: two-sample comparison (tests/test_trainer.py) ----------------------------------------------------------
LC_PHASES = {"early (epochs 4-8)": slice(3, 8), "first peak (9-16)": slice(8, 16), "dip (17-23)": slice(16, 23),
             "late (24-40)": slice(23, 40)}


def compare_learning_curves(a, b, p_min=0.01):
    """Two samples of per-epoch curves, `a` [seeds_a, epochs] and `b` [seeds_b, epochs] (EpLen/Mean or EpRet/Mean of independent
    training runs): are they draws from one distribution of runs?  Returns (list of failures, report dict).

    Epochs WITHIN a run are strongly correlated -- a run's level in epochs 24-40 is a persistent trait of its seed (reference:
    63 .. 113 steps over 12 seeds, SD 13) -- so the unit of every test is the SEED:
      * per phase (LC_PHASES): the per-seed mean over the phase's epochs, Welch's t-test between the two samples, p > p_min;
        this is the test that sees a one-sided offset: a shift of the late level by more than ~2.8 standard errors of the
        difference fails it;
      * per epoch: Welch's t-test with a Bonferroni factor of the number of epochs, min over epochs of p x epochs > p_min.
    A sign test over the epochs (how many of epochs 20-40 have mean(a) above mean(b)) is REPORTED, not asserted: applied to
    the reference pool split into two halves of 6 seeds it "fails" (more than 14 of 21 on one side) for 770 of the 924
    splits -- consecutive epochs of a run are one observation, not 21."""
    import numpy as np
    from scipy import stats
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    fails, report = [], {}
    for name, sl in LC_PHASES.items():
        x, y = a[:, sl].mean(axis=1), b[:, sl].mean(axis=1)
        t, p = stats.ttest_ind(x, y, equal_var=False)
        report[name] = dict(mean_a=float(x.mean()), mean_b=float(y.mean()), t=float(t), p=float(p))
        if not p > p_min:
            fails.append((name, float(x.mean()), float(y.mean()), float(t), float(p)))
    t, p = stats.ttest_ind(a, b, axis=0, equal_var=False)
    e = int(np.argmin(p))
    report["per epoch"] = dict(worst_epoch=e + 1, t=float(t[e]), p_bonferroni=float(min(1.0, p[e] * a.shape[1])))
    if not p[e] * a.shape[1] > p_min:
        fails.append(("epoch %d" % (e + 1), float(a[:, e].mean()), float(b[:, e].mean()), float(t[e]), float(p[e] * a.shape[1])))
    d = a[:, 19:40].mean(axis=0) - b[:, 19:40].mean(axis=0)
    report["sign count epochs 20-40 (reported only)"] = (int((d > 0).sum()), int((d < 0).sum()))
    return fails, report


# ---- one full update() of the reference's trainer (tests/golden/update.npz, oracle/refgen/gen_golden_update.py) --------------
def gae_torch(rew, val, terminated, truncated, final_val, last_val, gamma, lam, rew_scale=0.0, rew_clip=10.0):
    """csrc/pds_gae.hip restated with torch ops on [T, N] tensors of any device (test infrastructure: the CPU test of the
    trainer's update runs without the HIP library).  -> [adv, target_v, discounted_ret]"""
    import torch
    T, N = rew.shape
    adv, tv, dr = torch.empty_like(rew), torch.empty_like(rew), torch.empty_like(rew)
    zero = torch.zeros(N, dtype=rew.dtype, device=rew.device)
    next_val, next_ret, next_adv = last_val.clone(), last_val.clone(), zero.clone()
    gl = gamma * lam
    for t in range(T - 1, -1, -1):
        te, tr = terminated[t].bool(), truncated[t].bool()
        end = te | tr
        b = torch.where(tr, final_val[t] if final_val is not None else zero, zero)  # (cut wins: iwpg.py:374-379)
        next_val, next_ret, next_adv = torch.where(end, b, next_val), torch.where(end, b, next_ret), torch.where(end, zero, next_adv)
        rs = torch.clamp(rew[t] * rew_scale, -rew_clip, rew_clip) if rew_scale > 0 else rew[t]
        a = (rs + gamma * next_val - val[t]) + gl * next_adv
        g = rew[t] + gamma * next_ret
        adv[t], tv[t], dr[t] = a, a + val[t], g
        next_val, next_adv, next_ret = val[t], a, g
    return [adv, tv, dr]


def load_update_epoch(g, e, device):
    """The reference's rollout of epoch `e` in PPOTrainer's [T, N = 1] buffer layout: a path that ended by termination
    bootstraps with 0 (term = 1), one the epoch end or the TimeLimit cut with the recorded V (trunc = 1, fval = V; the last
    path: last_val).  -> dict of tensors."""
    import numpy as np
    import torch
    T = int(g["steps"])
    term = g[f"e{e}_terminated"].astype(np.uint8).copy()
    trunc = np.zeros(T, np.uint8)
    fval = np.zeros(T, np.float32)
    ends, lasts = g[f"e{e}_path_end"], g[f"e{e}_path_last_val"]
    last_val = 0.0
    for end, lv in zip(ends, lasts):
        t = int(end) - 1
        if end == T and not term[t]:
            last_val = float(lv)          # epoch end: PPOTrainer's last_val (no flag on the last step)
        elif term[t] and lv == 0.0:
            pass                           # terminated: bootstrap 0
        else:                              # the TimeLimit cut (also terminated AND cut: the reference takes V, iwpg.py:374-379)
            trunc[t], fval[t] = 1, lv
    f = lambda x, dt=torch.float32: torch.as_tensor(np.asarray(x), dtype=dt, device=device)  # noqa: E731
    return dict(obs=f(g[f"e{e}_obs_buf"])[:, None], act=f(g[f"e{e}_act_buf"])[:, None], rew=f(g[f"e{e}_rew_buf"])[:, None],
                val=f(g[f"e{e}_val_buf"])[:, None], logp=f(g[f"e{e}_logp_buf"])[:, None], term=f(term, torch.uint8)[:, None],
                trunc=f(trunc, torch.uint8)[:, None], fval=f(fval)[:, None], last_val=f([last_val]),
                adv=f(g[f"e{e}_adv_buf"]), target_v=f(g[f"e{e}_target_val_buf"]), disc_ret=f(g[f"e{e}_discounted_ret_buf"]),
                shuffles=[torch.as_tensor(p_, dtype=torch.int64, device=device) for p_ in g[f"e{e}_shuffles"]])
