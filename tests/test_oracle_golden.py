"""CPU oracle vs golden vectors produced by the reference itself.

The f64 build of the oracle must reproduce the reference (float64 numpy) to <= 1e-12 relative
(+1e-13 absolute for near-cancelling quantities); the f32 build -- what the HIP kernel is compared
with -- must stay within 1e-6 relative + small absolute single-step error.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as po
import golden_util as gu

SCENARIOS = gu.scenario_names()


def test_constants_match_reference():
    """G1: URDF scalars + derived constants (envs/agents.py:138-156)."""
    with open(os.path.join(gu.GOLDEN, "constants.json")) as f:
        ref = json.load(f)
    k = po.constants()
    for name, v in k.items():
        assert v == pytest.approx(ref[name], rel=1e-15, abs=0), name
    for task in ("hover", "circle", "takeoff"):
        for label, noise in (("noise_free", -1), ("noisy", 1)):
            env = po.OracleEnv(task, observation_noise=noise)
            assert env.obs_dim == ref["obs_dim"][task][label]
    e = po.OracleEnv("hover")
    assert e.get("A")[0] == pytest.approx(ref["A0"], rel=1e-15)
    assert e.get("B")[0] == pytest.approx(ref["B0"], rel=1e-15)
    assert e.get("K")[0] == pytest.approx(ref["K0"], rel=1e-15)


def test_reference_tables():
    """Circle / TakeOff reference trajectories (circle.py:45-56, takeoff.py:43-47) are recomputed
    analytically by the oracle's update_target."""
    with open(os.path.join(gu.GOLDEN, "constants.json")) as f:
        ref = json.load(f)
    circ = np.array(ref["circle_ref"])
    e = po.OracleEnv("circle", observation_noise=-1)
    import ctypes as C
    o = (C.c_double * 24)()
    for t in range(300):
        e.set("iteration", t)
        e.set("ref_offset", 0)
        e.L.po_compute_observation_f64(C.byref(e.cfg), C.byref(e.env), None, o)
        np.testing.assert_allclose(e.get("target_pos"), circ[t], rtol=0, atol=1e-16)
    zt = np.array(ref["takeoff_ref_z"])
    e = po.OracleEnv("takeoff", observation_noise=-1)
    for t in (0, 1, 17, 298, 299, 300, 450):
        e.set("iteration", t)
        e.L.po_compute_observation_f64(C.byref(e.cfg), C.byref(e.env), None, o)
        assert e.get("target_pos")[2] == zt[min(t, 299)]


def test_quaternion_convention():
    """pybullet boundary: order [x,y,z,w], ZYX (reference tests/test_quaternion.py:35-43 and
    envs/utils.py:32-56), plus an independent scipy cross-check of all three restated functions."""
    from scipy.spatial.transform import Rotation
    d = np.load(os.path.join(gu.GOLDEN, "quaternion.npz"))
    for rpy, q in zip(d["rpy"], d["quat"]):
        qo = po.quat_from_euler(rpy)
        np.testing.assert_allclose(qo, q, rtol=0, atol=3e-16)
        # round trip of the reference test: Euler(Q(rpy)) == rpy for |angles| < pi/2 in pitch
        if abs(rpy[1]) < 1.5:
            np.testing.assert_allclose(po.euler_from_quat(qo), rpy, rtol=0, atol=1e-13)
        rot = Rotation.from_euler("xyz", rpy)
        np.testing.assert_allclose(po.matrix_from_quat(qo), rot.as_matrix(), rtol=0, atol=1e-15)
        qs = rot.as_quat()
        qs = qs if np.dot(qs, qo) > 0 else -qs
        np.testing.assert_allclose(qo, qs, rtol=0, atol=1e-15)


def test_bullet_pose_readback_canonicalises_the_quaternion():
    """Row a17: getBasePositionAndOrientation answers through btTransform's 3x3 basis
    (btMatrix3x3::setRotation -> getRotation), so the quaternion update_information reads
    (envs/agents.py:443) is +-Q(sampled rpy) with Bullet's sign: w > 0 when the trace is positive
    (|w| > 1/2), otherwise the component of the largest diagonal element (largest of x^2, y^2, z^2) > 0."""
    from scipy.spatial.transform import Rotation
    rs = np.random.RandomState(3)
    n_flip = n_else = 0
    for i in range(4000):
        if i % 2:  # the Hover reset distribution: yaw in (-2 pi, 2 pi) (envs/hover.py:207-209)
            rpy = np.array([rs.uniform(-np.pi / 6, np.pi / 6), rs.uniform(-np.pi / 6, np.pi / 6), rs.uniform(-2 * np.pi, 2 * np.pi)])
            q = po.quat_from_euler(rpy)
        else:
            q = rs.standard_normal(4); q /= np.linalg.norm(q)
        r = po.bullet_readback_quat(q)
        same = np.dot(r, q) > 0
        np.testing.assert_allclose(r, q if same else -q, rtol=0, atol=2e-15)
        n_flip += not same
        # same rotation, and the sign rule stated on the quaternion itself
        np.testing.assert_allclose(po.matrix_from_quat(r), Rotation.from_quat(q).as_matrix(), rtol=0, atol=1e-15)
        if q[3] * q[3] > 0.25 + 1e-12:
            assert r[3] > 0
        elif q[3] * q[3] < 0.25 - 1e-12:
            n_else += 1
            assert r[int(np.argmax(np.abs(q[:3])))] > 0
        # idempotent up to rounding: what Bullet returns is what it would return again
        np.testing.assert_allclose(po.bullet_readback_quat(r), r, rtol=0, atol=2e-15)
        # f32 build: same branch, f32 accuracy
        np.testing.assert_allclose(po.bullet_readback_quat(q, "f32"), r, rtol=0, atol=3e-7)
    assert n_flip > 1000 and n_else > 1000


@pytest.mark.parametrize("name", ["hover_det", "hover_motor_dr", "takeoff_det", "takeoff_motor_dr", "circle_det"])
def test_golden_reset_quaternions_carry_bullets_sign(name):
    """The fixtures were generated through the read-back above: every recorded post-reset quaternion is the
    canonical one, and for Hover / TakeOff (yaw beyond +-2 pi / 3) a good part of them is -Q(wrapped rpy)."""
    g = gu.Golden(name)
    neg = 0
    for ep in range(g.E):
        q, rpy = g["reset_quat"][ep], g["reset_rpy"][ep]
        np.testing.assert_allclose(po.bullet_readback_quat(q), q, rtol=0, atol=2e-15)
        qw = po.quat_from_euler(rpy)  # what step_forward will write next (envs/physics.py:179)
        assert abs(abs(np.dot(qw, q)) - 1) < 1e-12
        neg += np.dot(qw, q) < 0
        if abs(rpy[2]) < 2 * np.pi / 3 - 0.2:
            assert np.dot(qw, q) > 0
        # the first observation carries it (noise-free scenarios: o = [p3, quat4, ...])
        np.testing.assert_array_equal(g["reset_obs"][ep][3:7], q)
    if name.startswith("circle"):
        assert neg == 0  # |yaw| <= 0.1 pi (envs/circle.py:234)
    elif g.E >= 8:
        assert neg > 0


def test_ground_effect_formula():
    """G7: BasePhysics.calculate_ground_effect (envs/physics.py:27-58)."""
    d = np.load(os.path.join(gu.GOLDEN, "ground_effect.npz"))
    import ctypes as C
    e = po.OracleEnv("takeoff", observation_noise=-1)
    for xyz, rpy, f, ge in zip(d["xyz"], d["rpy"], d["forces"], d["ge"]):
        e.set("xyz", xyz)
        e.set("rpy", rpy)
        e.set("quat", po.quat_from_euler(rpy))
        out = (C.c_double * 4)()
        e.L.po_ground_effect_f64(C.byref(e.env), (C.c_double * 4)(*f), out)
        np.testing.assert_allclose(np.array(out), ge, rtol=1e-12, atol=1e-18)


def _replay(name, precision, rtol, atol, resync):
    g = gu.Golden(name)
    env = gu.make_oracle(g, precision)
    assert env.obs_dim == g.D
    for ep in range(g.E):
        obs = gu.begin_episode(env, g, ep)
        gu.assert_close(obs, g["reset_obs"][ep], rtol, atol, f"{name} ep{ep} reset obs")
        for k in gu.DYN_FIELDS + ["x", "last_action", "dt", "m", "J", "ftf0", "ftf1", "A", "B", "K", "lpf", "gyro_bias"]:
            gu.assert_close(env.get(k), g["reset_" + k][ep], rtol, atol, f"{name} ep{ep} reset {k}")
        assert env.get("ref_offset") == g["reset_ref_offset"][ep]
        for t in range(g.n_valid(ep)):
            if resync and t > 0:
                # single-step parity: restart every step from the reference's recorded state
                for k in gu.DYN_FIELDS + ["x", "last_action", "env_last_action", "act_hist", "obs_hist", "ou", "gyro_bias", "lpf", "kf_state",
                          "rate_int", "rate_err", "att_int", "att_err", "action_buffer", "action_idx", "hist_alias",
                          "last_action_alias"]:
                    env.set(k, g["step_" + k][ep, t - 1])
            obs, r, term, trunc, cost = env.step(g["actions"][ep, t])
            w = f"{name} ep{ep} t{t}"
            gu.assert_close(obs, g["obs"][ep, t], rtol, atol, w + " obs")
            gu.assert_close(r, g["reward"][ep, t], rtol, atol * 10, w + " reward")
            for k in gu.DYN_FIELDS + ["x", "ou", "gyro_bias", "lpf", "rate_int", "att_int", "action_buffer", "act_hist"]:
                if k == "act_hist":  # what the reference's deque shows, views of action_buffer[-1] resolved
                    B = int(env.get("buf_size"))
                    got = np.array([env.get("action_buffer")[B - 1] if env.get("hist_alias")[h] else env.get("act_hist")[h]
                                    for h in range(2)])
                    gu.assert_close(got, g["step_act_hist"][ep, t], rtol, atol, w + " act_hist")
                    continue
                gu.assert_close(env.get(k), g["step_" + k][ep, t], rtol, atol, w + " " + k)
            assert int(env.get("action_idx")) == int(g["step_action_idx"][ep, t]), w
            assert [int(v) for v in env.get("hist_alias")] == [int(v) for v in g["step_hist_alias"][ep, t]], w
            assert term == bool(g["terminated"][ep, t]), w
            assert trunc == bool(g["truncated"][ep, t]), w
            assert cost == g["cost"][ep, t], w
            assert env.get("iteration") == g["step_iteration"][ep, t]


@pytest.mark.parametrize("name", SCENARIOS)
def test_oracle_f64_matches_reference(name):
    """Whole recorded trajectories (no re-synchronisation): float64 oracle == reference."""
    _replay(name, "f64", rtol=1e-12, atol=1e-13, resync=False)


@pytest.mark.parametrize("name", SCENARIOS)
def test_oracle_f32_single_step(name):
    """float32 oracle, re-synchronised to the reference state before every step: 1e-6 relative
    (north_star tolerance) + 2e-6 absolute for near-cancelling terms (yaw torque, tau_x/tau_y at
    equal thrusts integrate into rpy_dot with 1/J ~ 6e4 gain)."""
    rtol, atol = gu.tolerances(name)
    _replay(name, "f32", rtol=rtol, atol=atol, resync=True)


# Random123 known-answer vectors (kat_vectors of the Random123 distribution, D. E. Shaw Research):
# philox4x32 <rounds> <counter x4> <key x2> -> <output x4>
PHILOX_KAT = [
    (10, [0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    (10, [0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    (10, [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    (7, [0, 0, 0, 0], [0, 0], [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]),
    (7, [0xffffffff] * 4, [0xffffffff] * 2, [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]),
    (7, [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]),
]


def test_oracle_philox_known_answers():
    """The counter-based generator of the RNG contract (reset sampling: 10 rounds, per-step noise: 7) is
    the published Philox4x32: the oracle's restatement reproduces the Random123 known-answer vectors."""
    import ctypes as C
    L = po.lib()
    for rounds, ctr, key, want in PHILOX_KAT:
        out = (C.c_uint32 * 4)()
        L.po_philox4x32((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), rounds, out)
        assert [int(v) for v in out] == want, (rounds, ctr)
        if rounds == 10:
            assert po.philox4x32_10(ctr, key) == want
