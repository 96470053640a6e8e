"""Policy JSON interchange: a trained policy shipped with the reference
(experiments/07_control_structure_hypothesis/models/PWM/PWM_seed_00000_model.json, kept as a DATA
fixture) loads, passes its own check_sum, round-trips through the exporter, and -- on the GPU -- flies
the batched Circle env far longer than random actions do."""
import json
import os

import numpy as np
import pytest
import torch

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "policy_PWM_seed_00000_model.json")


def test_reference_policy_loads_and_checksum_holds():
    from phoenix_drone_simulation_amd.policy_io import load_network_json
    pol = load_network_json(FIX)
    assert pol.mean.shape == (40,) and pol.activation == "relu"
    sizes = [(l.in_features, l.out_features) for l in pol.net if isinstance(l, torch.nn.Linear)]
    assert sizes == [(40, 50), (50, 50), (50, 4)]
    data = json.load(open(FIX))
    assert float(pol.net(torch.ones(40)).sum()) == pytest.approx(float(data["check_sum"]), rel=1e-5)
    bad = dict(data, check_sum=float(data["check_sum"]) + 1.0)
    p = os.path.join(os.path.dirname(FIX), "_bad_tmp.json")
    try:
        json.dump(bad, open(p, "w"))
        with pytest.raises(ValueError):
            load_network_json(p)
    finally:
        os.remove(p)


def test_export_roundtrip(tmp_path):
    from phoenix_drone_simulation_amd.policy_io import load_network_json, dump_json, convert_actor_critic_to_json
    from phoenix_drone_simulation_amd.ppo import ActorCritic
    pol = load_network_json(FIX)
    sp = np.stack([pol.mean.numpy(), pol.std.numpy()])
    dump_json("relu", sp, pol.net, str(tmp_path / "m.json"))
    again = load_network_json(str(tmp_path / "m.json"))
    x = torch.randn(16, 40)
    assert torch.allclose(pol(x), again(x), atol=1e-6)
    ref = json.load(open(FIX)); new = json.load(open(tmp_path / "m.json"))
    assert set(ref) == set(new) and new["0"]["type"] == "standard"
    ac = ActorCritic(42, 4)
    ac.obs_oms.update(torch.randn(100, 42) * 2 + 1)
    convert_actor_critic_to_json(ac, str(tmp_path / "ac.json"))
    p2 = load_network_json(str(tmp_path / "ac.json"))
    ac.eval()
    obs = torch.randn(8, 42)
    assert torch.allclose(p2(obs), ac.step(obs)[0], atol=1e-5)


@pytest.mark.gpu
def test_reference_policy_flies_the_batched_circle_env():
    import phoenix_drone_simulation_amd as pds
    from phoenix_drone_simulation_amd.policy_io import load_network_json
    n = 4096
    kw = dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0)
    lens = {}
    for name in ("policy", "random"):
        env = pds.make("DroneCircleSimpleEnv-v0", num_envs=n, seed=3, **kw)
        pol = load_network_json(FIX).to(env.device)
        obs, _ = env.reset()
        steps = torch.zeros(n, device=env.device); total, count = 0.0, 0.0
        for t in range(300):
            a = pol(obs) if name == "policy" else (-0.11 + 0.1 * torch.randn(n, 4, device=env.device))
            obs, r, term, trunc, info = env.step(a.contiguous())
            steps += 1
            done = term | trunc
            total += float((steps * done).sum()); count += float(done.sum())
            steps = torch.where(done, torch.zeros_like(steps), steps)
        total += float(steps.sum()); count += n
        lens[name] = total / count
        env.close()
    assert lens["policy"] > 3 * lens["random"], lens


def test_bundled_checkpoint_exports_to_the_bundled_json(tmp_path):
    """The reference ships both the training checkpoint (checkpoints/PWM/.../seed_00000/torch_save/
    model.pt; tensors in tests/golden/ckpt_PWM_seed_00000.npz) and the firmware JSON exported from
    it (models/PWM/PWM_seed_00000_model.json).  Loading the former into our ActorCritic and
    exporting it must reproduce the latter: weights, scaling parameters and check_sum."""
    from phoenix_drone_simulation_amd.policy_io import convert_actor_critic_to_json, load_network_json
    from phoenix_drone_simulation_amd.ppo import ActorCritic
    ck = np.load(os.path.join(os.path.dirname(FIX), "ckpt_PWM_seed_00000.npz"))
    ac = ActorCritic.from_reference_state_dict({k: ck[k] for k in ck.files})
    assert ac.pi.net[0].in_features == 40 and ac.pi.net[4].out_features == 4
    out = tmp_path / "export.json"
    mine = convert_actor_critic_to_json(ac, str(out))
    with open(FIX) as f:
        ref = json.load(f)
    assert mine["activation"] == ref["activation"]
    np.testing.assert_allclose(np.array(mine["scaling_parameters"]), np.array(ref["scaling_parameters"]), rtol=1e-6, atol=1e-7)
    k = 0
    while str(k) in ref:
        assert ref[str(k)]["type"] == mine[str(k)]["type"] == "standard"
        np.testing.assert_allclose(np.array(mine[str(k)]["weights"]), np.array(ref[str(k)]["weights"]), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(np.array(mine[str(k)]["biases"]).reshape(-1), np.array(ref[str(k)]["biases"]).reshape(-1), rtol=1e-6, atol=1e-7)
        k += 1
    assert str(k) not in mine and k == 3
    assert abs(float(mine["check_sum"]) - float(ref["check_sum"])) < 1e-4
    # and the exported file drives the same actions as the bundled one
    a, b = load_network_json(str(out)), load_network_json(FIX)
    x = torch.randn(64, 40)
    assert torch.allclose(a(x), b(x), atol=1e-6)
    # deterministic (eval) ActorCritic.step == the JSON policy
    ac.eval()
    assert torch.allclose(ac.step(x)[0], b(x), atol=1e-5)


@pytest.mark.gpu
def test_batched_evaluator_matches_sequential_oracle_episodes(tmp_path):
    """evaluation.evaluate (EnvironmentEvaluator, utils/evaluation.py:15-117) with a bundled reference
    policy on the Circle env: every env plays one 500-step episode; the returns agree statistically
    with the same policy flying the f64 oracle (closed-loop fp32 drift: see
    test_closed_loop_with_reference_policy_500_steps), returns.csv has one value per line."""
    import phoenix_drone_simulation_amd as pds
    from oracle import oracle as po
    from phoenix_drone_simulation_amd.evaluation import evaluate
    from phoenix_drone_simulation_amd.policy_io import load_network_json
    n, seed = 128, 13
    base = dict(observation_noise=-1, domain_randomization=-1, motor_thrust_noise=0)
    env = pds.make("DroneCircleSimpleEnv-v0", num_envs=n, seed=seed, **base)
    pol = load_network_json(FIX).to(env.device)
    ret, length, cost = evaluate(env, pol, log_dir=str(tmp_path))
    assert ret.shape == (n,) and bool((length == 500).all()) and float(cost.sum()) == 0.0
    lines = open(tmp_path / "returns.csv").read().split()
    assert len(lines) == n and abs(float(lines[3]) - float(ret[3])) < 1e-4
    orc = po.OracleBatch("circle", n, precision="f64", **base)
    oobs = orc.reset(seed, 0)
    oret = np.zeros(n)
    for t in range(500):
        a = pol(torch.tensor(oobs, dtype=torch.float32, device=env.device)).cpu().numpy().astype(np.float32)
        oobs, r, te, tr, _ = orc.step(a, seed=seed, tick=1 + t, auto_reset=True)
        oret += r
    assert abs(float(ret.mean()) - oret.mean()) < 0.01 * abs(oret.mean())
    env.close()


@pytest.mark.gpu
def test_get_batch_pairs_observations_like_the_trajectory_generator():
    import phoenix_drone_simulation_amd as pds
    from phoenix_drone_simulation_amd.evaluation import get_batch
    from phoenix_drone_simulation_amd.policy_io import load_network_json
    n, steps = 64, 30
    env = pds.make("DroneCircleSimpleEnv-v0", num_envs=n, seed=3, max_episode_steps=12, observation_noise=-1,
                   domain_randomization=-1, motor_thrust_noise=0)
    pol = load_network_json(FIX).to(env.device)
    X, Y = get_batch(env, pol, steps)
    assert X.shape == Y.shape == (steps, n, 40)
    raw = X * (pol.std + pol.eps) + pol.mean  # un-standardise
    # history layout: the older half of the next input is the newer half of the current output
    # wherever the env did not finish (steps 11, 23 are TimeLimit truncations for everyone)
    for t in range(steps - 1):
        if (t + 1) % 12 == 0:
            assert not torch.allclose(raw[t + 1], Y[t], atol=1e-4)  # reset observation follows the terminal one
        else:
            assert torch.allclose(raw[t + 1], Y[t], atol=1e-4)
    env.close()
