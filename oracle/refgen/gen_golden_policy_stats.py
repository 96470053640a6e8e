#!/usr/bin/env python3
"""Sim-to-sim transfer check (build container only): two policies TRAINED ON THE HIP ENVS (tests/golden/hip_policy_{early,late}.npz,
the reference's ActorCritic state_dict layout; profiles/tools/train_export_policies.py) are loaded into the REFERENCE's own
`core.ActorCritic` (algs/core.py:313-412) and evaluated in the REFERENCE's own DroneHoverSimpleEnv-v0 (env defaults: sensor noise,
thrust noise, 10 % domain randomisation, reset distribution) the way `EnvironmentEvaluator.eval_once` does (utils/evaluation.py:
reset, act deterministically until terminated or truncated): per-episode length and return for EPISODES episodes each.
Only data is written: tests/golden/policy_eval_stats.json.  tests/test_gpu_noise.py plays the same policies in the HIP envs.
Stand-ins as for the other generators (pybullet*, gymnasium, mpi4py, tensorboard)."""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "standins"))
sys.path.insert(0, "/root/reference")
_tb = types.ModuleType("torch.utils.tensorboard")
_tb.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = _tb

import gymnasium as gym  # noqa: E402  (stand-in)
import phoenix_drone_simulation  # noqa: E402,F401
import phoenix_drone_simulation.algs.core as core  # noqa: E402

ENV_ID, EPISODES = "DroneHoverSimpleEnv-v0", {"early": 5000, "late": 600}
GOLD = os.path.join(HERE, "..", "..", "tests", "golden")


def _eval_actor_critic(args):
    """(numpy seed, episodes, env id, env kwargs, checkpoint file) -> (lengths, returns, terminated): one env instance, the
    policy played deterministically (EnvironmentEvaluator.eval_once, utils/evaluation.py:40-75)."""
    seed, episodes, env_id, kw, ckpt = args
    torch.set_num_threads(1)
    np.random.seed(seed)
    # `use_motor_dynamics` / `use_latency` are not reachable through the reference's env kwargs (CrazyFlieSimpleAgent is built
    # with both off, envs/agents.py:485-495; a kwarg of that name is swallowed): they are flipped on the agent the way the
    # reference's own debug script does (debug/compare_system_equations_with_PyBullet.py:26-30), as oracle/refgen/gen_golden.py
    # does for the golden scenarios; the ring keeps its ctor size max(1, int(latency // time_step)) (agents.py:180)
    flips = [k for k in ("use_motor_dynamics", "use_latency") if kw.get(k)]
    env = gym.make(env_id, **{k: v for k, v in kw.items() if k not in ("use_motor_dynamics", "use_latency")})
    for k in flips:
        setattr(env.unwrapped.drone, k, True)
    ac = core.ActorCritic('mlp', env.observation_space, env.action_space, use_standardized_obs=True, use_scaled_rewards=True,
                          use_shared_weights=False,
                          ac_kwargs={'pi': {'hidden_sizes': (50, 50), 'activation': 'relu'},
                                     'val': {'hidden_sizes': (64, 64), 'activation': 'tanh'}})
    sd = np.load(os.path.join(GOLD, ckpt))
    ac.load_state_dict({k: torch.as_tensor(sd[k]) for k in sd.files}, strict=True)
    ac.eval()
    lens, rets, terms = [], [], []
    for ep in range(episodes):
        o, _ = env.reset()
        n, ret = 0, 0.0
        while True:
            a, _, _ = ac.step(torch.as_tensor(o, dtype=torch.float32))
            o, r, te, tr, _ = env.step(a)
            n += 1; ret += float(r)
            if te or tr or n >= 500:
                break
        lens.append(n); rets.append(ret); terms.append(bool(te))
    return lens, rets, terms


def _eval_constant_action(args):
    """(numpy seed, episodes, env id, env kwargs, offset) -> (lengths, returns): a = HOVER_ACTION + offset on all four motors"""
    seed, episodes, env_id, kw, offset = args
    np.random.seed(seed)
    env = gym.make(env_id, **kw)
    a = np.full(4, env.unwrapped.drone.HOVER_ACTION + offset)
    lens, rets = [], []
    for ep in range(episodes):
        env.reset()
        n, ret = 0, 0.0
        while True:
            o, r, te, tr, _ = env.step(a)
            n += 1; ret += float(r)
            if te or tr or n >= 500:
                break
        lens.append(n); rets.append(ret)
    return lens, rets


def _circle_reference_policy(out):
    from phoenix_drone_simulation.utils import utils
    fix = os.path.join(GOLD, "policy_PWM_seed_00000_model.json")
    net = utils.load_network_json(fix)
    sp = np.array(json.load(open(fix))["scaling_parameters"])
    mean, std = torch.as_tensor(sp[0], dtype=torch.float32), torch.as_tensor(sp[1], dtype=torch.float32)
    kw = dict(aggregate_phy_steps=2, domain_randomization=0.10, observation_noise=1, motor_thrust_noise=0.05)
    np.random.seed(4321)
    env = gym.make("DroneCircleSimpleEnv-v0", **kw)
    episodes, lens, rets, terms = 2400, [], [], []
    for ep in range(episodes):
        o, _ = env.reset()
        n, ret = 0, 0.0
        while True:
            with torch.no_grad():
                a = net((torch.as_tensor(o, dtype=torch.float32) - mean) / (std + 1e-5)).numpy()  # utils/export.py:88-92 scaling
            o, r, te, tr, _ = env.step(a)
            n += 1; ret += float(r)
            if te or tr or n >= 500:
                break
        lens.append(n); rets.append(ret); terms.append(bool(te))
    out["circle_reference_policy"] = dict(episodes=episodes, env_id="DroneCircleSimpleEnv-v0", env_kwargs=kw, ep_len=lens,
                                          ep_ret=[round(x, 4) for x in rets], terminated=[int(t) for t in terms])
    print("circle reference policy: len", np.mean(lens), "ret", np.mean(rets), "+-", np.std(rets) / np.sqrt(episodes), "terminated", np.mean(terms))


def main():
    torch.set_num_threads(1)
    out = dict(what="per-episode length / return of two HIP-trained policies played deterministically in the reference's own "
                    "DroneHoverSimpleEnv-v0 (env defaults); a statistical sample, reproducible (numpy seeded before the env is built)",
               generator="oracle/refgen/gen_golden_policy_stats.py", env_id=ENV_ID)
    path = os.path.join(GOLD, "policy_eval_stats.json")
    only = os.environ.get("PO_ONLY")  # regenerate one part, keep the others from the existing file (they are deterministic)
    if only and os.path.exists(path):
        out.update({k: v for k, v in json.load(open(path)).items() if k in ("early", "late", "circle_reference_policy", "circle_attrate", "hover_latency_motor", "hover_hold", "circle_default", "hover_history4", "takeoff_const")})
    for name, episodes in EPISODES.items():
        if only and only != "hover":
            continue
        np.random.seed(4321)
        env = gym.make(ENV_ID)
        ac = core.ActorCritic('mlp', env.observation_space, env.action_space, use_standardized_obs=True, use_scaled_rewards=True,
                              use_shared_weights=False,
                              ac_kwargs={'pi': {'hidden_sizes': (50, 50), 'activation': 'relu'},
                                         'val': {'hidden_sizes': (64, 64), 'activation': 'tanh'}})
        sd = np.load(os.path.join(GOLD, f"hip_policy_{name}.npz"))
        ac.load_state_dict({k: torch.as_tensor(sd[k]) for k in sd.files}, strict=True)
        ac.eval()  # EnvironmentEvaluator: no exploration noise
        lens, rets, terms = [], [], []
        for ep in range(episodes):
            o, _ = env.reset()
            n, ret, term = 0, 0.0, False
            while True:
                a, _, _ = ac.step(torch.as_tensor(o, dtype=torch.float32))
                o, r, te, tr, _ = env.step(a)
                n += 1; ret += float(r)
                if te or tr or n >= 500:
                    term = bool(te)
                    break
            lens.append(n); rets.append(ret); terms.append(term)
        out[name] = dict(episodes=episodes, ep_len=lens, ep_ret=[round(x, 4) for x in rets], terminated=[int(t) for t in terms])
        print(name, "len", np.mean(lens), "+-", np.std(lens) / np.sqrt(episodes), "ret", np.mean(rets), "terminated", np.mean(terms))
    # ---- the other direction: a policy trained BY THE REFERENCE (exp-07 control_mode PWM, 2 physics sub-steps per step,
    # experiments/07_control_structure_hypothesis/run_control_structures.py:53-61; bundled JSON, tests/golden/
    # policy_PWM_seed_00000_model.json) loaded with the reference's own utils.load_network_json and flown in the reference's
    # stochastic DroneCircleSimpleEnv-v0 at that experiment's env settings
    if not only or only == "circle":
        _circle_reference_policy(out)
    if not only or only == "circle_attrate":
        # ---- a HIP-trained policy on exp-07's AttitudeRate configuration (PID rate controller under the policy, 4 physics
        # sub-steps per step: envs/control.py:120-287, run_control_structures.py:53-61) in the reference's Circle env.
        # SEVEN independent env instances (numpy seeds 11..17) x 1 400 episodes, in parallel: a single long-lived env is ONE
        # realisation of the slow sensor-bias walk, and one 2 400-episode run (seed 4321: 423.0 steps, 17.8 % falls) sat 2.7
        # of its own standard errors away from these seven (409 .. 418 steps, 18.7 .. 20.5 % falls)
        import multiprocessing as mp
        kw = dict(control_mode="AttitudeRate", aggregate_phy_steps=4)
        jobs = [(sd_, 1400, "DroneCircleSimpleEnv-v0", kw, "hip_policy_circle_attrate_late.npz") for sd_ in range(11, 18)]
        with mp.get_context("spawn").Pool(7) as pool:
            parts = pool.map(_eval_actor_critic, jobs)
        lens, rets, terms = (sum((p_[i] for p_ in parts), []) for i in range(3))
        out["circle_attrate"] = dict(episodes=len(lens), env_id="DroneCircleSimpleEnv-v0", env_kwargs=kw, numpy_seeds=list(range(11, 18)),
                                     ep_len=lens, ep_ret=[round(x, 4) for x in rets], terminated=[int(t) for t in terms])
        print("circle AttitudeRate HIP policy: len", np.mean(lens), "ret", np.mean(rets), "terminated", np.mean(terms),
              "| per instance:", [round(float(np.mean(p_[0])), 1) for p_ in parts])
    # ---- more of the env's variant families in the loop (HIP-trained policies, profiles/tools/train_export_policies.py EXTRA):
    # the latency ring + first-order motor model (envs/agents.py:259-298) and the Kalman hold (envs/hover.py:134-156)
    extra = {"hover_latency_motor": ("DroneHoverSimpleEnv-v0", dict(use_latency=True, latency=0.02, use_motor_dynamics=True)),
             "hover_hold": ("DroneHoverSimpleEnv-v0", dict(observation_frequency=50)),
             "circle_default": ("DroneCircleSimpleEnv-v0", dict()),
             "hover_history4": ("DroneHoverSimpleEnv-v0", dict(observation_history_size=4))}
    for name, (env_id, kw) in extra.items():
        if only and only != name and only != "extra":
            continue
        import multiprocessing as mp
        jobs = [(sd_, 1000, env_id, kw, f"hip_policy_{name}.npz") for sd_ in range(21, 28)]  # seven env instances x 1 000 episodes
        with mp.get_context("spawn").Pool(7) as pool:
            parts = pool.map(_eval_actor_critic, jobs)
        lens, rets, terms = (sum((p_[i] for p_ in parts), []) for i in range(3))
        out[name] = dict(episodes=len(lens), env_id=env_id, env_kwargs=kw, numpy_seeds=list(range(21, 28)), ep_len=lens,
                         ep_ret=[round(x, 4) for x in rets], terminated=[int(t) for t in terms])
        print(name, "len", np.mean(lens), "+-", np.std(lens) / np.sqrt(len(lens)), "ret", np.mean(rets), "terminated", np.mean(terms))
    if not only or only == "takeoff_const":
        # ---- the third task under a fixed open-loop command (no termination in TakeOff, envs/takeoff.py:100: every episode
        # runs 500 steps; PPO on it overflows early in training): all four motors at HOVER_ACTION + 0.04 --
        # lift-off, climb through the target height; the return depends on the randomised mass / thrust-to-weight / motor
        # noise and on the noisy observations only through the reward's state terms
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(7) as pool:
            parts = pool.map(_eval_constant_action, [(sd_, 300, "DroneTakeOffSimpleEnv-v0", {}, 0.04) for sd_ in range(51, 58)])
        lens, rets = (sum((p_[i] for p_ in parts), []) for i in range(2))
        out["takeoff_const"] = dict(episodes=len(lens), env_id="DroneTakeOffSimpleEnv-v0", env_kwargs={}, action_offset=0.04,
                                    numpy_seeds=list(range(51, 58)), ep_len=lens, ep_ret=[round(x, 4) for x in rets])
        print("takeoff constant action: len", np.mean(lens), "ret", np.mean(rets), "+-", np.std(rets) / np.sqrt(len(rets)), "sd", np.std(rets))
    with open(os.path.join(GOLD, "policy_eval_stats.json"), "w") as f:
        json.dump(out, f)
    print("wrote policy_eval_stats.json")


if __name__ == "__main__":
    main()
