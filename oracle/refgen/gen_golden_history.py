#!/usr/bin/env python3
"""Golden trajectories of the REFERENCE envs with observation_history_size != 2 (build container only):
tests/golden/history.npz.  Experiment 04 of the reference trains with history sizes 1, 2, 4, 6, 8
(experiments/04_history_of_state_action_inputs/04_train_with_history.py:34).

Deterministic configuration (no sensor / thrust noise, no domain randomisation, no reset
distribution), so the trajectory is a pure function of the action sequence: reset, step a recorded
action sequence until the env terminates or is truncated, reset again, a few more steps.  Only data
is written (actions in, observations / rewards / flags out)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "standins"))
sys.path.insert(0, "/root/reference")

import gymnasium as gym  # noqa: E402  (stand-in)
import phoenix_drone_simulation  # noqa: E402,F401

ENV_IDS = {"hover": "DroneHoverSimpleEnv-v0", "circle": "DroneCircleSimpleEnv-v0", "takeoff": "DroneTakeOffSimpleEnv-v0"}


def run(task, H, steps, max_steps, seed):
    env = gym.make(ENV_IDS[task], observation_history_size=H, observation_noise=-1, domain_randomization=-1,
                   motor_thrust_noise=0.0, enable_reset_distribution=False, max_episode_steps=max_steps)
    rs = np.random.RandomState(seed)
    obs0, _ = env.reset()
    hover = -1.0 + 2.0 / 2.25
    acts, obs, rew, term, trunc, cost, reset_obs = [], [], [], [], [], [], []
    for t in range(steps):
        a = hover + (0.25 if task == "takeoff" else 0.0) + 0.15 * rs.standard_normal(4) + (0.004 * t if task == "hover" else 0.0) * np.array([1, -1, -1, 1.0])
        o, r, te, tr, info = env.step(a)
        acts.append(a); obs.append(np.array(o, dtype=np.float64)); rew.append(float(r)); term.append(bool(te)); trunc.append(bool(tr))
        cost.append(float(info.get("cost", 0.0)))
        if te or tr:
            o2, _ = env.reset()
            reset_obs.append(np.array(o2, dtype=np.float64))
        else:
            reset_obs.append(np.zeros_like(np.array(o, dtype=np.float64)))
    return dict(obs0=np.array(obs0, dtype=np.float64), actions=np.array(acts), obs=np.array(obs), reward=np.array(rew),
                terminated=np.array(term), truncated=np.array(trunc), cost=np.array(cost), reset_obs=np.array(reset_obs))


def main():
    out = {}
    for task, H, steps, max_steps in (("hover", 1, 60, 25), ("hover", 4, 90, 500), ("circle", 4, 60, 22), ("circle", 6, 40, 500),
                                      ("takeoff", 8, 40, 15), ("hover", 2, 40, 500)):
        d = run(task, H, steps, max_steps, seed=100 + H)
        for k, v in d.items():
            out[f"{task}_h{H}_{k}"] = v
        out[f"{task}_h{H}_max_steps"] = np.int64(max_steps)
        print(task, H, "obs dim", d["obs"].shape[1], "terminated", int(d["terminated"].sum()), "truncated", int(d["truncated"].sum()))
    path = os.path.join(HERE, "..", "..", "tests", "golden", "history.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "kB")


if __name__ == "__main__":
    main()
