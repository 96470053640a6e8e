#!/usr/bin/env python3
"""Episode statistics of the REFERENCE envs under their own numpy randomness (build container only):
tests/golden/episode_stats.json.  The GPU path draws from Philox instead of the global MT19937 stream,
so bitwise comparison is impossible for the stochastic default configuration; this pins it at the level
that matters to a trainer: the distribution of episode lengths and returns under a fixed action
distribution (a = HOVER_ACTION + 0.1 N(0,1), the benchmark's recipe), default env config (sensor noise,
thrust noise, 10 % domain randomisation, reset distribution)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "standins"))
sys.path.insert(0, "/root/reference")

import gymnasium as gym  # noqa: E402  (stand-in)
import phoenix_drone_simulation  # noqa: E402,F401

ENV_IDS = {"hover": "DroneHoverSimpleEnv-v0", "circle": "DroneCircleSimpleEnv-v0"}


def _run(args):
    """(task, numpy seed, episodes) -> per-episode (lengths, returns, costs, first rewards) of ONE env instance"""
    task, seed, episodes = args
    np.random.seed(seed)  # before the env is built: its constructor already draws (envs/base.py:142)
    env = gym.make(ENV_IDS[task])
    arng = np.random.RandomState(99 + seed)  # actions from a private stream: the env's draws stay its own
    hover = -1.0 + 2.0 / 2.25
    lens, rets, costs, first_rew = [], [], [], []
    for ep in range(episodes):
        env.reset()
        done, n, ret, cost = False, 0, 0.0, 0.0
        while not done:
            a = hover + 0.1 * arng.standard_normal(4)
            o, r, te, tr, info = env.step(a)
            if n == 0:
                first_rew.append(float(r))
            n += 1; ret += float(r); cost += float(info.get("cost", 0.0))
            done = bool(te or tr) or n >= 500
        lens.append(n); rets.append(ret); costs.append(cost)
    return lens, rets, costs, first_rew


def main():
    import multiprocessing as mp
    out = {}
    for task, episodes in (("hover", 3000), ("circle", 3000)):
        # seven independent env instances (round 5: one long-lived env is one realisation of the slow sensor-bias walk)
        with mp.get_context("spawn").Pool(7) as pool:
            parts = pool.map(_run, [(task, 1234 + i, episodes) for i in range(7)])
        lens, rets, costs, first_rew = (np.array(sum((p[i] for p in parts), [])) for i in range(4))
        episodes = len(lens)
        out[task] = dict(episodes=episodes, instances=7, len_mean=float(lens.mean()), len_std=float(lens.std()),
                         ret_mean=float(rets.mean()), ret_std=float(rets.std()),
                         ret_per_step_mean=float((rets / lens).mean()), ret_per_step_std=float((rets / lens).std()),
                         cost_per_step_mean=float((costs / lens).mean()),
                         first_reward_mean=float(first_rew.mean()), first_reward_std=float(first_rew.std()),
                         len_quantiles=[float(q) for q in np.quantile(lens, [0.1, 0.25, 0.5, 0.75, 0.9])])
        print(task, out[task])
    path = os.path.join(HERE, "..", "..", "tests", "golden", "episode_stats.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
