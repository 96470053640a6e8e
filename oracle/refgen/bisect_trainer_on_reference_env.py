#!/usr/bin/env python3
"""Bisection tool for the late-epoch offset of the learning-curve pin: THIS repo's
trainer logic (`ppo.PPOTrainer`, PyTorch-op path, CPU tensors) on the REFERENCE's own envs.

Runs in the build container only (imports /root/reference like the other generators; test infrastructure,
writes data only).  Two questions, one run each:
  * the trainer against the real env at the reference's layout (1 env x 32 000 steps) -- does `PPOTrainer`
    restate IWPGAlgorithm.learn()?
  * the same at the GPU test's layout (8 envs x 4 000 steps) -- does the layout matter?
If both follow the reference's curve, the offset comes from the HIP env in closed loop or from the fused kernels.

usage: bisect_trainer_on_reference_env.py [--envs 1] [--steps 32000] [--seeds 6] [--first-seed 100] [--epochs 40]
                                          [--workers 6] [--out /tmp/bisect.json]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ENV_ID = "DroneHoverSimpleEnv-v0"


def cpu_gae(rew, val, terminated, truncated, final_val, last_val, gamma, lam, rew_scale=0.0, rew_clip=10.0):
    """csrc/pds_gae.hip in torch (backward scan over [T, N])."""
    import torch
    T, N = rew.shape
    adv, tv, dr = torch.empty_like(rew), torch.empty_like(rew), torch.empty_like(rew)
    next_val, next_ret, next_adv = last_val.clone(), last_val.clone(), torch.zeros(N)
    gl = gamma * lam
    for t in range(T - 1, -1, -1):
        te, tr = terminated[t].bool(), truncated[t].bool()
        end = te | tr
        b = torch.where(tr, final_val[t], torch.zeros(N))  # (cut wins over terminated: iwpg.py:374-379)
        next_val = torch.where(end, b, next_val)
        next_ret = torch.where(end, b, next_ret)
        next_adv = torch.where(end, torch.zeros(N), next_adv)
        rs = torch.clamp(rew[t] * rew_scale, -rew_clip, rew_clip) if rew_scale > 0 else rew[t]
        delta = rs + gamma * next_val - val[t]
        a = delta + gl * next_adv
        g = rew[t] + gamma * next_ret
        adv[t], tv[t], dr[t] = a, a + val[t], g
        next_val, next_adv, next_ret = val[t], a, g
    return adv, tv, dr


class RefVecEnv:
    """N reference envs behind the DroneVecEnv surface PPOTrainer uses (auto-reset + final_obs)."""

    def __init__(self, n, seed):
        import gymnasium as gym
        import numpy as np
        import torch
        self.np, self.torch = np, torch
        self.envs = [gym.make(ENV_ID) for _ in range(n)]
        self.num_envs, self.device = n, torch.device("cpu")
        self.obs_dim = int(self.envs[0].observation_space.shape[0])
        self.act_dim = 4
        self.max_steps = self.envs[0]._max_episode_steps
        self.len = [0] * n
        self.envs[0].reset(seed=seed)  # (np.random is global: IWPGAlgorithm seeds it once, iwpg.py:124-127)

    def reset(self):
        obs = [e.reset()[0] for e in self.envs]
        self.len = [0] * self.num_envs
        return self.torch.as_tensor(self.np.stack(obs), dtype=self.torch.float32), {}

    def step(self, a):
        np, torch = self.np, self.torch
        a = a.numpy()
        obs, rew, term, trunc, fin = [], [], [], [], []
        for i, e in enumerate(self.envs):
            o, r, te, _, _ = e.step(a[i])
            self.len[i] += 1
            tr = self.len[i] == self.max_steps  # iwpg.py:370: the trainer's own count
            fin.append(o)
            if te or tr:
                o, _ = e.reset()
                self.len[i] = 0
            obs.append(o); rew.append(r); term.append(te); trunc.append(tr)
        f = lambda x, dt=torch.float32: torch.as_tensor(np.asarray(x), dtype=dt)  # noqa: E731
        return f(np.stack(obs)), f(rew), f(term, torch.bool), f(trunc, torch.bool), {"final_obs": f(np.stack(fin))}


def run_seed(args):
    seed, n_envs, steps, epochs = args
    import torch
    torch.set_num_threads(1)
    sys.path.insert(0, os.path.join(HERE, "standins"))
    sys.path.insert(0, "/root/reference")
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules["torch.utils.tensorboard"] = tb
    import numpy as np
    import phoenix_drone_simulation  # noqa: F401
    import importlib.util
    # the package's ppo.py without its __init__ (which loads the HIP library)
    pkg = types.ModuleType("pds_amd"); pkg.__path__ = [os.path.join(HERE, "..", "..", "phoenix-drone-simulation_amd")]
    sys.modules["pds_amd"] = pkg
    for name in ("native", "fused", "ppo"):
        spec = importlib.util.spec_from_file_location(f"pds_amd.{name}", os.path.join(pkg.__path__[0], f"{name}.py"))
        mod = importlib.util.module_from_spec(spec); sys.modules[f"pds_amd.{name}"] = mod
        if name == "ppo":
            spec.loader.exec_module(mod)
        elif name == "fused":
            for fn in ("counter_add", "gaussian_sample", "random_permutation", "rollout_record"):
                setattr(mod, fn, None)
    ppo = sys.modules["pds_amd.ppo"]
    ppo.gae = cpu_gae
    torch.manual_seed(seed); np.random.seed(seed)
    env = RefVecEnv(n_envs, seed)
    t0 = time.time()
    tr = ppo.PPOTrainer(env, rollout_len=steps // n_envs, epochs=epochs, seed=seed, fused=False, reset_each_rollout=True)
    tr.learn()
    return dict(seed=seed, wall_s=time.time() - t0, ep_len=[r["ep_len"] for r in tr.log], ep_ret=[r["ep_ret"] for r in tr.log],
                loss_v=[r["loss_v"] for r in tr.log])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32000)
    ap.add_argument("--seeds", type=int, default=6)
    ap.add_argument("--first-seed", type=int, default=100)
    ap.add_argument("--epochs", type=int, default=40)
    ap.add_argument("--workers", type=int, default=6)
    ap.add_argument("--out", default="/tmp/bisect.json")
    a = ap.parse_args()
    jobs = [(s, a.envs, a.steps, a.epochs) for s in range(a.first_seed, a.first_seed + a.seeds)]
    with mp.get_context("spawn").Pool(min(a.workers, len(jobs))) as pool:
        res = pool.map(run_seed, jobs)
    json.dump(dict(envs=a.envs, steps=a.steps, epochs=a.epochs, runs=res), open(a.out, "w"))
    for r in res:
        print("seed", r["seed"], round(r["wall_s"]), "s EpLen", [round(x) for x in r["ep_len"]][::3])


if __name__ == "__main__":
    main()
