#!/usr/bin/env python3
"""Bisection of the learning-curve offset on the GPU box: PPOTrainer on the HIP envs at
the reference's configuration (tests/golden/learning_curve.json) in several VARIANTS of layout / code path, many
seeds each; prints the per-epoch mean EpLen / EpRet next to the reference's and a late-epoch summary
(mean over epochs 24-40 per seed -> mean +- SE over seeds).

usage (GPU box): python profiles/tools/learning_bisect.py [--seeds 12] [--variants base,torchops,n1,...]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from learning_curve import reference_curves  # noqa: E402

VARIANTS = {
    # name: (num_envs, trainer kwargs, env kwargs)
    "base": (8, {}, {}),
    "perstep": (8, dict(fused_rollout=False), {}),
    "torchops": (8, dict(fused=False), {}),
    "n1": (1, {}, {}),
    "n2": (2, {}, {}),
    "n32": (32, {}, {}),
    "seqvalue": (8, dict(overlap_value_update=False), {}),
    # mechanism check: without the sensor noise (no persistent gyro bias shared by all episodes of an env)
    "n1_quiet": (1, {}, dict(observation_noise=-1)),
    "n8_quiet": (8, {}, dict(observation_noise=-1)),
    "n64": (64, {}, {}),
}


def run(seed, E, spe, num_envs, env_id, tkw, ekw):
    import torch
    import phoenix_drone_simulation_amd as pds
    from phoenix_drone_simulation_amd.ppo import PPOTrainer
    env = pds.make(env_id, num_envs=num_envs, seed=seed, **ekw)
    tr = PPOTrainer(env, rollout_len=spe // num_envs, epochs=E, seed=seed, reset_each_rollout=True, **tkw)
    t0 = time.time()
    tr.learn()
    torch.cuda.synchronize()
    out = (np.array([r["ep_ret"] for r in tr.log]), np.array([r["ep_len"] for r in tr.log]),
           np.array([r["loss_v"] for r in tr.log]), time.time() - t0)
    env.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=12)
    ap.add_argument("--variants", default="base,perstep,torchops,n1,n32")
    ap.add_argument("--first-seed", type=int, default=100)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    ref, rret, rlen = reference_curves()
    E, spe = ref["epochs"], ref["steps_per_epoch"]
    late = slice(23, 40)
    res = {}
    for name in a.variants.split(","):
        n, tkw, ekw = VARIANTS[name]
        runs = [run(a.first_seed + s, E, spe, n, ref["env_id"], tkw, ekw) for s in range(a.seeds)]
        res[name] = dict(ret=np.array([r[0] for r in runs]), len=np.array([r[1] for r in runs]),
                         loss_v=np.array([r[2] for r in runs]), secs=float(np.mean([r[3] for r in runs])))
        lt = res[name]["len"][:, late].mean(axis=1)
        print(f"# {name}: {n} envs x {spe // n} steps, {a.seeds} seeds from {a.first_seed}, {res[name]['secs']:.1f} s each; late EpLen per seed "
              f"{np.round(lt, 1).tolist()} -> {lt.mean():.2f} +- {lt.std(ddof=1) / np.sqrt(len(lt)):.2f}", flush=True)
    names = list(res)
    print("epoch | ref EpLen mean (SE) | " + " | ".join(f"{n} EpLen mean (SE)" for n in names))
    se = lambda x: x.std(axis=0, ddof=1) / np.sqrt(x.shape[0])  # noqa: E731
    for e in range(E):
        print(f"{e + 1:5d} | {rlen[:, e].mean():6.1f} ({se(rlen)[e]:4.1f}) | " +
              " | ".join(f"{res[n]['len'][:, e].mean():6.1f} ({se(res[n]['len'])[e]:4.1f})" for n in names))
    print("late-epoch summary (epochs 24-40, mean over epochs per seed -> mean +- SE over seeds)")
    for key, rr in (("len", rlen), ("ret", rret)):
        r = rr[:, late].mean(axis=1)
        print(f"  {key}: reference {r.mean():8.2f} +- {r.std(ddof=1) / np.sqrt(len(r)):5.2f}  (n={len(r)})")
        for n in names:
            m = res[n][key][:, late].mean(axis=1)
            d = m.mean() - r.mean()
            s = np.sqrt(m.var(ddof=1) / len(m) + r.var(ddof=1) / len(r))
            print(f"  {key}: {n:10s} {m.mean():8.2f} +- {m.std(ddof=1) / np.sqrt(len(m)):5.2f}   diff {d:+7.2f} = {d / s:+5.1f} SE")
    if a.out:
        json.dump({n: {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in r.items()} for n, r in res.items()},
                  open(a.out, "w"))


if __name__ == "__main__":
    main()
