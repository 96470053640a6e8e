#!/usr/bin/env python3
"""PPOTrainer on the HIP envs at the configuration of the reference's own learn() run that
tests/golden/learning_curve.json records (oracle/refgen/gen_golden_learning.py): DroneHoverSimpleEnv-v0 with the
env's defaults (sensor noise, 10 % DR, thrust noise), 32 000 steps per epoch, EPOCHS epochs (the linear
exploration-noise / learning-rate schedules span exactly that), PPO defaults (algs/ppo/defaults.py:6-19),
`reset_each_rollout=True` (IWPGAlgorithm.roll_out starts every epoch with env.reset()).

Prints per epoch: the reference's mean / min / max over its seeds and this trainer's mean / min / max over its
seeds, for EpRet and EpLen.   usage (GPU box): python profiles/tools/learning_curve.py [--seeds 6] [--envs 8]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden", "learning_curve.json")


def reference_curves():
    ref = json.load(open(GOLD))
    seeds = [str(s) for s in ref["seeds"]]
    ret = np.array([ref["curves"][s]["EpRet/Mean"] for s in seeds])
    length = np.array([ref["curves"][s]["EpLen/Mean"] for s in seeds])
    return ref, ret, length


def run_trainer(seed, epochs, steps_per_epoch, num_envs, env_id="DroneHoverSimpleEnv-v0", **trainer_kw):
    """-> (EpRet/Mean [epochs], EpLen/Mean [epochs], seconds)"""
    import torch
    import phoenix_drone_simulation_amd as pds
    from phoenix_drone_simulation_amd.ppo import PPOTrainer
    assert steps_per_epoch % num_envs == 0
    env = pds.make(env_id, num_envs=num_envs, seed=seed)
    tr = PPOTrainer(env, rollout_len=steps_per_epoch // num_envs, epochs=epochs, seed=seed, reset_each_rollout=True,
                    **trainer_kw)
    t0 = time.time()
    tr.learn()
    torch.cuda.synchronize()
    dt = time.time() - t0
    ret = np.array([r["ep_ret"] for r in tr.log])
    length = np.array([r["ep_len"] for r in tr.log])
    env.close()
    return ret, length, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=6)
    ap.add_argument("--envs", type=int, default=8)
    a = ap.parse_args()
    ref, rret, rlen = reference_curves()
    E, spe = ref["epochs"], ref["steps_per_epoch"]
    mine = [run_trainer(100 + s, E, spe, a.envs, env_id=ref["env_id"]) for s in range(a.seeds)]
    mret, mlen = np.array([m[0] for m in mine]), np.array([m[1] for m in mine])
    print(f"# {ref['env_id']} (env defaults), {spe} steps per epoch, {E} epochs; reference: {len(ref['seeds'])} seeds of its own "
          f"learn() ({np.mean(ref['wall_s']):.0f} s each on one CPU core); HIP trainer: {a.seeds} seeds, {a.envs} envs x "
          f"{spe // a.envs} steps, {np.mean([m[2] for m in mine]):.1f} s each")
    print("epoch | reference EpRet mean [min, max] | HIP EpRet mean [min, max] | reference EpLen mean [min, max] | HIP EpLen mean [min, max]")
    for e in range(E):
        print(f"{e + 1:5d} | {rret[:, e].mean():9.2f} [{rret[:, e].min():9.2f}, {rret[:, e].max():9.2f}] | "
              f"{mret[:, e].mean():9.2f} [{mret[:, e].min():9.2f}, {mret[:, e].max():9.2f}] | "
              f"{rlen[:, e].mean():7.1f} [{rlen[:, e].min():7.1f}, {rlen[:, e].max():7.1f}] | "
              f"{mlen[:, e].mean():7.1f} [{mlen[:, e].min():7.1f}, {mlen[:, e].max():7.1f}]")


if __name__ == "__main__":
    main()
