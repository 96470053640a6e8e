"""Several PPOTrainer runs side by side on one GPU (one Python thread and one HIP stream each): a 1-env rollout is one
block on one of 256 CUs, so runs at the reference's layout (1 env x 32 000 steps) overlap almost for free -- IF the HIP
runtime is given enough hardware queues: GPU_MAX_HW_QUEUES=16 in the environment BEFORE the process touches the GPU (the
default of 4 serialises streams that share a queue: 8 runs x 8 threads 5.2 s per run with 4 queues, 1.9 s with 16; 12.4 s
sequentially).  Bit-identical to sequential runs.  tests/test_trainer.py starts this script as a child process.
usage: GPU_MAX_HW_QUEUES=16 python profiles/tools/learning_threads.py [--seeds 8] [--threads 8] [--envs 1] [--env-id ..] [--out f.json]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run_all(seeds, num_envs, threads, env_id, epochs=40, spe=32000):
    """-> {seed: (EpLen/Mean [epochs], EpRet/Mean [epochs])} through ppo.train_runs_side_by_side"""
    from phoenix_drone_simulation_amd.ppo import train_runs_side_by_side
    logs = train_runs_side_by_side(env_id, seeds, num_envs, spe // num_envs, epochs, threads=threads,
                                   trainer_kwargs=dict(reset_each_rollout=True))
    return {s: (np.array([r["ep_len"] for r in logs[s]]), np.array([r["ep_ret"] for r in logs[s]])) for s in seeds}


def run_all_columns(seeds, num_envs, threads, env_id, epochs=40, spe=32000):
    """-> {seed: {column: [epochs]}} with the reference's other progress.csv columns (PPOTrainer(log_reference_columns=True))"""
    from phoenix_drone_simulation_amd.ppo import train_runs_side_by_side
    logs = train_runs_side_by_side(env_id, seeds, num_envs, spe // num_envs, epochs, threads=threads,
                                   trainer_kwargs=dict(reset_each_rollout=True, log_reference_columns=True))
    cols = ("ep_len", "ep_ret", "loss_pi", "loss_v", "values_v_mean", "rew_scale_mean", "rew_scale_std", "kl")
    return {s: {c: [float(r.get(c, float("nan"))) for r in logs[s]] for c in cols} for s in seeds}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--envs", type=int, default=1)
    ap.add_argument("--env-id", default="DroneHoverSimpleEnv-v0")
    ap.add_argument("--first-seed", type=int, default=100)
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-check", action="store_true", help="skip the two sequential runs that check determinism")
    ap.add_argument("--columns", action="store_true", help="record the reference's other progress.csv columns too (implies --no-check)")
    a = ap.parse_args()
    import torch
    import phoenix_drone_simulation_amd as pds  # noqa: F401  (imported once, before the threads)
    from phoenix_drone_simulation_amd.ppo import PPOTrainer  # noqa: F401
    torch.cuda.init()
    seeds = list(range(a.first_seed, a.first_seed + a.seeds))
    if a.columns:
        import json
        t0 = time.time()
        out = run_all_columns(seeds, a.envs, a.threads, a.env_id)
        print(f"{a.seeds} seeds x {a.envs} envs in {a.threads} threads: {time.time() - t0:.1f} s")
        json.dump({str(s): out[s] for s in seeds}, open(a.out, "w"))
        return
    t0 = time.time()
    out = run_all(seeds, a.envs, a.threads, a.env_id)
    dt = time.time() - t0
    print(f"{a.seeds} seeds x {a.envs} envs in {a.threads} threads: {dt:.1f} s = {dt / a.seeds:.1f} s per seed")
    if a.out:
        import json
        json.dump({str(s): [out[s][0].tolist(), out[s][1].tolist()] for s in seeds}, open(a.out, "w"))
    if a.no_check:
        return
    t0 = time.time()
    seq = run_all(seeds[:2], a.envs, 1, a.env_id)
    print(f"sequential: {(time.time() - t0) / 2:.1f} s per seed; identical to the threaded runs: "
          f"{all(np.array_equal(seq[s][0], out[s][0]) and np.array_equal(seq[s][1], out[s][1]) for s in seeds[:2])}")
    print("late EpLen per seed", [round(float(out[s][0][23:].mean()), 1) for s in seeds])


if __name__ == "__main__":
    main()
