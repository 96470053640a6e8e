"""Several PPOTrainer runs side by side on one GPU (one Python thread and one HIP stream each): a 1-env rollout is one
block on one of 256 CUs, so runs at the reference's layout (1 env x 32 000 steps) overlap almost for free.
usage: python profiles/tools/learning_threads.py [--seeds 8] [--threads 4] [--envs 1]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


BUILD = threading.Lock()  # torch.manual_seed + the networks' initialisation use torch's GLOBAL generator


def run(seed, num_envs, out, epochs=40, spe=32000):
    import torch
    import phoenix_drone_simulation_amd as pds
    from phoenix_drone_simulation_amd.ppo import PPOTrainer
    with torch.cuda.stream(torch.cuda.Stream()):
        with BUILD:
            env = pds.make("DroneHoverSimpleEnv-v0", num_envs=num_envs, seed=seed)
            tr = PPOTrainer(env, rollout_len=spe // num_envs, epochs=epochs, seed=seed, reset_each_rollout=True)
            torch.cuda.current_stream().synchronize()
        tr.learn()
        torch.cuda.current_stream().synchronize()
        out[seed] = (np.array([r["ep_len"] for r in tr.log]), np.array([r["ep_ret"] for r in tr.log]))
        env.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--envs", type=int, default=1)
    a = ap.parse_args()
    import torch
    import phoenix_drone_simulation_amd as pds  # noqa: F401  (imported once, before the threads)
    from phoenix_drone_simulation_amd.ppo import PPOTrainer  # noqa: F401
    torch.cuda.init()
    seeds = list(range(100, 100 + a.seeds))
    out = {}
    t0 = time.time()
    pending = list(seeds)
    lock = threading.Lock()

    def worker():
        while True:
            with lock:
                if not pending:
                    return
                s = pending.pop(0)
            run(s, a.envs, out)

    ts = [threading.Thread(target=worker) for _ in range(a.threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    dt = time.time() - t0
    print(f"{a.seeds} seeds x {a.envs} envs in {a.threads} threads: {dt:.1f} s = {dt / a.seeds:.1f} s per seed")
    seq = {}
    t0 = time.time()
    for s in seeds[:2]:
        run(s, a.envs, seq)
    print(f"sequential: {(time.time() - t0) / 2:.1f} s per seed; identical to the threaded runs: "
          f"{all(np.array_equal(seq[s][0], out[s][0]) and np.array_equal(seq[s][1], out[s][1]) for s in seeds[:2])}")
    print("late EpLen per seed", [round(float(out[s][0][23:].mean()), 1) for s in seeds])


if __name__ == "__main__":
    main()
