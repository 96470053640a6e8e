"""Several PPOTrainer runs side by side on one GPU (one Python thread and one HIP stream each): a 1-env rollout is one
block on one of 256 CUs, so runs at the reference's layout (1 env x 32 000 steps) overlap almost for free -- IF the HIP
runtime is given enough hardware queues: GPU_MAX_HW_QUEUES=16 in the environment BEFORE the process touches the GPU (the
default of 4 serialises streams that share a queue: 8 runs x 8 threads 5.2 s per run with 4 queues, 1.9 s with 16; 12.4 s
sequentially).  Bit-identical to sequential runs.  tests/test_trainer.py starts this script as a child process.
usage: GPU_MAX_HW_QUEUES=16 python profiles/tools/learning_threads.py [--seeds 8] [--threads 8] [--envs 1] [--env-id ..] [--out f.json]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


BUILD = threading.Lock()  # torch.manual_seed + the networks' initialisation use torch's GLOBAL generator


def run(seed, num_envs, out, epochs=40, spe=32000, env_id="DroneHoverSimpleEnv-v0"):
    import torch
    import phoenix_drone_simulation_amd as pds
    from phoenix_drone_simulation_amd.ppo import PPOTrainer
    with torch.cuda.stream(torch.cuda.Stream()):
        with BUILD:
            env = pds.make(env_id, num_envs=num_envs, seed=seed)
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
    ap.add_argument("--env-id", default="DroneHoverSimpleEnv-v0")
    ap.add_argument("--first-seed", type=int, default=100)
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-check", action="store_true", help="skip the two sequential runs that check determinism")
    a = ap.parse_args()
    import torch
    import phoenix_drone_simulation_amd as pds  # noqa: F401  (imported once, before the threads)
    from phoenix_drone_simulation_amd.ppo import PPOTrainer  # noqa: F401
    torch.cuda.init()
    seeds = list(range(a.first_seed, a.first_seed + a.seeds))
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
            run(s, a.envs, out, env_id=a.env_id)

    ts = [threading.Thread(target=worker) for _ in range(a.threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    dt = time.time() - t0
    print(f"{a.seeds} seeds x {a.envs} envs in {a.threads} threads: {dt:.1f} s = {dt / a.seeds:.1f} s per seed")
    if a.out:
        import json
        json.dump({str(s): [out[s][0].tolist(), out[s][1].tolist()] for s in seeds}, open(a.out, "w"))
    if a.no_check:
        return
    seq = {}
    t0 = time.time()
    for s in seeds[:2]:
        run(s, a.envs, seq, env_id=a.env_id)
    print(f"sequential: {(time.time() - t0) / 2:.1f} s per seed; identical to the threaded runs: "
          f"{all(np.array_equal(seq[s][0], out[s][0]) and np.array_equal(seq[s][1], out[s][1]) for s in seeds[:2])}")
    print("late EpLen per seed", [round(float(out[s][0][23:].mean()), 1) for s in seeds])


if __name__ == "__main__":
    main()
