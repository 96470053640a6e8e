#!/usr/bin/env python3
"""Train PPO on a batched Simple env, evaluate it and leave the reference's run artefacts behind.

Counterpart of the reference's examples/train_drone_hover.py / `python -m phoenix_drone_simulation.train
--alg ppo --env DroneHoverSimpleEnv-v0`: same env ids and kwargs, same PPO hyper-parameters
(algs/ppo/defaults.py), but the rollout is one lockstep batch on the GPU (the env-step kernel of
csrc/pds_step.h, network inference and loss gradients on the f32 matrix cores, csrc/pds_mlp.hip).

    python examples/train_ppo.py --env DroneCircleSimpleEnv-v0 --num-envs 8192 --epochs 300 --log-dir /tmp/run
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phoenix_drone_simulation_amd as pds  # noqa: E402
from phoenix_drone_simulation_amd.evaluation import evaluate  # noqa: E402
from phoenix_drone_simulation_amd.ppo import PPOTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="DroneHoverSimpleEnv-v0")
    ap.add_argument("--num-envs", type=int, default=8192)
    ap.add_argument("--rollout-len", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--log-dir", default=None)
    ap.add_argument("--observation-history-size", type=int, default=2,
                    help="the reference's history of [o, u] pairs fed to the networks (envs/base.py:44; experiments/04_*: 1, 2, 4, 6, 8); "
                         "up to 192 network inputs stay on the fused kernels and the rollout is one launch (pds_rollout_history)")
    ap.add_argument("--pi-hidden", type=int, nargs=2, default=(50, 50), help="policy hidden sizes (experiments/04_*: 32 32 / 48 48 / 64 64)")
    args = ap.parse_args()
    env_kw = dict(observation_history_size=args.observation_history_size) if args.observation_history_size != 2 else {}
    ac_kwargs = {"pi": {"hidden_sizes": tuple(args.pi_hidden), "activation": "relu"}, "val": {"hidden_sizes": (64, 64), "activation": "tanh"}}
    # one process per GPU (the reference: mpi_fork over CPU cores, examples/train_with_multi_cores.py):
    #   python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_ppo.py ...
    # every rank steps its shard of the envs; gradients and running statistics are averaged over RCCL
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        from phoenix_drone_simulation_amd.sharding import make_sharded
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
        env = make_sharded(args.env, args.num_envs * world, seed=args.seed, **env_kw)
    else:
        env = pds.make(args.env, num_envs=args.num_envs, seed=args.seed, **env_kw)  # the reference's default config
    trainer = PPOTrainer(env, rollout_len=args.rollout_len, epochs=args.epochs, seed=args.seed, ac_kwargs=ac_kwargs)
    t0 = time.time()
    for e in range(args.epochs):
        i = trainer.learn_one_epoch()
        if rank == 0 and (e % max(1, args.epochs // 20) == 0 or e == args.epochs - 1):
            print(f"epoch {i['epoch']:4d}  EpRet {i['ep_ret']:9.2f}  EpLen {i['ep_len']:6.1f}  FPS {i['fps']:.3e}", flush=True)
    torch.cuda.synchronize()
    print(f"{args.epochs * args.num_envs * args.rollout_len} env-steps in {time.time() - t0:.1f} s")
    ret, length, cost = evaluate(env, trainer.ac, log_dir=args.log_dir if rank == 0 else None)
    print(f"evaluation: mean return {float(ret.mean()):.2f}  mean episode length {float(length.mean()):.1f}  mean cost {float(cost.mean()):.2f}")
    if args.log_dir and rank == 0:
        trainer.save_checkpoint(args.log_dir)          # torch_save/model.pt + model.json (firmware format)
        trainer.write_progress_csv(os.path.join(args.log_dir, "progress.csv"))
        print("saved to", args.log_dir)


if __name__ == "__main__":
    main()
