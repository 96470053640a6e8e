#!/usr/bin/env python3
"""Golden vectors for ONE FULL update() of the reference's PPO trainer, two consecutive epochs (VERDICT
round 4 item 1a: "a step that can be pinned exactly").

Runs, in the build container only, `ProximalPolicyOptimizationAlgorithm` (algs/ppo/ppo.py:12-63) on the reference's own
`DroneHoverSimpleEnv-v0` for two epochs of `IWPGAlgorithm.learn_one_epoch` (algs/iwpg/iwpg.py:282-300) -- exploration-noise
anneal, roll_out, update (value net mini-batches, then the policy steps, then the running statistics), LambdaLR step -- at a
small size (1 000 steps per epoch, 4 mini-batches x 2 value iterations, 10 policy iterations) and records everything a
restatement needs to repeat the two updates WITHOUT any randomness of its own:

  * the ActorCritic state_dict before epoch 0 and after each update (weights, obs_oms / ret_oms statistics, log_std);
  * per epoch the rollout buffer (obs, act, rew, val, logp), where each path ended and with which bootstrap value
    (Buffer.finish_path calls), the terminated flags, and the index arrays `np.random.shuffle` produced for the value net;
  * per epoch the buffer's own outputs (adv, target_v, discounted_ret), the policy learning rate, Loss/Pi and Loss/Value.

tests/test_trainer.py feeds the recorded rollouts and shuffles to `ppo.PPOTrainer.update()` (PyTorch-op path on CPU tensors,
fused MFMA path on the GPU) and compares the parameters after each update.  Only data is written: tests/golden/update.npz.
Stand-ins (absent modules): pybullet*, gymnasium, mpi4py (one rank), torch.utils.tensorboard -- as for the other generators."""
import copy
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "standins"))
sys.path.insert(0, "/root/reference")
_tb = types.ModuleType("torch.utils.tensorboard")
_tb.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = _tb

ENV_ID = "DroneHoverSimpleEnv-v0"
EPOCHS_TOTAL, STEPS, MINI, V_ITERS, PI_ITERS = 8, 1000, 4, 2, 10


def main():
    torch.set_num_threads(1)
    import phoenix_drone_simulation  # noqa: F401  (registers the env ids)
    from phoenix_drone_simulation.algs.ppo import ppo
    from phoenix_drone_simulation.utils import utils

    # IWPGAlgorithm.__init__ builds the env BEFORE it seeds numpy (iwpg.py:72-75 vs 124-127), and DroneBaseEnv.__init__ already
    # draws from the global generator (compute_observation at envs/base.py:142 advances the gyro-bias walk): without this line
    # the first observation differs from run to run by ~1e-3 and the fixture would not regenerate bit for bit
    np.random.seed(20261003)
    log_dir = tempfile.mkdtemp(prefix="ref_update_")
    kw = utils.get_defaults_kwargs(alg="ppo", env_id=ENV_ID)
    kw.update(epochs=EPOCHS_TOTAL, steps_per_epoch=STEPS, seed=3, verbose=False, save_freq=10 ** 9, num_mini_batches=MINI,
              train_v_iterations=V_ITERS, train_pi_iterations=PI_ITERS,
              logger_kwargs=dict(log_dir=log_dir, exp_name="golden", level=0, use_tensor_board=False, verbose=False))
    alg = ppo.ProximalPolicyOptimizationAlgorithm(env_id=ENV_ID, **kw)
    out = dict(steps=np.int64(STEPS), epochs_total=np.int64(EPOCHS_TOTAL), num_mini_batches=np.int64(MINI),
               train_v_iterations=np.int64(V_ITERS), train_pi_iterations=np.int64(PI_ITERS),
               obs_dim=np.int64(alg.env.observation_space.shape[0]), gamma=np.float64(alg.buf.gamma), lam=np.float64(alg.buf.lam),
               pi_lr=np.float64(alg.pi_lr), vf_lr=np.float64(alg.vf_lr), clip_ratio=np.float64(alg.clip_ratio))
    for k, v in alg.ac.state_dict().items():
        out["sd_init__" + k] = v.numpy().copy()

    # ---- recorders -------------------------------------------------------------------------------------------------------
    rec = dict(paths=[], term=[], shuffles=[])
    finish = alg.buf.finish_path

    def finish_path(last_val=0):
        rec["paths"].append((alg.buf.ptr, float(np.asarray(last_val).reshape(-1)[0])))
        return finish(last_val)
    alg.buf.finish_path = finish_path
    step = alg.env.step

    def env_step(a):
        r = step(a)
        rec["term"].append(bool(r[2]))
        return r
    alg.env.step = env_step
    shuffle = np.random.shuffle

    def rec_shuffle(x):
        shuffle(x)
        rec["shuffles"].append(np.array(x, dtype=np.int64).copy())
    np.random.shuffle = rec_shuffle

    for e in range(2):
        alg.epoch = e
        rec["paths"].clear(); rec["term"].clear(); rec["shuffles"].clear()
        out[f"e{e}_lr"] = np.float64(alg.pi_optimizer.param_groups[0]["lr"])
        alg.ac.update(frac=e / alg.epochs)                       # learn_one_epoch: exploration-noise anneal
        out[f"e{e}_log_std"] = alg.ac.pi.log_std.detach().numpy().copy()
        alg.roll_out()
        b = alg.buf
        for name in ("obs_buf", "act_buf", "rew_buf", "val_buf", "logp_buf", "adv_buf", "target_val_buf", "discounted_ret_buf"):
            out[f"e{e}_{name}"] = getattr(b, name).copy()
        out[f"e{e}_path_end"] = np.array([p for p, _ in rec["paths"]], dtype=np.int64)     # exclusive end index of each path
        out[f"e{e}_path_last_val"] = np.array([v for _, v in rec["paths"]], dtype=np.float32)
        out[f"e{e}_terminated"] = np.array(rec["term"], dtype=np.uint8)
        assert len(rec["term"]) == STEPS and rec["paths"][-1][0] == STEPS
        alg.update()
        out[f"e{e}_shuffles"] = np.stack(rec["shuffles"])                                  # [train_v_iterations, STEPS]
        out[f"e{e}_loss_pi"] = np.float64(alg.loss_pi_before)
        out[f"e{e}_loss_v"] = np.float64(alg.loss_v_before)
        for k, v in alg.ac.state_dict().items():
            out[f"e{e}_sd_after__" + k] = v.numpy().copy()
        alg.scheduler.step()                                     # IWPGAlgorithm.log (iwpg.py:307-309), without the logger
    np.random.shuffle = shuffle
    path = os.path.join(HERE, "..", "..", "tests", "golden", "update.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "kB;", len(out), "arrays; paths per epoch",
          [len(out[f"e{e}_path_end"]) for e in range(2)], "lr", [float(out[f"e{e}_lr"]) for e in range(2)])


if __name__ == "__main__":
    main()
