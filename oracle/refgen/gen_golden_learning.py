#!/usr/bin/env python3
"""Learning curves of the REFERENCE's own PPO trainer on the reference's own Simple env (SURVEY.md 8f
rank 1, "Hover return vs epochs"): the end-to-end pin of the on-device caller (`ppo.PPOTrainer`).

Runs, in the build container only, `ProximalPolicyOptimizationAlgorithm` (algs/ppo/ppo.py:12-63,
IWPGAlgorithm.learn / roll_out / update algs/iwpg/iwpg.py:259-485, defaults algs/ppo/defaults.py:6-19:
pi 50-50 relu, V 64-64 tanh, gamma .99, 32 000 steps per epoch) on `DroneHoverSimpleEnv-v0` with the
env's default configuration (sensor noise, 10 % domain randomisation, thrust noise), for SEEDS x EPOCHS,
and records what its logger writes per epoch (progress.csv: EpRet / EpLen / Loss/Value / exploration
noise / reward scale).  Only data is written: tests/golden/learning_curve.json.

The linear schedules (exploration-noise anneal core.py:268-276, LambdaLR iwpg.py:178-187) depend on the
`epochs` argument, so the run is made with epochs = EPOCHS exactly; the GPU test uses the same number.

Stand-ins (absent modules, as for the other generators): pybullet / pybullet_data / pybullet_utils (pure
math + a dict of base poses, see gen_golden.py), gymnasium, mpi4py (one rank), torch.utils.tensorboard.

The file is a STATISTICAL SAMPLE of the reference trainer's seed distribution, not a known answer; it feeds two-sample tests
only.  (Rounds 4-5 found re-runs of a seed to diverge from the first update on: the reference builds its env -- whose
constructor already draws from numpy's global generator -- before it seeds numpy.  run_seed() now seeds numpy first.)  `--first-seed K --merge` appends seeds K.. to an existing file.

usage: gen_golden_learning.py [--seeds 5] [--first-seed 0] [--merge] [--epochs 30] [--workers 5]
                              [--out tests/golden/learning_curve.json]
"""
import argparse
import csv
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ENV_ID = "DroneHoverSimpleEnv-v0"
COLUMNS = ["Epoch", "EpRet/Mean", "EpRet/Std", "EpRet/Min", "EpRet/Max", "EpLen/Mean", "EpLen/Min", "EpLen/Max",
           "Values/V/Mean", "Loss/Pi", "Loss/Value", "Entropy", "KL", "PolicyRatio", "LR", "Misc/RewScaleMean",
           "Misc/RewScaleStddev", "Misc/ExplorationNoiseStd", "TotalEnvSteps"]


def run_seed(args):
    seed, epochs, steps_per_epoch, env_id = args
    import torch
    torch.set_num_threads(1)
    sys.path.insert(0, os.path.join(HERE, "standins"))
    sys.path.insert(0, "/root/reference")
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules["torch.utils.tensorboard"] = tb
    import phoenix_drone_simulation  # noqa: F401  (registers the env ids)
    from phoenix_drone_simulation.algs.ppo import ppo
    from phoenix_drone_simulation.utils import utils, loggers

    # IWPGAlgorithm.__init__ builds the env BEFORE it seeds numpy (iwpg.py:72-75 vs 124-127) and DroneBaseEnv.__init__ already
    # draws from the global generator (compute_observation at envs/base.py:142 advances the gyro-bias walk from it): left alone,
    # the same `seed` gives a first observation that differs by ~1e-3 from run to run and curves that diverge from the first
    # update on.  Seeding numpy here makes a run a function of `seed` (oracle/refgen/gen_golden_update.py regenerates bit for
    # bit for the same reason); the runs stay a SAMPLE of the trainer's distribution either way.
    import numpy as np
    np.random.seed(977 * int(seed) + 13)
    log_dir = tempfile.mkdtemp(prefix=f"ref_ppo_s{seed}_")
    kw = utils.get_defaults_kwargs(alg="ppo", env_id=env_id)
    kw.update(epochs=epochs, steps_per_epoch=steps_per_epoch, seed=seed, verbose=False, save_freq=10 ** 9,
              logger_kwargs=dict(log_dir=log_dir, exp_name="golden", level=0, use_tensor_board=False, verbose=False))
    t0 = time.time()
    alg = ppo.ProximalPolicyOptimizationAlgorithm(env_id=env_id, **kw)
    hyper = {k: (list(v) if isinstance(v, tuple) else v) for k, v in alg.params.items()
             if isinstance(v, (int, float, str, bool, tuple)) and k not in ("seed",)}
    alg.learn()
    rows = []
    with open(os.path.join(log_dir, "progress.csv")) as f:
        for r in csv.DictReader(f, delimiter="," if "," in open(os.path.join(log_dir, "progress.csv")).readline() else "\t"):
            rows.append({c: float(r[c]) for c in COLUMNS if c in r and r[c] != ""})
    return dict(seed=seed, rows=rows, wall_s=time.time() - t0, hyper=hyper,
                obs_dim=int(alg.env.observation_space.shape[0]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--first-seed", type=int, default=0)
    ap.add_argument("--merge", action="store_true", help="append to the seeds already in --out")
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--steps-per-epoch", type=int, default=32 * 1000)
    ap.add_argument("--workers", type=int, default=5)
    ap.add_argument("--env", default=ENV_ID)
    ap.add_argument("--out", default=os.path.join(HERE, "..", "..", "tests", "golden", "learning_curve.json"))
    a = ap.parse_args()
    jobs = [(s, a.epochs, a.steps_per_epoch, a.env) for s in range(a.first_seed, a.first_seed + a.seeds)]
    with mp.get_context("spawn").Pool(min(a.workers, len(jobs))) as pool:
        res = pool.map(run_seed, jobs)
    old = None
    if a.merge and os.path.exists(a.out):
        with open(a.out) as f:
            old = json.load(f)
        assert old["epochs"] == a.epochs and old["steps_per_epoch"] == a.steps_per_epoch and old["env_id"] == a.env
    out = dict(
        what="per-epoch log of the reference's ProximalPolicyOptimizationAlgorithm.learn() on its own env, one entry per "
             "seed.  STATISTICAL SAMPLE of the trainer's run distribution, not a known answer: use it for two-sample tests "
             "only.  A run is a function of its seed since numpy is seeded before the env is built (the reference seeds it "
             "after the env's constructor has drawn from it)",
        generator="oracle/refgen/gen_golden_learning.py", env_id=a.env, epochs=a.epochs, steps_per_epoch=a.steps_per_epoch,
        obs_dim=res[0]["obs_dim"], hyper=res[0]["hyper"], columns=COLUMNS,
        seeds=[r["seed"] for r in res], wall_s=[round(r["wall_s"], 1) for r in res],
        curves={str(r["seed"]): {c: [row.get(c) for row in r["rows"]] for c in COLUMNS} for r in res})
    if old is not None:
        out["seeds"] = old["seeds"] + out["seeds"]
        out["wall_s"] = old["wall_s"] + out["wall_s"]
        out["curves"] = {**old["curves"], **out["curves"]}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", a.out, os.path.getsize(a.out) // 1024, "kB")
    for r in res:
        print("seed", r["seed"], "wall", round(r["wall_s"]), "s  EpRet", [round(x["EpRet/Mean"], 1) for x in r["rows"]][::3],
              " EpLen", [round(x["EpLen/Mean"]) for x in r["rows"]][::3])


if __name__ == "__main__":
    main()
