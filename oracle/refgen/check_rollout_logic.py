#!/usr/bin/env python3
"""Deterministic pin of the trainer's ROLLOUT logic: `ppo.PPOTrainer.roll_out` against
`IWPGAlgorithm.roll_out` (algs/iwpg/iwpg.py:350-385) on the reference's own env, element for element.

Runs in the build container only (imports /root/reference like the other generators; test infrastructure, writes data only).

Pass A -- the reference: `ProximalPolicyOptimizationAlgorithm` for EPOCHS epochs of `learn_one_epoch` (noise anneal, roll_out,
update, LambdaLR step) with recorders on the env, the Buffer, the logger and np.random.shuffle.  Recorded per epoch: the rollout
buffers, every transition (the observation `next_o` the step returned, the terminated flag, the observation of the reset that
followed a path end), every `finish_path(last_val)` call, the `EpRet` / `EpLen` values the logger received, the Buffer's adv /
target_v / discounted_ret, torch's generator state in front of the rollout, the shuffles, the state_dict after the update.

Pass B -- this repo's trainer: the SAME construction is repeated (same seeds => the same env object state, the same initial
state_dict and the same generator states: asserted), then `PPOTrainer` (PyTorch-op path, CPU tensors, `reset_each_rollout`)
drives the reference's env through a one-env adapter.  torch's CPU `Normal.sample()` with the same generator state gives the
same standard normals, so epoch 0's buffers must equal pass A's element for element, and after the replayed update epoch 1's.
Two things the adapter does to keep the two random streams aligned (neither changes a distribution):
  * the reference resets the env twice between two epochs -- once after the epoch-end cut (iwpg.py:385, in front of the update,
    whose shuffles draw from the same numpy stream), once at the top of the next roll_out (iwpg.py:353); `PPOTrainer` once.
    The adapter resets the env behind the last step of a rollout as well.
  * the reference's bootstrap call `self.ac(o)` (iwpg.py:376) is `ActorCritic.step`, which in training mode SAMPLES an action it
    throws away: 4 normal draws per cut path.  The adapter burns the same 4 draws when a path is cut.

Three scenarios: the registered env (TimeLimit 500: paths end by termination or at the epoch end); the same env class under a
TimeLimit of 12 steps (the reference reads `_max_episode_steps`, iwpg.py:79-80), where paths are cut by the limit every few
episodes and "terminated AND cut on the same step" occurs; and a seed whose epoch 0 ends on a terminated step.

Output: tests/golden/rollout.npz (pass A's records; tests/test_trainer.py replays them through `_roll_out_eager` on a
recorded-transition env) and a report of pass B on stdout (profiles/r06_rollout_logic.txt).

usage: check_rollout_logic.py [--out tests/golden/rollout.npz] [--no-write]
"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, os.path.join(HERE, "standins"))
sys.path.insert(0, "/root/reference")
_tb = types.ModuleType("torch.utils.tensorboard")
_tb.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = _tb

ENV_ID = "DroneHoverSimpleEnv-v0"
EPOCHS_TOTAL, EPOCHS_RUN, STEPS, MINI, V_ITERS, PI_ITERS = 8, 3, 1000, 4, 2, 10
# name -> (TimeLimit override, seed); seed 8 is the first whose epoch 0 ends on a TERMINATED step (the reference bootstraps that
# path with V(o) like any other epoch-end cut, iwpg.py:374-379)
SCENARIOS = {"limit500": (None, 3), "limit12": (12, 3), "limit500_term_at_cut": (None, 8)}
REPLAY_BUFFERS = False  # --full-size --replay-buffers


def build_reference(limit, seed=3, np_seed=20261003):
    """The reference's algorithm object, constructed the same way every time (see gen_golden_update.py for the numpy seed)."""
    import phoenix_drone_simulation  # noqa: F401  (registers the env ids)
    import gymnasium as gym
    from gymnasium.envs.registration import TimeLimit
    from phoenix_drone_simulation.algs.ppo import ppo
    from phoenix_drone_simulation.utils import utils
    np.random.seed(np_seed)
    log_dir = tempfile.mkdtemp(prefix="ref_rollout_")
    kw = utils.get_defaults_kwargs(alg="ppo", env_id=ENV_ID)
    kw.update(epochs=EPOCHS_TOTAL, steps_per_epoch=STEPS, seed=seed, verbose=False, save_freq=10 ** 9, num_mini_batches=MINI,
              train_v_iterations=V_ITERS, train_pi_iterations=PI_ITERS,
              logger_kwargs=dict(log_dir=log_dir, exp_name="golden", level=0, use_tensor_board=False, verbose=False))
    env = ENV_ID
    if limit is not None:  # an env INSTANCE (iwpg.py:74-77) of the same class under a shorter TimeLimit
        env = TimeLimit(gym.make(ENV_ID).unwrapped, limit)
    return ppo.ProximalPolicyOptimizationAlgorithm(env_id=env, **kw)


# ---- pass A ----------------------------------------------------------------------------------------------------------------
def record_reference(limit, seed):
    alg = build_reference(limit, seed)
    out = dict(steps=np.int64(STEPS), epochs_total=np.int64(EPOCHS_TOTAL), epochs_run=np.int64(EPOCHS_RUN),
               num_mini_batches=np.int64(MINI), train_v_iterations=np.int64(V_ITERS), train_pi_iterations=np.int64(PI_ITERS),
               max_ep_len=np.int64(alg.max_ep_len), obs_dim=np.int64(alg.env.observation_space.shape[0]),
               gamma=np.float64(alg.buf.gamma), lam=np.float64(alg.buf.lam), pi_lr=np.float64(alg.pi_lr),
               vf_lr=np.float64(alg.vf_lr), clip_ratio=np.float64(alg.clip_ratio))
    for k, v in alg.ac.state_dict().items():
        out["sd_init__" + k] = v.numpy().copy()
    states = dict(np0=np.random.get_state(), torch0=torch.get_rng_state().clone())

    rec = dict(paths=[], steps=[], resets=[], shuffles=[], ep=[])
    finish, step, reset, store, shuffle = alg.buf.finish_path, alg.env.step, alg.env.reset, alg.logger.store, np.random.shuffle

    def finish_path(last_val=0):
        rec["paths"].append((alg.buf.ptr, float(np.asarray(last_val).reshape(-1)[0])))
        return finish(last_val)

    def env_step(a):
        r = step(a)
        rec["steps"].append((np.asarray(r[0], np.float32).copy(), float(r[1]), bool(r[2]), bool(r[3])))
        return r

    def env_reset(**kw):
        r = reset(**kw)
        rec["resets"].append((len(rec["steps"]), np.asarray(r[0], np.float32).copy()))
        return r

    def logger_store(**kw):
        if "EpRet" in kw:
            rec["ep"].append((len(rec["steps"]), float(kw["EpRet"]), int(kw["EpLen"])))
        return store(**kw)

    def rec_shuffle(x):
        shuffle(x)
        rec["shuffles"].append(np.array(x, dtype=np.int64).copy())
    alg.buf.finish_path, alg.env.step, alg.env.reset, alg.logger.store = finish_path, env_step, env_reset, logger_store
    np.random.shuffle = rec_shuffle
    try:
        for e in range(EPOCHS_RUN):
            alg.epoch = e
            for v in rec.values():
                v.clear()
            out[f"e{e}_lr"] = np.float64(alg.pi_optimizer.param_groups[0]["lr"])
            alg.ac.update(frac=e / alg.epochs)
            out[f"e{e}_log_std"] = alg.ac.pi.log_std.detach().numpy().copy()
            out[f"e{e}_torch_rng"] = torch.get_rng_state().numpy().copy()
            alg.roll_out()
            b = alg.buf
            for name in ("obs_buf", "act_buf", "rew_buf", "val_buf", "logp_buf", "adv_buf", "target_val_buf",
                         "discounted_ret_buf"):
                out[f"e{e}_{name}"] = getattr(b, name).copy()
            assert len(rec["steps"]) == STEPS and rec["paths"][-1][0] == STEPS
            out[f"e{e}_path_end"] = np.array([p for p, _ in rec["paths"]], dtype=np.int64)
            out[f"e{e}_path_last_val"] = np.array([v for _, v in rec["paths"]], dtype=np.float32)
            # what env.step returned on the last step of each path (pre-reset; as f32: the trainer's cast, iwpg.py:356-357, 376);
            # on every other step it is the next row of obs_buf (asserted)
            nxt = np.stack([s[0] for s in rec["steps"]])
            ends = out[f"e{e}_path_end"] - 1
            inner = np.setdiff1d(np.arange(STEPS - 1), ends)
            assert np.array_equal(nxt[inner], b.obs_buf[inner + 1])
            out[f"e{e}_end_obs"] = nxt[ends]
            out[f"e{e}_step_rew"] = np.array([s[1] for s in rec["steps"]], np.float64)
            out[f"e{e}_terminated"] = np.array([s[2] for s in rec["steps"]], np.uint8)
            out[f"e{e}_env_truncated"] = np.array([s[3] for s in rec["steps"]], np.uint8)  # the wrapper's flag (ignored by iwpg)
            out[f"e{e}_reset_at"] = np.array([k for k, _ in rec["resets"]], np.int64)   # number of steps taken before the reset
            out[f"e{e}_reset_obs"] = np.stack([o for _, o in rec["resets"]])
            out[f"e{e}_ep_at"] = np.array([k for k, _, _ in rec["ep"]], np.int64)
            out[f"e{e}_ep_ret"] = np.array([r for _, r, _ in rec["ep"]], np.float64)
            out[f"e{e}_ep_len"] = np.array([n for _, _, n in rec["ep"]], np.int64)
            alg.update()
            out[f"e{e}_shuffles"] = np.stack(rec["shuffles"])
            out[f"e{e}_loss_pi"] = np.float64(alg.loss_pi_before)
            out[f"e{e}_loss_v"] = np.float64(alg.loss_v_before)
            for k, v in alg.ac.state_dict().items():
                out[f"e{e}_sd_after__" + k] = v.numpy().copy()
            alg.scheduler.step()
    finally:
        np.random.shuffle = shuffle
    return out, states


# ---- pass B ----------------------------------------------------------------------------------------------------------------
def load_ppo():
    """The package's ppo.py without its __init__ (which loads the HIP library); pds_gae restated with torch ops."""
    import importlib.util
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    pkg = types.ModuleType("pds_amd")
    pkg.__path__ = [os.path.join(ROOT, "phoenix-drone-simulation_amd")]
    sys.modules["pds_amd"] = pkg
    for name in ("native", "fused", "ppo"):
        spec = importlib.util.spec_from_file_location(f"pds_amd.{name}", os.path.join(pkg.__path__[0], f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"pds_amd.{name}"] = mod
        if name == "ppo":
            spec.loader.exec_module(mod)
        elif name == "fused":
            for fn in ("counter_add", "gaussian_sample", "random_permutation", "rollout_record"):
                setattr(mod, fn, None)
    ppo = sys.modules["pds_amd.ppo"]
    import golden_util as gu
    ppo.gae = gu.gae_torch
    return ppo


class LiveAdapter:
    """ONE reference env behind the DroneVecEnv surface `PPOTrainer` uses (auto-reset + final_obs + both flags)."""

    def __init__(self, env, max_ep_len, steps):
        self.env, self.max_ep_len, self.T = env, int(max_ep_len), int(steps)
        self.num_envs, self.device, self.act_dim, self.env_id_base = 1, torch.device("cpu"), 4, 0
        self.obs_dim = int(env.observation_space.shape[0])
        self.ep_len, self.t, self.dry = 0, 0, True

    def reset(self):
        if self.dry:  # PPOTrainer's constructor: leave the env and its random stream alone
            return torch.zeros(1, self.obs_dim), {}
        o, _ = self.env.reset()
        self.ep_len, self.t = 0, 0
        return torch.as_tensor(o, dtype=torch.float32)[None], {}

    def step(self, a):
        o, r, te, _, _ = self.env.step(a.numpy()[0])
        self.ep_len += 1
        self.t += 1
        tr = self.ep_len == self.max_ep_len  # iwpg.py:371: the trainer's own count
        fin = o
        if tr or self.t == self.T:
            torch.normal(torch.zeros(4), torch.ones(4))  # the action `self.ac(o)` samples and drops (iwpg.py:376)
        if te or tr or self.t == self.T:
            o, _ = self.env.reset()  # (iwpg.py:385: also behind the epoch-end cut, in front of the update's shuffles)
            self.ep_len = 0
            if self.t == self.T:
                o = fin  # PPOTrainer bootstraps the cut path from the observation it is handed back
        f = lambda x, dt=torch.float32: torch.as_tensor(np.asarray(x), dtype=dt)[None]  # noqa: E731
        return f(o), f(r), f(te, torch.bool), f(tr, torch.bool), {"final_obs": f(fin)}


def buffers_of(tr, ppo):
    """PPOTrainer's rollout of one epoch -> the reference Buffer's arrays (+ GAE outputs with the current reward scale)."""
    scale = float(1.0 / (tr.ac.ret_oms.std.item() + tr.ac.ret_oms.eps))
    adv, tv, dr = ppo.gae(tr.rew_buf, tr.val_buf, tr.term_buf, tr.trunc_buf, tr.fval_buf, tr.last_val, tr.gamma, tr.lam,
                          scale, float(tr.ac.ret_oms.bound))
    n = lambda x: x.reshape(x.shape[0], -1).squeeze(-1).numpy().copy()  # noqa: E731
    return dict(obs_buf=n(tr.obs_buf), act_buf=n(tr.act_buf), rew_buf=n(tr.rew_buf), val_buf=n(tr.val_buf),
                logp_buf=n(tr.logp_buf), adv_buf=n(adv), target_val_buf=n(tv), discounted_ret_buf=n(dr))


def run_ours_live(limit, seed, g, states, ppo):
    alg = build_reference(limit, seed)
    assert all(np.array_equal(a, b) for a, b in zip(np.random.get_state()[1:2], states["np0"][1:2])), "numpy stream differs"
    assert torch.equal(torch.get_rng_state(), states["torch0"]), "torch stream differs"
    env = LiveAdapter(alg.env, alg.max_ep_len, STEPS)
    t_state = torch.get_rng_state()
    tr = ppo.PPOTrainer(env, rollout_len=STEPS, epochs=EPOCHS_TOTAL, gamma=float(g["gamma"]), lam=float(g["lam"]),
                        train_pi_iterations=PI_ITERS, train_v_iterations=V_ITERS, num_mini_batches=MINI, seed=seed, fused=False,
                        reset_each_rollout=True)
    # PPOTrainer's constructor seeded torch and initialised its own networks (its reset was a dry one): the first real reset
    # is the one at the top of the reference's roll_out; take the reference's generator state and weights
    env.dry = False
    tr.obs, _ = env.reset()
    torch.set_rng_state(t_state)
    with torch.no_grad():
        for k, p_ in tr.ac.state_dict().items():
            p_.copy_(torch.as_tensor(g["sd_init__" + k]))
    tr.perm_fn = lambda B: torch.as_tensor(_np_shuffle(tr, B))
    report = []
    for e in range(EPOCHS_RUN):
        assert np.array_equal(torch.get_rng_state().numpy(), g[f"e{e}_torch_rng"]) or e > 0, "generator state differs"
        if tr.use_exploration_noise_anneal:
            tr.ac.update(frac=tr.epoch / tr.epochs)
        stats = tr.roll_out().tolist()
        got = buffers_of(tr, ppo)
        row = dict(epoch=e)
        for name, arr in got.items():
            want = g[f"e{e}_{name}"]
            row[name] = (float(np.max(np.abs(arr - want))), bool(np.array_equal(arr, want)))
        n_ep = len(g[f"e{e}_ep_len"])
        row["episodes"] = (stats[2], n_ep)
        row["ep_len_sum"] = (stats[1], int(g[f"e{e}_ep_len"].sum()))
        row["ep_ret_sum"] = (stats[0], float(g[f"e{e}_ep_ret"].sum()))
        report.append(row)
        if REPLAY_BUFFERS:  # the update from the REFERENCE's buffers (isolates update() from the rollout's 1e-6 differences)
            T = STEPS
            f = lambda name: torch.as_tensor(g[f"e{e}_{name}"])  # noqa: E731
            tr.obs_buf.copy_(f("obs_buf")[:, None]); tr.act_buf.copy_(f("act_buf")[:, None]); tr.rew_buf.copy_(f("rew_buf")[:, None])
            tr.val_buf.copy_(f("val_buf")[:, None]); tr.logp_buf.copy_(f("logp_buf")[:, None])
            got2 = buffers_of(tr, ppo)
            row["gae_from_reference_buffers"] = tuple(float(np.max(np.abs(got2[k] - g[f"e{e}_{k}"]))) for k in ("adv_buf", "target_val_buf", "discounted_ret_buf"))
        tr.update()
        diffs = {k: float(np.max(np.abs(p_.numpy() - g[f"e{e}_sd_after__" + k]))) for k, p_ in tr.ac.state_dict().items()}
        worst = max(diffs.values())
        row["state_dict_after_update"] = worst
        row["state_dict_worst_keys"] = sorted(diffs.items(), key=lambda kv: -kv[1])[:3]
        if REPLAY_BUFFERS:  # continue from the reference's parameters: every epoch is then an independent check
            with torch.no_grad():
                for k, p_ in tr.ac.state_dict().items():
                    p_.copy_(torch.as_tensor(g[f"e{e}_sd_after__" + k]))
        tr.scheduler.step()
        tr.epoch += 1
    return report


def _np_shuffle(tr, B):
    """update_value_net's index array (iwpg.py:463-466): ONE arange per update, shuffled in place again and again."""
    key = ("_idx", tr.epoch)
    if getattr(tr, "_shuffle_key", None) != key:
        tr._shuffle_key, tr._shuffle_idx = key, np.arange(B)
    np.random.shuffle(tr._shuffle_idx)
    return tr._shuffle_idx.copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "rollout.npz"))
    ap.add_argument("--no-write", action="store_true")
    ap.add_argument("--full-size", action="store_true",
                    help="the reference's own epoch (32 000 steps, 16 mini-batches x 5 value iterations, 80 policy iterations) instead of "
                         "the fixture's small one; first scenario only, implies --no-write (a few minutes per epoch)")
    ap.add_argument("--replay-buffers", action="store_true",
                    help="with --full-size: run update() from the REFERENCE's rollout buffers and re-synchronise the parameters after each "
                         "epoch (Adam amplifies the rollout's 1e-6 differences over 80 iterations otherwise)")
    ap.add_argument("--epochs-run", type=int, default=0, help="with --full-size: epochs to compare (default 3)")
    ap.add_argument("--seed", type=int, default=3, help="with --full-size: the run's seed (without --replay-buffers the two programs then run freely from the "
                    "same start with aligned random streams: their per-epoch EpLen differ by what 80 Adam steps make of rounding, nothing else)")
    a = ap.parse_args()
    if a.full_size:
        global STEPS, MINI, V_ITERS, PI_ITERS, EPOCHS_TOTAL, SCENARIOS, REPLAY_BUFFERS
        REPLAY_BUFFERS = a.replay_buffers
        if a.epochs_run:
            global EPOCHS_RUN
            EPOCHS_RUN = a.epochs_run
        STEPS, MINI, V_ITERS, PI_ITERS, EPOCHS_TOTAL = 32000, 16, 5, 80, 40
        SCENARIOS = {"limit500_full_size": (None, a.seed)}
        a.no_write = True
    torch.set_num_threads(1)
    ppo = load_ppo()
    out, failed = {}, []
    for name, (limit, seed) in SCENARIOS.items():
        g, states = record_reference(limit, seed)
        cuts = [int(np.sum((g[f"e{e}_path_last_val"] != 0))) for e in range(EPOCHS_RUN)]
        both = 0
        for e in range(EPOCHS_RUN):
            ends = g[f"e{e}_path_end"] - 1
            both += int(np.sum((g[f"e{e}_terminated"][ends] != 0) & (g[f"e{e}_path_last_val"] != 0)))
        last_term = [int(g[f"e{e}_terminated"][-1]) for e in range(EPOCHS_RUN)]
        print(f"[{name}] reference: terminated on the epoch's last step (bootstrapped with V all the same)", last_term)
        print(f"[{name}] reference: max_ep_len {int(g['max_ep_len'])}, paths per epoch",
              [len(g[f"e{e}_path_end"]) for e in range(EPOCHS_RUN)], "bootstrapped with V", cuts,
              "terminated AND cut", both, "episodes logged", [len(g[f"e{e}_ep_len"]) for e in range(EPOCHS_RUN)])
        for row in run_ours_live(limit, seed, g, states, ppo):
            print(f"[{name}] PPOTrainer on the live reference env, epoch {row['epoch']}:")
            for k, v in row.items():
                if k != "epoch":
                    print(f"    {k:24s} {v}")
            bad = [k for k, v in row.items() if k.endswith("_buf") and v[0] > 1e-4] + \
                  [k for k in ("episodes", "ep_len_sum") if row[k][0] != row[k][1]] + \
                  (["state_dict_after_update"] if row["state_dict_after_update"] > 1e-4 else [])
            if bad:
                failed.append((name, row["epoch"], bad))
        for k, v in g.items():
            out[f"{name}__{k}"] = v
    print("RESULT:", "every buffer, flag, episode statistic and post-update parameter equal to rounding (<= 1e-4 abs)"
          if not failed else f"MISMATCH {failed}")
    if failed:
        sys.exit(1)
    if not a.no_write:
        np.savez_compressed(a.out, **out)
        print("wrote", a.out, os.path.getsize(a.out) // 1024, "kB;", len(out), "arrays")


if __name__ == "__main__":
    main()
