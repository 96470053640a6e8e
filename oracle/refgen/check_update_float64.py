#!/usr/bin/env python3
"""The trainer's UPDATE at the reference's full epoch size, compared in float64 (round 6, behind the seed-paired learning curves).

check_rollout_logic.py --full-size finds the parameters after an update 1e-3 .. 5e-3 apart: 80 Adam steps on a clipped objective
amplify float32 rounding to that level (profiles/r06_update_sensitivity.txt), which could hide a small systematic difference -- an
iteration more or less, another mini-batch order, a learning rate a few per cent off.  Here BOTH updates run in float64 on the SAME
data: the reference's `IWPGAlgorithm.update` (algs/iwpg/iwpg.py:398-485; its actor-critic cast with `.double()`, `Buffer.get`
returning the recorded float32 batch as float64) and `PPOTrainer.update` (PyTorch-op path, the same cast, the same batch, the
reference's advantages / value targets handed in, numpy's shuffle stream re-seeded for each).  Rounding is then 1e-16 instead of
6e-8, and what is left between the two is what the two programs do differently.  Both are also run in float32 from the same start,
for the contrast.  Build container only (imports /root/reference); test infrastructure, prints a report
(profiles/r06_update_float64.txt).

usage: check_update_float64.py [--epochs 2] [--teeth]"""
import argparse
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import check_rollout_logic as crl  # noqa: E402


def run_reference_update(alg, raw, dtype, seed, start, opt_start):
    """alg.update() from `raw` (the Buffer's batch) in `dtype`, from the parameters `start` and the optimiser states `opt_start`
    -> state_dict after (as float64)."""
    alg.ac.float()
    alg.ac.load_state_dict(start)
    if dtype == torch.float64:
        alg.ac.double()
    alg.pi_optimizer.load_state_dict(copy.deepcopy(opt_start[0]))  # (casts Adam's moments to the parameters' type)
    alg.vf_optimizer.load_state_dict(copy.deepcopy(opt_start[1]))
    data = {k: v.to(dtype) for k, v in raw.items()}
    alg.buf.get = lambda: {k: v.clone() for k, v in data.items()}
    np.random.seed(seed)
    alg.update()
    return {k: v.detach().double().clone() for k, v in alg.ac.state_dict().items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--teeth", action="store_true", help="control: PPOTrainer with ONE policy iteration less and a learning rate 1 %% higher -- the comparison must see it")
    a = ap.parse_args()
    crl.STEPS, crl.MINI, crl.V_ITERS, crl.PI_ITERS, crl.EPOCHS_TOTAL = 32000, 16, 5, 80, 40
    torch.set_num_threads(1)
    ppo = crl.load_ppo()
    seed = 3
    alg = crl.build_reference(None, seed)
    print(f"{crl.ENV_ID}, env defaults, {crl.STEPS} steps per epoch, {crl.PI_ITERS} policy iterations, {crl.V_ITERS} x {crl.MINI} value mini-batches; float64 = both programs cast, float32 = as shipped")
    for e in range(a.epochs):
        alg.epoch = e
        alg.ac.update(frac=e / alg.epochs)
        alg.roll_out()
        raw = alg.buf.get()  # (float32 tensors: obs, act, target_v, adv, log_p, discounted_ret, ...)
        lr_pi, lr_v = alg.pi_optimizer.param_groups[0]["lr"], alg.vf_optimizer.param_groups[0]["lr"]
        start = copy.deepcopy(alg.ac.state_dict())
        opt_start = (copy.deepcopy(alg.pi_optimizer.state_dict()), copy.deepcopy(alg.vf_optimizer.state_dict()))
        out = {}
        for name, dtype in (("float64", torch.float64), ("float32", torch.float32)):
            # ---- the reference
            ref = alg
            ref_sd = run_reference_update(alg, raw, dtype, 1234 + e, start, opt_start)
            # ---- PPOTrainer.update on the same batch
            env = crl.LiveAdapter(ref.env, ref.max_ep_len, crl.STEPS)
            tr = ppo.PPOTrainer(env, rollout_len=crl.STEPS, epochs=crl.EPOCHS_TOTAL, gamma=float(alg.buf.gamma), lam=float(alg.buf.lam),
                                train_pi_iterations=crl.PI_ITERS - (1 if a.teeth and name == 'float64' else 0), train_v_iterations=crl.V_ITERS, num_mini_batches=crl.MINI, seed=seed,
                                fused=False, reset_each_rollout=True)
            tr.ac.load_state_dict(start)
            # Adam's moments of the reference's optimisers (its parameter lists differ from ours: matched by name)
            for opt, ref_opt, ref_sd_opt in ((tr.pi_opt, alg.pi_optimizer, opt_start[0]), (tr.vf_opt, alg.vf_optimizer, opt_start[1])):
                ref_params = [p_ for g_ in ref_opt.param_groups for p_ in g_["params"]]
                ref_name = {id(p_): n_ for n_, p_ in alg.ac.named_parameters()}
                ours = dict(tr.ac.named_parameters())
                for idx, st in ref_sd_opt["state"].items():
                    opt.state[ours[ref_name[id(ref_params[idx])]]] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
            for g_ in tr.pi_opt.param_groups:
                g_["lr"] = lr_pi
            for g_ in tr.vf_opt.param_groups:
                g_["lr"] = lr_v * (1.01 if a.teeth and name == "float32" else 1.0)
            tr.epoch = e
            if dtype == torch.float64:
                tr.ac.double()
                for opt in (tr.pi_opt, tr.vf_opt):  # Adam's moments follow their parameters' type
                    for st in opt.state.values():
                        for k, v in st.items():
                            if torch.is_tensor(v) and v.is_floating_point() and v.dim() > 0:
                                st[k] = v.double()
            T = crl.STEPS
            tr.obs_buf = raw["obs"].to(dtype).reshape(T, 1, -1).clone()
            tr.act_buf = raw["act"].to(dtype).reshape(T, 1, -1).clone()
            tr.logp_buf = raw["log_p"].to(dtype).reshape(T, 1).clone()
            handed = tuple(raw[k].to(dtype).reshape(T, 1).clone() for k in ("adv", "target_v", "discounted_ret"))
            tr.last_val = torch.zeros(1, dtype=dtype)  # (unused: the advantages are handed in)
            ppo.gae = lambda *args, **kw: handed  # the reference's own advantages / targets / discounted returns
            tr.perm_fn = lambda B: torch.as_tensor(crl._np_shuffle(tr, B))
            np.random.seed(1234 + e)
            tr.update()
            ours_sd = {k: v.detach().double().clone() for k, v in tr.ac.state_dict().items()}
            diffs = {k: float((ours_sd[k] - ref_sd[k]).abs().max()) for k in ref_sd if ref_sd[k].is_floating_point()}
            moved = {k: float((ref_sd[k] - start[k].double()).abs().max()) for k in ref_sd if ref_sd[k].is_floating_point()}
            worst = sorted(diffs.items(), key=lambda kv: -kv[1])[:3]
            out[name] = (max(diffs.values()), worst, max(moved[k] for k in moved if k.startswith(("pi.net", "v.net"))))
            print(f"epoch {e} {name}: max |PPOTrainer - reference| over the state_dict after the update = {out[name][0]:.3e}  (worst: "
                  + ", ".join(f"{k} {v:.1e}" for k, v in worst) + f"); the update itself moved the weights by up to {out[name][2]:.3e}", flush=True)
        # continue the reference's own run in float32 (what its learn() does)
        run_reference_update(alg, raw, torch.float32, 4321 + e, start, opt_start)
        alg.scheduler.step()
        del alg.buf.get  # the class method again
    print("RESULT: in float64 the two updates agree to", "rounding" if out["float64"][0] < 1e-9 else f"{out['float64'][0]:.1e}")


if __name__ == "__main__":
    main()
