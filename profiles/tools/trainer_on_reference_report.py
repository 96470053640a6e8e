#!/usr/bin/env python3
"""profiles/r06_trainer_on_reference_env.txt: `ppo.PPOTrainer`'s PyTorch path driving the REFERENCE's own
env (oracle/refgen/bisect_trainer_on_reference_env.py, 1 env x 32 000 steps, 40 epochs; CPU, build container) against the reference's
own `learn()` runs (tests/golden/learning_curve.json) -- the one comparison in which the env is identical and only the trainer
differs.  Per phase of the curve: per-seed phase means, Welch's t-test.  usage: trainer_on_reference_report.py runs.json [more.json]"""
import json
import os
import sys

import numpy as np
from scipy import stats

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu  # noqa: E402


def load_ref(path):
    d = json.load(open(path))
    seeds = [str(s) for s in d["seeds"]]
    return np.array([d["curves"][s]["EpLen/Mean"] for s in seeds]), np.array([d["curves"][s]["EpRet/Mean"] for s in seeds])


def main():
    L, R = load_ref(os.path.join(ROOT, "tests", "golden", "learning_curve.json"))
    for path in sys.argv[1:]:
        B = json.load(open(path))
        X = np.array([r["ep_len"] for r in B["runs"]])
        XR = np.array([r["ep_ret"] for r in B["runs"]])
        print(f"== {os.path.basename(path)}: {len(X)} runs (seeds {B['runs'][0]['seed']}..{B['runs'][-1]['seed']}), {B['envs']} env x {B['steps']} steps, "
              f"{B['epochs']} epochs, {np.mean([r['wall_s'] for r in B['runs']]) / 60:.0f} min per run; reference: {len(L)} runs")
        for what, a, b in (("EpLen", X, L), ("EpRet", XR, R)):
            fails, rep = gu.compare_learning_curves(a, b)
            for phase, v in rep.items():
                if isinstance(v, dict) and "mean_a" in v:
                    print(f"  {what} {phase:22s} trainer-on-reference-env {v['mean_a']:9.2f}   reference {v['mean_b']:9.2f}   "
                          f"t = {v['t']:+.2f}   Welch p = {v['p']:.3f}")
            print(f"  {what} per epoch: {rep['per epoch']};  sign count epochs 20-40 (reported only): "
                  f"{rep['sign count epochs 20-40 (reported only)']};  failed comparisons at p > 0.01: {len(fails)}")
        print("  epoch:      " + " ".join(f"{e:6d}" for e in range(1, X.shape[1] + 1)))
        print("  trainer:    " + " ".join(f"{v:6.1f}" for v in X.mean(0)))
        print("  reference:  " + " ".join(f"{v:6.1f}" for v in L.mean(0)))
        se = np.sqrt(X.var(0, ddof=1) / len(X) + L.var(0, ddof=1) / len(L))
        print("  diff / SE:  " + " ".join(f"{v:+6.2f}" for v in (X.mean(0) - L.mean(0)) / se))
        t, p = stats.ttest_ind(X[:, 9:19].mean(1), L[:, 9:19].mean(1), equal_var=False)
        print(f"  epochs 10-19 (where round 5's HIP sample sat below the reference in all ten): {X[:, 9:19].mean():.2f} vs {L[:, 9:19].mean():.2f}, "
              f"t = {t:+.2f}, p = {p:.3f}")


if __name__ == "__main__":
    main()
