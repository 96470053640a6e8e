#!/usr/bin/env python3
"""profiles/r05_learning_curve.txt: the reference's learning-curve sample (tests/golden/learning_curve.json) against the HIP runs of
profiles/tools/learning_bisect.py (gpurun_out/r05_bisect{1,2}.json) and the trainer-on-reference-env runs.  CPU only."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu  # noqa: E402


def load_ref(path):
    d = json.load(open(path))
    seeds = [str(s) for s in d["seeds"]]
    return (np.array([d["curves"][s]["EpLen/Mean"] for s in seeds]), np.array([d["curves"][s]["EpRet/Mean"] for s in seeds]))


def main():
    L, R = load_ref(os.path.join(ROOT, "tests", "golden", "learning_curve.json"))
    L0, R0 = load_ref(os.path.join(ROOT, "profiles", "r05_reference_runs_unseeded_env_ctor.json"))
    b1 = json.load(open(os.path.join(ROOT, "gpurun_out", "r05_bisect1.json")))
    b2 = json.load(open(os.path.join(ROOT, "gpurun_out", "r05_bisect2.json")))
    B = json.load(open(os.path.join(ROOT, "profiles", "r05_trainer_on_reference_env.json")))
    cat = lambda k, key, *src: np.array(sum([s[k][key] for s in src], []))  # noqa: E731
    runs = {
        "reference, the EARLIER sample of 24 runs (env built before numpy is seeded, as the reference does: not reproducible)": (L0, R0),
        "HIP 1 x 32000 (fused; the reference layout)": (cat("n1", "len", b1, b2), cat("n1", "ret", b1, b2)),
        "HIP 8 x 4000 (fused; round 4's layout)": (cat("base", "len", b1, b2), cat("base", "ret", b1, b2)),
        "HIP 8 x 4000 per-step kernels (bit-identical to the fused runs of the same seeds)": (cat("perstep", "len", b1), cat("perstep", "ret", b1)),
        "HIP 8 x 4000 PyTorch ops (autograd, torch Adam, randperm)": (cat("torchops", "len", b1), cat("torchops", "ret", b1)),
        "HIP 32 x 1000": (cat("n32", "len", b1), cat("n32", "ret", b1)),
        "HIP 64 x 500": (cat("n64", "len", b2), cat("n64", "ret", b2)),
        "HIP 1 x 32000, no sensor noise (another env: for the layout comparison only)": (cat("n1_quiet", "len", b2), cat("n1_quiet", "ret", b2)),
        "HIP 8 x 4000, no sensor noise (another env: for the layout comparison only)": (cat("n8_quiet", "len", b2), cat("n8_quiet", "ret", b2)),
        "PPOTrainer's PyTorch path on the REFERENCE's envs, 1 x 32000 (CPU, oracle/refgen/bisect_trainer_on_reference_env.py)":
            (np.array([r["ep_len"] for r in B["runs"]]), np.array([r["ep_ret"] for r in B["runs"]])),
    }
    late = slice(23, 40)
    print("# Round 5: where the round-4 late-epoch offset of the learning-curve pin came from (VERDICT round 4, item 1).")
    print("# DroneHoverSimpleEnv-v0, env defaults, 32 000 steps per epoch, 40 epochs, PPO defaults.  Reference = its own learn(), 24 runs")
    print("# (tests/golden/learning_curve.json: numpy seeded before the env is built, regenerates).  Late level = per-seed mean of EpLen")
    print("# (EpRet) over epochs 24-40; +- = standard error over seeds; diff in standard errors of the difference (Welch); 'fails' = number of")
    print("# failed comparisons of golden_util.compare_learning_curves (4 phases + per-epoch Bonferroni, EpLen and EpRet).")
    print("# HIP seeds 100-111 and 200-211 (profiles/r05_learning_bisect_seeds*.txt).")
    m, mr = L[:, late].mean(1), R[:, late].mean(1)
    print(f"reference, {len(m)} runs: EpLen {m.mean():.2f} +- {m.std(ddof=1) / np.sqrt(len(m)):.2f} (SD {m.std(ddof=1):.1f}, min {m.min():.1f}, max "
          f"{m.max():.1f}); EpRet {mr.mean():.2f} +- {mr.std(ddof=1) / np.sqrt(len(mr)):.2f}")
    m6 = L0[:6, late].mean(1)
    print(f"   (the six runs round 4 compared against, seeds 0-5 of the earlier sample: EpLen {m6.mean():.2f} +- {m6.std(ddof=1) / np.sqrt(6):.2f})")
    for name, (X, XR) in runs.items():
        x, xr = X[:, late].mean(1), XR[:, late].mean(1)
        dd, se = x.mean() - m.mean(), np.sqrt(x.var(ddof=1) / len(x) + m.var(ddof=1) / len(m))
        ddr, ser = xr.mean() - mr.mean(), np.sqrt(xr.var(ddof=1) / len(xr) + mr.var(ddof=1) / len(mr))
        nf = len(gu.compare_learning_curves(X, L)[0]) + len(gu.compare_learning_curves(XR, R)[0])
        print(f"{name}: {len(x)} runs, EpLen {x.mean():.2f} +- {x.std(ddof=1) / np.sqrt(len(x)):.2f} (SD {x.std(ddof=1):.1f}) diff {dd:+.2f} = "
              f"{dd / se:+.2f} SE; EpRet {xr.mean():.2f} diff {ddr:+.2f} = {ddr / ser:+.2f} SE; fails {nf}")
    print()
    print("first peak (epochs 9-16), mean EpLen: reference %.1f | HIP 1 env %.1f | 8 envs %.1f | 32 envs %.1f | 64 envs %.1f" % (
        L[:, 8:16].mean(), runs["HIP 1 x 32000 (fused; the reference layout)"][0][:, 8:16].mean(),
        runs["HIP 8 x 4000 (fused; round 4's layout)"][0][:, 8:16].mean(), runs["HIP 32 x 1000"][0][:, 8:16].mean(),
        runs["HIP 64 x 500"][0][:, 8:16].mean()))
    print()
    print("epoch | reference EpLen mean (SE), 24 runs | earlier reference sample, 24 runs | HIP 1 x 32000, 24 runs | HIP 8 x 4000, 24 runs | trainer on reference envs, 7 runs")
    se = lambda x: x.std(axis=0, ddof=1) / np.sqrt(x.shape[0])  # noqa: E731
    N1 = runs["HIP 1 x 32000 (fused; the reference layout)"][0]
    A = runs["HIP 8 x 4000 (fused; round 4's layout)"][0]
    BB = list(runs.values())[-1][0]
    for e in range(40):
        print(f"{e + 1:5d} | {L[:, e].mean():6.1f} ({se(L)[e]:4.1f}) | {L0[:, e].mean():6.1f} ({se(L0)[e]:4.1f}) | {N1[:, e].mean():6.1f} ({se(N1)[e]:4.1f}) | "
              f"{A[:, e].mean():6.1f} ({se(A)[e]:4.1f}) | {BB[:, e].mean():6.1f} ({se(BB)[e]:4.1f})")


if __name__ == "__main__":
    main()
