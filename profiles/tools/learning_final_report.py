#!/usr/bin/env python3
"""profiles/r06_learning_curve_paired.txt: whole PPO runs at the reference's layout (DroneHoverSimpleEnv-v0, env defaults, 1 env x 32 000
steps x 40 epochs), compared SEED BY SEED.  A run's network initialisation is a function of its seed (torch.manual_seed), and it decides
most of the early learning speed: the reference's two 24-run samples of round 5 share seeds 0-23, so they are not 48 independent runs, and
a comparison against runs at OTHER seeds carries the luck of those 24 initialisations.  Samples: the reference's own learn() (the 24-run
fixture, round 5's 24 runs with the unseeded env constructor -- both seeds 0-23 -- and the chunks of the round's last hours at seeds
100..: profiles/r06_reference_runs_more.json), HIP runs (fused path) at seeds 0-255 and 1000-1191, PPOTrainer's PyTorch path on the
reference's own env at seeds 100-147.  CPU only.  usage: learning_final_report.py [more_reference_runs.json ...]"""
import json
import os
import sys

import numpy as np
from scipy import stats

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PHASES = {"early (epochs 4-8)": slice(3, 8), "first peak (9-16)": slice(8, 16), "dip (17-23)": slice(16, 23), "late (24-40)": slice(23, 40)}


def load_ref(path):
    d = json.load(open(path))
    return {int(s): np.array(d["curves"][str(s)]["EpLen/Mean"]) for s in d["seeds"]}


def se(a):
    a = np.asarray(a)
    return a.std(ddof=1) / np.sqrt(len(a))


def paired(name, ours, ref_by_seed):
    """ours: {seed: curve}; ref_by_seed: {seed: [curves]} -- the per-seed difference ours - mean(reference runs of that seed)"""
    seeds = sorted(set(ours) & set(ref_by_seed))
    if len(seeds) < 4:
        return
    print(f"{name}: {len(seeds)} seeds ({seeds[0]}..{seeds[-1]})")
    for n, sl in PHASES.items():
        x = np.array([ours[s][sl].mean() for s in seeds])
        y = np.array([np.mean([c[sl].mean() for c in ref_by_seed[s]]) for s in seeds])
        d = x - y
        print(f"    {n:20s} ours {x.mean():6.2f}  reference {y.mean():6.2f}  correlation over seeds {np.corrcoef(x, y)[0, 1]:5.2f}  paired difference {d.mean():+6.2f} +- {se(d):.2f}"
              f"  p = {stats.ttest_1samp(d, 0.0).pvalue:.3f}")


def main():
    P = lambda *a: os.path.join(ROOT, *a)  # noqa: E731
    ref = {}
    parts = [("the 24-run fixture (tests/golden/learning_curve.json)", load_ref(P("tests", "golden", "learning_curve.json"))),
             ("round 5's 24 runs with the unseeded env constructor", load_ref(P("profiles", "r05_reference_runs_unseeded_env_ctor.json")))]
    for path in sys.argv[1:]:
        d = json.load(open(path))
        for chunk in (d["chunks"] if "chunks" in d else [d]):
            parts.append((f"round 6, seeds {chunk['seeds'][0]}-{chunk['seeds'][-1]} ({os.path.basename(path)})",
                          {int(s): np.array(chunk["curves"][str(s)]["EpLen/Mean"]) for s in chunk["seeds"]}))
    for _, p in parts:
        for s, c in p.items():
            ref.setdefault(s, []).append(c)
    hip = {}
    d = json.load(open(P("profiles", "r06_hip_runs_seeds0_255.json")))
    hip.update({int(k): np.array(v[0]) for k, v in d.items()})
    d = json.load(open(P("profiles", "r06_hip_runs_192.json")))
    hip.update({int(k): np.array(v[0]) for k, v in d.items()})
    tor = {}
    for f in ("r06_trainer_on_reference_env.json", "r06_trainer_on_reference_env_b.json"):
        tor.update({r["seed"]: np.array(r["ep_len"]) for r in json.load(open(P("profiles", f)))["runs"]})

    print("# Whole PPO runs at the reference's layout (DroneHoverSimpleEnv-v0, env defaults, 1 env x 32 000 steps x 40 epochs), EpLen/Mean, per-seed phase means.")
    print("# Samples of the reference's own learn() (25 CPU-minutes per run, oracle/refgen/gen_golden_learning.py):")
    for name, p in parts:
        v = np.array([c[8:16].mean() for c in p.values()])
        print(f"#   {len(p):3d} runs -- {name}: first peak {v.mean():.2f} +- {se(v):.2f}")
    print(f"# HIP (fused path, 1.5-1.9 s per run, twelve side by side: profiles/tools/learning_threads.py): {len(hip)} runs, seeds 0-255 and 1000-1191")
    print(f"# PPOTrainer's PyTorch path on the REFERENCE's own env (29 CPU-minutes per run, oracle/refgen/bisect_trainer_on_reference_env.py): {len(tor)} runs, seeds 100-147")
    print()
    print("1. The seed decides much of a run: correlation over seeds 0-23 between the reference's two samples AT THE SAME SEEDS (same torch seed = same initial networks and")
    print("   the same sampling stream; they differ in what the env constructor drew before numpy was seeded):")
    a, b = parts[0][1], parts[1][1]
    for n, sl in PHASES.items():
        x = np.array([a[s][sl].mean() for s in range(24)]); y = np.array([b[s][sl].mean() for s in range(24)])
        print(f"    {n:20s} correlation {np.corrcoef(x, y)[0, 1]:5.2f}   SD over seeds {x.std(ddof=1):5.2f} / {y.std(ddof=1):5.2f}   SD of the same-seed difference {(x - y).std(ddof=1):5.2f}")
    print("   -> those 48 runs are 24 initialisations run twice: the standard error of their first-peak mean is that of the 24 per-seed means, not SD / sqrt(48).")
    print()
    print("2. HIP runs over seeds: the seeds the reference samples used against the population")
    for n, sl in PHASES.items():
        v = {s: c[sl].mean() for s, c in hip.items()}
        allv = np.array(list(v.values()))
        line = f"    {n:20s} all {len(allv)} seeds {allv.mean():6.2f} +- {se(allv):.2f}"
        for lo, hi in ((0, 23), (24, 255), (1000, 1191), (100, 147), (100, 195)):
            w = np.array([v[s] for s in range(lo, hi + 1) if s in v])
            line += f" | {lo}-{hi}: {w.mean():6.2f} +- {se(w):.2f}"
        print(line)
    print()
    print("3. Seed by seed (ours - the mean of the reference's runs at that seed):")
    paired("  HIP at the reference's seeds", hip, ref)
    if any(s >= 100 for s in ref):
        paired("  HIP at seeds 0-23 only", {s: c for s, c in hip.items() if s < 24}, ref)
        paired("  HIP at seeds >= 100 only", {s: c for s, c in hip.items() if s >= 100}, ref)
    paired("  PPOTrainer's PyTorch path on the reference's own env", tor, ref)
    print("  ... and the two of ours against each other (same trainer and seeds, the HIP envs against the reference's env: 'reference' = the trainer on the reference's env here)")
    paired("  HIP against PPOTrainer's PyTorch path on the reference's own env", hip, {s: [c] for s, c in tor.items()})
    print()
    print("4. Populations (independent seeds; the reference by per-seed means):")
    rseed = {s: np.mean(cs, axis=0) for s, cs in ref.items()}
    R = np.array([rseed[s] for s in sorted(rseed)]); H = np.array([hip[s] for s in sorted(hip)]); X = np.array([tor[s] for s in sorted(tor)])
    for n, sl in PHASES.items():
        r, h, x = R[:, sl].mean(1), H[:, sl].mean(1), X[:, sl].mean(1)
        print(f"    {n:20s} reference ({len(r)} seeds) {r.mean():6.2f} +- {se(r):.2f} | HIP ({len(h)}) {h.mean():6.2f} +- {se(h):.2f}  diff {h.mean() - r.mean():+.2f} +- {np.hypot(se(h), se(r)):.2f}"
              f"  p = {stats.ttest_ind(h, r, equal_var=False).pvalue:.3f} | trainer on the reference's env ({len(x)}) {x.mean():6.2f} +- {se(x):.2f}  diff {x.mean() - r.mean():+.2f} +- {np.hypot(se(x), se(r)):.2f}"
              f"  p = {stats.ttest_ind(x, r, equal_var=False).pvalue:.3f}")
    new = sorted(s for s in rseed if s >= 100)
    if new:
        Rn = np.array([rseed[s] for s in new])
        for n, sl in PHASES.items():
            r, h = Rn[:, sl].mean(1), H[:, sl].mean(1)
            print(f"    {n:20s} reference at the NEW seeds only ({len(r)}) {r.mean():6.2f} +- {se(r):.2f} | HIP ({len(h)}) {h.mean():6.2f} +- {se(h):.2f}  p = {stats.ttest_ind(h, r, equal_var=False).pvalue:.3f}")
    print()
    print("epoch:                 " + " ".join(f"{e:5d}" for e in range(1, 41)))
    print(f"reference ({len(R):3d} seeds): " + " ".join(f"{v:5.1f}" for v in R.mean(0)))
    print(f"HIP ({len(H)}):             " + " ".join(f"{v:5.1f}" for v in H.mean(0)))
    print(f"trainer/ref env ({len(X)}):  " + " ".join(f"{v:5.1f}" for v in X.mean(0)))
    s = np.sqrt(H.var(0, ddof=1) / len(H) + R.var(0, ddof=1) / len(R))
    print("HIP - ref, in SE:      " + " ".join(f"{v:+5.1f}" for v in (H.mean(0) - R.mean(0)) / s))


if __name__ == "__main__":
    main()
