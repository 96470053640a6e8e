#!/usr/bin/env python3
"""Measured single-step parity margins (GPU): runs the single-step / reset parity tests of tests/test_gpu_parity.py
and tests/test_gpu_noise.py with a recording `assert_close` and prints, per field (and per observation column
group), the largest absolute error and the largest absolute slack needed on top of 1e-6 relative
(max(|got - want| - 1e-6 |want|)) over all reference scenarios.  DESIGN section 6 quotes this table.

usage (GPU box): python profiles/tools/parity_margins.py > gpurun_out/parity_margins.txt"""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_util as gu  # noqa: E402

REC = collections.defaultdict(lambda: [0.0, 0.0, ""])  # key -> [max err, max slack over 1e-6 rel, scenario]


def record(a, b, rtol, atol, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b)
    slack = err - 1e-6 * np.abs(b)
    scen, _, field = what.partition(" ")
    if field in ("obs", "reset obs") and a.ndim == 2:
        g = gu.Golden(scen)
        groups = gu.obs_groups(g.task, a.shape[1], gu.noisy(g) and "det" not in scen)
        for c, name in enumerate(groups):
            k = f"{field}[{name}]"
            if err[:, c].max() > REC[k][0]:
                REC[k][0] = float(err[:, c].max())
            if slack[:, c].max() > REC[k][1]:
                REC[k][1], REC[k][2] = float(slack[:, c].max()), scen
        return
    if field == "pid state":
        return
    fam = "pid " if any(t in scen for t in ("_rate", "_att")) else ""
    k = fam + field
    REC[k][0] = max(REC[k][0], float(err.max()))
    if slack.max() > REC[k][1]:
        REC[k][1], REC[k][2] = float(slack.max()), scen


def main():
    gu.assert_close = record
    import test_gpu_noise as tn
    import test_gpu_parity as tp
    for name in tp.DET_SCENARIOS:
        tp.test_single_step_vs_reference(name)
    for name in tn.NOISE_SCENARIOS:
        tn.test_noisy_single_step_vs_reference(name)
        tn.test_noisy_reset_vs_reference(name)
    single = dict(REC)
    REC.clear()
    # closed loop (tests/test_gpu_parity.py::test_closed_loop_short_horizon): error growth per step, as a multiple of
    # the single-step unit 1e-6 |want| + 1e-6
    CL = collections.defaultdict(float)

    def record_cl(a, b, rtol, atol, what=""):
        a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
        scen, tt, field = what.split(" ")
        CL[(field, int(tt[1:]))] = max(CL[(field, int(tt[1:]))], float((np.abs(a - b) / (1e-6 * np.abs(b) + 1e-6)).max()))
    gu.assert_close = record_cl
    for scen in ["hover_det", "circle_det", "takeoff_det"] + tp.LAT_DET:
        tp.test_closed_loop_short_horizon(scen)
    print("closed loop: max |err| / (1e-6 |want| + 1e-6) by step")
    for field in ("obs", "reward"):
        print(f"  {field:7s}", " ".join(f"t{t}:{CL[(field, t)]:.1f}" for t in sorted(t for f, t in CL if f == field)))
    REC.update(single)
    print(f"{'field':28s} {'max |err|':>12s} {'max slack over 1e-6 rel':>26s}  worst scenario")
    for k in sorted(REC):
        e, s, sc = REC[k]
        print(f"{k:28s} {e:12.3e} {s:26.3e}  {sc}")


if __name__ == "__main__":
    main()
