#!/usr/bin/env python3
"""Condense a profiles/run_profile.sh output directory into a small markdown summary
(kernel-trace stats + per-launch HBM traffic from the PMC passes)."""
import csv
import glob
import os
import sys


def find(root, suffix):
    hits = sorted(glob.glob(os.path.join(root, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def main(out):
    print(f"# rocprofv3 summary ({os.path.basename(out)})\n")
    stats = find(os.path.join(out, "trace"), "kernel_stats.csv")
    if stats:
        print("## --kernel-trace --stats (python3 bench.py --steps 100 --warmup 10)\n")
        print("| kernel | calls | total ns | avg ns | min ns | max ns | % |")
        print("|---|---|---|---|---|---|---|")
        for r in csv.DictReader(open(stats)):
            name = r.get("Name", "")[:90]
            print(f"| `{name}` | {r.get('Calls')} | {r.get('TotalDurationNs')} | {r.get('AverageNs')} | "
                  f"{r.get('MinNs')} | {r.get('MaxNs')} | {r.get('Percentage')} |")
    for label, sub, scale in (("FETCH_SIZE", "pmc_fetch", 2.0), ("WRITE_SIZE", "pmc_write", 1.0)):
        f = find(os.path.join(out, sub), "counter_collection.csv")
        if not f:
            continue
        vals = {}
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != label:
                continue
            vals.setdefault(r.get("Kernel_Name", "")[:90], []).append(float(r.get("Counter_Value", 0)))
        print(f"\n## --pmc {label} (KiB per dispatch; gfx950 correction factor x{scale})\n")
        print("| kernel | dispatches | mean counter | corrected MB per launch |")
        print("|---|---|---|---|")
        for k, v in vals.items():
            m = sum(v) / len(v)
            print(f"| `{k}` | {len(v)} | {m:.1f} | {m * 1024 * scale / 1e6:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
