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


def traffic_json(out):
    """Per-launch HBM bytes of pds::step_kernel from the two PMC passes (FETCH_SIZE x2 on gfx950,
    WRITE_SIZE exact; both in KiB) -> dict for bench.py's roofline.traffic."""
    res = {}
    for label, sub, scale in (("FETCH_SIZE", "pmc_fetch", 2.0), ("WRITE_SIZE", "pmc_write", 1.0)):
        f = find(os.path.join(out, sub), "counter_collection.csv")
        if not f:
            return None
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
             if r.get("Counter_Name") == label and "step_kernel" in r.get("Kernel_Name", "")]
        if not v:
            return None
        res[label] = sum(v) / len(v) * 1024 * scale
    return {"kernel": "pds::step_kernel", "read_bytes_per_launch": res["FETCH_SIZE"],
            "write_bytes_per_launch": res["WRITE_SIZE"],
            "hbm_bytes_per_launch": res["FETCH_SIZE"] + res["WRITE_SIZE"],
            "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; FETCH_SIZE doubled "
                      "(gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md HBM section)"}


def main(out):
    print(f"# rocprofv3 summary ({os.path.basename(out)})\n")
    stats = find(os.path.join(out, "trace"), "kernel_stats.csv")
    if stats:
        print("## --kernel-trace --stats (python3 bench.py --steps 500 --warmup 50: the default bench window)\n")
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
    if len(sys.argv) > 2:
        import json
        t = traffic_json(sys.argv[1])
        if t:
            json.dump(t, open(sys.argv[2], "w"), indent=1)
