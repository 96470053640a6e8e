#!/bin/bash
# rocprofv3 kernel trace of three PPO epochs (8192 envs x 64 steps): which kernels an epoch consists of
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_ppo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/profiles/tools/ppo_breakdown.py 8192 64 > $OUT.log 2>&1
cd $REPO && python3 - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("| kernel | calls | total ms | avg us | % |"); print("|---|---|---|---|---|")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f"| \`{r['Name'][:70]}\` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {100*float(r['TotalDurationNs'])/tot:.1f} |")
PY
tail -3 $OUT.log
