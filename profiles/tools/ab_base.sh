#!/bin/bash
# Same-box A/B of bench.py between this tree and the round-1 tree checked out in .ab_base/ (git worktree,
# built in place).  Usage: gpurun -- 'bash profiles/tools/ab_base.sh "0 2 3 4" 3'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
CFGS=${1:-"0 2 3 4"}
REPS=${2:-3}
for c in $CFGS; do
  for r in $(seq $REPS); do
    for tree in .ab_base .; do
      extra=""; [ "$tree" = "." ] && extra="--no-traffic"  # (the round-1 bench.py has no such flag)
      (cd $REPO/$tree && python3 bench.py --config $c --no-cpu-baseline $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('config $c tree %-8s us/step %.2f  kernel %.2f' % ('$tree', d['ms_per_step']*1e3, d['roofline']['avg_launch_ms']*1e3))")
    done
  done
done
