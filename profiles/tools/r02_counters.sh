#!/bin/bash
# Round 2, item 1: SQ counters (VALU, LDS bank conflicts, wait share) per tile for BASELINE configs 2/3/4.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
for c in 2 3 4; do
  echo "== config $c (default tile rule)"
  BENCH_ARGS="--config $c" bash $REPO/profiles/tools/pmc_sq.sh ${TAG}_c$c
done
echo "== config 3, full tile"
PDS_FORCE_TILE=full BENCH_ARGS="--config 3" bash $REPO/profiles/tools/pmc_sq.sh ${TAG}_c3full
echo "== config 4 at 262144 envs, half tile"
PDS_FORCE_TILE=half BENCH_ARGS="--config 4 --envs-per-gpu 262144" bash $REPO/profiles/tools/pmc_sq.sh ${TAG}_c4half
