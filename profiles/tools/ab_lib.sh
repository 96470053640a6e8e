#!/bin/bash
# Same-box A/B of two builds of libpds_hip (PDS_LIB): bash profiles/tools/ab_lib.sh libA.so libB.so "2 3 0" 3 [extra bench args]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
A=$1; B=$2; CFGS=${3:-"0 2 3 4"}; REPS=${4:-3}; shift 4
for c in $CFGS; do for r in $(seq $REPS); do for lib in $A $B; do
  PDS_LIB=$REPO/phoenix-drone-simulation_amd/$lib python3 bench.py --config $c --no-cpu-baseline --no-traffic "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('config $c %-24s us/step %.2f' % ('$lib', d['ms_per_step']*1e3))"
done; done; done
