#!/bin/bash
# eager vs hipGraph replay vs pds_step_k for the small-batch configs (launch / latency bound)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
run() { python3 bench.py --no-cpu-baseline --no-traffic "$@" 2>&1 | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-60s us/env-step-launch %.2f  value %.3e  frac %.3f' % (' '.join(sys.argv[1:]), d['ms_per_step']*1e3, d['value'], d['roofline']['frac']))
        break
else: print('FAILED', ' '.join(sys.argv[1:]))
" "$@"; }
for c in ${1:-2 3 0}; do
  run --config $c --steps 512 --warmup 64
  run --config $c --steps 512 --warmup 64 --mode graph
  for k in 4 8 16 64; do run --config $c --steps 512 --warmup 64 --mode stepk --k $k; done
done
