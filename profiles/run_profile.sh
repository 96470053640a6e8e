#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'bash profiles/run_profile.sh r01'
# 1) --kernel-trace --stats (per-kernel average duration), 2) and 3) separate --pmc passes for the
# HBM read / write byte counters (MI355X_MICROARCH.md "HBM": FETCH_SIZE is reported at 1/2 of the
# bytes of a wide coalesced read on gfx950 -> doubled by profiles/summarize.py; WRITE_SIZE exact).
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps ${PROFILE_STEPS:-500} --warmup ${PROFILE_WARMUP:-50} --no-cpu-baseline --no-traffic ${BENCH_ARGS:-}"  # (the default bench window, so that the profiled average is the one bench.py times)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/bench_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $ARGS > "$OUT/bench_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $ARGS > "$OUT/bench_pmc_write.log" 2>&1
cd "$REPO" && python3 profiles/summarize.py "$OUT" "$OUT/traffic_$TAG.json" > "$OUT/summary_$TAG.md" 2>&1
cat "$OUT/summary_$TAG.md"
