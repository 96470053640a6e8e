#!/bin/bash
# Round 6 evidence run (through gpurun from the repo root): GPU test suite, the bench line of every BASELINE
# configuration (full default run for the headline: in-run PMC traffic + cpu_baseline), rocprofv3 kernel stats,
# SQ counters per configuration, variant timings, trainer figures.  Results land in gpurun_out/r06/ and are copied to profiles/.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06
mkdir -p $O
cd $REPO
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -3 $O/gputest.log
timeout 900 python bench.py > $O/bench_headline.json 2> $O/bench_headline.err
for c in 2 3 4 6; do timeout 600 python bench.py --config $c > $O/bench_config$c.json 2> $O/bench_config$c.err; done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_driver_window.json 2>/dev/null
for c in 0 2 3 4 6; do
  BENCH_ARGS="--config $c" bash profiles/run_profile.sh r06_c$c > $O/profile_c$c.log 2>&1
  cp gpurun_out/prof_r06_c$c/summary_r06_c$c.md $O/ 2>/dev/null
  cp gpurun_out/prof_r06_c$c/traffic_r06_c$c.json $O/ 2>/dev/null
  f=$(ls gpurun_out/prof_r06_c$c/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_c$c.csv
done
( for c in 0 2 3 4 6; do echo "== config $c"; BENCH_ARGS="--config $c" bash profiles/tools/pmc_sq.sh r06_c$c; done ) > $O/pmc_sq_configs.txt 2>&1
( echo "== config 6, pds_step_k (K = 8)"; BENCH_ARGS="--config 6 --mode stepk --steps 64 --warmup 16" KERNEL_PATTERN=step_k_kernel bash profiles/tools/pmc_sq.sh r06_stepk6 ) > $O/pmc_sq_stepk.txt 2>&1
timeout 300 python bench.py --config 6 --also-envs 2097152 --no-cpu-baseline --no-traffic > $O/bench_config6_also_2pow21.json 2>/dev/null
timeout 900 python profiles/tools/time_variants.py > $O/variant_timings.txt 2>&1
( timeout 300 python profiles/tools/hist_breakdown.py 8192 64 DroneHoverSimpleEnv-v0 4; timeout 300 python profiles/tools/hist_breakdown.py 8192 64 DroneCircleSimpleEnv-v0 8 ) 2>&1 | grep epoch > $O/hist_breakdown.txt
AB_SPLIT=1 timeout 900 python profiles/tools/ab_variants.py libpds_hip.so > $O/ab_split_reset.txt 2>&1
python profiles/tools/kernel_resources.py step_ > $O/kernel_resources_all.txt 2>&1
timeout 600 python profiles/tools/mlp_bench.py > $O/mlp_bench.txt 2>&1
( timeout 300 python profiles/tools/ppo_breakdown.py 8192 64; timeout 300 python profiles/tools/ppo_breakdown.py 65536 32; timeout 300 python profiles/tools/ppo_breakdown.py 1048576 8 ) 2>&1 | grep launch > $O/ppo_breakdown.txt
ls -la $O
