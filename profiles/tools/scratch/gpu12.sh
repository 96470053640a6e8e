mkdir -p gpurun_out/r03
for c in 0 3 6; do
  BENCH_ARGS="--config $c" bash profiles/run_profile.sh r03b_c$c > gpurun_out/r03/profileb_c$c.log 2>&1
  cp gpurun_out/prof_r03b_c$c/summary_r03b_c$c.md gpurun_out/r03/ 2>/dev/null
  cp gpurun_out/prof_r03b_c$c/traffic_r03b_c$c.json gpurun_out/r03/ 2>/dev/null
  f=$(ls gpurun_out/prof_r03b_c$c/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/r03/kernel_statsb_c$c.csv
  timeout 300 python bench.py --config $c --no-cpu-baseline --no-traffic | grep "^{" > gpurun_out/r03/benchb_c$c.json
done
grep -h step_kernel gpurun_out/r03/kernel_statsb_c*.csv | cut -c1-230
python -c "
import json
for c in (0,3,6):
    d=json.loads(open(f'gpurun_out/r03/benchb_c{c}.json').read()); print(c, d['roofline']['avg_launch_ms']*1e3)
"
