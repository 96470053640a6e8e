set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r03_gputest2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_gputest2.log
tail -8 gpurun_out/r03_gputest2.log
for c in 0 2 3 4 6; do timeout 300 python bench.py --config $c --steps 500 --warmup 50 --no-cpu-baseline --no-traffic >> gpurun_out/r03_bench_cfgs.log 2>&1; done
for c in 0 2 3 4 6; do PDS_LIB=$PWD/phoenix-drone-simulation_amd/libpds_hip_mw4.so timeout 300 python bench.py --config $c --steps 500 --warmup 50 --no-cpu-baseline --no-traffic >> gpurun_out/r03_bench_cfgs_mw4.log 2>&1; done
for f in gpurun_out/r03_bench_cfgs.log gpurun_out/r03_bench_cfgs_mw4.log; do echo $f; python -c "
import sys, json
for l in open('$f'):
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print(d['config']['workload'][:40], d['config']['envs_per_gpu'], 'kernel us %.2f' % (r['avg_launch_ms']*1e3), 'frac %.3f' % r['frac'], 'frac_wall %.3f' % r['frac_wall'])
"; done
timeout 900 python profiles/tools/time_variants.py > gpurun_out/r03_variants.txt 2>&1
PDS_LIB=$PWD/phoenix-drone-simulation_amd/libpds_hip_mw4.so timeout 900 python profiles/tools/time_variants.py > gpurun_out/r03_variants_mw4.txt 2>&1
cat gpurun_out/r03_variants.txt; cat gpurun_out/r03_variants_mw4.txt
