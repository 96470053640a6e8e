mkdir -p gpurun_out
P=$PWD/phoenix-drone-simulation_amd
timeout 1500 python -m pytest tests -m gpu -q -x -k "parity or noise or stepk or properties or trainer" > gpurun_out/r03_gputest11.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_gputest11.log
tail -5 gpurun_out/r03_gputest11.log
rm -f gpurun_out/r03_ab8.txt
for lib in libpds_hip_drain.so libpds_hip.so libpds_hip_drain.so libpds_hip.so; do
  echo "=== $lib" >> gpurun_out/r03_ab8.txt
  for c in 6; do PDS_LIB=$P/$lib timeout 300 python bench.py --config $c --steps 500 --warmup 50 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('config $c', d['config']['envs_per_gpu'], 'kernel us %.2f' % (r['avg_launch_ms']*1e3), 'frac %.3f' % r['frac'])
" >> gpurun_out/r03_ab8.txt; done
done
for lib in libpds_hip_drain.so libpds_hip.so; do
  echo "=== $lib" >> gpurun_out/r03_ab8.txt
  PDS_LIB=$P/$lib timeout 900 python profiles/tools/time_variants.py 1048576 200 0 >> gpurun_out/r03_ab8.txt 2>&1
done
cat gpurun_out/r03_ab8.txt
