mkdir -p gpurun_out
P=$PWD/phoenix-drone-simulation_amd
rm -f gpurun_out/r03_ab6.txt
for lib in libpds_hip.so libpds_hip_b512.so libpds_hip.so libpds_hip_b512.so; do
  echo "=== $lib" >> gpurun_out/r03_ab6.txt
  for c in 2 3 0; do PDS_LIB=$P/$lib timeout 300 python bench.py --config $c --steps 500 --warmup 50 --no-cpu-baseline --no-traffic 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('config $c', d['config']['envs_per_gpu'], 'kernel us %.2f' % (r['avg_launch_ms']*1e3), 'frac %.3f' % r['frac'])
    elif 'rror' in l: print(l.strip()[:200])
" >> gpurun_out/r03_ab6.txt; done
  for n in 16384 32768 131072; do PDS_LIB=$P/$lib timeout 300 python bench.py --envs-per-gpu $n --steps 500 --warmup 50 --no-cpu-baseline --no-traffic 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('hover', d['config']['envs_per_gpu'], 'kernel us %.2f' % (r['avg_launch_ms']*1e3), 'frac %.3f' % r['frac'])
" >> gpurun_out/r03_ab6.txt; done
done
cat gpurun_out/r03_ab6.txt
