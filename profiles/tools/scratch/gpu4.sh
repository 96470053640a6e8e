mkdir -p gpurun_out
P=$PWD/phoenix-drone-simulation_amd
timeout 1500 python -m pytest tests -m gpu -q -x -k "parity or properties or noise or stepk" > gpurun_out/r03_gputest4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_gputest4.log
tail -5 gpurun_out/r03_gputest4.log
rm -f gpurun_out/r03_ab2.txt
for lib in libpds_hip_base.so libpds_hip.so libpds_hip_base.so libpds_hip.so; do
  echo "=== $lib" >> gpurun_out/r03_ab2.txt
  for c in 0 2 3 4 6; do PDS_LIB=$P/$lib timeout 300 python bench.py --config $c --steps 500 --warmup 50 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('config $c', d['config']['envs_per_gpu'], 'kernel us %.2f' % (r['avg_launch_ms']*1e3), 'frac %.3f' % r['frac'])
" >> gpurun_out/r03_ab2.txt; done
done
for g in 512 768 1024 1536 2048; do echo "=== grid $g" >> gpurun_out/r03_ab2.txt; for c in 0 6; do PDS_GRID_BLOCKS=$g timeout 300 python bench.py --config $c --steps 500 --warmup 50 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('config $c', d['config']['envs_per_gpu'], 'kernel us %.2f' % (r['avg_launch_ms']*1e3), 'frac %.3f' % r['frac'])
" >> gpurun_out/r03_ab2.txt; done; done
echo "=== 2^21, 2^22 base / new" >> gpurun_out/r03_ab2.txt
for lib in libpds_hip_base.so libpds_hip.so; do for n in 2097152 4194304; do PDS_LIB=$P/$lib timeout 300 python bench.py --envs-per-gpu $n --steps 300 --warmup 50 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$lib', d['config']['envs_per_gpu'], 'kernel us %.2f' % (r['avg_launch_ms']*1e3), 'frac %.3f' % r['frac'])
" >> gpurun_out/r03_ab2.txt; done; done
cat gpurun_out/r03_ab2.txt
