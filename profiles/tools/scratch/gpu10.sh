mkdir -p gpurun_out
rm -f gpurun_out/r03_ab7.txt
for rep in 1 2; do for tile in full half; do
  echo "=== tile $tile" >> gpurun_out/r03_ab7.txt
  for c in 0 4; do PDS_FORCE_TILE=$tile timeout 300 python bench.py --config $c --steps 500 --warmup 50 --no-cpu-baseline --no-traffic 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('config $c', d['config']['envs_per_gpu'], 'kernel us %.2f' % (r['avg_launch_ms']*1e3), 'frac %.3f' % r['frac'])
" >> gpurun_out/r03_ab7.txt; done
  for n in 524288 2097152; do PDS_FORCE_TILE=$tile timeout 300 python bench.py --envs-per-gpu $n --steps 300 --warmup 50 --no-cpu-baseline --no-traffic 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('hover', d['config']['envs_per_gpu'], 'kernel us %.2f' % (r['avg_launch_ms']*1e3), 'frac %.3f' % r['frac'])
" >> gpurun_out/r03_ab7.txt; done
done; done
cat gpurun_out/r03_ab7.txt
