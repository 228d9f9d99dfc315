mkdir -p gpurun_out/r4
for s in 5 6 7 8 9 10; do
  python bench.py --workload C3a --steps 8 --warmup 3 --no-cpu-baseline --no-extra-legs --split $s 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); c=j['config']; print('split', $s, 'ms', j['ms_per_step'], 'tests/ray', c['proxy_tests_per_ray'], 'boxes/ray', c['node_visits_per_ray'], 'prims', c['n_bvh_primitives'], 'build', c['bvh_build_ms'])
"
done
