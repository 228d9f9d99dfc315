#!/bin/bash
run() { L=$1; shift; GRT_LIB=$PWD/gaussian-ray-tracing_amd/libgrt_hip_$L.so python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); c=j['config']; print('%.4f (%s, tests/ray %.1f boxes/ray %.1f)' % (j['kernel_ms'], c['kernel_variant'], c['proxy_tests_per_ray'], c['node_visits_per_ray']), end=' ')
"; }
for W in C3 C2 C5 C1 C3a; do
  echo -n "$W base: "; run base --workload $W; run base --workload $W; echo
  echo -n "$W leaf8 (leaf max 8): "; run leaf8 --workload $W --leaf-max 8; run leaf8 --workload $W --leaf-max 8; echo
  echo -n "$W leaf8 lib, leaf max 4: "; run leaf8 --workload $W; echo
done
