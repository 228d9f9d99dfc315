#!/bin/bash
# profiles/tools/sweep.sh <workload> <steps> <flagname> v1 v2 ... : bench with --<flagname> v for each v (default library)
W=$1; S=$2; FL=$3; shift; shift; shift
mkdir -p gpurun_out/r3
for V in "$@"; do
  python bench.py --workload $W --steps $S --warmup 5 --no-cpu-baseline --no-extra-legs --$FL $V 2> gpurun_out/r3/sweep.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); c=j['config']; print('$FL=$V', '$W', 'ms/frame', j['ms_per_step'], 'kernel', j['kernel_ms'], 'tests/ray', c['proxy_tests_per_ray'], 'boxes/ray', c['node_visits_per_ray'], 'rounds', c['rounds_per_ray'], 'prims', c.get('n_bvh_primitives'), 'build_ms', c['bvh_build_ms'])
"
done
