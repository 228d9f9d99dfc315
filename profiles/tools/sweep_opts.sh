#!/bin/bash
# profiles/tools/sweep_opts.sh <out.log> <bench args...> -- "<opts A>" "<opts B>" ...   (run on the GPU box from the repo root)
# One bench.py run per option set (each a string of --opt ID=VALUE ...); prints ms/frame, kernel ms, synchronous latency.
OUT=$1; shift
ARGS=()
while [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
for O in "$@"; do
  python bench.py "${ARGS[@]}" --no-cpu-baseline $O 2>> $OUT.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); c=j['config']
        print('opts[$O]', '${ARGS[*]}', 'ms/frame', j['ms_per_step'], 'kernel', j['kernel_ms'], 'latency', c.get('latency_ms_per_frame'), 'sync', c.get('value_sync'), 'pipe', c.get('value_pipelined'), 'orbit', j.get('ms_per_step_orbit'), 'cold', c.get('kernel_ms_cold'), 'tests/ray', c['proxy_tests_per_ray'], 'boxes/ray', c['node_visits_per_ray'])
" | tee -a $OUT
done
