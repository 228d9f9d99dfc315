#!/bin/bash
# latency-bound cases per variant library: C1, C2 static kernel + orbit frame, rank 4 of 8 synchronous
run() { L=$1; shift; GRT_LIB=$PWD/gaussian-ray-tracing_amd/libgrt_hip_$L.so python bench.py --steps 20 --warmup 6 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%.4f/%s' % (j['kernel_ms'], j.get('ms_per_step_orbit')), end=' ')
"; }
for L in "$@"; do
  echo -n "$L: C1 "; run $L --workload C1; run $L --workload C1
  echo -n " C2 "; run $L --workload C2; run $L --workload C2
  echo -n " rank4of8 "; run $L --workload C3 --emulate-ranks 8 --inflight 1 --no-extra-legs; run $L --workload C3 --emulate-ranks 8 --inflight 1 --no-extra-legs
  echo
done
