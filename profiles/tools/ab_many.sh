#!/bin/bash
# profiles/tools/ab_many.sh label... : kernel ms of each variant library on C1, C2, C3, C5 and the emulated rank 4 of 8 (synchronous), twice
mkdir -p gpurun_out/r4
run() { L=$1; shift; GRT_LIB=$PWD/gaussian-ray-tracing_amd/libgrt_hip_$L.so python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%.4f' % j['kernel_ms'], end=' ')
"; }
for L in "$@"; do
  echo -n "$L: C1 "; run $L --workload C1; run $L --workload C1
  echo -n " C2 "; run $L --workload C2; run $L --workload C2
  echo -n " C3 "; run $L --workload C3; run $L --workload C3
  echo -n " C5 "; run $L --workload C5
  echo -n " rank4of8 "; run $L --workload C3 --emulate-ranks 8 --inflight 1; run $L --workload C3 --emulate-ranks 8 --inflight 1
  echo
done
