#!/bin/bash
# part-wave thresholds re-swept for a variant library: profiles/tools/sweep_parts2.sh <label>
L=$1
run() { GRT_LIB=$PWD/gaussian-ray-tracing_amd/libgrt_hip_$L.so python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%.4f' % j['kernel_ms'], end=' ')
"; }
for o in "27=60 --opt 28=75" "27=50 --opt 28=75" "27=40 --opt 28=75" "27=70 --opt 28=75" "27=60 --opt 28=50" "27=60 --opt 28=100" "27=50 --opt 28=50" "27=40 --opt 28=50"; do
  echo -n "$L opts $o: C1 "; run --workload C1 --opt $o; run --workload C1 --opt $o
  echo -n " C2 "; run --workload C2 --opt $o; run --workload C2 --opt $o
  echo -n " rank4of8 "; run --workload C3 --emulate-ranks 8 --inflight 1 --opt $o; run --workload C3 --emulate-ranks 8 --inflight 1 --opt $o
  echo
done
