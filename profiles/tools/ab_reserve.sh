#!/bin/bash
run() { python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%.4f' % j['kernel_ms'], end=' ')
"; }
for v in 24 20 16 14; do
  echo -n "reserve $v: C1 "; run --workload C1 --opt 11=$v; run --workload C1 --opt 11=$v
  echo -n " C2 "; run --workload C2 --opt 11=$v; run --workload C2 --opt 11=$v
  echo -n " C3 "; run --workload C3 --opt 11=$v; run --workload C3 --opt 11=$v
  echo -n " C5 "; run --workload C5 --opt 11=$v; run --workload C5 --opt 11=$v
  echo -n " C3a "; run --workload C3a --opt 11=$v --steps 8
  echo -n " rank4of8 "; run --workload C3 --emulate-ranks 8 --inflight 1 --opt 11=$v; run --workload C3 --emulate-ranks 8 --inflight 1 --opt 11=$v
  echo
done
