#!/bin/bash
for W in C3 C5 C2; do for v in 1 2 4 8 16 32; do python bench.py --workload $W --steps 20 --warmup 6 --no-cpu-baseline --no-extra-legs --opt 4=$v 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$W swizzle $v kernel', j['kernel_ms'])
"; done; done
