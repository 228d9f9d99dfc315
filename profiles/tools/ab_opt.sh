#!/bin/bash
# profiles/tools/ab_opt.sh "<bench options A>" "<bench options B>" ... : kernel ms (static) and orbit frame ms of the shipped library per option set
for W in C1 C2 C3 C5; do for o in "$@"; do python bench.py --workload $W --steps 20 --warmup 6 --no-cpu-baseline $o 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$W [$o] static frame', j['ms_per_step'], 'kernel', j['kernel_ms'], 'orbit frame', j['ms_per_step_orbit'], 'orbit kernel', j['config']['kernel_ms_orbit'], 'cold kernel', j['config']['kernel_ms_cold'])
"; done; done
