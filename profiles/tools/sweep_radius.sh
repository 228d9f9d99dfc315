#!/bin/bash
# moving camera: the dilation radius of the cost map (GRT_OPT_COST_RADIUS) re-swept; orbit frame / orbit kernel ms
for W in C2 C1 C3; do for r in 0 2 3 4 6 8; do python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --opt 13=$r 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$W radius $r orbit frame', j['ms_per_step_orbit'], 'orbit kernel', j['config']['kernel_ms_orbit'], 'static kernel', j['kernel_ms'])
"; done; done
