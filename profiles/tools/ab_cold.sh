#!/bin/bash
for W in C1 C2 C3; do for o in "--opt 15=1" "--opt 15=2 --opt 32=60" "--opt 15=2 --opt 32=40" "--opt 15=2 --opt 32=25" "--opt 15=2 --opt 32=15"; do python bench.py --workload $W --steps 5 --warmup 3 --no-cpu-baseline $o 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$W [$o] cold kernel', j['config']['kernel_ms_cold'], 'cold frame', j['config']['frame_ms_cold'], 'static kernel', j['kernel_ms'])
"; done; done
