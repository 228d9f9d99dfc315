for o in "--opt 31=1" "--opt 31=0" "--opt 31=1" "--opt 31=0" "--opt 29=0" "--opt 29=0 --opt 31=0"; do python bench.py --workload C4 --steps 20 --warmup 6 --no-cpu-baseline --no-extra-legs $o 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('C4 [$o] frame', j['ms_per_step'], 'kernel', j['kernel_ms'])
"; done
