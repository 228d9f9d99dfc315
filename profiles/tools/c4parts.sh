for o in "28=75" "28=50" "28=35" "28=20" "28=0" "28=35 --opt 27=40" "28=20 --opt 27=40"; do python bench.py --workload C4 --steps 20 --warmup 6 --no-cpu-baseline --no-extra-legs --opt $o 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('opts', '$o', 'ms/frame', j['ms_per_step'], 'kernel', j['kernel_ms'])
"; done
