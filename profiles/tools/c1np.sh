for L in base fill base fill; do GRT_LIB=$PWD/gaussian-ray-tracing_amd/libgrt_hip_$L.so python bench.py --workload C1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --opt 27=0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$L', 'C1 no parts kernel', j['kernel_ms'])
"; done
for L in base fill; do GRT_LIB=$PWD/gaussian-ray-tracing_amd/libgrt_hip_$L.so GRT_DEBUG_LAUNCH=1 python bench.py --workload C1 --steps 3 --warmup 5 --no-cpu-baseline --no-extra-legs 2>&1 | grep "grt launch" | tail -1 | cut -c1-220; done
for L in base fill base fill; do GRT_LIB=$PWD/gaussian-ray-tracing_amd/libgrt_hip_$L.so python bench.py --workload C2 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --opt 27=0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$L', 'C2 no parts kernel', j['kernel_ms'])
"; done
