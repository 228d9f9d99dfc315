#!/bin/bash
# profiles/tools/sweep_knobs.sh : the tile kernel's scheduling knobs re-swept on the current kernel (kernel ms; run on the GPU box from the repo root)
mkdir -p gpurun_out/r4
run() { # workload, label, options...
  W=$1; L=$2; shift; shift
  python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$W', '$L', 'kernel', j['kernel_ms'], 'ms/frame', j['ms_per_step'])
"
}
for W in C2 C1 C3 C5; do
  run $W default
  for v in 12 16 32 48; do run $W ready_min=$v --opt 8=$v; done
  for v in 32 96 128 192; do run $W band=$v --opt 9=$v; done
  for v in 32 96 128 192; do run $W look=$v --opt 10=$v; done
  for v in 12 16 32 40; do run $W reserve=$v --opt 11=$v; done
  for v in 1 4 8; do run $W swizzle=$v --opt 4=$v; done
  run $W default
done
