#!/bin/bash
# profiles/tools/ab.sh <workload> <steps> label...  : kernel ms of each variant library (run on the GPU box from the repo root)
W=$1; S=$2; shift; shift
mkdir -p gpurun_out/r4
for rep in 1 2; do
for L in "$@"; do
  GRT_LIB=$PWD/gaussian-ray-tracing_amd/libgrt_hip_$L.so python bench.py --workload $W --steps $S --warmup 5 --no-cpu-baseline --no-extra-legs --dump gpurun_out/r4/frame_${W}_$L.npy 2> gpurun_out/r4/ab_$L.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$L', '$W', 'ms/frame', j['ms_per_step'], 'kernel', j['kernel_ms'], 'tests/ray', j['config']['proxy_tests_per_ray'], 'boxes/ray', j['config']['node_visits_per_ray'], 'rounds', j['config']['rounds_per_ray'])
"
done
done
md5sum gpurun_out/r4/frame_${W}_*.npy   # the variants' frames: identical files = bit-identical frames
rm -f gpurun_out/r4/frame_${W}_*.npy
