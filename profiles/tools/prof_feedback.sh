#!/bin/bash
# profiles/tools/prof_feedback.sh <label> <workload> : rocprofv3 kernel stats of a bench run WITH the orbit leg (the kernels behind a moving camera's frame)
L=$1; W=$2
cd /tmp && export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/gpurun_out/r4/fb_${L}_$W
mkdir -p $D
GRT_LIB=$GRAFT_REPO_ROOT/gaussian-ray-tracing_amd/libgrt_hip_$L.so rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline > $D/bench.json 2> $D/err.log
f=$(find $D -name "*kernel_stats.csv" | head -1)
echo "== $L $W"; [ -n "$f" ] && grep -i "cost_order\|dilate\|check_costs\|frame_tail\|eye_records\|k_render_tile" $f | cut -d, -f1-4 | cut -c1-150
