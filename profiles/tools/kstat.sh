#!/bin/bash
# per-kernel averages of a C4 bench under rocprof for variant lib $1
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/r3/ks_$1; rm -rf $O; mkdir -p $O; cd /tmp
GRT_LIB=$R/gaussian-ray-tracing_amd/libgrt_hip_$1.so rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 $R/bench.py --workload C4 --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/log 2>&1
cd $R; f=$(find $O -name "*kernel_stats.csv" | head -1); echo "== $1"; head -6 $f | cut -d, -f1-4 | cut -c1-140
