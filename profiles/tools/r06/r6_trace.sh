#!/bin/bash
# usage: scratch/r6_trace.sh <tag> <bench args...>   -> gpurun_out/trace_<tag>/
TAG=$1; shift
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/trace_$TAG; rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra-legs "$@" > $O/bench.log 2>&1
tail -c 600 $O/bench.log
