#!/bin/bash
# profiles/tools/mkvar2.sh <label> [flags...] : only the tile kernel recompiled (current csrc source) with extra flags, linked with
# the other objects of csrc.  Built like the shipped object, through hipcc_via_asm.py (assembly checked and repaired);
# NOREPAIR=1 keeps the assembly as the compiler made it (the round-3 failure reproduced, profiles/r04_experiments_log.md).
set -e
L=$1; shift
R=/root/repo/gaussian-ray-tracing_amd
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden -Wall -Wno-unused-function -fopenmp"
cd $R/csrc
python3 hipcc_via_asm.py ${NOREPAIR:+--no-repair} --keep-asm /tmp/cs/var_$L /tmp/tile_$L.o grt_render_tile.hip $F "$@"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fopenmp -o $R/libgrt_hip_$L.so grt_api.o grt_bvh.o grt_render.o grt_render_wave.o grt_render_stream.o /tmp/tile_$L.o grt_render_tile_single.o grt_host.o
echo "built $L"
