#!/bin/bash
# profiles/tools/mkvar2.sh <label> [flags...] : only the tile kernel recompiled (current csrc source) with extra flags, linked with the other objects of csrc
set -e
L=$1; shift
R=/root/repo/gaussian-ray-tracing_amd
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden -Wall -Wno-unused-function -fopenmp"
cd $R/csrc
/opt/rocm/bin/hipcc $F "$@" -c grt_render_tile.hip -o /tmp/tile_$L.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fopenmp -o $R/libgrt_hip_$L.so grt_api.o grt_bvh.o grt_render.o grt_render_wave.o grt_render_stream.o /tmp/tile_$L.o grt_render_tile_single.o grt_host.o
echo "built $L"
