#!/bin/bash
# profiles/tools/mkvar_single.sh <label> [flags...] : only the one-ray-per-wave translation unit recompiled with extra flags
set -e
L=$1; shift
R=/root/repo/gaussian-ray-tracing_amd
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden -Wall -Wno-unused-function -fopenmp"
cd $R/csrc
python3 hipcc_via_asm.py --keep-asm /tmp/cs/var_$L /tmp/single_$L.o grt_render_tile_single.hip $F "$@"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fopenmp -o $R/libgrt_hip_$L.so grt_api.o grt_bvh.o grt_render.o grt_render_wave.o grt_render_stream.o grt_render_tile.o /tmp/single_$L.o grt_host.o
grep -h "group_segment_fixed_size\|\.vgpr_count" /tmp/cs/var_$L/*.s | sort | uniq -c | head -4
echo "built $L"
