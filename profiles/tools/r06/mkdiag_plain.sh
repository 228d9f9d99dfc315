#!/bin/bash
# the counted build of the tile kernel as it stands (make EXTRA=-DGRT_TILE_DIAG) -> gaussian-ray-tracing_amd/libgrt_hip_diag.so
set -e
D=/tmp/full_diag_plain
rm -rf $D; mkdir -p $D/gaussian-ray-tracing_amd $D/include
cd /root/repo
cp -r gaussian-ray-tracing_amd/csrc $D/gaussian-ray-tracing_amd/csrc; cp include/grt.h $D/include/; rm -f $D/gaussian-ray-tracing_amd/csrc/*.o
cd $D/gaussian-ray-tracing_amd/csrc
make -j8 OUT=$D/libgrt_hip.so EXTRA="-DGRT_TILE_DIAG" 2>&1 | grep -i "error\|moved behind" || true
cp $D/libgrt_hip.so /root/repo/gaussian-ray-tracing_amd/libgrt_hip_diag.so
echo built diag
