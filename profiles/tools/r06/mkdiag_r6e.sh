#!/bin/bash
# diag: exact tests of pieces — how many hit the particle on some lane, how many insert anything (wave-level)
# segments = particles fetched, proxy_tests = exact tests, rays = of those pieces, node_visits = piece tests where some lane HITS, rounds = piece tests where some lane INSERTS,
# stall_exits = NON-piece tests where some lane inserts
set -e
D=/tmp/full_diag_r6e
rm -rf $D; mkdir -p $D/gaussian-ray-tracing_amd $D/include
cd /root/repo
cp -r gaussian-ray-tracing_amd/csrc $D/gaussian-ray-tracing_amd/csrc; cp include/grt.h $D/include/; rm -f $D/gaussian-ray-tracing_amd/csrc/*.o
cd $D/gaussian-ray-tracing_amd/csrc
python3 - <<'PY'
p='grt_render_tile.hip'
s=open(p).read()
for f in ("node_visits","stall_exits","rays","rounds"):
    s=s.replace("GRT_D(%s, 1)"%f,"")
old="                        GRT_D(proxy_tests, 1)\n"
assert s.count(old)==1
s=s.replace(old,old+"                        if (PIECES && __float_as_uint(r3.w)) { GRT_D(rays, 1) }\n",1)
old="                        const uint32_t cellb = PIECES ? __float_as_uint(r3.w) : 0u;\n"
assert s.count(old)==1
s=s.replace(old,old+"                        if (PIECES && cellb && wave_any(hit)) { GRT_D(node_visits, 1) }\n",1)
old="                        if (wave_any(ins)) {\n"
assert s.count(old)==1
s=s.replace(old,"                        if (PIECES && cellb && wave_any(ins)) { GRT_D(rounds, 1) }\n                        if (!(PIECES && cellb) && wave_any(ins)) { GRT_D(stall_exits, 1) }\n"+old,1)
open(p,'w').write(s)
PY
make -j8 OUT=$D/libgrt_hip.so EXTRA="-DGRT_TILE_DIAG" 2>&1 | grep -i "error\|moved behind" || true
cp $D/libgrt_hip.so /root/repo/gaussian-ray-tracing_amd/libgrt_hip_diag_r6e.so
echo built
