#!/bin/bash
# scratch/mkdiag_r6b.sh : diag build for VERDICT r05 item 6: which tiles spill window overflow to their bags and never read it back?
# counters (wave-level, GRT_TILE_DIAG): rays = tiles that spilled, stall_exits = of those: no refill scan and one pass,
# segments = bag entries written (all tiles), proxy_tests = bag entries written by the tiles that never read back,
# rounds = passes, node_visits = refill scans, fetches/hit_evals as usual
set -e
D=/tmp/full_diag_r6b
rm -rf $D; mkdir -p $D/gaussian-ray-tracing_amd $D/include
cd /root/repo
cp -r gaussian-ray-tracing_amd/csrc $D/gaussian-ray-tracing_amd/csrc; cp include/grt.h $D/include/; rm -f $D/gaussian-ray-tracing_amd/csrc/*.o
cd $D/gaussian-ray-tracing_amd/csrc
python3 - <<'PY'
p='grt_render_tile.hip'
s=open(p).read()
for f in ("node_visits","stall_exits","rays","segments","proxy_tests"):
    s=s.replace("GRT_D(%s, 1)"%f,"")
old="        uint32_t npass = 0;\n"
assert s.count(old)==1
s=s.replace(old,old+"        uint32_t dg_sp_ = 0, dg_rf_ = 0;\n",1)
old="                                if (to_bag) {\n"
assert s.count(old)==1
s=s.replace(old,"                                dg_sp_ += (uint32_t)__popcll(wave_ballot(to_bag));\n"+old,1)
old="                            // lanes that do not need it yet but have room for four more keys come along: one scan instead\n"
assert s.count(old)==1
s=s.replace(old,"                            dg_rf_++; GRT_D(node_visits, 1)\n"+old,1)
old="        // (unit and part code are taken from the ONE scalar that lives across the passes, the order entry)\n"
assert s.count(old)==1
s=s.replace(old,"        if (dg_sp_) { GRT_D(rays, 1) GRT_D(segments, dg_sp_) if (dg_rf_ == 0u && npass == 1u) { GRT_D(stall_exits, 1) GRT_D(proxy_tests, dg_sp_) } }\n"+old,1)
open(p,'w').write(s)
PY
make -j8 OUT=$D/libgrt_hip.so EXTRA="-DGRT_TILE_DIAG" 2>&1 | grep -i "error\|moved behind" || true
cp $D/libgrt_hip.so /root/repo/gaussian-ray-tracing_amd/libgrt_hip_diag_r6b.so
echo built diag_r6b
