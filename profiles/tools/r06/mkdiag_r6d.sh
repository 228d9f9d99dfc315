#!/bin/bash
# scratch/mkdiag_r6d.sh : diag build for the lane-utilisation column of the section budget.  Wave-level counters:
# segments = particles fetched (trips), proxy_tests = exact tests run, hit_evals = compositing steps, fetches = leaf steps (as in the plain
# diag build); rays = SUM over exact tests of the lanes the sphere pre-test lets through, stall_exits = SUM of the lanes that hit,
# rounds = SUM over insert blocks of the lanes that insert, node_visits = SUM over compositing steps of the lanes that composite
set -e
D=/tmp/full_diag_r6d
rm -rf $D; mkdir -p $D/gaussian-ray-tracing_amd $D/include
cd /root/repo
cp -r gaussian-ray-tracing_amd/csrc $D/gaussian-ray-tracing_amd/csrc; cp include/grt.h $D/include/; rm -f $D/gaussian-ray-tracing_amd/csrc/*.o
cd $D/gaussian-ray-tracing_amd/csrc
python3 - <<'PY'
p='grt_render_tile.hip'
s=open(p).read()
for f in ("node_visits","stall_exits","rays","rounds"):
    s=s.replace("GRT_D(%s, 1)"%f,"")
old="                            if (!m_) continue;\n"
assert s.count(old)==1
s=s.replace(old,old+"                            GRT_D(rays, (uint32_t)__popcll(m_))\n",1)
old="                        const bool hit = proxy_slabs_pre(pa, d_g, r0.w, te, tx) && act_;\n"
assert s.count(old)==1
s=s.replace(old,old+"                        GRT_D(stall_exits, (uint32_t)__popcll(wave_ballot(hit)))\n",1)
old="                        if (wave_any(ins)) {\n"
assert s.count(old)==1
s=s.replace(old,old+"                            GRT_D(rounds, (uint32_t)__popcll(wave_ballot(ins)))\n",1)
old="                        if (!cm_) continue;\n                        GRT_D(hit_evals, 1)\n"
assert s.count(old)==1
s=s.replace(old,old+"                        GRT_D(node_visits, (uint32_t)__popcll(cm_))\n",1)
open(p,'w').write(s)
PY
make -j8 OUT=$D/libgrt_hip.so EXTRA="-DGRT_TILE_DIAG" 2>&1 | grep -i "error\|moved behind" || true
cp $D/libgrt_hip.so /root/repo/gaussian-ray-tracing_amd/libgrt_hip_diag_r6d.so
echo built diag_r6d
