#!/bin/bash
# scratch/mkdiag_r6.sh : diag build for VERDICT r05 item 5(b): how many exact-test trips of a leaf step could PAIR by 8x4 halves of the tile?
# counters (wave-level, GRT_TILE_DIAG): segments = particles fetched (trips), proxy_tests = trips whose sphere pre-test lets some lane through,
# rounds = of those, trips whose pre-test mask lies in ONE top / bottom half, node_visits = pairs (top-only with bottom-only) per leaf step summed,
# rays = trips whose mask lies in ONE left / right half, stall_exits = pairs (left-only with right-only), fetches = leaf steps, hit_evals = compositing steps
set -e
D=/tmp/full_diag_r6
rm -rf $D; mkdir -p $D/gaussian-ray-tracing_amd $D/include
cd /root/repo
cp -r gaussian-ray-tracing_amd/csrc $D/gaussian-ray-tracing_amd/csrc; cp include/grt.h $D/include/; rm -f $D/gaussian-ray-tracing_amd/csrc/*.o
cd $D/gaussian-ray-tracing_amd/csrc
python3 - <<'PY'
p='grt_render_tile.hip'
s=open(p).read()
for f in ("node_visits","stall_exits","rays","rounds"):
    s=s.replace("GRT_D(%s, 1)"%f,"")
old="                    bool trip = wm != 0ull; // MODE 2: ONE trip, lanes = particles\n"
assert s.count(old)==1
s=s.replace(old,old+"                    uint32_t dg_t_ = 0, dg_b_ = 0, dg_l_ = 0, dg_r_ = 0;\n",1)
old="                            if (!m_) continue;\n"
assert s.count(old)==1
new=old+"""                            { const bool t_ = (m_ >> 32) == 0ull, b_ = (uint32_t)m_ == 0u;
                              const uint64_t lm_ = 0x0F0F0F0F0F0F0F0Full; const bool l_ = (m_ & ~lm_) == 0ull, r_ = (m_ & lm_) == 0ull;
                              dg_t_ += t_; dg_b_ += b_; dg_l_ += l_; dg_r_ += r_; }
"""
s=s.replace(old,new,1)
old="                    continue; // (the step is over: on to the next trip of the step loop)\n"
assert s.count(old)==1
s=s.replace(old,"                    GRT_D(rounds, dg_t_ + dg_b_) GRT_D(node_visits, min(dg_t_, dg_b_)) GRT_D(rays, dg_l_ + dg_r_) GRT_D(stall_exits, min(dg_l_, dg_r_))\n"+old,1)
open(p,'w').write(s)
PY
make -j8 OUT=$D/libgrt_hip.so EXTRA="-DGRT_TILE_DIAG" 2>&1 | grep -i "error\|moved behind" || true
cp $D/libgrt_hip.so /root/repo/gaussian-ray-tracing_amd/libgrt_hip_diag_r6.so
echo built diag_r6
