#!/bin/bash
# profiles/tools/movcount.sh [flags...]: static count of compiler-made 64-bit register copies (v_mov_b64_e32), spills and
# size of the C3 kernel (k_render_tile<false,false,false,0,false>) per marked piece of the current source
cd /root/repo/gaussian-ray-tracing_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden -Wno-unused-function -DGRT_MARKS "$@" -S --cuda-device-only -o /tmp/cs/mc.s grt_render_tile.hip 2>/dev/null
python3 - <<'PY'
import re
L=open('/tmp/cs/mc.s').read().split('\n')
s=next(i for i,l in enumerate(L) if l.startswith('_ZN3grt12_GLOBAL__N_113k_render_tileILb0ELb0ELb0ELi0ELb0EEEvNS_10RenderArgsE:'))
e=next(i for i in range(s,len(L)) if L[i].strip().startswith('s_endpgm'))
piece='prologue'; cnt={}; order=[]
for l in L[s:e]:
    t=l.strip()
    m=re.match(r'; GRT_MARK (\w+)',t)
    if m:
        piece=m.group(1)+('2' if m.group(1) in cnt else ''); continue
    if piece not in cnt: cnt[piece]=[0,0,0,0,0]; order.append(piece)
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
    op=t.split()[0]
    cnt[piece][0]+=1
    if op=='v_mov_b64_e32' and not t.endswith('-1'): cnt[piece][1]+=1
    if op.startswith('scratch_'): cnt[piece][2]+=1
    if op.startswith('v_') : cnt[piece][3]+=1
    if op=='s_nop': cnt[piece][4]+=1+int(t.split()[1])
for p in order: print(f"{p:14s} instr={cnt[p][0]:5d} valu={cnt[p][3]:5d} mov64_e32={cnt[p][1]:3d} scratch={cnt[p][2]:3d} nop_waitstates={cnt[p][4]:3d}")
print('total', sum(c[0] for c in cnt.values()), 'mov64_e32', sum(c[1] for c in cnt.values()), 'scratch', sum(c[2] for c in cnt.values()), 'nop_ws', sum(c[4] for c in cnt.values()))
for l in L[e:e+80]:
    if 'vgpr_spill_count' in l or 'NumVgprs' in l or 'ScratchSize' in l or '.vgpr_count' in l: print(l.strip())
PY
