#!/bin/bash
# Runs bench.py on every workload (one GPU) and on the emulated multi-GPU rank layouts; writes profiles/<tag>_bench*.json
# (through gpurun only gpurun_out/ comes back from the GPU box: the raw lines are in gpurun_out/all_<tag>/*.json and the
#  python block at the end can be re-run on them locally; profiles/collect.sh likewise: re-run profiles/summarise.py on
#  gpurun_out/collect_<tag>[_<workload>])
TAG=${1:-rXX}
O=gpurun_out/all_$TAG
mkdir -p $O
python3 bench.py > $O/C3.json 2> $O/C3.err
for w in C1 C2 C3a C3b C4 C5; do python3 bench.py --no-cpu-baseline --workload $w > $O/$w.json 2> $O/$w.err; echo "$w done"; done
python3 bench.py --no-cpu-baseline --sh-degree 3 > $O/C3_sh3.json 2> $O/C3_sh3.err
python3 bench.py --no-cpu-baseline --kernel 3 --no-extra-legs > $O/C3_stream_kernel.json 2> $O/C3_stream_kernel.err
for n in 2 4 8; do python3 bench.py --no-cpu-baseline --emulate-ranks $n > $O/C3_emulated_rank_of_$n.json 2> $O/C3_emulated_rank_of_$n.err; echo "ranks $n done"; done
python3 - "$O" "$TAG" <<'PY'
import json, sys, os, glob
src, tag = sys.argv[1], sys.argv[2]
allj = {}
for f in sorted(glob.glob(src + "/*.json")):
    lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
    if lines: allj[os.path.basename(f)[:-5]] = json.loads(lines[-1])
json.dump(allj.get("C3"), open(f"profiles/{tag}_bench.json", "w"), indent=1)
json.dump({k: v for k, v in allj.items() if k != "C3"}, open(f"profiles/{tag}_bench_other_workloads.json", "w"), indent=1)
for k, j in allj.items():
    c = j["config"]
    print(f"{k:28s} value {j['value']:9.1f} Mrays/s  ms/frame {j['ms_per_step']:8.3f}  kernel {j['kernel_ms']:8.3f}  cold {c.get('kernel_ms_cold')}  orbit {c.get('kernel_ms_orbit')}  sync/pipe {c.get('value_sync')}/{c.get('value_pipelined')}  lat {c.get('latency_ms_per_frame')}  passes {c['rounds_per_ray']} tests/ray {c['proxy_tests_per_ray']} boxes/ray {c['node_visits_per_ray']}")
PY
