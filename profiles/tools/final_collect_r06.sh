#!/bin/bash
# Round 6: every workload's profile re-collected with ONE build — run on the GPU box from the repo root in two
# calls (a gpurun call is limited to 20 minutes):  profiles/tools/final_collect_r06.sh a | b
# Results come back through gpurun_out/profiles_r06/ (only gpurun_out/ is merged back): copy them into profiles/ afterwards.
set -e
PART=${1:-a}
mkdir -p gpurun_out/r6 gpurun_out/profiles_r06
if [ "$PART" = "a" ]; then
  bash profiles/collect.sh r06 > gpurun_out/r6/collect_C3.log 2>&1; echo "collected C3"
  for w in C2 C1 C4; do bash profiles/collect.sh r06 $w > gpurun_out/r6/collect_$w.log 2>&1; echo "collected $w"; done
  bash profiles/collect.sh r06 C3 "" sh3 --sh-degree 3 > gpurun_out/r6/collect_sh3.log 2>&1; echo "collected sh3"
else
  for w in C5 C3a C3b; do bash profiles/collect.sh r06 $w > gpurun_out/r6/collect_$w.log 2>&1; echo "collected $w"; done
  bash profiles/run_all_workloads.sh r06 > gpurun_out/r6/run_all.log 2>&1; tail -14 gpurun_out/r6/run_all.log
fi
cp profiles/r06* profiles/traffic.json gpurun_out/profiles_r06/ 2>/dev/null; echo copied
