mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4/t19_full.log 2>&1; tail -3 gpurun_out/r4/t19_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/smoke19.log 2>&1; tail -1 gpurun_out/r4/smoke19.log
bash profiles/collect.sh r04 > gpurun_out/r4/collect_C3.log 2>&1; echo "collected C3"
for w in C2 C1 C4 C5 C3a; do bash profiles/collect.sh r04 $w > gpurun_out/r4/collect_$w.log 2>&1; echo "collected $w"; done
bash profiles/collect.sh r04 C3 "" sh3 --sh-degree 3 > gpurun_out/r4/collect_sh3.log 2>&1; echo "collected sh3"
bash profiles/run_all_workloads.sh r04 > gpurun_out/r4/run_all.log 2>&1; tail -12 gpurun_out/r4/run_all.log
mkdir -p gpurun_out/r4/profiles_r04e && cp profiles/r04* profiles/traffic.json gpurun_out/r4/profiles_r04e/ 2>/dev/null; echo copied
