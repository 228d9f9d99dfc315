#!/bin/bash
# Collects the profiles this directory holds (run on the GPU box from the repo root):
#   profiles/collect.sh <tag> [workload] [quick]      e.g.  profiles/collect.sh r04        (C3, everything)
#                                                           profiles/collect.sh r03 C4 quick
# 1. rocprofv3 --kernel-trace --stats of the bench command without its untimed extra legs (cold / orbit frames, the
#    pipelined loop: their overlapping launches would pollute the per-kernel averages)
#                                              -> <tag>[_<workload>]_kernel_stats.csv, ..._bench_under_rocprof.json
# 2. separate --pmc passes (no trace domains combined with them): FETCH_SIZE | WRITE_SIZE | TCC hit/miss/EA | SQ (2 passes)
#    of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs`  -> ..._counters.json (via summarise.py)
#    ("quick": only the two SQ passes)
# 3. (C3, not quick) the issue-rate calibration microbenchmark (profiles/calib/valu_calib) alone and under the SQ --pmc pass
#                                              -> <tag>_valu_calib.json, calibration block of <tag>_counters.json
set -e
TAG=${1:-rXX}
WL=${2:-C3}
QUICK=${3:-}
LABEL=${4:-}            # optional: suffix for a variant of the workload, the remaining arguments go to bench.py
for _ in 1 2 3 4; do shift 2>/dev/null || true; done   #   e.g. profiles/collect.sh r03 C3 "" sh3 --sh-degree 3
BARGS="$@"
R=$PWD
export TMPDIR=/tmp
SUF=""; [ "$WL" != "C3" ] && SUF="_$WL"
[ -n "$LABEL" ] && SUF="${SUF}_$LABEL"
O=$R/gpurun_out/collect_$TAG$SUF
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs $BARGS > $O/bench_under_rocprof.log 2>&1
echo "trace done"
i=0
PASSES=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAIT_INST_ANY" \
        "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE")
for C in "${PASSES[@]}"; do
  i=$((i+1))
  if [ -n "$QUICK" ] && [ $i -le 3 ]; then continue; fi
  rocprofv3 --pmc $C --output-format csv -d $O/pmc$i -o p -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs $BARGS > $O/pmc$i.log 2>&1
  echo "pass $i done: $C"
done
if [ "$WL" = "C3" ] && [ -z "$QUICK" ] && [ -z "$LABEL" ]; then
  $R/profiles/calib/valu_calib > $O/valu_calib.json
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/calib_pmc -o p -- $R/profiles/calib/valu_calib > $O/calib_pmc.log 2>&1
  echo "calibration done"
fi
cd $R
python3 profiles/summarise.py $O $TAG $WL $LABEL
