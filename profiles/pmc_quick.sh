#!/bin/bash
# quick SQ counter pass of one bench command:  profiles/pmc_quick.sh <tag> <kernel-name-substring> <bench args...>
# (separate --pmc passes, no trace domains combined with them)
TAG=$1; KSUB=$2; shift 2
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
cd /tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAIT_INST_ANY" \
         "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $O/pmc$i -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/pmc$i.log 2>&1 || exit 1
done
cd $R
python3 - "$O" "$KSUB" <<'PY'
import csv, glob, sys, collections, json
src, ksub = sys.argv[1], sys.argv[2]
out = {}
for f in glob.glob(src + "/pmc*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            acc[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for name, d in acc.items():
        v = [d[k] for k in sorted(d)][4:]
        if v: out[name] = sum(v) / len(v)
w = out.get("SQ_WAVES", 1)
print(json.dumps({k: round(v / w, 1) for k, v in sorted(out.items())}, indent=1))
json.dump(out, open(src + "/summary.json", "w"), indent=1)
PY
