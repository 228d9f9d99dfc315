#!/bin/bash
# extra SQ / SQC counter passes of the default bench frame (instruction cache, scalar cache, LDS, latencies):
#   profiles/tools/pmc_extra.sh <tag> <kernel-name-substring> <bench args...>    -> gpurun_out/pmcx_<tag>/summary.json
TAG=$1; KSUB=$2; shift 2
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/pmcx_$TAG
rm -rf $O; mkdir -p $O
cd /tmp
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_REQ" \
         "SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
         "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $O/pmc$i -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs "$@" > $O/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/pmc$i.log; }
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
        v = [d[k] for k in sorted(d)][2:]
        if v: out[name] = sum(v) / len(v)
json.dump(out, open(src + "/summary.json", "w"), indent=1)
print(json.dumps({k: round(v, 1) for k, v in sorted(out.items())}, indent=1))
PY
