#!/usr/bin/env python3
"""Summarises one profiles/collect.sh run into the small files kept under profiles/ (the raw rocprofv3 output stays
in gpurun_out/).  Counters are averaged over the timed-loop dispatches of the render kernel (the instrumented and the
cold-order frames bench.py runs first are skipped: only dispatches after the first 4 are used)."""
import csv, glob, json, os, shutil, sys, collections

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
KERNEL = "k_render_stream<false, false, false>"

for f in glob.glob(src + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if "rocprim" not in r[0] and float(r[4]) >= 0.05][:12]
    with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w", newline="") as o:
        csv.writer(o).writerows(keep)
for line in open(src + "/bench_under_rocprof.log"):
    if line.startswith("{"):
        open(os.path.join(here, f"{tag}_bench_under_rocprof.json"), "w").write(json.dumps(json.loads(line), indent=1) + "\n")

out = {}
for f in glob.glob(src + "/pmc*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            acc[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for name, d in acc.items():
        v = [d[k] for k in sorted(d)][4:]
        if v:
            out[name] = sum(v) / len(v)
res = {"kernel": "grt::" + KERNEL, "workload": "C3 (bench.py default), steady-state frames", "per_dispatch": out}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    res["hbm_bytes_per_launch"] = int((2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024)  # MI355X_MICROARCH.md §HBM: gfx950 FETCH_SIZE x2
    res["hbm_bytes_per_launch_raw"] = int((out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024)
if "TCC_HIT_sum" in out:
    res["l2_hit_rate"] = round(out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"]), 4)
if "SQ_WAVES" in out and "SQ_INSTS_VALU" in out:
    w = out["SQ_WAVES"]
    res["per_wave"] = {k: round(v / w, 1) for k, v in out.items() if k.startswith("SQ_")}
    # SQ cycle counters tick once per 4 clocks; a SIMD holds 4 of these waves (128 VGPRs): the VALU is busy for
    # 4 * ACTIVE_INST_VALU of every WAVE_CYCLES a wave is resident
    res["valu"] = {"valu_insts_per_wave": round(out["SQ_INSTS_VALU"] / w), "salu_insts_per_wave": round(out["SQ_INSTS_SALU"] / w),
                   "valu_busy_frac": round(4 * out["SQ_ACTIVE_INST_VALU"] / out["SQ_WAVE_CYCLES"], 3),
                   "note": "4 waves/SIMD x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, rocprofv3 --pmc, same command"}
json.dump(res, open(os.path.join(here, f"{tag}_counters.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
