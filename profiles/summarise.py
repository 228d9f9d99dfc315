#!/usr/bin/env python3
"""Summarises one profiles/collect.sh run into the small files kept under profiles/ (the raw rocprofv3 output stays
in gpurun_out/).  Counters are averaged over the timed-loop dispatches of the render kernel (the instrumented frame
bench.py runs first is skipped: only dispatches after the first 2 are used)."""
import csv, datetime, glob, json, os, sys, collections

src, tag = sys.argv[1], sys.argv[2]
ROUND = tag.split("_")[0]  # "r06" of "r06", "r06pre", "r06_C4" ...: the round the counters were collected in (bench.py checks it)
wl = sys.argv[3] if len(sys.argv) > 3 else "C3"
label = sys.argv[4] if len(sys.argv) > 4 else ""
here = os.path.dirname(os.path.abspath(__file__))
if wl != "C3":
    tag = f"{tag}_{wl}"
if label:
    tag = f"{tag}_{label}"
bench_line = None
for line in open(src + "/bench_under_rocprof.log"):
    if line.startswith("{"):
        bench_line = json.loads(line)
# the dominant kernel is the one the bench line names (tile kernel: <COUNT, SH, MESH, MODE, PIECES>)
KERNEL = bench_line["roofline"]["kernel"].replace("grt::", "") if bench_line else "k_render_tile<false, false, false, 0, false>"
# mesh frames: the stages of the wavefront pipeline are kernels of their own
STAGES = ["k_primary_mesh<false>", "k_primary_mesh_wave<false>", "k_render_tile<false, false, true, 0,", "k_queue_mesh<false>", "k_render_tile<false, false, true, 1,",
          "k_render_tile<false, false, true, 2,", "k_bounce<false>"] if "true, 0" in KERNEL else []

for f in glob.glob(src + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if "rocprim" not in r[0] and float(r[4]) >= 0.05][:16]
    with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w", newline="") as o:
        csv.writer(o).writerows(keep)
for line in open(src + "/bench_under_rocprof.log"):
    if line.startswith("{"):
        open(os.path.join(here, f"{tag}_bench_under_rocprof.json"), "w").write(json.dumps(json.loads(line), indent=1) + "\n")

# Steady state of the dominant kernel from the per-dispatch trace: rocprofv3's --stats average runs over EVERY dispatch of the
# process, the cold first frames of a context included (round 3: average 1.831 ms, max 2.10, above the driver's ms_per_step);
# the timed loop of the bench command is the LAST `steps` dispatches of the kernel.
steady = None
for f in glob.glob(src + "/trace/**/*kernel_trace.csv", recursive=True):
    d = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"]]
    d = [x[1] for x in sorted(d)]
    n = bench_line["steps"] if bench_line else 20
    if len(d) >= n:
        t = sorted(d[-n:])
        steady = {"kernel": "grt::" + KERNEL, "dispatches_in_trace": len(d), "timed_loop_dispatches": n, "average_ms": round(sum(t) / n, 4),
                  "median_ms": round(t[n // 2], 4), "min_ms": round(t[0], 4), "max_ms": round(t[-1], 4),
                  "all_dispatches_average_ms": round(sum(d) / len(d), 4),
                  "bench_kernel_ms_hip_events": bench_line.get("kernel_ms") if bench_line else None,
                  "bench_ms_per_step": bench_line.get("ms_per_step") if bench_line else None,
                  "note": "rocprofv3 --kernel-trace of `bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs`; the last 20 dispatches of "
                          "the kernel are the timed loop (the earlier ones: first frames of the context, instrumented frame, warm-up)"}
        json.dump(steady, open(os.path.join(here, f"{tag}_kernel_steady.json"), "w"), indent=1)

def per_kernel(pattern, name_filter, skip):
    out = {}
    for f in glob.glob(pattern, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if name_filter(r["Kernel_Name"], r):
                acc[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        for name, d in acc.items():
            v = [d[k] for k in sorted(d)][skip:]
            if v:
                out[name] = sum(v) / len(v)
    return out

out = per_kernel(src + "/pmc*/**/*counter_collection.csv", lambda n, r: KERNEL in n, 2)
res = {"kernel": "grt::" + KERNEL, "workload": f"{wl} (bench.py --workload {wl}), steady-state frames", "collected": datetime.date.today().isoformat(),
       "per_dispatch": out}
if STAGES:  # SQ counters of every stage of the wavefront pipeline (per dispatch, averaged over the steady-state frames)
    res["stages"] = {}
    for st in STAGES:
        o = per_kernel(src + "/pmc*/**/*counter_collection.csv", lambda n, r, st=st: st in n, 3)
        if o.get("SQ_WAVES"):
            w_ = o["SQ_WAVES"]
            res["stages"][st] = {"waves_per_dispatch": round(w_, 1), "per_wave": {k: round(v / w_, 1) for k, v in o.items() if k.startswith("SQ_")}}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    res["hbm_bytes_per_launch"] = int((2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024)  # MI355X_MICROARCH.md §HBM: gfx950 FETCH_SIZE x2
    res["hbm_bytes_per_launch_raw"] = int((out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024)
if "TCC_HIT_sum" in out:
    res["l2_hit_rate"] = round(out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"]), 4)
calib = None
cpath = src + "/valu_calib.json"
if os.path.exists(cpath):
    cj = json.load(open(cpath))
    json.dump(cj, open(os.path.join(here, f"{tag}_valu_calib.json"), "w"), indent=1)
    calib = {(c["inst"], c["waves_per_simd"]): c for c in cj["cases"]}
if "SQ_WAVES" in out and "SQ_INSTS_VALU" in out:
    w = out["SQ_WAVES"]
    res["per_wave"] = {k: round(v / w, 1) for k, v in out.items() if k.startswith("SQ_")}
    # What is 100 %?  The calibration microbenchmark (profiles/calib/valu_calib.hip) at the SAME residency (4 waves
    # per SIMD): a SIMD issues one wave64 VALU instruction every 3.1 cycles from dependent chains, one every 2.0 from
    # independent ones; this kernel's waves issue one every `cycles_per_valu_inst_per_simd` (SQ_WAVE_CYCLES counts
    # quad-cycles of ONE wave; 4 waves share the SIMD).
    cyc = 4.0 * out["SQ_WAVE_CYCLES"] / out["SQ_INSTS_VALU"] / 4.0
    if "SQ_WAIT_ANY" in out and "SQ_ACTIVE_INST_ANY" in out and "SQ_WAIT_INST_ANY" in out:
        wc = out["SQ_WAVE_CYCLES"]
        res["wave_time_split"] = {"issuing": round(out["SQ_ACTIVE_INST_ANY"] / wc, 3), "issue_stalled": round(out["SQ_WAIT_INST_ANY"] / wc, 3),
                                  "waiting_on_waitcnt": round(out["SQ_WAIT_ANY"] / wc, 3),
                                  "note": "fractions of SQ_WAVE_CYCLES (the three are disjoint, MI355X_MICROARCH.md)"}
    # (since round 4 the heaviest tiles run as 2 / 4 part waves: a launch has more waves than 8x8 tiles, so the per-WAVE
    #  averages fell without the frame's work changing — the per-TILE figures and the totals are the comparable ones)
    try:
        sys.path.insert(0, os.path.dirname(here))
        import bench as _b
        _w = _b.WORKLOADS[wl]
        n_tiles = ((_w[2] + 7) // 8) * ((_w[3] + 7) // 8)
    except Exception:
        n_tiles = None
    v = {"valu_insts_per_wave": round(out["SQ_INSTS_VALU"] / w), "salu_insts_per_wave": round(out["SQ_INSTS_SALU"] / w),
         "waves_per_launch": round(w), "tiles_8x8_per_launch": n_tiles,
         "valu_insts_per_launch_millions": round(out["SQ_INSTS_VALU"] / 1e6, 1), "salu_insts_per_launch_millions": round(out["SQ_INSTS_SALU"] / 1e6, 1),
         "valu_insts_per_tile": round(out["SQ_INSTS_VALU"] / n_tiles) if n_tiles else None,
         "salu_insts_per_tile": round(out["SQ_INSTS_SALU"] / n_tiles) if n_tiles else None,
         "waves_per_simd": 4,
         "cycles_per_valu_inst_per_simd": round(cyc, 2),
         "raw_metric_4x_ACTIVE_INST_VALU_over_WAVE_CYCLES": round(4 * out["SQ_ACTIVE_INST_VALU"] / out["SQ_WAVE_CYCLES"], 3)}
    v["valu_pipe_utilisation"] = round(v["raw_metric_4x_ACTIVE_INST_VALU_over_WAVE_CYCLES"] / 2.0, 3)  # the SIMD-32 retires a wave64 op in 2 cycles
    if calib:
        dep = calib[("v_fma_f32_dependent", 4)]["cyc_per_inst_one_wave"] / 4.0
        ind = calib[("v_fma_f32", 4)]["cyc_per_inst_one_wave"] / 4.0
        mix = calib[("v_fma_f32+s_add_u32 interleaved (per pair)", 4)]["cyc_per_inst_one_wave"] / 4.0
        v["calibration_cycles_per_inst_per_simd_at_4_waves"] = {"dependent_valu": round(dep, 2), "independent_valu": round(ind, 2),
                                                                 "valu_salu_pair": round(mix, 2)}
        v["valu_issue_frac_of_dependent_chain_rate"] = round(dep / cyc, 3)
        v["valu_issue_frac_of_independent_rate"] = round(ind / cyc, 3)
    cal = per_kernel(src + "/calib_pmc/**/*counter_collection.csv", lambda n, r: "k_fma_dep" in n or n.startswith("k_fma("), 0)
    if cal.get("SQ_WAVE_CYCLES"):
        v["raw_metric_of_the_saturating_microkernels_all_W"] = round(4 * cal["SQ_ACTIVE_INST_VALU"] / cal["SQ_WAVE_CYCLES"], 3)
    v["note"] = ("rocprofv3 --pmc of the same command.  The raw metric round 1 quoted as 'VALU busy' (n_waves x SQ_ACTIVE_INST_VALU / "
                 "SQ_WAVE_CYCLES) tops out at 2.0, not 1.0: the SIMD-32 retires a wave64 instruction every 2 cycles, so the VALU PIPE "
                 "utilisation is half of it (valu_pipe_utilisation).  What a frame costs is its instruction count: adding work to every "
                 "exact test (profiles/r03_sensitivity.json) gives +0.036 ms per 1000 VALU and +0.028 ms per 1000 SALU instructions per "
                 "wave and 0.0009 ms per dependent scalar round trip, which reproduces the frame from these counters; round 3 took "
                 "40.1 k -> ~37 k VALU and 20.2 k -> ~16 k SALU per wave out of the kernel (DESIGN.md 5.2, 6) "
                 "(calibration: profiles/calib/valu_calib.hip, dep_dist.hip)")
    res["valu"] = v
json.dump(res, open(os.path.join(here, f"{tag}_counters.json"), "w"), indent=1)
tpath = os.path.join(here, "traffic.json")
tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
sh_deg = 3 if label == "sh3" else 0
tj[f"{wl}_sh{sh_deg}_k0_n1"] = {"hbm_bytes_per_launch": res.get("hbm_bytes_per_launch"), "valu": res.get("valu"), "collected": res["collected"],
                      "round": ROUND, "kernel": res["kernel"], "source": f"profiles/{tag}_counters.json"}
json.dump(tj, open(tpath, "w"), indent=1)
print(json.dumps(res, indent=1))
