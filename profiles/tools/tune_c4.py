"""Sweep of the bounce pipeline's knobs on C4 (pixels never depend on them): ms per frame (wall, 10 frames after 3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python"))
import numpy as np, torch, grt, bench
wl = "C4"
seed, n, W, H, fisheye, with_mesh, mb, aniso = bench.WORKLOADS[wl]
acts, center, mesh = bench.build_scene(grt, wl)
p = grt.default_params(W, H, center, fisheye=fisheye, max_bounces=mb)
tr = grt.Tracer(0); tr.upload(acts)
if mesh is not None: tr.set_meshes([mesh])
def t():
    for _ in range(3): tr.render(p)
    tr.sync(); t0 = time.perf_counter()
    for _ in range(10): tr.render(p)
    tr.sync(); return (time.perf_counter() - t0) * 100.0
base = {grt.OPT_BUNDLE_BUDGET: 1024, grt.OPT_BUNDLE_ROUNDS: 2, grt.OPT_SINGLE_LOOKAHEAD: 256, grt.OPT_SINGLE_BAND: 256, grt.OPT_LANE_BUDGET: 128, grt.OPT_TILE_READY_MIN: 24}
names = {grt.OPT_BUNDLE_BUDGET: "bundle_budget", grt.OPT_BUNDLE_ROUNDS: "bundle_rounds", grt.OPT_SINGLE_LOOKAHEAD: "single_look", grt.OPT_SINGLE_BAND: "single_band", grt.OPT_LANE_BUDGET: "lane_budget", grt.OPT_TILE_READY_MIN: "ready_min"}
print("base", round(t(), 4))
sweeps = {grt.OPT_BUNDLE_BUDGET: (640, 704, 768, 832, 896, 960), grt.OPT_BUNDLE_ROUNDS: (1, 3), grt.OPT_SINGLE_LOOKAHEAD: (128, 512), grt.OPT_SINGLE_BAND: (128, 512), grt.OPT_LANE_BUDGET: (64, 256),
          grt.OPT_TILE_READY_MIN: (12, 16, 32)}
for opt, vals in sweeps.items():
    for v in vals:
        tr.set_option(opt, v)
        print(names[opt], v, round(t(), 4), flush=True)
    tr.set_option(opt, base[opt])
print("base again", round(t(), 4))
tr.set_option(grt.OPT_BUNDLE_ROUNDS, 1)
for v in (704, 768, 832, 896, 1024):
    tr.set_option(grt.OPT_BUNDLE_BUDGET, v); print("rounds 1 budget", v, round(t(), 4), flush=True)
