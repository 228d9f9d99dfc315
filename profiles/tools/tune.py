"""Sweep of the tile kernel's scheduling knobs on one workload (pixels never depend on them): kernel ms, median of 8."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python"))
import numpy as np, torch, grt, bench
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
seed, n, W, H, fisheye, with_mesh, mb, aniso = bench.WORKLOADS[wl]
acts, center, mesh = bench.build_scene(grt, wl)
p = grt.default_params(W, H, center, fisheye=fisheye, max_bounces=mb)
tr = grt.Tracer(0); tr.upload(acts)
if mesh is not None: tr.set_meshes([mesh])
def t():
    for _ in range(3): tr.render(p)
    ms = []
    for _ in range(8): tr.render(p); ms.append(tr.last_kernel_ms())
    return float(np.median(ms))
base = {grt.OPT_TILE_READY_MIN: 24, grt.OPT_TILE_BAND: 64, grt.OPT_TILE_LOOKAHEAD: 64, grt.OPT_TILE_RESERVE: 24, grt.OPT_SWIZZLE: 2, grt.OPT_COST_RADIUS: 4}
names = {grt.OPT_TILE_READY_MIN: "ready_min", grt.OPT_TILE_BAND: "band", grt.OPT_TILE_LOOKAHEAD: "look", grt.OPT_TILE_RESERVE: "reserve", grt.OPT_SWIZZLE: "swizzle", grt.OPT_COST_RADIUS: "cost_radius"}
print("base", round(t(), 4))
sweeps = {grt.OPT_TILE_READY_MIN: (4, 8, 12, 24, 32), grt.OPT_TILE_BAND: (16, 32, 96, 128, 192), grt.OPT_TILE_LOOKAHEAD: (16, 32, 96, 128, 192),
          grt.OPT_TILE_RESERVE: (8, 16, 32, 40), grt.OPT_SWIZZLE: (0, 1, 4, 8)}
for opt, vals in sweeps.items():
    for v in vals:
        tr.set_option(opt, v)
        print(names[opt], v, round(t(), 4), flush=True)
    tr.set_option(opt, base[opt])
print("base again", round(t(), 4))
