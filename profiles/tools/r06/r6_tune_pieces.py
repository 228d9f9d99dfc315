import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python"))
import numpy as np, torch, grt, bench
for wl in sys.argv[1:]:
    seed, n, W, H, fisheye, with_mesh, mb, aniso = bench.WORKLOADS[wl]
    acts, center, mesh = bench.build_scene(grt, wl)
    p = grt.default_params(W, H, center, fisheye=fisheye, max_bounces=mb)
    def run(opts, build_opts=()):
        tr = grt.Tracer(0)
        for k, v in build_opts: tr.set_option(k, v)
        tr.upload(acts)
        for k, v in opts: tr.set_option(k, v)
        ms = []
        for _ in range(7): tr.render(p); tr.sync(); ms.append(tr.last_kernel_ms())
        tr.check(); tr.close()
        return float(np.median(ms[3:]))
    print(wl, "default", round(run(()), 3))
    for sw in (0, 2): print(wl, "rotation sweeps", sw, round(run((), ((grt.OPT_BVH_ROTATIONS, sw),)), 3))
    for r in (8, 16, 32, 40): print(wl, "reserve", r, round(run(((grt.OPT_TILE_RESERVE, r),)), 3))
    for b in (0, 128, 256, 1024, 2048): print(wl, "band_abs", b, round(run(((grt.OPT_TILE_BAND_ABS, b),)), 3))
    for lk in (32, 128): print(wl, "lookahead", lk, round(run(((grt.OPT_TILE_LOOKAHEAD, lk),)), 3))
    for bd in (32, 128): print(wl, "band", bd, round(run(((grt.OPT_TILE_BAND, bd),)), 3))
