import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python"))
import numpy as np, torch, grt, bench
for wl in sys.argv[1:]:
    seed, n, W, H, fisheye, with_mesh, mb, aniso = bench.WORKLOADS[wl]
    acts, center, mesh = bench.build_scene(grt, wl)
    p = grt.default_params(W, H, center, fisheye=fisheye, max_bounces=mb)
    ref = None
    for sweeps in (0, 1, 2, 3, 5):
        tr = grt.Tracer(0); tr.set_option(grt.OPT_BVH_ROTATIONS, sweeps); tr.upload(acts)
        info = tr.bvh_info()
        ms = []
        for _ in range(8):
            u8, _ = tr.render(p); tr.sync(); ms.append(tr.last_kernel_ms())
        if ref is None: ref = u8.clone()
        tr.set_option(grt.OPT_COUNTERS, 1); tr.render(p); c = tr.counters(); tr.check()
        print(wl, "sweeps", sweeps, "height", info["height"], "prims", info["n_primitives"], "build ms", round(info["build_ms"], 2), "kernel ms", round(float(np.median(ms[3:])), 3),
              "tests/ray", round(c["proxy_tests"] / c["rays"], 1), "boxes/ray", round(c["node_visits"] / c["rays"], 1), "same", bool((u8 == ref).all()))
        tr.close()
