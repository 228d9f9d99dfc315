import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python"))
import numpy as np, torch, grt, bench
for wl in sys.argv[1:]:
    seed, n, W, H, fisheye, with_mesh, mb, aniso = bench.WORKLOADS[wl]
    acts, center, mesh = bench.build_scene(grt, wl)
    p = grt.default_params(W, H, center, fisheye=fisheye, max_bounces=mb)
    trs = {}
    for sweeps in (0, 1):
        tr = grt.Tracer(0); tr.set_option(grt.OPT_BVH_ROTATIONS, sweeps); tr.upload(acts)
        if mesh is not None: tr.set_meshes([mesh])
        for _ in range(5): tr.render(p); tr.sync()
        trs[sweeps] = tr
    res = {0: [], 1: []}
    for rep in range(4):
        for sweeps in (0, 1):
            ms = []
            for _ in range(10):
                trs[sweeps].render(p); trs[sweeps].sync(); ms.append(trs[sweeps].last_kernel_ms())
            res[sweeps].append(float(np.median(ms)))
    print(wl, "no rotation:", [round(x, 4) for x in res[0]], "one sweep:", [round(x, 4) for x in res[1]], "ratio", round(np.mean(res[1]) / np.mean(res[0]), 4))
    # orbit leg: moving camera, 20 frames
    eye0 = np.float32(list(p.eye)) - center
    for sweeps in (0, 1):
        mm = []
        for i in range(24):
            ang = np.deg2rad(1.5 * (i + 1))
            eye = center + np.float32([eye0[0] * np.cos(ang) + eye0[2] * np.sin(ang), eye0[1], -eye0[0] * np.sin(ang) + eye0[2] * np.cos(ang)])
            q = grt.default_params(W, H, center, fisheye=fisheye, max_bounces=mb, eye=tuple(float(x) for x in eye))
            trs[sweeps].render(q); trs[sweeps].sync(); mm.append(trs[sweeps].last_kernel_ms())
        print("   orbit, sweeps", sweeps, round(float(np.median(mm[4:])), 4))
    for t in trs.values(): t.close()
