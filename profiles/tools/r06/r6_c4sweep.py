import sys, os, numpy as np
sys.path[:0] = ["gaussian-ray-tracing_amd/python", "."]
import grt, bench, torch
acts, center, mesh = bench.build_scene(grt, "C4")
tr = grt.Tracer(0); tr.upload(acts); tr.set_meshes([mesh])
for label, kw in (("mirror cap 2", dict(mesh_type=grt.MIRROR, max_bounces=2)), ("mirror cap 32", dict(mesh_type=grt.MIRROR, max_bounces=32)),
                  ("glass cap 2", dict(mesh_type=grt.GLASS, max_bounces=2)), ("glass cap 32", dict(mesh_type=grt.GLASS, max_bounces=32))):
    p = grt.default_params(1920, 1080, center, **kw)
    out = []
    for predict, budget in ((0, 896), (1, 896), (1, 1100), (1, 1300), (1, 1400), (1, 1500), (1, 1700)):
        tr.set_option(grt.OPT_BUNDLE_BUDGET, budget)
        tr.set_option(grt.OPT_BUNDLE_PREDICT, predict)
        ms = []
        for _ in range(9):
            tr.render(p); tr.sync()
            ms.append(tr.last_kernel_ms())
        out.append(f"{'on ' if predict else 'off'} {budget}: {ms[0]:.2f} / {float(np.median(ms[3:])):.3f}")
    print(label, "(first frame / steady):", " | ".join(out))
tr.check()
