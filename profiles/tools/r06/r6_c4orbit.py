import sys, os, numpy as np
sys.path[:0] = ["gaussian-ray-tracing_amd/python", "."]
import grt, bench, torch
acts, center, mesh = bench.build_scene(grt, "C4")
W, H = 1920, 1080
p = grt.default_params(W, H, center, mesh_type=grt.MIRROR, max_bounces=2)
eye0 = np.float32(list(p.eye)) - center
def orbit(i, step=1.5):
    ang = np.deg2rad(step * (i + 1))
    eye = center + np.float32([eye0[0] * np.cos(ang) + eye0[2] * np.sin(ang), eye0[1], -eye0[0] * np.sin(ang) + eye0[2] * np.cos(ang)])
    return grt.default_params(W, H, center, mesh_type=grt.MIRROR, max_bounces=2, eye=tuple(float(x) for x in eye))
for pred in (0, 1):
    tr = grt.Tracer(0); tr.set_option(grt.OPT_BUNDLE_PREDICT, pred); tr.upload(acts); tr.set_meshes([mesh])
    ms = []
    for _ in range(8): tr.render(p); tr.sync(); ms.append(tr.last_kernel_ms())
    out = [f"standing {np.median(ms[3:]):.3f}"]
    for step in (1.5, 0.3):
        mm = []
        for i in range(24): tr.render(orbit(i, step)); tr.sync(); mm.append(tr.last_kernel_ms())
        out.append(f"orbit {step} deg/frame: median {np.median(mm[4:]):.3f} max {max(mm[4:]):.3f}")
        # a standing view at the END of the orbit, then back home
        q = orbit(23, step); mm = []
        for _ in range(14): tr.render(q); tr.sync(); mm.append(tr.last_kernel_ms())
        out.append("then standing there: " + " ".join(f"{x:.2f}" for x in mm))
    mm = []
    for _ in range(14): tr.render(p); tr.sync(); mm.append(tr.last_kernel_ms())
    out.append("back home: " + " ".join(f"{x:.2f}" for x in mm))
    print("predict", pred, "|", " | ".join(out)); tr.check(); tr.close()
