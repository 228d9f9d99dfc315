import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python"))
import numpy as np, torch, grt, bench
W, H = 1920, 1080
for sigma in (0.0, 0.6, 0.7, 0.85, 1.0, 1.2, 1.4, 1.6):
    raw = grt.synth_scene(3, 1_000_000)
    if sigma:
        rng = np.random.default_rng(1003)
        raw["scale"] = (raw["scale"] + rng.normal(0.0, sigma, size=raw["scale"].shape)).astype(np.float32)
    acts = grt.activate(raw); center = grt.gaussian_center(acts["pos"])
    p = grt.default_params(W, H, center)
    out = []
    for split in (-1,):
        tr = grt.Tracer(0); tr.set_option(grt.OPT_SPLIT, split); tr.upload(acts)
        info = tr.bvh_info()
        ms = []
        for _ in range(7):
            tr.render(p); tr.sync(); ms.append(tr.last_kernel_ms())
        out.append(f"{'auto' if split < 0 else 'split 8'}: {float(np.median(ms[3:])):.3f} ms (prims x{info['n_primitives'] / info['n_proxies']:.3f}, build {info['build_ms']:.1f} ms)")
        tr.check(); tr.close()
    print("sigma", sigma, " | ".join(out))
