import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python"))
import numpy as np, torch, grt, bench
W, H = 1920, 1080
sigmas = [float(x) for x in sys.argv[1:]]
for sigma in sigmas:
    raw = grt.synth_scene(3, 1_000_000)
    if sigma:
        rng = np.random.default_rng(1003)
        raw["scale"] = (raw["scale"] + rng.normal(0.0, sigma, size=raw["scale"].shape)).astype(np.float32)
    acts = grt.activate(raw); center = grt.gaussian_center(acts["pos"])
    p = grt.default_params(W, H, center)
    for vol in (400,):
        out = []
        for split in (6, 8, 10, 12, 16, 24):
            tr = grt.Tracer(0); tr.set_option(grt.OPT_SPLIT, split); tr.set_option(grt.OPT_SPLIT_VOL_PCT, vol); tr.upload(acts)
            info = tr.bvh_info(); ms = []
            for _ in range(6): tr.render(p); tr.sync(); ms.append(tr.last_kernel_ms())
            out.append(f"{split}: {float(np.median(ms[3:])):.3f} (x{info['n_primitives'] / info['n_proxies']:.2f})")
            tr.check(); tr.close()
        print("sigma", sigma, "vol%", vol, "|", " | ".join(out), flush=True)
