"""Soak of the overflow pool's size classes: random camera cuts, orbits and standing phases on a scene with deep and shallow tiles;
every frame must equal a tracer that gives every tile a full bag (GRT_OPT_OVF_CLASSES = 0), and no frame may report an error."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grt
from common import synth
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
n, W, H = 300000, 960, 544
raw, acts = synth(11, n, 0.3)
center = grt.gaussian_center(acts["pos"])
a = grt.Tracer(0); a.upload(acts)
b = grt.Tracer(0); b.upload(acts); b.set_option(grt.OPT_OVF_CLASSES, 0)
bad = 0; frames = 0; t0 = time.time(); mem = []
eye = np.float32(center + np.float32([0.5, 0.4, 3.0]))
for phase in range(40):
    kind = rng.integers(0, 3)
    if kind == 0:  # camera cut
        d = rng.normal(size=3).astype(np.float32); d /= np.linalg.norm(d)
        eye = np.float32(center + d * np.float32(rng.uniform(0.3, 3.5)))
        steps, dang = 1, 0.0
    elif kind == 1:  # orbit
        steps, dang = int(rng.integers(4, 14)), float(rng.uniform(0.005, 0.06))
    else:  # stand
        steps, dang = int(rng.integers(6, 20)), 0.0
    for i in range(steps):
        if dang:
            e0 = eye - center
            eye = np.float32(center + np.float32([e0[0] * np.cos(dang) + e0[2] * np.sin(dang), e0[1], -e0[0] * np.sin(dang) + e0[2] * np.cos(dang)]))
        p = grt.default_params(W, H, center, eye=tuple(float(x) for x in eye))
        x8, xf = a.render(p, want_f32=True)
        y8, yf = b.render(p, want_f32=True)
        if rng.integers(0, 3) != 0: a.sync()  # (some frames queued without waiting)
        torch.cuda.synchronize()
        if not ((x8 == y8).all() and (xf == yf).all()):
            bad += 1; print("MISMATCH phase", phase, "kind", kind, "step", i, flush=True)
        frames += 1
    a.check(); b.check()
    m = a.memory_info(); mem.append(m["overflow_pool_bytes"] >> 20)
print("frames", frames, "mismatches", bad, "pool MB over time", mem, "full-bag tracer pool MB", b.memory_info()["overflow_pool_bytes"] >> 20, "s", round(time.time() - t0, 1))
sys.exit(1 if bad else 0)
