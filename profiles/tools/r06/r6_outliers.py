import sys, os, numpy as np
sys.path[:0] = ["gaussian-ray-tracing_amd/python", "oracle", "tests"]
import grt, oracle as O
from common import make_scene, usable_cores, threshold_flip_explains
W, H = 1920, 1080
acts, p, sc, op, _ = make_scene(3, 1_000_000, W, H)
tr = grt.Tracer(0); tr.upload(acts)
u8, f32 = tr.render(p, want_f32=True); tr.sync()
g = f32.cpu().numpy()
ru8, rf, rc = sc.render(op, threads=usable_cores())
d = np.abs(g - rf)
ys, xs = np.nonzero((d > 1e-5).any(-1))
print("pixels over 1e-5:", len(ys), "over 1e-4:", int((d > 1e-4).any(-1).sum()), "max", d.max())
for y, x in zip(ys, xs):
    print(x, y, g[y, x], rf[y, x], d[y, x].max(), threshold_flip_explains(sc, op, int(x), int(y), g[y, x]))
hist = np.histogram(d.max(-1).ravel(), bins=[0, 1e-8, 1e-7, 3e-7, 1e-6, 1e-5, 1e-4, 1])
print(hist)
