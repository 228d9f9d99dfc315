import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python"))
import numpy as np, torch, grt, bench
acts, center, mesh = bench.build_scene(grt, "C3")
W, H = 1920, 1080
h, edges = np.histogramdd(acts["pos"], bins=48, range=[(-1.5, 1.5)] * 3)
order = np.argsort(h.ravel())[::-1]
tr = grt.Tracer(0); tr.upload(acts)
i = np.unravel_index(order[1], h.shape)
eye = tuple(float((edges[k][i[k]] + edges[k][i[k] + 1]) / 2) for k in range(3))
p = grt.default_params(W, H, center, eye=eye)
for _ in range(4): tr.render(p); tr.sync()
tr.set_option(grt.OPT_COUNTERS, 1); tr.render(p); c = tr.counters()
nw = (W // 8) * (H // 8)
print({k: round(v / nw, 1) for k, v in c.items()}, "kernel ms", tr.last_kernel_ms())
