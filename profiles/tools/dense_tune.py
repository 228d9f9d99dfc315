"""Dense-core camera (C3 scene, eye in the second densest cell): kernel ms against the tile kernel's band / look-ahead."""
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
def t():
    for _ in range(3): tr.render(p); tr.sync()
    ms = []
    for _ in range(4): tr.render(p); ms.append(tr.last_kernel_ms())
    return round(float(np.median(ms)), 3)
print("base", t())
for band, look in ((256, 64), (1024, 64), (64, 256), (64, 1024), (256, 256), (1024, 1024), (4096, 4096), (1024, 256)):
    tr.set_option(grt.OPT_TILE_BAND, band); tr.set_option(grt.OPT_TILE_LOOKAHEAD, look)
    print("band", band, "look", look, t(), flush=True)
tr.set_option(grt.OPT_TILE_BAND, 64); tr.set_option(grt.OPT_TILE_LOOKAHEAD, 64)
for ready in (4, 32, 64):
    tr.set_option(grt.OPT_TILE_READY_MIN, ready); print("ready", ready, t(), flush=True)
tr.set_option(grt.OPT_TILE_READY_MIN, 16)
for res in (8, 40, 56):
    tr.set_option(grt.OPT_TILE_RESERVE, res); print("reserve", res, t(), flush=True)
