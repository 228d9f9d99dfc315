"""C3 scene with the eye inside the densest cluster cores (the 'dense-core camera' of DESIGN 8)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python"))
import numpy as np, torch, grt, bench
acts, center, mesh = bench.build_scene(grt, "C3")
W, H = 1920, 1080
h, edges = np.histogramdd(acts["pos"], bins=48, range=[(-1.5, 1.5)] * 3)
order = np.argsort(h.ravel())[::-1]
tr = grt.Tracer(0)
if len(sys.argv) > 1 and int(sys.argv[1]) >= 0: tr.set_option(grt.OPT_SPLIT, int(sys.argv[1]))
for kv in sys.argv[2:]:  # further arguments: option=value
    k_, v_ = kv.split("="); tr.set_option(int(k_), int(v_))
tr.upload(acts)
out = []
for rank in (0, 1, 2):
    i = np.unravel_index(order[rank], h.shape)
    eye = tuple(float((edges[k][i[k]] + edges[k][i[k] + 1]) / 2) for k in range(3))
    p = grt.default_params(W, H, center, eye=eye)
    for _ in range(4): tr.render(p); tr.sync()
    ms = []
    for _ in range(5): tr.render(p); ms.append(tr.last_kernel_ms())
    tr.set_option(grt.OPT_COUNTERS, 1); tr.render(p); c = tr.counters(); tr.set_option(grt.OPT_COUNTERS, 0)
    out.append({"eye": eye, "kernel_ms": float(np.median(ms)), "tests_per_ray": c["proxy_tests"] / c["rays"], "boxes_per_ray": c["node_visits"] / c["rays"],
                "passes": c["rounds"] / c["rays"], "hit_evals_per_ray": c["hit_evals"] / c["rays"]})
print(json.dumps(out, indent=1))
