import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python"))
import numpy as np, torch, grt, bench
out = {}
for wl in sys.argv[1:]:
    seed, n, W, H, fisheye, with_mesh, mb, aniso = bench.WORKLOADS[wl]
    acts, center, mesh = bench.build_scene(grt, wl)
    p = grt.default_params(W, H, center, fisheye=fisheye, max_bounces=mb)
    tr = grt.Tracer(0); tr.upload(acts)
    for _ in range(5): tr.render(p); tr.sync()
    tr.set_option(grt.OPT_COUNTERS, 1)
    tr.render(p); c = tr.counters()
    nw = (W // 8) * (H // 8)
    out[wl] = {"particles_fetched_per_tile": c["segments"] / nw, "exact_tests_per_tile": c["proxy_tests"] / nw, "of_which_pieces": c["rays"] / nw,
               "piece_tests_some_lane_hits": c["node_visits"] / nw, "piece_tests_some_lane_inserts": c["rounds"] / nw, "whole_proxy_tests_some_lane_inserts": c["stall_exits"] / nw,
               "compositing_steps": c["hit_evals"] / nw}
    tr.close()
print(json.dumps(out, indent=1))
