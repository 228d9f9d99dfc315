"""Steps of the tile kernel by the number of lanes that still want something (a -DGRT_TILE_DIAG -DGRT_TILE_DIAG4 -DGRT_TILE_DIAG5 build)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python"))
import numpy as np, torch, grt, bench
out = {}
for wl in sys.argv[1:]:
    seed, n, W, H, fisheye, with_mesh, mb, aniso = bench.WORKLOADS[wl]
    acts, center, mesh = bench.build_scene(grt, wl)
    p = grt.default_params(W, H, center, fisheye=fisheye, max_bounces=mb)
    tr = grt.Tracer(0); tr.upload(acts)
    for _ in range(6): tr.render(p); tr.sync()
    tr.set_option(grt.OPT_COUNTERS, 1)
    tr.render(p); c = tr.counters()
    nw = (W // 8) * (H // 8)
    allsteps = max(c["rec_fetches"] - 2 * c["node_visits"], 1)  # (counter 6 = fetches + 2 x node_visits)
    out[wl] = {"steps_per_tile": allsteps / nw, "share_with_at_most_2_wanting_lanes": c["rays"] / allsteps, "at_most_4": c["segments"] / allsteps,
               "at_most_8": c["hit_evals"] / allsteps, "at_most_16": c["proxy_tests"] / allsteps}
    tr.close()
print(json.dumps(out, indent=1))
