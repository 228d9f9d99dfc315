"""pairable exact-test trips by 8x4 halves (GRT_LIB = libgrt_hip_diag_r6.so)."""
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
    tr.set_option(grt.OPT_QUAD_PARTS, 0)
    for _ in range(6): tr.render(p); tr.sync()
    tr.set_option(grt.OPT_COUNTERS, 1)
    tr.render(p); c = tr.counters()
    nw = (W // 8) * (H // 8)
    leaf_steps = (c["rec_fetches"] - 2 * c["node_visits"] * 0) / nw  # (see below: fetches is folded into rec_fetches; not used)
    t = c["proxy_tests"] / nw
    out[wl] = {"tiles": nw, "particles_fetched_per_tile": c["segments"] / nw, "tests_with_a_lane_per_tile": t,
               "one_vertical_half_only": c["rounds"] / nw, "pairs_top_bottom": c["node_visits"] / nw, "pairs_top_bottom_share_of_tests": c["node_visits"] / max(c["proxy_tests"], 1),
               "one_horizontal_half_only": c["rays"] / nw, "pairs_left_right": c["stall_exits"] / nw, "pairs_left_right_share_of_tests": c["stall_exits"] / max(c["proxy_tests"], 1),
               "compositing_steps": c["hit_evals"] / nw}
    tr.close()
print(json.dumps(out, indent=1))
