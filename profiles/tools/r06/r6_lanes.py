import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python"))
import numpy as np, torch, grt, bench
which = os.environ.get("GRT_LIB", "")
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
    if "r6d" in which:
        out[wl] = {"tiles": nw, "particles_fetched": c["segments"] / nw, "exact_tests": c["proxy_tests"] / nw, "compositing_steps": c["hit_evals"] / nw,
                   "lanes_through_the_sphere_pre_test_per_exact_test": c["rays"] / max(c["proxy_tests"], 1), "lanes_that_hit_per_exact_test": c["stall_exits"] / max(c["proxy_tests"], 1),
                   "inserting_lanes_total_per_tile": c["rounds"] / nw, "lanes_per_compositing_step": c["node_visits"] / max(c["hit_evals"], 1)}
    else:
        out[wl] = {"tiles": nw, "node_steps": c["rays"] / nw, "particles_fetched": c["segments"] / nw, "compositing_steps": c["hit_evals"] / nw, "passes": c["rounds"] / nw,
                   "dfs_pops_and_refills": c["node_visits"] / nw, "exact_tests": c["proxy_tests"] / nw, "rebalances": c["stall_exits"] / nw}
    tr.close()
print(json.dumps(out, indent=1))
