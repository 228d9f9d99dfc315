"""Pair statistics of the tile kernel's exact test (a -DGRT_TILE_DIAG2 build: GRT_LIB=.../libgrt_hip_diag2.so python profiles/tools/diag2.py C3 C2 C5)."""
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
    nw = ((W + 7) // 8) * ((H + 7) // 8)
    tests = max(c["proxy_tests"], 1)
    out[wl] = {"waves": nw, "particles_fetched": c["segments"] / nw, "exact_tests": c["proxy_tests"] / nw,
               "leaf_steps": (c["rec_fetches"] - 2 * c["node_visits"]) / nw,
               "pretest_positive_lanes_per_test": c["rays"] / tests, "hit_lanes_per_test": c["hit_evals"] / tests,
               "tests_with_a_hit": c["node_visits"] / nw,
               "passes_if_pairs_packed_perfectly_per_leaf_step": c["rounds"] / nw,
               "passes_if_two_consecutive_survivors_share": c["stall_exits"] / nw}
    tr.close()
print(json.dumps(out, indent=1))
