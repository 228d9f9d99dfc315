"""Wave-level trip counts of the tile kernel (diagnostic build: GRT_LIB=.../libgrt_hip_diag.so)."""
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
    out[wl] = {"waves": nw, "node_steps": c["rays"] / nw, "particles_fetched": c["segments"] / nw, "compositing_steps": c["hit_evals"] / nw,
               "passes": c["rounds"] / nw, "dfs_pops_and_refills": c["node_visits"] / nw, "exact_tests": c["proxy_tests"] / nw,
               "rec_fetches_raw": c["rec_fetches"] / nw, "rebalances": c["stall_exits"] / nw, "kernel_ms": tr.last_kernel_ms()}
    tr.close()
print(json.dumps(out, indent=1))
