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
    out[wl] = {"tiles": nw, "tiles_that_spill": c["rays"], "of_which_never_read_back_one_pass": c["stall_exits"],
               "bag_entries_written": c["segments"], "MB_written": c["segments"] * 16 / 1e6, "entries_by_tiles_that_never_read_back": c["proxy_tests"],
               "share_of_writes_never_read": c["proxy_tests"] / max(c["segments"], 1), "passes_per_tile": c["rounds"] / nw, "refill_scans_per_tile": c["node_visits"] / nw}
    tr.close()
print(json.dumps(out, indent=1))
