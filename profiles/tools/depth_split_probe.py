"""What would splitting a heavy tile in DEPTH buy?  (profiles/r04_experiments_log.md, item 3)

The frame's critical path is its heaviest 8x8 tile; pixel sub-blocks shorten it by a fifth only.  This probe uses the library as
it is: it finds the slowest 8x8 tile of the workload (kernel ms of single-tile windows over a coarse grid, then refined), then
renders that tile with the ray interval cut at tau — [t_min, tau) and [tau, t_max) as two frames — for a sweep of tau, and prints
the two kernel times beside the whole tile's.  The far part starts with full transmittance (it does not know what the near
part absorbed), which is exactly what a depth-slab wave would have to do.
  python profiles/tools/depth_split_probe.py C3 [C2 C1]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python"))
import numpy as np, torch, grt, bench

def ms_of(tr, p, win, reps=5):
    out = torch.zeros((p.height, p.width, 3), dtype=torch.uint8, device="cuda:0")
    v = []
    for _ in range(reps):
        tr.render(p, window=win, out_u8=out); tr.sync(); v.append(tr.last_kernel_ms())
    return float(np.median(v[1:]))

res = {}
for wl in sys.argv[1:]:
    seed, n, W, H, fisheye, with_mesh, mb, aniso = bench.WORKLOADS[wl]
    acts, center, mesh = bench.build_scene(grt, wl)
    p = grt.default_params(W, H, center, fisheye=fisheye, max_bounces=mb)
    tr = grt.Tracer(0); tr.upload(acts)
    tr.set_option(grt.OPT_TILE_PARTS2_PCT, 0); tr.set_option(grt.OPT_TILE_PARTS4_PCT, 0)
    full = ms_of(tr, p, (0, 0, W, H))
    # coarse: 64x64 windows; then the 8x8 tiles of the slowest three
    coarse = []
    for y in range(0, H, 64):
        for x in range(0, W, 64):
            coarse.append((ms_of(tr, p, (x, y, min(x + 64, W), min(y + 64, H)), 3), x, y))
    coarse.sort(reverse=True)
    fine = []
    for _, x0, y0 in coarse[:3]:
        for y in range(y0, min(y0 + 64, H), 8):
            for x in range(x0, min(x0 + 64, W), 8):
                fine.append((ms_of(tr, p, (x, y, min(x + 8, W), min(y + 8, H)), 3), x, y))
    fine.sort(reverse=True)
    t_tile, tx, ty = fine[0]
    win = (tx, ty, tx + 8, ty + 8)
    sweep = []
    tmin, tmax = p.t_min, p.t_max
    for tau in (1.5, 2.0, 2.25, 2.5, 2.6, 2.7, 2.8, 2.9, 3.0, 3.1, 3.2, 3.3, 3.4, 3.5, 3.75, 4.0, 4.5):
        p.t_min, p.t_max = tmin, tau
        near = ms_of(tr, p, win)
        p.t_min, p.t_max = tau, tmax
        far = ms_of(tr, p, win)
        p.t_min, p.t_max = tmin, tmax
        sweep.append({"tau": tau, "near_ms": round(near, 4), "far_ms": round(far, 4)})
    res[wl] = {"frame_kernel_ms": round(full, 4), "slowest_64x64_windows_ms": [round(c[0], 4) for c in coarse[:5]],
               "slowest_tile": {"x": tx, "y": ty, "kernel_ms": round(t_tile, 4)}, "next_tiles_ms": [round(f[0], 4) for f in fine[1:6]],
               "depth_split_of_the_slowest_tile": sweep}
    tr.close()
print(json.dumps(res, indent=1))
