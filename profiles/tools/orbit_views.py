"""orbit_views.py [workload] — is a moving camera's frame slower because it MOVES, or because the views on its path cost more?
The kernel of a STANDING camera at the orbit leg's angles (bench.py: 1.5 degrees per frame about the look-at point), each view rendered
until its order has settled, beside the same views rendered one after the other as the orbit leg does."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python"))
import numpy as np, torch, grt, bench
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
seed, n, W, H, fisheye, with_mesh, mb, aniso = bench.WORKLOADS[wl]
acts, center, mesh = bench.build_scene(grt, wl)
tr = grt.Tracer(0); tr.upload(acts)
eye0 = np.float32([0, 0, 3]) - center
def view(i):
    ang = np.deg2rad(1.5 * i)
    eye = center + np.float32([eye0[0] * np.cos(ang) + eye0[2] * np.sin(ang), eye0[1], -eye0[0] * np.sin(ang) + eye0[2] * np.cos(ang)])
    return grt.default_params(W, H, center, fisheye=fisheye, max_bounces=mb, eye=tuple(float(x) for x in eye))
standing = []
for i in (0, 3, 8, 13, 18, 23):
    p = view(i)
    for _ in range(6): tr.render(p); tr.sync()
    ms = []
    for _ in range(8): tr.render(p); ms.append(tr.last_kernel_ms())
    standing.append((i, round(float(np.median(ms)), 4)))
moving = []
for i in range(24):
    tr.render(view(i)); moving.append(round(tr.last_kernel_ms(), 4))
print(json.dumps({"workload": wl, "standing_camera_kernel_ms_at_frame_angle": standing, "moving_camera_kernel_ms_per_frame": moving}))
