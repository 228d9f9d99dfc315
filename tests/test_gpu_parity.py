"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Tolerances (north star): radiance max |diff| <= 1e-4 per channel; 8-bit output EQUAL to the oracle's except
within that tolerance of a quantisation step (see compare())."""
import json
import os

import numpy as np
import pytest

import grt
import oracle as O
from common import acts_to_particles, make_scene, to_oracle_params, u8_matches

pytestmark = pytest.mark.gpu
TOL = 1e-4
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_probes.json")))


@pytest.fixture(scope="module", params=[0, 1, 2, 3], ids=["tile", "perlane", "wave", "stream"])
def tr(request):
    """Every parity test runs on each traversal path (GRT_OPT_KERNEL): 0 = default (tile kernel: BVH culling per
    child box against the tile frustum; stage 2 of the wavefront pipeline when meshes are present), 1 = per-lane
    megakernel everywhere, 2 = round-based wave kernel (per-lane megakernel when meshes are present), 3 = single-pass
    streaming wave kernel (round 1's default)."""
    t = grt.Tracer(0)
    t.set_option(grt.OPT_KERNEL, request.param)
    t.kernel_variant = request.param
    yield t
    t.close()


def compare(gpu_f32, ref_f32, gpu_u8=None, ref_u8=None, tol=TOL, max_outlier_frac=0.0, max_outlier=0.0):
    """Radiance within `tol` on every pixel (or on all but `max_outlier_frac` of them, each of which is still within
    `max_outlier`).  The 8-bit frame must EQUAL the oracle's, except on pixels whose radiance lies within `tol` of a
    quantisation step (x * 256 within tol * 256 of an integer, where a difference below the radiance tolerance may
    legitimately change the level by one) and on the radiance outliers."""
    g = gpu_f32.cpu().numpy() if hasattr(gpu_f32, "cpu") else gpu_f32
    d = np.abs(g - ref_f32)
    bad = (d > tol).any(-1)
    frac = bad.mean()
    assert frac <= max_outlier_frac, f"max diff {d.max():.3e}, {bad.sum()} px over {tol}"
    if bad.any():
        assert d[bad].max() <= max_outlier, f"outlier of {d[bad].max():.3e} (bound {max_outlier})"
    if gpu_u8 is not None:
        g8 = gpu_u8.cpu().numpy() if hasattr(gpu_u8, "cpu") else gpu_u8
        ok = u8_matches(g8, ref_u8, ref_f32, tol) | bad[..., None]
        assert ok.all(), f"{(~ok).sum()} 8-bit values differ away from a quantisation step"
    return d.max()


def test_kat3_golden_pixel(tr):
    import torch
    k = G["kat3"]["raw"]
    raw = dict(pos=np.float32([r["pos"] for r in k]), f_dc=np.float32([r["f_dc"] for r in k]),
               f_rest=np.zeros((3, 45), np.float32), opacity=np.float32([r["logit"] for r in k]),
               scale=np.float32([r["log_scale"] for r in k]), rot=np.float32([r["rot"] for r in k]))
    tr.upload(grt.activate(raw))
    o = np.float32(G["kat3"]["ray_o"]); d = np.float32(G["kat3"]["ray_d_unnormalised"])
    d = (d / np.float32(np.sqrt(np.float32((d * d).sum())))).astype(np.float32)
    p = grt.default_params(16, 16, (0, 0, 0))
    rays = torch.tensor(np.concatenate([o, d])[None], device="cuda:0")
    tr.set_option(grt.OPT_COUNTERS, 1)
    out = tr.render_rays(p, rays).cpu().numpy()[0]
    cnt = tr.counters()
    tr.set_option(grt.OPT_COUNTERS, 0)
    np.testing.assert_allclose(out, G["kat3"]["pixel"], atol=1e-5)
    assert [int(min(int(np.float32(min(max(x, 0), 1)) * np.float32(256)), 255)) for x in out] == G["kat3"]["u8"]
    assert cnt["hit_evals"] == 6 and cnt["rays"] == 1 and cnt["rounds"] == 1


def test_c1_10k_256x256_pinhole(tr):
    """BASELINE config C1: 10k-Gaussian synthetic scene (seed 1), 256x256 pinhole, single bounce."""
    acts, p, sc, op, _ = make_scene(1, 10000, 256, 256)
    tr.upload(acts)
    tr.set_option(grt.OPT_COUNTERS, 1)
    u8, f32 = tr.render(p, want_f32=True)
    cnt = tr.counters()
    tr.set_option(grt.OPT_COUNTERS, 0)
    ref_u8, ref_f32, rc = sc.render(op)
    compare(f32, ref_f32, u8, ref_u8)
    assert cnt["rays"] == rc["rays"] == 256 * 256 and cnt["stall_exits"] == 0
    assert abs(cnt["hit_evals"] - rc["hit_evals"]) <= 1e-4 * rc["hit_evals"]
    if tr.kernel_variant not in (0, 3):  # the streaming / tile kernels replace the k = 7 rounds by (mostly) one pass
        assert cnt["rounds"] == rc["rounds"] or abs(cnt["rounds"] - rc["rounds"]) <= 1e-4 * rc["rounds"]
    else:
        assert rc["rays"] <= cnt["rounds"] < rc["rounds"]
    # the un-instrumented kernel writes the same bytes
    u8b, f32b = tr.render(p, want_f32=True)
    assert (u8b == u8).all() and (f32b == f32).all()
    info = tr.bvh_info()
    assert info["n_particles"] == 10000 and 0 < info["n_proxies"] <= 10000 and info["height"] >= 14


@pytest.mark.parametrize("deg", [1, 2, 3])
def test_sh_degrees(tr, deg):
    acts, p, sc, op, _ = make_scene(4, 3000, 96, 96, scale_boost=0.5, sh_degree=deg)
    tr.upload(acts)
    u8, f32 = tr.render(p, want_f32=True)
    ref_u8, ref_f32, _ = sc.render(op)
    compare(f32, ref_f32, u8, ref_u8)


def test_dense_large_proxies_many_rounds(tr):
    """Large, overlapping proxies: dozens of k-buffer rounds per ray."""
    acts, p, sc, op, _ = make_scene(6, 1500, 64, 64, scale_boost=1.5)
    tr.upload(acts)
    tr.set_option(grt.OPT_COUNTERS, 1)
    u8, f32 = tr.render(p, want_f32=True)
    cnt = tr.counters()
    tr.set_option(grt.OPT_COUNTERS, 0)
    ref_u8, ref_f32, rc = sc.render(op)
    compare(f32, ref_f32, u8, ref_u8)
    assert rc["rounds"] > 3 * rc["rays"]
    assert cnt["stall_exits"] == 0 and abs(cnt["hit_evals"] - rc["hit_evals"]) <= 1e-4 * rc["hit_evals"]


def test_fisheye(tr):
    """Fisheye raygen uses sinf/cosf/asinf/atan2f whose device/host ulps differ: ray directions differ in the
    last bits, so allow a 2e-4 fraction of near-tie order flips (SURVEY §7 hard part 1)."""
    acts, p, sc, op, _ = make_scene(7, 4000, 128, 96, scale_boost=0.5, fisheye=True)
    tr.upload(acts)
    tr.set_option(grt.OPT_COUNTERS, 1)
    u8, f32 = tr.render(p, want_f32=True)
    cnt = tr.counters()
    tr.set_option(grt.OPT_COUNTERS, 0)
    ref_u8, ref_f32, rc = sc.render(op)
    compare(f32, ref_f32, u8, ref_u8, max_outlier_frac=2e-4, max_outlier=0.08)
    assert cnt["rays"] == rc["rays"] < 128 * 96
    g = f32.cpu().numpy()
    assert (g[0, 0] == 0).all() and (u8.cpu().numpy()[0, 0] == 0).all()  # r > 1 => black (decision vii)


def test_fisheye_exact_with_host_rays(tr):
    """Ray-buffer mode: with the oracle's own fisheye rays the integration is bit-reproducible."""
    import ctypes as C
    import torch
    acts, p, sc, op, _ = make_scene(7, 4000, 64, 48, scale_boost=0.5, fisheye=True)
    tr.upload(acts)
    L = O.lib(); fp = C.POINTER(C.c_float)
    L.grto_get_fisheye_ray.argtypes = [C.c_uint32, C.c_uint32, fp, fp, fp, fp, C.c_uint32, C.c_uint32, fp, fp]
    nU = np.float32([-x for x in p.U]); nV = np.float32([-x for x in p.V]); W = np.float32(list(p.W)); eye = np.float32(list(p.eye))
    rays = []
    for y in range(48):
        for x in range(64):
            o = np.zeros(3, np.float32); d = np.zeros(3, np.float32)
            if L.grto_get_fisheye_ray(x, y, nU.ctypes.data_as(fp), nV.ctypes.data_as(fp), W.ctypes.data_as(fp),
                                      eye.ctypes.data_as(fp), 64, 48, o.ctypes.data_as(fp), d.ctypes.data_as(fp)):
                rays.append(np.concatenate([o, d]))
    rays = np.float32(rays)
    ref, _ = sc.render_rays(op, rays)
    out = tr.render_rays(p, torch.tensor(rays, device="cuda:0"))
    compare(out, ref)


def test_axis_parallel_rays_with_signed_zero_directions(tr):
    """The slab test folds the sign of a direction projection in by its SIGN BIT (grt_device.h: proxy_slabs_pre,
    oracle/grt_oracle.c: proxy_slabs): projections that are exactly +0 or -0 — rays along the axes of unrotated Gaussians —
    are where the two readings of 'negative' differ.  Oracle and kernel must agree on them, whatever the sign of the zero."""
    import torch
    acts, p, sc, op, _ = make_scene(23, 600, 32, 32, scale_boost=0.8)
    acts = dict(acts)
    acts["quat"] = np.tile(np.float32([1, 0, 0, 0]), (len(acts["pos"]), 1))  # unrotated: the slab normals are the axes' own
    acts["pos"] = (np.round(acts["pos"] * 4.0) / 4.0).astype(np.float32)     # centres on a lattice the rays run through
    sc.close()
    sc = O.Scene(acts_to_particles(acts))
    tr.upload(acts)
    rays = []
    zeros = (np.float32(0.0), np.float32(-0.0))
    for axis in range(3):
        for sgn in (1.0, -1.0):
            for za in zeros:
                for zb in zeros:
                    for u in np.arange(-1.0, 1.01, 0.25):
                        for v in np.arange(-1.0, 1.01, 0.25):
                            o = np.zeros(3, np.float32); d = np.zeros(3, np.float32)
                            o[axis] = -6.0 * sgn; o[(axis + 1) % 3] = u; o[(axis + 2) % 3] = v
                            d[axis] = sgn; d[(axis + 1) % 3] = za; d[(axis + 2) % 3] = zb
                            rays.append(np.concatenate([o, d]))
    rays = np.float32(rays)
    ref, cnt = sc.render_rays(op, rays)
    out = tr.render_rays(p, torch.tensor(rays, device="cuda:0"))
    assert cnt["hit_evals"] > len(rays)  # the rays do run through Gaussians
    compare(out, ref)
    sc.close()


@pytest.mark.parametrize("mesh_type", [grt.MIRROR, grt.NORMAL, grt.GLASS])
def test_mesh_sphere_modes(tr, mesh_type):
    acts, p, sc, op, center = make_scene(8, 5000, 128, 128, scale_boost=0.5, mesh_type=mesh_type)
    pos = (0.25 * center + 0.75 * np.float32([0, 0, 3])).astype(np.float32)
    v, n, f = grt.sphere_mesh(pos, tess_u=48, tess_v=24)
    tr.upload(acts)
    tr.set_meshes([(v, n, f)])
    sc.set_mesh(v, n, f)
    tr.set_option(grt.OPT_COUNTERS, 1)
    u8, f32 = tr.render(p, want_f32=True)
    cnt = tr.counters()
    tr.set_option(grt.OPT_COUNTERS, 0)
    tr.set_meshes([])
    ref_u8, ref_f32, rc = sc.render(op)
    compare(f32, ref_f32, u8, ref_u8)
    assert cnt["segments"] == rc["segments"]
    if mesh_type != grt.NORMAL:
        assert rc["segments"] > rc["rays"]


@pytest.mark.parametrize("mesh_type", [grt.MIRROR, grt.GLASS])
def test_mesh_with_zero_normals_nan_bounce_directions(tr, mesh_type):
    """A mesh patch whose vertex normals are all zero interpolates to a NaN shading normal (normalize(0),
    shaders/tracer.cuh:167-185), so the bounced direction is NaN and the reference's loop guard `length(dir) > 0.1`
    (shaders/tracer.cu:59) ends the ray.  Such a ray must end HERE too without taking the other rays of its tile / bundle
    with it (the bundle frustum's wave reductions do not ignore NaN; the guard is what keeps NaN out of them):
    every pixel, the NaN ones' neighbours included, must match the oracle, and all pixels must be finite."""
    acts, p, sc, op, center = make_scene(8, 5000, 128, 128, scale_boost=0.5, mesh_type=mesh_type, max_bounces=4)
    pos = (0.25 * center + 0.75 * np.float32([0, 0, 3])).astype(np.float32)
    v, n, f = grt.sphere_mesh(pos, tess_u=48, tess_v=24)
    n = n.copy()
    front = (v[:, 2] - pos[2] > 0.2) & (np.abs(v[:, 0] - pos[0]) < 0.12)  # a band of the side that faces the camera
    assert 20 < front.sum() < len(v) // 2
    n[front] = 0.0
    tr.upload(acts)
    tr.set_meshes([(v, n, f)])
    sc.set_mesh(v, n, f)
    tr.set_option(grt.OPT_COUNTERS, 1)
    u8, f32 = tr.render(p, want_f32=True)
    cnt = tr.counters()
    tr.set_option(grt.OPT_COUNTERS, 0)
    tr.set_meshes([])
    ref_u8, ref_f32, rc = sc.render(op)
    assert np.isfinite(ref_f32).all() and bool(np.isfinite(f32.cpu().numpy()).all())
    compare(f32, ref_f32, u8, ref_u8)
    assert cnt["segments"] == rc["segments"] and cnt["stall_exits"] == 0


def test_two_meshes_plane_and_sphere_mirror_bounce_cap(tr):
    acts, p, sc, op, center = make_scene(9, 3000, 96, 96, scale_boost=0.5, mesh_type=grt.MIRROR, max_bounces=2)
    eye = np.float32([0, 0, 3])
    v1, n1, f1 = grt.sphere_mesh((0.25 * center + 0.75 * eye).astype(np.float32) + np.float32([0.35, 0, 0]), tess_u=32, tess_v=16)
    v2, n2, f2 = grt.plane_mesh((0.25 * center + 0.75 * eye).astype(np.float32) - np.float32([0.3, 0, 0]))
    tr.upload(acts)
    tr.set_meshes([(v1, n1, f1), (v2, n2, f2)])
    sc.set_mesh(np.concatenate([v1, v2]), np.concatenate([n1, n2]), np.concatenate([f1, f2 + len(v1)]))
    u8, f32 = tr.render(p, want_f32=True)
    tr.set_meshes([])
    ref_u8, ref_f32, _ = sc.render(op)
    compare(f32, ref_f32, u8, ref_u8)


def test_window_and_tiles_are_bit_identical_to_full_frame(tr):
    import torch
    acts, p, sc, op, _ = make_scene(10, 4000, 200, 120, scale_boost=0.5)
    tr.upload(acts)
    u8, f32 = tr.render(p, want_f32=True)
    # window
    w8 = torch.zeros_like(u8); wf = torch.zeros_like(f32)
    tr.render(p, window=(37, 11, 150, 97), out_u8=w8, out_f32=wf)
    assert (w8[11:97, 37:150] == u8[11:97, 37:150]).all() and (wf[11:97, 37:150] == f32[11:97, 37:150]).all()
    assert (w8[:11] == 0).all() and (w8[:, :37] == 0).all() and (w8[97:] == 0).all() and (w8[:, 150:] == 0).all()
    # interleaved tiles, 2 "ranks", ragged border (200 = 6*32 + 8, 120 = 3*32 + 24)
    tw = th = 32
    tx, ty = (200 + tw - 1) // tw, (120 + th - 1) // th
    full8 = torch.zeros((ty * th, tx * tw, 3), dtype=torch.uint8, device="cuda:0")
    fullf = torch.zeros((ty * th, tx * tw, 3), dtype=torch.float32, device="cuda:0")
    for rank in range(2):
        cnt = (tx * ty - rank + 1) // 2
        b8 = torch.full((cnt, th, tw, 3), 7, dtype=torch.uint8, device="cuda:0")
        bf = torch.full((cnt, th, tw, 3), 7.0, dtype=torch.float32, device="cuda:0")
        tr.render_tiles(p, tw, th, rank, 2, cnt, out_u8=b8, out_f32=bf)
        for j in range(cnt):
            t = rank + 2 * j
            full8[(t // tx) * th:(t // tx + 1) * th, (t % tx) * tw:(t % tx + 1) * tw] = b8[j]
            fullf[(t // tx) * th:(t // tx + 1) * th, (t % tx) * tw:(t % tx + 1) * tw] = bf[j]
    assert (full8[:120, :200] == u8).all() and (fullf[:120, :200] == f32).all()
    assert (full8[120:] == 0).all() and (full8[:, 200:] == 0).all()  # outside the frame: zeros


def test_edge_cases_empty_transparent_single(tr):
    import torch
    p = grt.default_params(32, 32, (0, 0, 0))
    # empty scene
    tr.upload(dict(pos=np.zeros((0, 3), np.float32), scale=np.zeros((0, 3), np.float32), quat=np.zeros((0, 4), np.float32),
                   opacity=np.zeros(0, np.float32), sh=np.zeros((0, 16, 3), np.float32)))
    u8, f32 = tr.render(p, want_f32=True)
    assert (u8 == 0).all() and (f32 == 0).all()
    # every particle below alpha_min: nothing hittable (decision vi)
    raw = grt.synth_scene(3, 50); raw["opacity"][:] = -8.0
    tr.upload(grt.activate(raw))
    assert tr.bvh_info()["n_proxies"] == 0
    u8, f32 = tr.render(p, want_f32=True)
    assert (u8 == 0).all()
    # one and two particles (degenerate BVHs)
    for n in (1, 2, 3):
        raw = grt.synth_scene(4, n); raw["pos"][:] *= 0.2; raw["scale"][:] = -1.5; raw["opacity"][:] = 2.0
        acts = grt.activate(raw)
        tr.upload(acts)
        sc = O.Scene(acts_to_particles(acts))
        u8, f32 = tr.render(p, want_f32=True)
        ref_u8, ref_f32, rc = sc.render(to_oracle_params(p))
        compare(f32, ref_f32, u8, ref_u8)
        assert rc["hit_evals"] > 0


def test_invalid_arguments_raise(tr):
    p = grt.default_params(32, 32, (0, 0, 0))
    raw = grt.synth_scene(4, 10)
    tr.upload(grt.activate(raw))
    with pytest.raises(grt.GrtError, match="window"):
        tr.render(p, window=(0, 0, 33, 32))
    p.sh_degree_max = 4
    with pytest.raises(grt.GrtError, match="sh_degree"):
        tr.render(p)
    p.sh_degree_max = 0
    with pytest.raises(grt.GrtError, match="multiple of 16"):
        import torch
        tr.render_tiles(p, 24, 24, 0, 1, 1, out_u8=torch.zeros((1, 24, 24, 3), dtype=torch.uint8, device="cuda:0"))


def test_c2_100k_window(tr):
    """BASELINE config C2 scene (100k, seed 2, 1280x720): centre 320x192 window against the oracle."""
    acts, p, sc, op, _ = make_scene(2, 100000, 1280, 720)
    tr.upload(acts)
    win = (480, 264, 800, 456)
    u8, f32 = tr.render(p, window=win, want_f32=True)
    ref_u8, ref_f32, rc = sc.render(op, window=win)
    x0, y0, x1, y1 = win
    compare(f32[y0:y1, x0:x1], ref_f32[y0:y1, x0:x1], u8[y0:y1, x0:x1], ref_u8[y0:y1, x0:x1])
    assert rc["hit_evals"] > 5 * rc["rays"]


@pytest.mark.parametrize("eye,fovy", [((0.05, -0.1, 0.2), 75.0), ((1.2, 0.9, -1.4), 40.0), ((0.0, 2.5, 0.3), 100.0)])
def test_camera_inside_and_around_the_cloud(tr, eye, fovy):
    """Ray origins inside proxies (exit-only hits), inside BVH boxes (negative box entry distances) and oblique
    views: the default camera of the other tests always looks down -z from outside the scene."""
    raw = grt.synth_scene(13, 6000); raw["scale"] = raw["scale"] + np.float32(0.6)
    acts = grt.activate(raw)
    center = grt.gaussian_center(acts["pos"])
    p = grt.default_params(112, 80, center, eye=eye, fovy=fovy)
    tr.upload(acts)
    u8, f32 = tr.render(p, want_f32=True)
    sc = O.Scene(acts_to_particles(acts))
    ref_u8, ref_f32, rc = sc.render(to_oracle_params(p))
    compare(f32, ref_f32, u8, ref_u8)
    assert rc["hit_evals"] > 4 * rc["rays"]


def test_needle_and_pancake_gaussians(tr):
    """Trained-scene-like anisotropy: scales spread over 3 decades per axis (needles, pancakes), large and tiny
    proxies mixed — stresses the LBVH (huge overlapping boxes) and the slot window."""
    rng = np.random.default_rng(5)
    raw = grt.synth_scene(14, 5000)
    raw["scale"] = (raw["scale"] + rng.normal(0.0, 1.6, size=raw["scale"].shape)).astype(np.float32)
    acts = grt.activate(raw)
    p = grt.default_params(96, 96, grt.gaussian_center(acts["pos"]))
    tr.upload(acts)
    u8, f32 = tr.render(p, want_f32=True)
    sc = O.Scene(acts_to_particles(acts))
    ref_u8, ref_f32, rc = sc.render(to_oracle_params(p))
    compare(f32, ref_f32, u8, ref_u8)


def test_feedback_scheduling_does_not_change_pixels(tr):
    """Heaviest-first block order (frame-to-frame feedback) is pure scheduling."""
    acts, p, sc, op, _ = make_scene(15, 20000, 320, 200, scale_boost=0.3)
    tr.upload(acts)
    tr.set_option(grt.OPT_FEEDBACK, 0)
    a8, af = tr.render(p, want_f32=True)
    tr.set_option(grt.OPT_FEEDBACK, 1)
    frames = [tr.render(p, want_f32=True) for _ in range(3)]  # cold order, then two fed-back frames
    for b8, bf in frames:
        assert (a8 == b8).all() and (af == bf).all()
    ref_u8, ref_f32, _ = sc.render(op)
    compare(af, ref_f32, a8, ref_u8)


def test_coincident_gaussians_deep_overlap(tr):
    """Collisions: 600 Gaussians share one centre exactly (identical Morton codes, a tall LBVH) with different sizes
    and low opacity, inside a random cloud — several hundred proxies overlap along the central rays, far more than any
    window holds, so the streaming kernel must make progress pass after pass and still composite in exact key order."""
    raw = grt.synth_scene(31, 1500)
    raw["pos"][:600] = np.float32([0.05, -0.02, 0.1])
    raw["scale"][:600] = np.log(np.float32(0.02) + np.float32(0.0006) * np.arange(600, dtype=np.float32))[:, None]
    raw["opacity"][:600] = np.float32(-1.5)  # sigmoid -> 0.18: the rays go deep before they saturate
    acts = grt.activate(raw)
    center = grt.gaussian_center(acts["pos"])
    p = grt.default_params(72, 64, center)
    sc = O.Scene(acts_to_particles(acts))
    tr.upload(acts)
    tr.set_option(grt.OPT_COUNTERS, 1)
    u8, f32 = tr.render(p, want_f32=True)
    cnt = tr.counters()
    tr.set_option(grt.OPT_COUNTERS, 0)
    ref_u8, ref_f32, rc = sc.render(to_oracle_params(p))
    compare(f32, ref_f32, u8, ref_u8)
    assert abs(cnt["hit_evals"] - rc["hit_evals"]) <= 1e-4 * rc["hit_evals"]
    assert rc["hit_evals"] > 20 * rc["rays"]
    assert cnt["stall_exits"] == 0  # no lane was given up on with transmittance left
    sc.close()


@pytest.mark.parametrize("kernel", [0, 3, 4])
@pytest.mark.parametrize("n", [1, 2, 4])
def test_tiny_scene_in_a_fresh_context(kernel, n):
    """<= leaf_max hittable proxies: the LBVH root is a leaf range and there is NO node array.  A fresh context (no
    stale buffers of an earlier, larger scene) must render it through grt_render on the streaming kernels."""
    t = grt.Tracer(0)
    try:
        t.set_option(grt.OPT_KERNEL, kernel)
        raw = grt.synth_scene(4, n); raw["pos"][:] *= 0.2; raw["scale"][:] = -1.5; raw["opacity"][:] = 2.0
        acts = grt.activate(raw)
        t.upload(acts)
        assert t.bvh_info()["n_nodes"] == n - 1
        p = grt.default_params(48, 40, (0, 0, 0))
        u8, f32 = t.render(p, want_f32=True)
        sc = O.Scene(acts_to_particles(acts))
        ref_u8, ref_f32, rc = sc.render(to_oracle_params(p))
        compare(f32, ref_f32, u8, ref_u8)
        assert rc["hit_evals"] > 0
        # all but <= 4 particles below alpha_min, after a larger scene in the same context
        raw = grt.synth_scene(5, 3000); t.upload(grt.activate(raw))
        t.render(p)
        raw = grt.synth_scene(5, 300); raw["opacity"][3:] = -9.0; raw["opacity"][:3] = 3.0; raw["scale"][:3] = -1.2
        raw["pos"][:3] *= 0.1
        acts = grt.activate(raw)
        t.upload(acts)
        assert t.bvh_info()["n_proxies"] == 3
        u8, f32 = t.render(p, want_f32=True)
        sc = O.Scene(acts_to_particles(acts))
        ref_u8, ref_f32, rc = sc.render(to_oracle_params(p))
        compare(f32, ref_f32, u8, ref_u8)
    finally:
        t.close()


def test_option_values_are_validated():
    t = grt.Tracer(0)
    try:
        for bad in (-1, 99):
            with pytest.raises(grt.GrtError, match="GRT_OPT_KERNEL"):
                t.set_option(grt.OPT_KERNEL, bad)
        raw = grt.synth_scene(4, 10)
        with pytest.raises(grt.GrtError, match="alpha_min"):
            t.upload(grt.activate(raw), alpha_min=0.0)
    finally:
        t.close()


def test_moved_mesh_refit_matches_rebuild_and_oracle(tr):
    """grt_update_meshes (a gizmo drag: reference updateInstanceTransforms, src/GaussianTracer.cpp:711-736): the mesh LBVH
    keeps its hierarchy and re-fits its boxes.  A moved / rotated sphere must render exactly as a freshly built one,
    and as the oracle's."""
    acts, p, sc, op, center = make_scene(41, 5000, 160, 120, scale_boost=0.5, mesh_type=grt.MIRROR, max_bounces=3)
    pos = (0.25 * center + 0.75 * np.float32([0, 0, 3])).astype(np.float32)
    v0, n0, f = grt.sphere_mesh((0, 0, 0), tess_u=48, tess_v=24)
    tr.upload(acts)
    tr.set_meshes([(v0 + pos, n0, f)])
    a8, af = tr.render(p, want_f32=True)
    a8, af = a8.clone(), af.clone()
    build_ms = tr.bvh_info()["mesh_update_ms"]
    # drag: rotate about z by 40 degrees, scale 1.3 in x, translate (positions by M, normals by mat3(M): GaussianTracer.cpp:659-662)
    ang = np.float32(np.deg2rad(40.0)); c_, s_ = np.cos(ang), np.sin(ang)
    M = (np.float32([[c_, -s_, 0], [s_, c_, 0], [0, 0, 1]]) @ np.diag(np.float32([1.3, 1.0, 1.0]))).astype(np.float32)
    v1 = (v0 @ M.T + pos + np.float32([0.25, -0.1, -0.2])).astype(np.float32)
    n1 = (n0 @ M.T).astype(np.float32)
    tr.update_meshes([(v1, n1, f)])
    refit_ms = tr.bvh_info()["mesh_update_ms"]
    b8, bf = tr.render(p, want_f32=True)
    b8, bf = b8.clone(), bf.clone()
    assert not (b8 == a8).all()
    tr.set_meshes([(v1, n1, f)])  # fresh build of the moved mesh
    c8, cf = tr.render(p, want_f32=True)
    assert (c8 == b8).all() and (cf == bf).all()
    sc.set_mesh(v1, n1, f)
    ref_u8, ref_f32, rc = sc.render(op)
    compare(bf, ref_f32, b8, ref_u8)
    assert rc["segments"] > rc["rays"] and refit_ms > 0 and build_ms > 0
    # a different topology is refused (the facade then rebuilds)
    v2, n2, f2 = grt.sphere_mesh(pos, tess_u=16, tess_v=8)
    with pytest.raises(grt.GrtError, match="counts differ"):
        tr.update_meshes([(v2, n2, f2)])
    tr.set_meshes([])


@pytest.mark.parametrize("mesh_type,sh_degree", [(grt.MIRROR, 0), (grt.GLASS, 0), (grt.GLASS, 3)])
def test_bounce_pipeline_every_route_gives_the_same_frame(mesh_type, sh_degree):
    """Mesh frames on the tile kernel: the bounced rays go through k_queue_mesh + the bundle kernel (a wave per 8x8
    tile's rays), chunks over budget through the one-ray-per-wave mode, whatever still bounces after the bundle rounds
    through the per-lane kernel, whose long segments send their ray to the retry queue, to be finished alone on a wave
    (which then traces the mesh itself).  Every split of the work must give the per-lane megakernel's frame bit for bit,
    and the oracle's within tolerance: 0 rounds (per-lane only; with retries; every ray retried), budget 1 (every ray
    alone on a wave, through all its bounces), default, 4 rounds with small budgets; also with degree-3 SH
    (the lone-ray mode evaluates an event's radiance when it is inserted, not when it is composited)."""
    acts, p, sc, op, center = make_scene(51, 20000, 192, 160, scale_boost=0.7, mesh_type=mesh_type, max_bounces=8,
                                         sh_degree=sh_degree)
    eye = np.float32([0, 0, 3])
    base = (0.25 * center + 0.75 * eye).astype(np.float32)
    v1, n1, f1 = grt.sphere_mesh(base + np.float32([0.25, 0, 0]), tess_u=48, tess_v=24)
    v2, n2, f2 = grt.sphere_mesh(base - np.float32([0.45, 0.1, 0.2]), tess_u=32, tess_v=16)
    meshes = [(v1, n1, f1), (v2, n2, f2)]
    ref_t = grt.Tracer(0)
    ref_t.set_option(grt.OPT_KERNEL, 1)
    ref_t.upload(acts)
    ref_t.set_meshes(meshes)
    ref_u8, ref_f32 = ref_t.render(p, want_f32=True)
    ref_t.close()
    t = grt.Tracer(0)
    t.upload(acts)
    t.set_meshes(meshes)
    t.set_option(grt.OPT_COUNTERS, 1)
    routes = {"per-lane only": (0, 1024, 1 << 30), "per-lane, long segments retried alone": (0, 1024, 64),
              "every ray retried alone (own mesh trace)": (0, 1024, 1), "all alone": (2, 1, 128), "default": (2, 1024, 128),
              "deep": (4, 24, 16), "one round": (1, 64, 128)}
    segs = set()
    # (round 6: bundle verdicts, GRT_OPT_BUNDLE_PREDICT — a tile whose bounced rays gave up as a bundle sends them one per wave at
    #  once in the following frames of the SAME view, beside the bundle kernel on a second stream.  Every route runs with verdicts off
    #  (one frame: round 5's pipeline) and on (three frames of a standing view: the first gives the verdicts, the next two use them),
    #  and under views that change — below)
    for name, (rounds, budget, lane_budget) in routes.items():
        for predict, n_frames in ((0, 1), (1, 3)):
            t.set_option(grt.OPT_BUNDLE_PREDICT, predict)
            t.set_option(grt.OPT_BUNDLE_ROUNDS, rounds)
            t.set_option(grt.OPT_BUNDLE_BUDGET, budget)
            t.set_option(grt.OPT_LANE_BUDGET, lane_budget)
            for k in range(n_frames):
                u8, f32 = t.render(p, want_f32=True)
                cnt = t.counters()
                assert bool((u8 == ref_u8).all()), (name, predict, k)
                assert bool((f32 == ref_f32).all()), (name, predict, k)
                assert cnt["stall_exits"] == 0, (name, predict, k)
                segs.add((cnt["segments"], cnt["hit_evals"]))
    assert len(segs) == 1  # the same segments and the same composited events on every route
    if sh_degree == 0:
        # views that change (the eye wobbles): a verdict counts only under the view it was given in, so a frame with other parameters
        # uses none, the second frame of a view that stands uses the first one's, and coming back to a view seen before starts over —
        # each frame equals the per-lane megakernel's frame of the same view
        t.set_option(grt.OPT_BUNDLE_ROUNDS, 2); t.set_option(grt.OPT_BUNDLE_BUDGET, 48); t.set_option(grt.OPT_LANE_BUDGET, 128)
        ref_t = grt.Tracer(0)
        ref_t.set_option(grt.OPT_KERNEL, 1)
        ref_t.upload(acts)
        ref_t.set_meshes(meshes)
        for k, repeats in enumerate((1, 1, 3, 1, 2, 1, 1, 2)):
            q = grt.default_params(192, 160, center, mesh_type=mesh_type, max_bounces=8, sh_degree=sh_degree,
                                   eye=(0.02 * (k % 3), 0.01 * (k % 2), 3.0))
            r8, rf = ref_t.render(q, want_f32=True)
            for j in range(repeats):
                u8, f32 = t.render(q, want_f32=True)
                assert bool((u8 == r8).all()) and bool((f32 == rf).all()), ("changing views", k, j)
        t.check()
        ref_t.close()
    with pytest.raises(grt.GrtError):
        t.set_option(grt.OPT_BUNDLE_ROUNDS, 5)
    t.close()
    sc.set_mesh(np.concatenate([v1, v2]), np.concatenate([n1, n2]), np.concatenate([f1, f2 + len(v1)]))
    o_u8, o_f32, rc = sc.render(op)
    compare(ref_f32, o_f32, ref_u8, o_u8)
    assert next(iter(segs))[0] == rc["segments"] and rc["segments"] > 1.2 * rc["rays"]


def test_camera_inside_a_dense_cluster_full_bags_are_pruned():
    """The eye sits in the densest cell of the scene, inside hundreds of overlapping proxies: their exit events are all
    pending at once and arrive in no order, every lane's window overflows into its bag, full bags keep their nearer half
    (grt_render_tile.hip: bag_prune) and some lanes still go again.  Tile kernel == streaming kernel == per-lane kernel
    bit for bit, and the oracle within tolerance."""
    W, H = 160, 128
    acts, p0, sc, op0, center = make_scene(61, 60000, W, H, scale_boost=1.3)
    pos = acts["pos"]
    h, edges = np.histogramdd(pos, bins=24, range=[(-1.5, 1.5)] * 3)
    i = np.unravel_index(np.argmax(h), h.shape)
    eye = tuple(float((edges[k][i[k]] + edges[k][i[k] + 1]) / 2) for k in range(3))
    p = grt.default_params(W, H, center, eye=eye)
    frames = {}
    for kernel in (0, 3, 1):
        t = grt.Tracer(0)
        t.set_option(grt.OPT_KERNEL, kernel)
        t.upload(acts)
        t.set_option(grt.OPT_COUNTERS, 1)
        u8, f32 = t.render(p, want_f32=True)
        frames[kernel] = (u8.clone(), f32.clone(), t.counters())
        t.close()
    u8, f32, cnt = frames[0]
    assert cnt["stall_exits"] == 0
    assert cnt["proxy_tests"] > 300 * cnt["rays"]        # hundreds of proxies around the eye
    assert cnt["rounds"] > cnt["rays"]                   # some lanes went again ...
    assert cnt["rounds"] < 2.5 * cnt["rays"]             # ... but not a pass per dozen events (3.9 before the pruning)
    for k in (3, 1):
        assert bool((frames[k][0] == u8).all()) and bool((frames[k][1] == f32).all()), k
        assert frames[k][2]["hit_evals"] == cnt["hit_evals"], k
    ref_u8, ref_f32, rc = sc.render(to_oracle_params(p), threads=8)
    compare(f32, ref_f32, u8, ref_u8)
    assert rc["hit_evals"] == cnt["hit_evals"]
    sc.close()
