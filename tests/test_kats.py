"""Analytic known-answer tests of the bounce state machine, written in fp64 HERE (not through the oracle) and checked
against BOTH the CPU oracle and the HIP path: refract entering / leaving / total internal reflection
(shaders/tracer.cuh:432-464), getBarycentricNormal's weight convention u -> v1, v -> v2 (tracer.cuh:167-185), a mirror
bounce with a Gaussian in front of and behind the mirror (shaders/tracer.cu:58-106), and KAT-3 (three
interleaved proxies) through a camera whose centre pixel is the KAT ray, so that the tile / streaming kernels — not
only the ray-buffer kernel — are pinned by it.

Every scene is one ray from the origin plus a mesh of one or two triangles and one or two small isotropic Gaussians
placed ON the analytically expected path: a wrong direction (wrong Snell branch, swapped barycentrics, missing
reflection) makes the ray miss them by many sigma and the pixel comes out black.
"""
import json
import os

import numpy as np
import pytest

import grt
import oracle as O
from common import acts_to_particles, to_oracle_params

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_probes.json")))
C0 = 0.28209479177387814
TOL = 2e-5  # fp64 closed form vs the fp32 pipelines


def unit(v):
    v = np.asarray(v, np.float64)
    return v / np.sqrt((v * v).sum())


def gaussian(pos, sigma, opacity, colour):
    """Raw PLY columns of one isotropic Gaussian whose degree-0 radiance is `colour`."""
    return dict(pos=np.float64(pos), scale=np.full(3, np.log(sigma)), rot=np.float64([1, 0, 0, 0]),
                opacity=np.log(opacity / (1 - opacity)), f_dc=(np.float64(colour) - 0.5) / C0)


def raw_of(gs):
    n = len(gs)
    return dict(pos=np.float32([g["pos"] for g in gs]), f_dc=np.float32([g["f_dc"] for g in gs]),
                f_rest=np.zeros((n, 45), np.float32), opacity=np.float32([g["opacity"] for g in gs]),
                scale=np.float32([g["scale"] for g in gs]), rot=np.float32([g["rot"] for g in gs]))


def through_centre(colour, opacity, T):
    """A ray through the centre of an isotropic Gaussian: response 1, entry and exit hit with the same alpha
    (shaders/tracer.cuh:352-367).  Returns (radiance of the segment, transmittance after it)."""
    a = min(0.99, opacity)
    c = np.float64(colour)
    return T * c * a + T * (1 - a) * c * a, T * (1 - a) ** 2


def big_quad(p0, normal, vnormals=None, half=4.0):
    """Two triangles in the plane through p0 with the given normal (vertex normals default to it)."""
    n = unit(normal)
    a = unit(np.cross(n, [0.3, 1.0, 0.2]))
    b = np.cross(n, a)
    c = p0 + 0.9 * a - 0.4 * b  # p0 well inside one triangle, off the shared diagonal
    v = np.float32([c + half * (-a - b), c + half * (a - b), c + half * (a + b), c + half * (-a + b)])
    vn = np.float32([n] * 4) if vnormals is None else np.float32(vnormals)
    return v, vn, np.uint32([[0, 1, 2], [2, 3, 0]])


def refract64(d, n, etai_over_etat):
    """shaders/tracer.cuh:432-464 in fp64: returns (direction, refracted?)."""
    d, n = np.float64(d), np.float64(n)
    if d @ n < 0:
        ri = 1.0 / etai_over_etat
    else:
        ri, n = etai_over_etat, -n
    cos_t = min(-(d @ n), 1.0)
    sin_t = np.sqrt(1.0 - cos_t * cos_t)
    if ri * sin_t > 1.0:
        rn = n if d @ n < 0 else -n
        return d - 2.0 * rn * (rn @ d), False
    perp = ri * (d + cos_t * n)
    return perp - np.sqrt(abs(1.0 - perp @ perp)) * n, True


# ---------------------------------------------------------------------------------------------------------------
# the KAT scenes: (gaussians, mesh, mesh_type, ray direction, expected pixel, max_bounces)
# ---------------------------------------------------------------------------------------------------------------
D = unit([0.28, -0.1, -0.95])  # the ray, from the origin


def kat_glass(case):
    p_hit = 1.5 * D
    if case == "enter":      # front side: n.d < 0, ri = 1 / 1.4995
        n = unit([-0.9, 0.2, 0.5])
    elif case == "leave":    # back side at a moderate angle: ri = 1.4995, refracts away from the normal
        n = -unit([-0.75, 0.1, 0.6])
    else:                    # "tir": back side beyond the critical angle (41.8 deg): reflects
        n = -unit([-0.95, 0.1, 0.45])
    assert (D @ n < 0) == (case == "enter")
    out, refracted = refract64(D, n, 1.5 / 1.0003)
    assert refracted == (case != "tir")
    out = unit(out)
    straight = p_hit + 0.9 * D
    centre = p_hit + 0.9 * out
    assert np.linalg.norm(centre - straight) > 0.15  # many sigma off the unrefracted path
    colour, opacity = [0.9, 0.55, 0.2], 0.7
    gs = [gaussian(centre, 0.02, opacity, colour)]
    rad, T = through_centre(colour, opacity, 1.0)
    density = 1.0 - T
    pixel = rad * density  # LastGaussianPass: directLight = radiance * alpha, blocking = 0 (shaders/tracer.cu:68-82,101)
    return gs, big_quad(p_hit, n), grt.GLASS, pixel


def kat_barycentric():
    # one triangle, three different vertex normals; the hit has barycentrics (u, v) = (0.55, 0.15):
    # normal = normalize((1-u-v) n0 + u n1 + v n2) with u weighting VERTEX 1 and v VERTEX 2 (tracer.cuh:167-185)
    u, v = 0.55, 0.15
    e1, e2 = np.float64([1.1, 0.2, 0.1]), np.float64([0.1, 1.2, -0.2])
    p_hit = 1.4 * D
    v0 = p_hit - u * e1 - v * e2
    tri = np.float32([v0, v0 + e1, v0 + e2])
    # fp32 vertices move the hit slightly: recompute (t, u, v) from the rounded triangle in fp64
    t0, t1, t2 = np.float64(tri)
    E1, E2 = t1 - t0, t2 - t0
    P = np.cross(D, E2); det = E1 @ P
    Tv = -t0
    uu = (Tv @ P) / det
    Q = np.cross(Tv, E1)
    vv = (D @ Q) / det
    t_hit = (E2 @ Q) / det
    flat = unit(np.cross(E1, E2))
    if flat @ D > 0:
        flat = -flat
    vn = np.float32([unit(flat + [0.25, 0.0, 0.05]), unit(flat + [-0.1, 0.3, 0.0]), unit(flat + [0.0, -0.2, 0.3])])
    n = unit((1 - uu - vv) * np.float64(vn[0]) + uu * np.float64(vn[1]) + vv * np.float64(vn[2]))
    n_swapped = unit((1 - uu - vv) * np.float64(vn[0]) + vv * np.float64(vn[1]) + uu * np.float64(vn[2]))
    out = unit(D - 2.0 * n * (n @ D))
    out_swapped = unit(D - 2.0 * n_swapped * (n_swapped @ D))
    hit = t_hit * D
    centre = hit + 0.8 * out
    assert np.linalg.norm(centre - (hit + 0.8 * out_swapped)) > 0.1  # the swapped convention misses by > 5 sigma
    colour, opacity = [0.3, 0.8, 0.6], 0.6
    gs = [gaussian(centre, 0.02, opacity, colour)]
    rad, T = through_centre(colour, opacity, 1.0)
    pixel = rad * (1.0 - T)
    return gs, (tri, vn, np.uint32([[0, 1, 2]])), grt.MIRROR, pixel


def kat_mirror_two_gaussians():
    # Gaussian A in front of the mirror (GaussianPass: accumColor += radiance, blocking = density), mirror,
    # Gaussian B on the reflected ray (LastGaussianPass with the density carried over): shaders/tracer.cu:58-106
    n = unit([-0.5, 0.1, 0.85])
    p_hit = 2.0 * D
    out = unit(D - 2.0 * n * (n @ D))
    cA, oA = [0.8, 0.3, 0.4], 0.35
    cB, oB = [0.2, 0.7, 0.9], 0.8
    gs = [gaussian(0.9 * D, 0.03, oA, cA), gaussian(p_hit + 1.1 * out, 0.03, oB, cB)]
    rad1, T1 = through_centre(cA, oA, 1.0)
    density1 = 1.0 - T1
    rad2, T2 = through_centre(cB, oB, T1)
    density2 = 1.0 - T2
    pixel = rad1 + rad2 * density2 * (1.0 - density1)
    return gs, big_quad(p_hit, n), grt.MIRROR, pixel


KATS = {"glass_enter": lambda: kat_glass("enter"), "glass_leave": lambda: kat_glass("leave"),
        "glass_tir": lambda: kat_glass("tir"), "barycentric": kat_barycentric, "mirror_two": kat_mirror_two_gaussians}


def camera_for(d, mesh_type):
    """17x17 frame whose centre pixel (8, 8) is the ray from the origin along d: 2 (8.5 / 17) - 1 = 0 exactly and the eye
    sits at the origin, so W = lookat - eye = d bit for bit (shaders/tracer.cuh:115-134)."""
    return grt.default_params(17, 17, np.float32(d), eye=(0.0, 0.0, 0.0), mesh_type=mesh_type, max_bounces=8)


@pytest.mark.parametrize("name", sorted(KATS))
def test_kat_oracle(name):
    gs, mesh, mesh_type, pixel = KATS[name]()
    acts = grt.activate(raw_of(gs))
    sc = O.Scene(acts_to_particles(acts))
    sc.set_mesh(*mesh)
    p = camera_for(D, mesh_type)
    ray = np.float32(np.concatenate([[0, 0, 0], D]))[None]
    out, cnt = sc.render_rays(to_oracle_params(p), ray)
    assert np.abs(out[0] - pixel).max() <= TOL, (out[0], pixel)
    assert pixel.max() > 0.05 and cnt["segments"] == 2
    _, f32, _ = sc.render(to_oracle_params(p))
    assert np.abs(f32[8, 8] - pixel).max() <= TOL
    sc.close()


def test_refract_function_level_against_fp64():
    """grto_refract (the oracle's restatement of tracer.cuh:432-464) on the three branches."""
    import ctypes as C
    L = O.lib(); fp = C.POINTER(C.c_float)
    for n, want_refracted in ((unit([-0.9, 0.2, 0.5]), True), (-unit([-0.75, 0.1, 0.6]), True),
                              (-unit([-0.95, 0.1, 0.45]), False)):
        d32, n32, out = np.float32(D), np.float32(n), np.zeros(3, np.float32)
        r = L.grto_refract(d32.ctypes.data_as(fp), n32.ctypes.data_as(fp), np.float32(1.5 / 1.0003), out.ctypes.data_as(fp))
        want, refracted = refract64(np.float64(d32), np.float64(n32), float(np.float32(1.5) / np.float32(1.0003)))
        assert bool(r) == refracted == want_refracted
        assert np.abs(out - want).max() <= 5e-7


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [0, 1, 3], ids=["tile", "perlane", "stream"])
@pytest.mark.parametrize("name", sorted(KATS))
def test_kat_gpu(name, kernel):
    import torch
    gs, mesh, mesh_type, pixel = KATS[name]()
    tr = grt.Tracer(0)
    try:
        tr.set_option(grt.OPT_KERNEL, kernel)
        tr.upload(grt.activate(raw_of(gs)))
        tr.set_meshes([mesh])
        p = camera_for(D, mesh_type)
        # the ray-buffer entry point (per-lane kernel) ...
        ray = torch.tensor(np.float32(np.concatenate([[0, 0, 0], D]))[None], device="cuda:0")
        out = tr.render_rays(p, ray).cpu().numpy()[0]
        assert np.abs(out - pixel).max() <= TOL, (out, pixel)
        # ... and the frame entry point: wavefront pipeline (primary segment on the tile / streaming kernel, bounces queued)
        _, f32 = tr.render(p, want_f32=True)
        got = f32.cpu().numpy()[8, 8]
        assert np.abs(got - pixel).max() <= TOL, (got, pixel)
    finally:
        tr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [0, 1, 2, 3], ids=["tile", "perlane", "wave", "stream"])
def test_kat3_through_the_camera(kernel):
    """KAT-3 (three interleaved proxies: hit order 2,2,0,1,0,1; pixel and 8-bit value from the survey's run of the
    reference's own device functions) as the centre pixel of a 17x17 frame, on every camera-ray kernel."""
    k = G["kat3"]
    raw = dict(pos=np.float32([r["pos"] for r in k["raw"]]), f_dc=np.float32([r["f_dc"] for r in k["raw"]]),
               f_rest=np.zeros((3, 45), np.float32), opacity=np.float32([r["logit"] for r in k["raw"]]),
               scale=np.float32([r["log_scale"] for r in k["raw"]]), rot=np.float32([r["rot"] for r in k["raw"]]))
    o = np.float32(k["ray_o"]); d = np.float32(k["ray_d_unnormalised"])
    # eye + d must be exact in fp32 for W = lookat - eye to be d: o = (0, 0, 3) and d = (0.01, -0.005, -1) are
    lookat = (o + d).astype(np.float32)
    assert ((lookat - o).astype(np.float32) == d).all()
    p = grt.default_params(17, 17, lookat, eye=tuple(float(x) for x in o))
    tr = grt.Tracer(0)
    try:
        tr.set_option(grt.OPT_KERNEL, kernel)
        tr.upload(grt.activate(raw))
        tr.set_option(grt.OPT_COUNTERS, 1)
        u8, f32 = tr.render(p, window=(8, 8, 9, 9), want_f32=True)
        cnt = tr.counters()
        tr.set_option(grt.OPT_COUNTERS, 0)
        assert cnt["hit_evals"] == 6 and cnt["rays"] == 1 and cnt["stall_exits"] == 0
        np.testing.assert_allclose(f32.cpu().numpy()[8, 8], k["pixel"], atol=1e-5)
        assert u8.cpu().numpy()[8, 8].tolist() == k["u8"]
        u8b, f32b = tr.render(p, want_f32=True)  # and inside a whole frame
        assert (f32b[8, 8] == f32[8, 8]).all() and (u8b[8, 8] == u8[8, 8]).all()
    finally:
        tr.close()
