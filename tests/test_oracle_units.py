"""Pins the CPU oracle (oracle/grt_oracle.c) function by function:
 (a) against the probe values the reference's own device code printed (tests/golden/survey_probes.json),
 (b) bit-for-bit against oracle/_ref (reference sources compiled in place) where that library exists,
 (c) against the analytic known-answer tests of SURVEY.md §4.2."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle as O

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_probes.json")))
fp = C.POINTER(C.c_float)


def f32(a):
    return np.ascontiguousarray(a, np.float32)


def ptr(a):
    return a.ctypes.data_as(fp)


def probe_particle():
    p = np.zeros(1, O.PARTICLE_DTYPE)
    q = f32(G["probe_particle"]["quat_wxyz_unnormalised"])
    q = q / np.float32(np.sqrt(np.float32((q * q).sum())))
    p["pos"] = G["probe_particle"]["pos"]
    p["scale"] = G["probe_particle"]["scale"]
    p["quat"] = q
    p["opacity"] = 0.7
    for i in range(16):
        p["sh"][0, i] = (0.1 * i - 0.5, 0.05 * i, -0.02 * i + 0.3)
    return p


def probe_ray():
    o = f32(G["probe_ray"]["o"])
    d = f32(G["probe_ray"]["d_unnormalised"])
    d = d / np.float32(np.sqrt(np.float32((d * d).sum())))
    return o, f32(d)


def test_struct_size():
    assert O.PARTICLE_DTYPE.itemsize == G["sizes"]["GaussianParticle"]


def test_compute_response_probe():
    p = probe_particle(); o, d = probe_ray()
    r = O.lib().grto_compute_response(p.ctypes.data, ptr(o), ptr(d))
    assert abs(r - G["computeResponse"]) < 2e-6


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_compute_radiance_probe(deg):
    p = probe_particle(); o, d = probe_ray()
    L = O.lib()
    L.grto_compute_radiance.argtypes = [C.c_void_p, fp, C.c_uint32, fp]
    rgb = np.zeros(3, np.float32)
    L.grto_compute_radiance(p.ctypes.data, ptr(d), deg, ptr(rgb))
    np.testing.assert_allclose(rgb, G["computeRadiance"][str(deg)], atol=2e-6)


def test_get_ray_probe():
    g = G["getRay"]; L = O.lib()
    U, V, W = f32(g["U"]), f32(g["V"]), f32(g["W"])
    eye = f32([0, 0, 3]); o = np.zeros(3, np.float32); d = np.zeros(3, np.float32)
    L.grto_get_ray.argtypes = [C.c_uint32, C.c_uint32, fp, fp, fp, fp, C.c_uint32, C.c_uint32, fp, fp]
    L.grto_get_ray(g["px"][0], g["px"][1], ptr(U), ptr(V), ptr(W), ptr(eye), g["dim"][0], g["dim"][1], ptr(o), ptr(d))
    np.testing.assert_allclose(d, g["dir"], atol=1e-6)
    np.testing.assert_array_equal(o, eye)


def test_fisheye_ray_probe():
    g = G["getRay"]; L = O.lib()
    U, V, W = f32(g["U"]), f32(g["V"]), f32(g["W"])
    eye = f32([0, 0, 3]); o = np.zeros(3, np.float32); d = np.zeros(3, np.float32)
    L.grto_get_fisheye_ray.argtypes = [C.c_uint32, C.c_uint32, fp, fp, fp, fp, C.c_uint32, C.c_uint32, fp, fp]
    px = G["getFishEyeRay"]["px"]
    assert L.grto_get_fisheye_ray(px[0], px[1], ptr(U), ptr(V), ptr(W), ptr(eye), g["dim"][0], g["dim"][1], ptr(o), ptr(d)) == 1
    np.testing.assert_allclose(d, G["getFishEyeRay"]["dir"], atol=1e-6)
    # corner pixel: r > 1 -> no ray (decision vii)
    assert L.grto_get_fisheye_ray(0, 0, ptr(U), ptr(V), ptr(W), ptr(eye), g["dim"][0], g["dim"][1], ptr(o), ptr(d)) == 0


def test_quantize_probe():
    for x, q in zip(G["quantize"]["in"], G["quantize"]["out"]):
        assert O.lib().grto_quantize(x) == q
    assert O.lib().grto_quantize(-1.0) == 0 and O.lib().grto_quantize(2.0) == 255
    assert O.lib().grto_quantize(1.0 / 256) == 1 and O.lib().grto_quantize(np.nextafter(np.float32(1 / 256), np.float32(0))) == 0


def test_uvw_probe():
    g = G["UVWFrame"]
    U, V, W = O.uvw_frame(g["eye"], g["lookat"], g["up"], g["fov"], g["aspect"])
    np.testing.assert_allclose(U, g["U"], atol=3e-7 * 4)
    np.testing.assert_allclose(V, g["V"], atol=3e-7 * 4)
    np.testing.assert_allclose(W, g["W"], atol=3e-7 * 4)


def test_icosahedron_and_slab_normals():
    L = O.lib()
    v = np.zeros((12, 3), np.float32); idx = np.zeros(60, np.uint32)
    L.grto_icosahedron(v.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p))
    np.testing.assert_allclose(v[0], G["icosahedron_v0"], atol=1e-7)
    n = np.zeros((10, 3), np.float32)
    L.grto_slab_normals(n.ctypes.data_as(C.c_void_p))
    # every face plane of the mesh is one of +-n_i at distance exactly 1 (fp64 check)
    vd = v.astype(np.float64)
    seen = set()
    for f in idx.reshape(20, 3):
        a, b, c = vd[f]
        fn = np.cross(b - a, c - a); fn /= np.linalg.norm(fn)
        assert abs(fn @ a - 1.0) < 1e-6
        assert fn @ a > 0  # CCW outward
        k = np.argmax(np.abs(n.astype(np.float64) @ fn))
        assert abs(abs(n[k].astype(np.float64) @ fn) - 1.0) < 1e-6
        seen.add((int(k), int(np.sign(n[k] @ fn))))
    assert len(seen) == 20


def _kat3_particles():
    k = G["kat3"]["raw"]
    n = len(k)
    return O.activate(f32([r["pos"] for r in k]), f32([r["f_dc"] for r in k]), np.zeros((n, 45), np.float32),
                      f32([r["logit"] for r in k]), f32([r["log_scale"] for r in k]), f32([r["rot"] for r in k]))


def test_kat3_function_level():
    k = G["kat3"]; L = O.lib()
    parts = _kat3_particles()
    o = f32(k["ray_o"]); d = f32(k["ray_d_unnormalised"]); d = f32(d / np.float32(np.sqrt(np.float32((d * d).sum()))))
    np.testing.assert_allclose(parts["opacity"], k["opacity"], atol=1e-6)
    L.grto_compute_radiance.argtypes = [C.c_void_p, fp, C.c_uint32, fp]
    for i in range(3):
        pi = parts[i:i + 1]
        s = L.grto_proxy_scale(float(pi["opacity"][0]), 0.01)
        assert abs(s - k["proxy_s"][i]) < 1e-5
        te = C.c_float(); tx = C.c_float()
        assert L.grto_proxy_hit(pi.ctypes.data, 0.01, ptr(o), ptr(d), C.byref(te), C.byref(tx)) == 1
        assert abs(te.value - k["t_entry_exit"][i][0]) < 1e-5
        assert abs(tx.value - k["t_entry_exit"][i][1]) < 1e-5
        r = L.grto_compute_response(pi.ctypes.data, ptr(o), ptr(d))
        assert abs(r - k["response"][i]) < 1e-5
        rgb = np.zeros(3, np.float32)
        L.grto_compute_radiance(pi.ctypes.data, ptr(d), 0, ptr(rgb))
        np.testing.assert_allclose(rgb, k["colour"][i], atol=1e-6)


@pytest.mark.parametrize("use_bvh", [0, 1])
def test_kat3_trace_and_pixel(use_bvh):
    k = G["kat3"]
    sc = O.Scene(_kat3_particles())
    sc.use_bvh(use_bvh)
    o = f32(k["ray_o"]); d = f32(k["ray_d_unnormalised"]); d = f32(d / np.float32(np.sqrt(np.float32((d * d).sum()))))
    n, ids, ts = sc.trace_gps(o, d, 1e-3, 1e5)
    assert n == 6
    assert list(ids[:6]) == k["hit_order"]
    assert ids[6] == 0xFFFFFFFF and ts[6] == np.float32(1e20)
    prm = O.make_params(1, 1, o, [1, 0, 0], [0, 1, 0], [0, 0, -1])
    rad, dens = sc.trace(prm, o, d, 1e-3, 1e5)
    np.testing.assert_allclose(rad, k["radiance"], atol=1e-5)
    assert abs(dens - k["density"]) < 1e-5
    rays = np.concatenate([o, d])[None]
    out, cnt = sc.render_rays(prm, rays)
    np.testing.assert_allclose(out[0], k["pixel"], atol=1e-5)
    assert [O.lib().grto_quantize(float(x)) for x in out[0]] == k["u8"]
    assert cnt["hit_evals"] == 6 and cnt["rays"] == 1


def test_kat_single_gaussian_through_centre():
    """one Gaussian, ray through its centre, opacity 0.8, colour c:
    alpha = 0.8, entry+exit => radiance 0.96 c, density 0.96, pixel 0.9216 c."""
    logit = float(np.log(0.8 / 0.2))
    parts = O.activate(f32([[0, 0, 0]]), f32([[1.0, 0.0, -1.0]]), np.zeros((1, 45), np.float32), f32([logit]),
                       f32([[-2, -2, -2]]), f32([[1, 0, 0, 0]]))
    sc = O.Scene(parts)
    o = f32([0, 0, 3]); d = f32([0, 0, -1])
    prm = O.make_params(1, 1, o, [1, 0, 0], [0, 1, 0], [0, 0, -1])
    c = np.maximum(0.0, 0.5 + 0.28209479177387814 * np.array([1.0, 0.0, -1.0]))
    rad, dens = sc.trace(prm, o, d, 1e-3, 1e5)
    np.testing.assert_allclose(rad, 0.96 * c, atol=2e-6)
    assert abs(dens - 0.96) < 2e-6
    out, _ = sc.render_rays(prm, np.concatenate([o, d])[None])
    np.testing.assert_allclose(out[0], 0.9216 * c, atol=3e-6)


def test_low_opacity_particle_is_unhittable():
    """decision (vi): opacity <= alpha_min => NaN/zero proxy scale => never hit (GaussianTracer.cpp:306)."""
    logit = float(np.log(0.005 / 0.995))
    parts = O.activate(f32([[0, 0, 0]]), f32([[1, 1, 1]]), np.zeros((1, 45), np.float32), f32([logit]),
                       f32([[-2, -2, -2]]), f32([[1, 0, 0, 0]]))
    sc = O.Scene(parts)
    n, ids, ts = sc.trace_gps([0, 0, 3], [0, 0, -1], 1e-3, 1e5)
    assert n == 0 and ids[0] == 0xFFFFFFFF


def test_origin_inside_proxy_gives_exit_only():
    parts = O.activate(f32([[0, 0, 0]]), f32([[1, 1, 1]]), np.zeros((1, 45), np.float32), f32([2.0]),
                       f32([[-1, -1, -1]]), f32([[1, 0, 0, 0]]))
    sc = O.Scene(parts)
    n, ids, ts = sc.trace_gps([0, 0, 0.1], [0, 0, -1], 1e-3, 1e5)
    assert n == 1 and ids[0] == 0 and ts[0] > 0


# ---------------------------------------------------------------------------------------------
# bit-exact pins against the reference sources compiled in place (oracle/_ref)
# ---------------------------------------------------------------------------------------------
needs_ref = pytest.mark.skipif(not os.path.exists(O._REF), reason="oracle/_ref/libgrt_ref.so not built")  # (file check only: the .so is loaded by the CPU tests that use it, never at collection)


@needs_ref
def test_ref_uvw_frame_bitexact():
    rng = np.random.default_rng(7)
    R = O.ref()
    for _ in range(200):
        eye = f32(rng.normal(size=3) * 3); look = f32(rng.normal(size=3)); up = f32([0, 1, 0])
        fov = float(rng.uniform(20, 100)); asp = float(rng.uniform(0.5, 2.5))
        U, V, W = O.uvw_frame(eye, look, up, fov, asp)
        u2 = np.zeros(3, np.float32); v2 = np.zeros(3, np.float32); w2 = np.zeros(3, np.float32)
        R.ref_uvw_frame(ptr(eye), ptr(look), ptr(up), C.c_float(fov), C.c_float(asp), ptr(u2), ptr(v2), ptr(w2))
        assert U.tobytes() == u2.tobytes() and V.tobytes() == v2.tobytes() and W.tobytes() == w2.tobytes()


@needs_ref
def test_ref_icosahedron_bitexact():
    v = np.zeros((12, 3), np.float32); idx = np.zeros(60, np.uint32)
    O.lib().grto_icosahedron(v.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p))
    v2 = np.zeros(36, np.float32); i2 = np.zeros(60, np.uint32)
    O.ref().ref_icosahedron(ptr(v2), i2.ctypes.data_as(C.c_void_p))
    assert v.tobytes() == v2.tobytes() and idx.tobytes() == i2.tobytes()


@needs_ref
def test_ref_glm_invcov_and_matvec_bitexact():
    rng = np.random.default_rng(11)
    R = O.ref(); L = O.lib()
    for _ in range(300):
        p = np.zeros(1, O.PARTICLE_DTYPE)
        q = f32(rng.normal(size=4)); q = f32(q / np.float32(np.sqrt(np.float32((q * q).sum()))))
        p["quat"] = q; p["scale"] = f32(np.exp(rng.normal(-3, 1, size=3))); p["pos"] = f32(rng.normal(size=3))
        A = np.zeros(9, np.float32); A2 = np.zeros(9, np.float32)
        L.grto_inv_cov(C.c_void_p(p.ctypes.data), ptr(A))
        sc = f32(p["scale"][0]); qq = f32(p["quat"][0])
        R.ref_inv_cov(ptr(sc), ptr(qq), ptr(A2))
        assert A.tobytes() == A2.tobytes()
        Rg = np.zeros(9, np.float32); Rg2 = np.zeros(9, np.float32)
        L.grto_mat3_cast(ptr(qq), ptr(Rg)); R.ref_mat3_cast(ptr(qq), ptr(Rg2))
        assert Rg.tobytes() == Rg2.tobytes()


@needs_ref
def test_ref_response_chain_bitexact():
    """computeResponse rebuilt from the reference's glm pieces (mat*vec, dot) equals the oracle's value."""
    rng = np.random.default_rng(13)
    R = O.ref(); L = O.lib()
    for _ in range(200):
        p = np.zeros(1, O.PARTICLE_DTYPE)
        q = f32(rng.normal(size=4)); q = f32(q / np.float32(np.sqrt(np.float32((q * q).sum()))))
        p["quat"] = q; p["scale"] = f32(np.exp(rng.normal(-2, 0.7, size=3))); p["pos"] = f32(rng.normal(size=3) * 0.3)
        o = f32([0, 0, 3]); d = f32(rng.normal(size=3) * 0.1 + [0, 0, -1])
        dn = np.zeros(3, np.float32); R.ref_normalize(ptr(d), ptr(dn))
        A = np.zeros(9, np.float32)
        R.ref_inv_cov(ptr(f32(p["scale"][0])), ptr(f32(p["quat"][0])), ptr(A))
        mu = f32(p["pos"][0])
        og = np.zeros(3, np.float32); dg = np.zeros(3, np.float32); pg = np.zeros(3, np.float32)
        R.ref_mat3_vec(ptr(A), ptr(f32(o - mu)), ptr(og)); R.ref_mat3_vec(ptr(A), ptr(dn), ptr(dg))
        dval = np.float32(-np.float32(R.ref_glm_dot(ptr(og), ptr(dg))) / max(np.float32(1e-6), np.float32(R.ref_glm_dot(ptr(dg), ptr(dg)))))
        pos = f32(o + f32(dval * dn))
        R.ref_mat3_vec(ptr(A), ptr(f32(mu - pos)), ptr(pg))
        expect_arg = np.float32(np.float32(-0.5) * np.float32(R.ref_glm_dot(ptr(pg), ptr(pg))))
        got = L.grto_compute_response(p.ctypes.data, ptr(o), ptr(dn))
        # same argument bit-for-bit => same expf(); compare through log to avoid libm identity assumptions
        assert abs(got - float(np.exp(np.float64(expect_arg)))) <= 1.2e-7 * max(1.0, got)


@needs_ref
def test_ref_vector_math_bitexact():
    rng = np.random.default_rng(5)
    R = O.ref(); L = O.lib()
    for _ in range(200):
        a = f32(rng.normal(size=3)); n = f32(rng.normal(size=3))
        nn = np.zeros(3, np.float32); R.ref_normalize(ptr(n), ptr(nn))
        r1 = np.zeros(3, np.float32); r2 = np.zeros(3, np.float32)
        L.grto_reflect(ptr(a), ptr(nn), ptr(r1)); R.ref_reflect(ptr(a), ptr(nn), ptr(r2))
        assert r1.tobytes() == r2.tobytes()


@needs_ref
def test_ref_proxy_vertices_inside_oracle_aabb_and_on_slab_planes():
    """The reference instance transform T*(R*S) puts the 12 proxy vertices where the oracle's slab
    polytope has its vertices: every vertex satisfies max_i |n_i.A(v-mu)| == s (to rounding)."""
    rng = np.random.default_rng(3)
    R = O.ref(); L = O.lib()
    v = np.zeros((12, 3), np.float32); idx = np.zeros(60, np.uint32)
    L.grto_icosahedron(v.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p))
    n = np.zeros((10, 3), np.float32); L.grto_slab_normals(n.ctypes.data_as(C.c_void_p))
    for _ in range(50):
        p = np.zeros(1, O.PARTICLE_DTYPE)
        q = f32(rng.normal(size=4)); q = f32(q / np.float32(np.sqrt(np.float32((q * q).sum()))))
        p["quat"] = q; p["scale"] = f32(np.exp(rng.normal(-2, 0.7, size=3))); p["pos"] = f32(rng.normal(size=3))
        p["opacity"] = 0.6
        s = L.grto_proxy_scale(0.6, 0.01)
        A = np.zeros(9, np.float32); L.grto_inv_cov(C.c_void_p(p.ctypes.data), ptr(A))
        A = A.reshape(3, 3).astype(np.float64)
        for k in range(12):
            w = np.zeros(3, np.float32)
            R.ref_instance_vertex(ptr(f32(p["pos"][0])), ptr(f32(p["scale"][0])), ptr(f32(p["quat"][0])), C.c_float(s), ptr(v[k]), ptr(w))
            g = A @ (w.astype(np.float64) - p["pos"][0].astype(np.float64))
            m = np.max(np.abs(n.astype(np.float64) @ g))
            assert abs(m - s) < 2e-3 * s


# ---------------------------------------------------------------------------------------------
# decision (v) on whole frames: ten slabs in Gaussian space == the reference's 20 instanced triangles
# ---------------------------------------------------------------------------------------------
def _small_scene(n=300, seed=21):
    rng = np.random.default_rng(seed)
    pos = f32(rng.normal(0.0, 0.45, size=(n, 3)))
    f_dc = f32(rng.uniform(-1.5, 1.5, size=(n, 3)))
    f_rest = f32(rng.normal(0.0, 0.1, size=(n, 45)))
    logit = f32(rng.normal(1.0, 2.0, size=n))
    log_scale = f32(rng.normal(np.log(0.12), 0.6, size=(n, 3)))  # anisotropic: axes differ by e^(+-0.6)
    rot = f32(rng.normal(size=(n, 4)))
    return O.activate(pos, f_dc, f_rest, logit, log_scale, rot)


def _ref_instanced_icosahedra(parts, alpha_min=0.01):
    """World-space vertices [n][12][3] of every particle's proxy: the reference's OWN mesh (src/geometry/Icosahedron.h:13-37) through the
    reference's OWN instance transform T * (R * S) (src/GaussianTracer.cpp:304-311: glm translate / mat4_cast / scale with
    s = sqrtf(2 logf(opacity / alpha_min)), :306), both compiled from the reference's sources in oracle/_ref."""
    R = O.ref()
    base = np.zeros(36, np.float32); idx = np.zeros(60, np.uint32)
    R.ref_icosahedron(ptr(base), idx.ctypes.data_as(C.c_void_p))
    base = base.reshape(12, 3)
    out = np.zeros((len(parts), 12, 3), np.float32)
    for i, p in enumerate(parts):
        if not p["opacity"] > alpha_min:
            continue  # NaN / zero transform in the reference: unhittable (decision (vi)); the oracle never tests it
        s = np.float32(np.sqrt(np.float32(2.0) * np.float32(np.log(np.float32(p["opacity"] / np.float32(alpha_min))))))
        assert abs(float(s) - O.lib().grto_proxy_scale(float(p["opacity"]), alpha_min)) <= 2e-7 * float(s)
        for k in range(12):
            w = np.zeros(3, np.float32)
            R.ref_instance_vertex(ptr(f32(p["pos"])), ptr(f32(p["scale"])), ptr(f32(p["quat"])), C.c_float(float(s)), ptr(f32(base[k])), ptr(w))
            out[i, k] = w
    return out, idx


@needs_ref
@pytest.mark.parametrize("mode", ["pinhole", "fisheye", "mirror", "glass_sh2"])
def test_decision_v_slabs_equal_the_reference_built_triangles_on_whole_frames(mode):
    """SURVEY 8(c) decision (v) replaces OptiX's ray / triangle tests against the instanced icosahedron by ten slab tests in Gaussian
    space; until round 6 the only evidence for it on whole frames was a probe the survey session ran.  Here the oracle renders small
    frames twice: with the slabs, and with every proxy intersected as its 20 TRIANGLES — vertices from the reference's own mesh and
    instance transform (oracle/_ref), double-precision Moeller-Trumbore, no culling, brute force over all particles (no BVH of ours
    in the way) — feeding the same k-buffer and integrator (shaders/tracer.cu:136-153, tracer.cuh:328-373).  0 differing 8-bit
    levels, radiance within 1e-6, identical hit counts."""
    parts = _small_scene()
    W = H = 48
    eye = f32([0, 0, 3]); look = f32(parts["pos"].mean(0)); up = f32([0, 1, 0])
    U, V, Wv = O.uvw_frame(eye, look, up, 60.0, 1.0)
    kw = dict(pinhole={}, fisheye=dict(fisheye=True), mirror=dict(mesh_type=0, max_bounces=4),
              glass_sh2=dict(mesh_type=2, max_bounces=6, sh_degree=2))[mode]
    prm = O.make_params(W, H, eye, U, V, Wv, **kw)
    sc = O.Scene(parts)
    if mode in ("mirror", "glass_sh2"):
        # a quad in the middle of the cloud with NON-flat vertex normals (the probe's set-up): rays bounce / refract into the Gaussians behind
        mv = f32([[-0.9, -0.9, 0.2], [0.9, -0.9, 0.2], [0.9, 0.9, 0.1], [-0.9, 0.9, 0.3]])
        mn = f32([[0.1, 0.0, 1.0], [-0.2, 0.1, 1.0], [0.0, -0.15, 1.0], [0.15, 0.1, 1.0]])
        mn = f32(mn / np.linalg.norm(mn, axis=1, keepdims=True))
        sc.set_mesh(mv, mn, np.uint32([[0, 1, 2], [0, 2, 3]]))
    u8_s, f_s, c_s = sc.render(prm, threads=8)
    verts, idx = _ref_instanced_icosahedra(parts)
    sc.set_proxy_triangles(verts, idx)
    sc.use_bvh(False)
    u8_t, f_t, c_t = sc.render(prm, threads=8)
    sc.close()
    assert c_s["hit_evals"] > 8 * W * H * (0.5 if mode == "fisheye" else 1.0)  # the frames are full of Gaussians
    assert c_s["hit_evals"] == c_t["hit_evals"] and c_s["segments"] == c_t["segments"]
    assert (u8_s == u8_t).all(), f"{(u8_s != u8_t).sum()} 8-bit levels differ"
    assert float(np.abs(f_s - f_t).max()) <= 1e-6
    if mode != "pinhole":
        assert c_s["segments"] > c_s["rays"] or mode == "fisheye"
