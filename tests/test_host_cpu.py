"""CPU-only checks of the product's host layer and of the C-ABI surface (no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import grt
import oracle as O
from common import acts_to_particles, make_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "grt.h")).read()
    declared = set(re.findall(r"GRT_API\s+[\w\s\*]+?\b(grt_\w+)\s*\(", hdr))
    assert declared == set(grt.EXPORTS), declared ^ set(grt.EXPORTS)
    L = grt.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = grt.lib().grt_create(C.byref(h), 0)
    assert rc == -3 and not h.value
    assert b"no CPU fallback" in grt.lib().grt_last_error(None)
    with pytest.raises(grt.GrtError):
        grt.Tracer(0)


def test_host_activation_bitexact_vs_oracle():
    raw = grt.synth_scene(11, 2000)
    acts = grt.activate(raw)
    ref = O.activate(raw["pos"], raw["f_dc"], raw["f_rest"], raw["opacity"], raw["scale"], raw["rot"])
    for k in ("pos", "scale", "quat", "opacity", "sh"):
        assert np.ascontiguousarray(acts[k]).tobytes() == np.ascontiguousarray(ref[k]).tobytes(), k
    assert np.allclose(np.linalg.norm(acts["quat"], axis=1), 1, atol=1e-6)
    assert (acts["opacity"] > 0).all() and (acts["opacity"] < 1).all()


def test_host_uvw_bitexact_vs_oracle():
    rng = np.random.default_rng(3)
    for _ in range(100):
        eye = rng.normal(size=3).astype(np.float32) * 3; look = rng.normal(size=3).astype(np.float32)
        fov, asp = float(rng.uniform(20, 100)), float(rng.uniform(0.5, 2.5))
        a = grt.uvw_frame(eye, look, (0, 1, 0), fov, asp)
        b = O.uvw_frame(eye, look, (0, 1, 0), fov, asp)
        for x, y in zip(a, b):
            assert x.tobytes() == y.tobytes()


def test_synth_scene_is_deterministic_and_matches_spec():
    a = grt.synth_scene(2, 20000); b = grt.synth_scene(2, 20000); c = grt.synth_scene(3, 20000)
    for k in a:
        assert a[k].tobytes() == b[k].tobytes()
    assert a["pos"].tobytes() != c["pos"].tobytes()
    # prefix property: particle i does not depend on n except through the scale mean
    d = grt.synth_scene(2, 100)
    assert d["pos"].tobytes() == a["pos"][:100].tobytes()
    assert abs(a["scale"].mean() - np.log(0.7 * 20000 ** (-1 / 3))) < 0.02
    assert abs(a["scale"].std() - 0.5) < 0.02
    assert abs(a["opacity"].mean() - 1.0) < 0.05 and abs(a["opacity"].std() - 2.0) < 0.05
    assert abs(a["f_rest"].std() - 0.1) < 0.005
    assert a["f_dc"].min() >= -1.5 and a["f_dc"].max() <= 1.5


def test_ply_roundtrip_binary_and_ascii(tmp_path):
    raw = grt.synth_scene(5, 300)
    p = str(tmp_path / "s.ply")
    grt.write_ply(p, raw)
    back = grt.read_ply(p)
    for k in raw:
        assert raw[k].tobytes() == back[k].tobytes(), k
    # ascii variant with shuffled property order and a double-typed column: lookup is by NAME
    names = ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{k}" for k in range(45)] + ["opacity"] + \
            [f"scale_{k}" for k in range(3)] + [f"rot_{k}" for k in range(4)]
    cols = np.concatenate([raw["pos"], raw["f_dc"], raw["f_rest"], raw["opacity"][:, None], raw["scale"], raw["rot"]], 1)
    perm = np.random.default_rng(0).permutation(len(names))
    q = str(tmp_path / "a.ply")
    with open(q, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment test\nelement vertex 300\n")
        for j in perm:
            f.write(f"property {'double' if names[j] == 'opacity' else 'float'} {names[j]}\n")
        f.write("end_header\n")
        for row in cols:
            f.write(" ".join(repr(float(row[j])) for j in perm) + "\n")
    back = grt.read_ply(q)
    for k in raw:
        assert raw[k].tobytes() == back[k].tobytes(), k


def test_ply_missing_property_is_an_error(tmp_path):
    q = str(tmp_path / "bad.ply")
    with open(q, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\nend_header\n0 0 0\n")
    with pytest.raises(grt.GrtError, match="missing property"):
        grt.read_ply(q)
    with pytest.raises(grt.GrtError):
        grt.read_ply(str(tmp_path / "nope.ply"))


def test_gaussian_center_matches_sequential_sum():
    raw = grt.synth_scene(1, 5000)
    c = grt.gaussian_center(raw["pos"])
    s = np.zeros(3, np.float32)
    for row in raw["pos"]:
        s += row
    assert (s / np.float32(5000)).astype(np.float32).tobytes() == c.tobytes()


# ---- oracle frame-level self-consistency (brute force over all proxies == BVH) ----
@pytest.mark.parametrize("cfg", [dict(), dict(sh_degree=3), dict(fisheye=True)])
def test_oracle_bvh_equals_bruteforce(cfg):
    acts, p, sc, op, center = make_scene(21, 600, 48, 48, scale_boost=1.0, **cfg)
    u8a, fa, ca = sc.render(op)
    sc.use_bvh(0)
    u8b, fb, cb = sc.render(op)
    assert fa.tobytes() == fb.tobytes() and u8a.tobytes() == u8b.tobytes()
    assert ca["hit_evals"] == cb["hit_evals"] and ca["rays"] == cb["rays"]
    assert ca["hit_evals"] > 2 * 48 * 48  # the scene is actually being hit
    if cfg.get("fisheye"):
        assert ca["rays"] < 48 * 48 and (fa[0, 0] == 0).all()


@pytest.mark.parametrize("mesh_type", [grt.MIRROR, grt.NORMAL, grt.GLASS])
def test_oracle_mesh_bvh_equals_bruteforce(mesh_type):
    acts, p, sc, op, center = make_scene(22, 400, 40, 40, scale_boost=1.0, mesh_type=mesh_type)
    eye = np.array([0, 0, 3], np.float32)
    pos = (0.25 * center + 0.75 * eye).astype(np.float32)  # GaussianTracer.cpp:580-588
    v, n, f = grt.sphere_mesh(pos, tess_u=24, tess_v=12)
    sc.set_mesh(v, n, f)
    _, fa, ca = sc.render(op)
    sc.use_bvh(0)
    _, fb, cb = sc.render(op)
    assert fa.tobytes() == fb.tobytes()
    assert ca["segments"] > ca["rays"] or mesh_type == grt.NORMAL  # secondary segments exist


def _write_ply_with_rest(path, raw, rest_cols):
    n = len(raw["pos"])
    k = rest_cols.shape[1]
    names = ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{j}" for j in range(k)] + ["opacity"] + \
            [f"scale_{j}" for j in range(3)] + [f"rot_{j}" for j in range(4)]
    cols = np.concatenate([raw["pos"], raw["f_dc"], rest_cols, raw["opacity"][:, None], raw["scale"], raw["rot"]], 1)
    with open(path, "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {n}\n" +
                 "".join(f"property float {m}\n" for m in names) + "end_header\n").encode())
        f.write(np.ascontiguousarray(cols, np.float32).tobytes())


@pytest.mark.parametrize("deg", [0, 1, 2])
def test_ply_with_fewer_sh_bands_maps_channels(tmp_path, deg):
    """3DGS exports trained at SH degree L < 3 carry 3K f_rest columns, K = (L+1)^2-1, channel-major with stride K
    (SURVEY §8(f) rank 4).  Column c*K+j must land where the degree-3 layout keeps channel c, coefficient j
    (slot c*15+j), so that GaussianData.cpp:113-128's sh[k] = (f_rest[k-1], f_rest[14+k], f_rest[29+k]) sees the
    right colour channel; the bands the file lacks are zero."""
    raw = grt.synth_scene(6, 40)
    K = (deg + 1) ** 2 - 1
    rest = np.random.default_rng(deg).normal(size=(40, 3 * K)).astype(np.float32)
    q = str(tmp_path / f"deg{deg}.ply")
    _write_ply_with_rest(q, raw, rest)
    back = grt.read_ply(q)
    want = np.zeros((40, 45), np.float32)
    for c in range(3):
        want[:, c * 15:c * 15 + K] = rest[:, c * K:(c + 1) * K]
    assert back["f_rest"].tobytes() == want.tobytes()
    for k in ("pos", "f_dc", "opacity", "scale", "rot"):
        assert back[k].tobytes() == raw[k].tobytes()
    # through the activation: sh[1+j] of a particle = (R_j, G_j, B_j) of the file
    acts = grt.activate(back)
    for j in range(K):
        assert (acts["sh"][:, 1 + j, 0] == rest[:, j]).all() and (acts["sh"][:, 1 + j, 1] == rest[:, K + j]).all() \
            and (acts["sh"][:, 1 + j, 2] == rest[:, 2 * K + j]).all()
    assert (acts["sh"][:, 1 + K:] == 0).all()


def test_ply_with_odd_f_rest_count_is_rejected(tmp_path):
    raw = grt.synth_scene(6, 8)
    q = str(tmp_path / "bad.ply")
    _write_ply_with_rest(q, raw, np.zeros((8, 10), np.float32))
    with pytest.raises(grt.GrtError, match="f_rest"):
        grt.read_ply(q)


def test_generated_slot_macros_are_current():
    """csrc/grt_slots_gen.inc is generated: the committed file must be what gen_slots.py prints."""
    import subprocess
    import sys
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussian-ray-tracing_amd", "csrc")
    out = subprocess.run([sys.executable, os.path.join(d, "gen_slots.py")], capture_output=True, text=True, check=True).stdout
    assert out == open(os.path.join(d, "grt_slots_gen.inc")).read()


def _fnv(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a).tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def test_procedural_primitives_are_unchanged():
    """grt_host_primitive_* (table-driven lattice) produce, bit for bit, the arrays the round-1 facade produced by
    following src/geometry/Primitives.cpp:6-140 statement by statement (FNV-1a of the raw bytes recorded then)."""
    v, n, f = grt.primitive_mesh(grt.PRIM_PLANE)
    assert (v.shape, f.shape) == ((4, 3), (2, 3))
    assert (_fnv(v), _fnv(n), _fnv(f)) == ("7a2c5c8c355ac993", "559fdd29b636fee3", "5e76009c85122600")
    v, n, f = grt.primitive_mesh(grt.PRIM_SPHERE)
    assert (v.shape, f.shape) == ((16290, 3), (32040, 3))
    assert (_fnv(v), _fnv(n), _fnv(f)) == ("644ab7773a63ff81", "1384e9db3e63bbe6", "45d53d77f94efe52")
    # unit normals, radius 0.3, south pole first (theta = 0 -> +y)
    assert np.abs(np.linalg.norm(n, axis=1) - 1).max() < 1e-6 and np.abs(np.linalg.norm(v, axis=1) - 0.3).max() < 1e-6
    assert v[0, 1] == np.float32(0.3)
    # the numpy formulation used for the small test meshes agrees to rounding
    v2, n2, f2 = grt.sphere_mesh((0, 0, 0))
    assert (f2 == f).all() and np.abs(v2 - v).max() < 1e-6


def test_obj_round_trip_flips_y_and_unindexes(tmp_path):
    """Primitives::createLoadMesh (src/geometry/Primitives.cpp:142-202): one vertex per face corner in file order, Y of
    positions and normals negated; polygons fan-triangulated; a corner without a normal is an error."""
    v, n, f = grt.primitive_mesh(grt.PRIM_SPHERE)
    q = str(tmp_path / "s.obj")
    grt.write_obj(q, v, n, f)
    mv, mn, mf = grt.load_obj(q, center=(0.5, 0.0, -1.0))
    flip = np.float32([1, -1, 1])
    assert (mf == np.arange(3 * len(f), dtype=np.uint32).reshape(-1, 3)).all()
    assert (mv == (v[f.reshape(-1)] * flip + np.float32([0.5, 0.0, -1.0])).astype(np.float32)).all()
    assert (mn == n[f.reshape(-1)] * flip).all()
    quad = str(tmp_path / "quad.obj")
    open(quad, "w").write("v 0 0 0\nv 1 0 0\nv 1 2 0\nv 0 2 0\nvn 0 0 1\nf 1//1 2//1 3//1 -1//-1\n")
    qv, qn, qf = grt.load_obj(quad)
    assert qv.tolist() == [[0, 0, 0], [1, 0, 0], [1, -2, 0], [0, 0, 0], [1, -2, 0], [0, -2, 0]] and (qn == [0, 0, 1]).all()
    bad = str(tmp_path / "bad.obj")
    open(bad, "w").write("v 0 0 0\nv 1 0 0\nv 1 1 0\nf 1 2 3\n")
    with pytest.raises(grt.GrtError, match="normal"):
        grt.load_obj(bad)
