"""The C++ facade (GaussianTracer mirror) through the headless CLI gaussian-ray-tracing_amd/grt_render."""
import os
import subprocess

import numpy as np
import pytest

import grt
import oracle as O
from common import acts_to_particles, to_oracle_params, u8_matches

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "gaussian-ray-tracing_amd", "grt_render")


def _read_ppm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P6"
        w, h = map(int, f.readline().split())
        assert f.readline().strip() == b"255"
        return np.frombuffer(f.read(), np.uint8).reshape(h, w, 3)[::-1]  # file is top-down, frame row 0 = bottom


def _read_png(path):
    import struct
    import zlib
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(typ + data) & 0xFFFFFFFF)
        if typ == b"IHDR":
            w, h, depth, colour = struct.unpack(">IIBB", data[:10])
            assert (depth, colour) == (8, 2)
        elif typ == b"IDAT":
            idat += data
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * 3 + 1)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)[::-1]  # file is top-down, frame row 0 = bottom


def _scene(tmp_path, seed=12, n=3000):
    raw = grt.synth_scene(seed, n)
    raw["scale"] = raw["scale"] + np.float32(0.6)
    ply = str(tmp_path / "scene.ply")
    grt.write_ply(ply, raw)
    return ply, grt.activate(raw)


def test_cli_exists_and_fails_loudly_without_gpu(tmp_path):
    import torch
    assert os.path.exists(CLI), "run __graft_entry__.build()"
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ply, _ = _scene(tmp_path, n=50)
    r = subprocess.run([CLI, "-p", ply, "--width", "32", "--height", "32"], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr
    r = subprocess.run([CLI, "-p", str(tmp_path / "missing.ply")], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr


@pytest.mark.gpu
def test_cli_frame_matches_oracle(tmp_path):
    ply, acts = _scene(tmp_path)
    out = str(tmp_path / "f.ppm")
    r = subprocess.run([CLI, "-p", ply, "--width", "160", "--height", "96", "--out", out, "--bench", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Mrays/s" in r.stdout
    p = grt.default_params(160, 96, grt.gaussian_center(acts["pos"]))
    sc = O.Scene(acts_to_particles(acts))
    ref, ref_f32, cnt = sc.render(to_oracle_params(p))
    got = _read_ppm(out)
    # the exact-u8 rule of the parity tests: equal, except within 1e-4 of a quantisation step
    assert u8_matches(got, ref, ref_f32).all() and cnt["hit_evals"] > 160 * 96
    # the frame as .npy (the renderer's own row order): numpy.load gives the same bytes
    npy = str(tmp_path / "f.npy")
    r = subprocess.run([CLI, "-p", ply, "--width", "160", "--height", "96", "--out", npy], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    arr = np.load(npy)
    assert arr.shape == (96, 160, 3) and arr.dtype == np.uint8 and (arr == got).all()
    # the same frame as PNG (stored-deflate writer in the CLI): decodes to the same bytes
    png = str(tmp_path / "f.png")
    r = subprocess.run([CLI, "-p", ply, "--width", "160", "--height", "96", "--out", png], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (_read_png(png) == got).all()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sphere", "obj"])
def test_cli_mirror_mesh_matches_oracle(tmp_path, kind):
    ply, acts = _scene(tmp_path)
    out = str(tmp_path / "f.ppm")
    center = grt.gaussian_center(acts["pos"])
    eye = np.float32([0, 0, 3])
    pos = (center * np.float32(0.25) + eye * np.float32(0.75)).astype(np.float32)  # src/GaussianTracer.cpp:580-588
    args = [CLI, "-p", ply, "--width", "128", "--height", "96", "--out", out, "--type", "mirror", "--bounces", "3"]
    if kind == "sphere":
        v, n, f = grt.sphere_mesh(pos)  # numpy sin/cos may differ from sinf in the last bit: geometry differs by ulps
        args += ["--sphere"]
        tol_frac = 2e-3
    else:
        v0, n0, f = grt.sphere_mesh((0, 0, 0), tess_u=40, tess_v=20)
        obj = str(tmp_path / "m.obj")
        with open(obj, "w") as fh:  # the loader flips Y of positions and normals (Primitives.cpp:175,179)
            for a in v0:
                fh.write(f"v {float(a[0])!r} {float(-a[1])!r} {float(a[2])!r}\n")
            for a in n0:
                fh.write(f"vn {float(a[0])!r} {float(-a[1])!r} {float(a[2])!r}\n")
            for t in f:
                fh.write("f " + " ".join(f"{i + 1}//{i + 1}" for i in t) + "\n")
        args += ["--obj", obj]
        # the loader emits one vertex per face corner, in file order
        v = (v0[f.reshape(-1)] + pos[None]).astype(np.float32); n = n0[f.reshape(-1)]
        f = np.arange(len(v), dtype=np.uint32).reshape(-1, 3)
        tol_frac = 0.0
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    p = grt.default_params(128, 96, center, mesh_type=grt.MIRROR, max_bounces=3)
    sc = O.Scene(acts_to_particles(acts))
    sc.set_mesh(v, n, f)
    ref, ref_f32, cnt = sc.render(to_oracle_params(p))
    got = _read_ppm(out)
    bad = (~u8_matches(got, ref, ref_f32)).any(-1).mean()  # (the numpy sphere's vertices differ from sinf's by ulps)
    assert bad <= tol_frac, bad
    assert cnt["segments"] > cnt["rays"]


@pytest.mark.gpu
def test_cli_viewer_glue_row_zero_is_the_bottom_and_drag_refits(tmp_path):
    """Viewer glue: render(CUDAOutputBuffer&) fills the pinned host mirror, whose row 0 is the
    renderer's row 0 and the BOTTOM row of the window image (src/Display.cpp:13,184); a gizmo drag through
    updateInstanceTransforms (mesh LBVH re-fit) gives the oracle's frame of the moved sphere."""
    ply, acts = _scene(tmp_path)
    W, H = 144, 80
    center = grt.gaussian_center(acts["pos"])
    pos = (center * np.float32(0.25) + np.float32([0, 0, 3]) * np.float32(0.75)).astype(np.float32)
    move = np.float32([0.2, 0.15, -0.1])
    out, raw = str(tmp_path / "f.ppm"), str(tmp_path / "f.rgb")
    r = subprocess.run([CLI, "-p", ply, "--width", str(W), "--height", str(H), "--out", out, "--raw", raw, "--type", "mirror",
                        "--bounces", "3", "--sphere", "--move", *(repr(float(x)) for x in move)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    buf = np.fromfile(raw, np.uint8).reshape(H, W, 3)       # the pinned mirror, row 0 first
    with open(out, "rb") as f:                                # the window image, top row first
        assert f.readline().strip() == b"P6"; f.readline(); f.readline()
        img = np.frombuffer(f.read(), np.uint8).reshape(H, W, 3)
    assert (img[H - 1] == buf[0]).all() and (img[0] == buf[H - 1]).all() and (img[::-1] == buf).all()
    v, n, fcs = grt.primitive_mesh(grt.PRIM_SPHERE)
    vv = ((v + pos[None]).astype(np.float32) + move[None]).astype(np.float32)  # translate(position) then the drag
    p = grt.default_params(W, H, center, mesh_type=grt.MIRROR, max_bounces=3)
    sc = O.Scene(acts_to_particles(acts))
    sc.set_mesh(vv, n, fcs)
    ref, ref_f32, cnt = sc.render(to_oracle_params(p))
    assert cnt["segments"] > cnt["rays"]
    bad = (~u8_matches(buf, ref, ref_f32)).any(-1).mean()
    assert bad <= 2e-3, bad  # float association of (v + pos) + move may differ by an ulp from the facade's matrix product
    top, bottom = buf[H - 8:].astype(int).sum(), buf[:8].astype(int).sum()
    assert top != bottom  # the frame is not symmetric: the orientation check above has teeth


@pytest.mark.gpu
def test_cli_gpus_n_tile_sharded_frame_equals_the_one_gpu_frame(tmp_path):
    """--gpus N: one GaussianTracer and one host thread per rank, the frame's 32x32 tiles dealt
    round-robin, peer copies of the compact tile buffers to rank 0, un-permute there.  On a one-GPU box the ranks share
    device 0 (--devices 0,0,0): the assembled frame must be the one-launch frame byte for byte (ragged frame, 3 ranks,
    mirror sphere)."""
    ply, acts = _scene(tmp_path)
    one, three = str(tmp_path / "one.npy"), str(tmp_path / "three.npy")
    common = [CLI, "-p", ply, "--width", "200", "--height", "136", "--type", "mirror", "--sphere", "--bounces", "3"]
    r = subprocess.run(common + ["--out", one], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(common + ["--out", three, "--gpus", "3", "--devices", "0,0,0", "--bench", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "gpus 3" in r.stdout
    a, b = np.load(one), np.load(three)
    assert a.shape == (136, 200, 3) and a.any() and (a == b).all()


@pytest.mark.gpu
def test_cli_rccl_gather_at_one_rank_equals_the_one_launch_frame(tmp_path):
    """--gather rccl (ncclGather, rccl.h:745, or grouped ncclSend / ncclRecv): the C++ host
    path of the N-rank frame — ncclCommInitAll, every rank's thread posting its send, rank 0 the receives into the slices of
    ONE [N][count][32][32][3] buffer, un-permute — executed on the hardware there is: one rank, so the frame goes
    render_tiles -> RCCL send / recv to itself -> assemble and must equal the one-launch frame byte for byte.  Asking for
    RCCL with ranks that share a device is refused (RCCL allows one rank per GPU: the peer-copy fallback is for that)."""
    ply, acts = _scene(tmp_path)
    one, coll = str(tmp_path / "one.npy"), str(tmp_path / "coll.npy")
    common = [CLI, "-p", ply, "--width", "200", "--height", "136", "--type", "mirror", "--sphere", "--bounces", "3"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(common + ["--out", one], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(common + ["--out", coll, "--gpus", "1", "--gather", "rccl", "--bench", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "gather rccl" in r.stdout
    a, b = np.load(one), np.load(coll)
    assert a.shape == (136, 200, 3) and a.any() and (a == b).all()
    r = subprocess.run(common + ["--gpus", "2", "--devices", "0,0", "--gather", "rccl"], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "distinct devices" in r.stderr
