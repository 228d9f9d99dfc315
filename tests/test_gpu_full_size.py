"""Parity at BASELINE.json's full sizes (C2: 100 k at 1280x720; C3: 1 M Gaussians at 1920x1080; C4: C3 + the mirror sphere,
<= 2 bounces; C5: 3 M at 3840x2160 fisheye):
  * THE WHOLE FRAME AGAINST THE ORACLE, every pixel (round 6; rounds 1-5 sampled 0.2 % of it): radiance within 1e-4 per channel,
    the 8-bit frame equal to the oracle's except within that tolerance of a quantisation step (compare()); every pixel runs
    shaders/tracer.cu:17-110 -> tracer.cuh:484-496.  The oracle builds its own midpoint-split BVH and shares no code with the
    device, so a common-mode error of the four kernels (which share grt_device.h) at scale — index width, tree height, pool
    exhaustion meeting compositing — shows here and nowhere else;
  * kernel independence: the tile kernel (default), the streaming kernel, the round-based wave kernel and (on a window)
    the per-lane kernel are four separate implementations of the traversal and must agree bit for bit;
  * shard independence: the 8-rank tile split, un-permuted, is the full frame;
  * schedule independence: frames launched in cost order (scheduling feedback, big-window split) are unchanged.
"""
import numpy as np
import pytest
import torch

import grt
import tiles
from common import make_scene, threshold_flip_explains, usable_cores
from test_gpu_parity import compare

pytestmark = pytest.mark.gpu


def whole_frame_against_the_oracle(sc, op, f32, u8, label, **tolerances):
    """The oracle renders EVERY pixel of the frame (threads = the cores this process may use) and compare() holds on all of them:
    radiance within 1e-4, 8-bit values equal except within 1e-4 of a quantisation step.  Pinhole frames: a pixel beyond the
    tolerance must be a ray that sits ON one of the reference's two hard thresholds (common.threshold_flip_explains: the oracle
    reproduces the GPU's value once minTransmittance or alpha_min moves by a relative 1e-6 .. 1e-4), there may be at most 1e-5 of
    the frame of them, each within 0.02 (round 6, first whole C3 frame: ONE pixel of 2 073 600, T within 1e-6 of minTransmittance,
    3.9e-4; every other pixel within 3e-7).  Returns the oracle's counters and the largest radiance difference off those pixels."""
    import time
    t0 = time.perf_counter()
    ref_u8, ref_f32, rc = sc.render(op, threads=usable_cores())
    dt = time.perf_counter() - t0
    g = f32.cpu().numpy()
    d = np.abs(g - ref_f32)
    over = (d > 1e-4).any(-1)
    flips = []
    if not tolerances:
        ys, xs = np.nonzero(over)
        assert len(ys) <= max(2, int(1e-5 * over.size)), f"{label}: {len(ys)} pixels beyond 1e-4"
        for y, x in zip(ys, xs):
            why = threshold_flip_explains(sc, op, int(x), int(y), g[y, x])
            assert why is not None, f"{label}: pixel ({x}, {y}) differs by {d[y, x].max():.3e} and no threshold explains it"
            flips.append((int(x), int(y), float(d[y, x].max()), why))
        tolerances = dict(max_outlier_frac=len(ys) / over.size, max_outlier=0.02)
    compare(g, ref_f32, u8, ref_u8, **tolerances)
    dmax = float(d[~over].max())
    n8 = int((u8.cpu().numpy() != ref_u8).sum())
    print(f"{label}: whole frame {op.width}x{op.height} = {op.width * op.height} pixels against the oracle ({dt:.1f} s on "
          f"{usable_cores()} threads): max |radiance diff| {dmax:.3e} on {int((~over).sum())} pixels, {int(over.sum())} beyond 1e-4 "
          f"{flips if flips else ''}, {n8} of {ref_u8.size} 8-bit values differ (each within 1e-4 of a quantisation step, or on those pixels)")
    return rc, dmax


def test_c2_full_size_whole_frame_against_the_oracle():
    """BASELINE config C2: 100 k Gaussians (seed 2), 1280x720 pinhole — all 921 600 pixels against the oracle, and the
    four kernels bit for bit (the quad kernel is not used at this size; C1's frame in test_gpu_parity runs through it)."""
    W, H = 1280, 720
    acts, p, sc, op, _ = make_scene(2, 100_000, W, H)
    tr = grt.Tracer(0)
    tr.upload(acts)
    tr.set_option(grt.OPT_COUNTERS, 1)
    u8, f32 = tr.render(p, want_f32=True)
    cnt = tr.counters()
    tr.set_option(grt.OPT_COUNTERS, 0)
    u8, f32 = u8.clone(), f32.clone()
    for _ in range(3):  # steady state: cost-ordered launches
        a8, af = tr.render(p, want_f32=True)
    assert (a8 == u8).all() and (af == f32).all()
    for kernel in (1, 2, 3):
        tr.set_option(grt.OPT_KERNEL, kernel)
        a8, af = tr.render(p, want_f32=True)
        assert (a8 == u8).all() and (af == f32).all(), kernel
    rc, _ = whole_frame_against_the_oracle(sc, op, f32, u8, "C2")
    assert rc["rays"] == W * H and cnt["stall_exits"] == 0 and abs(cnt["hit_evals"] - rc["hit_evals"]) <= 1e-4 * rc["hit_evals"]
    tr.close()
    sc.close()


def test_c3_full_size_whole_frame_against_the_oracle_and_properties():
    W, H = 1920, 1080
    acts, p, sc, op, _ = make_scene(3, 1_000_000, W, H)
    tr = grt.Tracer(0)
    tr.upload(acts)
    u8, f32 = tr.render(p, want_f32=True)
    u8, f32 = u8.clone(), f32.clone()
    assert int(u8.sum().item()) > 0
    # ---- kernel independence ----
    for kernel in (2, 3):
        tr.set_option(grt.OPT_KERNEL, kernel)
        a8, af = tr.render(p, want_f32=True)
        assert (a8 == u8).all() and (af == f32).all(), kernel
    tr.set_option(grt.OPT_KERNEL, 1)
    win = (640, 536, 832, 664)  # around the heaviest tiles of this frame
    w8 = torch.zeros_like(u8); wf = torch.zeros_like(f32)
    tr.render(p, window=win, out_u8=w8, out_f32=wf)
    x0, y0, x1, y1 = win
    assert (w8[y0:y1, x0:x1] == u8[y0:y1, x0:x1]).all() and (wf[y0:y1, x0:x1] == f32[y0:y1, x0:x1]).all()
    tr.set_option(grt.OPT_KERNEL, 0)
    # ---- schedule independence: steady-state frames (heaviest-first), forced big-window split ----
    for kernel, fb in ((0, 1), (3, 1), (3, 5)):
        tr.set_option(grt.OPT_KERNEL, kernel)
        tr.set_option(grt.OPT_FEEDBACK, fb)
        for _ in range(3):
            a8, af = tr.render(p, want_f32=True)
        assert (a8 == u8).all() and (af == f32).all(), (kernel, fb)
    tr.set_option(grt.OPT_KERNEL, 0)
    tr.set_option(grt.OPT_FEEDBACK, 1)
    # ---- shard independence: 8 ranks ----
    world = 8
    tx, ty = tiles.grid(W, H, 32)
    gathered = []
    for rank in range(world):
        _, _, cnt, max_cnt = tiles.my_tiles(tx * ty, world, rank)
        buf = torch.zeros((max_cnt, 32, 32, 3), dtype=torch.uint8, device="cuda:0")
        for _ in range(2):  # second frame runs in cost order
            tr.render_tiles(p, 32, 32, rank, world, cnt, out_u8=buf)
        gathered.append(buf)
    assert (tiles.assemble(gathered, W, H, 32) == u8).all()
    # ---- the oracle on the WHOLE frame (2 073 600 pixels) ----
    rc, _ = whole_frame_against_the_oracle(sc, op, f32, u8, "C3")
    tr.set_option(grt.OPT_COUNTERS, 1)
    tr.render(p)
    cnt = tr.counters()
    tr.set_option(grt.OPT_COUNTERS, 0)
    # (consumed hits may differ where expf's last bit moves a ray across T = minTransmittance: test_c1's allowance)
    assert rc["rays"] == W * H and rc["hit_evals"] > 30 * W * H and cnt["stall_exits"] == 0
    assert abs(cnt["hit_evals"] - rc["hit_evals"]) <= 1e-4 * rc["hit_evals"]
    tr.close()
    sc.close()


def test_c5_full_size_fisheye_kernel_and_shard_independence():
    W, H = 3840, 2160
    acts, p, sc, op, _ = make_scene(5, 3_000_000, W, H, fisheye=True)
    tr = grt.Tracer(0)
    tr.upload(acts)
    u8, f32 = tr.render(p, want_f32=True)
    u8, f32 = u8.clone(), f32.clone()
    assert (u8[:8, :8] == 0).all()  # fisheye: r > 1 is black
    # ---- the oracle on the WHOLE frame (8 294 400 pixels, 6.5 M of them inside the image circle).  Device and host rays differ
    #      in the last bits of sinf / cosf / asinf / atan2f here, and only here: a near-tie between two events may flip on a
    #      few pixels (test_fisheye's allowance: at most 2e-4 of the pixels beyond 1e-4, each bounded by 0.08; measured 7.2e-5: 597 pixels) ----
    rc, _ = whole_frame_against_the_oracle(sc, op, f32, u8, "C5", max_outlier_frac=2e-4, max_outlier=0.08)
    assert 0.75 * W * H < rc["rays"] < 0.80 * W * H and rc["hit_evals"] > 10 * rc["rays"]  # pi / 4 of the frame spawns rays
    rim = f32[1836:1852, 3272:3288].cpu().numpy()
    assert (rim[-1, -1] == 0).all()  # this window straddles r = 1 (corner radii 0.993 / 1.009): outside is black
    sc.close()
    for kernel in (2, 3):
        tr.set_option(grt.OPT_KERNEL, kernel)
        a8, _ = tr.render(p)
        assert (a8 == u8).all(), kernel
    tr.set_option(grt.OPT_KERNEL, 0)
    world = 8
    tx, ty = tiles.grid(W, H, 32)
    gathered = []
    for rank in range(world):
        _, _, cnt, max_cnt = tiles.my_tiles(tx * ty, world, rank)
        buf = torch.zeros((max_cnt, 32, 32, 3), dtype=torch.uint8, device="cuda:0")
        tr.render_tiles(p, 32, 32, rank, world, cnt, out_u8=buf)
        gathered.append(buf)
    assert (tiles.assemble(gathered, W, H, 32) == u8).all()
    tr.close()


def test_c4_full_size_mirror_sphere_through_the_obj_path(tmp_path):
    """BASELINE config C4: the 1 M-Gaussian scene (seed 3) at 1920x1080 with the reference's procedural sphere
    (src/geometry/Primitives.cpp:63-140) written once as OBJ with normals and loaded through the OBJ path (Y flip,
    Primitives.cpp:175,179; un-indexed soup), placed at 0.25 lookat + 0.75 eye (src/GaussianTracer.cpp:630-638),
    MIRROR, bounce cap 2.  The default pipeline (tile kernel + wavefront bounces), the streaming-kernel pipeline and,
    on windows with mirror pixels, the per-lane megakernel agree bit for bit; the oracle renders the whole frame."""
    W, H = 1920, 1080
    acts, p, sc, op, center = make_scene(3, 1_000_000, W, H, mesh_type=grt.MIRROR, max_bounces=2)
    v, n, f = grt.primitive_mesh(grt.PRIM_SPHERE)
    flip = np.float32([1, -1, 1])
    path = str(tmp_path / "sphere.obj")
    grt.write_obj(path, v * flip, n * flip, f)  # the loader flips Y back
    pos = (0.25 * center + 0.75 * np.float32([0, 0, 3])).astype(np.float32)
    mv, mn, mf = grt.load_obj(path, center=pos)
    assert len(mv) == 3 * len(f) and (mv == (v[f.reshape(-1)] + pos[None]).astype(np.float32)).all() and (mn == n[f.reshape(-1)]).all()
    tr = grt.Tracer(0)
    tr.upload(acts)
    tr.set_meshes([(mv, mn, mf)])
    sc.set_mesh(mv, mn, mf)
    tr.set_option(grt.OPT_COUNTERS, 1)
    u8, f32 = tr.render(p, want_f32=True)
    cnt = tr.counters()
    tr.set_option(grt.OPT_COUNTERS, 0)
    u8, f32 = u8.clone(), f32.clone()
    assert cnt["segments"] > 1.2 * cnt["rays"] and cnt["stall_exits"] == 0  # secondary segments exist
    for _ in range(2):  # steady state (cost-ordered launch) writes the same frame
        a8, af = tr.render(p, want_f32=True)
        assert (a8 == u8).all() and (af == f32).all()
    tr.set_option(grt.OPT_KERNEL, 3)
    a8, af = tr.render(p, want_f32=True)
    assert (a8 == u8).all() and (af == f32).all()
    tr.set_option(grt.OPT_KERNEL, 1)
    for win in ((896, 476, 1024, 604), (560, 440, 688, 568)):  # sphere centre; its left rim
        x0, y0, x1, y1 = win
        w8 = torch.zeros_like(u8); wf = torch.zeros_like(f32)
        tr.render(p, window=win, out_u8=w8, out_f32=wf)
        assert (w8[y0:y1, x0:x1] == u8[y0:y1, x0:x1]).all() and (wf[y0:y1, x0:x1] == f32[y0:y1, x0:x1]).all(), win
    tr.set_option(grt.OPT_KERNEL, 0)
    # ---- the oracle on the WHOLE frame: every pixel through mesh closest hit, mirror bounce and both Gaussian segments ----
    rc, _ = whole_frame_against_the_oracle(sc, op, f32, u8, "C4")
    assert rc["segments"] - rc["rays"] > 400_000 and rc["segments"] == cnt["segments"]  # the sphere's pixels carry a second segment
    tr.close()
    sc.close()


def test_c4_scale_glass_sphere_deep_bounces_pipeline_independence():
    """The 1 M scene at 1920x1080 with a GLASS sphere and up to 6 bounces (rays refract into the sphere, reflect inside,
    leave again): bundle rounds, chunks over budget, lone rays on the packed queue and the per-lane finish all carry
    rays here.  The default pipeline, the pipeline with 4 bundle rounds and a small budget, the streaming-kernel
    pipeline (per-lane bounces only) and, on a window, the per-lane megakernel agree bit for bit."""
    W, H = 1920, 1080
    acts, p, sc, op, center = make_scene(3, 1_000_000, W, H, mesh_type=grt.GLASS, max_bounces=6)
    sc.close()
    pos = (0.25 * center + 0.75 * np.float32([0, 0, 3])).astype(np.float32)
    v, n, f = grt.sphere_mesh(pos, tess_u=64, tess_v=32)
    tr = grt.Tracer(0)
    tr.upload(acts)
    tr.set_meshes([(v, n, f)])
    tr.set_option(grt.OPT_COUNTERS, 1)
    u8, f32 = tr.render(p, want_f32=True)
    cnt = tr.counters()
    u8, f32 = u8.clone(), f32.clone()
    assert cnt["segments"] > 1.5 * cnt["rays"] and cnt["stall_exits"] == 0  # several segments per sphere pixel
    tr.set_option(grt.OPT_BUNDLE_ROUNDS, 4)
    tr.set_option(grt.OPT_BUNDLE_BUDGET, 96)
    a8, af = tr.render(p, want_f32=True)
    c2 = tr.counters()
    assert (a8 == u8).all() and (af == f32).all()
    assert c2["segments"] == cnt["segments"] and c2["hit_evals"] == cnt["hit_evals"] and c2["stall_exits"] == 0
    tr.set_option(grt.OPT_BUNDLE_ROUNDS, 2)
    tr.set_option(grt.OPT_BUNDLE_BUDGET, 1024)
    tr.set_option(grt.OPT_KERNEL, 3)
    a8, af = tr.render(p, want_f32=True)
    assert (a8 == u8).all() and (af == f32).all()
    tr.set_option(grt.OPT_KERNEL, 1)
    x0, y0, x1, y1 = 880, 460, 1040, 620  # the sphere's centre
    w8 = torch.zeros_like(u8); wf = torch.zeros_like(f32)
    tr.render(p, window=(x0, y0, x1, y1), out_u8=w8, out_f32=wf)
    assert (w8[y0:y1, x0:x1] == u8[y0:y1, x0:x1]).all() and (wf[y0:y1, x0:x1] == f32[y0:y1, x0:x1]).all()
    tr.close()


def test_anisotropic_scene_crowded_frontier_kernel_independence():
    """Needle / pancake Gaussians at scale (per-axis log-scale noise sigma 1.2 on 300 k Gaussians): proxies that span a
    large part of the scene overlap by the hundred, the tile kernel's frontier spills to its LDS bag and is rebalanced
    over and over, windows overflow into the per-lane bags, full bags are pruned, some lanes take another pass.  The tile kernel (default), the
    streaming kernel and the round-based kernel must still agree bit for bit, with and without the size classes of the
    LBVH, and the oracle agrees on sampled windows."""
    W, H = 960, 540
    raw = grt.synth_scene(3, 300_000)
    rng = np.random.default_rng(1003)
    raw["scale"] = (raw["scale"] + rng.normal(0.0, 1.2, size=raw["scale"].shape)).astype(np.float32)
    acts = grt.activate(raw)
    center = grt.gaussian_center(acts["pos"])
    p = grt.default_params(W, H, center)
    frames = {}
    # (the first four builds cut their proxies as rounds 3-5 did — pieces 8 quarters of the typical diagonal long, and only where the
    #  cells' boxes hold under half the proxy's box: that tree is what crowds the frontier and overflows windows and bags here; the fifth
    #  is round 6's default — every long proxy cut, the length chosen by the scene — which must render the same bytes with less trouble)
    for kernel, size_classes, old_splits in ((0, 1, True), (3, 1, True), (2, 1, True), (0, 0, True), (0, 2, False)):
        tr = grt.Tracer(0)
        tr.set_option(grt.OPT_SIZE_CLASSES, 1 if size_classes else 0)
        tr.set_option(grt.OPT_KERNEL, kernel)
        if old_splits:
            tr.set_option(grt.OPT_SPLIT, 8)
            tr.set_option(grt.OPT_SPLIT_VOL_PCT, 50)
        tr.upload(acts)
        tr.set_option(grt.OPT_COUNTERS, 1)
        u8, f32 = tr.render(p, want_f32=True)
        cnt = tr.counters()
        frames[(kernel, size_classes)] = (u8.clone(), f32.clone(), cnt)
        assert cnt["stall_exits"] == 0, (kernel, size_classes)
        tr.set_option(grt.OPT_SIZE_CLASSES, 1)
        tr.close()
    u8, f32, cnt = frames[(0, 1)]
    # windows overflow, full bags are pruned to their nearer half, and some lanes still go again
    assert cnt["rounds"] > cnt["rays"] + 500
    for key, (a8, af, c2) in frames.items():
        assert (a8 == u8).all() and (af == f32).all(), key
        assert c2["hit_evals"] == cnt["hit_evals"], key
    from common import acts_to_particles, to_oracle_params
    import oracle as O
    sc = O.Scene(acts_to_particles(acts))
    for (x0, y0, x1, y1) in [(472, 262, 488, 278), (100, 400, 116, 416), (800, 100, 816, 116)]:
        ref_u8, ref_f32, _ = sc.render(to_oracle_params(p), window=(x0, y0, x1, y1), threads=8)
        compare(f32[y0:y1, x0:x1], ref_f32[y0:y1, x0:x1], u8[y0:y1, x0:x1], ref_u8[y0:y1, x0:x1])
    sc.close()
