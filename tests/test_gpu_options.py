"""Every tuning option of the library is pure scheduling / data layout: frames must not change by a single byte."""
import numpy as np
import pytest
import torch

import grt
import tiles
from common import make_scene

pytestmark = pytest.mark.gpu


def test_options_do_not_change_pixels():
    acts, p, sc, op, _ = make_scene(17, 30000, 256, 160, scale_boost=0.4)
    tr = grt.Tracer(0)
    tr.upload(acts)
    ref8, reff = tr.render(p, want_f32=True)
    ref8, reff = ref8.clone(), reff.clone()
    for leaf_max in (1, 2, 8):
        tr.set_option(grt.OPT_LEAF_MAX, leaf_max)
        tr.upload(acts)
        for kernel in (0, 1, 2, 3):
            tr.set_option(grt.OPT_KERNEL, kernel)
            a8, af = tr.render(p, want_f32=True)
            assert (a8 == ref8).all() and (af == reff).all(), (leaf_max, kernel)
    tr.set_option(grt.OPT_LEAF_MAX, 4); tr.set_option(grt.OPT_KERNEL, 0); tr.upload(acts)
    for swz in (0, 1, 8, 100000):
        tr.set_option(grt.OPT_SWIZZLE, swz)
        a8, af = tr.render(p, want_f32=True)
        assert (a8 == ref8).all() and (af == reff).all(), swz
    for kernel in (0, 3):
        tr.set_option(grt.OPT_KERNEL, kernel)
        for fb in (0, 1, 3, 5):  # 5 = heaviest-first + (streaming kernel) big-window kernel for the heaviest tiles
            tr.set_option(grt.OPT_FEEDBACK, fb)
            for _ in range(3):
                a8, af = tr.render(p, want_f32=True)
                assert (a8 == ref8).all() and (af == reff).all(), (kernel, fb)
    tr.set_option(grt.OPT_KERNEL, 0); tr.set_option(grt.OPT_FEEDBACK, 1)
    # tile kernel tuning: batch sizes, compositing deferral, frontier reserve, priority
    for opt, vals in ((grt.OPT_TILE_READY_MIN, (1, 64)), (grt.OPT_TILE_BAND, (0, 1024)), (grt.OPT_TILE_LOOKAHEAD, (0, 1024)),
                      (grt.OPT_TILE_RESERVE, (0, 60)), (grt.OPT_TILE_PRIO_DIV, (4,))):
        for v in vals:
            tr.set_option(opt, v)
            for _ in range(2):
                a8, af = tr.render(p, want_f32=True)
                assert (a8 == ref8).all() and (af == reff).all(), (opt, v)
        tr.set_option(opt, {grt.OPT_TILE_READY_MIN: 16, grt.OPT_TILE_BAND: 64, grt.OPT_TILE_LOOKAHEAD: 64,
                            grt.OPT_TILE_RESERVE: 24, grt.OPT_TILE_PRIO_DIV: 0}[opt])
    # a tree with pieces (needles and sheets): the absolute band floor of its kernel is scheduling only
    raw = grt.synth_scene(19, 20000)
    raw["scale"] = (raw["scale"] + np.random.default_rng(3).normal(0.0, 1.6, size=raw["scale"].shape)).astype(np.float32)
    acts2 = grt.activate(raw)
    tr.upload(acts2)
    assert tr.bvh_info()["n_primitives"] > tr.bvh_info()["n_proxies"]
    p2 = grt.default_params(256, 160, grt.gaussian_center(acts2["pos"]))
    b8, bf = tr.render(p2, want_f32=True)
    b8, bf = b8.clone(), bf.clone()
    for v in (0, 64, 100000):
        tr.set_option(grt.OPT_TILE_BAND_ABS, v)
        a8, af = tr.render(p2, want_f32=True)
        assert (a8 == b8).all() and (af == bf).all(), v
    tr.set_option(grt.OPT_TILE_BAND_ABS, 512)
    tr.upload(acts)
    with pytest.raises(grt.GrtError):
        tr.set_option(grt.OPT_LEAF_MAX, 9)
    with pytest.raises(grt.GrtError):
        tr.set_option(99, 1)
    tr.close()


def test_heavy_tiles_as_part_waves_give_the_same_frame():
    """GRT_OPT_TILE_PARTS4_PCT / _PARTS2_PCT / _PARTS_LOAD_PCT: an 8x8 tile that was heavy in the previous frame runs as four waves
    of 4x4 pixels or two of 4x8.  Pure scheduling: full frames, windows and rank tile lists must not change by a byte, whatever
    the thresholds (so low here that nearly every tile is split; four-way only; two-way only; a band of each; more parts than
    the launch has room for: the heaviest classes get them), frame after frame (a split tile's cost word is scaled back so that it
    stays split), and the per-ray counters (rays, segments, consumed hits) must be those of whole tiles."""
    import torch
    acts, p, sc, op, _ = make_scene(31, 30000, 200, 136, scale_boost=0.45)
    tr = grt.Tracer(0)
    tr.upload(acts)
    tr.set_option(grt.OPT_TILE_PARTS4_PCT, 0)  # the frame of whole tiles
    ref8, reff = tr.render(p, want_f32=True)
    ref8, reff = ref8.clone(), reff.clone()
    tr.set_option(grt.OPT_COUNTERS, 1)
    tr.render(p)
    c0 = tr.counters()
    tr.set_option(grt.OPT_COUNTERS, 0)
    w8 = torch.zeros_like(ref8)
    # (the four-way parts on the quad kernel — GRT_OPT_QUAD_PARTS = 2: whatever the launch's size; by default only launches of 2048 to 12 288
    #  tiles use it — and as part waves of the camera-ray kernel; 64: the quad kernel takes 64 parts at most, the rest stay part waves)
    for quad, v2, v4, vl in ((2, 1, 0, 0), (2, 1, 1, 0), (2, 0, 1, 0), (64, 0, 1, 0), (0, 0, 1, 0), (2, 10, 40, 0), (0, 10, 40, 0), (2, 20, 60, 50), (2, 0, 70, 75),
                             (0, 0, 70, 75), (2, 5, 5, 100), (1, 5, 5, 100)):
        tr.set_option(grt.OPT_QUAD_PARTS, quad)
        tr.set_option(grt.OPT_TILE_PARTS2_PCT, v2)
        tr.set_option(grt.OPT_TILE_PARTS4_PCT, v4)
        tr.set_option(grt.OPT_TILE_PARTS_LOAD_PCT, vl)
        for it in range(4):
            a8, af = tr.render(p, want_f32=True)
            assert (a8 == ref8).all() and (af == reff).all(), (v2, v4, it)
        tr.set_option(grt.OPT_COUNTERS, 1)
        tr.render(p)
        c1 = tr.counters()
        tr.set_option(grt.OPT_COUNTERS, 0)
        for k in ("rays", "segments", "hit_evals"):  # (passes may differ: which lane overflows its window depends on the wave)
            assert c1[k] == c0[k], (k, v2, v4)
        assert c1["stall_exits"] == 0
        for it in range(3):  # a window (its own launch geometry, own costs)
            w8.zero_()
            tr.render(p, window=(24, 16, 170, 120), out_u8=w8)
            assert (w8[16:120, 24:170] == ref8[16:120, 24:170]).all(), (v2, v4, it)
        # a rank's tile list: 32x32 tiles 1, 3, 5, ...
        tx, ty = (200 + 31) // 32, (136 + 31) // 32
        cnt = (tx * ty) // 2
        buf = torch.zeros((cnt, 32, 32, 3), dtype=torch.uint8, device="cuda:0")
        for it in range(3):
            tr.render_tiles(p, 32, 32, 1, 2, cnt, out_u8=buf)
            tr.check()
            for j in range(cnt):
                t = 1 + 2 * j
                x0, y0 = (t % tx) * 32, (t // tx) * 32
                h, w = min(32, 136 - y0), min(32, 200 - x0)
                assert (buf[j, :h, :w] == ref8[y0:y0 + h, x0:x0 + w]).all(), (v2, v4, it, j)
    tr.check()
    tr.close(); sc.close()


def test_part_waves_on_a_mesh_frame_give_the_same_frame():
    """GRT_OPT_MESH_PARTS (default on): the heavy tiles of a MESH frame's primary stage run as part waves too — each part queues its
    own chunk of continuation rays (<= 16), so the queues hold one chunk per launch-order entry.  Pure scheduling again: mirror and
    glass sphere, full frame / window / rank tile list, thresholds that split nearly every tile, frame after frame: byte-identical
    to the frame without part waves, per-ray counters equal, no give-up."""
    import torch
    for mesh_type, bounces in ((grt.MIRROR, 3), (grt.GLASS, 4)):
        acts, p, sc, op, center = make_scene(23, 30000, 200, 136, scale_boost=0.45, mesh_type=mesh_type, max_bounces=bounces)
        pos = (0.25 * center + 0.75 * np.float32([0, 0, 3])).astype(np.float32)
        tr = grt.Tracer(0)
        tr.upload(acts)
        tr.set_meshes([grt.sphere_mesh(pos, tess_u=32, tess_v=16)])
        tr.set_option(grt.OPT_MESH_PARTS, 0)
        ref8, reff = tr.render(p, want_f32=True)
        ref8, reff = ref8.clone(), reff.clone()
        tr.set_option(grt.OPT_COUNTERS, 1)
        tr.render(p)
        c0 = tr.counters()
        tr.set_option(grt.OPT_COUNTERS, 0)
        tr.set_option(grt.OPT_MESH_PARTS, 1)
        w8 = torch.zeros_like(ref8)
        for v2, v4, vl in ((0, 1, 0), (1, 0, 0), (10, 40, 0), (0, 60, 75)):
            tr.set_option(grt.OPT_TILE_PARTS2_PCT, v2)
            tr.set_option(grt.OPT_TILE_PARTS4_PCT, v4)
            tr.set_option(grt.OPT_TILE_PARTS_LOAD_PCT, vl)
            for it in range(4):
                a8, af = tr.render(p, want_f32=True)
                assert (a8 == ref8).all() and (af == reff).all(), (mesh_type, v2, v4, it)
            tr.set_option(grt.OPT_COUNTERS, 1)
            tr.render(p)
            c1 = tr.counters()
            tr.set_option(grt.OPT_COUNTERS, 0)
            for k in ("rays", "segments", "hit_evals"):
                assert c1[k] == c0[k], (k, mesh_type, v2, v4)
            assert c1["stall_exits"] == 0
            for it in range(3):
                w8.zero_()
                tr.render(p, window=(24, 16, 170, 120), out_u8=w8)
                assert (w8[16:120, 24:170] == ref8[16:120, 24:170]).all(), (mesh_type, v2, v4, it)
            tx, ty = (200 + 31) // 32, (136 + 31) // 32
            cnt = (tx * ty) // 2
            buf = torch.zeros((cnt, 32, 32, 3), dtype=torch.uint8, device="cuda:0")
            for it in range(3):
                tr.render_tiles(p, 32, 32, 1, 2, cnt, out_u8=buf)
                tr.check()
                for j in range(cnt):
                    t = 1 + 2 * j
                    x0, y0 = (t % tx) * 32, (t // tx) * 32
                    h, w = min(32, 136 - y0), min(32, 200 - x0)
                    assert (buf[j, :h, :w] == ref8[y0:y0 + h, x0:x0 + w]).all(), (mesh_type, v2, v4, it, j)
        tr.check()
        tr.close(); sc.close()


def test_launch_order_by_several_workgroups_gives_the_same_frame():
    """GRT_OPT_ORDER_MULTI_MIN: from 16384 tiles on the launch order is made by several workgroups in four kernels (grt_bvh.hip: k_ord_a-d)
    instead of one workgroup.  Forced here on a 4000-tile frame (two workgroups) and a 425-tile one (one): with every part-wave policy —
    including more parts than the launch has room for, the one-thread room rule — frames byte-identical frame after frame, per-ray
    counters those of whole tiles, no give-up; and a moving camera (costs of another view order the frame) likewise."""
    import torch
    for (w, h, n) in ((640, 400, 60000), (200, 136, 30000)):
        acts, p, sc, op, center = make_scene(37, n, w, h, scale_boost=0.45)
        tr = grt.Tracer(0)
        tr.upload(acts)
        ref8, reff = tr.render(p, want_f32=True)
        ref8, reff = ref8.clone(), reff.clone()
        tr.set_option(grt.OPT_COUNTERS, 1)
        tr.render(p)
        c0 = tr.counters()
        tr.set_option(grt.OPT_COUNTERS, 0)
        tr.set_option(grt.OPT_ORDER_MULTI_MIN, 1)
        for v2, v4, vl in ((0, 60, 75), (1, 1, 0), (0, 1, 0), (10, 40, 0), (5, 5, 100)):
            tr.set_option(grt.OPT_TILE_PARTS2_PCT, v2)
            tr.set_option(grt.OPT_TILE_PARTS4_PCT, v4)
            tr.set_option(grt.OPT_TILE_PARTS_LOAD_PCT, vl)
            for it in range(4):
                a8, af = tr.render(p, want_f32=True)
                assert (a8 == ref8).all() and (af == reff).all(), (w, v2, v4, it)
            tr.set_option(grt.OPT_COUNTERS, 1)
            tr.render(p)
            c1 = tr.counters()
            tr.set_option(grt.OPT_COUNTERS, 0)
            for k in ("rays", "segments", "hit_evals"):
                assert c1[k] == c0[k], (k, w, v2, v4)
            assert c1["stall_exits"] == 0
        # a moving camera: every frame is ordered by the costs of the view before it (dilated), the two ways of making the order must
        # both give the frame a fresh context gives
        e = np.float32([0.0, 0.0, 3.0]) - center
        for it in range(4):
            ang = 0.03 * (it + 1)
            eye = (float(center[0] + e[0] * np.cos(ang) + e[2] * np.sin(ang)), 0.0, float(center[2] - e[0] * np.sin(ang) + e[2] * np.cos(ang)))
            q = grt.default_params(w, h, center, eye=eye)
            a8, _ = tr.render(q)
            t2 = grt.Tracer(0)
            t2.upload(acts)
            b8, _ = t2.render(q)
            assert (a8 == b8).all(), (w, it)
            t2.close()
        tr.check()
        tr.close(); sc.close()


def test_cold_frame_with_part_waves_gives_the_same_frame():
    """GRT_OPT_COLD_ESTIMATE = 2 (default): the FIRST frame of a launch geometry, which has no costs of a previous frame, already runs the
    tiles with the largest estimated cost as part waves.  Pure scheduling: the first frame of a fresh context is byte-identical whatever
    the threshold (1 %: nearly every tile split, the room rule decides), also for a window, a tile list and a fisheye frame."""
    import torch
    for fisheye in (False, True):
        acts, p, sc, op, center = make_scene(41, 30000, 200, 136, scale_boost=0.45, fisheye=fisheye)
        frames = []
        for est, pct in ((1, 40), (2, 40), (2, 1), (2, 10), (0, 40)):
            tr = grt.Tracer(0)
            tr.upload(acts)
            tr.set_option(grt.OPT_COLD_ESTIMATE, est)
            tr.set_option(grt.OPT_COLD_PARTS_PCT, pct)
            a8, af = tr.render(p, want_f32=True)           # cold
            w8 = torch.zeros_like(a8)
            tr.render(p, window=(24, 16, 170, 120), out_u8=w8)  # cold again: another launch geometry
            tx, ty = (200 + 31) // 32, (136 + 31) // 32
            cnt = (tx * ty) // 2
            buf = torch.zeros((cnt, 32, 32, 3), dtype=torch.uint8, device="cuda:0")
            tr.render_tiles(p, 32, 32, 1, 2, cnt, out_u8=buf)   # and another
            tr.check()
            frames.append((a8.clone(), af.clone(), w8.clone(), buf.clone()))
            tr.close()
        for f in frames[1:]:
            for x, y in zip(frames[0], f):
                assert (x == y).all(), fisheye
        sc.close()


def test_split_launch_with_mesh_and_tiles():
    """Big-window split launch (GRT_OPT_FEEDBACK = 5) through the wavefront pipeline and the tile entry point."""
    import torch
    acts, p, sc, op, center = make_scene(18, 20000, 192, 128, scale_boost=0.4, mesh_type=grt.MIRROR, max_bounces=3)
    pos = (0.25 * center + 0.75 * np.float32([0, 0, 3])).astype(np.float32)
    tr = grt.Tracer(0)
    tr.upload(acts)
    tr.set_meshes([grt.sphere_mesh(pos, tess_u=32, tess_v=16)])
    tr.set_option(grt.OPT_FEEDBACK, 0)
    ref8, _ = tr.render(p)
    ref8 = ref8.clone()
    for kernel in (0, 3):
        tr.set_option(grt.OPT_KERNEL, kernel)
        tr.set_option(grt.OPT_FEEDBACK, 5)
        for _ in range(3):
            a8, _ = tr.render(p)
            assert (a8 == ref8).all(), kernel
        tiles8 = torch.zeros((6 * 4, 32, 32, 3), dtype=torch.uint8, device="cuda:0")
        for _ in range(3):
            tr.render_tiles(p, 32, 32, 0, 1, 24, out_u8=tiles8)
        img = tiles8.reshape(4, 6, 32, 32, 3).permute(0, 2, 1, 3, 4).reshape(128, 192, 3)
        assert (img == ref8).all(), kernel
    tr.close()


def test_eye_records_follow_the_camera_and_the_scene():
    """The streaming kernel caches per-eye proxy records in the context: a moved camera, a camera moved back, and a
    new scene on the same tracer must each give the oracle's frame (a stale cache would not)."""
    import copy
    from common import to_oracle_params
    from test_gpu_parity import compare as compare_frames
    acts, p, sc, op, center = make_scene(21, 8000, 128, 96, scale_boost=0.4)
    tr = grt.Tracer(0)
    tr.upload(acts)
    cams = []
    for eye in ((0.0, 0.0, 3.0), (1.5, 0.4, 2.2), (0.0, 0.0, 3.0), (-0.7, -1.1, 2.6)):
        q = grt.default_params(128, 96, center, eye=np.float32(eye))
        cams.append(q)
        u8, f32 = tr.render(q, want_f32=True)
        ref_u8, ref_f32, _ = sc.render(to_oracle_params(q))
        compare_frames(f32, ref_f32, u8, ref_u8)
    # same eye, different scene: the records must be rebuilt by the upload
    acts2, p2, sc2, op2, _ = make_scene(22, 6000, 128, 96, scale_boost=0.4)
    tr.upload(acts2)
    q = copy.copy(cams[-1])
    u8, f32 = tr.render(q, want_f32=True)
    ref_u8, ref_f32, _ = sc2.render(to_oracle_params(q))
    compare_frames(f32, ref_f32, u8, ref_u8)
    tr.close()


def test_assemble_tiles_equals_the_torch_unpermute():
    """grt_assemble_tiles (rank 0's un-permute kernel) against tiles.assemble on random bytes: ragged frame, 3 ranks."""
    import tiles
    import torch
    W, H, T, world = 200, 136, 32, 3  # 7 x 5 tiles, ragged on both sides
    tx, ty = tiles.grid(W, H, T)
    _, _, _, max_cnt = tiles.my_tiles(tx * ty, world, 0)
    g = torch.randint(0, 256, (world, max_cnt, T, T, 3), dtype=torch.uint8, device="cuda:0")
    ref = tiles.assemble([g[r] for r in range(world)], W, H, T)
    t = grt.Tracer(0)
    out = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
    t.assemble_tiles(g, world, max_cnt, T, W, H, out)
    t.sync()
    assert bool((out == ref).all())
    with pytest.raises(grt.GrtError):
        t.assemble_tiles(g, world, 1, T, W, H, out)  # world x max_cnt tiles do not cover the frame
    t.close()


def test_quad_parts_every_tile_under_window_overflow_sh_and_fisheye():
    """GRT_OPT_QUAD_PARTS with thresholds that send EVERY tile to the quad kernel (one 4x4 quadrant per wave, lanes = rays x slots),
    where its machinery is under load: a camera inside the densest cluster (every window overflows into its bag, bags are pruned,
    rays go again), SH degree 2, a fisheye frame with rayless pixels, a moving camera (costs of one view order the next), counters
    on (the instrumented instantiation).  The frame must be the frame of whole tiles, byte for byte; consumed hits per ray equal."""
    W, H = 160, 128
    acts, p0, sc, op0, center = make_scene(61, 60000, W, H, scale_boost=1.3)
    h, edges = np.histogramdd(acts["pos"], bins=24, range=[(-1.5, 1.5)] * 3)
    i = np.unravel_index(np.argmax(h), h.shape)
    eye = tuple(float((edges[k][i[k]] + edges[k][i[k] + 1]) / 2) for k in range(3))
    views = [("dense core", grt.default_params(W, H, center, eye=eye)), ("default", p0)]
    psh = grt.default_params(W, H, center); psh.sh_degree_max = 2
    views.append(("sh2", psh))
    pfe = grt.default_params(W, H, center, eye=(0.3, 0.2, 1.1)); pfe.mode_fisheye = 1
    views.append(("fisheye", pfe))
    tr = grt.Tracer(0)
    tr.upload(acts)
    for name, p in views:
        tr.set_option(grt.OPT_TILE_PARTS4_PCT, 0)
        tr.set_option(grt.OPT_COUNTERS, 1)
        ref8, reff = tr.render(p, want_f32=True)
        ref8, reff = ref8.clone(), reff.clone()
        c0 = tr.counters()
        tr.set_option(grt.OPT_TILE_PARTS4_PCT, 1)
        tr.set_option(grt.OPT_TILE_PARTS_LOAD_PCT, 0)
        for quad in (2, 0):
            tr.set_option(grt.OPT_QUAD_PARTS, quad)
            for counters in (1, 0):
                tr.set_option(grt.OPT_COUNTERS, counters)
                for it in range(3):
                    a8, af = tr.render(p, want_f32=True)
                    tr.check()
                    assert (a8 == ref8).all() and (af == reff).all(), (name, quad, counters, it)
                if counters:
                    c1 = tr.counters()
                    for k in ("rays", "segments", "hit_evals"):
                        assert c1[k] == c0[k], (name, quad, k)
                    assert c1["stall_exits"] == 0
    # a camera that moves: every frame is ordered (and split) by another view's costs
    tr.set_option(grt.OPT_QUAD_PARTS, 2)
    tr.set_option(grt.OPT_COUNTERS, 0)
    for j in range(6):
        pm = grt.default_params(W, H, center, eye=(0.4 * j, 0.1 * j, 3.0 - 0.3 * j))
        tr.set_option(grt.OPT_TILE_PARTS4_PCT, 0)
        r8 = tr.render(pm)[0].clone()
        tr.set_option(grt.OPT_TILE_PARTS4_PCT, 1)
        for it in range(2):
            assert (tr.render(pm)[0] == r8).all(), (j, it)
    tr.check()
    tr.close(); sc.close()


@pytest.mark.parametrize("fisheye", [False, True], ids=["pinhole", "fisheye"])
def test_mesh_primary_stage_per_tile_equals_per_lane(fisheye):
    """GRT_OPT_MESH_PRIMARY_WAVE: stage 1 of a mesh frame (camera ray -> closest mesh hit -> closest-hit shading,
    shaders/tracer.cuh:266-287, shaders/tracer.cu:155-187) with the 64 rays of an 8x8 tile walking the mesh tree together
    (k_primary_mesh_wave) against every lane walking it alone (k_primary_mesh): MIRROR / NORMAL / GLASS, two meshes of which one is a
    single triangle pair (a tree that is one leaf) and one a finely tessellated sphere seen at a grazing angle, a window launch, a tile
    list, the streaming-kernel pipeline, hit counters — the same bytes, and the per-lane megakernel's."""
    W, H = 200, 152
    acts, p0, sc, op, center = make_scene(23, 15000, W, H, scale_boost=0.5)
    sc.close()
    eye = np.float32([0, 0, 3])
    base = (0.25 * center + 0.75 * eye).astype(np.float32)
    v1, n1, f1 = grt.sphere_mesh(base + np.float32([0.2, 0.05, 0]), tess_u=96, tess_v=48)
    quad_v = np.float32([[-1.2, -0.9, -0.6], [0.1, -0.9, -0.4], [0.1, 0.4, -0.5], [-1.2, 0.4, -0.7]]) + center
    quad_n = np.float32([[0.1, 0.0, 1.0], [-0.2, 0.1, 1.0], [0.0, -0.15, 1.0], [0.15, 0.1, 1.0]])
    quad_n = (quad_n / np.linalg.norm(quad_n, axis=1, keepdims=True)).astype(np.float32)
    meshes = [(v1, n1, f1), (quad_v, quad_n, np.uint32([[0, 1, 2], [0, 2, 3]]))]
    for mesh_type in (grt.MIRROR, grt.NORMAL, grt.GLASS):
        p = grt.default_params(W, H, center, mesh_type=mesh_type, max_bounces=4, fisheye=fisheye)
        ref_t = grt.Tracer(0)
        ref_t.set_option(grt.OPT_KERNEL, 1)
        ref_t.upload(acts)
        ref_t.set_meshes(meshes)
        ref8, reff = ref_t.render(p, want_f32=True)
        ref_t.close()
        for kernel in (0, 3):
            t = grt.Tracer(0)
            t.set_option(grt.OPT_KERNEL, kernel)
            t.upload(acts)
            t.set_meshes(meshes)
            t.set_option(grt.OPT_COUNTERS, 1)
            cnts = {}
            for wave in (0, 1, 2):
                t.set_option(grt.OPT_MESH_PRIMARY_WAVE, wave)
                for _ in range(2):
                    a8, af = t.render(p, want_f32=True)
                    assert bool((a8 == ref8).all()) and bool((af == reff).all()), (mesh_type, kernel, wave)
                c = t.counters()
                cnts[wave] = (c["rays"], c["segments"], c["hit_evals"])
                assert c["stall_exits"] == 0
                # a window and a tile list of the same frame
                w8 = torch.zeros_like(ref8); wf = torch.zeros_like(reff)
                t.render(p, window=(40, 24, 168, 120), out_u8=w8, out_f32=wf)
                assert bool((w8[24:120, 40:168] == ref8[24:120, 40:168]).all()) and bool((wf[24:120, 40:168] == reff[24:120, 40:168]).all())
                tx, ty = tiles.grid(W, H, 32)
                _, _, cnt_t, max_cnt = tiles.my_tiles(tx * ty, 2, 1)
                buf = torch.zeros((max_cnt, 32, 32, 3), dtype=torch.uint8, device="cuda:0")
                t.render_tiles(p, 32, 32, 1, 2, cnt_t, out_u8=buf)
                full = tiles.assemble([torch.zeros_like(buf), buf], W, H, 32)
                for j in range(cnt_t):
                    k = 1 + 2 * j
                    y0, x0 = (k // tx) * 32, (k % tx) * 32
                    assert bool((full[y0:y0 + 32, x0:x0 + 32] == ref8[y0:y0 + 32, x0:x0 + 32]).all()), (mesh_type, kernel, wave, k)
            assert cnts[0] == cnts[1] == cnts[2]
            t.check()
            t.close()
