#!/usr/bin/env python3
"""subframe_probe.py [workload] — what would ONE mesh frame cost if its tiles ran as K independent wavefront pipelines on K streams?

A mesh frame (C4) is a chain of launches — primary segments, bundles, lone rays, twice — and every link ends in a tail of a few long
waves while the rest of the GPU idles (four FRAMES in flight hide those tails: 3.02 -> 2.31 ms per frame).  This probe does the same
inside one frame with what the C ABI already has: K frame slots (views of one scene), slot j renders the 32x32 tiles j, j + K, ... of
the frame (grt_render_tiles) on its own stream, all K are in flight together, and the frame is done when the last one is.  It prints
the wall time per frame for K = 1, 2, 3, 4, 6, 8 beside the plain one-launch frame.  (round 4, profiles/r04_experiments_log.md)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gaussian-ray-tracing_amd", "python"))
import torch
import grt, tiles
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
seed, n, W, H, fisheye, with_mesh, max_bounces, aniso = bench.WORKLOADS[wl]
acts, center, mesh = bench.build_scene(grt, wl)
p = grt.default_params(W, H, center, sh_degree=0, fisheye=fisheye, mesh_type=grt.MIRROR, max_bounces=max_bounces)
TILE = 32
KMAX = 8
trs = []
for k in range(KMAX):
    if k == 0:
        t = grt.Tracer(0)
        t.upload(acts)
        if mesh is not None:
            t.set_meshes([mesh])
    else:
        t = trs[0].view()
    trs.append(t)
tx, ty = tiles.grid(W, H, TILE)
n_tiles = tx * ty
frame = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
for _ in range(4):
    trs[0].render(p, out_u8=frame)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    trs[0].render(p, out_u8=frame)
    torch.cuda.synchronize()
print(f"{wl}: one launch chain: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms / frame", flush=True)
ref = frame.clone()
for K in (1, 2, 3, 4, 6, 8):
    streams = [torch.cuda.Stream(device="cuda:0") for _ in range(K)]
    cnts = [tiles.my_tiles(n_tiles, K, j)[2] for j in range(K)]
    max_cnt = max(cnts)
    gbuf = torch.zeros((K, max_cnt, TILE, TILE, 3), dtype=torch.uint8, device="cuda:0")
    out = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")

    def one():
        for j in range(K):
            with torch.cuda.stream(streams[j]):
                trs[j].render_tiles(p, TILE, TILE, j, K, cnts[j], out_u8=gbuf[j])
        torch.cuda.synchronize()
    for _ in range(5):
        one()
    t0 = time.perf_counter()
    for _ in range(20):
        one()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    trs[0].assemble_tiles(gbuf, K, max_cnt, TILE, W, H, out)
    torch.cuda.synchronize()
    same = bool((out == ref).all())
    print(f"{wl}: {K} sub-frames on {K} streams: {ms:.3f} ms / frame (without the un-permute), frame identical: {same}", flush=True)
