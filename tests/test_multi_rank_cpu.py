"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of bench.py's OWN step machinery (class FrameLoop: frame slots,
round-robin tile split, one gather per frame, un-permute on rank 0).  Only the renderer is swapped: each rank "renders"
its tiles with the CPU oracle into the compact buffer layout grt_render_tiles uses.  The assembled frame of every
slot must equal the single-rank frame bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import tiles
from common import make_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, W, H, slots, out_path, force=False):
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    acts, p, sc, op, _ = make_scene(31, 800, W, H, scale_boost=1.0)
    tile = bench.TILE
    tx, ty = tiles.grid(W, H, tile)

    def render_tiles(slot, first, stride, cnt, out):  # what grt_render_tiles does, on the CPU oracle
        out.zero_()
        for j in range(cnt):
            t = first + j * stride
            x0, y0 = (t % tx) * tile, (t // tx) * tile
            x1, y1 = min(x0 + tile, W), min(y0 + tile, H)
            u8, _, _ = sc.render(op, window=(x0, y0, x1, y1), threads=1, want_f32=False)
            out[j, : y1 - y0, : x1 - x0] = torch.from_numpy(u8[y0:y1, x0:x1])

    loop = bench.FrameLoop(torch, tiles, W, H, world, rank, slots, "cpu", render_full=None, render_tiles=render_tiles, dist=dist,
                           force_collective=force)
    for i in range(slots + 1):  # every slot, and slot 0 twice
        loop.step(i)
    if rank == 0:
        np.save(out_path, np.stack([f.numpy() for f in loop.frames]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bench_frame_loop_over_gloo_equals_single_rank(world, tmp_path):
    W, H, slots = 112, 80, 2  # ragged right and bottom tiles; two frames in flight
    out = str(tmp_path / "frames.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, slots, out), nprocs=world, join=True)
    acts, p, sc, op, _ = make_scene(31, 800, W, H, scale_boost=1.0)
    ref, _, cnt = sc.render(op, want_f32=False)
    got = np.load(out)
    assert got.shape == (slots,) + ref.shape
    for k in range(slots):
        assert (got[k] == ref).all(), k
    assert cnt["hit_evals"] > W * H


def test_forced_collective_at_world_size_one_over_gloo(tmp_path):
    """bench.py --force-collective: ONE rank still goes through tile list -> gather -> un-permute (the branch the GPU test
    runs over RCCL in a child process); here over gloo."""
    W, H, slots = 112, 80, 2
    out = str(tmp_path / "frames.npy")
    mp.spawn(_worker, args=(1, _free_port(), W, H, slots, out, True), nprocs=1, join=True)
    acts, p, sc, op, _ = make_scene(31, 800, W, H, scale_boost=1.0)
    ref, _, _ = sc.render(op, want_f32=False)
    got = np.load(out)
    for k in range(slots):
        assert (got[k] == ref).all(), k


def test_tile_bookkeeping():
    for n_tiles in (1, 7, 2040, 2041):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                first, stride, cnt, max_cnt = tiles.my_tiles(n_tiles, world, r)
                assert cnt <= max_cnt
                seen += [first + j * stride for j in range(cnt)]
            assert sorted(seen) == list(range(n_tiles))
