"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of the tile sharding used by bench.py.  Each rank
"renders" its round-robin tiles with the CPU oracle into the compact buffer layout grt_render_tiles uses, the
buffers are gathered on rank 0 and un-permuted; the result must equal the single-rank frame bit for bit."""
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


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, W, H, tile, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    acts, p, sc, op, _ = make_scene(31, 800, W, H, scale_boost=1.0)
    tx, ty = tiles.grid(W, H, tile)
    first, stride, cnt, max_cnt = tiles.my_tiles(tx * ty, world, rank)
    mine = torch.zeros((max_cnt, tile, tile, 3), dtype=torch.uint8)
    for j in range(cnt):
        t = first + j * stride
        x0, y0 = (t % tx) * tile, (t // tx) * tile
        x1, y1 = min(x0 + tile, W), min(y0 + tile, H)
        u8, _, _ = sc.render(op, window=(x0, y0, x1, y1), threads=1, want_f32=False)
        mine[j, : y1 - y0, : x1 - x0] = torch.from_numpy(u8[y0:y1, x0:x1])
    frame = tiles.gather_frame(dist, mine, W, H, rank, world, tile)
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_sharding_gloo_equals_single_rank(world, tmp_path):
    W, H, tile = 112, 80, 32  # ragged right and bottom tiles
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, tile, out), nprocs=world, join=True)
    acts, p, sc, op, _ = make_scene(31, 800, W, H, scale_boost=1.0)
    ref, _, cnt = sc.render(op, want_f32=False)
    got = np.load(out)
    assert got.shape == ref.shape and (got == ref).all()
    assert cnt["hit_evals"] > W * H


def test_tile_bookkeeping():
    for n_tiles in (1, 7, 2040, 2041):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                first, stride, cnt, max_cnt = tiles.my_tiles(n_tiles, world, r)
                assert cnt <= max_cnt
                seen += [first + j * stride for j in range(cnt)]
            assert sorted(seen) == list(range(n_tiles))
