"""Screen-tile sharding of one frame over N ranks: tiles of TILE x TILE pixels numbered
row-major, tile t belongs to rank t % world (round-robin = load balance for spatially varying ray cost), each
rank renders its tiles into a compact [count][TILE][TILE][3] buffer, rank 0 gathers and un-permutes.
Pure index logic + torch ops: the same code runs over RCCL on GPUs (bench.py) and over gloo on CPU (tests)."""
import torch

TILE = 32


def grid(width, height, tile=TILE):
    return (width + tile - 1) // tile, (height + tile - 1) // tile


def my_tiles(n_tiles, world, rank):
    """(first, stride, count) of this rank's tiles and the padded per-rank count used by the gather."""
    count = (n_tiles - rank + world - 1) // world
    return rank, world, count, (n_tiles + world - 1) // world


def assemble(gathered, width, height, tile=TILE):
    """gathered: list (len = world) of [max_cnt][tile][tile][C] tensors -> [height][width][C] frame."""
    world = len(gathered)
    tx, ty = grid(width, height, tile)
    n_tiles = tx * ty
    c = gathered[0].shape[-1]
    g = torch.stack(gathered, 1).reshape(-1, tile, tile, c)[:n_tiles]  # tile t = j*world + rank
    img = g.reshape(ty, tx, tile, tile, c).permute(0, 2, 1, 3, 4).reshape(ty * tile, tx * tile, c)
    return img[:height, :width]


def gather_frame(dist, mine, width, height, rank, world, tile=TILE):
    """One collective per frame: gather every rank's compact tile buffer on rank 0, return the frame there."""
    gathered = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, gathered, dst=0)
    return assemble(gathered, width, height, tile) if rank == 0 else None
