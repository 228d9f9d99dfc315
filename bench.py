#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: Mrays/s and ms/frame at 1080p on a 1M-Gaussian scene
(BASELINE.json metric, config C3), on 1..N MI355X of one node.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one frame: every rank traces its share of the frame's 32x32 screen tiles (scene replicated,
tiles dealt round-robin), the tile buffers are gathered to rank 0 over RCCL and un-permuted into the
frame (N = 1: one full-frame launch, no collective).  The frame is fixed, so scaling is STRONG.
Inputs (scene, BVH) are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

Frames in flight (--inflight D).  One GPU renders a frame in a few ms and the last part of that is a handful of
heavy 8x8 tiles finishing alone; a rank that owns 1/N of the tiles is bound by exactly those
tiles.  With D > 1 the bench keeps D consecutive frames in flight per rank — frame i goes to slot i % D, each slot
with its own context (its own scheduling feedback and buffers), HIP stream and output buffer — so the tail of one
frame overlaps the bulk of the next ones, as a viewer that double-buffers its display would run it.  Every one
of the K timed steps is still a complete frame (render, gather, un-permute) and all of them finish inside the timed
region.  Default: D = 1 on one GPU (each frame is synchronised, like the reference's render(); this is the run
the roofline and the rocprof summaries refer to), 4 on 2-4 GPUs, 8 on 8.  So that a scaling curve compares like
with like, EVERY run reports both figures next to `value`: `config.value_sync` (D = 1, every frame synchronised)
and `config.value_pipelined` (D = 4, or 8 from 8 GPUs on), each over its own K timed frames.

The step machinery (frame slots, tile split, gather, un-permute) is class FrameLoop: bench.py drives it with the HIP
library over RCCL, tests/test_multi_rank_cpu.py drives the same class with the CPU oracle over gloo.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python"))

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # one hardware queue per frame slot (HIP's default is 4)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (seed, n_gaussians, width, height, fisheye, mesh, max_bounces, per-axis log-scale noise)   — BASELINE.json configs
    "C1": (1, 10_000, 256, 256, False, False, 32, 0.0),
    "C2": (2, 100_000, 1280, 720, False, False, 32, 0.0),
    "C3": (3, 1_000_000, 1920, 1080, False, False, 32, 0.0),
    # C3 with trained-scene-like anisotropy: every axis' log-scale gets N(0, 1.6^2) on top (needles, pancakes)
    "C3a": (3, 1_000_000, 1920, 1080, False, False, 32, 1.6),
    # ... and the middle point: sigma 1.0 — a trained scene's scales are closer to this than to C3's 0.5 alone
    "C3b": (3, 1_000_000, 1920, 1080, False, False, 32, 1.0),
    "C4": (3, 1_000_000, 1920, 1080, False, True, 2, 0.0),
    "C5": (5, 3_000_000, 3840, 2160, True, False, 32, 0.0),
}
TILE = 32  # screen tiles dealt round-robin to the ranks (--tile: any multiple of 16; smaller tiles balance better, larger ones
           # keep neighbouring 8x8 wave tiles — which share BVH nodes and records — on one GPU)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
TRAFFIC_FILE = os.path.join("profiles", "traffic.json")
ROUND = "r06"  # `roofline.traffic` is printed only from counters collected in THIS round for THIS kernel (profiles/collect.sh tags its entries)


def kernel_name(variant, with_mesh, sh_degree, leaf_max=4, pieces=False):
    """Which kernel grt_render dispatches the Gaussian segment of camera rays to (csrc/grt_render.hip: launch_render)."""
    sh = "true" if sh_degree > 0 else "false"
    mesh = "true" if with_mesh else "false"
    if variant == 1:
        return "grt::k_render<false>"
    if variant == 2 and not with_mesh:
        return f"grt::k_render_wave<false, {sh}>"
    if variant in (0, 5) and leaf_max <= 4:
        return f"grt::k_render_tile<false, {sh}, {mesh}, 0, {'true' if pieces else 'false'}>"
    return f"grt::k_render_stream<false, {sh}, {mesh}>"


def contract_bytes(cnt, pixels, sh_degree, float_out=False):
    """SURVEY §8(d)'s algorithmic bytes per launch, the figure `roofline.achieved` / `roofline.frac` are computed from:
    B_alg = H * B_hit + V * B_node + P * B_out with H = consumed hit evaluations, B_hit = 44 + 12 (deg + 1)^2 (56 B at
    degree 0, 236 B at degree 3), V = BVH node visits (one 32-B child box per lane that tests it), P = pixels written
    (3 B, + 12 B with the float frame)."""
    return (cnt["hit_evals"] * (44 + 12 * (sh_degree + 1) ** 2) + cnt["node_visits"] * 32 + pixels * (3 + (12 if float_out else 0)))


def algorithmic_bytes(cnt, pixels, sh_degree, float_out=False):
    """`roofline.frac_as_fetched`: bytes at the granularity the kernel fetches them: every BVH child box / proxy record at the
    granularity the kernel fetches it (the counter is in 16-B units) — tile kernel: a child box is 32 B per LANE that
    tests it, a proxy record + its eye record 128 B per WAVE; streaming kernel: a 4-wide node 128 B, a proxy record +
    eye record 80 B, per wave — plus, for every consumed hit, its colour (16 B at degree 0, 192 B of SH above), and
    3 B per pixel (+12 B float)."""
    b_col = 16 if sh_degree == 0 else 192
    return cnt["rec_fetches"] * 16 + cnt["hit_evals"] * b_col + pixels * (3 + (12 if float_out else 0))


class FrameLoop:
    """The per-frame work of one rank: frame i runs in slot i % D (own stream / buffers); every rank renders its
    round-robin tiles into the slot's compact buffer, ONE gather per frame brings the buffers to rank 0, which
    un-permutes them into the slot's frame.  `render_full(slot, frame)` / `render_tiles(slot, first, stride, count,
    out)` do the rendering (HIP library in bench.py, the CPU oracle in the gloo test); `dist` is torch.distributed or
    None; `local_gather` replaces the collective by a local copy (one-rank emulation on one GPU)."""

    def __init__(self, torch, tiles, width, height, world, rank, slots, device, render_full, render_tiles, dist=None,
                 local_gather=False, streams=None, assemble=None, force_collective=False):
        self.torch, self.tiles, self.W, self.H = torch, tiles, width, height
        self.world, self.rank, self.D, self.dist, self.local = world, rank, slots, dist, local_gather
        # one rank normally renders the frame in one launch; force_collective sends it through the N-rank path all the
        # same (tile list -> compact buffer -> gather -> un-permute): the RCCL branch on a one-GPU box
        self.single = world == 1 and not force_collective
        self.render_full, self.render_tiles = render_full, render_tiles
        tx, ty = tiles.grid(width, height, TILE)
        self.n_tiles = tx * ty
        _, _, self.my_cnt, self.max_cnt = tiles.my_tiles(self.n_tiles, world, rank)
        self.frames = [torch.zeros((height, width, 3), dtype=torch.uint8, device=device) for _ in range(slots)]
        self.mines = self.gathereds = None
        if not self.single:
            self.mines = [torch.zeros((self.max_cnt, TILE, TILE, 3), dtype=torch.uint8, device=device) for _ in range(slots)]
            root = rank == 0 or local_gather
            # ONE buffer per slot, [world][max_cnt][tile][tile][3]; the collective writes rank r's tiles into its slice r
            self.gbufs = [torch.zeros((world,) + tuple(self.mines[0].shape), dtype=torch.uint8, device=device) if root else None
                          for _ in range(slots)]
            self.gathereds = [[g[r] for r in range(world)] if root else None for g in self.gbufs]
        self.streams = streams
        self.assemble = assemble  # assemble(slot, gbuf, world, max_cnt, frame): the HIP un-permute kernel; None = torch views

    def step(self, i=0):
        k = i % self.D
        ctx = self.torch.cuda.stream(self.streams[k]) if self.streams else _Null()
        with ctx:
            if self.single:
                self.render_full(k, self.frames[k])
                return
            self.render_tiles(k, self.rank, self.world, self.my_cnt, self.mines[k])
            if self.local:
                self.gathereds[k][self.rank].copy_(self.mines[k])  # stands in for the collective
            else:
                # RCCL: 7 peers -> 7 distinct xGMI links into rank 0, <= 0.8 MB each at 1080p
                self.dist.gather(self.mines[k], self.gathereds[k], dst=0)
            if self.rank == 0 or self.local:
                if self.assemble is not None:
                    self.assemble(k, self.gbufs[k], self.world, self.max_cnt, self.frames[k])
                else:
                    self.frames[k].copy_(self.tiles.assemble(self.gathereds[k], self.W, self.H, TILE))


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def build_scene(grt, workload):
    """Synthetic inputs of one BASELINE config: raw PLY columns -> activated arrays, camera, mesh."""
    seed, n, W, H, fisheye, with_mesh, max_bounces, aniso = WORKLOADS[workload]
    raw = grt.synth_scene(seed, n)
    if aniso:
        rng = np.random.default_rng(seed + 1000)
        raw["scale"] = (raw["scale"] + rng.normal(0.0, aniso, size=raw["scale"].shape)).astype(np.float32)
    acts = grt.activate(raw)
    center = grt.gaussian_center(acts["pos"])
    mesh = None
    if with_mesh:
        # the reference's procedural sphere written ONCE as OBJ with normals and loaded through the OBJ path (Y flip,
        # src/geometry/Primitives.cpp:175,179), placed at 0.25 lookat + 0.75 eye (src/GaussianTracer.cpp:630-638)
        v, nrm, f = grt.primitive_mesh(grt.PRIM_SPHERE)
        flip = np.float32([1, -1, 1])
        pos = (0.25 * center + 0.75 * np.float32([0, 0, 3])).astype(np.float32)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "sphere.obj")
            grt.write_obj(path, v * flip, nrm * flip, f)
            mesh = grt.load_obj(path, center=pos)
    return acts, center, mesh


def main():
    global TILE
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--sh-degree", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the untimed legs (cold frame, orbit, the other pipelining depth)")
    ap.add_argument("--kernel", type=int, default=0, help="traversal kernel variant (GRT_OPT_KERNEL)")
    ap.add_argument("--split", type=int, default=-1, help="GRT_OPT_SPLIT (piece length of the spatial splits, quarters of the typical proxy diagonal; 0 = off; -1 = library default)")
    ap.add_argument("--leaf-max", type=int, default=0, help="GRT_OPT_LEAF_MAX before the BVH is built (0 = library default)")
    ap.add_argument("--tile", type=int, default=TILE, help="edge of the screen tiles dealt round-robin to the ranks (multiple of 16)")
    ap.add_argument("--band-abs", type=int, default=-1, help="GRT_OPT_TILE_BAND_ABS (-1 = library default)")
    ap.add_argument("--opt", action="append", default=[], metavar="ID=VALUE",
                    help="grt_set_option(ID, VALUE) on every frame slot (tuning sweeps; include/grt.h lists the options), repeatable")
    ap.add_argument("--dump", default=None, help="write the frame as .npy (rank 0)")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="one process, one GPU: do the work of ONE rank of an N-rank run (tile list, frame slots, "
                         "un-permute; the collective is replaced by a local copy) - per-rank time without N GPUs")
    ap.add_argument("--emulate-rank", type=int, default=-1, help="which rank to emulate (default N // 2)")
    ap.add_argument("--force-collective", action="store_true",
                    help="with --gpus 1 under torch.distributed.run: init RCCL and send the frame through the N-rank path "
                         "(render_tiles -> dist.gather -> assemble) at world size 1")
    ap.add_argument("--inflight", type=int, default=0,
                    help="frames in flight per rank for `value` (0 = auto: 1 on one GPU, 4 on 2-4 GPUs, 8 on 8+)")
    args = ap.parse_args()
    if args.tile <= 0 or args.tile % 16:
        print("bench.py: --tile must be a positive multiple of 16", file=sys.stderr)
        sys.exit(2)
    TILE = args.tile

    import torch
    import grt
    import tiles

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dist = None
    force = bool(args.force_collective and world == 1 and not args.emulate_ranks)
    if force and "RANK" not in os.environ:
        print("bench.py: --force-collective needs the torch.distributed.run environment (RANK / WORLD_SIZE / MASTER_*)", file=sys.stderr)
        sys.exit(2)
    if world > 1 or force:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device(dev))

    emul = args.emulate_ranks if (args.emulate_ranks > 1 and world == 1) else 0
    t_world, t_rank = (emul, (args.emulate_rank if args.emulate_rank >= 0 else emul // 2)) if emul else (world, rank)

    seed, n, W, H, fisheye, with_mesh, max_bounces, aniso = WORKLOADS[args.workload]
    acts, center, mesh = build_scene(grt, args.workload)
    p = grt.default_params(W, H, center, sh_degree=args.sh_degree, fisheye=fisheye, mesh_type=grt.MIRROR,
                           max_bounces=max_bounces)
    D_pipe = 8 if t_world >= 8 else 4
    D = args.inflight if args.inflight > 0 else (1 if t_world == 1 else D_pipe)
    extra = not args.no_extra_legs
    D_other = (D_pipe if D == 1 else 1) if extra else 0  # the other pipelining depth, reported alongside
    n_ctx = max(D, D_other)
    t0 = time.time()
    trs = []
    for k in range(n_ctx):  # ONE scene; every further frame slot is a view of it (own stream state, eye records, scratch)
        if k == 0:
            t = grt.Tracer(local_rank)
            if args.split >= 0:
                t.set_option(grt.OPT_SPLIT, args.split)
            if args.leaf_max > 0:
                t.set_option(grt.OPT_LEAF_MAX, args.leaf_max)
            t.upload(acts)
            if mesh is not None:
                t.set_meshes([mesh])
        else:
            t = trs[0].view()
        t.set_option(grt.OPT_KERNEL, args.kernel)
        if args.band_abs >= 0:
            t.set_option(grt.OPT_TILE_BAND_ABS, args.band_abs)
        for kv in args.opt:
            k_, v_ = kv.split("=")
            t.set_option(int(k_), int(v_))
        trs.append(t)
    tr = trs[0]
    setup_s = (time.time() - t0) / n_ctx
    info = tr.bvh_info()

    def make_loop(depth):
        streams = [torch.cuda.Stream(device=dev) for _ in range(depth)] if depth > 1 else None
        return FrameLoop(torch, tiles, W, H, t_world, t_rank, depth, dev,
                         render_full=lambda k, frame: trs[k].render(p, out_u8=frame, want_u8=True),
                         render_tiles=lambda k, first, stride, cnt, out: trs[k].render_tiles(p, TILE, TILE, first, stride, cnt, out_u8=out),
                         dist=dist, local_gather=bool(emul), streams=streams, force_collective=force,
                         assemble=lambda k, g, world_, max_cnt, frame: trs[k].assemble_tiles(g, world_, max_cnt, TILE, W, H, frame))

    loop = make_loop(D)
    frame = loop.frames[0]
    my_cnt = loop.my_cnt

    def one_frame(t=tr, q=p):
        if t_world == 1:
            t.render(q, out_u8=frame, want_u8=True)
        else:
            t.render_tiles(q, TILE, TILE, t_rank, t_world, my_cnt, out_u8=loop.mines[0])

    # ---- instrumented frame (outside the timed region): counters for rays and algorithmic bytes; a steady-state frame
    #      (the first frames of a context size its scratch: their pass counts are not the timed frames') ----
    for _ in range(3):
        one_frame()
        tr.sync()
    tr.set_option(grt.OPT_COUNTERS, 1)
    one_frame()
    cnt = tr.counters()
    tr.set_option(grt.OPT_COUNTERS, 0)
    cold_ms = cold_wall_ms = cold_screen_ms = orbit_ms = None
    if extra:
        # a COLD frame: no costs of a previous frame (first frame, camera cut, new size).  The tiles are then ordered by a
        # projected particle count (one extra pass over the particle positions, inside the wall time below, outside
        # the kernel bracket); for the record also in plain screen order (GRT_OPT_COLD_ESTIMATE = 0)
        def cold_frame():
            tr.set_option(grt.OPT_FEEDBACK, 1)  # forgets the cost map
            torch.cuda.synchronize()
            tc = time.perf_counter()
            one_frame()
            torch.cuda.synchronize()
            return tr.last_kernel_ms(), (time.perf_counter() - tc) * 1e3
        cold = [cold_frame() for _ in range(3)]
        cold_ms, cold_wall_ms = float(np.median([x[0] for x in cold])), float(np.median([x[1] for x in cold]))
        tr.set_option(grt.OPT_COLD_ESTIMATE, 0)
        cold_screen_ms = float(np.median([cold_frame()[0] for _ in range(3)]))
        tr.set_option(grt.OPT_COLD_ESTIMATE, 1)
        one_frame(); one_frame()
        # a MOVING camera: the eye orbits the look-at point by 1.5 degrees per frame (the per-eye records are rebuilt
        # and the previous frame's tile costs order a slightly different frame); median over the last 8 of 10 frames
        mm = []
        eye0 = np.float32(list(p.eye)) - center
        for i in range(10):
            ang = np.deg2rad(1.5 * (i + 1))
            eye = center + np.float32([eye0[0] * np.cos(ang) + eye0[2] * np.sin(ang), eye0[1], -eye0[0] * np.sin(ang) + eye0[2] * np.cos(ang)])
            q = grt.default_params(W, H, center, sh_degree=args.sh_degree, fisheye=fisheye, mesh_type=grt.MIRROR,
                                   max_bounces=max_bounces, eye=tuple(float(x) for x in eye))
            one_frame(q=q)
            mm.append(tr.last_kernel_ms())
        orbit_ms = float(np.median(mm[2:]))
        one_frame(); one_frame()
    names = ("rays", "segments", "hit_evals", "rounds", "node_visits", "proxy_tests", "rec_fetches", "stall_exits")
    cnt_t = torch.tensor([cnt[k] for k in names], dtype=torch.int64, device=dev)
    if dist is not None:
        dist.all_reduce(cnt_t)
    tot = dict(zip(names, cnt_t.tolist()))
    if emul:  # the other ranks' rays are not traced here: scale this rank's share to the frame
        tot = {k: v * t_world for k, v in tot.items()}
    rays_per_frame = tot["segments"]  # primary rays + each secondary segment

    def timed(lp, steps, warmup):
        """K frames of one FrameLoop between barrier + synchronize: (elapsed s, synchronous single-frame latency ms)."""
        latency = None
        if lp.D > 1:
            for t in trs[:lp.D]:
                t.set_option(grt.OPT_FEEDBACK, 3)  # several frames in flight: no big-window split (a second stream per context)
            for i in range(2 * lp.D):
                lp.step(i)
            torch.cuda.synchronize()
            lat = []
            for _ in range(5):  # synchronous single-frame time of this rank layout (slot 0)
                torch.cuda.synchronize()
                tl = time.perf_counter()
                lp.step(0)
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - tl) * 1e3)
            latency = float(np.median(lat))
        for i in range(warmup):
            lp.step(i)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        kms = []
        ts = time.perf_counter()
        for i in range(steps):
            lp.step(i)
            if lp.D == 1:
                kms.append(tr.last_kernel_ms())  # HIP events on the launch stream; syncs like the reference's render()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - ts
        for t in trs[:lp.D]:
            t.set_option(grt.OPT_FEEDBACK, 1)
        return el, latency, kms

    elapsed, latency_ms, kern_ms = timed(loop, args.steps, args.warmup)
    if D > 1:
        # kernel duration for the roofline: synchronous launches of the same frame right after the timed region
        # (the launches inside it overlap each other, so their own durations are not launch durations)
        for _ in range(4):
            loop.step(0)
            kern_ms.append(trs[0].last_kernel_ms())
        kern_ms = kern_ms[1:]
    # ---- a timed leg under a MOVING camera (one GPU): the same K frames, the eye orbiting the look-at point by 1.5 degrees per
    #      frame, every frame synchronised like `value`'s — eye records rebuilt, the previous frame's costs ordering a different
    #      frame, the feedback kernels behind every frame.  The headline repeats one view (a frame that repeats the last one keeps
    #      its launch order and collects no costs); this is the same measurement for a viewer in motion: `value_orbit`.
    elapsed_orbit = None
    if extra and t_world == 1 and not force:
        eye0 = np.float32(list(p.eye)) - center

        def orbit_params(i):
            ang = np.deg2rad(1.5 * (i + 1))
            eye = center + np.float32([eye0[0] * np.cos(ang) + eye0[2] * np.sin(ang), eye0[1], -eye0[0] * np.sin(ang) + eye0[2] * np.cos(ang)])
            return grt.default_params(W, H, center, sh_degree=args.sh_degree, fisheye=fisheye, mesh_type=grt.MIRROR,
                                      max_bounces=max_bounces, eye=tuple(float(x) for x in eye))
        qs = [orbit_params(i) for i in range(args.warmup + args.steps)]
        for q in qs[:args.warmup]:
            one_frame(q=q)
            tr.sync()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for q in qs[args.warmup:]:
            one_frame(q=q)
            tr.last_kernel_ms()  # waits for the frame, as the synchronised loop of `value` does
        torch.cuda.synchronize()
        elapsed_orbit = time.perf_counter() - ts
        one_frame(); one_frame()
        tr.sync()
    other = None
    if D_other:
        lp2 = make_loop(D_other)
        el2, lat2, _ = timed(lp2, args.steps, args.warmup)
        other = (el2, lat2)
        if latency_ms is None:
            latency_ms = lat2
    for t in trs:
        t.check()  # raises when a wave gave up on live rays anywhere in the run (sticky device error word)
    mem = [t.memory_info() for t in trs]
    # (the overflow pool follows the median demand of the last eight frames it has READ BACK, and a loop that queues its frames without
    #  waiting reads few: sixteen more synchronised standing frames on slot 0, outside every timed region, show what the slot settles at)
    if extra:  # (not in the runs the profiles are taken from: their last K dispatches are the timed loop)
        for _ in range(16):
            one_frame()
            tr.sync()
    mem_settled = tr.memory_info()
    el = torch.tensor([elapsed, other[0] if other else 0.0], dtype=torch.float64, device=dev)
    km = torch.tensor([float(np.mean(kern_ms))], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
    elapsed, elapsed_other = float(el[0].item()), float(el[1].item())
    kernel_ms = float(km.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_per_frame * args.steps / elapsed / 1e6
        value_other = rays_per_frame * args.steps / elapsed_other / 1e6 if other else None
        value_sync, value_pipe = (value, value_other) if D == 1 else (value_other, value)
        # roofline of the dominant kernel: this rank's launch
        pix_mine = W * H if t_world == 1 else my_cnt * TILE * TILE
        b_alg = contract_bytes(cnt, pix_mine, args.sh_degree)       # H * B_hit + V * 32 + P * 3
        b_fetch = algorithmic_bytes(cnt, pix_mine, args.sh_degree)  # as fetched (records are shared by a wave: below the floor)
        b_min = cnt["hit_evals"] * (44 + 12 * (args.sh_degree + 1) ** 2) + pix_mine * 3  # SURVEY §8(d) floor
        achieved = b_alg / (kernel_ms * 1e-3) / 1e9
        traffic = valu = traffic_source = traffic_round = None
        kname = kernel_name(args.kernel, with_mesh, args.sh_degree, pieces=info["n_primitives"] > info["n_proxies"])
        tpath = os.path.join(ROOT, TRAFFIC_FILE)
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"{args.workload}_sh{args.sh_degree}_k{args.kernel}_n{t_world}"
                ent = tj.get(key, {})
                traffic_round = ent.get("round")
                #  counters of another round's build, or of another kernel, are not this build's traffic: not printed
                if ent and traffic_round == ROUND and ent.get("kernel") == kname:
                    traffic = ent.get("hbm_bytes_per_launch")
                    valu = ent.get("valu")  # SQ counters of the same launch: what actually bounds the kernel
                    if traffic is not None or valu is not None:
                        traffic_source = (f"{TRAFFIC_FILE} ['{key}']: rocprofv3 --pmc passes of this same command collected by "
                                          f"profiles/collect.sh in round {traffic_round} ({ent.get('collected', 'date not recorded')}); NOT measured by this run")
                elif ent:
                    traffic_source = (f"{TRAFFIC_FILE} ['{key}'] holds counters of round {traffic_round} for kernel {ent.get('kernel')}: not this "
                                      f"round's ({ROUND}) / this kernel ({kname}); traffic not printed")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s (+ ms/frame) @1080p, 1M-Gaussian PLY", "value": round(value, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "value_orbit": None if elapsed_orbit is None else round(rays_per_frame * args.steps / elapsed_orbit / 1e6, 3),
            "ms_per_step_orbit": None if elapsed_orbit is None else round(elapsed_orbit / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {n}-Gaussian synthetic 3DGS scene (seed {seed}), {W}x{H} "
                                   f"{'fisheye' if fisheye else 'pinhole'}, SH degree {args.sh_degree}"
                                   f"{', per-axis log-scale noise sigma %.1f (needles / pancakes)' % aniso if aniso else ''}"
                                   f"{', reflective sphere (reference primitive, through the OBJ path), <= 2 bounces' if with_mesh else ', no mesh'}",
                       "tile": (f"{TILE}x{TILE} round-robin over ranks, RCCL gather to rank 0" if (t_world > 1 or force)
                                else "full frame, one launch"),
                       "emulated_ranks": (f"rank {t_rank} of {t_world} on one GPU, no collective; value = this rank's rays x {t_world} / time"
                                          if emul else None),
                       "forced_collective": force or None,
                       "frames_in_flight": D,
                       "device_memory_bytes": {"scene": mem[0]["scene_bytes"], "frame_slots": [m["slot_bytes"] for m in mem],
                                               "frame_slot_0_settled": mem_settled["slot_bytes"],
                                               "overflow_pool_chunks": [m["overflow_chunks"] for m in mem],
                                               "overflow_demand_chunks": [m["overflow_demand"] for m in mem],
                                               "total": mem[0]["scene_bytes"] + sum(m["slot_bytes"] for m in mem),
                                               "note": "one scene shared by all frame slots (views); the pool of window-overflow bags (32 KiB chunks, "
                                                       "one to three per tile that overflows) follows the median demand of the last eight frames READ BACK, "
                                                       "both ways; frame_slots = as the legs of this run left them (cold frames, a moving camera, loops that "
                                                       "queue frames without waiting), frame_slot_0_settled = slot 0 after sixteen more synchronised standing frames"},
                       "value_sync": None if value_sync is None else round(value_sync, 3),
                       "value_pipelined": None if value_pipe is None else round(value_pipe, 3),
                       "pipelined_frames_in_flight": D if D > 1 else (D_other or None),
                       "latency_ms_per_frame": None if latency_ms is None else round(latency_ms, 4),
                       "rays_per_frame": rays_per_frame, "hit_evals_per_ray": round(tot["hit_evals"] / max(tot["segments"], 1), 2),
                       "rounds_per_ray": round(tot["rounds"] / max(tot["segments"], 1), 2),
                       "node_visits_per_ray": round(tot["node_visits"] / max(tot["segments"], 1), 1),
                       "proxy_tests_per_ray": round(tot["proxy_tests"] / max(tot["segments"], 1), 1),
                       "fetched_record_bytes_per_ray": round(16 * tot["rec_fetches"] / max(tot["segments"], 1), 1),
                       "stall_exits": tot["stall_exits"],
                       "bvh_height": info["height"], "n_proxies": info["n_proxies"], "n_bvh_primitives": info["n_primitives"], "bvh_build_ms": round(info["build_ms"], 2),
                       "setup_s": round(setup_s, 2), "kernel_variant": args.kernel, "options": args.opt or None,
                       "scheduling": "8x8 tiles launched heaviest-first from the previous frame's per-tile cost "
                                     "(steady state of an interactive viewer); launches of <= 12288 tiles with no other frame slot in flight run "
                                     "their heaviest tiles as 4x4 quadrants on the quad kernel (lanes = rays x slots) beside the camera-ray kernel; kernel_ms_cold / frame_ms_cold (wall, synchronised): a frame with no "
                                     "previous costs, tiles ordered by projected particle counts; kernel_ms_cold_screen_order: the same in screen order; "
                                     "kernel_ms_orbit: the eye orbits the look-at point by 1.5 degrees per frame "
                                     "(per-eye records rebuilt, last frame's costs order a different frame)",
                       "kernel_ms_cold": None if cold_ms is None else round(cold_ms, 4),
                       "frame_ms_cold": None if cold_wall_ms is None else round(cold_wall_ms, 4),
                       "kernel_ms_cold_screen_order": None if cold_screen_ms is None else round(cold_screen_ms, 4),
                       "kernel_ms_orbit": None if orbit_ms is None else round(orbit_ms, 4)},
            "kernel_ms": round(kernel_ms, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_collected_round": traffic_round,
                         "kernel": kname, "algorithmic_bytes_per_launch": int(b_alg),
                         "formula": "SURVEY 8(d): H * (44 + 12 (deg + 1)^2) + V * 32 + P * 3, H = consumed hit evaluations, V = child boxes tested, "
                                    "P = pixels; / kernel_ms (HIP events on the launch stream) / 8 TB/s",
                         "floor_bytes_per_launch": int(b_min),
                         "floor_frac": round(b_min / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "as_fetched_bytes_per_launch": int(b_fetch),
                         "frac_as_fetched": round(b_fetch / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "kernel_ms_median": round(float(np.median(kern_ms)), 4), "kernel_ms_min": round(float(np.min(kern_ms)), 4),
                         "valu_issue": valu},
        }
        if world == 1 and not args.no_cpu_baseline:
            # the oracle's frame is the CHECKER of this run's GPU frame as well as the baseline: every bench line carries its own
            # correctness figure.  Outside every timed region; the product path never calls it.
            g8 = gf = None
            if t_world == 1:
                g8, gf = tr.render(p, want_u8=True, want_f32=True)
                tr.sync()
                g8, gf = g8.cpu().numpy(), gf.cpu().numpy()
            out["cpu_baseline"], out["parity"] = cpu_baseline(acts, p, mesh, W, H, g8, gf)
        if args.dump:
            np.save(args.dump, frame.cpu().numpy())
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for t in reversed(trs):  # views first, then the scene
        t.close()


def cpu_baseline(acts, p, mesh, W, H, gpu_u8=None, gpu_f32=None):
    """Returns (cpu_baseline, parity).  The CPU oracle (oracle/grt_oracle.c, kind 'port': the reference itself needs OptiX and cannot run on a
    CPU) on a bounded sample of the same workload: the whole frame up to 1080p, the centred quarter-area window of it
    above (about 10-30 s of CPU work), on every core the process may run on (sched_getaffinity; `nproc` = os.cpu_count() is printed
    beside the thread count actually used).  `parity`: the oracle's pixels of that sample against the GPU frame of this run (u8 and
    float radiance) by the tests' rule — radiance within 1e-4 per channel, 8-bit values EQUAL except within 1e-4 of a quantisation
    step (tests/common.py: u8_matches)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from common import acts_to_particles, to_oracle_params, threshold_flip_explains, u8_matches
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    nproc = os.cpu_count() or cores  # all host cores the process may run on, nproc printed beside them
    affinity = cores
    # ... of which the container may only USE its CPU quota at a time (cgroup v2 cpu.max / v1 cfs quota): the GPU box shows 256
    # logical cores to a pod whose share is 16, and 256 oracle threads on 16 cores' worth of time run at half the speed of 16
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(round(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, int(round(q / per)))
        except Exception:
            pass
    if quota:
        cores = min(cores, quota)
    sc = O.Scene(acts_to_particles(acts))
    if mesh is not None:
        sc.set_mesh(*mesh)
    cw, ch = (W, H) if W * H <= 1920 * 1080 else (W // 2, H // 2)
    win = ((W - cw) // 2, (H - ch) // 2, (W - cw) // 2 + cw, (H - ch) // 2 + ch)
    t0 = time.perf_counter()
    ref_u8, ref_f32, c = sc.render(to_oracle_params(p), window=win, threads=cores, want_u8=True, want_f32=True)
    dt = time.perf_counter() - t0
    parity = None
    if gpu_u8 is not None:
        x0, y0, x1, y1 = win
        r8, rf, g8, gf = ref_u8[y0:y1, x0:x1], ref_f32[y0:y1, x0:x1], gpu_u8[y0:y1, x0:x1], gpu_f32[y0:y1, x0:x1]
        d = np.abs(gf - rf)
        over = (d > 1e-4).any(-1)
        ok = u8_matches(g8, r8, rf, 1e-4)
        # a pinhole pixel beyond the tolerance must be a ray ON one of the reference's hard thresholds (T > minTransmittance, tracer.cuh:341,353;
        # hitAlpha > alpha_min, :361): the oracle reproduces the GPU's value with that threshold moved by a relative 1e-6 .. 1e-4
        ys, xs = np.nonzero(over)
        flips = None
        if not p.mode_fisheye and len(ys) <= 64:
            q = to_oracle_params(p)
            flips = [threshold_flip_explains(sc, q, int(x0 + x), int(y0 + y), gf[y, x]) for y, x in zip(ys, xs)]
        n_flip = None if flips is None else sum(f is not None for f in flips)
        parity = {"pixels": int(cw * ch), "of_frame_pixels": int(W * H), "tolerance": 1e-4, "max_abs": float(d.max()),
                  "pixels_over_tolerance": int(over.sum()),
                  "of_which_threshold_flips": n_flip,  # explained by minTransmittance / alpha_min moved by <= 1e-4 relative (pinhole frames)
                  "max_abs_off_those_pixels": float(d[~over].max()) if (~over).any() else 0.0,
                  "u8_values_differing": int((g8 != r8).sum()),  # all of them, including the ones a radiance difference below 1e-4 explains
                  "u8_mismatch": int((~ok).sum()),               # 8-bit values that differ AWAY from a quantisation step
                  "u8_mismatch_outside_over_tolerance_pixels": int((~ok & ~over[..., None]).sum()),
                  "ok": bool((n_flip == int(over.sum()) <= max(2, int(1e-5 * over.size)) and not (~ok & ~over[..., None]).any()
                              and (d.max() <= 0.02)) if not p.mode_fisheye
                             else (over.mean() <= 2e-3 and (d[over].max() if over.any() else 0.0) <= 0.08 and not (~ok & ~over[..., None]).any())),
                  "rule": "GPU frame of this run vs the CPU oracle's (oracle/grt_oracle.c, the restatement of shaders/tracer.cu:17-110 -> "
                          "tracer.cuh:484-496) on the cpu_baseline sample: radiance within 1e-4 per channel; 8-bit equal except within 1e-4 "
                          "of a quantisation step; a pixel beyond that must be a ray on one of the reference's two hard thresholds (T > minTransmittance, "
                          "hitAlpha > alpha_min: expf's last bit decides), shown by the oracle reproducing the GPU value with the threshold "
                          "moved by a relative <= 1e-4; at most 1e-5 of the pixels, each within 0.02" + ("; fisheye: device and glibc trig differ in ulps, a near-tie may flip on <= 2e-3 of the "
                                                      "pixels, each within 0.08" if p.mode_fisheye else "")}
    sc.close()
    return {"value": round(c["segments"] / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "nproc": nproc, "affinity_cores": affinity,
            "cgroup_cpu_quota_cores": quota, "threads": cores, "kind": "port",
            "sample": f"{'whole' if (cw, ch) == (W, H) else 'centred'} {cw}x{ch} window of the same frame ({c['segments']} rays, {dt:.1f} s); "
                      f"full-frame estimate {W * H / (c['segments'] / dt) * 1e3:.0f} ms/frame",
            "hit_evals_per_ray": round(c["hit_evals"] / max(c["segments"], 1), 2)}, parity


if __name__ == "__main__":
    main()
