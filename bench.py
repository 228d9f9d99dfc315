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
the roofline and the rocprof summaries refer to), 4 on 2-4 GPUs, 8 on 8; `config.latency_ms_per_frame` is the
synchronous single-frame time of the same rank layout.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gaussian-ray-tracing_amd", "python"))

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # one hardware queue per frame slot (HIP's default is 4)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (seed, n_gaussians, width, height, fisheye, mesh, max_bounces)   — BASELINE.json configs
    "C1": (1, 10_000, 256, 256, False, False, 32),
    "C2": (2, 100_000, 1280, 720, False, False, 32),
    "C3": (3, 1_000_000, 1920, 1080, False, False, 32),
    "C4": (3, 1_000_000, 1920, 1080, False, True, 2),
    "C5": (5, 3_000_000, 3840, 2160, True, False, 32),
}
TILE = 32
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def kernel_name(variant, with_mesh, n_proxies, sh_degree):
    """Which kernel grt_render dispatches to (csrc/grt_render.hip: launch_render)."""
    sh = "true" if sh_degree > 0 else "false"
    if variant == 1:
        return "grt::k_render<false>"
    if variant == 2 and not with_mesh:
        return f"grt::k_render_wave<false, {sh}>"
    return f"grt::k_render_stream<false, {sh}, {'true' if with_mesh else 'false'}>"


def algorithmic_bytes(cnt, pixels, sh_degree, float_out=False):
    """Bytes the kernel's algorithm needs per frame: every BVH node / proxy record at the
    granularity the kernel fetches it — once per WAVE, by scalar load: a 4-wide node is 128 B, a proxy record
    64 B (mu/A/s/opacity/id) + its 16-B eye record; the counter is in 16-B units — plus, for every consumed
    hit, its colour (16 B at degree 0, 192 B of SH above), and 3 B per pixel (+12 B float)."""
    b_col = 16 if sh_degree == 0 else 192
    return cnt["rec_fetches"] * 16 + cnt["hit_evals"] * b_col + pixels * (3 + (12 if float_out else 0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--sh-degree", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel", type=int, default=0, help="traversal kernel variant (GRT_OPT_KERNEL)")
    ap.add_argument("--dump", default=None, help="write the frame as .npy (rank 0)")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="one process, one GPU: do the work of ONE rank of an N-rank run (tile list, frame slots, "
                         "un-permute; the collective is replaced by a local copy) - per-rank time without N GPUs")
    ap.add_argument("--emulate-rank", type=int, default=-1, help="which rank to emulate (default N // 2)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="frames in flight per rank (0 = auto: 1 on one GPU, 4 on 2-4 GPUs, 8 on 8+)")
    args = ap.parse_args()

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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device(dev))

    emul = args.emulate_ranks if (args.emulate_ranks > 1 and world == 1) else 0
    t_world, t_rank = (emul, (args.emulate_rank if args.emulate_rank >= 0 else emul // 2)) if emul else (world, rank)

    seed, n, W, H, fisheye, with_mesh, max_bounces = WORKLOADS[args.workload]
    raw = grt.synth_scene(seed, n)
    acts = grt.activate(raw)
    center = grt.gaussian_center(acts["pos"])
    p = grt.default_params(W, H, center, sh_degree=args.sh_degree, fisheye=fisheye, mesh_type=grt.MIRROR,
                           max_bounces=max_bounces)
    D = args.inflight if args.inflight > 0 else (1 if t_world == 1 else (8 if t_world >= 8 else 4))
    mesh = None
    if with_mesh:
        pos = (0.25 * center + 0.75 * np.float32([0, 0, 3])).astype(np.float32)  # src/GaussianTracer.cpp:630-638
        mesh = grt.sphere_mesh(pos)
    t0 = time.time()
    trs = []
    for _ in range(D):  # one context per frame slot (the scene is a few hundred MB: replicated per slot)
        t = grt.Tracer(local_rank)
        t.set_option(grt.OPT_KERNEL, args.kernel)
        t.upload(acts)
        if mesh is not None:
            t.set_meshes([mesh])
        trs.append(t)
    tr = trs[0]
    setup_s = (time.time() - t0) / D
    info = tr.bvh_info()

    # ---- work split ----
    tx, ty = tiles.grid(W, H, TILE)
    n_tiles = tx * ty
    _, _, my_cnt, max_cnt = tiles.my_tiles(n_tiles, t_world, t_rank)
    frames = [torch.zeros((H, W, 3), dtype=torch.uint8, device=dev) for _ in range(D)]
    frame = frames[0]
    if t_world > 1:
        mines = [torch.zeros((max_cnt, TILE, TILE, 3), dtype=torch.uint8, device=dev) for _ in range(D)]
        mine = mines[0]
        gathereds = [[torch.zeros_like(mine) for _ in range(t_world)] if (rank == 0 or emul) else None for _ in range(D)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(D)] if D > 1 else [torch.cuda.current_stream()]

    def step(i=0):
        k = i % D
        with torch.cuda.stream(streams[k]):
            if t_world == 1:
                trs[k].render(p, out_u8=frames[k], want_u8=True)
            else:
                trs[k].render_tiles(p, TILE, TILE, t_rank, t_world, my_cnt, out_u8=mines[k])
                if emul:
                    gathereds[k][t_rank].copy_(mines[k])  # stands in for the collective
                else:
                    dist.gather(mines[k], gathereds[k], dst=0)  # RCCL: 7 peers -> 7 distinct xGMI links into rank 0, <= 0.8 MB each
                if rank == 0:
                    frames[k].copy_(tiles.assemble(gathereds[k], W, H, TILE))

    # ---- instrumented frame (outside the timed region): counters for rays and algorithmic bytes ----
    tr.set_option(grt.OPT_COUNTERS, 1)
    if t_world == 1:
        tr.render(p, out_u8=frame, want_u8=True)
    else:
        tr.render_tiles(p, TILE, TILE, t_rank, t_world, my_cnt, out_u8=mine)
    cnt = tr.counters()
    tr.set_option(grt.OPT_COUNTERS, 0)
    # first frame without scheduling feedback (what a cold start / a camera cut costs), for the record
    tr.set_option(grt.OPT_FEEDBACK, 0)
    step_cold = (lambda: tr.render(p, out_u8=frame, want_u8=True)) if t_world == 1 else \
        (lambda: tr.render_tiles(p, TILE, TILE, t_rank, t_world, my_cnt, out_u8=mine))
    step_cold(); step_cold()
    cold_ms = tr.last_kernel_ms()
    tr.set_option(grt.OPT_FEEDBACK, 1)
    # a moving camera: the eye records (one pass over the particles) are rebuilt inside the timed kernel bracket
    moving_ms = None
    if t_world == 1:
        import copy
        mm = []
        for i in range(6):
            q = copy.copy(p)
            q.eye[0] = p.eye[0] + 1e-4 * (i + 1)
            tr.render(q, out_u8=frame, want_u8=True)
            mm.append(tr.last_kernel_ms())
        moving_ms = float(np.median(mm[2:]))
        tr.render(p, out_u8=frame, want_u8=True)
    names = ("rays", "segments", "hit_evals", "rounds", "node_visits", "proxy_tests", "rec_fetches")
    cnt_t = torch.tensor([cnt[k] for k in names], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(cnt_t)
    tot = dict(zip(names, cnt_t.tolist()))
    if emul:  # the other ranks' rays are not traced here: scale this rank's share to the frame
        tot = {k: v * t_world for k, v in tot.items()}
    rays_per_frame = tot["segments"]  # primary rays + each secondary segment

    # every frame slot gets its scheduling feedback before anything is timed (set-up, like the frames above);
    # with several frames in flight the big-window split (a second stream per context) is left out
    latency_ms = None
    if D > 1:
        for t in trs:
            t.set_option(grt.OPT_FEEDBACK, 3)
        for i in range(2 * D):
            step(i)
        torch.cuda.synchronize()
        # synchronous single-frame time of this rank layout (slot 0): the latency a frame has without the overlap
        lat = []
        for _ in range(5):
            torch.cuda.synchronize()
            tl = time.perf_counter()
            step(0)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - tl) * 1e3)
        latency_ms = float(np.median(lat))
    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    kern_ms = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
        if D == 1:
            kern_ms.append(tr.last_kernel_ms())  # HIP events on the launch stream; syncs like the reference's render()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if D > 1:
        # kernel duration for the roofline: synchronous launches of the same frame right after the timed region
        # (the launches inside it overlap each other, so their own durations are not launch durations)
        for _ in range(4):
            step(0)
            kern_ms.append(trs[0].last_kernel_ms())
        kern_ms = kern_ms[1:]
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    km = torch.tensor([float(np.mean(kern_ms))], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    kernel_ms = float(km.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_per_frame * args.steps / elapsed / 1e6
        # roofline of the dominant (only) kernel: this rank's launch
        pix_mine = W * H if t_world == 1 else my_cnt * TILE * TILE
        b_alg = algorithmic_bytes(cnt, pix_mine, args.sh_degree)
        b_min = cnt["hit_evals"] * (44 + 12 * (args.sh_degree + 1) ** 2) + pix_mine * 3  # SURVEY §8(d) floor
        achieved = b_alg / (kernel_ms * 1e-3) / 1e9
        traffic = None
        valu = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"{args.workload}_sh{args.sh_degree}_k{args.kernel}_n{t_world}"
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
                valu = tj.get(key, {}).get("valu")  # SQ counters of the same launch: what actually bounds the kernel
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s (+ ms/frame) @1080p, 1M-Gaussian PLY", "value": round(value, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {n}-Gaussian synthetic 3DGS scene (seed {seed}), {W}x{H} "
                                   f"{'fisheye' if fisheye else 'pinhole'}, SH degree {args.sh_degree}"
                                   f"{', reflective sphere mesh, <=2 bounces' if with_mesh else ', no mesh'}",
                       "tile": f"{TILE}x{TILE} round-robin over ranks, RCCL gather to rank 0" if t_world > 1 else "full frame, one launch",
                       "emulated_ranks": (f"rank {t_rank} of {t_world} on one GPU, no collective; value = this rank's rays x {t_world} / time"
                                          if emul else None),
                       "frames_in_flight": D,
                       "latency_ms_per_frame": None if latency_ms is None else round(latency_ms, 4),
                       "rays_per_frame": rays_per_frame, "hit_evals_per_ray": round(tot["hit_evals"] / max(tot["segments"], 1), 2),
                       "rounds_per_ray": round(tot["rounds"] / max(tot["segments"], 1), 2),
                       "node_visits_per_ray": round(tot["node_visits"] / max(tot["segments"], 1), 1),
                       "proxy_tests_per_ray": round(tot["proxy_tests"] / max(tot["segments"], 1), 1),
                       "fetched_record_bytes_per_ray": round(16 * tot["rec_fetches"] / max(tot["segments"], 1), 1),
                       "bvh_height": info["height"], "n_proxies": info["n_proxies"], "bvh_build_ms": round(info["build_ms"], 2),
                       "setup_s": round(setup_s, 2), "kernel_variant": args.kernel,
                       "scheduling": "8x8 tiles launched heaviest-first from the previous frame's per-tile cost "
                                     "(steady state of an interactive viewer); kernel_ms_cold is one frame without it, "
                                     "kernel_ms_moving_camera re-derives the per-eye records every frame",
                       "kernel_ms_cold": round(cold_ms, 4),
                       "kernel_ms_moving_camera": None if moving_ms is None else round(moving_ms, 4)},
            "kernel_ms": round(kernel_ms, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": kernel_name(args.kernel, with_mesh, info["n_proxies"], args.sh_degree), "algorithmic_bytes_per_launch": int(b_alg),
                         "floor_bytes_per_launch": int(b_min),
                         "floor_frac": round(b_min / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "valu_issue": valu},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(acts, p, mesh, W, H)
        if args.dump:
            np.save(args.dump, frame.cpu().numpy())
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for t in trs:
        t.close()


def cpu_baseline(acts, p, mesh, W, H):
    """The CPU oracle (oracle/grt_oracle.c, kind 'port': the reference itself needs OptiX and cannot run on a
    CPU) on a bounded sample of the same workload: the centred quarter-area crop of the same frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from common import acts_to_particles, to_oracle_params
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)  # the GPU box's CPU share for one GPU
    sc = O.Scene(acts_to_particles(acts))
    if mesh is not None:
        sc.set_mesh(*mesh)
    cw, ch = (W, H) if W * H <= 1920 * 1080 else (W // 2, H // 2)
    win = ((W - cw) // 2, (H - ch) // 2, (W - cw) // 2 + cw, (H - ch) // 2 + ch)
    t0 = time.perf_counter()
    _, _, c = sc.render(to_oracle_params(p), window=win, threads=cores, want_u8=True, want_f32=False)
    dt = time.perf_counter() - t0
    sc.close()
    return {"value": round(c["segments"] / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{'whole' if (cw, ch) == (W, H) else 'centred'} {cw}x{ch} window of the same frame ({c['segments']} rays, {dt:.1f} s); "
                      f"full-frame estimate {W * H / (c['segments'] / dt) * 1e3:.0f} ms/frame",
            "hit_evals_per_ray": round(c["hit_evals"] / max(c["segments"], 1), 2)}


if __name__ == "__main__":
    main()
