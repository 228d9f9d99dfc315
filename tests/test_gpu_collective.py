"""The RCCL branch of bench.py on the hardware there is: ONE rank under torch.distributed.run with --force-collective
goes through init_process_group("nccl"), render_tiles -> dist.gather -> grt_assemble_tiles — the N-rank frame path at
world size 1 — in a fresh child process (the launcher starts before anything touches the GPU).  The gathered frame
must be the one-launch frame, byte for byte."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import grt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_forced_collective_frame_equals_the_one_launch_frame(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    dump = str(tmp_path / "frame.npy")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collective",
           "--workload", "C2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra-legs", "--dump", dump]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["config"]["forced_collective"] is True and j["value"] > 0
    assert "RCCL gather" in j["config"]["tile"]
    got = np.load(dump)
    # the same frame in ONE launch, in this process
    seed, n, W, H, fisheye, with_mesh, max_bounces, aniso = bench.WORKLOADS["C2"]
    acts, center, mesh = bench.build_scene(grt, "C2")
    p = grt.default_params(W, H, center, fisheye=fisheye, max_bounces=max_bounces)
    tr = grt.Tracer(0)
    tr.upload(acts)
    u8, _ = tr.render(p)
    tr.check()
    assert got.shape == (H, W, 3) and bool((u8.cpu().numpy() == got).all())
    tr.close()
