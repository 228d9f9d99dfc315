"""bench.py end to end on the smallest workload: the JSON contract of the default run, and the multi-slot
(frames in flight) step of a multi-GPU rank, exercised on one GPU through --emulate-ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "C1", "--steps", "4", "--warmup", "1", *args],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # ONE JSON line
    return json.loads(lines[0])


def test_bench_json_contract():
    d = _run()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["unit"] == "Mrays/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["higher_is_better"] is True
    c = d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in c and c["frames_in_flight"] == 1
    # both pipelining depths are reported, whatever `value` was measured at
    assert abs(c["value_sync"] - d["value"]) < 1e-6 and c["value_pipelined"] > 0 and c["pipelined_frames_in_flight"] == 4
    assert c["kernel_ms_cold"] > 0 and c["kernel_ms_orbit"] > 0 and c["stall_exits"] == 0
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf and "traffic_source" in rf
    # traffic is printed only from counters collected in THIS round for THIS kernel (profiles/traffic.json entries carry both)
    assert "traffic_collected_round" in rf
    assert rf["traffic"] is None or (rf["traffic_collected_round"] == "r06" and rf["traffic_source"] is not None)
    assert "k_render_tile" in rf["kernel"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "Mrays/s" and cb["sample"]
    assert d["value"] > 5 * cb["value"]
    # every bench line carries its own correctness figure: this run's GPU frame against the oracle's on the cpu_baseline sample
    pr = d["parity"]
    assert pr["pixels"] == 256 * 256 == pr["of_frame_pixels"] and pr["ok"] is True
    assert pr["u8_mismatch_outside_over_tolerance_pixels"] == 0 and pr["pixels_over_tolerance"] == pr["of_which_threshold_flips"] <= 2
    assert 0 <= pr["max_abs_off_those_pixels"] <= 1e-4


def test_bench_frames_in_flight_on_an_emulated_rank():
    d = _run("--no-cpu-baseline", "--emulate-ranks", "2")
    c = d["config"]
    assert c["frames_in_flight"] == 4 and c["latency_ms_per_frame"] > 0 and "rank 1 of 2" in c["emulated_ranks"]
    assert d["value"] > 0 and d["kernel_ms"] > 0
    assert abs(c["value_pipelined"] - d["value"]) < 1e-6 and c["value_sync"] > 0
