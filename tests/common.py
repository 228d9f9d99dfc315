"""Shared helpers for the test-suite: scene construction for both the oracle and the HIP library."""
import numpy as np

import grt
import oracle as O


def acts_to_particles(acts):
    n = len(acts["pos"])
    p = np.zeros(n, O.PARTICLE_DTYPE)
    p["pos"] = acts["pos"]; p["scale"] = acts["scale"]; p["quat"] = acts["quat"]
    p["opacity"] = acts["opacity"]; p["sh"] = acts["sh"]
    return p


def to_oracle_params(p):
    q = O.Params()
    q.width, q.height, q.sh_degree_max = p.width, p.height, p.sh_degree_max
    for name in ("eye", "U", "V", "W"):
        for k in range(3):
            getattr(q, name)[k] = getattr(p, name)[k]
    q.t_min, q.t_max, q.min_transmittance, q.alpha_min = p.t_min, p.t_max, p.minTransmittance, p.alpha_min
    q.mode_fisheye, q.type, q.max_bounces = p.mode_fisheye, p.type, p.max_bounces
    return q


def synth(seed, n, scale_boost=0.0):
    raw = grt.synth_scene(seed, n)
    if scale_boost:
        raw["scale"] = raw["scale"] + np.float32(scale_boost)
    acts = grt.activate(raw)
    return raw, acts


def make_scene(seed, n, width, height, **kw):
    """(acts, grt.Params, oracle Scene, oracle Params) with the reference's default camera."""
    scale_boost = kw.pop("scale_boost", 0.0)
    raw, acts = synth(seed, n, scale_boost)
    center = grt.gaussian_center(acts["pos"])
    p = grt.default_params(width, height, center, **kw)
    sc = O.Scene(acts_to_particles(acts))
    return acts, p, sc, to_oracle_params(p), center


def usable_cores():
    """Threads the oracle may usefully run on: the affinity mask capped by the container's cgroup CPU quota (the GPU box shows 256
    logical cores to a pod whose share is 16; 256 oracle threads there run at half the speed of 16)."""
    import os
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = min(cores, max(1, int(round(int(q) / int(per)))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                cores = min(cores, max(1, int(round(q / per))))
        except Exception:
            pass
    return cores


def u8_matches(got_u8, ref_u8, ref_f32, tol=1e-4):
    """Per value: the 8-bit frame EQUALS the oracle's, except where the oracle's radiance lies within `tol` of a
    quantisation step (x * 256 within tol * 256 of an integer), where a radiance difference below the tolerance may
    legitimately move the level by one.  Returns the boolean array (use .all(), or a fraction where geometry differs
    by ulps)."""
    du = np.abs(np.asarray(got_u8).astype(np.int32) - np.asarray(ref_u8).astype(np.int32))
    x = np.clip(np.asarray(ref_f32, dtype=np.float64), 0.0, 1.0) * 256.0
    near_step = np.abs(x - np.round(x)) <= tol * 256.0
    return (du == 0) | ((du == 1) & near_step)


def threshold_flip_explains(sc, op, x, y, gpu_rgb, tol=1e-4, rels=(1e-6, 1e-5, 1e-4)):
    """A pixel where the GPU and the oracle differ by more than the tolerance: is it a ray that sits ON one of the reference's two
    hard thresholds?  `if (T > minTransmittance)` (tracer.cuh:341,353) and `if (hitAlpha > alpha_min)` (tracer.cuh:361) are
    discontinuities of the reference's own function: expf differs in its last bit between glibc, the ROCm device library and CUDA's,
    the difference is carried by T, and a ray whose T lands within ulps of the threshold consumes one hit more or fewer (<= minT * c
    = 1e-3 of radiance; alpha_min: <= 0.01 T c).  The oracle re-renders the pixel with minTransmittance and / or alpha_min moved by a
    relative 1e-6 .. 1e-4: returns the smallest such change (rel, fT, fA) that reproduces the GPU's value within `tol`, else None —
    a pixel that no such change explains is a real mismatch."""
    import oracle as O
    g = np.asarray(gpu_rgb, np.float32)
    for rel in rels:
        for fT in (1.0, 1.0 - rel, 1.0 + rel):
            for fA in (1.0, 1.0 - rel, 1.0 + rel):
                if fT == 1.0 and fA == 1.0:
                    continue
                q = O.Params.from_buffer_copy(op)
                q.min_transmittance = float(np.float32(op.min_transmittance) * np.float32(fT))
                q.alpha_min = float(np.float32(op.alpha_min) * np.float32(fA))
                if float(np.abs(sc.render_pixel(q, x, y) - g).max()) <= tol:
                    return rel, fT, fA
    return None
