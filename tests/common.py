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


def u8_matches(got_u8, ref_u8, ref_f32, tol=1e-4):
    """Per value: the 8-bit frame EQUALS the oracle's, except where the oracle's radiance lies within `tol` of a
    quantisation step (x * 256 within tol * 256 of an integer), where a radiance difference below the tolerance may
    legitimately move the level by one.  Returns the boolean array (use .all(), or a fraction where geometry differs
    by ulps)."""
    du = np.abs(np.asarray(got_u8).astype(np.int32) - np.asarray(ref_u8).astype(np.int32))
    x = np.clip(np.asarray(ref_f32, dtype=np.float64), 0.0, 1.0) * 256.0
    near_step = np.abs(x - np.round(x)) <= tol * 256.0
    return (du == 0) | ((du == 1) & near_step)
