"""ctypes binding of the CPU oracle (oracle/libgrt_oracle.so) and of oracle/_ref/libgrt_ref.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg — never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libgrt_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libgrt_ref.so")

PARTICLE_DTYPE = np.dtype(
    [("pos", "<f4", 3), ("scale", "<f4", 3), ("quat", "<f4", 4), ("opacity", "<f4"), ("sh", "<f4", (16, 3))]
)
assert PARTICLE_DTYPE.itemsize == 236


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("sh_degree_max", C.c_uint32),
        ("eye", C.c_float * 3), ("U", C.c_float * 3), ("V", C.c_float * 3), ("W", C.c_float * 3),
        ("t_min", C.c_float), ("t_max", C.c_float), ("min_transmittance", C.c_float), ("alpha_min", C.c_float),
        ("mode_fisheye", C.c_int32), ("type", C.c_int32), ("max_bounces", C.c_uint32),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "segments", "hit_evals", "rounds", "node_visits", "proxy_tests")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force=False):
    """(Re)build liboracle with the committed Makefile (gcc); returns the path."""
    src = os.path.join(_HERE, "grt_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libgrt_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        fp = C.POINTER(C.c_float)
        L.grto_proxy_scale.restype = C.c_float
        L.grto_proxy_scale.argtypes = [C.c_float, C.c_float]
        L.grto_compute_response.restype = C.c_float
        L.grto_compute_response.argtypes = [C.c_void_p, fp, fp]
        L.grto_quantize.restype = C.c_uint8
        L.grto_quantize.argtypes = [C.c_float]
        L.grto_scene_create.restype = C.c_void_p
        L.grto_scene_create.argtypes = [C.c_void_p, C.c_uint64, C.c_float]
        L.grto_scene_destroy.argtypes = [C.c_void_p]
        L.grto_scene_use_bvh.argtypes = [C.c_void_p, C.c_int]
        L.grto_scene_set_proxy_triangles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.grto_scene_set_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.grto_render.argtypes = [C.c_void_p, C.POINTER(Params), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_void_p, C.c_void_p, C.POINTER(Counters), C.c_int]
        L.grto_render_rays.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint64, C.c_void_p,
                                       C.POINTER(Counters), C.c_int]
        L.grto_trace_gps.restype = C.c_uint32
        L.grto_trace_gps.argtypes = [C.c_void_p, fp, fp, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.grto_trace.argtypes = [C.c_void_p, C.POINTER(Params), fp, fp, C.c_float, C.c_float, fp, fp, C.c_void_p]
        L.grto_render_pixel.argtypes = [C.c_void_p, C.POINTER(Params), C.c_uint32, C.c_uint32, fp, C.c_void_p]
        L.grto_proxy_hit.restype = C.c_int
        L.grto_proxy_hit.argtypes = [C.c_void_p, C.c_float, fp, fp, fp, fp]
        L.grto_refract.restype = C.c_int
        L.grto_refract.argtypes = [fp, fp, C.c_float, fp]
        L.grto_get_fisheye_ray.restype = C.c_int
        L.grto_tri_hit.restype = C.c_int
        _lib = L
    return _lib


_ref = None


def ref():
    """oracle/_ref (reference sources compiled in place) or None when it was never built."""
    global _ref
    if _ref is None and os.path.exists(_REF):
        R = C.CDLL(_REF)
        R.ref_length.restype = C.c_float
        R.ref_dot.restype = C.c_float
        R.ref_glm_dot.restype = C.c_float
        R.ref_clamp.restype = C.c_float
        R.ref_clamp.argtypes = [C.c_float, C.c_float, C.c_float]
        _ref = R
    return _ref


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def activate(pos, f_dc, f_rest, opacity_logit, log_scale, rot):
    """Raw PLY columns (N,3),(N,3),(N,45),(N,),(N,3),(N,4) -> activated particle array (GaussianData.cpp:97-128)."""
    L = lib()
    n = len(pos)
    out = np.zeros(n, dtype=PARTICLE_DTYPE)
    pos, f_dc, f_rest, log_scale, rot = (np.ascontiguousarray(x, np.float32) for x in (pos, f_dc, f_rest, log_scale, rot))
    opacity_logit = np.ascontiguousarray(opacity_logit, np.float32)
    fp = C.POINTER(C.c_float)
    L.grto_activate.argtypes = [fp, fp, fp, C.c_float, fp, fp, C.c_void_p]
    base = out.ctypes.data
    for i in range(n):
        L.grto_activate(pos[i].ctypes.data_as(fp), f_dc[i].ctypes.data_as(fp), f_rest[i].ctypes.data_as(fp),
                        float(opacity_logit[i]), log_scale[i].ctypes.data_as(fp), rot[i].ctypes.data_as(fp),
                        C.c_void_p(base + i * 236))
    return out


def make_params(width, height, eye, U, V, W, sh_degree=0, fisheye=False, mesh_type=0, max_bounces=32,
                t_min=1e-3, t_max=1e5, min_transmittance=1e-3, alpha_min=0.01):
    p = Params()
    p.width, p.height, p.sh_degree_max = width, height, sh_degree
    for name, v in (("eye", eye), ("U", U), ("V", V), ("W", W)):
        arr = getattr(p, name)
        for k in range(3):
            arr[k] = float(v[k])
    p.t_min, p.t_max, p.min_transmittance, p.alpha_min = t_min, t_max, min_transmittance, alpha_min
    p.mode_fisheye, p.type, p.max_bounces = int(fisheye), mesh_type, max_bounces
    return p


def uvw_frame(eye, lookat, up, fovy, aspect):
    L = lib()
    e, ep = _f(eye); l, lp = _f(lookat); u, up_ = _f(up)
    U = np.zeros(3, np.float32); V = np.zeros(3, np.float32); W = np.zeros(3, np.float32)
    fp = C.POINTER(C.c_float)
    L.grto_uvw_frame.argtypes = [fp, fp, fp, C.c_float, C.c_float, fp, fp, fp]
    L.grto_uvw_frame(ep, lp, up_, fovy, aspect, U.ctypes.data_as(fp), V.ctypes.data_as(fp), W.ctypes.data_as(fp))
    return U, V, W


class Scene:
    def __init__(self, particles, alpha_min=0.01):
        particles = np.ascontiguousarray(particles, dtype=PARTICLE_DTYPE)
        self._parts = particles
        self._h = lib().grto_scene_create(particles.ctypes.data, len(particles), alpha_min)

    def use_bvh(self, flag):
        lib().grto_scene_use_bvh(self._h, int(flag))

    def set_proxy_triangles(self, verts, idx):
        """Checking mode for decision (v): proxies are intersected as 20 triangles each (verts [n][12][3] world space, idx[60]) instead
        of ten slabs; verts=None switches back."""
        if verts is None:
            lib().grto_scene_set_proxy_triangles(self._h, None, None)
            return
        v = np.ascontiguousarray(verts, np.float32).reshape(len(self._parts), 12, 3)
        i = np.ascontiguousarray(idx, np.uint32).reshape(60)
        lib().grto_scene_set_proxy_triangles(self._h, v.ctypes.data, i.ctypes.data)

    def set_mesh(self, verts, normals, faces):
        v = np.ascontiguousarray(verts, np.float32); n = np.ascontiguousarray(normals, np.float32)
        f = np.ascontiguousarray(faces, np.uint32)
        lib().grto_scene_set_mesh(self._h, v.ctypes.data, n.ctypes.data, len(v), f.ctypes.data, len(f))

    def render(self, params, window=None, threads=0, want_u8=True, want_f32=True):
        w, h = params.width, params.height
        x0, y0, x1, y1 = window if window else (0, 0, w, h)
        u8 = np.zeros((h, w, 3), np.uint8) if want_u8 else None
        f32 = np.zeros((h, w, 3), np.float32) if want_f32 else None
        c = Counters()
        lib().grto_render(self._h, C.byref(params), x0, y0, x1, y1, u8.ctypes.data if want_u8 else None,
                          f32.ctypes.data if want_f32 else None, C.byref(c), threads)
        return u8, f32, c.as_dict()

    def render_rays(self, params, rays, threads=0):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros((len(rays), 3), np.float32)
        c = Counters()
        lib().grto_render_rays(self._h, C.byref(params), rays.ctypes.data, len(rays), out.ctypes.data, C.byref(c), threads)
        return out, c.as_dict()

    def trace_gps(self, o, d, tmin, tmax):
        o, op = _f(o); d, dp = _f(d)
        ids = np.zeros(7, np.uint32); ts = np.zeros(7, np.float32)
        n = lib().grto_trace_gps(self._h, op, dp, tmin, tmax, ids.ctypes.data, ts.ctypes.data)
        return n, ids, ts

    def trace(self, params, o, d, t_min, t_max, density=0.0):
        o, op = _f(o); d, dp = _f(d)
        dens = np.array([density], np.float32); rad = np.zeros(3, np.float32)
        fp = C.POINTER(C.c_float)
        lib().grto_trace(self._h, C.byref(params), op, dp, t_min, t_max, dens.ctypes.data_as(fp), rad.ctypes.data_as(fp), None)
        return rad, float(dens[0])

    def render_pixel(self, params, ix, iy):
        rgb = np.zeros(3, np.float32)
        lib().grto_render_pixel(self._h, C.byref(params), ix, iy, rgb.ctypes.data_as(C.POINTER(C.c_float)), None)
        return rgb

    def close(self):
        if self._h:
            lib().grto_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
