"""ctypes binding of libgrt_hip.so (include/grt.h) for bench.py and the tests.

PyTorch is used only as plumbing: device buffers (torch.empty on cuda:N), the current stream and
torch.distributed.  There is NO CPU fallback here: without the HIP library or a GPU every call raises.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (GRT_LIB: a diagnostic build of the same library, e.g. make EXTRA=-DGRT_TILE_DIAG — profiling only)
LIB_PATH = os.environ.get("GRT_LIB") or os.path.join(_PKG, "libgrt_hip.so")


class GrtError(RuntimeError):
    """Mirrors the reference's std::runtime_error on any CUDA/OptiX failure (src/Exception.h:19-80)."""


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("sh_degree_max", C.c_uint32),
        ("eye", C.c_float * 3), ("U", C.c_float * 3), ("V", C.c_float * 3), ("W", C.c_float * 3),
        ("t_min", C.c_float), ("t_max", C.c_float), ("minTransmittance", C.c_float), ("alpha_min", C.c_float),
        ("mode_fisheye", C.c_int32), ("type", C.c_int32), ("max_bounces", C.c_uint32),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "segments", "hit_evals", "rounds", "node_visits", "proxy_tests",
                                          "rec_fetches", "stall_exits")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class BvhInfo(C.Structure):
    _fields_ = [("n_particles", C.c_uint64), ("n_proxies", C.c_uint64), ("n_nodes", C.c_uint32),
                ("height", C.c_uint32), ("mesh_faces", C.c_uint32), ("mesh_height", C.c_uint32),
                ("build_ms", C.c_float), ("mesh_update_ms", C.c_float), ("scene_lo", C.c_float * 3), ("scene_hi", C.c_float * 3),
                ("n_primitives", C.c_uint64)]


class MemoryInfo(C.Structure):
    _fields_ = [("scene_bytes", C.c_uint64), ("slot_bytes", C.c_uint64), ("overflow_pool_bytes", C.c_uint64),
                ("overflow_chunks", C.c_uint32), ("overflow_demand", C.c_uint32)]


class Gaussians(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("pos", "scale", "quat", "opacity", "sh")]


class Mesh(C.Structure):
    _fields_ = [("verts", C.c_void_p), ("normals", C.c_void_p), ("nv", C.c_uint32), ("faces", C.c_void_p),
                ("nf", C.c_uint32)]


MIRROR, NORMAL, GLASS = 0, 1, 2
OPT_COUNTERS, OPT_KERNEL, OPT_LEAF_MAX, OPT_SWIZZLE, OPT_FEEDBACK = 1, 2, 3, 4, 5
OPT_TILE_READY_MIN, OPT_TILE_BAND, OPT_TILE_LOOKAHEAD, OPT_TILE_RESERVE, OPT_TILE_PRIO_DIV, OPT_COST_RADIUS, OPT_SIZE_CLASSES, OPT_COLD_ESTIMATE = 8, 9, 10, 11, 12, 13, 14, 15
OPT_BUNDLE_ROUNDS, OPT_BUNDLE_BUDGET, OPT_SINGLE_LOOKAHEAD, OPT_SINGLE_BAND, OPT_LANE_BUDGET = 16, 17, 18, 19, 20
OPT_OVF_CHUNKS, OPT_OVF_ENTRIES, OPT_MAX_ITERS, OPT_SPLIT, OPT_TILE_BAND_ABS = 21, 22, 23, 24, 25
OPT_TILE_PARTS2_PCT, OPT_TILE_PARTS4_PCT, OPT_TILE_PARTS_LOAD_PCT = 26, 27, 28
OPT_MESH_PARTS = 29
OPT_ORDER_MULTI_MIN = 30
OPT_STATIC_SHARP = 31
OPT_COLD_PARTS_PCT = 32
OPT_QUAD_PARTS = 33
OPT_OVF_CLASSES = 34
OPT_BVH_ROTATIONS = 35
OPT_BUNDLE_PREDICT = 36
OPT_MESH_PRIMARY_WAVE = 37
OPT_SPLIT_VOL_PCT = 38
ERR_LIMIT = -5
KERNEL_AUTO, KERNEL_PERLANE, KERNEL_WAVE, KERNEL_STREAM, KERNEL_STREAM_BIG, KERNEL_TILE = 0, 1, 2, 3, 4, 5

EXPORTS = [
    "grt_create", "grt_create_view", "grt_get_memory_info", "grt_destroy", "grt_last_error", "grt_set_option", "grt_upload_gaussians", "grt_build_bvh",
    "grt_set_meshes", "grt_update_meshes", "grt_get_bvh_info", "grt_debug_bvh_depth", "grt_render", "grt_render_tiles", "grt_assemble_tiles", "grt_render_rays", "grt_sync",
    "grt_get_counters", "grt_last_kernel_ms", "grt_host_activate", "grt_host_uvw_frame", "grt_host_synth_scene",
    "grt_host_ply_count", "grt_host_ply_read", "grt_host_ply_write", "grt_host_last_error",
    "grt_host_primitive_counts", "grt_host_primitive_fill", "grt_host_obj_count", "grt_host_obj_read", "grt_host_obj_write",
]

_lib = None


def lib():
    """Load libgrt_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        # torch ships its own HIP/HSA runtime (torch/lib/libamdhip64.so); it must be the one the process
        # initialises, so import torch BEFORE libgrt_hip.so resolves libamdhip64 (two runtimes opening the
        # KFD in one process => "no ROCm-capable device is detected").  Stand-alone C/C++ users bind to
        # /opt/rocm as usual.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise GrtError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp, u32, u64, fl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float
        L.grt_create.argtypes = [C.POINTER(vp), C.c_int]
        L.grt_create_view.argtypes = [vp, C.POINTER(vp)]
        L.grt_get_memory_info.argtypes = [vp, C.POINTER(MemoryInfo)]
        L.grt_destroy.argtypes = [vp]
        L.grt_last_error.restype = C.c_char_p
        L.grt_last_error.argtypes = [vp]
        L.grt_set_option.argtypes = [vp, C.c_int, C.c_int]
        L.grt_upload_gaussians.argtypes = [vp, C.POINTER(Gaussians), u64]
        L.grt_build_bvh.argtypes = [vp, fl]
        L.grt_set_meshes.argtypes = [vp, C.POINTER(Mesh), u32]
        L.grt_update_meshes.argtypes = [vp, C.POINTER(Mesh), u32]
        L.grt_get_bvh_info.argtypes = [vp, C.POINTER(BvhInfo)]
        L.grt_debug_bvh_depth.argtypes = [vp, C.POINTER(u32)]
        L.grt_render.argtypes = [vp, C.POINTER(Params), vp, vp, u32, u32, u32, u32, vp]
        L.grt_render_tiles.argtypes = [vp, C.POINTER(Params), vp, vp, u32, u32, u32, u32, u32, vp]
        L.grt_assemble_tiles.argtypes = [vp, vp, u32, u32, u32, u32, u32, u32, vp, vp]
        L.grt_render_rays.argtypes = [vp, C.POINTER(Params), vp, u64, vp, vp]
        L.grt_sync.argtypes = [vp]
        L.grt_sync.restype = C.c_int
        L.grt_get_counters.argtypes = [vp, C.POINTER(Counters)]
        L.grt_last_kernel_ms.argtypes = [vp, C.POINTER(fl)]
        L.grt_host_activate.argtypes = [u64] + [vp] * 11
        L.grt_host_uvw_frame.argtypes = [vp, vp, vp, fl, fl, vp, vp, vp]
        L.grt_host_uvw_frame.restype = None
        L.grt_host_synth_scene.argtypes = [u64, u64] + [vp] * 6
        L.grt_host_ply_count.argtypes = [C.c_char_p, C.POINTER(u64)]
        L.grt_host_ply_read.argtypes = [C.c_char_p, u64] + [vp] * 6
        L.grt_host_ply_write.argtypes = [C.c_char_p, u64] + [vp] * 6
        L.grt_host_last_error.restype = C.c_char_p
        L.grt_host_primitive_counts.argtypes = [C.c_int, C.POINTER(u32), C.POINTER(u32)]
        L.grt_host_primitive_fill.argtypes = [C.c_int, vp, vp, vp]
        L.grt_host_obj_count.argtypes = [C.c_char_p, C.POINTER(u32), C.POINTER(u32)]
        L.grt_host_obj_read.argtypes = [C.c_char_p, u32, vp, vp, vp]
        L.grt_host_obj_write.argtypes = [C.c_char_p, u32, vp, vp, u32, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _host_check(rc):
    if rc != 0:
        raise GrtError(f"grt host error {rc}: {lib().grt_host_last_error().decode()}")


# ---------------------------------------------------------------------------------------------
# host helpers (no GPU needed)
# ---------------------------------------------------------------------------------------------
def raw_columns(n):
    return dict(pos=np.zeros((n, 3), np.float32), f_dc=np.zeros((n, 3), np.float32),
                f_rest=np.zeros((n, 45), np.float32), opacity=np.zeros(n, np.float32),
                scale=np.zeros((n, 3), np.float32), rot=np.zeros((n, 4), np.float32))


def synth_scene(seed, n):
    """Deterministic synthetic 3DGS scene (raw PLY columns), SURVEY.md §8(d)."""
    r = raw_columns(n)
    _host_check(lib().grt_host_synth_scene(seed, n, _p(r["pos"]), _p(r["f_dc"]), _p(r["f_rest"]), _p(r["opacity"]),
                                           _p(r["scale"]), _p(r["rot"])))
    return r


def activate(raw):
    """Raw PLY columns -> activated attributes (src/GaussianData.cpp:97-131)."""
    n = len(raw["pos"])
    raw = {k: np.ascontiguousarray(v, np.float32) for k, v in raw.items()}
    out = dict(pos=np.zeros((n, 3), np.float32), scale=np.zeros((n, 3), np.float32), quat=np.zeros((n, 4), np.float32),
               opacity=np.zeros(n, np.float32), sh=np.zeros((n, 16, 3), np.float32))
    _host_check(lib().grt_host_activate(n, _p(raw["pos"]), _p(raw["f_dc"]), _p(raw["f_rest"]), _p(raw["opacity"]),
                                        _p(raw["scale"]), _p(raw["rot"]), _p(out["pos"]), _p(out["scale"]),
                                        _p(out["quat"]), _p(out["opacity"]), _p(out["sh"])))
    return out


def read_ply(path):
    n = C.c_uint64()
    _host_check(lib().grt_host_ply_count(path.encode(), C.byref(n)))
    r = raw_columns(n.value)
    _host_check(lib().grt_host_ply_read(path.encode(), n.value, _p(r["pos"]), _p(r["f_dc"]), _p(r["f_rest"]),
                                        _p(r["opacity"]), _p(r["scale"]), _p(r["rot"])))
    return r


def write_ply(path, raw):
    raw = {k: np.ascontiguousarray(v, np.float32) for k, v in raw.items()}
    _host_check(lib().grt_host_ply_write(path.encode(), len(raw["pos"]), _p(raw["pos"]), _p(raw["f_dc"]),
                                         _p(raw["f_rest"]), _p(raw["opacity"]), _p(raw["scale"]), _p(raw["rot"])))


def uvw_frame(eye, lookat, up, fovy_deg, aspect):
    e, l, u = (np.ascontiguousarray(x, np.float32) for x in (eye, lookat, up))
    U, V, W = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().grt_host_uvw_frame(_p(e), _p(l), _p(u), fovy_deg, aspect, _p(U), _p(V), _p(W))
    return U, V, W


def default_params(width, height, acts_pos_mean, sh_degree=0, fisheye=False, mesh_type=MIRROR, max_bounces=32,
                   eye=(0.0, 0.0, 3.0), fovy=60.0):
    """Reference defaults: eye (0,0,3), lookat = mean Gaussian position, up (0,1,0), fovY 60
    (src/gui.cpp:52-55); t_min 1e-3, t_max 1e5, minTransmittance 1e-3, alpha_min 0.01
    (src/GaussianTracer.cpp:479-482)."""
    U, V, W = uvw_frame(eye, acts_pos_mean, (0.0, 1.0, 0.0), fovy, float(np.float32(width) / np.float32(height)))
    p = Params()
    p.width, p.height, p.sh_degree_max = width, height, sh_degree
    for name, v in (("eye", eye), ("U", U), ("V", V), ("W", W)):
        arr = getattr(p, name)
        for k in range(3):
            arr[k] = float(v[k])
    p.t_min, p.t_max, p.minTransmittance, p.alpha_min = 1e-3, 1e5, 1e-3, 0.01
    p.mode_fisheye, p.type, p.max_bounces = int(fisheye), mesh_type, max_bounces
    return p


def gaussian_center(pos):
    """GaussianData::getCenter (src/GaussianData.cpp:139-151): sequential fp32 sum / n."""
    c = np.zeros(3, np.float32)
    for k in range(3):
        c[k] = np.cumsum(pos[:, k], dtype=np.float32)[-1] / np.float32(len(pos))
    return c


PRIM_PLANE, PRIM_SPHERE = 0, 1


def primitive_mesh(kind, center=(0.0, 0.0, 0.0)):
    """The reference's procedural plane / sphere (src/geometry/Primitives.cpp:6-140) placed by translate(center),
    from the C ABI (the same arrays the C++ facade's createPlane/createSphere hold): (verts, normals, faces)."""
    nv, nf = C.c_uint32(), C.c_uint32()
    _host_check(lib().grt_host_primitive_counts(kind, C.byref(nv), C.byref(nf)))
    v = np.zeros((nv.value, 3), np.float32); n = np.zeros((nv.value, 3), np.float32); f = np.zeros((nf.value, 3), np.uint32)
    _host_check(lib().grt_host_primitive_fill(kind, _p(v), _p(n), _p(f)))
    c = np.asarray(center, np.float32)
    return (v if not c.any() else (v + c[None]).astype(np.float32)), n, f  # (-0.0 + 0.0 would lose the sign bit)


def load_obj(path, center=(0.0, 0.0, 0.0)):
    """Primitives::createLoadMesh (src/geometry/Primitives.cpp:142-202): un-indexed soup, Y flipped, translate(center)."""
    nv, nf = C.c_uint32(), C.c_uint32()
    _host_check(lib().grt_host_obj_count(path.encode(), C.byref(nv), C.byref(nf)))
    v = np.zeros((nv.value, 3), np.float32); n = np.zeros((nv.value, 3), np.float32); f = np.zeros(nv.value, np.uint32)
    _host_check(lib().grt_host_obj_read(path.encode(), nv.value, _p(v), _p(n), _p(f)))
    return (v + np.asarray(center, np.float32)[None]).astype(np.float32), n, f.reshape(-1, 3)


def write_obj(path, verts, normals, faces):
    v = np.ascontiguousarray(verts, np.float32); n = np.ascontiguousarray(normals, np.float32)
    f = np.ascontiguousarray(faces, np.uint32)
    _host_check(lib().grt_host_obj_write(path.encode(), len(v), _p(v), _p(n), len(f), _p(f)))


def sphere_mesh(center, radius=0.3, tess_u=180, tess_v=90):
    """UV sphere with the reference's construction (src/geometry/Primitives.cpp:63-140) at any tessellation, placed by
    translate(center) — numpy formulation for the small test meshes; primitive_mesh(PRIM_SPHERE) is the reference's
    180 x 90 sphere itself."""
    f32 = np.float32
    phi_step = f32(2.0) * f32(np.pi) / f32(tess_u)
    theta_step = f32(np.pi) / f32(tess_v - 1)
    lat = np.arange(tess_v, dtype=np.float32) * theta_step
    lon = np.arange(tess_u + 1, dtype=np.float32) * phi_step
    st, ct = np.sin(lat).astype(f32), np.cos(lat).astype(f32)
    sp, cp = np.sin(lon).astype(f32), np.cos(lon).astype(f32)
    n = np.stack([np.outer(st, cp), np.repeat(ct[:, None], tess_u + 1, 1), np.outer(st, sp)], -1).astype(f32)
    normals = n.reshape(-1, 3)
    verts = (normals * f32(radius)).astype(f32) + np.asarray(center, f32)[None]
    cols = tess_u + 1
    la, lo = np.meshgrid(np.arange(tess_v - 1), np.arange(tess_u), indexing="ij")
    ll, lr = la * cols + lo, la * cols + lo + 1
    ur, ul = (la + 1) * cols + lo + 1, (la + 1) * cols + lo
    faces = np.stack([np.stack([ll, lr, ur], -1), np.stack([ur, ul, ll], -1)], 2).reshape(-1, 3).astype(np.uint32)
    return verts.astype(f32), normals.copy(), faces


def plane_mesh(center, width=0.3, height=0.5):
    """Primitives::createPlane (src/geometry/Primitives.cpp:6-61)."""
    f32 = np.float32
    c = np.asarray(center, f32)
    v = np.array([[-width / 2, -height / 2, 0], [width / 2, -height / 2, 0], [-width / 2, height / 2, 0],
                  [width / 2, height / 2, 0]], f32) + c[None]
    n = np.tile(np.array([[0, 0, 1]], f32), (4, 1))
    f = np.array([[0, 1, 3], [3, 2, 0]], np.uint32)
    return v.astype(f32), n, f


# ---------------------------------------------------------------------------------------------
# device side
# ---------------------------------------------------------------------------------------------
class Tracer:
    """One context per GPU (mirrors class GaussianTracer, src/GaussianTracer.h:27-111)."""

    def __init__(self, device=0, scene=None):
        """scene = another Tracer: this one is a VIEW of it (grt_create_view) — a frame slot of its own on that
        Tracer's Gaussians / BVHs / meshes."""
        import torch
        if not torch.cuda.is_available():
            raise GrtError("no GPU visible: libgrt_hip has no CPU fallback")
        self._torch = torch
        self.device = device if scene is None else scene.device
        self._h = C.c_void_p()
        self._scene = scene  # keeps the parent alive as long as the view
        rc = lib().grt_create(C.byref(self._h), device) if scene is None else lib().grt_create_view(scene._h, C.byref(self._h))
        if rc != 0:
            raise GrtError(f"grt_create failed ({rc}): {lib().grt_last_error(None).decode()}")

    def view(self):
        return Tracer(scene=self)

    def _check(self, rc):
        if rc != 0:
            e = GrtError(f"grt error {rc}: {lib().grt_last_error(self._h).decode()}")
            e.code = rc
            raise e

    def memory_info(self):
        o = MemoryInfo()
        self._check(lib().grt_get_memory_info(self._h, C.byref(o)))
        return {n: int(getattr(o, n)) for n, _ in o._fields_}

    def check(self):
        """grt_sync: waits for the last frame and raises (code ERR_LIMIT) when a wave gave up on live rays."""
        self._check(lib().grt_sync(self._h))

    def set_option(self, opt, val):
        self._check(lib().grt_set_option(self._h, opt, val))

    def upload(self, acts, alpha_min=0.01):
        a = {k: np.ascontiguousarray(v, np.float32) for k, v in acts.items()}
        g = Gaussians(*(a[k].ctypes.data for k in ("pos", "scale", "quat", "opacity", "sh")))
        self._check(lib().grt_upload_gaussians(self._h, C.byref(g), len(a["pos"])))
        self._check(lib().grt_build_bvh(self._h, alpha_min))

    def set_meshes(self, meshes):
        keep, arr = [], (Mesh * max(len(meshes), 1))()
        for i, (v, n, f) in enumerate(meshes):
            v = np.ascontiguousarray(v, np.float32); n = np.ascontiguousarray(n, np.float32)
            f = np.ascontiguousarray(f, np.uint32)
            keep += [v, n, f]
            arr[i] = Mesh(v.ctypes.data, n.ctypes.data, len(v), f.ctypes.data, len(f))
        self._check(lib().grt_set_meshes(self._h, arr, len(meshes)))

    def update_meshes(self, meshes):
        """Same topology, new positions / normals: the mesh LBVH is re-fitted, not rebuilt."""
        keep, arr = [], (Mesh * max(len(meshes), 1))()
        for i, (v, n, f) in enumerate(meshes):
            v = np.ascontiguousarray(v, np.float32); n = np.ascontiguousarray(n, np.float32)
            f = np.ascontiguousarray(f, np.uint32)
            keep += [v, n, f]
            arr[i] = Mesh(v.ctypes.data, n.ctypes.data, len(v), f.ctypes.data, len(f))
        self._check(lib().grt_update_meshes(self._h, arr, len(meshes)))

    def bvh_info(self):
        o = BvhInfo()
        self._check(lib().grt_get_bvh_info(self._h, C.byref(o)))
        return {n: (list(getattr(o, n)) if n.startswith("scene") else getattr(o, n)) for n, _ in o._fields_}

    def bvh_depth_walked(self):
        """(testing) depth of the Gaussian LBVH walked on the host; must be <= bvh_info()['height']."""
        d = C.c_uint32(0)
        self._check(lib().grt_debug_bvh_depth(self._h, C.byref(d)))
        return int(d.value)

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def render(self, params, window=None, want_u8=True, want_f32=False, out_u8=None, out_f32=None):
        t = self._torch
        w, h = params.width, params.height
        dev = f"cuda:{self.device}"
        if want_u8 and out_u8 is None:
            out_u8 = t.zeros((h, w, 3), dtype=t.uint8, device=dev)
        if want_f32 and out_f32 is None:
            out_f32 = t.zeros((h, w, 3), dtype=t.float32, device=dev)
        x0, y0, x1, y1 = window if window else (0, 0, w, h)
        self._check(lib().grt_render(self._h, C.byref(params), out_u8.data_ptr() if out_u8 is not None else None,
                                     out_f32.data_ptr() if out_f32 is not None else None, x0, y0, x1, y1,
                                     self._stream()))
        return out_u8, out_f32

    def render_tiles(self, params, tile_w, tile_h, first, stride, count, out_u8=None, out_f32=None):
        self._check(lib().grt_render_tiles(self._h, C.byref(params), out_u8.data_ptr() if out_u8 is not None else None,
                                           out_f32.data_ptr() if out_f32 is not None else None, tile_w, tile_h, first,
                                           stride, count, self._stream()))

    def assemble_tiles(self, gathered, world, max_cnt, tile, width, height, out_u8):
        """gathered: ONE uint8 tensor [world][max_cnt][tile][tile][3] (the ranks' compact buffers) -> out_u8 [height][width][3]."""
        self._check(lib().grt_assemble_tiles(self._h, gathered.data_ptr(), world, max_cnt, tile, tile, width, height,
                                             out_u8.data_ptr(), self._stream()))

    def render_rays(self, params, rays, out=None):
        t = self._torch
        if out is None:
            out = t.zeros((rays.shape[0], 3), dtype=t.float32, device=rays.device)
        self._check(lib().grt_render_rays(self._h, C.byref(params), rays.data_ptr(), rays.shape[0], out.data_ptr(),
                                          self._stream()))
        return out

    def counters(self):
        c = Counters()
        self._check(lib().grt_get_counters(self._h, C.byref(c)))
        return c.as_dict()

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(lib().grt_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def sync(self):
        self._torch.cuda.synchronize(self.device)

    def close(self):
        if self._h:
            lib().grt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
