"""ctypes binding of libpathtracer_amd.so (include/pt_api.h).

This is the only way Python reaches the GPU path.  There is no CPU fallback: if the shared
library is missing, or no GPU is visible, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT_LIB") or os.path.join(_HERE, "libpathtracer_amd.so")  # PT_LIB: a diagnostic build of the same ABI

PT_OK = 0
STATUS_NAMES = {0: "PT_OK", 1: "PT_ERR_INVALID", 2: "PT_ERR_NO_SCENE", 3: "PT_ERR_DEVICE", 4: "PT_ERR_LIMIT"}

SPHERE, XY_RECT, XZ_RECT, YZ_RECT, CONE, CYLINDER = range(6)
LAMBERTIAN, METAL, DIELECTRIC, DIFFUSE_LIGHT = range(4)
TRACE_BVH, TRACE_LINEAR, TRACE_BVH_IN_SCENE, TRACE_BVH_GEOMETRIC = 0, 1, 2, 3
RENDER_STAGE_TIMES, RENDER_NO_SORT, RENDER_COUNT_TESTS, RENDER_NO_LDS_SCENE, RENDER_SPLIT_KERNELS, RENDER_GEOMETRIC_CULL = 1, 2, 4, 8, 16, 32
KERNEL_NAMES = {0: "-", 1: "k_paths", 2: "k_paths_stream", 3: "k_raygen/k_intersect/k_shade/k_film", 4: "k_paths_ring"}
NOALLOC_SCRATCH_BYTES = 32 << 20
NO_PRIM = 0xFFFFFFFF
FLT_MAX = float(np.finfo(np.float32).max)

EXPORTS = [
    "pt_context_create", "pt_context_destroy", "pt_last_error", "pt_camera_look_at", "pt_scene_upload", "pt_scene_update",
    "pt_trace_rays", "pt_trace_rays_device", "pt_render", "pt_render_device", "pt_film_finalize",
    "pt_bvh_build_host", "pt_bvh_build_host_leaf", "pt_bvh_refit_host", "pt_bvh_wide_host", "pt_scene_ray_cone", "pt_version",
    "pt_tuning_default", "pt_context_configure", "pt_render_device_noalloc", "pt_scatter_batch",
    "pt_group_create", "pt_group_destroy", "pt_group_size", "pt_group_last_error", "pt_group_configure", "pt_group_scene_upload", "pt_group_render",
    "pt_group_render_device", "pt_group_shard", "pt_host_alloc", "pt_host_free",
]


class PtError(RuntimeError):
    def __init__(self, status, text):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), text))
        self.status = status


class PtCamera(C.Structure):
    _fields_ = [(n, C.c_float * 3) for n in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w")] + [
        ("lens_radius", C.c_float)
    ]


class PtRayCone(C.Structure):
    _fields_ = [("kappa_max", C.c_float), ("kappa_min", C.c_float), ("r_min", C.c_float), ("cluster_centre", C.c_float * 3),
                ("cluster_half_extent_sum", C.c_float), ("adaptive", C.c_uint32)]


class PtSceneDesc(C.Structure):
    _fields_ = [
        ("n_prims", C.c_uint32), ("prim_type", C.c_void_p), ("prim_param", C.c_void_p), ("prim_material", C.c_void_p),
        ("n_materials", C.c_uint32), ("mat_type", C.c_void_p), ("mat_param", C.c_void_p),
        ("camera", PtCamera), ("background", C.c_float * 3),
    ]


class PtRenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_uint32), ("sample_offset", C.c_uint32),
        ("max_depth", C.c_int32), ("seed", C.c_uint64), ("samples_per_pass", C.c_uint32), ("flags", C.c_uint32),
    ]


class PtStats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64), ("primary_rays", C.c_uint64), ("box_tests", C.c_uint64), ("prim_tests", C.c_uint64),
        ("exact_fallbacks", C.c_uint64), ("ms_total", C.c_double), ("ms_raygen", C.c_double),
        ("ms_intersect", C.c_double), ("ms_shade", C.c_double), ("ms_film", C.c_double),
        ("launches_intersect", C.c_uint32), ("passes", C.c_uint32), ("ms_paths", C.c_double), ("exact_replays", C.c_uint64),
        ("kernel", C.c_uint32), ("reserved", C.c_uint32),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


TUNING_FIELDS = ("leaf_max", "wide_nodes", "pipeline", "lds_scene", "block", "workgroups_per_cu", "slice_paths", "slice_taper",
                 "sort_by_material", "stream_serve_min", "stream_phase", "stream_tile", "stream_run", "ring_low_water", "split_queue")


class PtTuning(C.Structure):
    """pt_tuning: a negative field leaves the choice to the library."""
    _fields_ = [(n, C.c_int32) for n in TUNING_FIELDS]


HIT_DTYPE = np.dtype(
    [("t", "<f4"), ("prim", "<u4"), ("n", "<f4", (3,)), ("front", "<u4"), ("p", "<f4", (3,)), ("flags", "<u4")]
)
SCATTER_DTYPE = np.dtype([("scattered", "<u4"), ("attenuation", "<f4", (3,)), ("origin", "<f4", (3,)), ("direction", "<f4", (3,)), ("emitted", "<f4", (3,))])
NODE_DTYPE = np.dtype([("b", "<f4", (3, 4)), ("c", "<i4", (2,)), ("pad", "<u4", (2,))])  # b[axis] = (lo child 0, lo child 1, hi child 0, hi child 1)

_lib = None


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels carry their own libamdhip64 (same soname as /opt/rocm's, found through the
    wheel's RPATH under another file name).  A process that loads this library first and torch second
    ends up with two HIP runtimes, and the second one sees no GPU.  Loading the wheel's copy before
    this library makes both resolve to that one runtime, whichever is imported first."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass  # the system runtime is used; importing torch later in this process will then fail to see the GPU


def lib():
    """Loads the shared library; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libpathtracer_amd.so is missing (%s): build it with `make -C pathtracer_amd/csrc`; "
                "there is no CPU fallback" % LIB_PATH
            )
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.pt_last_error.restype = C.c_char_p
        L.pt_last_error.argtypes = [C.c_void_p]
        L.pt_version.restype = C.c_char_p
        L.pt_context_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.pt_context_destroy.argtypes = [C.c_void_p]
        L.pt_context_destroy.restype = None
        L.pt_scene_upload.argtypes = [C.c_void_p, C.POINTER(PtSceneDesc)]
        L.pt_scene_update.argtypes = [C.c_void_p, C.POINTER(PtSceneDesc)]
        L.pt_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_void_p]
        L.pt_trace_rays_device.argtypes = L.pt_trace_rays.argtypes + [C.c_void_p]
        L.pt_render.argtypes = [C.c_void_p, C.POINTER(PtRenderParams), C.c_void_p, C.POINTER(PtStats)]
        L.pt_render_device.argtypes = [C.c_void_p, C.POINTER(PtRenderParams), C.c_void_p, C.c_void_p, C.POINTER(PtStats)]
        L.pt_film_finalize.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.pt_camera_look_at.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(PtCamera)]
        L.pt_bvh_build_host.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]
        L.pt_scene_ray_cone.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(PtRayCone)]
        L.pt_bvh_wide_host.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p]
        L.pt_bvh_refit_host.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(PtSceneDesc), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]
        L.pt_bvh_build_host_leaf.argtypes = [C.POINTER(PtSceneDesc), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]
        L.pt_scatter_batch.argtypes = [C.c_void_p] + [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p]
        L.pt_group_create.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        L.pt_group_destroy.argtypes = [C.c_void_p]
        L.pt_group_destroy.restype = None
        L.pt_group_size.argtypes = [C.c_void_p]
        L.pt_group_last_error.argtypes = [C.c_void_p]
        L.pt_group_last_error.restype = C.c_char_p
        L.pt_group_configure.argtypes = [C.c_void_p, C.POINTER(PtTuning)]
        L.pt_group_scene_upload.argtypes = [C.c_void_p, C.POINTER(PtSceneDesc)]
        L.pt_group_render.argtypes = [C.c_void_p, C.POINTER(PtRenderParams), C.c_void_p, C.POINTER(PtStats)]
        L.pt_group_render_device.argtypes = [C.c_void_p, C.POINTER(PtRenderParams), C.POINTER(C.c_void_p), C.POINTER(PtStats)]
        L.pt_group_shard.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.pt_host_alloc.argtypes = [C.c_size_t]
        L.pt_host_alloc.restype = C.c_void_p
        L.pt_host_free.argtypes = [C.c_void_p]
        L.pt_host_free.restype = None
        L.pt_tuning_default.argtypes = [C.POINTER(PtTuning)]
        L.pt_tuning_default.restype = None
        L.pt_context_configure.argtypes = [C.c_void_p, C.POINTER(PtTuning)]
        L.pt_render_device_noalloc.argtypes = [C.c_void_p, C.POINTER(PtRenderParams), C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def version():
    return lib().pt_version().decode()


def _f3(a):
    return (C.c_float * 3)(*[float(x) for x in a])


def camera_look_at(lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist):
    """camera::camera (camera.h:38-62) through the library."""
    cam = PtCamera()
    st = lib().pt_camera_look_at(_f3(lookfrom), _f3(lookat), _f3(vup), vfov, aspect, aperture, focus_dist, C.byref(cam))
    if st != PT_OK:
        raise PtError(st, "pt_camera_look_at")
    return cam


def camera_fields(cam):
    d = {k: np.array(list(getattr(cam, k)), dtype=np.float32) for k in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w")}
    d["lens_radius"] = np.float32(cam.lens_radius)
    return d


class SceneDesc:
    """Owns the flattened arrays of a scene and exposes them as a pt_scene_desc."""

    def __init__(self, prim_type, prim_param, prim_material, mat_type, mat_param, camera, background=(0, 0, 0)):
        self.prim_type = np.ascontiguousarray(prim_type, dtype=np.uint8)
        self.prim_param = np.ascontiguousarray(prim_param, dtype=np.float32).reshape(-1, 6)
        self.prim_material = np.ascontiguousarray(prim_material, dtype=np.uint32)
        self.mat_type = np.ascontiguousarray(mat_type, dtype=np.uint8)
        self.mat_param = np.ascontiguousarray(mat_param, dtype=np.float32).reshape(-1, 4)
        if not (len(self.prim_type) == len(self.prim_param) == len(self.prim_material)):
            raise ValueError("object arrays differ in length")
        if len(self.mat_type) != len(self.mat_param):
            raise ValueError("material arrays differ in length")
        d = PtSceneDesc()
        d.n_prims = len(self.prim_type)
        d.prim_type = self.prim_type.ctypes.data
        d.prim_param = self.prim_param.ctypes.data
        d.prim_material = self.prim_material.ctypes.data
        d.n_materials = len(self.mat_type)
        d.mat_type = self.mat_type.ctypes.data
        d.mat_param = self.mat_param.ctypes.data
        d.camera = camera
        d.background = _f3(background)
        self.c = d


def bvh_build_host(desc, leaf_max=None):
    """Builds the tree pt_scene_upload builds (or one with leaves of up to leaf_max objects), on the host;
    returns (nodes, order, max_depth)."""
    n_nodes, depth = C.c_uint32(0), C.c_uint32(0)

    def call(nodes_p, order_p):
        if leaf_max is None:
            return lib().pt_bvh_build_host(C.byref(desc.c), C.byref(n_nodes), C.byref(depth), nodes_p, order_p)
        return lib().pt_bvh_build_host_leaf(C.byref(desc.c), int(leaf_max), C.byref(n_nodes), C.byref(depth), nodes_p, order_p)

    st = call(None, None)
    if st != PT_OK:
        raise PtError(st, "pt_bvh_build_host")
    nodes = np.zeros(n_nodes.value, dtype=NODE_DTYPE)
    order = np.zeros(desc.c.n_prims, dtype=np.uint32)
    st = call(nodes.ctypes.data, order.ctypes.data)
    if st != PT_OK:
        raise PtError(st, "pt_bvh_build_host")
    return nodes, order, depth.value


def bvh_refit_host(before, after):
    """The tree pt_scene_update leaves: built for `before`, refitted to `after`; returns (nodes, order, max_depth)."""
    n_nodes, depth = C.c_uint32(0), C.c_uint32(0)
    st = lib().pt_bvh_refit_host(C.byref(before.c), C.byref(after.c), C.byref(n_nodes), C.byref(depth), None, None)
    if st != PT_OK:
        raise PtError(st, "pt_bvh_refit_host")
    nodes = np.zeros(n_nodes.value, dtype=NODE_DTYPE)
    order = np.zeros(after.c.n_prims, dtype=np.uint32)
    st = lib().pt_bvh_refit_host(C.byref(before.c), C.byref(after.c), C.byref(n_nodes), C.byref(depth), nodes.ctypes.data, order.ctypes.data)
    if st != PT_OK:
        raise PtError(st, "pt_bvh_refit_host")
    return nodes, order, depth.value


WIDE_NODE_DTYPE = np.dtype([("origin", "<f4", (3,)), ("exp", "u1", (3,)), ("n_children", "u1"), ("qlo", "u1", (3, 4)), ("qhi", "u1", (3, 4)),
                            ("c", "<i4", (4,)), ("pad", "<u4", (2,))])
WIDE_ABSENT = -(2 ** 31) + 1


def bvh_wide_host(desc):
    """The four-wide, quantised form of the tree (pt_bvh_wide_host); returns (records, max_depth)."""
    n_nodes, depth = C.c_uint32(0), C.c_uint32(0)
    st = lib().pt_bvh_wide_host(C.byref(desc.c), C.byref(n_nodes), C.byref(depth), None)
    if st != PT_OK:
        raise PtError(st, "pt_bvh_wide_host")
    nodes = np.zeros(n_nodes.value, dtype=WIDE_NODE_DTYPE)
    st = lib().pt_bvh_wide_host(C.byref(desc.c), C.byref(n_nodes), C.byref(depth), nodes.ctypes.data)
    if st != PT_OK:
        raise PtError(st, "pt_bvh_wide_host")
    return nodes, depth.value


def scene_ray_cone(desc):
    """pt_scene_ray_cone as a dict (host only)."""
    out = PtRayCone()
    st = lib().pt_scene_ray_cone(C.byref(desc.c), C.byref(out))
    if st != PT_OK:
        raise PtError(st, "pt_scene_ray_cone")
    return {"kappa_max": out.kappa_max, "kappa_min": out.kappa_min, "r_min": out.r_min,
            "cluster_centre": tuple(out.cluster_centre), "cluster_half_extent_sum": out.cluster_half_extent_sum,
            "adaptive": bool(out.adaptive)}


def group_shard(spp, sample_offset, n_ranks, rank):
    """pt_group_shard: (first absolute sample, samples) of GPU `rank` of `n_ranks` in pt_group_render (host arithmetic)."""
    off, cnt = C.c_uint32(0), C.c_uint32(0)
    st = lib().pt_group_shard(spp, sample_offset, n_ranks, rank, C.byref(off), C.byref(cnt))
    if st != PT_OK:
        raise PtError(st, "pt_group_shard(spp=%d, n_ranks=%d, rank=%d)" % (spp, n_ranks, rank))
    return off.value, cnt.value


class PinnedFilm:
    """A (height, width, 3) float32 array in page-locked host memory (pt_host_alloc): pt_render and pt_group_render
    fill it at the link's rate.  Keep the object alive while the array is in use; close() frees the memory."""

    def __init__(self, height, width):
        n = height * width * 3
        self._p = lib().pt_host_alloc(n * 4)
        if not self._p:
            raise MemoryError("pt_host_alloc(%d bytes) failed" % (n * 4))
        self.array = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_float)), shape=(n,)).reshape(height, width, 3)

    def close(self):
        if self._p:
            self.array = None
            lib().pt_host_free(self._p)
            self._p = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def film_finalize(rgb_sum, total_spp):
    """main.cu:95-98 and :192-205: returns (gamma image bottom-up float32, 8-bit image top-down)."""
    rgb_sum = np.ascontiguousarray(rgb_sum, dtype=np.float32)
    h, w, _ = rgb_sum.shape
    gamma = np.zeros((h, w, 3), dtype=np.float32)
    rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
    st = lib().pt_film_finalize(rgb_sum.ctypes.data, w, h, total_spp, gamma.ctypes.data, rgb8.ctypes.data)
    if st != PT_OK:
        raise PtError(st, "pt_film_finalize")
    return gamma, rgb8


class Context:
    """One GPU, one scene at a time (pt_context)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        st = lib().pt_context_create(device, C.byref(self._h))
        if st != PT_OK:
            raise PtError(st, "pt_context_create(device=%d): no usable GPU" % device)

    def close(self):
        if self._h:
            lib().pt_context_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != PT_OK:
            raise PtError(st, lib().pt_last_error(self._h).decode())

    def upload(self, desc):
        self._check(lib().pt_scene_upload(self._h, C.byref(desc.c)))

    def configure(self, **fields):
        """pt_context_configure: scheduling / tree-shape parameters (TUNING_FIELDS); unnamed ones go back to the
        library's own choice.  leaf_max and wide_nodes apply from the next upload()."""
        t = PtTuning()
        lib().pt_tuning_default(C.byref(t))
        for k, v in fields.items():
            if k not in TUNING_FIELDS:
                raise ValueError("unknown tuning field %r" % k)
            setattr(t, k, int(v))
        self._check(lib().pt_context_configure(self._h, C.byref(t)))

    def tuned(self, **fields):
        """Context manager: configure(**fields) inside, the library's defaults again afterwards."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            self.configure(**fields)
            try:
                yield self
            finally:
                self.configure()

        return cm()

    def update(self, desc):
        """pt_scene_update: a changed scene with the same number of objects; the tree is refitted, not rebuilt."""
        self._check(lib().pt_scene_update(self._h, C.byref(desc.c)))

    def trace(self, o, d, t_min=0.001, t_max=FLT_MAX, mode=TRACE_BVH):
        o = np.ascontiguousarray(o, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, dtype=np.float32).reshape(-1, 3)
        if len(o) != len(d):
            raise ValueError("origins and directions differ in length")
        out = np.zeros(len(o), dtype=HIT_DTYPE)
        self._check(lib().pt_trace_rays(self._h, o.ctypes.data, d.ctypes.data, len(o), t_min, t_max, mode, out.ctypes.data))
        return out

    def scatter(self, o, d, t, prim, sphere_pt, xi):
        """pt_scatter_batch: material::scatter / emitted for given hits with the random numbers handed in."""
        o = np.ascontiguousarray(o, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, dtype=np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(t, dtype=np.float32).reshape(-1)
        prim = np.ascontiguousarray(prim, dtype=np.uint32).reshape(-1)
        sphere_pt = np.ascontiguousarray(sphere_pt, dtype=np.float32).reshape(-1, 3)
        xi = np.ascontiguousarray(xi, dtype=np.float32).reshape(-1)
        n = len(t)
        if not (len(o) == len(d) == len(prim) == len(sphere_pt) == len(xi) == n):
            raise ValueError("inputs differ in length")
        out = np.zeros(n, dtype=SCATTER_DTYPE)
        self._check(lib().pt_scatter_batch(self._h, o.ctypes.data, d.ctypes.data, t.ctypes.data, prim.ctypes.data, sphere_pt.ctypes.data,
                                           xi.ctypes.data, n, out.ctypes.data))
        return out

    def trace_device(self, d_o, d_d, n, d_out, stream=0, t_min=0.001, t_max=FLT_MAX, mode=TRACE_BVH):
        self._check(lib().pt_trace_rays_device(self._h, d_o, d_d, n, t_min, t_max, mode, d_out, stream))

    @staticmethod
    def params(width, height, spp, max_depth, seed=1984, sample_offset=0, samples_per_pass=0, flags=0):
        p = PtRenderParams()
        p.width, p.height, p.spp, p.sample_offset = width, height, spp, sample_offset
        p.max_depth, p.seed, p.samples_per_pass, p.flags = max_depth, seed, samples_per_pass, flags
        return p

    def render(self, width, height, spp, max_depth, seed=1984, sample_offset=0, samples_per_pass=0, flags=0, out=None):
        """Returns (rgb_sum[h, w, 3] float32 linear sums, stats dict).  `out`: a C-contiguous float32 (h, w, 3) array to
        fill instead of a new one (a PinnedFilm's array arrives at the link's rate)."""
        p = self.params(width, height, spp, max_depth, seed, sample_offset, samples_per_pass, flags)
        if out is None:
            out = np.zeros((height, width, 3), dtype=np.float32)
        elif out.shape != (height, width, 3) or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float32 array of shape (height, width, 3)")
        stats = PtStats()
        self._check(lib().pt_render(self._h, C.byref(p), out.ctypes.data, C.byref(stats)))
        return out, stats.as_dict()

    def render_device(self, params, d_rgb_sum, stream=0, want_stats=False):
        """Adds into a device buffer (e.g. a torch tensor's data_ptr()) on `stream`."""
        stats = PtStats() if want_stats else None
        self._check(lib().pt_render_device(self._h, C.byref(params), d_rgb_sum, stream, C.byref(stats) if want_stats else None))
        return stats.as_dict() if want_stats else None

    def render_device_noalloc(self, params, d_rgb_sum, stream=0):
        """pt_render_device_noalloc: launches and memsets only; scratch in the library's own per-GPU array."""
        self._check(lib().pt_render_device_noalloc(self._h, C.byref(params), d_rgb_sum, stream))


class Group:
    """pt_group: one image on several GPUs of this node in one process (samples cut per GPU, one RCCL all-reduce)."""

    def __init__(self, devices):
        devices = list(devices)
        self._h = C.c_void_p()
        ids = (C.c_int * len(devices))(*devices)
        st = lib().pt_group_create(len(devices), ids, C.byref(self._h))
        if st != PT_OK:
            raise PtError(st, "pt_group_create(%r): GPUs or RCCL not usable" % (devices,))

    def close(self):
        if self._h:
            lib().pt_group_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, st):
        if st != PT_OK:
            raise PtError(st, lib().pt_group_last_error(self._h).decode())

    def size(self):
        return int(lib().pt_group_size(self._h))

    def upload(self, desc):
        self._check(lib().pt_group_scene_upload(self._h, C.byref(desc.c)))

    def render(self, width, height, spp, max_depth, seed=1984, sample_offset=0, flags=0, out=None):
        """Returns (rgb_sum[h, w, 3] float32 linear sums over all GPUs' samples, stats dict)."""
        p = Context.params(width, height, spp, max_depth, seed, sample_offset, 0, flags)
        if out is None:
            out = np.zeros((height, width, 3), dtype=np.float32)
        elif out.shape != (height, width, 3) or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float32 array of shape (height, width, 3)")
        stats = PtStats()
        self._check(lib().pt_group_render(self._h, C.byref(p), out.ctypes.data, C.byref(stats)))
        return out, stats.as_dict()

    def render_device(self, width, height, spp, max_depth, seed=1984, sample_offset=0, flags=0):
        """pt_group_render_device: returns (device address of GPU 0's buffer of the summed image, stats dict); the buffer
        belongs to the group and is valid until the next call on it."""
        p = Context.params(width, height, spp, max_depth, seed, sample_offset, 0, flags)
        ptr = C.c_void_p()
        stats = PtStats()
        self._check(lib().pt_group_render_device(self._h, C.byref(p), C.byref(ptr), C.byref(stats)))
        return ptr.value, stats.as_dict()
