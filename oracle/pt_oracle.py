"""ctypes binding of the CPU oracle (oracle/pt_oracle.c).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under pathtracer_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpt_oracle.so")


class OrcScene(C.Structure):
    _fields_ = [
        ("n_prims", C.c_uint32),
        ("prim_type", C.c_void_p),
        ("prim_param", C.c_void_p),
        ("prim_material", C.c_void_p),
        ("n_materials", C.c_uint32),
        ("mat_type", C.c_void_p),
        ("mat_param", C.c_void_p),
        ("cam_origin", C.c_float * 3),
        ("cam_lower_left", C.c_float * 3),
        ("cam_horizontal", C.c_float * 3),
        ("cam_vertical", C.c_float * 3),
        ("cam_u", C.c_float * 3),
        ("cam_v", C.c_float * 3),
        ("cam_w", C.c_float * 3),
        ("cam_lens_radius", C.c_float),
        ("background", C.c_float * 3),
    ]


HIT_DTYPE = np.dtype(
    [("t", "<f4"), ("prim", "<u4"), ("n", "<f4", (3,)), ("front", "<u4"), ("p", "<f4", (3,)), ("nan_flag", "<u4")]
)

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_render.restype = C.c_uint64
        L.orc_render_region.restype = C.c_uint64
        L.orc_render_bvh.restype = C.c_uint64
        L.orc_u32_to_uniform.restype = C.c_float
        L.orc_u32_to_uniform.argtypes = [C.c_uint32]
        L.orc_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _f3(a):
    return (C.c_float * 3)(*[float(x) for x in a])


class Scene:
    """Holds the flattened arrays alive and exposes the orc_scene struct."""

    def __init__(self, prim_type, prim_param, prim_material, mat_type, mat_param, background=(0, 0, 0)):
        self.prim_type = np.ascontiguousarray(prim_type, dtype=np.uint8)
        self.prim_param = np.ascontiguousarray(prim_param, dtype=np.float32).reshape(-1, 6)
        self.prim_material = np.ascontiguousarray(prim_material, dtype=np.uint32)
        self.mat_type = np.ascontiguousarray(mat_type, dtype=np.uint8)
        self.mat_param = np.ascontiguousarray(mat_param, dtype=np.float32).reshape(-1, 4)
        s = OrcScene()
        s.n_prims = len(self.prim_type)
        s.prim_type = self.prim_type.ctypes.data
        s.prim_param = self.prim_param.ctypes.data
        s.prim_material = self.prim_material.ctypes.data
        s.n_materials = len(self.mat_type)
        s.mat_type = self.mat_type.ctypes.data
        s.mat_param = self.mat_param.ctypes.data
        s.background = _f3(background)
        self.c = s

    def set_camera(self, lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist):
        lib().orc_camera_make(
            _f3(lookfrom), _f3(lookat), _f3(vup), C.c_float(vfov), C.c_float(aspect), C.c_float(aperture),
            C.c_float(focus_dist), C.byref(self.c),
        )

    def camera_fields(self):
        c = self.c
        return {
            k: np.array(list(getattr(c, "cam_" + k)), dtype=np.float32)
            for k in ("origin", "lower_left", "horizontal", "vertical", "u", "v", "w")
        } | {"lens_radius": np.float32(c.cam_lens_radius)}

    # ---- queries
    def trace(self, o, d, t_min=0.001, t_max=np.finfo(np.float32).max, threads=0):
        o = np.ascontiguousarray(o, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(len(o), dtype=HIT_DTYPE)
        lib().orc_trace_batch(
            C.byref(self.c), C.c_void_p(o.ctypes.data), C.c_void_p(d.ctypes.data), C.c_size_t(len(o)),
            C.c_float(t_min), C.c_float(t_max), C.c_void_p(out.ctypes.data), C.c_int(threads),
        )
        return out

    def hit_prim(self, i, o, d, t_min=0.001, t_max=np.finfo(np.float32).max):
        out = np.zeros(1, dtype=HIT_DTYPE)
        h = lib().orc_hit_prim(
            C.byref(self.c), C.c_uint32(i), _f3(o), _f3(d), C.c_float(t_min), C.c_float(t_max),
            C.c_void_p(out.ctypes.data),
        )
        return bool(h), out[0]

    def scatter(self, mat, ray_o, ray_d, rec, sphere_pt=(0, 0, 0), xi=1.0):
        rec_arr = np.zeros(1, dtype=HIT_DTYPE)
        rec_arr[0] = rec
        atten, so, sd = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
        r = lib().orc_scatter(
            C.byref(self.c), C.c_uint32(mat), _f3(ray_o), _f3(ray_d), C.c_void_p(rec_arr.ctypes.data),
            _f3(sphere_pt), C.c_float(xi), atten, so, sd,
        )
        f = lambda a: np.array(list(a), dtype=np.float32)
        return bool(r), f(atten), f(so), f(sd)

    def get_ray(self, u, v, disk_pt=(0, 0)):
        o, d = (C.c_float * 3)(), (C.c_float * 3)()
        lib().orc_get_ray(C.byref(self.c), C.c_float(u), C.c_float(v), (C.c_float * 2)(*disk_pt), o, d)
        return np.array(list(o), dtype=np.float32), np.array(list(d), dtype=np.float32)

    def render(self, width, height, spp, depth, seed=1984, sample0=0, threads=0, region=None):
        if region is None:
            region = (0, 0, width, height)
        x0, y0, w, h = region
        out = np.zeros((h, w, 3), dtype=np.float32)
        rays = lib().orc_render_region(
            C.byref(self.c), C.c_int(width), C.c_int(height), C.c_int(x0), C.c_int(y0), C.c_int(w), C.c_int(h),
            C.c_uint32(sample0), C.c_uint32(spp), C.c_int(depth), C.c_uint64(seed), C.c_void_p(out.ctypes.data),
            C.c_int(threads),
        )
        return out, int(rays)


    def render_bvh(self, nodes, order, width, height, spp, depth, seed=1984, sample0=0, threads=0):
        """Timing baseline only: the same walk with nearest hits found through a tree (nodes: the 64-byte records of
        the GPU library's host-side build, order: leaf slot -> list index).  Not the reference's algorithm."""
        nodes = np.ascontiguousarray(nodes)
        order = np.ascontiguousarray(order, dtype=np.uint32)
        out = np.zeros((height, width, 3), dtype=np.float32)
        rays = lib().orc_render_bvh(
            C.byref(self.c), C.c_void_p(nodes.ctypes.data), C.c_uint32(len(nodes)), C.c_void_p(order.ctypes.data),
            C.c_int(width), C.c_int(height), C.c_uint32(sample0), C.c_uint32(spp), C.c_int(depth), C.c_uint64(seed),
            C.c_void_p(out.ctypes.data), C.c_int(threads),
        )
        return out, int(rays)


PHILOX_ROUNDS = 7  # ORC_PHILOX_ROUNDS: what the renderer's stream uses


def count_tree_work(scene, nodes, order, o, d, t_min=0.001, t_max=np.finfo(np.float32).max):
    """(box tests, object tests) of a plain stack traversal of the given tree over the rays (roofline bookkeeping)."""
    nodes = np.ascontiguousarray(nodes)
    order = np.ascontiguousarray(order, dtype=np.uint32)
    o = np.ascontiguousarray(o, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(d, dtype=np.float32).reshape(-1, 3)
    nb, npr = C.c_uint64(0), C.c_uint64(0)
    lib().orc_count_tree_work(
        C.byref(scene.c), C.c_void_p(nodes.ctypes.data), C.c_uint32(len(nodes)), C.c_void_p(order.ctypes.data),
        C.c_void_p(o.ctypes.data), C.c_void_p(d.ctypes.data), C.c_size_t(len(o)), C.c_float(t_min), C.c_float(t_max),
        C.byref(nb), C.byref(npr),
    )
    return int(nb.value), int(npr.value)


def philox(ctr, key, rounds=10):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32(c, k, C.c_int(rounds), o)
    return [int(x) for x in o]


def u32_to_uniform(x):
    return float(lib().orc_u32_to_uniform(C.c_uint32(x)))


def finalize(rgb_sum, spp):
    rgb_sum = np.ascontiguousarray(rgb_sum, dtype=np.float32)
    h, w, _ = rgb_sum.shape
    gamma = np.zeros((h, w, 3), dtype=np.float32)
    rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
    lib().orc_finalize(
        C.c_void_p(rgb_sum.ctypes.data), C.c_int(w), C.c_int(h), C.c_uint32(spp), C.c_void_p(gamma.ctypes.data),
        C.c_void_p(rgb8.ctypes.data),
    )
    return gamma, rgb8


def num_threads():
    return int(lib().orc_num_threads())
