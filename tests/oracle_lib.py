"""ctypes wrapper of the CPU oracle (oracle/_build/liboracle.so) — the CHECKER for tests, smoke()
and bench.py's cpu_baseline leg.  Never imported by the product package."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
EARTHMAP = os.path.join(ROOT, "ray-tracing_amd", "assets", "earthmap.ppm")
DEFAULT_SEED = 0x5EED0001

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError(f"{LIB} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(LIB)
        vp, u32, u64, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_double
        L.orc_last_error.restype = C.c_char_p
        L.orc_scene_build.argtypes, L.orc_scene_build.restype = [C.c_char_p, dbl, u64, C.c_char_p], vp
        L.orc_scene_free.argtypes = [vp]
        L.orc_scene_num_prims.argtypes = [vp]
        L.orc_scene_num_lights.argtypes = [vp]
        L.orc_scene_num_world.argtypes = [vp]
        L.orc_scene_camera.argtypes = [vp, vp]
        L.orc_trace_primary.argtypes = [vp, u32, u32, dbl, dbl, vp, vp]
        L.orc_render.argtypes = [vp, u32, u32, u32, u32, C.c_int, u64, C.c_int, vp, vp]
        L.orc_render_ex.argtypes = [vp, u32, u32, u32, u32, C.c_int, u64, C.c_int, C.c_int, vp, vp]
        L.orc_sample_pixel.argtypes = [vp, u32, u32, u32, u32, u32, u64, C.c_int, vp]
        L.orc_resolve.argtypes = [vp, u32, u32, u32, vp]
        L.orc_bvh_dump.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.orc_bvh_dump.restype = C.c_int
        L.orc_scene_from_desc.argtypes, L.orc_scene_from_desc.restype = [vp], vp
        L.orc_from_desc_error.restype = C.c_char_p
        _lib = L
    return _lib


COUNTER_NAMES = ("paths", "casts", "prim_tests", "node_visits", "nan_retries", "misses", "depth_exhausted", "flops",
                 "lambertian", "metal", "dielectric", "light", "isotropic", "frosted", "dropped")


class OracleScene:
    def __init__(self, name, aspect_ratio=1.0, seed=DEFAULT_SEED, image_path=None):
        self._h = lib().orc_scene_build(name.encode(), float(aspect_ratio), int(seed), (image_path or EARTHMAP).encode())
        if not self._h:
            raise RuntimeError(lib().orc_last_error().decode())
        self.name = name

    @classmethod
    def from_desc(cls, desc_ptr, name="flat"):
        """Oracle scene graph rebuilt from a flattened description (oracle/src/from_desc.cpp)."""
        self = cls.__new__(cls)
        self._h = lib().orc_scene_from_desc(desc_ptr)
        if not self._h:
            raise RuntimeError(lib().orc_from_desc_error().decode())
        self.name = name
        return self

    def close(self):
        if self._h:
            lib().orc_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_prims(self):
        return lib().orc_scene_num_prims(self._h)

    def camera(self):
        out = np.zeros(24)
        lib().orc_scene_camera(self._h, out.ctypes.data)
        return out

    def trace_primary(self, width, height, time=0.0, medium_xi=0.5):
        ids = np.zeros((height, width), dtype=np.int32)
        ts = np.zeros((height, width), dtype=np.float64)
        lib().orc_trace_primary(self._h, width, height, float(time), float(medium_xi), ids.ctypes.data, ts.ctypes.data)
        return ids, ts

    def render(self, width, height, spp_begin, spp_count, seed=DEFAULT_SEED, max_depth=50, threads=8, accum=None, max_nan_retries=64):
        if accum is None:
            accum = np.zeros((height, width, 3), dtype=np.float64)
        cnt = np.zeros(16)
        lib().orc_render_ex(self._h, width, height, spp_begin, spp_count, max_depth, int(seed), threads, int(max_nan_retries),
                            accum.ctypes.data, cnt.ctypes.data)
        return accum, dict(zip(COUNTER_NAMES, cnt.tolist()))

    def sample_pixel(self, width, height, x, y, spp_begin, spp_count, seed=DEFAULT_SEED, max_depth=50):
        out = np.zeros((spp_count, 3))
        one = np.zeros(3)
        for i in range(spp_count):
            lib().orc_sample_pixel(self._h, width, height, x, y, spp_begin + i, int(seed), max_depth, one.ctypes.data)
            out[i] = one
        return out

    def bvh_dump(self):
        n = lib().orc_bvh_dump(self._h, 0, None, None, None, None)
        boxes = np.zeros((n, 6))
        children = np.zeros((n, 2), dtype=np.int32)
        roots = np.zeros(8, dtype=np.int32)
        nr = C.c_int()
        lib().orc_bvh_dump(self._h, n, boxes.ctypes.data, children.ctypes.data, roots.ctypes.data, C.byref(nr))
        return boxes, children, roots[: nr.value]


def resolve(accum, spp):
    h, w = accum.shape[:2]
    accum = np.ascontiguousarray(accum, dtype=np.float64)
    rgb = np.zeros((h, w, 3), dtype=np.uint8)
    lib().orc_resolve(accum.ctypes.data, w, h, spp, rgb.ctypes.data)
    return rgb
