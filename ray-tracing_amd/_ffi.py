"""ctypes bindings of include/rt_hip.h (device boundary) and include/rt_host.h (host scene API).
The product fails loudly when a library is missing: there is no CPU fallback."""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_HIP = os.environ.get("RT_HIP_LIB") or os.path.join(PKG, "librt_hip.so")  # RT_HIP_LIB: A/B variant builds (tools/gpu_ab.py)
LIB_HOST = os.path.join(PKG, "librt_host.so")

RT_F32, RT_F64 = 0, 1
RT_STATS_WORDS = 4
RT_RENDER_NO_SORT, RT_RENDER_DETERMINISTIC = 1, 2  # RtRenderFlags
RT_COMM_ID_BYTES = 128
RT_NOALLOC_MAX_PIXELS = 1024 * 1024


class RtRenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("spp_begin", C.c_uint32), ("spp_count", C.c_uint32),
        ("max_depth", C.c_int32), ("max_nan_retries", C.c_uint32),
        ("seed", C.c_uint64),
        ("precision", C.c_uint32), ("flags", C.c_uint32),
    ]


class RtError(RuntimeError):
    pass


_hip = None
_host = None


def _missing(path):
    raise RtError(
        f"native library {path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950). There is no CPU fallback."
    )


def hip():
    """librt_hip.so (HIP kernels + C-ABI)."""
    global _hip
    if _hip is None:
        if not os.path.exists(LIB_HIP):
            _missing(LIB_HIP)
        lib = C.CDLL(LIB_HIP)
        vp, u32, u64, i32, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_double
        sig = {
            "rt_init": ([C.c_int], C.c_int),
            "rt_last_error": ([], C.c_char_p),
            "rt_describe": ([], C.c_char_p),
            "rt_scene_create": ([vp, C.POINTER(vp)], C.c_int),
            "rt_scene_destroy": ([vp], C.c_int),
            "rt_trace_primary": ([vp, u32, u32, dbl, dbl, u32, vp, vp, vp], C.c_int),
            "rt_render": ([vp, C.POINTER(RtRenderParams), vp, vp, vp], C.c_int),
            "rt_debug_sample_pixel": ([vp, C.POINTER(RtRenderParams), u32, u32, vp, vp], C.c_int),
            "rt_resolve": ([vp, u32, u32, u32, vp, vp], C.c_int),
            "rt_render_rgb8": ([vp, C.POINTER(RtRenderParams), vp, vp, vp], C.c_int),
            "rt_render_rgb8_noalloc": ([vp, C.POINTER(RtRenderParams), vp, vp, vp], C.c_int),
            "rt_comm_unique_id": ([vp], C.c_int),
            "rt_comm_create": ([vp, C.c_int, C.c_int, C.POINTER(vp)], C.c_int),
            "rt_comm_adopt": ([vp, C.POINTER(vp)], C.c_int),
            "rt_comm_destroy": ([vp], C.c_int),
            "rt_allreduce_accum": ([vp, vp, u64, vp], C.c_int),
            "rt_allreduce_stats": ([vp, vp, vp], C.c_int),
            "rt_debug_divide": ([vp, vp, u64, vp, vp, vp], C.c_int),
            "rt_debug_divide_fast": ([vp, vp, u64, vp, vp, vp], C.c_int),
            "rt_debug_convert": ([vp, u64, vp, vp], C.c_int),
            "rt_device_alloc": ([C.POINTER(vp), u64], C.c_int),
            "rt_device_free": ([vp], C.c_int),
            "rt_device_zero": ([vp, u64, vp], C.c_int),
            "rt_copy_to_host": ([vp, vp, u64, vp], C.c_int),
            "rt_stream_synchronize": ([vp], C.c_int),
        }
        for name, (args, res) in sig.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, res
        _hip = lib
    return _hip


def host():
    """librt_host.so (scene vocabulary, scenes.rs constructors, flattener)."""
    global _host
    if _host is None:
        if not os.path.exists(LIB_HOST):
            _missing(LIB_HOST)
        lib = C.CDLL(LIB_HOST)
        vp = C.c_void_p
        lib.rth_last_error.argtypes, lib.rth_last_error.restype = [], C.c_char_p
        lib.rth_scene_build.argtypes, lib.rth_scene_build.restype = [C.c_char_p, C.c_double, C.c_uint64, C.c_char_p], vp
        lib.rth_scene_free.argtypes, lib.rth_scene_free.restype = [vp], None
        lib.rth_scene_desc.argtypes, lib.rth_scene_desc.restype = [vp], vp
        lib.rth_scene_num_world_prims.argtypes, lib.rth_scene_num_world_prims.restype = [vp], C.c_uint32
        lib.rth_scene_roundtrip_check.argtypes, lib.rth_scene_roundtrip_check.restype = [vp], C.c_int
        lib.rth_scene_names.argtypes, lib.rth_scene_names.restype = [], C.POINTER(C.c_char_p)
        _host = lib
    return _host


def check(rc):
    if rc != 0:
        raise RtError(f"rt_hip error {rc}: {hip().rt_last_error().decode()}")


# ---- mirrors of the RtSceneDesc records (tests inspect / hand-build flat scenes with these) ----
class RtPrim(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("material", C.c_uint32), ("d", C.c_double * 10)]


class RtMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("texture", C.c_uint32), ("albedo", C.c_double * 3), ("fuzz", C.c_double), ("ref_idx", C.c_double)]


class RtTexture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("odd", C.c_uint32), ("even", C.c_uint32), ("table", C.c_uint32),
                ("color", C.c_double * 3), ("scale", C.c_double)]


class RtPerlin(C.Structure):
    _fields_ = [("perm_x", C.c_int32 * 256), ("perm_y", C.c_int32 * 256), ("perm_z", C.c_int32 * 256), ("ranvec", (C.c_double * 3) * 256)]


class RtImage(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb", C.POINTER(C.c_uint8))]


class RtXform(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("reserved", C.c_uint32), ("v", C.c_double * 3)]


class RtInstance(C.Structure):
    _fields_ = [("geom_kind", C.c_uint32), ("geom_index", C.c_uint32), ("xform_begin", C.c_uint32), ("xform_count", C.c_uint32),
                ("is_medium", C.c_uint32), ("phase_material", C.c_uint32), ("density", C.c_double)]


class RtBvh(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32)] + [(n, C.POINTER(C.c_double)) for n in ("min_x", "min_y", "min_z", "max_x", "max_y", "max_z")] + \
               [("left", C.POINTER(C.c_int32)), ("right", C.POINTER(C.c_int32))]


class RtCamera(C.Structure):
    _fields_ = [(n, C.c_double * 3) for n in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w")] + \
               [("lens_radius", C.c_double), ("t1", C.c_double), ("t2", C.c_double)]


class RtSceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_prims", C.c_uint32), ("prims", C.POINTER(RtPrim)),
        ("n_materials", C.c_uint32), ("materials", C.POINTER(RtMaterial)),
        ("n_textures", C.c_uint32), ("textures", C.POINTER(RtTexture)),
        ("n_perlins", C.c_uint32), ("perlins", C.POINTER(RtPerlin)),
        ("n_images", C.c_uint32), ("images", C.POINTER(RtImage)),
        ("n_xforms", C.c_uint32), ("xforms", C.POINTER(RtXform)),
        ("n_world", C.c_uint32), ("world", C.POINTER(RtInstance)),
        ("n_lights", C.c_uint32), ("lights", C.POINTER(RtInstance)),
        ("bvh", RtBvh),
        ("background", C.c_double * 3),
        ("camera", RtCamera),
    ]
