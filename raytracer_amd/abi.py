"""ctypes view of the C ABI in include/rt_abi.h (librt_amd.so)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB: diagnostic builds of the same library (profiling counters, tuning variants)
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(HERE, "csrc", "librt_amd.so")

RT_MAX_LIGHTS = 16
RT_MAX_DEPTH = 5
RT_MAX_XFORM_NEST = 8
RT_MAX_PATTERN_NEST = 8
RT_TILE = 8
RT_STATIC_MAX_TILES = 1 << 19
RT_OK, RT_ERR_INVALID, RT_ERR_HIP, RT_ERR_LIMIT, RT_ERR_NO_DEVICE = 0, -1, -2, -3, -4
RT_NODE_CASTS_SHADOW, RT_NODE_CLOSED = 1, 2
RT_MAX_CSG_NEST = 8
RT_MAX_CSG_HITS = 128
RT_NODE_CSG_DIFFERENCE, RT_NODE_CSG_INTERSECTION, RT_NODE_CSG_UNION, RT_NODE_CSG_OP_MASK = 0 << 4, 1 << 4, 2 << 4, 3 << 4

c_d = C.c_double
i32 = C.c_int32


class rt_node(C.Structure):
    _fields_ = [(n, i32) for n in ("type", "flags", "parent", "skip", "material", "xform", "data", "ctx")]


class rt_xform(C.Structure):
    _fields_ = [("inv", c_d * 12), ("inv_t", c_d * 9), ("reserved", c_d * 3)]


class rt_aabb(C.Structure):
    _fields_ = [("min", c_d * 3), ("max", c_d * 3)]


class rt_range(C.Structure):
    _fields_ = [("minimum", c_d), ("maximum", c_d)]


class rt_tri(C.Structure):
    _fields_ = [("p1", c_d * 3), ("e1", c_d * 3), ("e2", c_d * 3)]


class rt_tri_normals(C.Structure):
    _fields_ = [("n1", c_d * 3), ("n2", c_d * 3), ("n3", c_d * 3)]


class rt_material(C.Structure):
    _fields_ = [(n, c_d) for n in ("ambient", "diffuse", "specular", "shininess", "reflective",
                                   "transparency", "refractive_index")] + [("pattern", i32), ("reserved", i32)]


class rt_pattern(C.Structure):
    _fields_ = [("type", i32), ("a", i32), ("b", i32), ("xform", i32), ("color", c_d * 3), ("scale", c_d)]


class rt_light(C.Structure):
    _fields_ = [("position", c_d * 3), ("intensity", c_d * 3)]


class rt_scene_desc(C.Structure):
    _fields_ = [
        ("n_nodes", i32), ("nodes", C.POINTER(rt_node)),
        ("n_xforms", i32), ("xforms", C.POINTER(rt_xform)),
        ("n_aabbs", i32), ("aabbs", C.POINTER(rt_aabb)),
        ("n_ranges", i32), ("ranges", C.POINTER(rt_range)),
        ("n_tris", i32), ("tris", C.POINTER(rt_tri)), ("tri_normals", C.POINTER(rt_tri_normals)),
        ("n_materials", i32), ("materials", C.POINTER(rt_material)),
        ("n_patterns", i32), ("patterns", C.POINTER(rt_pattern)),
        ("n_pattern_xforms", i32), ("pattern_xforms", C.POINTER(c_d)),
        ("n_lights", i32), ("lights", C.POINTER(rt_light)),
    ]


class rt_camera(C.Structure):
    _fields_ = [("width", i32), ("height", i32), ("half_width", c_d), ("half_height", c_d),
                ("pixel_size", c_d), ("inv", c_d * 12)]


class rt_params(C.Structure):
    _fields_ = [("max_depth", i32), ("rank", i32), ("world_size", i32), ("reserved", i32)]


class rt_stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_reflect", C.c_uint64),
                ("rays_refract", C.c_uint64), ("kernel_ms", C.c_float), ("reserved", C.c_float)]

    def total_rays(self):
        return self.rays_primary + self.rays_shadow + self.rays_reflect + self.rays_refract


class rt_build_info(C.Structure):
    _fields_ = [("validate_ms", C.c_float), ("records_ms", C.c_float), ("upload_ms", C.c_float),
                ("n_nodes", i32), ("n_records", i32), ("n_cull_records", i32), ("reserved", i32),
                ("walk_bytes", C.c_uint64), ("table_bytes", C.c_uint64)]


class rt_multi_stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_reflect", C.c_uint64),
                ("rays_refract", C.c_uint64), ("kernel_ms", C.c_float), ("render_wall_ms", C.c_float),
                ("exchange_ms", C.c_float), ("total_ms", C.c_float)]

    def total_rays(self):
        return self.rays_primary + self.rays_shadow + self.rays_reflect + self.rays_refract


# Every entry point include/rt_abi.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "rt_device_count": (C.c_int, []),
    "rt_scene_create": (C.c_int, [C.POINTER(rt_scene_desc), C.c_int, C.POINTER(C.c_void_p)]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_scene_build_info": (C.c_int, [C.c_void_p, C.POINTER(rt_build_info)]),
    "rt_local_rows": (C.c_int64, [i32, C.POINTER(rt_params)]),
    "rt_global_row": (C.c_int64, [C.c_int64, C.POINTER(rt_params)]),
    "rt_render_device": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.c_void_p,
                                   C.c_void_p]),
    "rt_render_device_static": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.c_void_p,
                                          C.c_void_p]),
    "rt_render": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.POINTER(c_d),
                            C.POINTER(rt_stats)]),
    "rt_render_device_u8": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.c_void_p,
                                      C.c_void_p]),
    "rt_render_u8": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.POINTER(C.c_uint8),
                               C.POINTER(rt_stats)]),
    "rt_render_ppm": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.c_char_p, C.c_uint64,
                                C.POINTER(C.c_uint64), C.POINTER(rt_stats)]),
    "rt_ppm_encode": (C.c_int, [C.c_int, C.POINTER(C.c_uint8), i32, i32, C.c_char_p, C.c_uint64,
                                C.POINTER(C.c_uint64)]),
    "rt_color_at": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(c_d), C.POINTER(c_d), i32, C.POINTER(c_d),
                              C.POINTER(rt_stats)]),
    "rt_multi_create": (C.c_int, [C.POINTER(rt_scene_desc), i32, C.POINTER(i32), C.POINTER(C.c_void_p)]),
    "rt_multi_destroy": (None, [C.c_void_p]),
    "rt_multi_ranks": (i32, [C.c_void_p]),
    "rt_render_multi": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), i32, C.POINTER(c_d), C.POINTER(rt_multi_stats)]),
    "rt_render_multi_u8": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), i32, C.POINTER(C.c_uint8),
                                     C.POINTER(rt_multi_stats)]),
    "rt_render_multi_device": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), i32, C.c_void_p, i32,
                                         C.POINTER(rt_multi_stats)]),
    "rt_host_alloc": (C.c_void_p, [C.c_uint64]),
    "rt_host_free": (None, [C.c_void_p]),
    "rt_scene_launch_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "rt_scene_ray_counts": (C.c_int, [C.c_void_p, C.POINTER(rt_stats)]),
    "rt_selftest_division": (C.c_int, [C.c_int, C.c_int64, C.POINTER(c_d), C.POINTER(c_d), C.POINTER(c_d),
                                       C.POINTER(c_d)]),
    "rt_last_error": (C.c_char_p, []),
}

_LIB = None


def load():
    """Loads librt_amd.so and types its entry points.  Fails loudly when the library is missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("librt_amd.so is not built: run __graft_entry__.build() or `make -C "
                               "raytracer_amd/csrc`")
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rt error %d: %s" % (code, message))
        self.code = code
