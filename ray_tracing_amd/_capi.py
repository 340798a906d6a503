"""ctypes mirror of include/rt_amd.h and include/rt_host.h.

Plumbing only: every structure and prototype here restates the C headers, and
`load()` opens the in-tree ray_tracing_amd/librt_amd.so built by
`ray_tracing_amd/Makefile` (see __graft_entry__.build).  There is no Python
implementation of anything: if the library is missing, load() raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB selects another build of the same library (kernel experiments); the default is the in-tree one
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(HERE, "librt_amd.so")

RT_OK = 0
RT_ERR_INVALID, RT_ERR_NO_DEVICE, RT_ERR_UNSUPPORTED, RT_ERR_NOMEM = -1, -2, -3, -4
RT_RENDER_ACCUMULATE = 1
RT_TRACE_NOALLOC_MAX_RAYS = 65536
RT_OPT_RENDER_PATH, RT_OPT_TRACE_PATH = 1, 2
(RT_OPT_BLOCK_KERNEL, RT_OPT_CHAIN_TASKS, RT_OPT_TARGET_TASKS, RT_OPT_PATHS_PER_LANE, RT_OPT_NODE_STEPS,
 RT_OPT_CHAIN_BLOCKS_PER_CU, RT_OPT_SHADE_BATCH, RT_OPT_LEAF_BATCH, RT_OPT_LANES_REGROUP, RT_OPT_REGROUP_ROUNDS, RT_OPT_BLOCK_WAVES) = range(3, 14)
(RT_EVAL_LIGHT_PDF, RT_EVAL_ONB, RT_EVAL_ONB_FAST, RT_EVAL_OPTICS, RT_EVAL_PERLIN, RT_EVAL_TEXTURE, RT_EVAL_SAMPLERS, RT_EVAL_BOUNCE,
 RT_EVAL_BOUNCE_FAST) = range(1, 10)
RT_CREATE_KEEP_BOX_RECORDS = 1
RT_MULTI_REDUCE_ALWAYS = 1
RT_PATH_AUTO, RT_PATH_EXACT, RT_PATH_FAST = 0, 1, 2
RT_INFO_FAST_PATH_AVAILABLE, RT_INFO_LAST_RENDER_PATH, RT_INFO_RECORDS, RT_INFO_BVH_LEVELS = 1, 2, 3, 4

# enum values (include/rt_amd.h)
RT_ENTRY_INNER, RT_ENTRY_LEAF = 0, 1
(RT_PRIM_SPHERE, RT_PRIM_MOVING_SPHERE, RT_PRIM_XY_RECT, RT_PRIM_XZ_RECT, RT_PRIM_YZ_RECT, RT_PRIM_GROUP,
 RT_PRIM_MEDIUM) = range(7)
RT_PRIM_FLIP, RT_PRIM_NEEDS_UV = 1, 2
RT_XFORM_TRANSLATE, RT_XFORM_ROTATE_Y = 0, 1
RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE_LIGHT, RT_MAT_ISOTROPIC = range(5)
RT_TEX_CONSTANT, RT_TEX_CHECKER, RT_TEX_NOISE, RT_TEX_IMAGE = range(4)


class rt_entry(C.Structure):
    _fields_ = [("lo", C.c_float * 3), ("a", C.c_int32), ("hi", C.c_float * 3), ("b", C.c_int32)]


class rt_prim(C.Structure):
    _fields_ = [("kind", C.c_int32), ("flags", C.c_uint32), ("material", C.c_int32), ("chain", C.c_int32),
                ("first", C.c_int32), ("count", C.c_int32), ("user_id", C.c_int32), ("reserved", C.c_int32),
                ("f", C.c_float * 12)]


class rt_xform(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_float), ("b", C.c_float), ("c", C.c_float)]


class rt_chain(C.Structure):
    _fields_ = [("first", C.c_int32), ("count", C.c_int32)]


class rt_material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("texture", C.c_int32), ("p0", C.c_float), ("reserved", C.c_int32)]


class rt_texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("nx", C.c_int32), ("ny", C.c_int32),
                ("reserved", C.c_int32), ("offset", C.c_int64), ("color", C.c_float * 3), ("scale", C.c_float)]


class rt_camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
                ("lower_left_corner", C.c_float * 3), ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3),
                ("lens_radius", C.c_float), ("time0", C.c_float), ("time1", C.c_float)]


class rt_scene_desc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
                ("entries", C.POINTER(rt_entry)), ("n_entries", C.c_int32),
                ("prims", C.POINTER(rt_prim)), ("n_prims", C.c_int32),
                ("xforms", C.POINTER(rt_xform)), ("n_xforms", C.c_int32),
                ("chains", C.POINTER(rt_chain)), ("n_chains", C.c_int32),
                ("materials", C.POINTER(rt_material)), ("n_materials", C.c_int32),
                ("textures", C.POINTER(rt_texture)), ("n_textures", C.c_int32),
                ("texels", C.POINTER(C.c_uint8)), ("n_texel_bytes", C.c_int64),
                ("lights", C.POINTER(C.c_int32)), ("n_lights", C.c_int32),
                ("camera", rt_camera)]


class rt_render_params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("nx", C.c_int32), ("ny", C.c_int32), ("sample_begin", C.c_int32),
                ("sample_end", C.c_int32), ("sub_samples", C.c_int32), ("max_depth", C.c_int32),
                ("seed", C.c_uint32), ("flags", C.c_uint32)]


class rt_hit(C.Structure):
    _fields_ = [("prim_id", C.c_int32), ("material", C.c_int32), ("t", C.c_float), ("u", C.c_float),
                ("v", C.c_float), ("p", C.c_float * 3), ("normal", C.c_float * 3), ("reserved", C.c_float)]


assert C.sizeof(rt_entry) == 32 and C.sizeof(rt_prim) == 80 and C.sizeof(rt_texture) == 48
assert C.sizeof(rt_hit) == 48 and C.sizeof(rt_material) == 16

# name -> (restype, argtypes); every symbol include/*.h declares
PROTOTYPES = {
    # include/rt_amd.h
    "rt_scene_create": (C.c_int, [C.POINTER(rt_scene_desc), C.c_int, C.POINTER(C.c_void_p)]),
    "rt_scene_create_flags": (C.c_int, [C.POINTER(rt_scene_desc), C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_render": (C.c_int, [C.c_void_p, C.POINTER(rt_render_params), C.c_void_p, C.c_void_p]),
    "rt_resolve": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "rt_render_to_host": (C.c_int, [C.c_void_p, C.POINTER(rt_render_params), C.c_void_p, C.c_void_p]),
    "rt_sample_range": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rt_render_multi_to_host": (C.c_int, [C.POINTER(rt_scene_desc), C.POINTER(C.c_int32), C.c_int32, C.POINTER(rt_render_params),
                                          C.c_uint32, C.c_void_p, C.c_void_p]),
    "rt_multi_create": (C.c_int, [C.POINTER(rt_scene_desc), C.POINTER(C.c_int32), C.c_int32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "rt_multi_render_to_host": (C.c_int, [C.c_void_p, C.POINTER(rt_render_params), C.c_void_p, C.c_void_p]),
    "rt_multi_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rt_multi_last_render_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "rt_multi_destroy": (None, [C.c_void_p]),
    "rt_camera_rays": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_trace_rays": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_uint32, C.c_void_p]),
    "rt_trace_rays_noalloc": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_uint32, C.c_void_p]),
    "rt_debug_eval": (C.c_int, [C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_void_p]),
    "rt_last_error": (C.c_char_p, []),
    "rt_abi_version": (C.c_int, []),
    "rt_device_count": (C.c_int, []),
    "rt_last_render_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "rt_scene_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rt_scene_get_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    # include/rt_host.h
    "rt_host_scene_build": (C.c_int, [C.c_char_p, C.c_float, C.POINTER(C.c_void_p)]),
    "rt_host_scene_desc": (C.POINTER(rt_scene_desc), [C.c_void_p]),
    "rt_host_scene_free": (None, [C.c_void_p]),
    "rt_host_scene_names": (C.c_char_p, []),
    "rt_host_ran01_state": (C.c_uint32, []),
}

_lib = None


def load():
    """Opens librt_amd.so and sets the prototypes.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} is missing: build it with `make -C ray_tracing_amd` "
                          "(or python -c 'import __graft_entry__ as g; g.build()'); there is no fallback path")
        # One HIP runtime per process: torch ships its own libamdhip64.so.7 and librt_amd.so asks for
        # the same SONAME.  Importing torch first makes the loader hand librt_amd.so torch's copy, so
        # device pointers and streams can be shared; loaded the other way round the process would hold
        # two runtimes and torch would find no GPU.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib
