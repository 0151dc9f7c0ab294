"""ctypes binding of include/mrt_abi.h (libmrt_hip.so).

This is the only way Python reaches the hot path: every call goes through the C ABI, exactly as
a Swift/C++ host would.  If the library is missing the import fails loudly — there is no CPU
fallback and nothing under oracle/ is ever imported from here.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MRT_LIB_PATH") or os.path.join(_HERE, "libmrt_hip.so")   # MRT_LIB_PATH: A/B builds during tuning

MRT_OK = 0
MRT_ERR_INVALID_ARGUMENT, MRT_ERR_NO_DEVICE, MRT_ERR_HIP, MRT_ERR_IO, MRT_ERR_STATE, MRT_ERR_OUT_OF_MEMORY, MRT_ERR_UNSUPPORTED = 1, 2, 3, 4, 5, 6, 7


class MRTError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"mrt error {code}: {message}")
        self.code = code


# ---------------------------------------------------------------- data contract (ShaderTypes.h:60-107)
class Float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("_pad", C.c_float)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(float(x), float(y), float(z), 0.0)

    def tolist(self):
        return [self.x, self.y, self.z]


def _f3(v):
    return v if isinstance(v, Float3) else Float3(*v)


class Camera(C.Structure):  # ShaderTypes.h:60-65
    _fields_ = [("position", Float3), ("right", Float3), ("up", Float3), ("forward", Float3)]


class LightType:  # ShaderTypes.h:67-74
    unused, sunlight, spotlight, pointlight, areaLight = 0, 1, 2, 3, 4


class Light(C.Structure):  # ShaderTypes.h:76-87
    _fields_ = [("type", C.c_int32), ("_pad0", C.c_int32 * 3), ("position", Float3), ("color", Float3),
                ("forward", Float3), ("right", Float3), ("up", Float3), ("coneAngle", C.c_float),
                ("_pad1", C.c_float * 3), ("direction", Float3)]

    # Scene.swift:70-107
    @staticmethod
    def areaLight(position, forward, right, up, color):
        l = Light(); l.type = LightType.areaLight
        l.position, l.forward, l.right, l.up, l.color = _f3(position), _f3(forward), _f3(right), _f3(up), _f3(color)
        return l

    @staticmethod
    def sunLight(direction, color):
        l = Light(); l.type = LightType.sunlight
        l.direction, l.color = _f3(direction), _f3(color)
        return l

    @staticmethod
    def pointLight(position, color):
        l = Light(); l.type = LightType.pointlight
        l.position, l.color = _f3(position), _f3(color)
        return l

    @staticmethod
    def spotLight(position, direction, coneAngle, color):
        l = Light(); l.type = LightType.spotlight
        l.position, l.direction, l.coneAngle, l.color = _f3(position), _f3(direction), float(coneAngle), _f3(color)
        return l


class Uniforms(C.Structure):  # ShaderTypes.h:89-97
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("blocksWide", C.c_int32), ("frameIndex", C.c_uint32),
                ("lightCount", C.c_int32), ("_pad", C.c_int32 * 3), ("camera", Camera)]


class Material(C.Structure):  # ShaderTypes.h:99-107
    _fields_ = [("baseColor", Float3), ("specular", Float3), ("emission", Float3), ("specularExponent", C.c_float),
                ("refractionIndex", C.c_float), ("dissolve", C.c_float), ("_pad", C.c_float)]


class Ray(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("min_distance", C.c_float), ("direction", C.c_float * 3), ("max_distance", C.c_float)]


class Intersection(C.Structure):
    _fields_ = [("type", C.c_int32), ("distance", C.c_float), ("instance_id", C.c_int32), ("geometry_id", C.c_int32),
                ("primitive_id", C.c_int32), ("u", C.c_float), ("v", C.c_float), ("_pad", C.c_int32)]


class SceneStats(C.Structure):
    _fields_ = [("triangles", C.c_uint64), ("vertices", C.c_uint64), ("bvh_nodes", C.c_uint64), ("bvh_leaves", C.c_uint64),
                ("scene_bytes", C.c_uint64), ("build_ms", C.c_float), ("sah_cost", C.c_float), ("instances", C.c_int32),
                ("max_submeshes", C.c_int32), ("max_leaf_tris", C.c_int32), ("max_depth", C.c_int32), ("wide_layout", C.c_int32), ("wide_depth", C.c_int32),
                ("wide_cost", C.c_float), ("wide_cost_built", C.c_float), ("refits", C.c_uint32), ("leaf_growth", C.c_float)]


class RenderStats(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("primary_rays", C.c_uint64),
                ("bytes_alg", C.c_uint64), ("ms_gpu_last", C.c_float), ("ms_extend_last", C.c_float),
                ("extend_launches_last", C.c_uint32), ("_pad", C.c_uint32)]


class KernelTimes(C.Structure):
    _fields_ = [("ms", C.c_float * 4), ("launches", C.c_uint32 * 4)]


KERNEL_CLASSES = ("primary", "shade", "trace", "accumulate")

assert C.sizeof(Camera) == 64 and C.sizeof(Light) == 128 and C.sizeof(Uniforms) == 96 and C.sizeof(Material) == 64
assert C.sizeof(Ray) == 32 and C.sizeof(Intersection) == 32

# ---------------------------------------------------------------- function table: every symbol include/mrt_abi.h declares
_P, _I32, _U32, _F, _SZ = C.c_void_p, C.c_int32, C.c_uint32, C.c_float, C.c_size_t
_PF, _PU32, _PI32 = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
SIGNATURES = {
    "mrt_last_error": (C.c_char_p, []),
    "mrt_abi_version": (C.c_int, []),
    "mrt_context_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "mrt_context_destroy": (C.c_int, [_P]),
    "mrt_context_set_stream": (C.c_int, [_P, _P]),
    "mrt_context_device_name": (C.c_int, [_P, C.c_char_p, _SZ]),
    "mrt_scene_create": (C.c_int, [_P, C.POINTER(_P)]),
    "mrt_scene_destroy": (C.c_int, [_P]),
    "mrt_scene_add_mesh": (C.c_int, [_P, _P, _SZ, _P, _SZ, _SZ, _P, _PI32]),
    "mrt_scene_add_instance": (C.c_int, [_P, _I32, _P, _PI32]),
    "mrt_mesh_add_submesh": (C.c_int, [_P, _I32, _P, _SZ, C.POINTER(Material), _PI32]),
    "mrt_scene_add_obj": (C.c_int, [_P, C.c_char_p, _PF, _PF, _F, _PI32]),
    "mrt_scene_set_lights": (C.c_int, [_P, C.POINTER(Light), _I32]),
    "mrt_scene_commit": (C.c_int, [_P]),
    "mrt_scene_set_option": (C.c_int, [_P, C.c_char_p, C.c_double]),
    "mrt_scene_set_instance_transform": (C.c_int, [_P, _I32, _P]),
    "mrt_scene_update_mesh": (C.c_int, [_P, _I32, _P, _SZ, _P, _SZ, _SZ]),
    "mrt_debug_scene_refits": (C.c_int, [_P, C.POINTER(_U32)]),
    "mrt_scene_stats": (C.c_int, [_P, C.POINTER(SceneStats)]),
    "mrt_scene_instance_transform": (C.c_int, [_P, _I32, _PF]),
    "mrt_scene_intersect_closest": (C.c_int, [_P, _P, _SZ, _P]),
    "mrt_scene_intersect_any": (C.c_int, [_P, _P, _SZ, _P]),
    "mrt_obj_load": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "mrt_dragon_proxy": (C.c_int, [C.POINTER(_P)]),
    "mrt_dragon_proxy_irregular": (C.c_int, [C.POINTER(_P)]),
    "mrt_dragon_proxy_hostile": (C.c_int, [C.POINTER(_P)]),
    "mrt_bunny_proxy": (C.c_int, [C.POINTER(_P)]),
    "mrt_meshdata_free": (C.c_int, [_P]),
    "mrt_meshdata_counts": (C.c_int, [_P, C.POINTER(_SZ), _PI32]),
    "mrt_meshdata_vertices": (C.c_int, [_P, _P, _P]),
    "mrt_meshdata_submesh": (C.c_int, [_P, _I32, C.POINTER(_SZ), _P, C.POINTER(Material), C.c_char_p, _SZ]),
    "mrt_make_transform": (C.c_int, [_PF, _PF, _F, _PF]),
    "mrt_default_camera": (C.c_int, [_I32, _I32, C.POINTER(Camera)]),
    "mrt_renderer_create": (C.c_int, [_P, _P, _I32, _I32, _U32, _I32, C.POINTER(_P)]),
    "mrt_renderer_destroy": (C.c_int, [_P]),
    "mrt_renderer_resize": (C.c_int, [_P, _I32, _I32]),
    "mrt_renderer_set_camera": (C.c_int, [_P, C.POINTER(Camera)]),
    "mrt_renderer_set_uniforms": (C.c_int, [_P, C.POINTER(Uniforms)]),
    "mrt_renderer_get_uniforms": (C.c_int, [_P, C.POINTER(Uniforms)]),
    "mrt_renderer_frames_completed": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "mrt_renderer_set_option": (C.c_int, [_P, C.c_char_p, C.c_double]),
    "mrt_renderer_get_option": (C.c_int, [_P, C.c_char_p, _P]),
    "mrt_debug_renderer_set_option": (C.c_int, [_P, C.c_char_p, C.c_double]),
    "mrt_debug_renderer_get_option": (C.c_int, [_P, C.c_char_p, _P]),
    "mrt_renderer_set_shard": (C.c_int, [_P, _I32, _I32]),
    "mrt_renderer_set_frame_index": (C.c_int, [_P, _U32]),
    "mrt_renderer_frame_index": (C.c_int, [_P, _PU32]),
    "mrt_renderer_render": (C.c_int, [_P, _I32]),
    "mrt_renderer_wait": (C.c_int, [_P]),
    "mrt_renderer_read_accum": (C.c_int, [_P, _P, _SZ]),
    "mrt_renderer_copy_accum_to_device": (C.c_int, [_P, _P, _SZ]),
    "mrt_renderer_write_accum_from_device": (C.c_int, [_P, _P, _SZ]),
    "mrt_renderer_shard_tiles": (C.c_int, [_P, _I32, _I32, C.POINTER(C.c_uint64)]),
    "mrt_renderer_pack_owned_tiles": (C.c_int, [_P, _P, _SZ]),
    "mrt_renderer_unpack_tiles": (C.c_int, [_P, _P, _SZ, _I32, _I32]),
    "mrt_renderer_read_tonemapped_rgba8": (C.c_int, [_P, _P, _SZ]),
    "mrt_renderer_stats": (C.c_int, [_P, C.POINTER(RenderStats)]),
    "mrt_renderer_reset_stats": (C.c_int, [_P]),
    "mrt_renderer_kernel_times": (C.c_int, [_P, C.POINTER(KernelTimes)]),
    "mrt_group_create": (C.c_int, [_PI32, _I32, C.POINTER(_P)]),
    "mrt_group_destroy": (C.c_int, [_P]),
    "mrt_group_size": (C.c_int, [_P, _PI32]),
    "mrt_group_context": (C.c_int, [_P, _I32, C.POINTER(_P)]),
    "mrt_group_reduce_mode": (C.c_int, [_P, _PI32, C.c_char_p, _SZ]),
    "mrt_group_set_reduce_mode": (C.c_int, [_P, _I32]),
    "mrt_group_renderer_create": (C.c_int, [_P, _P, _I32, _I32, _U32, _I32, C.POINTER(_P)]),
    "mrt_group_renderer_destroy": (C.c_int, [_P]),
    "mrt_group_renderer_rank": (C.c_int, [_P, _I32, C.POINTER(_P)]),
    "mrt_group_set_option": (C.c_int, [_P, C.c_char_p, C.c_double]),
    "mrt_group_set_camera": (C.c_int, [_P, C.POINTER(Camera)]),
    "mrt_group_render": (C.c_int, [_P, _I32]),
    "mrt_group_wait": (C.c_int, [_P]),
    "mrt_group_frames_completed": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "mrt_group_gather": (C.c_int, [_P, _P, _SZ]),
    "mrt_group_gathered_device_ptr": (C.c_int, [_P, C.POINTER(_P)]),
    "mrt_group_stats": (C.c_int, [_P, C.POINTER(RenderStats)]),
    "mrt_debug_halton": (C.c_int, [_P, _P, _P, _SZ, _P]),
    "mrt_debug_hemisphere": (C.c_int, [_P, _P, _P, _SZ, _P]),
    "mrt_debug_seeds": (C.c_int, [_P, _U32, _I32, _I32, _P]),
    "mrt_debug_traversal_stats": (C.c_int, [_P, _P, _SZ, _I32, _P]),
    "mrt_debug_intersect_stream": (C.c_int, [_P, _P, _SZ, _I32, _P]),
    "mrt_debug_stream_stats": (C.c_int, [_P, _P, _SZ, _I32, _U32, _P, _SZ]),
    "mrt_debug_calibrate": (C.c_int, [_P, _SZ, _P]),
    "mrt_debug_wide_histogram": (C.c_int, [_P, _P]),
    "mrt_debug_commit_times": (C.c_int, [_P, _P]),
    "mrt_debug_read_wnodes": (C.c_int, [_P, _P, _SZ, C.POINTER(C.c_uint64)]),
    "mrt_debug_layout_limits": (C.c_int, [C.c_uint64, C.c_uint64]),
    "mrt_debug_host_sah": (C.c_int, [_P, _P, _U32, _P, _P, _P, _P]),
    "mrt_debug_validate": (C.c_int, [_P]),
    "mrt_debug_validate_patched": (C.c_int, [_P, _U32, _U32, _U32]),
}

# Frames in flight run on separate HIP streams; the runtime maps streams onto 4 hardware queues by default,
# which makes lanes collide (measured: 3.6 -> 4.5 Grays/s with 8 queues).  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: the HIP extension has not been built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C metal-raytracing_amd/csrc`. There is no CPU fallback for the render path.")



def _share_hip_runtime_with_torch():
    """One HIP runtime per process, whatever the import order.  PyTorch ships its own libamdhip64 / libhsa-runtime64 under
    torch/lib with the SAME sonames as /opt/rocm's: whichever copy is loaded first serves both.  libmrt_hip.so works on either; torch
    only works on its own (import torch after /opt/rocm's runtime is resident: "No HIP GPUs are available").  So when torch is
    installed and not yet imported, its copies are loaded here, before libmrt_hip.so pulls in the system ones.  A C / C++ / Swift
    host never sees this: it has one runtime.  MRT_HIP_RUNTIME=system skips it."""
    import sys
    if "torch" in sys.modules or os.environ.get("MRT_HIP_RUNTIME", "") == "system":
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.submodule_search_locations:
        return None
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    loaded = []
    # (not librccl: the library opens RCCL only when a device group is created, and then takes the copy already in the process)
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path)
                loaded.append(path)
            except OSError:
                return loaded
    return loaded


HIP_RUNTIME_PRELOADED = _share_hip_runtime_with_torch()
lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in SIGNATURES.items():
    if os.environ.get("MRT_LIB_PATH") and not hasattr(lib, _name):
        continue                       # an A/B build of an older tree (tools/build_variant.sh): entry points added since are simply absent
    _fn = getattr(lib, _name)          # AttributeError here = header/library mismatch: fail loudly
    _fn.restype, _fn.argtypes = _res, _args


def check(rc):
    if rc != MRT_OK:
        raise MRTError(rc, lib.mrt_last_error().decode("utf-8", "replace"))


def ptr(a):
    """void* of a C-contiguous numpy array."""
    return a.ctypes.data_as(C.c_void_p)
