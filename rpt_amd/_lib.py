"""ctypes binding of the C ABI in include/rpt_hip.h (librpt_hip.so).

The HIP library is the product; there is no CPU fallback.  Importing this module on a
machine where the library is missing, or calling into it without a gfx950 GPU, raises.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RPT_LIB", os.path.join(_HERE, "librpt_hip.so"))  # RPT_LIB: A/B builds


class RptError(RuntimeError):
    """Raised for every non-zero return code of the C ABI (message from rpt_last_error)."""


class ShapeDesc(C.Structure):
    pass


ShapeDesc._fields_ = [
    ("kind", C.c_int32),
    ("has_transform", C.c_int32),
    ("transform", C.c_double * 16),
    ("plane_normal", C.c_double * 3),
    ("plane_value", C.c_double),
    ("tris", C.POINTER(C.c_double)),
    ("n_tris", C.c_uint64),
    ("children", C.POINTER(ShapeDesc)),  # kind 4 (KdTree of bounded shapes)
    ("n_children", C.c_uint64),
]


class MaterialDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("_pad", C.c_int32),
        ("albedo", C.c_double * 3),
        ("emittance", C.c_double),
        ("shininess", C.c_double),
        ("ior", C.c_double),
    ]


class CameraDesc(C.Structure):
    _fields_ = [
        ("eye", C.c_double * 3),
        ("direction", C.c_double * 3),
        ("up", C.c_double * 3),
        ("fov", C.c_double),
        ("aperture", C.c_double),
        ("focal_distance", C.c_double),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("exposure_value", C.c_double),
        ("max_bounces", C.c_uint32),
        ("shard_rank", C.c_uint32),
        ("shard_count", C.c_uint32),
    ]


# Every symbol include/rpt_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
_D3 = C.POINTER(C.c_double)
SYMBOLS = [
    ("rpt_device_count", C.c_int, []),
    ("rpt_last_error", C.c_char_p, []),
    ("rpt_shard_tiles", C.c_int64, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P, C.c_uint64]),
    ("rpt_scene_create", _P, []),
    ("rpt_scene_destroy", None, [_P]),
    ("rpt_scene_add_object", C.c_int, [_P, C.POINTER(ShapeDesc), C.POINTER(MaterialDesc)]),
    ("rpt_scene_add_light_point", C.c_int, [_P, _D3, _D3]),
    ("rpt_scene_add_light_ambient", C.c_int, [_P, _D3]),
    ("rpt_scene_add_light_directional", C.c_int, [_P, _D3, _D3]),
    ("rpt_scene_add_light_object", C.c_int, [_P, C.POINTER(ShapeDesc), C.POINTER(MaterialDesc)]),
    ("rpt_scene_add_medium", C.c_int, [_P, C.c_int32, C.c_double, C.c_double]),
    ("rpt_scene_set_environment_color", C.c_int, [_P, _D3]),
    ("rpt_scene_set_environment_hdri", C.c_int, [_P, C.c_uint32, C.c_uint32, _D3]),
    ("rpt_scene_commit", C.c_int, [_P, C.c_int]),
    ("rpt_render_sample", C.c_int,
     [_P, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_uint32, C.c_uint64, C.c_uint32, _P]),
    ("rpt_render_sample_device", C.c_int,
     [_P, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_uint32, C.c_uint64, C.c_uint32, _P, _P]),
    ("rpt_intersect_batch", C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, _P]),
    ("rpt_scene_stats", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("rpt_get_counters", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("rpt_debug_section_counters", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("rpt_get_timing", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    ("rpt_get_timing_mean", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    ("rpt_render_chunking", C.c_int, [C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("rpt_scene_render_chunking", C.c_int, [_P, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("rpt_set_option", C.c_int, [C.c_char_p, C.c_int64]),
    ("rpt_scene_set_option", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("rpt_buffer_create", _P, [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]),
    ("rpt_buffer_destroy", None, [_P]),
    ("rpt_buffer_add_samples", C.c_int, [_P, _P]),
    ("rpt_buffer_add_samples_device", C.c_int, [_P, _P, _P]),
    ("rpt_buffer_image", C.c_int, [_P, _P]),
    ("rpt_buffer_variance", C.c_int, [_P, C.POINTER(C.c_double)]),
    ("rpt_buffer_batches", C.c_int, [_P, C.POINTER(C.c_uint32)]),
    ("rpt_render_into_buffer", C.c_int,
     [_P, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_uint32, C.c_uint64, C.c_uint32, _P]),
    ("rpt_photon_map_build", C.c_int, [_P, C.c_uint64, C.c_int32, C.c_double, C.c_uint64]),
    ("rpt_photon_shoot", C.c_int,
     [_P, C.c_uint64, C.c_int32, C.c_double, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    ("rpt_photon_records", C.c_int, [_P, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    ("rpt_photon_map_from_records", C.c_int, [_P, C.c_uint64, C.c_int32, _P, C.c_uint64, _P, C.c_uint64]),
    ("rpt_photon_map_stats", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("rpt_photon_map_download", C.c_int, [_P, C.c_int32, _P, C.c_uint64]),
    ("rpt_photon_render_sample", C.c_int,
     [_P, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, _P]),
    ("rpt_photon_render_sample_device", C.c_int,
     [_P, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, _P, _P]),
    ("rpt_comm_unique_id", C.c_int, [_P]),
    ("rpt_comm_create", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    ("rpt_comm_destroy", None, [_P]),
    ("rpt_comm_rank", C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("rpt_gather_frame_device", C.c_int, [_P, C.c_uint32, C.c_uint32, _P, _P, C.c_uint32, _P]),
    ("rpt_allgather_records_device", C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _P]),
    ("rpt_frame_pack_layout", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    ("rpt_frame_pack_device", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, _P]),
    ("rpt_frame_unpack_device", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, _P]),
    ("rpt_debug_epsilon_counters", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("rpt_debug_rng_u32", C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _P]),
    ("rpt_debug_material_sample_f", C.c_int,
     [C.POINTER(MaterialDesc), C.c_uint64, _P, _P, C.c_uint64, _P, _P, _P]),
    ("rpt_debug_material_bsdf", C.c_int, [C.POINTER(MaterialDesc), C.c_uint64, _P, _P, _P, _P]),
    ("rpt_debug_radix_sort", C.c_int, [C.c_uint64, _P, _P, _P]),
    ("rpt_debug_exclusive_scan2", C.c_int, [C.c_uint64, _P, _P, _P, _P, C.POINTER(C.c_uint64)]),
    ("rpt_debug_photon_positions64", C.c_int, [_P, _P, C.c_uint64]),
    ("rpt_debug_photon_selections", C.c_int, [_P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("rpt_debug_camera_rays", C.c_int,
     [C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_uint64, C.c_uint32, _P, _P]),
]

_lib = None


def load():
    """Load librpt_hip.so (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RptError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # torch (when used in the same process) bundles its own libamdhip64.so.7; load it first so
    # both share ONE HIP runtime instead of two copies with the same SONAME.
    if "torch" not in sys.modules and os.environ.get("RPT_IMPORT_TORCH", "0") == "1":
        import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc < 0:
        msg = load().rpt_last_error()
        raise RptError(f"rpt error {rc}: {msg.decode() if msg else ''}")
    return rc
