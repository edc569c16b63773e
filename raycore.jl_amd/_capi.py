"""ctypes binding of libraycore_mi355x.so (include/raycore_mi355x.h).

The library is the product: hand-written gfx950 HIP kernels behind a C ABI.  There is no CPU fallback
here or in the library -- a missing .so or a missing GPU is a loud error.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libraycore_mi355x.so")

# numpy views of the wire / export structs (byte-identical to the C structs)
RAY_DT = np.dtype([("o", "<f4", 3), ("tmin", "<f4"), ("d", "<f4", 3), ("tmax", "<f4")])
HIT_DT = np.dtype([("hit", "<u4"), ("t", "<f4"), ("primitive_id", "<u4"), ("instance_custom_index", "<u4"),
                   ("bary_u", "<f4"), ("bary_v", "<f4"), ("instance_id", "<u4"), ("_pad", "<u4")])
NODE_DT = np.dtype([("aabb0_min", "<f4", 3), ("aabb0_max", "<f4", 3), ("aabb1_min", "<f4", 3),
                    ("aabb1_max", "<f4", 3), ("child0", "<u4"), ("child1", "<u4"), ("parent", "<u4")])
INSTANCE_DT = np.dtype([("blas_index", "<u4"), ("instance_id", "<u4"), ("transform", "<f4", 12),
                        ("inv_transform", "<f4", 12), ("flags", "<u4")])
DESC_DT = np.dtype([("nodes_offset", "<u4"), ("primitives_offset", "<u4"), ("root_min", "<f4", 3),
                    ("root_max", "<f4", 3)])
PRIM_DT = np.dtype([("v", "<f4", (3, 3)), ("meta", "<u4")])
NODE4_DT = np.dtype([("child", "<u4", 4), ("aabb", "<f4", (4, 2, 3)), ("parent", "<u4"), ("child_count", "u1"),
                     ("primitive_count", "u1"), ("_pad1", "u1"), ("_pad2", "u1")])  # BVHNode4, src/bvh4.jl:40-69
assert NODE4_DT.itemsize == 120
TRIANGLE_DT = np.dtype([("vertices", "<f4", (3, 3)), ("normals", "<f4", (3, 3)), ("tangents", "<f4", (3, 3)), ("uv", "<f4", (3, 2)),
                        ("metadata", "<u4")])  # Triangle{UInt32}, src/triangle_mesh.jl:1-7
assert TRIANGLE_DT.itemsize == 136
assert RAY_DT.itemsize == 32 and HIT_DT.itemsize == 32 and NODE_DT.itemsize == 60
assert INSTANCE_DT.itemsize == 108 and DESC_DT.itemsize == 32 and PRIM_DT.itemsize == 40

RC_OK, RC_ERR_INVALID_ARGUMENT, RC_ERR_INVALID_HANDLE, RC_ERR_NO_DEVICE = 0, 1, 2, 3
RC_INVALID_ID = 0xFFFFFFFF

# every symbol include/raycore_mi355x.h declares: (name, restype, argtypes)
_vp, _u32, _u64, _i64, _int = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int64, C.c_int
_pu32, _pint, _pf = C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_float)
SYMBOLS = [
    ("rc_last_error", C.c_char_p, []),
    ("rc_range_push", C.c_int, [C.c_char_p]),
    ("rc_range_pop", C.c_int, []),
    ("rc_ranges_enabled", C.c_int, []),
    ("rc_device_count", _int, []),
    ("rc_scene_create", _int, [_int, C.POINTER(_vp)]),
    ("rc_scene_destroy", _int, [_vp]),
    ("rc_add_blas", _int, [_vp, _vp, _vp, _u32, _pu32]),
    ("rc_add_blas_device", _int, [_vp, _vp, _vp, _u32, _pu32]),
    ("rc_add_instances", _int, [_vp, _u32, _vp, _vp, _u32, _pu32]),
    ("rc_add_instances_with_inverse", _int, [_vp, _u32, _vp, _vp, _vp, _u32, _pu32]),
    ("rc_update_transforms", _int, [_vp, _u32, _vp, _u32]),
    ("rc_update_geometry", _int, [_vp, _u32, _vp, _vp, _u32]),
    ("rc_delete", _int, [_vp, _u32, _pint]),
    ("rc_is_valid", _int, [_vp, _u32, _pint]),
    ("rc_handle_instance_count", _int, [_vp, _u32, _pu32]),
    ("rc_get_instances", _int, [_vp, _u32, _vp, _u32, _pu32]),
    ("rc_sync", _int, [_vp, _pint]),
    ("rc_counts", _int, [_vp, _pu32, _pu32, _pu32, _pu32, _pu32, _pu32]),
    ("rc_world_bound", _int, [_vp, _vp]),
    ("rc_wait", _int, [_vp]),
    ("rc_export_tlas_nodes", _int, [_vp, _vp, _u32, _pu32]),
    ("rc_export_blas_nodes", _int, [_vp, _vp, _u32, _pu32]),
    ("rc_export_instances", _int, [_vp, _vp, _u32, _pu32]),
    ("rc_export_blas_descs", _int, [_vp, _vp, _u32, _pu32]),
    ("rc_export_prims", _int, [_vp, _vp, _u32, _pu32]),
    ("rc_trace_closest", _int, [_vp, _vp, _vp, _u64]),
    ("rc_trace_any", _int, [_vp, _vp, _vp, _u64]),
    ("rc_trace_closest_device", _int, [_vp, _vp, _vp, _u64, _vp]),
    ("rc_trace_any_device", _int, [_vp, _vp, _vp, _u64, _vp]),
    ("rc_trace_closest_device_batches", _int, [_vp, _vp, _vp, _vp, _int, _vp]),
    ("rc_trace_any_device_batches", _int, [_vp, _vp, _vp, _vp, _int, _vp]),
    ("rc_set_option", _int, [_vp, C.c_char_p, _i64]),
    ("rc_get_option", _int, [_vp, C.c_char_p, C.POINTER(_i64)]),
    ("rc_generate_ray_grid_device", _int, [_vp, _vp, _u32, _vp, _vp]),
    ("rc_get_illumination", _int, [_vp, _vp, _u32, _vp]),
    ("rc_get_illumination_device", _int, [_vp, _vp, _u32, _u64, _u64, _vp, _vp]),
    ("rc_view_factors_device", _int, [_vp, _u32, _u64, _u32, _u32, _u32, _u32, _vp, _u64, _u64, _u32, _u32, _vp]),
    ("rc_view_factors", _int, [_vp, _u32, _u64, _vp]),
    ("rc_view_factors_rows_host", _int, [_vp, _u32, _u64, _u32, _u32, _vp, _u64]),
    ("rc_view_factors_multi", _int, [C.POINTER(_vp), _int, _u32, _u64, _vp, _int]),
    ("rc_view_factor_totals", _int, [_vp, _u32, _u64, _vp, _vp]),
    ("rc_view_factor_totals_device", _int, [_vp, _u32, _u64, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    ("rc_view_factor_totals_multi", _int, [C.POINTER(_vp), _int, _u32, _u64, _vp, _vp]),
    ("rc_multi_prepare", _int, [C.POINTER(_vp), _int, _vp]),
    ("rc_multi_ranks", _int, [C.POINTER(_vp), _int, _vp]),
    ("rc_trace_closest_multi", _int, [C.POINTER(_vp), _int, _vp, _vp, _u64]),
    ("rc_trace_any_multi", _int, [C.POINTER(_vp), _int, _vp, _vp, _u64]),
    ("rc_get_illumination_multi", _int, [C.POINTER(_vp), _int, _vp, _u32, _vp]),
    ("rc_view_factor_rays_device", _int, [_vp, _u64, _u32, _u32, _u32, _vp, _vp]),
    ("rc_hit_points_device", _int, [_vp, _vp, _vp, _u64, _vp, _vp, _vp]),
    ("rc_shadow_rays_device", _int, [_vp, _vp, _vp, _u64, _vp, C.c_float, _vp, _vp]),
    ("rc_blas4_build", _int, [_vp, _u32, _pu32]),
    ("rc_export_blas4_nodes", _int, [_vp, _u32, _vp, _u32, _pu32]),
    ("rc_trace_closest4", _int, [_vp, _u32, _vp, _vp, _u64]),
    ("rc_trace_any4", _int, [_vp, _u32, _vp, _vp, _u64]),
    ("rc_trace_closest4_device", _int, [_vp, _u32, _vp, _vp, _u64, _vp]),
    ("rc_trace_any4_device", _int, [_vp, _u32, _vp, _vp, _u64, _vp]),
    ("rc_collide_instances", _int, [_vp, _vp, _u64, C.POINTER(_u64)]),
    ("rc_collide_instances_device", _int, [_vp, _vp, _u64, C.POINTER(_u64), _vp]),
    ("rc_collide_instances_any", _int, [_vp, _u32, _u32, _pint]),
    ("rc_add_mesh", _int, [_vp, _vp, _vp, _vp, _u32, _vp, _u32, _vp, _pu32]),
    ("rc_add_mesh_face_metadata", _int, [_vp, _vp, _vp, _vp, _u32, _vp, _u32, _vp, _pu32]),
    ("rc_update_geometry_mesh", _int, [_vp, _u32, _vp, _vp, _vp, _u32, _vp, _u32, _vp]),
    ("rc_export_triangles", _int, [_vp, _vp, _u32, _pu32]),
    ("rc_shading_attributes_device", _int, [_vp, _vp, _u64, _vp, _vp, _vp]),
    ("rc_primary_rays_lookat_device", _int, [_vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _u32, _u32, _u32, _u64, _int, _vp, _vp]),
    ("rc_reflection_rays_device", _int, [_vp, _vp, _vp, _u64, C.c_float, _vp, _vp]),
    ("rc_compact_hits_device", _int, [_vp, _vp, _u64, _vp, _vp, _vp]),
    ("rc_scene_save", _int, [_vp, C.c_char_p]),
    ("rc_scene_load", _int, [_int, C.c_char_p, C.POINTER(_vp)]),
    ("rc_instance_buffer_device", _int, [_vp, _u32, C.POINTER(_vp), _pu32]),
    ("rc_refit_device", _int, [_vp, _int]),
    ("rc_host_register", _int, [_vp, _vp, _u64]),
    ("rc_host_unregister", _int, [_vp, _vp]),
    ("rc_last_kernel_ms", _int, [_vp, _pf]),
    ("rc_recent_kernel_ms", _int, [_vp, _u32, _pf, _pu32]),
]

_lib = None


class RaycoreError(RuntimeError):
    """Raised for every non-zero status of the C ABI (the Julia wrapper raises ErrorException)."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RaycoreError(-1, f"{LIB_PATH} is missing: build it with `make -C raycore.jl_amd/csrc` "
                                   "(or __graft_entry__.build()); there is no fallback path")
        try:
            # PyTorch ships its own libamdhip64; a process must initialise ONE HIP runtime, and device
            # pointers are shared with torch tensors (torch.distributed / RCCL), so let torch load first.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)  # AttributeError here = the library does not export a declared symbol
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def check(status):
    if status != RC_OK:
        raise RaycoreError(status, lib().rc_last_error().decode("utf-8", "replace"))


def ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a))  # raw device pointer (e.g. torch.Tensor.data_ptr())
