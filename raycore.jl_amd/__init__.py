"""raycore.jl_amd -- MI355X-native TLAS/BLAS traversal behind Raycore.jl's accel API (host side).

`import raycore_jl_amd` (the alias module at the repo root) loads this package; the directory name
contains a dot and cannot be imported by that name.
"""
from . import scenes  # noqa: F401  (pure numpy; usable without the library)
from ._capi import HIT_DT, LIB_PATH, RAY_DT, SYMBOLS, TRIANGLE_DT, RaycoreError, lib  # noqa: F401
from .api import (BLAS4, CONTACT_DT, CollisionResult, ContactPair, collide_instances, collide_instances_any, EMPTY_TRIANGLE, RAYHIT_DT, INVALID_HANDLE, any_hit4, build_blas4, closest_hit4, Bounds3, Ray, RayHit, StaticTLAS, TLAS, TLAS_from_items,  # noqa: F401
                  TLAS_from_meshes, TLASHandle, Triangle, adapt, any_hit, closest_hit, generate_ray_grid,
                  get_centroid, get_illumination, hits_from_grid, typed_hit_metadata, mat4_to_mat3x4, sync, trace_rays, view_factors, view_factors_multi, view_factor_totals, view_factor_totals_multi, multi_prepare, trace_multi, get_illumination_multi,
                  world_bound, expand_faceviews)


def device_count():
    """Number of visible HIP devices according to the library (0 without a GPU)."""
    return lib().rc_device_count()


class profile_range:
    """`with profile_range("shade"):` -- one roctx range around a caller's phase (every entry point of the library is a range of its own; see
    rc_range_push in include/raycore_mi355x.h).  A no-op without a marker library or with RC_ROCTX=0."""

    def __init__(self, name):
        self.name = name.encode("utf-8")

    def __enter__(self):
        lib().rc_range_push(self.name)
        return self

    def __exit__(self, *exc):
        lib().rc_range_pop()
        return False


def ranges_enabled():
    return bool(lib().rc_ranges_enabled())
