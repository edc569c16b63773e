"""ctypes binding of oracle/librc_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (raycore.jl_amd) never does.  See oracle/rc_oracle.h for what the oracle is and how
its parity is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RC_ORACLE_VARIANT=asan|tsan: the sanitized builds (oracle/Makefile `san` / `tsan`), for tests/test_oracle_sanitized.py only -- the
# interpreter must have been started with the matching sanitizer runtime in LD_PRELOAD.
_VARIANT = os.environ.get("RC_ORACLE_VARIANT", "")
assert _VARIANT in ("", "asan", "tsan"), _VARIANT
_SO = os.path.join(_HERE, "librc_oracle%s.so" % ("_" + _VARIANT if _VARIANT else ""))
_TARGET = {"": "all", "asan": "san", "tsan": "tsan"}[_VARIANT]

NODE_DT = np.dtype([("aabb0_min", "<f4", 3), ("aabb0_max", "<f4", 3), ("aabb1_min", "<f4", 3),
                    ("aabb1_max", "<f4", 3), ("child0", "<u4"), ("child1", "<u4"), ("parent", "<u4")])
INSTANCE_DT = np.dtype([("blas_index", "<u4"), ("instance_id", "<u4"), ("transform", "<f4", 12),
                        ("inv_transform", "<f4", 12), ("flags", "<u4")])
DESC_DT = np.dtype([("nodes_offset", "<u4"), ("primitives_offset", "<u4"), ("root_min", "<f4", 3),
                    ("root_max", "<f4", 3)])
TRI_DT = np.dtype([("v", "<f4", (3, 3)), ("meta", "<u4")])
RAY_DT = np.dtype([("o", "<f4", 3), ("tmin", "<f4"), ("d", "<f4", 3), ("tmax", "<f4")])
HIT_DT = np.dtype([("hit", "<u4"), ("t", "<f4"), ("primitive_id", "<u4"), ("instance_custom_index", "<u4"),
                   ("bary_u", "<f4"), ("bary_v", "<f4"), ("instance_id", "<u4"), ("_pad", "<u4")])
NODE4_DT = np.dtype([("child", "<u4", 4), ("aabb", "<f4", (4, 2, 3)), ("parent", "<u4"), ("child_count", "u1"),
                     ("primitive_count", "u1"), ("_pad1", "u1"), ("_pad2", "u1")])  # BVHNode4, src/bvh4.jl:40-69
TRIANGLE_DT = np.dtype([("vertices", "<f4", (3, 3)), ("normals", "<f4", (3, 3)), ("tangents", "<f4", (3, 3)), ("uv", "<f4", (3, 2)),
                        ("metadata", "<u4")])  # Triangle{UInt32}, src/triangle_mesh.jl:1-7
assert NODE4_DT.itemsize == 120 and TRIANGLE_DT.itemsize == 136
assert NODE_DT.itemsize == 60 and INSTANCE_DT.itemsize == 108 and DESC_DT.itemsize == 32
assert TRI_DT.itemsize == 40 and RAY_DT.itemsize == 32 and HIT_DT.itemsize == 32

INVALID_NODE = 0xFFFFFFFF
IDENTITY = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float32)


def build(force=False):
    """Compile the oracle with its committed recipe (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(
            os.path.getmtime(os.path.join(_HERE, f)) for f in ("rc_oracle.c", "rc_oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "-s", _TARGET])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, u32, u64, i32, f32p = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.POINTER(C.c_float)
        L.rco_scene_new.restype = vp
        L.rco_scene_free.argtypes = [vp]
        L.rco_scene_add_blas.restype = u32
        L.rco_scene_add_blas.argtypes = [vp, vp, vp, u32, C.c_int]
        L.rco_scene_add_instance.argtypes = [vp, u32, u32, vp, vp]
        L.rco_scene_build.argtypes = [vp]
        for name in ("tlas_nodes", "instances", "blas_nodes", "blas_prims", "blas_descs"):
            f = getattr(L, "rco_scene_" + name)
            f.restype = u32
            f.argtypes = [vp, vp]
        L.rco_scene_world_bound.argtypes = [vp, vp]
        L.rco_scene_blas_morton.restype = u32
        L.rco_scene_blas_morton.argtypes = [vp, u32, vp]
        L.rco_closest_hit.argtypes = [vp, vp, vp, vp]
        L.rco_any_hit.argtypes = [vp, vp, vp, vp]
        L.rco_trace_batch.argtypes = [vp, vp, vp, u64, C.c_int, C.c_int, vp]
        L.rco_trace_deferred_batch.argtypes = [vp, vp, vp, u64, C.c_uint32, C.c_int]
        L.rco_pool_pin.argtypes = [C.c_int]
        L.rco_allowed_cpus.restype = C.c_int
        L.rco_brute_closest.argtypes = [vp, vp, vp]
        L.rco_corner.restype = None
        L.rco_corner.argtypes = [vp, vp, C.c_int, vp]
        L.rco_expand_bits.restype = u32
        L.rco_expand_bits.argtypes = [u32]
        L.rco_morton_code_30bit.restype = u32
        L.rco_morton_code_30bit.argtypes = [vp]
        L.rco_clz32.restype = i32
        L.rco_clz32.argtypes = [u32]
        L.rco_delta.restype = i32
        L.rco_delta.argtypes = [i32, i32, vp, i32]
        for name in ("mat3x4_inverse", "mat4_to_mat3x4", "safe_invdir"):
            getattr(L, "rco_" + name).argtypes = [vp, vp]
        L.rco_transform_point.argtypes = [vp, vp, vp]
        L.rco_transform_direction.argtypes = [vp, vp, vp]
        L.rco_is_degenerate.argtypes = [vp]
        L.rco_philox4x32_10.argtypes = [vp, vp, vp]
        L.rco_generate_ray_grid.argtypes = [vp, vp, u32, vp]
        L.rco_get_illumination.argtypes = [vp, vp, u32, vp, C.c_int]
        L.rco_view_factors.argtypes = [vp, u32, u64, u32, u32, u32, u32, vp, C.c_int]
        L.rco_view_factor_ray.argtypes = [vp, u32, u32, u64, vp]
        L.rco_view_factor_row.argtypes = [vp, u32, u64, u32, u32, u32, vp]
        L.rco_trace_entries.restype = u32
        L.rco_trace_entries.argtypes = [vp, vp, C.c_int, vp, vp, vp, u32]
        L.rco_trace_steps.restype = u32
        L.rco_trace_steps.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, u32]
        L.rco_trace_events.restype = u32
        L.rco_trace_events.argtypes = [vp, vp, C.c_int, vp, vp, u32]
        L.rco_hit_points.argtypes = [vp, vp, vp, u64, vp, vp]
        L.rco_shadow_rays.argtypes = [vp, vp, vp, u64, vp, C.c_float, vp]
        L.rco_blas4_nodes.restype = u32
        L.rco_blas4_nodes.argtypes = [vp, u32, vp]
        L.rco_trace4_batch.argtypes = [vp, u32, vp, vp, u64, C.c_int, C.c_int, vp]
        L.rco_collide_instances.restype = u64
        L.rco_collide_instances.argtypes = [vp, vp, vp]
        L.rco_collide_instances_any.argtypes = [vp, u32, u32, u32, u32]
        L.rco_scene_add_mesh.restype = u32
        L.rco_scene_add_mesh.argtypes = [vp, vp, vp, vp, u32, vp, u32, vp]
        L.rco_scene_triangles.restype = u32
        L.rco_scene_triangles.argtypes = [vp, vp]
        L.rco_shading_attributes.argtypes = [vp, vp, u64, vp, vp]
        L.rco_primary_rays_lookat.argtypes = [vp, vp, vp, vp, C.c_float, C.c_float, u32, u32, u32, u64, C.c_int, vp]
        L.rco_reflection_rays.argtypes = [vp, vp, vp, u64, C.c_float, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.reshape(shape) if shape is not None else a


def pool_pin(enable=True):
    """Pin the worker pool's threads to the CPUs this process may run on, one each (call before the pool's first use)."""
    lib().rco_pool_pin(1 if enable else 0)


def allowed_cpus():
    return int(lib().rco_allowed_cpus())


class Scene:
    """One-shot StaticTLAS: add_blas per geometry, add_instance, build (== build_tlas)."""

    def __init__(self):
        self._h = lib().rco_scene_new()

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:  # at interpreter shutdown the module globals may already be gone
            try:
                _lib.rco_scene_free(self._h)
            except Exception:  # noqa: BLE001
                pass
            self._h = None

    def add_blas(self, verts, meta=None, filter_degenerate=True):
        verts = _f32(verts).reshape(-1, 9)
        m = None if meta is None else np.ascontiguousarray(meta, dtype=np.uint32)
        idx = lib().rco_scene_add_blas(self._h, _p(verts), _p(m), len(verts), int(filter_degenerate))
        if idx == 0:
            raise ValueError("Geometry has no valid triangles")
        return idx

    def add_mesh(self, verts, faces, normals, uvs=None, face_meta=None):
        v, nrm = _f32(verts).reshape(-1, 3), _f32(normals).reshape(-1, 3)
        f = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
        uv = None if uvs is None else _f32(uvs).reshape(-1, 2)
        fm = None if face_meta is None else np.ascontiguousarray(face_meta, dtype=np.uint32)
        idx = lib().rco_scene_add_mesh(self._h, _p(v), _p(nrm), _p(uv), len(v), _p(f), len(f), _p(fm))
        if idx == 0:
            raise ValueError("Geometry has no valid triangles")
        return idx

    @property
    def triangles(self):
        n = lib().rco_scene_triangles(self._h, None)
        out = np.zeros(n, dtype=TRIANGLE_DT)
        if n:
            lib().rco_scene_triangles(self._h, _p(out))
        return out

    def shading_attributes(self, hits):
        hits = np.ascontiguousarray(hits)
        nrm, uv = np.zeros((len(hits), 3), np.float32), np.zeros((len(hits), 2), np.float32)
        lib().rco_shading_attributes(self._h, _p(hits), len(hits), _p(nrm), _p(uv))
        return nrm, uv

    def add_instance(self, blas_index, xform=None, instance_id=0, inv=None):
        x = IDENTITY if xform is None else _f32(xform).reshape(12)
        i = None if inv is None else _f32(inv).reshape(12)
        if lib().rco_scene_add_instance(self._h, blas_index, instance_id, _p(x), _p(i)) != 0:
            raise ValueError("bad blas index")

    def build(self):
        lib().rco_scene_build(self._h)
        return self

    def _arr(self, fn, dt):
        n = fn(self._h, None)
        out = np.zeros(n, dtype=dt)
        if n:
            fn(self._h, _p(out))
        return out

    @property
    def tlas_nodes(self):
        return self._arr(lib().rco_scene_tlas_nodes, NODE_DT)

    @property
    def instances(self):
        return self._arr(lib().rco_scene_instances, INSTANCE_DT)

    @property
    def blas_nodes(self):
        return self._arr(lib().rco_scene_blas_nodes, NODE_DT)

    @property
    def blas_prims(self):
        return self._arr(lib().rco_scene_blas_prims, TRI_DT)

    @property
    def blas_descs(self):
        return self._arr(lib().rco_scene_blas_descs, DESC_DT)

    def blas_morton(self, blas_index):
        n = lib().rco_scene_blas_morton(self._h, blas_index, None)
        out = np.zeros(n, dtype=np.uint32)
        lib().rco_scene_blas_morton(self._h, blas_index, _p(out))
        return out

    @property
    def world_bound(self):
        out = np.zeros(6, dtype=np.float32)
        lib().rco_scene_world_bound(self._h, _p(out))
        return out

    # ---- traversal ----
    def trace(self, rays, mode="closest", nthreads=1, counters=False, out=None):
        """`out`: a HIT_DT array to reuse (a timing loop must not pay for a fresh 32 B/ray array -- page faults under hundreds of threads --
        on every pass; every record is written)."""
        rays = np.ascontiguousarray(rays, dtype=RAY_DT)
        if out is None:
            hits = np.zeros(len(rays), dtype=HIT_DT)
        else:
            assert out.dtype == HIT_DT and len(out) == len(rays) and out.flags["C_CONTIGUOUS"]
            hits = out
        cnt = np.zeros((len(rays), 2), dtype=np.uint32) if counters else None
        lib().rco_trace_batch(self._h, _p(rays), _p(hits), len(rays), 0 if mode == "closest" else 1, nthreads, _p(cnt))
        return (hits, cnt) if counters else hits

    def trace_deferred(self, rays, lag, nthreads=1):
        """dev experiment: closest_hit whose leaf-test results arrive `lag` loop iterations late (lag 0 = the reference algorithm)."""
        rays = np.ascontiguousarray(rays, dtype=RAY_DT)
        hits = np.zeros(len(rays), dtype=HIT_DT)
        lib().rco_trace_deferred_batch(self._h, _p(rays), _p(hits), len(rays), int(lag), nthreads)
        return hits

    def brute(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_DT)
        hits = np.zeros(len(rays), dtype=HIT_DT)
        for i in range(len(rays)):
            lib().rco_brute_closest(self._h, _p(rays[i:i + 1]), _p(hits[i:i + 1]))
        return hits

    # ---- BVH4 (src/bvh4.jl) ----
    def blas4_nodes(self, blas_index):
        n = lib().rco_blas4_nodes(self._h, blas_index, None)
        out = np.zeros(n, dtype=NODE4_DT)
        lib().rco_blas4_nodes(self._h, blas_index, _p(out))
        return out

    def trace4(self, blas_index, rays, mode="closest", nthreads=1, counters=False):
        rays = np.ascontiguousarray(rays, dtype=RAY_DT)
        hits = np.zeros(len(rays), dtype=HIT_DT)
        cnt = np.zeros((len(rays), 2), dtype=np.uint32) if counters else None
        lib().rco_trace4_batch(self._h, blas_index, _p(rays), _p(hits), len(rays), 0 if mode == "closest" else 1, nthreads, _p(cnt))
        return (hits, cnt) if counters else hits

    # ---- collision broad phase (src/collision.jl) ----
    def collide_instances(self):
        """-> (contacts (m, 2) uint32 [instance_a, instance_b] 1-based, inclusive prefix counts (n,))"""
        n = len(self.instances)
        counts = np.zeros(n, dtype=np.uint32)
        m = lib().rco_collide_instances(self._h, None, _p(counts))
        out = np.zeros((m, 2), dtype=np.uint32)
        if m:
            lib().rco_collide_instances(self._h, _p(out), None)
        return out, counts

    def collide_instances_any(self, range_a, range_b):
        return bool(lib().rco_collide_instances_any(self._h, range_a[0], range_a[1], range_b[0], range_b[1]))

    # ---- drivers ----
    def ray_grid(self, viewdir, grid):
        out = np.zeros(grid * grid, dtype=RAY_DT)
        lib().rco_generate_ray_grid(self._h, _p(_f32(viewdir)), grid, _p(out))
        return out

    def get_illumination(self, viewdir, grid, nthreads=1):
        out = np.zeros(len(self.blas_prims), dtype=np.float32)
        lib().rco_get_illumination(self._h, _p(_f32(viewdir)), grid, _p(out), nthreads)
        return out

    def view_factors(self, rays_per_triangle, seed=0, src=None, rays=None, nthreads=1, out=None):
        n = len(self.blas_prims)
        if out is None:
            out = np.zeros((n, n), dtype=np.uint32, order="F")  # Julia Matrix: [src, dst] column-major
        s0, s1 = (0, n) if src is None else src
        r0, r1 = (0, rays_per_triangle) if rays is None else rays
        lib().rco_view_factors(self._h, rays_per_triangle, seed, s0, s1, r0, r1, _p(out), nthreads)
        return out

    def view_factor_row(self, rays_per_triangle, src_idx0, seed=0, rays=None):
        """Row of source primitive src_idx0 (0-based flat index) as a compact vector: row[hit_meta - 1] = counted rays."""
        row = np.zeros(len(self.blas_prims), dtype=np.uint32)
        r0, r1 = (0, rays_per_triangle) if rays is None else rays
        lib().rco_view_factor_row(self._h, rays_per_triangle, seed, src_idx0, r0, r1, _p(row))
        return row

    def trace_events(self, ray, mode="closest", cap=4096):
        """dev: (events, depths) byte arrays of one ray's traversal steps (tools/sched_sim.py)."""
        ray = np.ascontiguousarray(ray, dtype=RAY_DT).reshape(1)
        ev, dp = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
        n = lib().rco_trace_events(self._h, _p(ray), 0 if mode == "closest" else 1, _p(ev), _p(dp), cap)
        return ev[:min(n, cap)], dp[:min(n, cap)]

    def trace_steps(self, ray, mode="closest", cap=4096):
        """dev: (events, depths, node indices, closest t at the start of each step) of one ray (tools/tlas_subtree_bound.py)."""
        ray = np.ascontiguousarray(ray, dtype=RAY_DT).reshape(1)
        ev, dp, nd, ct = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8), np.zeros(cap, np.uint32), np.zeros(cap, np.float32)
        n = min(lib().rco_trace_steps(self._h, _p(ray), 0 if mode == "closest" else 1, _p(ev), _p(dp), _p(nd), _p(ct), cap), cap)
        return ev[:n], dp[:n], nd[:n], ct[:n]

    def trace_entries(self, ray, mode="closest", cap=512):
        """test: (instance index, closest_t at entry, triangle tests before leaving) for every instance entry of one ray."""
        ray = np.ascontiguousarray(ray, dtype=RAY_DT).reshape(1)
        inst, ct, lf = np.zeros(cap, np.uint32), np.zeros(cap, np.float32), np.zeros(cap, np.uint32)
        n = lib().rco_trace_entries(self._h, _p(ray), 0 if mode == "closest" else 1, _p(inst), _p(ct), _p(lf), cap)
        n = min(n, cap)
        return inst[:n], ct[:n], lf[:n]

    def hit_points(self, rays, hits):
        pts, nrm = np.zeros((len(rays), 3), np.float32), np.zeros((len(rays), 3), np.float32)
        lib().rco_hit_points(self._h, _p(np.ascontiguousarray(rays)), _p(np.ascontiguousarray(hits)), len(rays), _p(pts), _p(nrm))
        return pts, nrm

    def shadow_rays(self, rays, hits, light, bias=0.01):
        out = np.zeros(len(rays), dtype=RAY_DT)
        lib().rco_shadow_rays(self._h, _p(np.ascontiguousarray(rays)), _p(np.ascontiguousarray(hits)), len(rays), _p(_f32(light)), bias, _p(out))
        return out

    def reflection_rays(self, rays, hits, bias=0.01):
        out = np.zeros(len(rays), dtype=RAY_DT)
        lib().rco_reflection_rays(self._h, _p(np.ascontiguousarray(rays)), _p(np.ascontiguousarray(hits)), len(rays), bias, _p(out))
        return out

    def view_factor_ray(self, src_idx0, ray_idx, seed=0):
        out = np.zeros(1, dtype=RAY_DT)
        lib().rco_view_factor_ray(self._h, src_idx0, ray_idx, seed, _p(out))
        return out[0]


def primary_rays_lookat(pos, right, up, forward, half_width, half_height, width, height, samples=1, seed=0, jitter=True):
    out = np.zeros(width * height * samples, dtype=RAY_DT)
    v = [_f32(a) for a in (pos, right, up, forward)]
    lib().rco_primary_rays_lookat(_p(v[0]), _p(v[1]), _p(v[2]), _p(v[3]), half_width, half_height, width, height, samples, seed,
                                  1 if jitter else 0, _p(out))
    return out


def make_rays(origins, directions, tmin=0.0, tmax=np.inf):
    origins = np.asarray(origins, dtype=np.float32).reshape(-1, 3)
    directions = np.broadcast_to(np.asarray(directions, dtype=np.float32).reshape(-1, 3), origins.shape)
    r = np.zeros(len(origins), dtype=RAY_DT)
    r["o"] = origins
    r["d"] = directions
    r["tmin"] = tmin
    r["tmax"] = tmax
    return r


def mat4_to_mat3x4(m4_colmajor_16):
    """Julia Mat4f(args...) is column-major: pass the 16 constructor arguments in order."""
    out = np.zeros(12, dtype=np.float32)
    lib().rco_mat4_to_mat3x4(_p(_f32(m4_colmajor_16).reshape(16)), _p(out))
    return out


def mat3x4_inverse(m):
    out = np.zeros(12, dtype=np.float32)
    lib().rco_mat3x4_inverse(_p(_f32(m).reshape(12)), _p(out))
    return out
