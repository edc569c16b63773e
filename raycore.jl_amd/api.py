"""Host-side mirror of Raycore.jl's accel API for the TLAS/BLAS path, over the C ABI.

Names, argument meaning and error behaviour follow the reference (paths relative to the reference repo):
TLAS / push! / delete! / update_transform(s)! / update! / sync! / Adapt.adapt (src/instanced-bvh.jl:334-1102),
closest_hit / any_hit (:1902-2140), trace_rays (ext/RaycoreMakieExt.jl:81-87), get_illumination /
get_centroid / view_factors (src/kernels.jl:58-124).  Python has no `!`, so `push!` is `push`, etc.
Where the Julia API is 1-based (instance index returned by closest_hit, blas_index) this layer is 1-based
too; the C ABI underneath is 0-based.

Triangles are passed as (n, 9) float32 soup (v0 v1 v2) plus optional uint32 metadata: GeometryBasics mesh
decomposition is outside this path.  Transforms are 4x4 (rows 0..2 used, translation
in column 4, like Mat4f) or 12 floats in Mat3x4f byte order.
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _capi
from ._capi import HIT_DT, RAY_DT, RaycoreError, check, lib, ptr

TLASHandle = namedtuple("TLASHandle", ["id"])  # src/instanced-bvh.jl:180-182
INVALID_HANDLE = TLASHandle(0)
# Triangle{UInt32} (src/triangle_mesh.jl:1-7).  Positional order (vertices, metadata) is kept for the path's own use; the
# shading fields default to None when a caller builds one by hand.
Triangle = namedtuple("Triangle", ["vertices", "metadata", "normals", "tangents", "uv"], defaults=(None, None, None))
Bounds3 = namedtuple("Bounds3", ["p_min", "p_max"])
RayHit = namedtuple("RayHit", ["hit", "point", "metadata"])  # src/kernels.jl:1-5
Ray = namedtuple("Ray", ["o", "d", "t_min", "t_max"], defaults=(0.0, np.inf))  # src/ray.jl:1-7 (time unused on the path)

EMPTY_TRIANGLE = Triangle(np.zeros((3, 3), np.float32), np.uint32(0), np.zeros((3, 3), np.float32), np.zeros((3, 3), np.float32),
                          np.zeros((3, 2), np.float32))  # empty_triangle, src/triangle_mesh.jl:49-57


def _triangle(rec):
    return Triangle(rec["vertices"].copy(), rec["metadata"], rec["normals"].copy(), rec["tangents"].copy(), rec["uv"].copy())


def mat4_to_mat3x4(m):
    """mat4_to_mat3x4 (src/instanced-bvh.jl:1663-1669): upper three rows of the 4x4, row-major."""
    m = np.asarray(m, dtype=np.float32)
    if m.shape == (4, 4):
        return np.ascontiguousarray(m[:3, :].reshape(12))
    if m.size == 12:
        return np.ascontiguousarray(m.reshape(12))
    raise ValueError("transform must be 4x4 or 12 floats (Mat3x4f)")


def _as_xforms(transforms):
    if transforms is None:
        return None
    t = np.asarray(transforms, dtype=np.float32)
    if t.ndim == 2 and t.shape == (4, 4):
        t = t[None]
    if t.ndim == 3 and t.shape[1:] == (4, 4):
        return np.ascontiguousarray(t[:, :3, :].reshape(-1, 12))
    return np.ascontiguousarray(t.reshape(-1, 12))


def _as_rays(rays):
    if isinstance(rays, np.ndarray) and rays.dtype == RAY_DT:
        return np.ascontiguousarray(rays)
    if isinstance(rays, Ray):
        rays = [rays]
    out = np.zeros(len(rays), dtype=RAY_DT)
    for i, r in enumerate(rays):
        out[i] = (tuple(r.o), r.t_min, tuple(r.d), r.t_max)
    return out


class StaticTLAS:
    """The adapted form (StaticTLAS, src/instanced-bvh.jl:155-168): what kernels traverse.  Returned by
    TLAS.adapt(); arrays are read back lazily in the reference's layout."""

    def __init__(self, owner):
        self._owner = owner

    def _export(self, fn, dt):
        n = C.c_uint32()
        check(fn(self._owner._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=dt)
        if n.value:
            check(fn(self._owner._h, ptr(out), n.value, None))
        return out

    @property
    def nodes(self):
        return self._export(lib().rc_export_tlas_nodes, _capi.NODE_DT)

    @property
    def instances(self):
        return self._export(lib().rc_export_instances, _capi.INSTANCE_DT)

    @property
    def all_blas_nodes(self):
        return self._export(lib().rc_export_blas_nodes, _capi.NODE_DT)

    @property
    def all_blas_prims(self):
        return self._export(lib().rc_export_prims, _capi.PRIM_DT)

    @property
    def all_blas_triangles(self):
        """all_blas_prims as full 136-byte Triangle{UInt32} records (vertices, normals, tangents, uv, metadata)."""
        return self._export(lib().rc_export_triangles, _capi.TRIANGLE_DT)

    @property
    def blas_descriptors(self):
        return self._export(lib().rc_export_blas_descs, _capi.DESC_DT)

    @property
    def root_aabb(self):
        return self._owner.world_bound()


class TLAS:
    """Mutable two-level accel (TLAS{Backend}, src/instanced-bvh.jl:261-310) on one MI355X."""

    def __init__(self, device=0):  # TLAS(backend), :334-358
        h = C.c_void_p()
        check(lib().rc_scene_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self._static = StaticTLAS(self)
        self._prims_cache = None
        # Triangle{TMeta} for any TMeta (src/triangle_mesh.jl:1-7): the library keeps one uint32 per primitive; metadata that is not a
        # 32-bit unsigned integer is interned here and the word holds its 1-based index (`intern_metadata`, `typed_metadata`).
        self._meta_table = None
        self.meta_type = np.uint32

    # -- lifetime -------------------------------------------------------------------------------------
    def save(self, path):
        """Write the synced scene to a file (geometry BVHs, attributes, instances, handles); see rc_scene_save."""
        self.sync()
        check(lib().rc_scene_save(self._h, str(path).encode()))

    @classmethod
    def load(cls, path, device=0):
        """Read a scene file written by save(); handles keep their ids."""
        t = cls.__new__(cls)
        h = C.c_void_p()
        check(lib().rc_scene_load(int(device), str(path).encode(), C.byref(h)))
        t._h, t.device, t._static, t._prims_cache = h, device, StaticTLAS(t), None
        return t

    def free(self):  # free!, :383-399
        if getattr(self, "_h", None):
            for a in list(getattr(self, "_registered", {}).values()):  # arrays still pinned through host_register
                lib().rc_host_unregister(self._h, ptr(a))
            self._registered = {}
            lib().rc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    # -- mutation -------------------------------------------------------------------------------------
    def push(self, verts, transforms=None, instance_id=0, instance_ids=None, meta=None):
        """push!(tlas, mesh, transform; instance_id) / push!(tlas, mesh, transforms; instance_ids) (:639-676)."""
        verts = np.ascontiguousarray(np.asarray(verts, dtype=np.float32).reshape(-1, 9))
        m = None if meta is None else np.ascontiguousarray(meta, dtype=np.uint32)
        if m is not None and len(m) != len(verts):
            raise ValueError("meta length != triangle count")
        xf = _as_xforms(transforms)
        n_inst = 1 if xf is None else len(xf)
        if instance_ids is not None:
            ids = np.ascontiguousarray(instance_ids, dtype=np.uint32)
            if len(ids) != n_inst:  # ArgumentError, :664-666
                raise ValueError(f"instance_ids length {len(ids)} != transforms length {n_inst}")
        else:
            ids = np.full(n_inst, instance_id, dtype=np.uint32)
        blas_id, handle = C.c_uint32(), C.c_uint32()
        check(lib().rc_add_blas(self._h, ptr(verts), ptr(m), len(verts), C.byref(blas_id)))
        check(lib().rc_add_instances(self._h, blas_id.value, ptr(xf), ptr(ids), n_inst, C.byref(handle)))
        self._prims_cache = None
        return TLASHandle(handle.value)

    def push_instances(self, blas_index, transforms=None, instance_ids=None, inv_transforms=None):
        """More instances of an existing geometry (1-based blas_index, as InstanceDescriptor stores it, :90-96)."""
        xf, inv = _as_xforms(transforms), _as_xforms(inv_transforms)
        n_inst = 1 if xf is None else len(xf)
        ids = np.zeros(n_inst, np.uint32) if instance_ids is None else np.ascontiguousarray(instance_ids, dtype=np.uint32)
        handle = C.c_uint32()
        check(lib().rc_add_instances_with_inverse(self._h, int(blas_index) - 1, ptr(xf), ptr(inv), ptr(ids), n_inst, C.byref(handle)))
        return TLASHandle(handle.value)

    def add_geometry(self, verts, meta=None):
        """build_and_append_blas! alone (:581-608); returns the 1-based BLAS index."""
        verts = np.ascontiguousarray(np.asarray(verts, dtype=np.float32).reshape(-1, 9))
        m = None if meta is None else np.ascontiguousarray(meta, dtype=np.uint32)
        blas_id = C.c_uint32()
        check(lib().rc_add_blas(self._h, ptr(verts), ptr(m), len(verts), C.byref(blas_id)))
        self._prims_cache = None
        return blas_id.value + 1

    def add_mesh(self, verts, faces, normals, uvs=None, face_meta=None, metadata_per_face=None):
        """build_and_append_blas! on a decomposed mesh (:581-608): verts / normals (nv, 3), faces (nf, 3) 0-based, uvs (nv, 2) or
        None, face_meta per vertex (as after expand_faceviews, :595) or None.  `metadata_per_face` (nf words) is the
        TLAS(items, metadata_fn) convention instead -- metadata_fn(mesh_idx, face_idx) for every face (:2300-2306) -- which a per-vertex
        array cannot carry when faces share their first vertex (the two triangles of a quad).  Returns the 1-based BLAS index."""
        v = np.ascontiguousarray(np.asarray(verts, dtype=np.float32).reshape(-1, 3))
        nrm = np.ascontiguousarray(np.asarray(normals, dtype=np.float32).reshape(-1, 3))
        f = np.ascontiguousarray(np.asarray(faces, dtype=np.uint32).reshape(-1, 3))
        uv = None if uvs is None else np.ascontiguousarray(np.asarray(uvs, dtype=np.float32).reshape(-1, 2))
        fm = None if face_meta is None else np.ascontiguousarray(face_meta, dtype=np.uint32)
        if len(nrm) != len(v) or (uv is not None and len(uv) != len(v)) or (fm is not None and len(fm) != len(v)):
            raise ValueError("normals / uvs / face_meta must have one entry per vertex")
        blas_id = C.c_uint32()
        if metadata_per_face is not None:
            if fm is not None:
                raise ValueError("face_meta (per vertex) and metadata_per_face (per face) are alternatives")
            pf = np.ascontiguousarray(metadata_per_face, dtype=np.uint32)
            if len(pf) != len(f):
                raise ValueError("metadata_per_face must have one entry per face")
            check(lib().rc_add_mesh_face_metadata(self._h, ptr(v), ptr(nrm), ptr(uv), len(v), ptr(f), len(f), ptr(pf), C.byref(blas_id)))
        else:
            check(lib().rc_add_mesh(self._h, ptr(v), ptr(nrm), ptr(uv), len(v), ptr(f), len(f), ptr(fm), C.byref(blas_id)))
        self._prims_cache = None
        return blas_id.value + 1

    def update_mesh(self, handle, verts, faces, normals, uvs=None, face_meta=None):
        """update!(tlas, handle, new_mesh) (:808-857) for a decomposed mesh."""
        v = np.ascontiguousarray(np.asarray(verts, dtype=np.float32).reshape(-1, 3))
        nrm = np.ascontiguousarray(np.asarray(normals, dtype=np.float32).reshape(-1, 3))
        f = np.ascontiguousarray(np.asarray(faces, dtype=np.uint32).reshape(-1, 3))
        uv = None if uvs is None else np.ascontiguousarray(np.asarray(uvs, dtype=np.float32).reshape(-1, 2))
        fm = None if face_meta is None else np.ascontiguousarray(face_meta, dtype=np.uint32)
        if len(nrm) != len(v) or (uv is not None and len(uv) != len(v)) or (fm is not None and len(fm) != len(v)):
            raise ValueError("normals / uvs / face_meta must have one entry per vertex")
        check(lib().rc_update_geometry_mesh(self._h, handle.id, ptr(v), ptr(nrm), ptr(uv), len(v), ptr(f), len(f), ptr(fm)))
        self._prims_cache = None
        return handle

    def push_mesh(self, verts, faces, normals, transforms=None, uvs=None, face_meta=None, instance_id=0, instance_ids=None):
        """push!(tlas, mesh, transforms; instance_ids) (:639-676) for a decomposed mesh; returns the TLASHandle."""
        blas_index = self.add_mesh(verts, faces, normals, uvs, face_meta)
        if instance_ids is None and transforms is not None:
            n_inst = len(_as_xforms(transforms))
            instance_ids = np.full(n_inst, instance_id, np.uint32)
        elif instance_ids is None:
            instance_ids = np.array([instance_id], np.uint32)
        return self.push_instances(blas_index, transforms, instance_ids)

    def primary_rays_lookat_device(self, camera_pos, right, up, forward, half_width, half_height, width, height, d_rays, samples=1, seed=0,
                                   jitter=True, stream=None):
        """generate_primary_rays_lookat! (docs/src/wavefront-renderer.jl:219-254) into a device RTRay buffer of width*height*samples."""
        v = [np.ascontiguousarray(a, dtype=np.float32) for a in (camera_pos, right, up, forward)]
        check(lib().rc_primary_rays_lookat_device(self._h, ptr(v[0]), ptr(v[1]), ptr(v[2]), ptr(v[3]), float(half_width), float(half_height),
                                                  int(width), int(height), int(samples), int(seed), 1 if jitter else 0, ptr(d_rays), ptr(stream)))

    def reflection_rays_device(self, d_rays, d_hits, n, d_out, bias=0.01, stream=None):
        """Mirror-reflection rays from hits (generate_reflection_rays! with roughness 0; reflect, src/math.jl:80)."""
        check(lib().rc_reflection_rays_device(self._h, ptr(d_rays), ptr(d_hits), int(n), float(bias), ptr(d_out), ptr(stream)))

    def compact_hits_device(self, d_hits, n, d_indices, d_count, stream=None):
        """Ascending indices of the hit rays + their count (device u32): queue compaction between wavefront stages."""
        check(lib().rc_compact_hits_device(self._h, ptr(d_hits), int(n), ptr(d_indices), ptr(d_count), ptr(stream)))

    def shading_attributes_device(self, d_hits, n, d_normals=None, d_uvs=None, stream=None):
        """Interpolated shading normal / uv per hit on device buffers (docs/src/wavefront-renderer.jl:382-387)."""
        check(lib().rc_shading_attributes_device(self._h, ptr(d_hits), int(n), ptr(d_normals), ptr(d_uvs), ptr(stream)))

    def add_geometry_device(self, d_verts, n, d_meta=None):
        """build_and_append_blas! from device-resident soup (d_verts: device pointer to n x 9 f32); returns the 1-based BLAS index."""
        blas_id = C.c_uint32()
        check(lib().rc_add_blas_device(self._h, ptr(d_verts), ptr(d_meta) if d_meta else None, int(n), C.byref(blas_id)))
        self._prims_cache = None
        return blas_id.value + 1

    def delete(self, handle):  # delete!, :690-699
        d = C.c_int()
        check(lib().rc_delete(self._h, handle.id, C.byref(d)))
        self._prims_cache = None
        return bool(d.value)

    def update_transform(self, handle, transform):  # update_transform!, :755-770
        n = self.n_instances(handle) if self.is_valid(handle) else None
        if n is not None and n != 1:
            raise RaycoreError(_capi.RC_ERR_INVALID_ARGUMENT, f"Handle has {n} instances, use update_transforms! for multiple")
        xf = _as_xforms(transform)
        check(lib().rc_update_transforms(self._h, handle.id, ptr(xf), 1))

    def update_transforms(self, handle, transforms):  # update_transforms!, :784-797
        xf = _as_xforms(transforms)
        check(lib().rc_update_transforms(self._h, handle.id, ptr(xf), len(xf)))

    def update(self, handle, verts, meta=None):  # update!, :808-857
        verts = np.ascontiguousarray(np.asarray(verts, dtype=np.float32).reshape(-1, 9))
        m = None if meta is None else np.ascontiguousarray(meta, dtype=np.uint32)
        check(lib().rc_update_geometry(self._h, handle.id, ptr(verts), ptr(m), len(verts)))
        self._prims_cache = None

    def instance_buffer(self, handle):
        """instance_buffer(tlas, handle) (src/Raycore.jl:117-128): (device address, count) of the handle's 108-byte InstanceDescriptor
        records in the synced scene; rewrite them with your own kernels, then call refit_device()."""
        self.sync()
        p, n = C.c_void_p(), C.c_uint32()
        check(lib().rc_instance_buffer_device(self._h, handle.id, C.byref(p), C.byref(n)))
        return p.value, n.value

    def refit_device(self, recompute_inverse=True):
        """refit_tlas!(tlas) on descriptors rewritten in device memory: no host round trip."""
        check(lib().rc_refit_device(self._h, 1 if recompute_inverse else 0))
        return self

    def sync(self):  # sync!, :894-921
        a = C.c_int()
        check(lib().rc_sync(self._h, C.byref(a)))
        self.last_sync_action = ("noop", "refit", "rebuild")[a.value]
        if a.value == 2:
            self._prims_cache = None
        return self

    def adapt(self):  # Adapt.adapt(backend, tlas), :1085-1102: sync, then hand out the adapted form
        self.sync()
        return self._static

    @property
    def static_tlas(self):
        return self._static

    # -- queries ----------------------------------------------------------------------------------------
    def is_valid(self, handle):  # :524-526
        v = C.c_int()
        check(lib().rc_is_valid(self._h, handle.id, C.byref(v)))
        return bool(v.value)

    def _counts(self):
        c = [C.c_uint32() for _ in range(6)]
        check(lib().rc_counts(self._h, *[C.byref(x) for x in c]))
        return [x.value for x in c]

    def n_instances(self, handle=None):  # :533-537, :2391-2398
        if handle is None:
            return self._counts()[0]
        n = C.c_uint32()
        check(lib().rc_handle_instance_count(self._h, handle.id, C.byref(n)))
        return n.value

    def n_total_instances(self):  # :544
        return self._counts()[1]

    def n_geometries(self):  # :2405
        return self._counts()[2]

    def n_primitives(self):
        return self._counts()[3]

    def get_instances(self, handle):  # :732-738
        n = C.c_uint32()
        check(lib().rc_get_instances(self._h, handle.id, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=_capi.INSTANCE_DT)
        check(lib().rc_get_instances(self._h, handle.id, ptr(out), n.value, None))
        return out

    def get_instance(self, handle, instance_idx=1):  # :714-723
        inst = self.get_instances(handle)
        if not 1 <= instance_idx <= len(inst):
            raise RaycoreError(_capi.RC_ERR_INVALID_ARGUMENT, f"Instance index {instance_idx} out of range 1:{len(inst)}")
        return inst[instance_idx - 1]

    def world_bound(self):  # :2147-2149
        out = np.zeros(6, dtype=np.float32)
        check(lib().rc_world_bound(self._h, ptr(out)))
        return Bounds3(out[:3].copy(), out[3:].copy())

    def wait_for_gpu(self):  # wait_for_gpu!, :2418-2421
        check(lib().rc_wait(self._h))
        return self

    # -- batch tracing (device kernels) -------------------------------------------------------------------
    def set_option(self, name, value):
        check(lib().rc_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int64()
        check(lib().rc_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def trace(self, rays, mode="closest", out=None):
        """Batch closest_hit / any_hit: RAY_DT array in, HIT_DT array out (RTRay/RTHitResult, src/rt_transport.jl).  `out` reuses a
        HIT_DT array of the same length (a render loop saves the page faults of a fresh 32 B/ray array on every call)."""
        rays = _as_rays(rays)
        if out is None:
            hits = np.empty(len(rays), dtype=HIT_DT)  # every record is written by the library
        else:
            if out.dtype != HIT_DT or len(out) != len(rays) or not out.flags["C_CONTIGUOUS"]:
                raise ValueError("out must be a contiguous HIT_DT array with one record per ray")
            hits = out
        fn = lib().rc_trace_closest if mode == "closest" else lib().rc_trace_any
        check(fn(self._h, ptr(rays), ptr(hits), len(rays)))
        return hits

    def trace_device(self, d_rays, d_hits, n, mode="closest", stream=None):
        """Device-pointer form (e.g. torch tensors' data_ptr()); asynchronous on `stream`."""
        fn = lib().rc_trace_closest_device if mode == "closest" else lib().rc_trace_any_device
        check(fn(self._h, ptr(d_rays), ptr(d_hits), int(n), ptr(stream) if stream else None))

    def trace_device_batches(self, d_rays, d_hits, n, mode="closest", stream=None):
        """Several INDEPENDENT device batches in one call (rc_trace_*_device_batches): sequences of device pointers and ray counts.  The
        batches overlap on the scene's auxiliary streams, forked from and joined back into `stream`; asynchronous like trace_device."""
        k = len(n)
        if len(d_rays) != k or len(d_hits) != k:
            raise ValueError("d_rays, d_hits and n must have one entry per batch")
        rp = (C.c_void_p * k)(*[int(x) for x in d_rays])
        hp = (C.c_void_p * k)(*[int(x) for x in d_hits])
        nn = (C.c_uint64 * k)(*[int(x) for x in n])
        fn = lib().rc_trace_closest_device_batches if mode == "closest" else lib().rc_trace_any_device_batches
        check(fn(self._h, rp, hp, nn, k, ptr(stream) if stream else None))

    def hit_points_device(self, d_rays, d_hits, n, d_points, d_normals=None, stream=None):
        check(lib().rc_hit_points_device(self._h, ptr(d_rays), ptr(d_hits), int(n), ptr(d_points), ptr(d_normals) if d_normals else None,
                                         ptr(stream) if stream else None))

    def shadow_rays_device(self, d_rays, d_hits, n, light, d_shadow_rays, bias=0.01, stream=None):
        """generate_shadow_rays! for one point light (docs/src/wavefront-renderer.jl:288-333); output feeds trace_device(mode='any')."""
        lv = np.ascontiguousarray(light, dtype=np.float32)
        check(lib().rc_shadow_rays_device(self._h, ptr(d_rays), ptr(d_hits), int(n), ptr(lv), float(bias), ptr(d_shadow_rays),
                                          ptr(stream) if stream else None))

    def last_kernel_ms(self):
        ms = C.c_float()
        check(lib().rc_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def recent_kernel_ms(self, max_launches=47):
        """Durations (ms, oldest first) of the scene's most recent eager launches, from the events the launches carried themselves
        (rc_recent_kernel_ms): for timing a run of back-to-back launches without events between them."""
        out = (C.c_float * int(max_launches))()
        n = C.c_uint32(0)
        check(lib().rc_recent_kernel_ms(self._h, int(max_launches), out, C.byref(n)))
        return [float(out[i]) for i in range(n.value)]

    def host_register(self, array):
        """Page-lock a numpy array (rays, a reused `out=` hit array, triangle soup, a view-factor matrix) so the host-buffer calls move
        it by DMA at the full PCIe rate; undo with host_unregister.  The accel keeps a reference to the array while it is registered (an
        array collected while pinned would leave a stale pinned range behind); free() unregisters what is left."""
        check(lib().rc_host_register(self._h, ptr(array), array.nbytes))
        if not hasattr(self, "_registered"):
            self._registered = {}
        self._registered[array.ctypes.data] = array
        return array

    def host_unregister(self, array):
        check(lib().rc_host_unregister(self._h, ptr(array)))
        getattr(self, "_registered", {}).pop(array.ctypes.data, None)

    def intern_metadata(self, values):
        """Metadata words for `values`: the values themselves when they are all uint32-representable integers and the accel is still
        in uint32 mode (TMetadata = UInt32, the reference's default), otherwise 1-based indices into the accel's metadata table."""
        values = list(values)
        ints = all(isinstance(v, (int, np.integer)) and not isinstance(v, (bool, np.bool_)) and 0 <= int(v) <= 0xFFFFFFFF for v in values)
        if ints and self._meta_table is None:
            return np.array(values, dtype=np.uint32)
        if self._meta_table is None:
            if self.n_geometries() > 0:
                raise RaycoreError(_capi.RC_ERR_INVALID_ARGUMENT, "an accel's metadata type is fixed by its first geometry (Triangle{TMetadata})")
            self._meta_table = []
            self.meta_type = type(values[0]) if values else object
        base = len(self._meta_table)
        self._meta_table.extend(values)
        return np.arange(base + 1, base + 1 + len(values), dtype=np.uint32)

    def typed_metadata(self, word):
        """The metadata value a primitive's uint32 word stands for (identity in uint32 mode; 0 = the empty triangle's metadata)."""
        if self._meta_table is None:
            return word
        word = int(word)
        return self._meta_table[word - 1] if word >= 1 else None

    def eltype(self):  # Base.eltype(tlas), src/instanced-bvh.jl:2334-2345
        return ("Triangle", self.meta_type)

    def _prims(self):
        if self._prims_cache is None:
            self._prims_cache = self._static.all_blas_prims
        return self._prims_cache

    def _triangles(self):
        if self._prims_cache is None or getattr(self, "_tri_cache_for", None) is not self._prims_cache:
            self._tri_cache = self._static.all_blas_triangles
            self._tri_cache_for = self._prims()
        return self._tri_cache


def _owner(accel):
    if isinstance(accel, StaticTLAS):
        return accel._owner
    if isinstance(accel, TLAS):
        accel.sync()  # the reference requires adapt-per-dispatch; adapt == sync + read (:1085-1102)
        return accel
    raise TypeError("expected TLAS or StaticTLAS")


def sync(tlas):
    return tlas.sync()


def adapt(tlas):
    return tlas.adapt()


def world_bound(accel):
    return _owner(accel).world_bound() if isinstance(accel, StaticTLAS) else accel.world_bound()


def _tuple_from_hit(t, h, miss_prim):
    """(hit, Triangle, t, bary, instance_idx) exactly as closest_hit/any_hit return it (:2010-2023, :2106-2139)."""
    if not h["hit"]:
        return (False, miss_prim, np.float32(0), np.zeros(3, np.float32), np.uint32(0))
    u, v = h["bary_u"], h["bary_v"]
    w = (np.float32(1.0) - u) - v  # 1f0 - hit_u - hit_v (:2015)
    return (True, _typed(t, _triangle(t._triangles()[h["primitive_id"]])), h["t"], np.array([w, u, v], np.float32), np.uint32(h["instance_id"] + 1))


def _typed(t, tri):
    return tri if t._meta_table is None else tri._replace(metadata=t.typed_metadata(tri.metadata))


def closest_hit(accel, ray):
    """closest_hit(tlas, ray) -> (hit, primitive, distance, barycentric, instance_idx) (:1902-2024)."""
    t = _owner(accel)
    return _tuple_from_hit(t, t.trace(_as_rays(ray))[0], EMPTY_TRIANGLE)


def any_hit(accel, ray):
    """any_hit(tlas, ray) (:2034-2140); on a miss the dummy primitive is all_blas_prims[1] (:2137)."""
    t = _owner(accel)
    h = t.trace(_as_rays(ray), mode="any")[0]
    tris = t._triangles()
    dummy = _typed(t, _triangle(tris[0])) if len(tris) else EMPTY_TRIANGLE
    return _tuple_from_hit(t, h, dummy)


def trace_rays(accel, rays):
    """trace_rays(tlas, rays) = map(closest_hit) (ext/RaycoreMakieExt.jl:81-87), one device launch."""
    t = _owner(accel)
    return [_tuple_from_hit(t, h, EMPTY_TRIANGLE) for h in t.trace(_as_rays(rays))]


ContactPair = namedtuple("ContactPair", ["instance_a", "instance_b"])  # src/collision.jl:25-28 (1-based instance indices)
CollisionResult = namedtuple("CollisionResult", ["contacts", "num_contacts", "cache"])  # :40-44
CONTACT_DT = np.dtype([("instance_a", "<u4"), ("instance_b", "<u4")])


def collide_instances(tlas, cache=None):
    """collide_instances(tlas; cache) (src/collision.jl:189-233): all instance pairs with overlapping world AABBs, as a
    CONTACT_DT array (ContactPair bytes).  `cache` is accepted for signature parity; the library keeps its own buffer."""
    tlas.sync()
    n = C.c_uint64(0)
    check(lib().rc_collide_instances(tlas._h, None, 0, C.byref(n)))
    out = np.zeros(n.value, dtype=CONTACT_DT)
    if n.value:
        check(lib().rc_collide_instances(tlas._h, ptr(out), n.value, C.byref(n)))
    return CollisionResult(out, int(n.value), cache)


def collide_instances_any(tlas, handle_a, handle_b):
    """collide_instances_any(tlas, handle_a, handle_b) -> Bool (src/collision.jl:241-261)."""
    tlas.sync()
    ov = C.c_int(0)
    check(lib().rc_collide_instances_any(tlas._h, handle_a.id, handle_b.id, C.byref(ov)))
    return bool(ov.value)


class BLAS4:
    """BLAS4 (src/bvh4.jl:154-162): a 4-wide BVH over ONE geometry, traced in the geometry's own space (the reference has no
    instanced BVH4 path).  Built by build_blas4; owns a private scene holding the geometry's BVH2 and its collapse."""

    def __init__(self, scene, blas_id, n_nodes):
        self._scene, self._blas_id, self.num_interior = scene, blas_id, n_nodes  # num_interior = length(nodes4), :521
        self._prims_cache = None

    @property
    def nodes(self):
        out = np.zeros(self.num_interior, dtype=_capi.NODE4_DT)
        cnt = C.c_uint32(0)
        check(lib().rc_export_blas4_nodes(self._scene._h, self._blas_id, ptr(out), len(out), C.byref(cnt)))
        return out

    @property
    def primitives(self):
        if self._prims_cache is None:
            self._prims_cache = self._scene.adapt().all_blas_prims  # one geometry => its Morton-sorted primitives
        return self._prims_cache

    @property
    def root_aabb(self):
        d = self._scene.adapt().blas_descriptors[0]
        return Bounds3(d["root_min"].copy(), d["root_max"].copy())

    def trace(self, rays, mode="closest"):
        rays = _as_rays(rays)
        hits = np.zeros(len(rays), dtype=HIT_DT)
        fn = lib().rc_trace_closest4 if mode == "closest" else lib().rc_trace_any4
        check(fn(self._scene._h, self._blas_id, ptr(rays), ptr(hits), len(rays)))
        return hits

    def trace_device(self, d_rays, d_hits, n, mode="closest", stream=None):
        fn = lib().rc_trace_closest4_device if mode == "closest" else lib().rc_trace_any4_device
        check(fn(self._scene._h, self._blas_id, ptr(d_rays), ptr(d_hits), int(n), ptr(stream)))

    def last_kernel_ms(self):
        return self._scene.last_kernel_ms()


def build_blas4(verts, meta=None, device=0):
    """build_blas4(primitives) (src/bvh4.jl:511-522): LBVH (BVH2) build + collapse to BVH4, both on the device."""
    scene = TLAS(device)
    blas_index = scene.add_geometry(verts, meta)
    scene.push_instances(blas_index)  # identity instance: lets the private scene sync so primitives / root_aabb can be exported
    n = C.c_uint32(0)
    check(lib().rc_blas4_build(scene._h, blas_index - 1, C.byref(n)))
    return BLAS4(scene, blas_index - 1, n.value)


def _tuple4(blas, h, miss_prim):
    if not h["hit"]:
        return (False, miss_prim, np.float32(0), np.zeros(3, np.float32))
    p = blas.primitives[h["primitive_id"]]
    u, v = h["bary_u"], h["bary_v"]
    w = np.float32(1.0) - u - v  # :681
    return (True, Triangle(p["v"].copy(), p["meta"]), h["t"], np.array([w, u, v], np.float32))


def closest_hit4(blas, ray):
    """closest_hit4(blas::BLAS4, ray) -> (hit, primitive, distance, barycentric) (src/bvh4.jl:606-689)."""
    return _tuple4(blas, blas.trace(_as_rays(ray))[0], EMPTY_TRIANGLE)


def any_hit4(blas, ray):
    """any_hit4(blas::BLAS4, ray) (src/bvh4.jl:696-766); the miss dummy is primitives[1] (:763)."""
    p = blas.primitives[0]
    return _tuple4(blas, blas.trace(_as_rays(ray), mode="any")[0], Triangle(p["v"].copy(), p["meta"]))


def generate_ray_grid(accel, viewdir, grid_size):
    """generate_ray_grid as used by hits_from_grid (src/kernels.jl:10-72), computed on the device; returns RAY_DT rays."""
    import torch
    t = _owner(accel)
    n = grid_size * grid_size
    buf = torch.empty(n * 8, dtype=torch.float32, device=f"cuda:{t.device}")
    vd = np.ascontiguousarray(viewdir, dtype=np.float32)
    check(lib().rc_generate_ray_grid_device(t._h, ptr(vd), grid_size, ptr(buf.data_ptr()), None))
    torch.cuda.synchronize(t.device)
    return buf.cpu().numpy().view(RAY_DT).reshape(n)


RAYHIT_DT = np.dtype([("hit", "?"), ("point", "<f4", 3), ("metadata", "<u4")])  # RayHit{UInt32}, src/kernels.jl:1-5


def hits_from_grid(accel, viewdir, grid_size=32):
    """hits_from_grid (src/kernels.jl:58-72): a grid_size x grid_size array of RayHit records (RAYHIT_DT, indexed [i, j] like
    the Julia Matrix); point = sum_mul(bary, prim.vertices) in the primitive's local space, metadata = prim.metadata
    (a miss carries the zero triangle: point 0, metadata 0).  Always the one structured array: on an accel whose metadata type is
    not UInt32 the `metadata` field holds the interned words, and `typed_hit_metadata(accel, hits)` maps them to the values."""
    t = _owner(accel)
    rays = generate_ray_grid(t._static, viewdir, grid_size)
    hits = t.trace(rays)
    prims = t._prims()
    out = np.zeros(len(rays), dtype=RAYHIT_DT)
    m = hits["hit"] == 1
    out["hit"] = m
    if m.any():
        p = prims[hits["primitive_id"][m]]
        u, v = hits["bary_u"][m], hits["bary_v"][m]
        w = (np.float32(1) - u) - v
        # sum_mul (src/math.jl:52): a[1]*b[1] + a[2]*b[2] + a[3]*b[3], left to right, in Float32
        out["point"][m] = (w[:, None] * p["v"][:, 0] + u[:, None] * p["v"][:, 1]) + v[:, None] * p["v"][:, 2]
        out["metadata"][m] = p["meta"]
    return out.reshape(grid_size, grid_size, order="F")


def typed_hit_metadata(accel, hits):
    """RayHit{TMetadata}.metadata for an accel with a non-UInt32 metadata type: an object array shaped like `hits` holding the values
    the interned words stand for (None where the ray missed).  In UInt32 mode it is `hits["metadata"]` itself."""
    t = _owner(accel)
    if t._meta_table is None:
        return hits["metadata"]
    typed = np.empty(hits.shape, dtype=object)
    for idx in np.ndindex(hits.shape):
        typed[idx] = t.typed_metadata(hits["metadata"][idx]) if hits["hit"][idx] else None
    return typed


def get_centroid(accel, viewdir, grid_size=32):
    """get_centroid (src/kernels.jl:106-110): the hit points and their mean."""
    hits = hits_from_grid(accel, viewdir, grid_size)
    pts = hits["point"][hits["hit"]]
    return pts, (pts.mean(axis=0) if len(pts) else np.full(3, np.nan, np.float32))


def get_illumination(accel, viewdir, grid_size=1000):
    """get_illumination (src/kernels.jl:112-124): per-primitive hit counts of a grid_size^2 orthographic ray grid."""
    t = _owner(accel)
    out = np.zeros(t.n_primitives(), dtype=np.float32)
    vd = np.ascontiguousarray(viewdir, dtype=np.float32)
    check(lib().rc_get_illumination(t._h, ptr(vd), int(grid_size), ptr(out)))
    return out


def _vf_out(n, out):
    if out is None:
        return np.empty((n, n), dtype=np.uint32, order="F")  # every element is written by the library
    if out.dtype != np.uint32 or out.shape != (n, n) or not out.flags["F_CONTIGUOUS"]:
        raise ValueError("out must be an (N, N) uint32 array in column-major (Fortran) order, like Julia's Matrix")
    return out


def view_factors(accel, rays_per_triangle=10000, seed=0, out=None):
    """view_factors (src/kernels.jl:74-104): N x N UInt32 matrix, [src_meta, hit_meta] (column-major like Julia's Matrix).  Row chunks
    are traced while the finished ones travel to the host matrix, so the call costs about the PCIe time of 4 N^2 bytes.  `out` reuses
    a matrix (a render / solve loop saves the page faults of a fresh one)."""
    t = _owner(accel)
    out = _vf_out(t.n_primitives(), out)
    check(lib().rc_view_factors(t._h, int(rays_per_triangle), int(seed), ptr(out)))
    return out


VF_MODE_ROWS, VF_MODE_RAYS = 0, 1


def view_factors_multi(accels, rays_per_triangle=10000, seed=0, mode="rows", out=None):
    """view_factors on several devices of ONE process (rc_view_factors_multi): `accels` are synced accels holding the same geometry,
    one per device.  mode="rows": accel g traces matrix rows [gN/G, (g+1)N/G) and copies them into the host matrix over its own PCIe
    link; mode="rays": every device shoots rays_per_triangle / G rays of every source, RCCL ncclReduce over xGMI into the first
    device (BASELINE's north-star partition).  Same matrix either way, bit for bit."""
    owners = [_owner(a) for a in accels]
    n = owners[0].n_primitives()
    out = _vf_out(n, out)
    handles = (C.c_void_p * len(owners))(*[o._h for o in owners])
    check(lib().rc_view_factors_multi(handles, len(owners), int(rays_per_triangle), int(seed), ptr(out), {"rows": VF_MODE_ROWS, "rays": VF_MODE_RAYS}[mode]))
    return out


def view_factor_totals(accel, rays_per_triangle=10000, seed=0):
    """Per-triangle totals of the view-factor job WITHOUT the N x N matrix (rc_view_factor_totals): (received, emitted), uint64 vectors of
    length N with received[j] = sum(view_factors(...)[:, j]) -- the rays arriving at metadata j+1, what the reference's users compute from
    the matrix (docs/src/viewfactors_content.md:62-68) -- and emitted[i] = sum(view_factors(...)[i, :]).  Costs the tracing only."""
    t = _owner(accel)
    n = t.n_primitives()
    received, emitted = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    check(lib().rc_view_factor_totals(t._h, int(rays_per_triangle), int(seed), ptr(received), ptr(emitted)))
    return received, emitted


def view_factor_totals_multi(accels, rays_per_triangle=10000, seed=0):
    """The same totals with the RAYS sharded over several devices of one process (rc_view_factor_totals_multi): accel g shoots ray indices
    [g R / G, (g+1) R / G) of every source; one RCCL ncclReduce (uint64, 2 N elements) over xGMI into the first device when the accels sit
    on distinct devices, a host sum for replicas on one device.  Identical vectors for every G."""
    owners = [_owner(a) for a in accels]
    n = owners[0].n_primitives()
    received, emitted = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    handles = (C.c_void_p * len(owners))(*[o._h for o in owners])
    check(lib().rc_view_factor_totals_multi(handles, len(owners), int(rays_per_triangle), int(seed), ptr(received), ptr(emitted)))
    return received, emitted


def multi_prepare(accels):
    """Once per set of devices, before the *_multi calls that are timed (rc_multi_prepare): RCCL and its communicator, the scenes' auxiliary
    streams and staging vectors, a warm-up collective.  Returns the host milliseconds each item took and the number of RCCL ranks the
    accels form (0: replicas on one device, partial results are added on the host)."""
    owners = [_owner(a) for a in accels]
    handles = (C.c_void_p * len(owners))(*[o._h for o in owners])
    ms = np.zeros(4, np.float32)
    check(lib().rc_multi_prepare(handles, len(owners), ptr(ms)))
    ranks = C.c_int(0)
    check(lib().rc_multi_ranks(handles, len(owners), C.byref(ranks)))
    return {"comm_init_ms": float(ms[0]), "streams_and_buffers_ms": float(ms[1]), "warmup_collective_ms": float(ms[2]), "total_ms": float(ms[3]), "rccl_ranks": int(ranks.value)}


def trace_multi(accels, rays, mode="closest", out=None):
    """One host batch of rays on several devices of ONE process (rc_trace_closest_multi / rc_trace_any_multi): `accels` are synced accels
    holding the same scene, one per device; accel g uploads, traces and downloads the g-th contiguous shard of the batch over its own
    PCIe link (rays are independent -- replicas, no collective).  hits[i] is what any one accel's trace gives for rays[i]."""
    owners = [_owner(a) for a in accels]
    rays = _as_rays(rays)
    if out is None:
        hits = np.empty(len(rays), dtype=HIT_DT)
    else:
        if out.dtype != HIT_DT or len(out) != len(rays) or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a contiguous HIT_DT array with one record per ray")
        hits = out
    handles = (C.c_void_p * len(owners))(*[o._h for o in owners])
    fn = lib().rc_trace_closest_multi if mode == "closest" else lib().rc_trace_any_multi
    check(fn(handles, len(owners), ptr(rays), ptr(hits), len(rays)))
    return hits


def get_illumination_multi(accels, viewdir, grid_size=1000):
    """get_illumination with the ray grid cut into one share per device (rc_get_illumination_multi); the partial histograms are added
    on the host.  Same counts as get_illumination."""
    owners = [_owner(a) for a in accels]
    out = np.zeros(owners[0].n_primitives(), dtype=np.float32)
    vd = np.ascontiguousarray(viewdir, dtype=np.float32)
    handles = (C.c_void_p * len(owners))(*[o._h for o in owners])
    check(lib().rc_get_illumination_multi(handles, len(owners), ptr(vd), int(grid_size), ptr(out)))
    return out


def expand_faceviews(positions, position_faces, **attributes):
    """GeometryBasics.expand_faceviews as build_and_append_blas! uses it (src/instanced-bvh.jl:581-590): a mesh whose attributes are
    indexed by their own face arrays (a cube: 8 positions, 6 normals, one metadata value per face) becomes a mesh with ONE index set,
    which is what `add_mesh` / `rc_add_mesh` take.  Every face corner is the tuple of its indices into all attributes; each distinct
    tuple becomes one vertex, numbered in order of first appearance.

    attributes: name=(values, faces) with faces (nf, 3) indices into values, or name=(values, None) for one value per FACE (the
    `face_meta` convention, :595: after expansion it is per vertex).  Returns (positions, faces, {name: per-vertex values})."""
    pf = np.asarray(position_faces, dtype=np.int64).reshape(-1, 3)
    nf = len(pf)
    cols = [pf.reshape(-1)]
    names, values = [], []
    for name, (vals, faces) in attributes.items():
        vals = np.asarray(vals)
        if faces is None:
            if len(vals) != nf:
                raise ValueError(f"{name}: one value per face expected")
            idx = np.repeat(np.arange(nf, dtype=np.int64), 3)
        else:
            idx = np.asarray(faces, dtype=np.int64).reshape(-1)
            if len(idx) != 3 * nf:
                raise ValueError(f"{name}: faces must have one index triple per position face")
        cols.append(idx); names.append(name); values.append(vals)
    corners = np.stack(cols, axis=1)                                  # (3 nf, 1 + n_attributes) index tuples
    uniq, first, inverse = np.unique(corners, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")                          # first-appearance order, as GeometryBasics numbers the merged vertices
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    new_faces = rank[np.asarray(inverse).reshape(-1)].reshape(nf, 3).astype(np.uint32)
    tuples = uniq[order]
    out_pos = np.asarray(positions, dtype=np.float32).reshape(-1, 3)[tuples[:, 0]]
    out_attr = {name: vals[tuples[:, k + 1]] for k, (name, vals) in enumerate(zip(names, values))}
    return out_pos, new_faces, out_attr


def TLAS_from_items(items, metadata_fn, device=0):
    """TLAS(items, metadata_fn; backend) (src/instanced-bvh.jl:2276-2324): one BLAS + one identity instance per
    item, instance_id = item index, metadata = metadata_fn(item_idx, face_idx) (1-based); returns the adapted accel."""
    t = TLAS(device)
    for mi, verts in enumerate(items, start=1):
        verts = np.asarray(verts, dtype=np.float32).reshape(-1, 9)
        meta = t.intern_metadata([metadata_fn(mi, fi) for fi in range(1, len(verts) + 1)])  # TMetadata = typeof(metadata_fn(1, 1)), :2281-2282
        b = t.add_geometry(verts, meta)
        ident = np.array([[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]], dtype=np.float32)
        t.push_instances(b, ident, [mi], inv_transforms=ident)  # identity for both, :2314-2320
    return t.adapt()


def TLAS_from_meshes(meshes, device=0):
    """TLAS(meshes; backend) -> (tlas, handles) (src/instanced-bvh.jl:2361-2378)."""
    if len(meshes) == 0:
        raise RaycoreError(_capi.RC_ERR_INVALID_ARGUMENT, "Cannot create TLAS from empty mesh list")
    t = TLAS(device)
    handles = [t.push(m) for m in meshes]
    t.sync()
    return t, handles
