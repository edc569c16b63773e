/*
 * raycore_mi355x.h -- C ABI of the MI355X-native TLAS/BLAS traversal library (libraycore_mi355x.so).
 *
 * This is the drop-in boundary for ONE path of JuliaGeometry/Raycore.jl: the two-level BVH build,
 * closest_hit / any_hit traversal, and the get_illumination / view_factors drivers that sit on it.
 * The reference has no FFI of its own; its plug-in point is the Julia-level AbstractAccel contract
 * (src/Raycore.jl:14-49), already implemented by Raycore.TLAS and the external Lava.HWTLAS.  Each entry
 * point below names the reference function it replaces (paths relative to the reference repo); the
 * Julia `ccall` binding a maintainer would add is shown in INTEGRATION.md and
 * raycore.jl_amd/julia/RaycoreMI355X.jl.
 *
 * Conventions
 *  - Every function returns 0 on success, non-zero on error; rc_last_error() returns the message of the
 *    calling thread's last failure.  No C++ exception crosses this boundary.  The Julia wrapper turns a
 *    non-zero status into ErrorException (test/test_tlas_stress.jl:585-617 expects that type).
 *  - The caller owns every host buffer it passes.  The library owns all device memory behind the opaque
 *    rc_scene handle and frees it in rc_scene_destroy (replaces free!, src/instanced-bvh.jl:383-399).
 *  - Threading.  Mutations (rc_add_*, rc_update_*, rc_delete, rc_sync, rc_set_option, rc_blas4_build, rc_refit_device) on one scene
 *    must be externally serialised and must not overlap queries.  QUERIES ON A SYNCED SCENE ARE RE-ENTRANT: any number of host threads
 *    may call rc_trace_closest / rc_trace_any (each call stages through its own stream and buffers; a fifth concurrent call waits for a
 *    staging context) and the *_device entry points (rc_trace_*_device, rc_get_illumination_device, rc_view_factors_device,
 *    rc_trace_*4_device and the wavefront stage kernels) at once -- the reference's own drivers call closest_hit on one adapted accel
 *    from Threads.@threads (src/kernels.jl:64,82).  Only the enqueue of a launch is serialised inside the library (microseconds); the
 *    launches overlap on their streams, each with its own work counters and -- per stream -- its own stack spill area.  The other
 *    host-buffer queries (rc_get_illumination, rc_view_factors*, rc_collide_instances, rc_trace_*4) are safe to call concurrently and run
 *    one at a time per scene.  rc_compact_hits_device and rc_collide_instances_device use scene-owned scratch: keep their calls on one
 *    scene on one stream.  rc_last_kernel_ms reports the calling thread's latest launch.  No global mutable state.
 *  - hipGraph capture: trace / driver launches on a capturing stream are captured (no events, a counter slot of their own that eager
 *    launches never use); the stream must have run one eager launch on the scene before (stack spill area), see INTEGRATION.md.
 *    Entry points that allocate, copy or free (scene create / destroy / sync, the host-buffer queries) may run on any thread while a
 *    capture is open elsewhere in the process, also one in hipStreamCaptureModeGlobal: they set the calling thread's capture interaction
 *    mode to relaxed for their duration and touch only the scene's own streams.  rc_scene_destroy waits for the scene's own streams and
 *    for the launches it recorded on the callers' streams, not for the whole device.
 *  - There is NO CPU fallback: every compute entry point runs hand-written gfx950 HIP kernels and fails
 *    with RC_ERR_NO_DEVICE when no GPU is present.
 *  - Index bases at this boundary are 0-based (C); the host wrappers add 1 where the Julia API is 1-based.
 */
#ifndef RAYCORE_MI355X_H
#define RAYCORE_MI355X_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RC_OK 0
#define RC_ERR_INVALID_ARGUMENT 1
#define RC_ERR_INVALID_HANDLE 2   /* "Invalid handle" / "Handle has been deleted" (src/instanced-bvh.jl:715-716,756-757) */
#define RC_ERR_NO_DEVICE 3
#define RC_ERR_HIP 4
#define RC_ERR_EMPTY_GEOMETRY 5   /* "Geometry has no valid triangles" (src/instanced-bvh.jl:601) */
#define RC_ERR_NOT_SYNCED 6
#define RC_ERR_STACK_OVERFLOW 7

#define RC_INVALID_ID 0xFFFFFFFFu

/* Wire structs: byte-identical to RTRay / RTHitResult (src/rt_transport.jl:10-19, 33-42). */
typedef struct rc_ray {
    float origin_x, origin_y, origin_z, tmin;
    float dir_x, dir_y, dir_z, tmax;
} rc_ray;

/* hit: 1/0.  t: world-ray parameter (direction is NOT renormalised, src/instanced-bvh.jl:1974-1975).
 * primitive_id: 0-based index into the flat Morton-sorted primitive array (all_blas_prims).
 * instance_id: 0-based position in the instance array (Julia closest_hit returns this + 1).
 * instance_custom_index: InstanceDescriptor.instance_id of the hit instance.
 * bary_u/bary_v: Moeller-Trumbore u, v; Julia's bary = ((1-u)-v, u, v) (src/instanced-bvh.jl:2015-2016).
 * Miss: hit=0, t=0, bary 0, primitive_id=instance_id=RC_INVALID_ID (Julia: empty triangle, index 0). */
typedef struct rc_hit {
    uint32_t hit;
    float t;
    uint32_t primitive_id;
    uint32_t instance_custom_index;
    float bary_u, bary_v;
    uint32_t instance_id;
    uint32_t _pad;
} rc_hit;

/* Reference-layout records for introspection / parity checks (rc_export_*). */
typedef struct rc_bvh_node { /* BVHNode2, 60 bytes (src/instanced-bvh.jl:50-63) */
    float aabb0_min[3], aabb0_max[3], aabb1_min[3], aabb1_max[3];
    uint32_t child0, child1, parent;
} rc_bvh_node;
typedef struct rc_instance_desc { /* InstanceDescriptor, 108 bytes (src/instanced-bvh.jl:90-96) */
    uint32_t blas_index; /* 1-based, as stored by the reference */
    uint32_t instance_id;
    float transform[12];     /* Mat3x4f = Vulkan row-major 3x4 (src/instanced-bvh.jl:28-31) */
    float inv_transform[12];
    uint32_t flags;
} rc_instance_desc;
typedef struct rc_blas_desc { /* BLASDescriptor, 32 bytes (src/instanced-bvh.jl:132-136) */
    uint32_t nodes_offset, primitives_offset;
    float root_min[3], root_max[3];
} rc_blas_desc;
typedef struct rc_prim { /* the part of Triangle{UInt32} the path reads (src/triangle_mesh.jl:1-7) */
    float v[9];
    uint32_t meta;
} rc_prim;

typedef struct rc_scene rc_scene; /* opaque; plays the role of the mutable TLAS (src/instanced-bvh.jl:261-310) */

const char* rc_last_error(void);
/* Tracing (the reference's method: docs/src/hw_acceleration.md:198-218 times phases by hand; SURVEY.md section 5 asks for roctx ranges).  Every
 * entry point below is one roctx range named after itself while a marker library (rocprofiler-sdk's roctx, else libroctx64) can be loaded
 * and RC_ROCTX is not "0"; a caller brackets its own phases with these two, so that `rocprofv3 --marker-trace --kernel-trace` attributes
 * dispatches to phases.  Both always succeed (no-ops without a marker library); rc_ranges_enabled: 1 when ranges are being emitted. */
int rc_range_push(const char* name);
int rc_range_pop(void);
int rc_ranges_enabled(void);
/* Number of visible HIP devices (0 when none); never fails. */
int rc_device_count(void);

/* TLAS(backend) (src/instanced-bvh.jl:334-358).  device = HIP device ordinal. */
int rc_scene_create(int device, rc_scene** out);
/* free!(tlas) (src/instanced-bvh.jl:383-399) */
int rc_scene_destroy(rc_scene* scene);

/* build_and_append_blas! minus the GeometryBasics mesh decomposition (src/instanced-bvh.jl:581-608):
 * verts = n x 9 f32 triangle soup, meta = n u32 or NULL (=> face index 1..n, assigned BEFORE the
 * degenerate filter, :595).  Degenerate faces (is_degenerate, src/triangle_mesh.jl:14-17) are dropped,
 * the LBVH is built on the device (build_blas, :1376-1443).  *blas_id receives the 0-based geometry id; like the reference's
 * blas_index it is renumbered when an rc_sync after rc_delete drops unreferenced geometries (compact_instances!, :996-1065), so
 * use it right away (rc_add_instances, rc_blas4_build) or re-derive it from rc_get_instances(...).blas_index - 1. */
int rc_add_blas(rc_scene* scene, const float* verts, const uint32_t* meta, uint32_t n, uint32_t* blas_id);
/* Same with the triangle soup already in device memory (the reference builds BLASes "directly on the backend",
 * src/instanced-bvh.jl:16-21): no host staging; filter, sort and tree construction all run on the GPU. */
int rc_add_blas_device(rc_scene* scene, const float* d_verts, const uint32_t* d_meta, uint32_t n, uint32_t* blas_id);

/* push!(tlas, mesh, transforms; instance_ids) once the geometry exists (src/instanced-bvh.jl:639-676,
 * append_instances_with_handle! :612-623): m instances of blas_id.  xforms = m x 12 f32 (Mat3x4f bytes)
 * or NULL for identity; instance_ids = m u32 or NULL for 0.  *handle receives the TLASHandle id (>= 1). */
int rc_add_instances(rc_scene* scene, uint32_t blas_id, const float* xforms, const uint32_t* instance_ids,
                     uint32_t m, uint32_t* handle);
/* Same, but the caller supplies the inverse transforms too (the InstanceDescriptor(..., transform,
 * inv_transform, ...) constructor used with build_tlas, src/instanced-bvh.jl:90-102, 1605). */
int rc_add_instances_with_inverse(rc_scene* scene, uint32_t blas_id, const float* xforms, const float* inv_xforms,
                                  const uint32_t* instance_ids, uint32_t m, uint32_t* handle);

/* update_transform! / update_transforms! (src/instanced-bvh.jl:755-797): m must equal the handle's
 * instance count; marks the scene transforms-dirty (refit on the next rc_sync). */
int rc_update_transforms(rc_scene* scene, uint32_t handle, const float* xforms, uint32_t m);
/* update!(tlas, handle, new_geometry) (src/instanced-bvh.jl:808-857): replace the BLAS the handle uses. */
int rc_update_geometry(rc_scene* scene, uint32_t handle, const float* verts, const uint32_t* meta, uint32_t n);
/* delete!(tlas, handle) (src/instanced-bvh.jl:690-699): *deleted = 1 if it was live, 0 otherwise (idempotent). */
int rc_delete(rc_scene* scene, uint32_t handle, int* deleted);
/* is_valid (src/instanced-bvh.jl:524-526); n_instances(tlas, handle) (:533-537) */
int rc_is_valid(rc_scene* scene, uint32_t handle, int* valid);
int rc_handle_instance_count(rc_scene* scene, uint32_t handle, uint32_t* count);
/* get_instances(tlas, handle) (src/instanced-bvh.jl:732-738): copies the handle's descriptors. */
int rc_get_instances(rc_scene* scene, uint32_t handle, rc_instance_desc* out, uint32_t capacity, uint32_t* count);

/* sync!(tlas) (src/instanced-bvh.jl:894-921): dirty => compact + rebuild TLAS + flat BLAS arrays;
 * only transforms dirty => refit in place (refit_tlas!, :2197-2222); clean => no-op without a device
 * synchronise.  *action (optional) receives 0 = no-op, 1 = refit, 2 = rebuild. */
int rc_sync(rc_scene* scene, int* action);

/* n_instances(tlas) (live, :2391-2398), length(tlas.instances) (n_total_instances, :544), n_geometries (:2405),
 * number of flat primitives / TLAS nodes / BLAS nodes after the last sync. */
int rc_counts(rc_scene* scene, uint32_t* n_live_instances, uint32_t* n_total_instances, uint32_t* n_geometries,
              uint32_t* n_prims, uint32_t* n_tlas_nodes, uint32_t* n_blas_nodes);
/* world_bound(tlas) (src/instanced-bvh.jl:2147-2149): out = {min xyz, max xyz}. */
int rc_world_bound(rc_scene* scene, float out[6]);
/* wait_for_gpu! (src/instanced-bvh.jl:2418-2421).  Also the place where asynchronous launches (the *_device entry points) report a
 * traversal-stack overflow (RC_ERR_STACK_OVERFLOW): the host-buffer entry points check after every call, the device ones cannot. */
int rc_wait(rc_scene* scene);

/* Reference-layout copies of the synced StaticTLAS arrays (src/instanced-bvh.jl:155-168).  Pass NULL
 * buffers to query counts only. */
int rc_export_tlas_nodes(rc_scene* scene, rc_bvh_node* out, uint32_t capacity, uint32_t* count);
int rc_export_blas_nodes(rc_scene* scene, rc_bvh_node* out, uint32_t capacity, uint32_t* count);
int rc_export_instances(rc_scene* scene, rc_instance_desc* out, uint32_t capacity, uint32_t* count);
int rc_export_blas_descs(rc_scene* scene, rc_blas_desc* out, uint32_t capacity, uint32_t* count);
int rc_export_prims(rc_scene* scene, rc_prim* out, uint32_t capacity, uint32_t* count);

/* closest_hit / any_hit over a batch (src/instanced-bvh.jl:1902-2024, 2034-2140; batch shape =
 * trace_rays, ext/RaycoreMakieExt.jl:81-87, and Lava.trace_closest_hits!, docs/src/hw_acceleration.md:141-146).
 * Host-buffer form: copies rays in, hits out. */
int rc_trace_closest(rc_scene* scene, const rc_ray* rays, rc_hit* hits, uint64_t n);
int rc_trace_any(rc_scene* scene, const rc_ray* rays, rc_hit* hits, uint64_t n);
/* Device-buffer form: d_rays / d_hits are device pointers on the scene's device; the launch is enqueued
 * on `stream` (a hipStream_t, NULL = the null stream) and is asynchronous.  Launches on different streams
 * may be in flight together (each has its own work counters); for batches of about a million rays that is
 * worth 1.4x: the end of a launch -- waves waiting for their longest rays -- leaves most of the machine to
 * the next one (mid-size batches). */
int rc_trace_closest_device(rc_scene* scene, const rc_ray* d_rays, rc_hit* d_hits, uint64_t n, void* stream);
int rc_trace_any_device(rc_scene* scene, const rc_ray* d_rays, rc_hit* d_hits, uint64_t n, void* stream);
/* Several INDEPENDENT device batches in one call: d_rays / d_hits / n are HOST arrays of n_batches device pointers / counts.  The
 * reference's batch API takes one ray array per call (trace_rays(tlas, rays), ext/RaycoreMakieExt.jl:81-87; Lava.trace_closest_hits!,
 * docs/src/hw_acceleration.md:141-146), which gives a caller holding several mid-size batches -- tiles of a frame, the queues of a
 * wavefront tracer (docs/src/wavefront-renderer.jl:260-512) -- no way to say that they do not depend on each other.  Here the batches
 * go round-robin onto four auxiliary streams of the scene, forked from `stream` and joined back into it with events: to the caller the
 * call is one asynchronous operation on `stream`, and each batch's hits are exactly what rc_trace_*_device gives for it.  Worth ~1.3x
 * for batches of about a million rays (the tail of one launch overlaps the bulk of the next); nothing for batches that fill the machine
 * for long.  Capture-safe: on a capturing `stream` the auxiliary streams join the capture through the fork event and are all joined back
 * before the call returns (also when a launch fails); each captured launch holds a capture slot like a single captured launch does. */
int rc_trace_closest_device_batches(rc_scene* scene, const rc_ray* const* d_rays, rc_hit* const* d_hits, const uint64_t* n, int n_batches, void* stream);
int rc_trace_any_device_batches(rc_scene* scene, const rc_ray* const* d_rays, rc_hit* const* d_hits, const uint64_t* n, int n_batches, void* stream);

/* Kernel selection for the trace entry points (tuning / A-B measurement).  The default is the tuned
 * kernel; every variant returns identical results.  Names: "kernel" (-1 auto, 0..6),
 * "blocks_per_cu", "lds_stack", "refill", "sched_thr", "pool", "claim_shards" (scheduling knobs of the
 * persistent kernels), "taper" (guided claim sizes: towards the end of the claim order a 128-ray chunk is dealt in halves, quarters,
 * eighths; in eighths of (part size x waves) still to hand out per piece, default 12, 0 = whole chunks only), "cost_order" (1 = the
 * chunks that held long-lived rays in earlier launches of the same BATCH are claimed first: the batch is recognised on the device by sample
 * rays among up to four remembered per launch shape -- chunk count, mode, stream --, records its chunk costs in its launches 2-4 (never the first: a batch that does not come back pays nothing) and
 * then in one launch of eight; default 1),
 * "cost_thr" (its initial reporting threshold), "entry_cull" (1 = an instance whose conservative sphere the ray's segment misses is
 * not entered: the reference's traversal of it would test no triangle; 0 = off, 1 = closest_hit and the drivers (default),
 * 2 = any_hit batches too), "vf_chunk_bytes" (device block per row chunk of the host-matrix view factors), "vf_first_touch" (ROWS on several devices: each device's
 * host thread joins the device's NUMA node and faults in its own row block; default 1), "release_captures" (set to 1 when the hipGraphs that
 * captured launches of this scene have been destroyed: frees their stack spill regions and counter slots, 16 per scene; get: how many are held),
 * "blas_top" (1 = a scene with a single BLAS keeps that BLAS's top internal
 * nodes in LDS; takes effect at the next structural rc_sync), "onesweep_min" (key count from which
 * the builds sort with Onesweep radix passes instead of a merge sort), "stats" (dev counters),
 * "timeline_ptr" (dev: device address of 8 x u64 per wave that kernel 5 fills with its waves' event
 * times, tools/archive/timeline_probe.py; 0 = off).
 * None of them changes a result: every variant and every claim order returns identical hits.
 * Read-only: "n_cus", "blas_top_k", "tlas_top_k", "claim_drift", "stat0".."statf", "stat16".."stat23". */
int rc_set_option(rc_scene* scene, const char* name, int64_t value);
int rc_get_option(rc_scene* scene, const char* name, int64_t* value);

/* generate_ray_grid as used by hits_from_grid (src/kernels.jl:10-72): grid*grid rays written to the
 * device buffer d_rays (column-major (i-1) + grid*(j-1)), direction = normalize(viewdir). */
int rc_generate_ray_grid_device(rc_scene* scene, const float viewdir[3], uint32_t grid, rc_ray* d_rays, void* stream);
/* get_illumination(tlas, viewdir; grid_size) (src/kernels.jl:112-124): out_counts = n_prims f32 (host). */
int rc_get_illumination(rc_scene* scene, const float viewdir[3], uint32_t grid, float* out_counts);
/* Rays [ray_begin, ray_end) of the grid only, accumulated into a device histogram (n_prims f32) --
 * the shard unit for multi-GPU runs.  The kernel counts into private copies of the histogram (a few large, well-lit
 * triangles take most of the hits) and adds them to d_counts ATOMICALLY at the end of the launch: shards add up whether they are enqueued
 * on one stream or run concurrently on several.  A count saturates at 2^24 like the reference's f32 `+= 1`. */
int rc_get_illumination_device(rc_scene* scene, const float viewdir[3], uint32_t grid, uint64_t ray_begin,
                               uint64_t ray_end, float* d_counts, void* stream);

/* view_factors(tlas; rays_per_triangle) (src/kernels.jl:74-104) with the unseeded task-local RNG replaced
 * by Philox4x32-10: key = seed, counter = (ray_idx, src_prim_index, 0, 0) -> (r1, r2, xi1, xi2), so a result
 * does not depend on how the work is sharded.  Shoots rays [ray_begin, ray_end) of source primitives
 * [src_begin, src_end) and ACCUMULATES into d_matrix (device, u32): element (src_meta-1, hit_meta-1) lives
 * at (row - row_offset)*row_stride + (hit_meta-1)*col_stride with row = src_meta-1, or -- with
 * RC_VF_ROW_BY_PRIMITIVE in flags -- row = the source's index in the flat (Morton-sorted) primitive array,
 * which is what makes a contiguous source range own a contiguous row block.  Julia's column-major N x N
 * Matrix is row_stride=1, col_stride=N, row_offset=0, flags=0; a row-sharded block is row_stride=N,
 * col_stride=1, row_offset=src_begin, flags=RC_VF_ROW_BY_PRIMITIVE (the gatherer then permutes rows by
 * metadata once).  With RC_VF_SOURCES_BY_METADATA, [src_begin, src_end) are positions in the primitives' order
 * by ascending metadata (ties by flat index) and row = that position: when the metadata are a permutation of
 * 1..N -- the view-factor convention, src/kernels.jl:85 -- position p IS matrix row src_meta-1, so a contiguous
 * source range owns a contiguous block of FINAL rows and a multi-GPU gather / reduce needs no row permutation
 * and no second N x N buffer.
 * RC_VF_ROW_BY_METADATA_VALUE (with RC_VF_SOURCES_BY_METADATA): sources are addressed by position in the metadata order, but
 * row = src_meta-1 as without flags -- for metadata that are not a permutation of 1..N (duplicates share a row, gaps leave rows
 * empty) a range of matrix rows is still a contiguous range of positions, which is how the host-matrix entry points cut the job
 * into row chunks and row blocks. */
#define RC_VF_ROW_BY_PRIMITIVE 1u
#define RC_VF_SOURCES_BY_METADATA 2u
#define RC_VF_ROW_BY_METADATA_VALUE 4u
int rc_view_factors_device(rc_scene* scene, uint32_t rays_per_triangle, uint64_t seed, uint32_t src_begin,
                           uint32_t src_end, uint32_t ray_begin, uint32_t ray_end, uint32_t* d_matrix,
                           uint64_t row_stride, uint64_t col_stride, uint32_t row_offset, uint32_t flags, void* stream);
/* Whole job into a host N x N column-major matrix (the Julia return value, src/kernels.jl:74-78).  The matrix never exists on the
 * device: row chunks (192 MB blocks) are traced on alternating streams while the finished ones travel to out_matrix as 2-D copies,
 * so the call lasts about as long as N*N*4 bytes need over the PCIe link (C5: 10 GB, ~0.18 s; the tracing, 41 ms, hides inside).
 * out_matrix may be pageable or registered (rc_host_register); every element is written. */
int rc_view_factors(rc_scene* scene, uint32_t rays_per_triangle, uint64_t seed, uint32_t* out_matrix);
/* Matrix rows [row_begin, row_end) only, into a host column-major matrix with leading dimension ld >= N (element (r, c) at
 * out_matrix[r + ld*c]): the unit of a multi-PROCESS run -- every rank maps the same shared-memory matrix and brings its own row
 * block home over its own PCIe link (raycore.jl_amd/distributed.py, mode "rows_host"). */
int rc_view_factors_rows_host(rc_scene* scene, uint32_t rays_per_triangle, uint64_t seed, uint32_t row_begin, uint32_t row_end,
                              uint32_t* out_matrix, uint64_t ld);
/* The same job on several devices of one process: scenes[g] is a synced scene on device g's ordinal, all built from the same
 * geometry (view_factors shards across the GPUs of a node).
 *   RC_VF_MODE_ROWS: scene g traces matrix rows [g N / G, (g+1) N / G) and copies its chunks straight into out_matrix -- G PCIe links
 *     in parallel, nothing crosses xGMI, no collective.  Several scenes may share a device.
 *   RC_VF_MODE_RAYS: scene g shoots ray indices [g R / G, (g+1) R / G) of every source into a full device accumulator; row chunks are
 *     summed into scenes[0]'s device with RCCL (ncclReduce, ncclUint32, over xGMI; librccl.so is loaded on first use) while the next
 *     chunk is traced, and leave from there.  Needs distinct devices.
 * Both give the matrix rc_view_factors gives, bit for bit (Philox is keyed by ray index and source primitive). */
#define RC_VF_MODE_ROWS 0
#define RC_VF_MODE_RAYS 1
int rc_view_factors_multi(rc_scene* const* scenes, int n_scenes, uint32_t rays_per_triangle, uint64_t seed, uint32_t* out_matrix, int mode);
/* The per-triangle TOTALS of the view-factor job, without the matrix: out_received[j] = sum_i result[i, j] (the column sums -- how many
 * rays ARRIVE at the triangles with metadata j+1; the quantity the reference's users read off the matrix,
 * docs/src/viewfactors_content.md:62-68: sum(view(viewf_matrix, :, i))) and out_emitted[i] = sum_j result[i, j] (the row sums: counted
 * rays that LEFT triangles with metadata i+1).  Same rays (Philox keyed by seed, ray index, source primitive) and the same counting
 * rule as view_factors! (src/kernels.jl:93-97), so the vectors equal the sums of rc_view_factors' matrix exactly; u64, n_prims each;
 * either pointer may be NULL.  No N x N array exists anywhere: the call costs the tracing (C5: ~40 ms), not 10 GB over PCIe.
 * _device: rays [ray_begin, ray_end) of the sources with flat primitive indices [src_begin, src_end), ACCUMULATED into device vectors
 *   (the shard unit of a multi-process run: torch.distributed reduce of 2 N int64).  The kernel counts into private copies of the received
 *   vector (a few large triangles receive most rays of a closed scene, and per-ray atomics onto a few hundred cache lines run at half the
 *   rate of scattered ones) and adds them to d_received atomically at the end of the launch; d_emitted takes one atomic per wave and source:
 *   shards accumulate correctly on one stream or concurrently on several.
 * _multi: scenes[g] = synced copies of one scene on DISTINCT devices; device g shoots ray indices [g R / G, (g+1) R / G) of every
 *   source and ONE ncclReduce (ncclUint64, sum, 2 N elements, over xGMI; librccl.so loaded on first use) brings the vectors to
 *   scenes[0]'s device -- "rays sharded across the GPUs with an RCCL reduce of the per-triangle accumulators" (SURVEY.md 8e).  Scenes
 *   that share a device are replicas without a communicator: their partial vectors are added on the host. */
int rc_view_factor_totals(rc_scene* scene, uint32_t rays_per_triangle, uint64_t seed, uint64_t* out_received, uint64_t* out_emitted);
int rc_view_factor_totals_device(rc_scene* scene, uint32_t rays_per_triangle, uint64_t seed, uint32_t src_begin, uint32_t src_end,
                                 uint32_t ray_begin, uint32_t ray_end, uint64_t* d_received, uint64_t* d_emitted, void* stream);
int rc_view_factor_totals_multi(rc_scene* const* scenes, int n_scenes, uint32_t rays_per_triangle, uint64_t seed, uint64_t* out_received,
                                uint64_t* out_emitted);
/* Everything the *_multi entry points need ONCE per set of devices, taken out of the calls that are timed: librccl.so and
 * the communicator over the scenes' devices (ncclCommInitAll, when they are distinct), each scene's auxiliary streams, its staging vectors
 * for the totals, and a first tiny collective that makes RCCL set up its xGMI connections.  Optional -- the *_multi calls do the same
 * lazily -- and idempotent.  out_ms (may be NULL) = 4 floats of host wall time: [0] load RCCL + create the communicator, [1] streams and
 * staging buffers, [2] the warm-up collective (0 without a communicator), [3] total.  rc_multi_ranks: how many RCCL ranks the scenes form
 * (0 when they share a device: replicas, partial results added on the host). */
int rc_multi_prepare(rc_scene* const* scenes, int n_scenes, float* out_ms);
int rc_multi_ranks(rc_scene* const* scenes, int n_scenes, int* out_ranks);
/* closest_hit / any_hit over one HOST batch on several devices of one process (rays are independent -- replicas of the
 * scene, contiguous ray shards, no collective): scenes[g] is a synced copy of the same scene on device g; shard g is uploaded, traced
 * and downloaded by device g over its own PCIe link, which is what bounds a host-to-host batch (64 bytes per ray).  hits[i] is what
 * rc_trace_closest / rc_trace_any on any one of the scenes gives for rays[i]. */
int rc_trace_closest_multi(rc_scene* const* scenes, int n_scenes, const rc_ray* rays, rc_hit* hits, uint64_t n);
int rc_trace_any_multi(rc_scene* const* scenes, int n_scenes, const rc_ray* rays, rc_hit* hits, uint64_t n);
/* get_illumination on several devices: device g traces its share of the grid's rays into its own histogram; the n_prims-long partial
 * histograms are added on the host (a count stops at 2^24 like the reference's f32 `+= 1`).  Same result as rc_get_illumination. */
int rc_get_illumination_multi(rc_scene* const* scenes, int n_scenes, const float viewdir[3], uint32_t grid, float* out_counts);
/* The rays view_factors shoots for one source primitive (ray indices [ray_begin, ray_begin + n_rays)),
 * written to a device buffer: the body of the reference's inner loop up to the Ray constructor
 * (src/kernels.jl:89-92), exposed so ray generation can be checked on its own. */
int rc_view_factor_rays_device(rc_scene* scene, uint64_t seed, uint32_t src_prim, uint32_t ray_begin, uint32_t n_rays,
                               rc_ray* d_rays, void* stream);

/* Wavefront stages next to the trace (the reference's fastest renderer is a wavefront pipeline around closest_hit /
 * any_hit, docs/src/wavefront-renderer.jl:260-362).  All buffers are device pointers; slot i belongs to ray i.
 * rc_hit_points_device: hit_point = ray.o + ray.d * t (:302) and the geometric world-space normal of the hit primitive
 * (normalize(cross(v1-v0, v2-v0)) through the instance's inverse-transpose, flipped to face the ray origin); zeros on a
 * miss.  d_points / d_normals: n x 3 f32, d_normals may be NULL.
 * rc_shadow_rays_device: generate_shadow_rays! for one point light (:288-333): origin = hit_point + normal * bias,
 * direction = normalize(light - origin), t_max = distance; misses get the reference's dummy ray (d = (0,0,1), t_max = 0).
 * The output feeds rc_trace_any_device unchanged. */
int rc_hit_points_device(rc_scene* scene, const rc_ray* d_rays, const rc_hit* d_hits, uint64_t n, float* d_points,
                         float* d_normals, void* stream);
int rc_shadow_rays_device(rc_scene* scene, const rc_ray* d_rays, const rc_hit* d_hits, uint64_t n, const float light[3],
                          float bias, rc_ray* d_shadow_rays, void* stream);

/* ---- BVH4 (src/bvh4.jl; exported by the reference as BVHNode4 / BLAS4 / build_blas4 / closest_hit4 / any_hit4).
 * BLAS-level only, as in the reference: rays are traced in the geometry's own space, no instances.
 * rc_bvh4_node = BVHNode4, 120 bytes (src/bvh4.jl:40-69): interior nodes hold 1-based BVH4 child indices, a leaf
 * (child_count == 0) holds the 1-based index of its triangle in the BLAS's Morton-sorted primitive array in child[0]. */
typedef struct rc_bvh4_node {
    uint32_t child[4];
    float aabb[4][2][3]; /* aabbK_min, aabbK_max */
    uint32_t parent;
    uint8_t child_count, primitive_count, _pad1, _pad2;
} rc_bvh4_node;
/* build_blas4 (src/bvh4.jl:511-522) for geometry blas_id (from rc_add_blas*): collapses its binary LBVH into BVHNode4s on
 * the device (collapse_bvh2_to_bvh4 :314-497, same node numbering).  *n_nodes (optional) receives the node count. */
int rc_blas4_build(rc_scene* scene, uint32_t blas_id, uint32_t* n_nodes);
/* Copies the BLAS4's nodes out in the reference layout (count query when out == NULL). */
int rc_export_blas4_nodes(rc_scene* scene, uint32_t blas_id, rc_bvh4_node* out, uint32_t capacity, uint32_t* count);
/* closest_hit4 (src/bvh4.jl:606-689) / any_hit4 (:696-766) over a host ray batch.  Both ignore rc_ray.tmin (the
 * reference starts from t_min = 0, :610/:700).  rc_hit: primitive_id = 0-based index into the BLAS's sorted primitives,
 * instance_id = RC_INVALID_ID, instance_custom_index = 0; miss as rc_trace_closest. */
int rc_trace_closest4(rc_scene* scene, uint32_t blas_id, const rc_ray* rays, rc_hit* hits, uint64_t n);
int rc_trace_any4(rc_scene* scene, uint32_t blas_id, const rc_ray* rays, rc_hit* hits, uint64_t n);
/* Same on device buffers, enqueued on `stream` (NULL = default stream), no host synchronisation. */
int rc_trace_closest4_device(rc_scene* scene, uint32_t blas_id, const rc_ray* d_rays, rc_hit* d_hits, uint64_t n, void* stream);
int rc_trace_any4_device(rc_scene* scene, uint32_t blas_id, const rc_ray* d_rays, rc_hit* d_hits, uint64_t n, void* stream);

/* ---- collision broad phase (src/collision.jl) ------------------------------------------------------------------------
 * rc_contact_pair = ContactPair (:25-28): 1-based instance indices, instance_a < instance_b. */
typedef struct rc_contact_pair {
    uint32_t instance_a, instance_b;
} rc_contact_pair;
/* collide_instances(tlas) (:189-233) on a synced scene: all pairs of instances whose world AABBs overlap, found by one
 * TLAS walk per instance on the device (count pass, prefix sum, write pass).  *count receives the number of pairs; they
 * are written (in the reference algorithm's order) when out != NULL and capacity >= *count, else only counted.
 * rc_collide_instances copies to a host buffer, _device writes a device buffer on `stream` (the total is read back on
 * the host either way, as in the reference, :218). */
int rc_collide_instances(rc_scene* scene, rc_contact_pair* out, uint64_t capacity, uint64_t* count);
int rc_collide_instances_device(rc_scene* scene, rc_contact_pair* d_out, uint64_t capacity, uint64_t* count, void* stream);
/* collide_instances_any(tlas, handle_a, handle_b) (:241-261): do any two instances of the two handles overlap (AABB)?
 * Restated as written: the reference reads instance i's box from TLAS leaf position n-1+i (Morton-sorted order). */
int rc_collide_instances_any(rc_scene* scene, uint32_t handle_a, uint32_t handle_b, int* overlap);

/* ---- mesh ingestion and the full Triangle record -----------------------------------------------------------------------
 * rc_triangle = Triangle{UInt32}, 136 bytes (src/triangle_mesh.jl:1-7): what closest_hit returns by value. */
typedef struct rc_triangle {
    float vertices[3][3], normals[3][3], tangents[3][3], uv[3][2];
    uint32_t metadata;
} rc_triangle;
/* build_and_append_blas! after the GeometryBasics decomposition (src/instanced-bvh.jl:581-608): the caller passes what
 * expand_faceviews / decompose / decompose_normals / decompose_uv produced -- nv vertices (verts, normals: nv x 3 f32; uvs:
 * nv x 2 f32 or NULL), nf faces as 0-based index triples, and optionally the per-vertex face_meta array (the reference
 * reads face_meta[first vertex of the face], :595; NULL => face index 1..nf, assigned before the degenerate filter).
 * Expansion, is_degenerate_face filtering (:573-577), build_triangle (:555-566) and the LBVH build run on the device. */
int rc_add_mesh(rc_scene* scene, const float* verts, const float* normals, const float* uvs, uint32_t nv,
                const uint32_t* indices, uint32_t nf, const uint32_t* face_meta, uint32_t* blas_id);
/* The same with ONE METADATA WORD PER FACE (nf words): the TLAS(items, metadata_fn) constructor evaluates metadata_fn(mesh_idx, face_idx)
 * for every face (src/instanced-bvh.jl:2300-2306), which a per-vertex array cannot carry when faces share their first vertex (the two
 * triangles of a quad, the fans of a sphere).  Assigned before the degenerate filter, like the face index. */
int rc_add_mesh_face_metadata(rc_scene* scene, const float* verts, const float* normals, const float* uvs, uint32_t nv,
                              const uint32_t* indices, uint32_t nf, const uint32_t* metadata_per_face, uint32_t* blas_id);
/* update!(tlas, handle, new_mesh) (src/instanced-bvh.jl:808-857) for a decomposed mesh: replaces the BLAS the handle uses. */
int rc_update_geometry_mesh(rc_scene* scene, uint32_t handle, const float* verts, const float* normals, const float* uvs, uint32_t nv,
                            const uint32_t* indices, uint32_t nf, const uint32_t* face_meta);
/* all_blas_prims as full Triangles (synced scene; count query when out == NULL).  Geometry that came in as plain soup
 * (rc_add_blas*) has no mesh attributes: normals = geometric normal normalize((v1-v0) x (v2-v0)), uv = the reference's
 * default (0,0),(1,0),(1,1) (:561-565); tangents are NaN as in build_triangle. */
int rc_export_triangles(rc_scene* scene, rc_triangle* out, uint32_t capacity, uint32_t* count);
/* Shading epilogue next to the trace (docs/src/wavefront-renderer.jl:382-387), device buffers: per hit the interpolated
 * normal normalize(n0*b1 + n1*b2 + n2*b3) (d_normals, n x 3 f32) and uv0*b1 + uv1*b2 + uv2*b3 (d_uvs, n x 2 f32) with
 * (b1, b2, b3) = ((1-u)-v, u, v); zeros on a miss.  Either output may be NULL.  Saves returning 136-byte Triangles. */
int rc_shading_attributes_device(rc_scene* scene, const rc_hit* d_hits, uint64_t n, float* d_normals, float* d_uvs, void* stream);

/* More wavefront stages (device buffers, enqueued on `stream`):
 * rc_primary_rays_lookat_device: generate_primary_rays_lookat! (docs/src/wavefront-renderer.jl:219-254): width*height*samples
 * rays, ray (pixel-1)*samples + s for the row-major pixel (x, y); u = 2(x - 0.5 + j1)/width - 1, v = 1 - 2(y - 0.5 + j2)/height,
 * d = normalize(forward + right*(u*half_width) + up*(v*half_height)), t_min 0, t_max Inf.  The reference draws the jitter
 * from rand(Vec2f); here it is Philox4x32-10 keyed by `seed` with the ray index as counter (jitter != 0) or the pixel
 * centre (jitter == 0).
 * rc_compact_hits_device: ascending indices of the rays whose hit flag is set (d_indices, capacity n) and their number
 * (*d_count, a device u32) -- the queue compaction between stages; the reference's demo keeps dummy rays instead (:445). */
int rc_primary_rays_lookat_device(rc_scene* scene, const float camera_pos[3], const float camera_right[3], const float camera_up[3],
                                  const float camera_forward[3], float half_width, float half_height, uint32_t width, uint32_t height,
                                  uint32_t samples, uint64_t seed, int jitter, rc_ray* d_rays, void* stream);
int rc_compact_hits_device(rc_scene* scene, const rc_hit* d_hits, uint64_t n, uint32_t* d_indices, uint32_t* d_count, void* stream);
/* generate_reflection_rays! for perfect mirrors (docs/src/wavefront-renderer.jl:431-476, roughness 0) on reflect (src/math.jl:80):
 * per hit, origin = hit_point + shading_normal * bias, direction = reflect(-ray.d, shading_normal), t_max = Inf; misses get the
 * reference's dummy ray (d = (0,0,1), t_max = 0).  Material tests (metallic / roughness) stay with the caller. */
int rc_reflection_rays_device(rc_scene* scene, const rc_ray* d_rays, const rc_hit* d_hits, uint64_t n, float bias, rc_ray* d_out, void* stream);

/* Scene files.  The reference has no on-disk format; this one keeps what a rebuild would recompute (per geometry: sorted
 * primitives, BVH2 nodes, mesh attributes; plus instance descriptors and the handle table).  rc_scene_save needs a synced
 * scene; rc_scene_load returns an unsynced scene whose handles are the saved ones -- rc_sync rebuilds the TLAS, after
 * which it traces bit-identically to the saved scene. */
int rc_scene_save(rc_scene* scene, const char* path);
int rc_scene_load(int device, const char* path, rc_scene** out);

/* instance_buffer(tlas, handle) (a stub for GPU back ends in the reference, src/Raycore.jl:117-128) + refit_tlas!
 * (src/instanced-bvh.jl:2197-2222): *d_descs receives the device address of the handle's InstanceDescriptor records (108 bytes
 * each, rc_instance_desc layout) inside the synced scene.  The caller's own kernels may rewrite `transform` (and
 * `inv_transform`) there; rc_refit_device then commits: recompute_inverse != 0 => inv_transform = mat3x4_inverse(transform) is
 * recomputed on the device for every instance first; then instance boxes, TLAS refit and traversal records are refreshed
 * in place -- no host round trip.  The host mirror is refreshed lazily (rc_get_instances / rc_export_instances / the next
 * mutation).  The address is valid until the next rebuilding rc_sync. */
int rc_instance_buffer_device(rc_scene* scene, uint32_t handle, rc_instance_desc** d_descs, uint32_t* count);
int rc_refit_device(rc_scene* scene, int recompute_inverse);

/* Page-lock (pin) a caller-owned host array so that the host-buffer entry points (rc_trace_closest / rc_trace_any, rc_add_blas,
 * rc_view_factors ...) move it by DMA at the full PCIe rate instead of through the driver's staging copies: the option a Julia
 * `Vector{RTRay}` / `Vector{RTHitResult}` pair that is traced every frame wants (there is no counterpart in the reference, whose
 * arrays live in the backend's memory).  Thin wrappers over hipHostRegister / hipHostUnregister so that a caller without HIP
 * bindings can use them; the memory stays the caller's.  Registering an array twice, or unregistering one that is not
 * registered, is an error (non-zero status). */
int rc_host_register(rc_scene* scene, void* ptr, uint64_t bytes);
int rc_host_unregister(rc_scene* scene, void* ptr);

/* Timing of the most recent trace / driver launch or BLAS build (device pipeline, without staging copies) on this scene, measured with HIP events on the launch
 * stream (kernel only, no copies), in milliseconds.  For a trace launch the two events ride on the trace kernel's own dispatch: the figure is that
 * kernel's duration and does NOT include the pair of small claim-order rebuild kernels (5 + 8 us) the library enqueues in front of about one
 * launch in eight of a REPEATED batch (first launches enqueue nothing else).  Time a run of launches between two events of your own on the stream
 * to include them -- bench.py does so for every repeated-batch figure. */
int rc_last_kernel_ms(rc_scene* scene, float* ms);
/* The same for the scene's most recent eager launches on any stream, oldest first: up to max_launches (at most 47: a launch's events live until
 * the scene's 48th launch after it) durations into ms[], their number into *n.  Waits for the launches asked about.  For a caller that
 * enqueues a run of launches back to back and wants every one's duration afterwards without putting events of its own between them (the
 * reference times its kernels the same way, one by one: docs/src/hw_acceleration.md:198-218).  Captured launches have no events: 0. */
int rc_recent_kernel_ms(rc_scene* scene, uint32_t max_launches, float* ms, uint32_t* n);

#ifdef __cplusplus
}
#endif
#endif
