/*
 * rc_oracle.h -- CPU restatement of Raycore.jl's TLAS/BLAS hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU baseline.  The product (raycore.jl_amd/csrc) never links or calls it.
 *
 * PARITY PINNING: the reference is pure Julia and cannot be run here or on the GPU box (no julia
 * binary, no network), and it ships no golden-vector files.  This restatement follows the reference
 * source statement by statement (file:line cited at each function) and is pinned by every
 * known-answer test the reference's own test-suite holds for this path (tests/test_oracle_kats.py
 * restates them).  Exact t bits, tie-breaks, NaN behaviour, Morton-sort tie order and all of
 * get_illumination / view_factors are NOT pinned by any reference test ("parity unpinned" for those;
 * see DESIGN.md).
 *
 * All citations are relative to /root/reference/.
 */
#ifndef RC_ORACLE_H
#define RC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCO_INVALID_NODE 0xFFFFFFFFu       /* src/instanced-bvh.jl:65   */
#define RCO_TOP_LEVEL_SENTINEL 0xFFFFFFFEu /* src/instanced-bvh.jl:1733 */

/* BVHNode2, 60 bytes (src/instanced-bvh.jl:50-63) */
typedef struct {
    float aabb0_min[3], aabb0_max[3], aabb1_min[3], aabb1_max[3];
    uint32_t child0, child1, parent;
} rco_node;

/* InstanceDescriptor, 108 bytes (src/instanced-bvh.jl:90-96). Mat3x4f = 12 floats, Vulkan row-major
 * rows [r0 r1 r2 t] (src/instanced-bvh.jl:28-31). */
typedef struct {
    uint32_t blas_index; /* 1-based */
    uint32_t instance_id;
    float transform[12];
    float inv_transform[12];
    uint32_t flags;
} rco_instance;

/* BLASDescriptor, 32 bytes (src/instanced-bvh.jl:132-136) */
typedef struct {
    uint32_t nodes_offset, primitives_offset; /* 0-based */
    float root_min[3], root_max[3];
} rco_blas_desc;

/* The part of Triangle{UInt32} (src/triangle_mesh.jl:1-7) the path reads: vertices + metadata. */
typedef struct {
    float v[3][3];
    uint32_t meta;
} rco_tri;

/* Ray record = RTRay layout (src/rt_transport.jl:10-19); fields = Ray{o,d,t_min,t_max} (src/ray.jl:1-7) */
typedef struct {
    float ox, oy, oz, tmin, dx, dy, dz, tmax;
} rco_ray;

/* Hit record = RTHitResult layout (src/rt_transport.jl:33-42).
 * primitive_id: 0-based index into the flat all_blas_prims array; instance_id: 0-based position in
 * the instance array; instance_custom_index: InstanceDescriptor.instance_id.  Miss: hit=0, t=0,
 * u=v=0, primitive_id=instance_id=0xFFFFFFFF, instance_custom_index=0. */
typedef struct {
    uint32_t hit;
    float t;
    uint32_t primitive_id, instance_custom_index;
    float bary_u, bary_v;
    uint32_t instance_id, _pad;
} rco_hit;

typedef struct rco_scene rco_scene;

/* ---- scene assembly: build_blas per geometry, then build_tlas (src/instanced-bvh.jl:1376,1605) -- */
rco_scene* rco_scene_new(void);
void rco_scene_free(rco_scene*);
/* verts: n x 9 floats (v0 v1 v2); meta: n u32 or NULL (=> face index 1..n BEFORE filtering,
 * src/instanced-bvh.jl:595).  Degenerate faces dropped when filter_degenerate != 0 (:573-577,:599).
 * Returns the 1-based BLAS index, 0 on error (no valid triangle). */
uint32_t rco_scene_add_blas(rco_scene*, const float* verts, const uint32_t* meta, uint32_t n, int filter_degenerate);
/* xform: 12 floats (Mat3x4f); inv: 12 floats or NULL (=> mat3x4_inverse, :1675-1687). */
int rco_scene_add_instance(rco_scene*, uint32_t blas_index, uint32_t instance_id, const float* xform, const float* inv);
int rco_scene_build(rco_scene*); /* build_tlas (:1605-1651); 0 on success */

/* accessors (copy-out into caller buffers; pass NULL to only query the count) */
uint32_t rco_scene_tlas_nodes(const rco_scene*, rco_node* out);
uint32_t rco_scene_instances(const rco_scene*, rco_instance* out);
uint32_t rco_scene_blas_nodes(const rco_scene*, rco_node* out);
uint32_t rco_scene_blas_prims(const rco_scene*, rco_tri* out);
uint32_t rco_scene_blas_descs(const rco_scene*, rco_blas_desc* out);
void rco_scene_world_bound(const rco_scene*, float out[6]);
/* Morton codes (sorted) of BLAS `blas_index` (1-based), for build parity checks. */
uint32_t rco_scene_blas_morton(const rco_scene*, uint32_t blas_index, uint32_t* out);

/* ---- traversal ---------------------------------------------------------------------------------- */
/* closest_hit (:1902-2024) / any_hit (:2034-2140) for one ray.  counters (optional, 2 x u32):
 * [0] += BVHNode2 fetches, [1] += TLAS-leaf entries (instance + descriptor fetches) -- the algorithmic
 * byte model of SURVEY.md section 8(d). */
void rco_closest_hit(const rco_scene*, const rco_ray*, rco_hit*, uint32_t* counters);
void rco_any_hit(const rco_scene*, const rco_ray*, rco_hit*, uint32_t* counters);
/* Batch over n rays with nthreads pthreads (mirrors Threads.@threads, src/kernels.jl:64).
 * mode 0 = closest, 1 = any.  counters: NULL or n x 2 u32 (zeroed by the callee). */
/* dev experiment: closest_hit with leaf-test results arriving `lag` loop iterations late (lag 0 = the reference algorithm) */
void rco_trace_deferred_batch(const rco_scene*, const rco_ray* rays, rco_hit* hits, uint64_t n, uint32_t lag, int nthreads);
/* worker k of the thread pool pinned to the k-th allowed CPU (before the pool's first use); CPUs this process may run on */
void rco_pool_pin(int enable);
int rco_allowed_cpus(void);
void rco_trace_batch(const rco_scene*, const rco_ray* rays, rco_hit* hits, uint64_t n, int mode, int nthreads,
                     uint32_t* counters);
/* Independent check of the traversal: Moeller-Trumbore over every (instance, triangle) pair with the
 * reference's arithmetic, no BVH.  Ties keep the first minimum found. */
void rco_brute_closest(const rco_scene*, const rco_ray*, rco_hit*);

/* ---- small pieces exposed for the reference's unit KATs ------------------------------------------ */
void rco_corner(const float mn[3], const float mx[3], int c, float out[3]);  /* corner(b, c), c = 1..8, src/bounds.jl:53-59 */
uint32_t rco_expand_bits(uint32_t x);                       /* :1177-1183 */
uint32_t rco_morton_code_30bit(const float p[3]);           /* :1189-1200 */
int32_t rco_clz32(uint32_t x);                              /* :1203-1206 */
int32_t rco_delta(int32_t i1, int32_t i2, const uint32_t* codes, int32_t n); /* :1212-1229 */
void rco_mat3x4_inverse(const float m[12], float out[12]);  /* :1675-1687 */
void rco_mat4_to_mat3x4(const float m4_colmajor[16], float out[12]); /* :1663-1669 */
void rco_transform_point(const float m[12], const float p[3], float out[3]);     /* :1692-1698 */
void rco_transform_direction(const float m[12], const float v[3], float out[3]); /* :1711-1717 */
void rco_safe_invdir(const float d[3], float out[3]);       /* :1742-1748 */
int rco_is_degenerate(const float v[9]);                    /* src/triangle_mesh.jl:14-17 */
void rco_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void rco_set_histograms(uint32_t* tlas_hist, uint32_t* blas_hist); /* dev: per-node visit counts (single-threaded use) */
/* dev: trace one ray and record one byte per loop iteration (kind | flags, see rc_oracle.c) plus the stack depth after the step;
 * returns the number of steps (only the first `cap` are stored).  Feeds tools/sched_sim.py. */
uint32_t rco_trace_events(const rco_scene*, const rco_ray*, int any, uint8_t* events, uint8_t* depths, uint32_t cap);
uint32_t rco_trace_steps(const rco_scene*, const rco_ray*, int any, uint8_t* events, uint8_t* depths, uint32_t* nodes, float* closest, uint32_t cap);
/* test: the instance entries of one ray -- instance index, closest_t at entry, triangle tests before leaving (tests/test_entry_cull_predicate.py) */
uint32_t rco_trace_entries(const rco_scene*, const rco_ray*, int any, uint32_t* inst, float* closest_at_entry, uint32_t* leaf_tests, uint32_t cap);
int32_t rco_max_stack(int reset); /* dev: deepest traversal stack seen since the last reset (single-threaded use) */
void rco_sincos_f64(double x, double* s, double* c); /* sampler trig, see rc_oracle.c */
double rco_acos_f64(double x);

/* ---- drivers (src/kernels.jl) -------------------------------------------------------------------- */
/* generate_ray_grid (src/kernels.jl:10-56) as called from hits_from_grid (:58-72): writes grid*grid
 * rays (column-major: index (i-1) + grid*(j-1)), direction = normalize(viewdir), t_min 0, t_max Inf. */
void rco_generate_ray_grid(const rco_scene*, const float viewdir[3], uint32_t grid, rco_ray* out);
/* get_illumination (:112-124): out has n_prims floats. */
void rco_get_illumination(const rco_scene*, const float viewdir[3], uint32_t grid, float* out, int nthreads);
/* view_factors (:74-104) with the unseeded task-local RNG replaced by Philox4x32-10 keyed
 * (seed) with counter (ray_idx, src_prim_idx0, 0, 0) -> (r1, r2, xi1, xi2).  out: N x N u32,
 * column-major Julia Matrix: out[(src_meta-1) + N*(hit_meta-1)].  Only source primitives
 * [src_begin, src_end) and rays [ray_begin, ray_end) of each are shot (for shard tests); pass
 * 0,N,0,rays_per_triangle for the whole job.  Accumulates into out (caller zeroes). */
void rco_view_factors(const rco_scene*, uint32_t rays_per_triangle, uint64_t seed, uint32_t src_begin,
                      uint32_t src_end, uint32_t ray_begin, uint32_t ray_end, uint32_t* out, int nthreads);
/* One source primitive's row as a compact N-vector (row[hit_meta-1] += 1 per counted ray): lets a test check rows of a
 * 50 k x 50 k matrix without allocating it.  src = 0-based flat (Morton-sorted) primitive index. */
void rco_view_factor_row(const rco_scene*, uint32_t rays_per_triangle, uint64_t seed, uint32_t src, uint32_t ray_begin,
                         uint32_t ray_end, uint32_t* row);
/* The ray view_factors shoots for (src prim idx0, ray_idx): exposed so ray generation can be checked. */
int rco_view_factor_ray(const rco_scene*, uint32_t src_idx0, uint32_t ray_idx, uint64_t seed, rco_ray* out);

/* wavefront stages next to the trace (docs/src/wavefront-renderer.jl:296-333); points/normals: n x 3 floats */
void rco_hit_points(const rco_scene*, const rco_ray* rays, const rco_hit* hits, uint64_t n, float* points, float* normals);
void rco_shadow_rays(const rco_scene*, const rco_ray* rays, const rco_hit* hits, uint64_t n, const float light[3], float bias, rco_ray* out);

/* ---- BVH4 (src/bvh4.jl): BLAS-level 4-wide tree collapsed from the BVH2; no TLAS/instance support in the
 * reference.  PARITY UNPINNED: the reference has no test for this file. ------- */
/* BVHNode4, 120 bytes (src/bvh4.jl:40-69) */
typedef struct {
    uint32_t child[4];     /* child0..child3; interior: 1-based BVH4 node index; leaf: child0 = 1-based sorted prim index */
    float aabb[4][2][3];   /* aabbK_min, aabbK_max for K = 0..3 */
    uint32_t parent;
    uint8_t child_count;   /* 0 = leaf */
    uint8_t primitive_count;
    uint8_t _pad1, _pad2;
} rco_node4;
/* build_blas4 (:511-522) = build_blas + collapse_bvh2_to_bvh4 (:314-497) for BLAS `blas_index` (1-based).
 * Returns the node count; copies the nodes when out != NULL. */
uint32_t rco_blas4_nodes(rco_scene*, uint32_t blas_index, rco_node4* out);
/* closest_hit4 (:606-689) / any_hit4 (:696-766) on that BLAS4, in the BLAS's own space (no instance transform).
 * Both force t_min = 0 (:610,:700).  Hit record: primitive_id = 0-based index into the BLAS's Morton-sorted
 * primitives, instance_id = 0xFFFFFFFF, instance_custom_index = 0.  counters (NULL or n x 2 u32):
 * [0] BVHNode4 fetches, [1] triangle fetches. */
void rco_trace4_batch(rco_scene*, uint32_t blas_index, const rco_ray* rays, rco_hit* hits, uint64_t n, int mode,
                      int nthreads, uint32_t* counters);

/* ---- collision broad phase (src/collision.jl).  PARITY UNPINNED: no reference test. ------------------ */
typedef struct { uint32_t instance_a, instance_b; } rco_contact; /* ContactPair (:25-28), 1-based instance indices */
/* collide_instances (:189-233): pass 1 counts per sorted TLAS leaf, inclusive prefix sum, pass 2 writes each leaf's
 * pairs back-to-front inside its range (:135).  Returns the number of contacts; writes them when out != NULL and the
 * inclusive prefix sums (the `cache` buffer, n u32) when counts != NULL. */
uint64_t rco_collide_instances(const rco_scene*, rco_contact* out, uint32_t* counts);
/* collide_instances_any (:241-261) for two 0-based instance ranges.  As written in the reference, instance index i is
 * looked up at TLAS leaf position n-1+i, i.e. in Morton-sorted order, not at the leaf that holds instance i. */
int rco_collide_instances_any(const rco_scene*, uint32_t a_first, uint32_t a_count, uint32_t b_first, uint32_t b_count);

/* ---- mesh ingestion and the full Triangle record (src/instanced-bvh.jl:555-608, src/triangle_mesh.jl:1-7) ---------------- */
/* Triangle{UInt32}, 136 bytes: what closest_hit returns by value. */
typedef struct {
    float vertices[3][3], normals[3][3], tangents[3][3], uv[3][2];
    uint32_t metadata;
} rco_triangle;
/* build_and_append_blas! after the GeometryBasics decomposition (:581-600): per-vertex positions / normals / optional uvs,
 * 0-based triangle indices (3 per face), optional per-vertex face_meta (the reference reads face_meta[first vertex of the
 * face], :595; NULL => face index 1..nf assigned before the degenerate filter).  build_triangle (:555-566): tangents NaN,
 * default uv (0,0),(1,0),(1,1).  Returns the 1-based BLAS index, 0 when no valid triangle remains. */
uint32_t rco_scene_add_mesh(rco_scene*, const float* verts, const float* normals, const float* uvs, uint32_t nv,
                            const uint32_t* indices, uint32_t nf, const uint32_t* face_meta);
/* all_blas_prims as full Triangles (flat, Morton-sorted per BLAS).  Geometry added as plain soup (rco_scene_add_blas) has no
 * mesh attributes: its normals are the geometric normal normalize((v1-v0) x (v2-v0)) on all three vertices, uv the default. */
uint32_t rco_scene_triangles(const rco_scene*, rco_triangle* out);
/* Shading epilogue of the reference's renderers (docs/src/wavefront-renderer.jl:382-387): per hit, the interpolated normal
 * normalize(n0*b1 + n1*b2 + n2*b3) and uv0*b1 + uv1*b2 + uv2*b3 with (b1,b2,b3) = ((1-u)-v, u, v); zeros on a miss.
 * normals: n x 3 floats, uvs: n x 2 floats (either may be NULL). */
void rco_shading_attributes(const rco_scene*, const rco_hit* hits, uint64_t n, float* normals, float* uvs);

/* generate_primary_rays_lookat! (docs/src/wavefront-renderer.jl:219-254) with rand(Vec2f) replaced by Philox4x32-10 keyed by
 * `seed`, counter (ray_lo, ray_hi, 0, 0x50524159) -> (j1, j2); jitter == 0 => pixel centres.  out: width*height*samples rays. */
void rco_primary_rays_lookat(const float pos[3], const float right[3], const float up[3], const float forward[3], float half_width,
                             float half_height, uint32_t width, uint32_t height, uint32_t samples, uint64_t seed, int jitter, rco_ray* out);

/* generate_reflection_rays! for perfect mirrors (docs/src/wavefront-renderer.jl:431-476 with roughness 0) on reflect (src/math.jl:80):
 * per hit, normal = the interpolated shading normal of rco_shading_attributes, origin = (o + d*t) + normal*bias, direction =
 * reflect(-d, normal) = -wo + (2*(wo.n))*n, t_min 0, t_max Inf; misses get the dummy ray (o = 0, d = (0,0,1), t_max = 0). */
void rco_reflection_rays(const rco_scene*, const rco_ray* rays, const rco_hit* hits, uint64_t n, float bias, rco_ray* out);

#ifdef __cplusplus
}
#endif
#endif
