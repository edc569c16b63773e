/*
 * rc_oracle.c -- CPU restatement of Raycore.jl's TLAS/BLAS hot path (see rc_oracle.h header comment:
 * TEST INFRASTRUCTURE ONLY; parity pinned by the reference's own KATs, no reference run possible).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).  Julia never
 * contracts a*b+c and evaluates + left to right; every expression below keeps the reference's shape.
 * All citations are relative to /root/reference/.
 */
#include "rc_oracle.h"

#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * Julia Base float semantics
 * ---------------------------------------------------------------------------------------------- */
/* Base.min / Base.max on Float32: NaN-propagating, -0 < +0. */
static inline float jl_min(float a, float b) {
    if (a != a) return a;
    if (b != b) return b;
    if (a < b) return a;
    if (b < a) return b;
    return signbit(a) ? a : b;
}
static inline float jl_max(float a, float b) {
    if (a != a) return a;
    if (b != b) return b;
    if (a > b) return a;
    if (b > a) return b;
    return signbit(a) ? b : a;
}
/* Base.clamp(x, lo, hi) = ifelse(x > hi, hi, ifelse(x < lo, lo, x)); NaN passes through. */
static inline float jl_clamp(float x, float lo, float hi) { return x > hi ? hi : (x < lo ? lo : x); }
/* unsafe_trunc(UInt32, x): undefined for NaN in Julia; defined here as 0 (x86-64 cvttss2si low word,
 * and what v_cvt_u32_f32 returns on gfx950). */
static inline uint32_t jl_unsafe_trunc_u32(float x) { return (x != x) ? 0u : (uint32_t)x; }

typedef struct { float x, y, z; } v3;
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_from(const float* p) { return V(p[0], p[1], p[2]); }
static inline v3 v3_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
/* GeometryBasics 0.5 fixed_arrays: dot(a,b) = sum(a .* b) -> (a1b1 + a2b2) + a3b3 (assumed) */
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* norm(a) = sqrt(dot(a,a)); normalize(a) = a ./ norm(a) (GeometryBasics 0.5, assumed) */
static inline v3 v3_normalize(v3 a) {
    float n = sqrtf(v3_dot(a, a));
    return V(a.x / n, a.y / n, a.z / n);
}
static inline v3 v3_min(v3 a, v3 b) { return V(jl_min(a.x, b.x), jl_min(a.y, b.y), jl_min(a.z, b.z)); }
static inline v3 v3_max(v3 a, v3 b) { return V(jl_max(a.x, b.x), jl_max(a.y, b.y), jl_max(a.z, b.z)); }
static inline void v3_store(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

/* ------------------------------------------------------------------------------------------------
 * Transforms (src/instanced-bvh.jl:1663-1726).  Mat3x4f memory = Vulkan rows: m[4*r + c].
 * Julia index m[j+1, i+1] = memory[j + 4*i] = Vulkan (row i, col j).
 * ---------------------------------------------------------------------------------------------- */
void rco_mat4_to_mat3x4(const float m[16], float out[12]) { /* :1663-1669; m column-major 4x4 */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) out[4 * r + c] = m[r + 4 * c]; /* m[r+1, c+1] */
}

void rco_transform_point(const float m[12], const float p[3], float out[3]) { /* :1692-1698 */
    out[0] = m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3];
    out[1] = m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7];
    out[2] = m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11];
}
void rco_transform_direction(const float m[12], const float v[3], float out[3]) { /* :1711-1717 */
    out[0] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
    out[1] = m[4] * v[0] + m[5] * v[1] + m[6] * v[2];
    out[2] = m[8] * v[0] + m[9] * v[1] + m[10] * v[2];
}
static inline v3 xf_point(const float m[12], v3 p) {
    float in[3] = {p.x, p.y, p.z}, o[3];
    rco_transform_point(m, in, o);
    return v3_from(o);
}
static inline v3 xf_dir(const float m[12], v3 p) {
    float in[3] = {p.x, p.y, p.z}, o[3];
    rco_transform_direction(m, in, o);
    return v3_from(o);
}

/* mat3x4_inverse (:1675-1687).  R = m[1:3,1:3] (Julia indices) => R[i,j] = memory[(i-1) + 4(j-1)].
 * B = inv(R) with StaticArrays 1.9's 3x3 inverse (src/inv.jl, not vendored -- restated from its
 * published algorithm: x0,x1,x2 = columns; y0 = x1 x x2; d = x0 . y0; x0 /= d; y0 /= d;
 * y1 = x2 x x0; y2 = x0 x x1; inv = rows (y0,y1,y2)).  Assumed, unverifiable here. */
void rco_mat3x4_inverse(const float m[12], float out[12]) {
#define R_(i, j) m[((i)-1) + 4 * ((j)-1)]
    v3 x0 = V(R_(1, 1), R_(2, 1), R_(3, 1));
    v3 x1 = V(R_(1, 2), R_(2, 2), R_(3, 2));
    v3 x2 = V(R_(1, 3), R_(2, 3), R_(3, 3));
#undef R_
    v3 y0 = v3_cross(x1, x2);
    float d = v3_dot(x0, y0);
    x0 = V(x0.x / d, x0.y / d, x0.z / d);
    y0 = V(y0.x / d, y0.y / d, y0.z / d);
    v3 y1 = v3_cross(x2, x0);
    v3 y2 = v3_cross(x0, x1);
    /* B column-major tuple (y0[1],y1[1],y2[1], y0[2],y1[2],y2[2], y0[3],y1[3],y2[3]):
     * B[1,1]=y0.x B[2,1]=y1.x B[3,1]=y2.x B[1,2]=y0.y B[2,2]=y1.y B[3,2]=y2.y B[1,3]=y0.z ... */
    float B11 = y0.x, B21 = y1.x, B31 = y2.x;
    float B12 = y0.y, B22 = y1.y, B32 = y2.y;
    float B13 = y0.z, B23 = y1.z, B33 = y2.z;
    float tx = m[3], ty = m[7], tz = m[11]; /* m[4,1], m[4,2], m[4,3] */
    float tix = -(B11 * tx + B21 * ty + B31 * tz);
    float tiy = -(B12 * tx + B22 * ty + B32 * tz);
    float tiz = -(B13 * tx + B23 * ty + B33 * tz);
    out[0] = B11; out[1] = B21; out[2] = B31; out[3] = tix;
    out[4] = B12; out[5] = B22; out[6] = B32; out[7] = tiy;
    out[8] = B13; out[9] = B23; out[10] = B33; out[11] = tiz;
}

/* ------------------------------------------------------------------------------------------------
 * Morton / Karras LBVH pieces (src/instanced-bvh.jl:1177-1290)
 * ---------------------------------------------------------------------------------------------- */
uint32_t rco_expand_bits(uint32_t x) { /* :1177-1183 */
    x = (x * 0x00010001u) & 0xFF0000FFu;
    x = (x * 0x00000101u) & 0x0F00F00Fu;
    x = (x * 0x00000011u) & 0xC30C30C3u;
    x = (x * 0x00000005u) & 0x49249249u;
    return x;
}
uint32_t rco_morton_code_30bit(const float p[3]) { /* :1189-1200 */
    const float unit_side = 1024.0f;
    float x = jl_clamp(p[0] * unit_side, 0.0f, unit_side - 1.0f);
    float y = jl_clamp(p[1] * unit_side, 0.0f, unit_side - 1.0f);
    float z = jl_clamp(p[2] * unit_side, 0.0f, unit_side - 1.0f);
    return (rco_expand_bits(jl_unsafe_trunc_u32(x)) << 2) | (rco_expand_bits(jl_unsafe_trunc_u32(y)) << 1) |
           rco_expand_bits(jl_unsafe_trunc_u32(z));
}
int32_t rco_clz32(uint32_t x) { /* :1203-1206 */
    if (x == 0) return 32;
    return (int32_t)__builtin_clz(x);
}
int32_t rco_delta(int32_t i1, int32_t i2, const uint32_t* codes, int32_t n) { /* :1212-1229; 1-based */
    int32_t left = i1 < i2 ? i1 : i2;
    int32_t right = i1 < i2 ? i2 : i1;
    if (left < 1 || right > n) return -1;
    uint32_t lc = codes[left - 1], rc = codes[right - 1];
    if (lc != rc) return rco_clz32(lc ^ rc);
    return 32 + rco_clz32((uint32_t)left ^ (uint32_t)right);
}
static void find_span_for_node(int32_t idx, const uint32_t* codes, int32_t n, int32_t* lo, int32_t* hi) { /* :1232-1262 */
    int32_t d_left = rco_delta(idx, idx - 1, codes, n);
    int32_t d_right = rco_delta(idx, idx + 1, codes, n);
    int32_t d = d_right > d_left ? 1 : -1;
    int32_t delta_min = rco_delta(idx, idx - d, codes, n);
    int32_t l_max = 2;
    while (rco_delta(idx, idx + l_max * d, codes, n) > delta_min) l_max *= 2;
    int32_t l = 0, t = l_max;
    while (t > 1) {
        t = t / 2;
        if (rco_delta(idx, idx + (l + t) * d, codes, n) > delta_min) l = l + t;
    }
    int32_t j = idx + l * d;
    if (d > 0) { *lo = idx; *hi = j; } else { *lo = j; *hi = idx; }
}
static int32_t find_split_in_span(int32_t span_left, int32_t span_right, const uint32_t* codes, int32_t n) { /* :1265-1290 */
    int32_t numidentical = rco_delta(span_left, span_right, codes, n);
    int32_t left = span_left, right = span_right;
    while (right > left + 1) {
        int32_t newsplit = (right + left) / 2;
        if (rco_delta(left, newsplit, codes, n) > numidentical) left = newsplit; else right = newsplit;
    }
    return left;
}

static const rco_node EMPTY_NODE = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, RCO_INVALID_NODE, RCO_INVALID_NODE, RCO_INVALID_NODE};

/* emit_topology_kernel! + set_parent_pointers_kernel! (src/instanced-bvh-kernels.jl:119-191).
 * nodes is 0-based storage of the 1-based node array (node k lives at nodes[k-1]). */
static void emit_topology_and_parents(rco_node* nodes, const uint32_t* codes, int32_t n) {
    for (int32_t idx = 1; idx < n; ++idx) {
        int32_t lo, hi;
        find_span_for_node(idx, codes, n, &lo, &hi);
        int32_t split = find_split_in_span(lo, hi, codes, n);
        int32_t child0 = (split == lo) ? (n - 1 + split) : split;
        int32_t c1 = split + 1;
        int32_t child1 = (c1 == hi) ? (n - 1 + c1) : c1;
        rco_node nd = EMPTY_NODE;
        nd.child0 = (uint32_t)child0;
        nd.child1 = (uint32_t)child1;
        nodes[idx - 1] = nd;
    }
    for (int32_t idx = 1; idx < n; ++idx) {
        nodes[nodes[idx - 1].child0 - 1].parent = (uint32_t)idx;
        nodes[nodes[idx - 1].child1 - 1].parent = (uint32_t)idx;
    }
}

/* get_node_aabb (:1141-1160) / get_tlas_node_aabb (:1163-1174) */
static void node_aabb(const rco_node* nd, int is_interior, int tlas, v3* mn, v3* mx) {
    if (is_interior) {
        *mn = v3_min(v3_from(nd->aabb0_min), v3_from(nd->aabb1_min));
        *mx = v3_max(v3_from(nd->aabb0_max), v3_from(nd->aabb1_max));
    } else if (tlas) {
        *mn = v3_from(nd->aabb0_min);
        *mx = v3_from(nd->aabb0_max);
    } else {
        v3 v0 = v3_from(nd->aabb0_min), v1 = v3_from(nd->aabb0_max), v2 = v3_from(nd->aabb1_min);
        *mn = v3_min(v3_min(v0, v1), v2);
        *mx = v3_max(v3_max(v0, v1), v2);
    }
}

/* refit_aabbs_kernel! / refit_tlas_aabbs_kernel! (src/instanced-bvh-kernels.jl:239-286, 381-428),
 * executed sequentially: the second arrival at a node computes it, exactly as with the atomic flag. */
static void refit(rco_node* nodes, int32_t n, int tlas) {
    if (n < 2) return;
    uint32_t* flags = (uint32_t*)calloc((size_t)(n - 1), sizeof(uint32_t));
    for (int32_t prim = 1; prim <= n; ++prim) {
        uint32_t parent = nodes[(n - 1 + prim) - 1].parent;
        while (parent != RCO_INVALID_NODE) {
            uint32_t nv = ++flags[parent - 1];
            if (nv != 2) break;
            rco_node* nd = &nodes[parent - 1];
            uint32_t c0 = nd->child0, c1 = nd->child1;
            v3 mn0, mx0, mn1, mx1;
            node_aabb(&nodes[c0 - 1], c0 < (uint32_t)n, tlas, &mn0, &mx0);
            node_aabb(&nodes[c1 - 1], c1 < (uint32_t)n, tlas, &mn1, &mx1);
            v3_store(nd->aabb0_min, mn0); v3_store(nd->aabb0_max, mx0);
            v3_store(nd->aabb1_min, mn1); v3_store(nd->aabb1_max, mx1);
            parent = nd->parent;
        }
    }
    free(flags);
}

/* stable sortperm by code (Base.sortperm default is stable; AK.sortperm assumed stable) */
typedef struct { uint32_t code, idx; } code_idx;
static int cmp_code_idx(const void* a, const void* b) {
    const code_idx* x = (const code_idx*)a; const code_idx* y = (const code_idx*)b;
    if (x->code != y->code) return x->code < y->code ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* ------------------------------------------------------------------------------------------------
 * Scene containers
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    rco_node* nodes; uint32_t n_nodes;
    rco_tri* prims; uint32_t n_prims; /* Morton-sorted */
    uint32_t* morton;                 /* sorted codes */
    float root_min[3], root_max[3];
    rco_node4* nodes4; uint32_t n_nodes4; /* BLAS4 (src/bvh4.jl), built on demand */
    /* mesh attributes (rco_scene_add_mesh): per-vertex arrays + the source face of every sorted primitive */
    float* m_normals; float* m_uvs; uint32_t* m_indices; uint32_t* src_face; int has_attrs;
} blas_t;

struct rco_scene {
    blas_t* blas; uint32_t n_blas, cap_blas;
    rco_instance* inst; uint32_t n_inst, cap_inst;
    /* StaticTLAS (src/instanced-bvh.jl:155-168), valid after rco_scene_build */
    rco_node* nodes; uint32_t n_nodes;
    rco_node* blas_nodes; uint32_t n_blas_nodes;
    rco_tri* blas_prims; uint32_t n_blas_prims;
    rco_blas_desc* descs;
    float root_min[3], root_max[3];
    int built;
};

rco_scene* rco_scene_new(void) { return (rco_scene*)calloc(1, sizeof(rco_scene)); }
static void free_static(rco_scene* s) {
    free(s->nodes); free(s->blas_nodes); free(s->blas_prims); free(s->descs);
    s->nodes = s->blas_nodes = NULL; s->blas_prims = NULL; s->descs = NULL;
    s->n_nodes = s->n_blas_nodes = s->n_blas_prims = 0; s->built = 0;
}
void rco_scene_free(rco_scene* s) {
    if (!s) return;
    for (uint32_t i = 0; i < s->n_blas; ++i) { free(s->blas[i].nodes); free(s->blas[i].prims); free(s->blas[i].morton); free(s->blas[i].nodes4); free(s->blas[i].m_normals); free(s->blas[i].m_uvs); free(s->blas[i].m_indices); free(s->blas[i].src_face); }
    free(s->blas); free(s->inst); free_static(s); free(s);
}

/* is_degenerate (src/triangle_mesh.jl:14-17): v = (v3-v1) x (v2-v1); (v.v) ~ 0f0.  isapprox against
 * an exact zero with the default rtol and atol=0 is true only for v.v == 0. */
int rco_is_degenerate(const float p[9]) {
    v3 a = v3_from(p), b = v3_from(p + 3), c = v3_from(p + 6);
    v3 v = v3_cross(v3_sub(c, a), v3_sub(b, a));
    return v3_dot(v, v) == 0.0f;
}

/* build_blas (src/instanced-bvh.jl:1376-1443) */
static int build_blas(const rco_tri* in, uint32_t n, blas_t* out, const uint32_t* face_of_in) {
    if (n == 0) return -1;
    /* scene AABB: mapreduce(world_bound, U, prims, init=Bounds3()) (:1386) */
    v3 smin = V(INFINITY, INFINITY, INFINITY), smax = V(-INFINITY, -INFINITY, -INFINITY);
    for (uint32_t i = 0; i < n; ++i) {
        v3 v0 = v3_from(in[i].v[0]), v1 = v3_from(in[i].v[1]), v2 = v3_from(in[i].v[2]);
        v3 tmin = v3_min(v3_min(v0, v1), v2), tmax = v3_max(v3_max(v0, v1), v2); /* world_bound(tri), triangle_mesh.jl:37 */
        smin = v3_min(smin, tmin); smax = v3_max(smax, tmax);
    }
    v3 extent = v3_sub(smax, smin); /* :1388, NOT clamped for the BLAS */
    /* calculate_morton_code_for_prim (src/instanced-bvh-kernels.jl:88-98) */
    code_idx* ci = (code_idx*)malloc(sizeof(code_idx) * n);
    for (uint32_t i = 0; i < n; ++i) {
        v3 v0 = v3_from(in[i].v[0]), v1 = v3_from(in[i].v[1]), v2 = v3_from(in[i].v[2]);
        v3 tmin = v3_min(v3_min(v0, v1), v2), tmax = v3_max(v3_max(v0, v1), v2);
        v3 c = v3_scale(v3_add(tmin, tmax), 0.5f); /* 0.5f0 * (p_min + p_max) */
        v3 d = v3_sub(c, smin);
        float nrm[3] = {d.x / extent.x, d.y / extent.y, d.z / extent.z};
        ci[i].code = rco_morton_code_30bit(nrm);
        ci[i].idx = i;
    }
    qsort(ci, n, sizeof(code_idx), cmp_code_idx); /* :1399-1402 */
    out->n_prims = n;
    out->prims = (rco_tri*)malloc(sizeof(rco_tri) * n);
    out->morton = (uint32_t*)malloc(sizeof(uint32_t) * n);
    for (uint32_t i = 0; i < n; ++i) { out->prims[i] = in[ci[i].idx]; out->morton[i] = ci[i].code; }
    if (face_of_in) {
        out->src_face = (uint32_t*)malloc(sizeof(uint32_t) * n);
        for (uint32_t i = 0; i < n; ++i) out->src_face[i] = face_of_in[ci[i].idx];
    }
    free(ci);
    out->n_nodes = 2 * n - 1;
    out->nodes = (rco_node*)malloc(sizeof(rco_node) * out->n_nodes);
    for (uint32_t i = 0; i < out->n_nodes; ++i) out->nodes[i] = EMPTY_NODE; /* :1406-1412 */
    if (n > 1) emit_topology_and_parents(out->nodes, out->morton, (int32_t)n); /* :1415-1422 */
    /* create_leaf_for_prim (src/instanced-bvh-kernels.jl:198-215) */
    for (uint32_t j = 1; j <= n; ++j) {
        rco_node* nd = &out->nodes[(n - 1 + j) - 1];
        uint32_t parent = nd->parent;
        memcpy(nd->aabb0_min, out->prims[j - 1].v[0], 12);
        memcpy(nd->aabb0_max, out->prims[j - 1].v[1], 12);
        memcpy(nd->aabb1_min, out->prims[j - 1].v[2], 12);
        nd->aabb1_max[0] = nd->aabb1_max[1] = nd->aabb1_max[2] = 0.0f;
        nd->child0 = RCO_INVALID_NODE; nd->child1 = j; nd->parent = parent;
    }
    refit(out->nodes, (int32_t)n, 0); /* :1431-1433 */
    v3 rmin, rmax;
    node_aabb(&out->nodes[0], out->nodes[0].child0 != RCO_INVALID_NODE, 0, &rmin, &rmax); /* :1438-1440 */
    v3_store(out->root_min, rmin); v3_store(out->root_max, rmax);
    return 0;
}

uint32_t rco_scene_add_blas(rco_scene* s, const float* verts, const uint32_t* meta, uint32_t n, int filter) {
    rco_tri* tris = (rco_tri*)malloc(sizeof(rco_tri) * (n ? n : 1));
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i) { /* :593-600 */
        if (filter && rco_is_degenerate(verts + 9 * (size_t)i)) continue;
        memcpy(tris[m].v, verts + 9 * (size_t)i, 36);
        tris[m].meta = meta ? meta[i] : (i + 1);
        ++m;
    }
    if (m == 0) { free(tris); return 0; } /* error("Geometry has no valid triangles") :601 */
    if (s->n_blas == s->cap_blas) {
        s->cap_blas = s->cap_blas ? 2 * s->cap_blas : 8;
        s->blas = (blas_t*)realloc(s->blas, sizeof(blas_t) * s->cap_blas);
    }
    blas_t* b = &s->blas[s->n_blas];
    memset(b, 0, sizeof(*b));
    build_blas(tris, m, b, NULL);
    free(tris);
    free_static(s);
    return ++s->n_blas;
}

uint32_t rco_scene_add_mesh(rco_scene* s, const float* verts, const float* normals, const float* uvs, uint32_t nv,
                            const uint32_t* indices, uint32_t nf, const uint32_t* face_meta) {
    rco_tri* tris = (rco_tri*)malloc(sizeof(rco_tri) * (nf ? nf : 1));
    uint32_t* face_of = (uint32_t*)malloc(sizeof(uint32_t) * (nf ? nf : 1));
    uint32_t m = 0;
    for (uint32_t i = 0; i < nf; ++i) { /* :591-600 */
        float p[9];
        for (int k = 0; k < 3; ++k) memcpy(p + 3 * k, verts + 3 * (size_t)indices[3 * (size_t)i + k], 12);
        if (rco_is_degenerate(p)) continue; /* is_degenerate_face :573-577 */
        memcpy(tris[m].v, p, 36);
        tris[m].meta = face_meta ? face_meta[indices[3 * (size_t)i]] : (i + 1); /* :595 */
        face_of[m] = i;
        ++m;
    }
    if (m == 0) { free(tris); free(face_of); return 0; }
    if (s->n_blas == s->cap_blas) {
        s->cap_blas = s->cap_blas ? 2 * s->cap_blas : 8;
        s->blas = (blas_t*)realloc(s->blas, sizeof(blas_t) * s->cap_blas);
    }
    blas_t* b = &s->blas[s->n_blas];
    memset(b, 0, sizeof(*b));
    build_blas(tris, m, b, face_of);
    free(tris); free(face_of);
    b->has_attrs = 1;
    b->m_normals = (float*)malloc(sizeof(float) * 3 * (size_t)nv); memcpy(b->m_normals, normals, sizeof(float) * 3 * (size_t)nv);
    if (uvs) { b->m_uvs = (float*)malloc(sizeof(float) * 2 * (size_t)nv); memcpy(b->m_uvs, uvs, sizeof(float) * 2 * (size_t)nv); }
    b->m_indices = (uint32_t*)malloc(sizeof(uint32_t) * 3 * (size_t)nf); memcpy(b->m_indices, indices, sizeof(uint32_t) * 3 * (size_t)nf);
    free_static(s);
    return ++s->n_blas;
}

int rco_scene_add_instance(rco_scene* s, uint32_t blas_index, uint32_t instance_id, const float* xform, const float* inv) {
    if (blas_index < 1 || blas_index > s->n_blas) return -1;
    if (s->n_inst == s->cap_inst) {
        s->cap_inst = s->cap_inst ? 2 * s->cap_inst : 16;
        s->inst = (rco_instance*)realloc(s->inst, sizeof(rco_instance) * s->cap_inst);
    }
    rco_instance* d = &s->inst[s->n_inst++];
    d->blas_index = blas_index; d->instance_id = instance_id; d->flags = 0;
    memcpy(d->transform, xform, 48);
    if (inv) memcpy(d->inv_transform, inv, 48); else rco_mat3x4_inverse(xform, d->inv_transform); /* :644 */
    free_static(s);
    return 0;
}

/* corner(b, c), c = 1..8 (src/bounds.jl:53-59) */
static inline v3 corner(const float mn[3], const float mx[3], int c) {
    c -= 1;
    return V((c & 1) == 0 ? mn[0] : mx[0], (c & 2) == 0 ? mn[1] : mx[1], (c & 4) == 0 ? mn[2] : mx[2]);
}

void rco_corner(const float mn[3], const float mx[3], int c, float out[3]) { /* test hook: the reference pins corner() in test/bounds.jl:92-103 */
    v3 p = corner(mn, mx, c);
    out[0] = p.x; out[1] = p.y; out[2] = p.z;
}

/* build_tlas_topology (src/instanced-bvh.jl:1485-1594) + flat arrays of build_tlas (:1605-1651) */
int rco_scene_build(rco_scene* s) {
    free_static(s);
    uint32_t n = s->n_inst;
    /* flat BLAS arrays + descriptors (:1628-1648) */
    s->descs = (rco_blas_desc*)malloc(sizeof(rco_blas_desc) * (s->n_blas ? s->n_blas : 1));
    uint32_t tn = 0, tp = 0;
    for (uint32_t i = 0; i < s->n_blas; ++i) {
        s->descs[i].nodes_offset = tn; s->descs[i].primitives_offset = tp;
        memcpy(s->descs[i].root_min, s->blas[i].root_min, 12); memcpy(s->descs[i].root_max, s->blas[i].root_max, 12);
        tn += s->blas[i].n_nodes; tp += s->blas[i].n_prims;
    }
    s->n_blas_nodes = tn; s->n_blas_prims = tp;
    s->blas_nodes = (rco_node*)malloc(sizeof(rco_node) * (tn ? tn : 1));
    s->blas_prims = (rco_tri*)malloc(sizeof(rco_tri) * (tp ? tp : 1));
    for (uint32_t i = 0; i < s->n_blas; ++i) {
        memcpy(s->blas_nodes + s->descs[i].nodes_offset, s->blas[i].nodes, sizeof(rco_node) * s->blas[i].n_nodes);
        memcpy(s->blas_prims + s->descs[i].primitives_offset, s->blas[i].prims, sizeof(rco_tri) * s->blas[i].n_prims);
    }
    if (n == 0) { /* :1612-1620 */
        s->n_nodes = 0; s->nodes = NULL;
        s->root_min[0] = s->root_min[1] = s->root_min[2] = INFINITY;
        s->root_max[0] = s->root_max[1] = s->root_max[2] = -INFINITY;
        s->built = 1;
        return 0;
    }
    /* compute_instance_world_aabb (src/instanced-bvh-kernels.jl:38-62) */
    v3* amin = (v3*)malloc(sizeof(v3) * n); v3* amax = (v3*)malloc(sizeof(v3) * n);
    for (uint32_t i = 0; i < n; ++i) {
        const rco_instance* in = &s->inst[i];
        const blas_t* b = &s->blas[in->blas_index - 1];
        v3 c1 = xf_point(in->transform, corner(b->root_min, b->root_max, 1));
        v3 mn = c1, mx = c1;
        for (int c = 2; c <= 8; ++c) {
            v3 wc = xf_point(in->transform, corner(b->root_min, b->root_max, c));
            mn = v3_min(mn, wc); mx = v3_max(mx, wc);
        }
        amin[i] = mn; amax[i] = mx;
    }
    v3 smin = amin[0], smax = amax[0]; /* :1502-1511 */
    for (uint32_t i = 1; i < n; ++i) { smin = v3_min(smin, amin[i]); smax = v3_max(smax, amax[i]); }
    v3 ext = v3_sub(smax, smin);
    v3 extent = V(jl_max(ext.x, 1e-6f), jl_max(ext.y, 1e-6f), jl_max(ext.z, 1e-6f)); /* :1517-1521 */
    /* calculate_tlas_morton_code (src/instanced-bvh-kernels.jl:295-313) */
    code_idx* ci = (code_idx*)malloc(sizeof(code_idx) * n);
    for (uint32_t i = 0; i < n; ++i) {
        const rco_instance* in = &s->inst[i];
        const blas_t* b = &s->blas[in->blas_index - 1];
        v3 lc = v3_scale(v3_add(v3_from(b->root_min), v3_from(b->root_max)), 0.5f);
        v3 wc = xf_point(in->transform, lc);
        v3 d = v3_sub(wc, smin);
        float nrm[3] = {d.x / extent.x, d.y / extent.y, d.z / extent.z};
        ci[i].code = rco_morton_code_30bit(nrm); ci[i].idx = i;
    }
    qsort(ci, n, sizeof(code_idx), cmp_code_idx); /* :1533-1540 */
    uint32_t* codes = (uint32_t*)malloc(sizeof(uint32_t) * n);
    for (uint32_t i = 0; i < n; ++i) codes[i] = ci[i].code;
    s->n_nodes = 2 * n - 1;
    s->nodes = (rco_node*)malloc(sizeof(rco_node) * s->n_nodes);
    for (uint32_t i = 0; i < s->n_nodes; ++i) s->nodes[i] = EMPTY_NODE;
    if (n == 1) { /* :1553-1570 */
        rco_node* nd = &s->nodes[0];
        v3_store(nd->aabb0_min, smin); v3_store(nd->aabb0_max, smax);
        nd->child0 = RCO_INVALID_NODE; nd->child1 = ci[0].idx; nd->parent = RCO_INVALID_NODE;
        v3_store(s->root_min, smin); v3_store(s->root_max, smax);
    } else {
        emit_topology_and_parents(s->nodes, codes, (int32_t)n); /* :1574-1578 */
        /* create_tlas_leaf_for_instance (src/instanced-bvh-kernels.jl:332-357) */
        for (uint32_t j = 1; j <= n; ++j) {
            rco_node* nd = &s->nodes[(n - 1 + j) - 1];
            uint32_t parent = nd->parent;
            uint32_t orig = ci[j - 1].idx; /* 0-based original index */
            const rco_instance* in = &s->inst[orig];
            const blas_t* b = &s->blas[in->blas_index - 1];
            v3 mn = V(INFINITY, INFINITY, INFINITY), mx = V(-INFINITY, -INFINITY, -INFINITY);
            for (int c = 1; c <= 8; ++c) {
                v3 wc = xf_point(in->transform, corner(b->root_min, b->root_max, c));
                mn = v3_min(mn, wc); mx = v3_max(mx, wc);
            }
            *nd = EMPTY_NODE;
            v3_store(nd->aabb0_min, mn); v3_store(nd->aabb0_max, mx);
            nd->child0 = RCO_INVALID_NODE; nd->child1 = orig; nd->parent = parent;
        }
        refit(s->nodes, (int32_t)n, 1); /* :1585-1587 */
        v3 rmin, rmax;
        node_aabb(&s->nodes[0], 1, 1, &rmin, &rmax); /* :1590-1591 */
        v3_store(s->root_min, rmin); v3_store(s->root_max, rmax);
    }
    free(codes); free(ci); free(amin); free(amax);
    s->built = 1;
    return 0;
}

uint32_t rco_scene_tlas_nodes(const rco_scene* s, rco_node* out) { if (out && s->n_nodes) memcpy(out, s->nodes, sizeof(rco_node) * s->n_nodes); return s->n_nodes; }
uint32_t rco_scene_instances(const rco_scene* s, rco_instance* out) { if (out && s->n_inst) memcpy(out, s->inst, sizeof(rco_instance) * s->n_inst); return s->n_inst; }
uint32_t rco_scene_blas_nodes(const rco_scene* s, rco_node* out) { if (out && s->n_blas_nodes) memcpy(out, s->blas_nodes, sizeof(rco_node) * s->n_blas_nodes); return s->n_blas_nodes; }
uint32_t rco_scene_blas_prims(const rco_scene* s, rco_tri* out) { if (out && s->n_blas_prims) memcpy(out, s->blas_prims, sizeof(rco_tri) * s->n_blas_prims); return s->n_blas_prims; }
uint32_t rco_scene_blas_descs(const rco_scene* s, rco_blas_desc* out) { if (out && s->n_blas) memcpy(out, s->descs, sizeof(rco_blas_desc) * s->n_blas); return s->n_blas; }
void rco_scene_world_bound(const rco_scene* s, float out[6]) { memcpy(out, s->root_min, 12); memcpy(out + 3, s->root_max, 12); }
uint32_t rco_scene_blas_morton(const rco_scene* s, uint32_t bi, uint32_t* out) {
    if (bi < 1 || bi > s->n_blas) return 0;
    if (out) memcpy(out, s->blas[bi - 1].morton, sizeof(uint32_t) * s->blas[bi - 1].n_prims);
    return s->blas[bi - 1].n_prims;
}

/* ------------------------------------------------------------------------------------------------
 * Traversal (src/instanced-bvh.jl:1733-2140)
 * ---------------------------------------------------------------------------------------------- */
void rco_safe_invdir(const float d[3], float out[3]) { /* :1742-1748 */
    const float ooeps = 1.0e-5f;
    for (int i = 0; i < 3; ++i) out[i] = 1.0f / (fabsf(d[i]) > ooeps ? d[i] : copysignf(ooeps, d[i]));
}
static inline v3 safe_invdir(v3 d) {
    float in[3] = {d.x, d.y, d.z}, o[3];
    rco_safe_invdir(in, o);
    return v3_from(o);
}

/* fast_intersect_triangle (:1756-1797) */
static inline int fast_intersect_triangle(v3 ray_o, v3 ray_d, v3 v0, v3 v1, v3 v2, float t_min, float closest_t,
                                          float* t_out, float* u_out, float* v_out) {
    v3 e1 = v3_sub(v1, v0);
    v3 e2 = v3_sub(v2, v0);
    v3 s1 = v3_cross(ray_d, e2);
    float determinant = v3_dot(s1, e1);
    float invd = 1.0f / determinant;
    v3 d = v3_sub(ray_o, v0);
    float u = v3_dot(d, s1) * invd;
    if (u < 0.0f || u > 1.0f) return 0;
    v3 s2 = v3_cross(d, e1);
    float v = v3_dot(ray_d, s2) * invd;
    if (v < 0.0f || (u + v) > 1.0f) return 0;
    float t = v3_dot(e2, s2) * invd;
    if (t < t_min || t > closest_t) return 0;
    *t_out = t; *u_out = u; *v_out = v;
    return 1;
}

/* fast_intersect_bbox (:1841-1859) */
static inline void fast_intersect_bbox(v3 ray_o, v3 inv_d, const float pmin[3], const float pmax[3], float t_min,
                                       float t_max, float* min_t, float* max_t) {
    v3 ox = V(-ray_o.x * inv_d.x, -ray_o.y * inv_d.y, -ray_o.z * inv_d.z);
    v3 f = V(pmax[0] * inv_d.x + ox.x, pmax[1] * inv_d.y + ox.y, pmax[2] * inv_d.z + ox.z);
    v3 n = V(pmin[0] * inv_d.x + ox.x, pmin[1] * inv_d.y + ox.y, pmin[2] * inv_d.z + ox.z);
    v3 tmax_vec = v3_max(f, n), tmin_vec = v3_min(f, n);
    *max_t = jl_min(jl_min(jl_min(tmax_vec.x, tmax_vec.y), tmax_vec.z), t_max);
    *min_t = jl_max(jl_max(jl_max(tmin_vec.x, tmin_vec.y), tmin_vec.z), t_min);
}

/* intersect_internal_node (:1807-1832) */
static inline void intersect_internal_node(const rco_node* node, v3 inv_d, v3 ray_o, float t_min, float t_max,
                                           uint32_t* near_c, uint32_t* far_c) {
    float t0_min, t0_max, t1_min, t1_max;
    fast_intersect_bbox(ray_o, inv_d, node->aabb0_min, node->aabb0_max, t_min, t_max, &t0_min, &t0_max);
    fast_intersect_bbox(ray_o, inv_d, node->aabb1_min, node->aabb1_max, t_min, t_max, &t1_min, &t1_max);
    uint32_t traverse0 = (t0_min <= t0_max) ? node->child0 : RCO_INVALID_NODE;
    uint32_t traverse1 = (t1_min <= t1_max) ? node->child1 : RCO_INVALID_NODE;
    if (t0_min < t1_min && traverse0 != RCO_INVALID_NODE) { *near_c = traverse0; *far_c = traverse1; }
    else { *near_c = traverse1; *far_c = traverse0; }
}

/* The reference's stack is an unchecked 32-entry MVector (:1912); deeper trees are undefined behaviour
 * there.  The restatement uses 512 entries so every tree it can build is defined. */
#define RCO_STACK 512

/* dev instrumentation: per-node visit histograms (not thread-safe; used single-threaded) */
static uint32_t* g_hist_tlas = NULL;
static uint32_t* g_hist_blas = NULL;
void rco_set_histograms(uint32_t* tlas, uint32_t* blas) { g_hist_tlas = tlas; g_hist_blas = blas; }
static int32_t g_max_sp = 0;  /* dev: deepest stack seen (a maximum kept with relaxed atomics: the pool's threads all report into it) */
static inline void note_sp(int32_t sp) {
    int32_t cur = __atomic_load_n(&g_max_sp, __ATOMIC_RELAXED);
    while (sp > cur && !__atomic_compare_exchange_n(&g_max_sp, &cur, sp, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
/* dev: per-step event trace of one ray (tools/sched_sim.py replays these through wave-scheduling policies).  One byte per loop
 * iteration: low 3 bits = kind (0 TLAS interior, 1 BLAS interior, 2 instance entry, 3 leaf miss, 4 leaf hit), 0x20 = pushed the far
 * child, 0x40 = popped (no near child / after a leaf), 0x80 = that pop returned to the top level (sentinel); a second byte array
 * gets the stack depth after the step. */
static __thread uint8_t* tl_ev = NULL;
static __thread uint8_t* tl_ev_sp = NULL;
static __thread uint32_t tl_ev_cap = 0, tl_ev_n = 0;
/* dev: with an event trace, also the node index visited by each step and the closest t the step started with (tools/tlas_subtree_bound.py) */
static __thread uint32_t* tl_ev_node = NULL;
static __thread float* tl_ev_ct = NULL;
#define EV_STEP(nidx, ct) do { if (tl_ev_node && tl_ev_n < tl_ev_cap) { tl_ev_node[tl_ev_n] = (nidx); tl_ev_ct[tl_ev_n] = (ct); } } while (0)
#define EV_PUT(code, depth) do { if (tl_ev) { if (tl_ev_n < tl_ev_cap) { tl_ev[tl_ev_n] = (uint8_t)(code); if (tl_ev_sp) tl_ev_sp[tl_ev_n] = (uint8_t)((depth) > 255 ? 255 : (depth)); } ++tl_ev_n; } } while (0)
/* dev / test: per instance ENTRY of one ray -- which instance, the closest_t the ray carried when it entered, and how many triangle tests
 * (leaf visits) the reference performed before it left again.  tests/test_entry_cull_predicate.py checks the product's entry cull against
 * this: an entry the cull would skip must have zero leaf visits. */
typedef struct { uint32_t inst; float closest_at_entry; uint32_t leaf_tests; } rco_entry_rec;
static __thread rco_entry_rec* tl_ent = NULL;
static __thread uint32_t tl_ent_cap = 0, tl_ent_n = 0;
#define ENT_BEGIN(i, t) do { if (tl_ent) { if (tl_ent_n < tl_ent_cap) { tl_ent[tl_ent_n].inst = (uint32_t)(i); tl_ent[tl_ent_n].closest_at_entry = (t); tl_ent[tl_ent_n].leaf_tests = 0; } ++tl_ent_n; } } while (0)
#define ENT_LEAF() do { if (tl_ent && tl_ent_n > 0 && tl_ent_n <= tl_ent_cap) tl_ent[tl_ent_n - 1].leaf_tests += 1; } while (0)
int32_t rco_max_stack(int reset) { int32_t v = __atomic_load_n(&g_max_sp, __ATOMIC_RELAXED); if (reset) __atomic_store_n(&g_max_sp, 0, __ATOMIC_RELAXED); return v; }

static void set_miss(rco_hit* h) {
    h->hit = 0; h->t = 0.0f; h->primitive_id = 0xFFFFFFFFu; h->instance_custom_index = 0;
    h->bary_u = 0.0f; h->bary_v = 0.0f; h->instance_id = 0xFFFFFFFFu; h->_pad = 0;
}

/* closest_hit (:1902-2024) and any_hit (:2034-2140) share the loop; `any` selects the any_hit deltas
 * (t_min forced to 0 at :2039, early return at :2106-2115). */
static void traverse(const rco_scene* s, const rco_ray* r, rco_hit* out, uint32_t* counters, int any) {
    set_miss(out);
    if (s->n_nodes == 0) return; /* empty TLAS: reference is UB under @inbounds; its stress test expects a miss (test/test_tlas_stress.jl:808-831) */
    /* check_direction (src/ray.jl:39-49): components == 0 (either sign) become +0 */
    v3 world_o = V(r->ox, r->oy, r->oz);
    v3 world_d = V(r->dx == 0.0f ? 0.0f : r->dx, r->dy == 0.0f ? 0.0f : r->dy, r->dz == 0.0f ? 0.0f : r->dz);
    v3 ray_o = world_o, ray_d = world_d;
    float ray_mint = any ? 0.0f : r->tmin;
    float ray_maxt = r->tmax;
    v3 ray_inv_d = safe_invdir(ray_d);

    uint32_t stack[RCO_STACK];
    int32_t sp = 1;
    stack[sp - 1] = RCO_INVALID_NODE;

    int32_t current_instance = -1, closest_instance = -1;
    uint32_t closest_prim = RCO_INVALID_NODE;
    float hit_u = 0.0f, hit_v = 0.0f;
    uint32_t node_index = 1, current_blas_offset = 0;
    uint32_t n_node = 0, n_inst = 0;

    while (node_index != RCO_INVALID_NODE) {
        const rco_node* node = (current_instance < 0) ? &s->nodes[node_index - 1]
                                                      : &s->blas_nodes[current_blas_offset + node_index - 1];
        ++n_node;
        EV_STEP(node_index, ray_maxt);
        if (g_hist_tlas) { if (current_instance < 0) g_hist_tlas[node_index - 1]++; else g_hist_blas[current_blas_offset + node_index - 1]++; }
        int is_leaf = node->child0 == RCO_INVALID_NODE;
        uint32_t ev = 0;
        if (!is_leaf) {
            uint32_t near_c, far_c;
            intersect_internal_node(node, ray_inv_d, ray_o, ray_mint, ray_maxt, &near_c, &far_c);
            ev = current_instance < 0 ? 0u : 1u;
            if (far_c != RCO_INVALID_NODE) { if (sp < RCO_STACK) stack[sp++] = far_c; note_sp(sp); ev |= 0x20u; }
            if (near_c != RCO_INVALID_NODE) { node_index = near_c; EV_PUT(ev, sp); continue; }
        } else if (current_instance < 0) {
            current_instance = (int32_t)node->child1;
            if (sp < RCO_STACK) stack[sp++] = RCO_TOP_LEVEL_SENTINEL;
            node_index = 1;
            const rco_instance* inst = &s->inst[current_instance];
            const rco_blas_desc* desc = &s->descs[inst->blas_index - 1];
            ++n_inst;
            ENT_BEGIN(current_instance, ray_maxt);
            current_blas_offset = desc->nodes_offset;
            ray_o = xf_point(inst->inv_transform, world_o);
            ray_d = xf_dir(inst->inv_transform, world_d);
            ray_inv_d = safe_invdir(ray_d);
            EV_PUT(2u, sp);
            continue;
        } else {
            float t, u, v;
            ev = 3u;
            ENT_LEAF();
            int hit = fast_intersect_triangle(ray_o, ray_d, v3_from(node->aabb0_min), v3_from(node->aabb0_max),
                                              v3_from(node->aabb1_min), ray_mint, ray_maxt, &t, &u, &v);
            if (hit) {
                if (any) { /* :2106-2115 */
                    const rco_instance* inst = &s->inst[current_instance];
                    const rco_blas_desc* desc = &s->descs[inst->blas_index - 1];
                    out->hit = 1; out->t = t; out->bary_u = u; out->bary_v = v;
                    out->primitive_id = desc->primitives_offset + node->child1 - 1;
                    out->instance_id = (uint32_t)current_instance;
                    out->instance_custom_index = inst->instance_id;
                    if (counters) { counters[0] += n_node; counters[1] += n_inst; }
                    EV_PUT(4u, sp);
                    return;
                }
                ev = 4u;
                ray_maxt = t; closest_instance = current_instance; closest_prim = node->child1;
                hit_u = u; hit_v = v;
            }
        }
        node_index = stack[--sp];
        ev |= 0x40u;
        if (node_index == RCO_TOP_LEVEL_SENTINEL) { /* :1996-2006 */
            node_index = stack[--sp];
            current_instance = -1;
            ray_o = world_o; ray_d = world_d;
            ray_inv_d = safe_invdir(ray_d);
            ev |= 0x80u;
        }
        EV_PUT(ev, sp);
    }
    if (counters) { counters[0] += n_node; counters[1] += n_inst; }
    if (!any && closest_instance >= 0) { /* :2010-2017 */
        const rco_instance* inst = &s->inst[closest_instance];
        const rco_blas_desc* desc = &s->descs[inst->blas_index - 1];
        out->hit = 1; out->t = ray_maxt; out->bary_u = hit_u; out->bary_v = hit_v;
        out->primitive_id = desc->primitives_offset + closest_prim - 1;
        out->instance_id = (uint32_t)closest_instance;
        out->instance_custom_index = inst->instance_id;
    }
}

/* dev experiment ("why leaf tests cannot be taken off the critical path"): closest_hit with every leaf test's RESULT arriving
 * `lag` loop iterations late -- what a kernel would compute if a lane posted its triangle tests to a shared queue and went on walking
 * interior nodes until the answer came back.  The queued tests are resolved in visit order against the then-current closest t (so a hit
 * is still "the first minimum, later equal t replaces"), but the box tests in between prune with a STALE closest t: subtrees the reference
 * prunes are entered, and a triangle in such a subtree whose computed t is <= the closest t -- its box's computed entry was just above it,
 * the two are different roundings of nearly equal numbers, and exactly equal for duplicated geometry -- is accepted although the
 * reference never tested it.  tests/test_oracle_properties.py counts the rays this changes. */
typedef struct { const rco_node* node; int32_t inst; v3 o, d; uint32_t due; } pending_leaf;
static void traverse_deferred(const rco_scene* s, const rco_ray* r, rco_hit* out, uint32_t lag) {
    set_miss(out);
    if (s->n_nodes == 0) return;
    v3 world_o = V(r->ox, r->oy, r->oz);
    v3 world_d = V(r->dx == 0.0f ? 0.0f : r->dx, r->dy == 0.0f ? 0.0f : r->dy, r->dz == 0.0f ? 0.0f : r->dz);
    v3 ray_o = world_o, ray_d = world_d;
    float ray_mint = r->tmin, ray_maxt = r->tmax;
    v3 ray_inv_d = safe_invdir(ray_d);
    uint32_t stack[RCO_STACK];
    int32_t sp = 1;
    stack[0] = RCO_INVALID_NODE;
    int32_t current_instance = -1, closest_instance = -1;
    uint32_t closest_prim = RCO_INVALID_NODE, node_index = 1, current_blas_offset = 0, step = 0;
    float hit_u = 0.0f, hit_v = 0.0f;
    enum { QCAP = 64 };
    pending_leaf q[QCAP];
    uint32_t q_head = 0, q_tail = 0;
#define RESOLVE_DUE(all) while (q_head != q_tail && ((all) || q[q_head % QCAP].due <= step)) { \
        const pending_leaf* p = &q[q_head++ % QCAP]; float t, u, v; \
        if (fast_intersect_triangle(p->o, p->d, v3_from(p->node->aabb0_min), v3_from(p->node->aabb0_max), v3_from(p->node->aabb1_min), ray_mint, ray_maxt, &t, &u, &v)) { \
            ray_maxt = t; closest_instance = p->inst; closest_prim = p->node->child1; hit_u = u; hit_v = v; } }
    while (node_index != RCO_INVALID_NODE) {
        ++step;
        RESOLVE_DUE(0)
        const rco_node* node = (current_instance < 0) ? &s->nodes[node_index - 1] : &s->blas_nodes[current_blas_offset + node_index - 1];
        if (node->child0 != RCO_INVALID_NODE) {
            uint32_t near_c, far_c;
            intersect_internal_node(node, ray_inv_d, ray_o, ray_mint, ray_maxt, &near_c, &far_c);
            if (far_c != RCO_INVALID_NODE && sp < RCO_STACK) stack[sp++] = far_c;
            if (near_c != RCO_INVALID_NODE) { node_index = near_c; continue; }
        } else if (current_instance < 0) {
            current_instance = (int32_t)node->child1;
            if (sp < RCO_STACK) stack[sp++] = RCO_TOP_LEVEL_SENTINEL;
            node_index = 1;
            const rco_instance* inst = &s->inst[current_instance];
            current_blas_offset = s->descs[inst->blas_index - 1].nodes_offset;
            ray_o = xf_point(inst->inv_transform, world_o);
            ray_d = xf_dir(inst->inv_transform, world_d);
            ray_inv_d = safe_invdir(ray_d);
            continue;
        } else {
            if (q_tail - q_head == QCAP) { RESOLVE_DUE(1) }
            pending_leaf* p = &q[q_tail++ % QCAP];
            p->node = node; p->inst = current_instance; p->o = ray_o; p->d = ray_d; p->due = step + lag;
            if (lag == 0) { RESOLVE_DUE(1) }
        }
        node_index = stack[--sp];
        if (node_index == RCO_TOP_LEVEL_SENTINEL) {
            node_index = stack[--sp];
            current_instance = -1;
            ray_o = world_o; ray_d = world_d;
            ray_inv_d = safe_invdir(ray_d);
        }
    }
    RESOLVE_DUE(1)
#undef RESOLVE_DUE
    if (closest_instance >= 0) {
        const rco_instance* inst = &s->inst[closest_instance];
        const rco_blas_desc* desc = &s->descs[inst->blas_index - 1];
        out->hit = 1; out->t = ray_maxt; out->bary_u = hit_u; out->bary_v = hit_v;
        out->primitive_id = desc->primitives_offset + closest_prim - 1;
        out->instance_id = (uint32_t)closest_instance;
        out->instance_custom_index = inst->instance_id;
    }
}
typedef struct { const rco_scene* s; const rco_ray* rays; rco_hit* hits; uint32_t lag; } deferred_ctx;
static void deferred_range(void* p, uint64_t b, uint64_t e);
void rco_closest_hit(const rco_scene* s, const rco_ray* r, rco_hit* h, uint32_t* c) { traverse(s, r, h, c, 0); }
/* dev: trace one ray and record its step events (see tl_ev); returns the number of steps (may exceed cap: then only cap were stored) */
uint32_t rco_trace_events(const rco_scene* s, const rco_ray* r, int any, uint8_t* events, uint8_t* depths, uint32_t cap) {
    rco_hit h;
    tl_ev = events; tl_ev_sp = depths; tl_ev_cap = cap; tl_ev_n = 0;
    traverse(s, r, &h, NULL, any);
    tl_ev = NULL; tl_ev_sp = NULL;
    return tl_ev_n;
}
/* dev: rco_trace_events + per step the visited node index (1-based, in its own tree) and the closest t the step started with */
uint32_t rco_trace_steps(const rco_scene* s, const rco_ray* r, int any, uint8_t* events, uint8_t* depths, uint32_t* nodes, float* closest, uint32_t cap) {
    tl_ev_node = nodes; tl_ev_ct = closest;
    const uint32_t n = rco_trace_events(s, r, any, events, depths, cap);
    tl_ev_node = NULL; tl_ev_ct = NULL;
    return n;
}
void rco_any_hit(const rco_scene* s, const rco_ray* r, rco_hit* h, uint32_t* c) { traverse(s, r, h, c, 1); }
/* entries of ray r (see rco_entry_rec); returns their number (may exceed cap: then only cap were stored) */
uint32_t rco_trace_entries(const rco_scene* s, const rco_ray* r, int any, uint32_t* inst, float* closest_at_entry, uint32_t* leaf_tests, uint32_t cap) {
    rco_hit h;
    rco_entry_rec* rec = (rco_entry_rec*)malloc(sizeof(rco_entry_rec) * (cap ? cap : 1));
    tl_ent = rec; tl_ent_cap = cap; tl_ent_n = 0;
    traverse(s, r, &h, NULL, any);
    tl_ent = NULL;
    const uint32_t n = tl_ent_n, m = n < cap ? n : cap;
    for (uint32_t i = 0; i < m; ++i) { inst[i] = rec[i].inst; closest_at_entry[i] = rec[i].closest_at_entry; leaf_tests[i] = rec[i].leaf_tests; }
    free(rec);
    return n;
}


void rco_brute_closest(const rco_scene* s, const rco_ray* r, rco_hit* out) {
    set_miss(out);
    v3 world_o = V(r->ox, r->oy, r->oz);
    v3 world_d = V(r->dx == 0.0f ? 0.0f : r->dx, r->dy == 0.0f ? 0.0f : r->dy, r->dz == 0.0f ? 0.0f : r->dz);
    float best = r->tmax;
    for (uint32_t i = 0; i < s->n_inst; ++i) {
        const rco_instance* inst = &s->inst[i];
        const rco_blas_desc* desc = &s->descs[inst->blas_index - 1];
        const blas_t* b = &s->blas[inst->blas_index - 1];
        v3 o = xf_point(inst->inv_transform, world_o), d = xf_dir(inst->inv_transform, world_d);
        for (uint32_t j = 0; j < b->n_prims; ++j) {
            const rco_tri* tr = &s->blas_prims[desc->primitives_offset + j];
            float t, u, v;
            if (!fast_intersect_triangle(o, d, v3_from(tr->v[0]), v3_from(tr->v[1]), v3_from(tr->v[2]), r->tmin, best, &t, &u, &v)) continue;
            if (out->hit && !(t < best)) continue; /* first minimum wins */
            best = t;
            out->hit = 1; out->t = t; out->bary_u = u; out->bary_v = v;
            out->primitive_id = desc->primitives_offset + j; out->instance_id = i;
            out->instance_custom_index = inst->instance_id;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * pthread parallel-for (stands in for Threads.@threads, src/kernels.jl:64,82)
 * ---------------------------------------------------------------------------------------------- */
typedef void (*range_fn)(void* ctx, uint64_t begin, uint64_t end);
/* A persistent pool: workers are created once (lazily, grown on demand) and parked on a condition variable between calls, and a
 * call's range is handed out in dynamic chunks from one atomic cursor -- rays differ a lot in cost, so a static partition leaves
 * most threads idle behind the slowest slice, and creating + joining 256 threads per call costs more than a 4 M-ray pass is worth.
 * Calls are serialised by pf_call_mutex (the oracle is test infrastructure; concurrent callers simply queue). */
#define PF_MAX_THREADS 1024
#define PF_CHUNK 4096
static pthread_mutex_t pf_call_mutex = PTHREAD_MUTEX_INITIALIZER;
static pthread_mutex_t pf_mutex = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t pf_start = PTHREAD_COND_INITIALIZER, pf_done = PTHREAD_COND_INITIALIZER;
static pthread_t pf_threads[PF_MAX_THREADS];
static int pf_n_threads = 0;           /* workers created so far */
static uint64_t pf_generation = 0;     /* bumped per call */
static int pf_want = 0, pf_running = 0;/* workers taking part in the current call / still inside it */
static range_fn pf_fn; static void* pf_ctx; static uint64_t pf_n, pf_chunk;
static volatile uint64_t pf_cursor;
static void pf_drain(void) {
    for (;;) {
        uint64_t b = __atomic_fetch_add(&pf_cursor, pf_chunk, __ATOMIC_RELAXED);
        if (b >= pf_n) return;
        uint64_t e = b + pf_chunk < pf_n ? b + pf_chunk : pf_n;
        pf_fn(pf_ctx, b, e);
    }
}
/* Optional: worker k runs on the k-th CPU this process is allowed on (rco_pool_pin(1) before the pool's first use).  A timing run on a
 * 256-thread host should not depend on where the scheduler parks 255 freshly woken threads. */
static int pf_pin = 0;
void rco_pool_pin(int enable) { pf_pin = enable; }
int rco_allowed_cpus(void) {
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) != 0) return 1;
    return CPU_COUNT(&set);
}
static void pf_pin_self(int id) {
    cpu_set_t allowed, one;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    const int n = CPU_COUNT(&allowed);
    if (n <= 0) return;
    int want = (id + 1) % n, seen = 0;  /* worker 0 on the second allowed CPU: the calling thread (which works too) usually sits on the first */
    for (int c = 0; c < CPU_SETSIZE; ++c) {
        if (!CPU_ISSET(c, &allowed)) continue;
        if (seen++ == want) { CPU_ZERO(&one); CPU_SET(c, &one); (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one); return; }
    }
}
static void* pf_worker(void* arg) {
    const int id = (int)(intptr_t)arg;
    uint64_t seen = 0;
    if (pf_pin) pf_pin_self(id);
    pthread_mutex_lock(&pf_mutex);
    for (;;) {
        while (pf_generation == seen || id >= pf_want) {
            seen = pf_generation; /* a call this worker is not part of (or none): nothing to do for it */
            pthread_cond_wait(&pf_start, &pf_mutex);
        }
        seen = pf_generation;
        pthread_mutex_unlock(&pf_mutex);
        pf_drain();
        pthread_mutex_lock(&pf_mutex);
        if (--pf_running == 0) pthread_cond_signal(&pf_done);
    }
    return NULL;
}
static void parallel_for_chunked(uint64_t n, int nthreads, uint64_t chunk, range_fn fn, void* ctx) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > PF_MAX_THREADS) nthreads = PF_MAX_THREADS;
    if (chunk < 1) chunk = 1;
    if ((uint64_t)nthreads > (n + chunk - 1) / chunk) nthreads = n ? (int)((n + chunk - 1) / chunk) : 1;
    if (nthreads == 1) { fn(ctx, 0, n); return; }
    pthread_mutex_lock(&pf_call_mutex);
    pthread_mutex_lock(&pf_mutex);
    const int helpers = nthreads - 1;  /* the calling thread works too */
    while (pf_n_threads < helpers) {
        if (pthread_create(&pf_threads[pf_n_threads], NULL, pf_worker, (void*)(intptr_t)pf_n_threads) != 0) break;
        pthread_detach(pf_threads[pf_n_threads]);
        ++pf_n_threads;
    }
    pf_fn = fn; pf_ctx = ctx; pf_n = n; pf_chunk = chunk; pf_cursor = 0;
    pf_want = helpers < pf_n_threads ? helpers : pf_n_threads;
    pf_running = pf_want;
    ++pf_generation;
    pthread_cond_broadcast(&pf_start);
    pthread_mutex_unlock(&pf_mutex);
    pf_drain();
    pthread_mutex_lock(&pf_mutex);
    while (pf_running > 0) pthread_cond_wait(&pf_done, &pf_mutex);
    pf_want = 0;
    pthread_mutex_unlock(&pf_mutex);
    pthread_mutex_unlock(&pf_call_mutex);
}
static void parallel_for(uint64_t n, int nthreads, range_fn fn, void* ctx) { parallel_for_chunked(n, nthreads, PF_CHUNK, fn, ctx); }

typedef struct { const rco_scene* s; const rco_ray* rays; rco_hit* hits; int mode; uint32_t* counters; } trace_ctx;
static void trace_range(void* p, uint64_t b, uint64_t e) {
    trace_ctx* c = (trace_ctx*)p;
    for (uint64_t i = b; i < e; ++i) traverse(c->s, &c->rays[i], &c->hits[i], c->counters ? c->counters + 2 * i : NULL, c->mode);
}
static void deferred_range(void* p, uint64_t b, uint64_t e) {
    deferred_ctx* c = (deferred_ctx*)p;
    for (uint64_t i = b; i < e; ++i) traverse_deferred(c->s, &c->rays[i], &c->hits[i], c->lag);
}
void rco_trace_deferred_batch(const rco_scene* s, const rco_ray* rays, rco_hit* hits, uint64_t n, uint32_t lag, int nthreads) {
    deferred_ctx c = {s, rays, hits, lag};
    parallel_for(n, nthreads, deferred_range, &c);
}
void rco_trace_batch(const rco_scene* s, const rco_ray* rays, rco_hit* hits, uint64_t n, int mode, int nthreads, uint32_t* counters) {
    if (counters) memset(counters, 0, sizeof(uint32_t) * 2 * n);
    trace_ctx c = {s, rays, hits, mode, counters};
    parallel_for(n, nthreads, trace_range, &c);
}

/* ------------------------------------------------------------------------------------------------
 * Drivers (src/kernels.jl)
 * ---------------------------------------------------------------------------------------------- */
void rco_generate_ray_grid(const rco_scene* s, const float viewdir[3], uint32_t grid, rco_ray* out) {
    /* hits_from_grid (:58-61): ray_direction = normalize(viewdir); generate_ray_grid normalises again (:11) */
    v3 ray_direction = v3_normalize(v3_from(viewdir));
    v3 direction = v3_normalize(ray_direction);
    /* corners of Rect3f(p_min, p_max - p_min): origin + {0,1} .* widths (GeometryBasics coordinates(Rect3)) */
    v3 o = v3_from(s->root_min);
    v3 w = v3_sub(v3_from(s->root_max), v3_from(s->root_min));
    v3 temp = fabsf(direction.x) < 0.9f ? V(1.0f, 0.0f, 0.0f) : V(0.0f, 1.0f, 0.0f); /* :17-21 */
    v3 basis1 = v3_normalize(v3_cross(direction, temp));
    v3 basis2 = v3_normalize(v3_cross(direction, basis1));
    float min1 = INFINITY, max1 = -INFINITY, min2 = INFINITY, max2 = -INFINITY, mind = INFINITY;
    for (int c = 0; c < 8; ++c) {
        v3 p = V(o.x + ((c & 1) ? 1.0f : 0.0f) * w.x, o.y + ((c & 2) ? 1.0f : 0.0f) * w.y, o.z + ((c & 4) ? 1.0f : 0.0f) * w.z);
        float p1 = v3_dot(p, basis1), p2 = v3_dot(p, basis2), pd = v3_dot(p, direction);
        min1 = jl_min(min1, p1); max1 = jl_max(max1, p1);
        min2 = jl_min(min2, p2); max2 = jl_max(max2, p2);
        mind = jl_min(mind, pd);
    }
    float margin = 0.05f * jl_max(max1 - min1, max2 - min2); /* :33 */
    float grid_width = max1 - min1 + 2.0f * margin;
    float grid_height = max2 - min2 + 2.0f * margin;
    float min_depth = mind - margin; /* :39 */
    float c1 = (min1 + max1) / 2.0f, c2 = (min2 + max2) / 2.0f;
    v3 gc = v3_add(v3_add(v3_add(V(0, 0, 0), v3_scale(direction, min_depth)), v3_scale(basis1, c1)), v3_scale(basis2, c2)); /* :41-43 */
    float cell_w = grid_width / (float)grid, cell_h = grid_height / (float)grid; /* :45-46 */
    /* :49-55: u, v are Float64 ((grid_size+1)/2 is Float64), the sum is Float64 and rounds once into Point3f */
    double half = ((double)grid + 1.0) / 2.0;
    for (uint32_t j = 1; j <= grid; ++j)
        for (uint32_t i = 1; i <= grid; ++i) {
            double u = ((double)i - half) * (double)cell_w;
            double v = ((double)j - half) * (double)cell_h;
            rco_ray* r = &out[(size_t)(i - 1) + (size_t)grid * (j - 1)];
            r->ox = (float)(((double)gc.x + u * (double)basis1.x) + v * (double)basis2.x);
            r->oy = (float)(((double)gc.y + u * (double)basis1.y) + v * (double)basis2.y);
            r->oz = (float)(((double)gc.z + u * (double)basis1.z) + v * (double)basis2.z);
            r->tmin = 0.0f; r->dx = ray_direction.x; r->dy = ray_direction.y; r->dz = ray_direction.z; r->tmax = INFINITY;
        }
}

void rco_get_illumination(const rco_scene* s, const float viewdir[3], uint32_t grid, float* out, int nthreads) {
    size_t n = (size_t)grid * grid;
    rco_ray* rays = (rco_ray*)malloc(sizeof(rco_ray) * n);
    rco_hit* hits = (rco_hit*)malloc(sizeof(rco_hit) * n);
    rco_generate_ray_grid(s, viewdir, grid, rays);
    rco_trace_batch(s, rays, hits, n, 0, nthreads, NULL);
    for (uint32_t k = 0; k < s->n_blas_prims; ++k) out[k] = 0.0f;
    for (size_t i = 0; i < n; ++i) { /* :115-122 */
        if (!hits[i].hit) continue;
        uint32_t meta = s->blas_prims[hits[i].primitive_id].meta;
        if (meta >= 1 && meta <= s->n_blas_prims) out[meta - 1] += 1.0f;
    }
    free(rays); free(hits);
}

/* Philox4x32-10 (Salmon et al. 2011, Random123) */
void rco_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline float u32_to_unit(uint32_t x) { return (float)(x >> 8) * 0x1.0p-24f; } /* rand(Float32): 24-bit [0,1) */

/* Julia's Float32 sin/cos/acos evaluate in higher precision and round once.  The restatement evaluates
 * them in f64 with fixed-order polynomial kernels (fdlibm's __kernel_sin/__kernel_cos/acos coefficients)
 * and rounds once to f32; the product uses the same formulas so both sides generate bit-identical rays
 * (parity unpinned against Julia: the reference RNG is unseeded anyway).  Valid for the sampler's ranges:
 * sincos on [0, 2 pi], acos on [0, 1). */
static void rc_sincos_f64(double x, double* s, double* c) {
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, two_over_pi = 6.36619772367581382433e-01;
    int k = (int)(x * two_over_pi + 0.5);
    double r = (x - (double)k * pio2_hi) - (double)k * pio2_lo;
    double z = r * r;
    double sp = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    double ks = r + (z * r) * (-1.66666666666666324348e-01 + z * sp);
    double cp = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    double kc = 1.0 - (0.5 * z - z * cp);
    switch (k & 3) {
        case 0: *s = ks; *c = kc; break;
        case 1: *s = kc; *c = -ks; break;
        case 2: *s = -ks; *c = -kc; break;
        default: *s = -kc; *c = ks; break;
    }
}
static double rc_acos_f64(double x) {
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    if (x < 0.5) {
        double z = x * x;
        double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        double r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    double z = (1.0 - x) * 0.5;
    double s = sqrt(z);
    uint64_t bits; memcpy(&bits, &s, 8); bits &= 0xFFFFFFFF00000000ull;
    double df; memcpy(&df, &bits, 8);
    double c = (z - df * df) / (s + df);
    double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    double r = p / q;
    double w = r * s + c;
    return 2.0 * (df + w);
}
void rco_sincos_f64(double x, double* s, double* c) { rc_sincos_f64(x, s, c); }
double rco_acos_f64(double x) { return rc_acos_f64(x); }

int rco_view_factor_ray(const rco_scene* s, uint32_t src, uint32_t ray_idx, uint64_t seed, rco_ray* out) {
    if (src >= s->n_blas_prims) return -1;
    const rco_tri* tri = &s->blas_prims[src];
    v3 p1 = v3_from(tri->v[0]), p2 = v3_from(tri->v[1]), p3 = v3_from(tri->v[2]);
    v3 nn = v3_cross(v3_sub(p2, p1), v3_sub(p3, p1)); /* GB.orthogonal_vector (:86) */
    v3 normal = v3_normalize(nn);
    /* get_orthogonal_basis (src/math.jl:143-156) */
    v3 n = v3_normalize(normal);
    float ax = fabsf(normal.x), ay = fabsf(normal.y), az = fabsf(normal.z);
    int mi = 1; float mv = ax; /* argmin: first minimum */
    if (ay < mv) { mi = 2; mv = ay; }
    if (az < mv) { mi = 3; mv = az; }
    v3 cand = mi == 1 ? V(1, 0, 0) : (mi == 2 ? V(0, 1, 0) : V(0, 0, 1));
    v3 bv = v3_normalize(v3_cross(n, cand));
    v3 bu = v3_normalize(v3_cross(bv, n));
    uint32_t ctr[4] = {ray_idx, src, 0, 0}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, rnd[4];
    rco_philox4x32_10(ctr, key, rnd);
    float r1 = u32_to_unit(rnd[0]), r2 = u32_to_unit(rnd[1]), xi1 = u32_to_unit(rnd[2]), xi2 = u32_to_unit(rnd[3]);
    /* random_triangle_point (src/math.jl:158-174) */
    float sqrt_r1 = sqrtf(r1);
    float bu_ = 1.0f - sqrt_r1, bv_ = sqrt_r1 * (1.0f - r2), bw_ = sqrt_r1 * r2;
    v3 pt = v3_add(v3_add(v3_scale(p1, bu_), v3_scale(p2, bv_)), v3_scale(p3, bw_));
    v3 o = v3_add(pt, v3_scale(normal, 0.01f)); /* :91 */
    /* random_hemisphere_uniform (src/math.jl:125-141) */
    float theta = (float)rc_acos_f64((double)xi1);
    float phi = (2.0f * 3.1415927f) * xi2;
    double st, ct, sp, cp;
    rc_sincos_f64((double)theta, &st, &ct);
    rc_sincos_f64((double)phi, &sp, &cp);
    float sin_t = (float)st, cos_t = (float)ct, sin_p = (float)sp, cos_p = (float)cp;
    float xl = sin_t * cos_p, yl = sin_t * sin_p, zl = cos_t;
    v3 d = v3_add(v3_add(v3_scale(bu, xl), v3_scale(bv, yl)), v3_scale(normal, zl));
    out->ox = o.x; out->oy = o.y; out->oz = o.z; out->tmin = 0.0f;
    out->dx = d.x; out->dy = d.y; out->dz = d.z; out->tmax = INFINITY;
    return 0;
}

typedef struct { const rco_scene* s; uint32_t rpt; uint64_t seed; uint32_t src_begin, ray_begin, ray_end; uint32_t* out; } vf_ctx;
static void vf_range(void* p, uint64_t b, uint64_t e) {
    vf_ctx* c = (vf_ctx*)p;
    const rco_scene* s = c->s;
    uint32_t N = s->n_blas_prims;
    for (uint64_t k = b; k < e; ++k) {
        uint32_t src = c->src_begin + (uint32_t)k;
        uint32_t tri_idx = s->blas_prims[src].meta;
        for (uint32_t i = c->ray_begin; i < c->ray_end; ++i) {
            rco_ray ray; rco_hit hit;
            rco_view_factor_ray(s, src, i, c->seed, &ray);
            traverse(s, &ray, &hit, NULL, 0);
            if (!hit.hit) continue;
            uint32_t hit_idx = s->blas_prims[hit.primitive_id].meta;
            if (hit_idx != tri_idx && tri_idx >= 1 && tri_idx <= N && hit_idx >= 1 && hit_idx <= N)
                __atomic_fetch_add(&c->out[(size_t)(tri_idx - 1) + (size_t)N * (hit_idx - 1)], 1u, __ATOMIC_RELAXED);
        }
    }
}
/* One source primitive's row of the matrix as a compact N-vector: row[hit_meta - 1] += 1 per counted ray (view_factors! :85-97 for a
 * single src; lets a test check rows of a 50 k x 50 k matrix without allocating it).  `src` = 0-based flat (Morton-sorted) primitive. */
void rco_view_factor_row(const rco_scene* s, uint32_t rpt, uint64_t seed, uint32_t src, uint32_t ray_begin, uint32_t ray_end, uint32_t* row) {
    const uint32_t N = s->n_blas_prims;
    if (src >= N) return;
    if (ray_end > rpt) ray_end = rpt;
    const uint32_t tri_idx = s->blas_prims[src].meta;
    for (uint32_t i = ray_begin; i < ray_end; ++i) {
        rco_ray ray; rco_hit hit;
        rco_view_factor_ray(s, src, i, seed, &ray);
        traverse(s, &ray, &hit, NULL, 0);
        if (!hit.hit) continue;
        const uint32_t hit_idx = s->blas_prims[hit.primitive_id].meta;
        if (hit_idx != tri_idx && tri_idx >= 1 && tri_idx <= N && hit_idx >= 1 && hit_idx <= N) row[hit_idx - 1] += 1u;
    }
}
void rco_view_factors(const rco_scene* s, uint32_t rpt, uint64_t seed, uint32_t src_begin, uint32_t src_end,
                      uint32_t ray_begin, uint32_t ray_end, uint32_t* out, int nthreads) {
    if (src_end > s->n_blas_prims) src_end = s->n_blas_prims;
    if (ray_end > rpt) ray_end = rpt;
    if (src_begin >= src_end || ray_begin >= ray_end) return;
    vf_ctx c = {s, rpt, seed, src_begin, ray_begin, ray_end, out};
    {   /* one item = all rays of one source triangle: hand sources out a few at a time */
        const uint64_t n_src = src_end - src_begin;
        uint64_t chunk = n_src / ((uint64_t)(nthreads > 0 ? nthreads : 1) * 16u);
        parallel_for_chunked(n_src, nthreads, chunk < 1 ? 1 : (chunk > 64 ? 64 : chunk), vf_range, &c);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Wavefront stages next to the trace (docs/src/wavefront-renderer.jl:296-333): hit point, geometric world normal,
 * shadow rays toward one point light.  Same expression order as the product's stage kernels.
 * ---------------------------------------------------------------------------------------------- */
static void hit_frame(const rco_scene* s, const rco_ray* r, const rco_hit* h, v3* point, v3* normal) {
    *point = v3_add(V(r->ox, r->oy, r->oz), v3_scale(V(r->dx, r->dy, r->dz), h->t));
    const rco_tri* tri = &s->blas_prims[h->primitive_id];
    v3 v0 = v3_from(tri->v[0]), v1 = v3_from(tri->v[1]), v2 = v3_from(tri->v[2]);
    v3 nl = v3_cross(v3_sub(v1, v0), v3_sub(v2, v0));
    const float* m = s->inst[h->instance_id].inv_transform;
    v3 nw = V(m[0] * nl.x + m[4] * nl.y + m[8] * nl.z, m[1] * nl.x + m[5] * nl.y + m[9] * nl.z, m[2] * nl.x + m[6] * nl.y + m[10] * nl.z);
    nw = v3_normalize(nw);
    if (v3_dot(nw, V(r->dx, r->dy, r->dz)) > 0.0f) nw = V(-nw.x, -nw.y, -nw.z);
    *normal = nw;
}
void rco_hit_points(const rco_scene* s, const rco_ray* rays, const rco_hit* hits, uint64_t n, float* points, float* normals) {
    for (uint64_t i = 0; i < n; ++i) {
        v3 p = V(0, 0, 0), nn = V(0, 0, 0);
        if (hits[i].hit) hit_frame(s, &rays[i], &hits[i], &p, &nn);
        v3_store(points + 3 * i, p);
        if (normals) v3_store(normals + 3 * i, nn);
    }
}
void rco_shadow_rays(const rco_scene* s, const rco_ray* rays, const rco_hit* hits, uint64_t n, const float light[3], float bias, rco_ray* out) {
    for (uint64_t i = 0; i < n; ++i) {
        rco_ray sr = {0, 0, 0, 0, 0, 0, 1, 0};
        if (hits[i].hit) {
            v3 p, nn;
            hit_frame(s, &rays[i], &hits[i], &p, &nn);
            v3 o = v3_add(p, v3_scale(nn, bias));
            v3 lv = v3_sub(v3_from(light), o);
            float dist = sqrtf(v3_dot(lv, lv));
            sr.ox = o.x; sr.oy = o.y; sr.oz = o.z; sr.tmin = 0.0f;
            sr.dx = lv.x / dist; sr.dy = lv.y / dist; sr.dz = lv.z / dist; sr.tmax = dist;
        }
        out[i] = sr;
    }
}

/* ------------------------------------------------------------------------------------------------
 * BVH4 (src/bvh4.jl)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float mn[3], mx[3]; } b3;

static void bvh4_child_box(const rco_node* node, int interior, b3* out) { /* :265-274 / :288-295 */
    v3 mn, mx;
    node_aabb(node, interior, 0, &mn, &mx);
    v3_store(out->mn, mn); v3_store(out->mx, mx);
}

/* gather_children_bvh2 (:201-300).  Returns child_count. */
static int gather_children_bvh2(uint32_t root_idx, const rco_node* nodes2, uint32_t children[4], b3 aabbs[4], int child_is_leaf[4]) {
    uint32_t queue[8];
    int queue_size = 0, child_count = 0;
    for (int i = 0; i < 4; ++i) {
        children[i] = RCO_INVALID_NODE; child_is_leaf[i] = 0;
        for (int k = 0; k < 3; ++k) { aabbs[i].mn[k] = INFINITY; aabbs[i].mx[k] = -INFINITY; } /* Bounds3() */
    }
    const rco_node* root = &nodes2[root_idx - 1];
    if (root->child0 == RCO_INVALID_NODE) { /* :220-226 */
        children[0] = root_idx; bvh4_child_box(root, 0, &aabbs[0]); child_is_leaf[0] = 1;
        return 1;
    }
    queue[0] = root->child0; queue[1] = root->child1; queue_size = 2;
    while (child_count < 4 && queue_size > 0) { /* :234-277 */
        int best = 0;
        for (int i = 0; i < queue_size; ++i) {
            const rco_node* node = &nodes2[queue[i] - 1];
            if (node->child0 != RCO_INVALID_NODE && child_count + queue_size - 1 + 2 <= 4) { best = i; break; }
        }
        uint32_t node_idx = queue[best];
        queue[best] = queue[queue_size - 1];
        queue_size -= 1;
        const rco_node* node = &nodes2[node_idx - 1];
        int interior = node->child0 != RCO_INVALID_NODE;
        if (interior && child_count + queue_size + 2 <= 4) {
            queue[queue_size++] = node->child0;
            queue[queue_size++] = node->child1;
        } else {
            children[child_count] = node_idx;
            bvh4_child_box(node, interior, &aabbs[child_count]);
            child_is_leaf[child_count] = !interior;
            child_count += 1;
        }
    }
    while (queue_size > 0 && child_count < 4) { /* :280-297 */
        uint32_t node_idx = queue[queue_size - 1];
        queue_size -= 1;
        const rco_node* node = &nodes2[node_idx - 1];
        int interior = node->child0 != RCO_INVALID_NODE;
        children[child_count] = node_idx;
        bvh4_child_box(node, interior, &aabbs[child_count]);
        child_is_leaf[child_count] = !interior;
        child_count += 1;
    }
    return child_count;
}

static void bvh4_leaf(rco_node4* out, const rco_node* node2, uint32_t parent) { /* :368-387, :455-474 */
    b3 box;
    memset(out, 0, sizeof(*out));
    bvh4_child_box(node2, 0, &box);
    out->child[0] = node2->child1; out->child[1] = out->child[2] = out->child[3] = RCO_INVALID_NODE;
    memcpy(out->aabb[0][0], box.mn, 12); memcpy(out->aabb[0][1], box.mx, 12);
    out->parent = parent; out->child_count = 0; out->primitive_count = 1;
}

/* collapse_bvh2_to_bvh4 (:314-497): FIFO over interior subtrees; each task allocates its own node, then one leaf
 * node per leaf child in slot order; interior children are queued with (slot, parent) and patch the parent's
 * child pointer when they are dequeued. */
static uint32_t collapse_bvh2_to_bvh4(const rco_node* nodes2, uint32_t n_nodes2, rco_node4** out_nodes) {
    uint32_t max_nodes4 = n_nodes2 + 1;
    rco_node4* nodes4 = (rco_node4*)calloc(max_nodes4, sizeof(rco_node4));
    uint32_t node4_count = 0;
    typedef struct { uint32_t bvh2_idx; int slot; uint32_t parent4; } task;
    task* queue = (task*)malloc(sizeof(task) * (size_t)max_nodes4);
    size_t q_head = 0, q_tail = 0;
    const rco_node* root = &nodes2[0];
    if (root->child0 == RCO_INVALID_NODE) { /* :334-351 */
        node4_count += 1;
        bvh4_leaf(&nodes4[0], root, RCO_INVALID_NODE);
    } else {
        queue[q_tail++] = (task){1u, 0, RCO_INVALID_NODE}; /* the root task: same body as :404-490 minus the parent patch */
        while (q_head < q_tail) {
            task tk = queue[q_head++];
            uint32_t ch[4]; b3 boxes[4]; int is_leaf[4];
            int count = gather_children_bvh2(tk.bvh2_idx, nodes2, ch, boxes, is_leaf);
            node4_count += 1;
            uint32_t current4 = node4_count;
            if (tk.parent4 != RCO_INVALID_NODE) nodes4[tk.parent4 - 1].child[tk.slot] = current4; /* :415-444 */
            uint32_t child_indices[4] = {RCO_INVALID_NODE, RCO_INVALID_NODE, RCO_INVALID_NODE, RCO_INVALID_NODE};
            for (int i = 0; i < count; ++i) {
                if (is_leaf[i]) {
                    node4_count += 1;
                    child_indices[i] = node4_count;
                    bvh4_leaf(&nodes4[node4_count - 1], &nodes2[ch[i] - 1], current4);
                } else {
                    queue[q_tail++] = (task){ch[i], i, current4};
                }
            }
            rco_node4* nd = &nodes4[current4 - 1];
            memset(nd, 0, sizeof(*nd));
            for (int i = 0; i < 4; ++i) {
                nd->child[i] = child_indices[i];
                memcpy(nd->aabb[i][0], boxes[i].mn, 12); memcpy(nd->aabb[i][1], boxes[i].mx, 12);
            }
            nd->parent = tk.parent4; nd->child_count = (uint8_t)count; nd->primitive_count = 0;
        }
    }
    free(queue);
    *out_nodes = (rco_node4*)realloc(nodes4, sizeof(rco_node4) * node4_count); /* resize! :494 */
    return node4_count;
}

static blas_t* blas4_of(rco_scene* s, uint32_t bi) {
    if (bi == 0 || bi > s->n_blas) return NULL;
    blas_t* b = &s->blas[bi - 1];
    if (!b->nodes4) b->n_nodes4 = collapse_bvh2_to_bvh4(b->nodes, b->n_nodes, &b->nodes4);
    return b;
}

uint32_t rco_blas4_nodes(rco_scene* s, uint32_t bi, rco_node4* out) {
    blas_t* b = blas4_of(s, bi);
    if (!b) return 0;
    if (out) memcpy(out, b->nodes4, sizeof(rco_node4) * b->n_nodes4);
    return b->n_nodes4;
}

/* closest_hit4 (:606-689) and any_hit4 (:696-766) share the loop. */
static void traverse4(const blas_t* b, const rco_ray* r, rco_hit* out, uint32_t* counters, int any) {
    set_miss(out);
    v3 ray_o = V(r->ox, r->oy, r->oz);
    v3 ray_d = V(r->dx == 0.0f ? 0.0f : r->dx, r->dy == 0.0f ? 0.0f : r->dy, r->dz == 0.0f ? 0.0f : r->dz); /* check_direction */
    float ray_mint = 0.0f; /* :610 -- NOT ray.t_min */
    float ray_maxt = r->tmax;
    v3 ray_inv_d = safe_invdir(ray_d);
    uint32_t stack[RCO_STACK];
    int32_t sp = 0;
    uint32_t closest_prim = RCO_INVALID_NODE;
    float hit_u = 0.0f, hit_v = 0.0f;
    uint32_t node_idx = 1, n_node = 0, n_tri = 0;
    for (;;) {
        const rco_node4* node = &b->nodes4[node_idx - 1];
        ++n_node;
        if (node->child_count > 0) {
            /* intersect_all_children4 (:562-599) */
            uint32_t h_idx[4] = {RCO_INVALID_NODE, RCO_INVALID_NODE, RCO_INVALID_NODE, RCO_INVALID_NODE};
            float h_t[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
            int hit_count = 0;
            for (int i = 0; i < (int)node->child_count; ++i) {
                uint32_t child_idx = node->child[i];
                if (child_idx != RCO_INVALID_NODE) {
                    float min_t, max_t;
                    fast_intersect_bbox(ray_o, ray_inv_d, node->aabb[i][0], node->aabb[i][1], ray_mint, ray_maxt, &min_t, &max_t); /* :533-554 */
                    if (min_t <= max_t) { h_idx[hit_count] = child_idx; h_t[hit_count] = min_t; hit_count += 1; }
                }
            }
            for (int i = 1; i < hit_count; ++i) { /* insertion sort :590-596 */
                int j = i;
                while (j > 0 && h_t[j] < h_t[j - 1]) {
                    uint32_t ti = h_idx[j]; h_idx[j] = h_idx[j - 1]; h_idx[j - 1] = ti;
                    float tt = h_t[j]; h_t[j] = h_t[j - 1]; h_t[j - 1] = tt;
                    j -= 1;
                }
            }
            for (int i = hit_count - 1; i >= 1; --i) /* :637-642 */
                if (h_idx[i] != RCO_INVALID_NODE && sp < RCO_STACK) stack[sp++] = h_idx[i];
            note_sp(sp); /* dev: rco_max_stack */
            if (hit_count > 0 && h_idx[0] != RCO_INVALID_NODE) { node_idx = h_idx[0]; continue; }
        } else {
            uint32_t prim_idx = node->child[0];
            if (prim_idx != RCO_INVALID_NODE && prim_idx <= b->n_prims) { /* :652 */
                const rco_tri* tri = &b->prims[prim_idx - 1];
                ++n_tri;
                float t, u, v;
                if (fast_intersect_triangle(ray_o, ray_d, v3_from(tri->v[0]), v3_from(tri->v[1]), v3_from(tri->v[2]), ray_mint, ray_maxt, &t, &u, &v)) {
                    if (any) { /* :744-749 */
                        out->hit = 1; out->t = t; out->bary_u = u; out->bary_v = v; out->primitive_id = prim_idx - 1;
                        if (counters) { counters[0] += n_node; counters[1] += n_tri; }
                        return;
                    }
                    ray_maxt = t; closest_prim = prim_idx; hit_u = u; hit_v = v;
                }
            }
        }
        if (sp > 0) node_idx = stack[--sp]; else break;
    }
    if (counters) { counters[0] += n_node; counters[1] += n_tri; }
    if (!any && closest_prim != RCO_INVALID_NODE) { /* :679-683 */
        out->hit = 1; out->t = ray_maxt; out->bary_u = hit_u; out->bary_v = hit_v; out->primitive_id = closest_prim - 1;
    }
}

typedef struct { const blas_t* b; const rco_ray* rays; rco_hit* hits; int mode; uint32_t* counters; } trace4_ctx;
static void trace4_range(void* p, uint64_t b, uint64_t e) {
    trace4_ctx* c = (trace4_ctx*)p;
    for (uint64_t i = b; i < e; ++i) traverse4(c->b, &c->rays[i], &c->hits[i], c->counters ? c->counters + 2 * i : NULL, c->mode);
}
void rco_trace4_batch(rco_scene* s, uint32_t bi, const rco_ray* rays, rco_hit* hits, uint64_t n, int mode, int nthreads, uint32_t* counters) {
    blas_t* b = blas4_of(s, bi);
    if (!b) { for (uint64_t i = 0; i < n; ++i) set_miss(&hits[i]); return; }
    if (counters) memset(counters, 0, sizeof(uint32_t) * 2 * n);
    trace4_ctx c = {b, rays, hits, mode, counters};
    parallel_for(n, nthreads, trace4_range, &c);
}

/* ------------------------------------------------------------------------------------------------
 * Collision broad phase (src/collision.jl)
 * ---------------------------------------------------------------------------------------------- */
static inline int aabb_overlaps(v3 a_min, v3 a_max, v3 b_min, v3 b_max) { /* :51-53 */
    return (a_max.x >= b_min.x && a_max.y >= b_min.y && a_max.z >= b_min.z) &&
           (a_min.x <= b_max.x && a_min.y <= b_max.y && a_min.z <= b_max.z);
}

/* collide_instances_kernel! (:81-156) for sorted leaf i (1-based).  contacts == NULL: counting pass. */
static uint32_t collide_one(const rco_node* nodes, int32_t n_instances, int32_t i, const uint32_t* contact_counts, rco_contact* contacts) {
    const rco_node* leaf_node = &nodes[(n_instances - 1 + i) - 1];
    v3 a_min, a_max;
    node_aabb(leaf_node, leaf_node->child0 != RCO_INVALID_NODE, 1, &a_min, &a_max); /* tlas_node_aabb :56-66 */
    uint32_t instance_a = leaf_node->child1;
    uint32_t stack[RCO_STACK];
    int32_t sp = 0;
    uint32_t node_index = 1, count = 0;
    for (;;) {
        const rco_node* node = &nodes[node_index - 1];
        if (node->child0 != RCO_INVALID_NODE) {
            int overlap0 = aabb_overlaps(a_min, a_max, v3_from(node->aabb0_min), v3_from(node->aabb0_max));
            int overlap1 = aabb_overlaps(a_min, a_max, v3_from(node->aabb1_min), v3_from(node->aabb1_max));
            if (overlap0 && overlap1) { if (sp < RCO_STACK) stack[sp++] = node->child1; node_index = node->child0; continue; }
            else if (overlap0) { node_index = node->child0; continue; }
            else if (overlap1) { node_index = node->child1; continue; }
        } else {
            uint32_t instance_b = node->child1;
            if (instance_b > instance_a) {
                if (aabb_overlaps(a_min, a_max, v3_from(node->aabb0_min), v3_from(node->aabb0_max))) {
                    count += 1;
                    if (contacts) {
                        uint32_t write_idx = contact_counts[i - 1] - count + 1; /* :135 */
                        contacts[write_idx - 1].instance_a = instance_a + 1;
                        contacts[write_idx - 1].instance_b = instance_b + 1;
                    }
                }
            }
        }
        if (sp > 0) node_index = stack[--sp]; else break;
    }
    return count;
}

uint64_t rco_collide_instances(const rco_scene* s, rco_contact* out, uint32_t* counts_out) {
    int32_t n = (int32_t)s->n_inst;
    if (n == 0) return 0; /* :192-195 */
    uint32_t* counts = (uint32_t*)calloc((size_t)n, sizeof(uint32_t));
    for (int32_t i = 1; i <= n; ++i) counts[i - 1] = collide_one(s->nodes, n, i, NULL, NULL);
    for (int32_t i = 1; i < n; ++i) counts[i] += counts[i - 1]; /* AK.accumulate!(+) :215 */
    uint64_t total = counts[n - 1];
    if (out && total) for (int32_t i = 1; i <= n; ++i) collide_one(s->nodes, n, i, counts, out);
    if (counts_out) memcpy(counts_out, counts, sizeof(uint32_t) * (size_t)n);
    free(counts);
    return total;
}

int rco_collide_instances_any(const rco_scene* s, uint32_t a_first, uint32_t a_count, uint32_t b_first, uint32_t b_count) {
    uint32_t n = s->n_inst;
    for (uint32_t ia = a_first + 1; ia <= a_first + a_count; ++ia)
        for (uint32_t ib = b_first + 1; ib <= b_first + b_count; ++ib) {
            const rco_node* la = &s->nodes[(n - 1 + ia) - 1]; /* :252-253 */
            const rco_node* lb = &s->nodes[(n - 1 + ib) - 1];
            v3 amn, amx, bmn, bmx;
            node_aabb(la, la->child0 != RCO_INVALID_NODE, 1, &amn, &amx);
            node_aabb(lb, lb->child0 != RCO_INVALID_NODE, 1, &bmn, &bmx);
            if (aabb_overlaps(amn, amx, bmn, bmx)) return 1;
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Full Triangle records and the shading epilogue
 * ---------------------------------------------------------------------------------------------- */
static void full_triangle(const blas_t* b, uint32_t j, rco_triangle* t) { /* build_triangle :555-566 for sorted primitive j (0-based) */
    const rco_tri* p = &b->prims[j];
    memcpy(t->vertices, p->v, 36);
    for (int k = 0; k < 3; ++k) t->tangents[k][0] = t->tangents[k][1] = t->tangents[k][2] = NAN;
    static const float default_uv[3][2] = {{0, 0}, {1, 0}, {1, 1}}; /* :561-565 */
    memcpy(t->uv, default_uv, sizeof(default_uv));
    if (b->has_attrs) {
        const uint32_t* idx = b->m_indices + 3 * (size_t)b->src_face[j];
        for (int k = 0; k < 3; ++k) {
            memcpy(t->normals[k], b->m_normals + 3 * (size_t)idx[k], 12);
            if (b->m_uvs) memcpy(t->uv[k], b->m_uvs + 2 * (size_t)idx[k], 8);
        }
    } else {
        v3 v0 = v3_from(p->v[0]), v1 = v3_from(p->v[1]), v2 = v3_from(p->v[2]);
        v3 n = v3_normalize(v3_cross(v3_sub(v1, v0), v3_sub(v2, v0)));
        for (int k = 0; k < 3; ++k) v3_store(t->normals[k], n);
    }
    t->metadata = p->meta;
}

uint32_t rco_scene_triangles(const rco_scene* s, rco_triangle* out) {
    uint32_t n = 0;
    for (uint32_t i = 0; i < s->n_blas; ++i) {
        if (out) for (uint32_t j = 0; j < s->blas[i].n_prims; ++j) full_triangle(&s->blas[i], j, &out[n + j]);
        n += s->blas[i].n_prims;
    }
    return n;
}

void rco_shading_attributes(const rco_scene* s, const rco_hit* hits, uint64_t n, float* normals, float* uvs) {
    for (uint64_t i = 0; i < n; ++i) {
        float nn[3] = {0, 0, 0}, uv[2] = {0, 0};
        if (hits[i].hit) {
            /* locate the BLAS that owns flat primitive primitive_id */
            uint32_t pid = hits[i].primitive_id, bi = 0;
            while (bi + 1 < s->n_blas && s->descs[bi + 1].primitives_offset <= pid) ++bi;
            rco_triangle t;
            full_triangle(&s->blas[bi], pid - s->descs[bi].primitives_offset, &t);
            float u = hits[i].bary_u, v = hits[i].bary_v;
            float b1 = (1.0f - u) - v, b2 = u, b3 = v; /* :2015 */
            v3 sum = v3_add(v3_add(v3_scale(v3_from(t.normals[0]), b1), v3_scale(v3_from(t.normals[1]), b2)), v3_scale(v3_from(t.normals[2]), b3));
            v3 nrm = v3_normalize(sum);
            nn[0] = nrm.x; nn[1] = nrm.y; nn[2] = nrm.z;
            uv[0] = (t.uv[0][0] * b1 + t.uv[1][0] * b2) + t.uv[2][0] * b3;
            uv[1] = (t.uv[0][1] * b1 + t.uv[1][1] * b2) + t.uv[2][1] * b3;
        }
        if (normals) memcpy(normals + 3 * i, nn, 12);
        if (uvs) memcpy(uvs + 2 * i, uv, 8);
    }
}

void rco_primary_rays_lookat(const float pos[3], const float right[3], const float up[3], const float forward[3], float half_width,
                             float half_height, uint32_t width, uint32_t height, uint32_t samples, uint64_t seed, int jitter, rco_ray* out) {
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (uint32_t y = 1; y <= height; ++y)
        for (uint32_t x = 1; x <= width; ++x) {
            uint64_t pixel_idx = (uint64_t)(y - 1) * width + x; /* :234 */
            for (uint32_t sidx = 1; sidx <= samples; ++sidx) {
                uint64_t ray_idx = (pixel_idx - 1) * samples + sidx; /* 1-based :237 */
                uint64_t i = ray_idx - 1;
                float j1 = 0.5f, j2 = 0.5f;
                if (jitter) {
                    uint32_t ctr[4] = {(uint32_t)i, (uint32_t)(i >> 32), 0u, 0x50524159u}, rnd[4];
                    rco_philox4x32_10(ctr, key, rnd);
                    j1 = u32_to_unit(rnd[0]); j2 = u32_to_unit(rnd[1]);
                }
                float u = 2.0f * ((float)x - 0.5f + j1) / (float)width - 1.0f;   /* :241 */
                float v = 1.0f - 2.0f * ((float)y - 0.5f + j2) / (float)height;  /* :242 */
                v3 d = v3_normalize(v3_add(v3_add(v3_from(forward), v3_scale(v3_from(right), u * half_width)), v3_scale(v3_from(up), v * half_height)));
                rco_ray r = {pos[0], pos[1], pos[2], 0.0f, d.x, d.y, d.z, INFINITY};
                out[i] = r;
            }
        }
}

void rco_reflection_rays(const rco_scene* s, const rco_ray* rays, const rco_hit* hits, uint64_t n, float bias, rco_ray* out) {
    for (uint64_t i = 0; i < n; ++i) {
        rco_ray rr = {0, 0, 0, 0, 0, 0, 1, 0}; /* dummy_ray :445 */
        if (hits[i].hit) {
            float nn[3];
            rco_shading_attributes(s, &hits[i], 1, nn, NULL);
            v3 nrm = v3_from(nn);
            v3 o = V(rays[i].ox, rays[i].oy, rays[i].oz), d = V(rays[i].dx, rays[i].dy, rays[i].dz);
            v3 hp = v3_add(o, v3_scale(d, hits[i].t));                       /* ray.o + ray.d * dist :453 */
            v3 wo = V(-d.x, -d.y, -d.z);
            float k = 2.0f * v3_dot(wo, nrm);                                /* reflect (src/math.jl:80) */
            v3 rd = v3_add(V(-wo.x, -wo.y, -wo.z), v3_scale(nrm, k));
            v3 ro = v3_add(hp, v3_scale(nrm, bias));                         /* :467 */
            rr.ox = ro.x; rr.oy = ro.y; rr.oz = ro.z; rr.tmin = 0.0f;
            rr.dx = rd.x; rr.dy = rd.y; rr.dz = rd.z; rr.tmax = INFINITY;
        }
        out[i] = rr;
    }
}
