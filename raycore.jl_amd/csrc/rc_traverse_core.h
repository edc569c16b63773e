// rc_traverse_core.h -- the per-ray traversal (one statement-for-statement restatement of the reference's
// closest_hit / any_hit loop) shared by the trace kernels (rc_traverse.hip) and the drivers (rc_drivers.hip).
//
// Reference: closest_hit src/instanced-bvh.jl:1902-2024, any_hit :2034-2140, safe_invdir :1742-1748,
// fast_intersect_bbox :1841-1859, intersect_internal_node :1807-1832, fast_intersect_triangle :1756-1797.
#pragma once
#include "rc_internal.h"

namespace rc {

constexpr int kBlock = 256;
constexpr int kLdsStack = 24;    // entries per lane kept in LDS (24 KiB per 256-thread block)
constexpr int kTotalStack = 128; // Karras trees over (30-bit code, index) keys are <= 62 deep each => 126 max

struct SceneView {  // the StaticTLAS arrays a kernel reads (src/instanced-bvh.jl:155-168)
    const RcNode* tlas_nodes;
    const RcNode* blas_nodes;
    const RcInstRec* inst;
    const RcPrim* prims;
    uint32_t n_tlas_nodes;
    uint32_t n_prims;
    uint32_t tlas_off;        // a copy of the TLAS nodes sits at blas_nodes[tlas_off ...] (single-base addressing)
    uint32_t* overflow;       // [kTotalStack][total_threads] spill area of the lane stacks (entries below the LDS depth unused)
    uint32_t total_threads;
    uint32_t* status;         // [0] = stack overflow flag
};

struct TraceArgs {
    SceneView v;
    const RcRay* rays;
    RcHit* hits;
    uint64_t n_rays;
    unsigned long long* work_counter;  // persistent kernel: next unclaimed ray index
    int refill;                        // persistent kernel: refill when this many lanes are idle
    uint32_t pool;                     // persistent kernels: ray indices claimed per atomic
    int sched_thr;                     // scheduled kernel: run a leaf/entry batch once this many lanes wait for it
    unsigned long long* stats;         // optional instrumentation (dev builds), else nullptr
};

// Address-space-qualified pointers keep the two halves of the stack on their own instruction paths
// (ds_read/ds_write_b32 for LDS, global_load/store for the spill area); with generic pointers the compiler
// if-converts push/pop into a pointer select + flat_load, which sends every pop through the texture path.
typedef uint32_t __attribute__((address_space(3))) lds_u32;
typedef uint32_t __attribute__((address_space(1))) glb_u32;

template <int LDS_N>
struct LaneStackT {
    lds_u32* lds;        // &lds_stack[threadIdx.x]
    glb_u32* ovf;        // &overflow[global thread id]
    uint32_t ovf_stride;
    uint32_t* status;
    __device__ inline LaneStackT(uint32_t* lds_base, uint32_t* ovf_base, uint32_t stride, uint32_t* st)
        : lds((lds_u32*)lds_base), ovf((glb_u32*)ovf_base), ovf_stride(stride), status(st) {}
    __device__ inline void push(int& sp, uint32_t v) {
        if (__builtin_expect(sp < LDS_N, 1)) lds[sp * kBlock] = v;
        else if (sp < kTotalStack) ovf[(size_t)(sp - LDS_N) * ovf_stride] = v;
        else { *status = 1u; return; }
        ++sp;
    }
    __device__ inline uint32_t pop(int& sp) {
        --sp;
        if (__builtin_expect(sp < LDS_N, 1)) return lds[sp * kBlock];
        return ovf[(size_t)(sp - LDS_N) * ovf_stride];
    }
};
using LaneStack = LaneStackT<kLdsStack>;

struct NodeRegs {
    float4 a, b, c;
    uint4 d;
};
__device__ inline NodeRegs load_node(const RcNode* p) {
    const float4* q = reinterpret_cast<const float4*>(p);
    NodeRegs r;
    r.a = q[0]; r.b = q[1]; r.c = q[2];
    r.d = *reinterpret_cast<const uint4*>(q + 3);
    return r;
}

// Per-ray traversal state.  `cull_t` is closest_t with NaN mapped to -inf: Julia's min(x, NaN) = NaN makes
// every later box test fail once a NaN-t hit was accepted; comparing against -inf
// gives the same outcome with v_min_f32, which would otherwise drop the NaN.
struct RayState {
    float3_ wo, wd, winv;           // world ray (direction sanitised by check_direction), safe_invdir(world d)
    float3_ o, d, inv, ox;          // current-level ray, 1/d, (-o)*inv
    float tmin, closest_t, cull_t;
    float hit_u, hit_v;
    uint32_t closest_prim;
    int closest_inst, cur_inst;
    uint32_t node, blas_off;
    int sp;
};

template <class Stack>
__device__ inline void init_ray(RayState& s, const RcRay& r, bool any_hit, Stack& st, uint32_t tlas_off) {
    // check_direction (src/ray.jl:39-49): -0 and +0 both become +0
    s.wo = mk3(r.ox, r.oy, r.oz);
    s.wd = mk3(r.dx == 0.0f ? 0.0f : r.dx, r.dy == 0.0f ? 0.0f : r.dy, r.dz == 0.0f ? 0.0f : r.dz);
    s.winv = mk3(safe_inv1(s.wd.x), safe_inv1(s.wd.y), safe_inv1(s.wd.z));
    s.o = s.wo; s.d = s.wd; s.inv = s.winv;
    s.ox = mk3(-s.o.x * s.inv.x, -s.o.y * s.inv.y, -s.o.z * s.inv.z);
    s.tmin = any_hit ? 0.0f : r.tmin;  // any_hit forces t_min = 0 (:2039)
    s.closest_t = r.tmax;
    s.cull_t = (r.tmax != r.tmax) ? -INFINITY : r.tmax;
    s.hit_u = s.hit_v = 0.0f;
    s.closest_prim = RC_INVALID_NODE;
    s.closest_inst = -1; s.cur_inst = -1;
    s.node = 1; s.blas_off = tlas_off;
    s.sp = 0;
    st.push(s.sp, RC_INVALID_NODE);
}

// fast_intersect_bbox (:1841-1859)
__device__ inline void slab(const RayState& s, float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                            float& min_t, float& max_t) {
    float fx = mxx * s.inv.x + s.ox.x, fy = mxy * s.inv.y + s.ox.y, fz = mxz * s.inv.z + s.ox.z;
    float nx = mnx * s.inv.x + s.ox.x, ny = mny * s.inv.y + s.ox.y, nz = mnz * s.inv.z + s.ox.z;
    float tmaxx = fmaxf(fx, nx), tmaxy = fmaxf(fy, ny), tmaxz = fmaxf(fz, nz);
    float tminx = fminf(fx, nx), tminy = fminf(fy, ny), tminz = fminf(fz, nz);
    max_t = fminf(fminf(fminf(tmaxx, tmaxy), tmaxz), s.cull_t);
    min_t = fmaxf(fmaxf(fmaxf(tminx, tminy), tminz), s.tmin);
}

// One iteration of the reference's while loop (:1936-2007).  Returns false when the ray has terminated.
template <bool ANY, class Stack>
__device__ inline bool step(RayState& s, const SceneView& a, Stack& st) {
    const RcNode* np = a.blas_nodes + (s.blas_off + s.node - 1);  // packed traversal copy; TLAS nodes sit at blas_off = tlas_off
    NodeRegs nd = load_node(np);
    if (nd.d.x != RC_INVALID_NODE) {
        // intersect_internal_node (:1807-1832)
        float t0_min, t0_max, t1_min, t1_max;
        slab(s, nd.a.x, nd.a.y, nd.c.x, nd.a.z, nd.a.w, nd.c.y, t0_min, t0_max);  // packed order, see rc_pack_node
        slab(s, nd.b.x, nd.b.y, nd.c.z, nd.b.z, nd.b.w, nd.c.w, t1_min, t1_max);
        uint32_t trav0 = (t0_min <= t0_max) ? nd.d.x : RC_INVALID_NODE;
        uint32_t trav1 = (t1_min <= t1_max) ? nd.d.y : RC_INVALID_NODE;
        bool first0 = (t0_min < t1_min) && (trav0 != RC_INVALID_NODE);
        uint32_t near_c = first0 ? trav0 : trav1, far_c = first0 ? trav1 : trav0;
        if (far_c != RC_INVALID_NODE) st.push(s.sp, far_c);
        if (near_c != RC_INVALID_NODE) { s.node = near_c; return true; }
    } else if (s.cur_inst < 0) {
        // top-level leaf: enter the instance (:1961-1977)
        s.cur_inst = (int)nd.d.y;
        st.push(s.sp, RC_TOP_LEVEL_SENTINEL);
        s.node = 1;
        const float4* q = reinterpret_cast<const float4*>(a.inst + s.cur_inst);
        float4 m0 = q[0], m1 = q[1], m2 = q[2];
        uint4 m3 = *reinterpret_cast<const uint4*>(q + 3);
        s.blas_off = m3.x;
        s.o = mk3(m0.x * s.wo.x + m0.y * s.wo.y + m0.z * s.wo.z + m0.w, m1.x * s.wo.x + m1.y * s.wo.y + m1.z * s.wo.z + m1.w,
                  m2.x * s.wo.x + m2.y * s.wo.y + m2.z * s.wo.z + m2.w);
        s.d = mk3(m0.x * s.wd.x + m0.y * s.wd.y + m0.z * s.wd.z, m1.x * s.wd.x + m1.y * s.wd.y + m1.z * s.wd.z,
                  m2.x * s.wd.x + m2.y * s.wd.y + m2.z * s.wd.z);
        s.inv = mk3(safe_inv1(s.d.x), safe_inv1(s.d.y), safe_inv1(s.d.z));
        s.ox = mk3(-s.o.x * s.inv.x, -s.o.y * s.inv.y, -s.o.z * s.inv.z);
        return true;
    } else {
        // bottom-level leaf: fast_intersect_triangle (:1756-1797) on the vertices stored in the node
        float3_ v0 = mk3(nd.a.x, nd.a.y, nd.c.x), v1 = mk3(nd.a.z, nd.a.w, nd.c.y), v2 = mk3(nd.b.x, nd.b.y, nd.c.z);
        float3_ e1 = sub3(v1, v0), e2 = sub3(v2, v0);
        float3_ s1 = cross3(s.d, e2);
        float det = dot3(s1, e1);
        float invd = 1.0f / det;
        float3_ dd = sub3(s.o, v0);
        float u = dot3(dd, s1) * invd;
        float3_ s2 = cross3(dd, e1);
        float v = dot3(s.d, s2) * invd;
        float t = dot3(e2, s2) * invd;
        bool hit = !(u < 0.0f || u > 1.0f) && !(v < 0.0f || (u + v) > 1.0f) && !(t < s.tmin || t > s.closest_t);
        if (hit) {
            s.closest_t = t;
            s.cull_t = (t != t) ? -INFINITY : t;
            s.closest_inst = s.cur_inst;
            s.closest_prim = nd.d.y;
            s.hit_u = u; s.hit_v = v;
            if (ANY) return false;  // :2106-2115
        }
    }
    // pop (:1991-2006)
    s.node = st.pop(s.sp);
    if (s.node == RC_TOP_LEVEL_SENTINEL) {
        s.node = st.pop(s.sp);
        s.cur_inst = -1;
        s.blas_off = a.tlas_off;
        s.o = s.wo; s.d = s.wd; s.inv = s.winv;
        s.ox = mk3(-s.o.x * s.inv.x, -s.o.y * s.inv.y, -s.o.z * s.inv.z);
    }
    return s.node != RC_INVALID_NODE;
}

__device__ inline void write_hit(const RayState& s, const SceneView& a, RcHit* hits, uint64_t ray_index) {
    uint4 w0, w1;
    if (s.closest_inst >= 0) {  // :2010-2017
        const uint4 m3 = *(reinterpret_cast<const uint4*>(a.inst + s.closest_inst) + 3);
        w0 = make_uint4(1u, __float_as_uint(s.closest_t), m3.y + s.closest_prim - 1u, m3.z);
        w1 = make_uint4(__float_as_uint(s.hit_u), __float_as_uint(s.hit_v), (uint32_t)s.closest_inst, 0u);
    } else {  // :2018-2023
        w0 = make_uint4(0u, 0u, RC_INVALID_NODE, 0u);
        w1 = make_uint4(0u, 0u, RC_INVALID_NODE, 0u);
    }
    uint4* out = reinterpret_cast<uint4*>(hits + ray_index);
    out[0] = w0;
    out[1] = w1;
}

__device__ inline RcRay load_ray(const RcRay* rays, uint64_t i) {
    const float4* q = reinterpret_cast<const float4*>(rays + i);
    float4 a = q[0], b = q[1];
    return RcRay{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
}


// Whole-ray traversal for callers that do not interleave rays (drivers).
template <bool ANY, class Stack>
__device__ inline void trace_ray(RayState& s, const RcRay& r, const SceneView& a, Stack& st) {
    init_ray(s, r, ANY, st, a.tlas_off);
    if (a.n_tlas_nodes != 0)
        while (step<ANY>(s, a, st)) {}
}

// 0-based flat primitive index of the accepted hit (valid when s.closest_inst >= 0), :2012-2014
__device__ inline uint32_t hit_prim_index(const RayState& s, const SceneView& a) {
    const uint4 m3 = *(reinterpret_cast<const uint4*>(a.inst + s.closest_inst) + 3);
    return m3.y + s.closest_prim - 1u;
}

}  // namespace rc

rc::SceneView rc_scene_view(rc_scene* s, uint32_t total_threads);  // rc_traverse.hip
