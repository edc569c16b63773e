// rc_device.h -- device-side records and float helpers shared by the build, traversal and driver kernels.
//
// Everything here is compiled for gfx950 with -ffp-contract=off: the reference (Julia) never contracts
// a*b+c, and bit-exact hit ids need the same rounding sequence (SURVEY.md section 7 "hard parts").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RC_INVALID_NODE 0xFFFFFFFFu        // INVALID_NODE, src/instanced-bvh.jl:65
#define RC_TOP_LEVEL_SENTINEL 0xFFFFFFFEu  // TOP_LEVEL_SENTINEL, src/instanced-bvh.jl:1733

// BVHNode2 (src/instanced-bvh.jl:50-63) padded from 60 to 64 bytes so that a node is one aligned
// half cache line and is fetched with four 16-byte loads.  f[0..2]=aabb0_min f[3..5]=aabb0_max
// f[6..8]=aabb1_min f[9..11]=aabb1_max; a BLAS leaf keeps v0,v1,v2 in f[0..8] (BVH2IL layout,
// src/instanced-bvh-kernels.jl:198-215); a TLAS leaf keeps the instance's world AABB in f[0..5].
struct __attribute__((aligned(64))) RcNode {
    float f[12];
    uint32_t child0, child1, parent, pad;
};
static_assert(sizeof(RcNode) == 64, "RcNode must be 64 bytes");

// Traversal copy of a node: the same 12 floats, permuted so that the box test runs on packed-f32 pairs that
// are already adjacent (and 8-byte aligned) in the fetched registers:
//   dword 0..3  = aabb0_min.x, aabb0_min.y, aabb0_max.x, aabb0_max.y     (x,y of child 0)
//   dword 4..7  = aabb1_min.x, aabb1_min.y, aabb1_max.x, aabb1_max.y     (x,y of child 1)
//   dword 8..11 = aabb0_min.z, aabb0_max.z, aabb1_min.z, aabb1_max.z     (z of both children)
// so six v_pk_mul_f32 + six v_pk_add_f32 against (inv.x,inv.y)/(ox.x,ox.y) and a broadcast inv.z/ox.z replace
// twelve multiplies and twelve adds; each product/sum is the same IEEE operation as before (no fusion).
// A BLAS leaf (v0,v1,v2 in the canonical f[0..8]) therefore reads v0=(p0,p1,p8) v1=(p2,p3,p9) v2=(p4,p5,p10).
// The canonical (reference-layout) arrays are kept next to it for rc_export_* and the build kernels.
__host__ __device__ inline RcNode rc_pack_node(const RcNode& n) {
    RcNode p;
    p.f[0] = n.f[0]; p.f[1] = n.f[1]; p.f[2] = n.f[3]; p.f[3] = n.f[4];
    p.f[4] = n.f[6]; p.f[5] = n.f[7]; p.f[6] = n.f[9]; p.f[7] = n.f[10];
    p.f[8] = n.f[2]; p.f[9] = n.f[5]; p.f[10] = n.f[8]; p.f[11] = n.f[11];
    p.child0 = n.child0; p.child1 = n.child1; p.parent = n.parent; p.pad = 0;
    return p;
}

// Traversal copy of a BLAS LEAF (round 3).  The canonical record holds v0, v1, v2 (BVH2IL layout); the Moeller-Trumbore test needs v0 and
// the edges e1 = v1 - v0, e2 = v2 - v0, and its two cross products d x e2 and (o - v0) x e1 are, two components at a time,
//   (a.y, a.z) * (b.z, b.x) - (a.z, a.x) * (b.y, b.z)
// -- packed-f32 work on the pairs (y, z) and (z, x) of one operand and (z, x) and (y, z) of the other.  So the copy stores exactly those
// pairs, 8-byte aligned: three 16-byte loads land them in even-aligned register pairs and the test is 8 packed + ~30 plain VALU instead
// of ~70 with 21 register moves (the compiler's pairing of the straightforward form).  The edges are the SAME IEEE subtractions the test
// did per ray, done once per leaf at pack time: bit-identical results.
//   dword 0..3  = v0.y, v0.z, v0.z, v0.x        dword 4..7 = e1.z, e1.x, e1.y, e1.z        dword 8..11 = e2.z, e2.x, e2.y, e2.z
__host__ __device__ inline RcNode rc_pack_leaf(const RcNode& n) {
    RcNode p;
    const float e1x = n.f[3] - n.f[0], e1y = n.f[4] - n.f[1], e1z = n.f[5] - n.f[2];
    const float e2x = n.f[6] - n.f[0], e2y = n.f[7] - n.f[1], e2z = n.f[8] - n.f[2];
    p.f[0] = n.f[1]; p.f[1] = n.f[2]; p.f[2] = n.f[2]; p.f[3] = n.f[0];
    p.f[4] = e1z; p.f[5] = e1x; p.f[6] = e1y; p.f[7] = e1z;
    p.f[8] = e2z; p.f[9] = e2x; p.f[10] = e2y; p.f[11] = e2z;
    p.child0 = n.child0; p.child1 = n.child1; p.parent = n.parent; p.pad = 0;
    return p;
}

// Traversal record of a BVHNode4 (src/bvh4.jl:40-69), 128 bytes; word layout documented in rc_bvh4.hip.
struct __attribute__((aligned(128))) RcNode4 {
    uint32_t w[32];
};
static_assert(sizeof(RcNode4) == 128, "RcNode4 must be 128 bytes");

// What the traversal needs of an InstanceDescriptor (src/instanced-bvh.jl:90-96) + its BLASDescriptor
// (:132-136), folded into one 64-byte record: one aligned fetch per TLAS-leaf entry instead of the
// reference's 108-byte + 32-byte pair.
struct __attribute__((aligned(64))) RcInstRec {
    float inv[12];  // inv_transform, Vulkan row-major 3x4
    uint32_t nodes_offset, prims_offset, custom_index, n_prims;  // n_prims of the instance's BLAS: node index >= n_prims <=> leaf
};
static_assert(sizeof(RcInstRec) == 64, "RcInstRec must be 64 bytes");

// Reference-layout InstanceDescriptor (108 bytes) kept on the device for the TLAS build / refit kernels.
struct RcInstanceDesc {
    uint32_t blas_index;  // 1-based
    uint32_t instance_id;
    float transform[12];
    float inv_transform[12];
    uint32_t flags;
};
static_assert(sizeof(RcInstanceDesc) == 108, "InstanceDescriptor is 108 bytes");

struct RcBlasDesc {  // BLASDescriptor, 32 bytes
    uint32_t nodes_offset, primitives_offset;
    float root_min[3], root_max[3];
};
static_assert(sizeof(RcBlasDesc) == 32, "BLASDescriptor is 32 bytes");

struct RcPrim {  // vertices + metadata of a Triangle{UInt32}
    float v[9];
    uint32_t meta;
};
static_assert(sizeof(RcPrim) == 40, "RcPrim is 40 bytes");

struct RcRay {  // RTRay, src/rt_transport.jl:10-19
    float ox, oy, oz, tmin, dx, dy, dz, tmax;
};
struct RcHit {  // RTHitResult, src/rt_transport.jl:33-42
    uint32_t hit;
    float t;
    uint32_t primitive_id, instance_custom_index;
    float bary_u, bary_v;
    uint32_t instance_id, pad;
};

struct float3_ {
    float x, y, z;
};
__host__ __device__ inline float3_ mk3(float x, float y, float z) { return float3_{x, y, z}; }
__host__ __device__ inline float3_ sub3(float3_ a, float3_ b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ inline float3_ add3(float3_ a, float3_ b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ inline float3_ scale3(float3_ a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
// dot = (a1b1 + a2b2) + a3b3, cross in the textbook component order
__host__ __device__ inline float dot3(float3_ a, float3_ b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__host__ __device__ inline float3_ cross3(float3_ a, float3_ b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// Base.min / Base.max of Julia on Float32 (NaN-propagating, -0 < +0); used by the build kernels where
// the result is stored (node boxes) and therefore must match bit for bit.
__host__ __device__ inline float jl_min(float a, float b) {
    if (a != a) return a;
    if (b != b) return b;
    if (a < b) return a;
    if (b < a) return b;
    return (__builtin_signbit(a)) ? a : b;
}
__host__ __device__ inline float jl_max(float a, float b) {
    if (a != a) return a;
    if (b != b) return b;
    if (a > b) return a;
    if (b > a) return b;
    return (__builtin_signbit(a)) ? b : a;
}
__host__ __device__ inline float3_ min3v(float3_ a, float3_ b) { return mk3(jl_min(a.x, b.x), jl_min(a.y, b.y), jl_min(a.z, b.z)); }
__host__ __device__ inline float3_ max3v(float3_ a, float3_ b) { return mk3(jl_max(a.x, b.x), jl_max(a.y, b.y), jl_max(a.z, b.z)); }

// The same semantics in one instruction on the device: gfx950 has v_minimum3_f32 / v_maximum3_f32 (IEEE 754-2019 minimum / maximum:
// NaN-propagating, -0 < +0), which is what Julia's min / max are; v_min_f32 / v_max_f32 (fminf / fmaxf) drop NaNs instead.
__device__ inline float jl_minf(float a, float b) { return __builtin_elementwise_minimum(a, b); }
__device__ inline float jl_maxf(float a, float b) { return __builtin_elementwise_maximum(a, b); }

// transform_point / transform_direction for Mat3x4f (src/instanced-bvh.jl:1692-1698, 1711-1717)
__host__ __device__ inline float3_ xf_point(const float* m, float3_ p) {
    return mk3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
               m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
__host__ __device__ inline float3_ xf_dir(const float* m, float3_ v) {
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
               m[8] * v.x + m[9] * v.y + m[10] * v.z);
}

// mat3x4_inverse (src/instanced-bvh.jl:1675-1687): StaticArrays' 3x3 inverse (cross-product form) of the Julia-indexed upper block,
// then the translation.  Host and device: the same expression order either way.
__host__ __device__ inline void rc_mat3x4_inverse_hd(const float m[12], float out[12]) {
    float3_ x0 = mk3(m[0], m[1], m[2]), x1 = mk3(m[4], m[5], m[6]), x2 = mk3(m[8], m[9], m[10]);
    float3_ y0 = cross3(x1, x2);
    float d = dot3(x0, y0);
    x0 = mk3(x0.x / d, x0.y / d, x0.z / d);
    y0 = mk3(y0.x / d, y0.y / d, y0.z / d);
    float3_ y1 = cross3(x2, x0), y2 = cross3(x0, x1);
    float tx = m[3], ty = m[7], tz = m[11];
    out[0] = y0.x; out[1] = y1.x; out[2] = y2.x; out[3] = -(y0.x * tx + y1.x * ty + y2.x * tz);
    out[4] = y0.y; out[5] = y1.y; out[6] = y2.y; out[7] = -(y0.y * tx + y1.y * ty + y2.y * tz);
    out[8] = y0.z; out[9] = y1.z; out[10] = y2.z; out[11] = -(y0.z * tx + y1.z * ty + y2.z * tz);
}

// safe_invdir (src/instanced-bvh.jl:1742-1748)
__host__ __device__ inline float safe_inv1(float d) {
    const float ooeps = 1.0e-5f;
    return 1.0f / (__builtin_fabsf(d) > ooeps ? d : __builtin_copysignf(ooeps, d));
}
