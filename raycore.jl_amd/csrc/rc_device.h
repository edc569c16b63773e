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

// transform_point / transform_direction for Mat3x4f (src/instanced-bvh.jl:1692-1698, 1711-1717)
__host__ __device__ inline float3_ xf_point(const float* m, float3_ p) {
    return mk3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
               m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
__host__ __device__ inline float3_ xf_dir(const float* m, float3_ v) {
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
               m[8] * v.x + m[9] * v.y + m[10] * v.z);
}

// safe_invdir (src/instanced-bvh.jl:1742-1748)
__host__ __device__ inline float safe_inv1(float d) {
    const float ooeps = 1.0e-5f;
    return 1.0f / (__builtin_fabsf(d) > ooeps ? d : __builtin_copysignf(ooeps, d));
}
