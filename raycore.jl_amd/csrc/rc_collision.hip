// rc_collision.hip -- instance-level broad phase on the TLAS (src/collision.jl).
//
// Reference: ContactPair :25-28, aabb_overlaps :51-53, tlas_node_aabb :56-66, collide_instances_kernel! :81-156 (one
// work-item per Morton-sorted TLAS leaf walks the TLAS with AABB-vs-AABB tests; pairs (a < b) only), collide_instances
// :189-233 (count pass, inclusive prefix sum, write pass; a leaf's pairs land back-to-front inside its range, :135).
// Same two passes here: k_collide<false> -> hipCUB InclusiveSum -> k_collide<true>; the contact array is identical to the
// reference algorithm's, including its order.  The per-thread stack is 64 entries (a Karras tree over 62-bit keys is at
// most 62 deep; the reference's unchecked 16 would overflow on skewed scenes).
#include <hipcub/hipcub.hpp>

#include "rc_internal.h"

namespace {

constexpr int kCollideStack = 64;

__device__ inline bool aabb_overlaps(float3_ a_min, float3_ a_max, float3_ b_min, float3_ b_max) {  // :51-53
    return (a_max.x >= b_min.x && a_max.y >= b_min.y && a_max.z >= b_min.z) && (a_min.x <= b_max.x && a_min.y <= b_max.y && a_min.z <= b_max.z);
}

template <bool WRITE>
__global__ void k_collide(const RcNode* nodes, uint32_t n_instances, uint32_t* contact_counts, uint2* contacts, uint32_t* status) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x + 1u;  // 1-based sorted-leaf index
    if (i > n_instances) return;
    const RcNode leaf = nodes[(n_instances - 1u + i) - 1u];
    float3_ a_min, a_max;
    if (leaf.child0 != RC_INVALID_NODE) {  // tlas_node_aabb of an interior node (:57-61); never taken for a leaf position
        a_min = min3v(mk3(leaf.f[0], leaf.f[1], leaf.f[2]), mk3(leaf.f[6], leaf.f[7], leaf.f[8]));
        a_max = max3v(mk3(leaf.f[3], leaf.f[4], leaf.f[5]), mk3(leaf.f[9], leaf.f[10], leaf.f[11]));
    } else {
        a_min = mk3(leaf.f[0], leaf.f[1], leaf.f[2]);
        a_max = mk3(leaf.f[3], leaf.f[4], leaf.f[5]);
    }
    const uint32_t instance_a = leaf.child1;
    const uint32_t end_offset = WRITE ? contact_counts[i - 1u] : 0u;  // inclusive prefix sum
    uint32_t stack[kCollideStack];
    int sp = 0;
    uint32_t node_index = 1u, count = 0u;
    for (;;) {
        const RcNode* node = nodes + (node_index - 1u);
        const float4* q = reinterpret_cast<const float4*>(node);
        const float4 qa = q[0], qb = q[1], qc = q[2];
        const uint2 ch = *reinterpret_cast<const uint2*>(q + 3);
        if (ch.x != RC_INVALID_NODE) {
            const bool overlap0 = aabb_overlaps(a_min, a_max, mk3(qa.x, qa.y, qa.z), mk3(qa.w, qb.x, qb.y));
            const bool overlap1 = aabb_overlaps(a_min, a_max, mk3(qb.z, qb.w, qc.x), mk3(qc.y, qc.z, qc.w));
            if (overlap0 && overlap1) {
                if (sp < kCollideStack) stack[sp++] = ch.y; else *status = 1u;
                node_index = ch.x;
                continue;
            } else if (overlap0) { node_index = ch.x; continue; }
            else if (overlap1) { node_index = ch.y; continue; }
        } else {
            const uint32_t instance_b = ch.y;
            if (instance_b > instance_a && aabb_overlaps(a_min, a_max, mk3(qa.x, qa.y, qa.z), mk3(qa.w, qb.x, qb.y))) {
                count += 1u;
                if (WRITE) contacts[end_offset - count] = make_uint2(instance_a + 1u, instance_b + 1u);  // write_idx = counts[i] - count + 1 (1-based)
            }
        }
        if (sp > 0) node_index = stack[--sp]; else break;
    }
    if (!WRITE) contact_counts[i - 1u] = count;
}

}  // namespace

// collide_instances (:189-233).  d_out may be nullptr (count only).  Returns the total; throws if capacity is too small.
uint64_t rc_collide_instances_launch(rc_scene* s, uint2* d_out, uint64_t capacity, hipStream_t stream) {
    const uint32_t n = s->n_static_instances;
    if (n == 0) return 0;
    s->collide_counts.reserve(n);
    RcLaunchGuard launch(s, stream);  // (the scratch below is the scene's: collision queries on one scene are one at a time anyway)
    uint32_t* status = rc_status_word(s);
    const uint32_t blocks = (n + 127) / 128;
    launch.start();
    hipLaunchKernelGGL((k_collide<false>), dim3(blocks), dim3(128), 0, stream, s->tlas_nodes.p, n, s->collide_counts.p, (uint2*)nullptr, status);
    size_t tmp = 0;
    RC_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tmp, s->collide_counts.p, s->collide_counts.p, (int)n, stream));
    s->sort_tmp.reserve(tmp);
    RC_HIP(hipcub::DeviceScan::InclusiveSum(s->sort_tmp.p, tmp, s->collide_counts.p, s->collide_counts.p, (int)n, stream));
    uint32_t total = 0;
    RC_HIP(hipMemcpyAsync(&total, s->collide_counts.p + (n - 1), 4, hipMemcpyDeviceToHost, stream));
    RC_HIP(hipStreamSynchronize(stream));  // the reference reads contact_counts[end] on the host too (:218)
    if (total != 0 && d_out != nullptr) {
        if (capacity < total) throw RcError(1, "contact buffer too small: need " + std::to_string(total) + " pairs");
        hipLaunchKernelGGL((k_collide<true>), dim3(blocks), dim3(128), 0, stream, s->tlas_nodes.p, n, s->collide_counts.p, d_out, status);
    }
    launch.finish();
    return total;
}
