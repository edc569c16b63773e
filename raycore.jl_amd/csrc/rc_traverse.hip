// rc_traverse.hip -- closest_hit / any_hit over ray batches on gfx950 (wave64).
//
// Replaces the per-work-item traversal the reference runs through KernelAbstractions
// (closest_hit src/instanced-bvh.jl:1902-2024, any_hit :2034-2140, safe_invdir :1742-1748,
// fast_intersect_bbox :1841-1859, intersect_internal_node :1807-1832, fast_intersect_triangle :1756-1797).
//
// Per-ray semantics are the reference's, statement for statement: same BVH2 nodes, same near/far rule,
// same push/pop order, same Moeller-Trumbore expression order, no FMA contraction -- so hit ids and t are
// bit-identical to the reference algorithm regardless of how rays are scheduled onto lanes.  What is
// MI355X-specific is everything around that: 64-byte aligned node / instance records (4 x dwordx4 per
// fetch), the per-lane traversal stack in LDS ([entry][lane] layout => conflict-free ds_read/ds_write_b32)
// with a global spill area for the rare deep path, and a persistent-wave kernel that refills finished lanes
// from a global ray counter using ballot + mbcnt prefix sums instead of waiting for the slowest ray.
#include "rc_traverse_core.h"

namespace {

using namespace rc;

// ---- kernel 0: one ray per lane, grid-stride --------------------------------------------------------
template <bool ANY>
__global__ __launch_bounds__(kBlock) void k_trace_simple(TraceArgs a) {
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
    LaneStack st{lds_stack + threadIdx.x, a.v.overflow + gtid, a.v.total_threads, a.v.status};
    for (uint64_t i = gtid; i < a.n_rays; i += a.v.total_threads) {
        RayState s;
        init_ray(s, load_ray(a.rays, i), ANY, st);
        if (a.v.n_tlas_nodes != 0)
            while (step<ANY>(s, a.v, st)) {}
        write_hit(s, a.v, a.hits, i);
    }
}

// ---- kernel 1: persistent waves with lane refill ------------------------------------------------------
// Each wave owns a slice [pool_next, pool_end) of ray indices taken from the global counter kPool at a
// time.  Lanes whose ray has finished write their hit and go idle; when at least kRefill lanes are idle
// (or every lane is) the wave hands the idle lanes consecutive new indices: the idle mask comes from
// __ballot, each idle lane's rank from mbcnt (a prefix popcount), so no lane waits for the slowest ray of
// its original 64-ray packet.
constexpr int kPool = 512;
constexpr int kRefill = 20;

template <bool ANY>
__global__ __launch_bounds__(kBlock) void k_trace_persistent(TraceArgs a) {
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
    LaneStack st{lds_stack + threadIdx.x, a.v.overflow + gtid, a.v.total_threads, a.v.status};
    const int lane = threadIdx.x & 63;
    if (a.v.n_tlas_nodes == 0) {  // empty TLAS: every ray misses (test/test_tlas_stress.jl:808-831)
        RayState miss;
        miss.closest_inst = -1;
        for (uint64_t i = gtid; i < a.n_rays; i += a.v.total_threads) write_hit(miss, a.v, a.hits, i);
        return;
    }
    unsigned long long pool_next = 0, pool_end = 0;  // wave-uniform slice of ray indices
    bool exhausted = false;                          // wave-uniform: nothing left behind the global counter
    bool active = false;
    uint64_t my_ray = 0;
    RayState s;
    s.node = RC_INVALID_NODE;
    for (;;) {
        unsigned long long idle_mask = __ballot(!active);
        int n_idle = __popcll(idle_mask);
        const bool can_refill = !(exhausted && pool_next == pool_end);
        if (n_idle == 64 && !can_refill) break;
        if (can_refill && n_idle >= kRefill) {
            for (;;) {
                idle_mask = __ballot(!active);
                n_idle = __popcll(idle_mask);
                if (n_idle == 0) break;
                if (pool_next == pool_end) {
                    if (exhausted) break;
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(a.work_counter, (unsigned long long)kPool);
                    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base);
                    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
                    base = ((unsigned long long)hi << 32) | lo;
                    if (base >= a.n_rays) { exhausted = true; break; }
                    pool_next = base;
                    pool_end = base + kPool;
                    if (pool_end >= a.n_rays) { pool_end = a.n_rays; exhausted = true; }
                }
                const unsigned long long left = pool_end - pool_next;
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(idle_mask >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((unsigned)idle_mask, 0u));
                if (!active && rank < left) {
                    my_ray = pool_next + rank;
                    init_ray(s, load_ray(a.rays, my_ray), ANY, st);
                    active = true;
                }
                pool_next += ((unsigned long long)n_idle < left) ? (unsigned long long)n_idle : left;
            }
        }
        if (active) {
            if (!step<ANY>(s, a.v, st)) {
                write_hit(s, a.v, a.hits, my_ray);
                active = false;
            }
        }
    }
}

}  // namespace

// Scratch shared by every traversal launch: the lane-stack spill area (sized for the largest persistent
// grid: n_cus x 8 blocks) and the counter / status words, zeroed on the launch stream.
void rc_prepare_launch(rc_scene* s, hipStream_t stream) {
    s->overflow_stack.reserve((size_t)(kTotalStack - kLdsStack) * (size_t)s->n_cus * 8 * kBlock);
    s->counters.reserve(16);
    RC_HIP(hipMemsetAsync(s->counters.p, 0, 16 * sizeof(uint32_t), stream));
}

rc::SceneView rc_scene_view(rc_scene* s, uint32_t total_threads) {
    rc::SceneView v;
    v.tlas_nodes = s->tlas_nodes.p; v.blas_nodes = s->flat_nodes.p; v.inst = s->inst_recs.p; v.prims = s->flat_prims.p;
    v.n_tlas_nodes = s->n_tlas_nodes; v.n_prims = s->n_flat_prims;
    v.overflow = s->overflow_stack.p; v.total_threads = total_threads;
    v.status = s->counters.p + 4;
    return v;
}

uint32_t rc_persistent_blocks(rc_scene* s, uint64_t n_items) {
    const int per_cu = s->opt.blocks_per_cu > 0 ? (int)s->opt.blocks_per_cu : 6;  // 6 x 24 KiB LDS stacks per CU
    uint64_t want = (n_items + kBlock - 1) / kBlock, cap = (uint64_t)s->n_cus * per_cu;
    return (uint32_t)(want < cap ? want : cap);
}

void rc_launch_trace(rc_scene* s, const RcRay* d_rays, RcHit* d_hits, uint64_t n, int any_hit, hipStream_t stream) {
    if (n == 0) return;
    uint32_t blocks = rc_persistent_blocks(s, n);
    uint32_t total_threads = blocks * kBlock;
    rc_prepare_launch(s, stream);
    TraceArgs a;
    a.v = rc_scene_view(s, total_threads);
    a.rays = d_rays; a.hits = d_hits; a.n_rays = n;
    a.work_counter = reinterpret_cast<unsigned long long*>(s->counters.p);
    RC_HIP(hipEventRecord(s->ev0, stream));
    if (s->opt.kernel == 0) {
        if (any_hit) hipLaunchKernelGGL(k_trace_simple<true>, dim3(blocks), dim3(kBlock), 0, stream, a);
        else hipLaunchKernelGGL(k_trace_simple<false>, dim3(blocks), dim3(kBlock), 0, stream, a);
    } else {
        if (any_hit) hipLaunchKernelGGL(k_trace_persistent<true>, dim3(blocks), dim3(kBlock), 0, stream, a);
        else hipLaunchKernelGGL(k_trace_persistent<false>, dim3(blocks), dim3(kBlock), 0, stream, a);
    }
    RC_HIP(hipEventRecord(s->ev1, stream));
    RC_HIP(hipGetLastError());
}
