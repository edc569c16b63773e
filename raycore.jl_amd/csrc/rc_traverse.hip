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
template <bool ANY, int LDS_N, int MINW>
__global__ __launch_bounds__(kBlock, MINW) void k_trace_simple(TraceArgs a) {
    __shared__ uint32_t lds_stack[LDS_N * kBlock];
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
    LaneStackT<LDS_N> st{lds_stack + threadIdx.x, a.v.overflow + gtid, a.v.total_threads, a.v.status};
    for (uint64_t i = gtid; i < a.n_rays; i += a.v.total_threads) {
        RayState s;
        init_ray(s, load_ray(a.rays, i), ANY, st);
        if (a.v.n_tlas_nodes != 0)
            while (step<ANY>(s, a.v, st)) {}
        write_hit(s, a.v, a.hits, i);
    }
}

// ---- kernel 1: persistent waves with lane refill ------------------------------------------------------
// Each wave owns a slice [pool_next, pool_end) of ray indices taken from the global counter a.pool at a
// time (64..128, sized so that every resident wave gets several slices even for small batches).  Lanes whose ray has finished write their hit and go idle; when at least `refill` lanes are idle
// (or every lane is) the wave hands the idle lanes consecutive new indices: the idle mask comes from
// __ballot, each idle lane's rank from mbcnt (a prefix popcount), so no lane waits for the slowest ray of
// its original 64-ray packet.

template <bool ANY, int LDS_N, int MINW, bool STATS>
__global__ __launch_bounds__(kBlock, MINW) void k_trace_persistent(TraceArgs a) {
    __shared__ uint32_t lds_stack[LDS_N * kBlock];
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
    LaneStackT<LDS_N> st{lds_stack + threadIdx.x, a.v.overflow + gtid, a.v.total_threads, a.v.status};
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
    unsigned long long st_steps = 0, st_lanes = 0, st_maxsp = 0;
    for (;;) {
        unsigned long long idle_mask = __ballot(!active);
        int n_idle = __popcll(idle_mask);
        const bool can_refill = !(exhausted && pool_next == pool_end);
        if (n_idle == 64 && !can_refill) break;
        if (can_refill && n_idle >= a.refill) {
            for (;;) {
                idle_mask = __ballot(!active);
                n_idle = __popcll(idle_mask);
                if (n_idle == 0) break;
                if (pool_next == pool_end) {
                    if (exhausted) break;
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(a.work_counter, (unsigned long long)a.pool);
                    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base);
                    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
                    base = ((unsigned long long)hi << 32) | lo;
                    if (base >= a.n_rays) { exhausted = true; break; }
                    pool_next = base;
                    pool_end = base + a.pool;
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
        if (STATS) { st_steps += 1; st_lanes += active ? 1 : 0; if ((unsigned)s.sp > st_maxsp && active) st_maxsp = s.sp; }
        if (active) {
            if (!step<ANY>(s, a.v, st)) {
                write_hit(s, a.v, a.hits, my_ray);
                active = false;
            }
        }
    }
    if (STATS) {
        if (lane == 0) atomicAdd(&a.stats[0], st_steps);
        atomicAdd(&a.stats[1], st_lanes);
        atomicMax(&a.stats[2], st_maxsp);
    }
}

// ---- kernel 2: persistent waves + per-wave path scheduling --------------------------------------------
// Kernel 1 runs the reference's three-way loop body as written, so a wave executes the interior-node,
// triangle and instance-entry blocks whenever ANY of its lanes needs them -- measured (round 1, C3): ~100
// VALU instructions per wave step where a pure interior step needs ~45, i.e. the vector ALU (the actual
// limiter; the tree is cache resident) spends most of its issue slots on masked-off lanes.
//
// Here every lane carries the KIND of its next visit, known before the fetch because the node numbering
// encodes it (internal nodes 1..n-1, leaves n..2n-1, src/instanced-bvh.jl:1293-1295): interior (TLAS or
// BLAS), BLAS leaf (triangle), TLAS leaf (instance entry).  Per iteration the wave ballots the three
// kinds and runs ONLY the block most lanes are waiting for; the other lanes simply hold their state for
// a later iteration.  A lane's own sequence of visits, box tests, pushes and pops is unchanged -- only WHEN
// it happens relative to other lanes moves -- so results stay bit-identical to the reference order.
// Leaf blocks fetch just what they use (36 B of vertices; the primitive index is idx - n + 1).
enum : int { K_IDLE = 0, K_INTERIOR = 1, K_LEAF = 2, K_ENTRY = 3 };

struct SchedState {
    RayState r;
    uint32_t n_level;  // leaf threshold of the current level: n_instances (TLAS) or the BLAS's n_prims
    int kind;
};

__device__ inline int classify(const SchedState& s) {
    if (s.r.node == RC_INVALID_NODE) return K_IDLE;
    const bool leaf = s.r.node >= s.n_level;
    return leaf ? (s.r.cur_inst < 0 ? K_ENTRY : K_LEAF) : K_INTERIOR;
}

// pop (:1991-2006) incl. the return to the top level
template <class Stack>
__device__ inline void pop_next(SchedState& s, Stack& st, uint32_t n_instances) {
    s.r.node = st.pop(s.r.sp);
    if (s.r.node == RC_TOP_LEVEL_SENTINEL) {
        s.r.node = st.pop(s.r.sp);
        s.r.cur_inst = -1;
        s.r.o = s.r.wo; s.r.d = s.r.wd; s.r.inv = s.r.winv;
        s.r.ox = mk3(-s.r.o.x * s.r.inv.x, -s.r.o.y * s.r.inv.y, -s.r.o.z * s.r.inv.z);
        s.n_level = n_instances;
    }
}

template <bool ANY, int LDS_N, int MINW, bool STATS>
__global__ __launch_bounds__(kBlock, MINW) void k_trace_sched(TraceArgs a) {
    __shared__ uint32_t lds_stack[LDS_N * kBlock];
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
    LaneStackT<LDS_N> st{lds_stack + threadIdx.x, a.v.overflow + gtid, a.v.total_threads, a.v.status};
    const int lane = threadIdx.x & 63;
    if (a.v.n_tlas_nodes == 0) {
        RayState miss;
        miss.closest_inst = -1;
        for (uint64_t i = gtid; i < a.n_rays; i += a.v.total_threads) write_hit(miss, a.v, a.hits, i);
        return;
    }
    const uint32_t n_instances = (a.v.n_tlas_nodes + 1u) >> 1;
    unsigned long long pool_next = 0, pool_end = 0;
    bool exhausted = false;
    uint64_t my_ray = 0;
    SchedState s;
    s.r.node = RC_INVALID_NODE;
    s.kind = K_IDLE;
    unsigned long long st_iter[4] = {0, 0, 0, 0}, st_lane[4] = {0, 0, 0, 0};
    for (;;) {
        const unsigned long long m_int = __ballot(s.kind == K_INTERIOR), m_leaf = __ballot(s.kind == K_LEAF),
                                 m_ent = __ballot(s.kind == K_ENTRY);
        const int n_int = __popcll(m_int), n_leaf = __popcll(m_leaf), n_ent = __popcll(m_ent);
        int n_idle = 64 - n_int - n_leaf - n_ent;
        const bool can_refill = !(exhausted && pool_next == pool_end);
        if (n_idle == 64 && !can_refill) break;
        if (can_refill && n_idle >= a.refill) {
            for (;;) {
                unsigned long long idle_mask = __ballot(s.kind == K_IDLE);
                n_idle = __popcll(idle_mask);
                if (n_idle == 0) break;
                if (pool_next == pool_end) {
                    if (exhausted) break;
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(a.work_counter, (unsigned long long)a.pool);
                    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base);
                    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
                    base = ((unsigned long long)hi << 32) | lo;
                    if (base >= a.n_rays) { exhausted = true; break; }
                    pool_next = base;
                    pool_end = base + a.pool;
                    if (pool_end >= a.n_rays) { pool_end = a.n_rays; exhausted = true; }
                }
                const unsigned long long left = pool_end - pool_next;
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(idle_mask >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((unsigned)idle_mask, 0u));
                if (s.kind == K_IDLE && rank < left) {
                    my_ray = pool_next + rank;
                    init_ray(s.r, load_ray(a.rays, my_ray), ANY, st);
                    s.n_level = n_instances;
                    s.kind = classify(s);
                }
                pool_next += ((unsigned long long)n_idle < left) ? (unsigned long long)n_idle : left;
            }
            continue;
        }
        // pick a block: a leaf / entry batch runs once `thr` lanes wait for it (or nothing else can run);
        // otherwise the wave keeps walking interior nodes.  Waiting lanes cost no issue slots.
        int path;
        if (n_leaf >= a.sched_thr && n_leaf >= n_ent) path = K_LEAF;
        else if (n_ent >= a.sched_thr) path = K_ENTRY;
        else if (n_int > 0) path = K_INTERIOR;
        else path = (n_leaf >= n_ent) ? K_LEAF : K_ENTRY;
        if (STATS) { st_iter[path] += 1; st_lane[path] += (s.kind == path) ? 1 : 0; st_lane[0] += (s.kind != K_IDLE) ? 1 : 0; st_iter[0] += 1; }
        if (path == K_INTERIOR) {
            if (s.kind == K_INTERIOR) {
                const RcNode* np = (s.r.cur_inst < 0) ? (a.v.tlas_nodes + (s.r.node - 1)) : (a.v.blas_nodes + (s.r.blas_off + s.r.node - 1));
                const float4* q = reinterpret_cast<const float4*>(np);
                const float4 na = q[0], nb = q[1], nc = q[2];
                const uint2 ch = *reinterpret_cast<const uint2*>(q + 3);
                float t0_min, t0_max, t1_min, t1_max;
                slab(s.r, na.x, na.y, na.z, na.w, nb.x, nb.y, t0_min, t0_max);
                slab(s.r, nb.z, nb.w, nc.x, nc.y, nc.z, nc.w, t1_min, t1_max);
                const uint32_t trav0 = (t0_min <= t0_max) ? ch.x : RC_INVALID_NODE;
                const uint32_t trav1 = (t1_min <= t1_max) ? ch.y : RC_INVALID_NODE;
                const bool first0 = (t0_min < t1_min) && (trav0 != RC_INVALID_NODE);
                const uint32_t near_c = first0 ? trav0 : trav1, far_c = first0 ? trav1 : trav0;
                if (far_c != RC_INVALID_NODE) st.push(s.r.sp, far_c);
                if (near_c != RC_INVALID_NODE) s.r.node = near_c;
                else pop_next(s, st, n_instances);
                s.kind = classify(s);
                if (s.kind == K_IDLE) write_hit(s.r, a.v, a.hits, my_ray);
            }
        } else if (path == K_LEAF) {
            if (s.kind == K_LEAF) {
                const RcNode* np = a.v.blas_nodes + (s.r.blas_off + s.r.node - 1);
                const float4* q = reinterpret_cast<const float4*>(np);
                const float4 na = q[0], nb = q[1];
                const float v2z = np->f[8];
                float3_ v0 = mk3(na.x, na.y, na.z), v1 = mk3(na.w, nb.x, nb.y), v2 = mk3(nb.z, nb.w, v2z);
                float3_ e1 = sub3(v1, v0), e2 = sub3(v2, v0);
                float3_ s1 = cross3(s.r.d, e2);
                float det = dot3(s1, e1);
                float invd = 1.0f / det;
                float3_ dd = sub3(s.r.o, v0);
                float u = dot3(dd, s1) * invd;
                float3_ s2 = cross3(dd, e1);
                float v = dot3(s.r.d, s2) * invd;
                float t = dot3(e2, s2) * invd;
                const bool hit = !(u < 0.0f || u > 1.0f) && !(v < 0.0f || (u + v) > 1.0f) && !(t < s.r.tmin || t > s.r.closest_t);
                bool done = false;
                if (hit) {
                    s.r.closest_t = t;
                    s.r.cull_t = (t != t) ? -INFINITY : t;
                    s.r.closest_inst = s.r.cur_inst;
                    s.r.closest_prim = s.r.node - s.n_level + 1u;  // leaf of sorted primitive j sits at n-1+j
                    s.r.hit_u = u; s.r.hit_v = v;
                    done = ANY;
                }
                if (done) s.r.node = RC_INVALID_NODE;
                else pop_next(s, st, n_instances);
                s.kind = classify(s);
                if (s.kind == K_IDLE) write_hit(s.r, a.v, a.hits, my_ray);
            }
        } else {
            if (s.kind == K_ENTRY) {
                s.r.cur_inst = (int)(a.v.tlas_nodes + (s.r.node - 1))->child1;
                st.push(s.r.sp, RC_TOP_LEVEL_SENTINEL);
                s.r.node = 1;
                const float4* q = reinterpret_cast<const float4*>(a.v.inst + s.r.cur_inst);
                const float4 m0 = q[0], m1 = q[1], m2 = q[2];
                const uint4 m3 = *reinterpret_cast<const uint4*>(q + 3);
                s.r.blas_off = m3.x;
                s.n_level = m3.w;
                const float3_ wo = s.r.wo, wd = s.r.wd;
                s.r.o = mk3(m0.x * wo.x + m0.y * wo.y + m0.z * wo.z + m0.w, m1.x * wo.x + m1.y * wo.y + m1.z * wo.z + m1.w,
                            m2.x * wo.x + m2.y * wo.y + m2.z * wo.z + m2.w);
                s.r.d = mk3(m0.x * wd.x + m0.y * wd.y + m0.z * wd.z, m1.x * wd.x + m1.y * wd.y + m1.z * wd.z,
                            m2.x * wd.x + m2.y * wd.y + m2.z * wd.z);
                s.r.inv = mk3(safe_inv1(s.r.d.x), safe_inv1(s.r.d.y), safe_inv1(s.r.d.z));
                s.r.ox = mk3(-s.r.o.x * s.r.inv.x, -s.r.o.y * s.r.inv.y, -s.r.o.z * s.r.inv.z);
                s.kind = classify(s);
            }
        }
    }
    if (STATS) {
        for (int k = 0; k < 4; ++k) {
            if (lane == 0) atomicAdd(&a.stats[2 * k], st_iter[k]);
            atomicAdd(&a.stats[2 * k + 1], st_lane[k]);
        }
    }
}

}  // namespace

// Scratch shared by every traversal launch: the lane-stack spill area (sized for the largest persistent
// grid: n_cus x 8 blocks) and the counter / status words, zeroed on the launch stream.
void rc_prepare_launch(rc_scene* s, hipStream_t stream) {
    s->overflow_stack.reserve((size_t)kTotalStack * (size_t)s->n_cus * 8 * kBlock);
    s->counters.reserve(64);
    RC_HIP(hipMemsetAsync(s->counters.p, 0, 64 * sizeof(uint32_t), stream));
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

template <bool ANY>
static void launch_variant(rc_scene* s, const TraceArgs& a, uint32_t blocks, hipStream_t stream) {
    const int64_t lds = s->opt.lds_stack;
    const bool stats = s->opt.stats != 0;
#define RC_LAUNCH_P(L, W) hipLaunchKernelGGL((k_trace_persistent<ANY, L, W, false>), dim3(blocks), dim3(kBlock), 0, stream, a)
#define RC_LAUNCH_S(L, W) hipLaunchKernelGGL((k_trace_simple<ANY, L, W>), dim3(blocks), dim3(kBlock), 0, stream, a)
    if (s->opt.kernel == 2) {
        if (stats) hipLaunchKernelGGL((k_trace_sched<ANY, 24, 6, true>), dim3(blocks), dim3(kBlock), 0, stream, a);
        else if (lds == 16) hipLaunchKernelGGL((k_trace_sched<ANY, 16, 8, false>), dim3(blocks), dim3(kBlock), 0, stream, a);
        else hipLaunchKernelGGL((k_trace_sched<ANY, 24, 6, false>), dim3(blocks), dim3(kBlock), 0, stream, a);
    } else if (s->opt.kernel == 0) {
        if (lds == 12) RC_LAUNCH_S(12, 8); else if (lds == 16) RC_LAUNCH_S(16, 8); else if (lds == 32) RC_LAUNCH_S(32, 4); else RC_LAUNCH_S(24, 6);
    } else if (stats) {
        hipLaunchKernelGGL((k_trace_persistent<ANY, 24, 6, true>), dim3(blocks), dim3(kBlock), 0, stream, a);
    } else {
        if (lds == 12) RC_LAUNCH_P(12, 8); else if (lds == 16) RC_LAUNCH_P(16, 8); else if (lds == 32) RC_LAUNCH_P(32, 4); else RC_LAUNCH_P(24, 6);
    }
#undef RC_LAUNCH_P
#undef RC_LAUNCH_S
}

uint32_t rc_blocks_per_cu(rc_scene* s) {
    if (s->opt.blocks_per_cu > 0) return (uint32_t)s->opt.blocks_per_cu;
    switch (s->opt.lds_stack) { case 12: case 16: return 8; case 32: return 4; default: return 6; }
}

void rc_launch_trace(rc_scene* s, const RcRay* d_rays, RcHit* d_hits, uint64_t n, int any_hit, hipStream_t stream) {
    if (n == 0) return;
    uint64_t want = (n + kBlock - 1) / kBlock, cap = (uint64_t)s->n_cus * rc_blocks_per_cu(s);
    uint32_t blocks = (uint32_t)(want < cap ? want : cap);
    uint32_t total_threads = blocks * kBlock;
    rc_prepare_launch(s, stream);
    TraceArgs a;
    a.v = rc_scene_view(s, total_threads);
    a.rays = d_rays; a.hits = d_hits; a.n_rays = n;
    a.work_counter = reinterpret_cast<unsigned long long*>(s->counters.p);
    a.refill = (int)s->opt.refill;
    {
        uint64_t per = n / ((uint64_t)(total_threads / 64) * 4);
        per = (per / 64) * 64;
        a.pool = (uint32_t)(s->opt.pool > 0 ? s->opt.pool : (per < 64 ? 64 : (per > 128 ? 128 : per)));
    }
    a.sched_thr = (int)s->opt.sched_thr;
    a.stats = reinterpret_cast<unsigned long long*>(s->counters.p + 8);
    RC_HIP(hipEventRecord(s->ev0, stream));
    const int64_t saved_kernel = s->opt.kernel;
    if (saved_kernel < 0) s->opt.kernel = (n < (uint64_t)total_threads * 4) ? 0 : 1;  // auto: tiny batches gain nothing from refilling
    if (any_hit) launch_variant<true>(s, a, blocks, stream); else launch_variant<false>(s, a, blocks, stream);
    s->opt.kernel = saved_kernel;
    RC_HIP(hipEventRecord(s->ev1, stream));
    RC_HIP(hipGetLastError());
}
