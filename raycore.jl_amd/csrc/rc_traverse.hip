// rc_traverse.hip -- closest_hit / any_hit over ray batches on gfx950 (wave64).
//
// Replaces the per-work-item traversal the reference runs through KernelAbstractions
// (closest_hit src/instanced-bvh.jl:1902-2024, any_hit :2034-2140, safe_invdir :1742-1748,
// fast_intersect_bbox :1841-1859, intersect_internal_node :1807-1832, fast_intersect_triangle :1756-1797).
//
// Per-ray semantics are the reference's, statement for statement: same BVH2 nodes, same near/far rule,
// same push/pop order, same Moeller-Trumbore expression order, no FMA contraction -- so hit ids and t are
// bit-identical to the reference algorithm regardless of how rays are scheduled onto lanes.  What is
// MI355X-specific is everything around that: 64-byte aligned, pair-packed node / instance records fetched with
// raw buffer loads, the per-lane traversal stack in LDS ([entry][lane] layout => conflict-free ds_read/ds_write_b32)
// with a global spill area for the rare deep path, persistent waves that refill finished lanes from a global ray
// counter (ballot + mbcnt prefix sums), and phase-structured execution so a wave only issues the block its lanes
// actually need.  Kernel variants (rc_set_option "kernel"): 0 simple, 1 persistent, 2 voted scheduling, 3 phased
// (core in rc_traverse_core.h), 4 / 5 phased with the top level -- TLAS interior nodes, instance records, a single BLAS's top nodes --
// staged in LDS (5 = default when the scene has <= 256 instances).  All return identical results.
#include <algorithm>
#include <cstdlib>

#include <hip/hip_ext.h>
#include <hipcub/hipcub.hpp>

#include "rc_traverse_core.h"

namespace {

using namespace rc;

// ---- kernel 0: one ray per lane, grid-stride --------------------------------------------------------
template <bool ANY, int LDS_N, int MINW>
__global__ __launch_bounds__(kBlock, MINW) void k_trace_simple(TraceArgs a) {
    __shared__ uint32_t lds_stack[LDS_N * kBlock];
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
    LaneStackT<LDS_N> st(lds_stack + threadIdx.x, a.v.overflow + gtid, a.v.total_threads, a.v.status);
    for (uint64_t i = gtid; i < a.n_rays; i += a.v.total_threads) {
        RayState s;
        init_ray(s, load_ray(a.rays, i), ANY, st, a.v.tlas_off);
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
    LaneStackT<LDS_N> st(lds_stack + threadIdx.x, a.v.overflow + gtid, a.v.total_threads, a.v.status);
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
                    // one chunk of rays from this wave's shard of the interleaved chunk counters (RcClaim, rc_traverse_core.h)
                    if (!rc_claim_chunk(a.claim, nullptr, (blockIdx.x * kBlock + threadIdx.x) >> 6, lane, a.n_rays, pool_next, pool_end)) { exhausted = true; break; }
                }
                const unsigned long long left = pool_end - pool_next;
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(idle_mask >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((unsigned)idle_mask, 0u));
                if (!active && rank < left) {
                    my_ray = pool_next + rank;
                    init_ray(s, load_ray(a.rays, my_ray), ANY, st, a.v.tlas_off);
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

// ---- kernels 4 / 5: kernel 3 with the top level staged in LDS (LdsTop, rc_traverse_core.h) -------------------------------
// kernel 4 = <1024, 24>: one 1024-thread workgroup per CU (16 waves), 96 KiB of lane stacks; kernel 5 = <768, 16>: two workgroups per
// CU keep the 24 waves per CU of kernel 3 (the shallower LDS stack costs < 2 %, measured).  Used when the scene has <= 256 instances.
constexpr int kBigBlock = 1024;
constexpr size_t kBigStackBytes = (size_t)kLdsStack * kBigBlock * 4;
constexpr size_t kBigLdsBytes = kBigStackBytes + kLdsTopBytes;

template <bool ANY, int BLOCK, int LDS_N, int MINW, bool TIMELINE = false, bool STATS = false, bool STACK16 = false>
__global__ __launch_bounds__(BLOCK, MINW) void k_trace_phased_lds(TraceArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename std::conditional<STACK16, uint16_t, uint32_t>::type entry_t;
    constexpr size_t stack_bytes = (size_t)LDS_N * BLOCK * sizeof(entry_t);
    constexpr int kPlanes = STACK16 ? kLdsPlaneNodes16 : kLdsPlaneNodes;
    entry_t* lds_stack = reinterpret_cast<entry_t*>(smem);
    const LdsTop top(smem + stack_bytes, (size_t)7 * kPlanes * sizeof(float2));
    if (a.v.n_tlas_nodes) stage_lds_top<BLOCK, kPlanes>(top, a.v, a.blas_k, a.lds_blas_base);
    __syncthreads();
    PersistArgs p{a.n_rays, a.claim, a.refill, a.sched_thr, a.stats, a.blas_k, a.lds_blas_base, 0u, a.timeline};
    phased_trace<ANY, LDS_N, STATS, ArraySource, HitWriter, BLOCK, true, true, false, TIMELINE, STACK16>(a.v, p, lds_stack, ArraySource{a.rays}, HitWriter{a.v.inst, a.hits}, top);
}

// ---- kernel 6: kernel 5's shape for top levels that do not fit (more than 256 instances): only the breadth-first tops of the TLAS and
// of a single BLAS are staged (PARTIAL_LDS, rc_traverse_core.h); TLAS leaves and instance records come from memory as in kernel 3.
template <bool ANY, bool STACK16 = false>
__global__ __launch_bounds__(kMidBlock, 6) void k_trace_phased_partial(TraceArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename std::conditional<STACK16, uint16_t, uint32_t>::type entry_t;
    entry_t* lds_stack = reinterpret_cast<entry_t*>(smem);
    LdsTop top;
    top.tl = reinterpret_cast<float2*>(smem + (size_t)kMidStack * kMidBlock * sizeof(entry_t));
    stage_partial_top<kMidBlock, STACK16 ? kPartialPlaneNodes16 : kPartialPlaneNodes>(top.tl, a.v, a.tlas_k, a.blas_k, a.lds_blas_base);
    __syncthreads();
    PersistArgs p{a.n_rays, a.claim, a.refill, a.sched_thr, a.stats, a.blas_k, a.lds_blas_base, a.tlas_k};
    phased_trace<ANY, kMidStack, false, ArraySource, HitWriter, kMidBlock, false, false, true, false, STACK16>(a.v, p, lds_stack, ArraySource{a.rays}, HitWriter{a.v.inst, a.hits}, top);
}

// ---- kernel 2: persistent waves + per-wave path scheduling --------------------------------------------
// Kernel 1 runs the reference's three-way loop body as written, so a wave executes the interior-node,
// triangle, instance-entry and hit-write blocks whenever ANY of its lanes needs them.  Measured on C3
// (round 1): ~95 % of wave steps run the instance-entry block for 1-3 lanes -- half of all vector-memory
// instructions and ~45 % of the VALU issue slots go to masked-off lanes, and both the texture path
// (TA/TD ~75-85 % busy) and the VALU (~85 %) are what bound the kernel (the tree is cache resident).
//
// Here every lane carries the KIND of its next action, known before the fetch because the node numbering
// encodes it (internal nodes 1..n-1, leaves n..2n-1, src/instanced-bvh.jl:1293-1295):
//   INTERIOR  box tests of a TLAS/BLAS interior node          LEAF    Moeller-Trumbore on a BLAS leaf
//   SWITCH    enter an instance (TLAS leaf) or return to the top level (sentinel popped)
//   DONE      traversal finished, result not yet written      EMPTY   no ray
// Per iteration the wave ballots the kinds and runs ONE block: a LEAF / SWITCH batch once `sched_thr` lanes
// wait for it, otherwise the interior block; DONE lanes are written out and refilled together.  Waiting
// lanes cost no issue slots.  A lane's own sequence of visits, box tests, pushes and pops is exactly the
// reference's -- only WHEN it happens relative to other lanes moves -- so results stay bit-identical.
// TLAS and BLAS nodes live in one array (TLAS appended) so a lane needs a single offset, and the interior
// and leaf blocks never touch the ray registers (only SWITCH does), which keeps them free of copies.
enum : int { K_EMPTY = 0, K_INTERIOR = 1, K_LEAF = 2, K_SWITCH = 3, K_DONE = 4 };

template <bool ANY, int LDS_N, int MINW, bool STATS>
__global__ __launch_bounds__(kBlock, MINW) void k_trace_sched(TraceArgs a) {
    __shared__ uint32_t lds_stack[LDS_N * kBlock];
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
    LaneStackT<LDS_N> st(lds_stack + threadIdx.x, a.v.overflow + gtid, a.v.total_threads, a.v.status);
    const int lane = threadIdx.x & 63;
    if (a.v.n_tlas_nodes == 0) {
        RayState miss;
        miss.closest_inst = -1;
        for (uint64_t i = gtid; i < a.n_rays; i += a.v.total_threads) write_hit(miss, a.v, a.hits, i);
        return;
    }
    const uint32_t n_instances = (a.v.n_tlas_nodes + 1u) >> 1;
    const uint32_t tlas_off = a.v.tlas_off;          // TLAS nodes sit behind the BLAS nodes in a.v.blas_nodes
    const RcNode* const nodes = a.v.blas_nodes;
    unsigned long long pool_next = 0, pool_end = 0;
    bool exhausted = false;
    uint64_t my_ray = 0;
    // per-lane ray state (see RayState); kept in scalars so each block touches only what it owns
    float3_ wo = mk3(0, 0, 0), wd = mk3(0, 0, 0);
    float3_ o = mk3(0, 0, 0), d = mk3(0, 0, 0), inv = mk3(0, 0, 0), ox = mk3(0, 0, 0);
    float tmin = 0.f, closest_t = 0.f, hit_u = 0.f, hit_v = 0.f;
    uint32_t closest_prim = RC_INVALID_NODE, node = RC_INVALID_NODE, cur_off = 0, n_level = 0;
    int closest_inst = -1, cur_inst = -1, sp = 0, kind = K_EMPTY;
    unsigned long long st_iter[4] = {0, 0, 0, 0}, st_lane[4] = {0, 0, 0, 0};

    auto classify = [&](uint32_t nd) -> int {
        if (nd == RC_INVALID_NODE) return K_DONE;
        if (nd == RC_TOP_LEVEL_SENTINEL) return K_SWITCH;
        if (nd < n_level) return K_INTERIOR;
        return cur_inst < 0 ? K_SWITCH : K_LEAF;
    };

    for (;;) {
        const unsigned long long m_int = __ballot(kind == K_INTERIOR), m_leaf = __ballot(kind == K_LEAF),
                                 m_sw = __ballot(kind == K_SWITCH), m_done = __ballot(kind == K_DONE);
        const int n_int = __popcll(m_int), n_leaf = __popcll(m_leaf), n_sw = __popcll(m_sw), n_done = __popcll(m_done);
        const int n_live = n_int + n_leaf + n_sw;
        const bool can_refill = !(exhausted && pool_next == pool_end);
        if (n_live == 0 && n_done == 0 && !can_refill) break;
        if ((can_refill && 64 - n_live >= a.refill) || n_live == 0) {
            // write out finished lanes together, then hand the free lanes new rays
            if (kind == K_DONE) {
                uint4 w0, w1;
                if (closest_inst >= 0) {  // :2010-2017
                    const uint4 m3 = *(reinterpret_cast<const uint4*>(a.v.inst + closest_inst) + 3);
                    w0 = make_uint4(1u, __float_as_uint(closest_t), m3.y + closest_prim - 1u, m3.z);
                    w1 = make_uint4(__float_as_uint(hit_u), __float_as_uint(hit_v), (uint32_t)closest_inst, 0u);
                } else {  // :2018-2023
                    w0 = make_uint4(0u, 0u, RC_INVALID_NODE, 0u);
                    w1 = make_uint4(0u, 0u, RC_INVALID_NODE, 0u);
                }
                uint4* out = reinterpret_cast<uint4*>(a.hits + my_ray);
                out[0] = w0;
                out[1] = w1;
                kind = K_EMPTY;
            }
            for (;;) {
                const unsigned long long free_mask = __ballot(kind == K_EMPTY);
                const int n_free = __popcll(free_mask);
                if (n_free == 0) break;
                if (pool_next == pool_end) {
                    if (exhausted) break;
                    // one chunk of rays from this wave's shard of the interleaved chunk counters (RcClaim, rc_traverse_core.h)
                    if (!rc_claim_chunk(a.claim, nullptr, (blockIdx.x * kBlock + threadIdx.x) >> 6, lane, a.n_rays, pool_next, pool_end)) { exhausted = true; break; }
                }
                const unsigned long long left = pool_end - pool_next;
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(free_mask >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((unsigned)free_mask, 0u));
                if (kind == K_EMPTY && rank < left) {
                    my_ray = pool_next + rank;
                    const RcRay r = load_ray(a.rays, my_ray);
                    // init (:1904-1927); check_direction (src/ray.jl:39-49)
                    wo = mk3(r.ox, r.oy, r.oz);
                    wd = mk3(r.dx == 0.0f ? 0.0f : r.dx, r.dy == 0.0f ? 0.0f : r.dy, r.dz == 0.0f ? 0.0f : r.dz);
                    o = wo; d = wd;
                    inv = mk3(safe_inv1(d.x), safe_inv1(d.y), safe_inv1(d.z));
                    ox = mk3(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
                    tmin = ANY ? 0.0f : r.tmin;
                    closest_t = r.tmax;
                    hit_u = hit_v = 0.0f;
                    closest_prim = RC_INVALID_NODE;
                    closest_inst = -1; cur_inst = -1;
                    cur_off = tlas_off; n_level = n_instances;
                    sp = 0;
                    st.push(sp, RC_INVALID_NODE);
                    node = 1;
                    kind = classify(node);
                }
                pool_next += ((unsigned long long)n_free < left) ? (unsigned long long)n_free : left;
            }
            continue;
        }
        int path;
        if (n_leaf >= a.sched_thr && n_leaf >= n_sw) path = K_LEAF;
        else if (n_sw >= a.sched_thr) path = K_SWITCH;
        else if (n_int > 0) path = K_INTERIOR;
        else path = (n_leaf >= n_sw) ? K_LEAF : K_SWITCH;
        if (STATS) { st_iter[path] += 1; st_lane[path] += (kind == path) ? 1 : 0; st_lane[0] += (kind >= K_INTERIOR && kind <= K_SWITCH) ? 1 : 0; st_iter[0] += 1; }

        if (path == K_INTERIOR) {
            if (kind == K_INTERIOR) {
                // intersect_internal_node (:1807-1832) + push far / descend near / pop (:1946-1960, 1991-1993)
                const float4* q = reinterpret_cast<const float4*>(nodes + (cur_off + node - 1));
                const float4 na = q[0], nb = q[1], nc = q[2];
                const uint2 ch = *reinterpret_cast<const uint2*>(q + 3);
                // packed node (rc_pack_node): na = child-0 (min.x,min.y,max.x,max.y), nb = child-1 likewise, nc = z of both
                const v2f ixy = {inv.x, inv.y}, oxy = {ox.x, ox.y}, izz = {inv.z, inv.z}, ozz = {ox.z, ox.z};
                const v2f n0xy = v2f{na.x, na.y} * ixy + oxy, f0xy = v2f{na.z, na.w} * ixy + oxy;
                const v2f n1xy = v2f{nb.x, nb.y} * ixy + oxy, f1xy = v2f{nb.z, nb.w} * ixy + oxy;
                const v2f nf0z = v2f{nc.x, nc.y} * izz + ozz, nf1z = v2f{nc.z, nc.w} * izz + ozz;
                const float f0x = f0xy.x, f0y = f0xy.y, f0z = nf0z.y, n0x = n0xy.x, n0y = n0xy.y, n0z = nf0z.x;
                const float f1x = f1xy.x, f1y = f1xy.y, f1z = nf1z.y, n1x = n1xy.x, n1y = n1xy.y, n1z = nf1z.x;
                const float t0_max = jl_minf(jl_minf(jl_minf(jl_maxf(f0x, n0x), jl_maxf(f0y, n0y)), jl_maxf(f0z, n0z)), closest_t);
                const float t0_min = jl_maxf(jl_maxf(jl_maxf(jl_minf(f0x, n0x), jl_minf(f0y, n0y)), jl_minf(f0z, n0z)), tmin);
                const float t1_max = jl_minf(jl_minf(jl_minf(jl_maxf(f1x, n1x), jl_maxf(f1y, n1y)), jl_maxf(f1z, n1z)), closest_t);
                const float t1_min = jl_maxf(jl_maxf(jl_maxf(jl_minf(f1x, n1x), jl_minf(f1y, n1y)), jl_minf(f1z, n1z)), tmin);
                const uint32_t trav0 = (t0_min <= t0_max) ? ch.x : RC_INVALID_NODE;
                const uint32_t trav1 = (t1_min <= t1_max) ? ch.y : RC_INVALID_NODE;
                const bool first0 = (t0_min < t1_min) && (trav0 != RC_INVALID_NODE);
                const uint32_t near_c = first0 ? trav0 : trav1, far_c = first0 ? trav1 : trav0;
                if (far_c != RC_INVALID_NODE) st.push(sp, far_c);
                node = (near_c != RC_INVALID_NODE) ? near_c : st.pop(sp);
                kind = classify(node);
            }
        } else if (path == K_LEAF) {
            if (kind == K_LEAF) {
                // intersect_leaf_node -> fast_intersect_triangle (:1756-1797, 1868-1881), then pop
                const RcNode* np = nodes + (cur_off + node - 1);
                const float4* q = reinterpret_cast<const float4*>(np);
                const float4 na = q[0];
                const float4 nb = q[1];
                const float4 nc = q[2];
                const float3_ v0 = mk3(na.w, na.x, na.y), e1 = mk3(nb.y, nb.z, nb.x), e2 = mk3(nc.y, nc.z, nc.x);  // rc_pack_leaf: v0 and the edges
                const float3_ s1 = cross3(d, e2);
                const float det = dot3(s1, e1);
                const float invd = 1.0f / det;
                const float3_ dd = sub3(o, v0);
                const float u = dot3(dd, s1) * invd;
                const float3_ s2 = cross3(dd, e1);
                const float v = dot3(d, s2) * invd;
                const float t = dot3(e2, s2) * invd;
                const bool hit = !(u < 0.0f || u > 1.0f) && !(v < 0.0f || (u + v) > 1.0f) && !(t < tmin || t > closest_t);
                if (hit) {
                    closest_t = t;
                    closest_inst = cur_inst;
                    closest_prim = node - n_level + 1u;  // leaf of sorted primitive j sits at n-1+j (child1 = j)
                    hit_u = u; hit_v = v;
                }
                if (ANY && hit) node = RC_INVALID_NODE;  // :2106-2115
                else node = st.pop(sp);
                kind = classify(node);
            }
        } else {
            if (kind == K_SWITCH) {
                if (node == RC_TOP_LEVEL_SENTINEL) {
                    // back to the top level (:1996-2006)
                    node = st.pop(sp);
                    cur_inst = -1;
                    cur_off = tlas_off; n_level = n_instances;
                    o = wo; d = wd;
                    inv = mk3(safe_inv1(d.x), safe_inv1(d.y), safe_inv1(d.z));
                    ox = mk3(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
                } else {
                    // top-level leaf: enter the instance (:1961-1977)
                    cur_inst = (int)(nodes + (cur_off + node - 1))->child1;
                    st.push(sp, RC_TOP_LEVEL_SENTINEL);
                    node = 1;
                    const float4* q = reinterpret_cast<const float4*>(a.v.inst + cur_inst);
                    const float4 m0 = q[0], m1 = q[1], m2 = q[2];
                    const uint4 m3 = *reinterpret_cast<const uint4*>(q + 3);
                    cur_off = m3.x;
                    n_level = m3.w;
                    o = mk3(m0.x * wo.x + m0.y * wo.y + m0.z * wo.z + m0.w, m1.x * wo.x + m1.y * wo.y + m1.z * wo.z + m1.w,
                            m2.x * wo.x + m2.y * wo.y + m2.z * wo.z + m2.w);
                    d = mk3(m0.x * wd.x + m0.y * wd.y + m0.z * wd.z, m1.x * wd.x + m1.y * wd.y + m1.z * wd.z,
                            m2.x * wd.x + m2.y * wd.y + m2.z * wd.z);
                    inv = mk3(safe_inv1(d.x), safe_inv1(d.y), safe_inv1(d.z));
                    ox = mk3(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
                }
                kind = classify(node);
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

// ---- kernel 3: the phase-structured persistent core (rc_traverse_core.h) on a ray array ------------------------
template <bool ANY, int LDS_N, int MINW, bool STATS>
__global__ __launch_bounds__(kBlock, MINW) void k_trace_phased(TraceArgs a) {
    __shared__ uint32_t lds_stack[LDS_N * kBlock];
    PersistArgs p{a.n_rays, a.claim, a.refill, a.sched_thr, a.stats, 0u, 0u, 0u};
    phased_trace<ANY, LDS_N, STATS>(a.v, p, lds_stack, ArraySource{a.rays}, HitWriter{a.v.inst, a.hits});
}

}  // namespace

// ---- launch bookkeeping (RcLaunchGuard, rc_internal.h) ------------------------------------------------------------------------------
namespace {
struct TlTiming { uint64_t uid = 0; TimingRef ref; };
thread_local TlTiming tl_timing;  // the calling thread's latest timed operation (rc_last_kernel_ms reports it while its slot has not been reused)

void note_timing(rc_scene* s, const TimingRef& ref) {
    s->last_timing = ref;
    tl_timing.uid = s->uid;
    tl_timing.ref = ref;
}
}  // namespace

RcLaunchGuard::RcLaunchGuard(rc_scene* scene, hipStream_t st) : s(scene), stream(st), lock(scene->launch_mu) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    capturing = cap == hipStreamCaptureStatusActive;
    const size_t region_words = (size_t)kTotalStack * (size_t)s->n_cus * 8 * kBlock;  // the largest grid any option can ask for: rc_set_option clamps blocks_per_cu to 8 blocks of 256 threads per CU
    s->cur_capture = -1; s->cur_region = -1; s->cur_history = -1; s->cur_scratch = -1; s->cur_overflow = nullptr;
    s->guard_live = true;
    if (capturing) {
        // A captured launch bakes its addresses into the graph and may be replayed at any time, beside eager launches and beside other
        // graphs: it gets a lane-stack spill region AND a slot of claim counters that no other launch, eager or captured, will ever use
        // (graphs captured on one stream used to share that stream's region, so two of them replayed side by side on a tree
        // deeper than the LDS stack overwrote each other's entries).  Both come from a pool of kCounterSlots - kEagerSlots capture slots per
        // scene, handed back by option "release_capture" (one) / "release_captures" (all) once the caller's graphs are gone.  The region
        // is sized for the launch's own grid and allocated where that is known (rc_scene_view) -- inside the capture: the entry points run
        // with the thread's capture-interaction mode relaxed, in which hipMalloc is legal.
        if (s->capture_slots.empty()) s->capture_slots.resize(kCounterSlots - kEagerSlots);
        int free_slot = -1;
        for (size_t i = 0; i < s->capture_slots.size() && free_slot < 0; ++i) if (!s->capture_slots[i].in_use) free_slot = (int)i;
        if (free_slot < 0)
            throw RcError(1, "this scene already holds " + std::to_string(kCounterSlots - kEagerSlots) + " captured launches (each owns a stack spill region and a counter slot): "
                             "destroy a graph and hand its launches back (option \"release_capture\" with the token \"last_capture_token\" gave, or \"release_captures\" for all), "
                             "or capture several launches' worth of rays in one launch");
        s->capture_slots[free_slot].in_use = true;
        s->cur_capture = s->last_capture = free_slot;
        s->cur_overflow = nullptr;  // (rc_scene_view)
        s->cur_slot = kEagerSlots + free_slot;
    } else {
        // the lane-stack spill area of the launch's stream (eager launches on one stream are ordered and share it)
        size_t idx = 0;
        for (; idx < s->overflow_regions.size(); ++idx)
            if (s->overflow_regions[idx].stream == stream) break;
        if (idx == s->overflow_regions.size() || s->overflow_regions[idx].buf.cap < region_words) {
            if (idx == s->overflow_regions.size()) {
                if (idx == (size_t)rc_scene::kMaxOverflowRegions) {  // one stream too many: take over the region whose last launch is done (oldest first), else wait for the oldest region's
                    size_t victim = idx;
                    for (size_t v = 0; v < idx; ++v)
                        if (s->overflow_regions[v].last.idle()) { victim = v; break; }
                    if (victim == idx) { victim = 0; s->overflow_regions[victim].last.wait(); }
                    std::rotate(s->overflow_regions.begin() + victim, s->overflow_regions.begin() + victim + 1, s->overflow_regions.end());
                    idx -= 1;
                    s->overflow_regions[idx].stream = stream;
                } else {
                    s->overflow_regions.emplace_back();
                    s->overflow_regions.back().stream = stream;
                }
            }
            s->overflow_regions[idx].buf.reserve(region_words);
        }
        s->cur_region = (int)idx;
        s->cur_overflow = s->overflow_regions[idx].buf.p;
        // the launch's slot of chunk counters; a launch that reuses an eager slot from another stream waits for the slot's previous user
        s->launch_seq += 1;
        s->cur_slot = (int)(s->launch_seq % (uint64_t)kEagerSlots);
        rc_scene::LaunchSlot& slot = s->slots[s->cur_slot];
        if (slot.recorded && slot.stream != stream) RC_HIP(hipStreamWaitEvent(stream, slot.t1, 0));
    }
    if (s->opt.stats) RC_HIP(hipMemsetAsync(rc_stats_words(s), 0, kStatsWords * sizeof(unsigned long long), stream));
}

// The launch state is the guard's: once it is gone no capture slot, region or spill pointer is current (a stale cur_capture
// used to let a later unguarded call allocate into -- or index past -- the capture slots).  Runs with launch_mu still held.
RcLaunchGuard::~RcLaunchGuard() {
    s->cur_capture = -1; s->cur_region = -1; s->cur_history = -1; s->cur_scratch = -1; s->cur_overflow = nullptr;
    s->guard_live = false;
}

static int rc_event_mode() {  // dev (tools/archive/event_probe.py): RC_EVENT_MODE = 0 product, 1 hipEventDisableSystemFence on t1 too, 2 no t0 event, 3 no events at all (single-stream runs only), 4 system fence on t0 as well (rounds 3-4)
    static const int mode = [] { const char* e = getenv("RC_EVENT_MODE"); return e ? atoi(e) : 0; }();
    return mode;
}
void RcLaunchGuard::start() {
    if (capturing) return;
    rc_scene::LaunchSlot& slot = s->slots[s->cur_slot];
    if (!slot.t0) {
        // t0 only marks a time in front of the kernel: nothing is published by it, so it skips the system-scope fence (0.5 % of a back-to-back
        // launch sequence, profiles/r05_launch_fixed_costs.txt); t1 keeps the default -- the host reads pinned words behind it
        RC_HIP(hipEventCreateWithFlags(&slot.t0, rc_event_mode() == 4 ? hipEventDefault : hipEventDisableSystemFence));
        RC_HIP(hipEventCreateWithFlags(&slot.t1, rc_event_mode() == 1 ? hipEventDisableSystemFence : hipEventDefault));
    }
    if (rc_event_mode() < 2) RC_HIP(hipEventRecord(slot.t0, stream));
}

void RcLaunchGuard::bind() {  // instead of start(): the launch's kernel carries the events itself (launch_variant)
    if (capturing) return;
    if (rc_event_mode() >= 2) { start(); return; }  // (dev modes measure the event packets)
    rc_scene::LaunchSlot& slot = s->slots[s->cur_slot];
    if (!slot.t0) {
        RC_HIP(hipEventCreateWithFlags(&slot.t0, hipEventDisableSystemFence));
        RC_HIP(hipEventCreateWithFlags(&slot.t1, hipEventDefault));
    }
    s->bound_t0 = slot.t0; s->bound_t1 = slot.t1; s->bound_used = false;
}

void RcLaunchGuard::finish() {
    RC_HIP(hipGetLastError());
    const bool bound = s->bound_used;
    s->bound_t0 = s->bound_t1 = nullptr; s->bound_used = false;
    if (capturing) return;  // a captured launch has no events of its own: the graph orders it, and its duration is the replay's business
    rc_scene::LaunchSlot& slot = s->slots[s->cur_slot];
    if (rc_event_mode() < 3 && !bound) RC_HIP(hipEventRecord(slot.t1, stream));
    // the per-stream resources this launch used are busy until here (asked by the next stream that wants to take one over)
    const bool closed = rc_event_mode() < 3;  // (they follow the launch's closing event instead of recording their own)
    auto mark = [&](RcEvent& last) { if (closed) last.follow(slot.t1); else last.record(stream); };
    if (s->cur_region >= 0) mark(s->overflow_regions[s->cur_region].last);
    if (s->cur_history >= 0) mark(s->histories[s->cur_history].last);
    if (s->cur_scratch >= 0) mark(s->totals_scratch[s->cur_scratch].last);
    slot.stream = stream;
    slot.recorded = true;
    slot.seq = ++s->timing_seq;
    note_timing(s, TimingRef{s->cur_slot, slot.seq, 0.f});
}

void rc_timing_scene_begin(rc_scene* s, hipStream_t stream) { RC_HIP(hipEventRecord(s->ev0, stream)); }
void rc_timing_scene_end(rc_scene* s, hipStream_t stream) {
    RC_HIP(hipEventRecord(s->ev1, stream));
    std::lock_guard<std::mutex> g(s->launch_mu);
    rc_scene::LaunchSlot& slot = s->slots[kSceneTimingSlot];
    slot.t0 = s->ev0; slot.t1 = s->ev1; slot.stream = stream; slot.recorded = true;
    slot.seq = ++s->timing_seq;
    note_timing(s, TimingRef{kSceneTimingSlot, slot.seq, 0.f});
}
void rc_timing_fixed(rc_scene* s, float ms) {
    std::lock_guard<std::mutex> g(s->launch_mu);
    note_timing(s, TimingRef{-1, ++s->timing_seq, ms});
}
float rc_timing_read(rc_scene* s) {
    hipEvent_t a = nullptr, b = nullptr;
    {
        std::lock_guard<std::mutex> g(s->launch_mu);
        TimingRef ref = s->last_timing;
        if (tl_timing.uid == s->uid) {
            const TimingRef& mine = tl_timing.ref;
            if (mine.slot == -1 || (mine.slot >= 0 && s->slots[mine.slot].seq == mine.seq)) ref = mine;  // else: the slot has been reused since
        }
        if (ref.slot == -2) return 0.f;
        if (ref.slot == -1) return ref.fixed_ms;
        if (s->slots[ref.slot].seq != ref.seq) return 0.f;
        a = s->slots[ref.slot].t0; b = s->slots[ref.slot].t1;
    }
    float ms = 0.f;
    RC_HIP(hipEventSynchronize(b));
    RC_HIP(hipEventElapsedTime(&ms, a, b));
    return ms;
}

void rc_claim_fill(rc_scene* s, uint64_t n_items, uint32_t total_waves, rc::RcClaim& out) {
    out.counters = rc_counter_slot(s) + kShardBase;
    uint32_t shards = (uint32_t)s->opt.claim_shards, shift = 0;
    while (shards > 1 && shards > total_waves) shards >>= 1;  // every shard needs a wave: chunks dealt to a shard nobody drains would never be traced
    while ((2u << shift) <= shards) ++shift;
    out.shard_shift = shift;
    out.pool = (uint32_t)(s->opt.pool > 0 ? s->opt.pool : 128u);  // measured: 64 loses 10-16 % (a wave's lanes end up on rays of more image regions), 256+ unbalances the tail
    out.total_waves = total_waves;
    // guided claim sizes (RcClaim): the chunks of the claim order are dealt whole, then in halves, quarters, eighths; piece k ends where the
    // items left equal taper / 8 x its part size x waves
    const uint64_t P = out.pool, n_base = (n_items + P - 1) / P;
    uint64_t chunks[4] = {n_base, 0, 0, 0}, done = 0;
    if (s->opt.taper > 0 && (P & 7u) == 0) {  // parts must tile the chunk: a pool that is not a multiple of eight is dealt whole
        for (int k = 0; k < 3; ++k) {
            const uint64_t keep = (uint64_t)s->opt.taper * (P >> k) * total_waves / 8u;          // items left for the smaller parts
            const uint64_t upto = n_items > keep ? (n_items - keep) / P : 0;                          // whole chunks handed out before that point
            chunks[k] = upto > done ? upto - done : 0;
            done += chunks[k];
        }
        chunks[3] = n_base - done;
    }
    const uint64_t n_claims = chunks[0] + 2 * chunks[1] + 4 * chunks[2] + 8 * chunks[3];
    if (n_claims >= (1ull << 31)) throw RcError(1, "ray batch too large for 32-bit chunk ids");
    out.n_chunks = (uint32_t)n_claims;
    out.g1 = (uint32_t)chunks[0]; out.g2 = (uint32_t)(chunks[0] + 2 * chunks[1]); out.g3 = (uint32_t)(chunks[0] + 2 * chunks[1] + 4 * chunks[2]);
    out.c1 = (uint32_t)chunks[0]; out.c2 = (uint32_t)(chunks[0] + chunks[1]); out.c3 = (uint32_t)(chunks[0] + chunks[1] + chunks[2]);
    out.order = nullptr; out.cost = nullptr; out.hist = nullptr;
    out.sample_rays = nullptr; out.n_sample = 0; out.inv_l2 = 0.f; out.samples = nullptr; out.host_streak = nullptr; out.init_thr = 0; out.want_record = 0; out.host_gen = 0;
    for (int k = 0; k < 8; ++k) out.sample_host[k] = 0.f;
    out.pool_shift = 0;
    while ((2u << out.pool_shift) <= out.pool) ++out.pool_shift;
}

// The scene's arrays as the kernels see them: no launch state, no side effects.  What kernels that never spill a traversal stack get (the
// wavefront stages, which run OUTSIDE RcLaunchGuard and without launch_mu: ADVICE r5 -- they used to go through the allocating view
// below and could publish a tiny capture region into another thread's captured launch).
rc::SceneView rc_scene_view_static(rc_scene* s) {
    rc::SceneView v;
    v.tlas_nodes = s->tlas_nodes.p; v.blas_nodes = s->flat_nodes.p; v.inst = s->inst_recs.p; v.prims = s->flat_prims.p;
    v.n_tlas_nodes = s->n_tlas_nodes; v.n_prims = s->n_flat_prims; v.tlas_off = s->n_flat_nodes; v.n_nodes_total = s->n_flat_nodes + s->n_tlas_nodes; v.n_inst = s->n_static_instances;
    v.inst_cull = (s->opt.entry_cull && s->inst_cull.p && s->n_static_instances) ? s->inst_cull.p : nullptr;
    v.overflow = nullptr; v.total_threads = 0;
    v.status = rc_status_word(s);
    return v;
}

// The stack spill region of the launch being prepared (inside an RcLaunchGuard, launch_mu held).  An eager launch has its stream's
// region already; a captured launch gets its own here, sized for its own grid of `total_threads` threads (every capture used
// to pin the largest grid's 268 MB) -- allocated inside the capture: the entry points run with the thread's capture-interaction mode
// relaxed, in which hipMalloc is legal.  Never returns null.
uint32_t* rc_launch_overflow(rc_scene* s, uint32_t total_threads) {
    if (!s->guard_live) throw RcError(1, "internal: rc_launch_overflow outside RcLaunchGuard");
    if (s->cur_capture >= 0 && !s->cur_overflow) {
        rc_scene::CaptureSlot& cs = s->capture_slots[s->cur_capture];
        try {
            cs.region.reserve((size_t)kTotalStack * std::max<uint32_t>(total_threads, 64u));
        } catch (const RcError&) {
            (void)hipGetLastError();
            cs.in_use = false;
            throw RcError(1, "could not allocate the stack spill region of a captured launch inside the capture (capture mode does not allow hipMalloc here): "
                             "capture with hipStreamCaptureModeRelaxed / ThreadLocal, or run the launch eagerly");
        }
        s->cur_overflow = cs.region.p;
    }
    if (!s->cur_overflow) throw RcError(1, "internal: the launch has no stack spill region");
    return s->cur_overflow;
}

// The view of a traversal launch with `total_threads` threads (inside an RcLaunchGuard).
rc::SceneView rc_scene_view(rc_scene* s, uint32_t total_threads) {
    rc::SceneView v = rc_scene_view_static(s);
    v.overflow = rc_launch_overflow(s, total_threads); v.total_threads = total_threads;
    return v;
}

// Arguments of a persistent launch with `total_threads` threads (inside an RcLaunchGuard).
rc::PersistArgs rc_persist_args(rc_scene* s, uint64_t n_items, uint32_t total_threads) {
    rc::PersistArgs p;
    p.n_items = n_items;
    rc_claim_fill(s, n_items, total_threads / 64u, p.claim);
    p.refill = (int)s->opt.refill;
    p.int_thr = (int)s->opt.sched_thr;
    p.stats = rc_stats_words(s);
    return p;
}

// The drivers' LDS variants (768-thread workgroups, node planes) apply under the same conditions as trace kernel 5.
bool rc_lds_driver_ok(rc_scene* s) {
    return s->opt.kernel != 3 && s->n_tlas_nodes > 0 && s->n_tlas_nodes <= (uint32_t)kTlasLdsNodes && (uint64_t)(s->n_flat_nodes + s->n_tlas_nodes) * 64u < (1ull << 32);
}
uint32_t rc_lds_driver_blocks(rc_scene* s, uint64_t n_items) {
    return (uint32_t)std::min<uint64_t>((n_items + kMidBlock - 1) / kMidBlock, (uint64_t)s->n_cus * 2);
}
// ... and the partial-LDS variants under the conditions of trace kernel 6: a larger top level with something to stage
bool rc_partial_driver_ok(rc_scene* s) {
    return s->opt.kernel != 3 && s->n_tlas_nodes > (uint32_t)kTlasLdsNodes && s->tlas_top_k32 + s->blas_top_k32 > 0 &&
           (uint64_t)(s->n_flat_nodes + s->n_tlas_nodes) * 64u < (1ull << 32);
}
void rc_partial_driver_args(rc_scene* s, rc::PersistArgs& p) {  // (the drivers run the 32-bit shape: prefixes of the renumbered tops)
    p.tlas_k = s->tlas_top_k32; p.blas_k = s->opt.blas_top ? s->blas_top_k32 : 0; p.lds_blas_base = s->tlas_top_k32;
}
void rc_lds_driver_args(rc_scene* s, rc::PersistArgs& p) {
    if (s->opt.blas_top) { p.blas_k = s->blas_top_k32; p.lds_blas_base = (s->n_tlas_nodes + 1) / 2 - 1; }
}

uint32_t rc_persistent_blocks(rc_scene* s, uint64_t n_items) {
    const int per_cu = s->opt.blocks_per_cu > 0 ? (int)s->opt.blocks_per_cu : 6;  // 6 x 24 KiB LDS stacks per CU
    uint64_t want = (n_items + kBlock - 1) / kBlock, cap = (uint64_t)s->n_cus * per_cu;
    return (uint32_t)(want < cap ? want : cap);
}

// kernels 5 / 6 in their STACK16 shape: the scene's trees are all small enough, nobody asked for counters or a timeline (dev builds keep the 32-bit shape)
static bool rc_stack16(rc_scene* s) { return s->small_trees && s->opt.stack16 && !s->opt.timeline_ptr; }

template <bool ANY>
static void launch_variant(rc_scene* s, int64_t kernel, const TraceArgs& a, uint32_t blocks, hipStream_t stream) {
    const int64_t lds = s->opt.lds_stack;
    const bool stats = s->opt.stats != 0;
    // The launch's timing / ordering events ride on the kernel's own dispatch (hipExtLaunchKernelGGL binds them to its start and end): no
    // event packets of their own around the kernel -- they cost ~3 us each between two back-to-back launches.  (Null outside RcLaunchGuard::bind.)
    hipEvent_t e0 = s->bound_t0, e1 = s->bound_t1;
    s->bound_used = e0 != nullptr;
#define RC_LAUNCH_P(L, W) hipExtLaunchKernelGGL((k_trace_persistent<ANY, L, W, false>), dim3(blocks), dim3(kBlock), 0, stream, e0, e1, 0u, a)
#define RC_LAUNCH_S(L, W) hipExtLaunchKernelGGL((k_trace_simple<ANY, L, W>), dim3(blocks), dim3(kBlock), 0, stream, e0, e1, 0u, a)
    if (kernel == 4) {
        bool& attr_set = s->lds_attr_set[ANY ? 1 : 0];  // per scene = per device: the attribute belongs to the function on one device
        if (!attr_set) {
            RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_phased_lds<ANY, kBigBlock, kLdsStack, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBigLdsBytes));
            attr_set = true;
        }
        hipExtLaunchKernelGGL((k_trace_phased_lds<ANY, kBigBlock, kLdsStack, 4>), dim3(blocks), dim3(kBigBlock), kBigLdsBytes, stream, e0, e1, 0u, a);
    } else if (kernel == 5) {
        bool& attr_set = s->lds_attr_set[2 + (ANY ? 1 : 0)];
        if (!attr_set) {
            RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_phased_lds<ANY, kMidBlock, kMidStack, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMidLdsBytes));
            attr_set = true;
        }
        if (stats && rc_stack16(s)) {  // dev: the same kernel with per-phase pass / lane counters (option "stats"; tools/isa_mix.py weights the phases' static opcode histograms with them)
            RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_phased_lds<ANY, kMidBlock, kMidStack, 6, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMidLdsBytes16));
            hipExtLaunchKernelGGL((k_trace_phased_lds<ANY, kMidBlock, kMidStack, 6, false, true, true>), dim3(blocks), dim3(kMidBlock), kMidLdsBytes16, stream, e0, e1, 0u, a);
        } else if (stats) {
            RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_phased_lds<ANY, kMidBlock, kMidStack, 6, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMidLdsBytes));
            hipExtLaunchKernelGGL((k_trace_phased_lds<ANY, kMidBlock, kMidStack, 6, false, true>), dim3(blocks), dim3(kMidBlock), kMidLdsBytes, stream, e0, e1, 0u, a);
        } else if (a.timeline) {  // dev: the same kernel with per-wave event times written to the caller's buffer (option "timeline_ptr")
            RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_phased_lds<ANY, kMidBlock, kMidStack, 6, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMidLdsBytes));
            hipExtLaunchKernelGGL((k_trace_phased_lds<ANY, kMidBlock, kMidStack, 6, true>), dim3(blocks), dim3(kMidBlock), kMidLdsBytes, stream, e0, e1, 0u, a);
        } else if (rc_stack16(s)) {
            bool& set16 = s->lds_attr_set[12 + (ANY ? 1 : 0)];
            if (!set16) {
                RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_phased_lds<ANY, kMidBlock, kMidStack, 6, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMidLdsBytes16));
                set16 = true;
            }
            hipExtLaunchKernelGGL((k_trace_phased_lds<ANY, kMidBlock, kMidStack, 6, false, false, true>), dim3(blocks), dim3(kMidBlock), kMidLdsBytes16, stream, e0, e1, 0u, a);
        } else
        hipExtLaunchKernelGGL((k_trace_phased_lds<ANY, kMidBlock, kMidStack, 6>), dim3(blocks), dim3(kMidBlock), kMidLdsBytes, stream, e0, e1, 0u, a);
    } else if (kernel == 6 && rc_stack16(s)) {
        bool& attr_set = s->lds_attr_set[14 + (ANY ? 1 : 0)];
        if (!attr_set) {
            RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_phased_partial<ANY, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPartialLdsBytes16));
            attr_set = true;
        }
        hipExtLaunchKernelGGL((k_trace_phased_partial<ANY, true>), dim3(blocks), dim3(kMidBlock), kPartialLdsBytes16, stream, e0, e1, 0u, a);
    } else if (kernel == 6) {
        bool& attr_set = s->lds_attr_set[6 + (ANY ? 1 : 0)];
        if (!attr_set) {
            RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_phased_partial<ANY>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPartialLdsBytes));
            attr_set = true;
        }
        hipExtLaunchKernelGGL((k_trace_phased_partial<ANY>), dim3(blocks), dim3(kMidBlock), kPartialLdsBytes, stream, e0, e1, 0u, a);
    } else if (kernel == 3) {
        if (stats) hipExtLaunchKernelGGL((k_trace_phased<ANY, 24, 6, true>), dim3(blocks), dim3(kBlock), 0, stream, e0, e1, 0u, a);
        else if (lds == 16) hipExtLaunchKernelGGL((k_trace_phased<ANY, 16, 8, false>), dim3(blocks), dim3(kBlock), 0, stream, e0, e1, 0u, a);
        else if (lds == 20) hipExtLaunchKernelGGL((k_trace_phased<ANY, 20, 7, false>), dim3(blocks), dim3(kBlock), 0, stream, e0, e1, 0u, a);
        else if (lds == 17) hipExtLaunchKernelGGL((k_trace_phased<ANY, 16, 6, false>), dim3(blocks), dim3(kBlock), 0, stream, e0, e1, 0u, a);
        else if (lds == 13) hipExtLaunchKernelGGL((k_trace_phased<ANY, 12, 6, false>), dim3(blocks), dim3(kBlock), 0, stream, e0, e1, 0u, a);
        else hipExtLaunchKernelGGL((k_trace_phased<ANY, 24, 6, false>), dim3(blocks), dim3(kBlock), 0, stream, e0, e1, 0u, a);
    } else if (kernel == 2) {
        if (stats) hipExtLaunchKernelGGL((k_trace_sched<ANY, 24, 6, true>), dim3(blocks), dim3(kBlock), 0, stream, e0, e1, 0u, a);
        else if (lds == 16) hipExtLaunchKernelGGL((k_trace_sched<ANY, 16, 8, false>), dim3(blocks), dim3(kBlock), 0, stream, e0, e1, 0u, a);
        else hipExtLaunchKernelGGL((k_trace_sched<ANY, 24, 6, false>), dim3(blocks), dim3(kBlock), 0, stream, e0, e1, 0u, a);
    } else if (kernel == 0) {
        if (lds == 12) RC_LAUNCH_S(12, 8); else if (lds == 16) RC_LAUNCH_S(16, 8); else if (lds == 32) RC_LAUNCH_S(32, 4); else RC_LAUNCH_S(24, 6);
    } else if (stats) {
        hipExtLaunchKernelGGL((k_trace_persistent<ANY, 24, 6, true>), dim3(blocks), dim3(kBlock), 0, stream, e0, e1, 0u, a);
    } else {
        if (lds == 12) RC_LAUNCH_P(12, 8); else if (lds == 16) RC_LAUNCH_P(16, 8); else if (lds == 32) RC_LAUNCH_P(32, 4); else RC_LAUNCH_P(24, 6);
    }
#undef RC_LAUNCH_P
#undef RC_LAUNCH_S
}

uint32_t rc_blocks_per_cu(rc_scene* s) {
    if (s->opt.blocks_per_cu > 0) return (uint32_t)s->opt.blocks_per_cu;
    switch (s->opt.lds_stack) { case 12: case 16: return 8; case 20: return 7; case 32: return 4; default: return 6; }
}

// ---- cost-ordered claiming (RcClaim::order / cost / hist) -------------------------------------------------------------------------------
// Which batch a launch is, whether it claims through a learned order and whether it records is decided INSIDE the launch (order_select /
// order_commit, rc_traverse_core.h): nothing runs in front of a launch for that (round 5; rounds 3-4 spent three small dispatches per launch,
// ~3 % of a 4 M-ray step and ~4 % of a 1 M-ray one, profiles/r05_bench_with_extras_by_range.csv).  What is left to separate kernels is turning
// a finished RECORDING into an order:
//   k_order_count / k_order_scatter, for every batch slot whose cost array holds a recording no order has been built from yet (kHistPending):
//     the reported chunks in nine classes, linear in the lifetime of their longest ray between the reporting threshold and the top of the scale
//     the RECORDING launch worked with, longest first; then the chunks nobody reported; chunk ids ascending inside a class (a stable
//     counting sort: k_order_count tallies the classes per 1024-chunk block, k_order_scatter places every chunk into the slot's order array,
//     clears its cost for the slot's next recording and -- block 0 -- leaves the threshold and the scale of that next recording: the
//     threshold moves so that roughly 30-70 % of the chunks report).
//   The host enqueues the pair only in front of launches that MAY follow a recording (rc_cost_order_setup): the shape's first launches, the
//   launches after a batch that was not a repeat was reported, and the launch after one the host's cadence asked to record -- one launch in
//   eight of a repeating batch.  A recording the host misses (it learns of new batches through a pinned word, late when the caller enqueues
//   far ahead) simply waits for the next pair.
//   Recording costs 25-30 us of a 0.37 ms launch, so a slot records its launches 2-4 (never the first: a batch that does not come back must
//   not pay for it) and then one in eight; the launches in between claim through the slot's order as it stands.
//   (A first-launch PREDICTOR -- claim order from the number of top-of-tree boxes each chunk's middle ray passes -- was built and measured in
//   round 4: worth 5-12 % of a 1 M-ray launch, and the small kernels it needs cost as much: profiles/r04_first_launch_predictor.txt.  Not kept.)
// hist[kHistScale + 4 slot]: [0], [1] = the threshold / the top of the scale the slot's latest recording launch worked with; [2], [3] = the pair its next one will.
// dev knobs (tools/probes/build_variants.sh): the number of cost classes of the rebuilt order and the share of the chunks the reporting
// threshold aims at.  Round 5 (tools/probes/order_quality_probe.py, docs/EXPERIMENTS.md): 10 / 16 / 26 / 31 classes order equally well; the
// share matters -- with 30-70 % of the chunks reporting (rounds 3-4: 10-40 %) the same batches run 3-8 % faster, because the chunks below
// the threshold are claimed in natural order BEHIND the reported ones and most of a batch's medium-long chunks were among them.
#ifndef RC_ORDER_CLASSES
#define RC_ORDER_CLASSES 10
#endif
#ifndef RC_ORDER_RAISE_PCT
#define RC_ORDER_RAISE_PCT 70
#define RC_ORDER_LOWER_PCT 30
#endif
namespace {
constexpr int kOrderThreads = 256, kOrderPerThread = 4, kOrderTile = kOrderThreads * kOrderPerThread, kOrderClasses = RC_ORDER_CLASSES, kOrderMaxBlocks = 256;
constexpr int kOrderPacked = (kOrderClasses + 4) / 5;  // u64 words of 12-bit per-class counters (k_order_scatter)
static_assert(kOrderClasses + 1 <= 32, "k_order_scatter sums the blocks' counts with 32 threads per block group");
static_assert(rc_scene::ChunkHistory::kHeaderWords == (uint32_t)kHistHeaderWords, "rc_internal.h repeats the size of a header copy");
static_assert((uint32_t)(kOrderTile * kOrderMaxBlocks) == kHistSlotStride, "a slot's cost array holds the most chunks the rebuild kernels handle");
constexpr int kOrderCountWords = kOrderClasses + 1;  // per block: the classes' chunk counts, the largest cost
constexpr size_t kHistWords = kHistCounts + (size_t)kHistSlots * kOrderMaxBlocks * kOrderCountWords;

__device__ inline int order_class(uint32_t c, uint32_t thr, uint32_t top) {
    if (c == 0u) return kOrderClasses - 1;
    const uint32_t span = top > thr ? top - thr + 1u : 1u, above = c > thr ? c - thr : 0u;
    const uint32_t q = above * (uint32_t)(kOrderClasses - 1) / span;  // 0 .. kOrderClasses - 2 (and beyond when this launch's rays outlived the scale)
    return q >= (uint32_t)(kOrderClasses - 1) ? 0 : (int)(kOrderClasses - 2) - (int)q;
}
__device__ inline uint32_t* order_counts(uint32_t* hist, int slot) { return hist + kHistCounts + (size_t)slot * kOrderMaxBlocks * kOrderCountWords; }

__global__ __launch_bounds__(kOrderThreads) void k_order_count(const uint32_t* cost_base, uint32_t* ctl, uint32_t parity, uint32_t n_chunks) {
    __shared__ uint32_t cnt[kOrderCountWords];
    const uint32_t* hist = ctl + parity * kHistHeaderWords;  // the header copy the shape's next launch reads
    for (int slot = 0; slot < kHistSlots; ++slot) {
        if (hist[kHistPending + slot] == 0u) continue;  // (wave-uniform: a header word)
        if (threadIdx.x < kOrderCountWords) cnt[threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t* cost = cost_base + (size_t)slot * kHistSlotStride;
        const uint32_t thr = hist[kHistScale + 4 * slot], top = hist[kHistScale + 4 * slot + 1];  // the scale the recording launch worked with
        const uint32_t first = blockIdx.x * kOrderTile + threadIdx.x * kOrderPerThread;
        uint32_t mx = 0u;
        for (int j = 0; j < kOrderPerThread; ++j)
            if (first + j < n_chunks) { const uint32_t c = cost[first + j]; atomicAdd(&cnt[order_class(c, thr, top)], 1u); mx = c > mx ? c : mx; }
        if (mx) atomicMax(&cnt[kOrderClasses], mx);
        __syncthreads();
        if (threadIdx.x < kOrderCountWords) order_counts(ctl, slot)[blockIdx.x * kOrderCountWords + threadIdx.x] = cnt[threadIdx.x];  // [classes ..., block maximum]
        __syncthreads();
    }
}
__global__ __launch_bounds__(kOrderThreads) void k_order_scatter(uint32_t* cost_base, uint32_t* order_base, uint32_t* ctl, uint32_t parity, uint32_t n_chunks, uint32_t* host_words) {
    uint32_t* hist = ctl + parity * kHistHeaderWords;
    typedef hipcub::BlockScan<unsigned long long, kOrderThreads> Scan;
    __shared__ typename Scan::TempStorage scan_tmp;
    __shared__ uint32_t total[kOrderCountWords], before[kOrderClasses];  // chunks of class k in all blocks (last: the largest cost) / in the blocks before this one
    __shared__ uint32_t unreported_of[kHistSlots], longest_of[kHistSlots];  // per rebuilt slot, for the header update at the end (a serial walk over the blocks' counts there cost 20 us)
    const uint32_t first = blockIdx.x * kOrderTile + threadIdx.x * kOrderPerThread;
    for (int slot = 0; slot < kHistSlots; ++slot) {
        if (hist[kHistPending + slot] == 0u) continue;
        uint32_t* cost = cost_base + (size_t)slot * kHistSlotStride;
        uint32_t* order = order_base + (size_t)slot * kHistSlotStride;
        if (threadIdx.x < kOrderCountWords) { total[threadIdx.x] = 0u; if (threadIdx.x < kOrderClasses) before[threadIdx.x] = 0u; }
        __syncthreads();
        const uint32_t* counts = order_counts(ctl, slot);
        {   // thread (word w, group g) sums word w of the blocks g, g + 8, ... in registers, then one LDS atomic per thread.  (One thread per
            // BLOCK with an LDS atomic per word put 32 lanes on the same LDS address twenty times over: 17 of this kernel's 22 us.)
            const uint32_t w = threadIdx.x & 31u, g = threadIdx.x >> 5;
            if (w < (uint32_t)kOrderCountWords) {
                uint32_t all = 0u, mine = 0u;
                for (uint32_t b = g; b < gridDim.x; b += kOrderThreads / 32) {
                    const uint32_t c = counts[b * kOrderCountWords + w];
                    if (w == (uint32_t)kOrderClasses) all = c > all ? c : all;
                    else { all += c; mine += b < blockIdx.x ? c : 0u; }
                }
                if (w == (uint32_t)kOrderClasses) atomicMax(&total[w], all);
                else { if (all) atomicAdd(&total[w], all); if (mine) atomicAdd(&before[w], mine); }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) { unreported_of[slot] = total[kOrderClasses - 1]; longest_of[slot] = total[kOrderClasses]; }
        const uint32_t thr = hist[kHistScale + 4 * slot], top = hist[kHistScale + 4 * slot + 1];
        int cls[kOrderPerThread];
        unsigned long long packed[kOrderPacked];  // this thread's chunks per class, 12 bits each, five classes per word
#pragma unroll
        for (int q = 0; q < kOrderPacked; ++q) packed[q] = 0ull;
        for (int j = 0; j < kOrderPerThread; ++j) {
            cls[j] = first + j < n_chunks ? order_class(cost[first + j], thr, top) : -1;
#pragma unroll
            for (int q = 0; q < kOrderPacked; ++q) packed[q] += (cls[j] >= 0 && cls[j] / 5 == q) ? 1ull << (12 * (cls[j] % 5)) : 0ull;
        }
        unsigned long long prefix[kOrderPacked];
#pragma unroll
        for (int q = 0; q < kOrderPacked; ++q) {
            if (q) __syncthreads();
            Scan(scan_tmp).ExclusiveSum(packed[q], prefix[q]);
        }
        uint32_t pos[kOrderClasses], acc = 0;
        for (int k = 0; k < kOrderClasses; ++k) { pos[k] = acc + before[k] + (uint32_t)((prefix[k / 5] >> (12 * (k % 5))) & 0xFFFull); acc += total[k]; }
#pragma unroll
        for (int j = 0; j < kOrderPerThread; ++j) {
            if (cls[j] < 0) continue;
            uint32_t at = 0;  // pos[cls[j]]++ by selection: an array indexed by a run-time class would live in scratch memory
#pragma unroll
            for (int k = 0; k < kOrderClasses; ++k) { const bool mine = cls[j] == k; at = mine ? pos[k] : at; pos[k] += mine ? 1u : 0u; }
            order[at] = first + j;
            cost[first + j] = 0u;
        }
        __syncthreads();
    }
    // the header: every block has read what it needs of it above; the LAST block to get here turns the recordings into orders in use
    // (No fence: what the blocks wrote above -- orders, cleared costs -- is read by the NEXT kernel; the last block reads only header words
    // nobody writes in this kernel and its own LDS.  An agent-scope release here wrote back the whole L2 -- full of the previous launch's hit
    // records -- and made this kernel 25 us long.)
    __shared__ uint32_t last_block;
    if (threadIdx.x == 0) last_block = atomicAdd(ctl + kHistTicket, 1u) + 1u == gridDim.x ? 1u : 0u;
    __syncthreads();
    if (!last_block || threadIdx.x != 0) return;
    ctl[kHistTicket] = 0u;
    for (int slot = 0; slot < kHistSlots; ++slot) {
        if (hist[kHistPending + slot] == 0u) continue;
        const uint32_t unreported = unreported_of[slot], mx = longest_of[slot];
        uint32_t* scale = hist + kHistScale + 4 * slot;  // the scale of the slot's NEXT recording launch
        const uint32_t thr = scale[0], reported = n_chunks - unreported;
        uint32_t next = thr;
        if (reported * 100ull > (unsigned long long)n_chunks * RC_ORDER_RAISE_PCT) next += (next >> 2) + 1u;        // more than this share of the chunks reported: raise the bar
        else if (reported * 100ull < (unsigned long long)n_chunks * RC_ORDER_LOWER_PCT && next > 2u) next -= next >> 2;  // fewer than this: lower it
        scale[2] = next;
        scale[3] = mx > next ? mx : next + 8u;  // the longest lifetime just seen scales the next recording's classes
        hist[kHistPending + slot] = 0u;
        hist[kHistHasOrder + slot] = 1u;
    }
    if (host_words) __hip_atomic_store(host_words + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // nothing waits for a rebuild any more
}
}  // namespace

// The history entry of a launch shape (chunk count, mode, chunk geometry, stream): launches on one stream are ordered, so its buffers have
// one user at a time; another stream gets its own entry.  Returns false when cost ordering does not apply to this launch.
bool rc_cost_order_setup(rc_scene* s, uint64_t n, int any_hit, hipStream_t stream, rc::RcClaim& c, const RcRay* d_rays, const float* host_sample) {
    const uint64_t n_base64 = (n + c.pool - 1) / c.pool;  // chunks (the claim order permutes whole chunks; parts follow their chunk)
    // only where the order can matter: at least a claim per wave; the cost path maps rays to chunks with a shift (pool a power of two)
    if (s->n_tlas_nodes == 0) return false;  // (an empty scene's launches return before they look at the history: the header's two copies must flip with the host's parity)
    if (!s->opt.cost_order || c.n_chunks < c.total_waves || n_base64 < 64u || n_base64 > (uint64_t)(kOrderTile * kOrderMaxBlocks) || (c.pool & (c.pool - 1u)) != 0u || c.pool < 16u) return false;
    if (!d_rays && !host_sample) return false;
    const uint32_t n_base = (uint32_t)n_base64;
    // Keyed by the number of CHUNKS (a batch a few rays shorter or longer has the same chunks in the same places), mode, stream and chunk size.
    rc_scene::ChunkHistory* h = nullptr;
    for (auto& e : s->histories)
        if (e.n_chunks == n_base && e.any == any_hit && e.stream == stream && e.pool == c.pool) { h = &e; break; }
    if (!h) {
        // A shape this scene has no history for.  Nothing here may block or synchronise (the *_device entry points are
        // asynchronous, and a workload whose batch size changes with every launch -- a wavefront tracer compacting its rays bounce by
        // bounce -- misses every time): a free entry is taken at once; a full table gives up an entry only to a shape that has been seen
        // BEFORE (a one-off size learns nothing worth keeping), and only an entry whose buffers nobody can still be using -- its stream
        // is this launch's stream (stream order protects them) or whose last launch has completed (the entry's own event, queried, never waited for).  Otherwise the
        // launch simply runs in natural order.  Buffers have ONE size (the largest batch the rebuild kernels handle), so re-keying an
        // entry frees and allocates nothing.
        bool seen_before = false;
        for (const auto& r : s->recent_shapes) if (r.n_chunks == n_base && r.any == any_hit && r.stream == stream && r.pool == c.pool) { seen_before = true; break; }
        if (!seen_before) {
            if (s->recent_shapes.size() < (size_t)rc_scene::kRecentShapes) s->recent_shapes.emplace_back();
            auto& r = s->recent_shapes.size() < (size_t)rc_scene::kRecentShapes ? s->recent_shapes.back() : s->recent_shapes[s->recent_clock % (uint64_t)rc_scene::kRecentShapes];
            s->recent_clock += 1;
            r.n_chunks = n_base; r.any = any_hit; r.stream = stream; r.pool = c.pool;
        }
        if (s->histories.size() < (size_t)rc_scene::kMaxHistories) {
            s->histories.emplace_back();
            h = &s->histories.back();
            h->cost.reserve((size_t)kHistSlots * kHistSlotStride); h->order.reserve((size_t)kHistSlots * kHistSlotStride); h->ctl.reserve(kHistWords);
            h->samples.reserve(2 * kHistSampleFloats);
        } else {
            if (!seen_before) return false;
            size_t victim = s->histories.size();
            uint64_t oldest = ~0ull;
            for (size_t i = 0; i < s->histories.size(); ++i) {
                auto& e = s->histories[i];
                if (e.last_use >= oldest) continue;
                const bool reusable = e.stream == stream || e.last.idle();  // (stream order protects a same-stream entry; any other is free once its last launch is done)
                if (reusable) { victim = i; oldest = e.last_use; }
            }
            if (victim == s->histories.size()) return false;
            h = &s->histories[victim];
        }
        h->n_items = n; h->any = any_hit; h->stream = stream; h->n_chunks = n_base; h->pool = c.pool;
        h->rebuild_credit = 0; h->next_record = 8; h->records_asked = 0;
        if (h->fresh_streak.p) { reinterpret_cast<volatile uint32_t*>(h->fresh_streak.p)[0] = 0u; reinterpret_cast<volatile uint32_t*>(h->fresh_streak.p)[1] = 0u; reinterpret_cast<volatile uint32_t*>(h->fresh_streak.p)[2] = 0u; }
        RC_HIP(hipMemsetAsync(h->ctl.p, 0, sizeof(uint32_t) * kHistCounts, stream));  // every batch slot empty (the cost arrays are cleared when a slot is given out)
        h->gen = 0;
        h->parity = 0;
    }
    h->last_use = ++s->history_clock;
    s->cur_history = (int)(h - s->histories.data());
    // (A shape whose batches never repeat leaves the mechanism for a while: counted and decided on the device, order_select / order_commit.)
    h->fresh_streak.ensure();
    h->gen += 1;
    // Turning recordings into orders: the pair of rebuild kernels goes in front of this launch only when a slot MAY hold a recording --
    // the shape's first launches (a batch records its launches 2-4), the launches after the device reported a launch that was not a
    // repeat (a new batch starts its own launches 2-4), and the launch after one this host asked to record.  Everything else about the
    // launch's claim order is decided inside the launch itself (order_select).
    volatile uint32_t* pinned = reinterpret_cast<volatile uint32_t*>(h->fresh_streak.p);  // [0] the run of non-repeats, [1] a recording waits, [2] the pause (written by order_commit at the START of a launch: as of the latest launch that has STARTED -- possibly one still in flight; benign, the rebuild kernels are stream-ordered behind it)
    // [2] = launches of the pause still to go after the launch that wrote the word | that launch's number << 8.  A caller that enqueues far ahead of the
    // device reads an OLD word: every launch of the shape enqueued since takes one launch off the pause.  (Round 6: without the age a burst of 200
    // launches enqueued behind a pause -- bench.py's repeated-batch extra -- stayed "paused" on the host for its whole length: the recordings of the
    // batch's launches 2-4 were never turned into an order.)
    const uint32_t w2 = pinned[2], skip_seen = w2 & 0xFFu, gen_seen = w2 >> 8;
    const uint32_t since = ((uint32_t)(h->gen - 1) - gen_seen) & 0xFFFFFFu;   // launches of the shape enqueued after the one that wrote the word
    const bool paused = skip_seen > 1u + since;  // (this launch and the next are still inside the pause: nothing records, nothing to rebuild)
    if (pinned[0] > 0u) h->rebuild_credit = 6;
    if (paused) h->rebuild_credit = 0;
    const uint32_t blocks = (n_base + kOrderTile - 1) / kOrderTile;
    if (h->gen >= 2 && !paused && (h->gen <= 6 || h->rebuild_credit > 0 || pinned[1] != 0u)) {
        hipLaunchKernelGGL(k_order_count, dim3(blocks), dim3(kOrderThreads), 0, stream, h->cost.p, h->ctl.p, h->parity, n_base);
        hipLaunchKernelGGL(k_order_scatter, dim3(blocks), dim3(kOrderThreads), 0, stream, h->cost.p, h->order.p, h->ctl.p, h->parity, n_base, h->fresh_streak.p);
    }
    if (h->rebuild_credit > 0) h->rebuild_credit -= 1;
    // the recording cadence of a batch past its fourth launch: one launch in 7, 8, 9, 7, ... of the shape (not a fixed period: two or three
    // batches alternating on the shape must all get their turn)
    c.want_record = 0;
    if (h->gen >= h->next_record && !paused) {
        c.want_record = 1;
        h->next_record = h->gen + 7 + (h->records_asked % 3);
        h->records_asked += 1;
        if (h->rebuild_credit < 1) h->rebuild_credit = 1;  // (the next launch gets the rebuild pair)
    }
    const float ex = s->root_max[0] - s->root_min[0], ey = s->root_max[1] - s->root_min[1], ez = s->root_max[2] - s->root_min[2];
    const float l2 = ex * ex + ey * ey + ez * ez;
    c.inv_l2 = (l2 > 0.f && l2 < 1e30f) ? 1.0f / l2 : 0.f;
    c.sample_rays = d_rays;
    c.n_sample = n;
    for (int k = 0; k < 8; ++k) c.sample_host[k] = host_sample ? host_sample[k] : 0.f;
    c.samples = h->samples.p;
    c.host_streak = h->fresh_streak.p;
    c.host_gen = (uint32_t)h->gen & 0xFFFFFFu;
    c.init_thr = (uint32_t)s->opt.cost_thr;
    c.order = h->order.p;
    c.cost = h->cost.p;
    c.hist = h->ctl.p;
    c.parity = h->parity;  // the launch reads this copy of the header and writes the other: the shape's next launch reads that one
    h->parity ^= 1u;
    return true;
}

void rc_launch_trace(rc_scene* s, const RcRay* d_rays, RcHit* d_hits, uint64_t n, int any_hit, hipStream_t stream, bool learn_order) {
    if (n == 0) return;
    RcLaunchGuard launch(s, stream);  // serialises the enqueue: trace calls on one scene may come from several host threads
    uint64_t want = (n + kBlock - 1) / kBlock, cap = (uint64_t)s->n_cus * rc_blocks_per_cu(s);
    uint32_t blocks = (uint32_t)(want < cap ? want : cap);
    uint32_t total_threads = blocks * kBlock;
    int64_t kernel = s->opt.kernel;  // the fall-back rules below choose for this launch only
    if (kernel < 0)  // auto: tiny batches gain nothing from refilling; a TLAS that fits the LDS planes (<= 256 instances) is read from there
        kernel = (n < (uint64_t)total_threads * 5 / 4) ? 0 : (s->n_tlas_nodes <= (uint32_t)kTlasLdsNodes ? 5 : 6);  // measured crossover (tools/archive/small_batch_probe.py): ~1.2 rays per resident lane
    if (kernel == 4 && (s->n_tlas_nodes > (uint32_t)kTlasLdsNodes || s->n_static_instances > (uint32_t)kTlasLdsInst)) kernel = 3;
    if (kernel == 5 && s->n_tlas_nodes > (uint32_t)kTlasLdsNodes) kernel = 3;
    if (kernel == 6 && s->tlas_top_k + s->blas_top_k == 0) kernel = 3;  // nothing to stage
    if (kernel >= 3 && ((uint64_t)(s->n_flat_nodes + s->n_tlas_nodes) * 64u >= (1ull << 32) || n >= (1ull << 38))) kernel = 1;  // buffer offsets and chunk ids are 32-bit
    if (kernel == 4) {  // one 1024-thread workgroup per CU
        blocks = (uint32_t)std::min<uint64_t>((n + kBigBlock - 1) / kBigBlock, (uint64_t)s->n_cus);
        total_threads = blocks * kBigBlock;
    }
    if (kernel == 5 || kernel == 6) {  // two 768-thread workgroups per CU
        blocks = (uint32_t)std::min<uint64_t>((n + kMidBlock - 1) / kMidBlock, (uint64_t)s->n_cus * (s->opt.blocks_per_cu == 1 ? 1 : 2));
        total_threads = blocks * kMidBlock;
    }
    TraceArgs a;
    a.v = rc_scene_view(s, total_threads);
    if (any_hit && s->opt.entry_cull < 2) a.v.inst_cull = nullptr;  // any_hit rays stop at their first hit and enter few instances that a cull would spare: the test costs what it saves (shadow rays -2 %); 2 = cull there too
    a.rays = d_rays; a.hits = d_hits; a.n_rays = n;
    rc_claim_fill(s, n, total_threads / 64u, a.claim);
    a.refill = (int)s->opt.refill;
    a.sched_thr = kernel == 2 ? 32 : (int)s->opt.sched_thr;  // kernel 2's vote threshold is its own (lanes that must wait for a batch), tuned at 32
    a.stats = rc_stats_words(s);
    a.timeline = reinterpret_cast<unsigned long long*>(s->opt.timeline_ptr);
    const bool wide = rc_stack16(s);  // the STACK16 shape stages all of the renumbered tops, the 32-bit shape a prefix
    if ((kernel == 5 || kernel == 4) && s->opt.blas_top) { a.blas_k = (wide && kernel == 5) ? s->blas_top_k : s->blas_top_k32; a.lds_blas_base = (s->n_tlas_nodes + 1) / 2 - 1; }
    if (kernel == 6) {  // a plan made for the full-LDS kernels (<= 256 instances) has no TLAS renumbering: tlas_k = 0, its blas_k still fits
        a.tlas_k = wide ? s->tlas_top_k : s->tlas_top_k32; a.blas_k = s->opt.blas_top ? (wide ? s->blas_top_k : s->blas_top_k32) : 0; a.lds_blas_base = a.tlas_k;
    }
    launch.bind();
    if (learn_order && (kernel == 3 || kernel == 5 || kernel == 6) && !launch.capturing) rc_cost_order_setup(s, n, any_hit, stream, a.claim, d_rays);  // (the rebuild pair, when there is one, runs in front of the timed kernel)
    if (any_hit) launch_variant<true>(s, kernel, a, blocks, stream); else launch_variant<false>(s, kernel, a, blocks, stream);
    launch.finish();
}
