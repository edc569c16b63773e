// rc_bvh4.hip -- BVH4: collapse of a BLAS's binary LBVH into 4-wide nodes on the device, and closest_hit4 / any_hit4.
//
// Reference: src/bvh4.jl -- BVHNode4 :40-69, gather_children_bvh2 :201-300, collapse_bvh2_to_bvh4 :314-497 (a sequential
// host loop there), build_blas4 :511-522, fast_intersect_bbox4 :533-554, intersect_all_children4 :562-599,
// closest_hit4 :606-689, any_hit4 :696-766.  The reference's BVH4 is BLAS-level only (no instances, no TLAS traversal).
//
// Collapse on the GPU.  The reference numbers BVH4 nodes by a FIFO over interior subtrees: a task takes the next index
// for its own node, then one index per leaf child in slot order; interior children are queued.  A FIFO whose pushes
// happen in task order is a level-order walk, so the same numbering falls out of a level-synchronous pass: per level,
// one kernel runs gather_children_bvh2 for every task, an exclusive scan over (1 + #leaf children, #interior children)
// packed in one u64 gives every task its node index and its children's queue positions, and a second kernel writes the
// records, patches the parent's child slot and emits the next level's tasks.  The exported array is byte-identical to
// the reference algorithm's.
//
// Traversal record (device-internal, 128 bytes, two aligned 64-byte halves):
//   interior: child[4] @0 (1-based BVH4 index, bit 31 set when that child is a leaf, INVALID for unused slots),
//             four boxes @16 (min.xyz, max.xyz each), parent @112, child_count | primitive_count << 8 @116
//   leaf:     child[0] = 1-based sorted primitive index @0, v0 v1 v2 @16..51 (the reference fetches prims[prim_idx]; the
//             vertices are copied into the leaf so a leaf visit is one record fetch), primitive index again @52,
//             the leaf's AABB @64..87 (only for rc_export_blas4_nodes), parent/counts as above.
// The leaf bit lets a lane know which of the two loads to issue before it has fetched anything.
#include <hipcub/hipcub.hpp>

#include "rc_traverse_core.h"

namespace {

using namespace rc;

constexpr uint32_t kLeafBit = 0x80000000u;

struct Task4 { uint32_t bvh2, parent4, slot, pad; };       // bvh2 = 1-based BVH2 node of the subtree root
struct Gather4 { uint32_t ch[4]; uint32_t count, leaf_mask, pad0, pad1; };

struct Box6 { float3_ mn, mx; };

// get_node_aabb(node, is_interior) as the collapse uses it (src/bvh4.jl:265-274, src/instanced-bvh.jl:1141-1160)
__device__ inline Box6 child_box(const RcNode& n, bool interior) {
    Box6 b;
    if (interior) {
        b.mn = min3v(mk3(n.f[0], n.f[1], n.f[2]), mk3(n.f[6], n.f[7], n.f[8]));
        b.mx = max3v(mk3(n.f[3], n.f[4], n.f[5]), mk3(n.f[9], n.f[10], n.f[11]));
    } else {
        const float3_ v0 = mk3(n.f[0], n.f[1], n.f[2]), v1 = mk3(n.f[3], n.f[4], n.f[5]), v2 = mk3(n.f[6], n.f[7], n.f[8]);
        b.mn = min3v(min3v(v0, v1), v2);
        b.mx = max3v(max3v(v0, v1), v2);
    }
    return b;
}

// gather_children_bvh2 (:201-300) for an interior subtree root (the collapse never queues a leaf).
__device__ inline void gather_children(const RcNode* nodes2, uint32_t root_idx, Gather4& g) {
    uint32_t queue[8];
    int queue_size = 2, child_count = 0;
    uint32_t leaf_mask = 0;
    for (int i = 0; i < 4; ++i) g.ch[i] = RC_INVALID_NODE;
    queue[0] = nodes2[root_idx - 1].child0;
    queue[1] = nodes2[root_idx - 1].child1;
    while (child_count < 4 && queue_size > 0) {  // :234-277
        int best = 0;
        for (int i = 0; i < queue_size; ++i) {
            if (nodes2[queue[i] - 1].child0 != RC_INVALID_NODE && child_count + queue_size - 1 + 2 <= 4) { best = i; break; }
        }
        const uint32_t node_idx = queue[best];
        queue[best] = queue[queue_size - 1];
        queue_size -= 1;
        const uint32_t c0 = nodes2[node_idx - 1].child0;
        const bool interior = c0 != RC_INVALID_NODE;
        if (interior && child_count + queue_size + 2 <= 4) {
            queue[queue_size++] = c0;
            queue[queue_size++] = nodes2[node_idx - 1].child1;
        } else {
            g.ch[child_count] = node_idx;
            if (!interior) leaf_mask |= 1u << child_count;
            child_count += 1;
        }
    }
    while (queue_size > 0 && child_count < 4) {  // :280-297
        const uint32_t node_idx = queue[queue_size - 1];
        queue_size -= 1;
        g.ch[child_count] = node_idx;
        if (nodes2[node_idx - 1].child0 == RC_INVALID_NODE) leaf_mask |= 1u << child_count;
        child_count += 1;
    }
    g.count = (uint32_t)child_count;
    g.leaf_mask = leaf_mask;
    g.pad0 = g.pad1 = 0;
}

__global__ void k_collapse_gather(const RcNode* nodes2, const Task4* tasks, uint32_t n_tasks, Gather4* gathers, unsigned long long* counts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tasks) return;
    Gather4 g;
    gather_children(nodes2, tasks[i].bvh2, g);
    gathers[i] = g;
    const uint32_t n_leaf = __popc(g.leaf_mask);
    counts[i] = ((unsigned long long)(1u + n_leaf) << 32) | (unsigned long long)(g.count - n_leaf);
}

__device__ inline void write_leaf4(RcNode4* out, const RcNode& n2, uint32_t parent4) {  // :368-387, :455-474
    const Box6 b = child_box(n2, false);
    uint32_t* w = reinterpret_cast<uint32_t*>(out);
    float* f = reinterpret_cast<float*>(out);
    w[0] = n2.child1; w[1] = w[2] = w[3] = RC_INVALID_NODE;
    for (int k = 0; k < 9; ++k) f[4 + k] = n2.f[k];  // v0 v1 v2
    w[13] = n2.child1;
    w[14] = w[15] = 0;
    f[16] = b.mn.x; f[17] = b.mn.y; f[18] = b.mn.z; f[19] = b.mx.x; f[20] = b.mx.y; f[21] = b.mx.z;
    for (int k = 22; k < 28; ++k) w[k] = 0;
    w[28] = parent4;
    w[29] = 0u | (1u << 8);  // child_count 0, primitive_count 1
    w[30] = w[31] = 0;
}

// One task of the collapse: take node index base_nodes + (off >> 32) + 1, write the interior record and its leaf children, patch the
// parent's child slot, queue the interior children at next_tasks[(uint32_t)off ...] (next_tasks may point to LDS or global memory).
__device__ inline void emit_task(const RcNode* nodes2, const Task4 tk, const Gather4& g, unsigned long long off, uint32_t base_nodes,
                                 RcNode4* nodes4, Task4* next_tasks) {
    const uint32_t current4 = base_nodes + (uint32_t)(off >> 32) + 1u;  // 1-based
    uint32_t next_pos = (uint32_t)off;
    if (tk.parent4 != RC_INVALID_NODE) reinterpret_cast<uint32_t*>(nodes4 + (tk.parent4 - 1))[tk.slot] = current4;  // :415-444
    uint32_t* w = reinterpret_cast<uint32_t*>(nodes4 + (current4 - 1));
    float* f = reinterpret_cast<float*>(w);
    uint32_t leaf_idx = current4;
    for (int c = 0; c < 4; ++c) {
        Box6 b;
        b.mn = mk3(INFINITY, INFINITY, INFINITY); b.mx = mk3(-INFINITY, -INFINITY, -INFINITY);  // Bounds3()
        uint32_t child_word = RC_INVALID_NODE;
        if (c < (int)g.count) {
            const RcNode n2 = nodes2[g.ch[c] - 1];
            const bool is_leaf = (g.leaf_mask >> c) & 1u;
            b = child_box(n2, !is_leaf);
            if (is_leaf) {
                leaf_idx += 1;
                child_word = leaf_idx | kLeafBit;
                write_leaf4(nodes4 + (leaf_idx - 1), n2, current4);
            } else {
                next_tasks[next_pos++] = Task4{g.ch[c], current4, (uint32_t)c, 0u};  // patched when that task runs
            }
        }
        w[c] = child_word;
        f[4 + 6 * c + 0] = b.mn.x; f[4 + 6 * c + 1] = b.mn.y; f[4 + 6 * c + 2] = b.mn.z;
        f[4 + 6 * c + 3] = b.mx.x; f[4 + 6 * c + 4] = b.mx.y; f[4 + 6 * c + 5] = b.mx.z;
    }
    w[28] = tk.parent4;
    w[29] = g.count;  // child_count, primitive_count 0
    w[30] = w[31] = 0;
}

__global__ void k_collapse_emit(const RcNode* nodes2, const Task4* tasks, uint32_t n_tasks, const Gather4* gathers,
                                const unsigned long long* offsets, uint32_t base_nodes, RcNode4* nodes4, Task4* next_tasks) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tasks) return;
    emit_task(nodes2, tasks[i], gathers[i], offsets[i], base_nodes, nodes4, next_tasks);
}

// Narrow levels (<= kSmallLevel tasks: the top of the tree and the tails of skinny subtrees) run inside ONE workgroup, level after
// level, with the task queues in LDS and a block-wide scan -- no launches, no host round trip per level.  state[0] = tasks in `tasks`
// on entry / tasks left for the wide path on exit (0 = done, else > kSmallLevel and stored in `wide_out`), state[1] = nodes allocated.
constexpr int kSmallLevel = 1024;
__global__ __launch_bounds__(kSmallLevel) void k_collapse_small(const RcNode* nodes2, const Task4* tasks, uint32_t* state, RcNode4* nodes4, Task4* wide_out) {
    typedef hipcub::BlockScan<unsigned long long, kSmallLevel> Scan;
    __shared__ typename Scan::TempStorage scan_tmp;
    __shared__ Task4 q[2][kSmallLevel];
    uint32_t n_tasks = state[0], base_nodes = state[1];
    if (threadIdx.x < n_tasks) q[0][threadIdx.x] = tasks[threadIdx.x];
    __syncthreads();
    int cur = 0;
    while (n_tasks > 0 && n_tasks <= (uint32_t)kSmallLevel) {
        const bool mine = threadIdx.x < n_tasks;
        Task4 tk = Task4{0u, 0u, 0u, 0u};
        Gather4 g;
        unsigned long long cnt = 0ull, off = 0ull, total = 0ull;
        if (mine) {
            tk = q[cur][threadIdx.x];
            gather_children(nodes2, tk.bvh2, g);
            const uint32_t n_leaf = __popc(g.leaf_mask);
            cnt = ((unsigned long long)(1u + n_leaf) << 32) | (unsigned long long)(g.count - n_leaf);
        }
        Scan(scan_tmp).ExclusiveSum(cnt, off, total);
        const uint32_t next_n = (uint32_t)total;
        Task4* next_tasks = next_n <= (uint32_t)kSmallLevel ? q[cur ^ 1] : wide_out;
        if (mine) emit_task(nodes2, tk, g, off, base_nodes, nodes4, next_tasks);
        base_nodes += (uint32_t)(total >> 32);
        n_tasks = next_n;
        cur ^= 1;
        __syncthreads();
    }
    if (threadIdx.x == 0) { state[0] = n_tasks; state[1] = base_nodes; }
}

__global__ void k_collapse_totals(const unsigned long long* counts, const unsigned long long* offsets, uint32_t n_tasks, uint32_t* totals) {
    const unsigned long long t = offsets[n_tasks - 1] + counts[n_tasks - 1];
    totals[0] = (uint32_t)(t >> 32);
    totals[1] = (uint32_t)t;
}

__global__ void k_single_leaf4(const RcNode* nodes2, RcNode4* nodes4) {  // n == 1 (:334-351)
    if (blockIdx.x == 0 && threadIdx.x == 0) write_leaf4(nodes4, nodes2[0], RC_INVALID_NODE);
}

// device record -> the reference's 120-byte BVHNode4
__global__ void k_export4(const RcNode4* nodes4, uint32_t n, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(nodes4 + i);
    uint32_t* o = out + (size_t)i * 30;
    const uint32_t counts = w[29];
    if ((counts & 0xFFu) == 0) {  // leaf
        o[0] = w[0]; o[1] = o[2] = o[3] = RC_INVALID_NODE;
        for (int k = 0; k < 6; ++k) o[4 + k] = w[16 + k];
        for (int k = 10; k < 28; ++k) o[k] = 0;
    } else {
        for (int c = 0; c < 4; ++c) o[c] = (w[c] == RC_INVALID_NODE) ? RC_INVALID_NODE : (w[c] & ~kLeafBit);
        for (int k = 4; k < 28; ++k) o[k] = w[k];
    }
    o[28] = w[28];
    o[29] = counts & 0xFFFFu;
}

// ---- traversal -------------------------------------------------------------------------------------------------------
struct Trace4Args {
    const RcNode4* nodes;
    uint32_t n_nodes, root_word, n_prims;
    const RcRay* rays;
    RcHit* hits;
    uint64_t n_rays;
    RcClaim claim;             // how waves claim ray chunks (rc_traverse_core.h)
    int refill, int_thr;
    uint32_t* overflow;
    uint32_t total_threads;
    uint32_t* status;
};

// fast_intersect_bbox4 (:533-554); jl_minf / jl_maxf propagate NaNs like Julia's min / max
__device__ inline bool slab4(const float3_ inv, const float3_ ox, float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                             float tmin, float closest_t, float& t_entry) {
    const float fx = mxx * inv.x + ox.x, fy = mxy * inv.y + ox.y, fz = mxz * inv.z + ox.z;
    const float nx = mnx * inv.x + ox.x, ny = mny * inv.y + ox.y, nz = mnz * inv.z + ox.z;
    const float max_t = jl_minf(jl_minf(jl_minf(jl_maxf(fx, nx), jl_maxf(fy, ny)), jl_maxf(fz, nz)), closest_t);
    const float min_t = jl_maxf(jl_maxf(jl_maxf(jl_minf(fx, nx), jl_minf(fy, ny)), jl_minf(fz, nz)), tmin);
    t_entry = min_t;
    return min_t <= max_t;
}

// Persistent waves, two phases per round (the structure of phased_trace): an inner loop over interior nodes for as long as
// enough lanes have one pending, then one pass over the lanes that wait at a leaf, then write-out + refill.
template <bool ANY, int LDS_N>
__global__ __launch_bounds__(kBlock, 6) void k_trace4(Trace4Args a) {
    __shared__ uint32_t lds_stack[LDS_N * kBlock];
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
    LaneStackT<LDS_N> st(lds_stack + threadIdx.x, a.overflow + gtid, a.total_threads, a.status);
    const int lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t nrs = make_rsrc(a.nodes, a.n_nodes * 128u);
    unsigned long long pool_next = 0, pool_end = 0;
    bool exhausted = false;
    uint64_t my_ray = 0;
    float3_ o = mk3(0, 0, 0), d = mk3(0, 0, 0), inv = mk3(0, 0, 0), ox = mk3(0, 0, 0);
    float closest_t = 0.f, hit_u = 0.f, hit_v = 0.f;
    uint32_t closest_prim = RC_INVALID_NODE;
    uint32_t node = RC_INVALID_NODE;  // INVALID + !live = empty lane; INVALID + live = finished
    int sp = 0;
    bool live = false;
    const float tmin = 0.0f;  // closest_hit4 / any_hit4 both start from ray_mint = 0 (:610, :700)

    for (;;) {
        for (;;) {
            const bool is_int = (node & kLeafBit) == 0u;  // INVALID has the bit set
            const int n_int = __popcll(__ballot(is_int));
            if (n_int == 0) break;
            if (is_int) {
                const uint32_t off = (node - 1u) << 7;
                const u4v ch = __builtin_amdgcn_raw_buffer_load_b128(nrs, off, 0, 0);
                const float4 b0 = buf_f4(nrs, off + 16), b1 = buf_f4(nrs, off + 32), b2 = buf_f4(nrs, off + 48), b3 = buf_f4(nrs, off + 64),
                             b4 = buf_f4(nrs, off + 80), b5 = buf_f4(nrs, off + 96);
                // intersect_all_children4 (:562-599): slots in order, unused slots (child == INVALID) never hit
                float t0, t1, t2, t3;
                const bool h0 = slab4(inv, ox, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, tmin, closest_t, t0) && ch.x != RC_INVALID_NODE;
                const bool h1 = slab4(inv, ox, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, tmin, closest_t, t1) && ch.y != RC_INVALID_NODE;
                const bool h2 = slab4(inv, ox, b3.x, b3.y, b3.z, b3.w, b4.x, b4.y, tmin, closest_t, t2) && ch.z != RC_INVALID_NODE;
                const bool h3 = slab4(inv, ox, b4.z, b4.w, b5.x, b5.y, b5.z, b5.w, tmin, closest_t, t3) && ch.w != RC_INVALID_NODE;
                // position of each hit child after the reference's stable insertion sort by entry distance (:590-596):
                // hits from earlier slots with t <= mine and from later slots with t < mine come first
                const int r0 = (int)(h1 && t1 < t0) + (int)(h2 && t2 < t0) + (int)(h3 && t3 < t0);
                const int r1 = (int)(h0 && t0 <= t1) + (int)(h2 && t2 < t1) + (int)(h3 && t3 < t1);
                const int r2 = (int)(h0 && t0 <= t2) + (int)(h1 && t1 <= t2) + (int)(h3 && t3 < t2);
                const int r3 = (int)(h0 && t0 <= t3) + (int)(h1 && t1 <= t3) + (int)(h2 && t2 <= t3);
                const int hc = (int)h0 + (int)h1 + (int)h2 + (int)h3;
                // far children are pushed last-to-second (:637-642): sorted position r >= 1 lands at sp + (hc - 1 - r)
                const int top = sp + hc - 1;
                if (h0 && r0 > 0) st.store(top - r0, ch.x);
                if (h1 && r1 > 0) st.store(top - r1, ch.y);
                if (h2 && r2 > 0) st.store(top - r2, ch.z);
                if (h3 && r3 > 0) st.store(top - r3, ch.w);
                if (hc > 0) {
                    sp = top;
                    node = (h0 && r0 == 0) ? ch.x : (h1 && r1 == 0) ? ch.y : (h2 && r2 == 0) ? ch.z : ch.w;
                } else {
                    node = st.pop(sp);
                }
            }
            if (n_int < a.int_thr) break;
        }
        {
            const bool is_leaf = node != RC_INVALID_NODE && (node & kLeafBit) != 0u;
            if (is_leaf) {
                const uint32_t off = ((node & ~kLeafBit) - 1u) << 7;
                const float4 va = buf_f4(nrs, off + 16), vb = buf_f4(nrs, off + 32);
                const u2v vc = __builtin_amdgcn_raw_buffer_load_b64(nrs, off + 48, 0, 0);
                const uint32_t prim_idx = vc.y;
                const float3_ v0 = mk3(va.x, va.y, va.z), v1 = mk3(va.w, vb.x, vb.y), v2 = mk3(vb.z, vb.w, __uint_as_float(vc.x));
                // fast_intersect_triangle (src/instanced-bvh.jl:1756-1797)
                const float3_ e1 = sub3(v1, v0), e2 = sub3(v2, v0);
                const float3_ s1 = cross3(d, e2);
                const float det = dot3(s1, e1);
                const float invd = 1.0f / det;
                const float3_ dd = sub3(o, v0);
                const float u = dot3(dd, s1) * invd;
                const float3_ s2 = cross3(dd, e1);
                const float v = dot3(d, s2) * invd;
                const float t = dot3(e2, s2) * invd;
                const bool hit = prim_idx <= a.n_prims &&  // :652
                                 !(u < 0.0f || u > 1.0f) && !(v < 0.0f || (u + v) > 1.0f) && !(t < tmin || t > closest_t);
                closest_prim = hit ? prim_idx : closest_prim;
                closest_t = hit ? t : closest_t;
                hit_u = hit ? u : hit_u;
                hit_v = hit ? v : hit_v;
                if (ANY && hit) node = RC_INVALID_NODE;  // :744-749
                else node = st.pop(sp);
            }
        }
        {
            const bool fin = live && node == RC_INVALID_NODE;
            const int n_free = __popcll(__ballot(fin || !live));
            const bool can_refill = !(exhausted && pool_next == pool_end);
            if (n_free == 64 && !can_refill && !__ballot(fin)) break;
            if (n_free >= a.refill || n_free == 64 || !can_refill) {
                if (fin) {
                    uint4 w0, w1;
                    if (closest_prim != RC_INVALID_NODE) {  // :679-683
                        w0 = make_uint4(1u, __float_as_uint(closest_t), closest_prim - 1u, 0u);
                        w1 = make_uint4(__float_as_uint(hit_u), __float_as_uint(hit_v), RC_INVALID_NODE, 0u);
                    } else {
                        w0 = make_uint4(0u, 0u, RC_INVALID_NODE, 0u);
                        w1 = make_uint4(0u, 0u, RC_INVALID_NODE, 0u);
                    }
                    uint4* out = reinterpret_cast<uint4*>(a.hits + my_ray);
                    out[0] = w0;
                    out[1] = w1;
                    live = false;
                }
                while (can_refill) {
                    const unsigned long long free_mask = __ballot(!live);
                    const int nf = __popcll(free_mask);
                    if (nf == 0) break;
                    if (pool_next == pool_end) {
                        if (exhausted) break;
                        if (!rc_claim_chunk(a.claim, nullptr, (blockIdx.x * kBlock + threadIdx.x) >> 6, lane, a.n_rays, pool_next, pool_end)) { exhausted = true; break; }
                    }
                    const unsigned long long left = pool_end - pool_next;
                    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(free_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)free_mask, 0u));
                    if (!live && rank < left) {
                        my_ray = pool_next + rank;
                        const RcRay r = load_ray(a.rays, my_ray);
                        o = mk3(r.ox, r.oy, r.oz);  // check_direction (src/ray.jl:39-49), safe_invdir (:612)
                        d = mk3(r.dx == 0.0f ? 0.0f : r.dx, r.dy == 0.0f ? 0.0f : r.dy, r.dz == 0.0f ? 0.0f : r.dz);
                        inv = mk3(safe_inv1(d.x), safe_inv1(d.y), safe_inv1(d.z));
                        ox = mk3(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
                        closest_t = r.tmax;
                        hit_u = hit_v = 0.0f;
                        closest_prim = RC_INVALID_NODE;
                        sp = 0;
                        st.push(sp, RC_INVALID_NODE);  // stands for the reference's stack_ptr == 0 exit (:670-675)
                        node = a.root_word;
                        live = true;
                    }
                    pool_next += ((unsigned long long)nf < left) ? (unsigned long long)nf : left;
                }
            }
        }
    }
}

}  // namespace

// build_blas4 (:511-522) for a geometry whose BVH2 is already on the device.
void rc_build_blas4(rc_scene* s, Blas& b) {
    const uint32_t n = b.n_prims;
    if (n == 0) throw RcError(1, "Cannot build BLAS4 from empty primitive list");
    b.nodes4.reserve(b.n_nodes);  // interior tasks <= n - 1, leaves = n
    b.n_nodes4 = 0;
    b.root_word4 = 1u;
    hipStream_t st = s->stream;
    if (n == 1) {
        hipLaunchKernelGGL(k_single_leaf4, dim3(1), dim3(64), 0, st, b.nodes.p, b.nodes4.p);
        RC_HIP(hipGetLastError());
        b.n_nodes4 = 1;
        b.root_word4 = 1u | kLeafBit;
        return;
    }
    // scratch: two task queues, gather results, packed counts + offsets, totals
    const size_t cap = n;  // a level holds at most n - 1 interior subtrees
    s->c4_tasks_a.reserve(cap * 4); s->c4_tasks_b.reserve(cap * 4); s->c4_gather.reserve(cap * 8);
    s->c4_counts.reserve(cap); s->c4_offsets.reserve(cap); s->c4_totals.reserve(2);
    size_t tmp_bytes = 0;
    RC_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, s->c4_counts.p, s->c4_offsets.p, (int)cap, st));
    s->sort_tmp.reserve(tmp_bytes);
    Task4* cur = reinterpret_cast<Task4*>(s->c4_tasks_a.p);
    Task4* nxt = reinterpret_cast<Task4*>(s->c4_tasks_b.p);
    Gather4* gathers = reinterpret_cast<Gather4*>(s->c4_gather.p);
    const Task4 root{1u, RC_INVALID_NODE, 0u, 0u};
    RC_HIP(hipMemcpyAsync(cur, &root, sizeof(root), hipMemcpyHostToDevice, st));
    uint32_t n_tasks = 1, base_nodes = 0;
    while (n_tasks > 0) {
        if (n_tasks <= (uint32_t)kSmallLevel) {
            const uint32_t st_in[2] = {n_tasks, base_nodes};
            RC_HIP(hipMemcpyAsync(s->c4_totals.p, st_in, sizeof(st_in), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_collapse_small, dim3(1), dim3(kSmallLevel), 0, st, b.nodes.p, cur, s->c4_totals.p, b.nodes4.p, nxt);
            uint32_t st_out[2];
            RC_HIP(hipMemcpyAsync(st_out, s->c4_totals.p, sizeof(st_out), hipMemcpyDeviceToHost, st));
            RC_HIP(hipStreamSynchronize(st));
            n_tasks = st_out[0];
            base_nodes = st_out[1];
            std::swap(cur, nxt);
            continue;
        }
        const uint32_t blocks = (n_tasks + 255) / 256;
        hipLaunchKernelGGL(k_collapse_gather, dim3(blocks), dim3(256), 0, st, b.nodes.p, cur, n_tasks, gathers, s->c4_counts.p);
        size_t tb = s->sort_tmp.cap;
        RC_HIP(hipcub::DeviceScan::ExclusiveSum(s->sort_tmp.p, tb, s->c4_counts.p, s->c4_offsets.p, (int)n_tasks, st));
        hipLaunchKernelGGL(k_collapse_emit, dim3(blocks), dim3(256), 0, st, b.nodes.p, cur, n_tasks, gathers, s->c4_offsets.p, base_nodes, b.nodes4.p, nxt);
        hipLaunchKernelGGL(k_collapse_totals, dim3(1), dim3(1), 0, st, s->c4_counts.p, s->c4_offsets.p, n_tasks, s->c4_totals.p);
        uint32_t totals[2];
        RC_HIP(hipMemcpyAsync(totals, s->c4_totals.p, sizeof(totals), hipMemcpyDeviceToHost, st));
        RC_HIP(hipStreamSynchronize(st));
        base_nodes += totals[0];
        n_tasks = totals[1];
        std::swap(cur, nxt);
    }
    RC_HIP(hipGetLastError());
    b.n_nodes4 = base_nodes;
}

void rc_export_blas4(rc_scene* s, const Blas& b, void* host_out) {
    if (b.n_nodes4 == 0) return;
    DevBuf<uint32_t> tmp;
    tmp.reserve((size_t)b.n_nodes4 * 30);
    hipLaunchKernelGGL(k_export4, dim3((b.n_nodes4 + 255) / 256), dim3(256), 0, s->stream, b.nodes4.p, b.n_nodes4, tmp.p);
    RC_HIP(hipGetLastError());
    RC_HIP(hipMemcpyAsync(host_out, tmp.p, (size_t)b.n_nodes4 * 120, hipMemcpyDeviceToHost, s->stream));
    RC_HIP(hipStreamSynchronize(s->stream));
}

void rc_launch_trace4(rc_scene* s, const Blas& b, const RcRay* d_rays, RcHit* d_hits, uint64_t n, int any_hit, hipStream_t stream) {
    if (n == 0) return;
    if ((uint64_t)b.n_nodes4 * 128u >= (1ull << 32)) throw RcError(1, "BLAS4 too large for 32-bit buffer offsets");
    uint64_t want = (n + kBlock - 1) / kBlock, cap = (uint64_t)s->n_cus * 6;
    const uint32_t blocks = (uint32_t)(want < cap ? want : cap);
    const uint32_t total_threads = blocks * kBlock;
    RcLaunchGuard launch(s, stream);
    Trace4Args a;
    a.nodes = b.nodes4.p; a.n_nodes = b.n_nodes4; a.root_word = b.root_word4; a.n_prims = b.n_prims;
    a.rays = d_rays; a.hits = d_hits; a.n_rays = n;
    if (n >= (1ull << 38)) throw RcError(1, "ray batches of 2^38 rays or more are not supported by the BVH4 kernels");
    rc_claim_fill(s, n, total_threads / 64u, a.claim);
    a.refill = (int)s->opt.refill;
    a.int_thr = (int)s->opt.sched_thr;
    a.overflow = rc_launch_overflow(s, total_threads); a.total_threads = total_threads;  // (a captured launch: its own region, allocated here -- ADVICE r5)
    a.status = rc_status_word(s);
    launch.start();
    if (any_hit) hipLaunchKernelGGL((k_trace4<true, 24>), dim3(blocks), dim3(kBlock), 0, stream, a);
    else hipLaunchKernelGGL((k_trace4<false, 24>), dim3(blocks), dim3(kBlock), 0, stream, a);
    launch.finish();
}
