// NOT COMPILED INTO THE LIBRARY.  Kept as the record of a measured negative (round 2): bit-exact on the first run, 1.8x SLOWER than kernel 5
// on every workload (C3 1.17 vs 0.64 ms, C4 6.16 vs 3.49, C2 0.83 vs 0.45): three workgroup barriers per round put the 12 waves of a
// workgroup in lockstep, so at any moment most of them wait for the slowest one's interior loop and the CU runs at a fraction of the
// 6 waves / SIMD the kernel needs to hide its fetch latency.  To try it again: copy next to rc_traverse_core.h, include it from
// rc_traverse.hip, add a k_trace_coop<ANY> kernel that calls coop_trace, and a launch case with kCoopLdsBytes of dynamic LDS.
//
// rc_traverse_coop.h -- EXPERIMENTAL trace kernel 7: the phased persistent traversal of rc_traverse_core.h with its two sparse phases
// (triangle tests, instance entries) pooled across the WORKGROUP instead of run per wave.
//
// Why: the phased kernels are bound by VALU issue and only 44 % of the issued lane slots carry a ray; the worst offenders are the leaf
// phase (70 VALU instructions for ~11 of 64 lanes) and the instance-entry phase (80 for ~10).  Here a lane that reaches a BLAS leaf or a
// TLAS leaf posts a small request into an LDS ring shared by the 12 waves of the workgroup; after a workgroup barrier any wave takes
// 64 requests at a time and serves them at full lane fill, a second barrier later the owners read their results back.  Per ray the
// arithmetic is the reference's, operand for operand (the values travel through LDS bit for bit), so results stay bit-identical.
//
// Synchronisation is three s_barriers per round (posted | served | collected) and nothing else -- no polling, no flags: a wave that has run out of rays simply
// returns (s_barrier only counts the waves of a workgroup that are still running; tools/barrier_probe.hip), and the rings are
// double-buffered by the parity of the round so that the counters of one parity are cleared while the other is in use.
#pragma once
#include "rc_traverse_core.h"

namespace rc {

constexpr int kCoopStack = 12;                      // LDS lane-stack entries (the rings are paid for by 4 entries per lane)
constexpr int kCoopLeafSlots = 144, kCoopEntrySlots = 136;
constexpr size_t kCoopRingBytes = (size_t)5 * kCoopLeafSlots * 8 + (size_t)6 * kCoopEntrySlots * 8 + 32;
constexpr size_t kCoopLdsBytes = (size_t)kCoopStack * kMidBlock * 4 + kLdsTopBytes + kCoopRingBytes;
static_assert(kCoopLdsBytes <= 81920, "two workgroups per CU");

template <bool ANY, class Source, class Sink>
__device__ inline void coop_trace(const SceneView& av, const PersistArgs& a, unsigned char* smem, const Source& src, const Sink& sink) {
    constexpr int BLOCK = kMidBlock, LDS_N = kCoopStack;
    uint32_t* const lds_stack = reinterpret_cast<uint32_t*>(smem);
    const LdsTop top(smem + (size_t)LDS_N * BLOCK * 4);
    float2* const lq = reinterpret_cast<float2*>(smem + (size_t)LDS_N * BLOCK * 4 + kLdsTopBytes);  // leaf ring: 5 planes of kCoopLeafSlots float2
    float2* const eq = lq + 5 * kCoopLeafSlots;                                                        // entry ring: 6 planes of kCoopEntrySlots float2
    uint32_t* const ctr = reinterpret_cast<uint32_t*>(eq + 6 * kCoopEntrySlots);                       // [parity * 4 + {leaf count, leaf head, entry count, entry head}]
    const float2* const tl = top.tl;
    const uint32_t* const lt = top.lt;
    const float2* const il = top.il;
    const uint32_t gtid = blockIdx.x * BLOCK + threadIdx.x;
    LaneStackP<LDS_N, BLOCK> st(lds_stack + threadIdx.x, av.overflow + gtid, av.total_threads, av.status);
    const int lane = threadIdx.x & 63;
    if (av.n_tlas_nodes == 0) {  // empty TLAS: every ray misses; the whole workgroup leaves before the first barrier
        for (uint64_t i = gtid; i < a.n_items; i += av.total_threads) sink(i, false, 0.0f, 0.0f, 0.0f, RC_INVALID_NODE, -1);
        return;
    }
    if (threadIdx.x < 8) ctr[threadIdx.x] = 0u;
    stage_lds_top<BLOCK>(top, av, a.blas_k, a.lds_blas_base);
    __syncthreads();
    const uint32_t n_instances = (av.n_tlas_nodes + 1u) >> 1;
    const uint32_t tlas_off = av.tlas_off;
    const __amdgpu_buffer_rsrc_t nrs1 = make_rsrc(reinterpret_cast<const char*>(av.blas_nodes) - 64, (av.n_nodes_total + 1u) * 64u);
    unsigned long long pool_next = 0, pool_end = 0;
    bool exhausted = false;
    uint64_t my_ray = 0;
    float3_ wo = mk3(0, 0, 0), wd = mk3(0, 0, 0), winv = mk3(0, 0, 0);
    float3_ o = mk3(0, 0, 0), d = mk3(0, 0, 0), inv = mk3(0, 0, 0), ox = mk3(0, 0, 0);
    float tmin = 0.f, closest_t = 0.f, hit_u = 0.f, hit_v = 0.f;
    uint32_t closest_prim = RC_INVALID_NODE, cur_off = 0, n_level = 0;
    uint32_t node = RC_INVALID_NODE;
    int closest_inst = -1, cur_inst = -1;
    typename LaneStackP<LDS_N, BLOCK>::pos_t sp = st.empty();
    bool live = false;
    uint32_t pend = 0;       // 0 = nothing posted; (slot << 2) | 1 = leaf request in slot; (slot << 2) | 2 = entry request in slot
    uint32_t parity = 0;     // wave-uniform: which half of the counters this round uses
    int thr_eff = __builtin_amdgcn_readfirstlane(a.int_thr);

    for (;;) {
        // ---- A. interior phase, as in phased_trace
        for (;;) {
            const bool is_int = node < n_level;
            const int n_int = __popcll(__ballot(is_int));
            if (n_int == 0) break;
            if (is_int) {
                float4 na, nb, nc;
                u2v ch;
                constexpr int PS = kLdsPlaneNodes;
                if (cur_inst < 0 || node <= a.blas_k) {
                    const float2* q = tl + ((node - 1u) + (cur_inst < 0 ? 0u : a.lds_blas_base));
                    const float2 p0 = q[0], p1 = q[PS], p2 = q[2 * PS], p3 = q[3 * PS], p4 = q[4 * PS], p5 = q[5 * PS], p6 = q[6 * PS];
                    na = make_float4(p0.x, p0.y, p1.x, p1.y); nb = make_float4(p2.x, p2.y, p3.x, p3.y); nc = make_float4(p4.x, p4.y, p5.x, p5.y);
                    ch = u2v{__float_as_uint(p6.x), __float_as_uint(p6.y)};
                } else {
                    const uint32_t off = (cur_off + node) << 6;
                    na = buf_f4(nrs1, off); nb = buf_f4(nrs1, off, 16); nc = buf_f4(nrs1, off, 32);
                    ch = __builtin_amdgcn_raw_buffer_load_b64(nrs1, off, 48, 0);
                }
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
                const bool h0 = t0_min <= t0_max, h1 = t1_min <= t1_max;
                const bool first0 = (t0_min < t1_min) & h0;
                const uint32_t near_c = first0 ? ch.x : ch.y, far_c = first0 ? ch.y : ch.x;
                const bool near_ok = first0 | h1, far_ok = h0 & (h1 | !first0);
                if (far_ok) st.push(sp, far_c);
                node = near_ok ? near_c : st.pop(sp);
            }
            if (n_int < thr_eff) break;
        }
        // ---- B. post: BLAS leaves and TLAS leaves go into the workgroup's rings
        uint32_t* const c = ctr + parity * 4u;
        {
            const bool at_leaf = pend == 0u && node >= n_level && node < RC_TOP_LEVEL_SENTINEL;
            const bool want_leaf = at_leaf && cur_inst >= 0, want_entry = at_leaf && cur_inst < 0;
            const unsigned long long ml = __ballot(want_leaf), me = __ballot(want_entry);
            if (ml) {
                uint32_t base = 0;
                if (lane == 0) base = __hip_atomic_fetch_add(c + 0, (uint32_t)__popcll(ml), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                base = __builtin_amdgcn_readfirstlane(base);
                const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(ml >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ml, 0u));
                if (want_leaf && slot < (uint32_t)kCoopLeafSlots) {  // a full ring: the lane tries again next round
                    float2* q = lq + slot;
                    q[0] = make_float2(o.x, o.y); q[kCoopLeafSlots] = make_float2(o.z, d.x); q[2 * kCoopLeafSlots] = make_float2(d.y, d.z);
                    q[3 * kCoopLeafSlots] = make_float2(tmin, closest_t);
                    q[4 * kCoopLeafSlots] = make_float2(__uint_as_float((cur_off + node) << 6), 0.0f);
                    pend = (slot << 2) | 1u;
                }
            }
            if (me) {
                uint32_t base = 0;
                if (lane == 0) base = __hip_atomic_fetch_add(c + 2, (uint32_t)__popcll(me), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                base = __builtin_amdgcn_readfirstlane(base);
                const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(me >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)me, 0u));
                if (want_entry && slot < (uint32_t)kCoopEntrySlots) {
                    cur_inst = (int)lt[node - n_level];  // leaf of sorted instance j is node n - 1 + j; its child1 word
                    float2* q = eq + slot;
                    q[0] = make_float2(wo.x, wo.y); q[kCoopEntrySlots] = make_float2(wo.z, wd.x); q[2 * kCoopEntrySlots] = make_float2(wd.y, wd.z);
                    q[3 * kCoopEntrySlots] = make_float2(__uint_as_float((uint32_t)cur_inst), 0.0f);
                    pend = (slot << 2) | 2u;
                }
            }
        }
        __syncthreads();
        // ---- C. serve: any wave takes 64 requests at a time, at full lane fill
        {
            if (lane == 0) { uint32_t* const other = ctr + (parity ^ 1u) * 4u; other[0] = 0u; other[1] = 0u; other[2] = 0u; other[3] = 0u; }  // next round's counters (idle since the barrier before last)
            const uint32_t posted_l = c[0], posted_e = c[2];
            const uint32_t total_l = posted_l < (uint32_t)kCoopLeafSlots ? posted_l : (uint32_t)kCoopLeafSlots;
            const uint32_t total_e = posted_e < (uint32_t)kCoopEntrySlots ? posted_e : (uint32_t)kCoopEntrySlots;
            for (;;) {  // fast_intersect_triangle (:1756-1797) for the posted leaves
                uint32_t b = 0;
                if (lane == 0) b = __hip_atomic_fetch_add(c + 1, 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                b = __builtin_amdgcn_readfirstlane(b);
                if (b >= total_l) break;
                const uint32_t slot = b + (uint32_t)lane;
                if (slot < total_l) {
                    float2* q = lq + slot;
                    const float2 r0 = q[0], r1 = q[kCoopLeafSlots], r2 = q[2 * kCoopLeafSlots], r3 = q[3 * kCoopLeafSlots], r4 = q[4 * kCoopLeafSlots];
                    const float3_ ro = mk3(r0.x, r0.y, r1.x), rd = mk3(r1.y, r2.x, r2.y);
                    const uint32_t off = __float_as_uint(r4.x);
                    const float4 na = buf_f4(nrs1, off);
                    const float2 nb = buf_f2(nrs1, off, 16);
                    const float4 nc = buf_f4(nrs1, off, 32);
                    const float3_ v0 = mk3(na.x, na.y, nc.x), v1 = mk3(na.z, na.w, nc.y), v2 = mk3(nb.x, nb.y, nc.z);
                    const float3_ e1 = sub3(v1, v0), e2 = sub3(v2, v0);
                    const float3_ s1 = cross3(rd, e2);
                    const float det = dot3(s1, e1);
                    const float invd = 1.0f / det;
                    const float3_ dd = sub3(ro, v0);
                    const float u = dot3(dd, s1) * invd;
                    const float3_ s2 = cross3(dd, e1);
                    const float v = dot3(rd, s2) * invd;
                    const float t = dot3(e2, s2) * invd;
                    const bool hit = !(u < 0.0f || u > 1.0f) && !(v < 0.0f || (u + v) > 1.0f) && !(t < r3.x || t > r3.y);
                    q[0] = make_float2(t, u);
                    q[kCoopLeafSlots] = make_float2(v, __uint_as_float(hit ? 1u : 0u));
                }
            }
            for (;;) {  // instance entries (:1961-1977): world ray -> instance space
                uint32_t b = 0;
                if (lane == 0) b = __hip_atomic_fetch_add(c + 3, 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                b = __builtin_amdgcn_readfirstlane(b);
                if (b >= total_e) break;
                const uint32_t slot = b + (uint32_t)lane;
                if (slot < total_e) {
                    float2* q = eq + slot;
                    const float2 r0 = q[0], r1 = q[kCoopEntrySlots], r2 = q[2 * kCoopEntrySlots], r3 = q[3 * kCoopEntrySlots];
                    const float3_ rwo = mk3(r0.x, r0.y, r1.x), rwd = mk3(r1.y, r2.x, r2.y);
                    const float2* m = il + __float_as_uint(r3.x);
                    const float2 p0 = m[0], p1 = m[kTlasLdsInst], p2 = m[2 * kTlasLdsInst], p3 = m[3 * kTlasLdsInst], p4 = m[4 * kTlasLdsInst], p5 = m[5 * kTlasLdsInst];
                    const float3_ lo = mk3(p0.x * rwo.x + p0.y * rwo.y + p1.x * rwo.z + p1.y, p2.x * rwo.x + p2.y * rwo.y + p3.x * rwo.z + p3.y,
                                           p4.x * rwo.x + p4.y * rwo.y + p5.x * rwo.z + p5.y);
                    const float3_ ld = mk3(p0.x * rwd.x + p0.y * rwd.y + p1.x * rwd.z, p2.x * rwd.x + p2.y * rwd.y + p3.x * rwd.z, p4.x * rwd.x + p4.y * rwd.y + p5.x * rwd.z);
                    const float3_ li = mk3(safe_inv1(ld.x), safe_inv1(ld.y), safe_inv1(ld.z));
                    q[0] = make_float2(lo.x, lo.y); q[kCoopEntrySlots] = make_float2(lo.z, ld.x); q[2 * kCoopEntrySlots] = make_float2(ld.y, ld.z);
                    q[3 * kCoopEntrySlots] = make_float2(li.x, li.y);
                    q[4 * kCoopEntrySlots] = make_float2(li.z, -lo.x * li.x);
                    q[5 * kCoopEntrySlots] = make_float2(-lo.y * li.y, -lo.z * li.z);
                }
            }
        }
        __syncthreads();
        // ---- D. collect
        {
            const uint32_t kind = pend & 3u, slot = pend >> 2;
            if (kind == 1u) {
                const float2 r0 = lq[slot], r1 = lq[kCoopLeafSlots + slot];
                const bool hit = __float_as_uint(r1.y) != 0u;
                closest_prim = hit ? node - n_level + 1u : closest_prim;
                closest_inst = hit ? cur_inst : closest_inst;
                closest_t = hit ? r0.x : closest_t;
                hit_u = hit ? r0.y : hit_u;
                hit_v = hit ? r1.x : hit_v;
                if (ANY && hit) node = RC_INVALID_NODE;
                else node = st.pop(sp);
                pend = 0u;
            } else if (kind == 2u) {
                const float2* q = eq + slot;
                const float2 r0 = q[0], r1 = q[kCoopEntrySlots], r2 = q[2 * kCoopEntrySlots], r3 = q[3 * kCoopEntrySlots], r4 = q[4 * kCoopEntrySlots], r5 = q[5 * kCoopEntrySlots];
                const float2 p6 = il[6 * kTlasLdsInst + cur_inst];
                st.push(sp, RC_TOP_LEVEL_SENTINEL);
                node = 1;
                cur_off = __float_as_uint(p6.x);
                n_level = __float_as_uint(p6.y);
                o = mk3(r0.x, r0.y, r1.x); d = mk3(r1.y, r2.x, r2.y); inv = mk3(r3.x, r3.y, r4.x); ox = mk3(r4.y, r5.x, r5.y);
                pend = 0u;
            }
            // return to the top level (:1996-2006): cheap, stays with the owner
            if (node == RC_TOP_LEVEL_SENTINEL) {
                node = st.pop(sp);
                cur_inst = -1;
                cur_off = tlas_off; n_level = n_instances;
                inv = winv;
                ox = mk3(-wo.x * inv.x, -wo.y * inv.y, -wo.z * inv.z);
            }
        }
        __syncthreads();  // the ring's slots are reused by the next round's posts: nobody may still be reading this round's results
        parity ^= 1u;
        // ---- E. finished lanes: write out; refill when enough lanes are free (as in phased_trace)
        {
            const bool fin = live && node == RC_INVALID_NODE;
            const int n_free = __popcll(__ballot(fin || !live));
            const bool can_refill = !(exhausted && pool_next == pool_end);
            if (!can_refill) {
                const int half_live = (64 - n_free) / 2;
                thr_eff = __builtin_amdgcn_readfirstlane(half_live < a.int_thr ? (half_live > 1 ? half_live : 1) : a.int_thr);
            }
            if (n_free == 64 && !can_refill && !__ballot(fin)) break;  // this wave is done; the others go on without it
            if (n_free >= a.refill || n_free == 64 || !can_refill) {
                if (fin) {
                    sink(my_ray, closest_inst >= 0, closest_t, hit_u, hit_v, closest_prim, closest_inst);
                    live = false;
                }
                while (can_refill) {
                    const unsigned long long free_mask = __ballot(!live);
                    const int nf = __popcll(free_mask);
                    if (nf == 0) break;
                    if (pool_next == pool_end) {
                        if (exhausted) break;
                        if (!rc_claim_chunk(a.claim, (blockIdx.x * BLOCK + threadIdx.x) >> 6, lane, a.n_items, pool_next, pool_end)) { exhausted = true; break; }
                    }
                    const unsigned long long left = pool_end - pool_next;
                    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(free_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)free_mask, 0u));
                    if (!live && rank < left) {
                        my_ray = pool_next + rank;
                        const RcRay r = src(my_ray);
                        wo = mk3(r.ox, r.oy, r.oz);
                        wd = mk3(r.dx == 0.0f ? 0.0f : r.dx, r.dy == 0.0f ? 0.0f : r.dy, r.dz == 0.0f ? 0.0f : r.dz);
                        winv = mk3(safe_inv1(wd.x), safe_inv1(wd.y), safe_inv1(wd.z));
                        inv = winv;
                        ox = mk3(-wo.x * inv.x, -wo.y * inv.y, -wo.z * inv.z);
                        tmin = ANY ? 0.0f : r.tmin;
                        closest_t = r.tmax;
                        hit_u = hit_v = 0.0f;
                        closest_prim = RC_INVALID_NODE;
                        closest_inst = -1; cur_inst = -1;
                        cur_off = tlas_off; n_level = n_instances;
                        sp = st.empty();
                        st.push(sp, RC_INVALID_NODE);
                        node = 1;
                        live = true;
                        pend = 0u;
                    }
                    pool_next += ((unsigned long long)nf < left) ? (unsigned long long)nf : left;
                }
            }
        }
    }
}

}  // namespace rc
