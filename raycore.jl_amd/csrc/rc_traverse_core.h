// rc_traverse_core.h -- the per-ray traversal (one statement-for-statement restatement of the reference's
// closest_hit / any_hit loop) shared by the trace kernels (rc_traverse.hip) and the drivers (rc_drivers.hip).
//
// Reference: closest_hit src/instanced-bvh.jl:1902-2024, any_hit :2034-2140, safe_invdir :1742-1748,
// fast_intersect_bbox :1841-1859, intersect_internal_node :1807-1832, fast_intersect_triangle :1756-1797.
#pragma once
#include <type_traits>

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
    uint32_t n_nodes_total;   // BLAS + TLAS nodes in blas_nodes (sizes the buffer descriptor)
    uint32_t n_inst;          // instance records
    const float4* inst_cull;  // per instance (c_w, A), (B, ...): the entry-cull sphere (rc_build.hip k_inst_recs); nullptr = never skip an entry
    uint32_t* overflow;       // [kTotalStack][total_threads] spill area of the lane stacks (entries below the LDS depth unused)
    uint32_t total_threads;
    uint32_t* status;         // [0] = stack overflow flag
};

// How a persistent kernel's waves claim work (host side: RcLaunchGuard / rc_claim_fill, rc_traverse.hip).  A claim is one chunk
// of `pool` consecutive items (large claims keep a wave on neighbouring rays: coherent fetches).  The chunks are dealt out by
// n_shards counters -- in round c of a shard's counter the shards share chunks c * n .. c * n + n - 1, which of them a shard gets
// rotating with c: with a fixed assignment a shard would own one column band of a 2048-ray-wide image, and bands differ in cost by
// 2x -- because returning atomics on ONE address serialise at 12.6 ns each however many waves issue them (tools/archive/atomic_probe.hip).
// A wave stays with the shard it started on: the shards own the same number of interleaved chunks (+-1) and each is drained by
// 1/n_shards of the waves, so they run dry together, and probing other counters at the end costs more (every probe of a contended
// line queues behind the claims) than the few chunks' worth of imbalance it could recover.
// The counters reset themselves: every wave ends with exactly one failed claim, so shard s sees chunks(s) + waves(s) atomics per
// launch, and the wave whose failed claim is the last of them puts the counter back to zero -- no memset node between back-to-back
// launches (6 us, 1 % of a 4 M-ray launch, 3 % of a 1 M-ray one), nothing a later launch could clear under an earlier one, and
// a launch that is never enqueued leaves nothing behind.
// Work is cut into chunks of `pool` consecutive items.  Towards the END OF THE CLAIM ORDER a chunk is dealt out in parts (guided
// self-scheduling, option "taper"): the first G1 claims take a whole chunk each, then come chunks dealt in halves, quarters and eighths,
// with the boundaries placed where the items still to be handed out equal taper / 8 x (part size) x (waves of the launch).  Large claims
// keep a wave's lanes on neighbouring rays (coherent fetches) while there is plenty of work; small ones at the end let the waves run dry
// together -- a launch lasts until the LAST claimed part has been traced, and 128 rays are two generations of a wave's lanes (~150 us on
// C2).  taper = 0: every claim is a whole chunk.
// Cost-ordered claiming (host side: ChunkHistory, rc_traverse.hip): WHICH chunk the p-th position of the claim order stands for is either p
// itself or order[p], a permutation built on the device from what an earlier launch of the same BATCH recorded (the batch is recognised by
// sample rays; kHist* below).  A launch lasts until its longest rays are done, and a long ray that sits in a chunk claimed late STARTS late;
// chunks that held long rays before are therefore claimed first, whole (longest-processing-time-first with the batch's earlier launches as the
// predictor: render loops, repeated queries), and the cheap ones end the launch in small parts.
struct RcClaim {
    uint32_t* counters;            // kClaimShards words, kShardStrideWords apart, zero between launches
    uint32_t shard_shift;          // n_shards = 1 << shard_shift <= kClaimShards and <= the waves of the launch (every shard has a wave)
    uint32_t n_chunks;             // CLAIMS of the launch (whole chunks + parts), < 2^31
    uint32_t pool;                 // items per chunk
    uint32_t total_waves;          // waves of the launch: wave w claims from shard w & (n_shards - 1)
    uint32_t g1, g2, g3;           // first claim that takes half / a quarter / an eighth of a chunk (= n_chunks when there is none)
    uint32_t c1, c2, c3;           // position in the claim order of the first chunk dealt in halves / quarters / eighths
    const uint32_t* order;         // position in the claim order -> chunk, a permutation of 0 .. ceil(n_items / pool) - 1; nullptr = natural order
    uint32_t* cost;                // per chunk: the longest time in flight (interior iterations of its wave) of a ray of the chunk this launch; nullptr = not recorded
    uint32_t* hist;                // the history's header words (kHist*, below), two copies.  Every wave of the launch works out by itself which of the history's batch slots
                                   // the launch belongs to (order_select: a pure function of the header copy `parity` and the launch's sample rays) -- `cost` / `order` are slot
                                   // 0's arrays, slot k's lie k * kHistSlotStride words on --; wave 0 of workgroup 0 writes the next state into the other copy.  nullptr = no history
    uint32_t parity;               // which copy of the header and of the remembered samples this launch READS (the host flips it per launch of the shape)
    uint32_t pool_shift;           // log2(pool) when pool is a power of two (the cost path maps a ray to its chunk with a shift)
    // what order_select needs to recognise the batch
    const RcRay* sample_rays;      // the launch's ray array, or nullptr: generated rays, described by sample_host
    uint64_t n_sample;             // rays in it
    float sample_host[8];
    float inv_l2;                  // 1 / (scene diagonal)^2: origins are compared relative to the scene
    float* samples;                // 2 copies x kHistSlots x kHistSamples remembered sample rays (8 floats each)
    uint32_t* host_streak;         // pinned words for the host's decision about the rebuild kernels: [0] the run of launches that were not repeats, [1] a recording waits, [2] launches of a pause still to go | host_gen << 8
    uint32_t init_thr;             // reporting threshold a batch starts with
    uint32_t want_record;          // the host's cadence: a slot past its fourth launch records in this launch
    uint32_t host_gen;             // the host's count of launches of this shape (24 bits): written back next to the pause word, so that a host that has enqueued
                                   // far ahead of the device knows HOW OLD the pause count it reads is (each launch since takes one off it)
};
// A history (rc_scene::ChunkHistory) remembers up to kHistSlots different BATCHES of one launch shape, told apart on the device by
// kHistSamples sample rays (VERDICT r3 #5a: two cameras alternating on one stream each learn from their OWN previous launch, and a
// batch never seen before runs in natural order instead of in somebody else's).  Header words:
constexpr int kHistSlots = 4, kHistSamples = 64;
constexpr uint32_t kHistSlotStride = 1u << 18;  // words between the slots' cost arrays = the most chunks the rebuild kernels handle
// The header exists TWICE (kHistHeaderWords apart): a launch reads copy `parity` and writes the next state into the other one, so the one wave
// that writes (order_commit) can do it at the START of the launch, while every other wave is still reading, and nothing is left to do at the
// end.  The host flips `parity` with every launch of the shape; the rebuild kernels work in place on the copy the next launch reads.
constexpr int kHistSel = 0, kHistOrderValid = 1, kHistLifeThr = 2, kHistClock = 3, kHistFresh = 4, kHistRecorded = 5,  // of the latest launch (tests, tools)
              kHistFreshStreak = 36 /* consecutive launches that were not a repeat (identical sample rays) of a remembered batch */,
              kHistSkipLeft = 38 /* launches of the shape still to go out OUTSIDE the mechanism (its batches do not repeat: order_commit) */,
              kHistStamp = 8 /* [kHistSlots] */, kHistGen = 12 /* [kHistSlots] */,
              kHistScale = 16 /* [kHistSlots][4]: (threshold, top of the scale) the slot's latest recording launch worked with; the pair its next one will */,
              kHistPending = 32 /* [kHistSlots]: the slot's cost array holds a recording that no order has been built from yet */,
              kHistHasOrder = 40 /* [kHistSlots]: the slot's order array holds an order built from a recording of the slot's batch */,
              kHistHeaderWords = 48 /* one copy */,
              kHistTicket = 2 * kHistHeaderWords /* k_order_scatter: blocks that have finished (behind both copies) */, kHistCounts = 2 * kHistHeaderWords + 8;
constexpr size_t kHistSampleFloats = (size_t)kHistSlots * kHistSamples * 8;  // one copy of the remembered sample rays (they are doubled like the header)
// Round 5: nothing runs in front of a launch to prepare its claim order and nothing behind it.  order_select (below) is evaluated by every
// wave of the launch itself; wave 0 of workgroup 0 also writes the next state (order_commit) into the header's other copy;
// the one thing left to separate dispatches is turning a finished RECORDING into an order (k_order_count / k_order_scatter, rc_traverse.hip),
// which the host enqueues only in front of launches that may follow a recording -- one launch in eight of a repeating batch.
// RECORDING costs: every reporting ray is an atomic whose acknowledgement the wave's next s_waitcnt vmcnt waits for along with its node
// fetches -- 25-30 us of a 0.37 ms launch (profiles/r04_cost_order_recording.txt: the same learned order WITHOUT recording traces the
// shadow batch at 5.95 instead of 5.55 Grays/s).  So a batch slot records its launches 2-4 (the reporting threshold needs two rounds to
// settle; the FIRST launch of a batch never records: batches that do not come back pay nothing) and then one launch in kHistRecordEvery;
// the launches in between reuse the slot's order as it stands.
constexpr uint32_t kHistRecordEvery = 8;
// Wave-uniform: the next claim of this wave's shard, or false when the shard has run dry.  `wave_id` must be the same in all lanes.
// (A part that lies beyond the end of the batch -- in the last, incomplete chunk -- comes back empty; the caller simply claims again.)
__device__ inline bool rc_claim_chunk(const RcClaim& c, const uint32_t* order, uint32_t wave_id, int lane, uint64_t n_items, unsigned long long& pool_next,
                                      unsigned long long& pool_end) {
    const uint32_t n = 1u << c.shard_shift;
    const uint32_t my_shard = __builtin_amdgcn_readfirstlane(wave_id) & (n - 1u);  // wave-uniform (keeps the claim in scalar registers); neighbouring waves use different counters
    uint32_t* const counter = c.counters + my_shard * kShardStrideWords;
    uint32_t got = 0;
    if (lane == 0) got = atomicAdd(counter, 1u);
    const uint32_t cs = __builtin_amdgcn_readfirstlane(got);
    // round cs of the shards takes claims cs * n .. cs * n + n - 1, rotated per round; this shard's share of the claims:
    const uint32_t full_rounds = c.n_chunks >> c.shard_shift, rem = c.n_chunks & (n - 1u);
    const uint32_t my_chunks = full_rounds + ((((my_shard + full_rounds * 5u) & (n - 1u)) < rem) ? 1u : 0u);
    if (cs >= my_chunks) {  // dry.  The last of this shard's waves to find it so -- nobody touches the counter after it -- zeroes it for the next launch
        const uint32_t my_waves = (c.total_waves >> c.shard_shift) + (my_shard < (c.total_waves & (n - 1u)) ? 1u : 0u);
        if (cs + 1u == my_chunks + my_waves && lane == 0) atomicExch(counter, 0u);
        return false;
    }
    const uint32_t v = (cs << c.shard_shift) + ((my_shard + cs * 5u) & (n - 1u));  // claim index; everything below is scalar arithmetic
    uint32_t pos, part = 0u, shift = 0u;
    if (v < c.g1) pos = v;
    else if (v < c.g2) { const uint32_t u = v - c.g1; pos = c.c1 + (u >> 1); part = u & 1u; shift = 1u; }
    else if (v < c.g3) { const uint32_t u = v - c.g2; pos = c.c2 + (u >> 2); part = u & 3u; shift = 2u; }
    else { const uint32_t u = v - c.g3; pos = c.c3 + (u >> 3); part = u & 7u; shift = 3u; }
    const uint32_t chunk = order ? __builtin_amdgcn_readfirstlane(order[pos]) : pos;
    const uint32_t size = c.pool >> shift;
    pool_next = (unsigned long long)chunk * c.pool + (unsigned long long)part * size;
    pool_end = pool_next + size;
    if (pool_end > n_items) pool_end = n_items;
    if (pool_next > pool_end) pool_next = pool_end;
    return true;
}

// Which batch is this launch, and what does it do with the batch's slot?  A pure function of the history's header, the remembered sample
// rays and the launch's own: every wave evaluates it (64 lanes, one sample ray each) and gets the same answer.  The closest slot below the
// threshold is the batch's -- to continue if the launch repeats the slot's batch, to start over in otherwise; none: the least recently used
// slot is given to it.  A launch that starts a slot is "fresh": natural order, nothing recorded.  A slot records in
// its launches 2-4 (never the first: a batch that does not come back pays nothing; the reporting threshold needs two rounds to settle) and
// then when the host's cadence says so (one launch in eight).
struct OrderDecision {
    uint32_t sel, fresh, exact, gen, record, valid, life_thr, paused;
};
// A shape whose batches never repeat learns nothing and should pay nothing: after kGiveUpAfter consecutive launches that were not repeats
// the shape's next kGiveUpFor launches go out in natural order without looking at their rays, then the shape is tried again.  Counted and
// decided here, on the device (rounds 4-5 had the host do it from a pinned word, which a caller that enqueues ahead of the device sees late).
constexpr uint32_t kGiveUpAfter = 8, kGiveUpFor = 64;
__device__ inline OrderDecision order_select(const RcClaim& c, int lane, float r[8]) {
    const uint32_t* hist = c.hist + c.parity * kHistHeaderWords;
    const float* samples = c.samples + c.parity * kHistSampleFloats;
    if (hist[kHistSkipLeft] != 0u) {  // (a header word: the same answer in every wave)
        OrderDecision o;
        o.sel = 0u; o.fresh = 1u; o.exact = 0u; o.gen = 0u; o.record = 0u; o.valid = 0u; o.life_thr = 0xFFFFFFFFu; o.paused = 1u;
        for (int k = 0; k < 8; ++k) r[k] = 0.f;
        return o;
    }
    if (c.sample_rays) {
        uint64_t idx = (uint64_t)lane * c.n_sample / (uint64_t)kHistSamples + c.n_sample / (2u * kHistSamples);
        if (idx >= c.n_sample) idx = c.n_sample - 1;
        const float4* q = reinterpret_cast<const float4*>(c.sample_rays + idx);
        const float4 a = q[0], b = q[1];
        r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    } else {
        for (int k = 0; k < 8; ++k) r[k] = c.sample_host[k];
    }
    // (Loading all four slots' samples and header words up front -- one memory round trip instead of two or three -- was measured 0.7 %
    // SLOWER on C3: the launch's first claims wait behind 8 more vector loads per lane of every wave.)
    int best = -1;
    float best_d = 0.02f;  // mean over the samples of |dd|^2 / |d|^2 + |do|^2 / diagonal^2: ~0.1 rad of rotation, or a tenth of the scene of travel
    for (int k = 0; k < kHistSlots; ++k) {
        if (hist[kHistStamp + k] == 0u) continue;
        const float* sp = samples + ((size_t)k * kHistSamples + lane) * 8;
        const float ox = r[0] - sp[0], oy = r[1] - sp[1], oz = r[2] - sp[2], dx = r[4] - sp[4], dy = r[5] - sp[5], dz = r[6] - sp[6];
        const float na = r[4] * r[4] + r[5] * r[5] + r[6] * r[6], nb = sp[4] * sp[4] + sp[5] * sp[5] + sp[6] * sp[6];
        float d = (dx * dx + dy * dy + dz * dz) / fmaxf(fmaxf(na, nb), 1e-30f) + (ox * ox + oy * oy + oz * oz) * c.inv_l2;
        for (int m = 32; m > 0; m >>= 1) d += __shfl_xor(d, m);  // (a NaN anywhere: the sum is NaN and the comparison fails -- no match)
        d *= 1.0f / kHistSamples;
        if (d < best_d) { best_d = d; best = k; }
    }
    OrderDecision o;
    // Only a REPEAT -- the same sample rays bit for bit -- continues a slot's history.  A batch that merely resembles a remembered one (a
    // camera that moves every frame) starts over IN that slot (it replaces what it resembles instead of evicting somebody else): natural
    // order, nothing recorded, exactly like a batch never seen.  An order learned from similar rays was measured slower than natural order
    // (docs/EXPERIMENTS.md), and deciding this here rather than on the host makes it hold for a caller that enqueues far ahead of the device.
    o.exact = (best >= 0 && best_d == 0.0f) ? 1u : 0u;
    o.fresh = o.exact ? 0u : 1u;
    int sel = best;
    if (best < 0) {  // an empty slot, else the least recently used one
        uint32_t oldest = 0xFFFFFFFFu;
        for (int k = 0; k < kHistSlots; ++k) { const uint32_t st = hist[kHistStamp + k]; if (st < oldest) { oldest = st; sel = k; } }
    }
    o.sel = (uint32_t)__builtin_amdgcn_readfirstlane(sel);
    o.gen = o.fresh ? 1u : hist[kHistGen + o.sel] + 1u;
    o.record = ((o.gen >= 2u && o.gen <= 4u) || (c.want_record && o.gen >= 5u)) ? 1u : 0u;
    o.valid = (!o.fresh && hist[kHistHasOrder + o.sel] != 0u) ? 1u : 0u;
    o.life_thr = o.record ? hist[kHistScale + 4u * o.sel + 2u] : 0xFFFFFFFFu;
    o.paused = 0u;
    return o;
}
// The header after the launch, written into the OTHER copy by one wave (wave 0 of workgroup 0, at the start of the launch: it has just made
// the decision, `r` = this lane's sample ray of the launch; nobody reads that copy before the shape's next launch).
__device__ inline void order_commit(const RcClaim& c, const OrderDecision& o, int lane, const float r[8]) {
    const uint32_t* old_h = c.hist + c.parity * kHistHeaderWords;
    uint32_t* new_h = c.hist + (c.parity ^ 1u) * kHistHeaderWords;
    const float* old_s = c.samples + c.parity * kHistSampleFloats;
    float* new_s = c.samples + (c.parity ^ 1u) * kHistSampleFloats;
    // lane i carries header word i from the old copy to the new one, changed on the way where this launch changes it
    uint32_t w = lane < kHistHeaderWords ? old_h[lane] : 0u;
    auto put = [&](int idx, uint32_t v) { if (lane == idx) w = v; };
    for (int k = 0; k < kHistSlots; ++k) {  // the remembered samples: the launch's own in its slot, the others as they were
        const float4* from = reinterpret_cast<const float4*>(old_s + ((size_t)k * kHistSamples + lane) * 8);
        float4* to = reinterpret_cast<float4*>(new_s + ((size_t)k * kHistSamples + lane) * 8);
        const bool mine = !o.paused && (uint32_t)k == o.sel;
        to[0] = mine ? make_float4(r[0], r[1], r[2], r[3]) : from[0];
        to[1] = mine ? make_float4(r[4], r[5], r[6], r[7]) : from[1];
    }
    if (o.paused) {  // one launch of the pause gone; nothing else moves (the launch clock stands still)
        const uint32_t left = old_h[kHistSkipLeft] - 1u;
        put(kHistSkipLeft, left);
        if (lane < kHistHeaderWords) new_h[lane] = w;
        if (lane == 0 && c.host_streak) __hip_atomic_store(c.host_streak + 2, left | (c.host_gen << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    const uint32_t clock = old_h[kHistClock] + 1u;
    put(kHistClock, clock);
    put(kHistStamp + (int)o.sel, clock);
    put(kHistGen + (int)o.sel, o.gen);
    const int sc = kHistScale + 4 * (int)o.sel;  // [0], [1]: (threshold, top) of the slot's latest recording; [2], [3]: of its next one (k_order_scatter)
    if (o.fresh) { put(sc, c.init_thr); put(sc + 2, c.init_thr); put(sc + 1, c.init_thr + 8u); put(sc + 3, c.init_thr + 8u); put(kHistPending + (int)o.sel, 0u); put(kHistHasOrder + (int)o.sel, 0u); }
    else if (o.record) { put(sc, old_h[sc + 2]); put(sc + 1, old_h[sc + 3]); }  // the rebuild classes the costs with the scale they were recorded under
    if (o.record) put(kHistPending + (int)o.sel, 1u);
    put(kHistSel, o.sel); put(kHistFresh, o.fresh); put(kHistOrderValid, o.valid); put(kHistLifeThr, o.life_thr); put(kHistRecorded, o.record);
    // The run of launches that were not REPEATS of a remembered batch -- a path tracer's bounce rays (never matched), a camera that moves
    // every frame (matched, never identical) -- and the pause it leads to.  The host reads the pinned copies to decide about the rebuild
    // kernels: [0] the run (a new batch is starting: its launches 2-4 will record), [1] a recording waits, [2] the pause.
    uint32_t streak = o.exact ? 0u : old_h[kHistFreshStreak] + 1u;
    uint32_t skip = 0u;
    if (streak >= kGiveUpAfter) { streak = 0u; skip = kGiveUpFor; }
    put(kHistFreshStreak, streak);
    put(kHistSkipLeft, skip);
    if (lane < kHistHeaderWords) new_h[lane] = w;
    const bool is_pending = lane >= kHistPending && lane < kHistPending + kHistSlots && w != 0u;
    const uint32_t any_pending = __ballot(is_pending) ? 1u : 0u;  // ... whether a recording waits for the rebuild kernels
    if (lane == 0 && c.host_streak) {
        __hip_atomic_store(c.host_streak, streak, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(c.host_streak + 1, any_pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(c.host_streak + 2, skip | (c.host_gen << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (skip <= kGiveUpFor = 64: 8 bits)
    }
}

struct TraceArgs {
    SceneView v;
    const RcRay* rays;
    RcHit* hits;
    uint64_t n_rays;
    RcClaim claim;                     // persistent kernels: how waves claim ray chunks
    int refill;                        // persistent kernel: refill when this many lanes are idle
    int sched_thr;                     // scheduled kernel: run a leaf/entry batch once this many lanes wait for it
    unsigned long long* stats;         // optional instrumentation (dev builds), else nullptr
    uint32_t blas_k = 0, lds_blas_base = 0, tlas_k = 0;
    unsigned long long* timeline = nullptr;  // dev (option "timeline_ptr"): per-wave event record of kernel 5
};

// Address-space-qualified pointers keep the two halves of the stack on their own instruction paths
// (ds_read/ds_write_b32 for LDS, global_load/store for the spill area); with generic pointers the compiler
// if-converts push/pop into a pointer select + flat_load, which sends every pop through the texture path.
typedef uint32_t __attribute__((address_space(3))) lds_u32;
typedef uint32_t __attribute__((address_space(1))) glb_u32;

template <int LDS_N, int BLOCK = kBlock>
struct LaneStackT {
    lds_u32* lds;        // &lds_stack[threadIdx.x]
    glb_u32* ovf;        // &overflow[global thread id]
    uint32_t ovf_stride;
    uint32_t* status;
    __device__ inline LaneStackT(uint32_t* lds_base, uint32_t* ovf_base, uint32_t stride, uint32_t* st)
        : lds((lds_u32*)lds_base), ovf((glb_u32*)ovf_base), ovf_stride(stride), status(st) {}
    __device__ inline void push(int& sp, uint32_t v) {
        if (__builtin_expect(sp < LDS_N, 1)) lds[sp * BLOCK] = v;
        else if (sp < kTotalStack) ovf[(size_t)(sp - LDS_N) * ovf_stride] = v;
        else { *status = 1u; return; }
        ++sp;
    }
    __device__ inline void store(int pos, uint32_t v) {  // write entry `pos` without moving sp (multi-push callers)
        if (__builtin_expect(pos < LDS_N, 1)) lds[pos * BLOCK] = v;
        else if (pos < kTotalStack) ovf[(size_t)(pos - LDS_N) * ovf_stride] = v;
        else *status = 1u;
    }
    __device__ inline uint32_t pop(int& sp) {
        --sp;
        if (__builtin_expect(sp < LDS_N, 1)) return lds[sp * BLOCK];
        return ovf[(size_t)(sp - LDS_N) * ovf_stride];
    }
};
using LaneStack = LaneStackT<kLdsStack>;

// The same stack addressed by a POINTER to the next free LDS slot instead of an entry index (phased_trace): a push is compare +
// ds_write + add, a pop add + compare + ds_read -- the index form pays a 64-bit multiply-add (v_mad_u64_u32, 5 cycles of VALU issue
// on a kernel bound by exactly that) for every address.  Entries beyond the LDS depth live in the global spill area as before; the
// entry index is only reconstructed on that rare path.
typedef uint16_t __attribute__((address_space(3))) lds_u16;
// E = the type of an LDS entry: uint32_t, or uint16_t for scenes whose trees all have fewer than 65 534 nodes (STACK16 kernels: node values,
// INVALID and the sentinel are then 16-bit numbers throughout the kernel -- 0xFFFF and 0xFFFE are the low halves of the 32-bit ones -- and
// the LDS the stacks give up holds more of the tree; entries beyond the LDS depth spill to 32-bit words as before)
template <int LDS_N, int BLOCK, typename E = uint32_t>
struct LaneStackP {
    typedef E __attribute__((address_space(3))) lds_e;
    typedef lds_e* pos_t;
    lds_e* base;         // &lds_stack[threadIdx.x]: entry k at base + k * BLOCK
    lds_e* limit;        // base + LDS_N * BLOCK
    glb_u32* ovf;
    uint32_t ovf_stride;
    uint32_t* status;
    __device__ inline LaneStackP(E* lds_base, uint32_t* ovf_base, uint32_t stride, uint32_t* st)
        : base((lds_e*)lds_base), limit((lds_e*)lds_base + LDS_N * BLOCK), ovf((glb_u32*)ovf_base), ovf_stride(stride), status(st) {}
    __device__ inline pos_t empty() const { return base; }
    __device__ inline int depth(pos_t top) const { return (int)(top - base) / BLOCK; }
    __device__ inline void push(pos_t& top, uint32_t v) {
        if (__builtin_expect(top < limit, 1)) *top = (E)v;
        else {
            const int k = depth(top);
            if (k < kTotalStack) ovf[(size_t)(k - LDS_N) * ovf_stride] = v;
            else { *status = 1u; return; }
        }
        top += BLOCK;
    }
    __device__ inline uint32_t pop(pos_t& top) {
        top -= BLOCK;
        if (__builtin_expect(top < limit, 1)) return *top;
        return ovf[(size_t)(depth(top) - LDS_N) * ovf_stride];
    }
};

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

// Box tests under Julia's NaN-propagating min / max (fast_intersect_bbox :1841-1859): a NaN anywhere in a slab test -- a NaN
// (-o)*inv component (NaN origin or direction, inf*0), a NaN t_min, a NaN closest t (a NaN-t "hit") or a NaN
// node box (NaN vertices) -- makes min_t or max_t NaN and the test fail.  jl_minf / jl_maxf compile to gfx950's v_minimum3_f32 /
// v_maximum3_f32, which have exactly that semantics (v_min_f32 / v_max_f32 would drop the NaN), so the kernels need no special
// cases.  Triangle tests use plain comparisons on both sides.

// Per-ray traversal state.
struct RayState {
    float3_ wo, wd, winv;           // world ray (direction sanitised by check_direction), safe_invdir(world d)
    float3_ o, d, inv, ox;          // current-level ray, 1/d, (-o)*inv
    float tmin, closest_t;
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
    float tmaxx = jl_maxf(fx, nx), tmaxy = jl_maxf(fy, ny), tmaxz = jl_maxf(fz, nz);
    float tminx = jl_minf(fx, nx), tminy = jl_minf(fy, ny), tminz = jl_minf(fz, nz);
    max_t = jl_minf(jl_minf(jl_minf(tmaxx, tmaxy), tmaxz), s.closest_t);
    min_t = jl_maxf(jl_maxf(jl_maxf(tminx, tminy), tminz), s.tmin);
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
        // bottom-level leaf: fast_intersect_triangle (:1756-1797) on the leaf record (rc_pack_leaf: v0 and the edges e1 = v1 - v0, e2 = v2 - v0,
        // the subtractions of :1766-1767 done once at pack time)
        float3_ v0 = mk3(nd.a.w, nd.a.x, nd.a.y), e1 = mk3(nd.b.y, nd.b.z, nd.b.x), e2 = mk3(nd.c.y, nd.c.z, nd.c.x);
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


typedef float v2f __attribute__((ext_vector_type(2)));

// Ray source / hit sink of the plain trace entry points: RTRay array in, RTHitResult array out (:2010-2023).
struct ArraySource {
    const RcRay* rays;
    __device__ inline RcRay operator()(uint64_t i) const {
        const float4* q = reinterpret_cast<const float4*>(rays + i);
        float4 a = q[0], b = q[1];
        return RcRay{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    }
    // Round 6: a wave that has just claimed the range [first, end) touches one dword of every 64-byte line of it (lane L: rays first + 2 L and
    // first + 2 L + 1; 64 lanes = 128 rays = a whole chunk), in the same memory round trip as its first ray loads.  A batch traced for the first
    // time is read from HBM (through cold TLB entries), and a chunk is consumed in 3-4 refills of 20-64 rays -- each of which used to pay that
    // round trip; now the later ones find their lines in L2.  The value is never used (the caller only keeps it alive up to the ray loads' wait).
#ifdef RC_NO_RAY_PREFETCH   // dev: A/B builds (tools/ab_fresh.py)
    static constexpr bool kPrefetch = false;
#else
    static constexpr bool kPrefetch = true;
#endif
    __device__ inline uint32_t prefetch(unsigned long long first, unsigned long long end, int lane) const {
        const unsigned long long i = first + 2ull * (unsigned long long)lane;
        return i < end ? *reinterpret_cast<const uint32_t*>(rays + i) : 0u;
    }
};
struct HitWriter {
    const RcInstRec* inst;
    RcHit* hits;
    __device__ inline void operator()(uint64_t i, bool hit, float t, float u, float v, uint32_t prim, int instance) const {
        uint4 w0, w1;
        if (hit) {
            const uint4 m3 = *(reinterpret_cast<const uint4*>(inst + instance) + 3);
            w0 = make_uint4(1u, __float_as_uint(t), m3.y + prim - 1u, m3.z);
            w1 = make_uint4(__float_as_uint(u), __float_as_uint(v), (uint32_t)instance, 0u);
        } else {
            w0 = make_uint4(0u, 0u, RC_INVALID_NODE, 0u);
            w1 = make_uint4(0u, 0u, RC_INVALID_NODE, 0u);
        }
        uint4* out = reinterpret_cast<uint4*>(hits + i);
        out[0] = w0;
        out[1] = w1;
    }
};

// ---- persistent waves, phase-structured ("while-while") scheduling: the core of trace kernel 3 and of the drivers ----
// Same idea as kernel 2 -- run one block at a time for the lanes that need it -- but with a fixed phase order
// instead of a per-iteration vote: an inner loop walks interior nodes for as long as at least `int_thr`
// lanes have one pending (always at least once), then ONE pass each over the BLAS-leaf lanes, the level-switch
// lanes (instance entry / return to top level) and the finished lanes (write-out + refill).  The inner loop
// only touches {node, sp} so the compiler keeps the ray registers untouched across it.  Per-lane order of
// visits, tests, pushes and pops is the reference's; only the interleaving between lanes differs.
// Raw buffer loads (buffer_load_dwordx{1,2,4} ... offen): one wave-uniform descriptor per array in SGPRs and a 32-bit
// per-lane byte offset, so a node address is one shift instead of 64-bit pointer arithmetic, every fetch has exactly the
// width asked for, and an out-of-range offset reads 0 instead of faulting.  Arrays must stay below 4 GiB (64 M nodes).
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
// `soff` goes into the instruction's scalar-offset operand: the 16/32/48-byte pieces of one record share a single per-lane
// byte offset register instead of costing a v_add each.
__device__ inline float4 buf_f4(__amdgpu_buffer_rsrc_t r, uint32_t off, int soff = 0) {
    u4v v = __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ inline float2 buf_f2(__amdgpu_buffer_rsrc_t r, uint32_t off, int soff = 0) {
    u2v v = __builtin_amdgcn_raw_buffer_load_b64(r, off, soff, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}

// `src(i)` produces work item i's ray (a ray array, or a generator: grid rays, view-factor samples); `sink(i, hit, t, u,
// v, prim, inst)` consumes the result (prim = 1-based leaf primitive index inside the instance's BLAS as the reference
// keeps it, inst = 0-based instance, -1 on a miss).
struct PersistArgs {
    uint64_t n_items;                  // work items = rays
    RcClaim claim;                     // how waves claim chunks of items
    int refill;                        // refill when this many lanes are free
    int int_thr;                       // leave the interior loop when fewer lanes than this have an interior node pending
    unsigned long long* stats;
    uint32_t blas_k = 0;               // TLAS_LDS kernels: BLAS nodes 1..blas_k are staged in the planes at entry lds_blas_base + node - 1
    uint32_t lds_blas_base = 0;
    uint32_t tlas_k = 0;               // PARTIAL_LDS kernels: TLAS nodes 1..tlas_k are staged at entry node - 1 (the rest comes from memory)
    unsigned long long* timeline = nullptr;  // TIMELINE builds (dev): 8 words per wave, see the end of phased_trace
};

// TLAS_LDS / INST_LDS: the block has staged the top level in LDS before the call (LdsTop below; layout and sizes in rc_internal.h):
// packed nodes as seven float2 planes (dword pairs 0-1, 2-3, ... 12-13 of each node; a plane read is one ds_read_b64 with lane
// addresses 8 bytes apart per node), the TLAS leaf -> instance table, and the instance records as float2 planes.  TLAS-level visits
// and instance entries then never touch the vector-memory path, which is what bounds the kernel.
// kTlasLdsNodes, kTlasLdsInst, kLdsPlaneNodes: rc_internal.h (the TLAS build needs them too)

// Shape of the two-workgroups-per-CU LDS kernels (trace kernel 5 and the LDS variants of the drivers): 768 threads, 16-entry LDS lane
// stacks (48 KiB) + node planes (17 KiB) + leaf table (1 KiB) + instance planes (14 KiB) = 79.95 KiB per workgroup.
constexpr int kMidBlock = 768, kMidStack = 16;
constexpr size_t kNodePlaneBytes = (size_t)7 * kLdsPlaneNodes * sizeof(float2);
constexpr size_t kLeafTableBytes = (size_t)kTlasLdsInst * sizeof(uint32_t);
constexpr size_t kInstPlaneBytes = (size_t)7 * kTlasLdsInst * sizeof(float2);
constexpr size_t kLdsTopBytes = kNodePlaneBytes + kLeafTableBytes + kInstPlaneBytes;
constexpr size_t kMidLdsBytes = (size_t)kMidStack * kMidBlock * 4 + kLdsTopBytes;
// STACK16 shape (scenes whose trees all have fewer than 65 534 nodes: 16-bit lane stacks, 24 KiB instead of 48): kLdsPlaneNodes16 node-plane entries
constexpr size_t kNodePlaneBytes16 = (size_t)7 * kLdsPlaneNodes16 * sizeof(float2);
constexpr size_t kMidLdsBytes16 = (size_t)kMidStack * kMidBlock * 2 + kNodePlaneBytes16 + kLeafTableBytes + kInstPlaneBytes;
#if RC_LDS_PLANES16 == 748
static_assert(kMidLdsBytes <= 81920 && kMidLdsBytes16 <= 81920, "two workgroups per CU share 160 KiB of LDS");
#else   // dev variant (tools/lds_bound_probe.py): one workgroup per CU
static_assert(kMidLdsBytes <= 81920 && kMidLdsBytes16 <= 163840, "one workgroup per CU has 160 KiB of LDS");
#endif
struct LdsTop {
    float2* tl;    // node planes: plane p of entry e at tl[p * (plane entries) + e]
    uint32_t* lt;  // lt[j] = instance index of TLAS leaf n - 1 + j + 1 (node index n + j)
    float2* il;    // instance planes: plane p of instance i at il[p * kTlasLdsInst + i]; planes 0-5 = inverse transform, 6 = (nodes offset, leaf count)
    __device__ inline explicit LdsTop(unsigned char* base, size_t node_plane_bytes = kNodePlaneBytes)
        : tl(reinterpret_cast<float2*>(base)), lt(reinterpret_cast<uint32_t*>(base + node_plane_bytes)),
          il(reinterpret_cast<float2*>(base + node_plane_bytes + kLeafTableBytes)) {}
    __device__ inline LdsTop() : tl(nullptr), lt(nullptr), il(nullptr) {}
};
// Fill it (all threads of the workgroup; caller synchronises).  n_inst <= kTlasLdsInst, blas_k <= kLdsPlaneNodes - (n_inst - 1).
template <int BLOCK, int PLANE_NODES = kLdsPlaneNodes>
__device__ inline void stage_lds_top(const LdsTop& t, const SceneView& v, uint32_t blas_k, uint32_t lds_blas_base) {
    const RcNode* tnodes = v.blas_nodes + v.tlas_off;
    const uint32_t n_inst = (v.n_tlas_nodes + 1u) >> 1;
    for (uint32_t i = threadIdx.x; i < (n_inst - 1u) * 7u; i += BLOCK) {  // TLAS interior nodes 1..n-1
        const uint32_t nd = i / 7u, p = i % 7u;
        t.tl[p * PLANE_NODES + nd] = reinterpret_cast<const float2*>(tnodes + nd)[p];
    }
    for (uint32_t i = threadIdx.x; i < blas_k * 7u; i += BLOCK) {  // single-BLAS scene: its top internal nodes sit first in the traversal copy
        const uint32_t nd = i / 7u, p = i % 7u;
        t.tl[p * PLANE_NODES + lds_blas_base + nd] = reinterpret_cast<const float2*>(v.blas_nodes + nd)[p];
    }
    for (uint32_t j = threadIdx.x; j < n_inst; j += BLOCK)  // child1 word (dword 13) of leaf node n + j
        t.lt[j] = reinterpret_cast<const uint32_t*>(tnodes + (n_inst - 1u + j))[13];
    for (uint32_t i = threadIdx.x; i < v.n_inst * 7u; i += BLOCK) {
        const uint32_t in = i / 7u, p = i % 7u;
        const uint32_t* rec = reinterpret_cast<const uint32_t*>(v.inst + in);
        const uint32_t a = p < 6u ? rec[2u * p] : rec[12], b = p < 6u ? rec[2u * p + 1u] : rec[15];
        t.il[p * kTlasLdsInst + in] = make_float2(__uint_as_float(a), __uint_as_float(b));
    }
}

// PARTIAL_LDS (top levels too large for TLAS_LDS): the planes (kPartialPlaneNodes entries) hold only the breadth-first top of the TLAS
// -- rc_build_tlas renumbers the TLAS's internal nodes in the traversal copy the way it does for a single BLAS -- and of a single BLAS;
// TLAS leaves, instance records and everything below the tops are read from memory as in the plain kernel.
constexpr size_t kPartialPlaneBytes = (size_t)7 * kPartialPlaneNodes * sizeof(float2);
constexpr size_t kPartialLdsBytes = (size_t)kMidStack * kMidBlock * 4 + kPartialPlaneBytes;
constexpr size_t kPartialPlaneBytes16 = (size_t)7 * kPartialPlaneNodes16 * sizeof(float2);
constexpr size_t kPartialLdsBytes16 = (size_t)kMidStack * kMidBlock * 2 + kPartialPlaneBytes16;
static_assert(kPartialLdsBytes <= 81920 && kPartialLdsBytes16 <= 81920, "two workgroups per CU share 160 KiB of LDS");
template <int BLOCK, int PLANE_NODES = kPartialPlaneNodes>
__device__ inline void stage_partial_top(float2* tl, const SceneView& v, uint32_t tlas_k, uint32_t blas_k, uint32_t lds_blas_base) {
    const RcNode* tnodes = v.blas_nodes + v.tlas_off;
    for (uint32_t i = threadIdx.x; i < tlas_k * 7u; i += BLOCK) {
        const uint32_t nd = i / 7u, p = i % 7u;
        tl[p * PLANE_NODES + nd] = reinterpret_cast<const float2*>(tnodes + nd)[p];
    }
    for (uint32_t i = threadIdx.x; i < blas_k * 7u; i += BLOCK) {
        const uint32_t nd = i / 7u, p = i % 7u;
        tl[p * PLANE_NODES + lds_blas_base + nd] = reinterpret_cast<const float2*>(v.blas_nodes + nd)[p];
    }
}

// safe_invdir (src/instanced-bvh.jl:1742-1748) of three components with the division written out.  hipcc's correctly rounded 1.0f / x is
// v_div_scale x 2, v_rcp, six fma / mul steps, v_div_fmas, v_div_fixup; for a denominator in [1e-5, 2^60) (the clamp guarantees the lower
// end) neither v_div_scale scales nor v_div_fixup changes anything and 1.0f * r is r, so the same rcp + six fma give the same bits in 7
// instead of 11 VALU instructions -- an instance entry does three of these.  Any lane outside that range (huge, inf or NaN direction
// components) sends the whole wave down the generic division.
__device__ inline float rcp_rn_normal(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = r;
    const float e2 = __builtin_fmaf(-d, q, 1.0f);
    q = __builtin_fmaf(e2, r, q);
    const float e3 = __builtin_fmaf(-d, q, 1.0f);
    return __builtin_fmaf(e3, r, q);
}
__device__ inline float3_ safe_inv3(const float3_ d) {
    const float ooeps = 1.0e-5f;
    const float ax = __builtin_fabsf(d.x), ay = __builtin_fabsf(d.y), az = __builtin_fabsf(d.z);
    const bool tame = __builtin_fmaxf(__builtin_fmaxf(ax, ay), az) < 0x1p60f && ax == ax && ay == ay && az == az;
    if (__builtin_expect(__ballot(!tame) == 0ull, 1)) {
        const float cx = ax > ooeps ? d.x : __builtin_copysignf(ooeps, d.x), cy = ay > ooeps ? d.y : __builtin_copysignf(ooeps, d.y),
                    cz = az > ooeps ? d.z : __builtin_copysignf(ooeps, d.z);
        return mk3(rcp_rn_normal(cx), rcp_rn_normal(cy), rcp_rn_normal(cz));
    }
    return mk3(safe_inv1(d.x), safe_inv1(d.y), safe_inv1(d.z));
}

// Analysis builds (tools/isa_mix.py, -DRC_PHASE_MARKERS): comment lines in the generated assembly that delimit the phases, so that the
// static VALU opcode histogram of each phase can be read off the ISA.  Not compiled into the product.
#ifdef RC_PHASE_MARKERS
#define RC_MARK(name) asm volatile("; RC_MARK " name)
#else
#define RC_MARK(name) ((void)0)
#endif

template <bool ANY, int LDS_N, bool STATS, class Source, class Sink, int BLOCK = kBlock, bool TLAS_LDS = false, bool INST_LDS = TLAS_LDS, bool PARTIAL_LDS = false,
          bool TIMELINE = false, bool STACK16 = false>
__device__ inline void phased_trace(const SceneView& av, const PersistArgs& a, typename std::conditional<STACK16, uint16_t, uint32_t>::type* lds_stack, const Source& src,
                                    const Sink& sink, const LdsTop top = LdsTop()) {
    typedef typename std::conditional<STACK16, uint16_t, uint32_t>::type stack_entry_t;
    // node values that are not nodes: the 32-bit INVALID / sentinel of the reference, or their low halves where everything fits 16 bits
    constexpr uint32_t kInv = STACK16 ? 0xFFFFu : RC_INVALID_NODE, kSent = STACK16 ? 0xFFFEu : RC_TOP_LEVEL_SENTINEL;
    const float2* const tl = top.tl;
    const uint32_t* const lt = top.lt;
    const float2* const il = top.il;
    const uint32_t gtid = blockIdx.x * BLOCK + threadIdx.x;
    LaneStackP<LDS_N, BLOCK, stack_entry_t> st(lds_stack + threadIdx.x, av.overflow + gtid, av.total_threads, av.status);
    const int lane = threadIdx.x & 63;
    if (av.n_tlas_nodes == 0) {  // empty TLAS: every ray misses (test/test_tlas_stress.jl:808-831)
        for (uint64_t i = gtid; i < a.n_items; i += av.total_threads) sink(i, false, 0.0f, 0.0f, 0.0f, RC_INVALID_NODE, -1);
        return;
    }
    const uint32_t n_instances = (av.n_tlas_nodes + 1u) >> 1;
    const uint32_t tlas_off = av.tlas_off;
    // the node array addressed by 1-based node index: the base sits one record before element 0 (never dereferenced: index 0 is not a node)
    const __amdgpu_buffer_rsrc_t nrs1 = make_rsrc(reinterpret_cast<const char*>(av.blas_nodes) - 64, (av.n_nodes_total + 1u) * 64u);
    const __amdgpu_buffer_rsrc_t irs = make_rsrc(av.inst, av.n_inst * 64u);
    // entry cull (k_inst_recs, rc_build.hip): per ray 1 / (d . d) -- NaN when the ray is outside the regime the cull's bounds assume, which
    // fails every comparison below --, 1 / |d| and the ray's share of the margin
    const bool cull_on = av.inst_cull != nullptr;
    const __amdgpu_buffer_rsrc_t crs = make_rsrc(av.inst_cull, cull_on ? av.n_inst * 32u : 0u);
    float c_idd = 0.f, c_idl = 0.f, c_ray = 0.f, c_dd = 0.f;
    unsigned long long pool_next = 0, pool_end = 0;
    bool exhausted = false;
    uint64_t my_ray = 0;
    float3_ wo = mk3(0, 0, 0), wd = mk3(0, 0, 0), winv = mk3(0, 0, 0);
    float3_ inv = mk3(0, 0, 0), ox = mk3(0, 0, 0);
    // the instance-local ray, kept as the component pairs the packed triangle test multiplies (rc_pack_leaf): (y, z) and (z, x) of o and d
    v2f oyz = {0.f, 0.f}, ozx = {0.f, 0.f}, dyz = {0.f, 0.f}, dzx = {0.f, 0.f};
    float tmin = 0.f, closest_t = 0.f, hit_u = 0.f, hit_v = 0.f;
    uint32_t closest_prim = RC_INVALID_NODE, cur_off = 0, n_level = 0;
    uint32_t node = kInv;  // INVALID + !live = empty lane; INVALID + live = finished, result pending
    int closest_inst = -1, cur_inst = -1;
    // cost-ordered claiming (RcClaim::cost): a ray's cost = the interior-loop iterations its wave ran while the ray was in flight -- the time
    // the ray occupied its lane, in the unit the launch's tail is made of.  The wave's iteration count lives in a scalar register and every
    // lane remembers the count at which its ray started: no per-iteration vector work.
    uint32_t it_total = 0, start_it = 0;
    // which batch slot of the history this launch belongs to (order_select): its cost array, whether `order` is valid, its threshold
    const uint32_t* claim_order = a.claim.order;
    uint32_t* claim_cost = a.claim.cost;
    uint32_t life_thr = 0xFFFFFFFFu;
    if (a.claim.hist) {
        float smp[8];
        const OrderDecision od = order_select(a.claim, lane, smp);
        const uint32_t sel = od.sel;
        claim_order = __builtin_amdgcn_readfirstlane(od.valid) ? claim_order + (size_t)sel * kHistSlotStride : nullptr;
        life_thr = __builtin_amdgcn_readfirstlane(od.life_thr);  // 0xFFFFFFFF: this launch does not record
        claim_cost = life_thr == 0xFFFFFFFFu ? nullptr : claim_cost + (size_t)sel * kHistSlotStride;
        if (__builtin_amdgcn_readfirstlane(od.fresh) && !__builtin_amdgcn_readfirstlane(od.paused)) {  // the slot's cost array still holds what the evicted batch recorded: nobody reads or records it during a batch's first launch
            uint32_t* stale = a.claim.cost + (size_t)sel * kHistSlotStride;
            const uint64_t n_base = (a.n_items + a.claim.pool - 1u) / a.claim.pool;
            for (uint64_t i = gtid; i < n_base; i += av.total_threads) stale[i] = 0u;
        }
        if (blockIdx.x == 0 && threadIdx.x < 64) order_commit(a.claim, od, lane, smp);  // the header's other copy: read by the shape's next launch
    }
    typename LaneStackP<LDS_N, BLOCK, stack_entry_t>::pos_t sp = st.empty();
    bool live = false;
    unsigned long long st_iter[4] = {0, 0, 0, 0}, st_lane[4] = {0, 0, 0, 0}, st_outer = 0, st_sub[4] = {0, 0, 0, 0}, st_cull = 0;  // st_sub: passes with an exit lane / an entry lane / a result to write / rays to start
    unsigned long long st_t0 = STATS ? wall_clock64() : 0ull, st_tx = 0ull;
    int thr_eff = __builtin_amdgcn_readfirstlane(a.int_thr);  // wave-uniform: keeps the loop-exit compare on the scalar unit
    // TIMELINE (dev, tools/archive/timeline_probe.py): per-wave event times and scalar counts, cheap enough not to move the schedule
    unsigned long long tl_t0 = TIMELINE ? wall_clock64() : 0ull, tl_tx = 0ull, tl_t16 = 0ull, tl_t4 = 0ull;
    uint32_t tl_outer = 0, tl_outer_x = 0, tl_live_x = 0, tl_int = 0, tl_int_x = 0;

    for (;;) {
#if defined(RC_AGE_PRIO)  // dev experiment (VERDICT r4 #2a): a wave that carries a ray older than RC_AGE_PRIO interior iterations issues ahead of its SIMD's other waves
        {
            const bool old_ray = live && (it_total - start_it) > (uint32_t)RC_AGE_PRIO;
            if (__ballot(old_ray)) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
        }
#endif
        // ---- interior phase: intersect_internal_node (:1807-1832) + push far / descend near / pop (:1946-1960, 1991-1993)
        for (;;) {
            RC_MARK("interior_begin");
            const bool is_int = node < n_level;  // sentinels and INVALID are >= 0xFFFFFFFE, never below a leaf threshold
            const int n_int = __popcll(__ballot(is_int));
            if (n_int == 0) break;
            it_total += 1u;
            if (STATS) { st_iter[1] += 1; st_lane[1] += is_int ? 1 : 0; }
            if (TIMELINE) { tl_int += 1; if (tl_tx) tl_int_x += 1; }
            if (is_int) {
#if defined(RC_EXP_VALU)  // dev experiment (tools/probes/build_variants.sh; profiles/r05_bound_probe.txt): RC_EXP_VALU extra slow-class VALU instructions (one dependent chain) per interior pass
                { float dummy = tmin;
#pragma unroll
                  for (int k = 0; k < RC_EXP_VALU; ++k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(dummy) : "v"(closest_t));
                  if (dummy == 1.2345e-30f) tmin = dummy; }
#endif
#if defined(RC_EXP_NOP)   // ... RC_EXP_NOP x 16 idle cycles in the pass
#pragma unroll
                for (int k = 0; k < RC_EXP_NOP; ++k) asm volatile("s_nop 15");
#endif
                float4 na, nb, nc;
                u2v ch;
                constexpr int PS = PARTIAL_LDS ? (STACK16 ? kPartialPlaneNodes16 : kPartialPlaneNodes) : (STACK16 ? kLdsPlaneNodes16 : kLdsPlaneNodes);  // plane stride
                if ((TLAS_LDS && (cur_inst < 0 || node <= a.blas_k)) || (PARTIAL_LDS && node <= (cur_inst < 0 ? a.tlas_k : a.blas_k))) {
                    const float2* q = tl + ((node - 1u) + (cur_inst < 0 ? 0u : a.lds_blas_base));
                    const float2 p0 = q[0], p1 = q[PS], p2 = q[2 * PS], p3 = q[3 * PS], p4 = q[4 * PS], p5 = q[5 * PS], p6 = q[6 * PS];
                    na = make_float4(p0.x, p0.y, p1.x, p1.y); nb = make_float4(p2.x, p2.y, p3.x, p3.y); nc = make_float4(p4.x, p4.y, p5.x, p5.y);
                    ch = u2v{__float_as_uint(p6.x), __float_as_uint(p6.y)};
                } else {
                    const uint32_t off = (cur_off + node) << 6;  // the record of 1-based node index `node`: rsrc base is one record early
                    na = buf_f4(nrs1, off); nb = buf_f4(nrs1, off, 16); nc = buf_f4(nrs1, off, 32);
                    ch = __builtin_amdgcn_raw_buffer_load_b64(nrs1, off, 48, 0);
                }
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
                // intersect_internal_node's INVALID-or-child values, kept as predicates (an interior node has two valid children):
                // traverse0 = h0 ? child0 : INVALID, traverse1 likewise; near first iff t0_min < t1_min && traverse0 valid
                const bool h0 = t0_min <= t0_max, h1 = t1_min <= t1_max;
                const bool first0 = (t0_min < t1_min) & h0;
                const uint32_t near_c = first0 ? ch.x : ch.y, far_c = first0 ? ch.y : ch.x;
                const bool near_ok = first0 | h1, far_ok = h0 & (h1 | !first0);  // = first0 ? h1 : h0 (first0 implies h0), as lane-mask logic
                if (far_ok) st.push(sp, far_c);
                node = near_ok ? near_c : st.pop(sp);
            }
            RC_MARK("interior_end");
            if (n_int < thr_eff) break;  // too few interior lanes left: serve the waiting ones first
        }
        // ---- leaf phase: fast_intersect_triangle (:1756-1797) on BLAS leaves, then pop
        {
            const bool is_leaf = cur_inst >= 0 && node >= n_level && node < kSent;
            if (STATS && __ballot(is_leaf)) { st_iter[2] += 1; st_lane[2] += is_leaf ? 1 : 0; }
            RC_MARK("leaf_begin");
            if (is_leaf) {
                const uint32_t off = (cur_off + node) << 6;
                const float4 qa = buf_f4(nrs1, off), qb = buf_f4(nrs1, off, 16), qc = buf_f4(nrs1, off, 32);
                // rc_pack_leaf: qa = v0 (y, z | z, x), qb = e1 (z, x | y, z), qc = e2 (z, x | y, z).  The statements of :1766-1790 with each
                // cross product's x and y as one packed multiply-multiply-subtract; every component is the same IEEE operation on the same
                // operands as in the scalar form (no contraction), the dot products keep their (a.x b.x + a.y b.y) + a.z b.z order
                const v2f A0 = {qa.x, qa.y}, A1 = {qa.z, qa.w}, B0 = {qb.x, qb.y}, B1 = {qb.z, qb.w}, C0 = {qc.x, qc.y}, C1 = {qc.z, qc.w};
                const float e1x = B0.y, e1y = B1.x, e1z = B0.x, e2x = C0.y, e2y = C1.x, e2z = C0.x, dx = dzx.y, dy = dyz.x, dz = dyz.y;
                const v2f s1xy = dyz * C0 - dzx * C1;          // s1 = d x e2
                const float s1z = dx * e2y - dy * e2x;
                const float det = (s1xy.x * e1x + s1xy.y * e1y) + s1z * e1z;
                const float invd = 1.0f / det;
                const v2f ddyz = oyz - A0, ddzx = ozx - A1;    // dd = o - v0
                const float ddx = ddzx.y, ddy = ddyz.x, ddz = ddyz.y;
                const float u = ((ddx * s1xy.x + ddy * s1xy.y) + ddz * s1z) * invd;
                const v2f s2xy = ddyz * B0 - ddzx * B1;        // s2 = dd x e1
                const float s2z = ddx * e1y - ddy * e1x;
                const float v = ((dx * s2xy.x + dy * s2xy.y) + dz * s2z) * invd;
                const float t = ((e2x * s2xy.x + e2y * s2xy.y) + e2z * s2z) * invd;
                const bool hit = !(u < 0.0f || u > 1.0f) && !(v < 0.0f || (u + v) > 1.0f) && !(t < tmin || t > closest_t);
                closest_prim = hit ? node - n_level + 1u : closest_prim;  // leaf of sorted primitive j sits at n-1+j
                closest_inst = hit ? cur_inst : closest_inst;
                closest_t = hit ? t : closest_t;
                hit_u = hit ? u : hit_u;
                hit_v = hit ? v : hit_v;
                if (ANY && hit) node = kInv;  // :2106-2115
                else node = st.pop(sp);
            }
            RC_MARK("leaf_end");
        }
        // ---- switch phase: return to the top level (:1996-2006) or enter an instance (:1961-1977)
        bool was_skipped = false;
        {
            const bool is_exit = node == kSent;
            const bool is_entry = cur_inst < 0 && node >= n_level && node < kSent;
            if (STATS && __ballot(is_exit || is_entry)) { st_iter[3] += 1; st_lane[3] += (is_exit || is_entry) ? 1 : 0; }
            if (STATS) { if (__ballot(is_exit)) st_sub[0] += 1; if (__ballot(is_entry)) st_sub[1] += 1; }
            RC_MARK("switch_begin");
            if (is_exit) {
                node = st.pop(sp);
                cur_inst = -1;
                cur_off = tlas_off; n_level = n_instances;
                // back to the world ray (:2003-2005).  Only inv and (-o) * inv are restored: the top level has no triangle tests, so the
                // ray's o and d are not read again before the next instance entry overwrites them from wo / wd
                inv = winv;
                ox = mk3(-wo.x * inv.x, -wo.y * inv.y, -wo.z * inv.z);
                RC_MARK("switch_end");
            } else if (is_entry) {
                RC_MARK("entry_begin");
                float4 m0, m1, m2;
                u4v m3;
                if (TLAS_LDS) cur_inst = (int)lt[node - n_level];  // leaf of sorted instance j is node n - 1 + j; its child1 word
                else cur_inst = (int)__builtin_amdgcn_raw_buffer_load_b32(nrs1, (cur_off + node) << 6, 52, 0);
                bool skip = false;
                if (cull_on) {  // does the ray's segment stay clear of the instance's entry-cull sphere?  (derivation: k_inst_recs, rc_build.hip)
                    const float4 cs = buf_f4(crs, (uint32_t)cur_inst << 5);
                    const float cB = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(crs, (uint32_t)cur_inst << 5, 16, 0));
                    const float Lx = cs.x - wo.x, Ly = cs.y - wo.y, Lz = cs.z - wo.z;
                    const float LL = __builtin_fmaf(Lz, Lz, __builtin_fmaf(Ly, Ly, Lx * Lx));
                    const float bq = __builtin_fmaf(Lz, wd.z, __builtin_fmaf(Ly, wd.y, Lx * wd.x));
                    const float tc = bq * c_idd;                                  // parameter of the closest approach
                    const float d2 = __builtin_fmaf(-4.0e-6f, LL, __builtin_fmaf(-tc, bq, LL));  // distance^2 of the line, less its own rounding
                    // a slab test passes only for a parameter inside [tmin, closest_t] (:1853-1856): the SEGMENT has to stay clear of the
                    // sphere, not the whole line -- instances behind the hit found so far, behind a bounce ray's origin or beyond the light
                    const float ts = __builtin_amdgcn_fmed3f(tc, tmin, closest_t) - tc;
                    const float seg = __builtin_fmaf(0.999f * ts * ts, c_dd, d2);
                    const float Ae = cs.w + c_ray;
                    const float tb = 2.0f * __builtin_fmaf(Ae, c_idl, __builtin_fabsf(tc));
                    const float R = __builtin_fmaf(__builtin_fmaf(5.0e-6f, c_dd * c_idl, cB), tb, Ae);  // (B + 5e-6 |d|) t_bound + A + A_ray
                    skip = seg > R * R;                                           // (any NaN or Inf on the way: false, the instance is entered)
                }
                if (STATS) was_skipped = skip;
                if (skip) {
                    cur_inst = -1;
                    node = st.pop(sp);
                } else {
                if (INST_LDS) {
                    const float2* q = il + cur_inst;
                    const float2 p0 = q[0], p1 = q[kTlasLdsInst], p2 = q[2 * kTlasLdsInst], p3 = q[3 * kTlasLdsInst],
                                 p4 = q[4 * kTlasLdsInst], p5 = q[5 * kTlasLdsInst], p6 = q[6 * kTlasLdsInst];
                    m0 = make_float4(p0.x, p0.y, p1.x, p1.y); m1 = make_float4(p2.x, p2.y, p3.x, p3.y); m2 = make_float4(p4.x, p4.y, p5.x, p5.y);
                    m3 = u4v{__float_as_uint(p6.x), 0u, 0u, __float_as_uint(p6.y)};  // nodes offset, leaf count (the other two words are the sink's)
                } else {
                    const uint32_t ioff = (uint32_t)cur_inst << 6;
                    m0 = buf_f4(irs, ioff); m1 = buf_f4(irs, ioff, 16); m2 = buf_f4(irs, ioff, 32);
                    m3 = __builtin_amdgcn_raw_buffer_load_b128(irs, ioff, 48, 0);
                }
                st.push(sp, kSent);
                node = 1;
                cur_off = m3.x;
                n_level = m3.w;
                const float3_ o = mk3(m0.x * wo.x + m0.y * wo.y + m0.z * wo.z + m0.w, m1.x * wo.x + m1.y * wo.y + m1.z * wo.z + m1.w,
                                      m2.x * wo.x + m2.y * wo.y + m2.z * wo.z + m2.w);
                const float3_ d = mk3(m0.x * wd.x + m0.y * wd.y + m0.z * wd.z, m1.x * wd.x + m1.y * wd.y + m1.z * wd.z,
                                      m2.x * wd.x + m2.y * wd.y + m2.z * wd.z);
                oyz = v2f{o.y, o.z}; ozx = v2f{o.z, o.x}; dyz = v2f{d.y, d.z}; dzx = v2f{d.z, d.x};
                inv = safe_inv3(d);
                ox = mk3(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
                }
                RC_MARK("entry_end");
            }
        }
        if (STATS) st_cull += (unsigned long long)__popcll(__ballot(was_skipped));
        // ---- finished lanes: write out; refill when enough lanes are free
        {
            RC_MARK("finish_begin");
            if (STATS) st_outer += 1;
            const bool fin = live && node == kInv;
            const int n_free = __popcll(__ballot(fin || !live));
            const bool can_refill = !(exhausted && pool_next == pool_end);
            if (STATS && !can_refill && st_tx == 0) st_tx = wall_clock64();
            if (TIMELINE) {
                tl_outer += 1;
                if (!can_refill) {
                    const uint32_t n_live = 64u - (uint32_t)n_free;
                    if (tl_tx == 0) tl_tx = wall_clock64();
                    tl_outer_x += 1; tl_live_x += n_live;
                    if (n_live < 16u && tl_t16 == 0) tl_t16 = wall_clock64();
                    if (n_live < 4u && tl_t4 == 0) tl_t4 = wall_clock64();
                }
            }
            if (!can_refill) {  // drain: no more rays to hand out, so the interior loop's exit threshold follows the lanes still alive
                const int half_live = (64 - n_free) / 2;
                thr_eff = __builtin_amdgcn_readfirstlane(half_live < a.int_thr ? (half_live > 1 ? half_live : 1) : a.int_thr);
            }
            if (n_free == 64 && !can_refill && !__ballot(fin)) break;
            if (n_free >= a.refill || n_free == 64 || !can_refill) {
                if (STATS && __ballot(fin)) st_sub[2] += 1;
                RC_MARK("finish_end");
                RC_MARK("writeout_begin");
                if (fin) {
                    sink(my_ray, closest_inst >= 0, closest_t, hit_u, hit_v, closest_prim, closest_inst);
                    live = false;
                    const uint32_t life = it_total - start_it;
                    if (claim_cost && life >= life_thr) atomicMax(claim_cost + (uint32_t)(my_ray >> a.claim.pool_shift), life);
                }
                RC_MARK("writeout_end");
                RC_MARK("finish_begin");
                if (STATS) { st_iter[0] += 1; }
                while (can_refill) {
                    const unsigned long long free_mask = __ballot(!live);
                    const int nf = __popcll(free_mask);
                    if (nf == 0) break;
                    uint32_t touched = 0u;
                    if (pool_next == pool_end) {
                        if (exhausted) break;
                        if (!rc_claim_chunk(a.claim, claim_order, (blockIdx.x * BLOCK + threadIdx.x) >> 6, lane, a.n_items, pool_next, pool_end)) { exhausted = true; break; }
                        if constexpr (Source::kPrefetch) touched = src.prefetch(pool_next, pool_end, lane);  // the claimed range's lines, requested together with the first ray loads below
                    }
                    const unsigned long long left = pool_end - pool_next;
                    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(free_mask >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((unsigned)free_mask, 0u));
                    if (STATS) st_sub[3] += 1;
                    RC_MARK("finish_end");
                    RC_MARK("refill_begin");
                    if (!live && rank < left) {
                        my_ray = pool_next + rank;
                        const RcRay r = src(my_ray);
                        wo = mk3(r.ox, r.oy, r.oz);  // init (:1904-1927); check_direction (src/ray.jl:39-49)
                        wd = mk3(r.dx == 0.0f ? 0.0f : r.dx, r.dy == 0.0f ? 0.0f : r.dy, r.dz == 0.0f ? 0.0f : r.dz);
                        winv = safe_inv3(wd);
                        if (cull_on) {
                            const float dd = __builtin_fmaf(wd.z, wd.z, __builtin_fmaf(wd.y, wd.y, wd.x * wd.x));
                            const float o1 = __builtin_fabsf(wo.x) + __builtin_fabsf(wo.y) + __builtin_fabsf(wo.z);
                            const bool regime = dd >= 1.0e-2f && dd <= 1.0e6f && o1 < 1.0e30f;  // (false for NaN / Inf components)
                            c_idd = regime ? __builtin_amdgcn_rcpf(dd) : __builtin_nanf("");
                            c_idl = __builtin_amdgcn_rsqf(dd);
                            c_dd = dd;
                            c_ray = 8.0e-5f * o1;
                        }
                        inv = winv;  // (o, d are the instance-local ray: set at the first instance entry)
                        ox = mk3(-wo.x * inv.x, -wo.y * inv.y, -wo.z * inv.z);
                        tmin = ANY ? 0.0f : r.tmin;
                        closest_t = r.tmax;
                        hit_u = hit_v = 0.0f;
                        closest_prim = RC_INVALID_NODE;
                        closest_inst = -1; cur_inst = -1;
                        cur_off = tlas_off; n_level = n_instances;
                        sp = st.empty();
                        st.push(sp, kInv);
                        node = 1;
                        live = true;
                        start_it = it_total;
                    }
                    if constexpr (Source::kPrefetch) asm volatile("" ::"v"(touched));  // (keeps the touch alive; loads return in order, so the ray loads' wait has covered it)
                    RC_MARK("refill_end");
                    RC_MARK("finish_begin");
                    pool_next += ((unsigned long long)nf < left) ? (unsigned long long)nf : left;
                }
            }
            RC_MARK("finish_end");
        }
    }
    if (TIMELINE && lane == 0 && a.timeline) {  // [t0, tx (claims dry + own pool empty), t(<16 live), t(<4 live), t_end, outer | outer after tx << 32, interior iterations | after tx << 32, sum of live lanes after tx]
        unsigned long long* w = a.timeline + (size_t)(gtid >> 6) * 8u;
        w[0] = tl_t0; w[1] = tl_tx; w[2] = tl_t16; w[3] = tl_t4; w[4] = wall_clock64();
        w[5] = tl_outer | ((unsigned long long)tl_outer_x << 32); w[6] = tl_int | ((unsigned long long)tl_int_x << 32); w[7] = tl_live_x;
    }
    if (STATS) {
        const unsigned long long t_end = wall_clock64();
        for (int k = 0; k < 4; ++k) {
            unsigned long long v = st_lane[k];  // one atomic per wave: 64 same-address atomics per wave would stall the waves still tracing
            for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
            if (lane == 0) { atomicAdd(&a.stats[2 * k], st_iter[k]); atomicAdd(&a.stats[2 * k + 1], v); }
        }
        if (lane == 0) {  // drain timing, 100 MHz ticks: [8] ~first start, [9] ~first wave out of work, [10] last end, [11] sum(end - out of work), [12] sum(end - start), [13] waves
            atomicMax(&a.stats[8], ~st_t0);
            atomicMax(&a.stats[9], ~st_tx);
            atomicMax(&a.stats[10], t_end);
            atomicAdd(&a.stats[11], t_end - st_tx);
            atomicAdd(&a.stats[12], t_end - st_t0);
            atomicAdd(&a.stats[13], 1ull);
            atomicAdd(&a.stats[14], st_outer);  // outer iterations (one pass over the phases each)
            for (int k = 0; k < 4; ++k) atomicAdd(&a.stats[15 + k], st_sub[k]);
            atomicAdd(&a.stats[19], st_cull);  // instance entries skipped by the entry cull (counted once per wave-pass: lane 0's copy of the wave total)
        }
    }
}


}  // namespace rc

rc::SceneView rc_scene_view(rc_scene* s, uint32_t total_threads);  // rc_traverse.hip: a traversal launch's view, inside an RcLaunchGuard
rc::SceneView rc_scene_view_static(rc_scene* s);                    // the arrays only: no spill region, no side effects (stage kernels)
uint32_t* rc_launch_overflow(rc_scene* s, uint32_t total_threads);  // the spill region of the launch being prepared (never null)
rc::PersistArgs rc_persist_args(rc_scene* s, uint64_t n_items, uint32_t total_threads);
// cost-ordered claiming for a launch inside an RcLaunchGuard: `kind` separates the histories of launches that map items to rays differently
// (0 closest_hit, 1 any_hit, 2 the get_illumination grid)
// d_rays: the launch's ray array (the batch is recognised by sample rays read on the device), or nullptr with host_sample[8] = a description of
// generated rays (o.xyz, t_min, d.xyz, t_max of a stand-in ray: two launches whose stand-ins are close are the same batch)
bool rc_cost_order_setup(rc_scene* s, uint64_t n_items, int kind, hipStream_t stream, rc::RcClaim& claim, const RcRay* d_rays, const float* host_sample = nullptr);
bool rc_lds_driver_ok(rc_scene* s);
bool rc_partial_driver_ok(rc_scene* s);
void rc_partial_driver_args(rc_scene* s, rc::PersistArgs& p);
uint32_t rc_lds_driver_blocks(rc_scene* s, uint64_t n_items);
void rc_lds_driver_args(rc_scene* s, rc::PersistArgs& p);
