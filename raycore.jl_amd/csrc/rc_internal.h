// rc_internal.h -- host-side declarations shared by the translation units of libraycore_mi355x.so.
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "rc_device.h"

struct RcError : std::runtime_error {
    int code;
    RcError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define RC_HIP(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            throw RcError(4, std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" __FILE__ ":" + \
                                 std::to_string(__LINE__) + ")");                                         \
    } while (0)

struct PinnedU32 {  // one word of pinned host memory a kernel can write and the host can poll (no synchronisation implied)
    uint32_t* p = nullptr;
    PinnedU32() = default;
    PinnedU32(const PinnedU32&) = delete;
    PinnedU32& operator=(const PinnedU32&) = delete;
    PinnedU32(PinnedU32&& o) noexcept : p(o.p) { o.p = nullptr; }
    PinnedU32& operator=(PinnedU32&& o) noexcept { if (this != &o) { drop(); p = o.p; o.p = nullptr; } return *this; }
    ~PinnedU32() { drop(); }
    void drop() { if (p) (void)hipHostFree(p); p = nullptr; }
    void ensure() { if (!p) { RC_HIP(hipHostMalloc((void**)&p, 64, hipHostMallocDefault)); for (int k = 0; k < 16; ++k) p[k] = 0u; } }  // (a cache line: users keep a few words)
};

template <typename T>
struct DevBuf {  // owning device buffer, grow-only reuse
    T* p = nullptr;
    size_t cap = 0;
    void reserve(size_t n) {
        if (n <= cap) return;
        release();
        RC_HIP(hipMalloc((void**)&p, n * sizeof(T)));
        cap = n;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; }
        return *this;
    }
};

// "Is the last user of this per-stream resource done?" is asked of an event the LIBRARY recorded behind that user, never of the caller's
// stream handle (hipStreamQuery on a stream another thread is capturing into invalidates the capture, and on a destroyed
// stream it fails, so an entry keyed by a dead stream was never handed on).
struct RcEvent {
    hipEvent_t e = nullptr;
    bool recorded = false, owned = false;
    RcEvent() = default;
    RcEvent(const RcEvent&) = delete;
    RcEvent& operator=(const RcEvent&) = delete;
    RcEvent(RcEvent&& o) noexcept : e(o.e), recorded(o.recorded), owned(o.owned) { o.e = nullptr; o.recorded = o.owned = false; }
    RcEvent& operator=(RcEvent&& o) noexcept { if (this != &o) { drop(); e = o.e; recorded = o.recorded; owned = o.owned; o.e = nullptr; o.recorded = o.owned = false; } return *this; }
    ~RcEvent() { drop(); }
    void drop() { if (e && owned) (void)hipEventDestroy(e); e = nullptr; recorded = owned = false; }
    void record(hipStream_t st) {  // an event of its own behind the stream's work so far
        if (e && !owned) drop();
        if (!e) { if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); e = nullptr; return; } owned = true; }
        if (hipEventRecord(e, st) == hipSuccess) recorded = true; else (void)hipGetLastError();
    }
    // ... or the closing event the launch recorded anyway (its counter slot's t1, which lives as long as the scene): one event record per
    // launch fewer for each resource -- an event record is a packet of its own on the queue, ~3 us between two kernels.  The slot's next
    // launch records t1 again; that launch runs behind this one (same stream, or it waited for t1 first: RcLaunchGuard), so "t1 is done"
    // still implies "this launch is done" -- the answer can only come later than with an event of its own, never earlier.
    void follow(hipEvent_t closing) { drop(); e = closing; recorded = closing != nullptr; }
    bool idle() const {  // never blocks
        if (!recorded) return true;
        const hipError_t q = hipEventQuery(e);
        if (q != hipSuccess) (void)hipGetLastError();
        return q == hipSuccess;
    }
    void wait() const { if (recorded && hipEventSynchronize(e) != hipSuccess) (void)hipGetLastError(); }
};

struct Blas {  // one geometry: build_blas output (src/instanced-bvh.jl:111-118), device resident
    DevBuf<RcNode> nodes;
    DevBuf<RcPrim> prims;  // Morton-sorted
    uint32_t n_prims = 0;
    uint32_t n_nodes = 0;
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
    // mesh attributes (rc_add_mesh): per-vertex normals / uvs, the face indices and the source face of every sorted primitive
    DevBuf<float> m_normals, m_uvs;
    DevBuf<uint32_t> m_indices, src_face;
    bool has_attrs = false, has_uvs = false;
    uint32_t n_mesh_verts = 0, n_mesh_faces = 0;
    // BLAS4 (src/bvh4.jl:154-162), built on request by rc_blas4_build
    DevBuf<RcNode4> nodes4;
    uint32_t n_nodes4 = 0;
    uint32_t root_word4 = 1;  // root index, leaf bit set when the tree is a single leaf
};

struct HandleRange {
    uint32_t first = 0, count = 0;  // 0-based range in the instance array
};

namespace rc {
constexpr int kTlasLdsNodes = 511;    // largest TLAS (nodes) the LDS kernels take: kTlasLdsInst instances
constexpr int kTlasLdsInst = 256;
// What those kernels keep in LDS behind the lane stacks (rc_traverse_core.h, stage_lds_top):
//  * node planes, kLdsPlaneNodes entries of seven float2 each: the TLAS's interior nodes 1..n-1 (its leaves are never fetched as
//    nodes -- their boxes were tested in the parent), then, in a scene with a single BLAS, that BLAS's top `blas_k` internal nodes,
//    which rc_build_tlas renumbers to 1..blas_k in breadth-first order in the traversal copy (a private permutation of internal
//    node indices: visit order, tests and results are untouched, leaves keep their indices);
//  * the TLAS leaf -> instance index table (kTlasLdsInst words);
//  * the instance records as seven float2 planes of kTlasLdsInst entries (inverse transform, nodes offset, leaf count).
constexpr int kLdsPlaneNodes = 310;
constexpr int kPartialPlaneNodes = 585;  // node planes of the PARTIAL_LDS kernel (no leaf table, no instance planes): 32 760 bytes
// Scenes whose trees ALL have fewer than 65 534 nodes (every BLAS <= 32 767 triangles, <= 32 767 instances) run the STACK16 shape of the
// same kernels (round 5): lane-stack entries of 16 bits, and the 24 KiB of LDS that frees per workgroup holds more of the tree.
#ifndef RC_LDS_PLANES16        // dev: tools/lds_bound_probe.py builds a variant with 1 278 entries (every interior node of a 1 024-triangle BLAS; one workgroup per CU)
#define RC_LDS_PLANES16 748
#endif
constexpr int kLdsPlaneNodes16 = RC_LDS_PLANES16;  // 7 x 748 x 8 = 41 888 bytes of node planes (+ 24 576 of stacks + 1 024 + 14 336 = 81 824)
constexpr int kPartialPlaneNodes16 = 1023; // 7 x 1023 x 8 = 57 288 (+ 24 576 = 81 864)
constexpr uint32_t kStack16MaxLeaves = 32767;
}  // namespace rc

struct TraceOptions {
    int64_t kernel = -1;       // -1 = auto, 0 = one-ray-per-lane, 1 = persistent wave-refill, 2 = persistent + voted path scheduling, 3 = persistent + phased (while-while), 4 = 3 with the TLAS + instance records staged in LDS (1024-thread blocks, <= 256 instances), 5 = 3 with the TLAS staged in LDS at 24 waves/CU (2 x 768 threads)
    int64_t blocks_per_cu = 0; // 0 = derive from the LDS stack depth
    int64_t lds_stack = 24;    // per-lane stack entries kept in LDS: 12, 16, 24 or 32
    int64_t pool = 0;          // persistent kernels: ray indices per atomic claim (0 = auto 64..512)
    int64_t refill = 20;       // persistent kernel: refill when this many lanes of a wave are idle
    int64_t sched_thr = 36;    // kernel 2: lanes that must wait for a leaf/switch batch; kernel 3: interior lanes below which the wave serves the waiting lanes
    int64_t stats = 0;         // dev instrumentation (persistent kernels only)
    int64_t onesweep_min = 1000000;  // builds: key counts from here up are sorted by Onesweep radix passes, smaller ones by rocPRIM's merge sort (measured: 0.22 vs 0.25 ms at 250 k keys, 0.437 vs 0.425 ms at 1 M)
    int64_t stack16 = 1;       // scenes whose trees all have fewer than 65 534 nodes: 16-bit lane-stack entries, the freed LDS holds more of the tree (kernels 5 / 6)
    int64_t blas_top = 1;      // single-BLAS scenes: renumber the BLAS's top internal nodes to the front of the traversal copy and let kernel 5 read them from LDS
    int64_t host_pipeline = 1; // host-buffer trace calls of >= 1 Mi rays overlap upload / trace / download in chunks
    int64_t taper = 12;        // persistent kernels: guided chunk sizes at the end of a batch, in eighths of (chunk size x waves) still to hand out per piece (RcClaim); 0 = all chunks of `pool` items
    int64_t entry_cull = 1;    // phased kernels: an instance whose entry-cull sphere the ray's segment misses is not entered (k_inst_recs, rc_build.hip); results never change.  1 = closest_hit and the drivers, 2 = any_hit too
    int64_t cost_order = 1;    // phased kernels: chunks that held long rays in the previous launch of the same shape (batch size, mode, stream) are claimed first (RcClaim::order)
    int64_t cost_thr = 64;     //   initial reporting threshold, in interior-loop iterations of a ray's wave while the ray was in flight (adapted from launch to launch)
    int64_t claim_shards = 16; // phased kernels: chunk counters in use (a power of two <= kClaimShards)
    int64_t vf_first_touch = 1;          // ROWS on several devices: each device's host thread joins the device's NUMA node and faults in its own row block (rc_multi.hip)
    int64_t vf_chunk_bytes = 192 << 20;  // host-matrix view factors (rc_multi.hip): device block per row chunk -- large enough for full-rate launches and 2-D copies, small enough that the exposed first trace / last copy are a few ms
    int64_t timeline_ptr = 0;  // dev: device address of 8 x u64 per wave (n_cus x 24 waves) that kernel 5 fills with its waves' event times; 0 = off
};

// Where the duration of an operation can be read back (rc_last_kernel_ms): the event pair of a launch slot (valid while the slot has not
// been reused: `seq`), the scene's own pair (slot kSceneTimingSlot: builds, refits, BVH4 collapse), or a value computed by the call
// itself (slot -1: the chunked host-buffer trace sums its chunks' kernel times).
struct TimingRef {
    int slot = -2;  // -2 = nothing timed yet
    uint64_t seq = 0;
    float fixed_ms = 0.f;
};

// Staging of one host-buffer trace call in flight (rc_trace_closest / rc_trace_any / rc_trace_*4): its own stream and device copies of
// the caller's rays and hits, so that calls from several host threads on one synced scene do not share anything (SURVEY.md 8b:
// "trace calls are re-entrant on a synced scene"; the reference's drivers call closest_hit from Threads.@threads, src/kernels.jl:64,82).
// The calling thread's stream-capture interaction mode set to relaxed for a scope (see guarded() in rc_capi.hip): what every entry point and
// every worker thread of the library runs under, so that its allocations and copies neither fail nor invalidate a capture another thread
// has open in hipStreamCaptureModeGlobal.
struct RcCaptureRelaxed {
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    RcCaptureRelaxed() { (void)hipThreadExchangeStreamCaptureMode(&mode); }
    ~RcCaptureRelaxed() { (void)hipThreadExchangeStreamCaptureMode(&mode); }
    RcCaptureRelaxed(const RcCaptureRelaxed&) = delete;
    RcCaptureRelaxed& operator=(const RcCaptureRelaxed&) = delete;
};

struct CallCtx {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool busy = false;
    DevBuf<RcRay> rays;
    DevBuf<RcHit> hits;
};

struct rc_scene {
    int device = 0;
    int n_cus = 0;
    uint64_t uid = 0;                  // process-unique (a destroyed scene's address may be reused; thread-local timing references name the uid)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // the scene's own timing pair (LaunchSlot kSceneTimingSlot)

    // mutable TLAS state (src/instanced-bvh.jl:261-310)
    std::vector<Blas> blas;
    std::vector<RcInstanceDesc> instances;  // host mirror of tlas.instances
    std::map<uint32_t, HandleRange> handle_to_range;
    std::set<uint32_t> deleted_handles;
    uint32_t next_handle_id = 1;
    bool dirty = true, transforms_dirty = false, has_static = false;
    bool host_instances_stale = false;  // descriptors were rewritten on the device (rc_instance_buffer_device + rc_refit_device)

    // StaticTLAS (src/instanced-bvh.jl:155-168): the adapted form owned by rc_sync
    DevBuf<RcNode> tlas_nodes;
    uint32_t n_tlas_nodes = 0;
    DevBuf<RcInstanceDesc> d_instances;
    DevBuf<RcInstRec> inst_recs;
    DevBuf<uint32_t> blas_cull_bits;   // per BLAS: radius (float bits) of the sphere about the root box's centre that holds every leaf box
    DevBuf<float4> inst_cull;          // per instance: (c_w, A), (B, -, -, -): the entry-cull sphere, k_inst_recs (rc_build.hip)
    uint32_t n_static_instances = 0;
    DevBuf<RcNode> flat_nodes;
    uint32_t n_flat_nodes = 0;
    DevBuf<uint32_t> tlas_remap;       // same for the TLAS's internal nodes (top levels too large for the full LDS kernels), kept for refits
    uint32_t tlas_top_k = 0;
    uint32_t tlas_top_k32 = 0, blas_top_k32 = 0;  // what the kernels with 32-bit lane stacks (smaller node planes) stage of the same renumbering: prefixes of tlas_top_k / blas_top_k
    bool small_trees = false;          // every tree of the scene has fewer than 65 534 nodes: the STACK16 kernels apply
    DevBuf<uint32_t> top_remap;        // old -> new internal node index of that renumbering (scratch of rc_build_tlas)
    uint32_t blas_top_k = 0;           // single-BLAS scene: internal nodes 1..blas_top_k of the traversal copy are the tree's top in breadth-first order
    DevBuf<RcPrim> flat_prims;
    uint32_t n_flat_prims = 0;
    DevBuf<RcBlasDesc> d_descs;
    std::vector<RcBlasDesc> descs;
    DevBuf<uint32_t> d_blas_nprims;
    std::vector<uint32_t> blas_nprims;
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};

    // scratch
    DevBuf<uint32_t> keys_a, keys_b, vals_a, vals_b, flags, scene_enc, bounds_partials;
    DevBuf<unsigned char> sort_tmp;
    DevBuf<float> aabb_tmp;
    DevBuf<uint4> range_tmp;    // compact topology of the BLAS being built (k_topology -> k_refit): per internal node its sorted-leaf range, child0, parent; then one parent word per leaf
    DevBuf<uint4> tlas_ranges;  // same for the TLAS; kept, because refit_tlas! reuses the topology
    DevBuf<RcPrim> prim_tmp;
    // ---- launch bookkeeping: everything below is touched only with launch_mu held (RcLaunchGuard, rc_traverse.hip) ----
    // The ENQUEUE of launches on one scene is serialised (microseconds); the kernels themselves overlap freely on their streams.
    std::mutex launch_mu;
    // global spill areas of the traversal stacks: one per stream that has launched on this scene (launches on one stream are
    // ordered and share theirs; launches on different streams may overlap and must not), at most kMaxOverflowRegions (a ninth stream
    // takes over the region of an idle stream, or waits for the oldest).  Launches CAPTURED into a hipGraph never use these: each owns a
    // region of its own (capture_regions).
    static constexpr int kMaxOverflowRegions = 8;  // each is allocated on first use by a new stream
    struct OverflowRegion { hipStream_t stream = nullptr; DevBuf<uint32_t> buf; RcEvent last; };  // last: behind the latest launch that used the region
    std::vector<OverflowRegion> overflow_regions;
    // CAPTURED launches: each owns a counter slot (kEagerSlots + its index here), a spill region sized for ITS grid and, for the drivers, its
    // scratch counters -- a graph bakes the addresses in.  Released all at once (option "release_captures") or one by one (option
    // "release_capture" = the token option "last_capture_token" returned after the capture), once the caller's graph is gone.
    struct CaptureSlot { bool in_use = false; DevBuf<uint32_t> region; std::vector<std::unique_ptr<DevBuf<unsigned long long>>> scratch; };
    std::vector<CaptureSlot> capture_slots;  // kCounterSlots - kEagerSlots entries
    int cur_capture = -1;              // the capture slot of the launch being prepared (-1: an eager launch)
    int last_capture = -1;
    int cur_region = -1, cur_history = -1, cur_scratch = -1;  // what the launch being prepared uses: RcLaunchGuard::finish records their events
    uint32_t* cur_overflow = nullptr;  // the region of the launch being prepared (a captured launch's is allocated by rc_launch_overflow, which knows the grid)
    bool guard_live = false;           // an RcLaunchGuard exists (set / cleared by it under launch_mu): the cur_* fields are meaningful
    DevBuf<uint32_t> counters;        // kCounterSlots slots of claim counters (self-resetting, rc_claim_chunk), the sticky status word, dev statistics; zeroed at rc_scene_create
    uint64_t launch_seq = 0;          // eager launches so far; slot = launch_seq % kEagerSlots
    int cur_slot = 0;                 // slot of the launch being prepared
    hipEvent_t bound_t0 = nullptr, bound_t1 = nullptr;  // RcLaunchGuard::bind: the events the launch's kernel is to carry
    bool bound_used = false;
    struct LaunchSlot {               // per counter slot: the events of its latest launch
        hipEvent_t t0 = nullptr, t1 = nullptr;  // timing pair; t1 also orders the slot's next user when that one runs on another stream
        hipStream_t stream = nullptr;
        bool recorded = false;
        uint64_t seq = 0;
    };
    std::vector<LaunchSlot> slots;    // kCounterSlots + 1: the last entry is the scene's own pair (ev0 / ev1)
    uint64_t timing_seq = 0;
    TimingRef last_timing;            // most recent timed operation on the scene by any thread
    // cost-ordered claiming: what the last launch of a shape learned about its chunks (rc_cost_order_setup, rc_traverse.hip)
    struct ChunkHistory {
        uint64_t n_items = 0; int any = 0; hipStream_t stream = nullptr;
        uint32_t n_chunks = 0, pool = 0;  // chunks of `pool` items
        DevBuf<uint32_t> cost, order, ctl;  // cost: kHistSlots arrays (one per remembered batch); ctl: the header words kHist* (rc_traverse_core.h) + per-block class counts of the order kernels
        DevBuf<float> samples;              // kHistSlots x kHistSamples sample rays (8 floats each): how a launch's batch is recognised
        uint64_t gen = 0;                   // launches of this shape so far
        static constexpr uint32_t kHeaderWords = 48;  // = rc::kHistHeaderWords (rc_traverse_core.h; asserted in rc_traverse.hip)
        uint32_t parity = 0;                // which copy of the header / the samples the shape's next launch reads (order_commit writes the other)
        uint64_t last_use = 0;
        RcEvent last;                       // behind the latest launch that used the entry's buffers
        PinnedU32 fresh_streak;             // written by the launches (order_commit): [0] consecutive launches of this shape that were not repeats of a remembered batch, [1] a recording waits for the rebuild kernels
        uint32_t rebuild_credit = 0;        // launches that still get the rebuild kernel pair in front (a slot may hold a recording: rc_cost_order_setup)
        uint64_t next_record = 8;           // the launch of the shape that next asks its batch to record (cadence 7, 8, 9, ...)
        uint32_t records_asked = 0;
    };
    static constexpr int kMaxHistories = 8;
    static constexpr int kRecentShapes = 16;
    struct ShapeKey { uint32_t n_chunks = 0, pool = 0; int any = 0; hipStream_t stream = nullptr; };
    std::vector<ShapeKey> recent_shapes;  // shapes launched lately without a history entry: one that comes back may take over an entry
    uint64_t recent_clock = 0;
    std::vector<ChunkHistory> histories;
    uint64_t history_clock = 0;

    // host-buffer trace calls in flight (ctx_mu): a small pool of staging contexts; a call beyond kMaxCallCtx waits for one to come free
    static constexpr int kMaxCallCtx = 4;
    std::mutex ctx_mu;
    std::condition_variable ctx_cv;
    std::vector<std::unique_ptr<CallCtx>> call_ctx;
    hipStream_t aux_streams[4] = {nullptr, nullptr, nullptr, nullptr};  // rc_multi.hip: two compute streams, a copy stream, a communication stream (created on first use)
    std::mutex batches_mu;             // rc_trace_*_device_batches: one fork / join over the auxiliary streams is enqueued at a time
    hipEvent_t batch_fork = nullptr, batch_join[4] = {nullptr, nullptr, nullptr, nullptr};
    std::mutex stage_mu;               // stage kernels (hit points, shadow rays, ...) on the callers' streams: the last launch per stream, so that
    std::map<hipStream_t, hipEvent_t> stage_events;  // rc_scene_destroy can wait for exactly the work that still reads the scene
    std::mutex host_call_mu;          // the other host-buffer entry points (illumination, view factors, collisions, exports) run one at a time

    DevBuf<float> f32_stage;
    struct TotalsScratch { hipStream_t stream = nullptr; std::unique_ptr<DevBuf<unsigned long long>> buf; RcEvent last; };
    std::vector<TotalsScratch> totals_scratch;  // view-factor totals / illumination: private copies of the accumulators, one area per stream (rc_drivers.hip; launch_mu); a captured launch's live in its CaptureSlot
    DevBuf<unsigned long long> u64_stage;  // view-factor totals: received[N] then emitted[N] (rc_multi.hip)
    DevBuf<float> vert_stage;
    DevBuf<uint32_t> meta_stage;
    DevBuf<uint32_t> c4_tasks_a, c4_tasks_b, c4_gather, c4_totals;  // BVH4 collapse scratch (rc_bvh4.hip)
    DevBuf<unsigned long long> c4_counts, c4_offsets;
    DevBuf<uint32_t> slot_face;       // rc_add_mesh: source face of every compacted slot
    DevBuf<float> flat_attrs;         // 15 floats per flat primitive (normals 9, uv 6), built on demand after a rebuild
    bool flat_attrs_valid = false;
    DevBuf<uint32_t> vf_order;        // flat primitive indices by ascending metadata (RC_VF_SOURCES_BY_METADATA), built on demand after a rebuild
    std::vector<uint32_t> vf_meta_sorted;  // the metadata in that order (host): which source positions fall into a range of matrix rows
    bool vf_order_valid = false;
    DevBuf<uint32_t> compact_flags, compact_pos;  // rc_compact_hits scratch
    DevBuf<unsigned char> compact_tmp;
    DevBuf<uint32_t> collide_counts;  // collide_instances' per-leaf counts / prefix sums (the reference's `cache`)
    DevBuf<uint2> contact_stage;

    bool lds_attr_set[16] = {};  // hipFuncAttributeMaxDynamicSharedMemorySize done for kernels 4 / 5 (closest, any), illumination, view factors, [6, 7] kernel 6, [8, 9] the partial-LDS drivers, [10, 11] view-factor totals

    TraceOptions opt;
};

// rc_build.hip
uint32_t rc_ingest_faces(rc_scene* s, const float* d_verts, const uint32_t* d_meta, uint32_t n, bool keep_face_map = false);  // -> s->prim_tmp (+ s->slot_face)
void rc_build_blas(rc_scene* s, uint32_t n, Blas& out, bool keep_face_map = false);  // builds from s->prim_tmp
void rc_expand_mesh(rc_scene* s, const float* d_verts, const uint32_t* d_indices, const uint32_t* d_vertex_meta, bool meta_per_face, uint32_t nf, float* d_soup, uint32_t* d_meta);
void rc_ensure_flat_attrs(rc_scene* s);  // fills s->flat_attrs for the current flat primitive array
void rc_launch_export_triangles(rc_scene* s, void* d_out, hipStream_t stream);  // 136-byte Triangle{UInt32} records
void rc_launch_reflection_rays(rc_scene* s, const RcRay* d_rays, const RcHit* d_hits, uint64_t n, float bias, RcRay* d_out, hipStream_t stream);
void rc_launch_shading_attributes(rc_scene* s, const RcHit* d_hits, uint64_t n, float* d_normals, float* d_uvs, hipStream_t stream);
void rc_build_tlas(rc_scene* s);   // build_tlas_topology + flat arrays -> StaticTLAS
void rc_refit_tlas(rc_scene* s, bool from_device = false, bool recompute_inverse = false);   // refit_tlas!
void rc_mat3x4_inverse(const float m[12], float out[12]);

// rc_traverse.hip
// learn_order = false: this launch is one piece of a larger batch (the chunked host-buffer path): the next launch of the same size traces OTHER rays,
// so what this one learns about its chunks predicts nothing
void rc_launch_trace(rc_scene* s, const RcRay* d_rays, RcHit* d_hits, uint64_t n, int any_hit, hipStream_t stream, bool learn_order = true);

// rc_bvh4.hip
void rc_build_blas4(rc_scene* s, Blas& b);                            // build_blas4: collapse of b's BVH2 (src/bvh4.jl:511-522)
void rc_export_blas4(rc_scene* s, const Blas& b, void* host_out);     // reference-layout BVHNode4 array (120 B each)
void rc_launch_trace4(rc_scene* s, const Blas& b, const RcRay* d_rays, RcHit* d_hits, uint64_t n, int any_hit, hipStream_t stream);

// rc_collision.hip
uint64_t rc_collide_instances_launch(rc_scene* s, uint2* d_out, uint64_t capacity, hipStream_t stream);

// rc_drivers.hip
void rc_launch_ray_grid(rc_scene* s, const float viewdir[3], uint32_t grid, RcRay* d_rays, hipStream_t stream);
void rc_launch_illumination(rc_scene* s, const float viewdir[3], uint32_t grid, uint64_t ray_begin, uint64_t ray_end,
                            float* d_counts, hipStream_t stream);
void rc_launch_view_factors(rc_scene* s, uint32_t rays_per_triangle, uint64_t seed, uint32_t src_begin, uint32_t src_end,
                            uint32_t ray_begin, uint32_t ray_end, uint32_t* d_matrix, uint64_t row_stride,
                            uint64_t col_stride, uint32_t row_offset, uint32_t flags, hipStream_t stream);

void rc_launch_primary_rays(rc_scene* s, const float pos[3], const float right[3], const float up[3], const float forward[3], float half_width,
                            float half_height, uint32_t width, uint32_t height, uint32_t samples, uint64_t seed, int jitter, RcRay* d_out, hipStream_t stream);
void rc_launch_compact_hits(rc_scene* s, const RcHit* d_hits, uint64_t n, uint32_t* d_indices, uint32_t* d_count, hipStream_t stream);

// rc_traverse.hip helpers shared with rc_drivers.hip
namespace rc { struct SceneView; struct RcClaim; }
// One slot of chunk counters per launch, so that launches of one scene in flight on different streams never share a counter: from word
// kShardBase of a slot on, kClaimShards counters kShardStrideWords apart.  Eager launches rotate over slots 0 .. kEagerSlots - 1 (a
// launch that reuses a slot from another stream first waits for the event its previous user left there); launches that are being
// captured into a hipGraph rotate over the remaining slots, which eager launches never touch -- a graph bakes its slot in, and a
// replay must not meet an eager launch on the same counters.  A returning atomic on
// ONE address costs 12.6 ns on MI355X however many waves issue it (tools/archive/atomic_probe.hip): 6144 waves claiming their first rays
// wait up to 77 us, and the 32 768 claims of a 4 M-ray launch keep a single counter busy for 0.41 ms.  Sixteen counters 256 bytes
// apart run at 0.9 ns per claim.  The counters zero themselves at the end of every launch (rc_claim_chunk).  Words [4] and [8..] of SLOT 0 are the scene's
// sticky stack-overflow status (set by any launch, read and cleared by check_status / rc_wait: a later launch cannot clear an
// earlier launch's report) and the dev statistics (zeroed per launch only while the "stats" option is on).
constexpr int kCounterSlots = 64, kEagerSlots = 48, kCounterSlotWords = 2048, kCounterSlotUsedWords = 1088;
constexpr int kSceneTimingSlot = kCounterSlots;
constexpr int kClaimShards = 16, kShardBase = 64, kShardStrideWords = 64;
constexpr int kStatsWords = 24;  // u64 dev statistics behind the status word of slot 0 (u32 words 8 .. 55: below kShardBase)
inline uint32_t* rc_counter_slot(rc_scene* s) { return s->counters.p + (size_t)s->cur_slot * kCounterSlotWords; }  // slot of the launch being prepared (launch_mu held)
inline uint32_t* rc_status_word(rc_scene* s) { return s->counters.p + 4; }
inline unsigned long long* rc_stats_words(rc_scene* s) { return reinterpret_cast<unsigned long long*>(s->counters.p + 8); }
// Serialises the enqueue of one launch on a scene and does its bookkeeping.  Construction takes launch_mu, picks the stack spill region
// of `stream` and the launch's counter slot, and orders the launch behind the slot's previous user; start() records the slot's first
// timing event, finish() the second (both skipped while `stream` is being captured) and makes the launch the calling thread's -- and the
// scene's -- latest timed operation.  Everything between construction and destruction runs with the lock held.
struct RcLaunchGuard {
    rc_scene* s;
    hipStream_t stream;
    std::unique_lock<std::mutex> lock;
    bool capturing = false;
    RcLaunchGuard(rc_scene* scene, hipStream_t stream);
    ~RcLaunchGuard();
    RcLaunchGuard(const RcLaunchGuard&) = delete;
    RcLaunchGuard& operator=(const RcLaunchGuard&) = delete;
    void start();
    void bind();   // the launch's ONE kernel carries the slot's events (hipExtLaunchKernelGGL) instead of event records around it
    void finish();
};
void rc_claim_fill(rc_scene* s, uint64_t n_items, uint32_t total_waves, rc::RcClaim& out);  // the RcClaim of the launch being prepared
// timing of operations that are not launches through RcLaunchGuard (builds, refits: mutations, externally serialised)
void rc_timing_scene_begin(rc_scene* s, hipStream_t stream);
void rc_timing_scene_end(rc_scene* s, hipStream_t stream);
// Blocking copies and fills that stay off the legacy null stream: a null-stream operation synchronises with every BLOCKING stream of the
// device -- also one that another thread is capturing into a graph, which HIP answers by invalidating the capture (no capture
// interaction mode relaxes that).  These run on a per-device non-blocking utility stream and wait for it.
void rc_copy_now(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
void rc_memset_now(void* p, int value, size_t bytes);
void rc_note_stage_launch(rc_scene* s, hipStream_t stream);  // after a kernel outside RcLaunchGuard that reads scene memory
void rc_timing_fixed(rc_scene* s, float ms);
float rc_timing_read(rc_scene* s);  // the calling thread's latest timed operation on the scene, else the scene's latest
uint32_t rc_persistent_blocks(rc_scene* s, uint64_t n_items);
void rc_ensure_vf_order(rc_scene* s);  // builds vf_order / vf_meta_sorted if the scene has been rebuilt since
void rc_vf_source_range(rc_scene* s, uint32_t row_begin, uint32_t row_end, uint32_t& pos_begin, uint32_t& pos_end);  // positions in the metadata order whose metadata - 1 is in [row_begin, row_end)
// rc_multi.hip
void rc_launch_vf_totals(rc_scene* s, uint32_t rays_per_triangle, uint64_t seed, uint32_t src_begin, uint32_t src_end, uint32_t ray_begin,
                         uint32_t ray_end, unsigned long long* d_received, unsigned long long* d_emitted, hipStream_t stream);
void rc_multi_prepare_impl(rc_scene* const* scenes, int n_scenes, float out_ms[4]);
int rc_multi_ranks_impl(rc_scene* const* scenes, int n_scenes);
void rc_view_factor_totals_multi_impl(rc_scene* const* scenes, int n_scenes, uint32_t rays_per_triangle, uint64_t seed, uint64_t* out_received, uint64_t* out_emitted);
float rc_view_factors_rows_to_host(rc_scene* s, uint32_t rays_per_triangle, uint64_t seed, uint32_t row_begin, uint32_t row_end, uint32_t* out, uint64_t ld);
void rc_view_factors_multi_impl(rc_scene* const* scenes, int n_scenes, uint32_t rays_per_triangle, uint64_t seed, uint32_t* out, int mode);
struct rc_ray;  // include/raycore_mi355x.h (same 32 bytes as RcRay / RcHit)
struct rc_hit;
void rc_trace_multi_impl(rc_scene* const* scenes, int n_scenes, const rc_ray* rays, rc_hit* hits, uint64_t n, int any);
void rc_illumination_multi_impl(rc_scene* const* scenes, int n_scenes, const float viewdir[3], uint32_t grid, float* out);
void rc_trace_host_impl(rc_scene* s, const rc_ray* rays, rc_hit* hits, uint64_t n, int any);  // rc_capi.hip: one scene's host-buffer batch (throws)
void rc_launch_view_factor_rays(rc_scene* s, uint64_t seed, uint32_t src, uint32_t ray_begin, uint32_t n_ray, RcRay* d_out, hipStream_t stream);
void rc_launch_hit_points(rc_scene* s, const RcRay* d_rays, const RcHit* d_hits, uint64_t n, float* d_points, float* d_normals, hipStream_t stream);
void rc_launch_shadow_rays(rc_scene* s, const RcRay* d_rays, const RcHit* d_hits, uint64_t n, const float light[3], float bias, RcRay* d_out, hipStream_t stream);
