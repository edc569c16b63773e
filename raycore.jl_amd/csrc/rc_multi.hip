// rc_multi.hip -- view_factors into the caller's HOST matrix, on one device or on several, behind the C ABI.
//
// view_factors(tlas; rays_per_triangle) returns a host Matrix{UInt32}, column-major, result[src_meta, hit_meta] (src/kernels.jl:74-104).
// At BASELINE config C5 that matrix is 10 GB: one device traces the 204.9 M rays in 41 ms, and moving the result over one PCIe link
// (56 GB/s on this box) takes 180 ms -- the copy, not the tracing, is what the API's caller waits for.  So:
//
//  * the job runs in ROW CHUNKS: a chunk's rows are traced into a small device block (column-major inside, ld = rows of the chunk)
//    while the previous chunk's block travels to the host matrix as one 2-D copy (N pieces of rows*4 bytes, N*4 bytes apart), on
//    alternating compute streams and one copy stream -- the tracing hides entirely inside the transfer, and no N x N device matrix
//    (nor its 10 GB zero fill) exists;
//  * with several devices (one process, one scene per device, built from the same geometry), mode ROWS gives every device a block of
//    matrix rows and lets it copy its chunks straight into the caller's matrix over ITS OWN PCIe link -- G links in parallel, no xGMI
//    traffic, no collective; mode RAYS is the partition BASELINE's north star names: every device shoots rays_per_triangle / G rays of
//    every source into a full accumulator, row chunks are summed into device 0 by RCCL (ncclReduce, ncclUint32, called directly:
//    librccl is loaded at first use) over xGMI while the next chunk is traced, and device 0 copies them out.
//
// Philox is keyed by (seed; ray index, source primitive), so every partition produces the same matrix bit for bit.
// Reference: src/kernels.jl:74-104; SURVEY.md 8b / 8e.
#include <dlfcn.h>
#include <sched.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <thread>

#include "../../include/raycore_mi355x.h"
#include "rc_internal.h"
#include "rc_rccl_abi.h"

namespace {

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23  // Linux 5.14+
#endif
// A fresh N x N matrix (zeros(UInt32, N, N), np.zeros) has no pages yet; faulting 10 GB in from inside the copy path costs seconds.
// Ask the kernel for the range up front, in parallel slices (best effort: EINVAL on old kernels leaves the faults to the copies).
void populate_parallel(void* p, size_t bytes) {
    const uintptr_t page = 4096, a = reinterpret_cast<uintptr_t>(p) & ~(page - 1), b = (reinterpret_cast<uintptr_t>(p) + bytes + page - 1) & ~(page - 1);
    const size_t total = b - a;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nt = (unsigned)std::min<size_t>(std::min(16u, hw), std::max<size_t>(1, total >> 28));  // one thread per 256 MB, at most 16
    if (nt <= 1) { (void)madvise(reinterpret_cast<void*>(a), total, MADV_POPULATE_WRITE); return; }
    std::vector<std::thread> th;
    const size_t slice = ((total / nt) + page - 1) & ~(size_t)(page - 1);
    for (unsigned i = 0; i < nt; ++i) {
        const size_t off = (size_t)i * slice;
        if (off >= total) break;
        const size_t len = std::min(slice, total - off);
        th.emplace_back([=] { (void)madvise(reinterpret_cast<void*>(a + off), len, MADV_POPULATE_WRITE); });
    }
    for (auto& t : th) t.join();
}

// Several devices, ROWS: device g's thread owns rows [r0, r1) of a column-major matrix = one piece of (r1 - r0) * 4 bytes per column.
// First touch decides which NUMA node a fresh page lands on, and the device's DMA writes should land on the socket its PCIe root hangs
// off: the thread moves itself to the CPUs of the device's NUMA node (sysfs, best effort: containers often hide it) and faults in ITS
// pieces -- no numactl, no policy calls, the kernel's default first-touch placement does the rest.  Pieces shorter than two pages are
// left to the copies (neighbouring devices share those pages anyway).
void adopt_device_numa_node(int device) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return; }
    for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE* f = fopen(path, "r");
    if (!f) return;
    int node = -1;
    const int got = fscanf(f, "%d", &node);
    fclose(f);
    if (got != 1 || node < 0) return;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return;
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) { fclose(f); return; }
    int a = 0, b = 0;
    char sep = ',';
    while (fscanf(f, "%d", &a) == 1) {  // "0-63,128-191"
        b = a;
        int ch = fgetc(f);
        if (ch == '-') { if (fscanf(f, "%d", &b) != 1) break; ch = fgetc(f); }
        for (int c = a; c <= b && c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &allowed)) CPU_SET(c, &want);
        sep = (char)ch;
        if (sep != ',') break;
    }
    fclose(f);
    if (CPU_COUNT(&want) > 0) (void)sched_setaffinity(0, sizeof want, &want);  // this thread only (it ends with the job)
}
void populate_row_block(uint32_t* out, uint64_t ld, uint32_t n_cols, uint32_t r0, uint32_t r1) {
    const uintptr_t page = 4096;
    if ((uint64_t)(r1 - r0) * 4u < 2 * page) return;
    for (uint32_t c = 0; c < n_cols; ++c) {
        const uintptr_t a = (reinterpret_cast<uintptr_t>(out + r0 + ld * c) + page - 1) & ~(page - 1), b = reinterpret_cast<uintptr_t>(out + r1 + ld * c) & ~(page - 1);
        if (b > a && madvise(reinterpret_cast<void*>(a), b - a, MADV_POPULATE_WRITE) != 0) return;  // old kernel: the copies fault the pages in
    }
}

uint32_t chunk_rows_for(rc_scene* s, uint32_t n_cols, uint32_t rows) {  // option "vf_chunk_bytes"
    uint64_t c = (uint64_t)s->opt.vf_chunk_bytes / ((uint64_t)n_cols * 4u);
    if (c >= 64) c &= ~63ull;  // whole 256-byte pieces per column
    if (c < 1) c = 1;
    return (uint32_t)std::min<uint64_t>(c, rows);
}

void status_check(rc_scene* s) {  // after the job's streams have drained
    uint32_t st = 0;
    rc_copy_now(&st, rc_status_word(s), 4, hipMemcpyDeviceToHost);
    if (st) {
        rc_memset_now(rc_status_word(s), 0, 4);
        throw RcError(RC_ERR_STACK_OVERFLOW, "traversal stack overflow (tree deeper than 128 levels)");
    }
}

// Streams, events and device blocks of one device's share of a job.
struct DeviceJob {
    rc_scene* s = nullptr;
    hipStream_t compute[2] = {nullptr, nullptr}, copy = nullptr, comm = nullptr;
    static constexpr int kBlocks = 3;
    DevBuf<uint32_t> block[kBlocks];
    hipEvent_t traced[kBlocks] = {}, copied[kBlocks] = {};
    bool copied_valid[kBlocks] = {};
    hipEvent_t t_begin = nullptr, t_end = nullptr;
    void open(rc_scene* scene) {
        s = scene;
        RC_HIP(hipSetDevice(s->device));
        // the scene's own auxiliary streams, created once: every stream that launches a traversal gets a stack spill region of its own
        // (RcLaunchGuard), so per-call streams would churn through those
        for (auto& a : s->aux_streams) if (!a) RC_HIP(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
        compute[0] = s->aux_streams[0]; compute[1] = s->aux_streams[1]; copy = s->aux_streams[2]; comm = s->aux_streams[3];
        for (int i = 0; i < kBlocks; ++i) {
            RC_HIP(hipEventCreateWithFlags(&traced[i], hipEventDisableTiming));
            RC_HIP(hipEventCreateWithFlags(&copied[i], hipEventDisableTiming));
        }
        RC_HIP(hipEventCreate(&t_begin));
        RC_HIP(hipEventCreate(&t_end));
    }
    ~DeviceJob() {
        if (!s) return;
        (void)hipSetDevice(s->device);
        for (auto c : compute) if (c) (void)hipStreamSynchronize(c);  // (an error path may leave work in flight that uses this job's blocks and events)
        if (copy) (void)hipStreamSynchronize(copy);
        if (comm) (void)hipStreamSynchronize(comm);
        for (int i = 0; i < kBlocks; ++i) { if (traced[i]) (void)hipEventDestroy(traced[i]); if (copied[i]) (void)hipEventDestroy(copied[i]); }
        if (t_begin) (void)hipEventDestroy(t_begin);
        if (t_end) (void)hipEventDestroy(t_end);
    }
};

}  // namespace

// Rows [row_begin, row_end) of the view-factor matrix into a host column-major matrix with leading dimension ld (element (r, c) at
// out[r + ld * c], all n_prims columns): chunks of rows are traced into device blocks on alternating streams -- the tail of one
// chunk's launch overlaps the next chunk's bulk -- and each finished block leaves as one 2-D copy on the copy stream while the
// following chunks are traced.  Blocks the calling thread until the rows are in `out`.
float rc_view_factors_rows_to_host(rc_scene* s, uint32_t rays_per_triangle, uint64_t seed, uint32_t row_begin, uint32_t row_end, uint32_t* out, uint64_t ld) {
    RC_HIP(hipSetDevice(s->device));
    const uint32_t n = s->n_flat_prims;
    if (row_end > n) row_end = n;
    if (row_begin >= row_end || n == 0) return 0.f;
    rc_ensure_vf_order(s);
    const uint32_t rows = row_end - row_begin, C = chunk_rows_for(s, n, rows);
    const uint32_t n_chunks = (rows + C - 1) / C;
    DeviceJob job;
    job.open(s);
    for (int i = 0; i < DeviceJob::kBlocks && (uint32_t)i < n_chunks; ++i) job.block[i].reserve((size_t)C * n);
    auto chunk_range = [&](uint32_t k, uint32_t& r0, uint32_t& r1) { r0 = row_begin + k * C; r1 = std::min(row_end, r0 + C); };
    auto enqueue_trace = [&](uint32_t k) {
        uint32_t r0, r1; chunk_range(k, r0, r1);
        const int b = (int)(k % DeviceJob::kBlocks);
        hipStream_t cs = job.compute[k & 1];
        if (job.copied_valid[b]) RC_HIP(hipStreamWaitEvent(cs, job.copied[b], 0));  // the block's previous chunk has left
        if (k == 0) RC_HIP(hipEventRecord(job.t_begin, cs));
        RC_HIP(hipMemsetAsync(job.block[b].p, 0, sizeof(uint32_t) * (size_t)(r1 - r0) * n, cs));
        uint32_t p0, p1;
        rc_vf_source_range(s, r0, r1, p0, p1);  // the sources whose metadata - 1 lies in [r0, r1): a contiguous range of the metadata order
        if (p1 > p0)
            rc_launch_view_factors(s, rays_per_triangle, seed, p0, p1, 0, rays_per_triangle, job.block[b].p, 1, (uint64_t)(r1 - r0), r0,
                                   RC_VF_SOURCES_BY_METADATA | RC_VF_ROW_BY_METADATA_VALUE, cs);
        RC_HIP(hipEventRecord(job.traced[b], cs));
    };
    auto enqueue_copy = [&](uint32_t k) {  // may block the host until the copy is done when `out` is pageable: enqueue the next chunk's trace first
        uint32_t r0, r1; chunk_range(k, r0, r1);
        const int b = (int)(k % DeviceJob::kBlocks);
        RC_HIP(hipStreamWaitEvent(job.copy, job.traced[b], 0));
        RC_HIP(hipMemcpy2DAsync(out + r0, ld * 4u, job.block[b].p, (size_t)(r1 - r0) * 4u, (size_t)(r1 - r0) * 4u, n, hipMemcpyDeviceToHost, job.copy));
        RC_HIP(hipEventRecord(job.copied[b], job.copy));
        job.copied_valid[b] = true;
    };
    enqueue_trace(0);
    for (uint32_t k = 0; k < n_chunks; ++k) {
        if (k + 1 < n_chunks) enqueue_trace(k + 1);
        enqueue_copy(k);
    }
    RC_HIP(hipEventRecord(job.t_end, job.copy));
    RC_HIP(hipStreamSynchronize(job.copy));
    RC_HIP(hipStreamSynchronize(job.compute[0]));
    RC_HIP(hipStreamSynchronize(job.compute[1]));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, job.t_begin, job.t_end) != hipSuccess) { (void)hipGetLastError(); ms = 0.f; }  // first trace to last copy, on the device's clock
    status_check(s);
    return ms;
}

// ---- RCCL, called directly (no torch): loaded at first use so that the library itself has no link-time dependency on it ----------------
// The six entry points and the enum values are declared by hand in rc_rccl_abi.h and pinned to <rccl/rccl.h> by tests/test_rccl_abi.py.
namespace {
using rc_rccl::comm_t;
struct Rccl {
    void* handle = nullptr;
    rc_rccl::CommInitAllFn CommInitAll = nullptr;
    rc_rccl::CommDestroyFn CommDestroy = nullptr;
    rc_rccl::ReduceFn Reduce = nullptr;
    rc_rccl::GroupStartFn GroupStart = nullptr;
    rc_rccl::GroupEndFn GroupEnd = nullptr;
    rc_rccl::GetErrorStringFn GetErrorString = nullptr;
};
constexpr int kNcclUint32 = rc_rccl::kUint32, kNcclUint64 = rc_rccl::kUint64, kNcclSum = rc_rccl::kSum;
std::mutex g_rccl_mu;
Rccl g_rccl;
// One communicator set per device list, created on first use and kept for the life of the process -- or until a collective on it fails.  A job holds a
// reference for its duration, so a set that another thread's failure takes out of the cache is destroyed only when its last user is done with it.
struct CommSet {
    std::vector<comm_t> comm;
    ~CommSet();
};
std::map<std::vector<int>, std::shared_ptr<CommSet>> g_comms;

// Environment RC_RCCL_LIBRARY names the RCCL build to use (a path or a soname; once a process has mapped some other librccl.so -- torch
// brings its own -- LD_LIBRARY_PATH no longer decides which one dlopen("librccl.so") returns).  When it is set nothing else is tried: a
// caller who asks for a particular library must not silently get another.
Rccl& rccl() {
    if (g_rccl.handle) return g_rccl;
    const char* chosen = getenv("RC_RCCL_LIBRARY");
    if (chosen && chosen[0]) {
        g_rccl.handle = dlopen(chosen, RTLD_NOW | RTLD_GLOBAL);
        if (!g_rccl.handle) throw RcError(RC_ERR_HIP, std::string("RC_RCCL_LIBRARY=") + chosen + " could not be loaded: " + dlerror());
    } else {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.handle) break;
        }
        if (!g_rccl.handle) throw RcError(RC_ERR_HIP, std::string("RC_VF_MODE_RAYS needs RCCL and librccl.so could not be loaded: ") + dlerror());
    }
    void* h = g_rccl.handle;
    auto sym = [&](const char* n) {
        void* p = dlsym(h, n);
        if (!p) { g_rccl.handle = nullptr; throw RcError(RC_ERR_HIP, std::string("the RCCL library lacks ") + n); }
        return p;
    };
    g_rccl.CommInitAll = reinterpret_cast<rc_rccl::CommInitAllFn>(sym("ncclCommInitAll"));
    g_rccl.CommDestroy = reinterpret_cast<rc_rccl::CommDestroyFn>(sym("ncclCommDestroy"));
    g_rccl.Reduce = reinterpret_cast<rc_rccl::ReduceFn>(sym("ncclReduce"));
    g_rccl.GroupStart = reinterpret_cast<rc_rccl::GroupStartFn>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<rc_rccl::GroupEndFn>(sym("ncclGroupEnd"));
    g_rccl.GetErrorString = reinterpret_cast<rc_rccl::GetErrorStringFn>(sym("ncclGetErrorString"));
    return g_rccl;
}
void nccl_ok(int rc, const char* what) {
    if (rc != rc_rccl::kSuccess) throw RcError(RC_ERR_HIP, std::string(what) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"));
}

// The communicator set of a device list (rank g = devs[g]).
std::shared_ptr<CommSet> comms_for(const std::vector<int>& devs) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    Rccl& r = rccl();
    auto it = g_comms.find(devs);
    if (it == g_comms.end()) {
        auto fresh = std::make_shared<CommSet>();
        fresh->comm.resize(devs.size());
        nccl_ok(r.CommInitAll(fresh->comm.data(), (int)devs.size(), devs.data()), "ncclCommInitAll");
        it = g_comms.emplace(devs, fresh).first;
    }
    return it->second;
}
CommSet::~CommSet() {
    if (g_rccl.CommDestroy) for (comm_t c : comm) if (c) (void)g_rccl.CommDestroy(c);
}
// A communicator set whose collective failed is not used again (RCCL leaves it in an undefined state): forget it -- the next call builds a
// fresh one; it is destroyed when the last job that still holds it lets go.
void drop_comms(const std::vector<int>& devs, const std::shared_ptr<CommSet>& failed) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    auto it = g_comms.find(devs);
    if (it != g_comms.end() && it->second == failed) g_comms.erase(it);
}
// ncclGroupStart ... ncclGroupEnd around the per-rank calls of one collective; an exception between the two still closes the group
// (this thread's later RCCL calls would otherwise be queued for ever).
struct RcclGroup {
    Rccl& r;
    bool open = false;
    explicit RcclGroup(Rccl& lib) : r(lib) { nccl_ok(r.GroupStart(), "ncclGroupStart"); open = true; }
    void end() { open = false; nccl_ok(r.GroupEnd(), "ncclGroupEnd"); }
    ~RcclGroup() { if (open) (void)r.GroupEnd(); }
};
// Test hook (only under RC_ENABLE_DEBUG_HOOKS=1, like option debug_set_overflow): RC_DEBUG_RANKS_SHARE_DEVICE=1 lets scenes that live on
// ONE device count as one RCCL rank each, so that a one-GPU box can execute the multi-rank branches below against a stub communicator
// (tests/fake_rccl; real RCCL refuses a device list with duplicates).  Never set in production: replicas on one device are added on
// the host (totals) or refused (RC_VF_MODE_RAYS).
bool ranks_may_share_a_device() {
    const char* hooks = getenv("RC_ENABLE_DEBUG_HOOKS");
    const char* share = getenv("RC_DEBUG_RANKS_SHARE_DEVICE");
    return hooks && hooks[0] == '1' && share && share[0] == '1';
}
bool distinct_devices(rc_scene* const* scenes, int n) {
    if (ranks_may_share_a_device()) return true;  // (the C ABI has already refused the same SCENE twice)
    std::vector<int> d(n);
    for (int g = 0; g < n; ++g) d[g] = scenes[g]->device;
    std::sort(d.begin(), d.end());
    return std::adjacent_find(d.begin(), d.end()) == d.end();
}

void check_same_geometry(rc_scene* const* scenes, int n) {
    for (int g = 0; g < n; ++g) {
        rc_scene* s = scenes[g];
        if (!s) throw RcError(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
        if (!s->has_static || s->dirty || s->transforms_dirty) throw RcError(RC_ERR_NOT_SYNCED, "scene has pending mutations: call rc_sync first");
        if (s->n_flat_prims != scenes[0]->n_flat_prims || s->n_flat_nodes != scenes[0]->n_flat_nodes || s->n_static_instances != scenes[0]->n_static_instances ||
            memcmp(s->root_min, scenes[0]->root_min, 12) != 0 || memcmp(s->root_max, scenes[0]->root_max, 12) != 0)
            throw RcError(RC_ERR_INVALID_ARGUMENT, "the scenes of a multi-device call must hold the same geometry (one copy per device)");
    }
}
}  // namespace

// One host thread per scene (= per device: every thread drives its own device's streams and PCIe link); the first failure is rethrown
// on the calling thread once every thread is back.
template <typename F>
static void for_each_scene(rc_scene* const* scenes, int n_scenes, F&& body) {
    std::vector<std::thread> th;
    std::vector<std::string> err(n_scenes);
    std::vector<int> code(n_scenes, 0);
    for (int g = 0; g < n_scenes; ++g)
        th.emplace_back([&, g] {
            RcCaptureRelaxed relaxed;
            try {
                body(g);
            } catch (const RcError& e) { err[g] = e.what(); code[g] = e.code; }
            catch (const std::exception& e) { err[g] = e.what(); code[g] = RC_ERR_INVALID_ARGUMENT; }
        });
    for (auto& t : th) t.join();
    for (int g = 0; g < n_scenes; ++g)
        if (code[g]) throw RcError(code[g], "device " + std::to_string(scenes[g]->device) + ": " + err[g]);
}

// ROWS: device g owns matrix rows [g N / G, (g + 1) N / G) and brings them home over its own PCIe link; one host thread per device.
static void multi_rows(rc_scene* const* scenes, int n_scenes, uint32_t rays_per_triangle, uint64_t seed, uint32_t* out) {
    const uint32_t n = scenes[0]->n_flat_prims;
    std::vector<float> ms(n_scenes, 0.f);
    for_each_scene(scenes, n_scenes, [&](int g) {
        const uint32_t r0 = (uint32_t)((uint64_t)n * g / n_scenes), r1 = (uint32_t)((uint64_t)n * (g + 1) / n_scenes);
        if (n_scenes > 1 && scenes[g]->opt.vf_first_touch) {
            adopt_device_numa_node(scenes[g]->device);
            populate_row_block(out, n, n, r0, r1);
        }
        ms[g] = rc_view_factors_rows_to_host(scenes[g], rays_per_triangle, seed, r0, r1, out, n);
    });
    for (int g = n_scenes - 1; g >= 0; --g) rc_timing_fixed(scenes[g], ms[g]);  // on the CALLING thread (rc_last_kernel_ms is per thread); scenes[0] last
}

// RAYS: device g shoots ray indices [g R / G, (g + 1) R / G) of EVERY source into a full accumulator (row chunks stored one after the
// other, column-major inside a chunk); chunk k is summed into device 0 by ncclReduce on the communication streams as soon as every
// device has traced it -- while chunk k + 1 is being traced -- and leaves device 0 on its copy stream.
static void multi_rays(rc_scene* const* scenes, int n_scenes, uint32_t rays_per_triangle, uint64_t seed, uint32_t* out) {
    const uint32_t n = scenes[0]->n_flat_prims;
    std::vector<int> devs(n_scenes);
    for (int g = 0; g < n_scenes; ++g) devs[g] = scenes[g]->device;
    if (!distinct_devices(scenes, n_scenes))
        throw RcError(RC_ERR_INVALID_ARGUMENT, "RC_VF_MODE_RAYS reduces over RCCL, which needs one DISTINCT device per scene (use RC_VF_MODE_ROWS for several scenes on one device)");
    const std::shared_ptr<CommSet> comm_set = comms_for(devs);
    const std::vector<comm_t>& comms = comm_set->comm;
    const uint32_t C = chunk_rows_for(scenes[0], n, n), n_chunks = (n + C - 1) / C;
    std::vector<DeviceJob> jobs(n_scenes);
    std::vector<DevBuf<uint32_t>> acc(n_scenes);
    std::vector<std::vector<hipEvent_t>> traced(n_scenes);
    std::vector<hipEvent_t> reduced(n_chunks, nullptr);
    struct EventBin { std::vector<std::pair<int, hipEvent_t>> ev; ~EventBin() { for (auto& e : ev) { (void)hipSetDevice(e.first); (void)hipEventDestroy(e.second); } } } bin;
    for (int g = 0; g < n_scenes; ++g) {
        jobs[g].open(scenes[g]);
        rc_ensure_vf_order(scenes[g]);
        acc[g].reserve((size_t)n * n);
        traced[g].resize(n_chunks);
        for (uint32_t k = 0; k < n_chunks; ++k) { RC_HIP(hipEventCreateWithFlags(&traced[g][k], hipEventDisableTiming)); bin.ev.emplace_back(scenes[g]->device, traced[g][k]); }
    }
    RC_HIP(hipSetDevice(scenes[0]->device));
    for (uint32_t k = 0; k < n_chunks; ++k) { RC_HIP(hipEventCreateWithFlags(&reduced[k], hipEventDisableTiming)); bin.ev.emplace_back(scenes[0]->device, reduced[k]); }
    auto chunk_range = [&](uint32_t k, uint32_t& r0, uint32_t& r1) { r0 = k * C; r1 = std::min(n, r0 + C); };
    auto enqueue_trace = [&](uint32_t k) {
        uint32_t r0, r1; chunk_range(k, r0, r1);
        for (int g = 0; g < n_scenes; ++g) {
            rc_scene* s = scenes[g];
            RC_HIP(hipSetDevice(s->device));
            hipStream_t cs = jobs[g].compute[k & 1];
            if (k == 0 && g == 0) RC_HIP(hipEventRecord(jobs[0].t_begin, cs));
            uint32_t* blk = acc[g].p + (size_t)r0 * n;
            RC_HIP(hipMemsetAsync(blk, 0, sizeof(uint32_t) * (size_t)(r1 - r0) * n, cs));
            uint32_t p0, p1;
            rc_vf_source_range(s, r0, r1, p0, p1);
            const uint32_t q0 = (uint32_t)((uint64_t)rays_per_triangle * g / n_scenes), q1 = (uint32_t)((uint64_t)rays_per_triangle * (g + 1) / n_scenes);
            if (p1 > p0 && q1 > q0)
                rc_launch_view_factors(s, rays_per_triangle, seed, p0, p1, q0, q1, blk, 1, (uint64_t)(r1 - r0), r0, RC_VF_SOURCES_BY_METADATA | RC_VF_ROW_BY_METADATA_VALUE, cs);
            RC_HIP(hipEventRecord(traced[g][k], cs));
        }
    };
    auto enqueue_reduce = [&](uint32_t k) {
        uint32_t r0, r1; chunk_range(k, r0, r1);
        Rccl& r = rccl();
        for (int g = 0; g < n_scenes; ++g) {
            RC_HIP(hipSetDevice(scenes[g]->device));
            RC_HIP(hipStreamWaitEvent(jobs[g].comm, traced[g][k], 0));
        }
        RcclGroup group(r);
        for (int g = 0; g < n_scenes; ++g) {
            uint32_t* blk = acc[g].p + (size_t)r0 * n;
            nccl_ok(r.Reduce(blk, blk, (size_t)(r1 - r0) * n, kNcclUint32, kNcclSum, 0, comms[g], jobs[g].comm), "ncclReduce");
        }
        group.end();
        RC_HIP(hipSetDevice(scenes[0]->device));
        RC_HIP(hipEventRecord(reduced[k], jobs[0].comm));
    };
    auto enqueue_copy = [&](uint32_t k) {  // (copier thread) may block while the chunk travels (pageable `out`)
        uint32_t r0, r1; chunk_range(k, r0, r1);
        RC_HIP(hipSetDevice(scenes[0]->device));
        RC_HIP(hipStreamWaitEvent(jobs[0].copy, reduced[k], 0));
        RC_HIP(hipMemcpy2DAsync(out + r0, (size_t)n * 4u, acc[0].p + (size_t)r0 * n, (size_t)(r1 - r0) * 4u, (size_t)(r1 - r0) * 4u, n, hipMemcpyDeviceToHost, jobs[0].copy));
    };
    // The copies leave from their own host thread: a 2-D copy into PAGEABLE memory blocks its caller while the chunk travels, and the
    // thread that feeds every device's streams must never wait behind device 0's PCIe link (the accumulators are whole matrices, so
    // no trace or reduce ever waits for a copy: the enqueuing thread runs ahead freely; the copier follows the `reduced` events).
    std::atomic<uint32_t> ready{0};   // chunks whose reduce has been enqueued and whose `reduced` event has been recorded
    std::atomic<bool> abort_copies{false};
    std::string copy_err;
    int copy_code = 0;
    std::thread copier([&] {
        RcCaptureRelaxed relaxed;
        try {
            for (uint32_t k = 0; k < n_chunks; ++k) {
                while (ready.load(std::memory_order_acquire) <= k) {
                    if (abort_copies.load(std::memory_order_acquire)) return;
                    std::this_thread::yield();
                }
                enqueue_copy(k);
            }
        } catch (const RcError& e) { copy_err = e.what(); copy_code = e.code; }
        catch (const std::exception& e) { copy_err = e.what(); copy_code = RC_ERR_HIP; }
    });
    try {
        for (uint32_t k = 0; k < n_chunks; ++k) {
            enqueue_trace(k);
            enqueue_reduce(k);
            ready.store(k + 1, std::memory_order_release);
        }
    } catch (...) {
        abort_copies.store(true, std::memory_order_release);
        copier.join();
        drop_comms(devs, comm_set);
        throw;
    }
    copier.join();
    if (copy_code) throw RcError(copy_code, copy_err);
    RC_HIP(hipSetDevice(scenes[0]->device));
    RC_HIP(hipEventRecord(jobs[0].t_end, jobs[0].copy));
    RC_HIP(hipStreamSynchronize(jobs[0].copy));
    for (int g = 0; g < n_scenes; ++g) {
        RC_HIP(hipSetDevice(scenes[g]->device));
        RC_HIP(hipStreamSynchronize(jobs[g].comm));
        RC_HIP(hipStreamSynchronize(jobs[g].compute[0]));
        RC_HIP(hipStreamSynchronize(jobs[g].compute[1]));
        status_check(scenes[g]);
    }
    RC_HIP(hipSetDevice(scenes[0]->device));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, jobs[0].t_begin, jobs[0].t_end) == hipSuccess) rc_timing_fixed(scenes[0], ms);
}

void rc_view_factors_multi_impl(rc_scene* const* scenes, int n_scenes, uint32_t rays_per_triangle, uint64_t seed, uint32_t* out, int mode) {
    if (n_scenes < 1 || !scenes) throw RcError(RC_ERR_INVALID_ARGUMENT, "rc_view_factors_multi: no scenes");
    if (mode != RC_VF_MODE_ROWS && mode != RC_VF_MODE_RAYS) throw RcError(RC_ERR_INVALID_ARGUMENT, "rc_view_factors_multi: mode must be RC_VF_MODE_ROWS or RC_VF_MODE_RAYS");
    check_same_geometry(scenes, n_scenes);
    const uint64_t n = scenes[0]->n_flat_prims;
    if (n == 0) return;
    struct DeviceRestore { int dev = 0; DeviceRestore() { (void)hipGetDevice(&dev); } ~DeviceRestore() { (void)hipSetDevice(dev); } } restore;  // the caller's current device is the caller's
    if (!(mode == RC_VF_MODE_ROWS && n_scenes > 1 && scenes[0]->opt.vf_first_touch)) populate_parallel(out, n * n * 4u);  // (ROWS on several devices: every device's thread faults in its own rows)
    if (mode == RC_VF_MODE_ROWS) multi_rows(scenes, n_scenes, rays_per_triangle, seed, out);
    else multi_rays(scenes, n_scenes, rays_per_triangle, seed, out);
}

// ---- per-triangle totals: the partition under which "rays sharded + RCCL reduce of the per-triangle accumulators" scales ---------------
// What the reference's users read off the matrix are its per-triangle sums (docs/src/viewfactors_content.md:62-68: the column sums,
// "rays that arrive at triangle i").  Those can be accumulated directly: device g shoots ray indices [g R / G, (g + 1) R / G) of EVERY
// source into two N-vectors of u64 (received = column sums, emitted = row sums; ViewFactorTotalsSink), and ONE ncclReduce(sum,
// ncclUint64, 2 N elements -- 0.8 MB at C5) over xGMI brings them to scenes[0]'s device.  No N x N array exists on any device or on the
// host, so neither the 10 GB of PCIe traffic nor its zero fill bound the call: 1 / G of the tracing + one small collective.  Scenes that
// share a device (replicas) cannot form an RCCL communicator; their partial vectors are added on the host instead.
void rc_view_factor_totals_multi_impl(rc_scene* const* scenes, int n_scenes, uint32_t rays_per_triangle, uint64_t seed, uint64_t* out_received, uint64_t* out_emitted) {
    if (n_scenes < 1 || !scenes) throw RcError(RC_ERR_INVALID_ARGUMENT, "rc_view_factor_totals_multi: no scenes");
    check_same_geometry(scenes, n_scenes);
    const uint32_t n = scenes[0]->n_flat_prims;
    if (n == 0) return;
    struct DeviceRestore { int dev = 0; DeviceRestore() { (void)hipGetDevice(&dev); } ~DeviceRestore() { (void)hipSetDevice(dev); } } restore;
    const bool use_rccl = n_scenes > 1 && distinct_devices(scenes, n_scenes);
    std::vector<int> devs(n_scenes);
    for (int g = 0; g < n_scenes; ++g) devs[g] = scenes[g]->device;
    std::shared_ptr<CommSet> comm_set;
    if (use_rccl) comm_set = comms_for(devs);
    static const std::vector<comm_t> no_comms;
    const std::vector<comm_t>& comms = comm_set ? comm_set->comm : no_comms;
    std::vector<hipStream_t> stream(n_scenes, nullptr);
    hipEvent_t t_begin = nullptr, t_end = nullptr;
    RC_HIP(hipSetDevice(scenes[0]->device));
    RC_HIP(hipEventCreate(&t_begin));
    RC_HIP(hipEventCreate(&t_end));
    struct EventPair { int dev; hipEvent_t& a; hipEvent_t& b; ~EventPair() { (void)hipSetDevice(dev); if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } ev_guard{scenes[0]->device, t_begin, t_end};
    // trace: one host thread per device (allocation and launch set-up of G devices side by side); everything is enqueued, nothing waited for
    for_each_scene(scenes, n_scenes, [&](int g) {
        rc_scene* s = scenes[g];
        RC_HIP(hipSetDevice(s->device));
        for (auto& a : s->aux_streams) if (!a) RC_HIP(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
        stream[g] = s->aux_streams[0];
        s->u64_stage.reserve((size_t)2 * n);
        if (g == 0) RC_HIP(hipEventRecord(t_begin, stream[g]));
        RC_HIP(hipMemsetAsync(s->u64_stage.p, 0, sizeof(unsigned long long) * 2u * n, stream[g]));
        const uint32_t q0 = (uint32_t)((uint64_t)rays_per_triangle * g / n_scenes), q1 = (uint32_t)((uint64_t)rays_per_triangle * (g + 1) / n_scenes);
        rc_launch_vf_totals(s, rays_per_triangle, seed, 0, n, q0, q1, s->u64_stage.p, s->u64_stage.p + n, stream[g]);
    });
    std::vector<uint64_t> total((size_t)2 * n, 0);
    if (use_rccl) {  // stream-ordered behind each device's trace: RCCL itself waits for the slowest rank
        Rccl& r = rccl();
        try {
            RcclGroup group(r);
            for (int g = 0; g < n_scenes; ++g)
                nccl_ok(r.Reduce(scenes[g]->u64_stage.p, scenes[g]->u64_stage.p, (size_t)2 * n, kNcclUint64, kNcclSum, 0, comms[g], stream[g]), "ncclReduce");
            group.end();
        } catch (...) {
            for (int g = 0; g < n_scenes; ++g) { (void)hipSetDevice(scenes[g]->device); (void)hipStreamSynchronize(stream[g]); }  // the traces in flight write the scenes' staging vectors
            drop_comms(devs, comm_set);
            throw;
        }
    }
    RC_HIP(hipSetDevice(scenes[0]->device));
    if (use_rccl || n_scenes == 1) {
        RC_HIP(hipEventRecord(t_end, stream[0]));
        RC_HIP(hipMemcpyAsync(total.data(), scenes[0]->u64_stage.p, sizeof(uint64_t) * 2u * n, hipMemcpyDeviceToHost, stream[0]));
        // every stream drains BEFORE anything may throw: `total` is the target of the copy in flight on stream[0] (a status_check
        // that threw for a later device first unwound the vector under the copy)
        hipError_t first_error = hipSuccess;
        for (int g = n_scenes - 1; g >= 0; --g) {
            hipError_t e = hipSetDevice(scenes[g]->device);
            if (e == hipSuccess) e = hipStreamSynchronize(stream[g]);
            if (e != hipSuccess && first_error == hipSuccess) first_error = e;
        }
        RC_HIP(first_error);
        for (int g = n_scenes - 1; g >= 0; --g) { RC_HIP(hipSetDevice(scenes[g]->device)); status_check(scenes[g]); }
    } else {
        std::vector<uint64_t> part((size_t)2 * n);
        for (int g = 0; g < n_scenes; ++g) {
            RC_HIP(hipSetDevice(scenes[g]->device));
            RC_HIP(hipMemcpyAsync(part.data(), scenes[g]->u64_stage.p, sizeof(uint64_t) * 2u * n, hipMemcpyDeviceToHost, stream[g]));
            RC_HIP(hipStreamSynchronize(stream[g]));
            status_check(scenes[g]);
            for (size_t i = 0; i < total.size(); ++i) total[i] += part[i];
        }
        RC_HIP(hipSetDevice(scenes[0]->device));
        RC_HIP(hipEventRecord(t_end, stream[0]));  // every device's share is home: first launch to here, on device 0's clock
    }
    if (out_received) memcpy(out_received, total.data(), sizeof(uint64_t) * n);
    if (out_emitted) memcpy(out_emitted, total.data() + n, sizeof(uint64_t) * n);
    RC_HIP(hipSetDevice(scenes[0]->device));
    RC_HIP(hipEventSynchronize(t_end));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t_begin, t_end) == hipSuccess) rc_timing_fixed(scenes[0], ms); else (void)hipGetLastError();
}

// What a set of scenes needs once before its *_multi calls: see rc_multi_prepare in the header.  Each item is timed on the
// host clock so that bench.py can print the fixed costs next to the timed call instead of inside it.
void rc_multi_prepare_impl(rc_scene* const* scenes, int n_scenes, float out_ms[4]) {
    if (n_scenes < 1 || !scenes) throw RcError(RC_ERR_INVALID_ARGUMENT, "rc_multi_prepare: no scenes");
    check_same_geometry(scenes, n_scenes);
    struct DeviceRestore { int dev = 0; DeviceRestore() { (void)hipGetDevice(&dev); } ~DeviceRestore() { (void)hipSetDevice(dev); } } restore;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    const auto t_all = now();
    const uint32_t n = scenes[0]->n_flat_prims;
    const bool use_rccl = n_scenes > 1 && distinct_devices(scenes, n_scenes);
    std::vector<int> devs(n_scenes);
    for (int g = 0; g < n_scenes; ++g) devs[g] = scenes[g]->device;
    auto t0 = now();
    std::shared_ptr<CommSet> comm_set;
    if (use_rccl) comm_set = comms_for(devs);
    static const std::vector<comm_t> no_comms;
    const std::vector<comm_t>& comms = comm_set ? comm_set->comm : no_comms;
    const float ms_comm = ms_since(t0);
    t0 = now();
    for_each_scene(scenes, n_scenes, [&](int g) {
        rc_scene* s = scenes[g];
        RC_HIP(hipSetDevice(s->device));
        for (auto& a : s->aux_streams) if (!a) RC_HIP(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
        s->u64_stage.reserve((size_t)2 * std::max<uint32_t>(n, 1u));
        RC_HIP(hipMemsetAsync(s->u64_stage.p, 0, sizeof(unsigned long long) * 2u * std::max<uint32_t>(n, 1u), s->aux_streams[0]));
        RC_HIP(hipStreamSynchronize(s->aux_streams[0]));
    });
    const float ms_buffers = ms_since(t0);
    t0 = now();
    if (use_rccl && n > 0) {  // the first collective of a communicator sets up its xGMI connections: do it here, on zeros
        Rccl& r = rccl();
        try {
            RcclGroup group(r);
            for (int g = 0; g < n_scenes; ++g)
                nccl_ok(r.Reduce(scenes[g]->u64_stage.p, scenes[g]->u64_stage.p, (size_t)2 * n, kNcclUint64, kNcclSum, 0, comms[g], scenes[g]->aux_streams[0]), "ncclReduce");
            group.end();
        } catch (...) { drop_comms(devs, comm_set); throw; }
        for (int g = 0; g < n_scenes; ++g) { RC_HIP(hipSetDevice(scenes[g]->device)); RC_HIP(hipStreamSynchronize(scenes[g]->aux_streams[0])); }
    }
    const float ms_warm = use_rccl ? ms_since(t0) : 0.f;
    if (out_ms) { out_ms[0] = ms_comm; out_ms[1] = ms_buffers; out_ms[2] = ms_warm; out_ms[3] = ms_since(t_all); }
}
int rc_multi_ranks_impl(rc_scene* const* scenes, int n_scenes) {
    if (n_scenes < 1 || !scenes) throw RcError(RC_ERR_INVALID_ARGUMENT, "rc_multi_ranks: no scenes");
    for (int g = 0; g < n_scenes; ++g) if (!scenes[g]) throw RcError(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return (n_scenes > 1 && distinct_devices(scenes, n_scenes)) ? n_scenes : 0;
}

// closest_hit / any_hit over one host batch on several devices (rays are independent units -- replicas of the scene, the
// ray array cut into contiguous shards, no collective).  Device g uploads, traces and downloads shard g over its own PCIe link with
// the same three-stage pipeline a single-device call uses (rc_capi.hip), so a host-to-host batch -- bound by the link at 64 bytes per
// ray, not by the kernel -- runs G times as fast.  Shard boundaries are multiples of 64 rays; the hits are the single-device hits.
void rc_trace_multi_impl(rc_scene* const* scenes, int n_scenes, const rc_ray* rays, rc_hit* hits, uint64_t n, int any) {
    if (n_scenes < 1 || !scenes) throw RcError(RC_ERR_INVALID_ARGUMENT, "rc_trace_*_multi: no scenes");
    check_same_geometry(scenes, n_scenes);
    if (n == 0) return;
    if (!rays || !hits) throw RcError(RC_ERR_INVALID_ARGUMENT, "rays/hits is NULL");
    struct DeviceRestore { int dev = 0; DeviceRestore() { (void)hipGetDevice(&dev); } ~DeviceRestore() { (void)hipSetDevice(dev); } } restore;
    if (n_scenes == 1) { rc_trace_host_impl(scenes[0], rays, hits, n, any); return; }
    const uint64_t per = (((n + n_scenes - 1) / n_scenes) + 63) & ~63ull;
    std::vector<float> ms(n_scenes, 0.f);
    for_each_scene(scenes, n_scenes, [&](int g) {
        const uint64_t b = std::min<uint64_t>(n, per * g), e = std::min<uint64_t>(n, per * (g + 1));
        if (e <= b) return;
        rc_trace_host_impl(scenes[g], rays + b, hits + b, e - b, any);
        ms[g] = rc_timing_read(scenes[g]);  // this thread's launch(es) on this scene
    });
    for (int g = n_scenes - 1; g >= 0; --g) rc_timing_fixed(scenes[g], ms[g]);
}

// get_illumination on several devices: device g traces rays [g M / G, (g + 1) M / G) of the grid's M = grid^2 rays into
// its own histogram; the N-length partial histograms are summed on the host (N x 4 bytes per device -- nothing worth a collective).
// A count is a number of rays: the partial counts are integers in f32, exact up to 2^24, where the reference's `+= 1f0`
// (src/kernels.jl:119-121) and the device's f32 atomic add both stop growing -- the sum is clamped there to stay identical.
void rc_illumination_multi_impl(rc_scene* const* scenes, int n_scenes, const float viewdir[3], uint32_t grid, float* out) {
    if (n_scenes < 1 || !scenes) throw RcError(RC_ERR_INVALID_ARGUMENT, "rc_get_illumination_multi: no scenes");
    check_same_geometry(scenes, n_scenes);
    const uint32_t np = scenes[0]->n_flat_prims;
    if (np == 0) return;
    struct DeviceRestore { int dev = 0; DeviceRestore() { (void)hipGetDevice(&dev); } ~DeviceRestore() { (void)hipSetDevice(dev); } } restore;
    const uint64_t m = (uint64_t)grid * grid;
    std::vector<std::vector<float>> part(n_scenes);
    std::vector<float> ms(n_scenes, 0.f);
    for_each_scene(scenes, n_scenes, [&](int g) {
        rc_scene* s = scenes[g];
        const uint64_t b = m * g / n_scenes, e = m * (g + 1) / n_scenes;
        RC_HIP(hipSetDevice(s->device));
        std::lock_guard<std::mutex> one_at_a_time(s->host_call_mu);  // the scene's f32 staging buffer
        part[g].assign(np, 0.f);
        s->f32_stage.reserve(np);
        RC_HIP(hipMemsetAsync(s->f32_stage.p, 0, sizeof(float) * np, s->stream));
        if (e > b) rc_launch_illumination(s, viewdir, grid, b, e, s->f32_stage.p, s->stream);
        RC_HIP(hipMemcpyAsync(part[g].data(), s->f32_stage.p, sizeof(float) * np, hipMemcpyDeviceToHost, s->stream));
        RC_HIP(hipStreamSynchronize(s->stream));
        status_check(s);
        ms[g] = rc_timing_read(s);
    });
    for (uint32_t i = 0; i < np; ++i) {
        double sum = 0.0;
        for (int g = 0; g < n_scenes; ++g) sum += part[g][i];
        out[i] = (float)std::min(sum, 16777216.0);
    }
    for (int g = n_scenes - 1; g >= 0; --g) rc_timing_fixed(scenes[g], ms[g]);
}
