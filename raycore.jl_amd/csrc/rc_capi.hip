// rc_capi.hip -- the C ABI (include/raycore_mi355x.h) and the mutable-TLAS lifecycle behind it.
//
// Host-side mirror of the reference's TLAS management (src/instanced-bvh.jl:334-1102): handles, dirty /
// transforms_dirty flags, delete + compaction, sync! as the sole owner of the adapted (device) form.
// All compute happens in the HIP kernels of rc_build.hip / rc_traverse.hip / rc_drivers.hip; there is no
// CPU fallback -- without a device every entry point that needs one fails with RC_ERR_NO_DEVICE.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <dlfcn.h>
#include <sys/mman.h>

#include "../../include/raycore_mi355x.h"
#include "rc_internal.h"

static_assert(sizeof(rc_ray) == sizeof(RcRay) && sizeof(rc_hit) == sizeof(RcHit), "wire structs");
static_assert(sizeof(rc_bvh_node) == 60 && sizeof(rc_instance_desc) == sizeof(RcInstanceDesc) && sizeof(rc_blas_desc) == sizeof(RcBlasDesc) &&
                  sizeof(rc_prim) == sizeof(RcPrim), "export structs");

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

// Another thread of the process may have a stream capture open in hipStreamCaptureModeGlobal (torch.cuda.graph's default): a
// "potentially unsafe" call from ANY thread -- hipMalloc, hipFree, a blocking copy: what scene creation, rc_sync, the host-buffer queries
// and rc_scene_destroy do -- would then fail AND invalidate that capture.  Every entry point runs with the calling thread's capture
// interaction mode set to relaxed for its duration: its own allocations and copies touch only the scene's streams, never the capturing
// one, so they are safe beside a capture.  (A launch that is itself being captured makes no such call: rc_launch_trace refuses a
// capture that would need one.)
using CaptureRelaxed = RcCaptureRelaxed;  // rc_internal.h

// roctx ranges (SURVEY.md section 5 "Tracing / profiling"): every entry point that goes through guarded() is one range named
// after the entry point, so that `rocprofv3 --marker-trace --kernel-trace` attributes a run's dispatches to the calls that made them; a
// caller brackets its own phases with rc_range_push / rc_range_pop (bench.py does, per extra).  The marker library is looked up at run time
// -- rocprofiler-sdk's roctx first (what rocprofv3 listens to), the legacy libroctx64 second -- and never linked: without it, or with
// RC_ROCTX=0, a range costs one predictable branch.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* off = getenv("RC_ROCTX");
        if (off && off[0] == '0') return;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void* h = dlopen(name, RTLD_LAZY | RTLD_GLOBAL);
            if (!h) continue;
            push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
const Roctx& roctx() { static const Roctx r; return r; }
struct RcRange {
    bool on;
    explicit RcRange(const char* name) : on(roctx().push != nullptr) { if (on) (void)roctx().push(name); }
    ~RcRange() { if (on) (void)roctx().pop(); }
    RcRange(const RcRange&) = delete;
    RcRange& operator=(const RcRange&) = delete;
};

template <typename F>
int guarded(F&& f, const char* entry_point = __builtin_FUNCTION()) {
    CaptureRelaxed relaxed;
    RcRange range(entry_point);
    try {
        f();
        return RC_OK;
    } catch (const RcError& e) {
        return fail(e.code, e.what());
    } catch (const std::exception& e) {
        return fail(RC_ERR_INVALID_ARGUMENT, e.what());
    }
}

void use_device(rc_scene* s) { RC_HIP(hipSetDevice(s->device)); }

const float kIdentity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};

HandleRange& live_range(rc_scene* s, uint32_t handle) {
    auto it = s->handle_to_range.find(handle);
    if (it == s->handle_to_range.end()) throw RcError(RC_ERR_INVALID_HANDLE, "Invalid handle");
    if (s->deleted_handles.count(handle)) throw RcError(RC_ERR_INVALID_HANDLE, "Handle has been deleted");
    return it->second;
}

// compact_instances! (src/instanced-bvh.jl:996-1065).  The reference iterates a Dict (order unspecified);
// here surviving handles keep ascending handle-id order.
void compact_instances(rc_scene* s) {
    std::vector<RcInstanceDesc> fresh;
    std::map<uint32_t, HandleRange> ranges;
    for (auto& kv : s->handle_to_range) {
        if (s->deleted_handles.count(kv.first)) continue;
        HandleRange r{(uint32_t)fresh.size(), kv.second.count};
        fresh.insert(fresh.end(), s->instances.begin() + kv.second.first, s->instances.begin() + kv.second.first + kv.second.count);
        ranges[kv.first] = r;
    }
    s->deleted_handles.clear();
    s->handle_to_range.swap(ranges);
    std::set<uint32_t> used;
    for (auto& in : fresh) used.insert(in.blas_index);
    if (!s->blas.empty() && used.size() < s->blas.size()) {
        std::map<uint32_t, uint32_t> old_to_new;
        std::vector<Blas> kept;
        for (uint32_t old_idx : used) {  // ascending, like sort!(collect(used_blas_indices))
            old_to_new[old_idx] = (uint32_t)kept.size() + 1;
            kept.push_back(std::move(s->blas[old_idx - 1]));
        }
        for (auto& in : fresh) in.blas_index = old_to_new[in.blas_index];
        s->blas.swap(kept);  // dropped geometries free their device buffers here
    }
    s->instances.swap(fresh);
}

// The host mirror of tlas.instances is authoritative except after a device-side rewrite (rc_refit_device): pull it back
// before anything reads or edits it.
void sync_host_instances(rc_scene* s) {
    if (!s->host_instances_stale) return;
    RC_HIP(hipSetDevice(s->device));
    RC_HIP(hipStreamSynchronize(s->stream));
    if (!s->instances.empty())
        rc_copy_now(s->instances.data(), s->d_instances.p, sizeof(RcInstanceDesc) * s->instances.size(), hipMemcpyDeviceToHost);
    s->host_instances_stale = false;
}

void require_synced(rc_scene* s) {
    if (!s->has_static || s->dirty || s->transforms_dirty)
        throw RcError(RC_ERR_NOT_SYNCED, "scene has pending mutations: call rc_sync before tracing (Adapt.adapt does this per dispatch)");
}

// The scene's stack-overflow word is sticky: kernels only ever set it, and it is cleared here when it is reported -- so a launch can
// never clear another launch's report, and an overflow in an earlier asynchronous launch is reported by the next call that looks.
void check_status(rc_scene* s, hipStream_t stream) {
    uint32_t st = 0;
    if (!s->counters.p) return;
    RC_HIP(hipMemcpyAsync(&st, rc_status_word(s), 4, hipMemcpyDeviceToHost, stream));
    RC_HIP(hipStreamSynchronize(stream));
    if (st) {
        RC_HIP(hipMemsetAsync(rc_status_word(s), 0, 4, stream));
        RC_HIP(hipStreamSynchronize(stream));
        throw RcError(RC_ERR_STACK_OVERFLOW, "traversal stack overflow (tree deeper than 128 levels) in this or an earlier asynchronous launch");
    }
}

// A staging context for one host-buffer trace call (CallCtx, rc_internal.h): the first one runs on the scene's own stream, the others
// on streams of their own; a call that finds all kMaxCallCtx busy waits for one.
struct CtxLease {
    rc_scene* s;
    CallCtx* c = nullptr;
    explicit CtxLease(rc_scene* scene) : s(scene) {
        std::unique_lock<std::mutex> lk(s->ctx_mu);
        for (;;) {
            for (auto& p : s->call_ctx) if (!p->busy) { c = p.get(); break; }
            if (c) break;
            if ((int)s->call_ctx.size() < rc_scene::kMaxCallCtx) {
                std::unique_ptr<CallCtx> fresh(new CallCtx());
                if (s->call_ctx.empty()) fresh->stream = s->stream;
                else { RC_HIP(hipStreamCreateWithFlags(&fresh->stream, hipStreamNonBlocking)); fresh->own_stream = true; }
                s->call_ctx.push_back(std::move(fresh));
                c = s->call_ctx.back().get();
                break;
            }
            s->ctx_cv.wait(lk);
        }
        c->busy = true;
    }
    ~CtxLease() {
        { std::lock_guard<std::mutex> lk(s->ctx_mu); c->busy = false; }
        s->ctx_cv.notify_one();
    }
};

void export_nodes(rc_scene* s, const RcNode* d, uint32_t n, rc_bvh_node* out, uint32_t capacity, uint32_t* count) {
    if (count) *count = n;
    if (!out || n == 0) return;
    if (capacity < n) throw RcError(RC_ERR_INVALID_ARGUMENT, "export buffer too small");
    std::vector<RcNode> tmp(n);
    rc_copy_now(tmp.data(), d, sizeof(RcNode) * n, hipMemcpyDeviceToHost);
    for (uint32_t i = 0; i < n; ++i) {
        memcpy(&out[i], tmp[i].f, 48);
        out[i].child0 = tmp[i].child0; out[i].child1 = tmp[i].child1; out[i].parent = tmp[i].parent;
    }
}

}  // namespace

namespace {
constexpr char kSceneMagic[8] = {'R', 'C', 'M', 'I', '3', '5', '5', 'X'};
constexpr uint32_t kSceneVersion = 1;

struct FileCloser { FILE* f; ~FileCloser() { if (f) fclose(f); } };
void put(FILE* f, const void* p, size_t n) { if (n && fwrite(p, 1, n, f) != n) throw RcError(RC_ERR_INVALID_ARGUMENT, "scene file: write failed"); }
void get(FILE* f, void* p, size_t n) { if (n && fread(p, 1, n, f) != n) throw RcError(RC_ERR_INVALID_ARGUMENT, "scene file: truncated"); }
template <typename T> void put_dev(FILE* f, const T* d, size_t n, std::vector<unsigned char>& tmp) {
    if (!n) return;
    tmp.resize(n * sizeof(T));
    rc_copy_now(tmp.data(), d, n * sizeof(T), hipMemcpyDeviceToHost);
    put(f, tmp.data(), tmp.size());
}
// `check(host copy)` runs before the upload: a scene file is untrusted input, and the kernels index with what it holds
void bad_file(const char* what);
// bytes left between the read position and the end of the file: every count a header states is checked against it BEFORE anything is
// allocated for it (a scene file is untrusted input; a forged count must not drive a multi-gigabyte allocation)
size_t bytes_left(FILE* f) {
    const long here = ftell(f);
    if (here < 0 || fseek(f, 0, SEEK_END) != 0) bad_file("cannot seek");
    const long end = ftell(f);
    if (end < here || fseek(f, here, SEEK_SET) != 0) bad_file("cannot seek");
    return (size_t)(end - here);
}
template <typename T, typename Check> void get_dev(FILE* f, DevBuf<T>& d, size_t n, std::vector<unsigned char>& tmp, Check&& check) {
    if (n > bytes_left(f) / sizeof(T)) bad_file("truncated (an array is longer than the rest of the file)");
    d.reserve(n ? n : 1);
    if (!n) return;
    tmp.resize(n * sizeof(T));
    get(f, tmp.data(), tmp.size());
    check(reinterpret_cast<const T*>(tmp.data()));
    rc_copy_now(d.p, tmp.data(), n * sizeof(T), hipMemcpyHostToDevice);
}
template <typename T> void get_dev(FILE* f, DevBuf<T>& d, size_t n, std::vector<unsigned char>& tmp) { get_dev(f, d, n, tmp, [](const T*) {}); }
}  // namespace
namespace {
void bad_file(const char* what) { throw RcError(RC_ERR_INVALID_ARGUMENT, std::string("scene file: ") + what); }
}  // namespace

extern "C" {

const char* rc_last_error(void) { return g_last_error.c_str(); }

int rc_range_push(const char* name) {
    if (!name) return fail(RC_ERR_INVALID_ARGUMENT, "rc_range_push: null name");
    if (roctx().push) (void)roctx().push(name);
    return RC_OK;
}
int rc_range_pop(void) {
    if (roctx().pop) (void)roctx().pop();
    return RC_OK;
}
int rc_ranges_enabled(void) { return roctx().push != nullptr ? 1 : 0; }

int rc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rc_scene_create(int device, rc_scene** out) {
    if (!out) return fail(RC_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int n = rc_device_count();
    if (n <= 0) return fail(RC_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU fallback");
    if (device < 0 || device >= n) return fail(RC_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    rc_scene* s = nullptr;
    int rc = guarded([&] {
        s = new rc_scene();
        s->device = device;
        use_device(s);
        hipDeviceProp_t prop;
        RC_HIP(hipGetDeviceProperties(&prop, device));
        s->n_cus = prop.multiProcessorCount;
        RC_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        RC_HIP(hipEventCreate(&s->ev0));
        RC_HIP(hipEventCreate(&s->ev1));
        static std::atomic<uint64_t> next_uid{1};
        s->uid = next_uid.fetch_add(1);
        // claim counters, status word and statistics start at zero, and are zero before anything can launch: a launch on any stream --
        // or the replay of a captured one -- never meets uninitialised counters (a memset enqueued by the first launch would order only that launch's stream)
        s->counters.reserve((size_t)kCounterSlots * kCounterSlotWords);
        RC_HIP(hipMemsetAsync(s->counters.p, 0, sizeof(uint32_t) * (size_t)kCounterSlots * kCounterSlotWords, s->stream));
        RC_HIP(hipStreamSynchronize(s->stream));
        s->slots.assign(kCounterSlots + 1, rc_scene::LaunchSlot());
        if (const char* e = getenv("RC_ENTRY_CULL")) s->opt.entry_cull = e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1);  // test campaigns: the default of option "entry_cull" for every scene of the process
        if (const char* e = getenv("RC_STACK16")) s->opt.stack16 = e[0] == '0' ? 0 : 1;  // ... and of option "stack16"
    });
    if (rc != RC_OK) { delete s; return rc; }
    *out = s;
    return RC_OK;
}

int rc_scene_destroy(rc_scene* s) {
    if (!s) return RC_OK;
    CaptureRelaxed relaxed;  // a finaliser may run this while another thread captures a graph (see guarded())
    (void)hipSetDevice(s->device);
    // Wait for exactly the work that may still read the scene: the launches it recorded on the callers' streams (launch slots, stage
    // kernels) and its own streams.  Not hipDeviceSynchronize: HIP refuses that while any stream of the process is being captured, and
    // the refusal invalidates the capture.  (Replays of graphs that captured launches on this scene are the caller's to wait for.)
    for (auto& kv : s->stage_events) { (void)hipEventSynchronize(kv.second); (void)hipEventDestroy(kv.second); }
    s->stage_events.clear();
    for (int i = 0; i < (int)s->slots.size(); ++i)
        if (s->slots[i].recorded && s->slots[i].t1) (void)hipEventSynchronize(s->slots[i].t1);
    for (auto& c : s->call_ctx) if (c && c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto a : s->aux_streams) if (a) (void)hipStreamSynchronize(a);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    (void)hipGetLastError();
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    for (int i = 0; i < kCounterSlots && i < (int)s->slots.size(); ++i) {  // (the last entry aliases ev0 / ev1)
        if (s->slots[i].t0) (void)hipEventDestroy(s->slots[i].t0);
        if (s->slots[i].t1) (void)hipEventDestroy(s->slots[i].t1);
    }
    for (auto& c : s->call_ctx) if (c && c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    for (auto a : s->aux_streams) if (a) (void)hipStreamDestroy(a);
    if (s->batch_fork) (void)hipEventDestroy(s->batch_fork);
    for (auto e : s->batch_join) if (e) (void)hipEventDestroy(e);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return RC_OK;
}

// faces (device pointers) -> degenerate filter -> LBVH, all on the device
static void build_from_device_faces(rc_scene* s, const float* d_verts, const uint32_t* d_meta, uint32_t n, Blas& b) {
    if (n && !d_verts) throw RcError(RC_ERR_INVALID_ARGUMENT, "verts is NULL");
    uint32_t valid = rc_ingest_faces(s, d_verts, d_meta, n);
    if (valid == 0) throw RcError(RC_ERR_EMPTY_GEOMETRY, "Geometry has no valid triangles");  // :601
    rc_build_blas(s, valid, b);
}

// host soup -> staging buffers on the device
static void stage_faces(rc_scene* s, const float* verts, const uint32_t* meta, uint32_t n) {
    if (n && !verts) throw RcError(RC_ERR_INVALID_ARGUMENT, "verts is NULL");
    s->vert_stage.reserve(9 * (size_t)(n ? n : 1));
    if (n) RC_HIP(hipMemcpyAsync(s->vert_stage.p, verts, sizeof(float) * 9 * (size_t)n, hipMemcpyHostToDevice, s->stream));
    if (meta) {
        s->meta_stage.reserve(n ? n : 1);
        if (n) RC_HIP(hipMemcpyAsync(s->meta_stage.p, meta, sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice, s->stream));
    }
}

int rc_add_blas(rc_scene* s, const float* verts, const uint32_t* meta, uint32_t n, uint32_t* blas_id) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        stage_faces(s, verts, meta, n);
        Blas b;
        build_from_device_faces(s, s->vert_stage.p, meta ? s->meta_stage.p : nullptr, n, b);
        s->blas.push_back(std::move(b));
        if (blas_id) *blas_id = (uint32_t)s->blas.size() - 1;
    });
}

int rc_add_blas_device(rc_scene* s, const float* d_verts, const uint32_t* d_meta, uint32_t n, uint32_t* blas_id) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        Blas b;
        build_from_device_faces(s, d_verts, d_meta, n, b);
        s->blas.push_back(std::move(b));
        if (blas_id) *blas_id = (uint32_t)s->blas.size() - 1;
    });
}

// build_and_append_blas! body for a decomposed mesh (src/instanced-bvh.jl:581-600): upload, expand by index, filter, build
// face_meta: one word per VERTEX (read at each face's first vertex, :595) or, with meta_per_face, one word per FACE (:2300-2306)
static void build_mesh_blas(rc_scene* s, const float* verts, const float* normals, const float* uvs, uint32_t nv, const uint32_t* indices, uint32_t nf,
                            const uint32_t* face_meta, bool meta_per_face, Blas& b) {
    if ((nv && (!verts || !normals)) || (nf && !indices)) throw RcError(RC_ERR_INVALID_ARGUMENT, "verts / normals / indices is NULL");
    for (size_t i = 0; i < 3 * (size_t)nf; ++i)
        if (indices[i] >= nv) throw RcError(RC_ERR_INVALID_ARGUMENT, "face index out of range");
    DevBuf<float> d_verts;
    DevBuf<uint32_t> d_vmeta;
    d_verts.reserve(3 * (size_t)(nv ? nv : 1));
    b.m_normals.reserve(3 * (size_t)(nv ? nv : 1));
    b.m_indices.reserve(3 * (size_t)(nf ? nf : 1));
    if (nv) {
        RC_HIP(hipMemcpyAsync(d_verts.p, verts, sizeof(float) * 3 * (size_t)nv, hipMemcpyHostToDevice, s->stream));
        RC_HIP(hipMemcpyAsync(b.m_normals.p, normals, sizeof(float) * 3 * (size_t)nv, hipMemcpyHostToDevice, s->stream));
    }
    if (uvs && nv) {
        b.m_uvs.reserve(2 * (size_t)nv);
        RC_HIP(hipMemcpyAsync(b.m_uvs.p, uvs, sizeof(float) * 2 * (size_t)nv, hipMemcpyHostToDevice, s->stream));
        b.has_uvs = true;
    }
    const size_t n_meta = meta_per_face ? nf : nv;
    if (face_meta && n_meta) {
        d_vmeta.reserve(n_meta);
        RC_HIP(hipMemcpyAsync(d_vmeta.p, face_meta, sizeof(uint32_t) * n_meta, hipMemcpyHostToDevice, s->stream));
    }
    if (nf) RC_HIP(hipMemcpyAsync(b.m_indices.p, indices, sizeof(uint32_t) * 3 * (size_t)nf, hipMemcpyHostToDevice, s->stream));
    s->vert_stage.reserve(9 * (size_t)(nf ? nf : 1));
    s->meta_stage.reserve(nf ? nf : 1);
    rc_expand_mesh(s, d_verts.p, b.m_indices.p, face_meta && n_meta ? d_vmeta.p : nullptr, meta_per_face, nf, s->vert_stage.p, s->meta_stage.p);
    const uint32_t valid = rc_ingest_faces(s, s->vert_stage.p, s->meta_stage.p, nf, true);
    if (valid == 0) throw RcError(RC_ERR_EMPTY_GEOMETRY, "Geometry has no valid triangles");  // :601
    rc_build_blas(s, valid, b, true);  // synchronises the stream: the temporaries above may go
    b.has_attrs = true;
    b.n_mesh_verts = nv; b.n_mesh_faces = nf;
}

int rc_add_mesh(rc_scene* s, const float* verts, const float* normals, const float* uvs, uint32_t nv, const uint32_t* indices, uint32_t nf,
                const uint32_t* face_meta, uint32_t* blas_id) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        Blas b;
        build_mesh_blas(s, verts, normals, uvs, nv, indices, nf, face_meta, false, b);
        s->blas.push_back(std::move(b));
        if (blas_id) *blas_id = (uint32_t)s->blas.size() - 1;
    });
}

int rc_add_mesh_face_metadata(rc_scene* s, const float* verts, const float* normals, const float* uvs, uint32_t nv, const uint32_t* indices, uint32_t nf,
                              const uint32_t* metadata_per_face, uint32_t* blas_id) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        Blas b;
        build_mesh_blas(s, verts, normals, uvs, nv, indices, nf, metadata_per_face, true, b);
        s->blas.push_back(std::move(b));
        if (blas_id) *blas_id = (uint32_t)s->blas.size() - 1;
    });
}

int rc_update_geometry_mesh(rc_scene* s, uint32_t handle, const float* verts, const float* normals, const float* uvs, uint32_t nv, const uint32_t* indices,
                            uint32_t nf, const uint32_t* face_meta) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        sync_host_instances(s);
        use_device(s);
        HandleRange& r = live_range(s, handle);
        if (r.count == 0) throw RcError(RC_ERR_INVALID_HANDLE, "Handle has no instances");
        const uint32_t blas_idx = s->instances[r.first].blas_index;  // :814-816
        Blas b;
        build_mesh_blas(s, verts, normals, uvs, nv, indices, nf, face_meta, false, b);
        s->blas[blas_idx - 1] = std::move(b);
        s->dirty = true;
    });
}

int rc_export_triangles(rc_scene* s, rc_triangle* out, uint32_t capacity, uint32_t* count) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        if (count) *count = s->n_flat_prims;
        if (!out || s->n_flat_prims == 0) return;
        if (capacity < s->n_flat_prims) throw RcError(RC_ERR_INVALID_ARGUMENT, "export buffer too small");
        DevBuf<uint32_t> tmp;
        tmp.reserve(34 * (size_t)s->n_flat_prims);
        rc_launch_export_triangles(s, tmp.p, s->stream);
        RC_HIP(hipMemcpyAsync(out, tmp.p, sizeof(rc_triangle) * (size_t)s->n_flat_prims, hipMemcpyDeviceToHost, s->stream));
        RC_HIP(hipStreamSynchronize(s->stream));
    });
}

int rc_shading_attributes_device(rc_scene* s, const rc_hit* d_hits, uint64_t n, float* d_normals, float* d_uvs, void* stream) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        if (n && !d_hits) throw RcError(RC_ERR_INVALID_ARGUMENT, "hits is NULL");
        rc_launch_shading_attributes(s, reinterpret_cast<const RcHit*>(d_hits), n, d_normals, d_uvs, (hipStream_t)stream);
    });
}

int rc_add_instances_with_inverse(rc_scene* s, uint32_t blas_id, const float* xforms, const float* inv_xforms,
                                  const uint32_t* instance_ids, uint32_t m, uint32_t* handle) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        sync_host_instances(s);
        if (blas_id >= s->blas.size()) throw RcError(RC_ERR_INVALID_ARGUMENT, "blas_id out of range");
        HandleRange r{(uint32_t)s->instances.size(), m};
        for (uint32_t i = 0; i < m; ++i) {  // :670-674
            RcInstanceDesc d;
            d.blas_index = blas_id + 1;
            d.instance_id = instance_ids ? instance_ids[i] : 0u;
            memcpy(d.transform, xforms ? xforms + 12 * (size_t)i : kIdentity, 48);
            if (inv_xforms) memcpy(d.inv_transform, inv_xforms + 12 * (size_t)i, 48);
            else rc_mat3x4_inverse(d.transform, d.inv_transform);
            d.flags = 0;
            s->instances.push_back(d);
        }
        uint32_t h = s->next_handle_id++;  // append_instances_with_handle! (:612-623)
        s->handle_to_range[h] = r;
        s->dirty = true;
        if (handle) *handle = h;
    });
}

int rc_add_instances(rc_scene* s, uint32_t blas_id, const float* xforms, const uint32_t* instance_ids, uint32_t m, uint32_t* handle) {
    return rc_add_instances_with_inverse(s, blas_id, xforms, nullptr, instance_ids, m, handle);
}

int rc_update_transforms(rc_scene* s, uint32_t handle, const float* xforms, uint32_t m) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        sync_host_instances(s);
        HandleRange& r = live_range(s, handle);
        if (m != r.count) throw RcError(RC_ERR_INVALID_ARGUMENT, "Transform count (" + std::to_string(m) + ") != instance count (" + std::to_string(r.count) + ")");
        if (!xforms) throw RcError(RC_ERR_INVALID_ARGUMENT, "xforms is NULL");
        for (uint32_t i = 0; i < m; ++i) {  // update_instance_transforms_offset_kernel! (src/instanced-bvh-kernels.jl:455-476)
            RcInstanceDesc& d = s->instances[r.first + i];
            memcpy(d.transform, xforms + 12 * (size_t)i, 48);
            rc_mat3x4_inverse(d.transform, d.inv_transform);
        }
        s->transforms_dirty = true;
    });
}

int rc_update_geometry(rc_scene* s, uint32_t handle, const float* verts, const uint32_t* meta, uint32_t n) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        sync_host_instances(s);
        use_device(s);
        HandleRange& r = live_range(s, handle);
        if (r.count == 0) throw RcError(RC_ERR_INVALID_HANDLE, "Handle has no instances");
        uint32_t blas_idx = s->instances[r.first].blas_index;  // :814-816
        stage_faces(s, verts, meta, n);
        Blas b;
        build_from_device_faces(s, s->vert_stage.p, meta ? s->meta_stage.p : nullptr, n, b);
        s->blas[blas_idx - 1] = std::move(b);
        s->dirty = true;
    });
}

int rc_delete(rc_scene* s, uint32_t handle, int* deleted) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    int d = 0;
    if (s->handle_to_range.count(handle) && !s->deleted_handles.count(handle)) {  // :690-699
        s->deleted_handles.insert(handle);
        s->dirty = true;
        d = 1;
    }
    if (deleted) *deleted = d;
    return RC_OK;
}

int rc_is_valid(rc_scene* s, uint32_t handle, int* valid) {
    if (!s || !valid) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    *valid = s->handle_to_range.count(handle) && !s->deleted_handles.count(handle);
    return RC_OK;
}

int rc_handle_instance_count(rc_scene* s, uint32_t handle, uint32_t* count) {
    if (!s || !count) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    auto it = s->handle_to_range.find(handle);
    *count = (it == s->handle_to_range.end() || s->deleted_handles.count(handle)) ? 0u : it->second.count;  // :533-537
    return RC_OK;
}

int rc_get_instances(rc_scene* s, uint32_t handle, rc_instance_desc* out, uint32_t capacity, uint32_t* count) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        sync_host_instances(s);
        HandleRange& r = live_range(s, handle);
        if (count) *count = r.count;
        if (!out) return;
        if (capacity < r.count) throw RcError(RC_ERR_INVALID_ARGUMENT, "buffer too small");
        memcpy(out, s->instances.data() + r.first, sizeof(RcInstanceDesc) * r.count);
    });
}

int rc_sync(rc_scene* s, int* action) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (action) *action = 0;
    if (!s->dirty && !s->transforms_dirty && s->has_static) return RC_OK;  // :898-900, no device sync
    return guarded([&] {
        use_device(s);
        sync_host_instances(s);
        if (s->dirty || !s->has_static) {  // rebuild_bvh! + rebuild_static_tlas! (:902-905, :911-915)
            if (!s->deleted_handles.empty()) compact_instances(s);
            rc_build_tlas(s);
            s->dirty = false;
            s->transforms_dirty = false;
            s->has_static = true;
            if (action) *action = 2;
        } else {  // refit_tlas! (:906-911): in place, the adapted buffers keep their identity
            rc_refit_tlas(s);
            s->transforms_dirty = false;
            if (action) *action = 1;
        }
    });
}

int rc_counts(rc_scene* s, uint32_t* n_live, uint32_t* n_total, uint32_t* n_geom, uint32_t* n_prims, uint32_t* n_tlas_nodes, uint32_t* n_blas_nodes) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    uint32_t pending = 0;
    for (uint32_t h : s->deleted_handles) {
        auto it = s->handle_to_range.find(h);
        if (it != s->handle_to_range.end()) pending += it->second.count;
    }
    if (n_live) *n_live = (uint32_t)s->instances.size() - pending;  // :2391-2398
    if (n_total) *n_total = (uint32_t)s->instances.size();
    if (n_geom) *n_geom = (uint32_t)s->blas.size();
    if (n_prims) *n_prims = s->has_static ? s->n_flat_prims : 0;
    if (n_tlas_nodes) *n_tlas_nodes = s->has_static ? s->n_tlas_nodes : 0;
    if (n_blas_nodes) *n_blas_nodes = s->has_static ? s->n_flat_nodes : 0;
    return RC_OK;
}

int rc_world_bound(rc_scene* s, float out[6]) {
    if (!s || !out) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    memcpy(out, s->root_min, 12);
    memcpy(out + 3, s->root_max, 12);
    return RC_OK;
}

int rc_wait(rc_scene* s) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        RC_HIP(hipDeviceSynchronize());
        // asynchronous launches (the *_device entry points) cannot report a traversal-stack overflow themselves: the scene's sticky
        // status word is read (and cleared) here
        if (s->counters.p) {
            uint32_t st = 0;
            rc_copy_now(&st, rc_status_word(s), 4, hipMemcpyDeviceToHost);
            if (st) {
                rc_memset_now(rc_status_word(s), 0, 4);
                throw RcError(RC_ERR_STACK_OVERFLOW, "traversal stack overflow in an earlier asynchronous launch (tree deeper than 128 levels)");
            }
        }
    });
}

int rc_export_tlas_nodes(rc_scene* s, rc_bvh_node* out, uint32_t capacity, uint32_t* count) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] { use_device(s); require_synced(s); export_nodes(s, s->tlas_nodes.p, s->n_tlas_nodes, out, capacity, count); });
}
int rc_export_blas_nodes(rc_scene* s, rc_bvh_node* out, uint32_t capacity, uint32_t* count) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        if (count) *count = s->n_flat_nodes;
        if (!out || s->n_flat_nodes == 0) return;
        if (capacity < s->n_flat_nodes) throw RcError(RC_ERR_INVALID_ARGUMENT, "export buffer too small");
        for (size_t i = 0; i < s->blas.size(); ++i)  // the flat device array is the packed traversal copy; the reference layout lives per BLAS
            export_nodes(s, s->blas[i].nodes.p, s->blas[i].n_nodes, out + s->descs[i].nodes_offset, s->blas[i].n_nodes, nullptr);
    });
}
int rc_export_instances(rc_scene* s, rc_instance_desc* out, uint32_t capacity, uint32_t* count) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        sync_host_instances(s);
        require_synced(s);
        if (count) *count = (uint32_t)s->instances.size();
        if (!out) return;
        if (capacity < s->instances.size()) throw RcError(RC_ERR_INVALID_ARGUMENT, "export buffer too small");
        if (!s->instances.empty()) memcpy(out, s->instances.data(), sizeof(RcInstanceDesc) * s->instances.size());
    });
}
int rc_export_blas_descs(rc_scene* s, rc_blas_desc* out, uint32_t capacity, uint32_t* count) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        require_synced(s);
        if (count) *count = (uint32_t)s->descs.size();
        if (!out) return;
        if (capacity < s->descs.size()) throw RcError(RC_ERR_INVALID_ARGUMENT, "export buffer too small");
        if (!s->descs.empty()) memcpy(out, s->descs.data(), sizeof(RcBlasDesc) * s->descs.size());
    });
}
int rc_export_prims(rc_scene* s, rc_prim* out, uint32_t capacity, uint32_t* count) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        if (count) *count = s->n_flat_prims;
        if (!out || s->n_flat_prims == 0) return;
        if (capacity < s->n_flat_prims) throw RcError(RC_ERR_INVALID_ARGUMENT, "export buffer too small");
        rc_copy_now(out, s->flat_prims.p, sizeof(RcPrim) * s->n_flat_prims, hipMemcpyDeviceToHost);
    });
}

// A freshly allocated output array has no pages yet; faulting them in one by one from inside the transfer threads (which contend for
// the process's memory-map lock) is several times slower than asking the kernel for the whole range up front.
static void populate_pages(void* p, size_t bytes) {
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23  // Linux 5.14+
#endif
    const uintptr_t page = 4096, a = reinterpret_cast<uintptr_t>(p) & ~(page - 1), b = (reinterpret_cast<uintptr_t>(p) + bytes + page - 1) & ~(page - 1);
    (void)madvise(reinterpret_cast<void*>(a), b - a, MADV_POPULATE_WRITE);  // best effort: older kernels return EINVAL and the copies fault the pages in
}

// Host-buffer batches of three million rays or more (below that the chunking costs more than the overlap gains): upload, trace and download run as a three-stage pipeline over 512 Ki-ray chunks
// (16 MiB each way), each transfer direction on its own host thread and non-blocking stream, so the two directions of the link and
// the kernel overlap: a 4 M-ray batch takes about one direction's transfer time (2.4 ms at 56 GB/s) plus one chunk's latency instead
// of upload + kernel + download back to back.  The chunks are traced by the same kernels, so the results do not change.
static void trace_host_pipelined(rc_scene* s, CallCtx& cx, const rc_ray* rays, rc_hit* hits, uint64_t n, int any) {
    // at most 48 chunks of at least 512 Ki rays
    const uint64_t kChunk = std::max<uint64_t>(1ull << 19, ((n + 47) / 48 + 63) & ~63ull);
    const uint64_t n_chunks = (n + kChunk - 1) / kChunk;
    cx.rays.reserve(n);
    cx.hits.reserve(n);
    populate_pages(hits, sizeof(RcHit) * n);
    std::vector<hipEvent_t> ev_begin(n_chunks), ev_end(n_chunks);
    for (uint64_t c = 0; c < n_chunks; ++c) { RC_HIP(hipEventCreate(&ev_begin[c])); RC_HIP(hipEventCreate(&ev_end[c])); }
    std::atomic<uint64_t> uploaded{0}, launched{0};
    std::atomic<int> copy_error{0};
    std::atomic<bool> abort_all{false};
    auto span = [&](uint64_t c, uint64_t& off, uint64_t& cnt) { off = c * kChunk; cnt = std::min<uint64_t>(kChunk, n - off); };
    std::thread up([&] {
        RcCaptureRelaxed relaxed;
        hipStream_t st = nullptr;
        hipError_t e = hipSetDevice(s->device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (uint64_t c = 0; c < n_chunks && e == hipSuccess && !abort_all.load(); ++c) {
            uint64_t off, cnt; span(c, off, cnt);
            e = hipMemcpyAsync(cx.rays.p + off, reinterpret_cast<const RcRay*>(rays) + off, sizeof(RcRay) * cnt, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e == hipSuccess) uploaded.store(c + 1, std::memory_order_release);
        }
        if (e != hipSuccess) { copy_error.store((int)e); abort_all.store(true); }
        if (st) (void)hipStreamDestroy(st);
    });
    std::thread down([&] {
        RcCaptureRelaxed relaxed;
        hipStream_t st = nullptr;
        hipError_t e = hipSetDevice(s->device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (uint64_t c = 0; c < n_chunks && e == hipSuccess; ++c) {
            while (launched.load(std::memory_order_acquire) <= c && !abort_all.load()) std::this_thread::yield();
            if (launched.load(std::memory_order_acquire) <= c) break;  // aborted before this chunk was launched
            uint64_t off, cnt; span(c, off, cnt);
            e = hipEventSynchronize(ev_end[c]);
            if (e == hipSuccess) e = hipMemcpyAsync(reinterpret_cast<RcHit*>(hits) + off, cx.hits.p + off, sizeof(RcHit) * cnt, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
        }
        if (e != hipSuccess) { copy_error.store((int)e); abort_all.store(true); }
        if (st) (void)hipStreamDestroy(st);
    });
    std::string launch_error;
    int launch_code = 0;
    try {
        for (uint64_t c = 0; c < n_chunks; ++c) {
            while (uploaded.load(std::memory_order_acquire) <= c && !abort_all.load()) std::this_thread::yield();
            if (abort_all.load()) break;
            uint64_t off, cnt; span(c, off, cnt);
            RC_HIP(hipEventRecord(ev_begin[c], cx.stream));
            rc_launch_trace(s, cx.rays.p + off, cx.hits.p + off, cnt, any, cx.stream, false);
            RC_HIP(hipEventRecord(ev_end[c], cx.stream));
            launched.store(c + 1, std::memory_order_release);
        }
    } catch (const RcError& e) {
        launch_error = e.what(); launch_code = e.code; abort_all.store(true);
    }
    up.join();
    down.join();
    (void)hipStreamSynchronize(cx.stream);
    float total_ms = 0.f;
    uint32_t overflow = 0;
    const uint64_t done = launched.load();
    for (uint64_t c = 0; c < n_chunks; ++c) {
        if (c < done) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev_begin[c], ev_end[c]) == hipSuccess) total_ms += ms;
        }
        (void)hipEventDestroy(ev_begin[c]); (void)hipEventDestroy(ev_end[c]);
    }
    if (s->counters.p) {
        try {
            rc_copy_now(&overflow, rc_status_word(s), 4, hipMemcpyDeviceToHost);
            if (overflow) rc_memset_now(rc_status_word(s), 0, 4);
        } catch (const RcError&) { overflow = 0; (void)hipGetLastError(); }
    }
    rc_timing_fixed(s, total_ms);  // the chunks' kernel time, transfers excluded (as for the single-launch path)
    if (launch_code) throw RcError(launch_code, launch_error);
    if (copy_error.load()) throw RcError(RC_ERR_HIP, std::string("host-buffer transfer failed: ") + hipGetErrorString((hipError_t)copy_error.load()));
    if (overflow) throw RcError(RC_ERR_STACK_OVERFLOW, "traversal stack overflow (tree deeper than 128 levels)");
}

// One device's share of a host-buffer batch (throws; the C-ABI wrappers and rc_multi.hip's per-device threads catch).
extern "C++" void rc_trace_host_impl(rc_scene* s, const rc_ray* rays, rc_hit* hits, uint64_t n, int any) {
    use_device(s);
    require_synced(s);
    if (n == 0) return;
    if (!rays || !hits) throw RcError(RC_ERR_INVALID_ARGUMENT, "rays/hits is NULL");
    CtxLease lease(s);  // this call's own stream and staging buffers: host-buffer trace calls are re-entrant on a synced scene
    CallCtx& cx = *lease.c;
    if (n >= 3 * (1ull << 20) && s->opt.host_pipeline) { trace_host_pipelined(s, cx, rays, hits, n, any); return; }
    cx.rays.reserve(n);
    cx.hits.reserve(n);
    if (n >= (1ull << 16)) populate_pages(hits, sizeof(RcHit) * n);
    RC_HIP(hipMemcpyAsync(cx.rays.p, rays, sizeof(RcRay) * n, hipMemcpyHostToDevice, cx.stream));
    rc_launch_trace(s, cx.rays.p, cx.hits.p, n, any, cx.stream);
    RC_HIP(hipMemcpyAsync(hits, cx.hits.p, sizeof(RcHit) * n, hipMemcpyDeviceToHost, cx.stream));
    check_status(s, cx.stream);
}

static int trace_host(rc_scene* s, const rc_ray* rays, rc_hit* hits, uint64_t n, int any) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] { rc_trace_host_impl(s, rays, hits, n, any); });
}
int rc_trace_closest(rc_scene* s, const rc_ray* rays, rc_hit* hits, uint64_t n) { return trace_host(s, rays, hits, n, 0); }
int rc_trace_any(rc_scene* s, const rc_ray* rays, rc_hit* hits, uint64_t n) { return trace_host(s, rays, hits, n, 1); }

static int trace_device(rc_scene* s, const rc_ray* d_rays, rc_hit* d_hits, uint64_t n, void* stream, int any) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        if (n && (!d_rays || !d_hits)) throw RcError(RC_ERR_INVALID_ARGUMENT, "rays/hits is NULL");
        rc_launch_trace(s, reinterpret_cast<const RcRay*>(d_rays), reinterpret_cast<RcHit*>(d_hits), n, any, (hipStream_t)stream);
    });
}
int rc_trace_closest_device(rc_scene* s, const rc_ray* d_rays, rc_hit* d_hits, uint64_t n, void* stream) { return trace_device(s, d_rays, d_hits, n, stream, 0); }
int rc_trace_any_device(rc_scene* s, const rc_ray* d_rays, rc_hit* d_hits, uint64_t n, void* stream) { return trace_device(s, d_rays, d_hits, n, stream, 1); }

// Several INDEPENDENT device batches in one call.  A launch of about a million rays ends with most of the machine idle --
// its waves wait for their longest rays -- and the next launch's workgroups can fill that tail only if they are not ordered behind it:
// callers who kept four such launches in flight on four streams measured +28-44 %.  This entry point does that for them:
// the batches go round-robin onto the scene's auxiliary streams, forked from `stream` with an event and joined back into it, so to the
// caller the call is ONE asynchronous operation on `stream`.  Every batch is traced exactly as a single rc_trace_*_device call would.
static int trace_device_batches(rc_scene* s, const rc_ray* const* d_rays, rc_hit* const* d_hits, const uint64_t* n, int n_batches, void* stream_, int any) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        if (n_batches < 0) throw RcError(RC_ERR_INVALID_ARGUMENT, "n_batches is negative");
        if (n_batches == 0) return;
        if (!d_rays || !d_hits || !n) throw RcError(RC_ERR_INVALID_ARGUMENT, "the batch arrays are NULL");
        for (int b = 0; b < n_batches; ++b)
            if (n[b] && (!d_rays[b] || !d_hits[b])) throw RcError(RC_ERR_INVALID_ARGUMENT, "rays/hits of batch " + std::to_string(b) + " is NULL");
        hipStream_t stream = (hipStream_t)stream_;
        if (n_batches == 1) {  // nothing to overlap
            rc_launch_trace(s, reinterpret_cast<const RcRay*>(d_rays[0]), reinterpret_cast<RcHit*>(d_hits[0]), n[0], any, stream);
            return;
        }
        std::lock_guard<std::mutex> one(s->batches_mu);
        constexpr int kLanes = 4;
        for (auto& a : s->aux_streams) if (!a) RC_HIP(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
        if (!s->batch_fork) RC_HIP(hipEventCreateWithFlags(&s->batch_fork, hipEventDisableTiming));
        for (auto& e : s->batch_join) if (!e) RC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const int lanes = n_batches < kLanes ? n_batches : kLanes;
        RC_HIP(hipEventRecord(s->batch_fork, stream));
        for (int k = 0; k < lanes; ++k) RC_HIP(hipStreamWaitEvent(s->aux_streams[k], s->batch_fork, 0));
        // Whatever happens below, every forked stream is joined back: a caller that is CAPTURING `stream` must get all of them back before it
        // can end the capture, and an eager caller must never be left ordered behind nothing.
        std::string err;
        int code = 0;
        try {
            for (int b = 0; b < n_batches; ++b)
                rc_launch_trace(s, reinterpret_cast<const RcRay*>(d_rays[b]), reinterpret_cast<RcHit*>(d_hits[b]), n[b], any, s->aux_streams[b % kLanes]);
        } catch (const RcError& e) { err = e.what(); code = e.code; }
        for (int k = 0; k < lanes; ++k) {
            RC_HIP(hipEventRecord(s->batch_join[k], s->aux_streams[k]));
            RC_HIP(hipStreamWaitEvent(stream, s->batch_join[k], 0));
        }
        if (code) throw RcError(code, err);
    });
}
int rc_trace_closest_device_batches(rc_scene* s, const rc_ray* const* d_rays, rc_hit* const* d_hits, const uint64_t* n, int n_batches, void* stream) {
    return trace_device_batches(s, d_rays, d_hits, n, n_batches, stream, 0);
}
int rc_trace_any_device_batches(rc_scene* s, const rc_ray* const* d_rays, rc_hit* const* d_hits, const uint64_t* n, int n_batches, void* stream) {
    return trace_device_batches(s, d_rays, d_hits, n, n_batches, stream, 1);
}

// ---- BVH4 (src/bvh4.jl) ------------------------------------------------------------------------------------------
static_assert(sizeof(rc_bvh4_node) == 120, "BVHNode4 is 120 bytes");
static_assert(sizeof(rc_triangle) == 136, "Triangle{UInt32} is 136 bytes");

static Blas& blas4_of(rc_scene* s, uint32_t blas_id) {
    if (blas_id >= s->blas.size()) throw RcError(RC_ERR_INVALID_ARGUMENT, "blas_id out of range");
    Blas& b = s->blas[blas_id];
    if (b.n_nodes4 == 0) throw RcError(RC_ERR_NOT_SYNCED, "no BLAS4 for this geometry: call rc_blas4_build first");
    return b;
}

int rc_blas4_build(rc_scene* s, uint32_t blas_id, uint32_t* n_nodes) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        if (blas_id >= s->blas.size()) throw RcError(RC_ERR_INVALID_ARGUMENT, "blas_id out of range");
        Blas& b = s->blas[blas_id];
        rc_timing_scene_begin(s, s->stream);
        rc_build_blas4(s, b);
        rc_timing_scene_end(s, s->stream);
        RC_HIP(hipStreamSynchronize(s->stream));
        if (n_nodes) *n_nodes = b.n_nodes4;
    });
}

int rc_export_blas4_nodes(rc_scene* s, uint32_t blas_id, rc_bvh4_node* out, uint32_t capacity, uint32_t* count) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        Blas& b = blas4_of(s, blas_id);
        if (count) *count = b.n_nodes4;
        if (!out) return;
        if (capacity < b.n_nodes4) throw RcError(RC_ERR_INVALID_ARGUMENT, "export buffer too small");
        rc_export_blas4(s, b, out);
    });
}

static int trace4_host(rc_scene* s, uint32_t blas_id, const rc_ray* rays, rc_hit* hits, uint64_t n, int any) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        Blas& b = blas4_of(s, blas_id);
        if (n == 0) return;
        if (!rays || !hits) throw RcError(RC_ERR_INVALID_ARGUMENT, "rays/hits is NULL");
        CtxLease lease(s);
        CallCtx& cx = *lease.c;
        cx.rays.reserve(n);
        cx.hits.reserve(n);
        RC_HIP(hipMemcpyAsync(cx.rays.p, rays, sizeof(RcRay) * n, hipMemcpyHostToDevice, cx.stream));
        rc_launch_trace4(s, b, cx.rays.p, cx.hits.p, n, any, cx.stream);
        RC_HIP(hipMemcpyAsync(hits, cx.hits.p, sizeof(RcHit) * n, hipMemcpyDeviceToHost, cx.stream));
        check_status(s, cx.stream);
    });
}
int rc_trace_closest4(rc_scene* s, uint32_t blas_id, const rc_ray* rays, rc_hit* hits, uint64_t n) { return trace4_host(s, blas_id, rays, hits, n, 0); }
int rc_trace_any4(rc_scene* s, uint32_t blas_id, const rc_ray* rays, rc_hit* hits, uint64_t n) { return trace4_host(s, blas_id, rays, hits, n, 1); }

static int trace4_device(rc_scene* s, uint32_t blas_id, const rc_ray* d_rays, rc_hit* d_hits, uint64_t n, void* stream, int any) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        Blas& b = blas4_of(s, blas_id);
        if (n && (!d_rays || !d_hits)) throw RcError(RC_ERR_INVALID_ARGUMENT, "rays/hits is NULL");
        rc_launch_trace4(s, b, reinterpret_cast<const RcRay*>(d_rays), reinterpret_cast<RcHit*>(d_hits), n, any, (hipStream_t)stream);
    });
}
int rc_trace_closest4_device(rc_scene* s, uint32_t blas_id, const rc_ray* d_rays, rc_hit* d_hits, uint64_t n, void* stream) { return trace4_device(s, blas_id, d_rays, d_hits, n, stream, 0); }
int rc_trace_any4_device(rc_scene* s, uint32_t blas_id, const rc_ray* d_rays, rc_hit* d_hits, uint64_t n, void* stream) { return trace4_device(s, blas_id, d_rays, d_hits, n, stream, 1); }

// ---- collision broad phase (src/collision.jl) --------------------------------------------------------------------
static_assert(sizeof(rc_contact_pair) == sizeof(uint2), "ContactPair is two u32");

int rc_collide_instances_device(rc_scene* s, rc_contact_pair* d_out, uint64_t capacity, uint64_t* count, void* stream) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        const uint64_t total = rc_collide_instances_launch(s, reinterpret_cast<uint2*>(d_out), capacity, (hipStream_t)stream);
        if (count) *count = total;
    });
}

int rc_collide_instances(rc_scene* s, rc_contact_pair* out, uint64_t capacity, uint64_t* count) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        std::lock_guard<std::mutex> one_at_a_time(s->host_call_mu);
        uint64_t total = rc_collide_instances_launch(s, nullptr, 0, s->stream);
        if (count) *count = total;
        if (!out || total == 0) return;
        if (capacity < total) throw RcError(RC_ERR_INVALID_ARGUMENT, "contact buffer too small");
        s->contact_stage.reserve(total);
        total = rc_collide_instances_launch(s, s->contact_stage.p, total, s->stream);
        RC_HIP(hipMemcpyAsync(out, s->contact_stage.p, sizeof(uint2) * total, hipMemcpyDeviceToHost, s->stream));
        check_status(s, s->stream);
    });
}

int rc_collide_instances_any(rc_scene* s, uint32_t handle_a, uint32_t handle_b, int* overlap) {
    if (!s || !overlap) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        const HandleRange ra = live_range(s, handle_a), rb = live_range(s, handle_b);
        const uint32_t n = s->n_static_instances;
        *overlap = 0;
        if (n == 0) return;
        // "Small download for TLAS nodes" (:247): the leaves only
        std::vector<RcNode> leaves(n);
        rc_copy_now(leaves.data(), s->tlas_nodes.p + (n - 1), sizeof(RcNode) * n, hipMemcpyDeviceToHost);
        for (uint32_t ia = ra.first; ia < ra.first + ra.count; ++ia)
            for (uint32_t ib = rb.first; ib < rb.first + rb.count; ++ib) {
                const float* a = leaves[ia].f; const float* b = leaves[ib].f;  // leaf position n-1+i (1-based) = leaves[i-1]
                if ((a[3] >= b[0] && a[4] >= b[1] && a[5] >= b[2]) && (a[0] <= b[3] && a[1] <= b[4] && a[2] <= b[5])) { *overlap = 1; return; }
            }
    });
}

int rc_primary_rays_lookat_device(rc_scene* s, const float camera_pos[3], const float camera_right[3], const float camera_up[3], const float camera_forward[3],
                                  float half_width, float half_height, uint32_t width, uint32_t height, uint32_t samples, uint64_t seed, int jitter,
                                  rc_ray* d_rays, void* stream) {
    if (!s || !camera_pos || !camera_right || !camera_up || !camera_forward || !d_rays) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        rc_launch_primary_rays(s, camera_pos, camera_right, camera_up, camera_forward, half_width, half_height, width, height, samples, seed, jitter,
                               reinterpret_cast<RcRay*>(d_rays), (hipStream_t)stream);
    });
}

int rc_reflection_rays_device(rc_scene* s, const rc_ray* d_rays, const rc_hit* d_hits, uint64_t n, float bias, rc_ray* d_out, void* stream) {
    if (!s || (n && (!d_rays || !d_hits || !d_out))) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        rc_launch_reflection_rays(s, reinterpret_cast<const RcRay*>(d_rays), reinterpret_cast<const RcHit*>(d_hits), n, bias,
                                  reinterpret_cast<RcRay*>(d_out), (hipStream_t)stream);
    });
}

int rc_compact_hits_device(rc_scene* s, const rc_hit* d_hits, uint64_t n, uint32_t* d_indices, uint32_t* d_count, void* stream) {
    if (!s || !d_count || (n && (!d_hits || !d_indices))) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        rc_launch_compact_hits(s, reinterpret_cast<const RcHit*>(d_hits), n, d_indices, d_count, (hipStream_t)stream);
    });
}

// ---- scene files ------------------------------------------------------------------------------------------------------
// The reference has no on-disk format; this one stores what a rebuild would otherwise recompute: per
// geometry the Morton-sorted primitives, the BVH2 nodes and the mesh attributes, plus the instance descriptors and the handle
// table.  Loading uploads the arrays and lets rc_sync rebuild the (cheap) TLAS, so a loaded scene traces bit-identically.
int rc_scene_save(rc_scene* s, const char* path) {
    if (!s || !path) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        sync_host_instances(s);
        use_device(s);
        require_synced(s);  // compaction has run: no deleted handles, no unreferenced geometry
        RC_HIP(hipStreamSynchronize(s->stream));
        FileCloser fc{fopen(path, "wb")};
        if (!fc.f) throw RcError(RC_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path);
        std::vector<unsigned char> tmp;
        const uint32_t hdr[6] = {kSceneVersion, (uint32_t)s->blas.size(), (uint32_t)s->instances.size(), s->next_handle_id,
                                 (uint32_t)s->handle_to_range.size(), 0u};
        put(fc.f, kSceneMagic, 8);
        put(fc.f, hdr, sizeof(hdr));
        for (auto& kv : s->handle_to_range) {
            const uint32_t rec[3] = {kv.first, kv.second.first, kv.second.count};
            put(fc.f, rec, sizeof(rec));
        }
        put(fc.f, s->instances.data(), sizeof(RcInstanceDesc) * s->instances.size());
        for (const Blas& b : s->blas) {
            const uint32_t bh[6] = {b.n_prims, b.n_nodes, b.has_attrs ? 1u : 0u, b.has_uvs ? 1u : 0u, b.n_mesh_verts, b.n_mesh_faces};
            put(fc.f, bh, sizeof(bh));
            put(fc.f, b.root_min, 12);
            put(fc.f, b.root_max, 12);
            put_dev(fc.f, b.prims.p, b.n_prims, tmp);
            put_dev(fc.f, b.nodes.p, b.n_nodes, tmp);
            if (b.has_attrs) {
                put_dev(fc.f, b.m_normals.p, 3 * (size_t)b.n_mesh_verts, tmp);
                if (b.has_uvs) put_dev(fc.f, b.m_uvs.p, 2 * (size_t)b.n_mesh_verts, tmp);
                put_dev(fc.f, b.m_indices.p, 3 * (size_t)b.n_mesh_faces, tmp);
                put_dev(fc.f, b.src_face.p, b.n_prims, tmp);
            }
        }
        if (fflush(fc.f) != 0) throw RcError(RC_ERR_INVALID_ARGUMENT, "scene file: flush failed");
    });
}

int rc_scene_load(int device, const char* path, rc_scene** out) {
    if (!path || !out) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    rc_scene* s = nullptr;
    int rc = rc_scene_create(device, &s);
    if (rc != RC_OK) return rc;
    rc = guarded([&] {
        use_device(s);
        FileCloser fc{fopen(path, "rb")};
        if (!fc.f) throw RcError(RC_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path);
        char magic[8];
        uint32_t hdr[6];
        get(fc.f, magic, 8);
        get(fc.f, hdr, sizeof(hdr));
        if (memcmp(magic, kSceneMagic, 8) != 0 || hdr[0] != kSceneVersion) throw RcError(RC_ERR_INVALID_ARGUMENT, "not a raycore-mi355x scene file (or wrong version)");
        const uint32_t n_blas = hdr[1], n_inst = hdr[2], n_handles = hdr[4];
        s->next_handle_id = hdr[3];
        if ((size_t)n_handles > bytes_left(fc.f) / 12 || (size_t)n_inst > bytes_left(fc.f) / sizeof(RcInstanceDesc) || (size_t)n_blas > bytes_left(fc.f) / 48)
            bad_file("truncated (the header's counts exceed the file)");
        for (uint32_t i = 0; i < n_handles; ++i) {
            uint32_t rec[3];
            get(fc.f, rec, sizeof(rec));
            if ((uint64_t)rec[1] + rec[2] > n_inst) throw RcError(RC_ERR_INVALID_ARGUMENT, "scene file: handle range out of bounds");
            if (rec[0] == 0 || rec[0] >= s->next_handle_id || s->handle_to_range.count(rec[0])) bad_file("handle id out of range or repeated");
            s->handle_to_range[rec[0]] = HandleRange{rec[1], rec[2]};
        }
        s->instances.resize(n_inst);
        get(fc.f, s->instances.data(), sizeof(RcInstanceDesc) * n_inst);
        for (auto& in : s->instances)
            if (in.blas_index < 1 || in.blas_index > n_blas) throw RcError(RC_ERR_INVALID_ARGUMENT, "scene file: instance refers to a missing geometry");
        std::vector<unsigned char> tmp;
        s->blas.resize(n_blas);
        for (Blas& b : s->blas) {
            uint32_t bh[6];
            get(fc.f, bh, sizeof(bh));
            b.n_prims = bh[0]; b.n_nodes = bh[1]; b.has_attrs = bh[2] != 0; b.has_uvs = bh[3] != 0; b.n_mesh_verts = bh[4]; b.n_mesh_faces = bh[5];
            if (b.n_prims == 0 || b.n_nodes != 2 * b.n_prims - 1) throw RcError(RC_ERR_INVALID_ARGUMENT, "scene file: inconsistent geometry header");
            get(fc.f, b.root_min, 12);
            get(fc.f, b.root_max, 12);
            std::vector<RcPrim> host_prims;
            get_dev(fc.f, b.prims, b.n_prims, tmp, [&](const RcPrim* p) { host_prims.assign(p, p + b.n_prims); });
            get_dev(fc.f, b.nodes, b.n_nodes, tmp, [&](const RcNode* nd) {  // internal nodes 1..n-1 point at nodes, leaves n..2n-1 at a sorted primitive
                // The links must form a TREE: a traversal pushes one child and descends into the other, so a node that points at
                // itself or at an ancestor would make a ray loop for ever (the lane stack stops growing at 128 entries, the descent
                // does not stop) -- a hung GPU.  With 2n - 1 nodes, 2(n - 1) child links, every non-root node referenced exactly once
                // and the root never, every node is reached from the root along exactly one finite path.
                std::vector<unsigned char> refs(b.n_nodes, 0);
                for (uint32_t i = 0; i < b.n_nodes; ++i) {
                    const bool leaf = nd[i].child0 == RC_INVALID_NODE;
                    if (leaf != (i + 1 >= b.n_prims)) bad_file("node kinds do not follow the LBVH numbering");
                    if (leaf ? (nd[i].child1 < 1 || nd[i].child1 > b.n_prims)
                             : (nd[i].child0 < 1 || nd[i].child0 > b.n_nodes || nd[i].child1 < 1 || nd[i].child1 > b.n_nodes))
                        bad_file("node child index out of range");
                    if (!leaf)
                        for (uint32_t c : {nd[i].child0, nd[i].child1})
                            if (++refs[c - 1] > 1) bad_file("node links do not form a tree (a node has two parents)");
                }
                if (refs[0] != 0) bad_file("node links do not form a tree (the root is somebody's child)");
                for (uint32_t i = 1; i < b.n_nodes; ++i)
                    if (refs[i] != 1) bad_file("node links do not form a tree (an unreferenced node)");
                // The NODES are what a ray meets, the primitive array is what the entry cull's spheres and the epilogues are derived from
                // (ADVICE r3): a file whose leaves carry other vertices than its primitives, or whose boxes are not the refit's boxes (a
                // leaf's box = min / max of its three vertices, an interior child's = the union of that child's two boxes), would make
                // "no option changes a result" false.  min / max are exact, so a file written by rc_scene_save matches bit for bit
                // (NaN components compare as equal to NaN).
                auto same = [](float a, float c) { return memcmp(&a, &c, 4) == 0 || (a != a && c != c); };
                for (uint32_t i = b.n_prims - 1; i < b.n_nodes; ++i) {  // the leaves
                    const RcPrim& pr = host_prims[nd[i].child1 - 1];
                    for (int k = 0; k < 9; ++k) if (!same(nd[i].f[k], pr.v[k])) bad_file("a leaf node's vertices differ from the primitive it names");
                }
                for (uint32_t i = 0; i + 1 < b.n_prims; ++i) {  // the interior nodes
                    for (int side = 0; side < 2; ++side) {
                        const RcNode& ch = nd[(side ? nd[i].child1 : nd[i].child0) - 1];
                        float3_ lo, hi;
                        if (ch.child0 == RC_INVALID_NODE) {
                            const float3_ v0 = mk3(ch.f[0], ch.f[1], ch.f[2]), v1 = mk3(ch.f[3], ch.f[4], ch.f[5]), v2 = mk3(ch.f[6], ch.f[7], ch.f[8]);
                            lo = min3v(min3v(v0, v1), v2); hi = max3v(max3v(v0, v1), v2);
                        } else {
                            lo = min3v(mk3(ch.f[0], ch.f[1], ch.f[2]), mk3(ch.f[6], ch.f[7], ch.f[8]));
                            hi = max3v(mk3(ch.f[3], ch.f[4], ch.f[5]), mk3(ch.f[9], ch.f[10], ch.f[11]));
                        }
                        const float* box = nd[i].f + 6 * side;
                        if (!same(box[0], lo.x) || !same(box[1], lo.y) || !same(box[2], lo.z) || !same(box[3], hi.x) || !same(box[4], hi.y) || !same(box[5], hi.z))
                            bad_file("a node's child box is not the box of that child (the tree was not refitted from these vertices)");
                    }
                }
            });
            if (b.has_attrs) {
                get_dev(fc.f, b.m_normals, 3 * (size_t)b.n_mesh_verts, tmp);
                if (b.has_uvs) get_dev(fc.f, b.m_uvs, 2 * (size_t)b.n_mesh_verts, tmp);
                get_dev(fc.f, b.m_indices, 3 * (size_t)b.n_mesh_faces, tmp, [&](const uint32_t* ix) {
                    for (size_t i = 0; i < 3 * (size_t)b.n_mesh_faces; ++i) if (ix[i] >= b.n_mesh_verts) bad_file("face index out of range");
                });
                get_dev(fc.f, b.src_face, b.n_prims, tmp, [&](const uint32_t* sf) {
                    for (uint32_t i = 0; i < b.n_prims; ++i) if (sf[i] >= b.n_mesh_faces) bad_file("source face out of range");
                });
            }
        }
        s->dirty = true;  // the TLAS and the flat arrays are rebuilt by the next rc_sync
    });
    if (rc != RC_OK) { rc_scene_destroy(s); return rc; }
    *out = s;
    return RC_OK;
}

// instance_buffer(tlas, handle) + refit_tlas!(tlas) (src/Raycore.jl:117-128, src/instanced-bvh.jl:2197-2222): the handle's
// descriptors as they sit in device memory, to be rewritten by the caller's own kernels and committed without a host round trip.
int rc_instance_buffer_device(rc_scene* s, uint32_t handle, rc_instance_desc** d_descs, uint32_t* count) {
    if (!s || !d_descs) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        const HandleRange r = live_range(s, handle);
        *d_descs = reinterpret_cast<rc_instance_desc*>(s->d_instances.p + r.first);
        if (count) *count = r.count;
    });
}

int rc_refit_device(rc_scene* s, int recompute_inverse) {
    if (!s) return fail(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        rc_timing_scene_begin(s, s->stream);
        rc_refit_tlas(s, true, recompute_inverse != 0);
        rc_timing_scene_end(s, s->stream);
    });
}

int rc_set_option(rc_scene* s, const char* name, int64_t value) {
    if (!s || !name) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    std::string k(name);
    if (k == "kernel") s->opt.kernel = value < -1 ? -1 : (value > 6 ? 6 : value);
    else if (k == "blocks_per_cu") s->opt.blocks_per_cu = value < 0 ? 0 : (value > 8 ? 8 : value);  // 0 = derive from the stack depth; the stack spill area is sized for 8 blocks of 256 threads per CU
    else if (k == "lds_stack") s->opt.lds_stack = value;
    else if (k == "refill") s->opt.refill = value < 1 ? 1 : (value > 64 ? 64 : value);
    else if (k == "stack16") s->opt.stack16 = value != 0;
    else if (k == "stats") s->opt.stats = value;
    else if (k == "pool") s->opt.pool = value <= 0 ? 0 : (value < 16 ? 16 : (value > (1 << 20) ? (1 << 20) : value));  // 0 = default (128 rays per claim)
    else if (k == "onesweep_min") s->opt.onesweep_min = value < 0 ? 0 : value;
    else if (k == "blas_top") s->opt.blas_top = value != 0;
    else if (k == "claim_shards") { int64_t p2 = 1; while (p2 * 2 <= value && p2 * 2 <= kClaimShards) p2 *= 2; s->opt.claim_shards = p2; }  // a power of two
    else if (k == "host_pipeline") s->opt.host_pipeline = value != 0;
    else if (k == "taper") s->opt.taper = value < 0 ? 0 : (value > 64 ? 64 : value);
    else if (k == "entry_cull") s->opt.entry_cull = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (k == "cost_order") s->opt.cost_order = value != 0;
    else if (k == "cost_thr") s->opt.cost_thr = value < 1 ? 1 : (value > 4096 ? 4096 : value);
    else if (k == "sched_thr") s->opt.sched_thr = value < 1 ? 1 : (value > 64 ? 64 : value);
    else if (k == "vf_first_touch") s->opt.vf_first_touch = value ? 1 : 0;
    else if (k == "release_captures") {  // the caller's hipGraphs that captured launches of this scene are destroyed: free their spill regions, hand their counter slots out again
        if (value) {
            (void)hipSetDevice(s->device);
            std::lock_guard<std::mutex> g(s->launch_mu);
            s->capture_slots.clear();
            s->last_capture = -1;
        }
    }
    else if (k == "release_capture") {  // ONE captured launch handed back: value = the token option "last_capture_token" returned right after that capture
        (void)hipSetDevice(s->device);
        std::lock_guard<std::mutex> g(s->launch_mu);
        if (value < 1 || value > (int64_t)s->capture_slots.size() || !s->capture_slots[value - 1].in_use) return fail(RC_ERR_INVALID_ARGUMENT, "release_capture: not the token of a captured launch this scene still holds");
        s->capture_slots[value - 1] = rc_scene::CaptureSlot();
    }
    else if (k == "vf_chunk_bytes") s->opt.vf_chunk_bytes = value < 4096 ? 4096 : (value > (int64_t(4) << 30) ? (int64_t(4) << 30) : value);
    else if (k == "timeline_ptr") s->opt.timeline_ptr = value;  // dev instrumentation: the caller owns the buffer and its size (8 x u64 per wave of the launch)
    else if (k == "debug_set_overflow") {  // test hook: raise the sticky stack-overflow word as a kernel would (no LBVH is deep enough to do it for real)
        const char* hooks = getenv("RC_ENABLE_DEBUG_HOOKS");  // not part of the product's interface: only a process that asks for the hooks gets them
        if (!hooks || hooks[0] != '1') return fail(RC_ERR_INVALID_ARGUMENT, "unknown option debug_set_overflow (test hook: set RC_ENABLE_DEBUG_HOOKS=1)");
        if (!s->counters.p) return fail(RC_ERR_INVALID_ARGUMENT, "debug_set_overflow: no launch has run on this scene yet");
        const uint32_t one = value ? 1u : 0u;
        (void)hipSetDevice(s->device);
        if (hipMemcpy(rc_status_word(s), &one, 4, hipMemcpyHostToDevice) != hipSuccess) return fail(RC_ERR_HIP, "status write failed");
    }
    else return fail(RC_ERR_INVALID_ARGUMENT, "unknown option " + k);
    return RC_OK;
}
int rc_get_option(rc_scene* s, const char* name, int64_t* value) {
    if (!s || !name || !value) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    std::string k(name);
    if (k == "kernel") *value = s->opt.kernel;
    else if (k == "blocks_per_cu") *value = s->opt.blocks_per_cu;
    else if (k == "n_cus") *value = s->n_cus;
    else if (k == "lds_stack") *value = s->opt.lds_stack;
    else if (k == "refill") *value = s->opt.refill;
    else if (k == "pool") *value = s->opt.pool;
    else if (k == "claim_shards") *value = s->opt.claim_shards;
    else if (k == "sched_thr") *value = s->opt.sched_thr;
    else if (k == "onesweep_min") *value = s->opt.onesweep_min;
    else if (k == "stack16") *value = s->opt.stack16;
    else if (k == "stack16_in_use") *value = (s->small_trees && s->opt.stack16) ? 1 : 0;
    else if (k == "blas_top") *value = s->opt.blas_top;
    else if (k == "claim_drift") {
        // dev: chunk counters, over all slots, that are not back at zero once the device is idle -- always 0 unless the
        // every-wave-fails-exactly-once accounting of rc_claim_chunk is broken
        (void)hipSetDevice(s->device);
        (void)hipDeviceSynchronize();
        int64_t drift = 0;
        if (s->counters.p) {
            std::vector<uint32_t> w((size_t)kCounterSlots * kCounterSlotWords);
            if (hipMemcpy(w.data(), s->counters.p, w.size() * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return fail(RC_ERR_HIP, "counter read failed");
            for (int slot = 0; slot < kCounterSlots; ++slot)
                for (int sh = 0; sh < kClaimShards; ++sh)
                    if (w[(size_t)slot * kCounterSlotWords + kShardBase + (size_t)sh * kShardStrideWords] != 0) drift += 1;
        }
        *value = drift;
    }
    else if (k == "host_pipeline") *value = s->opt.host_pipeline;
    else if (k == "taper") *value = s->opt.taper;
    else if (k == "entry_cull") *value = s->opt.entry_cull;
    else if (k == "cost_order") *value = s->opt.cost_order;
    else if (k == "debug_order_ptr" || k == "debug_order_n" || k == "debug_ctl_ptr" || k == "debug_cost_ptr") {  // dev: the claim order of the most recently used launch shape
        const rc_scene::ChunkHistory* h = nullptr;
        for (const auto& e : s->histories) if (!h || e.last_use > h->last_use) h = &e;
        *value = !h ? 0 : (k == "debug_order_n" ? (int64_t)h->n_chunks : (k == "debug_ctl_ptr" ? (int64_t)(uintptr_t)(h->ctl.p + h->parity * rc_scene::ChunkHistory::kHeaderWords) : (k == "debug_cost_ptr" ? (int64_t)(uintptr_t)h->cost.p : (int64_t)(uintptr_t)h->order.p)));
    }
    else if (k == "debug_inst_cull_ptr") *value = (int64_t)(uintptr_t)s->inst_cull.p;  // dev: the entry-cull spheres, 2 x float4 per instance
    else if (k == "cost_thr") *value = s->opt.cost_thr;
    else if (k == "vf_first_touch") *value = s->opt.vf_first_touch;
    else if (k == "release_captures") {  // captured launches currently holding a region and a slot
        std::lock_guard<std::mutex> g(s->launch_mu);
        int64_t held = 0;
        for (const auto& c : s->capture_slots) held += c.in_use ? 1 : 0;
        *value = held;
    }
    else if (k == "last_capture_token") { std::lock_guard<std::mutex> g(s->launch_mu); *value = s->last_capture >= 0 && s->capture_slots[s->last_capture].in_use ? s->last_capture + 1 : 0; }  // of the scene's most recent captured launch (0: none held)
    else if (k == "capture_bytes") {  // device memory held by the scene's captured launches
        std::lock_guard<std::mutex> g(s->launch_mu);
        int64_t b = 0;
        for (const auto& c : s->capture_slots) { b += (int64_t)c.region.cap * 4; for (const auto& x : c.scratch) b += (int64_t)x->cap * 8; }
        *value = b;
    }
    else if (k == "vf_chunk_bytes") *value = s->opt.vf_chunk_bytes;
    else if (k == "blas_top_k") *value = s->blas_top_k;
    else if (k == "tlas_top_k") *value = s->tlas_top_k;
    else if (k.rfind("stat", 0) == 0 && ((k.size() == 5 && ((k[4] >= '0' && k[4] <= '9') || (k[4] >= 'a' && k[4] <= 'f'))) ||
                                           (k.size() == 6 && k[4] >= '1' && k[4] <= '2' && k[5] >= '0' && k[5] <= '9'))) {  // "stat0".."statf", "stat16".."stat23"
        unsigned long long st[kStatsWords] = {0};
        const int idx = k.size() == 6 ? (k[4] - '0') * 10 + (k[5] - '0') : (k[4] <= '9' ? k[4] - '0' : k[4] - 'a' + 10);
        if (idx >= kStatsWords) return fail(RC_ERR_INVALID_ARGUMENT, "unknown option " + k);
        (void)hipSetDevice(s->device);
        (void)hipDeviceSynchronize();
        if (s->counters.p && hipMemcpy(st, rc_stats_words(s), sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) return fail(RC_ERR_HIP, "stats read failed");
        *value = (int64_t)st[idx];
    }
    else return fail(RC_ERR_INVALID_ARGUMENT, "unknown option " + k);
    return RC_OK;
}

int rc_generate_ray_grid_device(rc_scene* s, const float viewdir[3], uint32_t grid, rc_ray* d_rays, void* stream) {
    if (!s || !viewdir || (grid && !d_rays)) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        rc_launch_ray_grid(s, viewdir, grid, reinterpret_cast<RcRay*>(d_rays), (hipStream_t)stream);
    });
}

int rc_get_illumination_device(rc_scene* s, const float viewdir[3], uint32_t grid, uint64_t ray_begin, uint64_t ray_end, float* d_counts, void* stream) {
    if (!s || !viewdir || !d_counts) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        uint64_t n = (uint64_t)grid * grid;
        if (ray_end > n) ray_end = n;
        rc_launch_illumination(s, viewdir, grid, ray_begin, ray_end, d_counts, (hipStream_t)stream);
    });
}

int rc_get_illumination(rc_scene* s, const float viewdir[3], uint32_t grid, float* out_counts) {
    if (!s || !viewdir || !out_counts) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        std::lock_guard<std::mutex> one_at_a_time(s->host_call_mu);
        uint32_t np = s->n_flat_prims;
        s->f32_stage.reserve(np ? np : 1);
        RC_HIP(hipMemsetAsync(s->f32_stage.p, 0, sizeof(float) * (np ? np : 1), s->stream));
        rc_launch_illumination(s, viewdir, grid, 0, (uint64_t)grid * grid, s->f32_stage.p, s->stream);
        if (np) RC_HIP(hipMemcpyAsync(out_counts, s->f32_stage.p, sizeof(float) * np, hipMemcpyDeviceToHost, s->stream));
        check_status(s, s->stream);
    });
}

int rc_view_factors_device(rc_scene* s, uint32_t rays_per_triangle, uint64_t seed, uint32_t src_begin, uint32_t src_end, uint32_t ray_begin,
                           uint32_t ray_end, uint32_t* d_matrix, uint64_t row_stride, uint64_t col_stride, uint32_t row_offset, uint32_t flags, void* stream) {
    if (!s || !d_matrix) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        rc_launch_view_factors(s, rays_per_triangle, seed, src_begin, src_end, ray_begin, ray_end, d_matrix, row_stride, col_stride, row_offset, flags, (hipStream_t)stream);
    });
}

// view_factors(tlas; rays_per_triangle) -> host Matrix{UInt32} (src/kernels.jl:74-78): row chunks traced on the device while the finished
// ones travel to the caller's matrix (rc_multi.hip); no N x N device matrix exists.
int rc_view_factors(rc_scene* s, uint32_t rays_per_triangle, uint64_t seed, uint32_t* out_matrix) {
    return rc_view_factors_multi(&s, 1, rays_per_triangle, seed, out_matrix, RC_VF_MODE_ROWS);
}

int rc_view_factors_rows_host(rc_scene* s, uint32_t rays_per_triangle, uint64_t seed, uint32_t row_begin, uint32_t row_end, uint32_t* out_matrix, uint64_t ld) {
    if (!s || !out_matrix) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        if (ld < s->n_flat_prims) throw RcError(RC_ERR_INVALID_ARGUMENT, "leading dimension smaller than the number of primitives");
        std::lock_guard<std::mutex> one_at_a_time(s->host_call_mu);
        rc_timing_fixed(s, rc_view_factors_rows_to_host(s, rays_per_triangle, seed, row_begin, row_end, out_matrix, ld));
    });
}

// The scene list of a multi-device call: no NULL, no scene twice.
static void check_scene_list(rc_scene* const* scenes, int n_scenes, const char* who) {
    for (int g = 0; g < n_scenes; ++g) {
        if (!scenes[g]) throw RcError(RC_ERR_INVALID_ARGUMENT, "scene is NULL");
        for (int h = 0; h < g; ++h) if (scenes[h] == scenes[g]) throw RcError(RC_ERR_INVALID_ARGUMENT, std::string(who) + ": the same scene twice");
    }
}

int rc_view_factors_multi(rc_scene* const* scenes, int n_scenes, uint32_t rays_per_triangle, uint64_t seed, uint32_t* out_matrix, int mode) {
    if (!scenes || n_scenes < 1 || !out_matrix) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        check_scene_list(scenes, n_scenes, "rc_view_factors_multi");
        // one job at a time per scene (the scenes' auxiliary streams and blocks are the job's); locked in address order, so that two
        // callers naming the same scenes in different orders cannot wait for each other
        std::vector<rc_scene*> by_address(scenes, scenes + n_scenes);
        std::sort(by_address.begin(), by_address.end());
        std::vector<std::unique_lock<std::mutex>> locks;
        for (rc_scene* s : by_address) locks.emplace_back(s->host_call_mu);
        rc_view_factors_multi_impl(scenes, n_scenes, rays_per_triangle, seed, out_matrix, mode);
    });
}

// Per-triangle totals of the view-factor job (column sums "received", row sums "emitted") without the matrix: rc_multi.hip.
int rc_view_factor_totals_device(rc_scene* s, uint32_t rays_per_triangle, uint64_t seed, uint32_t src_begin, uint32_t src_end, uint32_t ray_begin,
                                 uint32_t ray_end, uint64_t* d_received, uint64_t* d_emitted, void* stream) {
    if (!s || (!d_received && !d_emitted)) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        rc_launch_vf_totals(s, rays_per_triangle, seed, src_begin, src_end, ray_begin, ray_end, reinterpret_cast<unsigned long long*>(d_received),
                            reinterpret_cast<unsigned long long*>(d_emitted), (hipStream_t)stream);
    });
}
int rc_view_factor_totals_multi(rc_scene* const* scenes, int n_scenes, uint32_t rays_per_triangle, uint64_t seed, uint64_t* out_received, uint64_t* out_emitted) {
    if (!scenes || n_scenes < 1 || (!out_received && !out_emitted)) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        check_scene_list(scenes, n_scenes, "rc_view_factor_totals_multi");
        std::vector<rc_scene*> by_address(scenes, scenes + n_scenes);  // (lock order: see rc_view_factors_multi)
        std::sort(by_address.begin(), by_address.end());
        std::vector<std::unique_lock<std::mutex>> locks;
        for (rc_scene* s : by_address) locks.emplace_back(s->host_call_mu);
        rc_view_factor_totals_multi_impl(scenes, n_scenes, rays_per_triangle, seed, out_received, out_emitted);
    });
}
int rc_multi_prepare(rc_scene* const* scenes, int n_scenes, float* out_ms) {
    if (!scenes || n_scenes < 1) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        check_scene_list(scenes, n_scenes, "rc_multi_prepare");
        std::vector<rc_scene*> by_address(scenes, scenes + n_scenes);  // (lock order: see rc_view_factors_multi)
        std::sort(by_address.begin(), by_address.end());
        std::vector<std::unique_lock<std::mutex>> locks;
        for (rc_scene* s : by_address) locks.emplace_back(s->host_call_mu);
        float ms[4] = {0, 0, 0, 0};
        rc_multi_prepare_impl(scenes, n_scenes, ms);
        if (out_ms) memcpy(out_ms, ms, sizeof ms);
    });
}
int rc_multi_ranks(rc_scene* const* scenes, int n_scenes, int* out_ranks) {
    if (!scenes || n_scenes < 1 || !out_ranks) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] { *out_ranks = rc_multi_ranks_impl(scenes, n_scenes); });
}
int rc_view_factor_totals(rc_scene* s, uint32_t rays_per_triangle, uint64_t seed, uint64_t* out_received, uint64_t* out_emitted) {
    return rc_view_factor_totals_multi(&s, 1, rays_per_triangle, seed, out_received, out_emitted);
}

static int trace_host_multi(rc_scene* const* scenes, int n_scenes, const rc_ray* rays, rc_hit* hits, uint64_t n, int any) {
    if (!scenes || n_scenes < 1) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        check_scene_list(scenes, n_scenes, "rc_trace_*_multi");
        rc_trace_multi_impl(scenes, n_scenes, rays, hits, n, any);
    });
}
int rc_trace_closest_multi(rc_scene* const* scenes, int n_scenes, const rc_ray* rays, rc_hit* hits, uint64_t n) { return trace_host_multi(scenes, n_scenes, rays, hits, n, 0); }
int rc_trace_any_multi(rc_scene* const* scenes, int n_scenes, const rc_ray* rays, rc_hit* hits, uint64_t n) { return trace_host_multi(scenes, n_scenes, rays, hits, n, 1); }

int rc_get_illumination_multi(rc_scene* const* scenes, int n_scenes, const float viewdir[3], uint32_t grid, float* out_counts) {
    if (!scenes || n_scenes < 1 || !viewdir || !out_counts) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        check_scene_list(scenes, n_scenes, "rc_get_illumination_multi");
        rc_illumination_multi_impl(scenes, n_scenes, viewdir, grid, out_counts);
    });
}

int rc_view_factor_rays_device(rc_scene* s, uint64_t seed, uint32_t src_prim, uint32_t ray_begin, uint32_t n_rays, rc_ray* d_rays, void* stream) {
    if (!s || !d_rays) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        if (src_prim >= s->n_flat_prims) throw RcError(RC_ERR_INVALID_ARGUMENT, "src_prim out of range");
        rc_launch_view_factor_rays(s, seed, src_prim, ray_begin, n_rays, reinterpret_cast<RcRay*>(d_rays), (hipStream_t)stream);
    });
}

int rc_hit_points_device(rc_scene* s, const rc_ray* d_rays, const rc_hit* d_hits, uint64_t n, float* d_points, float* d_normals, void* stream) {
    if (!s || !d_rays || !d_hits || !d_points) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        rc_launch_hit_points(s, reinterpret_cast<const RcRay*>(d_rays), reinterpret_cast<const RcHit*>(d_hits), n, d_points, d_normals, (hipStream_t)stream);
    });
}

int rc_shadow_rays_device(rc_scene* s, const rc_ray* d_rays, const rc_hit* d_hits, uint64_t n, const float light[3], float bias, rc_ray* d_shadow_rays, void* stream) {
    if (!s || !d_rays || !d_hits || !light || !d_shadow_rays) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        require_synced(s);
        rc_launch_shadow_rays(s, reinterpret_cast<const RcRay*>(d_rays), reinterpret_cast<const RcHit*>(d_hits), n, light, bias, reinterpret_cast<RcRay*>(d_shadow_rays), (hipStream_t)stream);
    });
}

// (the HIP runtime accepts a second hipHostRegister of the same range silently; the library keeps its own table so that the two
// calls have defined semantics: one registration per array, unregister only what was registered through this interface)
static std::mutex g_host_reg_mutex;
static std::map<void*, uint64_t> g_host_reg;
int rc_host_register(rc_scene* s, void* p, uint64_t bytes) {
    if (!s || !p || bytes == 0) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        std::lock_guard<std::mutex> lock(g_host_reg_mutex);
        if (g_host_reg.count(p)) throw RcError(RC_ERR_INVALID_ARGUMENT, "rc_host_register: this array is already registered");
        RC_HIP(hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault));
        g_host_reg[p] = bytes;
    });
}
int rc_host_unregister(rc_scene* s, void* p) {
    if (!s || !p) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        std::lock_guard<std::mutex> lock(g_host_reg_mutex);
        if (!g_host_reg.count(p)) throw RcError(RC_ERR_INVALID_ARGUMENT, "rc_host_unregister: this array was not registered with rc_host_register");
        RC_HIP(hipStreamSynchronize(s->stream));  // no transfer of the scene may still be reading or writing the array
        RC_HIP(hipHostUnregister(p));
        g_host_reg.erase(p);
    });
}

int rc_last_kernel_ms(rc_scene* s, float* ms) {
    if (!s || !ms) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        *ms = rc_timing_read(s);
    });
}

int rc_recent_kernel_ms(rc_scene* s, uint32_t max_launches, float* ms, uint32_t* n_out) {
    if (!s || !ms || !n_out) return fail(RC_ERR_INVALID_ARGUMENT, "NULL argument");
    return guarded([&] {
        use_device(s);
        *n_out = 0;
        // the events of a launch stay in its counter slot until the slot's next launch (kEagerSlots launches later).  Only the handles are
        // copied under launch_mu; the waits happen without it, so other threads keep launching on the scene meanwhile -- a slot
        // that such a launch reuses before it is read here reports the newer launch's duration.
        struct Pair { hipEvent_t t0, t1; };
        std::vector<Pair> ev;
        uint64_t n = 0;
        {
            std::lock_guard<std::mutex> g(s->launch_mu);
            const uint64_t have = s->launch_seq < (uint64_t)(kEagerSlots - 1) ? s->launch_seq : (uint64_t)(kEagerSlots - 1);
            n = (uint64_t)max_launches < have ? (uint64_t)max_launches : have;
            for (uint64_t i = 0; i < n; ++i) {
                const uint64_t launch = s->launch_seq - n + 1 + i;  // (launch numbers start at 1)
                const rc_scene::LaunchSlot& slot = s->slots[launch % (uint64_t)kEagerSlots];
                ev.push_back((slot.recorded && slot.t0 && slot.t1) ? Pair{slot.t0, slot.t1} : Pair{nullptr, nullptr});
            }
        }
        for (uint64_t i = 0; i < n; ++i) {
            float t = 0.f;
            if (ev[i].t1) {
                RC_HIP(hipEventSynchronize(ev[i].t1));
                if (hipEventElapsedTime(&t, ev[i].t0, ev[i].t1) != hipSuccess) { (void)hipGetLastError(); t = 0.f; }
            }
            ms[i] = t;
        }
        *n_out = (uint32_t)n;
    });
}

}  // extern "C"
