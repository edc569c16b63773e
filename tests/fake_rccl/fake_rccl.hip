// fake_rccl.hip -- TEST INFRASTRUCTURE, never shipped: a stub communicator that lets a ONE-GPU box execute the product's multi-rank
// RCCL branches (raycore.jl_amd/csrc/rc_multi.hip: multi_rays' chunked ncclReduce + copier thread, rc_view_factor_totals_multi's single
// ncclReduce, rc_multi_prepare's warm-up collective).  The product finds it through RC_RCCL_LIBRARY; the "ranks" are G scenes that share
// device 0 (RC_ENABLE_DEBUG_HOOKS=1 + RC_DEBUG_RANKS_SHARE_DEVICE=1).  Built by tests/test_gpu_fake_rccl.py with hipcc.
//
// It defines exactly the six functions the product resolves, WITH <rccl/rccl.h>'s own prototypes (so a drift between the header and the
// product's hand-declared ABI shows up at compile time here and in tests/rccl_abi_check.cpp), and the stream semantics of the real call:
//   ncclReduce(send, recv, count, type, sum, root, comm, stream) is ENQUEUED on `stream`; the collective starts when every rank's stream
//   has reached its call, the root's recv buffer holds the element-wise sum once the root's stream passes the call, and a non-root
//   rank's stream does not pass it before its send buffer has been read.
// Implementation: one event per rank recorded on its stream; the root's stream waits for all of them, copies / adds the peers' buffers
// with a grid-stride kernel, records a `done` event every peer stream then waits for.  Calls inside ncclGroupStart / ncclGroupEnd are
// queued and matched at the outermost ncclGroupEnd; calls outside a group are matched across host threads (k-th call of each rank of a
// communicator set = collective k).  Arguments that must agree across ranks (count, type, op, root) are CHECKED: a mismatch returns
// ncclInvalidArgument, which is how a wrong per-rank call pattern in the product would surface here instead of as a hang on real RCCL.
//
// Test controls (plain C, looked up with ctypes on the same path):
//   fake_rccl_stats(uint64_t out[6])  = {communicator sets created, ncclReduce calls, collectives launched, elements reduced (sum over
//                                        collectives of count), ncclGroupEnd calls that launched something, largest rank count seen}
//   fake_rccl_fail_after(int64_t k)   : the (k+1)-th ncclReduce call from now returns ncclInternalError (k < 0: never) -- error-path tests
//   fake_rccl_reset_stats()
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdint>
#include <deque>
#include <mutex>
#include <vector>

namespace {

struct World;
struct Op {
    const void* send; void* recv; size_t count; ncclDataType_t type; ncclRedOp_t op; int root; hipStream_t stream;
};
struct World {
    int n = 0;
    std::vector<int> device;
    std::vector<std::deque<Op>> queue;  // per rank, calls not yet matched
    std::mutex mu;
};

}  // namespace

struct ncclComm {  // (the opaque type of rccl.h)
    World* world;
    int rank;
};

namespace {

std::atomic<uint64_t> g_worlds{0}, g_calls{0}, g_collectives{0}, g_elements{0}, g_group_launches{0}, g_max_ranks{0};
std::atomic<int64_t> g_fail_after{-1};
thread_local int t_group_depth = 0;
thread_local std::vector<std::pair<ncclComm*, Op>> t_group_ops;

template <typename T>
__global__ void k_add(T* __restrict__ dst, const T* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclUint32: case ncclInt32: return 4;
        case ncclUint64: case ncclInt64: return 8;
        default: return 0;  // the product reduces u32 matrices and u64 totals only
    }
}

#define FK_HIP(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); return ncclUnhandledCudaError; } } while (0)

// every rank's k-th call is there: enqueue the collective on the ranks' streams
ncclResult_t launch(World* w, const std::vector<Op>& op) {
    const Op& r = op[op[0].root];
    for (int g = 0; g < w->n; ++g)
        if (op[g].count != r.count || op[g].type != r.type || op[g].op != r.op || op[g].root != r.root) return ncclInvalidArgument;
    if (r.root < 0 || r.root >= w->n || r.op != ncclSum || type_size(r.type) == 0) return ncclInvalidArgument;
    for (int g = 0; g < w->n; ++g) if (w->device[g] != w->device[0]) return ncclInvalidUsage;  // one device only: this is a stub
    int prev = 0;
    FK_HIP(hipGetDevice(&prev));
    FK_HIP(hipSetDevice(w->device[0]));
    std::vector<hipEvent_t> arrived(w->n, nullptr);
    hipEvent_t done = nullptr;
    ncclResult_t rc = ncclSuccess;
    auto body = [&]() -> ncclResult_t {
        for (int g = 0; g < w->n; ++g) {
            if (g == r.root) continue;
            FK_HIP(hipEventCreateWithFlags(&arrived[g], hipEventDisableTiming));
            FK_HIP(hipEventRecord(arrived[g], op[g].stream));
            FK_HIP(hipStreamWaitEvent(r.stream, arrived[g], 0));
        }
        if (r.count) {
            if (r.recv != r.send) FK_HIP(hipMemcpyAsync(r.recv, r.send, r.count * type_size(r.type), hipMemcpyDeviceToDevice, r.stream));
            const unsigned blocks = (unsigned)std::min<size_t>(4096, (r.count + 255) / 256);
            for (int g = 0; g < w->n; ++g) {
                if (g == r.root) continue;
                if (type_size(r.type) == 4) hipLaunchKernelGGL(k_add<uint32_t>, dim3(blocks), dim3(256), 0, r.stream, (uint32_t*)r.recv, (const uint32_t*)op[g].send, r.count);
                else hipLaunchKernelGGL(k_add<unsigned long long>, dim3(blocks), dim3(256), 0, r.stream, (unsigned long long*)r.recv, (const unsigned long long*)op[g].send, r.count);
                FK_HIP(hipGetLastError());
            }
        }
        FK_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        FK_HIP(hipEventRecord(done, r.stream));
        for (int g = 0; g < w->n; ++g) if (g != r.root) FK_HIP(hipStreamWaitEvent(op[g].stream, done, 0));
        return ncclSuccess;
    };
    rc = body();
    for (auto e : arrived) if (e) (void)hipEventDestroy(e);  // (released by the runtime once the recorded work has passed)
    if (done) (void)hipEventDestroy(done);
    (void)hipSetDevice(prev);
    if (rc == ncclSuccess) { g_collectives++; g_elements += r.count; }
    return rc;
}

ncclResult_t submit(ncclComm* c, const Op& op, bool* launched) {
    World* w = c->world;
    std::lock_guard<std::mutex> lk(w->mu);
    w->queue[c->rank].push_back(op);
    for (;;) {
        for (int g = 0; g < w->n; ++g) if (w->queue[g].empty()) return ncclSuccess;
        std::vector<Op> set(w->n);
        for (int g = 0; g < w->n; ++g) { set[g] = w->queue[g].front(); w->queue[g].pop_front(); }
        const ncclResult_t rc = launch(w, set);
        if (rc != ncclSuccess) return rc;
        if (launched) *launched = true;
    }
}

}  // namespace

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (!comms || ndev < 1) return ncclInvalidArgument;
    World* w = new World;  // lives as long as the process, like the product's communicator cache
    w->n = ndev;
    w->device.resize(ndev);
    w->queue.resize(ndev);
    for (int g = 0; g < ndev; ++g) {
        w->device[g] = devlist ? devlist[g] : g;
        comms[g] = new ncclComm{w, g};
    }
    g_worlds++;
    uint64_t m = g_max_ranks.load();
    while ((uint64_t)ndev > m && !g_max_ranks.compare_exchange_weak(m, (uint64_t)ndev)) {}
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream) {
    if (!comm) return ncclInvalidArgument;
    g_calls++;
    int64_t f = g_fail_after.load();
    while (f >= 0) {
        if (g_fail_after.compare_exchange_weak(f, f - 1)) { if (f == 0) return ncclInternalError; break; }
    }
    const Op o{sendbuff, recvbuff, count, datatype, op, root, stream};
    if (t_group_depth > 0) { t_group_ops.emplace_back(comm, o); return ncclSuccess; }
    return submit(comm, o, nullptr);
}

ncclResult_t ncclGroupStart() {
    ++t_group_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (t_group_depth <= 0) return ncclInvalidUsage;
    if (--t_group_depth > 0) return ncclSuccess;
    std::vector<std::pair<ncclComm*, Op>> ops;
    ops.swap(t_group_ops);
    bool launched = false;
    for (auto& p : ops) {
        const ncclResult_t rc = submit(p.first, p.second, &launched);
        if (rc != ncclSuccess) return rc;
    }
    if (launched) g_group_launches++;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t result) {
    switch (result) {
        case ncclSuccess: return "no error (fake_rccl)";
        case ncclUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
        case ncclInternalError: return "internal error (fake_rccl: injected)";
        case ncclInvalidArgument: return "invalid argument (fake_rccl: the ranks' calls disagree)";
        case ncclInvalidUsage: return "invalid usage (fake_rccl)";
        default: return "error (fake_rccl)";
    }
}

void fake_rccl_stats(uint64_t out[6]) {
    out[0] = g_worlds; out[1] = g_calls; out[2] = g_collectives; out[3] = g_elements; out[4] = g_group_launches; out[5] = g_max_ranks;
}
void fake_rccl_reset_stats() { g_calls = 0; g_collectives = 0; g_elements = 0; g_group_launches = 0; }
void fake_rccl_fail_after(int64_t k) { g_fail_after = k; }

}  // extern "C"
