// rc_rccl_abi.h -- the slice of RCCL's C ABI that rc_multi.hip calls through dlopen'd function pointers, declared by hand so that the
// product library has no build- or link-time dependency on RCCL (a box without librccl.so still loads it; only the RCCL branches fail,
// loudly).  Every value and prototype here is pinned to <rccl/rccl.h> at build-check time by tests/rccl_abi_check.cpp (compiled by
// tests/test_rccl_abi.py with the image's header: static_asserts on the enum values, the enum sizes and each prototype's shape) and the
// test-only stub communicator tests/fake_rccl/fake_rccl.hip defines the same six functions WITH rccl.h's own prototypes.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>

namespace rc_rccl {

struct Comm;                 // ncclComm (opaque)
typedef Comm* comm_t;        // ncclComm_t = struct ncclComm*

// ncclResult_t, ncclDataType_t and ncclRedOp_t are plain C enums (4 bytes, passed as int)
constexpr int kSuccess = 0;  // ncclSuccess
constexpr int kUint32 = 3;   // ncclUint32
constexpr int kUint64 = 5;   // ncclUint64
constexpr int kSum = 0;      // ncclSum

typedef int (*CommInitAllFn)(comm_t* comms, int ndev, const int* devlist);                                   // ncclCommInitAll
typedef int (*CommDestroyFn)(comm_t comm);                                                                   // ncclCommDestroy
typedef int (*ReduceFn)(const void* sendbuff, void* recvbuff, size_t count, int datatype, int op, int root,  // ncclReduce
                        comm_t comm, hipStream_t stream);
typedef int (*GroupStartFn)();                                                                               // ncclGroupStart
typedef int (*GroupEndFn)();                                                                                 // ncclGroupEnd
typedef const char* (*GetErrorStringFn)(int result);                                                         // ncclGetErrorString

}  // namespace rc_rccl
