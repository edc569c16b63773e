// rccl_abi_check.cpp -- compiled (syntax only) by tests/test_rccl_abi.py against the image's <rccl/rccl.h>: pins every value and
// prototype that raycore.jl_amd/csrc/rc_rccl_abi.h declares by hand (the product reaches RCCL through dlopen'd function pointers and
// never includes rccl.h itself).  A C enum argument / result is passed as a 4-byte int, ncclComm_t is a pointer to an opaque struct:
// `abi_t` maps rccl.h's parameter types to those, and the mapped prototype must be the hand-declared function-pointer type exactly.
#include <rccl/rccl.h>

#include <type_traits>

#include "../raycore.jl_amd/csrc/rc_rccl_abi.h"

static_assert(ncclSuccess == rc_rccl::kSuccess, "ncclSuccess");
static_assert(ncclUint32 == rc_rccl::kUint32, "ncclUint32");
static_assert(ncclUint64 == rc_rccl::kUint64, "ncclUint64");
static_assert(ncclSum == rc_rccl::kSum, "ncclSum");
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int), "enums travel as int");
static_assert(std::is_pointer<ncclComm_t>::value && std::is_class<std::remove_pointer<ncclComm_t>::type>::value, "ncclComm_t is a pointer to an opaque struct");

template <typename T, bool IsEnum = std::is_enum<T>::value> struct abi { typedef T type; };
template <typename T> struct abi<T, true> { static_assert(sizeof(T) == sizeof(int), "enum wider than int"); typedef int type; };
template <> struct abi<ncclComm_t, false> { typedef rc_rccl::comm_t type; };
template <> struct abi<ncclComm_t*, false> { typedef rc_rccl::comm_t* type; };
template <typename F> struct abi_fn;
template <typename R, typename... A> struct abi_fn<R (*)(A...)> { typedef typename abi<R>::type (*type)(typename abi<A>::type...); };

#define SAME_SHAPE(fn, Hand) static_assert(std::is_same<abi_fn<decltype(&fn)>::type, rc_rccl::Hand>::value, #fn " differs from rc_rccl::" #Hand)
SAME_SHAPE(ncclCommInitAll, CommInitAllFn);
SAME_SHAPE(ncclCommDestroy, CommDestroyFn);
SAME_SHAPE(ncclReduce, ReduceFn);
SAME_SHAPE(ncclGroupStart, GroupStartFn);
SAME_SHAPE(ncclGroupEnd, GroupEndFn);
SAME_SHAPE(ncclGetErrorString, GetErrorStringFn);

int main() { return 0; }
