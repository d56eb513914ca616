// rccl_check.cpp -- compile-time proof that host/rccl_slice.hpp's hand-written prototypes ARE <rccl/rccl.h>'s.
//
// host/dist.cpp calls RCCL through function pointers it fills with dlsym; nothing checks a dlsym'ed pointer's type, and the
// N > 1 exchange has never met a second GPU before the driver's multi-GPU run.  Here every real declaration is rewritten with
// the slice's stand-in types (ncclResult_t / ncclDataType_t / ncclRedOp_t -> int, ncclComm_t -> void*, ncclUniqueId -> NcclId)
// and compared, type for type, with the member of tyr::Rccl that holds its address; the stand-ins themselves are checked for
// size, alignment and value.  Nothing in this file runs; it only has to compile.
#if __has_include(<rccl/rccl.h>)
#include <type_traits>

#include <rccl/rccl.h>

#include "rccl_slice.hpp"

namespace {
using namespace tyr;

// the stand-in of one parameter / return type
template <class T> struct Slice { typedef T type; };
template <> struct Slice<ncclResult_t> { typedef int type; };
template <> struct Slice<ncclDataType_t> { typedef int type; };
template <> struct Slice<ncclRedOp_t> { typedef int type; };
template <> struct Slice<ncclComm_t> { typedef nccl_comm type; };
template <> struct Slice<ncclComm_t*> { typedef nccl_comm* type; };
template <> struct Slice<ncclUniqueId> { typedef NcclId type; };
template <> struct Slice<ncclUniqueId*> { typedef NcclId* type; };
// ... and of a whole function type
template <class F> struct SliceFn;
template <class R, class... A> struct SliceFn<R (*)(A...)> { typedef typename Slice<R>::type (*type)(typename Slice<A>::type...); };

#define TYR_RCCL_SAME(member, real) static_assert(std::is_same<SliceFn<decltype(&real)>::type, decltype(Rccl::member)>::value, "host/rccl_slice.hpp: Rccl::" #member " no longer matches " #real " in <rccl/rccl.h>")
TYR_RCCL_SAME(GetUniqueId, ncclGetUniqueId);
TYR_RCCL_SAME(CommInitRank, ncclCommInitRank);
TYR_RCCL_SAME(CommDestroy, ncclCommDestroy);
TYR_RCCL_SAME(CommCount, ncclCommCount);
TYR_RCCL_SAME(Send, ncclSend);
TYR_RCCL_SAME(Recv, ncclRecv);
TYR_RCCL_SAME(Reduce, ncclReduce);
TYR_RCCL_SAME(GroupStart, ncclGroupStart);
TYR_RCCL_SAME(GroupEnd, ncclGroupEnd);
TYR_RCCL_SAME(GetErrorString, ncclGetErrorString);
#undef TYR_RCCL_SAME

// the stand-ins are passed exactly as what they stand for: same size and alignment, trivially copyable, same values
static_assert(sizeof(ncclUniqueId) == sizeof(NcclId) && sizeof(NcclId) == TYR_DIST_ID_BYTES && alignof(ncclUniqueId) == alignof(NcclId), "ncclUniqueId is no longer 128 bytes of char");
static_assert(std::is_trivially_copyable<ncclUniqueId>::value && std::is_standard_layout<ncclUniqueId>::value, "ncclUniqueId is passed BY VALUE to ncclCommInitRank: it must stay a plain aggregate");
static_assert(sizeof(ncclComm_t) == sizeof(nccl_comm) && std::is_pointer<ncclComm_t>::value, "ncclComm_t is no longer a pointer");
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int), "an RCCL enum is no longer int-sized");
static_assert(std::is_same<std::underlying_type<ncclResult_t>::type, int>::value || std::is_same<std::underlying_type<ncclResult_t>::type, unsigned>::value, "ncclResult_t's underlying type");
static_assert(static_cast<int>(ncclSuccess) == kNcclSuccess && static_cast<int>(ncclFloat32) == kNcclFloat && static_cast<int>(ncclFloat) == kNcclFloat && static_cast<int>(ncclSum) == kNcclSum, "an RCCL enumerator changed its value");
static_assert(NCCL_UNIQUE_ID_BYTES == TYR_DIST_ID_BYTES, "include/tyr_c.h: TYR_DIST_ID_BYTES");
} // namespace
#else
#pragma message("rccl/rccl.h not found: host/rccl_slice.hpp's prototypes are NOT checked in this build")
#endif
