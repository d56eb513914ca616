// rccl_slice.hpp -- the slice of <rccl/rccl.h> that host/dist.cpp binds at run time (dlopen + dlsym: the render library must load
// on a box without RCCL, so it cannot link librccl or include its header where it calls it).  The prototypes below are written
// by hand; host/rccl_check.cpp includes the real <rccl/rccl.h> beside this file and static_asserts every one of them against
// the declaration it stands for, so a signature that drifts in a later ROCm fails THIS library's build -- not the first
// multi-GPU run.
#pragma once

#include <cstddef>

#include <hip/hip_runtime.h>

#include "../../../include/tyr_c.h"

namespace tyr {

// types by value: ncclUniqueId is 128 opaque bytes, ncclComm_t a pointer, the enums (ncclResult_t, ncclDataType_t, ncclRedOp_t) ints
struct NcclId {
	char internal[TYR_DIST_ID_BYTES];
};
typedef void* nccl_comm;
constexpr int kNcclSuccess = 0;
constexpr int kNcclFloat = 7; // ncclFloat32
constexpr int kNcclSum = 0;

struct Rccl {
	void* handle = nullptr;
	int (*GetUniqueId)(NcclId*) = nullptr;
	int (*CommInitRank)(nccl_comm*, int, NcclId, int) = nullptr;
	int (*CommDestroy)(nccl_comm) = nullptr;
	int (*CommCount)(nccl_comm, int*) = nullptr;
	int (*Send)(const void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
	int (*Recv)(void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
	int (*Reduce)(const void*, void*, size_t, int, int, int, nccl_comm, hipStream_t) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
	bool ok = false;
};

} // namespace tyr
