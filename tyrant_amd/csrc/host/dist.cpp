// dist.cpp -- tyr_dist_*: the one exchange that ends a multi-GPU render, in C++ on RCCL (xGMI).
//
// The reference has no multi-GPU path (main.cpp:94 computes `multi_gpu` and drops it); SURVEY.md section 8e and
// BASELINE.json's north_star define this one: pixel rows dealt y % nranks == rank, a full scene replica per GPU, the
// unmodified wavefront loop per rank, and one combine at the end.  Two forms, same frame on the root:
//
//   TYR_DIST_REDUCE  ncclReduce(sum) of the zero-padded full-frame buffers: every rank ships the whole frame
//                    (33 MB at 1080p, mostly zeros) through a ring that is bound by ONE xGMI link (~153 GB/s).
//   TYR_DIST_GATHER  every rank packs the rows it owns (1/nranks of the frame) and sends that slab to the root,
//                    point to point: xGMI is point-to-point, so the root takes nranks-1 slabs over nranks-1
//                    different links at once (4 MB per link at 1080p on 8 GPUs).  The slab is packed on the render
//                    stream into one of two staging buffers and shipped on the communicator's own stream, so the
//                    next render may reset and refill the blit buffer while the exchange is still in flight.
//
// librccl is opened with dlopen when the first communicator is made, so the render library itself loads and runs on a
// box without RCCL.  WHICH librccl matters: a process may hold a second HIP runtime (PyTorch's wheels bundle their own
// libamdhip64 and librccl), and streams, events and allocations of one runtime mean nothing to the other.  The copy
// opened here is the one that sits next to whichever libamdhip64 this process BOUND this library's HIP calls to (found
// with dladdr on hipGetDeviceCount: the ROCm installation's when the library is alone, PyTorch's bundled one when torch
// was imported first; TYR_VERBOSE=1 prints the path), and RCCL
// is only ever handed buffers and streams this library made itself: what the caller owns (blit_buffer, frame_out --
// possibly another runtime's allocations) is touched by this library's own copy kernels alone.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include "ctx.hpp"
#include "host.hpp"
#include "rccl_slice.hpp"

namespace tyr {
void launch_pack_rows(const float4* frame, float4* slab, uint32_t W, uint32_t localRows, uint32_t rank, uint32_t nranks, hipStream_t stream);
void launch_scatter_rows(const float4* slabs, const float4* own, uint32_t ownRank, float4* frame, uint32_t W, uint32_t localRows, uint32_t nranks, hipStream_t stream);
} // namespace tyr

using namespace tyr;

namespace {

Rccl& rccl() {
	static Rccl R; // the dynamic loader's handle: process-wide by nature (function-local static: thread-safe init)
	static const bool once = [] {
		// the librccl beside the HIP runtime this library uses (same ROCm installation, same libamdhip64); by absolute
		// path, so that a differently built copy already in the process is not picked up by its SONAME
		std::string dir;
		Dl_info info;
		if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
			dir = info.dli_fname;
			const size_t slash = dir.rfind('/');
			dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
		}
		const std::string candidates[] = { dir + "librccl.so.1", dir + "librccl.so", "/opt/rocm/lib/librccl.so.1", "librccl.so.1" };
		for (const std::string& n : candidates)
			if (!n.empty() && (R.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL))) {
				if (std::getenv("TYR_VERBOSE"))
					std::fprintf(stderr, "[tyrant] RCCL: %s\n", n.c_str());
				break;
			}
		if (!R.handle)
			return false;
		auto sym = [&](const char* s) { return dlsym(R.handle, s); };
		R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
		R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
		R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
		R.CommCount = reinterpret_cast<decltype(R.CommCount)>(sym("ncclCommCount"));
		R.Send = reinterpret_cast<decltype(R.Send)>(sym("ncclSend"));
		R.Recv = reinterpret_cast<decltype(R.Recv)>(sym("ncclRecv"));
		R.Reduce = reinterpret_cast<decltype(R.Reduce)>(sym("ncclReduce"));
		R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(sym("ncclGroupStart"));
		R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(sym("ncclGroupEnd"));
		R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
		R.ok = R.GetUniqueId && R.CommInitRank && R.CommDestroy && R.Send && R.Recv && R.Reduce && R.GroupStart && R.GroupEnd;
		return R.ok;
	}();
	(void)once;
	return R;
}

int nccl_status(int rc, const char* what) {
	if (rc == kNcclSuccess)
		return TYR_OK;
	if (std::getenv("TYR_VERBOSE"))
		std::fprintf(stderr, "[tyrant] %s: RCCL error %d (%s)\n", what, rc, rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
	return TYR_ERR_DEVICE;
}

// The first HIP call of a thread into THIS library's runtime must not be a kernel launch (it fails with
// hipErrorNoDevice when another runtime initialised the GPU): make the runtime current on its device first.
int ensure_runtime() {
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || hipSetDevice(dev) != hipSuccess)
		return TYR_ERR_NO_DEVICE;
	(void)hipGetLastError(); // and start from a clean last-error slot
	return TYR_OK;
}

#define HIPCHK(expr)                       \
	do {                                   \
		hipError_t e_ = (expr);            \
		if (e_ != hipSuccess)              \
			return static_cast<int>(e_);   \
	} while (0)

} // namespace

struct tyr_dist {
	tyr_ctx* ctx = nullptr;
	int device = 0; // the ctx's device ordinal, copied: tyr_dist_destroy must not look into a ctx that may be gone already
	nccl_comm comm = nullptr;
	int rank = 0, nranks = 1;
	hipStream_t commStream = nullptr;
	float4* staging[2] = { nullptr, nullptr }; // this rank's packed rows, alternating
	// every exchange buffer is allocated by tyr_dist_create (collective: it fails on every rank or on none), never inside a
	// combine, where a failure on one rank would leave its peers waiting in ncclSend / ncclRecv
	float4* recvSlabs = nullptr;               // nranks slabs (any rank may be asked to be the root)
	float4* fullStage = nullptr, *fullRecv = nullptr; // reduce mode: whole frames RCCL may touch
	hipEvent_t evFullShipped = nullptr;        // comm stream: fullStage has been read by the last reduce
	bool fullShippedValid = false;
	hipEvent_t evPacked = nullptr;             // ctx stream: the slab is packed (or, reduce: the render is complete)
	hipEvent_t evShipped[2] = { nullptr, nullptr }; // comm stream: staging[i] has left (it may be packed again)
	bool shippedValid[2] = { false, false };
	hipEvent_t evDone = nullptr;               // comm stream: the last combine has finished
	int cur = 0;
	size_t slabPixels = 0;
};

extern "C" {

int tyr_dist_owned_rows(uint32_t height, uint32_t rank, uint32_t nranks, uint32_t* first_row, uint32_t* n_rows) {
	if (nranks == 0 || rank >= nranks || height % nranks != 0)
		return TYR_ERR_INVALID;
	if (first_row)
		*first_row = rank;
	if (n_rows)
		*n_rows = height / nranks;
	return TYR_OK;
}

int tyr_dist_row_owner(uint32_t y, uint32_t nranks, uint32_t* rank_out, uint32_t* local_row_out) {
	if (nranks == 0)
		return TYR_ERR_INVALID;
	if (rank_out)
		*rank_out = y % nranks;
	if (local_row_out)
		*local_row_out = y / nranks;
	return TYR_OK;
}

int tyr_dist_pack_rows(const void* frame_device, void* slab_device, uint32_t width, uint32_t height, uint32_t rank, uint32_t nranks, void* stream) {
	if (!frame_device || !slab_device || width == 0 || nranks == 0 || rank >= nranks || height == 0 || height % nranks != 0)
		return TYR_ERR_INVALID;
	if (int rc = ensure_runtime())
		return rc;
	launch_pack_rows(static_cast<const float4*>(frame_device), static_cast<float4*>(slab_device), width, height / nranks, rank, nranks, static_cast<hipStream_t>(stream));
	HIPCHK(hipGetLastError());
	return TYR_OK;
}

int tyr_dist_scatter_rows(const void* slabs_device, void* frame_device, uint32_t width, uint32_t height, uint32_t nranks, void* stream) {
	if (!slabs_device || !frame_device || width == 0 || nranks == 0 || height == 0 || height % nranks != 0)
		return TYR_ERR_INVALID;
	const float4* slabs = static_cast<const float4*>(slabs_device);
	if (int rc = ensure_runtime())
		return rc;
	launch_scatter_rows(slabs, slabs, nranks /* no rank's slab is replaced */, static_cast<float4*>(frame_device), width, height / nranks, nranks, static_cast<hipStream_t>(stream));
	HIPCHK(hipGetLastError());
	return TYR_OK;
}

int tyr_dist_unique_id(void* id_out128) {
	if (!id_out128)
		return TYR_ERR_INVALID;
	Rccl& R = rccl();
	if (!R.ok)
		return TYR_ERR_UNSUPPORTED;
	NcclId id;
	const int rc = nccl_status(R.GetUniqueId(&id), "ncclGetUniqueId");
	if (rc)
		return rc;
	std::memcpy(id_out128, &id, sizeof id);
	return TYR_OK;
}

int tyr_dist_create(tyr_dist** out, tyr_ctx* ctx, const void* id128, int32_t rank, int32_t nranks) {
	if (!out || !ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks)
		return TYR_ERR_INVALID;
	*out = nullptr;
	if (static_cast<uint32_t>(rank) != ctx->cfg.rank || static_cast<uint32_t>(nranks) != ctx->cfg.nranks)
		return TYR_ERR_INVALID; // the communicator's rank IS the pixel shard
	Rccl& R = rccl();
	if (!R.ok)
		return TYR_ERR_UNSUPPORTED;
	HIPCHK(hipSetDevice(ctx->cfg.device));
	tyr_dist* d = new (std::nothrow) tyr_dist();
	if (!d)
		return TYR_ERR_OOM;
	d->ctx = ctx;
	d->device = ctx->cfg.device;
	d->rank = rank;
	d->nranks = nranks;
	d->slabPixels = static_cast<size_t>(ctx->cfg.width) * ctx->localRows;
	auto fail = [&](int code) {
		tyr_dist_destroy(d);
		return code;
	};
	if (hipStreamCreateWithFlags(&d->commStream, hipStreamNonBlocking) != hipSuccess)
		return fail(TYR_ERR_NO_DEVICE);
	if (hipEventCreateWithFlags(&d->evPacked, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&d->evDone, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&d->evShipped[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&d->evShipped[1], hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&d->evFullShipped, hipEventDisableTiming) != hipSuccess)
		return fail(TYR_ERR_NO_DEVICE);
	{
		const size_t framePixels = static_cast<size_t>(ctx->cfg.width) * ctx->cfg.height;
		struct {
			float4** p;
			size_t pixels;
		} bufs[] = { { &d->staging[0], d->slabPixels }, { &d->staging[1], d->slabPixels }, { &d->recvSlabs, d->slabPixels * static_cast<size_t>(nranks) }, { &d->fullStage, framePixels }, { &d->fullRecv, framePixels } };
		for (auto& b : bufs) {
			void* p = nullptr;
			if (hipMalloc(&p, b.pixels * sizeof(float4)) != hipSuccess)
				return fail(TYR_ERR_OOM);
			*b.p = static_cast<float4*>(p);
		}
	}
	NcclId id;
	std::memcpy(&id, id128, sizeof id);
	const int rc = nccl_status(R.CommInitRank(&d->comm, nranks, id, rank), "ncclCommInitRank");
	if (rc)
		return fail(rc);
	*out = d;
	return TYR_OK;
}

int tyr_dist_destroy(tyr_dist* d) {
	if (!d)
		return TYR_OK;
	(void)hipSetDevice(d->device);
	if (d->commStream)
		(void)hipStreamSynchronize(d->commStream);
	if (d->comm)
		(void)rccl().CommDestroy(d->comm);
	for (auto& s : d->staging)
		if (s)
			(void)hipFree(s);
	for (float4* p : { d->recvSlabs, d->fullStage, d->fullRecv })
		if (p)
			(void)hipFree(p);
	for (hipEvent_t e : { d->evPacked, d->evDone, d->evShipped[0], d->evShipped[1], d->evFullShipped })
		if (e)
			(void)hipEventDestroy(e);
	if (d->commStream)
		(void)hipStreamDestroy(d->commStream);
	delete d;
	return TYR_OK;
}

int tyr_dist_combine(tyr_dist* d, int32_t mode, int32_t root, void* frame_out_device) {
	if (!d || root < 0 || root >= d->nranks || (mode != TYR_DIST_GATHER && mode != TYR_DIST_REDUCE))
		return TYR_ERR_INVALID;
	tyr_ctx* c = d->ctx;
	if (!c->blit)
		return TYR_ERR_NO_BUFFER;
	const bool isRoot = (d->rank == root);
	if (isRoot && !frame_out_device)
		return TYR_ERR_INVALID;
	Rccl& R = rccl();
	HIPCHK(hipSetDevice(c->cfg.device));
	(void)hipGetLastError(); // start from a clean last-error slot (shared with every other HIP user of this thread)
	const uint32_t W = c->cfg.width, H = c->cfg.height, rows = c->localRows;
	float4* frameOut = static_cast<float4*>(frame_out_device);

	if (mode == TYR_DIST_REDUCE) {
		// the sum over ranks IS the frame: ranks own disjoint pixels and every other element of their buffers is zero.
		// RCCL works on this library's own copies (see the note on runtimes at the top): blit -> fullStage on the render
		// stream, ncclReduce fullStage -> fullRecv and fullRecv -> frame_out on the communicator's stream.
		const size_t pixels = static_cast<size_t>(W) * H;
		if (d->fullShippedValid) // one staging frame: the previous reduce must have read it
			HIPCHK(hipStreamWaitEvent(c->stream, d->evFullShipped, 0));
		launch_pack_rows(c->blit, d->fullStage, W, H, 0u, 1u, c->stream); // nranks = 1: a plain copy of the frame
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventRecord(d->evPacked, c->stream)); // from here on the blit buffer is the renderer's again
		HIPCHK(hipStreamWaitEvent(d->commStream, d->evPacked, 0));
		const int rc = nccl_status(R.Reduce(d->fullStage, isRoot ? static_cast<void*>(d->fullRecv) : nullptr, pixels * 4, kNcclFloat, kNcclSum, root, d->comm, d->commStream), "ncclReduce");
		if (rc)
			return rc;
		if (isRoot) {
			launch_pack_rows(d->fullRecv, frameOut, W, H, 0u, 1u, d->commStream);
			HIPCHK(hipGetLastError());
		}
		HIPCHK(hipEventRecord(d->evFullShipped, d->commStream));
		d->fullShippedValid = true;
		HIPCHK(hipEventRecord(d->evDone, d->commStream));
		return TYR_OK;
	}

	// ---- gather of owned rows ----
	const int s = d->cur;
	d->cur ^= 1;
	if (d->shippedValid[s]) // staging[s] was shipped two combines ago: pack again only after it has left
		HIPCHK(hipStreamWaitEvent(c->stream, d->evShipped[s], 0));
	launch_pack_rows(c->blit, d->staging[s], W, rows, static_cast<uint32_t>(d->rank), static_cast<uint32_t>(d->nranks), c->stream);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(d->evPacked, c->stream)); // from here on the blit buffer is the renderer's again
	HIPCHK(hipStreamWaitEvent(d->commStream, d->evPacked, 0));
	const size_t slabFloats = d->slabPixels * 4;
	if (d->nranks > 1) {
		int rc = nccl_status(R.GroupStart(), "ncclGroupStart");
		if (rc)
			return rc;
		if (isRoot) {
			for (int r = 0; r < d->nranks && !rc; ++r)
				if (r != root)
					rc = nccl_status(R.Recv(d->recvSlabs + static_cast<size_t>(r) * d->slabPixels, slabFloats, kNcclFloat, r, d->comm, d->commStream), "ncclRecv");
		} else {
			rc = nccl_status(R.Send(d->staging[s], slabFloats, kNcclFloat, root, d->comm, d->commStream), "ncclSend");
		}
		const int rce = nccl_status(R.GroupEnd(), "ncclGroupEnd");
		if (rc || rce)
			return rc ? rc : rce;
	}
	if (isRoot) {
		launch_scatter_rows(d->nranks > 1 ? d->recvSlabs : d->staging[s], d->staging[s], static_cast<uint32_t>(root), frameOut, W, rows, static_cast<uint32_t>(d->nranks), d->commStream);
		HIPCHK(hipGetLastError());
	}
	HIPCHK(hipEventRecord(d->evShipped[s], d->commStream));
	d->shippedValid[s] = true;
	HIPCHK(hipEventRecord(d->evDone, d->commStream));
	return TYR_OK;
}

int tyr_dist_wait(tyr_dist* d) {
	if (!d)
		return TYR_ERR_INVALID;
	HIPCHK(hipSetDevice(d->device));
	HIPCHK(hipStreamSynchronize(d->commStream));
	return TYR_OK;
}

int tyr_dist_info(tyr_dist* d, int32_t* comm_ranks_out, int32_t* rank_out) {
	if (!d)
		return TYR_ERR_INVALID;
	int n = -1;
	if (rccl().CommCount && d->comm)
		if (int rc = nccl_status(rccl().CommCount(d->comm, &n), "ncclCommCount"))
			return rc;
	if (comm_ranks_out)
		*comm_ranks_out = n;
	if (rank_out)
		*rank_out = d->rank;
	return TYR_OK;
}

} // extern "C"
