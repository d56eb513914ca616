// driver.cpp -- tyr_ctx and the C ABI of include/tyr_c.h: the host driver of the wavefront
// loop (reference: launch_kernels, kernel.cu:664-748, and its caller main.cpp:112-170).
//
// State that the reference keeps in function statics and device globals (kernel.cu:211-224,
// 665-667, 688-691) lives in tyr_ctx; constants arrive in the kernels as one by-value
// argument block (FrameParams) instead of cudaMemcpyToSymbol (kernel.cu:681-684, 707-709).
// There is no CPU fallback: without a HIP device tyr_create fails with TYR_ERR_NO_DEVICE.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include <hip/hip_runtime.h>

#include "ctx.hpp"
#include <thread>

#include "host.hpp"

using namespace tyr;

#define HIPCHK(expr)                       \
	do {                                   \
		hipError_t e_ = (expr);            \
		if (e_ != hipSuccess)              \
			return static_cast<int>(e_);   \
	} while (0)

namespace {

template <class T>
int dev_alloc(T*& p, size_t count) {
	void* v = nullptr;
	hipError_t e = hipMalloc(&v, count * sizeof(T));
	if (e != hipSuccess)
		return e == hipErrorOutOfMemory ? TYR_ERR_OOM : static_cast<int>(e);
	p = static_cast<T*>(v);
	return TYR_OK;
}
template <class T>
void dev_free(T*& p) {
	if (p)
		(void)hipFree(p);
	p = nullptr;
}

int alloc_rayq(RayQ& q, size_t n) {
	int rc;
	if ((rc = dev_alloc(q.o_dx, n)))
		return rc;
	if ((rc = dev_alloc(q.dyz, n)))
		return rc;
	if ((rc = dev_alloc(q.direct_ix, n)))
		return rc;
	if ((rc = dev_alloc(q.flags, n)))
		return rc;
	if ((rc = dev_alloc(q.hit, n)))
		return rc;
	if ((rc = dev_alloc(q.key, n)))
		return rc;
	return TYR_OK;
}
void free_rayq(RayQ& q) {
	dev_free(q.o_dx);
	dev_free(q.dyz);
	dev_free(q.direct_ix);
	dev_free(q.flags);
	dev_free(q.hit);
	dev_free(q.key);
}

void default_spheres(tyr_sphere* s) {
	// kernel.cu:674-680
	const tyr_sphere t[TYR_NUM_SPHERES] = {
		{ 16.5f, { 0, 40, 16.5f }, { 1, 1, 1 }, { 0, 0, 0 }, TYR_DIFF },
		{ 16.5f, { 40, 0, 16.5f }, { 0.5f, 0.5f, 0.06f }, { 0, 0, 0 }, TYR_REFR },
		{ 16.5f, { -40, -50, 36.5f }, { 0.6f, 0.5f, 0.4f }, { 0, 0, 0 }, TYR_PHONG },
		{ 16.5f, { -40, -50, 16.5f }, { 0.6f, 0.5f, 0.4f }, { 0, 0, 0 }, TYR_SPEC },
		{ 1e4f, { 0, 0, -1e4f - 20 }, { 1, 1, 1 }, { 0, 0, 0 }, TYR_DIFF },
		{ 20, { 0, -80, 20 }, { 1.0f, 0.0f, 0.0f }, { 0, 0, 0 }, TYR_DIFF },
		{ 9, { 0, -80, 120.0f }, { 0.0f, 1.0f, 0.0f }, { 3, 3, 3 }, TYR_LIGHT },
	};
	std::memcpy(s, t, sizeof(t));
}

bool finite_n(const float* p, int n) {
	for (int i = 0; i < n; ++i)
		if (!std::isfinite(p[i]))
			return false;
	return true;
}

// Every entry point starts here.  Also empties the thread's last-error slot: it is shared with every other HIP user of
// the thread (a stale code left by another library would otherwise surface at this library's next hipGetLastError check).
int use_device(tyr_ctx* c) {
	const hipError_t e = hipSetDevice(c->cfg.device);
	(void)hipGetLastError();
	return static_cast<int>(e);
}

// refresh the pinned host mirror of the device counters; the stream is idle afterwards
int sync_counters(tyr_ctx* c) {
	HIPCHK(hipMemcpyAsync(c->hK, c->dK, sizeof(DevCounters), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return TYR_OK;
}
int push_counters(tyr_ctx* c) {
	HIPCHK(hipMemcpyAsync(c->dK, c->hK, sizeof(DevCounters), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return TYR_OK;
}


FrameParams make_params(const tyr_ctx* c) {
	FrameParams P{};
	P.W = c->cfg.width;
	P.H = c->cfg.height;
	P.N = c->cfg.queue_size;
	P.rank = c->cfg.rank;
	P.nranks = c->cfg.nranks;
	P.localRows = c->localRows;
	P.localPixels = c->localPixels;
	P.flags = c->cfg.flags;
	P.frame = c->frame;
	std::memcpy(P.camPos, c->cam.position, 12);
	std::memcpy(P.camDir, c->cam.direction, 12);
	std::memcpy(P.camRight, c->camRight, 12);
	std::memcpy(P.camUp, c->camUp, 12);
	P.focalDistance = c->cam.focalDistance;
	P.lensRadius = c->cam.lensRadius;
	std::memcpy(P.spheres, c->spheres, sizeof(P.spheres));
	P.sun = c->sun;
	P.scene = c->scene;
	P.scene.nStaged = std::min(c->scene.nStaged, static_cast<uint32_t>(std::max(c->tuning.stagedNodes, 0)));
	P.work = c->q[c->cur];
	P.next = c->q[c->cur ^ 1];
	P.shadow = c->shadow[c->iter & 1u];
	P.shadowPrev = c->shadow[(c->iter ^ 1u) & 1u];
	P.blit = c->blit;
	P.k = c->dK;
	P.kc = c->dKc + (c->iter & 1u);
	P.kcPrev = c->dKc + ((c->iter ^ 1u) & 1u);
	P.segWork = &c->dK->seg[c->cur][0][0];
	P.segNext = &c->dK->seg[c->cur ^ 1][0][0];
	P.segCap = c->segCap;
	P.classStride = c->segCap * tyr::kSegs;
	P.survFlag = c->survFlag;
	P.scanLive = &c->dK->n_live;
	P.foldSpheres = 0u;
	P.stream = c->dStream;
	P.streamIter = 0u;
	P.fillWork = c->fillRay[c->cur];
	P.fillNext = c->fillRay[c->cur ^ 1];
	P.fillShadow = c->fillSh[c->iter & 1u];
	P.fillShadowPrev = c->fillSh[(c->iter ^ 1u) & 1u];
	P.doneWork = c->doneRay[c->cur];
	P.doneNext = c->doneRay[c->cur ^ 1];
	{
		const int out = static_cast<int>(c->iter & 1u), prev = out ^ 1;
		P.vPrev = tyr::VTable{ c->vWord[prev], c->vPre[prev], c->vBlk[prev] };
		P.vWordOut = c->vWord[out];
		P.vPreOut = c->vPre[out];
		P.vBlkOut = c->vBlk[out];
	}
	P.refillMinIdle = static_cast<uint32_t>(std::min(std::max(c->tuning.refillMinIdle, 1), 64));
	P.minTraversing = static_cast<uint32_t>(std::min(std::max(c->tuning.minTraversing, 1), 64));
	P.ticketChunk = static_cast<uint32_t>(std::min(std::max(c->tuning.ticketChunk, 64), 65536));
	P.raysPerBlock = tyr::kCountRaysPerBlock;
	P.staticShare = static_cast<uint32_t>(std::min(std::max(c->tuning.staticShare, 0), 15));
	P.staticInterleave = c->tuning.staticInterleave ? 1u : 0u;
	P.wideDrain = c->tuning.wideDrain ? 1u : 0u;
	P.lights = c->dLights;
	P.nLights = c->nLights;
	std::memcpy(P.triEmission, c->triEmission, 12);
	P.palette = c->dPalette;
	return P;
}

struct KernelTimer {
	tyr_ctx* c;
	int k;
	bool on;
	KernelTimer(tyr_ctx* c_, int k_) : c(c_), k(k_), on((c_->cfg.flags & TYR_FLAG_PROFILE) != 0 && ((c_->tuning.profileMask >> k_) & 1) != 0) {
		if (on)
			(void)hipEventRecord(c->ev[c->iter & 1u][2 * k], c->stream);
	}
	~KernelTimer() {
		if (on) {
			(void)hipEventRecord(c->ev[c->iter & 1u][2 * k + 1], c->stream);
			c->evUsed[c->iter & 1u][k] = true;
		}
	}
};
// fold the recorded event pairs of one set into the running sums (the stream must have passed them)
void collect_timings_of(tyr_ctx* c, int set) {
	if (!(c->cfg.flags & TYR_FLAG_PROFILE))
		return;
	for (int k = 0; k < TYR_K_COUNT; ++k) {
		if (!c->evUsed[set][k])
			continue;
		float ms = 0.0f;
		hipError_t e = hipEventElapsedTime(&ms, c->ev[set][2 * k], c->ev[set][2 * k + 1]);
		if (e == hipErrorNotReady) { // (TYR_TUNE_KERNEL_SNAPSHOT: the host may be here a moment before the stage's closing event has been processed)
			(void)hipGetLastError();
			if (hipEventSynchronize(c->ev[set][2 * k + 1]) == hipSuccess)
				e = hipEventElapsedTime(&ms, c->ev[set][2 * k], c->ev[set][2 * k + 1]);
		}
		if (e == hipSuccess) {
			c->timings.ms[k] += ms;
			c->timings.launches[k] += 1;
		}
		c->evUsed[set][k] = false;
	}
}
// after the stream went idle
void collect_timings(tyr_ctx* c) {
	collect_timings_of(c, 0);
	collect_timings_of(c, 1);
}

uint32_t planned_new(const tyr_ctx* c) {
	const uint64_t room = c->cfg.queue_size - c->hK->primary_ray_cnt;
	const uint64_t budget = c->hK->budget_remaining;
	return static_cast<uint32_t>(std::min(room, budget));
}

// host prologue of launch_kernels, kernel.cu:671-718
int stage_begin(tyr_ctx* c) {
	if (!c->blit)
		return TYR_ERR_NO_BUFFER;
	c->firstTime = false;
	const f3 dir = ld3(c->cam.direction), up = ld3(c->cam.up);
	// kernel.cu:699-700
	const f3 right = normalize(cross(dir, up)) * 1.5f * (static_cast<float>(c->cfg.width) / static_cast<float>(c->cfg.height));
	const f3 upv = normalize(cross(right, dir)) * 1.5f;
	c->camRight[0] = right.x;
	c->camRight[1] = right.y;
	c->camRight[2] = right.z;
	c->camUp[0] = upv.x;
	c->camUp[1] = upv.y;
	c->camUp[2] = upv.z;
	// kernel.cu:702
	bool reset = false;
	for (int k = 0; k < 3; ++k)
		reset = reset || c->lastPos[k] != c->cam.position[k] || c->lastDir[k] != c->cam.direction[k];
	reset = reset || c->lastFocal != c->cam.focalDistance || c->cam.lensRadius != c->lastLens;
	if (c->sunChanged) { // kernel.cu:704-710
		c->sunChanged = false;
		reset = true;
		sun_setup(c->sunPos[0], c->sunPos[1], c->sun);
	}
	if (reset) { // kernel.cu:712-718
		HIPCHK(hipMemsetAsync(c->blit, 0, sizeof(float4) * static_cast<size_t>(c->cfg.width) * c->cfg.height, c->stream));
		c->hK->primary_ray_cnt = 0;
		HIPCHK(hipMemcpyAsync(&c->dK->primary_ray_cnt, &c->hK->primary_ray_cnt, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
		// ... and with it the survivors in the work queue (the next top-up regenerates all N slots, quirk 16)
		std::memset(&c->hK->seg[c->cur][0][0], 0, sizeof c->hK->seg[0]);
		std::memset(c->hK->segSurv, 0, sizeof c->hK->segSurv);
		HIPCHK(hipMemsetAsync(&c->dK->seg[c->cur][0][0], 0, sizeof c->dK->seg[0], c->stream));
		HIPCHK(hipMemsetAsync(c->dK->segSurv, 0, sizeof c->dK->segSurv, c->stream));
	}
	return TYR_OK;
}

// enqueue one stage; the host mirror hK must be current for the sizes used here (upper bounds will do: every kernel takes
// its counts from the device)
void enqueue_primary(tyr_ctx* c, const FrameParams& P, uint32_t nNew) {
	{
		KernelTimer t(c, TYR_K_PRIMARY);
		launch_primary(P, nNew, c->stream);
	}
}
// nSurvivors: how many of the nLive rays were in the queue before this iteration's primary rays (they still need their
// sphere pre-pass)
void enqueue_extend(tyr_ctx* c, const FrameParams& P0, uint32_t nLive, uint32_t nSurvivors) {
	FrameParams P = P0;
	P.prevFolded = c->lastShadeFolded ? 1u : 0u;
	KernelTimer t(c, TYR_K_EXTEND);
	if (c->cfg.flags & TYR_FLAG_DEBUG_BVH) { // the reference's BVH_DEBUG build: kernel.cu:721-722
		launch_extend_debug(P, c->segCap * tyr::kSegs * tyr::kClasses, c->stream);
		return;
	}
	launch_extend(P, nLive, nSurvivors, (c->cfg.flags & TYR_FLAG_COUNT_VISITS) != 0, c->tuning, c->numCUs, c->launchCache, c->stream);
}
// extend of this iteration and connect of the previous one in one launch (tyr_render, TYR_TUNE_MERGE_TRACE)
void enqueue_trace(tyr_ctx* c, const FrameParams& P0, uint32_t nLive, uint32_t nSurvivors, uint32_t maxShadowPrev) {
	FrameParams P = P0;
	P.traceShadow = maxShadowPrev != 0 ? 1u : 0u;
	P.prevFolded = c->lastShadeFolded ? 1u : 0u;
	if (c->scanCarried) { // the iteration before left its slot scan to this launch (TYR_TUNE_SCAN_IN_TRACE)
		P.scanPrevInTrace = 1u;
		P.scanLivePrev = &c->dK->scan_live[c->scanCarriedSet];
		c->scanCarried = false;
	}
	KernelTimer t(c, TYR_K_EXTEND);
	launch_trace(P, nLive, nSurvivors, maxShadowPrev, c->tuning, c->numCUs, c->launchCache, c->stream);
}
// shade, then the scan that turns its survive bytes into the next iteration's slots
void enqueue_shade(tyr_ctx* c, const FrameParams& P, uint32_t nLive) {
	c->shadowSet = c->iter & 1u;
	c->lastShadeFolded = P.foldSpheres != 0u;
	KernelTimer t(c, TYR_K_SHADE);
	launch_shade(P, nLive, c->numCUs, c->launchCache, c->stream);
	if (P.shadeOpensNext != 0u) {
		// shade's last block has opened the next iteration, and the scan's tables are read by the NEXT shade launch only: no k_scan_words
		// launch -- the next traversal launch's waves do the scan on their way in (hip/scan_wave.hpp; enqueue_trace hands it over)
		c->scanCarried = true;
		c->scanCarriedSet = P.scanSet & 1u;
	} else {
		launch_scan(P, nLive, c->stream);
	}
}
void enqueue_connect(tyr_ctx* c, const FrameParams& P0, uint32_t maxShadow) {
	FrameParams P = P0;
	P.prevFolded = c->lastShadeFolded ? 1u : 0u;
	KernelTimer t(c, TYR_K_CONNECT);
	launch_connect(P, maxShadow, (c->cfg.flags & TYR_FLAG_COUNT_VISITS) != 0, c->tuning, c->numCUs, c->launchCache, c->stream);
}

bool merged_render(const tyr_ctx* c) { return c->tuning.mergeTrace != 0 && !(c->cfg.flags & (TYR_FLAG_COUNT_VISITS | TYR_FLAG_DEBUG_BVH)) && c->scene.rootRef != tyr::kRefDone; }

// the shadow rays a merged render still owes (those of the last shaded iteration): a launch of their own
int flush_pending_shadow(tyr_ctx* c) {
	if (!c->shadowPending)
		return TYR_OK;
	c->shadowPending = false;
	FrameParams P = make_params(c);
	P.kc = P.kcPrev; // stage_end has advanced `iter`: the rays belong to the previous iteration's counter set and shadow queue
	P.shadow = P.shadowPrev;
	enqueue_connect(c, P, c->shadowPendingMax);
	HIPCHK(hipGetLastError());
	return TYR_OK;
}

void stage_end(tyr_ctx* c) {
	// kernel.cu:735-745
	if (c->frame == 0xFFFFFFFFu)
		c->frame = 0;
	c->frame++;
	std::memcpy(c->lastPos, c->cam.position, 12);
	std::memcpy(c->lastDir, c->cam.direction, 12);
	c->lastFocal = c->cam.focalDistance;
	c->lastLens = c->cam.lensRadius;
	c->cur ^= 1; // main.cpp:169
	c->iter++;
}

int check_device_error(const tyr_ctx* c) {
	if (!c->hK->device_error)
		return TYR_OK;
	if (std::getenv("TYR_VERBOSE"))
		std::fprintf(stderr, "[tyrant] device_error bits 0x%x (1 = traversal stack overflow, 4 = a traversal wave made no progress, 8 = a queue segment overflowed); n_live %u\n", c->hK->device_error, c->hK->n_live);
	return TYR_ERR_DEVICE;
}

} // namespace

namespace {
// physical slots that hold a record, per segment counter array `seg` (device pointer)
int valid_slots(const uint32_t* dSeg, std::vector<uint32_t>& slots, uint32_t* total = nullptr) {
	uint32_t cnt[tyr::kSegs * tyr::kSegStride];
	HIPCHK(hipMemcpy(cnt, dSeg, sizeof cnt, hipMemcpyDeviceToHost));
	slots.clear();
	uint32_t n = 0;
	for (uint32_t w = 0; w < tyr::kSegs; ++w) {
		const uint32_t c = cnt[w * tyr::kSegStride];
		n += c;
		for (uint32_t j = 0; j < c; ++j)
			slots.push_back(((((j >> 6) * tyr::kSegs) + w) << 6) | (j & 63u));
	}
	if (total)
		*total = n;
	return TYR_OK;
}
// dense layout of n records: record i in physical slot i (chunk i / 64 belongs to segment (i / 64) % 8)
void dense_counts(uint32_t n, uint32_t* cnt /* [kSegs * kSegStride] */) {
	std::memset(cnt, 0, sizeof(uint32_t) * tyr::kSegs * tyr::kSegStride);
	for (uint32_t chunk = 0; chunk * 64u < n; ++chunk)
		cnt[(chunk % tyr::kSegs) * tyr::kSegStride] += std::min<uint32_t>(64u, n - chunk * 64u);
}
template <class T>
int gather(const T* dev, const std::vector<uint32_t>& slots, uint32_t extent, std::vector<T>& out) {
	std::vector<T> all(extent);
	if (extent)
		HIPCHK(hipMemcpy(all.data(), dev, extent * sizeof(T), hipMemcpyDeviceToHost));
	out.resize(slots.size());
	for (size_t i = 0; i < slots.size(); ++i)
		out[i] = all[slots[i]];
	return TYR_OK;
}
} // namespace

extern "C" {

const char* tyr_status_string(int status) {
	switch (status) {
	case TYR_OK: return "ok";
	case TYR_ERR_INVALID: return "invalid argument";
	case TYR_ERR_NO_DEVICE: return "no HIP device (the product path has no CPU fallback)";
	case TYR_ERR_NO_SCENE: return "no scene uploaded";
	case TYR_ERR_NO_BUFFER: return "no blit_buffer bound";
	case TYR_ERR_OOM: return "out of device memory";
	case TYR_ERR_DEVICE: return "device-side error (a render: traversal stack overflow, a queue segment out of room or a stuck traversal wave; tyr_dist_*: an RCCL call failed -- TYR_VERBOSE=1 prints which)";
	case TYR_ERR_UNSUPPORTED: return "unsupported";
	case TYR_ERR_IO: return "file I/O error";
	default: return status > 0 ? hipGetErrorString(static_cast<hipError_t>(status)) : "unknown status";
	}
}

int tyr_abi_version(void) { return TYR_ABI_VERSION; }

int tyr_create(tyr_ctx** out, const tyr_config* cfg) {
	if (!out || !cfg)
		return TYR_ERR_INVALID;
	*out = nullptr;
	if (cfg->width == 0 || cfg->height == 0 || cfg->queue_size == 0 || cfg->nranks == 0 || cfg->rank >= cfg->nranks || (cfg->height % cfg->nranks) != 0)
		return TYR_ERR_INVALID;
	if (static_cast<uint64_t>(cfg->width) * cfg->height >= (1ull << 31) || cfg->queue_size >= (1u << 30))
		return TYR_ERR_INVALID;
	if ((cfg->flags & (TYR_FLAG_LIGHT_LIST | TYR_FLAG_TRIANGLE_COLORS)) && !(cfg->flags & TYR_FLAG_TRIANGLE_MATERIALS))
		return TYR_ERR_INVALID; // an emissive or coloured triangle is a triangle material
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
		return TYR_ERR_NO_DEVICE;
	tyr_ctx* c = new (std::nothrow) tyr_ctx();
	if (!c)
		return TYR_ERR_OOM;
	c->cfg = *cfg;
	c->localRows = cfg->height / cfg->nranks;
	c->localPixels = cfg->width * c->localRows;
	default_spheres(c->spheres);
	const tyr_camera cam = { { 1, 30, 90 }, { 1, 0, 0 }, { 0, 0, 1 }, 1.0f, 0.0f }; // camera.h:4-9
	c->cam = cam;
	int rc = use_device(c);
	if (rc) {
		delete c;
		return rc;
	}
	auto fail = [&](int code) {
		tyr_destroy(c);
		return code;
	};
	{
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0)
			c->numCUs = prop.multiProcessorCount; // the reference's sm_cores (main.cpp:102)
	}
	if (cfg->stream) {
		c->stream = static_cast<hipStream_t>(cfg->stream);
	} else {
		int least = 0, greatest = 0;
		(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
		if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest) != hipSuccess)
			return fail(TYR_ERR_NO_DEVICE);
		c->ownStream = true;
	}
	const size_t N = cfg->queue_size;
	// A queue is eight segments (hip/kernels.hpp "Queues"), sized so that none can run out whatever the rays do.  What one
	// segment of one class can receive in an iteration: from shade, the survivors of the tile pairs dealt to it, which
	// between them hold an eighth of every segment of both classes, N/8 + 1024 records at most; from the top-up an eighth
	// of the new primaries, (N - survivors)/8 + 256.  The sum is largest when a segment's survivors are all there are:
	// N/8 + 1024 + 7/8 (N/8 + 1024)... < 15 N/64 + 4096.  (kErrQueueOverflow stays as a check: a segment never writes past its end.)
	c->segCap = static_cast<uint32_t>(((N / 8 + 7 * (N / 64) + 4096) + 63) & ~size_t(63));
	const size_t cap = static_cast<size_t>(c->segCap) * tyr::kSegs;
	// slots are uint32 everywhere on the device: class 1's last slot is 2 * cap - 1.  At the admitted maximum
	// (queue_size < 2^30) 2 * cap = 16 * segCap < 3.75 * 2^30 + 2^20 < 2^32; checked rather than assumed.
	if (static_cast<uint64_t>(cap) * tyr::kClasses >= (1ull << 32))
		return fail(TYR_ERR_INVALID);
	if ((rc = alloc_rayq(c->q[0], cap * tyr::kClasses)) || (rc = alloc_rayq(c->q[1], cap * tyr::kClasses))) // class 0 (may enter the tree), class 1 (cannot)
		return fail(rc);
	for (auto& sq : c->shadow) // two: shade(i) fills one while the traversal launch of iteration i still reads shade(i - 1)'s
		if ((rc = dev_alloc(sq.o_dx, cap)) || (rc = dev_alloc(sq.dyz_cd_ix, cap)) || (rc = dev_alloc(sq.color, cap)) || (rc = dev_alloc(sq.key, cap)))
			return fail(rc);
	// one survive byte per virtual slot, and the two sets of scan tables made from them (iteration i writes set i & 1)
	const size_t entries = (N + 63) / 64 + kBlock, blocks = (N + 16383) / 16384 + 1;
	if ((rc = dev_alloc(c->survFlag, N + 64)))
		return fail(rc);
	for (int t = 0; t < 2; ++t) {
		if ((rc = dev_alloc(c->vWord[t], entries)) || (rc = dev_alloc(c->vPre[t], entries)) || (rc = dev_alloc(c->vBlk[t], blocks)))
			return fail(rc);
		if (hipMemset(c->vWord[t], 0, entries * 8) != hipSuccess || hipMemset(c->vPre[t], 0, entries * 4) != hipSuccess || hipMemset(c->vBlk[t], 0, blocks * 4) != hipSuccess)
			return fail(TYR_ERR_NO_DEVICE);
	}
	if ((rc = dev_alloc(c->dK, 1)) || (rc = dev_alloc(c->dKc, 2)))
		return fail(rc);
	{
		// the streamed tail's hand-off counters (self-resetting: whoever consumes a chunk / tile zeroes its counter)
		const size_t chunks = cap / 64 + 8, tiles = cap / 256 + 8;
		if ((rc = dev_alloc(c->dStream, 1)))
			return fail(rc);
		for (int t = 0; t < 2; ++t) {
			if ((rc = dev_alloc(c->fillRay[t], chunks)) || (rc = dev_alloc(c->fillSh[t], chunks)) || (rc = dev_alloc(c->doneRay[t], tiles)))
				return fail(rc);
			if (hipMemset(c->fillRay[t], 0, chunks * 4) != hipSuccess || hipMemset(c->fillSh[t], 0, chunks * 4) != hipSuccess || hipMemset(c->doneRay[t], 0, tiles * 4) != hipSuccess)
				return fail(TYR_ERR_NO_DEVICE);
		}
		if (hipHostMalloc(reinterpret_cast<void**>(&c->hStream), sizeof(StreamState), hipHostMallocDefault) != hipSuccess)
			return fail(TYR_ERR_OOM);
		if (hipEventCreateWithFlags(&c->evTail, hipEventDisableTiming) != hipSuccess)
			return fail(TYR_ERR_NO_DEVICE);
	}
	if (hipMemset(c->dKc, 0, 2 * sizeof(ConnectCounters)) != hipSuccess)
		return fail(TYR_ERR_NO_DEVICE);
	{
		// the side stream carries the streamed tail's shade and scan launches, which run BESIDE the traversal kernel (the
		// traversal grid leaves room for them: four of its blocks per CU instead of five)
		int least = 0, greatest = 0;
		(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
		if (hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, greatest) != hipSuccess)
			return fail(TYR_ERR_NO_DEVICE);
	}
	if (cfg->flags & TYR_FLAG_TRIANGLE_COLORS) {
		// the palette's defaults are the reference's constants: white triangles (kernel.cu:383), emission (3,3,3) (kernel.cu:680)
		if ((rc = dev_alloc(c->dPalette, 512)))
			return fail(rc);
		std::vector<float4> pal(512);
		for (int i = 0; i < 256; ++i) {
			pal[2 * i] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
			pal[2 * i + 1] = make_float4(3.0f, 3.0f, 3.0f, 0.0f);
		}
		if (hipMemcpy(c->dPalette, pal.data(), pal.size() * sizeof(float4), hipMemcpyHostToDevice) != hipSuccess)
			return fail(TYR_ERR_NO_DEVICE);
	}
	if (hipEventCreateWithFlags(&c->evSnapshot, hipEventDisableTiming) != hipSuccess)
		return fail(TYR_ERR_NO_DEVICE);
	if (hipHostMalloc(reinterpret_cast<void**>(&c->hK), sizeof(DevCounters), hipHostMallocDefault) != hipSuccess)
		return fail(TYR_ERR_OOM);
	std::memset(c->hK, 0, sizeof(DevCounters));
	c->hK->budget_remaining = ~0ull;
	if (hipMemcpy(c->dK, c->hK, sizeof(DevCounters), hipMemcpyHostToDevice) != hipSuccess)
		return fail(TYR_ERR_NO_DEVICE);
	for (auto& set : c->ev)
		for (auto& e : set)
			if (hipEventCreate(&e) != hipSuccess)
				return fail(TYR_ERR_NO_DEVICE);
	for (int s = 0; s < 2; ++s) {
		if (hipHostMalloc(reinterpret_cast<void**>(&c->hSnap[s]), sizeof(DevCounters), hipHostMallocDefault) != hipSuccess)
			return fail(TYR_ERR_OOM);
		std::memset(c->hSnap[s], 0, sizeof(DevCounters));
		if (hipEventCreateWithFlags(&c->evSnap[s], hipEventDisableTiming) != hipSuccess)
			return fail(TYR_ERR_NO_DEVICE);
		if (hipHostMalloc(reinterpret_cast<void**>(&c->hostSnap[s]), sizeof(tyr::HostSnap), hipHostMallocMapped) != hipSuccess)
			return fail(TYR_ERR_OOM);
		std::memset(c->hostSnap[s], 0, sizeof(tyr::HostSnap));
		if (hipHostGetDevicePointer(reinterpret_cast<void**>(&c->hostSnapDev[s]), c->hostSnap[s], 0) != hipSuccess)
			return fail(TYR_ERR_NO_DEVICE);
	}
	c->scene.rootRef = kRefDone;
	*out = c;
	return TYR_OK;
}

int tyr_destroy(tyr_ctx* c) {
	if (!c)
		return TYR_OK;
	(void)hipSetDevice(c->cfg.device);
	if (c->side)
		(void)hipStreamSynchronize(c->side);
	if (c->stream)
		(void)hipStreamSynchronize(c->stream);
	free_rayq(c->q[0]);
	free_rayq(c->q[1]);
	for (auto& sq : c->shadow) {
		dev_free(sq.o_dx);
		dev_free(sq.dyz_cd_ix);
		dev_free(sq.color);
		dev_free(sq.key);
	}
	dev_free(c->survFlag);
	for (int t = 0; t < 2; ++t) {
		dev_free(c->vWord[t]);
		dev_free(c->vPre[t]);
		dev_free(c->vBlk[t]);
	}
	dev_free(c->dK);
	dev_free(c->dKc);
	dev_free(c->dStream);
	for (int t = 0; t < 2; ++t) {
		dev_free(c->fillRay[t]);
		dev_free(c->fillSh[t]);
		dev_free(c->doneRay[t]);
	}
	if (c->hStream)
		(void)hipHostFree(c->hStream);
	if (c->evTail)
		(void)hipEventDestroy(c->evTail);
	dev_free(c->dNodes);
	dev_free(c->dQuads);
	dev_free(c->dTris);
	dev_free(c->dLights);
	dev_free(c->dPalette);
	if (c->ownBlit)
		dev_free(c->blit);
	if (c->hK)
		(void)hipHostFree(c->hK);
	for (auto& set : c->ev)
		for (auto& e : set)
			if (e)
				(void)hipEventDestroy(e);
	for (int s = 0; s < 2; ++s) {
		if (c->hSnap[s])
			(void)hipHostFree(c->hSnap[s]);
		if (c->evSnap[s])
			(void)hipEventDestroy(c->evSnap[s]);
		if (c->hostSnap[s])
			(void)hipHostFree(c->hostSnap[s]);
	}
	if (c->evSnapshot)
		(void)hipEventDestroy(c->evSnapshot);

	if (c->side)
		(void)hipStreamDestroy(c->side);
	if (c->ownStream && c->stream)
		(void)hipStreamDestroy(c->stream);
	delete c;
	return TYR_OK;
}

namespace {

// the light array the reference leaves as a TODO (kernel.cu:420): LIGHT triangles in (reordered) array order
int upload_light_list(tyr_ctx* c, const tyr_triangle* prims, int32_t nPrims) {
	if (!(c->cfg.flags & TYR_FLAG_LIGHT_LIST))
		return TYR_OK;
	std::vector<uint32_t> lights;
	for (int32_t i = 0; i < nPrims; ++i)
		if (prims[i].materialType == TYR_LIGHT)
			lights.push_back(static_cast<uint32_t>(i));
	if (lights.empty())
		return TYR_OK;
	int rc;
	if ((rc = dev_alloc(c->dLights, lights.size())))
		return rc;
	HIPCHK(hipMemcpy(c->dLights, lights.data(), lights.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	c->nLights = static_cast<uint32_t>(lights.size());
	return TYR_OK;
}

// the ctx lets go of its scene (the stream is idle afterwards)
int drop_scene(tyr_ctx* c) {
	HIPCHK(hipStreamSynchronize(c->stream));
	dev_free(c->dNodes);
	dev_free(c->dQuads);
	dev_free(c->dTris);
	dev_free(c->dLights);
	c->nLights = 0;
	c->scene = DevScene{};
	c->scene.rootRef = kRefDone;
	c->haveScene = true;
	return TYR_OK;
}

// a layout made on the device (hip/bvh_layout_dev.hip) becomes the ctx's scene: L's arrays change owner
int adopt_device_layout(tyr_ctx* c, DeviceTreeLayout& L, int32_t nPrims) {
	int rc = drop_scene(c);
	if (rc == TYR_OK)
		rc = dev_alloc(c->dNodes, 4); // (no pair nodes; one element so the pointer is never null)
	if (rc) {
		(void)hipFree(L.quads);
		(void)hipFree(L.tris);
		L.quads = L.tris = nullptr;
		return rc;
	}
	c->dQuads = L.quads;
	c->dTris = L.tris;
	L.quads = L.tris = nullptr;
	c->scene.quads = c->dQuads;
	c->scene.quadRootRef = L.quadRootRef;
	c->scene.nQuads = L.nQuads;
	c->scene.nStaged = L.nStaged;
	c->scene.quadMaxStack = L.quadMaxStack;
	c->scene.nodes = c->dNodes;
	c->scene.tris = c->dTris;
	std::memcpy(c->scene.rootMin, L.rootMin, 12);
	std::memcpy(c->scene.rootMax, L.rootMax, 12);
	c->scene.rootRef = 0u; // the pair layout's root (pair 0), as the host pass answers without pair nodes
	c->scene.nPairs = 0;
	c->scene.nPrims = static_cast<uint32_t>(nPrims);
	return TYR_OK;
}

// tyr_scene_upload with the layout pass on the device (TYR_TUNE_LAYOUT_ON_DEVICE): the reference's arrays cross the bus as they
// are -- 32 + 40 bytes per node / triangle instead of 128 + 48 -- and hip/bvh_layout_dev.hip writes the records in HBM.
// TYR_ERR_UNSUPPORTED: the host pass has to do this tree (and names the error of a malformed one).
int scene_upload_device_layout(tyr_ctx* c, const tyr_bvh_node* nodes, int32_t nNodes, const tyr_triangle* prims, int32_t nPrims) {
	const auto t0 = std::chrono::steady_clock::now();
	tyr_bvh_node* dRawNodes = nullptr;
	tyr_triangle* dRawPrims = nullptr;
	struct Guard {
		tyr_bvh_node*& a;
		tyr_triangle*& b;
		~Guard() {
			dev_free(a);
			dev_free(b);
		}
	} guard{ dRawNodes, dRawPrims };
	int rc;
	if ((rc = dev_alloc(dRawNodes, static_cast<size_t>(nNodes))) || (rc = dev_alloc(dRawPrims, static_cast<size_t>(nPrims))))
		return rc;
	HIPCHK(hipMemcpy(dRawNodes, nodes, static_cast<size_t>(nNodes) * sizeof(tyr_bvh_node), hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(dRawPrims, prims, static_cast<size_t>(nPrims) * sizeof(tyr_triangle), hipMemcpyHostToDevice));
	const auto t1 = std::chrono::steady_clock::now();
	DeviceTreeLayout L;
	if ((rc = layout_on_device(dRawNodes, nNodes, dRawPrims, nPrims, L, c->stream)))
		return rc;
	if ((rc = adopt_device_layout(c, L, nPrims)))
		return rc;
	c->uploadCopyS = std::chrono::duration<double>(t1 - t0).count();
	c->uploadLayoutS = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
	c->layoutOnDevice = true;
	return TYR_OK;
}

} // namespace

int tyr_scene_upload(tyr_ctx* c, const tyr_bvh_node* nodes, int32_t nNodes, const tyr_triangle* prims, int32_t nPrims) {
	if (!c)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	// the pair nodes are what the counting build and the BVH_DEBUG picture traverse (the reference's visit counts,
	// bvh.h:164-209); a ctx without those flags never reads them: they are neither laid out nor kept in HBM (64 MB on C3, 0.4 GB on C5)
	const bool wantPairs = (c->cfg.flags & (TYR_FLAG_COUNT_VISITS | TYR_FLAG_DEBUG_BVH)) != 0;
	c->layoutOnDevice = false;
	if (!wantPairs && c->tuning.layoutOnDevice != 0 && nodes && prims && nNodes >= 3 && nPrims > 0) {
		try {
			rc = scene_upload_device_layout(c, nodes, nNodes, prims, nPrims);
		} catch (...) {
			rc = TYR_ERR_UNSUPPORTED;
		}
		if (rc == TYR_OK)
			return upload_light_list(c, prims, nPrims);
		if (rc != TYR_ERR_UNSUPPORTED)
			return rc;
	}
	DeviceLayout L;
	const auto t0 = std::chrono::steady_clock::now();
	try { // (the layout pass allocates and starts threads: nothing may leave a C entry point as an exception)
		rc = build_device_layout(nodes, nNodes, prims, nPrims, L, wantPairs);
	} catch (const std::bad_alloc&) {
		rc = TYR_ERR_OOM;
	} catch (...) {
		rc = TYR_ERR_UNSUPPORTED;
	}
	if (rc)
		return rc;
	const auto t1 = std::chrono::steady_clock::now();
	c->uploadLayoutS = std::chrono::duration<double>(t1 - t0).count();
	c->uploadCopyS = 0.0;
	if ((rc = drop_scene(c)))
		return rc;
	if (L.rootRef == kRefDone)
		return TYR_OK; // Scene.cpp:49-52
	// at least one element so the pointers are never null
	const size_t nodeFloats = std::max<size_t>(L.pairNodes.size(), 16), quadFloats = std::max<size_t>(L.quadNodes.size(), 32), triFloats = L.tris.size();
	if ((rc = dev_alloc(c->dNodes, nodeFloats / 4)) || (rc = dev_alloc(c->dQuads, quadFloats / 4)) || (rc = dev_alloc(c->dTris, triFloats / 4)))
		return rc;
	if (!L.pairNodes.empty())
		HIPCHK(hipMemcpy(c->dNodes, L.pairNodes.data(), L.pairNodes.size() * sizeof(float), hipMemcpyHostToDevice));
	if (!L.quadNodes.empty())
		HIPCHK(hipMemcpy(c->dQuads, L.quadNodes.data(), L.quadNodes.size() * sizeof(float), hipMemcpyHostToDevice));
	c->scene.quads = c->dQuads;
	c->scene.quadRootRef = L.quadRootRef;
	c->scene.nQuads = L.nQuads;
	c->scene.nStaged = L.nStaged;
	c->scene.quadMaxStack = L.quadMaxStack;
	HIPCHK(hipMemcpy(c->dTris, L.tris.data(), triFloats * sizeof(float), hipMemcpyHostToDevice));
	c->scene.nodes = c->dNodes;
	c->scene.tris = c->dTris;
	std::memcpy(c->scene.rootMin, L.rootMin, 12);
	std::memcpy(c->scene.rootMax, L.rootMax, 12);
	c->scene.rootRef = L.rootRef;
	c->scene.nPairs = L.nPairs;
	c->scene.nPrims = static_cast<uint32_t>(nPrims);
	c->uploadCopyS = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count(); // (hipMemcpy from pageable memory returns when the data is on its way from a staging buffer at the latest; the three arrays, allocation included)
	return upload_light_list(c, prims, nPrims);
}

// Scene::Load's two halves in one call (Scene.cpp:49-67): the tree built on the ctx's device (hip/bvh_build_dev.hip: the reference's
// bytes) and laid out there (hip/bvh_layout_dev.hip) without the nodes ever leaving HBM.  prims is reordered in place as the
// reference's BVH constructor does (bvh.cpp:24); nodes_out (may be null, else 2n - 1 entries) receives the reference's node array.
int tyr_scene_build_upload(tyr_ctx* c, tyr_triangle* prims, int32_t n, const tyr_bbox* bboxes, tyr_bvh_node* nodes_out, int32_t* n_nodes_out, double* seconds_out3) {
	if (!c || n < 0 || (n > 0 && (!prims || !bboxes)))
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	if (n_nodes_out)
		*n_nodes_out = 0;
	if (seconds_out3)
		seconds_out3[0] = seconds_out3[1] = seconds_out3[2] = 0.0;
	if (n == 0)
		return tyr_scene_upload(c, nullptr, 0, nullptr, 0);
	const bool wantPairs = (c->cfg.flags & (TYR_FLAG_COUNT_VISITS | TYR_FLAG_DEBUG_BVH)) != 0;
	std::vector<tyr_bvh_node> own;
	auto host_nodes = [&]() -> tyr_bvh_node* {
		if (nodes_out)
			return nodes_out;
		if (own.empty())
			own.resize(2 * static_cast<size_t>(n) - 1);
		return own.data();
	};
	try {
		// the ways round the device: both halves on the host (its builder, then tyr_scene_upload) ...
		auto all_on_host = [&]() -> int {
			const auto t0 = std::chrono::steady_clock::now();
			const int nn = bvh_build(prims, n, bboxes, host_nodes(), 2);
			if (nn < 0)
				return nn;
			if (seconds_out3)
				seconds_out3[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			if (n_nodes_out)
				*n_nodes_out = nn;
			return tyr_scene_upload(c, host_nodes(), nn, prims, n);
		};
		if (wantPairs)
			return all_on_host();
		DeviceBuild B;
		double secs[2] = { 0.0, 0.0 };
		const int nNodes = bvh_build_device_keep(c->cfg.device, prims, n, bboxes, B, secs);
		if (nNodes == TYR_ERR_UNSUPPORTED)
			return all_on_host();
		if (nNodes < 0)
			return nNodes;
		const auto t1 = std::chrono::steady_clock::now();
		HIPCHK(hipMemcpy(prims, B.prims, static_cast<size_t>(n) * sizeof(tyr_triangle), hipMemcpyDeviceToHost)); // bvh.cpp:24: the caller's array in its new order
		if (nodes_out)
			HIPCHK(hipMemcpy(nodes_out, B.nodes, static_cast<size_t>(nNodes) * sizeof(tyr_bvh_node), hipMemcpyDeviceToHost));
		double copyS = secs[1] + std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
		if (n_nodes_out)
			*n_nodes_out = nNodes;
		if (seconds_out3)
			seconds_out3[0] = secs[0];
		const auto t2 = std::chrono::steady_clock::now();
		DeviceTreeLayout L;
		rc = c->tuning.layoutOnDevice != 0 ? layout_on_device(B.nodes, nNodes, B.prims, n, L, c->stream) : TYR_ERR_UNSUPPORTED;
		if (rc == TYR_ERR_UNSUPPORTED) { // ... or the tree built here and laid out there
			if (!nodes_out)
				HIPCHK(hipMemcpy(host_nodes(), B.nodes, static_cast<size_t>(nNodes) * sizeof(tyr_bvh_node), hipMemcpyDeviceToHost));
			return tyr_scene_upload(c, host_nodes(), nNodes, prims, n);
		}
		if (rc)
			return rc;
		if ((rc = adopt_device_layout(c, L, n)))
			return rc;
		c->uploadLayoutS = std::chrono::duration<double>(std::chrono::steady_clock::now() - t2).count();
		c->uploadCopyS = copyS;
		c->layoutOnDevice = true;
		if (seconds_out3) {
			seconds_out3[1] = c->uploadLayoutS;
			seconds_out3[2] = copyS;
		}
		return upload_light_list(c, prims, n);
	} catch (const std::bad_alloc&) {
		return TYR_ERR_OOM;
	} catch (...) {
		return TYR_ERR_UNSUPPORTED;
	}
}

int tyr_set_triangle_emission(tyr_ctx* c, const float* rgb) {
	if (!c || !rgb || !finite_n(rgb, 3))
		return TYR_ERR_INVALID;
	std::memcpy(c->triEmission, rgb, 12);
	return TYR_OK;
}

int tyr_set_triangle_palette(tyr_ctx* c, const float* color_rgb256, const float* emission_rgb256) {
	if (!c || !color_rgb256 || !finite_n(color_rgb256, 768) || (emission_rgb256 && !finite_n(emission_rgb256, 768)))
		return TYR_ERR_INVALID;
	if (!c->dPalette)
		return TYR_ERR_UNSUPPORTED; // the ctx was created without TYR_FLAG_TRIANGLE_COLORS
	int rc = use_device(c);
	if (rc)
		return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	std::vector<float4> pal(512);
	HIPCHK(hipMemcpy(pal.data(), c->dPalette, pal.size() * sizeof(float4), hipMemcpyDeviceToHost));
	for (int i = 0; i < 256; ++i) {
		pal[2 * i] = make_float4(color_rgb256[3 * i], color_rgb256[3 * i + 1], color_rgb256[3 * i + 2], 0.0f);
		if (emission_rgb256)
			pal[2 * i + 1] = make_float4(emission_rgb256[3 * i], emission_rgb256[3 * i + 1], emission_rgb256[3 * i + 2], 0.0f);
	}
	HIPCHK(hipMemcpy(c->dPalette, pal.data(), pal.size() * sizeof(float4), hipMemcpyHostToDevice));
	return TYR_OK;
}

int tyr_set_spheres(tyr_ctx* c, const tyr_sphere* spheres) {
	if (!c)
		return TYR_ERR_INVALID;
	if (!spheres) {
		default_spheres(c->spheres);
		return TYR_OK;
	}
	for (int i = 0; i < TYR_NUM_SPHERES; ++i) {
		const tyr_sphere& s = spheres[i];
		if (!finite_n(&s.radius, 10) || s.refl < TYR_DIFF || s.refl > TYR_LIGHT)
			return TYR_ERR_INVALID;
	}
	std::memcpy(c->spheres, spheres, sizeof(c->spheres));
	return TYR_OK;
}

int tyr_set_camera(tyr_ctx* c, const tyr_camera* cam) {
	if (!c || !cam || !finite_n(cam->position, 11))
		return TYR_ERR_INVALID;
	c->cam = *cam;
	return TYR_OK;
}

int tyr_set_sun_position(tyr_ctx* c, float x, float y) {
	if (!c || !std::isfinite(x) || !std::isfinite(y))
		return TYR_ERR_INVALID;
	c->sunPos[0] = x;
	c->sunPos[1] = y;
	c->sunChanged = true;
	return TYR_OK;
}

int tyr_set_blit_buffer(tyr_ctx* c, void* device_float4) {
	if (!c)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	if (c->ownBlit)
		dev_free(c->blit);
	c->ownBlit = false;
	c->blit = static_cast<float4*>(device_float4);
	if (!c->blit) {
		const size_t n = static_cast<size_t>(c->cfg.width) * c->cfg.height;
		if ((rc = dev_alloc(c->blit, n)))
			return rc;
		c->ownBlit = true;
		HIPCHK(hipMemset(c->blit, 0, n * sizeof(float4)));
	}
	return TYR_OK;
}

void* tyr_get_blit_buffer(tyr_ctx* c) { return c ? c->blit : nullptr; }

int tyr_set_budget(tyr_ctx* c, uint64_t primary_rays) {
	if (!c)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	c->hK->budget_remaining = primary_rays;
	return push_counters(c);
}

int tyr_set_frame(tyr_ctx* c, uint32_t frame) {
	if (!c || frame == 0)
		return TYR_ERR_INVALID;
	c->frame = frame;
	return TYR_OK;
}

int tyr_get_counters(tyr_ctx* c, tyr_counters* out) {
	if (!c || !out)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	const DevCounters& k = *c->hK;
	out->primary_ray_cnt = k.primary_ray_cnt;
	out->start_position = k.start_position;
	out->shadow_ray_cnt = k.shadow_ray_cnt;
	out->n_live = k.n_live;
	out->frame = c->frame;
	out->device_error = k.device_error;
	out->budget_remaining = k.budget_remaining;
	out->total_extend_rays = k.total_extend_rays;
	out->total_shadow_rays = k.total_shadow_rays;
	out->total_primary_rays = k.total_primary_rays;
	out->nodes_extend = k.nodes_extend;
	out->tris_extend = k.tris_extend;
	out->nodes_connect = k.nodes_connect;
	out->tris_connect = k.tris_connect;
	out->n_survive = k.n_survive;
	out->n_shadow_visible = k.n_shadow_visible;
	out->rays_in_tree_extend = k.rays_in_tree_extend;
	out->rays_in_tree_connect = k.rays_in_tree_connect;
	for (int i = 0; i < 16; ++i)
		out->debug[i] = k.debug[i];
	return TYR_OK;
}

// ---- stage-level API -----------------------------------------------------------------------
int tyr_stage_begin(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	if (!c->haveScene)
		return TYR_ERR_NO_SCENE;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	if ((rc = stage_begin(c)))
		return rc;
	return sync_counters(c);
}
int tyr_stage_primary(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	const uint32_t nNew = planned_new(c), nLive = c->hK->primary_ray_cnt + nNew;
	(void)nLive;
	enqueue_primary(c, make_params(c), nNew);
	HIPCHK(hipGetLastError());
	rc = sync_counters(c);
	collect_timings(c);
	return rc;
}
int tyr_stage_extend(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	enqueue_extend(c, make_params(c), c->hK->n_live, c->hK->n_live); // the host mirror no longer has the survivor count: upper bound
	HIPCHK(hipGetLastError());
	rc = sync_counters(c);
	collect_timings(c);
	return rc ? rc : check_device_error(c);
}
int tyr_stage_shade(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	enqueue_shade(c, make_params(c), c->hK->n_live);
	HIPCHK(hipGetLastError());
	rc = sync_counters(c);
	collect_timings(c);
	return rc ? rc : check_device_error(c);
}
int tyr_stage_connect(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	enqueue_connect(c, make_params(c), c->hK->shadow_ray_cnt);
	HIPCHK(hipGetLastError());
	rc = sync_counters(c);
	collect_timings(c);
	return rc ? rc : check_device_error(c);
}
int tyr_stage_end(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	stage_end(c);
	return TYR_OK;
}
int tyr_sync(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	return sync_counters(c);
}

// ---- the per-frame entry point --------------------------------------------------------------
// One wavefront iteration.  pipelined = false is launch_kernels as the reference has it: primary, extend, shade, connect on
// one stream, done when it returns (kernel.cu:719-733).  Inside tyr_render (pipelined, merged launches) connect(i) rides in
// the traversal launch of iteration i + 1 and the call returns as soon as shade's counts are on the host.
#ifdef TYR_LAUNCH_ANATOMY
// TYR_ANATOMY=2: the per-wave records k_trace_flat's anatomy build leaves in the next queue's hit column
static void print_wave_anatomy(const float2* dHit, unsigned long long feedTicks) {
	std::vector<float2> rec(5 * 8192);
	if (hipMemcpy(rec.data(), dHit, rec.size() * sizeof(float2), hipMemcpyDeviceToHost) != hipSuccess)
		return;
	std::vector<float> drain, normal, wide, perTrip, perStep, liveExh, liveWide, trips, steps, passes, handoffs;
	for (uint32_t w = 0; w < 8192; ++w) {
		const float tExh = rec[w].x, tEnd = rec[w].y, tWide = rec[8192 + w].x;
		if (!(tExh > 0.0f) || !(tEnd >= tExh) || !(tEnd < 1e5f))
			continue;
		const uint32_t lv = (uint32_t)rec[8192 + w].y;
		const float nTrips = rec[16384 + w].x, nSteps = rec[16384 + w].y;
		drain.push_back(tEnd - tExh);
		normal.push_back((tWide > 0.0f ? tWide : tEnd) - tExh);
		wide.push_back(tWide > 0.0f ? tEnd - tWide : 0.0f);
		if (nTrips > 0.0f)
			perTrip.push_back(((tWide > 0.0f ? tWide : tEnd) - tExh) / nTrips);
		if (nSteps > 0.0f && tWide > 0.0f)
			perStep.push_back((tEnd - tWide) / nSteps);
		liveExh.push_back((float)(lv & 255u));
		liveWide.push_back((float)(lv >> 8));
		trips.push_back(nTrips);
		steps.push_back(nSteps);
		passes.push_back(rec[24576 + w].x);
		handoffs.push_back(rec[32768 + w].x);
	}
	auto pct = [](std::vector<float>& v, double p) {
		if (v.empty())
			return 0.0f;
		const size_t k = (size_t)(p * (v.size() - 1));
		std::nth_element(v.begin(), v.begin() + k, v.end());
		return v[k];
	};
	auto line = [&](const char* name, std::vector<float>& v) { std::fprintf(stderr, "[anatomy]    %-44s n %5zu  median %8.2f  90 %% %8.2f  99 %% %8.2f  max %8.2f\n", name, v.size(), pct(v, 0.5), pct(v, 0.9), pct(v, 0.99), pct(v, 1.0)); };
	std::fprintf(stderr, "[anatomy]  per wave, after the queue ran out (feed %.1f us):\n", feedTicks / 100.0);
	line("drain: exit - 'used up' [us]", drain);
	line("  one ray to a lane [us]", normal);
	line("  four lanes to a ray [us]", wide);
	line("rays held when the queue ran out", liveExh);
	line("rays held on going wide", liveWide);
	line("descent trips one ray to a lane", trips);
	line("outer passes (leaf rounds) one ray to a lane", passes);
	line("steps four lanes to a ray", steps);
	line("hand-offs four lanes to a ray (steal build)", handoffs);
	line("us per trip, one ray to a lane", perTrip);
	line("us per step, four lanes to a ray", perStep);
	{
		// the feed phase: microseconds per descent trip while the queue lasted
		std::vector<float> feedBusy;
		for (uint32_t w = 0; w < 8192; ++w) {
			const float tExh = rec[w].x, n = rec[24576 + w].y;
			if (!(tExh > 0.0f) || !(n > 0.0f))
				continue;
			feedBusy.push_back(tExh / n);
		}
		line("feed phase: us per trip", feedBusy);
	}
	// the launch ends with these: the five waves that left last
	std::vector<uint32_t> order;
	for (uint32_t w = 0; w < 8192; ++w)
		if (rec[w].x > 0.0f && rec[w].y >= rec[w].x && rec[w].y < 1e5f)
			order.push_back(w);
	std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return rec[a].y > rec[b].y; });
	for (size_t i = 0; i < order.size() && i < 5; ++i) {
		const uint32_t w = order[i];
		const float tExh = rec[w].x, tEnd = rec[w].y, tWide = rec[8192 + w].x;
		const uint32_t lv = (uint32_t)rec[8192 + w].y;
		std::fprintf(stderr, "[anatomy]    last wave %zu: exit at %.1f us; queue used up at %.1f (%u rays held), %.1f us / %.0f trips one ray to a lane, %.1f us / %.0f steps four lanes to a ray (from %u rays)\n", i + 1, tEnd, tExh,
		             lv & 255u, (tWide > 0.0f ? tWide : tEnd) - tExh, rec[16384 + w].x, tWide > 0.0f ? tEnd - tWide : 0.0f, rec[16384 + w].y, lv >> 8);
	}
}
#endif

static int launch_iteration(tyr_ctx* c, bool pipelined) {
	// hK is current: every entry point that enqueues work ends with sync_counters
	int rc = stage_begin(c);
	if (rc)
		return rc;
	const uint32_t nNew = planned_new(c), nLive = c->hK->primary_ray_cnt + nNew;
	FrameParams P = make_params(c);
	if (c->cfg.flags & TYR_FLAG_DEBUG_BVH) {
		// kernel.cu:720-722 under BVH_DEBUG: primary_rays, set_wavefront_globals, extend_debug_BVH -- no shade, no connect
		// (nothing survives: the next call regenerates the whole queue from the cursor)
		enqueue_primary(c, P, nNew);
		enqueue_extend(c, P, nLive, nLive - nNew);
		HIPCHK(hipGetLastError());
		rc = sync_counters(c);
		collect_timings(c);
		stage_end(c);
		return rc ? rc : check_device_error(c);
	}
	const bool merge = pipelined && merged_render(c);
	if (merge && c->tuning.foldSpheres) {
		P.foldSpheres = 1u; // this iteration's shade does the sphere halves for the rays it emits
		P.resolveShadows = c->tuning.resolveShadows ? 1u : 0u; // ... and answers the shadow rays that cannot reach a triangle
		P.retireGhosts = (c->tuning.retireSky && c->unboundedRender) ? 1u : 0u; // ... and finishes the survivors that will hit nothing (a render cut short would see their pixels an iteration early)
	}
	if (merge && c->tuning.retireSky)
		P.retireSky = 1u;   // ... and k_primary finishes the camera rays that hit nothing
	enqueue_primary(c, P, nNew);
	if (merge) { // every traversal launch of a merged render is k_trace_flat; the first one has no shadow rays to carry yet
		const uint32_t carried = c->shadowPending ? c->shadowPendingMax : 0u;
		c->shadowPending = false;
		enqueue_trace(c, P, nLive, nLive - nNew, carried);
		enqueue_shade(c, P, nLive);
	} else {
		if ((rc = flush_pending_shadow(c))) // (a render whose merge setting changed between iterations: never, but cheap)
			return rc;
		enqueue_extend(c, P, nLive, nLive - nNew);
		enqueue_shade(c, P, nLive);
	}
	if (merge) {
		// everything the host needs to launch iteration i + 1 (survivors, budget, the shadow-ray count) is final here
		HIPCHK(hipMemcpyAsync(c->hK, c->dK, sizeof(DevCounters), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipEventRecord(c->evSnapshot, c->stream));
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventSynchronize(c->evSnapshot));
		c->shadowPending = c->hK->shadow_ray_cnt != 0;
		c->shadowPendingMax = c->hK->shadow_ray_cnt;
#if defined(TYR_QUAD_STATS) || defined(TYR_LAUNCH_ANATOMY)
		if (std::getenv("TYR_ANATOMY")) {
			// launch anatomy of this iteration's traversal launch (s_memrealtime, 100 MHz): first wave's start, first wave to
			// find the queue used up, last wave's exit
			const unsigned long long t0 = ~c->hK->debug[13], tx = ~c->hK->debug[14], t1 = c->hK->debug[15];
			std::fprintf(stderr, "[anatomy] iteration %u: %u rays: feed %.1f us, drain %.1f us", c->iter, nLive, (tx - t0) / 100.0, (t1 - tx) / 100.0);
#ifdef TYR_QUAD_STATS
			std::fprintf(stderr, "; longest ray %llu quad steps, rays with > 64 / 128 / 256 steps: %llu / %llu / %llu (running totals)", c->hK->debug[12], c->hK->debug[9], c->hK->debug[10], c->hK->debug[11]);
#endif
			std::fprintf(stderr, "\n");
#ifdef TYR_LAUNCH_ANATOMY
			if (std::getenv("TYR_ANATOMY")[0] == '2' && P.N > 40960u)
				print_wave_anatomy(P.next.hit, tx - t0);
#endif
		}
#endif
	} else {
		enqueue_connect(c, P, nLive); // at most one shadow ray per live ray
		HIPCHK(hipGetLastError());
		rc = sync_counters(c); // kernel.cu:733 cudaDeviceSynchronize
	}
	collect_timings(c);
	stage_end(c);
	return rc ? rc : check_device_error(c);
}

int tyr_launch_kernels(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	if (!c->haveScene)
		return TYR_ERR_NO_SCENE;
	int rc = use_device(c);
	if (rc)
		return rc;
	return launch_iteration(c, false);
}

// ---- tyr_render, one iteration ahead of the counts (TYR_TUNE_RUN_AHEAD) -------------------------
// Between shade(i) and the first kernel of iteration i + 1 the stream used to run dry for ~20 us: the counters travel
// to the host, the host wakes up, sizes the grids and launches.  Nothing in iteration i + 1 needs the host for that:
// k_primary (and set_wavefront_globals in its last block) compute the top-up from the device's counters (kernel.cu:253, 227-244 do the same), the
// persistent kernels read their item counts there, k_shade its tile count.  So the host queues iteration i + 1 right
// behind iteration i, sizing every grid from upper bounds it can already compute -- survivors(i) <= live(i), shadow rays
// (i) <= live(i), and live(i), the budget and the top-up of i + 1's predecessors follow exactly from the last counts that
// DID arrive -- and waits for iteration i's counts afterwards, with iteration i + 1 already running or queued.
// It learns one iteration late that the render has ended (no survivors, no budget): that last iteration has no rays of
// its own and traces the final shadow rays -- the connect launch a merged render needs at its end anyway -- and the
// host takes back its frame counter, queue swap and iteration parity, so that the ctx is where the reference's loop
// would have left it (kernel.cu:735-745, main.cpp:169).
struct IterationPlan {
	uint32_t nNew, nLive, nSurvivors, carried; // exact values or upper bounds; carried: shadow rays of the iteration before (0: none to trace)
};
// foldNext: this iteration's k_scan_words also opens the next one (no top-up can follow and the next one IS going to be queued);
// prologueDone: the previous iteration's did that for this one -- no k_primary launch, no k_pad_holes
static int enqueue_merged_iteration(tyr_ctx* c, const IterationPlan& p, bool begun, bool foldNext = false, bool prologueDone = false) {
	int rc = begun ? TYR_OK : stage_begin(c);
	if (rc)
		return rc;
	const int set = static_cast<int>(c->iter & 1u);
	FrameParams P = make_params(c);
	if (c->tuning.foldSpheres) {
		P.foldSpheres = 1u;
		P.resolveShadows = c->tuning.resolveShadows ? 1u : 0u;
		P.retireGhosts = (c->tuning.retireSky && c->unboundedRender) ? 1u : 0u;
	}
	if (c->tuning.retireSky)
		P.retireSky = 1u;
	const bool aside = foldNext && c->tuning.scanInTrace != 0;
	P.foldNextPrologue = (foldNext && !aside) ? 1u : 0u;
	P.shadeOpensNext = aside ? 1u : 0u; // k_shade's last block opens the next iteration, whose traversal launch does this iteration's slot scan on its way in (TYR_TUNE_SCAN_IN_TRACE)
	if (aside) {
		P.scanSet = static_cast<uint32_t>(set);
		P.scanLive = &c->dK->scan_live[set];
	}
	P.prologueDone = prologueDone ? 1u : 0u;
	// the counts the loop waits for: written by k_shade's last block into pinned host memory (nothing in the stream between this shade
	// launch and the next traversal launch; a ctx that times its stages still has their event pairs there)
	const bool kernelSnap = c->tuning.kernelSnapshot != 0;
	c->snapSeqOf[set] = 0;
	if (kernelSnap) {
		if (++c->snapSeq == 0u)
			++c->snapSeq;
		c->snapSeqOf[set] = c->snapSeq;
		P.hostSnap = c->hostSnapDev[set];
		P.snapSeq = c->snapSeq;
	}
	if (!prologueDone)
		enqueue_primary(c, P, p.nNew);
	enqueue_trace(c, P, p.nLive, p.nSurvivors, p.carried);
	enqueue_shade(c, P, p.nLive);
	if (!kernelSnap) {
		HIPCHK(hipMemcpyAsync(c->hSnap[set], c->dK, sizeof(DevCounters), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipEventRecord(c->evSnap[set], c->stream));
	}
	HIPCHK(hipGetLastError());
	stage_end(c);
	return TYR_OK;
}
static bool run_ahead_eligible(const tyr_ctx* c) {
#if defined(TYR_QUAD_STATS) || defined(TYR_LAUNCH_ANATOMY)
	return false; // the instrumented builds print per-iteration records from the host mirror (launch_iteration)
#else
	const bool wanted = c->tuning.runAhead != 0; // (2 meant "queues of at most 6 Mi slots" while a render's last iteration was followed by an empty one: render_run_ahead's lastBirth)
	return wanted && merged_render(c) && c->blit != nullptr;
#endif
}
static int render_run_ahead(tyr_ctx* c, uint32_t max_iterations, uint32_t& it) {
	it = 0;
	if (max_iterations == 0)
		return TYR_OK;
	int rc = flush_pending_shadow(c);
	if (rc)
		return rc;
	if ((rc = stage_begin(c))) // may reset the accumulation and the survivor count (kernel.cu:712-718): before the plan is made
		return rc;
	const uint64_t N = c->cfg.queue_size;
	// exact state in front of iteration 0 (hK is current: every entry point ends with sync_counters)
	uint64_t s = c->hK->primary_ray_cnt, budget = c->hK->budget_remaining;
	uint32_t nNew = static_cast<uint32_t>(std::min<uint64_t>(N - s, budget));
	uint32_t live = static_cast<uint32_t>(s) + nNew; // live(enq - 1), exact
	budget -= nNew;                                   // budget left behind iteration enq - 1, exact
	const uint32_t iter0 = c->iter; // iteration j of this render is the ctx's iteration iter0 + j: its events and its counters use set (iter0 + j) & 1
	// Will iteration j + 1 be queued without a look at iteration j's counts, and can it do without a top-up?  Then iteration j's
	// last kernel opens it (FrameParams::foldNextPrologue): set_wavefront_globals and the hole padding cost a ~5 us launch and a
	// gap between dependent kernels each, every iteration.  Both answers follow from what the host knows when it queues j: the
	// budget left behind j (exact once it is zero) and the last iteration that gave birth to rays.
	const bool mayFold = c->tuning.foldPrologue != 0 && c->tuning.foldSpheres != 0;
	// INVARIANT the kernels rely on: an iteration that turns out to have no rays (n_live == 0: the one queued ahead of its predecessor's
	// counts for nothing) is never followed by another -- the loop below returns when it sees "budget == 0 && s == 0" -- so the kernels that
	// would open its successor (k_scan_words' and k_shade's last blocks) skip that when n_live is 0, and the counters of the last real
	// iteration stay what tyr_shadow_export reads.
	auto queued_ahead_behind = [&](uint32_t j, uint64_t budgetBehindJ, uint32_t lastBirthAtJ) { return j + 1 < max_iterations && (budgetBehindJ != 0 || j < lastBirthAtJ + static_cast<uint32_t>(kMaxBounces)); };
	bool folded = mayFold && budget == 0 && queued_ahead_behind(0, budget, 0); // (of the iteration queued last: its k_scan_words has opened the next one)
	if ((rc = enqueue_merged_iteration(c, IterationPlan{ nNew, live, static_cast<uint32_t>(s), 0u }, true, folded, false)))
		return rc;
	uint32_t enq = 1;
	// The last iteration (of this render) that gave birth to rays: a primary ray survives at most kMaxBounces times
	// (kernel.cu:600-607), so shade of iteration lastBirth + kMaxBounces leaves no survivor -- once the budget is spent the
	// render's end is known in advance and no iteration has to be queued ahead for nothing.  (Survivors the ctx held when the
	// render began count as born in iteration 0: their bounce counts are not known here.)
	uint32_t lastBirth = 0;
	for (;;) {
		// iterations 0 .. enq - 1 are queued; the counts of 0 .. enq - 2 have arrived
		bool ahead = false;
		bool foldedPrev = folded; // whether the iteration whose counts are awaited below (enq - 1) opened its successor
		uint32_t frameBefore = c->frame;
		const uint32_t shadowSetBefore = c->shadowSet; // (enqueue_shade of an iteration queued ahead moves it: an iteration that turns out empty must give it back, or tyr_shadow_export would read the empty iteration's counters)
		const bool foldedBefore = c->lastShadeFolded;
		const bool canHaveSurvivors = budget != 0 || enq - 1 < lastBirth + static_cast<uint32_t>(kMaxBounces); // of iteration enq - 1
		if (enq < max_iterations && canHaveSurvivors) {
			const uint32_t liveMax = static_cast<uint32_t>(std::min<uint64_t>(N, static_cast<uint64_t>(live) + budget));
			const uint32_t newMax = static_cast<uint32_t>(std::min<uint64_t>(N, budget));
			const bool opened = folded; // iteration enq - 1's k_scan_words has done this one's set_wavefront_globals and hole padding
			foldedPrev = folded;
			folded = mayFold && budget == 0 && queued_ahead_behind(enq, 0, lastBirth); // (budget == 0: iteration enq tops nothing up, gives birth to nothing)
			if ((rc = enqueue_merged_iteration(c, IterationPlan{ newMax, liveMax, live, live }, false, folded, opened))) {
				(void)hipStreamSynchronize(c->stream); // (the failed iteration may be partly queued; nothing of it is the render's)
				c->scanCarried = false;
				c->shadowSet = shadowSetBefore;
				c->lastShadeFolded = foldedBefore;
				return rc;
			}
			ahead = true;
		}
		// The render's end is known (the budget is spent, iteration enq - 1 cannot leave a survivor): the launch that traces its last
		// shadow rays goes out now, sized from an upper bound (at most one shadow ray per ray; the kernel takes its counts from the
		// device), instead of after the ~25 us it takes the counts to reach the host and the launch to reach the GPU.
		bool flushedEarly = false;
		if (!ahead && mayFold && budget == 0 && !canHaveSurvivors) {
			c->shadowPending = true;
			c->shadowPendingMax = live;
			if ((rc = flush_pending_shadow(c)))
				return rc;
			flushedEarly = true;
		}
		const int set = static_cast<int>((iter0 + enq - 1) & 1u);
		// a failure from here on leaves an iteration queued that the render will never own: drain the stream and take the
		// host's bookkeeping of it back, so that the ctx is where its last completed iteration left it
		auto abandon = [&](int code) {
			// (also when nothing was queued ahead: kernels of iteration enq - 1 may still be running and would go on writing the snapshot
			// record and the blit buffer behind an error return)
			(void)hipStreamSynchronize(c->stream);
			c->scanCarried = false;
			if (ahead) {
				c->frame = frameBefore;
				c->cur ^= 1;
				c->iter--;
				c->shadowPending = false;
				c->shadowSet = shadowSetBefore;
				c->lastShadeFolded = foldedBefore;
			}
			return code;
		};
		if (c->snapSeqOf[set] != 0u) {
			// the kernel-written snapshot: poll its stamp (the stream is looked at now and then: a fault must not hang the host)
			volatile tyr::HostSnap* const hs = c->hostSnap[set];
			const uint32_t want = c->snapSeqOf[set];
			// An iteration is tens to hundreds of microseconds: spin.  The stream is looked at every 16 K spins -- idle (or failed)
			// without the stamp is the only verdict; a slow iteration (a serialising profiler, a very large scene) is waited for as
			// hipStreamSynchronize would, and once the wait is past a few milliseconds the core is given back between looks.
			for (uint32_t spins = 0; __atomic_load_n(&hs->seq, __ATOMIC_ACQUIRE) != want; ++spins) {
				if ((spins & 0x3fffu) == 0x3fffu) {
					const hipError_t q = hipStreamQuery(c->stream);
					if (q != hipErrorNotReady && __atomic_load_n(&hs->seq, __ATOMIC_ACQUIRE) != want) // idle (or failed) without the stamp
						return abandon(q == hipSuccess ? TYR_ERR_DEVICE : static_cast<int>(q));
					if (spins >= (1u << 20))
						std::this_thread::sleep_for(std::chrono::microseconds(50));
				}
#if defined(__x86_64__)
				__builtin_ia32_pause();
#endif
			}
			c->hK->primary_ray_cnt = hs->survivors;
			c->hK->shadow_ray_cnt = hs->shadows;
			c->hK->device_error = hs->device_error;
			c->hK->n_live = live;
		} else {
			const hipError_t e = hipEventSynchronize(c->evSnap[set]);
			if (e != hipSuccess)
				return abandon(static_cast<int>(e));
			std::memcpy(c->hK, c->hSnap[set], sizeof(DevCounters));
		}
		if (c->snapSeqOf[set] == 0u && foldedPrev) {
			// iteration enq - 1's k_scan_words ran the next iteration's set_wavefront_globals in front of this snapshot: the two counts
			// the host steers by were kept aside (DevCounters::reserved0 / reserved1), n_live already reads the next iteration's
			c->hK->primary_ray_cnt = c->hK->reserved0;
			c->hK->shadow_ray_cnt = c->hK->reserved1;
			c->hK->n_live = live;
		}
		collect_timings_of(c, set);
		s = c->hK->primary_ray_cnt; // survivors of iteration enq - 1
		const uint32_t shadows = c->hK->shadow_ray_cnt;
		if ((rc = check_device_error(c)))
			return abandon(rc);
		it = enq; // (counted once it is known to have completed without a device error)
		if (budget == 0 && s == 0) { // kernel loop of the reference's caller: nothing left to trace or to start
			if (ahead) {
				// iteration enq was queued for nothing but the shadow rays of iteration enq - 1: take the host state back
				c->frame = frameBefore;
				c->cur ^= 1;
				c->iter--;
				c->shadowPending = false;
				c->shadowSet = shadowSetBefore;
				c->lastShadeFolded = foldedBefore;
				c->scanCarried = false; // (the empty iteration's shade launch left no scan behind: its last block opens nothing when n_live is 0)
				c->runAheadUndo = true;
				c->undoLive = live;
				c->undoShadows = shadows;
			} else {
				c->shadowPending = !flushedEarly && shadows != 0;
				c->shadowPendingMax = shadows;
			}
			return TYR_OK;
		}
		if (!ahead && enq >= max_iterations) { // max_iterations reached
			c->shadowPending = !flushedEarly && shadows != 0;
			c->shadowPendingMax = shadows;
			return TYR_OK;
		}
		// iteration enq is real; what it does, exactly, now that its predecessor's survivors are known
		nNew = static_cast<uint32_t>(std::min<uint64_t>(N - s, budget));
		if (nNew != 0)
			lastBirth = enq;
		live = static_cast<uint32_t>(s) + nNew;
		budget -= nNew;
		if (!ahead) {
			// (it was not queued ahead because no survivor was expected, and there are some: cannot happen while a ray survives
			// at most kMaxBounces times -- queued now, from the exact counts, rather than trusted)
			folded = false;
			if ((rc = enqueue_merged_iteration(c, IterationPlan{ nNew, live, static_cast<uint32_t>(s), flushedEarly ? 0u : shadows }, false, false, false)))
				return rc;
		}
		++enq;
	}
}

// ---- tyr_render, the streamed tail (TYR_TUNE_STREAM_TAIL; hip/kernels.hpp "the STREAMED TAIL of a render") -----------------
// Once the budget is spent no iteration tops the queue up any more and each is thinner than the last: as launches they
// cost a traversal drain (~300 us), a shade launch, a scan and a host round trip apiece.  Here the remaining iterations
// are queued in one go: k_trace_stream on the ctx stream -- one launch that lives until the render ends -- and, on the side
// stream, one k_shade_stream + k_scan_words per iteration the tail can still have (a ray that has survived b times ends by
// iteration kMaxBounces + 1 - b of the tail; launches beyond the render's end return at once).  The host only waits for
// the two streams and reads how far the tail went.
constexpr uint32_t kTailLaunches = static_cast<uint32_t>(kMaxBounces) + 1u;
static_assert(kTailLaunches + 1u <= tyr::kStreamMaxIters, "the streamed tail writes StreamIter[streamIter + 1] from each of its shade launches");
static bool stream_tail_eligible(const tyr_ctx* c, uint32_t iterationsLeft) {
#if defined(TYR_QUAD_STATS) || defined(TYR_LAUNCH_ANATOMY)
	return false; // the instrumented builds stamp k_trace_flat's launches
#else
	if (const char* e = std::getenv("TYR_STREAM_TAIL"))
		if (e[0] == '0')
			return false;
	return c->tuning.streamTail != 0 && merged_render(c) && c->blit != nullptr && iterationsLeft >= kTailLaunches && c->hK->budget_remaining == 0 && c->hK->primary_ray_cnt != 0;
#endif
}
static int render_stream_tail(tyr_ctx* c, uint32_t& it) {
	int rc = stage_begin(c); // (the camera has not moved inside a render: no reset)
	if (rc)
		return rc;
	// What the host will have changed by the time anything can fail: taken back by bail().  From the first launch on a failure
	// leaves work queued on both streams -- k_trace_stream polls for chunks nobody will publish until its bounded waits
	// (kStreamTimeoutTicks) run out -- so the way out is always: wait for both streams, put the host's bookkeeping back where the
	// last completed iteration left it, and have the next tail re-zero its hand-off counters.
	const uint32_t frame0 = c->frame, iter0 = c->iter, shadowSet0 = c->shadowSet;
	const int cur0 = c->cur;
	const bool folded0 = c->lastShadeFolded, carried = c->shadowPending;
	auto bail = [&](int code) {
		(void)hipStreamSynchronize(c->side);
		(void)hipStreamSynchronize(c->stream);
		c->frame = frame0;
		c->iter = iter0;
		c->cur = cur0;
		c->shadowSet = shadowSet0;
		c->lastShadeFolded = folded0;
		c->shadowPending = false; // (the carried shadow rays may or may not have been traced: the render is void either way)
		c->streamDirty = true;
		(void)sync_counters(c); // hK follows what the device did, whatever it was
		(void)hipGetLastError();
		return code > 0 ? TYR_ERR_DEVICE : code; // (a raw hipError_t of a launch or copy: the device-side failure code of this API)
	};
#define TAILCHK(expr)                \
	do {                             \
		if ((expr) != hipSuccess)    \
			return bail(TYR_ERR_DEVICE); \
	} while (0)
	const size_t chunks = static_cast<size_t>(c->segCap) * tyr::kSegs / 64 + 8, tiles = static_cast<size_t>(c->segCap) * tyr::kSegs / 256 + 8;
	if (c->streamDirty) {
		for (int t = 0; t < 2; ++t) {
			TAILCHK(hipMemsetAsync(c->fillRay[t], 0, chunks * 4, c->stream));
			TAILCHK(hipMemsetAsync(c->fillSh[t], 0, chunks * 4, c->stream));
			TAILCHK(hipMemsetAsync(c->doneRay[t], 0, tiles * 4, c->stream));
		}
		c->streamDirty = false;
	}
	TAILCHK(hipMemsetAsync(c->dStream, 0, sizeof(StreamState), c->stream));
	c->shadowPending = false;
	{
		FrameParams P = make_params(c);
		{
			// the sphere halves of the tail's first iteration, unless the shade launch that made its rays has done them (every
			// later iteration's are done by k_shade_stream)
			FrameParams Pp = P;
			Pp.traceShadow = carried ? 1u : 0u;
			Pp.prevFolded = c->lastShadeFolded ? 1u : 0u;
			if (!c->lastShadeFolded)
				launch_trace_prepasses(Pp, c->hK->primary_ray_cnt, carried ? c->shadowPendingMax : 0u, c->stream);
		}
		launch_stream_begin(P, carried, c->stream);
		TAILCHK(hipEventRecord(c->evTail, c->stream));
		TAILCHK(hipStreamWaitEvent(c->side, c->evTail, 0));
		KernelTimer t(c, TYR_K_EXTEND);
		launch_trace_stream(P, c->tuning.streamTracePerCU, c->numCUs, c->stream);
	}
	for (uint32_t jj = 0; jj < kTailLaunches; ++jj) {
		FrameParams P = make_params(c);
		P.streamIter = jj;
		P.scanLive = &c->dStream->it[jj].nLive;
		P.resolveShadows = c->tuning.resolveShadows ? 1u : 0u; // (k_shade<.., true> always does the sphere halves)
		launch_shade_stream(P, c->tuning.streamShadePerCU, c->numCUs, c->side);
		launch_scan(P, c->hK->primary_ray_cnt, c->side); // (an upper bound of every later iteration's rays: nothing is topped up)
		stage_end(c);
	}
	TAILCHK(hipMemcpyAsync(c->hStream, c->dStream, sizeof(StreamState), hipMemcpyDeviceToHost, c->side));
	TAILCHK(hipGetLastError());
	if (hipStreamSynchronize(c->side) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
		return bail(TYR_ERR_DEVICE);
#undef TAILCHK
	// the iterations that had rays of their own are the render's (the one behind them at most traced the last shadow rays)
	uint32_t real = 0;
	while (real < kTailLaunches && c->hStream->it[real].nLive != 0)
		++real;
	c->frame = frame0;
	c->iter = iter0;
	c->cur = cur0;
	for (uint32_t jj = 0; jj < real; ++jj)
		stage_end(c);
	c->shadowSet = (c->iter - 1u) & 1u;
	c->lastShadeFolded = true;
	it += real;
	rc = sync_counters(c);
	collect_timings(c);
	if (!rc)
		rc = check_device_error(c);
	if (rc || !c->hStream->ended)
		c->streamDirty = true;
	if (!rc && !c->hStream->ended)
		rc = TYR_ERR_DEVICE; // (cannot happen: the tail has as many shade launches as a ray can have bounces left)
	return rc;
}

int tyr_render(tyr_ctx* c, uint32_t spp, uint32_t max_iterations, uint32_t* iterations_out) {
	if (!c)
		return TYR_ERR_INVALID;
	int rc = tyr_set_budget(c, static_cast<uint64_t>(spp) * c->localPixels);
	if (rc)
		return rc;
	if (!c->haveScene)
		return TYR_ERR_NO_SCENE;
	uint32_t it = 0;
	c->unboundedRender = max_iterations == 0xFFFFFFFFu;
	if (run_ahead_eligible(c)) {
		if ((rc = use_device(c)))
			return rc;
		rc = render_run_ahead(c, max_iterations, it);
	} else {
		while (it < max_iterations) {
			if (stream_tail_eligible(c, max_iterations - it)) {
				rc = render_stream_tail(c, it); // every remaining iteration
				break;
			}
			if ((rc = launch_iteration(c, true)))
				break;
			++it;
			if (c->hK->budget_remaining == 0 && c->hK->primary_ray_cnt == 0)
				break;
		}
	}
	{
		// the last shadow rays; counters refreshed (connect's included), nothing in flight when this returns
		int rcj = rc ? TYR_OK : flush_pending_shadow(c);
		c->shadowPending = false;
		if (!rcj)
			rcj = sync_counters(c);
		if (!rcj)
			collect_timings(c); // (an iteration queued ahead may still have had its event pairs out)
		if (c->runAheadUndo) {
			// the empty iteration's set_wavefront_globals zeroed the live and shadow counts of the last real iteration
			c->runAheadUndo = false;
			if (!rcj) {
				c->hK->n_live = c->undoLive;
				c->hK->shadow_ray_cnt = c->undoShadows;
				rcj = push_counters(c);
			}
		}
		if (!rc)
			rc = rcj ? rcj : check_device_error(c);
	}
	if (iterations_out)
		*iterations_out = it;
	return rc;
}

int tyr_resolve(tyr_ctx* c, void* device_rgba_out) {
	if (!c || !device_rgba_out)
		return TYR_ERR_INVALID;
	if (!c->blit)
		return TYR_ERR_NO_BUFFER;
	int rc = use_device(c);
	if (rc)
		return rc;
	{
		KernelTimer t(c, TYR_K_RESOLVE);
		launch_resolve(c->blit, static_cast<float4*>(device_rgba_out), c->cfg.width * c->cfg.height, c->stream);
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(c->stream));
	collect_timings(c);
	return TYR_OK;
}

int tyr_reset_accum(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	if (!c->blit)
		return TYR_ERR_NO_BUFFER;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	HIPCHK(hipMemsetAsync(c->blit, 0, sizeof(float4) * static_cast<size_t>(c->cfg.width) * c->cfg.height, c->stream));
	c->hK->primary_ray_cnt = 0;
	std::memset(&c->hK->seg[c->cur][0][0], 0, sizeof c->hK->seg[0]);
	std::memset(c->hK->segSurv, 0, sizeof c->hK->segSurv);
	return push_counters(c);
}

int tyr_read_accum(tyr_ctx* c, float* host_float4) {
	if (!c || !host_float4)
		return TYR_ERR_INVALID;
	if (!c->blit)
		return TYR_ERR_NO_BUFFER;
	int rc = use_device(c);
	if (rc)
		return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(hipMemcpy(host_float4, c->blit, sizeof(float4) * static_cast<size_t>(c->cfg.width) * c->cfg.height, hipMemcpyDeviceToHost));
	return TYR_OK;
}

// ---- AoS import / export (fixtures, parity tests) -------------------------------------------
// The device's queues are physically unordered (hip/kernels.hpp "Queues"); the ABI's queues are the reference's: record i
// is the ray in slot i of the serial order.  Export gathers the records the segments hold and sorts them by virtual slot
// (survivors of the previous iteration first, in the order of the slots they had there -- their rank -- then this
// iteration's primary rays by ticket); import lays the records down in order, slot = position.
int tyr_queue_export(tyr_ctx* c, int which, tyr_ray_queue* host, uint32_t count) {
	if (!c || !host || (which != 0 && which != 1) || count > c->cfg.queue_size)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	const int qi = which == 0 ? c->cur : (c->cur ^ 1);
	const RayQ& q = c->q[qi];
	std::vector<uint32_t> slots;
	for (uint32_t cls = 0; cls < tyr::kClasses; ++cls) { // both classes: where a record lies says nothing about its place in the order
		std::vector<uint32_t> part;
		if ((rc = valid_slots(&c->dK->seg[qi][cls][0], part)))
			return rc;
		for (uint32_t sl : part)
			slots.push_back(cls * c->segCap * tyr::kSegs + sl);
	}
	uint32_t extent = 0;
	for (uint32_t sl : slots)
		extent = std::max(extent, sl + 1);
	std::vector<float4> a, d;
	std::vector<float2> b, h;
	std::vector<uint32_t> f, key;
	if ((rc = gather(q.o_dx, slots, extent, a)) || (rc = gather(q.dyz, slots, extent, b)) || (rc = gather(q.direct_ix, slots, extent, d)) || (rc = gather(q.flags, slots, extent, f)) ||
	    (rc = gather(q.hit, slots, extent, h)) || (rc = gather(q.key, slots, extent, key)))
		return rc;
	std::vector<uint32_t> order(slots.size());
	for (uint32_t i = 0; i < order.size(); ++i)
		order[i] = i;
	auto rankOf = [&](uint32_t i) { return (static_cast<uint64_t>((key[i] & tyr::kKeyIndirect) ? 0u : 1u) << 32) | (key[i] & tyr::kKeyMask); };
	std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return rankOf(x) < rankOf(y); });
	std::memset(host, 0, sizeof(tyr_ray_queue) * count);
	for (uint32_t k = 0; k < count && k < order.size(); ++k) {
		const uint32_t i = order[k];
		tyr_ray_queue& r = host[k];
		r.origin[0] = a[i].x;
		r.origin[1] = a[i].y;
		r.origin[2] = a[i].z;
		r.direction[0] = a[i].w;
		r.direction[1] = b[i].x;
		r.direction[2] = b[i].y;
		r.direct[0] = d[i].x;
		r.direct[1] = d[i].y;
		r.direct[2] = d[i].z;
		std::memcpy(&r.index, &d[i].w, 4);
		r.bounces = static_cast<int32_t>(f[i] & 0xffu);
		r.lastSpecular = static_cast<uint8_t>((f[i] >> 8) & 1u);
		r.distance = h[i].x;
		uint32_t id;
		std::memcpy(&id, &h[i].y, 4);
		r.geometry_type = (id & kHitSphere) ? 0 : 1;
		r.identifier = static_cast<int32_t>(id & ~kHitSphere);
	}
	return TYR_OK;
}

// Test hook: the device's OWN rank tables against the order tyr_queue_export presents.  The export sorts the records by
// their key on the host; the kernels never sort -- k_shade turns a key into the ray's slot with v_lookup() over the scan
// tables of the iteration before (hip/device_common.hpp).  Here the same three-part sum is taken from copies of those
// tables for every record of the queue, and compared with the record's place in the sorted order: a wrong table shows
// here, not one iteration later as wrong random numbers.
int tyr_queue_rank_check(tyr_ctx* c, int which, uint32_t* checked_out, uint32_t* mismatches_out) {
	if (!c || (which != 0 && which != 1) || !checked_out || !mismatches_out)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	const int qi = which == 0 ? c->cur : (c->cur ^ 1);
	// the tables the keys of this queue point into: written by the scan of the iteration that made its survivors
	const int t = which == 1 ? static_cast<int>(c->iter & 1u) : static_cast<int>((c->iter & 1u) ^ 1u);
	const size_t N = c->cfg.queue_size, entries = (N + 63) / 64 + kBlock, blocks = (N + 16383) / 16384 + 1;
	std::vector<unsigned long long> word(entries);
	std::vector<uint32_t> pre(entries), blk(blocks);
	HIPCHK(hipMemcpy(word.data(), c->vWord[t], entries * 8, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(pre.data(), c->vPre[t], entries * 4, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(blk.data(), c->vBlk[t], blocks * 4, hipMemcpyDeviceToHost));
	std::vector<uint32_t> keys;
	for (uint32_t cls = 0; cls < tyr::kClasses; ++cls) {
		std::vector<uint32_t> part, k;
		if ((rc = valid_slots(&c->dK->seg[qi][cls][0], part)))
			return rc;
		for (uint32_t& sl : part)
			sl += cls * c->segCap * tyr::kSegs;
		uint32_t extent = 0;
		for (uint32_t sl : part)
			extent = std::max(extent, sl + 1);
		if ((rc = gather(c->q[qi].key, part, extent, k)))
			return rc;
		keys.insert(keys.end(), k.begin(), k.end());
	}
	auto sortKey = [&](uint32_t key) { return (static_cast<uint64_t>((key & tyr::kKeyIndirect) ? 0u : 1u) << 32) | (key & tyr::kKeyMask); };
	std::sort(keys.begin(), keys.end(), [&](uint32_t x, uint32_t y) { return sortKey(x) < sortKey(y); });
	uint32_t bad = 0;
	for (uint32_t i = 0; i < keys.size(); ++i) {
		const uint32_t v = keys[i] & tyr::kKeyMask;
		uint32_t slot = v; // a fresh primary ray carries its slot itself
		if (keys[i] & tyr::kKeyIndirect) {
			const uint32_t e = v >> 6;
			if (e >= entries || (e >> 8) >= blocks) {
				++bad;
				continue;
			}
			slot = blk[e >> 8] + pre[e] + static_cast<uint32_t>(__builtin_popcountll(word[e] & ((1ull << (v & 63u)) - 1ull)));
			if (!((word[e] >> (v & 63u)) & 1ull))
				++bad; // the record's own survive bit must be set
		}
		if (slot != i)
			++bad;
	}
	*checked_out = static_cast<uint32_t>(keys.size());
	*mismatches_out = bad;
	return TYR_OK;
}

int tyr_queue_import(tyr_ctx* c, const tyr_ray_queue* host, uint32_t n) {
	if (!c || (!host && n) || n > c->cfg.queue_size)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	c->lastShadeFolded = false; // imported rays carry no sphere record: the pre-pass kernels do them
	const RayQ& q = c->q[c->cur];
	std::vector<float4> a(n), d(n);
	std::vector<float2> b(n), h(n);
	std::vector<uint32_t> f(n), key(n);
	for (uint32_t i = 0; i < n; ++i) {
		const tyr_ray_queue& r = host[i];
		a[i] = make_float4(r.origin[0], r.origin[1], r.origin[2], r.direction[0]);
		b[i] = make_float2(r.direction[1], r.direction[2]);
		float ix;
		std::memcpy(&ix, &r.index, 4);
		d[i] = make_float4(r.direct[0], r.direct[1], r.direct[2], ix);
		f[i] = (static_cast<uint32_t>(r.bounces) & 0xffu) | ((r.lastSpecular ? 1u : 0u) << 8);
		const uint32_t id = (r.geometry_type == 0 ? kHitSphere : 0u) | static_cast<uint32_t>(r.identifier);
		float idf;
		std::memcpy(&idf, &id, 4);
		h[i] = make_float2(r.distance, idf);
		key[i] = i; // slot = position; no kKeySphereDone: extend's pre-pass computes the sphere half as for any survivor
	}
	if (n) {
		HIPCHK(hipMemcpy(q.o_dx, a.data(), n * sizeof(float4), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(q.dyz, b.data(), n * sizeof(float2), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(q.direct_ix, d.data(), n * sizeof(float4), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(q.flags, f.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(q.hit, h.data(), n * sizeof(float2), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(q.key, key.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
	}
	c->hK->primary_ray_cnt = n;
	// all of them in class 0 (the traversal's own root test sorts out those that miss the tree)
	dense_counts(n, &c->hK->seg[c->cur][0][0]);
	std::memset(&c->hK->seg[c->cur][1][0], 0, sizeof c->hK->seg[0][0]);
	for (uint32_t w = 0; w < tyr::kSegs; ++w) {
		c->hK->segSurv[0][w] = c->hK->seg[c->cur][0][w * tyr::kSegStride];
		c->hK->segSurv[1][w] = 0;
	}
	return push_counters(c);
}

int tyr_shadow_export(tyr_ctx* c, tyr_shadow_queue* host, uint32_t count) {
	if (!c || !host || count > c->cfg.queue_size)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	std::vector<uint32_t> slots;
	if ((rc = valid_slots(&(c->dKc + c->shadowSet)->seg[0], slots))) // the set of the iteration that was shaded last
		return rc;
	uint32_t extent = 0;
	for (uint32_t s : slots)
		extent = std::max(extent, s + 1);
	std::vector<float4> a, b, col;
	std::vector<uint32_t> key;
	const ShadowQ& sq = c->shadow[c->shadowSet];
	if ((rc = gather(sq.o_dx, slots, extent, a)) || (rc = gather(sq.dyz_cd_ix, slots, extent, b)) || (rc = gather(sq.color, slots, extent, col)) || (rc = gather(sq.key, slots, extent, key)))
		return rc;
	std::vector<uint32_t> order(slots.size());
	for (uint32_t i = 0; i < order.size(); ++i)
		order[i] = i;
	std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return key[x] < key[y]; }); // the emitting rays' slots: the serial order
	std::memset(host, 0, sizeof(tyr_shadow_queue) * count);
	for (uint32_t k = 0; k < count && k < order.size(); ++k) {
		const uint32_t i = order[k];
		tyr_shadow_queue& s = host[k];
		s.origin[0] = a[i].x;
		s.origin[1] = a[i].y;
		s.origin[2] = a[i].z;
		s.direction[0] = a[i].w;
		s.direction[1] = b[i].x;
		s.direction[2] = b[i].y;
		s.closestDistance = b[i].z;
		std::memcpy(&s.buffer_index, &b[i].w, 4);
		s.color[0] = col[i].x;
		s.color[1] = col[i].y;
		s.color[2] = col[i].z;
	}
	return TYR_OK;
}

int tyr_shadow_import(tyr_ctx* c, const tyr_shadow_queue* host, uint32_t n) {
	if (!c || (!host && n) || n > c->cfg.queue_size)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	std::vector<float4> a(n), b(n), col(n);
	std::vector<uint32_t> key(n);
	for (uint32_t i = 0; i < n; ++i) {
		const tyr_shadow_queue& s = host[i];
		float ix;
		std::memcpy(&ix, &s.buffer_index, 4);
		a[i] = make_float4(s.origin[0], s.origin[1], s.origin[2], s.direction[0]);
		b[i] = make_float4(s.direction[1], s.direction[2], s.closestDistance, ix);
		col[i] = make_float4(s.color[0], s.color[1], s.color[2], 0.0f);
		key[i] = i;
	}
	if (n) {
		const ShadowQ& sq = c->shadow[c->iter & 1u];
		HIPCHK(hipMemcpy(sq.o_dx, a.data(), n * sizeof(float4), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(sq.dyz_cd_ix, b.data(), n * sizeof(float4), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(sq.color, col.data(), n * sizeof(float4), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(sq.key, key.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
	}
	// what shade leaves behind (kernel.cu:416-417): the counts connect reads, in this iteration's set
	c->hK->shadow_ray_cnt = n;
	c->shadowSet = c->iter & 1u;
	c->lastShadeFolded = false; // (imported shadow rays carry no sphere verdict)
	ConnectCounters* kc = c->dKc + (c->iter & 1u);
	uint32_t cnt[tyr::kSegs * tyr::kSegStride];
	dense_counts(n, cnt);
	HIPCHK(hipMemcpy(&kc->shadow_cnt, &n, sizeof(uint32_t), hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(&kc->seg[0], cnt, sizeof cnt, hipMemcpyHostToDevice));
	return push_counters(c);
}

int tyr_vecmath_probe(int32_t device, int32_t op, const float* a, const float* b, const float* c, uint32_t n, float* out) {
	if (!a || !b || !c || !out || n == 0 || op < 0 || op > 19)
		return TYR_ERR_INVALID;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
		return TYR_ERR_NO_DEVICE;
	HIPCHK(hipSetDevice(device));
	float* d[4] = { nullptr, nullptr, nullptr, nullptr };
	const size_t bytes = static_cast<size_t>(n) * 3 * sizeof(float);
	int rc = TYR_OK;
	for (auto& p : d)
		if (!rc && hipMalloc(reinterpret_cast<void**>(&p), bytes) != hipSuccess)
			rc = TYR_ERR_OOM;
	const float* src[3] = { a, b, c };
	for (int i = 0; i < 3 && !rc; ++i)
		if (hipMemcpy(d[i], src[i], bytes, hipMemcpyHostToDevice) != hipSuccess)
			rc = TYR_ERR_DEVICE;
	if (!rc) {
		(void)hipGetLastError();
		launch_vecmath_probe(op, d[0], d[1], d[2], n, d[3], nullptr);
		if (hipGetLastError() != hipSuccess || hipMemcpy(out, d[3], bytes, hipMemcpyDeviceToHost) != hipSuccess)
			rc = TYR_ERR_DEVICE;
	}
	for (auto& p : d)
		if (p)
			(void)hipFree(p);
	return rc;
}

int tyr_sun_setup(float sun_x, float sun_y, float* out25) {
	if (!out25)
		return TYR_ERR_INVALID;
	SunParams S;
	sun_setup(sun_x, sun_y, S);
	static_assert(sizeof(SunParams) == 25 * sizeof(float), "tyr_sun_setup hands out SunParams as 25 floats");
	std::memcpy(out25, &S, sizeof S);
	return TYR_OK;
}

int tyr_sunsky_probe(int32_t device, float sun_x, float sun_y, int32_t which, const float* dirs, uint32_t n, float* out) {
	if (!dirs || !out || n == 0 || which < 0 || which > 3)
		return TYR_ERR_INVALID;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
		return TYR_ERR_NO_DEVICE;
	HIPCHK(hipSetDevice(device));
	SunParams S;
	sun_setup(sun_x, sun_y, S);
	const size_t inBytes = (which == 3 ? 1 : static_cast<size_t>(n) * 3) * sizeof(float);
	const size_t outBytes = (static_cast<size_t>(n) * 3 + (which == 3 ? 1 : 0)) * sizeof(float);
	float *dIn = nullptr, *dOut = nullptr;
	int rc = TYR_OK;
	if (hipMalloc(reinterpret_cast<void**>(&dIn), inBytes) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&dOut), outBytes) != hipSuccess)
		rc = TYR_ERR_OOM;
	if (!rc && hipMemcpy(dIn, dirs, inBytes, hipMemcpyHostToDevice) != hipSuccess)
		rc = TYR_ERR_DEVICE;
	if (!rc) {
		(void)hipGetLastError();
		launch_sunsky_probe(S, which, dIn, n, dOut, nullptr);
		if (hipGetLastError() != hipSuccess || hipMemcpy(out, dOut, outBytes, hipMemcpyDeviceToHost) != hipSuccess)
			rc = TYR_ERR_DEVICE;
	}
	if (dIn)
		(void)hipFree(dIn);
	if (dOut)
		(void)hipFree(dOut);
	return rc;
}

int tyr_get_scene_info(tyr_ctx* c, tyr_scene_info* out) {
	if (!c || !out)
		return TYR_ERR_INVALID;
	if (!c->haveScene)
		return TYR_ERR_NO_SCENE;
	std::memset(out, 0, sizeof *out);
	out->n_prims = c->scene.nPrims;
	out->n_pair_nodes = c->scene.nPairs;
	out->n_quad_nodes = c->scene.nQuads;
	out->n_staged_nodes = c->scene.nStaged;
	out->quad_max_stack = c->scene.quadMaxStack;
	out->n_lights = c->nLights;
	out->max_quad_nodes = 1u << kQuadOrderShift;
	out->max_prim_offset = kMaxPrimOffset;
	const bool havePairs = (c->cfg.flags & (TYR_FLAG_COUNT_VISITS | TYR_FLAG_DEBUG_BVH)) != 0;
	out->device_bytes = static_cast<uint64_t>(c->scene.nQuads) * 128 + (havePairs ? static_cast<uint64_t>(c->scene.nPairs) * 64 : 0) + static_cast<uint64_t>(c->scene.nPrims) * 48;
	out->upload_layout_s = c->uploadLayoutS;
	out->upload_copy_s = c->uploadCopyS;
	out->layout_on_device = c->layoutOnDevice ? 1u : 0u;
	return TYR_OK;
}

int tyr_layout_probe(const tyr_bvh_node* nodes, int32_t nNodes, const tyr_triangle* prims, int32_t nPrims, int32_t want_pairs, tyr_layout_stats* out) {
	if (!out)
		return TYR_ERR_INVALID;
	std::memset(out, 0, sizeof *out);
	DeviceLayout L;
	const auto t0 = std::chrono::steady_clock::now();
	int rc;
	try {
		rc = build_device_layout(nodes, nNodes, prims, nPrims, L, want_pairs != 0);
	} catch (const std::bad_alloc&) {
		rc = TYR_ERR_OOM;
	} catch (...) {
		rc = TYR_ERR_UNSUPPORTED;
	}
	out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	if (rc)
		return rc;
	auto fnv = [](const FloatBuf& v) {
		uint64_t h = 1469598103934665603ull;
		const unsigned char* p = reinterpret_cast<const unsigned char*>(v.data());
		for (size_t i = 0, n = v.size() * sizeof(float); i < n; ++i)
			h = (h ^ p[i]) * 1099511628211ull;
		return h;
	};
	out->n_pair_nodes = L.nPairs;
	out->n_quad_nodes = L.nQuads;
	out->n_staged_nodes = L.nStaged;
	out->quad_max_stack = L.quadMaxStack;
	out->root_ref = L.rootRef;
	out->quad_root_ref = L.quadRootRef;
	out->hash_pairs = fnv(L.pairNodes);
	out->hash_quads = fnv(L.quadNodes);
	out->hash_tris = fnv(L.tris);
	return TYR_OK;
}

int tyr_scene_hash(tyr_ctx* c, tyr_layout_stats* out) {
	if (!c || !out)
		return TYR_ERR_INVALID;
	if (!c->haveScene)
		return TYR_ERR_NO_SCENE;
	int rc = use_device(c);
	if (rc)
		return rc;
	std::memset(out, 0, sizeof *out);
	out->n_pair_nodes = c->scene.nPairs;
	out->n_quad_nodes = c->scene.nQuads;
	out->n_staged_nodes = c->scene.nStaged;
	out->quad_max_stack = c->scene.quadMaxStack;
	out->root_ref = c->scene.rootRef;
	out->quad_root_ref = c->scene.quadRootRef;
	out->seconds = c->uploadLayoutS;
	HIPCHK(hipStreamSynchronize(c->stream));
	try {
		auto fnv_device = [&](const void* d, size_t bytes, uint64_t& h) -> int {
			h = 1469598103934665603ull;
			std::vector<unsigned char> buf(std::min<size_t>(bytes, size_t(64) << 20));
			for (size_t at = 0; at < bytes; at += buf.size()) {
				const size_t k = std::min(buf.size(), bytes - at);
				HIPCHK(hipMemcpy(buf.data(), static_cast<const char*>(d) + at, k, hipMemcpyDeviceToHost));
				for (size_t i = 0; i < k; ++i)
					h = (h ^ buf[i]) * 1099511628211ull;
			}
			return TYR_OK;
		};
		if ((rc = fnv_device(c->dNodes, static_cast<size_t>(c->scene.nPairs) * 64, out->hash_pairs)) ||
		    (rc = fnv_device(c->dQuads, static_cast<size_t>(c->scene.nQuads) * 128, out->hash_quads)) ||
		    (rc = fnv_device(c->dTris, static_cast<size_t>(c->scene.nPrims) * 48, out->hash_tris)))
			return rc;
	} catch (const std::bad_alloc&) {
		return TYR_ERR_OOM;
	}
	return TYR_OK;
}

int tyr_set_tuning(tyr_ctx* c, int key, int value) {
	if (!c)
		return TYR_ERR_INVALID;
	struct Knob {
		int key, lo, hi;
		int Tuning::*field;
	};
	static const Knob knobs[] = {
		{ TYR_TUNE_REFILL_MIN_IDLE, 1, 64, &Tuning::refillMinIdle },
		{ TYR_TUNE_WAVES_PER_SIMD, 0, 8, &Tuning::wavesPerSimd },
		{ TYR_TUNE_MIN_TRAVERSING, 1, 64, &Tuning::minTraversing },
		{ TYR_TUNE_TICKET_CHUNK, 64, 65536, &Tuning::ticketChunk },
		{ TYR_TUNE_STATIC_SHARE, 0, 15, &Tuning::staticShare },
		{ TYR_TUNE_STAGED_NODES, 0, static_cast<int>(kStagedNodes), &Tuning::stagedNodes },
		{ TYR_TUNE_PROFILE_MASK, 0, (1 << TYR_K_COUNT) - 1, &Tuning::profileMask },
		{ TYR_TUNE_MERGE_TRACE, 0, 1, &Tuning::mergeTrace },
		{ TYR_TUNE_STATIC_INTERLEAVE, 0, 1, &Tuning::staticInterleave },
		{ TYR_TUNE_RUN_AHEAD, 0, 2, &Tuning::runAhead },
		{ TYR_TUNE_WIDE_DRAIN, 0, 1, &Tuning::wideDrain },
		{ TYR_TUNE_STREAM_TAIL, 0, 1, &Tuning::streamTail },
		{ TYR_TUNE_STREAM_SHADE_PER_CU, 1, 2, &Tuning::streamShadePerCU },
		{ TYR_TUNE_STREAM_TRACE_PER_CU, 1, 4, &Tuning::streamTracePerCU }, // (five traversal blocks of 31.7 KB leave no LDS for the 27.8 KB shade block beside them: no shade, no progress)
		{ TYR_TUNE_FOLD_SPHERES, 0, 1, &Tuning::foldSpheres },
		{ TYR_TUNE_RETIRE_SKY, 0, 1, &Tuning::retireSky },
		{ TYR_TUNE_RESOLVE_SHADOWS, 0, 1, &Tuning::resolveShadows },
		{ TYR_TUNE_WIDE_BLOCK_MIN_ITEMS, -1, 0x7fffffff, &Tuning::wideBlockMinItems },
		{ TYR_TUNE_FOLD_PROLOGUE, 0, 1, &Tuning::foldPrologue },
		{ TYR_TUNE_LAYOUT_ON_DEVICE, 0, 1, &Tuning::layoutOnDevice },
		{ TYR_TUNE_SCAN_IN_TRACE, 0, 1, &Tuning::scanInTrace },
		{ TYR_TUNE_KERNEL_SNAPSHOT, 0, 1, &Tuning::kernelSnapshot },
	};
	for (const Knob& k : knobs) {
		if (k.key != key)
			continue;
		if (value < k.lo || value > k.hi)
			return TYR_ERR_INVALID;
		c->tuning.*(k.field) = value;
		return TYR_OK;
	}
	return TYR_ERR_INVALID;
}

int tyr_get_timings(tyr_ctx* c, tyr_timings* out, int reset) {
	if (!c || !out)
		return TYR_ERR_INVALID;
	*out = c->timings;
	if (reset)
		c->timings = tyr_timings{};
	return TYR_OK;
}

// ---- host side of the hot path --------------------------------------------------------------
int tyr_bvh_build(tyr_triangle* prims, int32_t n, const tyr_bbox* bboxes, tyr_bvh_node* nodes_out, int32_t algo) { return bvh_build(prims, n, bboxes, nodes_out, algo); }

int tyr_bvh_build_device(int32_t device, tyr_triangle* prims, int32_t n, const tyr_bbox* bboxes, tyr_bvh_node* nodes_out, double* seconds_out2) {
	try { // (a box that is not finite is refused by the first kernel that reads the boxes: TYR_ERR_INVALID, nothing reordered)
		return bvh_build_device(device, prims, n, bboxes, nodes_out, seconds_out2);
	} catch (const std::bad_alloc&) {
		return TYR_ERR_OOM;
	} catch (...) {
		return TYR_ERR_UNSUPPORTED;
	}
}

int tyr_set_build_threads(int32_t threads) {
	if (threads < 0 || threads > 256)
		return TYR_ERR_INVALID;
	set_build_threads(threads);
	return TYR_OK;
}

int tyr_triangle_bboxes(const tyr_triangle* prims, int32_t n, tyr_bbox* out) {
	if (n < 0 || (n > 0 && (!prims || !out)))
		return TYR_ERR_INVALID;
	triangle_bboxes(prims, n, out);
	return TYR_OK;
}

int tyr_camera_update(double horizontal_angle, double vertical_angle, float direction_out[3]) {
	if (!direction_out)
		return TYR_ERR_INVALID;
	// camera.cpp:46-52
	f3 d = mk3(static_cast<float>(std::cos(vertical_angle) * std::sin(horizontal_angle)), static_cast<float>(std::cos(vertical_angle) * std::cos(horizontal_angle)),
		static_cast<float>(std::sin(vertical_angle)));
	d = normalize(d);
	direction_out[0] = d.x;
	direction_out[1] = d.y;
	direction_out[2] = d.z;
	return TYR_OK;
}

int tyr_camera_handle_input(tyr_camera_pose* cam, const tyr_input_state* in, double delta) {
	if (!cam || !in || !std::isfinite(delta))
		return TYR_ERR_INVALID;
	// camera.cpp:3-44, statement by statement (glm's vec3 * float * float associates to the left)
	const float dt = static_cast<float>(delta);
	float speed = 1;
	if (in->key_left_shift)
		speed = 40;
	f3 position = ld3(cam->position);
	const f3 direction = ld3(cam->direction), up = ld3(cam->up);
	if (in->key_w)
		position = position + (direction * speed) * dt;
	else if (in->key_s)
		position = position - (direction * speed) * dt;
	const f3 displacement = (normalize(cross(direction, up)) * speed) * dt;
	if (in->key_a)
		position = position - displacement;
	else if (in->key_d)
		position = position + displacement;
	if (in->key_space)
		position.z += (1 * speed) * dt;
	else if (in->key_left_control)
		position.z -= (1 * speed) * dt;
	cam->position[0] = position.x;
	cam->position[1] = position.y;
	cam->position[2] = position.z;
	if (in->key_left_alt)
		return TYR_OK;
	const double diffx = in->cursor_x - in->window_w * 0.5;
	const double diffy = in->cursor_y - in->window_h * 0.5;
	cam->horizontal_angle += diffx * 0.012;
	cam->vertical_angle -= diffy * 0.012;
	// std::max(-pi / 2 + 0.001, std::min(vertical_angle, pi / 2 - 0.001)): pi is a float (variables.h:3), the sums are doubles
	const double lo = static_cast<double>(-kPi / 2) + 0.001, hi = static_cast<double>(kPi / 2) - 0.001;
	cam->vertical_angle = std::max(lo, std::min(cam->vertical_angle, hi));
	return TYR_OK;
}

int tyr_default_spheres(tyr_sphere* out7) {
	if (!out7)
		return TYR_ERR_INVALID;
	default_spheres(out7);
	return TYR_OK;
}

} // extern "C"
