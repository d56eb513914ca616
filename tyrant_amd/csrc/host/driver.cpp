// driver.cpp -- tyr_ctx and the C ABI of include/tyr_c.h: the host driver of the wavefront
// loop (reference: launch_kernels, kernel.cu:664-748, and its caller main.cpp:112-170).
//
// State that the reference keeps in function statics and device globals (kernel.cu:211-224,
// 665-667, 688-691) lives in tyr_ctx; constants arrive in the kernels as one by-value
// argument block (FrameParams) instead of cudaMemcpyToSymbol (kernel.cu:681-684, 707-709).
// There is no CPU fallback: without a HIP device tyr_create fails with TYR_ERR_NO_DEVICE.
// (The render loop, the staged test hooks and the tuning / probe entry points are host/render_loop.cpp, host/staged_api.cpp and
// host/tuning_probes.cpp; what they share is host/driver_internal.hpp.)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "driver_internal.hpp"

using namespace tyr;
using namespace tyr::drv;

namespace tyr {
namespace drv {

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

// physical slots that hold a record, per segment counter array `seg` (device pointer)
int valid_slots(const uint32_t* dSeg, std::vector<uint32_t>& slots, uint32_t* total) {
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

} // namespace drv
} // namespace tyr

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
	if (hipMemset(c->dKc, 0, 2 * sizeof(ConnectCounters)) != hipSuccess)
		return fail(TYR_ERR_NO_DEVICE);
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
