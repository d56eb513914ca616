// tuning_probes.cpp -- tyr_set_tuning (the launch-shape knobs), the stage timings, and the probes the parity
// tests pin the device's arithmetic and the upload's layout pass with (tyr_vecmath_probe, tyr_sunsky_probe / tyr_sun_setup,
// tyr_get_scene_info, tyr_layout_probe, tyr_scene_hash).
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

extern "C" {

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

} // extern "C"
