// ctx.hpp -- tyr_ctx, the state behind the C ABI's opaque handle (private to the library: host/driver.cpp owns it,
// host/dist.cpp reads the stream, the blit buffer and the sharding from it).
#pragma once

#include <hip/hip_runtime.h>

#include "host.hpp"

using tyr::ConnectCounters;
using tyr::DevCounters;
using tyr::DevScene;
using tyr::LaunchCache;
using tyr::RayQ;
using tyr::ShadowQ;
using tyr::SunParams;
using tyr::Tuning;

struct tyr_ctx {
	tyr_config cfg{};
	hipStream_t stream = nullptr;
	bool ownStream = false;
	uint32_t localRows = 0, localPixels = 0;

	RayQ q[2]{};
	int cur = 0; // q[cur] = work queue, q[cur ^ 1] = next (the caller's std::swap, main.cpp:169)
	ShadowQ shadow[2]{}; // iteration i writes and connects shadow[i & 1]
	DevCounters* dK = nullptr;
	DevCounters* hK = nullptr; // pinned host mirror
	DevCounters* hSnap[2] = { nullptr, nullptr }; // tyr_render one iteration ahead: where the counters of iteration i land (set i & 1)
	hipEvent_t evSnap[2] = { nullptr, nullptr };
	tyr::HostSnap* hostSnap[2] = { nullptr, nullptr }; // TYR_TUNE_KERNEL_SNAPSHOT: written by k_shade's last block (pinned, device-visible)
	tyr::HostSnap* hostSnapDev[2] = { nullptr, nullptr }; // ... as the device addresses them
	uint32_t snapSeq = 0;                               // stamp of the last iteration queued with a kernel-written snapshot
	uint32_t snapSeqOf[2] = { 0, 0 };                   // ... per set (0: that iteration uses the copy + event)
	ConnectCounters* dKc = nullptr; // two sets, iteration i uses set i & 1
	uint32_t iter = 0;
	uint32_t shadowSet = 0; // which of the two sets holds the counts of the shadow queue's current content (tyr_shadow_export)

	hipEvent_t evSnapshot = nullptr;
	// TYR_TUNE_SCAN_IN_TRACE
	bool scanCarried = false;      // the last shade launch left its slot scan to the next traversal launch (hip/scan_wave.hpp)
	uint32_t scanCarriedSet = 0;
	// tyr_render with TYR_TUNE_MERGE_TRACE: the shadow rays of the last shaded iteration have not been traced yet (they
	// ride in the next iteration's trace launch, or in a connect of their own when the render ends)
	bool shadowPending = false;
	uint32_t shadowPendingMax = 0;
	// tyr_render one iteration ahead: the last queued iteration turned out to be empty; once the stream is idle the device's
	// counters get back what that iteration's set_wavefront_globals zeroed (the live and shadow counts of the last real one)
	bool runAheadUndo = false;
	uint32_t undoLive = 0, undoShadows = 0;
	// hip/kernels.hpp "Queues": room per queue segment, the survive bytes and the two sets of scan tables
	uint32_t segCap = 0;
	uint8_t* survFlag = nullptr;
	unsigned long long* vWord[2] = { nullptr, nullptr };
	uint32_t* vPre[2] = { nullptr, nullptr };
	uint32_t* vBlk[2] = { nullptr, nullptr };
	bool unboundedRender = false; // tyr_render was called without an iteration limit: contributions may arrive an iteration early (TYR_TUNE_RETIRE_SKY's survivors)
	bool lastShadeFolded = false; // the shade launch that filled the current work / shadow queues did the sphere pre-passes' work too (TYR_TUNE_FOLD_SPHERES)
	float4* blit = nullptr;
	bool ownBlit = false;

	float4* dNodes = nullptr;
	float4* dQuads = nullptr;
	float4* dTris = nullptr;
	uint32_t* dLights = nullptr; // TYR_FLAG_LIGHT_LIST: emissive triangles, array order
	uint32_t nLights = 0;
	float triEmission[3] = { 3.0f, 3.0f, 3.0f }; // kernel.cu:680
	float4* dPalette = nullptr;                   // TYR_FLAG_TRIANGLE_COLORS: 256 x { colour, emission }
	DevScene scene{};
	bool haveScene = false;
	double uploadLayoutS = 0.0, uploadCopyS = 0.0; // the last tyr_scene_upload: host layout passes, allocation + copies to HBM (tyr_scene_info)
	bool layoutOnDevice = false; // ... and where its layout pass ran

	tyr_sphere spheres[TYR_NUM_SPHERES]{};
	tyr_camera cam{};
	float sunPos[2] = { 0.05f, 0.3f }; // variables.cpp:3
	bool sunChanged = true;             // variables.cpp:4
	SunParams sun{};

	// launch_kernels statics, kernel.cu:665-667, 688-691
	bool firstTime = true;
	uint32_t frame = 1;
	float lastPos[3] = { 0, 0, 0 }, lastDir[3] = { 0, 0, 0 };
	float lastFocal = 1.0f, lastLens = 0.02f;
	float camRight[3]{}, camUp[3]{};

	Tuning tuning{};
	LaunchCache launchCache{}; // occupancy answers of the persistent kernels, per ctx (not process-wide)
	int numCUs = 256;

	hipEvent_t ev[2][2 * TYR_K_COUNT]{}; // TYR_FLAG_PROFILE: start / stop per stage, two sets (iteration i uses set i & 1: two iterations may be queued)
	bool evUsed[2][TYR_K_COUNT]{};
	tyr_timings timings{};
};

