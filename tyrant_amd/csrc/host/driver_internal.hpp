// driver_internal.hpp -- what the translation units behind the C ABI share (private to the library).  host/driver.cpp owns the ctx
// (create / destroy, scene, setters, the accumulation buffer) and defines these helpers; host/render_loop.cpp is launch_kernels' loop
// (tyr_launch_kernels, tyr_render one iteration ahead of the counts); host/staged_api.cpp the test hooks (tyr_stage_*, AoS queue
// import / export); host/tuning_probes.cpp tyr_set_tuning, the timings and the device / layout probes.
#pragma once

#include <cstdint>
#include <vector>

#include <hip/hip_runtime.h>

#include "ctx.hpp"
#include "host.hpp"

#define HIPCHK(expr)                       \
	do {                                   \
		hipError_t e_ = (expr);            \
		if (e_ != hipSuccess)              \
			return static_cast<int>(e_);   \
	} while (0)

namespace tyr {
namespace drv {

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

int alloc_rayq(RayQ& q, size_t n);
void free_rayq(RayQ& q);
void default_spheres(tyr_sphere* s);
bool finite_n(const float* p, int n);
int use_device(tyr_ctx* c);
int sync_counters(tyr_ctx* c);
int push_counters(tyr_ctx* c);
FrameParams make_params(const tyr_ctx* c);
void collect_timings_of(tyr_ctx* c, int set);
void collect_timings(tyr_ctx* c);
uint32_t planned_new(const tyr_ctx* c);
int stage_begin(tyr_ctx* c);
void enqueue_primary(tyr_ctx* c, const FrameParams& P, uint32_t nNew);
void enqueue_extend(tyr_ctx* c, const FrameParams& P0, uint32_t nLive, uint32_t nSurvivors);
void enqueue_trace(tyr_ctx* c, const FrameParams& P0, uint32_t nLive, uint32_t nSurvivors, uint32_t maxShadowPrev);
void enqueue_shade(tyr_ctx* c, const FrameParams& P, uint32_t nLive);
void enqueue_connect(tyr_ctx* c, const FrameParams& P0, uint32_t maxShadow);
bool merged_render(const tyr_ctx* c);
int flush_pending_shadow(tyr_ctx* c);
void stage_end(tyr_ctx* c);
int check_device_error(const tyr_ctx* c);
// AoS import / export (host/staged_api.cpp): physical slots that hold a record, per segment counter array `seg` (device pointer)
int valid_slots(const uint32_t* dSeg, std::vector<uint32_t>& slots, uint32_t* total = nullptr);
void dense_counts(uint32_t n, uint32_t* cnt /* [kSegs * kSegStride] */);
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

} // namespace drv
} // namespace tyr
