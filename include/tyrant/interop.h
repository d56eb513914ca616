// tyrant/interop.h -- launch_kernels with the reference's argument list (interop.h:25).
//
//   cudaError launch_kernels(cudaArray_const_t array, glm::vec4* blit_buffer, Scene::GPUScene gpuScene,
//                            RayQueue* queue, RayQueue* queue2, ShadowQueue* shadowQueue);
//
// `array` (the GL surface) becomes a linear RGBA32F device buffer, or nullptr to skip the blit.
// `blit_buffer` is the caller's device float4[W*H] as in main.cpp:129-130.  The three queue
// pointers are accepted and ignored: the library keeps its queues as structure-of-arrays inside
// the ctx and ping-pongs them itself, so the caller's std::swap (main.cpp:169) is harmless.
// Reads the globals `camera`, `sun_position`, `sun_position_changed` like kernel.cu:699-710.
// Returns 0 or a hipError_t / tyr_status (the reference returns cudaError).
#pragma once
#include "Scene.h"
#include "camera.h"
#include "variables.h"

namespace tyrant {

inline int launch_kernels(void* array_rgba32f, vec4* blit_buffer, Scene::GPUScene gpuScene, RayQueue* /*queue*/, RayQueue* /*queue2*/, ShadowQueue* /*shadowQueue*/) {
	tyr_ctx* ctx = gpuScene.CUDACachedBVH.ctx;
	if (!ctx)
		return TYR_ERR_NO_SCENE;
	int rc;
	// bind the caller's buffer when it is not the one the ctx holds (no function static: several ctxs / threads)
	void* const have = tyr_get_blit_buffer(ctx);
	if (blit_buffer ? have != static_cast<void*>(blit_buffer) : !have) {
		if ((rc = tyr_set_blit_buffer(ctx, blit_buffer)))
			return rc;
	}
	const tyr_camera cam = { { camera.position.x, camera.position.y, camera.position.z }, { camera.direction.x, camera.direction.y, camera.direction.z },
		{ camera.up.x, camera.up.y, camera.up.z }, camera.focalDistance, camera.lensRadius };
	if ((rc = tyr_set_camera(ctx, &cam)))
		return rc;
	if (sun_position_changed) { // kernel.cu:704-710
		sun_position_changed = false;
		if ((rc = tyr_set_sun_position(ctx, sun_position.x, sun_position.y)))
			return rc;
	}
	if ((rc = tyr_launch_kernels(ctx)))
		return rc;
	if (array_rgba32f)
		rc = tyr_resolve(ctx, array_rgba32f); // kernel.cu:729-731
	return rc;
}

#ifdef TYRANT_IMPLEMENTATION
// variables.cpp:3-5, camera.cpp:54 -- define TYRANT_IMPLEMENTATION in exactly one translation unit
vec2 sun_position = { 0.05f, 0.3f };
bool sun_position_changed = true;
Camera camera;
#endif

} // namespace tyrant
