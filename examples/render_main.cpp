// examples/render_main.cpp -- the reference's main() (main.cpp:87-205) without the window: pick a
// device, load a scene, allocate the buffers the caller owns, then per frame
//     camera.update(); launch_kernels(...); std::swap(ray_buffer_work, ray_buffer_next);
// written against include/tyrant/*.h, i.e. the reference's own names over libtyrant_hip.so.
//
//   render_main [device] [frames] [out.ppm] [scene.ply]
#define TYRANT_IMPLEMENTATION
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include <hip/hip_runtime.h>

#include "tyrant/interop.h"

using namespace tyrant;

#define TYR_CHECK(x)                                                                            \
	do {                                                                                        \
		int rc_ = (x);                                                                          \
		if (rc_) { /* assert_cuda.cpp:3-14: print and exit */                                   \
			std::fprintf(stderr, "tyr_assert: %s %s %d\n", tyr_status_string(rc_), __FILE__, __LINE__); \
			std::exit(rc_);                                                                     \
		}                                                                                       \
	} while (0)

// a small height field standing in for Data/castle.ply (absent from the reference checkout)
static std::vector<vec3> make_mesh(int cells) {
	std::vector<vec3> v;
	auto P = [&](int i, int j) {
		const float x = -80.0f + 160.0f * i / cells, y = -80.0f + 160.0f * j / cells;
		return vec3{ x, y, -18.0f + 6.0f * std::sin(x * 0.09f) * std::cos(y * 0.08f) };
	};
	for (int j = 0; j < cells; ++j)
		for (int i = 0; i < cells; ++i) {
			const vec3 p00 = P(i, j), p10 = P(i + 1, j), p11 = P(i + 1, j + 1), p01 = P(i, j + 1);
			v.insert(v.end(), { p00, p10, p11, p00, p11, p01 });
		}
	return v;
}

int main(int argc, char** argv) {
	const int device = argc > 1 ? std::atoi(argv[1]) : 0; // main.cpp:91
	const int frames = argc > 2 ? std::atoi(argv[2]) : 64;
	const char* out_path = argc > 3 ? argv[3] : nullptr;
	const unsigned W = 640, H = 360, N = 262144;

	tyr_config cfg{};
	cfg.width = W;
	cfg.height = H;
	cfg.queue_size = N;
	cfg.device = device;
	cfg.nranks = 1;
	tyr_ctx* ctx = nullptr;
	TYR_CHECK(tyr_create(&ctx, &cfg));

	set_default_ctx(ctx); // the reference's device state is process-wide: the handle-less calls below use this context
	if (std::getenv("TYR_BUILD_ON_DEVICE"))
		set_build_device(device); // Scene.cpp:53-67 on the GPU: `BVH bvh(...)` builds there (tyr_bvh_build_device), Scene::Load builds AND lays out there (tyr_scene_build_upload): the same bytes

	Scene scene;
	if (argc > 4)
		scene.Load(argv[4]); // main.cpp:112-113: Scene scene; scene.Load("Data/castle.ply");
	else
		scene.Load(make_mesh(96));

	// main.cpp:119-130: the caller owns the buffers.  The ray queues are opaque here (see tyrant/interop.h).
	RayQueue* ray_buffer_work = nullptr;
	RayQueue* ray_buffer_next = nullptr;
	ShadowQueue* shadow_queue_buffer = nullptr;
	vec4* blit_buffer = nullptr;
	vec4* surface = nullptr;
	TYR_CHECK(hipSetDevice(device));
	TYR_CHECK(hipMalloc(reinterpret_cast<void**>(&blit_buffer), sizeof(vec4) * W * H));
	TYR_CHECK(hipMalloc(reinterpret_cast<void**>(&surface), sizeof(vec4) * W * H));

	camera.position = { 0.0f, -250.0f, 95.0f };
	camera.horizontal_angle = 0.0;
	camera.vertical_angle = -0.273;

	const auto t0 = std::chrono::steady_clock::now();
	for (int f = 0; f < frames; ++f) { // main.cpp:139-170
		camera.update();
		TYR_CHECK(launch_kernels(f + 1 == frames ? surface : nullptr, blit_buffer, scene.gpuScene, ray_buffer_work, ray_buffer_next, shadow_queue_buffer));
		std::swap(ray_buffer_work, ray_buffer_next);
	}
	const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	tyr_counters k;
	TYR_CHECK(tyr_get_counters(ctx, &k));
	std::printf("%d frames in %.3f s (%.2f ms/frame), %.1f Mrays/s, frame counter %u\n", frames, dt, dt / frames * 1e3,
		(k.total_extend_rays + k.total_shadow_rays) / dt / 1e6, k.frame);

	if (out_path) {
		std::vector<vec4> img(static_cast<size_t>(W) * H);
		TYR_CHECK(hipMemcpy(img.data(), surface, sizeof(vec4) * W * H, hipMemcpyDeviceToHost));
		if (FILE* fp = std::fopen(out_path, "wb")) {
			std::fprintf(fp, "P6 %u %u 255\n", W, H);
			for (const vec4& p : img) {
				const float c[3] = { p.x, p.y, p.z };
				for (float ch : c)
					std::fputc(static_cast<int>(std::clamp(ch != ch ? 0.0f : ch, 0.0f, 1.0f) * 255.0f), fp);
			}
			std::fclose(fp);
		}
	}
	TYR_CHECK(tyr_destroy(ctx));
	(void)hipFree(blit_buffer);
	(void)hipFree(surface);
	return 0;
}
